#!/usr/bin/env python3
"""Does hipMemsetAsync clear a region larger than 4 GiB completely on this runtime?"""
import ctypes as C
import torch
hip = C.CDLL('libamdhip64.so')
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetAsync.restype = C.c_int
for gb in (3.0, 4.5, 6.4, 8.6):
    n = int(gb * (1 << 30))
    t = torch.full((n,), 255, dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    rc = hip.hipMemsetAsync(C.c_void_p(t.data_ptr()), 0, n, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    nz = int((t.view(torch.int64)[: n // 8] != 0).sum())
    first = int((t.view(torch.int64)[: n // 8] != 0).nonzero()[0]) * 8 if nz else -1
    print('%.1f GiB: rc %d, nonzero 8-byte words after memset: %d (first at byte %d)' % (gb, rc, nz, first), flush=True)
    del t
    torch.cuda.empty_cache()
