"""Helpers for the -m gpu tests: thin torch<->C-ABI call wrappers (tests always go through the
C ABI of libsr3_mi355x.so) and float64 CPU references of single ops."""
import ctypes as C

import torch
import torch.nn.functional as F

from sr3_hip import lib as L


def dev():
    return torch.device('cuda:0')


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def nhwc(x):      # NCHW -> NHWC contiguous
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):      # NHWC -> NCHW contiguous
    return x.permute(0, 3, 1, 2).contiguous()


def ohwi(w):      # OIHW -> OHWI contiguous
    return w.permute(0, 2, 3, 1).contiguous()


def conv_call(src0, src1, w, bias=None, ss=None, act=0, film=None, res0=None, res1=None, ups=0, stride=1,
              tile_cfg=0, ksplit=0, want_stats=False):
    """All tensor args are CPU fp32 in reference layouts: src NCHW, w OIHW, ss [B,Cin,2], film [B,Cout],
    res NCHW.  Returns (out NCHW cpu, stats [B,Cout,2] cpu double or None)."""
    lib = L.load()
    d = dev()
    B, C0, Hs, Ws = src0.shape
    C1 = 0 if src1 is None else src1.shape[1]
    Cout, Cin, k, _ = w.shape
    assert Cin == C0 + C1
    pad = k // 2
    Ho = ((Hs << ups) + 2 * pad - k) // stride + 1
    Wo = ((Ws << ups) + 2 * pad - k) // stride + 1
    g = lambda t: None if t is None else t.to(d)
    s0, s1 = g(nhwc(src0)), (None if src1 is None else g(nhwc(src1)))
    wd, bd, ssd, fd = g(ohwi(w)), g(bias), g(None if ss is None else ss.contiguous()), g(None if film is None else film.contiguous())
    r0 = None if res0 is None else g(nhwc(res0))
    r1 = None if res1 is None else g(nhwc(res1))
    out = torch.full((B, Ho, Wo, Cout), float('nan'), device=d)
    stats = None
    if want_stats:
        Tst = int(lib.sr3_conv_stats_slices(B, Hs, Ws, ups, Cin, Cout, tile_cfg, ksplit))
        stats = torch.full((B, max(Tst, 1), Cout, 2), float('nan'), dtype=torch.float64, device=d)
    nb = int(lib.sr3_conv_scratch_bytes(B, Ho, Wo, Cin, Cout, k, tile_cfg, ksplit))
    scratch = torch.empty(max(nb, 16), dtype=torch.uint8, device=d)
    L.check(lib.sr3_conv_f32(L.ptr(s0), C0, L.ptr(s1), C1, B, Hs, Ws, ups, stride, k, Cout, L.ptr(wd), L.ptr(bd),
                             L.ptr(ssd), act, L.ptr(fd), 0 if film is None else film.shape[1], L.ptr(r0),
                             0 if res0 is None else res0.shape[1], L.ptr(r1), 0 if res1 is None else res1.shape[1],
                             L.ptr(out), L.ptr(stats), tile_cfg, ksplit, L.ptr(scratch), nb, stream()))
    torch.cuda.synchronize()
    return nchw(out).cpu(), (None if stats is None else stats.cpu().sum(1))


def wino_expected_refusal(B, Cin, H, W, ups, tile, ksplit, k=3, stride=1):
    """Which error message the Winograd kernels (tile 11 / 12: conv3x3_wino.hip, 13: conv3x3_wino2.hip) MUST raise for this problem,
    or None when they must run -- the geometric rules of wino_geometry / conv3x3_wino_forward restated, so that a test can assert
    the refusal instead of skipping on whatever message came back (a production tile that raises unexpectedly then FAILS)."""
    Ho, Wo = H << ups, W << ups
    if k != 3 or stride != 1:
        return 'does not fit'
    nb4 = tile != 13 and Ho == 8 and Wo == 8 and ups == 0 and B % 4 == 0 and Cin > 16
    if tile == 13:
        if Wo < 16 or Wo % 16 or Ho % 8:
            return 'does not fit'
    elif not nb4 and (Wo < 16 or Wo % 16 or Ho % 16):
        return 'does not fit'
    if ksplit == 0:
        return None
    chunks = -(-Cin // 16)
    if -(-chunks // ksplit) > (16 if nb4 else 64):          # (checked first by conv3x3_wino_forward)
        return 'per K split'
    if nb4 and ksplit < 2:
        return 'split-K only'
    if ksplit > 1 and (ksplit - 1) * -(-chunks // ksplit) >= chunks:
        return 'empty split'
    return None


def conv_ref(src0, src1, w, bias=None, ss=None, act=0, film=None, res0=None, res1=None, ups=0, stride=1):
    """float64 reference of the same fused op."""
    x = src0 if src1 is None else torch.cat([src0, src1], 1)
    x = x.double()
    if act:
        x = x * ss[:, :, 0].double()[:, :, None, None] + ss[:, :, 1].double()[:, :, None, None]
        if act == 2:
            x = x * torch.sigmoid(x)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    y = F.conv2d(x, w.double(), None if bias is None else bias.double(), stride=stride, padding=w.shape[2] // 2)
    if film is not None:
        y = y + film.double()[:, :, None, None]
    if res0 is not None:
        r = res0 if res1 is None else torch.cat([res0, res1], 1)
        y = y + r.double()
    return y


def assert_close(got, ref, tol=2e-5, what=''):
    ref = ref.double()
    err = (got.double() - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= tol * scale, '%s: max abs err %.3e > %.1e * %.3g' % (what, err, tol, scale)
    return err
