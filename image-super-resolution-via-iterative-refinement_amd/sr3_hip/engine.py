"""Host-side handle of one UNet plan: parameter arena packing and the forward call.

The compute is entirely inside libsr3_mi355x.so; torch supplies device memory, the current
stream and tensor views -- nothing else.
"""
import ctypes as C
import math

import torch

from . import lib as L


def make_desc(variant, in_channel, out_channel, inner_channel, norm_groups, channel_mults, attn_res,
              res_blocks, image_size):
    d = L.UnetDesc()
    d.variant = {'sr3': 0, 'ddpm': 1}[variant]
    d.in_channel = int(in_channel)
    d.out_channel = int(out_channel if out_channel is not None else in_channel)
    d.inner_channel = int(inner_channel)
    d.norm_groups = int(norm_groups)
    mults = list(channel_mults)
    attn = [attn_res] if isinstance(attn_res, int) else list(attn_res)
    if len(mults) > 8 or len(attn) > 8:
        raise L.Sr3Error('at most 8 channel multipliers / attention resolutions')
    d.n_mults = len(mults)
    for i, m in enumerate(mults):
        d.channel_mults[i] = int(m)
    d.n_attn_res = len(attn)
    for i, a in enumerate(attn):
        d.attn_res[i] = int(a)
    d.res_blocks = int(res_blocks)
    d.image_size = int(image_size)
    return d


class Plan(object):
    """Owns an sr3_plan*; exposes the parameter table."""

    def __init__(self, variant, in_channel, out_channel, inner_channel, norm_groups, channel_mults, attn_res,
                 res_blocks, image_size):
        self.lib = L.load()
        self.variant = variant
        self.inner = int(inner_channel)
        self.in_channel = int(in_channel)
        self.out_channel = int(out_channel if out_channel is not None else in_channel)
        self.image_size = int(image_size)
        self.desc = make_desc(variant, in_channel, out_channel, inner_channel, norm_groups, channel_mults,
                              attn_res, res_blocks, image_size)
        h = C.c_void_p()
        L.check(self.lib.sr3_plan_create(C.byref(self.desc), C.byref(h)))
        self.handle = h
        self.options = {}          # plan options set through set_option (key -> value)
        self.generation = 0        # bumped by every set_option: part of the reverse-loop graph cache key
        self.param_floats = int(self.lib.sr3_plan_param_floats(h))
        self.table = []
        pi = L.ParamInfo()
        for i in range(self.lib.sr3_plan_num_params(h)):
            L.check(self.lib.sr3_plan_param_info(h, i, C.byref(pi)))
            self.table.append(dict(name=pi.name.decode(), shape=tuple(pi.shape[:pi.ndim]), pack=int(pi.pack),
                                   offset=int(pi.offset), numel=int(pi.numel)))

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.sr3_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def set_option(self, key, value):
        rc = self.lib.sr3_plan_set_option(self.handle, key.encode(), int(value))
        if rc < 0:
            L.check(rc)
        self.options[key] = int(value)
        self.generation += 1       # anything compiled / captured against the previous options is stale
        return rc

    def workspace_bytes(self, batch):
        n = int(self.lib.sr3_workspace_bytes(self.handle, int(batch)))
        if n == 0:
            raise L.Sr3Error('sr3_workspace_bytes failed: %s' % (self.lib.sr3_last_error() or b'').decode())
        return n

    def forward_flops(self, batch):
        return float(self.lib.sr3_plan_forward_flops(self.handle, int(batch)))

    def num_ops(self, batch):
        return int(self.lib.sr3_plan_num_ops(self.handle, int(batch)))

    def op_list(self, batch):
        """The ordered launch list of one forward at this batch size (host-only inspection)."""
        info = L.OpInfo()
        side, wait = C.c_int(), C.c_int()
        out = []
        for i in range(self.num_ops(batch)):
            L.check(self.lib.sr3_plan_op_info(self.handle, int(batch), i, C.byref(info)))
            o = {k: getattr(info, k) for k, _ in L.OpInfo._fields_}
            L.check(self.lib.sr3_plan_op_side(self.handle, int(batch), i, C.byref(side), C.byref(wait)))
            if side.value >= 0 or wait.value >= 0:      # plan option fork_side: ops on the side stream / the consumers that join them
                o['side_id'], o['wait_id'] = side.value, wait.value
            out.append(o)
        return out

    def taps(self):
        out = []
        name = C.create_string_buffer(64)
        off = C.c_size_t()
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        for i in range(self.lib.sr3_plan_num_taps(self.handle)):
            L.check(self.lib.sr3_plan_tap_info(self.handle, i, name, 64, C.byref(off), C.byref(c), C.byref(h), C.byref(w)))
            out.append((name.value.decode(), int(off.value), c.value, h.value, w.value))
        return out

    # ---- arena <-> reference state-dict views ---------------------------------------------
    def view(self, arena, entry):
        """A view of `arena` with the reference shape (OIHW for convs) -- writes go through."""
        flat = arena[entry['offset']:entry['offset'] + entry['numel']]
        shape = entry['shape']
        if entry['pack'] == 1:
            o, i, kh, kw = shape
            return flat.view(o, kh, kw, i).permute(0, 3, 1, 2)
        return flat.view(*shape)

    def default_freq(self):
        """Frequency table of PositionalEncoding (sr3 unet.py:24-28) / TimeEmbedding.inv_freq
        (ddpm unet.py:23-26), computed with the same fp32 torch expressions as the reference."""
        dim = self.inner
        if self.variant == 'sr3':
            count = dim // 2
            step = torch.arange(count, dtype=torch.float32) / count
            return torch.exp(-math.log(1e4) * step)
        return torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-math.log(10000) / dim))


class Workspace(object):
    """Per-(device, batch) scratch owned by torch's caching allocator (graph-capture friendly)."""

    def __init__(self):
        self.buf = None
        self.batch = None

    def get(self, plan, batch, device):
        need = plan.workspace_bytes(batch)
        if self.buf is None or self.buf.numel() < need or self.buf.device != device:
            self.buf = torch.empty(need + 256, dtype=torch.uint8, device=device)
        self.batch = batch
        off = (-self.buf.data_ptr()) % 256
        return self.buf[off:off + need], need


def reverse_step(plan, arena, freq, ws, x, z, tables, step2, cond=None, level_table=None, clip_denoised=True, eps_out=None):
    """One whole reverse step in place on `x` (sr3_reverse_step): eps = UNet(cat([cond, x], 1), level(t)); x <- p_sample update;
    t <- t - 1, with t = step2[1] (int32 tensor of two).  `tables` = (a, b, c1, c2, sigma) schedule tables on the device."""
    if not x.is_cuda or not x.is_contiguous() or x.dtype != torch.float32:
        raise L.Sr3Error('reverse_step needs a contiguous fp32 GPU tensor (got %s, %s); there is no CPU fallback' % (x.device, x.dtype))
    B = x.shape[0]
    cc = 0
    if cond is not None:
        cond = cond.contiguous()
        cc = cond.shape[1]
    if x.shape[1] + cc != plan.in_channel or x.shape[1] != plan.out_channel or x.shape[2] != plan.image_size or x.shape[3] != plan.image_size:
        raise L.Sr3Error('input shape %s (+%d cond channels) does not match the plan (in_channel %d, out_channel %d, size %d)'
                         % (tuple(x.shape), cc, plan.in_channel, plan.out_channel, plan.image_size))
    if step2.dtype != torch.int32 or step2.numel() != 2 or step2.device != x.device:
        raise L.Sr3Error('step2 must be two int32 on the device of x')
    wsbuf, need = ws.get(plan, B, x.device)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    a, b, c1, c2, sg = tables
    L.check(plan.lib.sr3_reverse_step(plan.handle, L.ptr(x), L.ptr(cond), cc, L.ptr(freq), L.ptr(level_table), L.ptr(step2),
                                      L.ptr(arena), L.ptr(wsbuf), need, L.ptr(z), L.ptr(a), L.ptr(b), L.ptr(c1), L.ptr(c2), L.ptr(sg),
                                      1 if clip_denoised else 0, L.ptr(eps_out), B, C.c_void_p(stream)))
    return x


def unet_forward(plan, arena, freq, ws, x, cond=None, noise_level=None, timestep=None, level_table=None,
                 step_dev=None, out=None):
    """eps = UNet(cat([cond, x], 1), level)   (all tensors on the GPU, fp32, NCHW contiguous)."""
    if not x.is_cuda:
        raise L.Sr3Error('the MI355X engine only runs on a GPU tensor (got %s); there is no CPU fallback' % x.device)
    B = x.shape[0]
    x = x.contiguous()
    cc = 0
    if cond is not None:
        cond = cond.contiguous()
        cc = cond.shape[1]
    if x.dtype != torch.float32 or (cond is not None and cond.dtype != torch.float32):
        raise L.Sr3Error('fp32 tensors expected')
    if x.shape[1] + cc != plan.in_channel or x.shape[2] != plan.image_size or x.shape[3] != plan.image_size:
        raise L.Sr3Error('input shape %s (+%d cond channels) does not match the plan (in_channel %d, size %d)'
                         % (tuple(x.shape), cc, plan.in_channel, plan.image_size))
    if noise_level is not None:
        noise_level = noise_level.reshape(-1).contiguous().float()
        if noise_level.numel() != B:
            raise L.Sr3Error('noise_level must have one value per sample')
    if timestep is not None:
        timestep = timestep.reshape(-1).contiguous().long()
        if timestep.numel() != B:
            raise L.Sr3Error('timestep must have one value per sample')
    wsbuf, need = ws.get(plan, B, x.device)
    if out is None:
        out = torch.empty(B, plan.out_channel, plan.image_size, plan.image_size, device=x.device, dtype=torch.float32)
    stream = torch.cuda.current_stream(x.device).cuda_stream
    L.check(plan.lib.sr3_unet_forward(plan.handle, L.ptr(x), L.ptr(cond), cc, L.ptr(noise_level), L.ptr(timestep),
                                      L.ptr(freq), L.ptr(level_table), L.ptr(step_dev), L.ptr(arena), L.ptr(wsbuf),
                                      need, L.ptr(out), B, C.c_void_p(stream)))
    return out
