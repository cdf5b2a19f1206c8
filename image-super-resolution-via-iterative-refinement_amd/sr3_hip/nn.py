"""nn.Module faces of the engine: the objects `define_G` hands to the reference's callers.

`EngineUNet` is what `GaussianDiffusion.denoise_fn` is in the reference
(model/sr3_modules/unet.py:161-259, model/ddpm_modules/unet.py:147-243): same constructor
arguments, same `forward(x, time)`, same state-dict keys and shapes -- but it owns one packed
parameter arena and every FLOP runs in libsr3_mi355x.so.
"""
import math
from collections import OrderedDict

import torch
from torch import nn

from . import engine as E
from . import lib as L


def _is_norm(name):
    return '.block.0.' in name or '.norm.' in name


class EngineUNet(nn.Module):
    variant = 'sr3'

    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                 channel_mults=(1, 2, 4, 8, 8), attn_res=(8), res_blocks=3, dropout=0,
                 with_noise_level_emb=True, image_size=128):
        super().__init__()
        if not with_noise_level_emb:
            raise NotImplementedError('the engine always conditions on the noise level / timestep '
                                      '(define_G never disables it: model/networks.py:91-101)')
        self.plan = E.Plan(self.variant, in_channel, out_channel, inner_channel, norm_groups, channel_mults,
                           attn_res, res_blocks, image_size)
        self.dropout = float(dropout)
        self.arena = nn.Parameter(torch.zeros(self.plan.param_floats, dtype=torch.float32), requires_grad=False)
        self.register_buffer('freq', self.plan.default_freq(), persistent=False)
        self._ws = E.Workspace()
        # derived weights (Winograd-transformed 3x3 filters): engine-owned device buffer, never part of a state dict;
        # rebuilt whenever the arena content, its storage or the plan options changed
        self._derived = None
        self._derived_key = None
        self._weights_epoch = 0
        self.reset_parameters()

    # ---- initialisation: same distributions AND same RNG consumption order as the reference ----
    def reset_parameters(self):
        """PyTorch default init of Conv2d / Linear / GroupNorm, drawn in the reference's module
        construction order so a given torch seed yields the reference's weights."""
        fan_in = 1
        for e in self.plan.table:
            v = self.plan.view(self.arena.detach(), e)
            name = e['name']
            if _is_norm(name):
                v.fill_(1.0 if name.endswith('weight') else 0.0)
            elif len(e['shape']) >= 2:
                t = torch.empty(e['shape'], dtype=torch.float32)
                nn.init.kaiming_uniform_(t, a=math.sqrt(5))
                fan_in = t[0].numel()
                v.copy_(t)
            else:
                bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
                t = torch.empty(e['shape'], dtype=torch.float32)
                nn.init.uniform_(t, -bound, bound)
                v.copy_(t)

    def init_scheme(self, init_type='orthogonal', scale=1, std=0.02):
        """weights_init_normal / _kaiming / _orthogonal (model/networks.py:14-55) over every Conv2d / Linear, drawn in
        `net.apply` order so a torch seed gives the reference's weights; biases are zeroed, GroupNorm is left alone
        (the reference's BatchNorm2d branch never matches a module of these networks)."""
        if init_type not in ('normal', 'kaiming', 'orthogonal'):
            raise NotImplementedError('initialization method [{:s}] not implemented'.format(init_type))
        for e in self.plan.table:
            name = e['name']
            if _is_norm(name):
                continue
            v = self.plan.view(self.arena.detach(), e)
            if len(e['shape']) >= 2:
                t = torch.empty(e['shape'], dtype=torch.float32)
                if init_type == 'orthogonal':
                    nn.init.orthogonal_(t, gain=1)
                elif init_type == 'normal':
                    nn.init.normal_(t, 0.0, std)
                else:
                    nn.init.kaiming_normal_(t, a=0, mode='fan_in')
                    t *= scale
                v.copy_(t.to(v.device))
            else:
                v.zero_()

    def init_orthogonal(self):
        self.init_scheme('orthogonal')

    # ---- derived weights ---------------------------------------------------------------------------
    def weights_changed(self):
        """Engine-side writes to the arena through raw pointers (fused Adam) are invisible to torch's version counter:
        the writer calls this."""
        self._weights_epoch += 1
        # the library refuses to run on filters it was told are stale (include/sr3_mi355x.h); ensure_derived re-prepares
        L.check(self.plan.lib.sr3_plan_invalidate_derived(self.plan.handle))

    def ensure_derived(self):
        """(Re)build the Winograd filters if the parameters moved or changed since the last build.  Views handed out by
        named_parameters()/state-dict loading share the arena's version counter, so in-place edits through them are
        seen; a no-op (one tuple compare) otherwise."""
        arena = self.arena
        if not arena.is_cuda:
            return
        key = (arena.data_ptr(), arena._version, self._weights_epoch, self.plan.generation)
        if key == self._derived_key:
            return
        import ctypes as C
        lib = self.plan.lib
        need = int(lib.sr3_plan_derived_bytes(self.plan.handle))
        if need == 0:
            self._derived_key = key
            return
        if self._derived is None or self._derived.device != arena.device or self._derived.numel() * 4 < need:
            self._derived = torch.empty((need + 3) // 4, dtype=torch.float32, device=arena.device)
            L.check(lib.sr3_plan_bind_derived(self.plan.handle, L.ptr(self._derived), self._derived.numel() * 4))
        stream = C.c_void_p(torch.cuda.current_stream(arena.device).cuda_stream)
        L.check(lib.sr3_plan_prepare_derived(self.plan.handle, L.ptr(arena), stream))
        self._derived_key = key

    # ---- parameter / state-dict surface --------------------------------------------------------
    def named_parameters(self, prefix='', recurse=True, remove_duplicate=True):
        for e in self.plan.table:
            yield (prefix + ('.' if prefix else '') + e['name'], self.plan.view(self.arena.detach(), e))

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self.variant == 'ddpm':
            destination[prefix + 'time_mlp.0.inv_freq'] = self.freq.detach().clone()
        for e in self.plan.table:
            destination[prefix + e['name']] = self.plan.view(self.arena.detach(), e).detach().clone().contiguous()

    def state_dict(self, *args, destination=None, prefix='', keep_vars=False):
        # reference key order: buffers of a submodule come after its parameters; rebuild in table order
        if destination is None:
            destination = OrderedDict()
        self._save_to_state_dict(destination, prefix, keep_vars)
        return destination

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        known = set()
        for e in self.plan.table:
            key = prefix + e['name']
            known.add(key)
            if key not in state_dict:
                missing_keys.append(key)
                continue
            src = state_dict[key]
            if tuple(src.shape) != tuple(e['shape']):
                error_msgs.append('size mismatch for %s: checkpoint %s vs model %s'
                                  % (key, tuple(src.shape), tuple(e['shape'])))
                continue
            self.plan.view(self.arena.detach(), e).copy_(src.to(self.arena.device, torch.float32))
        fkey = prefix + 'time_mlp.0.inv_freq'
        if self.variant == 'ddpm':
            known.add(fkey)
            if fkey in state_dict:
                self.freq.copy_(state_dict[fkey].to(self.freq.device, torch.float32))
            else:
                missing_keys.append(fkey)
        if strict:
            for k in state_dict.keys():
                if k.startswith(prefix) and k not in known:
                    unexpected_keys.append(k)

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, x, time, *, cond=None, level_table=None, step_dev=None, out=None, ws=None):
        """eps = UNet(x, time).  `x` may already contain the conditioning channels (reference call
        convention `denoise_fn(torch.cat([cond, x], 1), level)`), or they can be passed separately as
        `cond`, which the input conv reads as a virtual concat (nothing is materialised).  `ws`: a caller-owned
        engine.Workspace (the captured reverse loop keeps its own so no other call can move the buffer its graph
        has baked in)."""
        self.ensure_derived()
        kw = {}
        if step_dev is None:
            if self.variant == 'sr3':
                kw['noise_level'] = time
            else:
                kw['timestep'] = time
        return E.unet_forward(self.plan, self.arena.data, self.freq, self._ws if ws is None else ws, x, cond=cond,
                              level_table=level_table, step_dev=step_dev, out=out, **kw)

    def reverse_step(self, x, z, tables, step2, *, cond=None, level_table=None, clip_denoised=True, eps_out=None, ws=None):
        """One whole iteration of the reverse loop in place on `x` (engine.reverse_step): what `GaussianDiffusion.p_sample_loop`
        captures into its hipGraph."""
        self.ensure_derived()
        return E.reverse_step(self.plan, self.arena.data, self.freq, self._ws if ws is None else ws, x, z, tables, step2,
                              cond=cond, level_table=level_table, clip_denoised=clip_denoised, eps_out=eps_out)

    # ---- training step (forward + backward inside the engine) ------------------------------------
    def train_step(self, hr, cond, z, ca, cb, level, tstep, grad_scale, drop_seed=None):
        """One fused forward + backward: returns the sum-reduced loss (0-dim tensor, summed over all data-parallel
        ranks) and leaves d(loss * grad_scale)/d params -- rank-summed -- in `grad_arena`."""
        p_drop = self.dropout if self.training else 0.0
        if drop_seed is None:          # a fresh mask every step, drawn from torch's CPU generator
            drop_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p_drop > 0 else 0
        dev = hr.device
        if getattr(self, 'grad_arena', None) is None or self.grad_arena.device != dev:
            self.grad_arena = torch.zeros_like(self.arena.data)
        loss = torch.zeros(1, device=dev)
        # data parallel (one process per GPU): gradients are summed over ranks, so the 1/(b c h w) factor
        # uses the GLOBAL batch (model/model.py:52-53 under DataParallel); buckets reduce as they get ready
        import torch.distributed as tdist
        from . import dist as _dist
        dp = _dist.dp_world_size() > 1 or ((getattr(self, 'force_dp', False) or _dist.force_collectives)
                                          and tdist.is_available() and tdist.is_initialized())
        red, marks = None, (0, None, None)
        if dp:
            from .dist import GradReducer
            red = getattr(self, '_reducer', None)
            if red is None or red.device != dev:
                red = self._reducer = GradReducer(self.arena.numel(), dev, tdist)
            grad_scale = grad_scale / tdist.get_world_size()
            marks = red.mark_args()
        self._engine_train_step(hr, cond, z, ca, cb, level, tstep, grad_scale, p_drop, drop_seed, marks, loss)
        if dp:
            red.reduce(self.grad_arena, extra=[loss])
        return loss[0]

    def _engine_train_step(self, hr, cond, z, ca, cb, level, tstep, grad_scale, p_drop, drop_seed, marks, loss):
        """The sr3_train_step call itself (q_sample -> UNet forward -> loss -> backward, `marks` = the gradient-ready
        events of the data-parallel buckets)."""
        import ctypes as C
        dev = hr.device
        if dev.type != 'cuda':
            raise L.Sr3Error('training needs the model on a GPU; there is no CPU fallback')
        self.ensure_derived()          # the train plan's block1 / Upsample convs run on the Winograd kernel
        B = hr.shape[0]
        plan = self.plan
        cc = 0 if cond is None else cond.shape[1]
        need = int(plan.lib.sr3_train_workspace_bytes(plan.handle, B, cc))
        if need == 0:
            raise L.Sr3Error('sr3_train_workspace_bytes failed: %s' % (plan.lib.sr3_last_error() or b'').decode())
        ws = getattr(self, '_train_ws', None)
        if ws is None or ws.numel() < need + 256 or ws.device != dev:
            ws = self._train_ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        off = (-ws.data_ptr()) % 256
        wsv = ws[off:off + need]
        n_marks, offs, evs = marks
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(plan.lib.sr3_train_step(plan.handle, L.ptr(hr), L.ptr(cond), cc, L.ptr(z), L.ptr(ca), L.ptr(cb),
                                        L.ptr(level), L.ptr(tstep), L.ptr(self.freq), L.ptr(self.arena.data),
                                        L.ptr(self.grad_arena), L.ptr(wsv), need, L.ptr(loss), C.c_float(grad_scale),
                                        C.c_float(p_drop), C.c_uint(drop_seed & 0xFFFFFFFF), n_marks, offs, evs, B, stream))

    def named_gradients(self):
        """(reference key, gradient view in the reference shape) after a train_step."""
        for e in self.plan.table:
            yield e['name'], self.plan.view(self.grad_arena, e)

    def extra_repr(self):
        d = self.plan.desc
        return 'variant=%s, in=%d, out=%d, inner=%d, groups=%d, mults=%s, attn_res=%s, res_blocks=%d, image=%d, ' \
               'params=%d (packed arena, libsr3_mi355x)' % (
                   self.variant, d.in_channel, d.out_channel, d.inner_channel, d.norm_groups,
                   list(d.channel_mults[:d.n_mults]), list(d.attn_res[:d.n_attn_res]), d.res_blocks, d.image_size,
                   sum(e['numel'] for e in self.plan.table))
