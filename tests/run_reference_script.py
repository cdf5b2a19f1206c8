"""Runs one of the REFERENCE's own scripts (`infer.py`, `sr.py`, `sample.py` -- the files themselves, through runpy) on top
of the drop-in packages, the way INTEGRATION.md section 1 lays a maintainer's tree out:

    <work>/infer.py, sr.py, sample.py, config/, dataset/, core/logger.py, core/wandb_logger.py  -> symlinks into the reference
    <work>/model, <work>/data, <work>/sr3_hip, <work>/core/metrics.py                            -> symlinks into this repo

`tensorboardX` is not in this image: a stub SummaryWriter stands in.  Nothing of the reference is copied.
Usage: python run_reference_script.py [--force-cpu] [--standins] <reference root> <work dir> <script> [script args...]
  --force-cpu  the parsed options get gpu_ids = None (the reference's own CPU switch, model/base_model.py:9-10; its command
               line cannot express it: core/logger.py:49-55 joins the id list) by wrapping Logger.dict_to_nonedict, the
               function the scripts pass their options through right after Logger.parse
  --standins   the engine calls (train step, fused Adam, reverse loop, uint8 batch transform) are replaced by deterministic
               CPU stand-ins, as tests/dp_dropin_worker.py does: everything AROUND them is the shipped code driven by the
               reference's own script
Exit code 0: the script ran to its end; 3: it stopped in an engine call (`Sr3Error`, printed as `SR3ERROR: ...`) -- what a
host without a GPU must do at the first engine call, as there is no CPU fallback."""
import os
import runpy
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'image-super-resolution-via-iterative-refinement_amd')


def lay_out(ref, work):
    os.makedirs(os.path.join(work, 'core'), exist_ok=True)

    def link(src, dst):
        if not os.path.lexists(dst):
            os.symlink(src, dst)
    for name in os.listdir(ref):
        if name in ('model', 'data', 'core', 'experiments', '.git', '__pycache__'):
            continue
        link(os.path.join(ref, name), os.path.join(work, name))
    for name in os.listdir(os.path.join(ref, 'core')):
        if name.endswith('.py') and name != 'metrics.py':
            link(os.path.join(ref, 'core', name), os.path.join(work, 'core', name))
    link(os.path.join(PKG, 'core', 'metrics.py'), os.path.join(work, 'core', 'metrics.py'))
    for name in ('model', 'data', 'sr3_hip'):
        link(os.path.join(PKG, name), os.path.join(work, name))


def stub_tensorboardx(calls):
    m = types.ModuleType('tensorboardX')

    class SummaryWriter(object):
        def __init__(self, log_dir=None, **kw):
            calls.append(('SummaryWriter', log_dir))

        def add_scalar(self, tag, value, step=None):
            calls.append(('add_scalar', tag, float(value), step))

        def add_image(self, *a, **kw):
            calls.append(('add_image',))

        def close(self):
            pass
    m.SummaryWriter = SummaryWriter
    sys.modules['tensorboardX'] = m


def force_cpu():
    import core.logger as Logger      # the reference's own file (symlink)
    inner = Logger.dict_to_nonedict
    state = {'top': True}

    def wrapped(opt):
        top, state['top'] = state['top'], False
        out = inner(opt)
        if top:
            out['gpu_ids'] = None
            state['top'] = True
        return out
    Logger.dict_to_nonedict = wrapped


def install_standins(log):
    import torch
    import data.util as Util
    from sr3_hip.nn import EngineUNet
    from sr3_hip.optim import EngineAdam
    from sr3_hip.diffusion import EngineDiffusion

    def u8_to_f32(v, flip, min_max, device, out=None):
        x = v.permute(0, 3, 1, 2).float() / 255.0 * (min_max[1] - min_max[0]) + min_max[0]
        if flip is not None:
            f = torch.as_tensor(flip).bool().view(-1, 1, 1, 1)
            x = torch.where(f, x.flip(-1), x)
        return x

    def train_step(self, hr, cond, z, ca, cb, level, tstep, grad_scale, p_drop, drop_seed, marks, loss):
        log.append(('train_step', tuple(hr.shape)))
        self.grad_arena.copy_(torch.linspace(1.0, 2.0, self.arena.numel()) * grad_scale)
        loss[0] = float((hr - (cond if cond is not None else 0)).abs().double().sum())

    def adam(self):
        un = self.netG.denoise_fn
        un.arena.data.sub_(self.defaults['lr'] * un.grad_arena)
        un.weights_changed()
        log.append(('adam',))

    def loop(self, x_in, continous=False, **kw):
        T = self.num_timesteps
        n_snap = sum(1 for i in range(T) if i % (1 | (T // 10)) == 0)
        if not self.conditional:
            shape = tuple(x_in)
            img = torch.zeros(shape)
            first = img
        else:
            first = x_in.float()
            img = first * 0.5
        log.append(('reverse_loop', tuple(img.shape), T, bool(continous)))
        if (not self.conditional) and self.variant == 'ddpm':
            return img.clone()
        ret = torch.cat([first] + [img] * n_snap, 0)
        return ret if continous else ret[-1]
    import core.metrics as Metrics                 # the drop-in's file; its GPU calls get the oracle's CPU restatement
    sys.path.insert(0, REPO)
    from oracle import io_metrics_oracle as IO
    Metrics.tensor2img = IO.tensor2img
    Metrics.calculate_psnr = IO.calculate_psnr
    Metrics.calculate_ssim = IO.calculate_ssim
    Util.u8_batch_to_f32 = u8_to_f32
    EngineUNet._engine_train_step = train_step
    EngineAdam.step = adam
    EngineDiffusion.p_sample_loop = loop


def main():
    args = sys.argv[1:]
    want_cpu = want_standins = False
    while args and args[0].startswith('--'):
        want_cpu |= args[0] == '--force-cpu'
        want_standins |= args[0] == '--standins'
        args = args[1:]
    ref, work, script = args[0], args[1], args[2]
    lay_out(ref, work)
    os.chdir(work)
    calls = []
    stub_tensorboardx(calls)
    sys.argv = [script] + args[3:]
    sys.path.insert(0, work)          # what `python infer.py` does: the script's directory first
    if want_cpu:
        force_cpu()
    if want_standins:
        install_standins(calls)
    try:
        runpy.run_path(os.path.join(work, script), run_name='__main__')
    except Exception as e:            # noqa: BLE001
        from sr3_hip.lib import Sr3Error
        if isinstance(e, Sr3Error):
            import traceback
            tb = traceback.extract_tb(e.__traceback__)
            frames = [f for f in tb if os.path.basename(f.filename) == script]
            print('SR3ERROR: %s' % e)
            print('SCRIPT_LINE: %d' % (frames[-1].lineno if frames else -1))
            sys.exit(3)
        raise
    for c in calls:
        print('CALL: %r' % (c,))


if __name__ == '__main__':
    main()
