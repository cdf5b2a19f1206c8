"""Drop-in `model` package.  `create_model(opt)` is the entry point sr.py / infer.py / sample.py call
(reference model/__init__.py:5-9): it builds the DDPM wrapper around the MI355X engine and logs its class name.

Multi-GPU: the reference switches nn.DataParallel on from `opt['distributed']` (core/logger.py:56-59,
model/networks.py:113-115).  Here a script started once per GPU (`python -m torch.distributed.run --nproc-per-node N
sr.py ...`) joins the data-parallel job in this call -- `sr3_hip.dist.bootstrap()` reads the launcher's environment,
selects `cuda:LOCAL_RANK` and creates the RCCL process group -- so the callers stay unchanged."""
import logging


def create_model(opt):
    from sr3_hip import dist as _dist
    rank, world, _ = _dist.bootstrap()
    from . import model as _model
    wrapper = _model.DDPM(opt)
    logging.getLogger('base').info('Model [%s] is created.' % type(wrapper).__name__ +
                                   ('' if world == 1 else ' (data-parallel rank %d of %d)' % (rank, world)))
    return wrapper
