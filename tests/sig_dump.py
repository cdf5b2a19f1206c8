"""Dump the public call surface of a `model` package (the reference's or the drop-in's) as JSON: used by
tests/test_boundary_cpu.py in two subprocesses, because both trees call their package `model`.
usage: python sig_dump.py <root that contains model/>"""
import inspect
import json
import logging
import sys

root = sys.argv[1]
sys.path = [root] + [p for p in sys.path if p not in ('', root)]
logging.disable(logging.CRITICAL)
import model as M                                    # noqa: E402
import model.model as MM                             # noqa: E402
import model.networks as N                           # noqa: E402
import model.base_model as BM                        # noqa: E402
from model.sr3_modules import diffusion as SD, unet as SU      # noqa: E402
from model.ddpm_modules import diffusion as DD, unet as DU     # noqa: E402


def sig(f):
    out = []
    for p in inspect.signature(f).parameters.values():
        d = None if p.default is inspect.Parameter.empty else repr(p.default)
        out.append([p.name, p.kind.name, d])
    return out


def methods(cls, names=None):
    res = {}
    for n, f in inspect.getmembers(cls, predicate=inspect.isfunction):
        if n.startswith('_') and n != '__init__':
            continue
        if names is None or n in names:
            res[n] = sig(f)
    return res


out = {
    'create_model': sig(M.create_model),
    'define_G': sig(N.define_G),
    'init_weights': sig(N.init_weights),
    'DDPM': methods(MM.DDPM),
    'BaseModel': methods(BM.BaseModel),
    'sr3.GaussianDiffusion': methods(SD.GaussianDiffusion),
    'ddpm.GaussianDiffusion': methods(DD.GaussianDiffusion),
    'sr3.UNet': methods(SU.UNet, ('__init__', 'forward')),
    'ddpm.UNet': methods(DU.UNet, ('__init__', 'forward')),
    'sr3.make_beta_schedule': sig(SD.make_beta_schedule),
    'ddpm.make_beta_schedule': sig(DD.make_beta_schedule),
}
print(json.dumps(out))
