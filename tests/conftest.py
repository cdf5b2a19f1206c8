import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'experiments: needs a library built with -DSR3_EXPERIMENTS (skipped otherwise)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(autouse=True)
def _skip_experiment_tiles(request):
    """Cases parametrised on an experiment kernel (tile_cfg / tile 7, 8, 10 = round 1's split_bf16 halo tiles) run only against a library built with -DSR3_EXPERIMENTS; so do tests marked `experiments`."""
    cs = getattr(request.node, 'callspec', None)
    tiles = [cs.params.get(k) for k in ('tile_cfg', 'tile')] if cs else []
    marked = request.node.get_closest_marker('experiments') is not None
    if marked or any(t in (7, 8, 10) for t in tiles):
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from helpers import experiments_built
        if not experiments_built():
            pytest.skip('experiment kernel: the library was built without -DSR3_EXPERIMENTS')
