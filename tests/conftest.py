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
    config.addinivalue_line('markers', 'slow: full-length variants of tests that also run shortened by default; selected only when the '
                                       '-m expression names `slow` (e.g. -m "gpu and slow") or SR3_SLOW=1')


def pytest_collection_modifyitems(config, items):
    if 'slow' in (config.getoption('-m') or '') or os.environ.get('SR3_SLOW') == '1':
        return
    skip = pytest.mark.skip(reason='slow: full-length variant (run with -m "gpu and slow"); its shortened form runs by default')
    for it in items:
        if it.get_closest_marker('slow') is not None:
            it.add_marker(skip)


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
