import os
import sys

import pytest

# the parity cases are small: let every level that has at least one brick run the brick cell loop
# (the production default switches to it from 512 bricks on (1024 for p <= 2); test_gpu_parity.py has one case at the
# default threshold)
os.environ.setdefault("MGX_BRICK_MIN", "1")
# likewise the colour-by-colour (atomic-free) restriction, production default from 16384 coarse cells on
os.environ.setdefault("MGX_RESTRICT_COLOUR_MIN", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


def pytest_collection_modifyitems(config, items):
    """test_cell_by_cell_brick_kernel needs kernels that only the cross-check build of the library carries: in a process
    that runs the production library its six cases are taken out of the collection (not skipped) -- they run in the
    child process of test_cell_by_cell_brick_kernel_in_the_crosscheck_library, which loads libmgx_crosscheck.so and
    asserts "6 passed"."""
    try:
        import multigrid_amd as mg
        has = bool(mg._lib.load().mgx_has_cells_form())
    except Exception:
        return
    if has:
        return
    keep, drop = [], []
    for it in items:
        (drop if it.originalname == "test_cell_by_cell_brick_kernel" else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep
