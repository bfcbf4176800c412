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
