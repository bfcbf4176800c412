"""Domain decomposition on the GPU: 2 and 4 ranks (one process each, gloo transport; on a one-GPU
box they share the device) against the single-domain oracle on the same global mesh."""
import pytest

from test_decomposition import launch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,p,nr", [(2, 4, 3), (4, 4, 2), (2, 2, 3)])
def test_decomposed_solver_matches_oracle(world, p, nr):
    outs = launch("gpu", world, p, nr)
    assert all("gpu ok" in o for o in outs), outs
