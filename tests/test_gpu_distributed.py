"""Domain decomposition on the GPU: 2 and 4 ranks (one process each, gloo transport; on a one-GPU
box they share the device) against the single-domain oracle on the same global mesh."""
import pytest

from test_decomposition import launch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,p,nr", [(2, 4, 3), (4, 4, 2), (2, 2, 3), (2, 5, 2), (2, 8, 1)])
def test_decomposed_solver_matches_oracle(world, p, nr):
    outs = launch("gpu", world, p, nr)
    assert all("gpu ok" in o for o in outs), outs


def test_decomposed_mixed_precision_solver_matches_oracle():
    """the reference's default: fp32 V-cycle inside the fp64 outer iteration, on two ranks"""
    outs = launch("gpu", 2, 4, 3, extra=("f32",))
    assert all("gpu ok" in o for o in outs), outs


def test_native_rccl_transport_single_rank():
    """torch's RCCL process group and the library's own RCCL communicator in one process (one rank:
    the box has one GPU); tools/rccl_selftest.py exercises send/recv inside a group on the stream."""
    outs = launch("nccl", 1, 4, 2)
    assert "gpu ok" in outs[0] and "native RCCL" in outs[0], outs


def test_native_rccl_send_recv_to_self():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_selftest.py")], cwd=root,
                         capture_output=True, timeout=300)
    assert out.returncode == 0 and b"rccl selftest ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
