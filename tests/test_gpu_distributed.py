"""Domain decomposition on the GPU: 2 and 4 ranks (one process each, gloo transport; on a one-GPU
box they share the device) against the single-domain oracle on the same global mesh."""
import pytest

from test_decomposition import launch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,p,nr", [(2, 4, 3), (4, 4, 2), (2, 2, 3), (2, 5, 2), (2, 8, 1)])
def test_decomposed_solver_matches_oracle(world, p, nr):
    outs = launch("gpu", world, p, nr)
    assert all("gpu ok" in o for o in outs), outs


@pytest.mark.parametrize("world,p,nr", [(2, 4, 2), (4, 4, 2), (4, 2, 3)])
def test_block_split_cube_matches_oracle(world, p, nr):
    """strong scaling layout of bench.py: the square mesh (n_subdiv = 2) block-split 2x1x1 / 2x2x1,
    against the single-domain oracle on the same mesh (2x2x2: host tables and exchange protocol in
    tests/test_decomposition.py -- a one-GPU box takes at most 6 GPU processes)"""
    outs = launch("gpu", world, p, nr, extra=("strong",))
    assert all("gpu ok" in o for o in outs), outs


@pytest.mark.parametrize("world,p,nr,geometry", [(2, 4, 2, "shell_sector"), (4, 3, 2, "shell_sector"), (2, 2, 2, "sheared")])
def test_decomposed_variable_coefficient_on_mapped_mesh(world, p, nr, geometry):
    """BASELINE config 4 on several ranks: the general tensor branch (six coefficients per quadrature point,
    shell sector with the poisson_shell coefficient; full constant tensor on the sheared box) block-split
    over the ranks, operator / smoother parameters / V-cycle / FMG / PCG against the single-domain oracle"""
    outs = launch("gpu", world, p, nr, extra=(geometry,))
    assert all("gpu ok" in o for o in outs), outs


@pytest.mark.parametrize("world,p,nr,extra", [(2, 4, 3, ()), (4, 4, 3, ("strong",)), (2, 8, 2, ()), (2, 4, 3, ("f32",))])
def test_interface_bricks_first_with_overlapped_exchange(monkeypatch, world, p, nr, extra):
    """split schedule forced on every brick level (MGX_OVERLAP_MIN_BRICKS=1): the bricks on the rank
    interface run first, the exchange and the interface post-operation go to the side stream and
    overlap with the interior bricks; results against the single-domain oracle as above"""
    monkeypatch.setenv("MGX_OVERLAP_MIN_BRICKS", "1")
    outs = launch("gpu", world, p, nr, extra=extra)
    assert all("gpu ok" in o for o in outs), outs


@pytest.mark.parametrize("world,p,nr,extra,limit", [(2, 4, 3, (), "0"), (4, 4, 3, ("strong",), "3000"), (2, 4, 3, ("f32",), "40000"),
                                                     (2, 2, 4, ("strong",), "600000")])
def test_agglomerated_coarse_levels(monkeypatch, world, p, nr, extra, limit):
    """the coarse levels on every rank as a whole (mgx_solver_set_agglomeration), default in the other
    tests of this file with the threshold of 600 000 global DoFs: here switched off ("0") and with
    thresholds that put the seam at other levels; same comparisons with the single-domain oracle"""
    if limit == "0":
        monkeypatch.setenv("MGX_AGGLOMERATE", "0")
    else:
        monkeypatch.setenv("MGX_AGGLOMERATE_MAX_DOFS", limit)
    outs = launch("gpu", world, p, nr, extra=extra)
    assert all("gpu ok" in o for o in outs), outs
    assert all(("agglomerated" in o) == (limit != "0") for o in outs), outs


@pytest.mark.parametrize("world,p,nr,extra,overlap", [(2, 4, 3, (), True), (4, 4, 3, ("strong",), True), (2, 2, 4, ("strong",), False),
                                                       (2, 5, 2, (), True), (2, 4, 3, ("f32",), False), (4, 3, 3, (), False)])
def test_fused_transfers_on_decomposed_levels(monkeypatch, world, p, nr, extra, overlap):
    """residual + restriction and prolongation + first post-smoothing iteration inside the brick loop on decomposed
    levels (eight-colour schedule forced on every level, no agglomeration): the DoFs on the rank interface are
    restricted by their owners / corrected by every holder in list kernels after the exchange.  V-cycle, FMG and PCG
    against the single-domain oracle; the same runs with the separate kernels are the other tests of this file"""
    monkeypatch.setenv("MGX_FREE_ONE_MAX", "0")
    monkeypatch.setenv("MGX_FREE_MAX_BRICKS", "0")
    monkeypatch.setenv("MGX_AGGLOMERATE", "0")
    monkeypatch.setenv("MGX_TRACE", "1")
    if overlap:
        monkeypatch.setenv("MGX_OVERLAP_MIN_BRICKS", "1")
    outs = launch("gpu", world, p, nr, extra=extra)
    assert all("gpu ok" in o for o in outs), outs
    assert all("transfer_create: interface rows" in o for o in outs), [o[-2000:] for o in outs]


@pytest.mark.parametrize("world,p,nr,extra", [(2, 4, 3, ()), (4, 3, 2, ("strong",)), (2, 4, 2, ("shell_sector",))])
def test_right_hand_side_assembled_on_the_device(monkeypatch, world, p, nr, extra):
    """mgx_solver_compute_rhs on a decomposed mesh: every rank integrates over its cells on the GPU, the interface
    entries are summed over the ranks; the vectors against the single-domain oracle, then FMG / PCG as usual"""
    monkeypatch.setenv("MGX_TEST_DEVICE_RHS", "1")
    outs = launch("gpu", world, p, nr, extra=extra)
    assert all("gpu ok" in o for o in outs), outs


def test_poisson_shell_harness_on_ranks():
    """tools/poisson_shell.py --gpus N (poisson_shell/program.cc on several ranks): the convergence table of the program --
    FMG reduction, L2 errors, PCG iterations and reduction -- is the single-rank one"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MGX_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)

    def table(extra, cycles="2:6"):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "poisson_shell.py"), "2", "40000", "--cycles", cycles] + extra,
                             cwd=root, env=env, capture_output=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = out.stdout.decode().strip().splitlines()
        rows = [ln.split() for ln in lines[[i for i, ln in enumerate(lines) if ln.strip().startswith("cells")][-1] + 1:]]
        return [(int(r[0]), int(r[1]), float(r[4]), float(r[5]), float(r[7]), int(r[9]), float(r[10])) for r in rows]

    one, three = table([]), table(["--gpus", "3"])
    assert len(one) == len(three) == 4
    for a, b in zip(one, three):
        assert a[:2] == b[:2] and a[5] == b[5]
        assert all(abs(x - y) <= 1e-6 * abs(x) for x, y in zip(a[2:5] + a[6:], b[2:5] + b[6:])), (a, b)
    # no cycle is skipped on a rank count that does not divide the coarse cells: four ranks take the unrefined six-cell
    # shell as a whole, the cells of level 1 of the refined one (12 of 48 each), the coarse cells of the twelve-cell shell
    one, four = table([], "0:6"), table(["--gpus", "4"], "0:6")
    assert len(one) == len(four) == 6
    for a, b in zip(one, four):
        assert a[:2] == b[:2] and a[5] == b[5]
        assert all(abs(x - y) <= 1e-6 * abs(x) for x, y in zip(a[2:5] + a[6:], b[2:5] + b[6:])), (a, b)


def test_bench_launches_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` starts two ranks itself; here over gloo, both on the one GPU"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MGX_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--cells", "32", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["global_dofs"] == (32 * 4 + 1) ** 3
    assert "gloo" in line["config"]["transport"]


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


def test_callback_device_transport_takes_library_owned_buffers():
    """with the torch-NCCL callback transport (native RCCL off or unavailable) the DG ghost exchange hands
    the communicator device pointers that did not come from its alloc callback, and different receive pointers
    for every vector under one plan id: they are wrapped in place and cached per buffer set."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "callback_transport_worker.py")], cwd=root, env=env,
                         capture_output=True, timeout=600)
    assert out.returncode == 0 and b"callback transport ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("world,n_coarse,p,nr,problem", [(2, 6, 4, 2, "shell"), (3, 6, 2, 2, "shell"), (4, 12, 3, 1, "cube"),
                                                          (4, 6, 2, 2, "shell")])
def test_decomposed_hyper_shell_matches_oracle(world, n_coarse, p, nr, problem):
    """BASELINE config 4 on its own mesh on several ranks: the coarse cells of hyper_shell(6 | 12) dealt out to the
    ranks, interface DoFs on the block faces exchanged; operator, diagonal, smoother parameters, V-cycle, FMG and
    PCG against the single-domain oracle on the whole shell (owner weights in the transfers across ranks)"""
    outs = launch("", world, 0, 0, extra=("gpu", str(n_coarse), str(p), str(nr), problem), worker="shell_dist_worker.py")
    assert all("gpu ok" in o for o in outs), outs
