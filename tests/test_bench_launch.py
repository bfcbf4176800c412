"""bench.py --gpus N must really start N ranks: launcher logic on the CPU, no GPU needed
(--dry-run: rendezvous over gloo, host-side decomposition tables, one interface exchange)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env_extra=None):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), cwd=ROOT, env=env,
                          capture_output=True, timeout=600)


@pytest.mark.parametrize("n,scaling,grid", [(2, "strong", [2, 1, 1]), (4, "strong", [2, 2, 1]), (2, "weak", [2, 1, 1])])
def test_gpus_flag_launches_ranks(n, scaling, grid):
    out = run_bench("--gpus", str(n), "--dry-run", "--cells", "16", "--scaling", scaling)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == n and line["scaling"] == scaling and line["exchange_ok"]
    assert line["config"]["process_grid"] == grid
    cells = [16 * (g if scaling == "weak" else 1) for g in grid]
    assert line["config"]["global_dofs"] == (cells[0] * 4 + 1) * (cells[1] * 4 + 1) * (cells[2] * 4 + 1)
    if scaling == "strong":  # the ranks split one mesh: less than the whole on each
        assert line["config"]["n_dofs_per_gpu"] < line["config"]["global_dofs"]
    # the real run verifies itself after the timed region (README.md:135-159): on the split mesh the PCG must
    # take the single-domain iteration count and reach its L2 error; the weak family has no README row
    want = {"cg_its": 8, "l2_error": 1.319e-5} if scaling == "strong" else None
    assert line["verify"] == {"expected": want, "ok": None}


def test_no_verify_flag():
    out = run_bench("--gpus", "2", "--dry-run", "--cells", "8", "--no-verify")
    assert out.returncode == 0, out.stderr[-3000:]
    assert json.loads(out.stdout.decode().strip().splitlines()[-1])["verify"] is None


def test_gpus_flag_must_match_world_size():
    out = run_bench("--gpus", "4", "--dry-run", "--cells", "16",
                    env_extra={"WORLD_SIZE": "1", "RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert out.returncode != 0 and b"--gpus 4 but WORLD_SIZE=1" in out.stderr


def test_under_torch_distributed_run():
    """the driver's launch line for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- the ranks exist already, bench.py joins them"""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                          "--cells", "16"], cwd=ROOT, env=env, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.decode().strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1   # rank 0 alone prints the JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["exchange_ok"]
