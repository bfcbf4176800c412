"""The problem of `bench.py --gpus 8` (FE_Q(4), 128^3 cells block-split 2x2x2) solved by eight ranks that are eight
threads on ONE GPU (tests/thread_ranks.py): PCG iterations and global L2 error of every rank.  No timing value --
the ranks share the device -- but the whole decomposed code path of the 8-GPU run at its real size.
usage: eight_ranks_128.py [log2 cells = 7]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "2")
import test_gpu_eight_ranks as t  # noqa: E402
k = int(sys.argv[1]) if len(sys.argv) > 1 else 7
t0 = time.time()
res = t.benchmark_problem(k)
print("%d^3 cells over 8 thread-ranks: (PCG iterations, L2 error, coarse levels agglomerated) per rank" % 2 ** k)
for r in res:
    print(r)
print("%.1f s; README.md:135-159: 8 iterations, 4.207e-10 at 128^3 cells" % (time.time() - t0))
