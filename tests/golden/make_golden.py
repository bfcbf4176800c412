#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the CPU oracle (oracle/mg_oracle.c).

The reference itself cannot be run (deal.II is not available), so these vectors are
the oracle's outputs on seeded inputs; the oracle in turn is pinned against the reference's README
transcript (tests/test_oracle_readme.py).  They freeze today's results so that later rounds detect
any drift of either the oracle or the HIP path.  Inputs are produced by the numbering-independent
seeded generator (value = f(seed, lexicographic grid index)), stored in lexicographic grid order.

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import Oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def seeded_lex(n, seed):
    """splitmix64 of (seed, lexicographic grid id) -> uniform [-1,1): same generator as
    mgx_cube_seeded_vector (include/mgx_cube.h)."""
    g = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (g + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def to_lex(o, l, v):
    out = np.empty_like(v)
    out[o.dof_grid(l)] = v
    return out


def from_lex(o, l, vlex):
    return vlex[o.dof_grid(l)]


def main():
    out = {}
    # seeded vmult / residual in/out for p in {1,2,3,4,8} on 2^3 and 4^3 cells
    for p in (1, 2, 3, 4, 8):
        for nr in (1, 2):
            o = Oracle(p, 1, nr)
            l = nr
            n = o.n_dofs(l)
            x = from_lex(o, l, seeded_lex(n, 42))
            b = from_lex(o, l, seeded_lex(n, 43))
            out["vmult_p%d_n%d" % (p, 2 ** nr)] = to_lex(o, l, o.vmult(l, x))
            out["residual_p%d_n%d" % (p, 2 ** nr)] = to_lex(o, l, o.vmult_residual(l, b, x))
            if nr == 2:
                out["prolongate_p%d" % p] = to_lex(o, 2, o.prolongate(2, from_lex(o, 1, seeded_lex(o.n_dofs(1), 44))))
            o.close()
    # C1 (8^3 cells, p=4): Chebyshev parameters, per-cycle residual norms and L2 errors, both
    # V-cycle precisions, program defaults (degree 3, 1 cycle) and README settings (2 cycles)
    for vfloat in (False, True):
        for ncyc in (1, 2):
            o = Oracle(4, 1, 3, degree=3, n_cycles=ncyc, vfloat=vfloat)
            tag = "c1_%s_cyc%d" % ("f32" if vfloat else "f64", ncyc)
            rate, trace = o.solve(True)
            out[tag + "_trace"] = trace
            out[tag + "_fmg"] = np.array([rate, o.l2_error()])
            its, red = o.solve_cg()
            out[tag + "_cg"] = np.array([its, red, o.l2_error()])
            if ncyc == 1:
                out[tag + "_cheb"] = np.array([[o.cheb_info(l)[k] for k in ("lambda_max", "theta", "delta", "degree", "cg_its")]
                                               for l in range(o.n_levels)])
                n = o.n_dofs(3)
                out[tag + "_vcycle"] = to_lex(o, 3, o.vcycle(from_lex(o, 3, seeded_lex(n, 45))))
            o.close()
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_golden.npz"), "with", len(out), "arrays,",
          os.path.getsize(os.path.join(HERE, "oracle_golden.npz")) // 1024, "kB")


if __name__ == "__main__":
    main()
