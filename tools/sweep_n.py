#!/usr/bin/env python3
"""Per-kernel time of one iteration against the block width n (GL7d19 shape, p = 2^61-1 and p = 2^31-1).
n in {1,2,4,8,16} takes the specialised kernels (fused inner products, register-resident update); the other widths
the generic ones.  Usage: python tools/sweep_n.py [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python")]
import blz
from bench import WORKLOADS
w = WORKLOADS["gl7d19"]
ns = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 8, 12, 16, 24, 32, 64]
for p in ((1 << 61) - 1, (1 << 31) - 1):
    M = blz.Matrix.synth(w["rows"], w["cols"], w["nnz"], w["seed"], p, pattern=w["pattern"])
    for n in ns:
        with blz.Context(p, n) as ctx:
            ctx.set_matrix(M, False)
            ctx.init_v()
            ctx.iterate(3)
            ctx.profile(True)
            steps = 10
            ctx.iterate(steps)
            prof = ctx.profile_read()
            tot = sum(v["ms_total"] for v in prof.values()) / steps
            parts = "  ".join(f"{k} {v['ms_total'] / steps * 1e3:7.0f}" for k, v in prof.items() if v["launches"])
            print(f"p=2^{p.bit_length()}-1 n={n:2d}: {tot:7.3f} ms/iteration  {2 * M.nnz * n / tot / 1e6:9.1f} G MAC/s   [us] {parts}", flush=True)
