#!/usr/bin/env python3
"""A/B sweep of the SpMV grid size (blocks per CU) on the bench workloads; prints mean kernel times (HIP events)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, "block-lanczos-algorithm-parallelization_amd", "python"))
sys.path.insert(0, %r)
import blz, bench
w = bench.WORKLOADS[sys.argv[1]]
M = blz.Matrix.synth(w["rows"], w["cols"], w["nnz"], w["seed"], w["prime"], pattern=w["pattern"])
ctx = blz.Context(w["prime"], w["n"]); ctx.set_matrix(M, w["right"]); ctx.init_v(); ctx.iterate(2)
print(sys.argv[1], "blocks/CU", os.environ.get("BLZ_SPMV_BLOCKS_PER_CU"), "spmv1 %%.1f us  spmv2 %%.1f us" %% (ctx.time_kernel(0, 20)*1e3, ctx.time_kernel(1, 20)*1e3))
''' % (ROOT, ROOT)
for wl in sys.argv[1:] or ["gl7d19", "relat9"]:
    for b in (2, 3, 4, 5, 6, 8, 12, 16):
        env = dict(os.environ, BLZ_SPMV_BLOCKS_PER_CU=str(b))
        subprocess.run([sys.executable, "-c", code, wl], env=env)
