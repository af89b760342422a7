#!/usr/bin/env python3
"""Sweep of the staged SpMV's launch parameters on one workload: gathers in flight per lane (U), workgroups per CU and the
staging window (capw).  tools/exp_staged_sweep.py relat9"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python")); sys.path.insert(0, ROOT)
import blz, bench
name = sys.argv[1] if len(sys.argv) > 1 else "relat9"
w = bench.WORKLOADS[name]
M, _ = bench.make_matrix(blz, w, w["prime"])
P = None
grid = itertools.product((4, 8), (3, 4, 5, 6, 8), (512, 1024))
if len(sys.argv) > 2 and sys.argv[2] == "coarse":
    grid = itertools.product((4, 8), (2, 4, 8), (512, 1024))
for U, pc, capw in grid:
    os.environ.update(BLZ_STAGE_U=str(U), BLZ_SPMV_BLOCKS_PER_CU=str(pc), BLZ_STAGE_CAPW=str(capw), BLZ_STAGE_ALWAYS="1")
    ctx = blz.Context(w["prime"], w["n"]); ctx.set_matrix(M, w["right"]); ctx.init_v(); ctx.iterate(2)
    a, b = ctx.time_kernel(0, 10) * 1e3, ctx.time_kernel(1, 10) * 1e3
    print(f"{name} U {U} blocks/CU {pc} capw {capw:5d}: spmv1 {a:7.1f} us  spmv2 {b:7.1f} us", flush=True)
    ctx.close()
