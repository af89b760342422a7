#!/usr/bin/env python3
"""The launch-bound shape (relat8: five kernels of 6-18 us) under the library's plan switches, one at a time:
whole iterations by the wall clock (best of 3 x 800, one sync each) and the two products by HIP events.
tools/exp_relat8_switches.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python")); sys.path.insert(0, ROOT)
import blz, bench
name = sys.argv[1] if len(sys.argv) > 1 else "relat8"
w = bench.WORKLOADS[name]
M, _ = bench.make_matrix(blz, w, w["prime"])
sets = [{}, {"BLZ_SPMV_BLOCKS_PER_CU": "2"}, {"BLZ_SPMV_BLOCKS_PER_CU": "4"}, {"BLZ_SPMV_BLOCKS_PER_CU": "8"},
        {"BLZ_SPMV_BLOCKS_PER_CU": "16"}, {"BLZ_STAGE_ALWAYS": "1"}, {"BLZ_STAGE_ALWAYS": "1", "BLZ_STAGE_U": "4"},
        {"BLZ_NO_FUSE": "1"}, {"BLZ_NO_PACK": "1"}, {"BLZ_NO_REORDER": "1"}, {"BLZ_NO_PANEL": "1"}, {"BLZ_NO_PAD": "1"},
        {"BLZ_NO_MFMA": "1"}, {"BLZ_GRAPH": "1"}, {}]
keys = sorted({k for s in sets for k in s})
for s in sets:
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(s)
    ctx = blz.Context(w["prime"], w["n"]); ctx.set_matrix(M, w["right"]); ctx.init_v(); ctx.iterate(100); ctx.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); done, stopped, _ = ctx.iterate(800); ctx.sync(); best = min(best, (time.perf_counter() - t0) / 800)
        assert done == 800 and not stopped        # (the solve ends near 3087 iterations: stay short of it)
    a, b = ctx.time_kernel(0, 50) * 1e3, ctx.time_kernel(1, 50) * 1e3
    print(f"{name} {str(s):60s}: {best*1e6:7.2f} us/iteration   spmv1 {a:6.1f} us  spmv2 {b:6.1f} us", flush=True)
    ctx.close()
