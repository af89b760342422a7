#!/usr/bin/env python3
"""Round 3: two words per lane at n = 16 by row length and operand size (shapes of relat9 and GL7d19 at n = 16, both orientations)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python"), ROOT]
import blz
import bench

n = 16
for name in ("relat9", "gl7d19"):
    w = bench.WORKLOADS[name]
    p = w["prime"]
    M = blz.Matrix.synth(w["rows"], w["cols"], w["nnz"], w["seed"], p, pattern=w["pattern"])
    for right in (False, True):
        for env in ({"BLZ_STAGE_ALWAYS": "1"}, {"BLZ_STAGE_ALWAYS": "1", "BLZ_NO_PAIR": "1"}, {"BLZ_STAGE_ALWAYS": "1", "BLZ_SPMV_BLOCKS_PER_CU": "8"}, {}):
            for k in ("BLZ_NO_PAIR", "BLZ_STAGE_ALWAYS", "BLZ_SPMV_BLOCKS_PER_CU"):
                os.environ.pop(k, None)
            os.environ.update(env)
            with blz.Context(p, n) as ctx:
                ctx.set_matrix(M, right)
                ctx.init_v()
                ctx.iterate(3)
                ctx.profile(True)
                _, _, ms = ctx.iterate(10)
                prof = ctx.profile_read()
                print(f"{name:7s} n=16 {'right' if right else 'left ':5s} {str(env):75s} {ms / 10 * 1e3:8.1f} us/iteration ",
                      {k: round(v['ms_total'] / 10 * 1e3, 1) for k, v in prof.items() if v['launches'] and k.startswith('spmv')}, flush=True)
