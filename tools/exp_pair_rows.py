#!/usr/bin/env python3
"""Round 3: two words per lane (16-byte gathers) at n = 8 by row length: the relat9 shape solved from the LEFT (first product =
rows of CSR(M^T), 71 entries each, gathering 64-byte rows out of 791 MB) and from the right (first product = rows of 3), with
and without BLZ_NO_PAIR.  Usage: python tools/exp_pair_rows.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python"), ROOT]
import blz
import bench

for name in ("relat9", "gl7d19"):
    w = bench.WORKLOADS[name]
    p, n = w["prime"], w["n"]
    M = blz.Matrix.synth(w["rows"], w["cols"], w["nnz"], w["seed"], p, pattern=w["pattern"])
    for right in (False, True):
        for env in ({}, {"BLZ_NO_PAIR": "1"}, {"BLZ_STAGE_ALWAYS": "1"}, {"BLZ_STAGE_ALWAYS": "1", "BLZ_NO_PAIR": "1"}):
            for k in ("BLZ_NO_PAIR", "BLZ_STAGE_ALWAYS"):
                os.environ.pop(k, None)
            os.environ.update(env)
            with blz.Context(p, n) as ctx:
                ctx.set_matrix(M, right)
                ctx.init_v()
                ctx.iterate(3)
                ctx.profile(True)
                _, _, ms = ctx.iterate(20)
                prof = ctx.profile_read()
                print(f"{name:7s} {'right' if right else 'left ':5s} {str(env):55s} {ms / 20 * 1e3:8.1f} us/iteration ",
                      {k: round(v['ms_total'] / 20 * 1e3, 1) for k, v in prof.items() if v['launches'] and k.startswith('spmv')}, flush=True)
