#!/usr/bin/env python3
"""Experiment: what the SpMV kernels reach when the gathers HIT (banded supports), per shape -- the part of the run time
that is not the fabric.  Entries of row r uniformly in a band of `band` columns centred on r * C / R; band = 0 is the
uniform generator (the bench matrix).  Reorder off (the band IS the order)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python")); sys.path.insert(0, ROOT)
import blz, bench
os.environ["BLZ_NO_REORDER"] = "1"
for name in sys.argv[1:] or ["gl7d19", "relat9"]:
    w = bench.WORKLOADS[name]
    for band in (0, 1024, 16384, 65536, 524288):
        if band:
            M = blz.Matrix.synth_structured(w["rows"], w["cols"], w["nnz"], w["seed"], w["prime"], pattern=w["pattern"],
                                            hot_pct=0, band_pct=100, band=band)
        else:
            M = blz.Matrix.synth(w["rows"], w["cols"], w["nnz"], w["seed"], w["prime"], pattern=w["pattern"])
        for env in ({}, {"BLZ_NO_STAGE": "1"}, {"BLZ_STAGE_ALWAYS": "1"}):
            os.environ.update(env)
            ctx = blz.Context(w["prime"], w["n"]); ctx.set_matrix(M, w["right"]); ctx.init_v(); ctx.iterate(2)
            a, b = ctx.time_kernel(0, 10) * 1e3, ctx.time_kernel(1, 10) * 1e3
            print(f"{name:8s} band {band:7d} {str(env):28s} spmv1 {a:7.1f} us ({M.nnz / a / 1e3:6.1f} G gathers/s)  "
                  f"spmv2 {b:7.1f} us ({M.nnz / b / 1e3:6.1f} G/s)", flush=True)
            ctx.close()
            for k in env:
                del os.environ[k]
        del M
