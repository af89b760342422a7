#!/usr/bin/env python3
"""Round 3: one rank's two products of BASELINE config 5 at their real size -- 6.25 M rows / columns of the 50 M x 50 M, 2e9-entry
matrix (2 x 2.5e8 entries) against the FULL 6.4 GB gathered operand, n = 16 -- timed under the SpMV's switches.  The quarter-scale
shape of the bench (1.6 GB operand) under-states what the TLB costs at 6.4 GB, and this, not the quarter
shape, is what each of config 5's 8 GPUs executes.  Usage: python tools/exp_config5_rank.py [rank]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python"), ROOT]
import blz
import bench

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
w = bench.WORKLOADS["synth5"]
p, n, nranks = w["prime"], w["n"], 8
R_, C_, nnz = w["rows"], w["cols"], w["nnz"]
rb = [R_ * g // nranks for g in range(nranks + 1)]
cb = [C_ * g // nranks for g in range(nranks + 1)]
t0 = time.time()
rows_part = blz.Matrix.synth_part(R_, C_, nnz, w["seed"], p, rows=(rb[rank], rb[rank + 1]), pattern=True)
cols_part = blz.Matrix.synth_part(R_, C_, nnz, w["seed"], p, cols=(cb[rank], cb[rank + 1]), pattern=True)
print(f"rank {rank}: {rows_part.nnz} + {cols_part.nnz} entries generated in {time.time() - t0:.1f} s", flush=True)
rng = np.random.default_rng(5)
x = rng.integers(0, p, R_ * n, dtype=np.uint64)
variants = [dict(), dict(BLZ_SPMV_BLOCKS_PER_CU="8"), dict(BLZ_NO_PAIR="1"), dict(BLZ_NO_PAIR="1", BLZ_SPMV_BLOCKS_PER_CU="4"),
            dict(BLZ_SPMV_BLOCKS_PER_CU="6"), dict(BLZ_SPMV_BLOCKS_PER_CU="8", BLZ_STAGE_CAPW="1024")]
if len(sys.argv) > 2:
    variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in sys.argv[2:]]
for env in variants:
    for k in ("BLZ_SPMV_BLOCKS_PER_CU", "BLZ_NO_PAIR", "BLZ_STAGE_CAPW", "BLZ_STAGE_U", "BLZ_STAGE_RPG", "BLZ_NO_STAGE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with blz.Context(p, n) as ctx:
        with blz.Prepared.prepare_rank(rows_part, cols_part, R_, C_, nnz, False, rank, nranks, rb, cb) as P:
            ctx.set_matrix_prepared(P, rank)
        ctx.set_exchange_mode(True)
        ctx.set_block(blz.V, x)
        ctx.set_block(blz.TMP, x)
        ctx.time_kernel(0, 1)
        ctx.time_kernel(1, 1)
        t1 = ctx.time_kernel(0, 5)          # tmp_g = M^T[C_g, :] v   (rows of CSR(M^T): binomial lengths around 40)
        t2 = ctx.time_kernel(1, 5)          # Av_g = M[R_g, :] tmp    (rows of CSR(M): exactly 40)
        g1, g2 = cols_part.nnz / t1 / 1e6, rows_part.nnz / t2 / 1e6
        print(f"{str(env):70s} product 1 {t1:7.3f} ms ({g1:5.1f} G gathers/s)   product 2 {t2:7.3f} ms ({g2:5.1f} G gathers/s)", flush=True)
