#!/usr/bin/env python3
"""Round 3: what the iterated-sweeps ordering is worth on the GPU where it is chosen: a band matrix (2 M x 2 M, 20 entries per
row in a band of 4096 columns) whose rows and columns were scrambled, n = 8, p = 2^61-1; per-kernel times with the sweeps among
the candidates (default) and without (BLZ_REORDER_SWEEPS=0)."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python"))
import blz

p, n = (1 << 61) - 1, 8
rng = np.random.default_rng(7)
R = C = 2000000
per, band = 20, 4096
i = np.repeat(np.arange(R), per)
j = (i + rng.integers(-band // 2, band // 2, size=R * per)) % C
x = rng.choice(np.array([1, 1, 1, 2, 3], dtype=np.uint32), size=R * per)
pr, pc = rng.permutation(R), rng.permutation(C)
for name, M in (("scrambled", blz.Matrix(R, C, pr[i], pc[j], x)), ("own order", blz.Matrix(R, C, i, j, x))):
    for sw in ("1", "0"):
        os.environ["BLZ_REORDER_SWEEPS"] = sw
        with blz.Context(p, n) as ctx:
            ctx.set_matrix(M, False)
            ctx.init_v()
            ctx.iterate(5)
            ctx.profile(True)
            done, stopped, ms = ctx.iterate(20)
            prof = ctx.profile_read()
            loc, kind = ctx.locality()
            print(f"{name:10s} sweeps={sw}: order {kind}, lines per entry {loc[0]:.3f} / {loc[1]:.3f}, {ms / 20 * 1e3:7.1f} us per iteration,",
                  {k: round(v['ms_total'] / 20 * 1e3, 1) for k, v in prof.items() if v['launches']}, flush=True)
