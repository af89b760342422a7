import os, sys
sys.path.insert(0, "block-lanczos-algorithm-parallelization_amd/python"); __file__="tools/exp_skew.py"
src=open('tools/exp_skew.py').read().split("uniform = np.full")[0]
exec(src)
for sigma in (1.0,):
    L = np.clip(rng.lognormal(np.log(20) - sigma * sigma / 2, sigma, R), 1, 20000).astype(np.int64)
    M = build(L)
    for b in ("2","3","4","6","8","12","16"):
        timeit(M, f"sigma={sigma} blocks/CU={b}", {"BLZ_SPMV_BLOCKS_PER_CU": b})
M = build(np.full(R,20))
for b in ("4","8","16"):
    timeit(M, f"uniform blocks/CU={b}", {"BLZ_SPMV_BLOCKS_PER_CU": b})
