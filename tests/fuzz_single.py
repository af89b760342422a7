#!/usr/bin/env python3
"""Soak of the single-GPU path at sizes where the plan picks its large-matrix forms (test infrastructure; not collected by pytest):
random uniform, banded, scrambled-band, heavy-tailed and short-row / long-row matrices of 20 k ... 300 k rows, widths 4 ... 16,
four classes of primes, both orientations; a few iterations each and both products on their own, against the oracle word for
word, with the plan left to itself (no switches).  Usage: python tests/fuzz_single.py [cases] [seed]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python"), os.path.join(ROOT, "oracle")]
import blz
import oracle as orc

P61 = (1 << 61) - 1
PRIMES = [P61, P61, 2305843009213693907, (1 << 31) - 1, 4294967311]


def make(rng):
    kind = int(rng.integers(6))
    R = int(rng.integers(20000, 300000))
    if kind == 0:       # uniform, squarish
        C, per = int(R * rng.uniform(0.7, 1.3)), int(rng.integers(2, 45))
        i = np.repeat(np.arange(R), per)
        j = rng.integers(0, C, R * per)
    elif kind == 1:     # band in its own order
        C, per, band = R, int(rng.integers(8, 40)), int(rng.integers(200, 3000))
        i = np.repeat(np.arange(R), per)
        j = (i + rng.integers(-band // 2, band // 2, R * per)) % C
    elif kind == 2:     # scrambled band
        C, per, band = R, int(rng.integers(8, 30)), int(rng.integers(200, 3000))
        i = np.repeat(np.arange(R), per)
        j = (i + rng.integers(-band // 2, band // 2, R * per)) % C
        i, j = rng.permutation(R)[i], rng.permutation(C)[j]
    elif kind == 3:     # heavy-tailed columns + a few very long rows
        C, per = int(R * rng.uniform(0.5, 1.0)), int(rng.integers(5, 25))
        i = np.repeat(np.arange(R), per)
        j = np.minimum((C * rng.random(R * per) ** 4).astype(np.int64), C - 1)
        i[: 20000] = rng.integers(0, 3, 20000)
    elif kind == 4:     # tall: rows of 3, columns of many
        C, per = max(64, R // int(rng.integers(10, 40))), 3
        i = np.repeat(np.arange(R), per)
        j = rng.integers(0, C, R * per)
    else:               # wide
        C, per = R * int(rng.integers(4, 12)), int(rng.integers(20, 90))
        R = max(64, R // 8)
        i = np.repeat(np.arange(R), per)
        j = rng.integers(0, C, R * per)
    names = ["uniform", "band", "scrambled band", "heavy-tailed", "tall", "wide"]
    return names[kind], R, C, np.asarray(i, np.int32), np.asarray(j, np.int32)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    threads = min(16, os.cpu_count() or 1)
    t0 = time.time()
    for case in range(cases):
        name, R, C, i, j = make(rng)
        p = int(PRIMES[rng.integers(len(PRIMES))])
        n = int(rng.choice([4, 8, 8, 12, 16, 16]))
        right = bool(rng.integers(2))
        vals = rng.choice(np.array([1, 1, 1, 2, 3, 2 ** 32 - 1, 2 ** 32 - 2], dtype=np.uint64), size=len(i)) if rng.integers(3) else \
            np.ones(len(i), dtype=np.uint64)
        x = (vals % p).astype(np.uint32)
        M = blz.Matrix(R, C, i, j, x)
        Mo = orc.Matrix(R, C, M.i, M.j, M.x)
        tag = f"case {case}: {name} {R}x{C} nnz {M.nnz} p {p} n {n} {'right' if right else 'left'}"
        pair = orc.CsrPair(Mo)
        with blz.Context(p, n) as ctx:
            ctx.set_matrix(M, right)
            ctx.init_v()
            nv, nt = ctx.rows(blz.V), ctx.rows(blz.TMP)
            vv, tt = ctx.get_block(blz.V).copy(), np.zeros(max(nv, nt) * n, np.uint64)
            aa, pp = np.zeros(nv * n, np.uint64), np.zeros(nv * n, np.uint64)
            its = 3
            for _ in range(its):
                pair.iteration(n, p, right, vv, tt, aa, pp, threads)
            ok = ctx.iterate(its)[0] == its and np.array_equal(ctx.get_block(blz.V), vv) and np.array_equal(ctx.get_block(blz.P), pp)
            for t, src, dst, rows_in in ((not right, blz.V, blz.TMP, nv), (right, blz.TMP, blz.AV, nt)):
                xin = rng.integers(0, p, rows_in * n, dtype=np.uint64)
                ctx.set_block(src, xin)
                ctx.spmv(t, src, dst)
                ok = ok and np.array_equal(ctx.get_block(dst), pair.spmv(xin, t, n, p, threads))
            loc, order = ctx.locality()
        pair.close()
        print(("ok   " if ok else "FAIL ") + tag + f"  (order {order}, lines per entry {loc[0]:.2f} / {loc[1]:.2f})", flush=True)
        if not ok:
            sys.exit(1)
    print(f"{cases} cases in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
