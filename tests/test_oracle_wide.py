"""Beyond the reference's domain (p > 2^32: it stores u32 and caps p at 2^30-35,
sequential/lanczos_modp.c:189-193) no reference output can exist.  Here the
oracle's SAME code path (already pinned to the reference at small p by
test_oracle_golden.py) is cross-checked at p = 2^61-1 and other wide primes against
an independent restatement in exact Python integers, and against the reference's
own in-loop invariants (correctness_tests / final_check,
sequential/lanczos_modp.c:532-582).
"""
import os

import numpy as np
import pytest

import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P61 = (1 << 61) - 1
WIDE_PRIMES = [P61, (1 << 31) - 1, 4294967311, 2305843009213693907, (1 << 62) - 57]


def py_spmv(M, x, transpose, n, p):
    rows_out = M.ncols if transpose else M.nrows
    y = [0] * (rows_out * n)
    for i, j, a in zip(M.i.tolist(), M.j.tolist(), M.x.tolist()):
        r, c = (j, i) if transpose else (i, j)
        for l in range(n):
            y[r * n + l] = (y[r * n + l] + a * x[c * n + l]) % p
    return y


def py_semi_inverse(M_, n, p):
    def sweep(a, w, d):
        cnt = 0
        for j in range(n):
            piv = next((i for i in range(j, n) if a[i * n + j] != 0), None)
            if piv is None:
                continue
            d[j] = 1
            cnt += 1
            inv = pow(a[piv * n + j], -1, p)
            for k in range(n):
                a[piv * n + k] = a[piv * n + k] * inv % p
                if w is not None:
                    w[piv * n + k] = w[piv * n + k] * inv % p
            for k in range(n):
                a[j * n + k], a[piv * n + k] = a[piv * n + k], a[j * n + k]
                if w is not None:
                    w[j * n + k], w[piv * n + k] = w[piv * n + k], w[j * n + k]
            for i in range(n):
                if i == j:
                    continue
                m = a[i * n + j]
                for k in range(n):
                    a[i * n + k] = (a[i * n + k] - m * a[j * n + k]) % p
                    if w is not None:
                        w[i * n + k] = (w[i * n + k] - m * w[j * n + k]) % p
        return cnt

    a = list(M_)
    sel = [0] * n
    sweep(a, None, sel)
    a = [M_[i * n + j] if sel[i] and sel[j] else 0 for i in range(n) for j in range(n)]
    w = [1 if i == j and sel[i] else 0 for i in range(n) for j in range(n)]
    d = [0] * n
    return sweep(a, w, d), w, d


def py_iteration(M, n, p, right, v, pb):
    nrows = M.ncols if right else M.nrows
    tmp = py_spmv(M, v, not right, n, p)
    Av = py_spmv(M, tmp, right, n, p)
    vtAv = [sum(v[r * n + a] * Av[r * n + b] for r in range(nrows)) % p for a in range(n) for b in range(n)]
    vtAAv = [sum(Av[r * n + a] * Av[r * n + b] for r in range(nrows)) % p for a in range(n) for b in range(n)]
    npiv, winv, d = py_semi_inverse(vtAv, n, p)
    if npiv == 0:
        return npiv, v, pb, tmp, (vtAv, vtAAv, winv, d)
    spl = [vtAAv[i * n + j] if d[j] else vtAv[i * n + j] for i in range(n) for j in range(n)]
    c = [(-sum(winv[i * n + k] * spl[k * n + j] for k in range(n))) % p for i in range(n) for j in range(n)]
    vd = [(-vtAv[i * n + j]) % p if d[j] else 0 for i in range(n) for j in range(n)]
    nv, npb = [0] * (nrows * n), [0] * (nrows * n)
    for r in range(nrows):
        for j in range(n):
            t = Av[r * n + j] if d[j] else v[r * n + j]
            t += sum(v[r * n + k] * c[k * n + j] + pb[r * n + k] * vd[k * n + j] for k in range(n))
            nv[r * n + j] = t % p
            q = 0 if d[j] else pb[r * n + j]
            npb[r * n + j] = (q + sum(v[r * n + k] * winv[k * n + j] for k in range(n))) % p
    return npiv, nv, npb, tmp, (vtAv, vtAAv, winv, d)


@pytest.mark.parametrize("p", WIDE_PRIMES)
@pytest.mark.parametrize("name,n,right", [("quirks40x30", 2, False), ("quirks40x30", 3, True),
                                          ("trefethen20", 4, False)])
def test_wide_prime_trajectory_vs_python_ints(p, name, n, right):
    M = orc.Matrix.load(os.path.join(GOLDEN, name + ".mtx"), p)
    nrows = M.ncols if right else M.nrows
    recs = []
    res = orc.block_lanczos(M, n, p, right=right, trace=recs.append)
    v = [int(x) for x in orc.init_v(nrows, n, p)]
    pb = [0] * (nrows * n)
    for k, r in enumerate(recs):
        assert [int(x) for x in r["v"]] == v
        npiv, v, pb, tmp, (vtAv, vtAAv, winv, d) = py_iteration(M, n, p, right, v, pb)
        assert r["npiv"] == npiv
        assert [int(x) for x in r["vtAv"]] == vtAv and [int(x) for x in r["vtAAv"]] == vtAAv
        assert [int(x) for x in r["winv"]] == winv and [int(x) for x in r["d"]] == d
        assert [int(x) for x in r["tmp"]] == tmp
    assert recs[-1]["npiv"] == 0
    assert [int(x) for x in res["v"]] == v


@pytest.mark.parametrize("p", [P61, 1073741789])
def test_invariants_of_reference_correctness_tests(p):
    """sequential/lanczos_modp.c:532-557 evaluated on the oracle at 64 bits, and final_check :560-582."""
    n = 4
    M = orc.Matrix.load(os.path.join(GOLDEN, "rand300x200.mtx"), p)
    recs = []
    res = orc.block_lanczos(M, n, p, trace=recs.append)
    for r in recs:
        A, B, W, d = (r[k].astype(object) for k in ("vtAv", "vtAAv", "winv", "d"))
        A, B, W = A.reshape(n, n), B.reshape(n, n), W.reshape(n, n)
        assert (A == A.T).all() and (B == B.T).all() and (W == W.T).all()
        for i in range(n):
            for j in range(n):
                assert W[i, j] == 0 or d[i] or d[j]
        D = np.diag(d)
        assert ((W.dot(A.dot(D))) % p == D).all()
    assert orc.final_check(M.nrows, M.ncols, n, res["v"], res["tmp"]) == 3  # v != 0 and v^T M == 0
    y = orc.spmv(M, res["v"], True, n, p)
    assert not y.any()


def test_omp_kernels_equal_sequential_at_wide_prime():
    p, n = P61, 8
    M = orc.Matrix.load(os.path.join(GOLDEN, "rand3000x2000.mtx"), p)
    v = orc.init_v(M.nrows, n, p)
    t = orc.spmv(M, v, True, n, p)
    assert np.array_equal(orc.spmv_omp(M, v, True, n, p, 4), t)
    Av = orc.spmv(M, t, False, n, p)
    assert np.array_equal(orc.spmv_omp(M, t, False, n, p, 4), Av)
    a, b = orc.block_dot(M.nrows, Av, v, n, p)
    a2, b2 = orc.block_dot(M.nrows, Av, v, n, p, omp_threads=4)
    assert np.array_equal(a, a2) and np.array_equal(b, b2)
    # one whole OpenMP iteration == one oracle iteration
    res = orc.block_lanczos(M, n, p, stop_after=1)
    vv, tt, aa, pp = v.copy(), np.zeros(max(M.nrows, M.ncols) * n, np.uint64), np.zeros(M.nrows * n, np.uint64), \
        np.zeros(M.nrows * n, np.uint64)
    assert orc.iteration_omp(M, n, p, False, vv, tt, aa, pp, 4) > 0
    assert np.array_equal(vv, res["v"]) and np.array_equal(pp, res["p"])


@pytest.mark.parametrize("p,n,right", [(P61, 8, False), (1073741789, 4, True), ((1 << 62) - 57, 3, False)])
def test_by_rows_omp_iteration_equals_sequential(p, n, right):
    """orc_iteration_csr_omp (CSR built once, one output row per loop iteration, no per-thread copies of the output):
    the form bench.py times as cpu_baseline.  Same words as the sequential restatement, for both products and over
    several whole iterations."""
    M = orc.Matrix.load(os.path.join(GOLDEN, "rand3000x2000.mtx"), p)
    pair = orc.CsrPair(M)
    rows_v = M.ncols if right else M.nrows
    x = orc.init_v(M.ncols, n, p)
    assert np.array_equal(pair.spmv(x, False, n, p, 4), orc.spmv(M, x, False, n, p))
    y = orc.init_v(M.nrows, n, p)
    assert np.array_equal(pair.spmv(y, True, n, p, 3), orc.spmv(M, y, True, n, p))
    want = orc.block_lanczos(M, n, p, right=right, stop_after=4)
    v = orc.init_v(rows_v, n, p)
    tmp, Av, pb = np.zeros(max(M.nrows, M.ncols) * n, np.uint64), np.zeros(rows_v * n, np.uint64), np.zeros(rows_v * n, np.uint64)
    for _ in range(4):
        assert pair.iteration(n, p, right, v, tmp, Av, pb, 4) > 0
    pair.close()
    assert np.array_equal(v, want["v"]) and np.array_equal(pb, want["p"])
