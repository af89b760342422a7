"""BASELINE.json's configurations at FULL size (seeded synthetic stand-ins of SURVEY 8(d)) on the GPU.

At these sizes the checks are (a) direct equality with the C oracle where it finishes in seconds (one SpMV per
orientation, one whole iteration with its OpenMP kernels), and (b) size-independent properties of the domain:
linearity of the block SpMV mod p, the adjoint identity U^T (M W) = (M^T U)^T W, and the reference's own per-iteration
invariants (correctness_tests, sequential/lanczos_modp.c:532-557) on the n x n operands.
"""
import os
import sys

import numpy as np
import pytest

import blz
import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402  (shapes of BASELINE configs 2, 3, 4)


def addmod(a, b, p):
    s = a + b          # p < 2^62: no wrap in uint64
    return np.where(s >= p, s - np.uint64(p), s)


@pytest.mark.parametrize("name", ["relat8", "gl7d19", "relat9", "synth5q"])
def test_full_size_config(name):
    """synth5q is config 5's shape at 1/4 linear scale (12.5 M x 12.5 M, 5e8 entries, all ones, n = 16): columns beyond
    2^24 (the stream cannot be packed), 1.6 GB operands, 128-byte block rows, multi-millisecond launches."""
    w = WORKLOADS[name]
    p, n, right = w["prime"], w["n"], w["right"]
    M = blz.Matrix.synth(w["rows"], w["cols"], w["nnz"], w["seed"], p, pattern=w["pattern"])
    Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
    pair = orc.CsrPair(Mo)          # the oracle's by-rows OpenMP kernels (no per-thread copies of the output block)
    threads = min(16, os.cpu_count() or 1)
    big = name == "synth5q"
    with blz.Context(p, n) as ctx:
        ctx.set_matrix(M, right)
        ctx.init_v()
        nv, nt = ctx.rows(blz.V), ctx.rows(blz.TMP)
        v0 = ctx.get_block(blz.V)
        assert np.array_equal(v0[:64 * n], orc.init_v(64, n, p))

        # one whole iteration == the oracle's (OpenMP kernels; they equal the sequential ones, test_oracle_wide.py)
        vv, tt = v0.copy(), np.zeros(max(nv, nt) * n, np.uint64)
        aa, pp = np.zeros(nv * n, np.uint64), np.zeros(nv * n, np.uint64)
        assert pair.iteration(n, p, right, vv, tt, aa, pp, threads) > 0
        done, stopped, _ = ctx.iterate(1)
        assert (done, stopped) == (1, False)
        assert np.array_equal(ctx.get_block(blz.V), vv) and np.array_equal(ctx.get_block(blz.P), pp)
        assert np.array_equal(ctx.get_block(blz.AV), aa)
        del vv, tt, aa, pp

        # the reference's in-loop invariants on the n x n operands of that iteration (exact integers)
        A, B, Wi, d = (ctx.get_small(k).astype(object) for k in (blz.VTAV, blz.VTAAV, blz.WINV, blz.D))
        A, B, Wi = A.reshape(n, n), B.reshape(n, n), Wi.reshape(n, n)
        assert (A == A.T).all() and (B == B.T).all() and (Wi == Wi.T).all()
        D = np.diag(d)
        assert ((Wi.dot(A.dot(D))) % p == D).all()

        # linearity of both products on fresh random blocks: M(a+b) = Ma + Mb
        rng = np.random.default_rng(1)
        for transpose, rows_in, src, dst in ((not right, nv, blz.V, blz.TMP), (right, nt, blz.TMP, blz.AV)):
            a = rng.integers(0, p, rows_in * n, dtype=np.uint64)
            b = rng.integers(0, p, rows_in * n, dtype=np.uint64)
            ctx.set_block(src, a)
            ctx.spmv(transpose, src, dst)
            ya = ctx.get_block(dst)
            ctx.set_block(src, b)
            ctx.spmv(transpose, src, dst)
            yb = ctx.get_block(dst)
            ctx.set_block(src, addmod(a, b, p))
            ctx.spmv(transpose, src, dst)
            assert np.array_equal(ctx.get_block(dst), addmod(ya, yb, p))
            if transpose == (not right) or big:
                # and the product against the oracle at full size (both orientations for the config-5 shape)
                assert np.array_equal(ya, pair.spmv(a, transpose, n, p, threads))
            del a, b, ya, yb

        # adjoint identity through the block products: U^T (B^T W) = (B U)^T W with B = the first product's matrix
        U = rng.integers(0, p, nv * n, dtype=np.uint64)
        Wb = rng.integers(0, p, nt * n, dtype=np.uint64)
        ctx.set_block(blz.V, U)
        ctx.set_block(blz.TMP, Wb)
        ctx.spmv(right, blz.TMP, blz.AV)                 # AV = B^T W   (rows of v)
        lhs, _ = ctx.block_dot()                         # U^T (B^T W)
        ctx.spmv(not right, blz.V, blz.TMP)              # TMP = B U    (rows of tmp)
        BU = ctx.get_block(blz.TMP)
        rhs, _ = orc.block_dot(nt, Wb, BU, n, p, omp_threads=threads)   # (B U)^T W, summed on the host
        assert np.array_equal(lhs, rhs)
    pair.close()


@pytest.mark.parametrize("rank", [0, 7])
def test_config5_one_ranks_share_at_full_size(rank):
    """BASELINE config 5 at ITS OWN size, as one of its 8 GPUs sees it: the 50 M x 50 M / 2e9-entry all-ones matrix, n = 16,
    p = 2^61-1, 8-way row partition.  The rank generates only its own 6.25 M rows and its own 6.25 M columns (2 x 2.5e8
    entries, column indices up to 5e7 > 2^24, blz_synth_coo_part), prepares from that share alone (blz_prepare_rank) and
    multiplies its slabs by the FULL 6.4 GB gathered operand (external-exchange mode: set_block hands over what the
    all-gather would deliver).  One product per orientation against the oracle's by-rows kernel on those rows (the
    reference's per-term semantics, sequential/lanczos_modp.c:277-286), plus linearity on one of them.  Ranks 0 and 7: the
    first and the last slab of the gathered layout."""
    w = WORKLOADS["synth5"]
    p, n, nranks = w["prime"], w["n"], 8
    R_, C_, nnz = w["rows"], w["cols"], w["nnz"]
    shape = (R_, C_, nnz, w["seed"])
    rb = [R_ * g // nranks for g in range(nranks + 1)]
    cb = [C_ * g // nranks for g in range(nranks + 1)]
    threads = min(16, os.cpu_count() or 1)
    rows_part = blz.Matrix.synth_part(*shape, p, rows=(rb[rank], rb[rank + 1]), pattern=True)
    cols_part = blz.Matrix.synth_part(*shape, p, cols=(cb[rank], cb[rank + 1]), pattern=True)
    assert rows_part.nnz == (rb[rank + 1] - rb[rank]) * (nnz // R_) and rows_part.j.max() >= 1 << 24
    assert abs(cols_part.nnz - nnz // nranks) < 1e-3 * nnz // nranks          # binomial column degrees
    with blz.Context(p, n) as ctx:
        with blz.Prepared.prepare_rank(rows_part, cols_part, R_, C_, nnz, False, rank, nranks, rb, cb) as P:
            ctx.set_matrix_prepared(P, rank)
        ctx.set_exchange_mode(True)
        assert ctx.local_nnz(False) == rows_part.nnz and ctx.local_nnz(True) == cols_part.nnz
        rng = np.random.default_rng(50 + rank)
        # product 0: rows [rb] of M x (operand on the column side, TMP); product 1: rows [cb] of M^T x (operand V)
        for transpose, src, dst, part, lo, hi in ((False, blz.TMP, blz.AV, rows_part, rb[rank], rb[rank + 1]),
                                                  (True, blz.V, blz.TMP, cols_part, cb[rank], cb[rank + 1])):
            rows_in = ctx.rows(src)
            x = rng.integers(0, p, rows_in * n, dtype=np.uint64)              # the whole gathered operand: 6.4 GB
            ctx.set_block(src, x)
            ctx.spmv(transpose, src, dst)
            y = ctx.get_block(dst)[lo * n:hi * n].copy()
            # the same rows on the host, by rows, from the share alone (local row numbers)
            if transpose:
                Mo = orc.Matrix(R_, hi - lo, part.i, part.j - np.int32(lo), part.x)
            else:
                Mo = orc.Matrix(hi - lo, C_, part.i - np.int32(lo), part.j, part.x)
            A = orc.CsrOne(Mo, transpose)
            want = A.spmv(x, n, p, threads)
            A.close()
            del Mo
            assert np.array_equal(y, want), (rank, transpose)
            del want
            if transpose == (rank == 7):
                # linearity at this size: A (x + b) = A x + A b
                b = rng.integers(0, p, rows_in * n, dtype=np.uint64)
                ctx.set_block(src, b)
                ctx.spmv(transpose, src, dst)
                yb = ctx.get_block(dst)[lo * n:hi * n].copy()
                ctx.set_block(src, addmod(x, b, p))
                ctx.spmv(transpose, src, dst)
                assert np.array_equal(ctx.get_block(dst)[lo * n:hi * n], addmod(y, yb, p))
                del b, yb
            del x, y


def test_full_solve_finds_verified_kernel_vectors():
    """A whole solve at moderate scale (24 k iterations): 200 000 x 190 000, 2 M entries, n=8, p=2^61-1.  More rows than
    columns, so a left kernel exists; the block returned must be non-zero and annihilate M -- checked on the host
    with the oracle's SpMV, i.e. independently of every GPU kernel."""
    p, n = (1 << 61) - 1, 8
    M = blz.Matrix.synth(200000, 190000, 2000000, 0x534F4C56, p)
    res = blz.solve(M, p, n, batch=512)
    assert res["final_check"] == (True, True)
    assert 190000 // n - 50 <= res["iterations"] <= 190000 // n + 1
    Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
    assert res["v"].any()
    assert not orc.spmv_omp(Mo, res["v"], True, n, p, min(16, os.cpu_count() or 1)).any()
