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
