"""GPU parity: every hot-path entry point of libblz_hip.so, called through the C ABI, against
(a) golden vectors produced by the reference itself and (b) the CPU oracle on the same seeded inputs.
Integer path => bit-exact equality everywhere (no tolerances).
"""
import glob
import hashlib
import os

import numpy as np
import pytest

import blz
import oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MATRIX_OF = {"tref": "trefethen20", "r300": "rand300x200", "wide": "wide120x260",
             "quirks": "quirks40x30", "r3000": "rand3000x2000"}
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))
KERN = sorted(glob.glob(os.path.join(GOLDEN, "kern_*.npz")))
P61 = (1 << 61) - 1


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def load_both(name, prime):
    path = os.path.join(GOLDEN, name + ".mtx")
    return blz.Matrix.load(path, prime), orc.Matrix.load(path, prime)


def as_orc(M):
    return orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)


@pytest.mark.parametrize("path", KERN, ids=[os.path.basename(p)[5:-4] for p in KERN])
def test_each_kernel_against_reference_vectors(path):
    g = np.load(path)
    p, n, right = int(g["prime"]), int(g["n"]), bool(g["right"])
    M, _ = load_both(MATRIX_OF[os.path.basename(path).split("_")[1]], p)
    with blz.Context(p, n) as ctx:
        ctx.set_matrix(M, right)
        assert ctx.word_bytes == (4 if p < 2 ** 32 else 8)
        for it in sorted({k.split("_")[0] for k in g.files if k.startswith("it")}):
            v, tmp, Av, pb = (g[f"{it}_{k}"] for k in ("v", "tmp", "Av", "p"))
            ctx.set_block(blz.V, v)
            ctx.set_block(blz.P, pb)
            ctx.spmv(not right, blz.V, blz.TMP)             # sequential/lanczos_modp.c:635
            assert np.array_equal(ctx.get_block(blz.TMP), tmp)
            ctx.spmv(right, blz.TMP, blz.AV)                # :636
            assert np.array_equal(ctx.get_block(blz.AV), Av)
            a, b = ctx.block_dot()                          # :640
            assert np.array_equal(a, g[f"{it}_vtAv"]) and np.array_equal(b, g[f"{it}_vtAAv"])
            npiv, winv, d = ctx.semi_inverse()              # :644
            assert np.array_equal(winv, g[f"{it}_winv"]) and np.array_equal(d, g[f"{it}_d"])
            assert npiv == int(d.sum())
            ctx.orthogonalize()                             # :652-656
            assert np.array_equal(ctx.get_block(blz.V), g[f"{it}_vnext"])
            assert np.array_equal(ctx.get_block(blz.P), g[f"{it}_pnext"])


def test_semi_inverse_against_reference_vectors():
    g = np.load(os.path.join(GOLDEN, "semi_inverse.npz"))
    for key in sorted(k[:-2] for k in g.files if k.endswith("_M")):
        n, p = int(key.split("_")[0][1:]), int(key.split("_")[1][1:])
        with blz.Context(p, n) as ctx:
            for M, winv, d, npiv in zip(g[key + "_M"], g[key + "_winv"], g[key + "_d"], g[key + "_npiv"]):
                ctx.set_small(blz.VTAV, M)
                got = ctx.semi_inverse()
                assert got[0] == npiv and np.array_equal(got[1], winv) and np.array_equal(got[2], d), key


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[5:-4] for p in TRAJ])
def test_trajectory_against_reference(path):
    """Iteration by iteration: the n x n operands, the pivot sets and sha256(v) of the reference run."""
    g = np.load(path)
    p, n, right, stop = int(g["prime"]), int(g["n"]), bool(g["right"]), int(g["stop_after"])
    M, _ = load_both(MATRIX_OF[os.path.basename(path).split("_")[1]], p)
    step_checks = len(g["npiv"]) <= 80
    with blz.Context(p, n) as ctx:
        ctx.set_matrix(M, right)
        ctx.init_v()
        assert np.array_equal(ctx.get_block(blz.V), g["v0"])
        if step_checks:
            for k in range(len(g["npiv"])):
                if stop > 0 and k == stop:
                    break
                assert sha(ctx.get_block(blz.V)) == str(g["vhash"][k])
                done, stopped, _ = ctx.iterate(1)
                for which, name in ((blz.VTAV, "vtAv"), (blz.VTAAV, "vtAAv"), (blz.WINV, "winv"), (blz.D, "d")):
                    assert np.array_equal(ctx.get_small(which), g[name][k]), (name, k)
                assert stopped == (g["npiv"][k] == 0) and done == (0 if stopped else 1)
        else:
            while True:
                todo = 64 if stop <= 0 else min(64, stop - ctx.iterations)
                if todo <= 0:
                    break
                _, stopped, _ = ctx.iterate(todo)
                if stopped:
                    break
        assert ctx.iterations == int(g["iterations"])
        assert np.array_equal(ctx.get_block(blz.V), g["final_v"])
        if stop <= 0:
            assert np.array_equal(ctx.get_block(blz.TMP), g["final_tmp"])
            nz, zero = ctx.final_check()
            assert nz == bool(g["final_v"].any()) and zero == (not g["final_tmp"].any())
        # a stopped context stays put (later calls are no-ops)
        if stop <= 0:
            before = ctx.get_block(blz.V)
            assert ctx.iterate(3)[:2] == (0, True) and np.array_equal(ctx.get_block(blz.V), before)


@pytest.mark.parametrize("p", [P61, 4294967311, (1 << 62) - 57, 2305843009213693907, 2147483647, 65537, 7, 2])
@pytest.mark.parametrize("name,n,right", [("quirks40x30", 2, False), ("wide120x260", 4, True), ("rand300x200", 8, False)])
def test_full_solve_against_oracle_all_prime_classes(p, name, n, right):
    """Mersenne-61, Mersenne-31, Barrett 32/33/61/62-bit and tiny primes; oracle = same code path that the
    golden tests pin to the reference."""
    M, Mo = load_both(name, p)
    want = orc.block_lanczos(Mo, n, p, right=right)
    got = blz.solve(M, p, n, right=right, batch=7)
    assert got["iterations"] == want["iterations"]
    assert np.array_equal(got["v"], want["v"]) and np.array_equal(got["tmp"], want["tmp"])
    assert np.array_equal(got["p"], want["p"])


@pytest.mark.parametrize("pad", ["padded", "exact"])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 8, 11, 16, 24, 32, 48, 64])
def test_every_block_width(monkeypatch, n, pad):
    """Blocks live in HBM with their width rounded up to a power of two (zero columns), so every width runs the
    specialised kernels; BLZ_NO_PAD=1 keeps the exact width and the generic kernels.  Same words either way."""
    monkeypatch.setenv("BLZ_NO_PAD", "1" if pad == "exact" else "0")
    p = P61 if n % 2 else 1073741789
    M, Mo = load_both("rand300x200", p)
    its = 6 if n <= 32 else 3        # rank(M M^T) <= 200: wider blocks exhaust the Krylov space earlier
    want = orc.block_lanczos(Mo, n, p, stop_after=its)
    got = blz.solve(M, p, n, stop_after=its, batch=4)
    assert got["iterations"] == want["iterations"] == its
    assert np.array_equal(got["v"], want["v"]) and np.array_equal(got["p"], want["p"])


@pytest.mark.parametrize("p", [P61, (1 << 62) - 57, 2147483647, 1073741789])
@pytest.mark.parametrize("n", [16, 32, 64])
def test_wide_blocks_with_every_kind_of_word(n, p):
    """n = 16, 32 and 64 take their own inner-product and update kernels (49 accumulators per lane at n = 32, four
    passes of 16 rotations at n = 64; DPP broadcasts, half- and quarter-sums joined across lanes or through LDS):
    every reducer (Mersenne 61 / 31, Barrett at 64 and 32 bits) on a matrix large enough for several workgroups."""
    M, Mo = load_both("rand3000x2000", p)
    want = orc.block_lanczos(Mo, n, p, stop_after=5)
    got = blz.solve(M, p, n, stop_after=5, batch=5)
    assert got["iterations"] == 5
    assert np.array_equal(got["v"], want["v"]) and np.array_equal(got["p"], want["p"])


def test_n_out_of_range_and_bad_prime_are_errors():
    for (p, n) in ((65537, 0), (65537, 65), (1, 4), (0, 4), (1 << 62, 4)):
        with pytest.raises(blz.BlzError) as e:
            blz.Context(p, n)
        assert e.value.code == blz.EINVAL


def test_degenerate_matrices():
    p, n = P61, 4
    # no entries at all: v^T A v = 0 -> stops at once, v stays the random block, tmp = 0
    Z = blz.Matrix(30, 20, [], [], [])
    got = blz.solve(Z, p, n)
    want = orc.block_lanczos(as_orc(Z), n, p)
    assert got["iterations"] == want["iterations"] == 0 and np.array_equal(got["v"], want["v"])
    assert got["final_check"] == (True, True)
    # a single row / single column
    for (nr, nc) in ((1, 9), (9, 1)):
        A = blz.Matrix(nr, nc, [0] * 3 if nr == 1 else [0, 4, 8], [0, 4, 8] if nr == 1 else [0] * 3, [1, 2, 5])
        for right in (False, True):
            got = blz.solve(A, p, 2, right=right)
            want = orc.block_lanczos(as_orc(A), 2, p, right=right)
            assert got["iterations"] == want["iterations"] and np.array_equal(got["v"], want["v"])


@pytest.mark.parametrize("shape,nnz,n,p,right,pattern", [
    ((40000, 1500, 160000), None, 4, 2147483647, False, False),   # relat8-like aspect, config 2 arithmetic
    ((60000, 2700, 190000), None, 8, P61, True, False),           # relat9-like aspect, config 3 arithmetic
    ((20000, 20500, 390000), None, 8, P61, False, False),         # GL7d19-like density (19.5 / row), config 4
    ((30000, 30000, 1200000), None, 16, P61, False, True),        # config 5: all-ones pattern, 40 / row
])
def test_synthetic_config_shapes_against_oracle(shape, nnz, n, p, right, pattern):
    nr, nc, nz = shape
    M = blz.Matrix.synth(nr, nc, nz, 0x474C3764, p, pattern=pattern)
    Mo = as_orc(M)
    its = 5
    want = orc.block_lanczos(Mo, n, p, right=right, stop_after=its)
    got = blz.solve(M, p, n, right=right, stop_after=its, batch=2)
    assert got["iterations"] == its
    assert np.array_equal(got["v"], want["v"]) and np.array_equal(got["p"], want["p"])


def test_heavy_rows_do_not_overflow():
    """openMP/lanczos_modp.c:352-365 overflows its u64 sums on long rows of large values;
    the 128-bit accumulators here must not: one row of 5000 entries, all 2^32-1, p just below 2^62."""
    p, n = (1 << 62) - 57, 4
    nc = 6000
    i = np.zeros(5000, dtype=np.int32)
    j = np.arange(5000, dtype=np.int32)
    x = np.full(5000, 2 ** 32 - 1, dtype=np.uint32)
    i = np.concatenate([i, np.arange(1, 50, dtype=np.int32)])
    j = np.concatenate([j, np.arange(5000, 5049, dtype=np.int32)])
    x = np.concatenate([x, np.full(49, 3, dtype=np.uint32)])
    M = blz.Matrix(50, nc, i, j, x)
    with blz.Context(p, n) as ctx:
        ctx.set_matrix(M, False)
        xin = np.full(nc * n, p - 1, dtype=np.uint64)
        ctx.set_block(blz.TMP, xin)
        ctx.spmv(False, blz.TMP, blz.AV)
        assert np.array_equal(ctx.get_block(blz.AV), orc.spmv(as_orc(M), xin, False, n, p))
        vin = np.full(50 * n, p - 2, dtype=np.uint64)
        ctx.set_block(blz.V, vin)
        ctx.spmv(True, blz.V, blz.TMP)
        assert np.array_equal(ctx.get_block(blz.TMP), orc.spmv(as_orc(M), vin, True, n, p))


def outlier_matrix(p):
    rng = np.random.default_rng(11)
    nr, nc = 6000, 5000
    ii, jj = [], []
    for r in range(nr):
        k = 600 if r < 100 else (20000 if r in (777, 4242) else (3000 if r % 1500 == 7 else 4))
        cols = rng.choice(nc, size=min(k, nc), replace=False) if k < nc else np.arange(nc)
        ii.append(np.full(len(cols), r, dtype=np.int32))
        jj.append(cols.astype(np.int32))
    ii, jj = np.concatenate(ii), np.concatenate(jj)
    jj[::97] = 13                                        # a dense column too (duplicates within a row are legal)
    xx = rng.choice(np.array([1, 2, 3, 2 ** 32 - 1, 2 ** 32 - 2], dtype=np.uint64), size=len(ii)) % p
    return blz.Matrix(nr, nc, ii, jj, xx.astype(np.uint32))


@pytest.mark.parametrize("n,p", [(8, P61), (4, 2147483647), (16, 1073741789), (3, P61)])
def test_outlier_rows_get_a_workgroup_each(n, p):
    """A few very dense rows and columns among short ones (the shape of real relation matrices): rows above
    max(64, 4 x mean) entries are listed at upload and skipped by the streaming kernels; k_spmv_wave gives a wavefront
    to the medium ones (here the 100 ADJACENT rows of 600, which the renumbering keeps together), k_spmv_heavy a
    workgroup per 4096-entry segment to the long ones (3000 and 5000 entries: one and two segments, the latter
    combined by k_spmv_heavy_combine) -- including their share of the fused inner products (n = 8, 4).  Both
    orientations, through a whole solve."""
    M = outlier_matrix(p)
    for right in (False, True):
        want = orc.block_lanczos(as_orc(M), n, p, right=right, stop_after=3)
        got = blz.solve(M, p, n, right=right, stop_after=3, batch=3)
        assert got["iterations"] == 3
        assert np.array_equal(got["v"], want["v"]) and np.array_equal(got["p"], want["p"])


@pytest.mark.parametrize("n,p", [(8, P61), (5, 1073741789)])
def test_outlier_rows_in_column_chunked_products(monkeypatch, n, p):
    """Same matrix, products cut into 3 column pieces (the multi-GPU pipeline, forced on one rank): a row can be an
    outlier in one piece and ordinary in the next; every piece accumulates into the same output rows."""
    monkeypatch.setenv("BLZ_FORCE_COMM", "1")
    monkeypatch.setenv("BLZ_AG_CHUNKS", "3")
    M = outlier_matrix(p)
    want = orc.block_lanczos(as_orc(M), n, p, right=False, stop_after=3)
    with blz.Context(p, n) as c:
        c.comm_init(blz.comm_unique_id(), 0, 1)
        c.set_matrix(M, False, 0, 1)
        c.init_v()
        c.iterate(3)
        assert np.array_equal(c.get_block(blz.V), want["v"]) and np.array_equal(c.get_block(blz.P), want["p"])


@pytest.mark.parametrize("name,p,n,right", [("rand3000x2000", P61, 8, False), ("rand300x200", 65537, 4, True),
                                            ("quirks40x30", 1073741789, 2, False)])
def test_results_do_not_depend_on_the_internal_renumbering(monkeypatch, name, p, n, right):
    """The solver renumbers rows for locality (blz_reorder); with BLZ_NO_REORDER=1 it keeps the file's numbering.
    Blocks cross the ABI in the original numbering either way and are bit-identical."""
    M, Mo = load_both(name, p)
    want = orc.block_lanczos(Mo, n, p, right=right, stop_after=9)
    for flag in ("0", "1"):
        monkeypatch.setenv("BLZ_NO_REORDER", flag)
        with blz.Context(p, n) as ctx:
            ctx.set_matrix(M, right)
            ctx.init_v()
            assert np.array_equal(ctx.get_block(blz.V), orc.init_v(ctx.rows(blz.V), n, p))
            ctx.iterate(want["iterations"])
            assert np.array_equal(ctx.get_block(blz.V), want["v"]) and np.array_equal(ctx.get_block(blz.P), want["p"])
            # set_block / get_block round trip in the original numbering
            x = np.arange(ctx.rows(blz.TMP) * n, dtype=np.uint64) % p
            ctx.set_block(blz.TMP, x)
            assert np.array_equal(ctx.get_block(blz.TMP), x)
            ctx.spmv(right, blz.TMP, blz.AV)
            assert np.array_equal(ctx.get_block(blz.AV), orc.spmv(Mo, x, right, n, p))


@pytest.mark.parametrize("distinct", [5, 256, 300])
def test_packed_and_plain_matrix_streams_agree(monkeypatch, distinct):
    """Slabs with <= 256 distinct values and < 2^24 columns travel as one u32 per entry (column | palette index);
    more values fall back to the two-array form; BLZ_NO_PACK=1 forces it.  Same words either way."""
    p, n = P61, 8
    rng = np.random.default_rng(5)
    nr, nc, nz = 4000, 3500, 40000
    vals = rng.integers(1, 2 ** 32, size=distinct, dtype=np.uint64)
    M = blz.Matrix(nr, nc, rng.integers(0, nr, nz), rng.integers(0, nc, nz), vals[rng.integers(0, distinct, nz)].astype(np.uint32))
    want = orc.block_lanczos(as_orc(M), n, p, stop_after=4)
    for flag in ("0", "1"):
        monkeypatch.setenv("BLZ_NO_PACK", flag)
        got = blz.solve(M, p, n, stop_after=4, batch=4)
        assert np.array_equal(got["v"], want["v"]) and np.array_equal(got["p"], want["p"])


def test_graph_replay_gives_the_same_trajectory(monkeypatch):
    """BLZ_GRAPH=1: the loop body is captured once into a hipGraph and replayed (all state lives in device memory, so
    the launches have fixed arguments).  Off by default -- measured slower than plain stream launches on ROCm 7.2,
    DESIGN.md section 5 -- but it must give the same words, including the no-op iterations after termination."""
    monkeypatch.setenv("BLZ_GRAPH", "1")
    M, Mo = load_both("rand300x200", 65537)
    want = orc.block_lanczos(Mo, 4, 65537, right=False)
    with blz.Context(65537, 4) as ctx:
        ctx.set_matrix(M, False)
        ctx.init_v()
        done, stopped, _ = ctx.iterate(want["iterations"] + 7)
        assert stopped and done == want["iterations"]
        assert np.array_equal(ctx.get_block(blz.V), want["v"]) and np.array_equal(ctx.get_block(blz.P), want["p"])
        ctx.set_matrix(M, True)                       # a new matrix drops the recorded graph
        ctx.init_v()
        want_r = orc.block_lanczos(Mo, 4, 65537, right=True, stop_after=5)
        ctx.iterate(5)
        assert np.array_equal(ctx.get_block(blz.V), want_r["v"])


def test_abi_misuse_is_reported_not_crashed():
    with blz.Context(P61, 4) as ctx:
        for call in (lambda: ctx.init_v(), lambda: ctx.iterate(1), lambda: ctx.spmv(False, blz.V, blz.TMP),
                     lambda: ctx.set_block(blz.V, np.zeros(4, np.uint64)) if False else ctx.orthogonalize(),
                     lambda: ctx.final_check()):
            with pytest.raises(blz.BlzError) as e:
                call()
            assert e.value.code == blz.EINVAL and "no matrix" in str(e.value)
        M = blz.Matrix.load(os.path.join(GOLDEN, "quirks40x30.mtx"), P61)
        ctx.set_matrix(M, False)
        with pytest.raises(blz.BlzError):
            ctx.spmv(False, blz.V, blz.V)                # src == dst
        with pytest.raises(blz.BlzError):
            blz.check(blz.lib().blz_set_small(ctx.h, 9, blz.ptr(np.zeros(16, np.uint64))))
        with pytest.raises(blz.BlzError):
            ctx.set_matrix(M, False, 2, 2)               # rank out of range
        ctx.set_matrix(M, True)                          # re-loading another orientation is fine
        assert ctx.rows(blz.V) == 30 and ctx.rows(blz.TMP) == 40


@pytest.mark.parametrize("capw,rpg", [(None, None), ("64", "8"), ("128", "1"), ("4096", "3")])
@pytest.mark.parametrize("n,p,kind", [(8, P61, "packed"), (4, 2147483647, "packed"), (16, P61, "ones"), (1, 65537, "array"),
                                      (2, P61, "array"), (8, (1 << 62) - 57, "array"), (32, 1073741789, "packed"),
                                      (64, P61, "ones")])
def test_staged_matrix_stream_against_oracle(monkeypatch, n, p, kind, capw, rpg):
    """k_spmv_staged: the tile's piece of col_idx (and val) is copied into LDS by LDS-DMA one tile ahead, tiles are
    dealt to the XCDs in nnz-balanced ranges, row tails are predicated batches.  Every value mode (all ones, palette,
    separate array), window sizes that make rows overflow the staged window (they fall back to reading the stream from
    global memory) and tile heights from 1 row per lane group up, both orientations, plain and fused-with-block_dot
    forms, against the oracle; and BLZ_NO_STAGE=1 (the round-1 kernels) gives the same words."""
    monkeypatch.setenv("BLZ_STAGE_ALWAYS", "1")         # rows of ~19 entries would otherwise keep the round-1 kernels
    if capw:
        monkeypatch.setenv("BLZ_STAGE_CAPW", capw)
        monkeypatch.setenv("BLZ_STAGE_RPG", rpg)
    rng = np.random.default_rng(n * 1000 + len(kind))
    nr, nc, nz = 9000, 9500, 170000
    ii, jj = rng.integers(0, nr, nz), rng.integers(0, nc, nz)
    ii[:3000] = 17                                        # one outlier row, and rows of very different lengths
    ii[3000:5000] = rng.integers(100, 140, 2000)
    if kind == "ones":
        xx = np.ones(nz, dtype=np.uint32)
    elif kind == "packed":
        xx = rng.choice(np.array([1, 2, 3, 2 ** 32 - 1, 2 ** 32 - 2], dtype=np.uint64), size=nz).astype(np.uint32)
    else:
        xx = rng.integers(1, 2 ** 32, size=nz, dtype=np.uint64).astype(np.uint32)
    xx = (xx.astype(np.uint64) % p).astype(np.uint32)
    M = blz.Matrix(nr, nc, ii, jj, xx)
    Mo = as_orc(M)
    for right in (False, True):
        want = orc.block_lanczos(Mo, n, p, right=right, stop_after=3)
        for flag in ("0", "1"):
            monkeypatch.setenv("BLZ_NO_STAGE", flag)
            with blz.Context(p, n) as ctx:
                ctx.set_matrix(M, right)
                x = (np.arange(ctx.rows(blz.TMP) * n, dtype=np.uint64) * 2654435761) % p
                ctx.set_block(blz.TMP, x)
                ctx.spmv(right, blz.TMP, blz.AV)
                assert np.array_equal(ctx.get_block(blz.AV), orc.spmv(Mo, x, right, n, p))
                ctx.init_v()
                ctx.iterate(3)
                assert np.array_equal(ctx.get_block(blz.V), want["v"]) and np.array_equal(ctx.get_block(blz.P), want["p"])


@pytest.mark.parametrize("n,p", [(8, P61), (16, P61), (4, 2147483647)])
def test_a_local_matrix_takes_the_staged_form_by_itself(n, p):
    """Round 3: rows of ~20 entries normally keep k_spmv, but when the renumbering's sample finds the gathers mostly hitting
    (a band matrix: < 0.3 lines per entry) the first product of an iteration runs the staged form.  No switch set here: the
    plan decides; the words must be the oracle's for a banded matrix in its own order and for the same matrix scrambled."""
    rng = np.random.default_rng(n)
    R = C = 40000
    per, band = 20, 400
    i = np.repeat(np.arange(R), per)
    j = (i + rng.integers(-band // 2, band // 2, size=R * per)) % C
    x = (rng.choice(np.array([1, 2, 3, 2 ** 32 - 1], dtype=np.uint64), size=R * per) % p).astype(np.uint32)
    pr, pc = rng.permutation(R), rng.permutation(C)
    for M in (blz.Matrix(R, C, i, j, x), blz.Matrix(R, C, pr[i], pc[j], x)):
        Mo = as_orc(M)
        want = orc.block_lanczos(Mo, n, p, stop_after=4)
        with blz.Context(p, n) as ctx:
            ctx.set_matrix(M, False)
            ctx.init_v()
            ctx.iterate(4)
            assert np.array_equal(ctx.get_block(blz.V), want["v"]) and np.array_equal(ctx.get_block(blz.P), want["p"])
            # and both products on their own (the reference's tmp holds the new v after an iteration, so it is not compared above)
            xin = (np.arange(ctx.rows(blz.V) * n, dtype=np.uint64) * 2654435761 + 7) % p
            ctx.set_block(blz.V, xin)
            ctx.spmv(True, blz.V, blz.TMP)
            assert np.array_equal(ctx.get_block(blz.TMP), orc.spmv(Mo, xin, True, n, p))


@pytest.mark.parametrize("capw,rpg", [(None, None), ("64", "1"), ("128", "3"), ("4096", "8")])
@pytest.mark.parametrize("n,p,kind", [(16, P61, "ones"), (16, P61, "packed"), (16, (1 << 62) - 57, "array"), (12, 2305843009213693907, "packed"),
                                      (8, P61, "packed"), (8, (1 << 62) - 57, "array"), (8, P61, "ones")])
def test_two_words_per_lane_at_n16_against_oracle(monkeypatch, n, p, kind, capw, rpg):
    """Round 3: at n = 16 (64-bit words) a block row is gathered by 8 lanes of two words (16 bytes per lane, 8 rows per
    wavefront) instead of 16 lanes of one.  Same sums per word, so the same words as the oracle and as the one-word form
    (BLZ_NO_PAIR=1): every value mode, windows that rows overflow, an outlier row, both orientations, whole iterations.
    n = 12 is padded to 16 in HBM; n = 8 (4 lanes of 16 bytes per 64-byte row) pairs the first product of an
    iteration only -- the second carries the inner products and keeps one word per lane."""
    monkeypatch.setenv("BLZ_STAGE_ALWAYS", "1")
    if capw:
        monkeypatch.setenv("BLZ_STAGE_CAPW", capw)
        monkeypatch.setenv("BLZ_STAGE_RPG", rpg)
    rng = np.random.default_rng(len(kind) * 31 + p % 97)
    nr, nc, nz = 8000, 7700, (160000 if n > 8 else 60000)    # at n = 8 the plan pairs lanes on rows of a few entries only
    ii, jj = rng.integers(0, nr, nz), rng.integers(0, nc, nz)
    ii[:3000] = 23
    ii[3000:5000] = rng.integers(200, 230, 2000)
    if kind == "ones":
        xx = np.ones(nz, dtype=np.uint32)
    elif kind == "packed":
        xx = rng.choice(np.array([1, 2, 3, 2 ** 32 - 1, 2 ** 32 - 2], dtype=np.uint64), size=nz).astype(np.uint32)
    else:
        xx = rng.integers(1, 2 ** 32, size=nz, dtype=np.uint64).astype(np.uint32)
    xx = (xx.astype(np.uint64) % p).astype(np.uint32)
    M = blz.Matrix(nr, nc, ii, jj, xx)
    Mo = as_orc(M)
    for right in (False, True):
        want = orc.block_lanczos(Mo, n, p, right=right, stop_after=3)
        for flag in ("0", "1"):
            monkeypatch.setenv("BLZ_NO_PAIR", flag)
            with blz.Context(p, n) as ctx:
                ctx.set_matrix(M, right)
                for t, src, dst in ((right, blz.TMP, blz.AV), (not right, blz.V, blz.TMP)):
                    x = (np.arange(ctx.rows(src) * n, dtype=np.uint64) * 2654435761 + 99) % p
                    ctx.set_block(src, x)
                    ctx.spmv(t, src, dst)
                    assert np.array_equal(ctx.get_block(dst), orc.spmv(Mo, x, t, n, p)), (right, flag, t)
                ctx.init_v()
                ctx.iterate(3)
                assert np.array_equal(ctx.get_block(blz.V), want["v"]) and np.array_equal(ctx.get_block(blz.P), want["p"])
            if capw is None and kind in ("ones", "array"):
                # products cut into three column pieces (the exchange pipeline's form): pieces 2 and 3 ADD to the rows piece 1 wrote
                monkeypatch.setenv("BLZ_FORCE_COMM", "1")
                monkeypatch.setenv("BLZ_AG_CHUNKS", "3")
                with blz.Context(p, n) as ctx:
                    ctx.comm_init(blz.comm_unique_id(), 0, 1)
                    ctx.set_matrix(M, right, 0, 1)
                    ctx.init_v()
                    ctx.iterate(3)
                    assert np.array_equal(ctx.get_block(blz.V), want["v"]) and np.array_equal(ctx.get_block(blz.P), want["p"])
                monkeypatch.delenv("BLZ_FORCE_COMM")
                monkeypatch.delenv("BLZ_AG_CHUNKS")


@pytest.mark.parametrize("capw,tr", [(None, None), ("64", "5"), ("256", "64"), ("4096", "1")])
@pytest.mark.parametrize("n,p,kind", [(16, P61, "ones"), (16, P61, "packed"), (8, P61, "packed"), (8, 2147483647, "array"),
                                      (12, (1 << 62) - 57, "array"), (32, 1073741789, "ones")])
def test_dynamic_rows_in_the_staged_stream_against_oracle(monkeypatch, n, p, kind, capw, tr):
    """k_spmv_staged<..., DYN> (round 3): the lane groups of a wavefront take the rows of a tile from a shared counter
    instead of one row each in lockstep.  Rows stay whole and owned by one group, so the words must be the oracle's: rows
    of very different lengths (empty rows, one outlier row, a band of long rows), every value mode, tiles of 1 ... 64 rows,
    windows that rows overflow (those read their entries from global memory), both orientations; and the same words with the
    form switched off.  With n <= 8 the second product of an iteration keeps the lockstep form (it carries the inner
    products), the first runs the dynamic one."""
    monkeypatch.setenv("BLZ_STAGE_ALWAYS", "1")
    if capw:
        monkeypatch.setenv("BLZ_STAGE_CAPW", capw)
        monkeypatch.setenv("BLZ_STAGE_TR", tr)
    rng = np.random.default_rng(n * 77 + len(kind))
    nr, nc, nz = 7000, 7300, 150000
    ii, jj = rng.integers(0, nr, nz), rng.integers(0, nc, nz)
    ii[:3000] = 17                                        # one outlier row
    ii[3000:6000] = rng.integers(100, 130, 3000)          # a band of rows of ~100 entries
    ii[(ii >= 2000) & (ii < 2100)] = 2100                 # a hundred empty rows in a row
    if kind == "ones":
        xx = np.ones(nz, dtype=np.uint32)
    elif kind == "packed":
        xx = rng.choice(np.array([1, 2, 3, 2 ** 32 - 1, 2 ** 32 - 2], dtype=np.uint64), size=nz).astype(np.uint32)
    else:
        xx = rng.integers(1, 2 ** 32, size=nz, dtype=np.uint64).astype(np.uint32)
    xx = (xx.astype(np.uint64) % p).astype(np.uint32)
    M = blz.Matrix(nr, nc, ii, jj, xx)
    Mo = as_orc(M)
    for right in (False, True):
        want = orc.block_lanczos(Mo, n, p, right=right, stop_after=3)
        for flag in ("1", "0"):
            monkeypatch.setenv("BLZ_STAGE_DYN", flag)
            with blz.Context(p, n) as ctx:
                ctx.set_matrix(M, right)
                for t, src, dst in ((right, blz.TMP, blz.AV), (not right, blz.V, blz.TMP)):
                    x = (np.arange(ctx.rows(src) * n, dtype=np.uint64) * 2654435761 + 12345) % p
                    ctx.set_block(src, x)
                    ctx.spmv(t, src, dst)
                    assert np.array_equal(ctx.get_block(dst), orc.spmv(Mo, x, t, n, p)), (right, flag, t)
                ctx.init_v()
                ctx.iterate(3)
                assert np.array_equal(ctx.get_block(blz.V), want["v"]) and np.array_equal(ctx.get_block(blz.P), want["p"])


@pytest.mark.parametrize("n,p,kind", [(8, P61, "packed"), (4, 2147483647, "packed"), (16, P61, "ones"), (2, (1 << 62) - 57, "array"),
                                      (1, 65537, "packed"), (32, 1073741789, "ones")])
def test_panel_of_dense_block_rows_against_oracle(monkeypatch, n, p, kind):
    """k_spmv_panel: on a matrix with heavy-tailed column degrees the solver numbers the densest columns (rows, for the
    transposed product) first and the SpMV keeps that many block rows of its operand in LDS; entries outside the
    panel are gathered as usual; rows are dealt to the XCDs in nnz-balanced ranges.  Both orientations of a structured
    matrix and of its transpose (so that each product meets a panel), a full and a tiny panel, every value mode, the
    plain and the fused-with-block_dot forms, against the oracle; BLZ_NO_PANEL=1 gives the same words."""
    S = blz.Matrix.synth_structured(7000, 8000, 140000, 0x4E465331 + n, p, pattern=(kind == "ones"), hot_pct=45,
                                    band_pct=25, band=512)
    ii, jj, xx = S.i.copy(), S.j.copy(), S.x.copy()
    if kind == "array":
        rng = np.random.default_rng(n)
        xx = (rng.integers(1, 2 ** 32, size=len(xx), dtype=np.uint64) % p).astype(np.uint32)
    ii[:2500] = 5                                       # an outlier row as well
    for transposed in (False, True):
        M = blz.Matrix(8000, 7000, jj, ii, xx) if transposed else blz.Matrix(7000, 8000, ii, jj, xx)
        Mo = as_orc(M)
        for right in (False, True):
            want = orc.block_lanczos(Mo, n, p, right=right, stop_after=3)
            for env in ({"BLZ_NO_PANEL": "0"}, {"BLZ_NO_PANEL": "0", "BLZ_PANEL_ROWS": "37", "BLZ_PANEL_MIN_PCT": "1"},
                        {"BLZ_NO_PANEL": "0", "BLZ_PANEL_STRIPES": "1"}, {"BLZ_NO_PANEL": "1"}):
                for k_ in ("BLZ_PANEL_ROWS", "BLZ_PANEL_MIN_PCT", "BLZ_PANEL_STRIPES"):
                    monkeypatch.delenv(k_, raising=False)
                for k_, v_ in env.items():
                    monkeypatch.setenv(k_, v_)
                with blz.Context(p, n) as ctx:
                    ctx.set_matrix(M, right)
                    rows_a, share_a = ctx.panel_rows(False)
                    rows_b, share_b = ctx.panel_rows(True)
                    if env["BLZ_NO_PANEL"] == "1":
                        assert rows_a == rows_b == 0
                    else:
                        assert (rows_b if transposed else rows_a) > 0 and max(share_a, share_b) > 0.01
                    x = (np.arange(ctx.rows(blz.TMP) * n, dtype=np.uint64) * 2654435761) % p
                    ctx.set_block(blz.TMP, x)
                    ctx.spmv(right, blz.TMP, blz.AV)
                    assert np.array_equal(ctx.get_block(blz.AV), orc.spmv(Mo, x, right, n, p))
                    y = (np.arange(ctx.rows(blz.V) * n, dtype=np.uint64) * 40503 + 7) % p
                    ctx.set_block(blz.V, y)
                    ctx.spmv(not right, blz.V, blz.TMP)
                    assert np.array_equal(ctx.get_block(blz.TMP), orc.spmv(Mo, y, not right, n, p))
                    ctx.init_v()
                    ctx.iterate(3)
                    assert np.array_equal(ctx.get_block(blz.V), want["v"]) and np.array_equal(ctx.get_block(blz.P), want["p"])


@pytest.mark.parametrize("n", [8, 16])
@pytest.mark.parametrize("rows", [1, 15, 16, 17, 4099, 70001])
def test_block_update_on_the_matrix_cores_against_oracle(monkeypatch, n, rows):
    """orthogonalize() for p = 2^61-1 at n = 8 / 16 runs on v_mfma_i32_16x16x64_i8 (exact integer contractions of the
    base-256 digits of the block rows against signed digits of the n x n coefficients, folded mod 2^61-1): row counts
    around the 16-row tile, several iterations so that d takes different patterns, against the oracle; BLZ_NO_MFMA=1
    (the vector-ALU kernels) gives the same words; and one update with hand-made extreme operands (all words p-1)."""
    p = P61
    cols = max(rows // 2, 3)
    rng = np.random.default_rng(rows * n)
    nz = max(4 * rows, 8)
    M = blz.Matrix(rows, cols, rng.integers(0, rows, nz), rng.integers(0, cols, nz),
                   rng.choice(np.array([1, 2, 3, 2 ** 32 - 1], dtype=np.uint64), size=nz).astype(np.uint32))
    Mo = as_orc(M)
    want = orc.block_lanczos(Mo, n, p, stop_after=4)
    monkeypatch.setenv("BLZ_MFMA_MIN_ROWS", "0")            # (by default small blocks stay on the vector ALU: launch cost)
    # flag "8": matrix cores with the LDS-staged row loads at n = 8 as well (the default there loads fragments directly)
    for flag in ("0", "1", "8") if n == 8 else ("0", "1"):
        monkeypatch.setenv("BLZ_NO_MFMA", "1" if flag == "1" else "0")
        monkeypatch.setenv("BLZ_MFMA_STAGE8", "1" if flag == "8" else "0")
        with blz.Context(p, n) as ctx:
            ctx.set_matrix(M, False)
            ctx.init_v()
            ctx.iterate(4)
            assert ctx.iterations == want["iterations"]
            assert np.array_equal(ctx.get_block(blz.V), want["v"]) and np.array_equal(ctx.get_block(blz.P), want["p"])
        with blz.Context(p, n) as ctx:                      # (a fresh context: the solve above may have raised the stop flag)
            ctx.set_matrix(M, False)
            # extreme operands: every word of v, Av, p at p-1, coefficients from a semi_inverse of a dense symmetric matrix
            big = np.full(rows * n, p - 1, dtype=np.uint64)
            A = rng.integers(0, p, size=(n, n), dtype=np.uint64)
            A = ((A.astype(object) + A.T.astype(object)) % p).astype(np.uint64).reshape(-1)
            B = ((A.astype(object) * 3 + 1) % p).astype(np.uint64)
            ctx.set_block(blz.V, big)
            ctx.set_block(blz.AV, big)
            ctx.set_block(blz.P, big)
            ctx.set_small(blz.VTAV, A)
            ctx.set_small(blz.VTAAV, B)
            npiv, winv, d = ctx.semi_inverse()
            ctx.orthogonalize()
            tmp_o, p_o = orc.orthogonalize(big, big.copy(), d, A, B, winv, rows, big, n, p)
            assert np.array_equal(ctx.get_block(blz.V), tmp_o) and np.array_equal(ctx.get_block(blz.P), p_o)


@pytest.mark.parametrize("n", [8, 16])
@pytest.mark.parametrize("rows", [4096, 4097, 70001, 600000])
def test_inner_products_on_the_matrix_cores_against_oracle(monkeypatch, n, rows):
    """block_dot_products for p = 2^61-1 at n = 8 / 16 on v_mfma_i32_16x16x64_i8 (rows as the K dimension, both operands
    biased, the bias terms turned into column sums, i32 accumulators folded mod p every 64 tiles): random blocks and the
    extreme block (all words p-1), row counts around the 64-row tile and past the first fold, against the oracle;
    BLZ_NO_MFMA=1 gives the same words."""
    p = P61
    rng = np.random.default_rng(rows + n)
    M = blz.Matrix(rows, 8, rng.integers(0, rows, 64), rng.integers(0, 8, 64), np.ones(64, dtype=np.uint32))
    for flag in ("0", "1"):
        monkeypatch.setenv("BLZ_NO_MFMA", flag)
        with blz.Context(p, n) as ctx:
            ctx.set_matrix(M, False)
            for kind in ("random", "max"):
                if kind == "random":
                    v = rng.integers(0, p, rows * n, dtype=np.uint64)
                    a = rng.integers(0, p, rows * n, dtype=np.uint64)
                else:
                    v = np.full(rows * n, p - 1, dtype=np.uint64)
                    a = np.full(rows * n, p - 1, dtype=np.uint64)
                ctx.set_block(blz.V, v)
                ctx.set_block(blz.AV, a)
                got = ctx.block_dot()
                want = orc.block_dot(rows, a, v, n, p, omp_threads=8)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (kind, flag)
