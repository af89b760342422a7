"""Multi-rank device code on ONE GPU.

(1) P contexts on device 0, each holding rank g's slabs exactly as on a P-GPU node, in external-exchange mode: the
    test moves slabs between them through the C ABI (get_block / set_block / get_small / set_small) where RCCL
    would, so every kernel runs on rank-local slabs in the padded layout.  Result must equal the oracle.
(2) The RCCL plumbing itself on a 1-rank communicator (BLZ_FORCE_COMM=1 makes the library issue its in-place
    ncclAllGather / ncclAllReduce even though they are no-ops): dlopen, symbol table, comm init, stream use.
RCCL refuses a communicator with two ranks on one GPU, so a true 2-rank run needs the driver's multi-GPU node.
"""
import os

import numpy as np
import pytest

import blz
import oracle as orc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P61 = (1 << 61) - 1


def sharded_solve(M, p, n, right, world, max_iters):
    ctxs = [blz.Context(p, n) for _ in range(world)]
    try:
        for g, c in enumerate(ctxs):
            c.set_matrix(M, right, g, world)
            c.set_exchange_mode(True)
            c.init_v()

        def gather(block):
            # get_block writes only the rows a rank owns (original numbering; not contiguous once the solver has
            # renumbered rows for locality), so the disjoint parts simply add up
            full = np.zeros(ctxs[0].rows(block) * n, dtype=np.uint64)
            for c in ctxs:
                full += c.get_block(block)
            return full

        def product(transpose, src, dst):
            """one product of the iteration with the exchange done by hand: all-gather of the operand, or -- for a product in
            its short-side form -- every rank multiplies its OWN slab and the partial products are summed (= reduce-scatter)"""
            if ctxs[0].short_side(transpose):
                assert all(c.short_side(transpose) for c in ctxs)
                for c in ctxs:
                    c.spmv(transpose, src, dst)
                total = np.zeros(ctxs[0].rows(dst) * n, dtype=object)
                for c in ctxs:
                    total = total + c.get_partial(transpose).astype(object)
                total = np.array(total % p, dtype=np.uint64)
                for c in ctxs:
                    c.set_block(dst, total)                    # each rank keeps its rows of the sum
            else:
                full = gather(src)
                for c in ctxs:
                    c.set_block(src, full)                     # = all-gather
                    c.spmv(transpose, src, dst)

        its = 0
        while its < max_iters:
            product(not right, blz.V, blz.TMP)
            product(right, blz.TMP, blz.AV)
            parts = [c.block_dot() for c in ctxs]              # rank-local products
            a = np.zeros(n * n, dtype=object)
            b = np.zeros(n * n, dtype=object)
            for (x, y) in parts:
                a, b = a + x.astype(object), b + y.astype(object)
            a, b = np.array(a % p, dtype=np.uint64), np.array(b % p, dtype=np.uint64)   # = all-reduce, then mod p
            npivs = []
            for c in ctxs:
                c.set_small(blz.VTAV, a)
                c.set_small(blz.VTAAV, b)
                npivs.append(c.semi_inverse()[0])
            assert len(set(npivs)) == 1                        # replicated decision
            if npivs[0] == 0:
                break
            for c in ctxs:
                c.orthogonalize()
            its += 1
        return dict(v=gather(blz.V), p=gather(blz.P), tmp=gather(blz.TMP), iterations=its)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("world,chunks", [(2, 1), (3, 1), (8, 1), (2, 4), (3, 3), (8, 2)])
@pytest.mark.parametrize("name,p,n,right,iters", [
    ("rand300x200", P61, 8, False, 10 ** 9), ("rand300x200", 65537, 4, True, 10 ** 9),
    ("wide120x260", 2147483647, 4, True, 10 ** 9), ("rand3000x2000", P61, 8, False, 8),
    ("rand3000x2000", 1073741789, 16, True, 5)])
def test_rank_local_kernels_with_emulated_exchange(monkeypatch, name, p, n, right, iters, world, chunks):
    monkeypatch.setenv("BLZ_AG_CHUNKS", str(chunks))     # pieces per all-gather = column pieces of every product
    path = os.path.join(GOLDEN, name + ".mtx")
    M, Mo = blz.Matrix.load(path, p), orc.Matrix.load(path, p)
    want = orc.block_lanczos(Mo, n, p, right=right, stop_after=iters if iters < 10 ** 8 else -1)
    got = sharded_solve(M, p, n, right, world, iters)
    assert got["iterations"] == want["iterations"]
    assert np.array_equal(got["v"], want["v"]) and np.array_equal(got["p"], want["p"])
    if iters > 10 ** 8:
        assert np.array_equal(got["tmp"], want["tmp"])


def test_every_row_has_exactly_one_owner():
    M = blz.Matrix.load(os.path.join(GOLDEN, "rand300x200.mtx"), P61)
    ctxs = [blz.Context(P61, 4) for _ in range(3)]
    try:
        for g, c in enumerate(ctxs):
            c.set_matrix(M, False, g, 3)
        for block, rows in ((blz.V, 300), (blz.TMP, 200)):
            owners = [ctxs[0].owner_of_row(block, r) for r in range(rows)]
            assert set(owners) <= {0, 1, 2} and all(ctxs[1].owner_of_row(block, r) == owners[r] for r in range(rows))
            for g, c in enumerate(ctxs):
                assert owners.count(g) == c.local_rows(block)[1]
    finally:
        for c in ctxs:
            c.close()


def test_multi_rank_context_without_communicator_fails_loudly():
    M = blz.Matrix.load(os.path.join(GOLDEN, "rand300x200.mtx"), P61)
    with blz.Context(P61, 4) as c:
        c.set_matrix(M, False, 1, 2)
        c.init_v()
        with pytest.raises(blz.BlzError) as e:
            c.iterate(1)
        assert e.value.code == blz.ECOMM


@pytest.mark.parametrize("chunks", [1, 4])
def test_rccl_plumbing_on_one_rank(monkeypatch, chunks):
    monkeypatch.setenv("BLZ_FORCE_COMM", "1")
    monkeypatch.setenv("BLZ_AG_CHUNKS", str(chunks))
    p, n = P61, 8
    path = os.path.join(GOLDEN, "rand3000x2000.mtx")
    M, Mo = blz.Matrix.load(path, p), orc.Matrix.load(path, p)
    uid = blz.comm_unique_id()
    assert len(uid) == 128
    with blz.Context(p, n) as c:
        c.comm_init(uid, 0, 1)
        c.set_matrix(M, False, 0, 1)
        c.init_v()
        c.profile(True)
        c.iterate(6)
        prof = c.profile_read()
        assert prof["allgather_v"]["launches"] == 6 * chunks and prof["allgather_tmp"]["launches"] == 6 * chunks
        assert prof["spmv1"]["launches"] == 6 * chunks and prof["spmv2"]["launches"] == 6 * chunks
        assert prof["allreduce"]["launches"] == 6
        want = orc.block_lanczos(Mo, n, p, stop_after=6)
        assert np.array_equal(c.get_block(blz.V), want["v"])
        # and a whole solve through the two-stream pipeline (events are re-used every iteration)
        c.profile(False)
        c.init_v()
        while not c.iterate(64)[1]:
            pass
        full = orc.block_lanczos(Mo, n, p)
        assert c.iterations == full["iterations"]
        assert np.array_equal(c.get_block(blz.V), full["v"]) and np.array_equal(c.get_block(blz.TMP), full["tmp"])


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("shape,right,forced", [((9000, 400), False, None), ((9000, 400), True, None), ((400, 9000), False, None),
                                                ((3000, 2000), False, "1")])
def test_short_side_exchange_with_emulated_ranks(monkeypatch, shape, right, world, forced):
    """Tall and wide matrices on several ranks: the product whose operand lives on the long side runs in its short-side form
    (transpose of the rank's own rows times its own slab, partial products summed over the ranks instead of an all-gather of
    the long block); the other product keeps the all-gather of the short block.  Emulated exchange on one GPU; the trajectory
    must be the oracle's.  BLZ_SHORT_SIDE=1 forces the form on both products of a squarish matrix."""
    if forced:
        monkeypatch.setenv("BLZ_SHORT_SIDE", forced)
    p, n = P61, 8
    M = blz.Matrix.synth(shape[0], shape[1], 12 * max(shape), 0x54414C4C, p)
    Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
    want = orc.block_lanczos(Mo, n, p, right=right, stop_after=5)
    probe = blz.Context(p, n)
    probe.set_matrix(M, right, 0, world)
    flags = (probe.short_side(False), probe.short_side(True))
    probe.close()
    if forced:
        assert flags == (True, True)
    else:
        assert flags.count(True) == 1                         # exactly the product that would gather the long block
    got = sharded_solve(M, p, n, right, world, 5)
    assert got["iterations"] == want["iterations"]
    assert np.array_equal(got["v"], want["v"]) and np.array_equal(got["p"], want["p"])


@pytest.mark.parametrize("shape,right", [((9000, 400), False), ((9000, 400), True), ((3000, 2000), False)])
def test_short_side_exchange_through_rccl_on_one_rank(monkeypatch, shape, right):
    """The real call path (ncclReduceScatter on the compute stream, then the mod-p pass) on a 1-rank communicator."""
    monkeypatch.setenv("BLZ_FORCE_COMM", "1")
    monkeypatch.setenv("BLZ_SHORT_SIDE", "1")
    p, n = P61, 8
    M = blz.Matrix.synth(shape[0], shape[1], 12 * max(shape), 0x54414C4C, p)
    Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
    want = orc.block_lanczos(Mo, n, p, right=right, stop_after=7)
    with blz.Context(p, n) as c:
        c.comm_init(blz.comm_unique_id(), 0, 1)
        c.set_matrix(M, right, 0, 1)
        assert c.short_side(False) and c.short_side(True)
        c.init_v()
        c.profile(True)
        c.iterate(7)
        prof = c.profile_read()
        assert prof["reduce_scatter"]["launches"] == 14 and prof["allgather_v"]["launches"] == 0
        assert np.array_equal(c.get_block(blz.V), want["v"]) and np.array_equal(c.get_block(blz.P), want["p"])


@pytest.mark.parametrize("shape,right", [((1500, 900), False), ((900, 1500), True)])
def test_short_side_batch_past_the_stop_leaves_tmp_alone(monkeypatch, shape, right):
    """A batch always runs past the stop (the CLI doubles its batch).  The reduce-scatter of a short-side product is
    enqueued by the host whatever the flag says and, past the stop, sums STALE partial products -- with both products in
    that form the stale buffer even belongs to the other product.  The collective therefore lands in a buffer of its own
    and the stop-aware mod-p pass is what writes the slab: TMP after the batch must be the oracle's
    M^T v, word for word residues, and blz_final_check must say what the reference's final_check says."""
    monkeypatch.setenv("BLZ_FORCE_COMM", "1")
    monkeypatch.setenv("BLZ_SHORT_SIDE", "1")
    p, n = P61, 8
    M = blz.Matrix.synth(shape[0], shape[1], 10 * max(shape), 0x53544F50, p)
    Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
    full = orc.block_lanczos(Mo, n, p, right=right)
    with blz.Context(p, n) as c:
        c.comm_init(blz.comm_unique_id(), 0, 1)
        c.set_matrix(M, right, 0, 1)
        assert c.short_side(False) and c.short_side(True)
        c.init_v()
        done, stopped, _ = c.iterate(full["iterations"] + 37)
        assert stopped and done == full["iterations"]
        tmp = c.get_block(blz.TMP)
        assert (tmp < p).all()
        assert np.array_equal(tmp, full["tmp"]) and np.array_equal(c.get_block(blz.V), full["v"])
        assert (c.get_small(blz.VTAV) < p).all() and (c.get_small(blz.VTAAV) < p).all()
        v_nonzero, vtm_zero = c.final_check()
        assert v_nonzero == bool(full["v"].any()) and vtm_zero == (not full["tmp"].any())
