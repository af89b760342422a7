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

        its = 0
        while its < max_iters:
            full = gather(blz.V)
            for c in ctxs:
                c.set_block(blz.V, full)                       # = all-gather(v)
                c.spmv(not right, blz.V, blz.TMP)
            full = gather(blz.TMP)
            for c in ctxs:
                c.set_block(blz.TMP, full)                     # = all-gather(tmp)
                c.spmv(right, blz.TMP, blz.AV)
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
