"""The multi-GPU schedule of libblz_hip.so (enqueue_iteration in csrc/blz_api.hip), restated for the test-suite
with a pluggable exchange and the CPU oracle as the compute backend.

It consumes EXACTLY the data layout the product uses -- blz_shard_matrix(): nnz-balanced row slabs of M and M^T,
column indices rewritten to the rank-major padded layout, equal-sized slabs for an in-place all-gather -- and runs
the same sequence of steps:

    all-gather(v) -> tmp_g = B[C_g,:] v -> all-gather(tmp) -> Av_g = B^T[R_g,:] tmp
    -> local v^T Av, Av^T Av -> all-reduce(sum, u64) then mod p -> semi_inverse (replicated) -> local orthogonalize

`exchange` provides allgather(padded_slab) -> concatenated array and allreduce_sum(int array) -> array.
Used by tests/test_sharded_gloo.py (torch.distributed gloo, world_size 2, CPU only).  Test infrastructure only.
"""
import numpy as np

import blz
import oracle as orc


def slab_as_coo(slab):
    rows = slab["rows"]
    ri = np.repeat(np.arange(rows, dtype=np.int32), np.diff(slab["row_ptr"].astype(np.int64)))
    return orc.Matrix(rows, slab["cols"], ri, slab["col_idx"], slab["val"])


def run_rank(M, prime, n, right, rank, world, exchange, max_iters=10 ** 9, chunks=1):
    sh = blz.shard_matrix(M, right, rank, world, chunks)
    b0, b1 = sh["bounds"]
    s0, s1 = sh["stride"]
    first0, cnt0 = b0[rank], b0[rank + 1] - b0[rank]
    cnt1 = b1[rank + 1] - b1[rank]
    # SpMV 1 uses M^T for a left kernel and M for a right kernel (sequential/lanczos_modp.c:635); SpMV 2 the other
    A1 = slab_as_coo(sh["slabs"][0 if right else 1])
    A2 = slab_as_coo(sh["slabs"][1 if right else 0])
    assert A1.nrows == cnt1 and A2.nrows == cnt0
    nrows_v = M.ncols if right else M.nrows
    v = blz.rng_fill(nrows_v * n, prime)[first0 * n:(first0 + cnt0) * n].copy()
    pb = np.zeros(cnt0 * n, dtype=np.uint64)

    def padded(x, stride):
        out = np.zeros(stride * n, dtype=np.uint64)
        out[:x.size] = x
        return out

    def gathered(x, stride):
        """all-gather of the padded slabs, laid out as the product wants it: [piece k][rank g][stride/chunks rows]"""
        allg = exchange.allgather(padded(x, stride))                      # rank-major: [g][stride]
        return allg.reshape(world, chunks, stride // chunks, n).transpose(1, 0, 2, 3).reshape(-1).copy()

    its = 0
    tmp = np.zeros(cnt1 * n, dtype=np.uint64)
    while its < max_iters:
        vg = gathered(v, s0)
        tmp = orc.spmv(A1, vg, False, n, prime)
        tg = gathered(tmp, s1)
        Av = orc.spmv(A2, tg, False, n, prime)
        a, b = orc.block_dot(cnt0, Av, v, n, prime)
        tot = exchange.allreduce_sum(np.concatenate([a, b]))
        vtAv, vtAAv = tot[:n * n] % prime, tot[n * n:] % prime
        npiv, winv, d = orc.semi_inverse(vtAv, n, prime)
        if npiv == 0:
            break
        v, pb = orc.orthogonalize(v, pb, d, vtAv, vtAAv, winv, cnt0, Av, n, prime)
        its += 1
    return dict(v=v, p=pb, tmp=tmp, first=first0, count=cnt0, iterations=its, bounds=sh["bounds"], stride=sh["stride"])


def run_rank_prepared(P, prime, n, rank, exchange, max_iters=10 ** 9, short=(False, False)):
    """The same schedule on a PREPARED matrix (blz_prepare: made once by rank 0, mapped from its cache file by the others),
    with the short-side form for the products flagged in `short` (index = product t: 0 = M x, 1 = M^T x): the rank
    multiplies the transpose of its own rows of the other orientation by its OWN slab, the full-length partial products
    are summed over the ranks (`exchange.reduce_scatter`) and reduced mod p.  The prepared matrix must have been made
    without renumbering (reorder = 0) and in one piece, so that the blocks here are in the file's numbering."""
    right, world, chunks, b0, b1, (s0, s1) = P.layout()
    assert chunks == 1
    bounds, stride = (b0, b1), (s0, s1)
    first0, cnt0 = b0[rank], b0[rank + 1] - b0[rank]
    cnt1 = b1[rank + 1] - b1[rank]
    t1, t2 = (0, 1) if right else (1, 0)          # product of SpMV 1 / SpMV 2 (sequential/lanczos_modp.c:635-636)
    nrows_v = sum(b0[g + 1] - b0[g] for g in range(world))
    v = blz.rng_fill(nrows_v * n, prime)[first0 * n:(first0 + cnt0) * n].copy()
    pb = np.zeros(cnt0 * n, dtype=np.uint64)

    def padded(x, st):
        out = np.zeros(st * n, dtype=np.uint64)
        out[:x.size] = x
        return out

    def product(t, x, side_in, side_out, cnt_out):
        if short[t]:
            A = slab_as_coo(P.slab_short(rank, t))            # rows: padded rank-major numbering of the output side
            part = orc.spmv(A, x, False, n, prime) if A.ncols > 0 else np.zeros(A.nrows * n, np.uint64)
            mine = exchange.reduce_scatter(part, stride[side_out] * n)     # u64 sums of residues
            return (mine % np.uint64(prime))[:cnt_out * n]
        A = slab_as_coo(P.slab(rank, t))
        xg = exchange.allgather(padded(x, stride[side_in]))
        return orc.spmv(A, xg, False, n, prime)

    its = 0
    tmp = np.zeros(cnt1 * n, dtype=np.uint64)
    while its < max_iters:
        tmp = product(t1, v, 0, 1, cnt1)
        Av = product(t2, tmp, 1, 0, cnt0)
        a, b = orc.block_dot(cnt0, Av, v, n, prime)
        tot = exchange.allreduce_sum(np.concatenate([a, b]))
        vtAv, vtAAv = tot[:n * n] % prime, tot[n * n:] % prime
        npiv, winv, d = orc.semi_inverse(vtAv, n, prime)
        if npiv == 0:
            break
        v, pb = orc.orthogonalize(v, pb, d, vtAv, vtAAv, winv, cnt0, Av, n, prime)
        its += 1
    return dict(v=v, p=pb, tmp=tmp, first=first0, count=cnt0, iterations=its, bounds=[b0, b1], stride=[s0, s1])
