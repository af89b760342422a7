"""Seeded random systems through the whole GPU path against the oracle: shapes from 1 x 1 to a few hundred rows, any
density (empty rows and columns, duplicate entries, rows longer than the outlier threshold, rows that hold every
column), every class of prime, widths 1..12 and both orientations.  Each case runs to termination, so the whole
trajectory -- including the iteration where the n x n system loses rank and the last, partial one -- must agree word
for word.  Integer path: equality, no tolerance."""
import numpy as np
import pytest

import blz
import oracle as orc

pytestmark = pytest.mark.gpu

PRIMES = [2, 3, 251, 65537, 1073741789, 2147483647, 4294967291, 4294967311, (1 << 61) - 1, (1 << 62) - 57]


def random_case(seed):
    rng = np.random.default_rng(seed)
    nr, nc = (int(x) for x in rng.choice([1, 2, 3, 7, 31, 64, 65, 150, 400], size=2))
    style = seed % 5
    if style == 0:        # sparse
        nnz = int(rng.integers(0, 3 * max(nr, nc) + 1))
        i, j = rng.integers(0, nr, nnz), rng.integers(0, nc, nnz)
    elif style == 1:      # dense-ish
        nnz = int(nr * nc * rng.uniform(0.3, 1.0))
        i, j = rng.integers(0, nr, nnz), rng.integers(0, nc, nnz)
    elif style == 2:      # one row and one column hold almost everything (outlier rows in both products)
        nnz = int(rng.integers(1, 6 * max(nr, nc) + 1))
        i, j = rng.integers(0, nr, nnz), rng.integers(0, nc, nnz)
        i[: nnz // 2] = rng.integers(0, nr)
        j[nnz // 3: 2 * nnz // 3] = rng.integers(0, nc)
    elif style == 3:      # low rank: few distinct rows repeated (large kernels, early termination)
        base = rng.integers(0, nc, size=(3, 4))
        i = np.repeat(np.arange(nr), 4)
        j = base[rng.integers(0, 3, nr)].reshape(-1)
    else:                 # banded with duplicates
        i = np.repeat(np.arange(nr), 3)
        j = (i * nc // max(nr, 1) + rng.integers(0, 2, i.size)) % nc
    p = PRIMES[int(rng.integers(0, len(PRIMES)))]
    x = rng.integers(0, 2 ** 32, size=len(i), dtype=np.uint64) % p
    n = int(rng.integers(1, 13))
    return blz.Matrix(nr, nc, np.asarray(i, np.int32), np.asarray(j, np.int32), x.astype(np.uint32)), p, n, bool(seed & 1)


@pytest.mark.parametrize("seed", range(60))
def test_random_system_to_termination(seed):
    M, p, n, right = random_case(seed)
    Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
    want = orc.block_lanczos(Mo, n, p, right=right, stop_after=400)
    got = blz.solve(M, p, n, right=right, stop_after=400, batch=7)
    assert got["iterations"] == want["iterations"], (M.nrows, M.ncols, M.nnz, p, n, right)
    assert np.array_equal(got["v"], want["v"]) and np.array_equal(got["p"], want["p"])
