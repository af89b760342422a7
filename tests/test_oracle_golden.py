"""Pin the CPU oracle (oracle/blz_oracle.c) to the reference.

Every expectation here was produced by the reference's own code
(tests/golden/make_golden.py through oracle/_ref, built from
/root/reference/sequential/*.c).  Bit-exact comparisons throughout: the path
is integer arithmetic mod p.
"""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRAJ = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))
KERN = sorted(glob.glob(os.path.join(GOLDEN, "kern_*.npz")))
MATRIX_OF = {"tref": "trefethen20", "r300": "rand300x200", "wide": "wide120x260",
             "quirks": "quirks40x30", "r3000": "rand3000x2000"}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def matrix_for(path, prime):
    key = os.path.basename(path).split("_")[1]
    return orc.Matrix.load(os.path.join(GOLDEN, MATRIX_OF[key] + ".mtx"), prime)


def test_rng_and_invmod():
    g = json.load(open(os.path.join(GOLDEN, "rng.json")))
    assert orc.rng_draws(len(g["draws"])) == g["draws"]
    # SURVEY section 4 lists these two values (in printf-argument order, i.e. reversed)
    assert g["draws"][:2] == [3611617879039042869, 7395509291022964594]
    for a, p, inv in g["invmod"]:
        assert orc.invmod(a, p) == inv
        assert (a * inv) % p == 1


def test_semi_inverse_matches_reference():
    g = np.load(os.path.join(GOLDEN, "semi_inverse.npz"))
    keys = sorted(k[:-2] for k in g.files if k.endswith("_M"))
    assert keys
    for key in keys:
        n = int(key.split("_")[0][1:])
        p = int(key.split("_")[1][1:])
        for M, winv, d, npiv in zip(g[key + "_M"], g[key + "_winv"], g[key + "_d"], g[key + "_npiv"]):
            got_npiv, got_winv, got_d = orc.semi_inverse(M, n, p)
            assert got_npiv == npiv
            assert np.array_equal(got_winv, winv) and np.array_equal(got_d, d)
    # KAT: semi_inverse([[0,1],[1,5]]) mod 65537
    npiv, winv, d = orc.semi_inverse([0, 1, 1, 5], 2, 65537)
    assert (npiv, list(winv), list(d)) == (2, [65532, 1, 1, 0], [1, 1])


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[5:-4] for p in TRAJ])
def test_loader_matches_reference(path):
    g = np.load(path)
    M = matrix_for(path, int(g["prime"]))
    assert np.array_equal(M.i, g["coo_i"]) and np.array_equal(M.j, g["coo_j"])
    assert np.array_equal(M.x, g["coo_x"])  # includes the "%d into u32, then % p" quirk on negatives


@pytest.mark.parametrize("path", TRAJ, ids=[os.path.basename(p)[5:-4] for p in TRAJ])
def test_trajectory_matches_reference(path):
    g = np.load(path)
    p, n, right, stop = int(g["prime"]), int(g["n"]), bool(g["right"]), int(g["stop_after"])
    M = matrix_for(path, p)
    assert np.array_equal(orc.init_v(int(g["nrows"]), n, p), g["v0"])
    recs = []
    res = orc.block_lanczos(M, n, p, right=right, stop_after=stop, trace=recs.append)
    assert res["iterations"] == int(g["iterations"])
    assert np.array_equal(res["v"], g["final_v"])
    assert np.array_equal(res["tmp"], g["final_tmp"])
    assert len(recs) == len(g["npiv"])
    for k, r in enumerate(recs):
        assert r["npiv"] == g["npiv"][k]
        for name in ("vtAv", "vtAAv", "winv", "d"):
            assert np.array_equal(r[name], g[name][k]), (name, k)
        assert sha(r["v"]) == str(g["vhash"][k])


@pytest.mark.parametrize("path", KERN, ids=[os.path.basename(p)[5:-4] for p in KERN])
def test_kernels_match_reference(path):
    g = np.load(path)
    p, n, right = int(g["prime"]), int(g["n"]), bool(g["right"])
    M = matrix_for(path, p)
    nrows = M.ncols if right else M.nrows
    its = sorted({k.split("_")[0] for k in g.files if k.startswith("it")})
    assert its
    for it in its:
        v, tmp, Av, pb = (g[f"{it}_{k}"] for k in ("v", "tmp", "Av", "p"))
        # SpMV both orientations (sequential/lanczos_modp.c:635-636)
        assert np.array_equal(orc.spmv(M, v, not right, n, p), tmp)
        assert np.array_equal(orc.spmv(M, tmp, right, n, p), Av)
        assert np.array_equal(orc.spmv_omp(M, v, not right, n, p, 3), tmp)
        a, b = orc.block_dot(nrows, Av, v, n, p)
        assert np.array_equal(a, g[f"{it}_vtAv"]) and np.array_equal(b, g[f"{it}_vtAAv"])
        a, b = orc.block_dot(nrows, Av, v, n, p, omp_threads=3)
        assert np.array_equal(a, g[f"{it}_vtAv"]) and np.array_equal(b, g[f"{it}_vtAAv"])
        npiv, winv, d = orc.semi_inverse(g[f"{it}_vtAv"], n, p)
        assert np.array_equal(winv, g[f"{it}_winv"]) and np.array_equal(d, g[f"{it}_d"])
        for thr in (None, 2):
            vn, pn = orc.orthogonalize(v, pb, d, g[f"{it}_vtAv"], g[f"{it}_vtAAv"], winv, nrows, Av, n, p,
                                       omp_threads=thr)
            assert np.array_equal(vn, g[f"{it}_vnext"]) and np.array_equal(pn, g[f"{it}_pnext"])


def test_cli_outputs_match_reference(tmp_path):
    """End to end: the oracle's writer reproduces the reference binary's output file byte for byte,
    and the widened checker agrees with checker_modp's exit code."""
    cli = json.load(open(os.path.join(GOLDEN, "cli.json")))
    v = cli.pop("_validation")
    assert v == dict(missing_prime=0, out_and_stop=0, unknown_opt=1, prime_cap=1)
    for tag, c in cli.items():
        mpath = os.path.join(GOLDEN, c["matrix"] + ".mtx")
        M = orc.Matrix.load(mpath, c["prime"])
        res = orc.block_lanczos(M, c["n"], c["prime"], right=c["right"])
        assert res["iterations"] == c["iterations"]
        nrows = M.ncols if c["right"] else M.nrows
        ncols = M.nrows if c["right"] else M.ncols
        out = str(tmp_path / (tag + ".mtx"))
        orc.save_block(out, nrows, c["n"], res["v"])
        assert hashlib.sha256(open(out, "rb").read()).hexdigest() == c["out_sha256"]
        rc, _ = orc.check_kernel(mpath, out, c["prime"], c["right"])
        assert (rc != 0) == (c["checker_exit"] != 0)
        fc = orc.final_check(nrows, ncols, c["n"], res["v"], res["tmp"])
        want = ["- OK:    v != 0" if fc & 1 else "- KO:    v == 0",
                "- OK: vt*M == 0" if fc & 2 else "- KO: vt*M != 0"]
        assert [ln for ln in c["lines"] if ln.startswith(("- OK", "- KO"))] == want


def test_trefethen_config1():
    """BASELINE config[0]: Trefethen_20 as `general`, p=65537, n=1 -> 20 iterations, zero vector."""
    M = orc.Matrix.load(os.path.join(GOLDEN, "trefethen20.mtx"), 65537)
    assert (M.nrows, M.ncols, M.nnz) == (20, 20, 158)
    res = orc.block_lanczos(M, 1, 65537)
    assert res["iterations"] == 20 and not res["v"].any() and not res["tmp"].any()


def test_loader_rejects_what_reference_rejects(tmp_path):
    bad = tmp_path / "sym.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate integer symmetric\n2 2 1\n1 1 1\n")
    with pytest.raises(ValueError):
        orc.Matrix.load(str(bad), 65537)
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n1 1 1.0\n")
    with pytest.raises(ValueError):
        orc.Matrix.load(str(bad), 65537)
    bad.write_text("%%MatrixMarket matrix array integer general\n2 2\n1\n1\n1\n1\n")
    with pytest.raises(ValueError):
        orc.Matrix.load(str(bad), 65537)
