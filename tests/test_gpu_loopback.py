"""The multi-rank path of blz_iterate run with REAL multi-rank sums on ONE GPU (round 3).

RCCL refuses two ranks on one device, and no multi-GPU box has been available to any round, so until now the exchange code had
only ever run on a 1-rank communicator (where the sum of one residue is a residue and an all-gather is a copy) or as a Python
restatement.  The loopback communicator (include/blz.h: blz_loop_group) stands in for RCCL only: the contexts of one process,
one host thread each, meet in it; everything else -- rank-local slabs sized by their side, the gathered layouts, K pieces per
exchange on the second stream with their events, the fused inner products on the last piece, the all-reduce into its own landing
buffer, the short-side reduce-scatter, what a batch does past the stop -- is the code an 8-GPU job runs.  Checked against the
oracle word for word (mpi/lanczos_modp.c:967-1149 and :1209-1247 are what this replaces in the reference).
"""
import os
import threading

import numpy as np
import pytest

import blz
import oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
P61 = (1 << 61) - 1


def run_ranks(nranks, fn):
    """fn(rank) in one thread per rank; the first exception of any rank is re-raised"""
    errs = [None] * nranks

    def go(g):
        try:
            fn(g)
        except BaseException as e:          # noqa: BLE001 (re-raised below)
            errs[g] = e

    ths = [threading.Thread(target=go, args=(g,)) for g in range(nranks)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(600)
    for e in errs:
        if e is not None:
            raise e


def loopback_solve(M, p, n, right, nranks, batch=16, extra=0, stop_after=-1):
    """every rank: context, loopback communicator, its slabs, blz_iterate in batches until the (replicated) stop; returns what
    the ranks hold"""
    group = blz.LoopGroup(nranks)
    out = [None] * nranks

    def rank_main(g):
        with blz.Context(p, n) as ctx:
            ctx.comm_init_loopback(group, g)
            assert ctx.comm_info() == (nranks, g)
            ctx.set_matrix(M, right, g, nranks)
            ctx.init_v()
            stopped = False
            while not stopped:
                todo = batch if stop_after < 0 else min(batch, stop_after - ctx.iterations)
                if todo <= 0:
                    break
                stopped = ctx.iterate(todo)[1]
            if extra:
                assert ctx.iterate(extra)[:2] == (0, True)           # a whole batch past the stop
            out[g] = dict(v=ctx.get_block(blz.V), p=ctx.get_block(blz.P), tmp=ctx.get_block(blz.TMP), its=ctx.iterations,
                          vtav=ctx.get_small(blz.VTAV), vtaav=ctx.get_small(blz.VTAAV), winv=ctx.get_small(blz.WINV),
                          check=ctx.final_check() if stop_after < 0 else None,
                          pieces=(ctx.exchange_pieces(False), ctx.exchange_pieces(True)),
                          short=(ctx.short_side(False), ctx.short_side(True)))

    try:
        run_ranks(nranks, rank_main)
    finally:
        group.close()
    return out


def together(parts, key):
    full = np.zeros_like(parts[0][key])
    for q in parts:                      # a rank returns its own rows, zeros elsewhere
        full |= q[key]
    return full


@pytest.mark.parametrize("nranks,chunks", [(2, None), (3, "1"), (8, None), (2, "4"), (3, "3")])
@pytest.mark.parametrize("name,p,n,right", [("rand3000x2000", P61, 8, False), ("rand300x200", 65537, 4, True),
                                            ("wide120x260", 2147483647, 4, True), ("rand3000x2000", 2305843009213693907, 16, False)])
def test_whole_solve_with_loopback_ranks(monkeypatch, name, p, n, right, nranks, chunks):
    """blz_iterate on 2 / 3 / 8 ranks to termination and a batch beyond: v, p, tmp word for word the oracle's, the n x n operands
    residues and identical on every rank (they are all-reduced sums of the ranks' residues), the final check the reference's."""
    if chunks:
        monkeypatch.setenv("BLZ_AG_CHUNKS", chunks)
    path = os.path.join(GOLDEN, name + ".mtx")
    M, Mo = blz.Matrix.load(path, p), orc.Matrix.load(path, p)
    want = orc.block_lanczos(Mo, n, p, right=right)
    got = loopback_solve(M, p, n, right, nranks, batch=13, extra=29)
    assert all(q["its"] == want["iterations"] for q in got)
    assert np.array_equal(together(got, "v"), want["v"]) and np.array_equal(together(got, "p"), want["p"])
    assert np.array_equal(together(got, "tmp"), want["tmp"])
    for q in got:
        assert (q["vtav"] < p).all() and (q["vtaav"] < p).all() and (q["winv"] < p).all()
        assert np.array_equal(q["vtav"], got[0]["vtav"]) and np.array_equal(q["winv"], got[0]["winv"])
        assert q["check"] == (bool(want["v"].any()), not want["tmp"].any())
    if chunks:
        assert got[0]["pieces"] == (int(chunks), int(chunks))


@pytest.mark.parametrize("nranks", [2, 3, 8])
@pytest.mark.parametrize("shape,right,forced", [((9000, 400), False, None), ((400, 9000), True, None), ((2500, 1800), False, "1")])
def test_short_side_exchange_with_loopback_ranks(monkeypatch, shape, right, nranks, forced):
    """Tall / wide matrices: the product over the long side runs in its short-side form -- every rank multiplies its own slab, the
    full-length partial products are REDUCE-SCATTERED (real sums of up to 8 ranks' residues, < 8 p < 2^64) into a landing buffer
    and brought back to residues -- whole solve plus a batch past the stop, against the oracle."""
    if forced:
        monkeypatch.setenv("BLZ_SHORT_SIDE", forced)
    p, n = P61, 8
    M = blz.Matrix.synth(shape[0], shape[1], 10 * max(shape), 0x4C4F4F50, p)
    Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
    want = orc.block_lanczos(Mo, n, p, right=right)
    got = loopback_solve(M, p, n, right, nranks, batch=32, extra=40)
    assert got[0]["short"].count(True) == (2 if forced else 1)
    assert all(q["its"] == want["iterations"] for q in got)
    assert np.array_equal(together(got, "v"), want["v"]) and np.array_equal(together(got, "p"), want["p"])
    tmp = together(got, "tmp")
    assert (tmp < p).all() and np.array_equal(tmp, want["tmp"])
    for q in got:
        assert (q["vtav"] < p).all() and (q["vtaav"] < p).all()
        assert q["check"] == (bool(want["v"].any()), not want["tmp"].any())


def test_trajectory_with_loopback_ranks_at_config_shaped_sizes():
    """A GL7d19-like shape scaled down (200 k x 205 k, 4 M entries, n = 8, p = 2^61-1) on 4 ranks, 12 iterations: large enough
    for several pieces per exchange by the plan's own choice and for every kernel's multi-workgroup form."""
    p, n = P61, 8
    M = blz.Matrix.synth(200000, 205000, 4000000, 0x474C3764, p)
    Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
    want = orc.block_lanczos(Mo, n, p, stop_after=12)
    got = loopback_solve(M, p, n, False, 4, batch=5, stop_after=12)
    assert all(q["its"] == 12 for q in got)
    assert np.array_equal(together(got, "v"), want["v"]) and np.array_equal(together(got, "p"), want["p"])


def test_the_matrix_must_be_set_with_the_rank_the_communicator_knows():
    p, n = P61, 4
    M = blz.Matrix.load(os.path.join(GOLDEN, "rand300x200.mtx"), p)
    group = blz.LoopGroup(3)
    try:
        with blz.Context(p, n) as ctx:
            ctx.comm_init_loopback(group, 1)
            for rank, nranks in ((0, 3), (1, 2), (2, 3)):
                with pytest.raises(blz.BlzError):
                    ctx.set_matrix(M, False, rank, nranks)
            ctx.set_matrix(M, False, 1, 3)
            with pytest.raises(blz.BlzError):
                ctx.comm_init_loopback(group, 2)          # one communicator per context
    finally:
        group.close()


def test_a_rank_that_never_arrives_fails_the_others_instead_of_hanging():
    """Only rank 0 of a 2-rank group iterates: its first collective must come back with BLZ_ECOMM (after the group's time-out,
    shortened here), not hang the suite."""
    p, n = P61, 4
    M = blz.Matrix.load(os.path.join(GOLDEN, "rand300x200.mtx"), p)
    os.environ["BLZ_LOOP_TIMEOUT_S"] = "3"
    group = blz.LoopGroup(2)
    try:
        with blz.Context(p, n) as ctx:
            ctx.comm_init_loopback(group, 0)
            ctx.set_matrix(M, False, 0, 2)
            ctx.init_v()
            with pytest.raises(blz.BlzError) as e:
                ctx.iterate(1)
            assert e.value.code == blz.ECOMM
    finally:
        del os.environ["BLZ_LOOP_TIMEOUT_S"]
        group.close()


@pytest.mark.parametrize("gpus", [2, 3, 8])
def test_cli_with_several_contexts_on_one_gpu_reproduces_the_reference_hashes(tmp_path, gpus):
    """lib/lanczos_modp --gpus G with BLZ_LOOPBACK=1: the G contexts of the process share this box's one GPU and meet in the
    loopback communicator; the host side is what a G-GPU run executes (one thread per context and operation, the prepared
    matrix cut G ways, the output assembled from G slabs, checkpoints collected from G contexts).  The output file must be the one
    the unmodified reference binary wrote (tests/golden/cli.json)."""
    import hashlib
    import json
    import subprocess
    exe = os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "lib", "lanczos_modp")
    cli = json.load(open(os.path.join(GOLDEN, "cli.json")))
    cli.pop("_validation")
    ran = 0
    for tag, c in cli.items():
        if c["matrix"] not in ("rand300x200", "wide120x260"):
            continue
        out = str(tmp_path / f"{tag}_g{gpus}.mtx")
        r = subprocess.run([exe, "--matrix", os.path.join(GOLDEN, c["matrix"] + ".mtx"), "--prime", str(c["prime"]),
                            "--n", str(c["n"]), "--output-file", out, "--gpus", str(gpus), "--checkpoint", "0"]
                           + (["--right"] if c["right"] else []), capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, BLZ_LOOPBACK="1"), cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-1500:]
        assert "loopback communicator" in r.stderr
        assert f"after {c['iterations']} iterations" in r.stdout, (tag, r.stdout[-400:])
        assert hashlib.sha256(open(out, "rb").read()).hexdigest() == c["out_sha256"], tag
        ran += 1
    assert ran >= 2


def test_a_checkpoint_written_by_three_contexts_is_resumed_by_two_and_by_one(tmp_path):
    """The checkpoint file holds whole blocks in the file's row order (assembled from the ranks' slabs), so the number of GPUs that
    wrote it does not bind the run that resumes it: 37 iterations on 3 contexts, then the rest on 2 contexts and on 1, against
    the uninterrupted single-GPU run -- byte-identical kernel files (openMP/lanczos_modp.c:571-676 is the reference's
    checkpoint; it has one process, so no such question)."""
    import subprocess
    exe = os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "lib", "lanczos_modp")
    base = ["--matrix", os.path.join(GOLDEN, "rand3000x2000.mtx"), "--prime", "1073741789", "--n", "8"]
    env = dict(os.environ, BLZ_LOOPBACK="1")

    def run(args, cwd):
        return subprocess.run([exe] + base + args, capture_output=True, text=True, timeout=600, env=env, cwd=str(cwd))
    full = str(tmp_path / "full.mtx")
    assert run(["--output-file", full], tmp_path).returncode == 0
    work = tmp_path / "ck"
    work.mkdir()
    r = run(["--gpus", "3", "--checkpoint", "0", "--stop-after", "37"], work)
    assert r.returncode == 0 and os.path.exists(work / "lanczos_modp.ckpt"), r.stdout + r.stderr
    assert "after 37 iterations" in r.stdout
    saved = open(work / "lanczos_modp.ckpt", "rb").read()
    for gpus in (2, 1):
        open(work / "lanczos_modp.ckpt", "wb").write(saved)
        out = str(tmp_path / f"resumed{gpus}.mtx")
        r = run(["--load-checkpoint", "--output-file", out] + (["--gpus", str(gpus)] if gpus > 1 else []), work)
        assert r.returncode == 0, r.stderr[-1500:]
        assert open(full, "rb").read() == open(out, "rb").read(), gpus


def test_the_prepared_matrix_cache_is_keyed_by_the_number_of_contexts(tmp_path):
    """--cache with --gpus G: the cache holds the partition as well, so a file made for three contexts is mapped by the next run on
    three and NOT by a run on two (which prepares afresh and keeps its own file); every run writes the reference binary's
    output (tests/golden/cli.json)."""
    import hashlib
    import json
    import shutil
    import subprocess
    exe = os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "lib", "lanczos_modp")
    c = json.load(open(os.path.join(GOLDEN, "cli.json")))["rand300x200_p65537_n4_right"]
    mpath = str(tmp_path / "m.mtx")
    shutil.copy(os.path.join(GOLDEN, c["matrix"] + ".mtx"), mpath)
    args = ["--matrix", mpath, "--prime", str(c["prime"]), "--n", str(c["n"]), "--cache"] + (["--right"] if c["right"] else [])
    env = dict(os.environ, BLZ_LOOPBACK="1")
    for k, (gpus, mapped, files) in enumerate(((3, False, 1), (3, True, 1), (2, False, 2), (2, True, 2), (3, True, 2))):
        out = str(tmp_path / f"k{k}.mtx")
        r = subprocess.run([exe] + args + ["--gpus", str(gpus), "--output-file", out], capture_output=True, text=True, timeout=600,
                           env=env, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-1500:]
        assert ("mapped from" in r.stderr) == mapped, (k, r.stderr[-800:])
        assert hashlib.sha256(open(out, "rb").read()).hexdigest() == c["out_sha256"], k
        assert len([f for f in os.listdir(tmp_path) if f.endswith(".blzcache")]) == files, k


@pytest.mark.parametrize("gpus", [2, 5])
def test_verify_mode_with_several_contexts(tmp_path, gpus):
    """--verify runs one iteration at a time and checks the reference's invariants (correctness_tests,
    sequential/lanczos_modp.c:532-557) on the n x n operands rank 0 holds after the all-reduce: with several contexts they only
    hold if every rank's products, both all-gathers and the all-reduce were right in EVERY iteration.  The output is the
    reference binary's file."""
    import hashlib
    import json
    import subprocess
    exe = os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "lib", "lanczos_modp")
    cli = json.load(open(os.path.join(GOLDEN, "cli.json")))
    for tag in ("rand300x200_p65537_n4_left", "rand300x200_p65537_n4_right"):
        c = cli[tag]
        out = str(tmp_path / f"{tag}.mtx")
        r = subprocess.run([exe, "--matrix", os.path.join(GOLDEN, c["matrix"] + ".mtx"), "--prime", str(c["prime"]), "--n", str(c["n"]),
                            "--verify", "--gpus", str(gpus), "--output-file", out] + (["--right"] if c["right"] else []),
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, BLZ_LOOPBACK="1"), cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-1500:]
        assert f"after {c['iterations']} iterations" in r.stdout
        assert hashlib.sha256(open(out, "rb").read()).hexdigest() == c["out_sha256"], tag


REF_OMP = os.path.join(ROOT, "oracle", "_ref", "lanczos_modp_omp_ref")


@pytest.mark.skipif(not os.path.exists(REF_OMP), reason="oracle/_ref not built (reference checkout absent)")
def test_the_reference_text_checkpoint_both_ways_with_several_contexts(tmp_path):
    """tests/test_gpu_cli.py::test_resume_from_the_reference_text_checkpoint with three contexts on this side: the reference's own
    OpenMP binary (1 thread) starts the solve and writes v.txt tmp.txt Av.txt p.txt verbosity.txt (openMP/lanczos_modp.c:571-676),
    three contexts finish it; then three contexts start one, export the text files (assembled from their slabs), and the
    reference finishes it.  Both must end in the reference's uninterrupted result."""
    import hashlib
    import json
    import subprocess
    exe = os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "lib", "lanczos_modp")
    mpath = os.path.join(GOLDEN, "rand300x200.mtx")
    want = json.load(open(os.path.join(GOLDEN, "cli.json")))["rand300x200_p65537_n4_left"]["out_sha256"]
    base = ["--matrix", mpath, "--prime", "65537", "--n", "4"]
    env1 = dict(os.environ, OMP_NUM_THREADS="1")
    work = tmp_path / "ref"
    work.mkdir()
    r = subprocess.run([REF_OMP] + base + ["--checkpoint", "0", "--stop-after", "10"], cwd=str(work), env=env1, capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(work / "v.txt")
    out = str(tmp_path / "resumed.mtx")
    r = subprocess.run([exe] + base + ["--gpus", "3", "--load-checkpoint", "--output-file", out], cwd=str(work),
                       env=dict(os.environ, BLZ_LOOPBACK="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == want
    work2 = tmp_path / "ours"
    work2.mkdir()
    r = subprocess.run([exe] + base + ["--gpus", "3", "--checkpoint", "0", "--stop-after", "10"], cwd=str(work2),
                       env=dict(os.environ, BLZ_LOOPBACK="1", BLZ_REF_CHECKPOINT="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and os.path.exists(work2 / "p.txt"), r.stdout + r.stderr[-1500:]
    out2 = str(tmp_path / "ref_resumed.mtx")
    r = subprocess.run([REF_OMP] + base + ["--load-checkpoint", "--output-file", out2], cwd=str(work2), env=env1, capture_output=True, text=True)
    assert r.returncode == 0
    assert hashlib.sha256(open(out2, "rb").read()).hexdigest() == want
