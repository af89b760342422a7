"""Real multi-GPU runs of the library's own exchange code (csrc/blz_api.hip: enqueue_product's piece-major
ncclAllGather on the exchange stream, the u64 ncclAllReduce of the n x n partials, the event choreography between
the two streams) -- what mpi/lanczos_modp.c:967-1149 and :1209-1247 do through rank 0.

Every test here but the last needs at least two visible devices and SKIPS below that (the development pool has one GPU per box):
the moment a multi-GPU box runs the suite, the N > 1 path is validated against the reference's output hashes and
against the oracle, with no further work.
"""
import hashlib
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import blz
import oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
LIBDIR = os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "lib")
EXE = os.path.join(LIBDIR, "lanczos_modp")
NDEV = blz.device_count()
need2 = pytest.mark.skipif(NDEV < 2, reason=f"needs >= 2 GPUs ({NDEV} visible)")
GPU_COUNTS = [g for g in (2, 3, 4, 8) if g <= max(NDEV, 2)]


@need2
@pytest.mark.parametrize("gpus", GPU_COUNTS)
def test_cli_with_several_gpus_reproduces_the_reference_hashes(tmp_path, gpus):
    """lib/lanczos_modp --gpus G (one host thread per context, real RCCL communicator) on the golden matrices: the
    output file must be the one the unmodified reference binary wrote (tests/golden/cli.json)."""
    if gpus > NDEV:
        pytest.skip(f"{gpus} GPUs asked, {NDEV} visible")
    cli = json.load(open(os.path.join(GOLDEN, "cli.json")))
    cli.pop("_validation")
    for tag, c in cli.items():
        if c["matrix"] not in ("rand300x200", "wide120x260"):
            continue
        out = str(tmp_path / f"{tag}_g{gpus}.mtx")
        r = subprocess.run([EXE, "--matrix", os.path.join(GOLDEN, c["matrix"] + ".mtx"), "--prime", str(c["prime"]),
                            "--n", str(c["n"]), "--output-file", out, "--gpus", str(gpus)]
                           + (["--right"] if c["right"] else []), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert f"after {c['iterations']} iterations" in r.stdout, (tag, r.stdout[-400:])
        assert hashlib.sha256(open(out, "rb").read()).hexdigest() == c["out_sha256"], tag


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def _ranks(nranks, fn):
    """run fn(rank) in one thread per rank (ctypes releases the GIL: the ranks really meet inside RCCL)"""
    errs = [None] * nranks

    def go(g):
        try:
            fn(g)
        except BaseException as exc:    # noqa: BLE001 -- re-raised in the caller
            errs[g] = exc

    ts = [threading.Thread(target=go, args=(g,)) for g in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    for e in errs:
        if e is not None:
            raise e


@need2
@pytest.mark.parametrize("chunks", ["1", "4"])
@pytest.mark.parametrize("nranks", GPU_COUNTS)
@pytest.mark.parametrize("name,p,n,right", [("rand3000x2000", (1 << 61) - 1, 8, False), ("rand3000x2000", 1073741789, 4, True),
                                            ("rand300x200", 65537, 16, False)])
def test_contexts_with_a_real_communicator_against_the_oracle(monkeypatch, name, p, n, right, nranks, chunks):
    """One context per GPU, blz_comm_init over a real ncclUniqueId, blz_iterate: the pipelined exchange with K = 1
    and K = 4 pieces per all-gather.  Every rank's rows of v and p must be the oracle's, and the all-reduced n x n
    operands identical on every rank (also after the batch has run past the stop)."""
    if nranks > NDEV:
        pytest.skip(f"{nranks} GPUs asked, {NDEV} visible")
    monkeypatch.setenv("BLZ_AG_CHUNKS", chunks)
    M = blz.Matrix.load(os.path.join(GOLDEN, name + ".mtx"), p)
    Mo = orc.Matrix.load(os.path.join(GOLDEN, name + ".mtx"), p)
    want = orc.block_lanczos(Mo, n, p, right=right, stop_after=6)
    uid = blz.comm_unique_id()
    got_v, got_p, small = [None] * nranks, [None] * nranks, [None] * nranks

    def rank_main(g):
        with blz.Context(p, n, device=g) as ctx:
            ctx.comm_init(uid, g, nranks)
            ctx.set_matrix(M, right, g, nranks)
            ctx.init_v()
            done, stopped, _ = ctx.iterate(6)
            assert done == want["iterations"] and not stopped
            got_v[g], got_p[g] = ctx.get_block(blz.V), ctx.get_block(blz.P)
            small[g] = (ctx.get_small(blz.VTAV), ctx.get_small(blz.WINV))

    _ranks(nranks, rank_main)
    v = np.zeros_like(want["v"])
    pb = np.zeros_like(want["p"])
    for g in range(nranks):             # a rank returns its own rows, zeros elsewhere
        v |= got_v[g]
        pb |= got_p[g]
    assert np.array_equal(v, want["v"]) and np.array_equal(pb, want["p"])
    for g in range(1, nranks):
        assert np.array_equal(small[g][0], small[0][0]) and np.array_equal(small[g][1], small[0][1])


@need2
def test_full_solve_past_the_stop_keeps_the_small_operands_sane():
    """Several ranks run a batch that goes past termination: the n x n operands each rank reads back afterwards are
    still residues (the all-reduce is out of place and idempotent once the stop flag is up), and the kernel vectors are
    the reference's."""
    p, n, nranks = 65537, 4, 2
    path = os.path.join(GOLDEN, "rand300x200.mtx")
    M, Mo = blz.Matrix.load(path, p), orc.Matrix.load(path, p)
    want = orc.block_lanczos(Mo, n, p)
    uid = blz.comm_unique_id()
    got = [None] * nranks

    def rank_main(g):
        with blz.Context(p, n, device=g) as ctx:
            ctx.comm_init(uid, g, nranks)
            ctx.set_matrix(M, False, g, nranks)
            ctx.init_v()
            done, stopped, _ = ctx.iterate(want["iterations"] + 40)
            assert stopped and done == want["iterations"]
            got[g] = (ctx.get_block(blz.V), ctx.get_small(blz.VTAV), ctx.get_small(blz.VTAAV))

    _ranks(nranks, rank_main)
    v = got[0][0] | got[1][0]
    assert np.array_equal(v, want["v"])
    for g in range(nranks):
        assert (got[g][1] < p).all() and (got[g][2] < p).all()
        assert np.array_equal(got[g][1], got[0][1])


@need2
@pytest.mark.parametrize("world", [g for g in (2, 4, 8) if g <= max(NDEV, 2)])
def test_bench_under_torch_distributed_run_checks_itself(world):
    """bench.py as the driver launches it for N > 1 (one process per GPU): exit code 0 and the line says that the sharded
    run equals a single-GPU solve of the same system."""
    if world > NDEV:
        pytest.skip(f"{world} GPUs asked, {NDEV} visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", str(world),
                        "--workload", "tiny", "--steps", "5", "--warmup", "1", "--cpu-seconds", "0"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["value"] and d["sharded_equals_single_gpu"]["equal"] is True


@need2
@pytest.mark.parametrize("world", [g for g in (2, 8) if g <= max(NDEV, 2)])
def test_bench_starts_its_own_ranks(world):
    """`python bench.py --gpus N` with no launcher around it (how the driver starts the N = 1 run): the parent starts one
    process per GPU itself and relays their line.  The line must say how many ranks RCCL saw, how each exchange was cut and
    what every collective cost."""
    if world > NDEV:
        pytest.skip(f"{world} GPUs asked, {NDEV} visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--workload", "tiny", "--steps", "5",
                        "--warmup", "1", "--cpu-seconds", "0"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["value"] and d["sharded_equals_single_gpu"]["equal"] is True
    mg = d["multi_gpu"]
    assert mg["rccl_ranks_seen"] == world
    assert mg["pieces"]["spmv1"] >= 1 and mg["pieces"]["spmv2"] >= 1
    assert mg["collectives"]["allreduce"]["calls_per_step"] == 1
    assert d["roofline"]["traffic"] is None            # no PMC run of this (workload, N) is committed


def test_bench_per_rank_set_up_path_on_one_rank():
    """Config 5's way of starting (`bench.py --gpus 8 --workload synth5`): every rank generates its own rows and columns
    (blz_synth_coo_part), prepares from that share alone (blz_prepare_rank) and nobody holds the whole matrix.  Run here on a
    small matrix of the same kind with ONE rank under the launcher and the exchange code forced on; the line must say that the
    reference's in-loop invariants hold on the n x n operands after the measured iterations."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BLZ_BENCH_PER_RANK="1", BLZ_FORCE_COMM="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1",
                        "--workload", "tiny5", "--steps", "4", "--warmup", "1", "--cpu-seconds", "0", "--ref-iterations", "0"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["value"] and d["sharded_equals_single_gpu"]["invariants_hold"] is True
    assert "every rank generates" in d["data"]
    assert d["multi_gpu"]["collectives"]["allgather_v"]["calls_per_step"] >= 1


def test_bench_shared_set_up_path_on_one_rank():
    """The set-up bench.py uses for N > 1 -- rank 0 prepares once, writes the cache file, every rank maps it and uploads
    its slabs -- run with ONE rank under torch.distributed.run (BLZ_BENCH_SHARE=1 makes rank 0 map the file as the other
    ranks would) and the RCCL exchange code forced on (BLZ_FORCE_COMM=1), so that a one-GPU box covers everything but the
    wire.  The line must say that the run equals a plain single-GPU solve."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BLZ_BENCH_SHARE="1", BLZ_FORCE_COMM="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1",
                        "--workload", "tiny", "--steps", "5", "--warmup", "1", "--cpu-seconds", "0", "--ref-iterations", "0"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] and d["sharded_equals_single_gpu"]["equal"] is True
    assert d["kernels"]["allgather_v"]["launches"] > 0 and d["kernels"]["allreduce"]["launches"] > 0
    # the N > 1 diagnostics, on the 1-rank communicator: what RCCL says it has, the pieces, a time per collective
    mg = d["multi_gpu"]
    assert mg["rccl_ranks_seen"] == 1 and mg["pieces"]["spmv1"] >= 1
    assert mg["collectives"]["allgather_v"]["ms_per_call"] > 0 and mg["collectives"]["allreduce"]["calls_per_step"] == 1
    assert mg["exposed_exchange_ms_per_step"] >= 0 and mg["compute_ms_per_step"] > 0
