#!/usr/bin/env python3
"""bench.py -- the hot path of block-Lanczos mod p on MI355X, measured as BASELINE.json asks.

A "step" is one pass of the loop body of block_lanczos() (sequential/lanczos_modp.c:631-659):
two block SpMVs (tmp = M^T v, Av = M tmp), the n x n block inner products, the semi-inverse and the
row-local block update, all on the GPU with every operand resident in HBM before the clock starts.

  metric  = nnz*n mod-p MAC/s : (2 SpMV * nnz * n) * steps / time   (block products are NOT counted)
  value   = whole-job rate over all ranks (strong scaling: the matrix is fixed, rows are partitioned);
            the timed region (exactly K steps between barrier + synchronize) is repeated 5 times and the
            MEDIAN region is reported; every region's time is listed under "repeats"
  roofline= the SpMV kernel (dominant): algorithmic bytes of SURVEY 8(d) / mean launch duration measured
            with HIP events on the solver's stream inside this run, against the 8 TB/s HBM3E peak
  cpu_baseline = the oracle's by-rows OpenMP kernels timed on this box's host cores on a bounded sample of
            the SAME workload, at 16 threads and at all cores; a reported baseline, not the target.
  extra.workloads = the other single-GPU configs (relat9 and relat8 shapes) in the same line (N = 1 only)

Real matrices: if $BLZ_MTX_DIR holds GL7d19.mtx / relat9.mtx / relat8.mtx they are loaded instead of the seeded
synthetic stand-ins of the same shape ("data" says which).

Launch: python bench.py [--gpus N --steps K --warmup W].  N > 1 runs one process per GPU: either the caller starts
them (python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...: RANK / WORLD_SIZE in the
environment), or `python bench.py --gpus N` alone does it itself -- the parent, which never touches the GPU, starts
that same torch.distributed.run command as a CHILD process on a free port of 127.0.0.1, relays the one JSON line and
the exit code (--dry-run prints the command instead).  torch.distributed (gloo) carries only the control plane (RCCL
id, barrier, max of the times, verdicts); the data path's collectives are RCCL calls made by libblz_hip.so on its own
streams.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python"))

P61 = (1 << 61) - 1
P31 = (1 << 31) - 1
# Shapes of BASELINE.json's configs (SuiteSparse headers as recorded in SURVEY 8); the matrices
# themselves are not on the box, so seeded synthetic stand-ins of the same shape are generated.
WORKLOADS = {
    "gl7d19": dict(desc="JGD_GL7d/GL7d19-shape synthetic", rows=1911130, cols=1955309, nnz=37322725, prime=P61,
                   n=8, right=False, seed=0x474C3764, pattern=False, mtx="GL7d19.mtx"),
    "relat9": dict(desc="JGD_Relat/relat9-shape synthetic", rows=12360060, cols=549336, nnz=38955420, prime=P61,
                   n=8, right=True, seed=0x52454C39, pattern=False, mtx="relat9.mtx"),
    "relat8": dict(desc="JGD_Relat/relat8-shape synthetic", rows=345688, cols=12347, nnz=1334038, prime=P31,
                   n=4, right=False, seed=0x52454C38, pattern=False, mtx="relat8.mtx"),
    # config 5 is 50M x 50M, 2e9 nnz over 8 GPUs; this is one GPU's quarter-scale share of it (same density 40/row,
    # all-ones pattern path, n=16, 128-byte block rows)
    "synth5q": dict(desc="config-5-shape synthetic at 1/4 linear scale (all-ones pattern)", rows=12500000, cols=12500000,
                    nnz=500000000, prime=P61, n=16, right=False, seed=0x53594E35, pattern=True),
    # config 5 at FULL size on one GPU (fits: ~42 GB of the 288 GB HBM); minutes of host-side set-up
    # (with N > 1 every rank generates and prepares ONLY its own rows and columns -- blz_synth_coo_part, blz_prepare_rank --
    # so no process ever holds the 24 GB of triplets: `python bench.py --gpus 8 --workload synth5` is BASELINE config 5)
    "synth5": dict(desc="config-5 synthetic (all-ones pattern), full size", rows=50000000, cols=50000000,
                   nnz=2000000000, prime=P61, n=16, right=False, seed=0x53594E35, pattern=True, per_rank=True),
    # EXTRA workload, not a BASELINE config and never the headline: a matrix WITH structure (heavy-tailed column degrees,
    # banded supports -- the shape of a sieve relation matrix), for what the uniform stand-ins cannot show: the renumbering
    # chosen by line footprint, the per-XCD row ranges and the LDS panel of the SpMV
    "nfs": dict(desc="EXTRA (not a BASELINE config): structured synthetic, 40 % of a row's entries ~1/(c+16), 30 % in a band "
                     "of 4096 columns, 30 % uniform", rows=2000000, cols=2000000, nnz=40000000, prime=P61, n=8, right=False,
                seed=0x4E465331, pattern=False, structured=dict(hot_pct=40, band_pct=30, band=4096)),
    # EXTRA, never the headline: every entry in a band of 4096 columns around r*C/R -- all gathers hit L1 / L2, so the products are
    # paced by the kernels themselves, not by the fabric (what a well-ordered real matrix would look like at best)
    "band": dict(desc="EXTRA (not a BASELINE config): banded synthetic, every entry within 2048 columns of r*C/R", rows=2000000,
                 cols=2000000, nnz=40000000, prime=P61, n=8, right=False, seed=0x42414E44, pattern=False,
                 structured=dict(hot_pct=0, band_pct=100, band=4096)),
    "tiny": dict(desc="tiny synthetic (self-test)", rows=20000, cols=15000, nnz=200000, prime=P61,
                 n=8, right=False, seed=0x54494E59, pattern=False),
    "tiny5": dict(desc="tiny config-5-like synthetic (self-test of the per-rank set-up)", rows=60000, cols=60000, nnz=1800000, prime=P61,
                  n=16, right=False, seed=0x54494E35, pattern=True, per_rank=True),
}
LIVE_BUDGET_S = 200          # no further profiler pass of an EXTRA workload is started once the run has taken this long
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s HBM3E peak
REPEATS = 5             # timed regions of K steps each; the median one is reported


def prime_name(p):
    return {P61: "2^61-1", P31: "2^31-1"}.get(p, str(p))


def spmv_alg_bytes(nnz, rows_out, rows_in, n, w, pattern):
    """every array counted once -- (col_idx + val) per entry, row_ptr, X read, Y written."""
    return nnz * (4 + (0 if pattern else 4)) + 4 * (rows_out + 1) + (rows_in + rows_out) * n * w


def make_matrix(blz, w, p):
    """The workload's matrix: the real file when $BLZ_MTX_DIR has it, else the seeded synthetic of the same shape."""
    mdir = os.environ.get("BLZ_MTX_DIR")
    if mdir and w.get("mtx") and os.path.exists(os.path.join(mdir, w["mtx"])):
        path = os.path.join(mdir, w["mtx"])
        return blz.Matrix.load(path, p), f"real: {path}"
    if w.get("structured"):
        return (blz.Matrix.synth_structured(w["rows"], w["cols"], w["nnz"], w["seed"], p, pattern=w["pattern"], **w["structured"]),
                "synthetic")
    return blz.Matrix.synth(w["rows"], w["cols"], w["nnz"], w["seed"], p, pattern=w["pattern"]), "synthetic"


class Stopped(Exception):
    pass


def launcher_command(argv, gpus, port):
    """the command `bench.py --gpus N` runs as its child when nobody has started the ranks: what the driver would type"""
    rest = [a for a in argv if a != "--dry-run"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + rest


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args):
    """N > 1 asked and no launcher around us: start one process per GPU as a child (this process has made no GPU call
    and imports neither torch nor the library), pass its stderr through, relay its single JSON line and its exit code."""
    import subprocess
    cmd = launcher_command(sys.argv[1:], args.gpus, free_port())
    if args.dry_run:
        print(json.dumps({"dry_run": True, "n_gpus": args.gpus, "cmd": cmd}))
        return 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))    # (the launcher would set 1)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT, text=True)
    line = None
    for ln in child.stdout:                 # rank 0 writes exactly one line that starts with '{'; anything else is noise
        if ln.startswith("{"):
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
    code = child.wait()
    if line is not None:
        print(line, flush=True)
    elif code == 0:
        code = 1
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
    return code


def ctx_width(ctx, n):
    """block width in HBM (the next power of two unless BLZ_NO_PAD=1): the n x n operands travel padded"""
    if os.environ.get("BLZ_NO_PAD") == "1":
        return n
    g = 1
    while g < n:
        g <<= 1
    return g


def host_cpu():
    """model name and logical CPU count of this host (the CPU baseline is quoted with both)"""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model, "logical_cpus": os.cpu_count()}


def gather_ceiling(row_bytes, table_bytes):
    """The micro-benchmarked ceiling for random block-row gathers (tools/ubench2.hip), read from the committed run
    profiles/r02_ubench2_gather_requests_tlb_panel.txt: the best plain-load rate for this row size (32 / 64 / 128 B) at
    the table size nearest to the operand's.  Returns (rows per second, the line it came from) or (None, None)."""
    import re
    path = os.path.join(ROOT, "profiles", "r02_ubench2_gather_requests_tlb_panel.txt")
    want = min((32, 64, 128), key=lambda b: abs(b - row_bytes))
    best = None
    try:
        for ln in open(path):
            mt = re.match(r"gather\s+hipMalloc\s+plain\s+(\d+)B rows\s+table\s+([0-9.]+) MB.*?([0-9.]+) G rows/s\s*(\S*)", ln)
            if not mt or int(mt.group(1)) != want or mt.group(4).startswith("window"):
                continue
            dist_ = abs(float(mt.group(2)) * 1e6 - table_bytes)
            key = (dist_, -float(mt.group(3)))
            if best is None or key < best[0]:
                best = (key, float(mt.group(3)) * 1e9, ln.strip())
    except OSError:
        return None, None
    return (best[1], f"{os.path.relpath(path, ROOT)}: {best[2]}") if best else (None, None)


def live_gather_ceiling(row_bytes, table_bytes, avg_len):
    """The same ceiling measured on THIS box in THIS run (the fill rate differs by up to 10 % between the pool's boxes): the bare
    gather loop of tools/gather_ceiling.hip -- index load, gather, add -- for this row size, an operand of this size and
    output rows of this many gathers, as a child process (a few seconds).  Only for 64- and 128-byte block rows of 8-byte
    words; returns the program's JSON or {"error": ...}."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "gather_ceiling")
    if row_bytes not in (64, 128) or not os.path.exists(exe):
        return {"error": "not measured (row size or tools/gather_ceiling missing)"}
    count = 40 if table_bytes < 1e9 else 100          # million gathers per launch: ~1 ms / ~2 ms
    cmd = ["timeout", "-k", "5", "60", exe, str(row_bytes), f"{max(table_bytes / 1e6, 1.0):.1f}", str(max(1, int(round(avg_len)))),
           str(count)]
    try:
        rr = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=90)
        if rr.returncode != 0:
            return {"error": f"tools/gather_ceiling ended with {rr.returncode}: {(rr.stderr or rr.stdout)[-200:]}"}
        return json.loads(rr.stdout.strip().splitlines()[-1])
    except Exception as exc:
        return {"error": repr(exc)}


def measure(blz, torch, dist, ctx, info, w, steps, warmup, repeats):
    """Warm up, time `repeats` regions of exactly `steps` steps (barrier + synchronize on both sides, MAX over ranks),
    then the same steps once more with HIP-event spans per kernel class.  Returns the numbers of one workload."""
    n, right = w["n"], w["right"]

    def barrier():
        if dist is not None:
            dist.barrier()

    def all_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    bad = [0.0]

    def run(k):
        done, stopped, dev_ms = ctx.iterate(k)
        if stopped or done != k:
            bad[0] = 1.0
        return dev_ms

    def check():
        # every rank takes the same decision (outside the timed regions): a solve that terminated early is no benchmark
        if all_max(bad[0]) > 0:
            raise Stopped("the solve terminated inside the measured steps; use a larger workload")

    if warmup > 0:
        run(warmup)
    check()
    times, dev = [], []
    for _ in range(repeats):
        ctx.sync()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        dev_ms = run(steps)
        ctx.sync()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        barrier()
        times.append(all_max(elapsed))
        dev.append(dev_ms)
    check()
    elapsed = statistics.median(times)

    ctx.profile(True)
    run(steps)
    prof = ctx.profile_read()
    ctx.profile(False)
    check()

    word = ctx.word_bytes
    rows_v, rows_t = ctx.rows(blz.V), ctx.rows(blz.TMP)
    _, loc_v = ctx.local_rows(blz.V)
    _, loc_t = ctx.local_rows(blz.TMP)
    # per-rank algorithmic bytes of the two SpMV launches (X is read whole by every rank)
    nnz1, nnz2 = ctx.local_nnz(not right), ctx.local_nnz(right)   # tmp = (right ? M : M^T) v, then Av = the other one
    pattern = info["pattern"]
    bytes1 = spmv_alg_bytes(nnz1, loc_t, rows_v, n, word, pattern)
    bytes2 = spmv_alg_bytes(nnz2, loc_v, rows_t, n, word, pattern)
    kernels = {k: dict(ms_mean=(v["ms_total"] / steps) if v["launches"] else None, launches=v["launches"])
               for k, v in prof.items()}     # ms_mean = per step (a step may issue several launches of one class)
    kernels["spmv1"]["alg_bytes"] = bytes1
    kernels["spmv2"]["alg_bytes"] = bytes2
    for k_ in ("spmv1", "spmv2"):
        if kernels[k_]["ms_mean"]:
            kernels[k_]["alg_GBps"] = kernels[k_]["alg_bytes"] / (kernels[k_]["ms_mean"] * 1e-3) / 1e9
    kernels["block_dot"]["alg_bytes"] = 2 * loc_v * n * word
    kernels["orthogonalize"]["alg_bytes"] = 5 * loc_v * n * word
    t_spmv_ms = prof["spmv1"]["ms_total"] / max(steps, 1)
    macs_per_step = 2 * info["nnz"] * n
    loc, kind = ctx.locality()
    return dict(elapsed=elapsed, times=times, device_ms_per_step=statistics.median(dev) / steps, kernels=kernels, prof=prof,
                loc_rows=(loc_v, loc_t),
                t_spmv_ms=t_spmv_ms, alg_bytes=bytes1, achieved=bytes1 / (t_spmv_ms * 1e-3) / 1e9, nnz1=nnz1,
                macs_per_step=macs_per_step, value=macs_per_step * steps / elapsed, word=word, rows_v=rows_v, rows_t=rows_t,
                renumbering=dict(lines_per_entry=dict(M=loc[0], Mt=loc[1]), order=("smallest", "file", "mean", "sweeps")[kind]),
                lds_panel={"spmv1": dict(zip(("rows", "share"), ctx.panel_rows(not right))),
                           "spmv2": dict(zip(("rows", "share"), ctx.panel_rows(right)))})


def first_spmv_kernel(name):
    """rocprofv3's kernel name -> is it the plain (no fused block_dot) SpMV, i.e. the first product of a step"""
    name = name.split("(")[0].replace("void ", "")
    if name.startswith("k_spmv<"):
        return True
    if name.startswith("k_spmv_staged<") or name.startswith("k_spmv_panel<"):
        targs = [a.strip() for a in name[name.index("<") + 1:].rstrip(">").split(",")]
        return len(targs) > 3 and targs[3] == "false"        # <W, G, MERS, DOT, ...>
    return False


def spmv_traffic(workload, world):
    """HBM-side traffic of the first SpMV's kernel from the committed rocprofv3 PMC passes of this same command
    (tools/gpu_profile.sh: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs).  The default run measures it again itself
    (live_traffic below) and keeps this figure beside it as the cross-check."""
    tpath = os.path.join(ROOT, "profiles", f"traffic_{workload}_n{world}.json")
    if not os.path.exists(tpath):
        return None, None
    for name, rec in json.load(open(tpath)).items():
        if first_spmv_kernel(name) and "FETCH_SIZE_bytes_per_launch" in rec and "WRITE_SIZE_bytes_per_launch" in rec:
            # gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B although every request is a
            # 128-byte line fill, so it is doubled; WRITE_SIZE is exact.  profiles/r01_v6_gl7d19_pmc_l2_fabric.txt and
            # profiles/r02_ubench2_pmc.txt confirm it: every L2->fabric read request of a gather is tallied under
            # TCC_EA0_RDREQ_128B, whatever the allocation kind or load policy.
            return 2 * rec["FETCH_SIZE_bytes_per_launch"] + rec["WRITE_SIZE_bytes_per_launch"], os.path.relpath(tpath, ROOT)
    return None, None


def under_profiler():
    return any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def live_traffic(workload, limit_s=150):
    """The roofline's `traffic` measured by THIS run: two child processes of this same program under rocprofv3, `--pmc
    FETCH_SIZE` and `--pmc WRITE_SIZE` in separate passes (MI355X_MICROARCH.md, HBM section) with nothing else traced, a few
    steps each; per launch of the first SpMV's kernel: 2 x FETCH_SIZE + WRITE_SIZE, both x 1024 bytes (same corrections as
    spmv_traffic).  Children, never an exec: this process has initialised the GPU.  Each pass has a hard limit; a pass that
    fails or is killed ends the measurement (no further GPU step after a kill) and the caller keeps the committed figure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    got, t0 = {}, time.time()
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        env = dict(os.environ, TMPDIR="/tmp")
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["timeout", "-k", "5", str(limit_s), prof, "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(td, ctr), "--",
                   sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", "3", "--warmup", "1", "--repeats", "1",
                   "--cpu-seconds", "0", "--ref-iterations", "0", "--extras", "0", "--live-traffic", "0"]
            try:
                rr = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=limit_s + 30)
            except Exception as exc:
                return {"error": f"{ctr} pass: {exc!r}"}
            if rr.returncode != 0:
                return {"error": f"{ctr} pass ended with {rr.returncode}: {(rr.stderr or '')[-200:]}"}
            tot, cnt = 0.0, 0
            for f in glob.glob(os.path.join(td, ctr, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == ctr and first_spmv_kernel(row["Kernel_Name"]):
                        tot += float(row["Counter_Value"])
                        cnt += 1
            if not cnt:
                return {"error": f"{ctr} pass: no launch of the first SpMV's kernel in the counter file"}
            got[ctr] = (tot / cnt * 1024, cnt)
    fetch, wr = got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    return {"traffic": 2 * fetch + wr, "FETCH_SIZE_bytes_per_launch": fetch, "WRITE_SIZE_bytes_per_launch": wr,
            "launches": got["FETCH_SIZE"][1], "seconds": time.time() - t0}


def main():
    t_main = time.time()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="gl7d19", choices=sorted(WORKLOADS))
    ap.add_argument("--repeats", type=int, default=REPEATS, help="timed regions of --steps steps; the median is reported")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = 16 threads and all cores, else this many only")
    ap.add_argument("--ref-iterations", type=int, default=50,
                    help="iterations of the unmodified reference OpenMP binary on a 1/4-scale sample (0 = skip)")
    ap.add_argument("--extras", type=int, default=-1,
                    help="1/0: also run the other single-GPU workloads (relat9, relat8, config-5 quarter shape, structured); "
                         "default: only with the default workload at N=1")
    ap.add_argument("--live-traffic", type=int, default=-1,
                    help="1/0: measure roofline.traffic in this run (two rocprofv3 --pmc child passes of the same workload); "
                         "default: only with the default workload at N=1 and not under a profiler")
    ap.add_argument("--dry-run", action="store_true",
                    help="print what would be started (for N > 1 without a launcher: the torch.distributed.run command) and exit")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args))
    if args.dry_run:
        print(json.dumps({"dry_run": True, "n_gpus": int(os.environ.get("WORLD_SIZE", "1")),
                          "cmd": [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--dry-run"]}))
        return

    # stdout carries exactly ONE line (the JSON): RCCL and gloo print banners to fd 1 from native code, so fd 1 is
    # pointed at stderr for the duration of the run and the result is written to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    # the host driver of this pool only supports dmabuf IPC: without this RCCL's buffer exchange between the ranks fails with
    # "hipIpcGetMemHandle: invalid argument"; it must be in the environment before the first HIP call of the process
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:          # a launcher's WORLD_SIZE wins over the flag
        args.gpus = world

    if world > 1 and ("OMP_NUM_THREADS" not in os.environ or
                      (os.environ["OMP_NUM_THREADS"] == "1" and "TORCHELASTIC_RUN_ID" in os.environ
                       and os.environ.get("BLZ_BENCH_KEEP_OMP") != "1")):
        # the host-side set-up (generation, renumbering, CSR builds) is OpenMP code: share the cores between the ranks.
        # torch.distributed.run sets OMP_NUM_THREADS=1 for its workers when the caller has not set it -- that would make rank 0
        # prepare the matrix on one core; the OpenMP runtime has not been loaded yet at this point, so this still takes effect.
        os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 8) // world))

    import numpy as np
    import blz
    blz.lib()       # load libblz_hip.so (and with it /opt/rocm's HIP runtime) before torch brings its own copy
    import torch

    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("BLZ_BENCH_DIST") == "1":
        # launched by torch.distributed.run (also with one process: the same code path as N > 1)
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", init_method="env://")
    if blz.device_count() < 1:
        sys.exit("bench.py: no GPU visible -- libblz_hip has no CPU path")
    # one GPU per rank; if the launcher narrowed the visibility to one device per process, that device is number 0
    local_rank %= blz.device_count()
    torch.cuda.set_device(local_rank)

    def leave(code, msg):
        """every rank leaves together (a lone sys.exit would leave the others in a barrier until the gloo timeout)"""
        print(f"bench.py: {msg}", file=sys.stderr)
        if dist is not None:
            try:
                dist.destroy_process_group()
            except Exception:
                pass
        sys.exit(code)

    w = WORKLOADS[args.workload]
    p, n, right = w["prime"], w["n"], w["right"]
    t0 = time.time()
    # Only rank 0 holds the matrix: it does the rank-independent set-up once (renumbering, CSR(M), CSR(M^T), partition) and
    # shares it through a cache file that the other ranks map (one copy of the pages per node); every rank then uploads
    # its own slabs.  Round 1 generated, renumbered and built the whole matrix in every process.
    M, data, info = None, None, None
    # every rank makes its own share; nobody holds the whole matrix (BLZ_BENCH_PER_RANK=1: also with one rank under a launcher,
    # the way to run that code on a one-GPU box)
    per_rank = bool(w.get("per_rank")) and (world > 1 or (os.environ.get("BLZ_BENCH_PER_RANK") == "1" and "RANK" in os.environ))
    if per_rank:
        data = "synthetic (every rank generates its own rows and columns of the same seeded matrix)"
        info = dict(nrows=w["rows"], ncols=w["cols"], nnz=w["nnz"], pattern=bool(w["pattern"]), data=data)
    elif rank == 0:
        M, data = make_matrix(blz, w, p)
        info = dict(nrows=M.nrows, ncols=M.ncols, nnz=M.nnz, pattern=bool((M.x == 1).all()), data=data)
    if dist is not None:
        box = [info]
        dist.broadcast_object_list(box, src=0)
        info = box[0]
        data = info["data"]
    t_gen = time.time() - t0

    ctx = blz.Context(p, n, device=local_rank)
    if dist is not None:
        uid = [blz.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], rank, world)
    t0 = time.time()
    # BLZ_BENCH_SHARE=1 takes the multi-rank set-up path (prepare, cache file, map, upload) with one rank as well, and makes
    # rank 0 map the file like the others: the way to run that code on a one-GPU box (tests/test_gpu_cli.py)
    share_test = dist is not None and os.environ.get("BLZ_BENCH_SHARE") == "1"
    if per_rank:
        mine = None
        try:
            R_, C_, nnz_ = w["rows"], w["cols"], w["nnz"]
            rb = [R_ * g // world for g in range(world + 1)]        # equal rows = equal entries (every row has nnz / R of them)
            cb = [C_ * g // world for g in range(world + 1)]
            rows_part = blz.Matrix.synth_part(R_, C_, nnz_, w["seed"], p, rows=(rb[rank], rb[rank + 1]), pattern=w["pattern"])
            cols_part = blz.Matrix.synth_part(R_, C_, nnz_, w["seed"], p, cols=(cb[rank], cb[rank + 1]), pattern=w["pattern"])
            K = ctx.exchange_pieces_for(R_, C_, nnz_, world)
            with blz.Prepared.prepare_rank(rows_part, cols_part, R_, C_, nnz_, right, rank, world, rb, cb, chunks=K) as P:
                ctx.set_matrix_prepared(P, rank)
            del rows_part, cols_part
        except Exception as exc:
            mine = repr(exc)
        errs = [None] * world
        dist.all_gather_object(errs, mine)
        if any(errs):
            leave(1, f"per-rank set-up failed: {[e for e in errs if e]}")
    elif world == 1 and not share_test:
        ctx.set_matrix(M, right, 0, 1)
    else:
        import shutil
        import tempfile
        key = 0x42454E43 ^ (world << 40) ^ info["nnz"]
        name = f"blz_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getpid()}_{args.workload}_{world}.blzcache"
        status = [None]
        P = None
        if rank == 0:
            try:
                P = blz.Prepared.prepare_for(ctx, M, right, world)
                need = 24 * info["nnz"] + 64 * (info["nrows"] + info["ncols"]) + (1 << 20)      # generous bound on the file size
                dirs = [d for d in ("/dev/shm", tempfile.gettempdir())
                        if os.path.isdir(d) and os.access(d, os.W_OK) and shutil.disk_usage(d).free > need]
                if not dirs:
                    raise OSError("no room for the shared set-up file in /dev/shm or the temp directory")
                cache = os.path.join(dirs[0], name)
                P.save(cache, key)
                if share_test:
                    P.close()
                    P = blz.Prepared.load(cache, key)
                status = [cache]
            except Exception as exc:            # the other ranks are waiting: tell them instead of leaving them in a barrier
                status = [exc]
        dist.broadcast_object_list(status, src=0)
        if isinstance(status[0], Exception):
            leave(1, f"set-up on rank 0 failed: {status[0]!r}")
        cache = status[0]
        try:
            if rank != 0:
                P = blz.Prepared.load(cache, key)
            ctx.set_matrix_prepared(P, rank)
            mine = None
        except Exception as exc:
            mine = repr(exc)
        finally:
            if P is not None:
                P.close()
        errs = [None] * world
        dist.all_gather_object(errs, mine)
        if rank == 0:
            os.unlink(cache)
        if any(errs):
            leave(1, f"matrix upload failed: {[e for e in errs if e]}")
    ctx.init_v()
    ctx.sync()
    t_setup = time.time() - t0

    try:
        r = measure(blz, torch, dist, ctx, info, w, args.steps, args.warmup, max(1, args.repeats))
    except Stopped as exc:
        leave(1, str(exc))
    traffic, traffic_src = spmv_traffic(args.workload, world)      # None for any (workload, N) without a committed PMC run
    t_spmv_ms = r["t_spmv_ms"]
    # the first SpMV gathers block rows of v (n words each) out of the whole block
    ceiling, ceiling_src = gather_ceiling(n * r["word"], r["rows_v"] * n * r["word"])

    out = {
        "metric": "nnz*n mod-p MAC/s",
        "value": r["value"],
        "unit": "MAC/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": r["elapsed"] / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u64" if r["word"] == 8 else "u32",
        "data": data,
        "config": {
            "workload": f"{w['desc']}: {info['nrows']}x{info['ncols']}, {info['nnz']} nnz, --prime {prime_name(p)} --n {n} "
                        f"{'--right' if right else '--left'}",
            "step": "one block-Lanczos iteration: 2 block SpMV + block_dot + semi_inverse + orthogonalize",
            "parallelism": "single GPU" if world == 1 else f"row-partition x{world} + RCCL all-gather/all-reduce",
            "matrix": (data if data != "synthetic" else
                       ("seeded synthetic WITH structure (extra workload)" if w.get("structured") else
                        "seeded synthetic, uniform columns (SURVEY 8(d)); real .mtx not on the box")),
            "timing": f"median of {len(r['times'])} regions of {args.steps} steps each, max over ranks per region",
        },
        "repeats": {"ms_per_step": [t / args.steps * 1e3 for t in r["times"]]},
        "roofline": {
            "kernel": "the first SpMV of each step (tmp = M^T v): k_spmv, or k_spmv_staged / k_spmv_panel where the slab's plan chose them",
            "bound": "hbm",
            "achieved": r["achieved"],
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": r["achieved"] / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            # the same launch measured in bytes that actually cross the L2 <-> fabric boundary
            "traffic_GBps": (traffic / (t_spmv_ms * 1e-3) / 1e9) if (traffic and t_spmv_ms) else None,
            "traffic_frac_of_peak": (traffic / (t_spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (traffic and t_spmv_ms) else None,
            "alg_bytes_per_launch": r["alg_bytes"],
            "ms_per_launch": t_spmv_ms,
            # what this kernel is actually limited by: one 128-byte line fill per gathered block row;
            # the micro-benchmarked ceiling for random block-row gathers on MI355X is ~55 G rows/s whatever the row size
            # up to 128 B, the allocation kind or the load policy (profiles/r02_ubench2_*) = 7.0 TB/s of line traffic
            "gathers_per_s": r["nnz1"] / (t_spmv_ms * 1e-3) if t_spmv_ms else None,
            "gather_ceiling_per_s": ceiling,
            "gather_ceiling_source": ceiling_src,
        },
        "host": host_cpu(),
        "device_ms_per_step": r["device_ms_per_step"],
        "kernels": r["kernels"],
        "setup_s": {"generate_or_load": t_gen, "csr_upload_init": t_setup,
                    "note": "rank 0 prepares once and shares the result through an mmapped cache file when N > 1"},
        "renumbering": r["renumbering"],
        "lds_panel": r["lds_panel"],
    }
    macs_per_step, rows_v, rows_t = r["macs_per_step"], r["rows_v"], r["rows_t"]

    # ---- several ranks (or the exchange forced on one): what RCCL saw, how the exchange was cut, what each collective
    # cost and how much of it the products hid -- enough to fit the piece model (t0 per call, gather rate) from ONE run
    if dist is not None:
        seen, me = ctx.comm_info()
        prof, word = r["prof"], r["word"]
        # padded slab rows of each side = the largest rank's share (ncclAllGather moves equal counts)
        pad = torch.tensor(list(r["loc_rows"]), dtype=torch.int64)
        dist.all_reduce(pad, op=dist.ReduceOp.MAX)
        stride_v, stride_t = int(pad[0]), int(pad[1])
        seen_t = torch.tensor([seen], dtype=torch.int64)
        dist.all_reduce(seen_t, op=dist.ReduceOp.MIN)
        nn2 = 2 * ctx_width(ctx, n) ** 2 * 8
        per_call = {"allgather_v": stride_v * n * word * world, "allgather_tmp": stride_t * n * word * world,
                    "allreduce": nn2, "reduce_scatter": min(stride_v, stride_t) * n * 8 * world}       # (the short side's block)
        factor = {"allgather_v": (world - 1) / world, "allgather_tmp": (world - 1) / world,
                  "allreduce": 2 * (world - 1) / world, "reduce_scatter": (world - 1) / world}
        coll, exch_ms = {}, 0.0
        for name in ("allgather_v", "allgather_tmp", "allreduce", "reduce_scatter"):
            v_ = prof.get(name)
            if not v_ or not v_["launches"]:
                continue
            calls = v_["launches"] / args.steps
            ms_step = v_["ms_total"] / args.steps
            exch_ms += ms_step
            ent = {"calls_per_step": calls, "ms_per_step": ms_step, "ms_per_call": v_["ms_total"] / v_["launches"]}
            total = per_call[name]
            if name.startswith("allgather"):
                total = total / max(calls, 1)         # K pieces per step: each call moves 1/K of the block
            if total:
                ent["bytes_per_call_all_ranks"] = total
                ent["alg_GBps"] = total / (ent["ms_per_call"] * 1e-3) / 1e9
                ent["bus_GBps"] = ent["alg_GBps"] * factor[name]
            coll[name] = ent
        comp_ms = sum(prof[k_]["ms_total"] for k_ in ("spmv1", "spmv2", "block_dot", "semi_inverse", "orthogonalize")) / args.steps
        step_ms = r["device_ms_per_step"]
        exposed = max(0.0, step_ms - comp_ms)
        out["multi_gpu"] = {
            "rccl_ranks_seen": int(seen_t[0]), "rccl_rank_of_rank0": me,
            "pieces": {"spmv1": ctx.exchange_pieces(not right), "spmv2": ctx.exchange_pieces(right)},
            "short_side": {"spmv1": ctx.short_side(not right), "spmv2": ctx.short_side(right)},
            "collectives": coll,
            "note": "HIP-event spans on rank 0: collectives on the exchange stream (all-gathers) or the compute stream "
                    "(all-reduce, reduce-scatter); compute = the five kernel classes on the compute stream",
            "exchange_ms_per_step": exch_ms, "compute_ms_per_step": comp_ms, "device_ms_per_step": step_ms,
            "exposed_exchange_ms_per_step": exposed, "hidden_exchange_ms_per_step": max(0.0, exch_ms - exposed),
        }

    # ---- CPU baseline on this box's host cores: bounded sample of the same workload ------------
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        ncpu = os.cpu_count() or 1
        thread_sets = [args.cpu_threads] if args.cpu_threads else sorted({min(16, ncpu), ncpu})
        Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
        pair = orc.CsrPair(Mo)                       # built once, outside the timed sample (as the GPU's CSR slabs are)
        big = max(rows_v, rows_t) * n
        runs, v = [], None
        for threads in thread_sets:
            v = orc.init_v(rows_v, n, p)
            tmp, Av, pb = np.zeros(big, np.uint64), np.zeros(rows_v * n, np.uint64), np.zeros(rows_v * n, np.uint64)
            its, t_cpu, budget = 0, 0.0, args.cpu_seconds / len(thread_sets)
            while True:
                t0 = time.perf_counter()
                pair.iteration(n, p, right, v, tmp, Av, pb, threads)
                dt = time.perf_counter() - t0
                its += 1
                t_cpu += dt
                if t_cpu + dt > budget or its >= args.steps:
                    break
            runs.append(dict(cores=threads, value=macs_per_step * its / t_cpu, s_per_iteration=t_cpu / its, iterations=its))
        pair.close()
        best = max(runs, key=lambda q: q["value"])
        out["cpu_baseline"] = {
            "value": best["value"],
            "unit": "MAC/s",
            "cores": best["cores"],
            "kind": "port",
            "sample": f"{best['iterations']} full iteration(s) of the same workload with the oracle's by-rows OpenMP kernels "
                      f"(CSR built once, 128-bit sums; restating openMP/lanczos_modp.c with the output rows as the parallel "
                      f"loop), {best['s_per_iteration'] * best['iterations']:.1f} s on {ncpu} host CPUs "
                      f"({host_cpu()['model']}); the better of "
                      f"{[q['cores'] for q in runs]} threads",
            "s_per_iteration": best["s_per_iteration"],
            "runs": runs,
        }
        # the CPU's iterations must equal the GPU's (same seed, same matrix): compare after the last sample
        its = runs[-1]["iterations"]
        chk = blz.Context(p, n, device=local_rank)
        chk.set_matrix(M, right)
        chk.init_v()
        chk.iterate(its)
        same = bool(np.array_equal(chk.get_block(blz.V), v))
        chk.close()
        out["cpu_baseline"]["gpu_equals_cpu_after_sample"] = same
        if not same:
            leave(1, "GPU and CPU baseline disagree after the sampled iterations")

    # ---- the reference's own OpenMP program (oracle/_ref, compiled from its sources) on this box's host cores.
    # It cannot run the benchmark's configuration (p is capped at 2^30-35, u32 words, its u64 sums overflow on large
    # values, and it only reads files), so it gets a 1/4-scale all-ones sample of the same shape and density with
    # the largest prime it accepts.  Reported next to cpu_baseline, which is the same-workload port.
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "lanczos_modp_omp_ref")
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and args.ref_iterations > 0 and os.path.exists(ref_exe):
        import re
        import subprocess
        import tempfile
        p_ref, scale = 1073741789, 4
        S = blz.Matrix.synth(max(w["rows"] // scale, 64), max(w["cols"] // scale, 64), max(w["nnz"] // scale, 64), w["seed"], p_ref,
                             pattern=True)
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "sample.mtx")
            S.save(path)
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_STACKSIZE="1G")
            cmd = [ref_exe, "--matrix", path, "--prime", str(p_ref), "--n", str(n), "--stop-after", str(args.ref_iterations)]
            if right:
                cmd.append("--right")
            try:
                rr = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=td, timeout=240)
                mt = re.search(r"Terminated in ([0-9.]+)s after (\d+) iterations", rr.stdout)
                if rr.returncode == 0 and mt and float(mt.group(1)) > 0:
                    t_ref, k_ref = float(mt.group(1)), int(mt.group(2))
                    out["cpu_reference"] = {
                        "value": 2 * S.nnz * n * k_ref / t_ref, "unit": "MAC/s", "cores": threads, "kind": "reference",
                        "sample": f"openMP/lanczos_modp.c (unmodified, oracle/_ref) --stop-after {k_ref} on a 1/{scale}-scale all-ones "
                                  f"sample of the same shape ({S.nrows}x{S.ncols}, {S.nnz} nnz), --prime {p_ref} --n {n}: "
                                  f"{t_ref:.1f} s main loop",
                        "s_per_iteration": t_ref / max(k_ref, 1)}
                else:
                    out["cpu_reference"] = {"error": (rr.stderr or rr.stdout)[-300:]}
            except Exception as exc:   # the reference is a reported extra, never a reason to lose the bench line
                out["cpu_reference"] = {"error": repr(exc)}

    # ---- several ranks (or the exchange code forced on one): the sharded run must hold the same block as one GPU
    # solving the whole system.  Every rank sums the words of its slab of v; rank 0 repeats the same number of
    # iterations on a plain single-GPU context (tens of milliseconds) and compares the totals mod 2^64.  The verdict is
    # broadcast: on a mismatch the line carries value = null and EVERY rank exits non-zero.
    verdict = 1
    if dist is not None:
        iters_done = ctx.iterations
        mine = 0 if per_rank else int(ctx.get_block(blz.V).sum(dtype=np.uint64))          # rows of other ranks are left at zero
        tot = torch.tensor([mine & 0x7FFFFFFF, (mine >> 31) & 0x7FFFFFFF, mine >> 62], dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        its_t = torch.tensor([iters_done], dtype=torch.int64)
        dist.all_reduce(its_t, op=dist.ReduceOp.MAX)
        if rank == 0 and per_rank:
            # nobody holds the whole matrix: no single-GPU solve to compare with.  What every rank CAN check after its iterations
            # are the reference's own in-loop invariants on the (replicated) n x n operands: vtAv, vtAAv, winv symmetric and
            # winv * vtAv * D = D (correctness_tests, sequential/lanczos_modp.c:532-557) -- they only hold if the all-gathers,
            # the all-reduce and every rank's products were right.
            A_, B_, Wi, d_ = (ctx.get_small(k_).astype(object) for k_ in (blz.VTAV, blz.VTAAV, blz.WINV, blz.D))
            A_, B_, Wi = A_.reshape(n, n), B_.reshape(n, n), Wi.reshape(n, n)
            D_ = np.diag(d_)
            ok = bool((A_ == A_.T).all() and (B_ == B_.T).all() and (Wi == Wi.T).all() and ((Wi.dot(A_.dot(D_))) % p == D_).all()
                      and A_.any())
            out["sharded_equals_single_gpu"] = {"equal": None, "iterations": iters_done, "invariants_hold": ok,
                                                "check": "the whole matrix is never made: the reference's in-loop invariants "
                                                         "(symmetry, winv*vtAv*D = D) on the replicated n x n operands instead"}
            verdict = 1 if ok else 0
        elif rank == 0:
            total = (int(tot[0]) + (int(tot[1]) << 31) + (int(tot[2]) << 62)) & ((1 << 64) - 1)
            saved = {k_: os.environ.pop(k_) for k_ in ("BLZ_FORCE_COMM",) if k_ in os.environ}
            one = blz.Context(p, n, device=local_rank)
            one.set_matrix(M, right)
            one.init_v()
            one.iterate(iters_done)
            want = int(one.get_block(blz.V).sum(dtype=np.uint64))
            same = bool(want == total and one.iterations == iters_done == int(its_t[0]))
            one.close()
            os.environ.update(saved)
            out["sharded_equals_single_gpu"] = {"equal": same, "iterations": iters_done,
                                                "check": "sum of the words of v mod 2^64, all ranks, vs one GPU on the whole matrix"}
            verdict = 1 if same else 0
        vt = torch.tensor([verdict], dtype=torch.int64)
        dist.broadcast(vt, src=0)
        verdict = int(vt.item())
        if not verdict and rank == 0:
            out["value"] = None
            out["invalid"] = "the sharded run and the single-GPU run of the same system disagree"

    ctx.close()
    del M

    # ---- the other single-GPU configs in the same line (default run at N = 1): relat9 shape (config 3) and relat8
    # shape (config 2), same measurement, fewer regions
    want_extras = (args.extras == 1) or (args.extras < 0 and args.workload == "gl7d19" and world == 1 and dist is None)
    if want_extras and verdict:
        out["extra"] = {"workloads": {}}
        # relat9 / relat8 shapes = configs 3 / 2; synth5q = ONE GPU's quarter-scale share of config 5 (n = 16, 128-byte block
        # rows out of a 1.6 GB operand: the shape 91 % of whose iteration is the SpMV); nfs = the structured extra workload
        for name in ("relat9", "relat8", "synth5q", "nfs"):
            we = WORKLOADS[name]
            try:
                t0 = time.time()
                Me, data_e = make_matrix(blz, we, we["prime"])
                ce = blz.Context(we["prime"], we["n"], device=local_rank)
                ce.set_matrix(Me, we["right"])
                ce.init_v()
                t_set = time.time() - t0
                steps_e = min(args.steps, 10) if Me.nnz > 200000000 else args.steps      # 25 ms per step there
                re_ = measure(blz, torch, None, ce, dict(nnz=Me.nnz, pattern=bool((Me.x == 1).all())), we, steps_e, args.warmup, 3)
                ce.close()
                tr_e, tr_src = spmv_traffic(name, 1)
                ceil_e, ceil_src = gather_ceiling(we["n"] * re_["word"], re_["rows_v"] * we["n"] * re_["word"])
                out["extra"]["workloads"][name] = {
                    "workload": f"{we['desc']}: {Me.nrows}x{Me.ncols}, {Me.nnz} nnz, --prime {prime_name(we['prime'])} --n {we['n']} "
                                f"{'--right' if we['right'] else '--left'}",
                    "data": data_e, "value": re_["value"], "unit": "MAC/s", "steps": steps_e, "ms_per_step": re_["elapsed"] / steps_e * 1e3,
                    "roofline_frac": re_["achieved"] / HBM_PEAK_GBPS, "spmv1_GBps": re_["achieved"],
                    "spmv1_alg_bytes": re_["alg_bytes"], "spmv1_traffic": tr_e, "spmv1_traffic_source": tr_src,
                    "spmv1_ms": re_["t_spmv_ms"], "gathers_per_s": re_["nnz1"] / (re_["t_spmv_ms"] * 1e-3),
                    "spmv2_gathers_per_s": (Me.nnz / (re_["kernels"]["spmv2"]["ms_mean"] * 1e-3)
                                            if re_["kernels"]["spmv2"]["ms_mean"] else None),
                    "gather_ceiling_per_s": ceil_e, "gather_ceiling_source": ceil_src,
                    "kernels_ms": {k_: v_["ms_mean"] for k_, v_ in re_["kernels"].items() if v_["ms_mean"]},
                    "setup_s": t_set,
                    "ceiling_args": (we["n"] * re_["word"], re_["rows_v"] * we["n"] * re_["word"], re_["nnz1"] / max(re_["rows_t"], 1)),
                }
                del Me
            except Exception as exc:    # an extra never costs the headline line
                out["extra"]["workloads"][name] = {"error": repr(exc)}

    want_live = (args.live_traffic == 1) or (args.live_traffic < 0 and args.workload == "gl7d19" and world == 1 and dist is None
                                             and args.cpu_seconds > 0 and not under_profiler())
    # ---- the gather ceiling of this box, by the bare loop, for the headline product and for each extra one
    if want_live and verdict and rank == 0 and world == 1:
        rf = out["roofline"]
        lc = live_gather_ceiling(n * r["word"], r["rows_v"] * n * r["word"], r["nnz1"] / max(r["rows_t"], 1))
        rf["gather_ceiling_live"] = lc
        if "bare_8B_per_lane" in lc and rf["gathers_per_s"]:
            top = max(lc["bare_8B_per_lane"], lc["bare_16B_per_lane"])
            rf["gathers_frac_of_live_ceiling"] = rf["gathers_per_s"] / top
            rf["gathers_frac_of_live_ceiling_with_output_rows"] = rf["gathers_per_s"] / max(lc["with_output_rows_8B_per_lane"],
                                                                                           lc["with_output_rows_16B_per_lane"])
        for name, rec in out.get("extra", {}).get("workloads", {}).items():
            if "error" in rec or "ceiling_args" not in rec:
                continue
            le = live_gather_ceiling(*rec.pop("ceiling_args"))
            rec["gather_ceiling_live"] = le
            if "bare_8B_per_lane" in le:
                top = max(le["bare_8B_per_lane"], le["bare_16B_per_lane"])
                rec["gathers_frac_of_live_ceiling"] = rec["gathers_per_s"] / top

    # ---- roofline.traffic from this run's own PMC passes (the last GPU work of the run); the committed figure stays beside it
    if want_live and verdict and rank == 0 and world == 1:
        lt = live_traffic(args.workload)
        rf = out["roofline"]
        rf["traffic_committed"], rf["traffic_committed_source"] = rf["traffic"], rf["traffic_source"]
        if "traffic" in lt:
            rf["traffic"] = lt["traffic"]
            rf["traffic_source"] = ("live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this run, "
                                    f"{lt['launches']} launches, 2 x FETCH_SIZE + WRITE_SIZE (gfx950: 64 B counted per 128-byte fill)")
            rf["traffic_live"] = lt
            if t_spmv_ms:
                rf["traffic_GBps"] = lt["traffic"] / (t_spmv_ms * 1e-3) / 1e9
                rf["traffic_frac_of_peak"] = rf["traffic_GBps"] / HBM_PEAK_GBPS
        else:
            rf["traffic_live"] = lt         # why not; the committed figure stands
        # the same two passes for each extra workload, while every pass before it ended by itself
        for name, rec in (out.get("extra", {}).get("workloads", {}).items() if "traffic" in lt else ()):
            if "error" in rec:
                continue
            if time.time() - t_main > LIVE_BUDGET_S:
                # the default run has to end within minutes also on a box where the first `import torch` took two of them:
                # the extra workloads keep their committed figures then (the headline's passes have been made)
                rec["spmv1_traffic_live"] = {"skipped": f"the run had used {time.time() - t_main:.0f} s (budget {LIVE_BUDGET_S} s)"}
                continue
            le = live_traffic(name)
            rec["spmv1_traffic_committed"], rec["spmv1_traffic_committed_source"] = rec["spmv1_traffic"], rec["spmv1_traffic_source"]
            rec["spmv1_traffic_live"] = le
            if "traffic" not in le:
                break
            rec["spmv1_traffic"] = le["traffic"]
            rec["spmv1_traffic_source"] = f"live: rocprofv3 --pmc child passes of this run, {le['launches']} launches"

    for rec in out.get("extra", {}).get("workloads", {}).values():
        rec.pop("ceiling_args", None)
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not verdict:
        sys.exit(1)


if __name__ == "__main__":
    main()
