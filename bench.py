#!/usr/bin/env python3
"""bench.py -- the hot path of block-Lanczos mod p on MI355X, measured as BASELINE.json asks.

A "step" is one pass of the loop body of block_lanczos() (sequential/lanczos_modp.c:631-659):
two block SpMVs (tmp = M^T v, Av = M tmp), the n x n block inner products, the semi-inverse and the
row-local block update, all on the GPU with every operand resident in HBM before the clock starts.

  metric  = nnz*n mod-p MAC/s : (2 SpMV * nnz * n) * steps / time   (block products are NOT counted)
  value   = whole-job rate over all ranks (strong scaling: the matrix is fixed, rows are partitioned)
  roofline= the SpMV kernel (dominant): algorithmic bytes of SURVEY 8(d) / mean launch duration measured
            with HIP events on the solver's stream inside this run, against the 8 TB/s HBM3E peak
  cpu_baseline = the oracle's OpenMP kernels (restating openMP/lanczos_modp.c) timed on this box's
            host cores on a bounded sample of the SAME workload; a reported baseline, not the target.

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 under torch.distributed.run, one
process per GPU.  torch.distributed (gloo) carries only the control plane (RCCL id, barrier, max of the
times); the data path's collectives are RCCL calls made by libblz_hip.so on its own stream.
"""
import argparse
import json
import os
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
                   n=8, right=False, seed=0x474C3764, pattern=False),
    "relat9": dict(desc="JGD_Relat/relat9-shape synthetic", rows=12360060, cols=549336, nnz=38955420, prime=P61,
                   n=8, right=True, seed=0x52454C39, pattern=False),
    "relat8": dict(desc="JGD_Relat/relat8-shape synthetic", rows=345688, cols=12347, nnz=1334038, prime=P31,
                   n=4, right=False, seed=0x52454C38, pattern=False),
    # config 5 is 50M x 50M, 2e9 nnz over 8 GPUs; this is one GPU's quarter-scale share of it (same density 40/row,
    # all-ones pattern path, n=16, 128-byte block rows)
    "synth5q": dict(desc="config-5-shape synthetic at 1/4 linear scale (all-ones pattern)", rows=12500000, cols=12500000,
                    nnz=500000000, prime=P61, n=16, right=False, seed=0x53594E35, pattern=True),
    # config 5 at FULL size on one GPU (fits: ~42 GB of the 288 GB HBM); minutes of host-side set-up
    "synth5": dict(desc="config-5 synthetic (all-ones pattern), full size on ONE GPU", rows=50000000, cols=50000000,
                   nnz=2000000000, prime=P61, n=16, right=False, seed=0x53594E35, pattern=True),
    # EXTRA workload, not a BASELINE config and never the headline: a matrix WITH structure (heavy-tailed column degrees,
    # banded supports -- the shape of a sieve relation matrix), for what the uniform stand-ins cannot show: the LDS panel
    # of dense block rows and the per-XCD row ranges of the SpMV
    "nfs": dict(desc="EXTRA (not a BASELINE config): structured synthetic, 40 % of a row's entries ~1/(c+16), 30 % in a band "
                     "of 4096 columns, 30 % uniform", rows=2000000, cols=2000000, nnz=40000000, prime=P61, n=8, right=False,
                seed=0x4E465331, pattern=False, structured=dict(hot_pct=40, band_pct=30, band=4096)),
    "tiny": dict(desc="tiny synthetic (self-test)", rows=20000, cols=15000, nnz=200000, prime=P61,
                 n=8, right=False, seed=0x54494E59, pattern=False),
}
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s HBM3E peak


def prime_name(p):
    return {P61: "2^61-1", P31: "2^31-1"}.get(p, str(p))


def spmv_alg_bytes(nnz, rows_out, rows_in, n, w, pattern):
    """every array counted once -- (col_idx + val) per entry, row_ptr, X read, Y written."""
    return nnz * (4 + (0 if pattern else 4)) + 4 * (rows_out + 1) + (rows_in + rows_out) * n * w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="gl7d19", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--ref-iterations", type=int, default=50,
                    help="iterations of the unmodified reference OpenMP binary on a 1/4-scale sample (0 = skip)")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): RCCL and gloo print banners to fd 1 from native code, so fd 1 is
    # pointed at stderr for the duration of the run and the result is written to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    if world > 1 and "OMP_NUM_THREADS" not in os.environ:
        # the host-side set-up (generation, renumbering, CSR builds) is OpenMP code: share the cores between the ranks
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

    w = WORKLOADS[args.workload]
    p, n, right = w["prime"], w["n"], w["right"]
    t0 = time.time()
    if w.get("structured"):
        M = blz.Matrix.synth_structured(w["rows"], w["cols"], w["nnz"], w["seed"], p, pattern=w["pattern"], **w["structured"])
    else:
        M = blz.Matrix.synth(w["rows"], w["cols"], w["nnz"], w["seed"], p, pattern=w["pattern"])
    t_gen = time.time() - t0

    ctx = blz.Context(p, n, device=local_rank)
    if dist is not None:
        uid = [blz.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], rank, world)
    t0 = time.time()
    ctx.set_matrix(M, right, rank, world)
    ctx.init_v()
    ctx.sync()
    t_setup = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- warmup, then EXACTLY K steps between barrier + synchronize on both sides -------------
    if args.warmup > 0:
        ctx.iterate(args.warmup)
    ctx.sync()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    done, stopped, dev_ms = ctx.iterate(args.steps)
    ctx.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if stopped or done != args.steps:
        sys.exit(f"bench.py: the solve terminated after {done} of {args.steps} steps; use a larger workload")

    # ---- per-kernel durations with HIP events on the solver's stream, same K steps again ---------
    ctx.profile(True)
    ctx.iterate(args.steps)
    prof = ctx.profile_read()
    ctx.profile(False)

    word = ctx.word_bytes
    rows_v, rows_t = ctx.rows(blz.V), ctx.rows(blz.TMP)
    _, loc_v = ctx.local_rows(blz.V)
    _, loc_t = ctx.local_rows(blz.TMP)
    # per-rank algorithmic bytes of the two SpMV launches (X is read whole by every rank)
    nnz1, nnz2 = ctx.local_nnz(not right), ctx.local_nnz(right)   # tmp = (right ? M : M^T) v, then Av = the other one
    pattern = bool((M.x == 1).all())
    bytes1 = spmv_alg_bytes(nnz1, loc_t, rows_v, n, word, pattern)
    bytes2 = spmv_alg_bytes(nnz2, loc_v, rows_t, n, word, pattern)
    # roofline kernel = k_spmv, the first SpMV of every step (the second one carries block_dot as its epilogue and is
    # listed under "kernels"); HIP-event spans on the solver's stream, collected inside blz_iterate
    # (with several ranks a product is cut into column pieces that overlap the exchange: time per product, not per piece)
    t_spmv_ms = prof["spmv1"]["ms_total"] / max(args.steps, 1)
    alg_bytes = bytes1
    achieved = alg_bytes / (t_spmv_ms * 1e-3) / 1e9
    kernels = {k: dict(ms_mean=(v["ms_total"] / args.steps) if v["launches"] else None, launches=v["launches"])
               for k, v in prof.items()}     # ms_mean = per step (a step may issue several launches of one class)
    kernels["spmv1"]["alg_bytes"] = bytes1
    kernels["spmv2"]["alg_bytes"] = bytes2
    for k_ in ("spmv1", "spmv2"):
        if kernels[k_]["ms_mean"]:
            kernels[k_]["alg_GBps"] = kernels[k_]["alg_bytes"] / (kernels[k_]["ms_mean"] * 1e-3) / 1e9
    kernels["block_dot"]["alg_bytes"] = 2 * loc_v * n * word
    kernels["orthogonalize"]["alg_bytes"] = 5 * loc_v * n * word

    # HBM-side traffic of the SpMV kernel from the committed rocprofv3 PMC passes of this same command
    # (tools/gpu_profile.sh: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs; bench.py cannot collect PMCs itself).
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}_n{world}.json")
    if os.path.exists(tpath):
        for name, rec in json.load(open(tpath)).items():
            if name.startswith("k_spmv<") and "FETCH_SIZE_bytes_per_launch" in rec and "WRITE_SIZE_bytes_per_launch" in rec:
                # gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE = TCC_EA0_RDREQ x 64 B although every request
                # is a 128-byte line fill, so it is doubled; WRITE_SIZE is exact.  The separate pass in
                # profiles/r01_v6_gl7d19_pmc_l2_fabric.txt confirms it for this kernel: all of its 35.7 M L2->fabric
                # read requests per launch are counted under TCC_EA0_RDREQ_128B -- a gathered 64-byte block row costs a
                # whole 128-byte line.
                traffic = 2 * rec["FETCH_SIZE_bytes_per_launch"] + rec["WRITE_SIZE_bytes_per_launch"]
                traffic_src = os.path.relpath(tpath, ROOT)

    macs_per_step = 2 * M.nnz * n
    value = macs_per_step * args.steps / elapsed

    out = {
        "metric": "nnz*n mod-p MAC/s",
        "value": value,
        "unit": "MAC/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u64" if word == 8 else "u32",
        "data": "synthetic",
        "config": {
            "workload": f"{w['desc']}: {w['rows']}x{w['cols']}, {M.nnz} nnz, --prime {prime_name(p)} --n {n} "
                        f"{'--right' if right else '--left'}",
            "step": "one block-Lanczos iteration: 2 block SpMV + block_dot + semi_inverse + orthogonalize",
            "parallelism": "single GPU" if world == 1 else f"row-partition x{world} + RCCL all-gather/all-reduce",
            "matrix": ("seeded synthetic WITH structure (extra workload)" if w.get("structured") else
                       "seeded synthetic, uniform columns (SURVEY 8(d)); real .mtx not on the box"),
        },
        "roofline": {
            "kernel": "k_spmv (first SpMV of each step: tmp = M^T v)",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            # the same launch measured in bytes that actually cross the L2 <-> fabric boundary
            "traffic_GBps": (traffic / (t_spmv_ms * 1e-3) / 1e9) if (traffic and t_spmv_ms) else None,
            "traffic_frac_of_peak": (traffic / (t_spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (traffic and t_spmv_ms) else None,
            "alg_bytes_per_launch": alg_bytes,
            "ms_per_launch": t_spmv_ms,
            # what this kernel is actually limited by: one 128-byte line fill per entry; the
            # micro-benchmarked ceiling for random block-row gathers on MI355X is 54.7 G rows/s whatever the row
            # size up to 128 B (profiles/r01_ubench_alu_and_gather.txt) = 7.0 TB/s of line traffic
            "gathers_per_s": nnz1 / (t_spmv_ms * 1e-3) if t_spmv_ms else None,
            "gather_ceiling_per_s": 54.7e9,
        },
        "device_ms_per_step": dev_ms / args.steps,
        "kernels": kernels,
        "setup_s": {"generate": t_gen, "csr_upload_init": t_setup},
        # block rows of each product's operand kept in LDS and the share of the entries they serve (0 on uniform matrices)
        "renumbering": dict(zip(("lines_per_entry", "order"), (lambda l, k: (dict(zip(("M", "Mt"), l)), ("smallest", "file", "mean")[k]))(*ctx.locality()))),
        "lds_panel": {"spmv1": dict(zip(("rows", "share"), ctx.panel_rows(not right))),
                      "spmv2": dict(zip(("rows", "share"), ctx.panel_rows(right)))},
    }

    # ---- CPU baseline on this box's host cores: bounded sample of the same workload ------------
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
        big = max(rows_v, rows_t) * n
        v = orc.init_v(rows_v, n, p)
        tmp, Av, pb = np.zeros(big, np.uint64), np.zeros(rows_v * n, np.uint64), np.zeros(rows_v * n, np.uint64)
        its, t_cpu = 0, 0.0
        while True:
            t0 = time.perf_counter()
            orc.iteration_omp(Mo, n, p, right, v, tmp, Av, pb, threads)
            dt = time.perf_counter() - t0
            its += 1
            t_cpu += dt
            if t_cpu + dt > args.cpu_seconds or its >= args.steps:
                break
        out["cpu_baseline"] = {
            "value": macs_per_step * its / t_cpu,
            "unit": "MAC/s",
            "cores": threads,
            "kind": "port",
            "sample": f"{its} full iteration(s) of the same workload with the oracle's OpenMP kernels "
                      f"(strategy of openMP/lanczos_modp.c, 128-bit sums), {t_cpu:.1f} s on {os.cpu_count()} host CPUs",
            "s_per_iteration": t_cpu / its,
        }
        # the CPU's first iteration must equal the GPU's first iteration (same seed, same matrix)
        chk = blz.Context(p, n, device=local_rank)
        chk.set_matrix(M, right)
        chk.init_v()
        chk.iterate(its)
        same = bool(np.array_equal(chk.get_block(blz.V), v))
        chk.close()
        out["cpu_baseline"]["gpu_equals_cpu_after_sample"] = same
        if not same:
            sys.exit("bench.py: GPU and CPU baseline disagree after the sampled iterations")

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
                r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=td, timeout=240)
                mt = re.search(r"Terminated in ([0-9.]+)s after (\d+) iterations", r.stdout)
                if r.returncode == 0 and mt and float(mt.group(1)) > 0:
                    t_ref, k_ref = float(mt.group(1)), int(mt.group(2))
                    out["cpu_reference"] = {
                        "value": 2 * S.nnz * n * k_ref / t_ref, "unit": "MAC/s", "cores": threads, "kind": "reference",
                        "sample": f"openMP/lanczos_modp.c (unmodified, oracle/_ref) --stop-after {k_ref} on a 1/{scale}-scale all-ones "
                                  f"sample of the same shape ({S.nrows}x{S.ncols}, {S.nnz} nnz), --prime {p_ref} --n {n}: "
                                  f"{t_ref:.1f} s main loop",
                        "s_per_iteration": t_ref / max(k_ref, 1)}
                else:
                    out["cpu_reference"] = {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as exc:   # the reference is a reported extra, never a reason to lose the bench line
                out["cpu_reference"] = {"error": repr(exc)}

    # ---- several ranks (or the exchange code forced on one): the sharded run must hold the same block as one GPU
    # solving the whole system.  Every rank sums the words of its slab of v; rank 0 repeats the same number of
    # iterations on a plain single-GPU context (tens of milliseconds) and compares the totals mod 2^64.
    if dist is not None:
        iters_done = ctx.iterations
        mine = int(ctx.get_block(blz.V).sum(dtype=np.uint64))          # rows of other ranks are left at zero
        tot = torch.tensor([mine & 0x7FFFFFFF, (mine >> 31) & 0x7FFFFFFF, mine >> 62], dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        its_t = torch.tensor([iters_done], dtype=torch.int64)
        dist.all_reduce(its_t, op=dist.ReduceOp.MAX)
        if rank == 0:
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
            if not same:
                print("bench.py: WARNING the sharded run and the single-GPU run disagree", file=sys.stderr)

    ctx.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
