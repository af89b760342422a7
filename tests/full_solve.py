#!/usr/bin/env python3
"""Whole solve of a bench workload on one GPU, result verified on the host with the oracle's SpMV (test infrastructure).
Usage: python tests/full_solve.py gl7d19 [--right]
       python tests/full_solve.py relat9 --cli     (through files and the lanczos_modp / checker_modp executables)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ is where oracle users live
sys.path[:0] = [os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python"), os.path.join(ROOT, "oracle"), ROOT]
import blz, bench, oracle as orc
name = sys.argv[1]
w = dict(bench.WORKLOADS[name])
right = (("--right" in sys.argv) or w["right"]) and "--left" not in sys.argv
p, n = w["prime"], w["n"]
M = blz.Matrix.synth(w["rows"], w["cols"], w["nnz"], w["seed"], p, pattern=w["pattern"])
if "--cli" in sys.argv:
    # the drop-in path a user of the reference takes: .mtx in, lanczos_modp, kernel .mtx out, checker_modp
    import subprocess, tempfile
    lib = os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "lib")
    work = tempfile.mkdtemp(prefix="blz_full_")
    mpath, kpath = os.path.join(work, name + ".mtx"), os.path.join(work, "kernel.mtx")
    t0 = time.time(); M.save(mpath); t_save = time.time() - t0
    args = ["--matrix", mpath, "--prime", str(p), "--n", str(n)] + (["--right"] if right else [])
    t0 = time.time()
    r = subprocess.run([os.path.join(lib, "lanczos_modp")] + args + ["--output-file", kpath], capture_output=True, text=True)
    t_solve = time.time() - t0
    keep = [ln for ln in r.stdout.replace("\r", "\n").split("\n") if ln.strip().startswith(("- OK", "- KO", "Terminated", "Final", "Saving", "Expecting"))]
    t0 = time.time()
    c = subprocess.run([os.path.join(lib, "checker_modp")] + args[:4] + ["--kernel", kpath] + (["--right"] if right else []),
                       capture_output=True, text=True)
    t_check = time.time() - t0
    print(json.dumps(dict(workload=name, right=right, rows=M.nrows, cols=M.ncols, nnz=M.nnz, n=n, prime=str(p),
                          matrix_file_MB=round(os.path.getsize(mpath) / 1e6), kernel_file_MB=round(os.path.getsize(kpath) / 1e6) if os.path.exists(kpath) else None,
                          write_matrix_s=round(t_save, 1), lanczos_modp_exit=r.returncode, lanczos_modp_wall_s=round(t_solve, 1),
                          lanczos_modp_lines=keep, stderr_tail=r.stderr.strip().split("\n")[-6:],
                          checker_modp_exit=c.returncode, checker_modp_says=c.stdout.strip().split("\n")[-1:], checker_wall_s=round(t_check, 1))))
    for f in (mpath, kpath):
        if os.path.exists(f):
            os.remove(f)
    os.rmdir(work)
    sys.exit(0 if r.returncode == 0 and c.returncode == 0 else 1)
ctx = blz.Context(p, n)
t0 = time.time()
ctx.set_matrix(M, right)
ctx.init_v()
t_setup = time.time() - t0
t0 = time.time()
dev_ms, last = 0.0, t0
while True:
    done, stopped, ms = ctx.iterate(2048)
    dev_ms += ms
    if time.time() - last > 30:
        print(f"  iteration {ctx.iterations}  ({dev_ms/1e3:.1f} s of device time)", flush=True)
        last = time.time()
    if stopped:
        break
wall = time.time() - t0
nz, zero = ctx.final_check()
v = ctx.get_block(blz.V)
Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
chk = orc.spmv_omp(Mo, v, not right, n, p, min(16, os.cpu_count() or 1))      # x^T M (left) or M x (right)
res = dict(workload=name, right=right, rows=M.nrows, cols=M.ncols, nnz=M.nnz, n=n, prime=str(p), iterations=ctx.iterations,
           setup_s=round(t_setup, 2), solve_wall_s=round(wall, 2), device_s=round(dev_ms / 1e3, 2),
           ms_per_iteration=round(dev_ms / max(ctx.iterations, 1), 4), final_check=dict(v_nonzero=nz, vtM_zero=zero),
           host_verification=dict(v_nonzero=bool(v.any()), product_is_zero=bool(not chk.any())),
           nonzero_columns=int((v.reshape(-1, n) != 0).any(axis=0).sum()))
print(json.dumps(res))
