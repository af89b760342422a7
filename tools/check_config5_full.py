#!/usr/bin/env python3
"""Config 5 at its OWN size on one MI355X (50 M x 50 M, 2e9 entries, all ones, n = 16, p = 2^61-1): one whole iteration on the
GPU against the oracle's by-rows OpenMP iteration on the host cores (word for word), then a few timed iterations.
Too heavy for the test-suite (about 100 GB of host memory, several minutes); run once per round, output kept under profiles/.
Usage: python tools/check_config5_full.py [workload]        (default synth5; synth5q for a rehearsal)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python"), os.path.join(ROOT, "oracle"), ROOT]
import numpy as np
import blz, bench, oracle as orc

name = sys.argv[1] if len(sys.argv) > 1 else "synth5"
w = bench.WORKLOADS[name]
p, n, right = w["prime"], w["n"], w["right"]
T0 = time.time()


def say(msg):
    print(f"[{time.time() - T0:7.1f} s] {msg}", flush=True)


say(f"generating {name}: {w['rows']} x {w['cols']}, {w['nnz']} entries")
M = blz.Matrix.synth(w["rows"], w["cols"], w["nnz"], w["seed"], p, pattern=w["pattern"])
say("matrix generated; GPU set-up (renumbering, CSR(M), CSR(M^T), upload)")
ctx = blz.Context(p, n)
t0 = time.time()
ctx.set_matrix(M, right)
ctx.init_v()
ctx.sync()
t_setup = time.time() - t0
say(f"set-up {t_setup:.1f} s; one iteration on the GPU")
nv, nt = ctx.rows(blz.V), ctx.rows(blz.TMP)
v0 = ctx.get_block(blz.V)
done, stopped, ms1 = ctx.iterate(1)
assert (done, stopped) == (1, False)
gv = ctx.get_block(blz.V)
gp = ctx.get_block(blz.P)
small = {k: ctx.get_small(c).tolist() for k, c in (("vtAv", blz.VTAV), ("d", blz.D))}
say(f"GPU iteration {ms1:.1f} ms; building the oracle's CSR pair on the host")
Mo = orc.Matrix(M.nrows, M.ncols, M.i, M.j, M.x)
pair = orc.CsrPair(Mo)
say("oracle iteration (by-rows OpenMP kernels)")
threads = min(32, os.cpu_count() or 1)
vv, tt = v0, np.zeros(max(nv, nt) * n, np.uint64)
aa, pp = np.zeros(nv * n, np.uint64), np.zeros(nv * n, np.uint64)
t0 = time.time()
npiv = pair.iteration(n, p, right, vv, tt, aa, pp, threads)
t_cpu = time.time() - t0
pair.close()
same_v, same_p = bool(np.array_equal(gv, vv)), bool(np.array_equal(gp, pp))
say(f"oracle iteration {t_cpu:.1f} s on {threads} threads; v equal: {same_v}, p equal: {same_p}")
del vv, tt, aa, pp, gv, gp, Mo
# a short timed run with the per-kernel spans
ctx.iterate(2)
ctx.profile(True)
steps = 5
_, _, ms = ctx.iterate(steps)
prof = ctx.profile_read()
res = dict(workload=name, rows=M.nrows, cols=M.ncols, nnz=M.nnz, n=n, prime=str(p), setup_s=round(t_setup, 1),
           gpu_equals_oracle_after_one_iteration=dict(v=same_v, p=same_p, npiv=int(npiv)), oracle_iteration_s=round(t_cpu, 1),
           oracle_threads=threads, first_gpu_iteration_ms=round(ms1, 2), ms_per_iteration=round(ms / steps, 3),
           mac_per_s=2 * M.nnz * n / (ms / steps * 1e-3),
           kernels_us={k: round(v_["ms_total"] / steps * 1e3, 1) for k, v_ in prof.items() if v_["launches"]},
           d=small["d"])
print(json.dumps(res), flush=True)
sys.exit(0 if same_v and same_p else 1)
