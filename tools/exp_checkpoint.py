#!/usr/bin/env python3
"""What an asynchronous checkpoint costs the loop (run on the GPU box): the config-5 quarter shape (or any bench workload),
`iters` iterations without checkpoints, then the same with a snapshot every `every` seconds whose collection and file write
happen on a helper thread (what lanczos_modp --checkpoint does).  Usage: tools/exp_checkpoint.py [workload] [seconds] [every]"""
import json, os, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python"), ROOT]
import numpy as np
import blz, bench
name = sys.argv[1] if len(sys.argv) > 1 else "synth5q"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
every = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
w = bench.WORKLOADS[name]
p, n = w["prime"], w["n"]
M, _ = bench.make_matrix(blz, w, p)
ctx = blz.Context(p, n)
ctx.set_matrix(M, w["right"])
ctx.init_v()
ctx.iterate(3)
batch = max(1, int(0.25 / (ctx.iterate(4)[2] / 4 / 1e3)))      # ~0.25 s per blz_iterate call, as the CLI aims for


def run(with_ckpt):
    rows = ctx.rows(blz.V)
    v, pb = np.zeros(rows * n, np.uint64), np.zeros(rows * n, np.uint64)
    state = dict(busy=False, written=0, write_s=0.0)
    path = os.path.join(tempfile.gettempdir(), "exp_checkpoint.ckpt")

    def writer():
        t0 = time.time()
        _, _, its = ctx.snapshot_wait(v, pb)
        blz.checkpoint_save(path, p, n, w["right"], rows, its, v, pb)
        state["write_s"] += time.time() - t0
        state["written"] += 1
        state["busy"] = False

    th = None
    ctx.sync()
    t0 = last = time.time()
    its = 0
    while its < target:
        done, stopped, _ = ctx.iterate(batch)
        its += done
        assert not stopped
        if with_ckpt and not state["busy"] and time.time() - last >= every:
            if th is not None:
                th.join()
            ctx.snapshot_begin()
            state["busy"] = True
            th = threading.Thread(target=writer)
            th.start()
            last = time.time()
    ctx.sync()
    wall = time.time() - t0
    if th is not None:
        th.join()
    if os.path.exists(path):
        os.remove(path)
    return dict(iterations=its, wall_s=round(wall, 3), ms_per_iteration=round(wall / its * 1e3, 4), checkpoints=state["written"],
                writer_s=round(state["write_s"], 2))


target = max(batch, int(seconds / 0.25) * batch)      # the same number of iterations in both runs
plain = run(False)
ck = run(True)
print(json.dumps(dict(workload=name, n=n, block_MB=round(ctx.rows(blz.V) * n * 8 / 1e6), batch=batch, every_s=every, plain=plain,
                      with_checkpoints=ck, loss_pct=round((ck["ms_per_iteration"] / plain["ms_per_iteration"] - 1) * 100, 2))))
