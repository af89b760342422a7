"""bench.py as the driver starts it: `python bench.py --gpus N` with NO launcher around it must start its own ranks.

CPU-only checks (no GPU here): the parent never touches the GPU, builds the torch.distributed.run command for one process
per GPU on a free port of 127.0.0.1, and relays the ranks' exit code; --dry-run prints the command instead of running it.
The run itself on two real devices is tests/test_gpu_multirank.py::test_bench_starts_its_own_ranks (skips below 2 GPUs).
mpi/lanczos_modp.c is started by mpiexec; this is the same hand-over for one process per GPU.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(*args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def test_dry_run_prints_the_launcher_command_for_several_gpus():
    r = run("--gpus", "4", "--steps", "7", "--warmup", "2", "--dry-run")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    cmd = d["cmd"]
    assert d["dry_run"] is True and d["n_gpus"] == 4
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 < port < 65536
    at = cmd.index(BENCH)
    assert cmd[at + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]          # the ranks get the same flags, minus --dry-run
    # the port is one the kernel just handed out as free (a fixed one made concurrent suites on one box collide)
    def port_of(c):
        return c[c.index("--master-port") + 1]
    ports = {port_of(json.loads(run("--gpus", "4", "--dry-run").stdout)["cmd"]) for _ in range(3)} | {str(port)}
    assert len(ports) >= 2


def test_dry_run_with_one_gpu_and_under_a_launcher_stays_in_process():
    d = json.loads(run("--dry-run", "--workload", "tiny").stdout)
    assert d["n_gpus"] == 1 and d["cmd"][1] == BENCH and "--dry-run" not in d["cmd"]
    # ranks started by a launcher (WORLD_SIZE set) must not start ranks of their own
    d = json.loads(run("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}).stdout)
    assert d["n_gpus"] == 2 and "torch.distributed.run" not in d["cmd"]


def test_self_launch_relays_the_exit_code_of_its_ranks():
    """No GPU in this container: the two ranks stop with "no GPU visible" and the parent must come back non-zero with no
    result line -- not hang, not print a half-made line, not swallow the failure."""
    import pytest
    sys.path.insert(0, os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd", "python"))
    import blz
    if blz.device_count() > 0:
        pytest.skip("a GPU is visible: the run itself is covered by tests/test_gpu_multirank.py")
    r = run("--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "no GPU visible" in r.stderr or "ChildFailedError" in r.stderr


def test_the_roofline_kernel_is_picked_by_its_template_arguments(monkeypatch):
    """roofline.traffic is the PMC figure of the plain SpMV (no fused block_dot): the fourth template argument of the staged /
    panel kernels decides, not any `false` in the name (the later arguments have some too)."""
    sys.path.insert(0, ROOT)
    import bench
    yes = ["void k_spmv<unsigned long, 8, 61, false>(DevCsr, unsigned long const*)", "k_spmv<unsigned int, 4, 31>",
           "k_spmv_staged<unsigned long, 8, 61, false, 0, 8, false, 2>", "void k_spmv_panel<unsigned long, 8, 61, false, 1>(int)"]
    no = ["k_spmv_dot<unsigned long, 61, 8>", "k_spmv_staged<unsigned long, 16, 61, true, 0, 8, false, 1>",
          "k_spmv_panel<unsigned long, 8, 61, true, 1>", "k_ortho_mfma<8, false>", "k_dot_finalize"]
    assert all(bench.first_spmv_kernel(k) for k in yes) and not any(bench.first_spmv_kernel(k) for k in no)
    # every committed traffic file names exactly one such kernel
    for wl in ("gl7d19", "relat9", "relat8", "synth5q"):
        d = json.load(open(os.path.join(ROOT, "profiles", f"traffic_{wl}_n1.json")))
        assert sum(bench.first_spmv_kernel(k) for k in d) == 1, wl
        assert bench.spmv_traffic(wl, 1)[0] > 0
    # a run that is itself being profiled does not start profiler passes of its own
    for k in [k for k in os.environ if k.startswith("ROCPROF")]:
        monkeypatch.delenv(k)
    monkeypatch.setenv("LD_PRELOAD", "")
    assert not bench.under_profiler()
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    assert bench.under_profiler()


def test_the_ceiling_program_checks_its_arguments_before_it_touches_the_gpu():
    """tools/gather_ceiling (built by `make tools`, i.e. by __graft_entry__.build()) is started by bench.py as a child; bad
    arguments end it with 2 and no HIP call, and bench.py only asks for the row sizes it was written for."""
    exe = os.path.join(ROOT, "tools", "gather_ceiling")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "block-lanczos-algorithm-parallelization_amd"), "tools"])
    for bad in ([], ["64", "100"], ["48", "100", "10"], ["64", "0", "10"], ["128", "100", "0"], ["64", "100", "10", "5000"]):
        r = subprocess.run([exe] + bad, capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and not r.stdout, bad
    sys.path.insert(0, ROOT)
    import bench
    assert "error" in bench.live_gather_ceiling(16, 5e6, 100)


def test_a_failed_profiler_pass_is_reported_not_raised():
    """bench.py measures roofline.traffic with two rocprofv3 child passes of itself; a pass that fails (here: no GPU, so the child
    ends with an error) must come back as {"error": ...} -- the caller then keeps the committed figure and still prints its line --
    and no second pass is started after a failed one."""
    import pytest
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: the pass would succeed (tests/test_gpu_*.py cover that side)")
    sys.path.insert(0, ROOT)
    import bench
    got = bench.live_traffic("relat8", limit_s=120)
    assert set(got) == {"error"}, got
    assert got["error"].startswith(("FETCH_SIZE pass", "rocprofv3 not found")), got
