#!/bin/bash
# round 3, GPU step f: whole GPU suite with the fused finalize / Fermat-31 / pair-lane defaults, default bench line, kernel stats
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3f
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
tail -12 "$out/pytest.log"
timeout -k 10 900 python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err" || { echo "default bench failed"; tail -5 "$out/bench_default.err"; exit 1; }
python3 - "$out/bench_default.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("default ms/step %.4f value %.4g frac %.4f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"]))
print({a: round(v["ms_mean"]*1e3,1) for a,v in d["kernels"].items() if v.get("ms_mean")})
for n,e in d["extra"]["workloads"].items():
    print(n, {k: (round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k in ("ms_per_step","roofline_frac","gathers_per_s","spmv2_gathers_per_s","setup_s","error")}, {k: round(v*1e3,1) for k,v in (e.get("kernels_ms") or {}).items()})
PY
for wl in gl7d19 relat8; do
	timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$wl" -- python3 bench.py --workload $wl --steps 20 --warmup 3 --repeats 2 --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/prof_$wl.json" 2> "$out/prof_$wl.err" || { echo "rocprof $wl failed"; tail -5 "$out/prof_$wl.err"; exit 1; }
	f=$(find "$out/prof_$wl" -name "*kernel_stats.csv" | head -1)
	cp "$f" "$out/${wl}_kernel_stats.csv"
	head -8 "$out/${wl}_kernel_stats.csv" | cut -c1-60,150-260
done
