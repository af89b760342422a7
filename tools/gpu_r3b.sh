#!/bin/bash
# round 3, GPU step b: parity of the dynamic-rows SpMV, its A/B on the config-5 quarter shape, the default bench line, kernel stats
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3b
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "dynamic_rows or staged_matrix" > "$out/pytest_dyn.log" 2>&1 || { tail -30 "$out/pytest_dyn.log"; exit 1; }
tail -3 "$out/pytest_dyn.log"
B="python3 bench.py --workload synth5q --steps 10 --warmup 2 --repeats 3 --cpu-seconds 0 --ref-iterations 0 --extras 0"
for v in "dyn0:BLZ_STAGE_DYN=0" "dyn1:BLZ_STAGE_DYN=1" "dyn1_cu4:BLZ_STAGE_DYN=1 BLZ_SPMV_BLOCKS_PER_CU=4" "dyn1_u4:BLZ_STAGE_DYN=1 BLZ_STAGE_U=4" "dyn1_tr6:BLZ_STAGE_DYN=1 BLZ_STAGE_TR=6" "dyn1_tr12:BLZ_STAGE_DYN=1 BLZ_STAGE_TR=12" "dyn0_cu4:BLZ_STAGE_DYN=0 BLZ_SPMV_BLOCKS_PER_CU=4"; do
	tag=${v%%:*}; envs=${v#*:}
	env $envs timeout -k 10 300 $B > "$out/s5q_$tag.json" 2> "$out/s5q_$tag.err" || { echo "$tag failed"; tail -5 "$out/s5q_$tag.err"; exit 1; }
	python3 - "$out/s5q_$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], {a: round(k[a]["ms_mean"]*1e3,1) for a in ("spmv1","spmv2","block_dot","semi_inverse","orthogonalize") if k[a]["ms_mean"]}, flush=True)
PY
done
timeout -k 10 900 python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err" || { echo "default bench failed"; tail -5 "$out/bench_default.err"; exit 1; }
python3 - "$out/bench_default.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("default ms/step %.4f value %.4g frac %.4f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"]))
print({a: round(v["ms_mean"]*1e3,1) for a,v in d["kernels"].items() if v.get("ms_mean")})
for n,e in d["extra"]["workloads"].items():
    print(n, {k: (round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k in ("ms_per_step","roofline_frac","gathers_per_s","spmv2_gathers_per_s","setup_s","error")}, e.get("kernels_ms"))
PY
for wl in gl7d19 relat8; do
	timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$wl" -- python3 bench.py --workload $wl --steps 20 --warmup 3 --repeats 2 --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/prof_$wl.json" 2> "$out/prof_$wl.err" || { echo "rocprof $wl failed"; tail -5 "$out/prof_$wl.err"; exit 1; }
	f=$(find "$out/prof_$wl" -name "*kernel_stats.csv" | head -1)
	cp "$f" "$out/${wl}_kernel_stats.csv"
	head -12 "$out/${wl}_kernel_stats.csv" | cut -c1-200
done
