#!/bin/bash
# round 3, GPU step i: head-of-kernel loads hoisted (relat8 shape), regression check on the big shapes, staged form on a band matrix
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3i
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "not dynamic_rows and not two_words and not staged_matrix" > "$out/pytest.log" 2>&1 || { tail -30 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
for wl in relat8 gl7d19 relat9 nfs; do
	timeout -k 10 300 python3 bench.py --workload $wl --steps 20 --warmup 3 --repeats 5 --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/$wl.json" 2> "$out/$wl.err" || { echo "$wl failed"; tail -5 "$out/$wl.err"; exit 1; }
	python3 - "$out/$wl.json" "$wl" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], [round(x,4) for x in d["repeats"]["ms_per_step"]], {a: round(k[a]["ms_mean"]*1e3,1) for a in ("spmv1","spmv2","block_dot","semi_inverse","orthogonalize") if k[a]["ms_mean"]}, flush=True)
PY
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_relat8" -- python3 bench.py --workload relat8 --steps 20 --warmup 3 --repeats 2 --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/prof_relat8.json" 2> "$out/prof_relat8.err" || { echo "rocprof failed"; tail -5 "$out/prof_relat8.err"; exit 1; }
cp "$(find "$out/prof_relat8" -name "*kernel_stats.csv" | head -1)" "$out/relat8_kernel_stats.csv"
python3 - "$out/relat8_kernel_stats.csv" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("  %-60s calls %4s avg %9.1f ns" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
PY
BLZ_STAGE_ALWAYS=1 timeout -k 10 300 python3 tools/exp_sweeps.py > "$out/exp_band_staged.txt" 2>&1 || { tail -5 "$out/exp_band_staged.txt"; exit 1; }
cat "$out/exp_band_staged.txt"
