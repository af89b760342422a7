#!/bin/bash
# bench.py A/B lines only (no test suite): tools/gpu_ab2.sh <tag> "<env assignments A>" "<env assignments B>" workloads...
set -o pipefail
tag=$1; envA=$2; envB=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
for w in "$@"; do
	i=0
	for e in "$envA" "$envB"; do
		i=$((i + 1))
		env $e timeout -k 10 300 python bench.py --workload $w --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/bench_${w}_$i.json" 2> "$out/bench_${w}_$i.err" || { echo "bench $w [$e] failed"; tail -5 "$out/bench_${w}_$i.err"; exit 1; }
		python3 - "$out/bench_${w}_$i.json" "$e" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print(sys.argv[1].split("/")[-1], "[%s]"%sys.argv[2], "ms/step %.4f"%d["ms_per_step"], "frac %.4f"%d["roofline"]["frac"], {a:round(b["ms_mean"]*1e3,1) for a,b in k.items() if b.get("ms_mean")})
PY
	done
done
