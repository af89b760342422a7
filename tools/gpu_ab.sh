#!/bin/bash
# Run on the GPU box: GPU test suite, then bench.py A/B lines (staged stream on / off) for the single-GPU workloads.
# Usage: tools/gpu_ab.sh <tag> [workloads...]
set -o pipefail
tag=${1:-ab}; shift
wl=${@:-gl7d19 relat9}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1
rc=$?
tail -15 "$out/pytest.log"
[ $rc -eq 0 ] || exit $rc
for w in $wl; do
	for st in 0 1; do
		BLZ_NO_STAGE=$st timeout -k 10 300 python bench.py --workload $w --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/bench_${w}_nostage$st.json" 2> "$out/bench_${w}_nostage$st.err" || { echo "bench $w $st failed"; tail -5 "$out/bench_${w}_nostage$st.err"; exit 1; }
		python3 - "$out/bench_${w}_nostage$st.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print(sys.argv[1].split("/")[-1], "ms/step %.4f"%d["ms_per_step"], "frac %.4f"%d["roofline"]["frac"], {a:round(b["ms_mean"]*1e3,1) for a,b in k.items() if b.get("ms_mean")})
PY
	done
done
