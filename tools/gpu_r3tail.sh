#!/bin/bash
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3tail
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "each_kernel or trajectory or random_system or outlier or prime_classes or every_block_width or degenerate" > "$out/pytest.log" 2>&1 || { tail -30 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
for wl in relat8 gl7d19 nfs relat9; do
	timeout -k 10 300 python3 bench.py --workload $wl --steps 20 --warmup 3 --repeats 5 --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/$wl.json" 2> "$out/$wl.err" || { echo "$wl failed"; tail -5 "$out/$wl.err"; exit 1; }
	python3 - "$out/$wl.json" "$wl" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print("%-8s" % sys.argv[2], "ms/step %.4f" % d["ms_per_step"], {a: round(k[a]["ms_mean"]*1e3,1) for a in ("spmv1","spmv2","block_dot","semi_inverse","orthogonalize") if k[a]["ms_mean"]}, flush=True)
PY
done
