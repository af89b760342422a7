#!/bin/bash
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3x
mkdir -p "$out"
cd "$root"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -x -q -k "local_matrix or trajectory or cache or each_kernel" > "$out/pytest.log" 2>&1 || { tail -30 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
for wl in band gl7d19 nfs; do
	timeout -k 10 300 python3 bench.py --workload $wl --steps 20 --warmup 3 --repeats 3 --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/$wl.json" 2> "$out/$wl.err" || { echo "$wl failed"; tail -5 "$out/$wl.err"; exit 1; }
	python3 - "$out/$wl.json" "$wl" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print("%-8s" % sys.argv[2], "ms/step %.4f" % d["ms_per_step"], {a: round(k[a]["ms_mean"]*1e3,1) for a in ("spmv1","spmv2","block_dot","semi_inverse","orthogonalize") if k[a]["ms_mean"]}, flush=True)
PY
done
