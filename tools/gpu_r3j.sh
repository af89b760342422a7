#!/bin/bash
# round 3, GPU step j: the locality-aware choice of the staged form; relat8 shape without the LDS panel
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3j
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "local_matrix or trajectory or two_words" > "$out/pytest.log" 2>&1 || { tail -30 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
timeout -k 10 300 python3 tools/exp_sweeps.py > "$out/exp_band_default.txt" 2>&1 || { tail -5 "$out/exp_band_default.txt"; exit 1; }
cat "$out/exp_band_default.txt"
for v in "panel:BLZ_NO_PANEL=0" "nopanel:BLZ_NO_PANEL=1" "nopanel_cu6:BLZ_NO_PANEL=1 BLZ_SPMV_BLOCKS_PER_CU=6"; do
	tag=${v%%:*}; envs=${v#*:}
	env $envs timeout -k 10 300 python3 bench.py --workload relat8 --steps 50 --warmup 5 --repeats 5 --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/relat8_$tag.json" 2> "$out/relat8_$tag.err" || { echo "$tag failed"; tail -5 "$out/relat8_$tag.err"; exit 1; }
	python3 - "$out/relat8_$tag.json" "relat8 $tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], [round(x,4) for x in d["repeats"]["ms_per_step"]], {a: round(k[a]["ms_mean"]*1e3,1) for a in ("spmv1","spmv2","block_dot","semi_inverse","orthogonalize") if k[a]["ms_mean"]}, flush=True)
PY
done
