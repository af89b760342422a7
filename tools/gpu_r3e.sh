#!/bin/bash
# round 3, GPU step e: the two-words-per-lane SpMV at n = 16: parity, then A/B on the config-5 quarter shape
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3e
mkdir -p "$out"
cd "$root"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "two_words or dynamic_rows or staged_matrix" > "$out/pytest.log" 2>&1 || { tail -30 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
B="python3 bench.py --workload synth5q --steps 10 --warmup 2 --repeats 3 --cpu-seconds 0 --ref-iterations 0 --extras 0"
for v in "pair0:BLZ_NO_PAIR=1" "pair1:BLZ_NO_PAIR=0" "pair1_cu4:BLZ_NO_PAIR=0 BLZ_SPMV_BLOCKS_PER_CU=4" "pair1_cu6:BLZ_NO_PAIR=0 BLZ_SPMV_BLOCKS_PER_CU=6" "pair1_rpg2:BLZ_NO_PAIR=0 BLZ_SPMV_BLOCKS_PER_CU=4 BLZ_STAGE_RPG=2" "pair0_b:BLZ_NO_PAIR=1" "pair1_b:BLZ_NO_PAIR=0"; do
	tag=${v%%:*}; envs=${v#*:}
	env $envs timeout -k 10 300 $B > "$out/s5q_$tag.json" 2> "$out/s5q_$tag.err" || { echo "$tag failed"; tail -5 "$out/s5q_$tag.err"; exit 1; }
	python3 - "$out/s5q_$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], {a: round(k[a]["ms_mean"]*1e3,1) for a in ("spmv1","spmv2","block_dot","semi_inverse","orthogonalize") if k[a]["ms_mean"]}, "G gathers/s %.1f / %.1f" % (5e8/k["spmv1"]["ms_mean"]/1e6, 5e8/k["spmv2"]["ms_mean"]/1e6), flush=True)
PY
done
