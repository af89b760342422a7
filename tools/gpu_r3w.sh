#!/bin/bash
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3w
mkdir -p "$out"
cd "$root"
B="python3 bench.py --workload band --steps 20 --warmup 3 --repeats 3 --cpu-seconds 0 --ref-iterations 0 --extras 0"
for v in "default:BLZ_X=0" "cu6:BLZ_SPMV_BLOCKS_PER_CU=6" "cu8:BLZ_SPMV_BLOCKS_PER_CU=8" "u4:BLZ_STAGE_U=4" "u4_cu8:BLZ_STAGE_U=4 BLZ_SPMV_BLOCKS_PER_CU=8" "nopair_cu8:BLZ_NO_PAIR=1 BLZ_SPMV_BLOCKS_PER_CU=8" "nostage:BLZ_NO_STAGE=1" "nostage_cu8:BLZ_NO_STAGE=1 BLZ_SPMV_BLOCKS_PER_CU=8" "nofuse:BLZ_NO_FUSE=1" "nofuse_cu8:BLZ_NO_FUSE=1 BLZ_SPMV_BLOCKS_PER_CU=8"; do
	tag=${v%%:*}; envs=${v#*:}
	env $envs timeout -k 10 300 $B > "$out/band_$tag.json" 2> "$out/band_$tag.err" || { echo "$tag failed"; tail -5 "$out/band_$tag.err"; exit 1; }
	python3 - "$out/band_$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print("%-12s" % sys.argv[2], "ms/step %.4f" % d["ms_per_step"], {a: round(k[a]["ms_mean"]*1e3,1) for a in ("spmv1","spmv2","block_dot","semi_inverse","orthogonalize") if k[a]["ms_mean"]}, flush=True)
PY
done
