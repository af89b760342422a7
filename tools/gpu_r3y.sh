#!/bin/bash
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3y
mkdir -p "$out"
cd "$root"
for wl in relat9 gl7d19; do
for v in "fused:BLZ_X=0" "nofuse:BLZ_NO_FUSE=1" "nofuse_stage:BLZ_NO_FUSE=1 BLZ_STAGE_ALWAYS=1"; do
	tag=${v%%:*}; envs=${v#*:}
	env $envs timeout -k 10 300 python3 bench.py --workload $wl --steps 20 --warmup 3 --repeats 3 --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/${wl}_$tag.json" 2> "$out/${wl}_$tag.err" || { echo "$tag failed"; tail -5 "$out/${wl}_$tag.err"; exit 1; }
	python3 - "$out/${wl}_$tag.json" "$wl $tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print("%-22s" % sys.argv[2], "ms/step %.4f" % d["ms_per_step"], {a: round(k[a]["ms_mean"]*1e3,1) for a in ("spmv1","spmv2","block_dot","semi_inverse","orthogonalize") if k[a]["ms_mean"]}, flush=True)
PY
done
done
