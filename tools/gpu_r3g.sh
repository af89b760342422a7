#!/bin/bash
# round 3, GPU step g: parity of the pair-lane form at n = 8, its A/B on the relat9 / GL7d19 shapes, the image by 16 wavefronts,
# sweeps ordering on a scrambled band
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3g
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -k "two_words or trajectory or semi_inverse or each_kernel or block_update or past_the_stop or rccl" > "$out/pytest.log" 2>&1 || { tail -30 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
for wl in relat9 gl7d19; do
for v in "pair8_0:BLZ_PAIR8=0" "pair8_1:BLZ_PAIR8=1" "pair8_1_always:BLZ_PAIR8=1 BLZ_STAGE_ALWAYS=1" "pair8_0_always:BLZ_PAIR8=0 BLZ_STAGE_ALWAYS=1"; do
	tag=${v%%:*}; envs=${v#*:}
	env $envs timeout -k 10 300 python3 bench.py --workload $wl --steps 20 --warmup 3 --repeats 3 --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/${wl}_$tag.json" 2> "$out/${wl}_$tag.err" || { echo "$tag failed"; tail -5 "$out/${wl}_$tag.err"; exit 1; }
	python3 - "$out/${wl}_$tag.json" "$wl $tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
k=d["kernels"]
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], {a: round(k[a]["ms_mean"]*1e3,1) for a in ("spmv1","spmv2","block_dot","semi_inverse","orthogonalize") if k[a]["ms_mean"]}, flush=True)
PY
done
done
timeout -k 10 300 python3 tools/exp_sweeps.py > "$out/exp_sweeps.txt" 2>&1 || { tail -5 "$out/exp_sweeps.txt"; exit 1; }
cat "$out/exp_sweeps.txt"
