#!/bin/bash
# Run on the GPU box: the round's profile set -- kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes for every single-GPU
# workload (tools/gpu_profile.sh), the extra PMC passes for the config-5 quarter shape, and the default bench line.
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
for w in gl7d19 relat9 relat8 synth5q; do
	steps=20; [ $w = synth5q ] && steps=5
	tools/gpu_profile.sh prof_$w --workload $w --steps $steps --warmup 2 --ref-iterations 0 > gpurun_out/prof_$w.log 2>&1 || { echo "profile $w failed"; tail -5 gpurun_out/prof_$w.log; exit 1; }
	echo "== $w done"
done
tools/gpu_pmc_extra.sh pmc_synth5q --workload synth5q --steps 3 --warmup 1 > gpurun_out/pmc_synth5q.log 2>&1 || { echo "pmc synth5q failed"; tail -5 gpurun_out/pmc_synth5q.log; exit 1; }
echo "== pmc synth5q done"
timeout -k 10 600 python bench.py > gpurun_out/bench_default_r03.json 2> gpurun_out/bench_default_r03.err || { echo "default bench failed"; tail -5 gpurun_out/bench_default_r03.err; exit 1; }
echo "== default bench done"
