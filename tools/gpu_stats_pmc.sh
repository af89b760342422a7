#!/bin/bash
# Run on the GPU box: kernel-trace stats + one PMC pass (L2 hits / fabric requests) of one bench.py workload under an
# environment.  Usage: tools/gpu_stats_pmc.sh <tag> "<env>" [bench args...]
set -o pipefail
tag=$1; e=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
for kv in $e; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py "$@" --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/bench_stats.json" 2> "$out/stats.err" || { echo "stats pass failed"; tail -5 "$out/stats.err"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$out/pmc1" -- python3 bench.py "$@" --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/bench_pmc1.json" 2> "$out/pmc1.err" || { echo "pmc pass failed"; tail -5 "$out/pmc1.err"; }
echo "== [$e] $@ ==" > "$out/summary.txt"
python3 tools/summarize_prof.py "$out" >> "$out/summary.txt" 2>&1
python3 tools/pmc_summary.py "$out" k_spmv >> "$out/summary.txt" 2>&1
cat "$out/summary.txt"
