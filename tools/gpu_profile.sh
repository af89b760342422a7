#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats and, in separate passes, the HBM PMC counters
# for one bench.py workload.  Usage: tools/gpu_profile.sh <tag> [bench args...]
# Outputs: gpurun_out/<tag>/{stats,fetch,write}/ (rocprofv3 csv) and gpurun_out/<tag>/summary.txt
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py "$@" --cpu-seconds 0 --extras 0 > "$out/bench_stats.json" 2> "$out/stats.err" || { echo "stats pass failed"; tail -5 "$out/stats.err"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py "$@" --cpu-seconds 0 --extras 0 > "$out/bench_fetch.json" 2> "$out/fetch.err" || { echo "fetch pass failed"; tail -5 "$out/fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py "$@" --cpu-seconds 0 --extras 0 > "$out/bench_write.json" 2> "$out/write.err" || { echo "write pass failed"; tail -5 "$out/write.err"; exit 1; }
python3 tools/summarize_prof.py "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
