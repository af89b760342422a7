#!/bin/bash
# Run on the GPU box: extra PMC passes for one bench.py workload -- L2 -> fabric read requests by size, L2 hits and
# misses, outstanding-request level and credit stalls, VALU / memory-unit busy.  One rocprofv3 run per counter set
# (PMC only, no trace domains).  Usage: tools/gpu_pmc_extra.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum" \
           "VALUBusy MemUnitStalled MeanOccupancyPerCU" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum"; do
	i=$((i + 1))
	rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py "$@" --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/bench_pass$i.json" 2> "$out/pass$i.err" || { echo "pass $i ($set) failed"; tail -5 "$out/pass$i.err"; }
done
python3 tools/pmc_summary.py "$out" k_spmv k_orthogonalize k_block_dot > "$out/pmc_summary.txt" 2>&1
cat "$out/pmc_summary.txt"
