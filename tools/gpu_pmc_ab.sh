#!/bin/bash
# Run on the GPU box: a few PMC passes of one bench.py workload under two environments (A/B), PMC only.
# Usage: tools/gpu_pmc_ab.sh <tag> "<env A>" "<env B>" [bench args...]
set -o pipefail
tag=$1; envA=$2; envB=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$root"
v=0
for e in "$envA" "$envB"; do
	v=$((v + 1))
	out=$root/gpurun_out/$tag/v$v
	mkdir -p "$out"
	for kv in $e; do export "$kv"; done
	i=0
	for set in "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
	           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
	           "VALUBusy MemUnitStalled MeanOccupancyPerCU" \
	           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
		i=$((i + 1))
		timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py "$@" --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/bench_pass$i.json" 2> "$out/pass$i.err" || { echo "pass $i ($set) failed"; tail -5 "$out/pass$i.err"; }
	done
	echo "== [$e] ==" > "$out/pmc_summary.txt"
	python3 tools/pmc_summary.py "$out" k_spmv >> "$out/pmc_summary.txt" 2>&1
	cat "$out/pmc_summary.txt"
done
