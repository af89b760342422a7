#!/bin/bash
# PMC passes for the dense MFMA kernels of one bench workload: tools/gpu_pmc_dense.sh <tag> [bench args]
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "VALUBusy MemUnitBusy MemUnitStalled MeanOccupancyPerCU" "GRBM_GUI_ACTIVE"; do
	i=$((i + 1))
	timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py "$@" --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/bench_pass$i.json" 2> "$out/pass$i.err" || { echo "pass $i ($set) failed"; tail -3 "$out/pass$i.err"; }
done
python3 tools/pmc_summary.py "$out" k_ortho_mfma k_block_dot_mfma > "$out/pmc_summary.txt" 2>&1
cat "$out/pmc_summary.txt"
