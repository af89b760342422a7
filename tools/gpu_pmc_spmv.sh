#!/bin/bash
# PMC passes (wave scheduler + texture path + L2) for the SpMV kernels of one bench workload:
# tools/gpu_pmc_spmv.sh <tag> [bench args]
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
           "VALUBusy MemUnitBusy MemUnitStalled MeanOccupancyPerCU" "GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do	# (a TA_* pass hung the profiler on this pool: left out)
	i=$((i + 1))
	timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py "$@" --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/bench_pass$i.json" 2> "$out/pass$i.err" || { echo "pass $i ($set) failed"; tail -3 "$out/pass$i.err"; }
done
python3 tools/pmc_summary.py "$out" k_spmv > "$out/pmc_summary.txt" 2>&1
cat "$out/pmc_summary.txt"
