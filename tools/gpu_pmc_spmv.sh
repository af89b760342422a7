#!/bin/bash
# PMC passes (wave scheduler + texture path + L2) for the SpMV kernels of one bench workload:
#   tools/gpu_pmc_spmv.sh <tag> [bench args]
# Every counter set is first checked against the per-block counter registers of the device (tools/pmc_sets.py, from
# `rocprofv3 --list-avail` of THIS box) and split where it does not fit one pass -- round 2's TA set asked for three TA
# hardware counters where the block has two, rocprofiler aborted with error 38 inside the program's set-up and the pass
# sat in the profiler's signal handler for the step's whole limit.  A pass that fails or runs into its 120 s limit ends
# the script: no further GPU step after a kill.
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
python3 -c "import torch" > /dev/null 2>&1		# first import on a fresh box pages the image in (1-2 min): not inside a timed pass
rocprofv3 --list-avail > "$out/avail.txt" 2> "$out/avail.err" || { echo "rocprofv3 --list-avail failed"; tail -3 "$out/avail.err"; exit 1; }
if [ -n "$PMC_SHORT" ]; then	# A/B of two builds or switches: wave scheduler, busy / occupancy, requests in flight only
python3 tools/pmc_sets.py "$out/avail.txt" \
	"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
	"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
	"VALUBusy MemUnitStalled MeanOccupancyPerCU" \
	"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" > "$out/passes.txt"
else
python3 tools/pmc_sets.py "$out/avail.txt" \
	"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
	"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
	"VALUBusy MemUnitStalled MeanOccupancyPerCU" "GRBM_GUI_ACTIVE" \
	"TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
	"TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
	"TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
	"TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
	"TCP_UTCL1_TRANSLATION_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum" > "$out/passes.txt"
fi
cat "$out/passes.txt"
i=0
while IFS= read -r set; do
	case "$set" in \#*|"") continue;; esac
	i=$((i + 1))
	timeout -k 5 120 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- python3 bench.py "$@" --cpu-seconds 0 --ref-iterations 0 --extras 0 > "$out/bench_pass$i.json" 2> "$out/pass$i.err"
	rc=$?
	if [ $rc -ne 0 ]; then
		echo "pass $i ($set) failed with $rc: stopping here"; tail -3 "$out/pass$i.err"
		break
	fi
done < "$out/passes.txt"
python3 tools/pmc_summary.py "$out" k_spmv > "$out/pmc_summary.txt" 2>&1
cat "$out/pmc_summary.txt"
