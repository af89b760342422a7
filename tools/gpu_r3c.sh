#!/bin/bash
# round 3, GPU step c: what separates the n = 16 SpMV from the bare gather loop (ubench5), and the PMC script (fixed) on the
# config-5 quarter shape, lockstep and dynamic rows
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3c
mkdir -p "$out"
cd "$root"
timeout -k 10 300 tools/ubench5 1600 200 > "$out/ubench5.txt" 2>&1 || { tail -5 "$out/ubench5.txt"; exit 1; }
cat "$out/ubench5.txt"
B="--workload synth5q --steps 4 --warmup 1 --repeats 1"
BLZ_STAGE_DYN=0 bash tools/gpu_pmc_spmv.sh r3c/pmc_s5q_lockstep $B > "$out/pmc_lockstep.log" 2>&1 || { tail -20 "$out/pmc_lockstep.log"; exit 1; }
tail -60 "$out/pmc_lockstep.log"
PMC_SHORT=1 BLZ_STAGE_DYN=1 bash tools/gpu_pmc_spmv.sh r3c/pmc_s5q_dynamic $B > "$out/pmc_dynamic.log" 2>&1 || { tail -20 "$out/pmc_dynamic.log"; exit 1; }
tail -40 "$out/pmc_dynamic.log"
