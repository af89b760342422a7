#!/bin/bash
# Run on the GPU box: tools/ubench2 plain, then under rocprofv3 --pmc (one counter set per pass, PMC only).
# Usage: tools/gpu_ubench2.sh <tag>
set -o pipefail
tag=${1:-ub2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
timeout -k 10 400 tools/ubench2 > "$out/ubench2.txt" 2>&1 || { echo "ubench2 failed"; tail -5 "$out/ubench2.txt"; exit 1; }
cat "$out/ubench2.txt"
i=0
for spec in "q1|TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
            "q1|TCC_EA0_RD_UNCACHED_32B_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum" \
            "q1|TCP_TCC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_RW_READ_REQ_sum" \
            "q2|TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
            "q2|TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
	i=$((i + 1))
	which=${spec%%|*}; set=${spec#*|}
	timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -- tools/ubench2 $which > "$out/pass$i.txt" 2> "$out/pass$i.err" || { echo "pass $i ($set) failed"; tail -5 "$out/pass$i.err"; }
done
python3 tools/pmc_summary.py "$out" > "$out/pmc_summary.txt" 2>&1
echo done
