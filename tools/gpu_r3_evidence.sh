#!/bin/bash
# round 3, evidence runs with the final code: config 5 at full size on one GPU against the oracle, the GL7d19-shape solve to
# termination, config 3's shape through the executables (file in, lanczos_modp, checker_modp)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3e2
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python3 tools/check_config5_full.py > "$out/config5_full.txt" 2>&1 || { echo "config 5 full failed"; tail -8 "$out/config5_full.txt"; exit 1; }
tail -12 "$out/config5_full.txt"
