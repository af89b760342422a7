#!/bin/bash
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3e2
mkdir -p "$out"
cd "$root"
( while true; do sleep 60; echo "[progress] $(date +%T)"; done ) &
TICK=$!
timeout -k 10 600 python3 tests/full_solve.py gl7d19 --right > "$out/full_solve_gl7d19_right.txt" 2>&1; rc1=$?
tail -4 "$out/full_solve_gl7d19_right.txt"
if [ $rc1 -eq 0 ]; then
	timeout -k 10 500 python3 tests/full_solve.py relat9 --cli > "$out/full_solve_relat9_cli_left.json" 2>&1; rc2=$?
	tail -3 "$out/full_solve_relat9_cli_left.json" | cut -c1-600
fi
kill $TICK
exit $(( rc1 + ${rc2:-0} ))
