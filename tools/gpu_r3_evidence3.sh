#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3e2
mkdir -p "$out"
cd "$root"
( while true; do sleep 60; echo "[progress] $(date +%T)"; done ) &
TICK=$!
timeout -k 10 700 python3 tests/full_solve.py relat9 --cli --left > "$out/full_solve_relat9_cli_left.json" 2>&1
rc=$?
kill $TICK
tail -2 "$out/full_solve_relat9_cli_left.json" | cut -c1-1200
exit $rc
