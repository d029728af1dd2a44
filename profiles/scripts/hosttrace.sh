#!/bin/bash
# HIP API + kernel trace of a short default bench run; prints the host/GPU timeline around one large main-queue gap.
set -u
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
export TMPDIR=/tmp
T="${1:-h}"; shift
for kv in "$@"; do export "$kv"; done
O="gpurun_out/host_$T"
rm -rf "$O"; mkdir -p "$O"
timeout -k 10 500 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d "$O/tr" -- python3 bench.py --steps 1 --warmup 1 --blocks-per-step 2 --no-cpu-baseline > "$O/bench.json" 2> "$O/tr.err" || exit 1
ls -la "$O"/tr/* | head
python3 profiles/scripts/x_hostctx.py "$O/tr" k_scan_lookback 250 1 > "$O/hostctx.txt" 2>&1
rm -rf "$O/tr"
wc -l "$O/hostctx.txt"
