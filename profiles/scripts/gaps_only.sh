#!/bin/bash
# kernel trace of a short default bench run -> stream gaps + kernel stats.  usage: gaps_only.sh <tag> [ENV=VALUE ...]
set -u
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
export TMPDIR=/tmp
T="${1:-g}"; shift
for kv in "$@"; do export "$kv"; done
O="gpurun_out/gaps_$T"
rm -rf "$O"; mkdir -p "$O"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$O/bench.json" 2> "$O/stats.err" || exit 1
python3 profiles/scripts/x_gaps.py "$O/stats" > "$O/stream_gaps.txt" 2>&1 || true
cp "$O"/stats/*/*kernel_stats.csv "$O/kernel_stats.csv"
python3 profiles/scripts/x_timeline.py "$O/stats" 0.6 24 > "$O/timeline.txt" 2>&1 || true
python3 profiles/scripts/x_gapctx.py "$O/stats" "${GAPK:-k_chain<64}" ${GAPMIN:-300} 8 > "$O/gapctx.txt" 2>&1 || true
python3 profiles/scripts/x_gaphist.py "$O/stats" > "$O/gaphist.txt" 2>&1 || true
rm -rf "$O/stats"
python3 -c "
import json
j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(j['value'], j['config']['ms_per_read_block'], j['host_seconds'])"
head -12 "$O/stream_gaps.txt"; cat "$O/gaphist.txt"
