#!/bin/bash
# the other workloads of round 4, each with its CPU-reference leg and the parity check of its timed output.  usage: evidence_workloads.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
export TMPDIR=/tmp
T="${1:-r04d}"
O="gpurun_out/ev_$T"
mkdir -p "$O"
show() { python3 -c "
import json
j=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1]); print('$1', j['value'], j['config']['ms_per_read_block'], (j.get('cpu_baseline') or {}).get('value'), (j.get('parity') or {}).get('differing'), (j.get('parity') or {}).get('records'), j.get('alignment_ladder'))"; }
for w in ecoli chr1 grch38-random; do
  timeout -k 10 500 python3 bench.py --workload $w --steps 3 --warmup 1 > "$O/bench_$w.json" 2> "$O/bench_$w.err" || exit 1
  show $w
done
timeout -k 10 560 python3 bench.py --flags n95p --steps 3 --warmup 1 > "$O/bench_n95p.json" 2> "$O/bench_n95p.err" || exit 1
show n95p
timeout -k 10 560 python3 bench.py --read-lengths lognormal --steps 3 --warmup 1 > "$O/bench_lognormal.json" 2> "$O/bench_lognormal.err" || exit 1
show lognormal
