#!/bin/bash
# round-4 evidence: default bench (+ measured CPU baseline + parity of the timed output), rocprof kernel stats, PMC passes.
# usage: evidence_r04.sh <tag> [skip-default]
set -u
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
export TMPDIR=/tmp
# the library asks for twelve hardware queues when it is the first to use HIP in the process (dmh_core.hip init_device);
# under rocprofv3 the profiler is first, so the setting comes from here
export GPU_MAX_HW_QUEUES=12
T="${1:-r04a}"
O="gpurun_out/ev_$T"
rm -rf "$O"; mkdir -p "$O"
if [ "${2:-}" != "skip-default" ]; then
  timeout -k 10 560 python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err" || exit 1
  echo "default done"
fi
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/stats.err" || exit 1
echo "stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_f" -- python3 bench.py --steps 1 --warmup 1 --blocks-per-step 2 --no-cpu-baseline --no-stage-timing > "$O/pmc_f.json" 2> "$O/pmc_f.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_w" -- python3 bench.py --steps 1 --warmup 1 --blocks-per-step 2 --no-cpu-baseline --no-stage-timing > "$O/pmc_w.json" 2> "$O/pmc_w.err" || exit 1
echo "pmc done"
python3 profiles/pmc_summary.py "$O"/pmc_f/*/*counter_collection.csv "$O"/pmc_w/*/*counter_collection.csv grch38 > "$O/pmc_fetch_write_summary.json" || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d "$O/pmc_sq" -- python3 bench.py --steps 1 --warmup 1 --blocks-per-step 2 --no-cpu-baseline --no-stage-timing > "$O/pmc_sq.json" 2> "$O/pmc_sq.err" || exit 1
python3 - "$O" <<'PY' > "$O/pmc_sq_summary.txt"
import csv, collections, glob, sys
O = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for p in glob.glob(O + "/pmc_sq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
print("per kernel, summed over its launches of one bench step (2 read blocks x 18 passes, GRCh38-sized repeat-rich reference)")
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:18]:
    print(k, "launches", n[k], " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())))
PY
# stream gaps of the traced run (kernel trace per HIP queue)
python3 profiles/scripts/x_gaps.py "$O/stats" > "$O/stream_gaps.txt" 2>&1 || true
cp "$O"/stats/*/*kernel_stats.csv "$O/kernel_stats.csv"
rm -rf "$O/stats" "$O/pmc_f" "$O/pmc_w" "$O/pmc_sq"
for f in "$O"/bench_*.json; do echo "$f"; python3 -c "
import json,sys
j=json.loads(open('$f').read().strip().splitlines()[-1]); print('  ', j['value'], j['ms_per_step'], j['roofline'], (j.get('cpu_baseline') or {}).get('value'), j.get('parity'))"; done
head -14 "$O/kernel_stats.csv"
cat "$O/pmc_sq_summary.txt"
head -30 "$O/stream_gaps.txt"
# the other workloads, each with its CPU-reference leg and the parity check of the timed output
if [ "${3:-}" = "workloads" ]; then
  for w in ecoli chr1 grch38-random; do
    timeout -k 10 500 python3 bench.py --workload $w --steps 3 --warmup 1 > "$O/bench_$w.json" 2> "$O/bench_$w.err" || exit 1
    python3 -c "
import json
j=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', j['value'], j['config']['ms_per_read_block'], (j.get('cpu_baseline') or {}).get('value'), (j.get('parity') or {}).get('differing'), (j.get('parity') or {}).get('records'))"
  done
  timeout -k 10 560 python3 bench.py --flags n95p --steps 3 --warmup 1 > "$O/bench_n95p.json" 2> "$O/bench_n95p.err" || exit 1
  python3 -c "
import json
j=json.loads(open('$O/bench_n95p.json').read().strip().splitlines()[-1]); print('n95p', j['value'], j['config']['ms_per_read_block'], (j.get('cpu_baseline') or {}).get('value'), j.get('parity'))"
  timeout -k 10 560 python3 bench.py --read-lengths lognormal --steps 3 --warmup 1 > "$O/bench_lognormal.json" 2> "$O/bench_lognormal.err" || exit 1
  python3 -c "
import json
j=json.loads(open('$O/bench_lognormal.json').read().strip().splitlines()[-1]); print('lognormal', j['value'], j['config']['ms_per_read_block'], (j.get('cpu_baseline') or {}).get('value'), (j.get('parity') or {}).get('differing'), (j.get('parity') or {}).get('records'))"
fi
