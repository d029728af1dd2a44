#!/bin/bash
# SQ counters per kernel of one short bench step (own rocprofv3 pass, counters only): pmc_sq.sh <outdir> [extra bench args]
set -u
cd "${GRAFT_REPO_ROOT:?not on a gpurun box}" || exit 1
export TMPDIR=/tmp
O="$1"; shift
mkdir -p "$O"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d "$O/pmc_sq" -- python3 bench.py --steps 1 --warmup 1 --blocks-per-step 2 --no-cpu-baseline --no-stage-timing "$@" > "$O/pmc_sq.json" 2> "$O/pmc_sq.err" || exit 1
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
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:20]:
    print(k, "launches", n[k], " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items())))
PY
rm -rf "$O/pmc_sq"
cat "$O/pmc_sq_summary.txt"
