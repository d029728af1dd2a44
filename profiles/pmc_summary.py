#!/usr/bin/env python3
"""Builds the HBM-traffic summary bench.py uses for roofline.traffic from two rocprofv3 PMC passes over the SAME
bench command (collected separately, as the MI355X guide prescribes):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- python3 bench.py --steps 1 --warmup 1 --blocks-per-step 2 --no-cpu-baseline --no-stage-timing
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- python3 bench.py ...same...
    python profiles/pmc_summary.py out_f/*/*counter_collection.csv out_w/*/*counter_collection.csv "<workload>" > profiles/rNN_pmc_fetch_write_summary.json
gfx950: FETCH_SIZE counts half of a streamed read (MI355X_MICROARCH.md, HBM section) -> doubled;
WRITE_SIZE is exact.  Both are in KB.  Per kernel: launches, corrected bytes and the mean HBM traffic per launch."""
import collections
import csv
import json
import sys


def per_kernel(path):
    out = collections.defaultdict(lambda: {"launches": 0, "sum_KB": 0.0, "threads": 0})
    for r in csv.DictReader(open(path)):
        # (templated kernels -- k_radix_scatter<true>, k_merge_match<4, true> -- are summed under their plain name)
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        out[k]["launches"] += 1
        out[k]["sum_KB"] += float(r["Counter_Value"])
        out[k]["threads"] += int(r["Grid_Size"])
    return out


f, w = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
workload = sys.argv[3] if len(sys.argv) > 3 else ""
kern = {}
for k in sorted(set(f) | set(w)):
    fb = 2.0 * f[k]["sum_KB"] * 1024 if k in f else 0.0
    wb = w[k]["sum_KB"] * 1024 if k in w else 0.0
    n = max(f[k]["launches"] if k in f else 0, w[k]["launches"] if k in w else 0, 1)
    kern[k] = {"launches": n, "fetch_bytes_corrected": fb, "write_bytes": wb, "traffic_bytes_per_launch": (fb + wb) / n}
sc_f, sc_w = f["dmh::k_radix_scatter"], w["dmh::k_radix_scatter"]
hist = f["dmh::k_radix_hist"]
recs = sc_f["threads"] * 16            # 16 records per thread (upper bound: the last tile of a launch is partial)
# (keys-and-values launches move 12 + 12 B per record, keys-only launches 8 + 8: the figure below is the mean over
#  the launches of the profiled run)
fetch = 2.0 * sc_f["sum_KB"] * 1024
write = sc_w["sum_KB"] * 1024
print(json.dumps({
    "workload": workload,
    "kernels": kern,
    "derived": {
        "note": "gfx950: FETCH_SIZE counts half of the streamed bytes -> doubled; WRITE_SIZE exact. Check: k_radix_hist reads "
                "8 B per record, FETCH_SIZE x 2 per record = %.2f B" % (2.0 * hist["sum_KB"] * 1024 / max(1, hist["threads"] * 16)),
        "scatter_records_upper_bound": recs,
        "scatter_fetch_bytes_corrected": fetch, "scatter_write_bytes": write,
        "scatter_traffic_bytes_per_record": (fetch + write) / max(1, recs),
        "algorithmic_bytes_per_record": 32,
    }}, indent=1))
