import csv, sys, glob
# kernel timeline of a window of the trace across queues: x_timeline.py <dir> [fraction=0.6] [window_ms=14]
files = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
win = float(sys.argv[3]) if len(sys.argv) > 3 else 14.0
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dmh::", "")))
rows.sort()
t0 = rows[int(len(rows) * frac)][0]
qs = sorted({q for _, _, q, _ in rows})
for s, e, q, k in rows:
    if s < t0: continue
    if s > t0 + win * 1e6: break
    print("%9.1f %9.1f %7.1f  q%-2s %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, "    " * qs.index(q), k[:40]))
