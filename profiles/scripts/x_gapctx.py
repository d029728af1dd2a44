import csv, sys, glob
# context of the large gaps on the main queue: x_gapctx.py <dir> <prev-kernel substring> [min_us=300] [instances=2]
files = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
sub = sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
inst = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dmh::", "")))
rows.sort()
qs = sorted({q for _, _, q, _ in rows})
mainq = max(qs, key=lambda q: sum(1 for r in rows if r[2] == q))
start = len(rows) // 2
last = None; found = 0
for idx in range(start, len(rows)):
    s, e, q, k = rows[idx]
    if q != mainq: continue
    if last is not None and min_us * 1e3 < s - last[1] < 2e6 and sub in last[3]:
        found += 1
        t0 = last[1]
        print("---- gap of %.0f us after %s (before %s)" % ((s - last[1]) / 1e3, last[3], k))
        for s2, e2, q2, k2 in rows[max(0, idx - 60): idx + 12]:
            if s2 < t0 - 3.5e6: continue
            print("%9.1f %9.1f %7.1f  q%-2s %s%s" % ((s2 - t0) / 1e3, (e2 - t0) / 1e3, (e2 - s2) / 1e3, q2, "    " * qs.index(q2), k2[:40]))
        if found >= inst: break
    last = (s, e, q, k)
