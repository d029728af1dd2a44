import csv, sys, glob
# what the host was doing around the large main-queue gaps: x_hostctx.py <dir> <prev-kernel substring> [min_us=300] [skip=3]
d = sys.argv[1]
sub = sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rows = []
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dmh::", "")))
rows.sort()
api = []
for f in glob.glob(d + "/*/*hip_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Thread_Id"], r["Function"]))
api.sort()
qs = sorted({q for _, _, q, _ in rows})
mainq = max(qs, key=lambda q: sum(1 for r in rows if r[2] == q))
start = len(rows) // 3
print("rows", len(rows), "api", len(api), "queues", {q: sum(1 for r in rows if r[2] == q) for q in qs}, "main", mainq)
last = None; found = 0
for idx in range(start, len(rows)):
    s, e, q, k = rows[idx]
    if q != mainq: continue
    if last is not None and s - last[1] > min_us * 1e3 and s - last[1] < 2e6 and sub in last[3]:
        found += 1
        if found > skip:
            t0 = last[1]
            print("---- gap of %.0f us after %s (before %s)" % ((s - last[1]) / 1e3, last[3], k))
            ev = []
            for s2, e2, q2, k2 in rows[max(0, idx - 80): idx + 12]:
                if s2 >= t0 - 4.0e6: ev.append((s2, "GPU  q%-2s %s%-28s %7.1f us" % (q2, "    " * qs.index(q2), k2[:28], (e2 - s2) / 1e3)))
            for s2, e2, th, fn in api:
                if s2 < t0 - 4.0e6 or s2 > t0 + 0.6e6: continue
                ev.append((s2, "HOST t%s %-34s %7.1f us" % (th[-3:], fn, (e2 - s2) / 1e3)))
            ev.sort()
            for t, line in ev: print("%9.1f  %s" % ((t - t0) / 1e3, line))
            break
    if last is not None and s - last[1] > 150e3: print("gap %.0f us after %s" % ((s - last[1]) / 1e3, last[3]))
    last = (s, e, q, k)
