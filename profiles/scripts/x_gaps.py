import csv, sys, collections, glob
# per queue: busy time, span, gaps histogram by the kernel that FOLLOWS the gap
files = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
t0 = rows[len(rows) // 3][0]          # skip setup / warm-up: last two thirds
byq = collections.defaultdict(list)
for s, e, q, k in rows:
    if s >= t0: byq[q].append((s, e, k))
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:4]:
    busy = sum(e - s for s, e, k in v)
    span = v[-1][1] - v[0][0]
    gaps = collections.Counter(); gcnt = collections.Counter()
    last_e = v[0][1]; prevk = v[0][2]
    for s, e, k in v[1:]:
        if s > last_e:
            gaps[(prevk, k)] += s - last_e; gcnt[(prevk, k)] += 1
        if e > last_e: last_e = e; prevk = k
    print("queue", q, "kernels", len(v), "span %.1f ms busy %.1f ms (%.0f%%)" % (span / 1e6, busy / 1e6, 100.0 * busy / span))
    for (a, b), g in gaps.most_common(14):
        print("   %8.2f ms in %6d gaps (%.0f us each)  %s -> %s" % (g / 1e6, gcnt[(a, b)], g / 1e3 / gcnt[(a, b)], a[-28:], b[-28:]))
