import csv, sys, glob, collections
# distribution of the main queue's gaps that follow the last chain kernel of a pass
files = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
t0 = rows[len(rows) // 3][0]
byq = collections.defaultdict(list)
for s, e, q, k in rows:
    if s >= t0: byq[q].append((s, e, k))
main = max(byq.values(), key=len)
gaps = []
last_e, prevk = main[0][1], main[0][2]
for s, e, k in main[1:]:
    if s > last_e and "k_chain<64" in prevk: gaps.append((s - last_e) / 1e3)
    if e > last_e: last_e, prevk = e, k
gaps.sort()
n = len(gaps)
print("gaps behind k_chain<64>: %d, total %.1f ms" % (n, sum(gaps) / 1e3))
for lo, hi in ((0, 100), (100, 300), (300, 1000), (1000, 3000), (3000, 10000), (10000, 1e9)):
    sel = [g for g in gaps if lo <= g < hi]
    print("  %6d - %6s us: %4d gaps, %8.1f ms" % (lo, "inf" if hi > 1e8 else int(hi), len(sel), sum(sel) / 1e3))
