import csv, sys, glob
# the main stream's kernels in a window of the trace, with the gap in front of each: x_mainq.py <dir> [fraction=0.6] [window_ms=25]
files = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
win = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dmh::", "")))
rows.sort()
# the main stream: the queue that runs k_merge_match
mq = [q for _, _, q, k in rows if "k_merge_match" in k][0]
main = [r for r in rows if r[2] == mq]
t0 = main[int(len(main) * frac)][0]
prev = None
for s, e, q, k in main:
    if s < t0:
        prev = e
        continue
    if s > t0 + win * 1e6: break
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%9.1f %8.1f us  gap %7.1f %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, "<<<<" if gap > 60 else "    ", k[:50]))
    prev = e
