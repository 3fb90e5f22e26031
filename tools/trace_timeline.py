"""Compact timeline of one rocprofv3 run (--kernel-trace --memory-copy-trace --output-format csv): every dispatch / copy longer than
a threshold as `start end dur queue name`, ms relative to the first record of the window.  python tools/trace_timeline.py DIR [min_ms] [t0_ms t1_ms]"""
import csv
import glob
import sys

d = sys.argv[1]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "dma", r.get("Direction", "copy")))
rows.sort()
if not rows:
    sys.exit("no trace rows under " + d)
t_last = rows[-1][1]
win0 = float(sys.argv[3]) if len(sys.argv) > 4 else None
win1 = float(sys.argv[4]) if len(sys.argv) > 4 else None
base = rows[0][0]
for s, e, q, n in rows:
    a, b = (s - base) / 1e6, (e - base) / 1e6
    if b - a < min_ms:
        continue
    if win0 is not None and (b < win0 or a > win1):
        continue
    print("%9.1f %9.1f %7.1f  %-5s %s" % (a, b, b - a, q, n))
print("span %.1f ms, %d records" % ((t_last - base) / 1e6, len(rows)))
