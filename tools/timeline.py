"""Print the last `--last MS` milliseconds of a rocprofv3 --kernel-trace --memory-copy-trace run as one merged timeline
(start, duration, queue / direction, name).  python3 tools/timeline.py DIR [--last MS]"""
import csv
import glob
import sys

d = sys.argv[1]
last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q%s" % r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-48:]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy", "%s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
if not ev:
    raise SystemExit("no trace rows under " + d)
t_end = max(e[1] for e in ev)
t0 = t_end - int(last_ms * 1e6)
first = None
for s, e, q, name in ev:
    if s < t0:
        continue
    if first is None:
        first = s
    print("%9.1f us  +%8.1f us  %-6s %s" % ((s - first) / 1e3, (e - s) / 1e3, q, name))
