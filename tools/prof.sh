#!/bin/bash
# tools/prof.sh NAME -- PROGRAM ARGS...   rocprofv3 kernel trace + stats of one program run, csv under gpurun_out/NAME;
# prints the per-kernel summary (name, calls, total ns, average ns, percentage).
set -e
name=$1; shift; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$name
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -o p -- "$@" > gpurun_out/$name/run.log 2>&1 || { tail -20 gpurun_out/$name/run.log; exit 1; }
grep -v "^W2026\|^E2026" gpurun_out/$name/run.log | tail -12
f=$(find gpurun_out/$name -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print("%-70s calls %6s  avg %10.1f us  total %8.2f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
