#!/bin/bash
# tools/prof.sh NAME -- PROGRAM ARGS...   rocprofv3 kernel trace + stats of one program run, csv under gpurun_out/NAME;
# prints the per-kernel summary (name, calls, total ns, average ns, percentage).
set -e
name=$1; shift; shift
# The program after `--` must be the interpreter binary itself (python3 PROGRAM ...): rocprofv3's preloaded library has the GPU
# initialised before the program starts, so a hop through `env`, a shell or a `#!/usr/bin/env` script would exec from a process that
# already holds the GPU -- forbidden on this pool.
case "$(basename -- "$1")" in
  python3|python|python3.*) ;;
  *) echo "$0: run the interpreter directly after -- (python3 PROGRAM ARGS), not '$1'" >&2; exit 2;;
esac
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
