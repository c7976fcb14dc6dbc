#!/bin/bash
# tools/pmc_sq.sh NAME COUNTERS -- python3 PROGRAM ARGS   one --pmc pass (no trace), per-kernel mean of each counter
set -e
name=$1; ctrs=$2; shift; shift; shift
# The program after `--` must be the interpreter binary itself (python3 PROGRAM ...): rocprofv3's preloaded library has the GPU
# initialised before the program starts, so a hop through `env`, a shell or a `#!/usr/bin/env` script would exec from a process that
# already holds the GPU -- forbidden on this pool.
case "$(basename -- "$1")" in
  python3|python|python3.*) ;;
  *) echo "$0: run the interpreter directly after -- (python3 PROGRAM ARGS), not '$1'" >&2; exit 2;;
esac
export TMPDIR=/tmp
mkdir -p gpurun_out/$name
rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/$name -o p -- "$@" > gpurun_out/$name/run.log 2>&1 || { tail -20 gpurun_out/$name/run.log; exit 1; }
f=$(find gpurun_out/$name -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "ntt" not in k and "msm" not in k: continue
    print(k, "  ".join("%s=%.4g (n=%d)" % (c, sum(x) / len(x), len(x)) for c, x in sorted(v.items())))
PY
