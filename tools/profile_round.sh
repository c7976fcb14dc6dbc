#!/bin/bash
# tools/profile_round.sh TAG -- the round's committed evidence, run on the GPU box:
#   1. rocprofv3 --kernel-trace --stats of the headline bench command (python3 directly after `--`)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; counters never together with a trace) of a short bench run
# Outputs under gpurun_out/TAG_*; tools/pmc_summary.py turns the counter CSVs into profiles/pmc_traffic.json.
set -e -o pipefail
tag=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_bench -o p -- python3 bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 300 gpurun_out/${tag}_bench.err
# the timed step alone (its kernels only): the average of msm_accum_kernel here is what bench.py's roofline.kernel_ms must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_step -o p -- python3 bench.py --only-step --steps 20 --warmup 5 > gpurun_out/${tag}_step.json 2> gpurun_out/${tag}_step.err
for leg in step_fixed step_plain ntt; do
  case $leg in
    step_fixed) flags="--only-step --steps 5 --warmup 2 --form fixed";;
    step_plain) flags="--only-step --steps 5 --warmup 2 --form plain";;
    ntt)        flags="--only-ntt";;
  esac
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch_$leg -o p -- python3 bench.py $flags > /dev/null 2> gpurun_out/${tag}_pmc_fetch_$leg.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write_$leg -o p -- python3 bench.py $flags > /dev/null 2> gpurun_out/${tag}_pmc_write_$leg.err
  echo "== $leg"
  python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_fetch_$leg gpurun_out/${tag}_pmc_write_$leg gpurun_out/${tag}_pmc_$leg.json
done
