#!/bin/bash
# tools/profile_round4.sh -- round 4's committed evidence, run on the GPU box (python3 directly after `--` everywhere):
#   1. rocprofv3 --kernel-trace --stats of the timed step alone at 2^20 and at 2^24 (fixed-base form), and of the NTT leg alone
#      (2^22): per-kernel averages that can be recomputed per kernel AND size
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never together with a trace) of the 2^20 and 2^24 steps and the NTT leg
#   3. the kernel + copy timeline of a streamed host-pointer MSM (2^20 pairs, 3 chunks)
# Outputs under gpurun_out/r04/; copy the summaries into profiles/ (tools/profile_round4.sh prints the file names).
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r04
mkdir -p $out
stats() {  # NAME FLAGS...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o p -- python3 bench.py "$@" > $out/$name.json 2> $out/$name.err
  cp $(find $out/$name -name "*kernel_stats.csv" | head -1) $out/r04_${name}_kernel_stats.csv
  echo "== $name"; head -8 $out/r04_${name}_kernel_stats.csv | cut -c1-150
}
stats step_2p20_fixed --only-step --steps 20 --warmup 5
stats step_2p24_fixed --only-step --log-n 24 --steps 5 --warmup 2
stats ntt_2p22 --only-ntt
for leg in step_2p20_fixed step_2p24_fixed ntt_2p22; do
  case $leg in
    step_2p20_fixed) flags="--only-step --steps 5 --warmup 2";;
    step_2p24_fixed) flags="--only-step --log-n 24 --steps 3 --warmup 1";;
    ntt_2p22)        flags="--only-ntt";;
  esac
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$leg -o p -- python3 bench.py $flags > /dev/null 2> $out/pmc_fetch_$leg.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$leg -o p -- python3 bench.py $flags > /dev/null 2> $out/pmc_write_$leg.err
  echo "== pmc $leg"
  python3 tools/pmc_summary.py $out/pmc_fetch_$leg $out/pmc_write_$leg $out/r04_pmc_$leg.json
done
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/stream -o p -- python3 tools/stream_trace.py 20 3 600 > $out/stream.log 2>&1
python3 tools/timeline.py $out/stream --last 2.0 > $out/r04_stream_2p20_timeline.txt
echo "== stream timeline"; head -50 $out/r04_stream_2p20_timeline.txt
# round 4 additions: the evaluate_h call with the generated gates kernel (and, before it, the interpreter), a lone 2^17 fixed-base MSM
# (the host-finished tail: msm_planes_kernel), and the copy / kernel timeline of a pipelined host-pointer batch (8 x 2^20 columns)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/evalh -o p -- python3 tools/evalh_bench.py --check-k 0 > $out/evalh.json 2> $out/evalh.err
cp $(find $out/evalh -name "*kernel_stats.csv" | head -1) $out/r04_evalh_k18_kernel_stats.csv
echo "== evalh"; head -12 $out/r04_evalh_k18_kernel_stats.csv | cut -c1-150
rocprofv3 --kernel-trace --stats --output-format csv -d $out/msm17 -o p -- python3 tools/msm_bench.py --log-n 17 --no-plain --no-stages --reps 100 > $out/msm17.json 2> $out/msm17.err
cp $(find $out/msm17 -name "*kernel_stats.csv" | head -1) $out/r04_msm_2p17_fixed_kernel_stats.csv
echo "== msm17"; head -14 $out/r04_msm_2p17_fixed_kernel_stats.csv | cut -c1-150
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/hostbatch -o p -- python3 tools/host_ntt_batch.py 20 > $out/hostbatch.log 2>&1
python3 tools/timeline.py $out/hostbatch --last 8.0 > $out/r04_host_batch_2p20_timeline.txt || true
echo "== host batch timeline"; head -40 $out/r04_host_batch_2p20_timeline.txt
ls $out/r04_*
