#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the artefacts of round 5 kept under profiles/.  usage: tools/r05_final.sh <part: 1 | 2 | 3>
#   1  headline: kernel trace + stats, the whole line under the tracer, HBM counter passes (tools/profile_gpu.sh), the line with the driver's flags
#   2  config 5 and config 2: kernel stats under the tracer, config 5's traffic counters for both store variants
#   3  the speculative / recorded / eager policy turns (tools/latency_bench.py spec), breakdown and kernel timeline of a speculative turn,
#      the eager turn's loops A/B (tools/fast_loop_ab.py), sweep + rows in one launch (tools/sweep_rows_bench.py), kernel time of an eager turn
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
PART=${1:-1}
python3 -c 'import sys; sys.path.insert(0, "'$REPO'"); import __graft_entry__ as g; g.build()' || exit 1
if [ "$PART" = "1" ]; then
  bash $REPO/tools/profile_gpu.sh r05_c3 > $OUT/profile_r05_c3.log 2>&1 || exit 1
  find $OUT/prof_r05_c3 -name "*kernel_trace.csv" -delete; find $OUT/prof_r05_c3 -name "*counter_collection.csv" -delete
  echo headline profile done
  cd $REPO && timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/r05_bench_driver_flags.json 2> $OUT/r05_bench_driver_flags.err || exit 1
  echo driver-flags line done
elif [ "$PART" = "2" ]; then
  cd /tmp && export TMPDIR=/tmp
  for cfg in c5 c2; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r05_$cfg -- python3 $REPO/bench.py --config $cfg --steps 300 --warmup 10 --prewarm-steps 700 --no-cpu-baseline --no-series --no-side-configs --turns-per-launch 0 > $OUT/prof_r05_$cfg.json 2> $OUT/prof_r05_$cfg.err || exit 1
    find $OUT/prof_r05_$cfg -name "*kernel_trace.csv" -delete
  done
  echo kernel stats done
  cd $REPO && bash tools/r05_c5_traffic.sh > /dev/null 2>&1 || exit 1
  rm -rf $OUT/pmc_r05_c5_walk/g* $OUT/pmc_r05_c5_staged/g*
  echo traffic done
else
  cd $REPO
  timeout -k 10 700 python3 tools/latency_bench.py spec > $OUT/r05_speculative_latency.txt 2>&1 || exit 1
  timeout -k 10 300 python3 tools/spec_breakdown.py > $OUT/r05_spec_breakdown.txt 2>&1 || exit 1
  timeout -k 10 300 python3 tools/resolve_probe.py > $OUT/r05_resolve_probe.txt 2>&1 || exit 1
  bash tools/spec_trace.sh > $OUT/r05_spec_trace.txt 2>&1 || exit 1
  { timeout -k 10 300 python3 tools/fast_loop_ab.py 32 32 8 3 65536 3 && timeout -k 10 300 python3 tools/fast_loop_ab.py 32 32 8 3 16384 2 && timeout -k 10 300 python3 tools/fast_loop_ab.py 32 32 8 3 1024 2 \
    && timeout -k 10 300 python3 tools/fast_loop_ab.py 128 128 64 5 2048 2 && timeout -k 10 300 python3 tools/sweep_rows_bench.py 16384 65536; } > $OUT/r05_eager_turn.txt 2>&1 || exit 1
  bash tools/eager_trace.sh 65536 > $OUT/r05_eager_trace.txt 2>&1 || exit 1
  echo policy-turn measurements done
fi
