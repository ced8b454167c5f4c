#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the artefacts of round 6 kept under profiles/.  usage: tools/r06_final.sh <part: 1 | 2 | 3>
#   1  headline: kernel trace + stats, the whole line under the tracer, HBM counter passes (tools/profile_gpu.sh), the line with the driver's flags
#   2  config 5 and config 2: kernel stats under the tracer, config 5's traffic counters for both store variants
#   3  round 6's A/B tools on the final sources: sweep + rows in one launch (tools/rows_fused_ab.py), sgw_act behind what (tools/act_after_probe.py), the generic
#      speculative turn (tools/spec_generic_bench.py), more than 64 agents (tools/many_agents_bench.py), the turn loops through the Python API (tools/latency_bench.py spec)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
PART=${1:-1}
python3 -c 'import sys; sys.path.insert(0, "'$REPO'"); import __graft_entry__ as g; g.build()' || exit 1
if [ "$PART" = "1" ]; then
  bash $REPO/tools/profile_gpu.sh r06_c3 > $OUT/profile_r06_c3.log 2>&1 || exit 1
  find $OUT/prof_r06_c3 -name "*kernel_trace.csv" -delete; find $OUT/prof_r06_c3 -name "*counter_collection.csv" -delete
  echo headline profile done
  cd $REPO && timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/r06_bench_driver_flags.json 2> $OUT/r06_bench_driver_flags.err || exit 1
  echo driver-flags line done
elif [ "$PART" = "2" ]; then
  cd /tmp && export TMPDIR=/tmp
  for cfg in c5 c2; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r06_$cfg -- python3 $REPO/bench.py --config $cfg --steps 300 --warmup 10 --prewarm-steps 700 --no-cpu-baseline --no-series --no-side-configs --turns-per-launch 0 > $OUT/prof_r06_$cfg.json 2> $OUT/prof_r06_$cfg.err || exit 1
    find $OUT/prof_r06_$cfg -name "*kernel_trace.csv" -delete
  done
  echo kernel stats done
  cd $REPO && bash tools/r06_c5_traffic.sh > /dev/null 2>&1 || exit 1
  rm -rf $OUT/pmc_r06_c5_walk/g* $OUT/pmc_r06_c5_staged/g*
  echo traffic done
else
  cd $REPO
  timeout -k 10 300 python3 tools/rows_fused_ab.py > $OUT/r06_rows_fused_ab.txt 2>&1 || exit 1
  timeout -k 10 300 python3 tools/act_after_probe.py 65536 1024 > $OUT/r06_act_after.txt 2>&1 || exit 1
  timeout -k 10 300 python3 tools/spec_generic_bench.py 1024 4096 > $OUT/r06_spec_generic.txt 2>&1 || exit 1
  timeout -k 10 300 python3 tools/many_agents_bench.py > $OUT/r06_many_agents.txt 2>&1 || exit 1
  timeout -k 10 300 python3 tools/latency_bench.py spec > $OUT/r06_speculative_latency.txt 2>&1 || exit 1
  echo round-6 measurements done
fi
