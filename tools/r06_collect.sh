#!/bin/bash
# Run HERE after tools/r06_final.sh 1 / 2 / 3 have been merged back into gpurun_out/: copies the artefacts into profiles/ and refreshes the traffic tags.
set -e
cd "$(dirname "$0")/.."
if [ -d gpurun_out/prof_r06_c3 ]; then
  cp gpurun_out/prof_r06_c3/summary.txt profiles/r06_c3_rocprofv3_summary.txt
  newest() { ls -t $1 | head -1; }      # (gpurun merges into gpurun_out/: an earlier run's files of the same kind may still be there)
  cp $(newest "gpurun_out/prof_r06_c3/trace/*/*_kernel_stats.csv") profiles/r06_c3_kernel_stats.csv
  cp gpurun_out/prof_r06_c3/trace_bench.json profiles/r06_c3_bench_under_rocprof.json
  cp $(newest "gpurun_out/prof_r06_c3/trace_full/*/*_kernel_stats.csv") profiles/r06_c3_full_line_kernel_stats.csv
  cp gpurun_out/prof_r06_c3/trace_full_bench.json profiles/r06_c3_full_line_bench_under_rocprof.json
  cp gpurun_out/r06_bench_driver_flags.json profiles/r06_c3_bench_driver_flags.json
  python tools/make_traffic_json.py c3 65536 gpurun_out/prof_r06_c3/summary.txt profiles/r06_c3_rocprofv3_summary.txt
fi
if [ -d gpurun_out/prof_r06_c5 ]; then
  newest() { ls -t $1 | head -1; }
  cp $(newest "gpurun_out/prof_r06_c5/*/*_kernel_stats.csv") profiles/r06_c5_kernel_stats.csv
  cp gpurun_out/prof_r06_c5.json profiles/r06_c5_bench_under_rocprof.json
  cp $(newest "gpurun_out/prof_r06_c2/*/*_kernel_stats.csv") profiles/r06_c2_kernel_stats.csv
  cp gpurun_out/prof_r06_c2.json profiles/r06_c2_bench_under_rocprof.json
  cp gpurun_out/r06_c5_traffic.txt profiles/r06_c5_traffic.txt
  python tools/make_traffic_json.py c5 2048 profiles/r06_c5_traffic.txt profiles/r06_c5_traffic.txt
fi
if [ -f gpurun_out/r06_speculative_latency.txt ]; then
  { echo "== tools/latency_bench.py spec on round 6's final sources: Environment.take_turn() through the Python API, one-layer policy, replay memories (us per turn)"
    grep -v amdgpu.ids gpurun_out/r06_speculative_latency.txt; } > profiles/r06_turn_loops_latency.txt
fi
ls -la profiles/r06_* profiles/traffic_*.json | awk '{print $5, $9}'
