#!/bin/bash
# Run HERE after tools/r05_final.sh 1 / 2 / 3 have been merged back into gpurun_out/: copies the artefacts into profiles/ and refreshes the traffic tags.
set -e
cd "$(dirname "$0")/.."
if [ -d gpurun_out/prof_r05_c3 ]; then
  cp gpurun_out/prof_r05_c3/summary.txt profiles/r05_c3_rocprofv3_summary.txt
  newest() { ls -t $1 | head -1; }      # (gpurun merges into gpurun_out/: an earlier run's files of the same kind may still be there)
  cp $(newest "gpurun_out/prof_r05_c3/trace/*/*_kernel_stats.csv") profiles/r05_c3_kernel_stats.csv
  cp gpurun_out/prof_r05_c3/trace_bench.json profiles/r05_c3_bench_under_rocprof.json
  cp $(newest "gpurun_out/prof_r05_c3/trace_full/*/*_kernel_stats.csv") profiles/r05_c3_full_line_kernel_stats.csv
  cp gpurun_out/prof_r05_c3/trace_full_bench.json profiles/r05_c3_full_line_bench_under_rocprof.json
  cp gpurun_out/r05_bench_driver_flags.json profiles/r05_c3_bench_driver_flags.json
  python tools/make_traffic_json.py c3 65536 gpurun_out/prof_r05_c3/summary.txt profiles/r05_c3_rocprofv3_summary.txt
fi
if [ -d gpurun_out/prof_r05_c5 ]; then
  newest() { ls -t $1 | head -1; }
  cp $(newest "gpurun_out/prof_r05_c5/*/*_kernel_stats.csv") profiles/r05_c5_kernel_stats.csv
  cp gpurun_out/prof_r05_c5.json profiles/r05_c5_bench_under_rocprof.json
  cp $(newest "gpurun_out/prof_r05_c2/*/*_kernel_stats.csv") profiles/r05_c2_kernel_stats.csv
  cp gpurun_out/prof_r05_c2.json profiles/r05_c2_bench_under_rocprof.json
  cp gpurun_out/r05_c5_traffic.txt profiles/r05_c5_traffic.txt
  python tools/make_traffic_json.py c5 2048 profiles/r05_c5_traffic.txt profiles/r05_c5_traffic.txt
fi
if [ -f gpurun_out/r05_resolve_probe.txt ]; then
  { echo "== tools/latency_bench.py spec: Environment.take_turn() through the Python API, one-layer policy, replay memories (us per turn; final kernels)"
    grep -v amdgpu.ids gpurun_out/r05_speculative_latency.txt
    echo; echo "== tools/spec_breakdown.py: config 5's shape, one shared model: every segment between two synchronisations (each segment carries ~15-20 us of synchronisation)"
    grep -v amdgpu.ids gpurun_out/r05_spec_breakdown.txt
    echo; echo "== tools/resolve_probe.py: what the first resolve pass is made of (the same pass repeated on the same state)"
    grep -v amdgpu.ids gpurun_out/r05_resolve_probe.txt
    echo; echo "== tools/spec_trace.sh: the kernels of the last turns (rocprofv3 --kernel-trace; 'gap' = idle time of the GPU before the kernel; the tracer slows the host)"
    tail -80 gpurun_out/r05_spec_trace.txt; } > profiles/r05_speculative_turn.txt
fi
if [ -f gpurun_out/r05_eager_turn.txt ]; then
  { echo "== tools/fast_loop_ab.py: the eager policy turn through the Python API, ONE process per shape (same tensors), alternating: the generic"
    echo "== Agent.transition loop, Environment.fast_policy_loop, and the latter with sweep + every window in one launch (sgw_sweep_observe_rows);"
    echo "== then tools/sweep_rows_bench.py: that launch against the two it replaces, engine kernels only"
    grep -v amdgpu.ids gpurun_out/r05_eager_turn.txt
    echo; echo "== tools/eager_trace.sh 65536: device time of one eager turn by kernel (rocprofv3 --kernel-trace)"
    grep -v amdgpu.ids gpurun_out/r05_eager_trace.txt; } > profiles/r05_eager_turn.txt
fi
ls -la profiles/r05_* profiles/traffic_*.json | awk '{print $5, $9}'
