#!/bin/bash
# Run HERE after tools/refresh_profiles.sh and tools/final_profiles.sh 1 / 2 have been merged back into gpurun_out/: copies the
# artefacts into profiles/ under their round-3 names and refreshes the traffic tag.
set -e
cd "$(dirname "$0")/.."
python tools/assemble_profiles.py > /dev/null
cp gpurun_out/prof_r03_c3/summary.txt profiles/r03_c3_rocprofv3_summary.txt
cp gpurun_out/prof_r03_c3/trace/*/*_kernel_stats.csv profiles/r03_c3_kernel_stats.csv
cp gpurun_out/prof_r03_c3/trace_bench.json profiles/r03_c3_bench_under_rocprof.json
cp gpurun_out/prof_r03_c3/trace_full/*/*_kernel_stats.csv profiles/r03_c3_full_line_kernel_stats.csv
cp gpurun_out/prof_r03_c3/trace_full_bench.json profiles/r03_c3_full_line_bench_under_rocprof.json
cp gpurun_out/bench_driver_flags.json profiles/r03_c3_bench_driver_flags.json
python tools/make_traffic_json.py c3 65536 gpurun_out/prof_r03_c3/summary.txt profiles/r03_c3_rocprofv3_summary.txt
cp gpurun_out/prof_phased_c3/*/*_kernel_stats.csv profiles/r03_phased_c3_kernel_stats.csv
cp gpurun_out/prof_phased_c5/*/*_kernel_stats.csv profiles/r03_phased_c5_kernel_stats.csv
cp gpurun_out/prof_misc/*/*_kernel_stats.csv profiles/r03_misc_kernel_stats.csv
cp gpurun_out/prof_misc_big/*/*_kernel_stats.csv profiles/r03_big_rule_worlds_kernel_stats.csv
(echo "# MISC_ONLY=cleanup CLEANUP_E=65536 tools/pmc_misc.sh, round 3 final kernels: SQ counters of the Cleanup 21x31x3 turn at 65 536 envs (per-dispatch averages)"; grep "avg=" gpurun_out/pmc_cleanup_r03.txt) > profiles/r03_cleanup_pmc.txt
(echo "# MISC_ONLY=big tools/pmc_misc.sh, round 3 final kernels: SQ counters of the rule worlds above 4 KiB (Tag 72x72, Tag 128x128 on step_big<..., TAG>; Cleanup 48x48x3 on the wave-per-env RULES kernel)"; grep "avg=" gpurun_out/pmc_big_r03.txt) > profiles/r03_big_rule_worlds_pmc.txt
