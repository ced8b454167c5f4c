#!/bin/bash
# Run ON THE GPU BOX (through gpurun): every diagnostic bench whose output is kept under profiles/ (timings only; the
# rocprofv3 passes are tools/profile_gpu.sh, tools/pmc_misc.sh, tools/pmc_rt_shapes.sh).  usage: tools/refresh_profiles.sh <outdir>
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$REPO/gpurun_out/refresh}
mkdir -p $OUT
cd $REPO
python3 -c 'import __graft_entry__ as g; g.build()' || exit 1
T="timeout -k 10"
$T 300 python3 tools/bench_misc.py > $OUT/misc_bench.txt 2>&1 && CLEANUP_E=65536 MISC_ONLY=cleanup $T 100 python3 tools/bench_misc.py >> $OUT/misc_bench.txt 2>&1 && MISC_ONLY=big $T 200 python3 tools/bench_misc.py >> $OUT/misc_bench.txt 2>&1 || exit 1
echo misc done
$T 200 python3 tools/phased_bench.py > $OUT/phased_path.txt 2>&1 && $T 200 python3 tools/phased_bench.py 128 128 64 5 2048 >> $OUT/phased_path.txt 2>&1 || exit 1
echo phased done
$T 200 python3 tools/cleanup_rollout_bench.py > $OUT/cleanup_rollout_and_policy.txt 2>&1 && $T 200 python3 tools/cleanup_rollout_bench.py 65536 >> $OUT/cleanup_rollout_and_policy.txt 2>&1 && $T 100 python3 tools/act_probe.py >> $OUT/cleanup_rollout_and_policy.txt 2>&1 || exit 1
echo cleanup done
$T 300 python3 tools/latency_bench.py > $OUT/api_latency.txt 2>&1 || exit 1
echo latency done
$T 200 python3 tools/rt_shape_probe.py > $OUT/runtime_shapes.txt 2>&1 || exit 1
$T 400 python3 tools/group_sweep.py > $OUT/group_sweep.txt 2>&1 || exit 1
echo shapes done
$T 400 python3 tools/big_stage_probe.py > $OUT/big_world_staging.txt 2>&1 && PROBE_WALK=1 $T 400 python3 tools/big_stage_probe.py >> $OUT/big_world_staging.txt 2>&1 || exit 1
echo big done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/region_writer tools/micro/region_writer.hip && $T 100 /tmp/region_writer > $OUT/store_alignment_micro.txt 2>&1 || exit 1
$T 100 python3 tools/hbm_floor.py > $OUT/hbm_floor.txt 2>&1 || exit 1
$T 200 python3 tools/cleanup_observe_probe.py 1 3 5 10 > $OUT/cleanup_observe_probe.txt 2>&1 || exit 1
$T 300 python3 tools/generic_tables_probe.py > $OUT/generic_tables.txt 2>&1 && PROBE_SMALL=1 $T 300 python3 tools/generic_tables_probe.py >> $OUT/generic_tables.txt 2>&1 || exit 1
$T 600 python3 tools/tag_group_probe.py > $OUT/tag_group_probe.txt 2>&1 || exit 1
$T 900 python3 tools/mid_world_probe.py > $OUT/mid_worlds.txt 2>&1 || exit 1
echo all done
