#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace + stats, then HBM counters in separate passes.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
# build BEFORE any profiled process exists: under rocprofv3 the preload has initialised the GPU before python starts,
# and bench.py refuses to start a compiler from such a process (it exits non-zero on a stale library instead)
python3 -c 'import sys; sys.path.insert(0, "'$REPO'"); import __graft_entry__ as g; g.build()' || exit 1
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --prewarm-steps 0 --no-series --turns-per-launch 0 --no-cpu-baseline --no-side-configs $@"   # (counter passes: the headline kernel only)
# the timing pass runs long enough for the clocks to settle (the counter passes below stay short); it keeps the side configs and the
# policy-turn leg of bench.py, so the kernel stats of ONE trace name every kernel the line's numbers come from
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 500 --warmup 20 --no-series --no-cpu-baseline --no-side-configs "$@" > $OUT/trace_bench.json 2> $OUT/trace.err
# the whole line once more (side configs: config 2, config 5's share, config 3 at 524 288 envs; the policy-turn leg): names the
# kernels those numbers come from (the headline kernel's average in THIS trace mixes batch sizes and is not the headline's)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -- python3 $REPO/bench.py --steps 100 --warmup 20 --no-series --no-cpu-baseline "$@" > $OUT/trace_full_bench.json 2> $OUT/trace_full.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq2.err
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
