#!/bin/bash
# Run ON THE GPU BOX (through gpurun): round 4's measurements of BASELINE config 2 (what its ~10 us launch is made of) and config 5
# (the two speeds of its walking variant): ablation timings, stamp breakdowns (needs tools/libsgw_stamps.so = a -DSGW_STAMPS
# build), kernel traces and SQ / TCC counter passes.  usage: tools/r04_diag.sh <part: c2 | c5 | lat>
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
PART=${1:-c2}
cd $REPO
python3 -c 'import sys; sys.path.insert(0, "'$REPO'"); import __graft_entry__ as g; g.build()' > /dev/null || exit 1
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
if [ "$PART" = "c2" ]; then
  timeout -k 10 300 python3 tools/c2_breakdown.py > $OUT/r04_c2_breakdown.txt 2>&1 || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/dispatch_rate tools/micro/dispatch_rate.hip && timeout -k 10 60 /tmp/dispatch_rate 4096 >> $OUT/r04_c2_breakdown.txt 2>&1
  PYTHONPATH=$REPO SGW_LIB=$REPO/tools/libsgw_stamps.so SGW_OPTIONS="jit=0;group=64" timeout -k 10 120 python3 tools/stamps.py 4096 16 16 4 2 > $OUT/r04_c2_stamps.txt 2>&1
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r04_c2 -- python3 $REPO/bench.py --config c2 --steps 500 --warmup 10 --prewarm-steps 3000 --no-cpu-baseline --no-series --no-side-configs --turns-per-launch 0 > $OUT/prof_r04_c2.json 2> $OUT/prof_r04_c2.err) || exit 1
  timeout -k 10 700 bash tools/pmc_pass.sh r04_c2 "--config c2" "$SQ1" "$SQ2" "FETCH_SIZE" "WRITE_SIZE" > $OUT/r04_c2_pmc.txt 2>&1 || exit 1
elif [ "$PART" = "c5" ]; then
  (hostname; rocm-smi --showserial --showuniqueid 2>/dev/null | grep -i -E "serial|unique" | head -4) > $OUT/r04_c5_modes.txt 2>&1
  for v in "" "big_walk_share=1048576" "big_walk=0" "big_walk=0;big_stage=1"; do
    echo "== options: ${v:-default (walking workgroups at 2 048 envs)}" >> $OUT/r04_c5_modes.txt
    SGW_OPTIONS="$v" timeout -k 10 200 python3 -c "
import sys; sys.path.insert(0, '$REPO'); sys.path.insert(0, '$REPO/tools')
import torch
from sorrel_amd.engine import GridEngine
from sorrel_amd.spec import treasurehunt_spec
from _warm import timed_us
spec = treasurehunt_spec(128, 128, 64, 5, spawn_prob=0.05, seed=0, dense_prob=0.25)
eng = GridEngine(spec, 2048, device='cuda:0'); eng.reset(0)
for _ in range(700): eng.step(random_actions=True)
us = [timed_us(lambda: eng.step(random_actions=True), 200) for _ in range(4)]
by = spec.algorithmic_bytes_per_env_step() * 2048
print('us per launch', ['%.1f' % u for u in us], ' %.3f of 8 TB/s' % (by / min(us) / 1e3 / 8000), eng.launch_info().split(' threads')[0])
" >> $OUT/r04_c5_modes.txt 2>&1 || exit 1
  done
  PYTHONPATH=$REPO SGW_LIB=$REPO/tools/libsgw_stamps.so SGW_OPTIONS="jit=0" timeout -k 10 120 python3 tools/stamps_big.py 2048 > $OUT/r04_c5_stamps_walk.txt 2>&1 || exit 1
  PYTHONPATH=$REPO SGW_LIB=$REPO/tools/libsgw_stamps.so SGW_OPTIONS="jit=0;big_walk=0" timeout -k 10 120 python3 tools/stamps_big.py 2048 > $OUT/r04_c5_stamps_plain.txt 2>&1 || exit 1
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r04_c5 -- python3 $REPO/bench.py --config c5 --steps 300 --warmup 10 --prewarm-steps 700 --no-cpu-baseline --no-series --no-side-configs --turns-per-launch 0 > $OUT/prof_r04_c5.json 2> $OUT/prof_r04_c5.err) || exit 1
  timeout -k 10 700 bash tools/pmc_pass.sh r04_c5 "--config c5" "$SQ1" "$SQ2" "FETCH_SIZE" "WRITE_SIZE" > $OUT/r04_c5_pmc.txt 2>&1 || exit 1
  rocm-smi --showclocks --showpower > $OUT/r04_c5_smi.txt 2>&1
elif [ "$PART" = "jit" ]; then
  # the specialised instances: us per turn next to the prebuilt ones, create-time cost cold / from the disk cache / in memory, and a
  # kernel trace that names them
  rm -rf $REPO/sorrel_amd/csrc/jit_cache
  timeout -k 10 600 python3 tools/jit_probe.py > $OUT/r04_jit_probe.txt 2>&1 || exit 1
  timeout -k 10 300 python3 tools/generic_tables_probe.py > $OUT/r04_generic_tables.txt 2>&1 || exit 1
  PROBE_SMALL=1 timeout -k 10 300 python3 tools/generic_tables_probe.py >> $OUT/r04_generic_tables.txt 2>&1 || exit 1
  timeout -k 10 300 python3 tools/rt_shape_probe.py >> $OUT/r04_generic_tables.txt 2>&1 || exit 1
  timeout -k 10 300 python3 tools/jit_create_cost.py > $OUT/r04_jit_create_cost.txt 2>&1 || exit 1
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r04_jit -- python3 $REPO/tools/generic_tables_probe.py > $OUT/prof_r04_jit.out 2>&1) || exit 1
else
  timeout -k 10 600 python3 tools/latency_bench.py > $OUT/r04_api_latency.txt 2>&1 || exit 1
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r04_captured -- python3 $REPO/tools/profile_captured_turn.py 1024 > $OUT/prof_r04_captured.out 2>&1) || exit 1
fi
echo "part $PART done"
