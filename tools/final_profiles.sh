#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the rocprofv3-based artefacts kept under profiles/ (kernel traces of the bench line, of the
# policy-driven protocols and of the side shapes; counter passes).  usage: tools/final_profiles.sh <part: 1 | 2>
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
PART=${1:-1}
python3 -c 'import sys; sys.path.insert(0, "'$REPO'"); import __graft_entry__ as g; g.build()' || exit 1
if [ "$PART" = "1" ]; then
  bash $REPO/tools/profile_gpu.sh r03_c3 > $OUT/profile_r03_c3.log 2>&1 || exit 1
  echo headline profile done
  cd $REPO && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err || exit 1
  echo driver-flags line done
  timeout -k 10 300 python3 tools/ramp.py > $OUT/launch_ramp.txt 2>&1 || exit 1
  echo ramp done
else
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_phased_c3 -- python3 $REPO/tools/phased_bench.py > $OUT/prof_phased_c3.out 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_phased_c5 -- python3 $REPO/tools/phased_bench.py 128 128 64 5 2048 > $OUT/prof_phased_c5.out 2>&1 || exit 1
  echo phased traces done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_misc -- python3 $REPO/tools/bench_misc.py > $OUT/prof_misc.out 2>&1 || exit 1
  MISC_ONLY=big timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_misc_big -- python3 $REPO/tools/bench_misc.py > $OUT/prof_misc_big.out 2>&1 || exit 1
  echo misc traces done
  cd $REPO
  MISC_ONLY=cleanup CLEANUP_E=65536 timeout -k 10 700 bash tools/pmc_misc.sh cleanup_r03 > $OUT/pmc_cleanup_r03.txt 2>&1 || exit 1
  MISC_ONLY=big timeout -k 10 700 bash tools/pmc_misc.sh big_r03 > $OUT/pmc_big_r03.txt 2>&1 || exit 1
  echo counter passes done
fi
