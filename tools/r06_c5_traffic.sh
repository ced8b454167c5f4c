#!/bin/bash
# Run ON THE GPU BOX: config 5's HBM-side counters (FETCH_SIZE, WRITE_SIZE: separate passes) for the walking variant (a dword store per
# lane and channel) and for the staged one-env-per-workgroup variant (line-aligned 16-byte streaming stores), same batch.
# Question (round-4 review (kept for round 6: the same passes on the final sources)): WRITE_SIZE read 357 MB against 449 MB of algorithmic writes on the walking variant -- is that the counter
# (the guide calibrates WRITE_SIZE for 16-byte-per-lane streaming stores only) or the kernel?
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd $REPO
echo "== walking variant (default dispatch at 2 048 envs): direct dword stores" > $OUT/r06_c5_traffic.txt
timeout -k 10 500 bash tools/pmc_pass.sh r06_c5_walk "--config c5 --no-side-configs" "FETCH_SIZE" "WRITE_SIZE" >> $OUT/r06_c5_traffic.txt 2>&1 || exit 1
echo "== one env per workgroup, staged windows (options big_walk=0,big_stage=1): 16-byte streaming stores on 128-byte lines" >> $OUT/r06_c5_traffic.txt
export SGW_OPTIONS="big_walk=0,big_stage=1"
timeout -k 10 500 bash tools/pmc_pass.sh r06_c5_staged "--config c5 --no-side-configs" "FETCH_SIZE" "WRITE_SIZE" >> $OUT/r06_c5_traffic.txt 2>&1 || exit 1
echo "== algorithmic bytes per launch (2 048 envs): grid read 67 108 864, grid write 67 108 864, observations 380 633 088, rewards + positions + total + drawn actions 933 888; writes 448 675 840" >> $OUT/r06_c5_traffic.txt
cat $OUT/r06_c5_traffic.txt
