#!/bin/bash
# Run ON THE GPU BOX (round 6): config 5's observation stores alone, varying only which addresses are open at once (c5_emit.hip modes 6, 8-10 + remap),
# next to step_big's own pattern (mode 0) and torch's fill_.  Output: gpurun_out/c5_emit3.txt
cd "$(dirname "$0")"
[ -x ./c5_emit ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o c5_emit c5_emit.hip || exit 1
E=${1:-2048}
echo "# step_big's pattern and whole-env bursts, plain and with the XCD remap"
for remap in 0 1; do
  ./c5_emit 0 $E 0 39936 1 $remap
  ./c5_emit 0 $E 0 39936 0 $remap
  ./c5_emit 4 $E 0 39936 0 $remap
  ./c5_emit 1 $E 0 80000 0 $remap
  ./c5_emit 6 $E 0 0 0 $remap
  ./c5_emit 6 $E 0 0 1 $remap
done
echo "# block per piece: k pieces per env (k = 1: a block per env), remap 0 / 1 (XCD-contiguous pieces) / 2 (an env's pieces on one XCD)"
for k in 1 2 4 8 16; do
  for remap in 0 1 2; do
    ./c5_emit 8 $E $k 0 0 $remap
  done
done
./c5_emit 8 $E 4 0 1 0
./c5_emit 8 $E 8 0 1 0
echo "# persistent grid-stride fill: grid blocks x 256 threads, 1 / 4 stores per iteration"
for g in 1024 2048 4096 8192; do
  for u in 1 4; do
    ./c5_emit 9 $E $u $g 0 0
  done
done
./c5_emit 9 $E 4 2048 1 0
echo "# fill_'s launch shape: 256-thread blocks, spin consecutive float4 per thread"
for u in 1 2 4; do
  ./c5_emit 10 $E $u 0 0 0
done
./c5_emit 5 $E 0 0 0 0
./c5_emit 5 $E 0 0 1 0
echo "# step_big's direct pattern with aligned wave-wide stores (11: 256-byte dword chunks, 12: 16-byte stores from a 128-byte line)"
for remap in 0 1; do
  for nt in 1 0; do
    ./c5_emit 11 $E 0 39936 $nt $remap
    ./c5_emit 12 $E 0 39936 $nt $remap
  done
done
