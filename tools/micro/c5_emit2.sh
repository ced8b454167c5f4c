#!/bin/bash
# Run ON THE GPU BOX: what separates config 5's emit patterns from torch's fill_ (tools/micro/c5_emit.hip, modes 5-7).
cd "$(dirname "$0")"
[ -x ./c5_emit ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o c5_emit c5_emit.hip || exit 1
for nt in 0 1; do
  ./c5_emit 5 2048 0 0 $nt
  ./c5_emit 5 2048 1 0 $nt
  ./c5_emit 6 2048 0 0 $nt
  ./c5_emit 7 2048 0 0 $nt
  ./c5_emit 1 2048 0 80000 $nt
  ./c5_emit 1 2048 0 39936 $nt
done
