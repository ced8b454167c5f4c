#!/bin/bash
# Run ON THE GPU BOX: config 5's emit patterns alone (tools/micro/c5_emit.hip).
cd "$(dirname "$0")"
[ -x ./c5_emit ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o c5_emit c5_emit.hip || exit 1
for E in 1024 2048; do
  for spin in 0 400; do
    ./c5_emit 0 $E $spin 39936 1      # direct dword stores, four workgroups per CU
    ./c5_emit 0 $E $spin 52000 1      # ... three
    ./c5_emit 3 $E $spin 39936 1      # per-window runs, 8-byte stores
    ./c5_emit 4 $E $spin 39936 1      # eight windows per burst, all waves
    ./c5_emit 4 $E $spin 52000 1
    ./c5_emit 2 $E $spin 60000 1      # half-env bursts, two workgroups per CU
    ./c5_emit 1 $E $spin 80000 1      # whole-env burst, two per CU (one per CU with the grid in LDS as well)
    ./c5_emit 1 $E $spin 100000 1     # ... one per CU
    ./c5_emit 4 $E $spin 39936 0      # (temporal stores)
    ./c5_emit 1 $E $spin 80000 0
  done
done
