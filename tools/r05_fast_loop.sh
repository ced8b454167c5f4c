#!/bin/bash
# the eager policy turn: generic Agent.transition loop against Environment.fast_policy_loop (host share per agent).  GPU box.
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
for E in 1024 16384 65536; do
  LAT_GENERIC_LOOP=1 python tools/latency_bench.py one 32 32 8 3 $E 1
  python tools/latency_bench.py one 32 32 8 3 $E 1
done
LAT_GENERIC_LOOP=1 python tools/latency_bench.py one 64 64 64 7 2048 1
python tools/latency_bench.py one 64 64 64 7 2048 1
LAT_GENERIC_LOOP=1 python tools/latency_bench.py one 32 32 8 3 1024 3
python tools/latency_bench.py one 32 32 8 3 1024 3
LAT_GENERIC_LOOP=1 python tools/latency_bench.py one 32 32 8 3 1024 5
python tools/latency_bench.py one 32 32 8 3 1024 5
python tools/host_profile.py 1024 1500 2>&1 | head -50
