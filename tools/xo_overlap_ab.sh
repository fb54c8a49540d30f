#!/bin/bash
# A/B of where the small kernels' stream waits for the full-width deferred crossover
# (GNX_XO_WAIT: 1 before the next sort, 2 at once, 3 after the compaction) against the
# narrow crossover beside the whole step; gpurun from the repo root.
for cfg in "GNX_XO_WAIT=1" "GNX_XO_WAIT=2" "GNX_XO_WAIT=3" "GNX_XO_SORT_WAIT=0" "GNX_XO_WAIT=2 GNX_XO_NT=0" "GNX_XO_WAIT=2 GNX_XO_UNROLL=8" "GNX_DEFER_XO=0"; do
  echo "== $cfg"
  for i in 1 2; do env $cfg timeout -k 10 200 python tools/kbench.py --genomes --steps 100 --no-profile 2>&1 | grep "ind-steps"; done
  env $cfg timeout -k 10 200 python tools/kbench.py --genomes --steps 100 2>&1 | grep -E "crossover|move|compact|sum"
done
