#!/bin/bash
# A/B of when and how wide the deferred crossover runs beside the next step's small
# kernels (launch policy, resident workgroups per CU, chunks in flight per lane, CUs kept
# free of it); gpurun from the repo root:  bash tools/xo_overlap_ab.sh > gpurun_out/xo_overlap.txt
for cfg in "GNX_DEFER_XO=0" "GNX_XO_LAUNCH=0 GNX_XO_DROP=1 GNX_XO_BPC=32" "GNX_XO_LAUNCH=0 GNX_XO_DROP=2 GNX_XO_BPC=32" "GNX_XO_LAUNCH=0 GNX_XO_DROP=1 GNX_XO_BPC=4" "GNX_XO_LAUNCH=1 GNX_XO_DROP=1 GNX_XO_BPC=32" "GNX_XO_LAUNCH=0 GNX_XO_BPC=2 GNX_XO_UNROLL=6"; do
  echo "== $cfg"
  for i in 1 2; do env $cfg timeout -k 10 200 python tools/kbench.py --genomes --steps 100 --no-profile 2>&1 | grep "ind-steps"; done
  env $cfg timeout -k 10 200 python tools/kbench.py --genomes --steps 100 2>&1 | grep -E "crossover|move|sort |pairs|compact|find|density|death|sum"
done
