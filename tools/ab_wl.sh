#!/bin/bash
# like tools/ab.sh for another workload: tools/ab_wl.sh <tag> <workload> <rounds> "ENV=.." ...
TAG=$1; WL=$2; ROUNDS=$3; shift; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/$TAG
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    res=$(env $v python3 tools/kbench.py --workload $WL --genomes --steps ${AB_STEPS:-200} --no-profile 2>/dev/null | head -1)
    echo "[$WL $v] $res" | tee -a gpurun_out/$TAG/ab_$WL.txt
  done
done
