#!/bin/bash
# A/B of environment variants on ONE box: tools/ab.sh <tag> <rounds> "VAR=1 VAR2=x" "VAR=0" ...
# every variant runs tools/kbench.py (metric workload, genomes, no per-kernel events) once per
# round, variants interleaved; prints ms/step per run.  An empty string "" is the default build.
TAG=$1; ROUNDS=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/$TAG
OUT=gpurun_out/$TAG/ab.txt
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    res=$(env $v python3 tools/kbench.py --workload ${AB_WORKLOAD:-c4_metric} --genomes --steps ${AB_STEPS:-100} --no-profile 2>/dev/null | head -1)
    echo "[$v] $res" | tee -a $OUT
  done
done
