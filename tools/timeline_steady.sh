#!/bin/bash
# timeline of one step of gnx_walk at the metric workload's steady state: tools/timeline_steady.sh <tag> [warm] [env...]
TAG=$1; WARM=${2:-1000}; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT; rm -rf $OUT/trace
cd $ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o run -- python3 tools/steady_ab.py $WARM 30 > $OUT/out.txt 2> $OUT/rocprof.err || { tail -5 $OUT/rocprof.err; exit 1; }
TR=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_timeline.py $TR 5 > $OUT/timeline.txt
rm -rf $OUT/trace
cat $OUT/out.txt | tail -1
head -30 $OUT/timeline.txt
