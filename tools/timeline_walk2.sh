#!/bin/bash
# timeline of one step of gnx_walk at the metric workload (tools/kbench.py --genomes --walk under rocprofv3):
#   tools/timeline_walk2.sh <tag> [env assignments...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT; rm -rf $OUT/trace
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 tools/kbench.py --genomes --steps 30 --no-profile --walk > $OUT/kbench.txt 2> $OUT/rocprof.err || { tail -5 $OUT/rocprof.err; exit 1; }
TR=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_timeline.py $TR 5 > $OUT/timeline.txt
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
head -1 $OUT/kbench.txt
head -48 $OUT/timeline.txt
