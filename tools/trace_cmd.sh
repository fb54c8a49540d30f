#!/bin/bash
# rocprofv3 kernel trace of any python command of this repo + the timeline of one steady step
# (tools/trace_timeline.py) and the GPU-busy share (tools/busy.py):
#   tools/trace_cmd.sh <tag> <script.py> [args...]      (env assignments: export them before)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd "$ROOT"
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 "$@" > $OUT/stdout.txt 2> $OUT/rocprof.err || { tail -5 $OUT/rocprof.err; exit 1; }
TR=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
ST=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 tools/trace_timeline.py $TR > $OUT/timeline.txt 2>&1
python3 tools/busy.py $TR ${BUSY_TAIL_MS:-30} > $OUT/busy.txt 2>&1
cp $ST $OUT/kernel_stats.csv
rm -rf $OUT/trace
head -3 $OUT/stdout.txt
