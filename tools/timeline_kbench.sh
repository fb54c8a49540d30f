#!/bin/bash
# rocprofv3 kernel trace of tools/kbench.py --walk + the timeline of one steady step:
#   tools/timeline_kbench.sh <tag> <workload> [env assignments...]
TAG=${1:-tlk}
WL=${2:-c4_metric}
shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd "$ROOT"
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 tools/kbench.py --workload $WL --genomes --steps 30 --no-profile --walk > $OUT/kbench.txt 2> $OUT/rocprof.err || { tail -5 $OUT/rocprof.err; exit 1; }
TR=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_timeline.py $TR > $OUT/timeline.txt
rm -rf $OUT/trace
head -1 $OUT/kbench.txt
