#!/bin/bash
# rocprofv3 kernel trace of a short bench run + the timeline of one steady step
# (tools/trace_timeline.py).  Run through gpurun from the repo root:
#   tools/timeline.sh <tag> [workload] [extra env assignments...]
TAG=${1:-tl}
WL=${2:-c4_metric}
shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
export GNX_BENCH_NO_ALT=1
export GNX_BENCH_NO_PHASES=1      # (the stepper's phase marks synchronise the device)
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd "$ROOT"
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-model-api --no-other-workloads --steady-warmup 0 > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err || { tail -5 $OUT/rocprof.err; exit 1; }
TR=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
ST=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 tools/trace_timeline.py $TR > $OUT/timeline.txt
cp $ST $OUT/kernel_stats.csv
rm -rf $OUT/trace
python3 - <<PY
import json
j = json.load(open('$OUT/bench_under_rocprof.json'))
print('under rocprof: %.3f ms/step  %.3e ind-steps/s  xo %.3f ms frac %.3f' % (j['ms_per_step'], j['value'], j['roofline']['avg_launch_ms'], j['roofline']['frac']))
PY
head -60 $OUT/timeline.txt
