#!/bin/bash
# per-kernel average durations of a short kbench run under rocprofv3: tools/kstat.sh <tag> [env assignments...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT; rm -rf $OUT/trace
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 tools/kbench.py --genomes --steps 30 --no-profile > $OUT/kbench.txt 2> $OUT/rocprof.err || { tail -5 $OUT/rocprof.err; exit 1; }
ST=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $ST $OUT/kernel_stats.csv
rm -rf $OUT/trace
head -1 $OUT/kbench.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/kernel_stats.csv')))
for r in rows[:24]:
    print('%-60s calls %6s  avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
