#!/bin/bash
# per-kernel totals of any python command under rocprofv3: tools/kstat_cmd.sh <tag> <script> [args...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT; rm -rf $OUT/trace
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 "$@" > $OUT/out.txt 2> $OUT/rocprof.err || { tail -5 $OUT/rocprof.err; exit 1; }
ST=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $ST $OUT/kernel_stats.csv
rm -rf $OUT/trace
tail -4 $OUT/out.txt
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:34]:
    print('%-58s calls %6s  avg %8.1f us  total %8.2f ms  %5.1f %%' % (r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
