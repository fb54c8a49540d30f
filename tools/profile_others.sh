#!/bin/bash
# rocprofv3 kernel stats of the other BASELINE workloads (run through gpurun from the repo
# root): tools/profile_others.sh <tag>  ->  gpurun_out/prof_<tag>_<workload>/...
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
export GNX_BENCH_NO_ALT=1
OUT=$ROOT/gpurun_out
cd "$ROOT"
for WL in c2 c3 c4_dense; do
  rm -rf $OUT/prof_${TAG}_$WL
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$WL -o run -- python3 bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-model-api --no-other-workloads --steady-warmup 0 > $OUT/prof_${TAG}_${WL}_bench.json 2> $OUT/prof_${TAG}_${WL}.err || tail -3 $OUT/prof_${TAG}_${WL}.err
  ST=$(find $OUT/prof_${TAG}_$WL -name '*kernel_stats.csv' | head -1)
  TR=$(find $OUT/prof_${TAG}_$WL -name '*kernel_trace.csv' | head -1)
  cp $ST $OUT/${TAG}_${WL}_kernel_stats.csv
  python3 tools/trace_timeline.py $TR > $OUT/${TAG}_${WL}_timeline.txt 2>&1
  rm -rf $OUT/prof_${TAG}_$WL
  echo "$WL done: $(python3 -c "import json; j=json.load(open('$OUT/prof_${TAG}_${WL}_bench.json')); print(j['ms_per_step'], j['value'])")"
done
