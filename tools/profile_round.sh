#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 kernel trace + stats of the default bench command
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE), each in its own run with --kernel-trace only
# Outputs land under gpurun_out/; tools/pmc_summary.py condenses them into profiles/.
set -e
TAG=${1:-r02}
WL=${2:-c4_metric}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
# only the contract's timed region: no second measurement in the other overlap mode
export GNX_BENCH_NO_ALT=1
OUT=$ROOT/gpurun_out
cd "$ROOT"
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o run -- python3 bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-model-api --no-other-workloads > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}.err
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -o run -- python3 bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-model-api --no-other-workloads > $OUT/pmc_fetch_${TAG}.json 2> $OUT/pmc_fetch_${TAG}.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -o run -- python3 bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-model-api --no-other-workloads > $OUT/pmc_write_${TAG}.json 2> $OUT/pmc_write_${TAG}.err
echo "pmc write done"
python3 tools/pmc_summary.py $TAG $WL
