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
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o run -- python3 bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-model-api --no-other-workloads --steady-warmup 0 > $OUT/prof_${TAG}_bench.json 2> $OUT/prof_${TAG}.err
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -o run -- python3 bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-model-api --no-other-workloads --steady-warmup 0 > $OUT/pmc_fetch_${TAG}.json 2> $OUT/pmc_fetch_${TAG}.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -o run -- python3 bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-model-api --no-other-workloads --steady-warmup 0 > $OUT/pmc_write_${TAG}.json 2> $OUT/pmc_write_${TAG}.err
echo "pmc write done"
# 3./4. the request counters FETCH_SIZE / WRITE_SIZE are derived from, to attribute the read side
# (requests by size: 32-byte ones, the rest tallied at 64; TCC_BUBBLE = 128-byte reads)
rm -rf $OUT/pmc_rdreq_$TAG $OUT/pmc_wrreq_$TAG
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace --output-format csv -d $OUT/pmc_rdreq_$TAG -o run -- python3 bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-model-api --no-other-workloads --steady-warmup 0 > $OUT/pmc_rdreq_${TAG}.json 2> $OUT/pmc_rdreq_${TAG}.err
echo "pmc rdreq done"
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/pmc_wrreq_$TAG -o run -- python3 bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-model-api --no-other-workloads --steady-warmup 0 > $OUT/pmc_wrreq_${TAG}.json 2> $OUT/pmc_wrreq_${TAG}.err
echo "pmc wrreq done"
python3 tools/pmc_summary.py $TAG $WL
# what the summary left under profiles/ travels back with gpurun_out/ (profiles/ on the box does not)
mkdir -p $OUT/profiles_$TAG
cp profiles/${TAG}_* $OUT/profiles_$TAG/ 2>/dev/null
rm -rf $OUT/prof_$TAG/*/ $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/pmc_rdreq_$TAG $OUT/pmc_wrreq_$TAG 2>/dev/null; find $OUT/prof_$TAG -name "*kernel_trace.csv" -delete 2>/dev/null; true
