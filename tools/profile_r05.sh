#!/bin/bash
# Round 5's profile set (run through gpurun from the repo root): everything DESIGN.md quotes,
# from the tree as it is.  Outputs under gpurun_out/r05/ (copied to profiles/ afterwards).
#   tools/profile_r05.sh a | b | c   (gpurun limits a call to 20 minutes)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
PART=${1:-abc}
if [[ $PART == *a* ]]; then
# kernel stats + the four PMC passes of the default bench command, the counters' calibration
tools/profile_round.sh r05 > $O/profile_round.log 2>&1; tail -1 $O/profile_round.log | cut -c1-200
[ -x tools/pmc_calib ] || hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip
rm -rf gpurun_out/pmc_calib_rd gpurun_out/pmc_calib_wr
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --kernel-trace --output-format csv -d gpurun_out/pmc_calib_rd -o run -- ./tools/pmc_calib > $O/pmc_calib_rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d gpurun_out/pmc_calib_wr -o run -- ./tools/pmc_calib > $O/pmc_calib_wr.log 2>&1
echo "pmc done"
tools/timeline.sh r05_tl c4_metric > $O/timeline_c4.log 2>&1; head -1 $O/timeline_c4.log
for wl in c2 c3; do tools/timeline_walk.sh r05_${wl}_walk $wl > $O/timeline_${wl}_walk.log 2>&1; head -2 $O/timeline_${wl}_walk.log | tail -1; done
fi
if [[ $PART == *b* ]]; then
# tiles: two tiles of the metric workload as threads through gnx_tile_step; GPU busy share
python3 tools/tile_thread_bench.py 2 40 2>&1 | grep -E "rank|ms/step" > $O/tile_threads.txt
rocprofv3 --kernel-trace --output-format csv -d $O/busy_trace -o run -- python3 tools/tile_thread_bench.py 2 30 > $O/busy_out.txt 2>&1
TR=$(find $O/busy_trace -name "*kernel_trace.csv" | head -1); python3 tools/busy.py $TR 14 > $O/tile_threads_busy.txt; rm -rf $O/busy_trace
head -1 $O/tile_threads_busy.txt
# one tile through the tile protocol (the Model API's and the multi-GPU bench's path) against the plain step
GNX_BENCH_FORCE_STEPPER=1 python3 bench.py --no-cpu-baseline --no-model-api --no-other-workloads --steady-warmup 0 > $O/stepper_one_tile.json 2>/dev/null
{ for v in "" "--tile-step"; do echo "[kbench.py $v]"; GNX_HOST_TIMES=2 python3 tools/kbench.py --genomes --steps 300 --no-profile $v 2>&1 | grep -E "^N=|host marks"; done; } > $O/tile_step_one_rank.txt
tools/timeline.sh r05_stepper_tl c4_metric GNX_BENCH_FORCE_STEPPER=1 > $O/timeline_stepper.log 2>&1; head -1 $O/timeline_stepper.log
echo "tiles done"
fi
if [[ $PART == *c* ]]; then
python3 bench.py > $O/bench_final.json 2> $O/bench_final.err
python3 - <<PY
import json
j = json.load(open('$O/bench_final.json'))
print('final:', j['ms_per_step'], j['value'], j['roofline']['frac'], json.dumps(j.get('summary')))
PY
fi
