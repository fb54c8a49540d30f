#!/bin/bash
# Round 4's profile set (run through gpurun from the repo root): everything DESIGN.md quotes,
# from the tree as it is.  Outputs under gpurun_out/r04/ (copied to profiles/ afterwards).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
# (gpurun limits a call to 20 minutes: tools/profile_r04.sh a | b | c, or everything)
PART=${1:-abc}
if [[ $PART == *a* ]]; then
tools/launch_micro > $O/launch_micro.txt 2>&1
echo "micro done"
tools/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -1 $O/profile_round.log | cut -c1-200
tools/profile_others.sh r04 > $O/profile_others.log 2>&1; tail -3 $O/profile_others.log
tools/timeline.sh r04_tl c4_metric > $O/timeline_c4.log 2>&1; head -1 $O/timeline_c4.log
for wl in c2 c3; do tools/timeline_walk.sh r04_${wl}_walk $wl > $O/timeline_${wl}_walk.log 2>&1; head -2 $O/timeline_${wl}_walk.log | tail -1; done
fi
if [[ $PART == *b* ]]; then
# the host's share of a host-driven step at 10^5 individuals
GNX_HOST_TIMES=1 python3 tools/kbench.py --workload c2 --genomes --steps 300 --no-profile 2>&1 | grep -E "host times|^N=" > $O/host_times_c2.txt
# several handles on one GPU
{ GNX_DD_STREAMS=1 python3 tools/its_raw.py c2 8 300; python3 tools/its_raw.py c2 2 300 walk_many; python3 tools/its_raw.py c2 4 300 walk_many; } > $O/its_raw.txt 2>&1
python3 tools/its_bench.py c2 --its 8 --T 1500 > $O/its_bench.txt 2>&1
echo "iterations done"
fi
if [[ $PART == *c* ]]; then
# tiles: two tiles as threads, the library-driven step against the Python-driven one; GPU busy share
{ python3 tools/tile_thread_bench.py 2 20; echo "--- GNX_TILE_V3=0 (TiledStepper._step_v2)"; GNX_TILE_PROFILE= GNX_TILE_V3=0 python3 tools/tile_thread_bench.py 2 20; } 2>&1 | grep -E "rank|---" > $O/tile_threads.txt
rocprofv3 --kernel-trace --output-format csv -d $O/busy_trace -o run -- python3 tools/tile_thread_bench.py 2 30 > $O/busy_out.txt 2>&1
TR=$(find $O/busy_trace -name "*kernel_trace.csv" | head -1); python3 tools/busy.py $TR 12 > $O/tile_threads_busy.txt; rm -rf $O/busy_trace
head -1 $O/tile_threads_busy.txt
GNX_BENCH_FORCE_STEPPER=1 python3 bench.py --no-cpu-baseline --no-model-api --no-other-workloads --steady-warmup 0 > $O/stepper_one_tile.json 2>/dev/null
{ for v in "" "--tile-step"; do echo "[kbench.py $v]"; GNX_HOST_TIMES=2 python3 tools/kbench.py --genomes --steps 300 --no-profile $v 2>&1 | grep -E "^N=|host marks"; done; } > $O/tile_step_one_rank.txt
echo "tiles done"
python3 bench.py > $O/bench_final.json 2> $O/bench_final.err
python3 - <<PY
import json
j = json.load(open('$O/bench_final.json'))
print('final:', j['ms_per_step'], j['value'], j['roofline']['frac'], json.dumps(j.get('summary')))
PY
fi
