#!/bin/bash
# SQ counters of ONE kernel of the metric step after <warm> steps (kernels serialised; each counter set in a run of its own):
#   tools/pmc_kernel_steady.sh <tag> <kernel substring> <warm>   -> gpurun_out/<tag>/summary.txt (means of the last 10 launches)
TAG=${1:-pmck}; KERN=${2:-k_xo_jobs_lanes}; WARM=${3:-1500}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd "$ROOT"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" \
           ; do
  rm -rf $O/run$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/run$i -o run -- python3 tools/steady_ab.py $WARM 12 > $O/run$i.txt 2> $O/run$i.err || { tail -5 $O/run$i.err; exit 1; }
  i=$((i+1))
done
python3 - "$O" "$KERN" <<'PY'
import collections, csv, glob, sys
o, kern = sys.argv[1], sys.argv[2]
out = []
for path in sorted(glob.glob(o + '/run*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(path)):
        if kern in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for c, v in acc.items():
        v = v[-10:]
        out.append('%-40s %16.1f' % (c, sum(v) / len(v)))
    if dur:
        out.append('%-40s %16.1f' % ('duration_us', sum(dur[-70:]) / len(dur[-70:])))
open(o + '/summary.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/run0 $O/run1
