#!/bin/bash
# Counters of ONE kernel of the metric step, kernels serialised (rocprofv3 --pmc, each set of
# counters in a run of its own, --kernel-trace only): tools/pmc_kernel.sh <tag> <kernel substring>
#   -> gpurun_out/<tag>/summary.txt: per launch means of the last 10 launches
TAG=${1:-pmck}
KERN=${2:-k_xo_jobs_fused}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd "$ROOT"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" \
           ; do
  # (a set with TA_BUSY_avr / TA_*_STALLED_* aborted inside rocprofv3 on this image and hung the
  # run: do not add it back without trying it alone under a short timeout)
  rm -rf $O/run$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/run$i -o run -- python3 tools/kbench.py --genomes --steps 12 --no-profile > $O/run$i.txt 2> $O/run$i.err
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
    for c, v in acc.items():
        v = v[-10:]
        out.append('%-40s %16.1f' % (c, sum(v) / len(v)))
open(o + '/summary.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
