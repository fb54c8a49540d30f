#!/bin/bash
# Every kernel of the metric step ALONE (under --pmc the dispatches are serialised) with its HBM traffic, after <warm> steps:
#   tools/pmc_steady.sh <tag> <warm>  -> gpurun_out/<tag>/summary.txt: per kernel mean us, FETCH_SIZE and WRITE_SIZE (KB) of the last launches
TAG=${1:-pmcs}; WARM=${2:-1500}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
cd "$ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o run -- python3 tools/steady_ab.py $WARM 12 > $O/$c.txt 2> $O/$c.err || { tail -5 $O/$c.err; exit 1; }
done
python3 - "$O" <<'PY'
import collections, csv, glob, sys
o = sys.argv[1]
dur = collections.defaultdict(list)
val = {'FETCH_SIZE': collections.defaultdict(list), 'WRITE_SIZE': collections.defaultdict(list)}
for c in val:
    for path in glob.glob(o + '/' + c + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if r['Counter_Name'] == c:
                val[c][name].append(float(r['Counter_Value']))
                if c == 'FETCH_SIZE':
                    dur[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
rows = []
for name, d in dur.items():
    n = max(1, min(10, len(d) // 2))
    rows.append((sum(d[-n:]) / n, name, len(d), sum(val['FETCH_SIZE'][name][-n:]) / n,
                 sum(val['WRITE_SIZE'][name][-n:]) / n if val['WRITE_SIZE'][name] else float('nan')))
rows.sort(reverse=True)
out = ['%-44s %8s %8s %12s %12s' % ('kernel (alone: --pmc serialises)', 'launches', 'us', 'FETCH KB', 'WRITE KB')]
for us, name, n, f, w in rows[:32]:
    out.append('%-44s %8d %8.1f %12.0f %12.0f' % (name[:44], n, us, f, w))
open(o + '/summary.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
