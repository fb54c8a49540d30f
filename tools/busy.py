#!/usr/bin/env python3
"""GPU busy share of a rocprofv3 kernel trace: tools/busy.py <kernel_trace.csv> [tail ms]
(union of the kernels' intervals over the last `tail ms` of the trace) and the kernels by time."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
t1 = iv[-1][1]
t0 = t1 - int(tail_ms * 1e6)
iv = [v for v in iv if v[0] >= t0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print('window %.2f ms, %d dispatches, GPU busy %.2f ms = %.0f %%' % (span / 1e6, len(iv), busy / 1e6,
                                                                   100.0 * busy / span))
acc = collections.defaultdict(float)
cnt = collections.Counter()
for s, e, k in iv:
    acc[k.split('(')[0][:60]] += (e - s) / 1e6
    cnt[k.split('(')[0][:60]] += 1
n_steps = max(1, cnt.get('k_alive', 1))          # (one per tile and step)
print('  (%d tile-steps in the window: per tile-step below)' % n_steps)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:40]:
    print('  %8.2f ms  %6.1f us  x%-5.1f %s' % (v, 1e3 * v / n_steps, cnt[k] / n_steps, k))
