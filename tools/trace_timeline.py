#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV: every
dispatch between two consecutive crossover launches with its start offset, its
duration and the idle gap of the chip before it (no kernel of any stream
running).  Usage: trace_timeline.py <kernel_trace.csv> [step index from the end]"""
import csv
import sys

path = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                     r.get('Queue_Id', '')))
rows.sort()
xo = [i for i, r in enumerate(rows) if 'k_xo_' in r[2] and 'jobs' not in r[2]]
if len(xo) < back + 2:
    sys.exit('not enough crossover launches in the trace')
a, b = xo[-back - 1], xo[-back]
t0 = rows[a][0]
period = (rows[b][0] - t0) * 1e-3
print('step period %.1f us (crossover start to crossover start)' % period)
busy_end = rows[a][0]
idle = 0.0
queues = {}
for s, e, name, q in rows[a:b]:
    gap = max(0, s - busy_end) * 1e-3
    idle += gap
    busy_end = max(busy_end, e)
    qi = queues.setdefault(q, len(queues))
    short = name.split('(')[0]
    if len(short) > 70:
        short = short[:67] + '...'
    print('%9.1f  q%d  %8.1f us  gap %6.1f  %s' % ((s - t0) * 1e-3, qi, (e - s) * 1e-3, gap, short))
print('chip idle inside the step: %.1f us of %.1f' % (idle, period))
# totals over the last 10 steps by kernel
lo = rows[xo[-11]][0] if len(xo) > 11 else rows[xo[0]][0]
hi = rows[xo[-1]][0]
tot = {}
for s, e, name, q in rows:
    if lo <= s < hi:
        k = name.split('(')[0].split('<')[0]
        tot[k] = tot.get(k, 0) + (e - s)
n = min(10, len(xo) - 1)
print('\nper-step kernel time over the last %d steps (us):' % n)
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print('%9.1f  %s' % (v * 1e-3 / n, k[:90]))
