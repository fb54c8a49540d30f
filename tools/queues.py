#!/usr/bin/env python3
"""Which hardware queues the kernels of a rocprofv3 kernel trace ran on and how much they
overlapped: tools/queues.py <kernel_trace.csv>"""
import csv
import sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', ''),
                     r['Kernel_Name'].split('(')[0][:40]))
rows.sort()
rows = rows[len(rows) // 2:]                      # the steady half
t0, t1 = rows[0][0], max(r[1] for r in rows)
per_q = {}
for s, e, q, _ in rows:
    per_q[q] = per_q.get(q, 0) + (e - s)
busy = 0
end = t0
for s, e, q, _ in rows:
    if e > end:
        busy += e - max(s, end)
        end = e
span = t1 - t0
print('span %.1f us, chip busy %.1f us (%.0f %%), sum of kernel times %.1f us -> mean overlap x%.2f' % (
    span * 1e-3, busy * 1e-3, 100.0 * busy / span, sum(per_q.values()) * 1e-3,
    sum(per_q.values()) / max(busy, 1)))
for q, v in sorted(per_q.items()):
    print('  queue %s: %.1f us of kernels' % (q, v * 1e-3))
for s, e, q, n in rows[:60]:
    print('%9.1f %7.1f q%s %s' % ((s - t0) * 1e-3, (e - s) * 1e-3, q, n))
