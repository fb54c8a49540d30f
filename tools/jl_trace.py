#!/usr/bin/env python3
"""When each workgroup of the job builder ran (a -DGNX_JL_TRACE build: tools/build_variant.sh jltrace gnx_kernels_demog.hip -DGNX_JL_TRACE;
GNX_LIB=tools/_variants/libgnxhip_jltrace.so python tools/jl_trace.py <warm>)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cfg = bench.WORKLOADS['c4_metric']
dev, _, _ = bench.build_device(cfg, 42, 0)
for _ in range(3):
    dev.step(True, False)
bench.setup_genomes(dev, cfg, 42)
dev.walk(warm, False, True)
dev.synchronize()
buf = np.zeros(4 * 4096, np.uint64)
rc = dev.lib.gnx_debug_jl_trace(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
t = buf.reshape(4096, 4)
on = t[:, 1] > 0
t = t[on]
t0 = t[:, 0].min()
st = (t[:, 0] - t0) / 100.0          # wall clock: 100 MHz
en = (t[:, 1] - t0) / 100.0
cyc = t[:, 2]
xcc = (t[:, 3] >> np.uint64(32)).astype(int) & 0xf
print('N', dev.N, 'workgroups', len(t), 'span %.1f us' % en.max())
print('start  us: p0 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f' % tuple(np.percentile(st, [0, 50, 90, 99, 100])))
print('length us: p0 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f' % tuple(np.percentile(en - st, [0, 50, 90, 99, 100])))
print('cycles   : p50 %d p99 %d' % tuple(np.percentile(cyc, [50, 99])))
order = np.argsort(st)
for k in list(range(0, len(t), max(1, len(t) // 24))):
    i = order[k]
    print('  wg %4d  xcc %d  start %6.1f  end %6.1f' % (np.nonzero(on)[0][i], xcc[i], st[i], en[i]))
for x in range(8):
    m = xcc == x
    if m.any():
        print('  xcc %d: %3d workgroups, last end %.1f' % (x, m.sum(), en[m].max()))
