import sys, os
sys.path.insert(0, '/root/repo')
import bench
from geonomics_amd import _native as nat
cfg = bench.WORKLOADS['c4_metric']
dev, _, _ = bench.build_device(cfg, 42, 0)
for _ in range(3):
    dev.step(True, False)
bench.setup_genomes(dev, cfg, 42)
dev.walk(int(sys.argv[1]), False, True)
if len(sys.argv) > 2:
    dev.set_crossover_overlap(int(sys.argv[2]))      # 2: nothing runs beside the crossover
fam = bench.kernel_profile(dev, lambda burn: dev.step(burn, not burn), 10)
print('N', dev.N)
for k, v in fam.items():
    print('%-12s %.4f ms' % (k, v['ms_per_step']))
