import sys, os, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
cfg = bench.WORKLOADS['c4_metric']
for steps in (40, 160):
    out, _ = bench.model_api_measure(cfg, 'c4_metric', steps)
    print(steps, round(out['ms_per_step'], 4), '%.3e' % out['value'], out['mean_N'], flush=True)
