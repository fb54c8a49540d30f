#!/usr/bin/env python3
"""Requests per 640-byte block of tools/pmc_calib's kernels, from the two rocprofv3 --pmc passes
(gpurun_out/pmc_calib_rd, gpurun_out/pmc_calib_wr) -> profiles/<tag>_pmc_calibration.json"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r05'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_blocks = 350000
res = collections.defaultdict(dict)
for d in ('pmc_calib_rd', 'pmc_calib_wr'):
    for path in glob.glob(os.path.join(root, 'gpurun_out', d, '**', '*counter_collection.csv'),
                          recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if 'k_calib' in r['Kernel_Name']:
                acc[(r['Kernel_Name'], r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in acc.items():
            name = k[k.index('k_calib'):].split('(')[0]
            res[name][c + '_per_block'] = round(sum(v[1:]) / len(v[1:]) / n_blocks, 4)
out = {'what': 'a wave copies one random 128-byte-aligned 640-byte block (40 lanes x 16 B): 5 lines '
               'read, 5 written; <pool GiB, extra single-lane 16-B load from a third block>; '
               'mean of launches 2-4',
       'blocks_per_launch': n_blocks, 'kernels': res}
json.dump(out, open(os.path.join(root, 'profiles', '%s_pmc_calibration.json' % tag), 'w'), indent=1)
print(json.dumps(out, indent=1))
