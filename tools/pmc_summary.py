#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_round.sh into profiles/:
kernel-stats CSV (copied), the crossover rows of the PMC passes (copied), and a
JSON with the per-launch HBM traffic of the crossover kernel, corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts half of a
16-B-per-lane coalesced stream -> x2; WRITE_SIZE exact; both in KB)."""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
workload = sys.argv[2] if len(sys.argv) > 2 else 'c4_metric'
wl = workload.split('_')[0] if workload == 'c4_metric' else workload
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'gpurun_out')
prof = os.path.join(root, 'profiles')
os.makedirs(prof, exist_ok=True)


def find(d, suffix):
    hits = glob.glob(os.path.join(out, d, '**', '*' + suffix), recursive=True)
    return hits[0] if hits else None


def counter_rows(d, counter, dst):
    path = find(d, 'counter_collection.csv')
    vals, dur = [], []
    if not path:
        return vals, dur
    with open(path) as f, open(dst, 'w', newline='') as g:
        rd = csv.DictReader(f)
        wr = csv.DictWriter(g, rd.fieldnames)
        wr.writeheader()
        for r in rd:
            if ('k_xo_sparse' in r['Kernel_Name'] or 'k_xo_dense' in r['Kernel_Name']) \
                    and r['Counter_Name'] == counter:
                wr.writerow(r)
                vals.append(float(r['Counter_Value']))
                dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
    return vals, dur


ks = find('prof_' + tag, 'kernel_stats.csv')
if ks:
    shutil.copy(ks, os.path.join(prof, '%s_%s_kernel_stats.csv' % (tag, wl)))
fv, _ = counter_rows('pmc_fetch_' + tag, 'FETCH_SIZE',
                     os.path.join(prof, '%s_pmc_fetch_crossover.csv' % tag))
wv, _ = counter_rows('pmc_write_' + tag, 'WRITE_SIZE',
                     os.path.join(prof, '%s_pmc_write_crossover.csv' % tag))
# timed launches are the last 6 of each run (2 warm-up steps before them)
fv, wv = fv[-6:], wv[-6:]
alg = None
try:
    line = [l for l in open(os.path.join(out, 'pmc_fetch_%s.json' % tag)) if l.startswith('{"metric"')][-1]
    b = json.loads(line)
    alg = b['roofline']['algorithmic_bytes_per_launch']
    kernel = b['roofline']['kernel']
except Exception:
    kernel = 'k_xo'
if fv and wv:
    rd = 2.0 * 1024.0 * sum(fv) / len(fv)
    wrb = 1024.0 * sum(wv) / len(wv)
    js = {'workload': workload, 'kernel': kernel,
          'FETCH_SIZE_KB_mean': sum(fv) / len(fv), 'WRITE_SIZE_KB_mean': sum(wv) / len(wv),
          'correction': 'gfx950: FETCH_SIZE reports 1/2 of a 16-B-per-lane coalesced stream '
                        '(MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact; separate --pmc passes',
          'hbm_read_bytes_per_launch': rd, 'hbm_write_bytes_per_launch': wrb,
          'hbm_bytes_per_launch': rd + wrb, 'algorithmic_bytes_per_launch': alg,
          'note': 'separate runs (6 timed launches each)'}
    with open(os.path.join(prof, '%s_pmc_crossover.json' % tag), 'w') as f:
        json.dump(js, f, indent=1)
    print(json.dumps(js))
for name in ('prof_%s_bench.json' % tag,):
    src = os.path.join(out, name)
    if os.path.exists(src):
        line = [l for l in open(src) if l.startswith('{"metric"')]
        if line:
            open(os.path.join(prof, '%s_%s_bench_under_rocprof.json' % (tag, wl)), 'w').write(line[-1])
