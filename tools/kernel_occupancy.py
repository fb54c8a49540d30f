#!/usr/bin/env python3
"""Registers, LDS and waves per SIMD of the step's kernels, from the code objects (no GPU):
    python tools/kernel_occupancy.py [substring ...]
A kernel whose grid is just above what the chip holds at once runs in two rounds (the job builder did:
profiles/r06_ab_runs.txt)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin/'
want = sys.argv[1:]
tmp = tempfile.mkdtemp()
for f in ['gnx_kernels_pop', 'gnx_kernels_demog', 'gnx_kernels_genome', 'gnx_prim', 'gnx_tile']:
    co, elf = '%s/%s.co' % (tmp, f), '%s/%s.elf' % (tmp, f)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
                    '--cuda-device-only', '-Wno-unused-function', '-c', '%s/geonomics_amd/csrc/%s.hip' % (ROOT, f),
                    '-o', co], check=True, stderr=subprocess.DEVNULL)
    subprocess.run([LLVM + 'clang-offload-bundler', '--unbundle', '--type=o', '--input=' + co,
                    '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + elf], check=True)
    out = subprocess.run([LLVM + 'llvm-readelf', '--notes', elf], capture_output=True, text=True).stdout
    for p in re.split(r'\n  - ', out):
        m = re.search(r'\.name:\s+(\S+)', p)
        if not m:
            continue
        sym = m.group(1)
        name = subprocess.run(['c++filt', sym], capture_output=True, text=True).stdout.split('(')[0].strip()
        if want and not any(w in name for w in want):
            continue
        g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, p).group(1))
        vg = g('vgpr_count') + g('agpr_count')
        wps = min(8, 512 // max(8, (vg + 7) // 8 * 8))
        print('%-56s vgpr %3d  waves/SIMD %d  lds %6d  wg<=%4d  scratch %d' %
              (name[:56], vg, wps, g('group_segment_fixed_size'), g('max_flat_workgroup_size'),
               g('private_segment_fixed_size')))
