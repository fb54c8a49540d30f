"""Build libgnxhip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m geonomics_amd.build          # or: python geonomics_amd/build.py

hipcc cross-compiles without a GPU; the .so lands next to this file so it
travels with the tree (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libgnxhip.so')
SOURCES = ['gnx_api.hip', 'gnx_kernels_pop.hip', 'gnx_kernels_genome.hip',
           'gnx_kernels_demog.hip', 'gnx_tile.hip', 'gnx_stats.hip', 'gnx_prim.hip', 'gnx_dd.hip', 'gnx_comm.hip']


def _headers():
    # every header under csrc/ plus the public C-ABI header: editing any of
    # them marks every object stale (an old .so must never travel to the GPU box)
    import glob
    return sorted(glob.glob(os.path.join(CSRC, '*.h'))) + \
        sorted(glob.glob(os.path.join(HERE, '..', 'include', '*.h')))


FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
         # IEEE-exact f32/f64 arithmetic (no fma contraction): the parity tests
         # compare against numpy evaluating the same expressions
         '-ffp-contract=off', '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found')


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, '_obj')
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers()
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        ob = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(ob)
        if force or _stale(ob, [sp] + hdrs):
            cmd = [hipcc] + FLAGS + ['-c', sp, '-o', ob]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl', '-lpthread']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print('built', LIB)
