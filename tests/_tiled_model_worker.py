"""One rank of a Model run over several ranks (tests/test_gpu_tiling.py).  The script is
what a user's model script would be: it builds the params and calls make_model / run;
the tiling comes from the torch.distributed environment alone.

    python tests/_tiled_model_worker.py <out.npz> <traits 0|1> <workdir> [mutate] [poisson] [panmixia]

`run_model` is also called directly by the tests that rehearse the ranks as THREADS of one
process (the library's own tile protocol needs a transport that moves device memory between
the ranks: RCCL on a node - which refuses two ranks on one device - or the in-process one).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def run_model(traits, mutate=False, poisson=False, rank=0, world=1, panmixia=False):
    """builds the model, burns it in, walks 15 main steps; returns what the tests compare
    (every rank returns the same: accessors are global)"""
    import geonomics_amd as gnx
    from geonomics_amd.sim.params import ParametersDict
    from test_gpu_model_api import small_params

    p = small_params(seed=4, traits=traits, L=48, T=15, dim=(32, 32))
    if mutate:
        p['comm']['species']['spp_0']['gen_arch'].update({'start_neut_zero': True,
                                                          'mu_neut': 5e-4, 'use_tskit': True})
    if poisson:
        p['comm']['species']['spp_0']['mating'].update({'n_births_fixed': False,
                                                        'n_births_distr_lambda': 1.5})
    if panmixia:      # mating_radius None: pairs drawn from the whole population
        p['comm']['species']['spp_0']['mating']['mating_radius'] = None
    p['model']['stats'] = ParametersDict({'Nt': {'calc': True, 'freq': 1},
                                          'het': {'calc': True, 'freq': 5, 'mean': False},
                                          'maf': {'calc': True, 'freq': 5},
                                          'mean_fit': {'calc': True, 'freq': 5}})
    p['model']['data'] = ParametersDict({
        'sampling': {'scheme': 'random', 'n': 20, 'points': None, 'transect_endpoints': None,
                     'n_transect_points': None, 'radius': None, 'when': None,
                     'include_landscape': False, 'include_fixed_sites': True},
        'format': {'gen_format': 'vcf', 'geo_vect_format': 'csv', 'geo_rast_format': 'txt',
                   'nonneut_loc_format': None}})
    mod = gnx.make_model(p)
    spp = mod.comm[0]
    mod.walk(10000, 'burn', verbose=False)
    nburn = len(spp.Nt)
    n_at_assign = spp.Nt[-1]
    g0 = spp._get_genotypes()                 # right after the genome assignment
    mod.walk(15, 'main', verbose=False)
    ids = np.array([*spp])
    xy = mod.get_coords()
    g = spp._get_genotypes()
    z = mod.get_z() if traits else np.zeros((len(ids), 0))
    het = mod._stats_collector.stats['spp_0']['het']['vals'][14]
    from geonomics_amd.sim.stats import _calc_ld
    ld = _calc_ld(spp, loci=np.arange(0, 48, 3))
    ped_ok = -1
    if spp._tt is not None:        # genotypes read back through the recorded pedigree
        ped_ok = int((spp._tt.genotypes_of(ids) == g).all())
    dens = spp._calc_density()          # (collective on tiles: every rank calls it)
    stepper = getattr(spp, '_stepper', None)
    return dict(Nt=np.array(spp.Nt), births=np.array(spp.n_births),
                deaths=np.array(spp.n_deaths), nburn=nburn, n_at_assign=n_at_assign,
                site_counts0=g0.sum(axis=(0, 2)), n0=g0.shape[0], ids=ids, xy=xy, g=g, z=z,
                het=np.asarray(het), K=spp.K, ped_ok=ped_ok, ld=ld, N_rast=spp.N, dens=dens,
                world=world, v3=int(bool(stepper is not None and stepper.v3)),
                id_order=int(spp._dev.id_order))


if __name__ == '__main__':
    out, traits, workdir = sys.argv[1], bool(int(sys.argv[2])), sys.argv[3]
    flags = sys.argv[4:]
    os.chdir(workdir)
    res = run_model(traits, mutate='mutate' in flags, poisson='poisson' in flags,
                    panmixia='panmixia' in flags,
                    rank=int(os.environ.get('RANK', '0')),
                    world=int(os.environ.get('WORLD_SIZE', 1)))
    if int(os.environ.get('RANK', '0')) == 0:
        np.savez(out, **res)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
