"""Community: serial-integer-keyed dict of Species (reference:
geonomics/structs/community.py:20-149)."""
import numpy as np

from ..sim import burnin
from .species import _make_species


class Community(dict):
    def __init__(self, land, spps):
        self.update(spps)
        self.n_spps = len(spps)
        self.t = -1
        self.burned = False

    def __str__(self):
        return '%s\n%i Species: %s' % (str(type(self)), len(self), ', '.join(
            "'%s' (%i inds.)" % (v.name, len(v)) for v in self.values()))

    __repr__ = __str__

    def _set_t(self):
        self.t += 1

    def _reset_t(self):
        self.t = -1

    def _check_burned(self, burn_T, params=None):
        """reference structs/community.py:107-131: minimum burn-in time, then ADF
        and paired-t tests on Nt and the spatial test, for every species."""
        status = bool(np.all([len(spp.Nt) >= burn_T for spp in self.values()]))
        if status:
            adf_tests = np.all([burnin._test_adf_threshold(spp, burn_T)
                                for spp in self.values()])
            t_tests = np.all([burnin._test_t_threshold(spp, burn_T)
                              for spp in self.values()])
            spat_tests = np.all([spp._do_spatial_burnin_test(burn_T)
                                 for spp in self.values()])
            status = bool(adf_tests and t_tests and spat_tests)
        self.burned = status
        for spp in self.values():
            spp.burned = status


def _make_community(land, params, burn=False, verbose=False, seed=0, device=0, rng=None,
                    comm=None):
    if verbose:
        print('\tMAKING COMMUNITY...\n')
    spps = {n: _make_species(land=land, name=name, idx=n,
                             spp_params=params.comm.species[name], burn=burn,
                             verbose=verbose, seed=seed + 7919 * n, device=device, rng=rng,
                             comm=comm)
            for n, name in enumerate(params.comm.species.keys())}
    return Community(land, spps)
