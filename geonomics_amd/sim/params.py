"""Parameters-file format of Geonomics (reference: geonomics/sim/params.py).

A parameters file is Python source that defines one nested dict called
`params` with the sections 'landscape', 'comm' and 'model'; it is exec'd and
wrapped in a ParametersDict (dot access at every level).  This module keeps
that format and the key names of the reference's template (sim/params.py:
36-710) so existing files load unchanged; the template text itself is
generated from the defaults tables below rather than stored as one string.

Differences kept deliberately:
  * the exec namespace is pre-seeded with `np`, because the reference's own
    'defined'-layer template uses np.ones(...) (sim/params.py:180 vs 1129-1131);
  * both params.model.seed.num (read by the code, sim/model.py:98-99) and
    params.model.num (documented) seed the model.
"""
import copy
import os
import time

import numpy as np

_DICT_METHODS = ['clear', 'copy', 'fromkeys', 'get', 'items', 'keys', 'pop',
                 'popitem', 'setdefault', 'update', 'values']


class _DynAttrDict(dict):
    """dict whose keys are also attributes (reference sim/params.py:719-727)."""

    def __getattr__(self, item):
        try:
            return self[item]
        except KeyError:
            raise AttributeError(item)

    def __setattr__(self, key, value):
        self[key] = value

    def __dir__(self):
        return list(super().__dir__()) + [str(k) for k in self.keys()]

    def __deepcopy__(self, memo):
        return _wrap(copy.deepcopy(dict(self), memo))


def _wrap(d):
    for k, v in d.items():
        if k in _DICT_METHODS:
            raise ValueError('The key "%s" in your params file is disallowed '
                             'because it would clobber a Python dict method.' % k)
        if isinstance(v, dict):
            d[k] = _wrap(dict(v))
    return _DynAttrDict(d)


class ParametersDict(_DynAttrDict):
    """Nested parameters with dot access (reference sim/params.py:730-755)."""

    def __init__(self, params):
        super().__init__()
        self.update(_wrap(dict(params)))

    def __deepcopy__(self, memo):
        return ParametersDict(copy.deepcopy(dict(self), memo))

    def __str__(self):
        name = self.get('model', {}).get('name', None)
        return "%s\nModel name:%s%s" % (str(type(self)), ' ' * 30, name)

    __repr__ = __str__


# ---------------------------------------------------------------------------
# defaults (same keys and default values as the reference's template)
# ---------------------------------------------------------------------------

class _Code(str):
    """A value emitted verbatim into the generated file."""


def lyr_init_defaults(kind):
    if kind == 'random':
        return {'random': {'n_pts': 500, 'interp_method': 'linear'}}
    if kind == 'defined':
        return {'defined': {'rast': np.ones((20, 20)), 'pts': None,
                            'vals': None, 'interp_method': None}}
    if kind == 'file':
        return {'file': {'filepath': '/PATH/TO/FILE.EXT', 'scale_min_val': None,
                         'scale_max_val': None, 'coord_prec': 5, 'units': None}}
    if kind == 'nlmpy':
        return {'nlmpy': {'function': 'mpd', 'nRow': 20, 'nCol': 20, 'h': 1}}
    raise ValueError("layer type must be 'random', 'defined', 'file' or 'nlmpy'")


def lyr_change_event_defaults():
    return {'change_rast': '/PATH/TO/FILE.EXT', 'start_t': 49, 'end_t': 99,
            'n_steps': 5}


def spp_defaults():
    return {
        'init': {'N': 250, 'K_layer': 'lyr_0', 'K_factor': 1},
        'mating': {'repro_age': 0, 'sex': False, 'sex_ratio': 1 / 1, 'R': 0.5,
                   'b': 0.2, 'n_births_distr_lambda': 1, 'n_births_fixed': True,
                   'mating_radius': 10, 'choose_nearest_mate': False,
                   'inverse_dist_mating': False},
        'mortality': {'max_age': None, 'd_min': 0, 'd_max': 1,
                      'density_grid_window_width': None},
        'movement': {'move': True, 'direction_distr_mu': 0,
                     'direction_distr_kappa': 0,
                     'movement_distance_distr_param1': 0.01,
                     'movement_distance_distr_param2': 0.5,
                     'movement_distance_distr': 'lognormal',
                     'dispersal_distance_distr_param1': -1,
                     'dispersal_distance_distr_param2': 0.05,
                     'dispersal_distance_distr': 'lognormal'},
    }


def surf_defaults():
    return {'layer': 'lyr_0', 'mixture': True, 'vm_distr_kappa': 12,
            'approx_len': 5000}


def gen_arch_defaults():
    return {'gen_arch_file': None, 'L': 100, 'start_p_fixed': 0.5,
            'start_neut_zero': False, 'mu_neut': 0, 'mu_delet': 0,
            'delet_alpha_distr_shape': 0.2, 'delet_alpha_distr_scale': 0.2,
            'r_distr_alpha': 0.5, 'r_distr_beta': None, 'dom': False,
            'pleiotropy': False, 'recomb_rate_custom_fn': None,
            'n_recomb_paths_mem': int(1e4), 'n_recomb_paths_tot': int(1e5),
            'n_recomb_sims': 10_000, 'allow_ad_hoc_recomb': False,
            'jitter_breakpoints': False, 'mut_log': False, 'use_tskit': True,
            'tskit_simp_interval': 100}


def trait_defaults():
    return {'layer': 'lyr_0', 'phi': 0.05, 'n_loci': 1, 'mu': 0,
            'alpha_distr_mu': 0.1, 'alpha_distr_sigma': 0, 'max_alpha_mag': None,
            'gamma': 1, 'univ_adv': False}


def dem_change_event_defaults():
    return {'kind': 'monotonic', 'start_t': 49, 'end_t': 99, 'rate': 1.02,
            'interval': 1, 'distr': 'uniform', 'n_cycles': 10,
            'size_range': (0.5, 1.5), 'timesteps': [50, 90, 95],
            'sizes': [2, 5, 0.5]}


def its_defaults():
    return {'n_its': 1, 'rand_landscape': False, 'rand_comm': False,
            'rand_genarch': True, 'repeat_burn': False}


def data_defaults():
    return {'sampling': {'scheme': 'random', 'n': 250, 'points': None,
                         'transect_endpoints': None, 'n_transect_points': None,
                         'radius': None, 'when': None,
                         'include_landscape': False,
                         'include_fixed_sites': False},
            'format': {'gen_format': ['vcf', 'fasta'], 'geo_vect_format': 'csv',
                       'geo_rast_format': 'geotiff',
                       'nonneut_loc_format': None}}


def stats_defaults():
    return {'Nt': {'calc': True, 'freq': 1},
            'het': {'calc': True, 'freq': 5, 'mean': False},
            'maf': {'calc': True, 'freq': 5},
            'mean_fit': {'calc': True, 'freq': 5},
            'ld': {'calc': False, 'freq': 100}}


def default_params_dict(layers=1, species=1, data=None, stats=None):
    """The nested dict that make_parameters_file writes out."""
    lyrs = {}
    if isinstance(layers, int):
        assert layers > 0, 'The number of Layers must be a positive integer.'
        layers = [{'type': 'random'} for _ in range(layers)]
    assert isinstance(layers, list) and all(isinstance(d, dict) for d in layers), (
        "'layers' must be an int or a list of dicts")
    for i, ld in enumerate(layers):
        kind = ld.get('type', 'random')
        lyr = {'init': lyr_init_defaults(kind)}
        ch = ld.get('change', False)
        if ch is not False and ch is not None:
            n = 1 if ch is True else int(ch)
            lyr['change'] = {k: lyr_change_event_defaults() for k in range(n)}
        lyrs['lyr_%i' % i] = lyr
    spps = {}
    if isinstance(species, int):
        assert species > 0, 'The number of Species must be a positive integer.'
        species = [{'genomes': True} for _ in range(species)]
    assert isinstance(species, list) and all(isinstance(d, dict) for d in species), (
        "'species' must be an int or a list of dicts")
    for i, sd in enumerate(species):
        sp = spp_defaults()
        if sd.get('movement', True) is False:
            sp.pop('movement')
        else:
            if sd.get('movement_surface', False):
                sp['movement']['move_surf'] = surf_defaults()
            if sd.get('dispersal_surface', False):
                sp['movement']['disp_surf'] = surf_defaults()
        if sd.get('genomes', False) in (True, 'custom'):
            ga = gen_arch_defaults()
            nt = int(sd.get('n_traits', 0) or 0)
            if nt > 0:
                ga['traits'] = {'trait_%i' % t: trait_defaults() for t in range(nt)}
            sp['gen_arch'] = ga
        change = {}
        ndem = int(sd.get('demographic_change', 0) or 0)
        if ndem > 0:
            change['dem'] = {k: dem_change_event_defaults() for k in range(ndem)}
        if sd.get('parameter_change', False):
            change['life_hist'] = {'<life_hist_param>': {'timesteps': [], 'vals': []}}
        if change:
            sp['change'] = change
        spps['spp_%i' % i] = sp
    model = {'T': 100, 'burn_T': 30, 'num': None, 'its': its_defaults()}
    if data:
        model['data'] = data_defaults()
    if stats:
        model['stats'] = stats_defaults()
    return {
        'landscape': {'main': {'dim': (20, 20), 'res': (1, 1), 'ulc': (0, 0),
                               'prj': None},
                      'layers': lyrs},
        'comm': {'species': spps},
        'model': model,
    }


# short explanatory comments written next to keys of the generated file
_COMMENTS = {
    'dim': 'x,y (a.k.a. j,i) dimensions of the Landscape',
    'res': 'x,y resolution of the Landscape',
    'ulc': 'x,y coords of upper-left corner of the Landscape',
    'prj': 'projection of the Landscape',
    'N': 'starting number of individuals',
    'K_layer': 'carrying-capacity Layer name',
    'K_factor': 'multiplicative factor for carrying-capacity layer',
    'repro_age': 'age(s) at sexual maturity (if tuple, female first)',
    'sex': 'whether to assign sexes',
    'sex_ratio': 'ratio of males to females',
    'R': 'intrinsic growth rate',
    'b': 'intrinsic birth rate (probability that a pair mates)',
    'n_births_distr_lambda': 'expectation of the number of births per pair',
    'n_births_fixed': 'whether n_births is fixed at n_births_distr_lambda',
    'mating_radius': 'radius of mate-search area (None = panmixia)',
    'choose_nearest_mate': 'whether individs choose their nearest neighbour as mate',
    'inverse_dist_mating': 'whether mate choice is weighted by inverse distance',
    'max_age': 'maximum age (None = no maximum)',
    'd_min': 'min probability of death',
    'd_max': 'max probability of death',
    'density_grid_window_width': 'width of window used to estimate local density',
    'move': 'whether or not the species is mobile',
    'direction_distr_mu': 'mode of the von Mises distribution of movement direction',
    'direction_distr_kappa': 'concentration of that distribution',
    'movement_distance_distr': "'lognormal', 'levy' or 'wald'",
    'dispersal_distance_distr': "'lognormal', 'levy' or 'wald'",
    'L': 'total genome length (number of loci)',
    'start_p_fixed': 'starting 1-allele frequency (float), True = 0.5, False/None = random',
    'r_distr_alpha': 'recombination-rate Beta alpha (or fixed rate if beta is None)',
    'n_recomb_sims': 'number of recombination paths simulated and cached',
    'use_tskit': 'tree-sequence recording (not implemented here: genotypes are '
                 'tracked in full on the GPU)',
    'T': 'total Model runtime (in timesteps)',
    'burn_T': 'min burn-in runtime (in timesteps)',
    'num': 'seed number',
}


def _fmt_value(v):
    if isinstance(v, _Code):
        return str(v)
    if isinstance(v, np.ndarray):
        if v.size and np.all(v == 1):
            return 'np.ones(%s)' % repr(tuple(v.shape)).replace(' ', '')
        return 'np.array(%s)' % repr(v.tolist())
    return repr(v)


def _emit(d, out, indent):
    pad = ' ' * indent
    for k, v in d.items():
        if isinstance(v, dict):
            out.append('%s%r: {' % (pad, k))
            _emit(v, out, indent + 4)
            out.append("%s    }, # <END> %r" % (pad, k))
        else:
            if k in _COMMENTS:
                out.append('%s#%s' % (pad, _COMMENTS[k]))
            out.append('%s%-44s%s,' % (pad, repr(k) + ':', _fmt_value(v)))


def params_file_text(filename, layers=1, species=1, data=None, stats=None):
    d = default_params_dict(layers, species, data, stats)
    out = ['# %s' % filename, '',
           '# This is a parameters file in the Geonomics format',
           '# (written by make_parameters_file of geonomics_amd).', '',
           'params = {']
    _emit(d, out, 4)
    out.append('    } # <END> params')
    return '\n'.join(out) + '\n'


def _make_params_file(filepath=None, layers=1, species=1, data=None, stats=None):
    if filepath is None:
        filepath = 'GNX_params_%s.py' % time.strftime('%d-%m-%Y_%H:%M:%S',
                                                      time.localtime())
    d = os.path.split(filepath)[0]
    assert os.path.isdir(d) or d == '', (
        'The filepath to which to write the parameters file does not point '
        'to a valid directory.')
    filepath = os.path.splitext(filepath)[0] + '.py'
    with open(filepath, 'w') as f:
        f.write(params_file_text(os.path.split(filepath)[1], layers, species,
                                 data, stats))
    return filepath


def _read_params_file(filepath):
    ns = {'np': np, 'numpy': np}
    with open(filepath, 'r') as f:
        exec(f.read(), ns)
    if 'params' not in ns:
        raise ValueError("the parameters file must define a dict called 'params'")
    params = ParametersDict(ns['params'])
    if 'name' in params['model'] and params['model']['name'] is not None:
        name = params['model']['name']
    else:
        name = os.path.splitext(os.path.split(filepath)[-1])[0]
    params.model['name'] = name
    return params
