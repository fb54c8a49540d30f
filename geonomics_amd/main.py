"""Public functions of the Geonomics API (reference: geonomics/main.py
make_parameters_file:76, read_parameters_file:308, make_params_dict:403,
make_model:442, run_default_model:608)."""
import os
import re
import sys
import traceback

from .sim.model import Model
from .sim.params import (ParametersDict, _make_params_file, _read_params_file)


def make_parameters_file(filepath=None, layers=1, species=1, data=None, stats=None):
    """Write a template parameters file (same sections and keys as the
    reference's template).  `layers` / `species` may be ints or lists of dicts
    ({'type': 'random'|'defined'|'file'|'nlmpy', 'change': ...} /
    {'movement', 'movement_surface', 'dispersal_surface', 'genomes', 'n_traits',
    'demographic_change', 'parameter_change'})."""
    return _make_params_file(filepath=filepath, layers=layers, species=species,
                             data=data, stats=stats)


def read_parameters_file(filepath):
    """Read a parameters file into a ParametersDict; Layer and Species names
    must be unique (reference main.py:336-396)."""
    with open(filepath, 'r') as f:
        txt = f.read()
    for kind in ('lyr', 'spp'):
        names = re.findall(r"^\s*'(%s_\w+)'\s*:\s*\{" % kind, txt, flags=re.M)
        dup = sorted({n for n in names if names.count(n) > 1})
        if dup:
            raise ValueError('The following %s names appear more than once in the '
                             'parameters file: %s' % ('Layer' if kind == 'lyr'
                                                      else 'Species', dup))
    return _read_params_file(filepath)


def make_params_dict(params, model_name=None):
    """dict -> ParametersDict (reference main.py:403-439)"""
    params_dict = ParametersDict(params)
    if model_name is not None:
        params_dict['model']['name'] = model_name
    elif 'name' in params['model'] and params['model']['name'] is not None:
        pass
    else:
        params_dict.model['name'] = 'unnamed_model'
    return params_dict


def make_model(parameters=None, name=None, verbose=False):
    """Create a Model from a parameters file path, a dict or a ParametersDict;
    with no argument, from the single 'GNX_params_*.py' file in the working
    directory (reference main.py:442-605)."""
    if parameters is None:
        cands = [f for f in os.listdir('.') if re.match(r'^GNX_params_.*\.py$', f)]
        if len(cands) != 1:
            raise ValueError("The 'parameters' argument was not provided and the current "
                             "working directory does not contain exactly one "
                             "'GNX_params_<...>.py' file (found %i)." % len(cands))
        parameters = cands[0]
        print('NOTE: Using the following file, in the current working directory, to '
              'create the Model object:\n\t%s' % parameters)
    if isinstance(parameters, str):
        if not os.path.isfile(parameters):
            raise ValueError("If the 'parameters' argument is a string it must point to a "
                             "valid Geonomics parameters file.")
        try:
            parameters = read_parameters_file(parameters)
        except Exception as e:
            traceback.print_exc(file=sys.stdout)
            raise ValueError('Failed to read the parameters file at the filepath that was '
                             'provided. The following error was raised: \n\t%s\n\n' % e)
    elif isinstance(parameters, dict) and not isinstance(parameters, ParametersDict):
        parameters = make_params_dict(parameters, name)
    elif not isinstance(parameters, ParametersDict):
        raise ValueError("'parameters' must be a filepath, a dict or a ParametersDict")
    if 'name' not in parameters['model'] or parameters['model']['name'] is None:
        parameters.model['name'] = 'unnamed_model'
    try:
        if name is None:
            name = parameters['model']['name']
        return Model(name, parameters, verbose=verbose)
    except Exception as e:
        traceback.print_exc(file=sys.stdout)
        raise ValueError('Failed to create a Model object from the ParametersDict object '
                         'being used. The following error was raised: \n\t%s\n\n' % e)


def run_default_model(selection=False, delete_params_file=True, animate=False):
    """Create, burn in and walk the default model for 50 main steps
    (reference main.py:608-676).  The neutral default is the template model
    (20x20, 1 random layer, N=250, L=100); the selection variant is a 2-layer
    model with a 4-locus trait selected on the second layer.  Plotting is not
    part of the hot path."""
    import numpy as np
    from .sim import params as P
    if not selection:
        filename = 'GNX_default_model_params_NEUTRAL.py'
        make_parameters_file(filename)
        mod = make_model(parameters=filename)
        if delete_params_file:
            os.remove(filename)
    else:
        d = P.default_params_dict(layers=[{'type': 'defined'}, {'type': 'defined'}],
                                  species=[{'genomes': True, 'n_traits': 1}])
        dim = (35, 35)
        d['landscape']['main']['dim'] = dim
        rng = np.random.RandomState(1)
        from scipy.ndimage import gaussian_filter
        for k, key in enumerate(('lyr_0', 'lyr_1')):
            f = gaussian_filter(rng.rand(dim[1], dim[0]), 4)
            d['landscape']['layers'][key]['init']['defined']['rast'] = \
                (f - f.min()) / (f.max() - f.min())
        spp = d['comm']['species']['spp_0']
        spp['init']['N'] = 500
        spp['gen_arch']['L'] = 10
        spp['gen_arch']['traits']['trait_0'].update({'layer': 'lyr_1', 'n_loci': 4})
        mod = make_model(d, name='GNX_default_model_params_SELECTION')
    mod.walk(T=10000, mode='burn', verbose=True)
    mod.walk(T=50, mode='main', verbose=True)
    return mod
