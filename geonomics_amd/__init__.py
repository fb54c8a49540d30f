"""geonomics_amd - MI355X-native implementation of Geonomics' per-generation
simulation loop behind the Geonomics API (gnx.make_model / Model.walk /
Model.run and the parameters-file format).

    import geonomics_amd as gnx
    mod = gnx.make_model('GNX_params_xyz.py')
    mod.walk(10000, 'burn'); mod.walk(100, 'main')

Hand-written HIP (gfx950) through a C-ABI (libgnxhip.so, include/gnx_hip.h);
the product has no CPU fallback."""
__version__ = '0.1.0'

from .sim.params import ParametersDict                      # noqa: F401
from .sim.model import Model                                # noqa: F401
from .main import (make_parameters_file, read_parameters_file,   # noqa: F401
                   make_params_dict, make_model, run_default_model)
