"""geonomics_amd - MI355X-native implementation of Geonomics' per-generation
simulation loop behind the Geonomics API (make_model / Model.walk / Model.run
and the parameters-file format).  Hand-written HIP (gfx950) through a C-ABI
(libgnxhip.so, include/gnx_hip.h); no CPU fallback."""
__version__ = '0.1.0'
