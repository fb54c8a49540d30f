"""CPU-side checks of the C-ABI: the library loads and exports every symbol
include/gnx_hip.h declares; without a HIP device creation fails loudly."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from geonomics_amd import _native, build
    if not os.path.exists(_native.LIB_PATH):
        build.build(verbose=False)
    return _native.load()


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'gnx_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(gnx_[a-z0-9_]+)\s*\(', txt)))


def test_every_declared_symbol_is_exported(lib):
    from geonomics_amd import _native
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), 'libgnxhip.so does not export %s' % s
    assert sorted(_native.EXPORTS) == syms


def test_struct_layouts_match_header(lib):
    from geonomics_amd import _native
    assert ctypes.sizeof(_native.Config) == 56
    sp = _native.default_species_params()
    assert ctypes.sizeof(sp) % 8 == 0
    assert lib.gnx_words_per_hom(1) == 16
    assert lib.gnx_words_per_hom(1024) == 16
    assert lib.gnx_words_per_hom(1025) == 32
    assert lib.gnx_words_per_hom(100000) == 1568


def test_no_cpu_fallback(lib):
    """Without a HIP device the product refuses to run (no silent fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('a HIP device is present')
    from geonomics_amd import _native
    with pytest.raises(_native.GnxError, match='no HIP device|hip'):
        _native.Device(8, 8, 1)
