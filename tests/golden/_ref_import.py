"""Import the Geonomics reference (read-only, /root/reference) in THIS container.

Used ONLY by tests/golden/make_golden.py to generate golden vectors. The GPU
box has no /root/reference, so nothing under tests/ imports this at test time.

The reference lists third-party packages that this image lacks (bitarray,
shapely, tskit, msprime, statsmodels, geopandas, rasterio, PyVCF). None of them
is on the `use_tskit=False` hot path except:
  * bitarray.bitarray  - container of 0/1 used for the recombination
    "subsetters" (structs/genome.py:56,220-225; ops/mating.py:165-167)
  * shapely Polygon    - axis-aligned rectangle intersection areas for the
    density-grid windows (utils/spatial.py:299-314)
Minimal behavioural stand-ins for those two are registered in sys.modules
before the import; the rest are empty modules. These stand-ins live in the
golden-vector generator only; nothing of the reference is copied.
"""
import sys
import types

REF_ROOT = '/root/reference'


class _BitArray(list):
    """list-of-ints stand-in for bitarray.bitarray (str ctor, +, slicing)."""

    def __init__(self, init=''):
        if isinstance(init, str):
            super().__init__(int(c) for c in init)
        else:
            super().__init__(int(c) for c in init)

    def __add__(self, other):
        return _BitArray(list(self) + list(_BitArray(other)))

    def __getitem__(self, k):
        out = super().__getitem__(k)
        return _BitArray(out) if isinstance(k, slice) else out


class _Rect:
    """Axis-aligned rectangle stand-in for shapely.geometry.Polygon."""

    def __init__(self, coords=None, bounds=None):
        if bounds is not None:
            self.x0, self.y0, self.x1, self.y1 = bounds
        else:
            xs = [c[0] for c in coords]
            ys = [c[1] for c in coords]
            self.x0, self.x1 = min(xs), max(xs)
            self.y0, self.y1 = min(ys), max(ys)

    def intersection(self, other):
        x0, x1 = max(self.x0, other.x0), min(self.x1, other.x1)
        y0, y1 = max(self.y0, other.y0), min(self.y1, other.y1)
        if x1 <= x0 or y1 <= y0:
            return _Rect(bounds=(0, 0, 0, 0))
        return _Rect(bounds=(x0, y0, x1, y1))

    @property
    def area(self):
        return float((self.x1 - self.x0) * (self.y1 - self.y0))


def _adfuller_stub(x, *a, **k):
    # p-value 0 => ADF never blocks burn-in in the generator (the paired
    # t-tests still gate); goldens do not depend on when burn-in ends.
    return (0.0, 0.0)


def import_reference():
    """Return the imported reference package `geonomics` (v1.4.9)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod('bitarray', bitarray=_BitArray)
    geom = mod('shapely.geometry', Polygon=_Rect, Point=object)
    mod('shapely', geometry=geom)
    for name in ('tskit', 'msprime', 'geopandas', 'rasterio', 'vcf',
                 'statsmodels.api'):
        mod(name)
    st = mod('statsmodels.tsa.stattools', adfuller=_adfuller_stub)
    tsa = mod('statsmodels.tsa', stattools=st)
    mod('statsmodels', tsa=tsa)
    import matplotlib
    matplotlib.use('Agg')
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import geonomics
    return geonomics
