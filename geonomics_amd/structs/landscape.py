"""Landscape containers (reference: geonomics/structs/landscape.py Layer:34,
Landscape:245, _make_landscape:522).

The hot path only reads `lyr.rast`; the rasters are mirrored to the GPU as
float32 [n_layers][H][W] by the Species' device state.  Layer construction is
one-off host work: 'defined' and 'random' layers are built here; 'file' and
'nlmpy' layers need rasterio / nlmpy, which are file-I/O and third-party
generators outside the hot path - pass such rasters in as
'defined' layers instead.
"""
import numpy as np


class Layer:
    """One environmental raster in [0, 1]; rast is indexed [y, x], dim = (x, y)."""

    def __init__(self, rast, lyr_type, name, dim, res=(1, 1), ulc=(0, 0),
                 prj=None, coord_prec=0, units='', scale_min=0, scale_max=1):
        self.idx = None
        self.type = lyr_type
        self.name = str(name)
        assert type(dim) in [tuple, list], 'dim must be expressed as a tuple or a list'
        self.dim = tuple(dim)
        self.res = res
        self.ulc = ulc
        self.prj = prj
        self.coord_prec = coord_prec
        self.units = units
        assert isinstance(rast, np.ndarray), 'rast should be a numpy.ndarray'
        self.rast = rast
        self._scale_min = scale_min
        self._scale_max = scale_max
        self._is_K = []

    def __str__(self):
        return "<Layer '%s' (%s), dim %s>" % (self.name, self.type, str(self.dim))

    __repr__ = __str__


class Landscape(dict):
    """Serial-integer-keyed dict of Layers (reference structs/landscape.py:245)."""

    def __init__(self, lyrs, res=(1, 1), ulc=(0, 0), prj=None, mod=None):
        assert all(lyr.__class__.__name__ == 'Layer' for lyr in lyrs.values()), (
            'All layers supplied in lyrs must be of type landscape.Layer.')
        self.update(lyrs)
        self.n_lyrs = len(self)
        for k, v in self.items():
            v.idx = k
        assert len(set([lyr.dim for lyr in self.values()])) == 1, (
            'Dimensions of all layers must be equal.')
        self.dim = list(self.values())[0].dim
        self._dim_om = max([len(str(d)) for d in self.dim])
        self.res = res
        self._res_ratio = tuple(np.abs([val / max(self.res) for val in self.res]))
        self.ulc = ulc
        self.prj = prj
        self._x_cell_bds, self._y_cell_bds = [
            np.linspace(self.ulc[i], self.ulc[i] + (self.res[i] * self.dim[i]),
                        self.dim[i] + 1) for i in range(2)]
        for lyr in lyrs.values():
            assert lyr.rast.shape == (self.dim[1], self.dim[0]), (
                "Layer '%s' has raster shape %s; expected (y, x) = %s" % (
                    lyr.name, str(lyr.rast.shape), str((self.dim[1], self.dim[0]))))
            assert np.all(0 <= lyr.rast), "Layer '%s' contains values less than 0." % lyr.name
            assert np.all(lyr.rast <= 1), "Layer '%s' contains values greater than 1." % lyr.name
        self._changer = None
        self._changed_lyrs = set()     # layers whose raster changed since the last device sync

    def _set_raster(self, lyr_num, rast):
        """reference structs/landscape.py:353-354; the Model mirrors the layer to the
        species' devices right after the change (sim/model.py _make_land_change)"""
        self[lyr_num].rast = rast
        self._changed_lyrs.add(lyr_num)

    def _make_change(self, t, verbose=False):
        """reference structs/landscape.py:363-365"""
        self._changer._make_change(t=t, additional_args={'land': self}, verbose=verbose)

    def _get_lyr_num(self, lyr_id):
        if isinstance(lyr_id, int) or lyr_id is None:
            return lyr_id
        nums = [k for k, lyr in self.items() if lyr.name == lyr_id]
        assert len(nums) == 1, ("Expected to find a single Layer with a name "
                                "matching the name provided (%s). Instead found "
                                "%i.") % (lyr_id, len(nums))
        return nums[0]

    def _stack(self):
        """float32 [n_layers][H][W] for the device."""
        return np.stack([np.asarray(self[k].rast, dtype=np.float32)
                         for k in sorted(self.keys())])

    def __str__(self):
        return '%s\n%i Layer%s: %s' % (str(type(self)), self.n_lyrs,
                                       's' * (len(self) > 1),
                                       ', '.join("'%s'" % l.name for l in self.values()))

    __repr__ = __str__


def _scattered_to_raster(pts, vals, dim, method, n_classes, rng):
    """Raster of a layer given by scattered points: the values are interpolated
    (scipy griddata) at the nodes of a square max(dim) x max(dim) lattice running from 1 to
    max(dim) along both axes, first axis = first coordinate of `pts`, and the landscape's
    own extent is cut out of it afterwards.  'nearest' makes patches of n_classes integer
    habitat classes (the values are proportions, scaled to 0 .. n_classes - 1 and rounded);
    'cubic' surfaces are shifted and scaled into (0, 1), with a small random margin that
    keeps them off exactly 0 and 1.  Both 'random' and 'defined' layers are built this way
    (reference structs/landscape.py:417-519)."""
    from scipy.interpolate import griddata
    side = max(dim)
    axis = np.linspace(1, side, side)
    nodes = tuple(np.meshgrid(axis, axis, indexing='ij'))
    vals = np.asarray(vals, dtype=float)
    patches = method == 'nearest'
    surface = griddata(np.asarray(pts, dtype=float), vals * (n_classes - 1) if patches else vals,
                       nodes, method=method)
    if patches:
        surface = np.rint(surface).astype(float)
    elif method == 'cubic':
        lift = abs(surface.min()) + 0.01 * rng.rand()
        surface = (surface + lift) / ((surface + lift).max() + 0.01 * rng.rand())
    W, H = dim
    return surface if W == H else surface[:H, :W]


def _make_random_lyr(dim, n_pts, interp_method='cubic', num_hab_types=2, dist='beta',
                     alpha=0.05, beta=0.05, rng=None):
    """'random' layer: n_pts seed values (beta(alpha, beta) or uniform) at points drawn
    well beyond the landscape, N(max_dim / 2, 2 max_dim) per coordinate, so that the
    interpolation spans the whole extent; cells the hull still misses are 0"""
    rng = np.random if rng is None else rng
    seeds = rng.rand(n_pts) if dist == 'unif' else rng.beta(alpha, beta, n_pts)
    where = rng.normal(max(dim) / 2, max(dim) * 2, [n_pts, 2])
    rast = _scattered_to_raster(where, seeds, dim, interp_method, num_hab_types, rng)
    return np.clip(np.nan_to_num(rast, nan=0.0), 0, 1)


def _make_defined_lyr(dim, rast=None, pts=None, vals=None, interp_method='cubic',
                      num_hab_types=2, rng=None):
    """'defined' layer: the raster itself, or scattered (pts, vals) to interpolate"""
    if rast is not None:
        return np.asarray(rast, dtype=np.float64)
    return _scattered_to_raster(pts, vals, dim, interp_method, num_hab_types,
                                np.random if rng is None else rng)


def _make_landscape(mod, params, num_hab_types=2, verbose=False):
    """reference structs/landscape.py:522-700 (layer types 'random' and
    'defined'; change events: ops/change.py)."""
    if verbose:
        print('\tMAKING LANDSCAPE...\n')
    main = params.landscape.main
    dim = tuple(main.dim)
    res = main.res if main.get('res', None) is not None else (1, 1)
    ulc = main.ulc if main.get('ulc', None) is not None else (0, 0)
    prj = main.get('prj', None)
    lyrs = {}
    for n, (lyr_name, lyr_params) in enumerate(params.landscape.layers.items()):
        init = dict(lyr_params['init'])
        keys = [*init]
        if len(keys) > 1:
            raise ValueError(("The %ith layer ('%s') appears to have parameters for "
                              "more than one layer type.") % (n, str(lyr_name)))
        kind = keys[0]
        if kind == 'random':
            rast = _make_random_lyr(dim, num_hab_types=num_hab_types, **dict(init[kind]))
        elif kind == 'defined':
            rast = _make_defined_lyr(dim, num_hab_types=num_hab_types, **dict(init[kind]))
        elif kind in ('file', 'nlmpy'):
            raise NotImplementedError(
                "Layer type '%s' needs %s, which is outside the GPU hot path; read "
                "the raster yourself and pass it as a 'defined' layer ('rast')." % (
                    kind, 'rasterio/GDAL' if kind == 'file' else 'nlmpy'))
        else:
            raise ValueError("invalid layer type '%s' (valid: 'random', 'defined', "
                             "'file', 'nlmpy')" % kind)
        lyrs[n] = Layer(np.asarray(rast, dtype=np.float64), lyr_type=kind, name=lyr_name,
                        dim=dim, res=res, ulc=ulc, prj=prj)
    land = Landscape(lyrs, res=res, ulc=ulc, prj=prj, mod=mod)
    # change events (reference structs/landscape.py:655-672)
    change_params = {land._get_lyr_num(k): v.change for k, v in params.landscape.layers.items()
                     if 'change' in v.keys()}
    if len(change_params) > 0:
        from ..ops.change import _LandscapeChanger
        land._changer = _LandscapeChanger(land, change_params, mod=mod)
    return land
