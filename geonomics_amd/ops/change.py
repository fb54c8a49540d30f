"""Landscape- and species-change events (reference geonomics/ops/change.py).

A changer holds a time-ordered list of (timestep, function) pairs; the Model's
main queue calls `_make_change(t)` after the hot path of each step (reference
sim/model.py:644-656) and every function due at `t` runs.  What a change does
on the device:

  * layer change      - the new raster replaces `land[lyr].rast` and is uploaded
                        to every species' device mirror (one H x W f32 copy);
                        the Species' K follows through `_set_K`.  Movement and
                        dispersal surfaces need no series of their own: the
                        kernels sample the conductance neighbourhood from the
                        raster that is resident (reference builds one LUT per
                        step of the event, change.py:576-609, at the same
                        timesteps).
  * demographic change - `spp.K` is scaled (change.py:633-651); the K setter
                        uploads the raster as the device's explicit K.
  * life-history change - `setattr(spp, parameter, val)` (change.py:744-752);
                        the Species re-uploads its parameter block.

Kept from the reference: event parameters and their meaning; timesteps
`int64(round(linspace(start_t, end_t, n_steps)))` and per-cell
`linspace(start, end, n_steps + 1)[1:]` rasters (:302-353); K_mode 'current'
for monotonic events and 'base' for the others, base K captured when
`spp.t == ` the event's first timestep (:633-651); the last stochastic size
forced to 1 (:683-684); the sine-table construction of cyclical sizes
(:690-730); a change fires only when `t ==` its timestep, so a timestep that
is already past blocks the ones behind it (:56-86).
Not kept: `change_rast` given as a GIS file or directory of GIS files needs
rasterio (absent here) - arrays, .npy and .txt rasters are read; the plotting
helper `_plot_dem_changes`.
"""
import copy
import os

import numpy as np


class _Changer:
    def __init__(self, params):
        self.type = None
        self.change_params = copy.deepcopy(params)
        self._list = []          # [(t, fn)] in chronological order
        self._pos = 0

    # the reference keeps an iterator + `next_change`; same observable state
    @property
    def next_change(self):
        return self._list[self._pos] if self._pos < len(self._list) else None

    @property
    def changes(self):
        return iter(self._list[self._pos + 1:])

    def _set_changes_list(self, changes):
        self._list = list(changes)
        self._pos = 0

    def _make_change(self, t, additional_args, for_plotting=False, verbose=False):
        while (self.next_change is not None and t == self.next_change[0]
               and not for_plotting):
            if verbose:
                print('\t**** Running the next change\t%s\n\n' % str(self.next_change))
            self.next_change[1](changer=self, **additional_args)
            self._pos += 1

    def _add_change(self, change):
        """insert (timestep, fn) after the pending changes with timestep <= its own"""
        pend = self._list[self._pos:]
        k = 0
        while k < len(pend) and pend[k][0] <= change[0]:
            k += 1
        self._list = self._list[:self._pos] + pend[:k] + [change] + pend[k:]


# ---------------------------------------------------------------------------
# landscape
# ---------------------------------------------------------------------------
def _read_change_rast(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npy':
        return np.load(path)
    if ext in ('.txt', '.csv'):
        return np.loadtxt(path, delimiter=',' if ext == '.csv' else None)
    raise NotImplementedError(
        "change_rast '%s': GIS rasters need rasterio, which this build does not use; "
        "pass a numpy array, or a .npy / .txt raster" % path)


def _make_lyr_series(lyr, change_rast, start_t, end_t, n_steps, coord_prec=0):
    """[(timestep, raster)] of one change event (reference change.py:302-495)"""
    start_rast = lyr.rast
    timesteps = np.int64(np.round(np.linspace(start_t, end_t, n_steps)))
    if isinstance(change_rast, str) and os.path.isdir(change_rast):
        files = os.listdir(change_rast)
        assert len(files) == n_steps, (
            "The number of files in the directory provided for the 'change_rast' "
            "parameter is not equal to the number provided for the 'n_steps' parameter.")
        by_t = {}
        for f in files:
            head = os.path.splitext(f.split('_')[0])[0]
            assert head.isnumeric(), (
                "In the directory provided for the 'change_rast' parameter, the file %s "
                "does not start with an integer followed by an underscore." % f)
            by_t[int(head)] = f
        assert len(by_t) == len(files), 'Not all timesteps in the directory are unique.'
        timesteps = sorted(by_t)
        assert timesteps[0] == start_t and timesteps[-1] == end_t, (
            'The timesteps of the files must start at start_t and end at end_t.')
        rast_series = [np.asarray(_read_change_rast(os.path.join(change_rast, by_t[t])),
                                  dtype=np.float64) for t in timesteps]
    else:
        if isinstance(change_rast, str):
            if not os.path.isfile(change_rast):
                raise ValueError(
                    "The value provided for the 'change_rast' parameter must be either a "
                    "numpy.ndarray or a path to a valid file (the endpoint raster of the "
                    "change event), or a path to a directory of valid files (one for "
                    "each timestep in the change event).")
            change_rast = _read_change_rast(change_rast)
        if not isinstance(change_rast, np.ndarray):
            raise ValueError("The value provided for the 'change_rast' parameter must be "
                             "a numpy.ndarray, a file or a directory.")
        assert change_rast.shape == start_rast.shape, (
            'Dimensionality of the change raster does not match that of the Layer.')
        # per cell linspace(start, end, n_steps + 1)[1:]; linspace broadcasts the same
        # arithmetic over the raster
        stack = np.linspace(np.asarray(start_rast, dtype=np.float64),
                            np.asarray(change_rast, dtype=np.float64), n_steps + 1)[1:]
        rast_series = [stack[k] for k in range(n_steps)]
    assert len(rast_series) == n_steps == len(timesteps)
    for r in rast_series:
        assert r.shape == start_rast.shape, (
            'Dimensionality of the change rasters does not match that of the Layer.')
    return list(zip([int(t) for t in timesteps], rast_series))


def _make_conglom_lyr_series(land, lyr_num, change_params_one_lyr):
    """all events of one Layer, concatenated (reference change.py:498-560)"""
    spans = [t for v in change_params_one_lyr.values() for t in range(v['start_t'], v['end_t'])]
    assert len(set(spans)) == len(spans), (
        'Some of the change events for Layer number %i overlap in time.' % lyr_num)
    out = []
    for v in change_params_one_lyr.values():
        out.extend(_make_lyr_series(land[lyr_num], coord_prec=land[lyr_num].coord_prec,
                                    **dict(v)))
    return out


def _get_lyr_change_fn(lyr_num, new_lyr_rast):
    def fn(changer, land, lyr_num=lyr_num, new_lyr_rast=new_lyr_rast):
        land._set_raster(lyr_num, new_lyr_rast)
    return fn


class _LandscapeChanger(_Changer):
    def __init__(self, land, land_change_params, mod=None):
        super().__init__(land_change_params)
        self.type = 'land'
        self.change_info = {}
        self._set_changes(land)

    def _set_changes(self, land):
        lyr_changes = []
        for lyr_num in self.change_params.keys():
            series = _make_conglom_lyr_series(land, lyr_num, self.change_params[lyr_num])
            self.change_info[lyr_num] = {**self.change_params[lyr_num]}
            lyr_changes.extend((t, lyr_num, rast) for t, rast in series)
        lyr_changes.sort(key=lambda c: c[0])
        self._set_changes_list((t, _get_lyr_change_fn(n, r)) for t, n, r in lyr_changes)


# ---------------------------------------------------------------------------
# species: demography
# ---------------------------------------------------------------------------
def _make_dem_change_fns(sizes, timesteps, K_mode='base'):
    """reference change.py:633-651"""
    fns = []
    if K_mode == 'current':
        for size in sizes:
            def fn(changer, spp, size=size):
                spp.K = spp.K * size
            fns.append(fn)
    elif K_mode == 'base':
        t0 = timesteps[0]
        for size in sizes:
            def fn(changer, spp, size=size, t0=t0):
                if spp.t == t0:
                    changer._set_base_K(spp)
                spp.K = changer.base_K * size
            fns.append(fn)
    return list(zip([int(t) for t in timesteps], fns))


def _get_monotonic_dem_change_fns(rate, start_t, end_t):
    timesteps = range(start_t, end_t + 1)
    return _make_dem_change_fns([rate] * len(timesteps), timesteps, K_mode='current')


def _get_stochastic_dem_change_fns(start_K, size_range, start_t, end_t, interval,
                                   distr='uniform', rng=None):
    rng = np.random if rng is None else rng
    if interval is None:
        interval = 1
    timesteps = range(start_t, end_t + 1, interval)
    if distr == 'uniform':
        sizes = rng.uniform(*size_range, len(timesteps))
    elif distr == 'normal':
        mean = np.mean(size_range)
        sd = (size_range[1] - size_range[0]) / 6
        sizes = rng.normal(loc=mean, scale=sd, size=len(timesteps))
    else:
        raise ValueError("Argument 'distr' must be a value among ['uniform', 'normal']")
    sizes[-1] = 1                      # return to the starting size
    return _make_dem_change_fns(sizes, timesteps, K_mode='base')


def _cyclical_sizes(start_t, end_t, n_cycles, min_size, max_size, increase_first=True):
    """sizes and timesteps of the sine cycles (reference change.py:705-730)"""
    assert n_cycles <= (end_t - start_t) / 2, (
        'The number of cycles requested must be no more than half the number of time '
        'steps over which the cycling should take place.')
    base = np.sin(np.linspace(0, 2 * np.pi, 1000))
    if not increase_first:
        base = base[::-1]
    # positive lobe scaled towards max_size ...
    scaled = np.array([1 + n * (max_size - 1) if n >= 0 else n for n in base])
    # ... then whatever is (still) negative towards min_size
    scaled = np.array([1 + n * (1 - min_size) if n < 0 else n for n in scaled])
    cycle_timesteps = np.int32(np.linspace(start_t, end_t, n_cycles + 1))
    lengths = np.diff(cycle_timesteps)
    sizes = np.hstack([scaled[np.int32(np.linspace(1, len(scaled) - 1, l))]
                       for l in lengths] + [1])
    timesteps = range(int(cycle_timesteps[0]), int(cycle_timesteps[-1]) + 1)
    return sizes, timesteps


def _get_cyclical_dem_change_fns(start_t, end_t, n_cycles, size_range=None, min_size=None,
                                 max_size=None, increase_first=True):
    if size_range is not None and min_size is None and max_size is None:
        min_size, max_size = size_range
    elif size_range is None and min_size is not None and max_size is not None:
        pass
    else:
        raise ValueError('Must either provide size_range (as a tuple of minimum and maximum '
                         'sizes), or provide min_size and max_size separately, but not both.')
    sizes, timesteps = _cyclical_sizes(start_t, end_t, n_cycles, min_size, max_size,
                                       increase_first)
    return _make_dem_change_fns(sizes, timesteps, K_mode='base')


def _get_custom_dem_change_fns(timesteps, sizes):
    assert len(timesteps) == len(sizes), (
        'For custom demographic changes, timesteps and sizes must be iterables of equal '
        'length.')
    return _make_dem_change_fns(sizes, timesteps, K_mode='base')


def _get_dem_change_fns(spp, kind, start_t=None, end_t=None, rate=None, interval=None,
                        n_cycles=None, size_range=None, distr='uniform', min_size=None,
                        max_size=None, timesteps=None, sizes=None, increase_first=True,
                        rng=None):
    """reference change.py:612-630"""
    if kind == 'monotonic':
        return _get_monotonic_dem_change_fns(rate=rate, start_t=start_t, end_t=end_t)
    if kind == 'stochastic':
        return _get_stochastic_dem_change_fns(start_K=spp.K, start_t=start_t, end_t=end_t,
                                              interval=interval, size_range=size_range,
                                              distr=distr, rng=rng)
    if kind == 'cyclical':
        return _get_cyclical_dem_change_fns(start_t=start_t, end_t=end_t, n_cycles=n_cycles,
                                            size_range=size_range, min_size=min_size,
                                            max_size=max_size, increase_first=increase_first)
    if kind == 'custom':
        return _get_custom_dem_change_fns(timesteps=timesteps, sizes=sizes)
    raise ValueError("invalid kind of demographic change '%s'" % str(kind))


# ---------------------------------------------------------------------------
# species: life history
# ---------------------------------------------------------------------------
def _get_parameter_change_fns(parameter, timesteps, vals):
    """reference change.py:744-760"""
    assert len(timesteps) == len(vals), (
        "For custom changes of the '%s' parameter, timesteps and vals must be iterables "
        "of equal length." % parameter)
    fns = []
    for val in vals:
        def fn(changer, spp, parameter=parameter, val=val):
            setattr(spp, parameter, val)
        fns.append(fn)
    return list(zip([int(t) for t in timesteps], fns))


class _SpeciesChanger(_Changer):
    def __init__(self, spp, spp_change_params, land=None, rng=None):
        super().__init__(spp_change_params)
        self.type = 'spp'
        self.base_K = None
        self._rng = rng
        self._set_changes(spp, land)

    def _set_base_K(self, spp):
        self.base_K = spp.K

    def _set_changes(self, spp, land):
        cp = self.change_params
        dem = cp.get('dem', None) if cp is not None else None
        life = cp.get('life_hist', None) if cp is not None else None
        fns = []
        if dem is not None:
            for event_params in dem.values():
                if any(v is not None for v in event_params.values()):
                    fns.extend(_get_dem_change_fns(spp, rng=self._rng, **dict(event_params)))
        if life is not None:
            for parameter, pp in life.items():
                if any(v is not None for v in pp.values()):
                    fns.extend(_get_parameter_change_fns(parameter, **dict(pp)))
        # (movement / dispersal surfaces follow the layer on the device: no series)
        fns.sort(key=lambda c: c[0])
        self._set_changes_list(fns)

    def _plot_dem_changes(self, spp):
        raise NotImplementedError('plotting is outside the device hot path')
