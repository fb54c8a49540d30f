"""Species: the reference's per-species API (geonomics/structs/species.py) over a
GPU-resident struct-of-arrays population.

The reference Species is an OrderedDict {id -> Individual}; every hot method
iterates Python objects.  Here the individuals live on the MI355X behind a
libgnxhip.so handle (x, y, age, sex, id, e, z, fit, bit-packed genomes) and the
methods the Model's function queue calls - _set_age_stage, _do_movement,
_do_pop_dynamics, _set_Nt, _set_genomes_and_tables (structs/species.py:567,
582, 822, 554, 956) - are thin calls into the C-ABI.  Read accessors download
on demand and return arrays ordered by ascending individual id, like the
reference's (dict order == id order there).
"""
import copy
import os
import warnings

import numpy as np

from .. import _native as nat
from . import genome as _genome
from ..sim import burnin as _burnin


class Individual:
    """Snapshot of one individual (reference structs/individual.py:100).  Assigning
    `x` or `y` is remembered by the Species and written to the device by
    `Species._set_coords_and_cells()` (the reference's scripts move individuals that way:
    tests/validation/wf/wf_test.py:69-76); everything else is read-only."""

    def __init__(self, idx, x, y, age, sex, e, z, fit, g, spp=None):
        d = self.__dict__
        d['_spp'] = spp
        d['idx'] = idx
        d['x'] = x
        d['y'] = y
        d['age'] = age
        d['sex'] = sex
        d['e'] = e
        d['z'] = z
        d['fit'] = fit
        d['g'] = g

    def __setattr__(self, name, val):
        self.__dict__[name] = val
        if name in ('x', 'y') and self.__dict__.get('_spp') is not None:
            self._spp._pending_xy.setdefault(int(self.idx), [None, None])[
                0 if name == 'x' else 1] = float(val)

    def __repr__(self):
        return '<Individual %i at (%.3f, %.3f), age %i>' % (self.idx, self.x, self.y,
                                                           self.age)


class _ParamsVals:
    def __init__(self, spp_name):
        self.spp_name = spp_name


def _sum_K(land, K_layer_idx, K_factor):
    return float(np.sum(land[K_layer_idx].rast) * K_factor)


class Species:
    def __init__(self, name, idx, land, spp_params, genomic_architecture=None,
                 seed=0, device=0, rng=None):
        self.idx = idx
        self.name = str(name)
        self._rng = np.random if rng is None else rng
        self._land_dim = tuple(land.dim)
        self._land_res = land.res
        self._land_res_ratio = land._res_ratio
        self._land_ulc = land.ulc
        self._land_prj = land.prj
        self._it = None
        self.t = -1
        self.burned = False
        self.extinct = False
        self.Nt = []
        self.n_births = []
        self.n_deaths = []
        self._K = None
        self.K_layer = None
        self.K_factor = None
        self._move = False
        self._move_surf = None
        self._disp_surf = None
        self._changer = None
        self.sex_ratio = 0.5
        self._spp_params = spp_params
        # hoist mating / mortality / movement parameters to attributes
        # (reference structs/species.py:405-425)
        self._pv = _ParamsVals(self.name)
        for section in ['mating', 'mortality', 'movement']:
            if section in [*spp_params]:
                for att, val in spp_params[section].items():
                    if not isinstance(val, dict):
                        if att == 'sex_ratio':
                            val = val / (val + 1)
                        setattr(self._pv, att, val)
                if section == 'movement' and spp_params[section].move:
                    self._move = True
        if self.sex and type(self.repro_age) in [float, int]:
            self._pv.repro_age = (self.repro_age, self.repro_age)
        self.gen_arch = genomic_architecture
        self.selection = (self.gen_arch is not None and
                          (self.gen_arch.mu_delet > 0 or self.gen_arch.traits is not None))
        self.mutate = (self.gen_arch is not None and self.gen_arch._mu_tot is not None
                       and self.gen_arch._mu_tot > 0)
        self.mut_log = spp_params.gen_arch.get('mut_log', None) if 'gen_arch' in [
            *spp_params] else None
        # spatial pedigree (reference use_tskit=True): genotypes are tracked in full on
        # the device either way; the tree-sequence tables are an observer kept on the
        # host (structs/pedigree.py) for models small enough to hold them
        self._pending_xy = {}       # id -> [x, y] assigned through Individual objects
        self._tt = None
        self._record_pedigree = bool(self.gen_arch is not None and
                                     getattr(self.gen_arch, 'use_tskit', False))
        self._seed = int(seed)
        self._device_ordinal = device
        self._dev = None
        self._order = None          # cached argsort of ids
        self._burnin_spat_stats = {'mean': [], 'std': []}
        self.start_N = None
        self.max_ind_idx = None

    # -- carrying capacity ------------------------------------------------------------
    @property
    def K(self):
        return self._K

    @K.setter
    def K(self, val):
        """Assigning Species.K (demographic change events do, ops/change.py:633-651)
        makes the raster the device's explicit K."""
        self._K = val
        if self.__dict__.get('_dev', None) is not None and val is not None:
            self._dev.set_K_raster(np.asarray(val, dtype=np.float64))
            self._K_explicit = True

    def __setattr__(self, attr, val):
        """life-history parameters live in the hoisted params block; changing one
        (ops/change.py:744-752 does setattr(spp, parameter, val)) refreshes the
        device's parameter block"""
        pv = self.__dict__.get('_pv', None)
        if pv is not None and attr in pv.__dict__ and attr != 'spp_name':
            setattr(pv, attr, val)
            if self.__dict__.get('_dev', None) is not None and \
                    self.__dict__.get('_land_ref', None) is not None:
                self._dev.set_species_params(self._species_params_struct(self._land_ref))
            return
        object.__setattr__(self, attr, val)

    def _make_change(self, verbose=False):
        """reference structs/species.py:836-838"""
        self._changer._make_change(t=self.t, additional_args={'spp': self}, verbose=verbose)

    # -- attribute fall-through to the hoisted params (species.py:529-534) ----
    def __getattr__(self, attr):
        if attr.startswith('__') or attr in ('_pv',):
            raise AttributeError(attr)
        try:
            return self.__dict__['_pv'].__getattribute__(attr)
        except Exception:
            raise AttributeError('The Species has no attribute %s' % attr)

    # -- device construction ---------------------------------------------------
    def _species_params_struct(self, land):
        pv = self._pv
        sp = nat.default_species_params()
        sp.b = float(pv.b)
        sp.R = float(pv.R)
        sp.n_births_lambda = float(pv.n_births_distr_lambda)
        sp.n_births_fixed = int(bool(pv.n_births_fixed))
        sp.sexed = int(bool(pv.sex))
        sp.p_male = float(getattr(pv, 'sex_ratio', 0.5))
        sp.mating_radius = -1.0 if pv.mating_radius is None else float(pv.mating_radius)
        sp.mate_mode = (nat.MATE_NEAREST if getattr(pv, 'choose_nearest_mate', False)
                        else nat.MATE_INVERSE if getattr(pv, 'inverse_dist_mating', False)
                        else nat.MATE_UNIFORM)
        ra = pv.repro_age
        if ra is None:
            ra = 0
        ra = tuple(ra) if np.iterable(ra) else (ra, ra)
        sp.repro_age[0], sp.repro_age[1] = int(ra[0]), int(ra[1])
        sp.max_age = -1 if pv.max_age is None else int(pv.max_age)
        sp.d_min = float(pv.d_min)
        sp.d_max = float(pv.d_max)
        ww = getattr(pv, 'density_grid_window_width', None)
        sp.window_width = -1.0 if ww is None else float(ww)
        sp.move = int(self._move)
        if 'movement' in [*self._spp_params]:
            mv = self._spp_params.movement
            sp.dir_mu = float(pv.direction_distr_mu)
            sp.dir_kappa = float(pv.direction_distr_kappa)
            sp.move_distr = nat.DIST[pv.movement_distance_distr]
            sp.move_p1 = float(pv.movement_distance_distr_param1)
            sp.move_p2 = float(pv.movement_distance_distr_param2)
            sp.disp_distr = nat.DIST[pv.dispersal_distance_distr]
            sp.disp_p1 = float(pv.dispersal_distance_distr_param1)
            sp.disp_p2 = float(pv.dispersal_distance_distr_param2)
            for key, pre in (('move_surf', 'move_surf'), ('disp_surf', 'disp_surf')):
                if key in mv.keys():
                    ms = mv[key]
                    lyr = land._get_lyr_num(ms['layer'])
                    setattr(sp, pre, nat.SURF_MIXTURE if ms.get('mixture', True)
                            else nat.SURF_UNIMODAL)
                    setattr(sp, pre + '_layer', int(lyr))
                    kap = ms.get('vm_distr_kappa', 12)
                    setattr(sp, pre + '_kappa', float(12 if kap is None else kap))
                    setattr(self, '_' + key, True)
        sp.res_ratio[0], sp.res_ratio[1] = [float(v) for v in land._res_ratio]
        sp.K_layer = int(self.K_layer)
        sp.K_factor = float(self.K_factor)
        return sp

    def _make_device(self, land, N0, cap=None):
        L = self.gen_arch.L if self.gen_arch is not None else 0
        n_traits = (len(self.gen_arch.traits) if (self.gen_arch is not None and
                                                  self.gen_arch.traits is not None) else 0)
        if cap is None:
            factor = float(os.environ.get('GNX_CAP_FACTOR', '2.5'))
            cap = int(factor * max(N0, _sum_K(land, self.K_layer, self.K_factor))) + 1024
        self._cap = cap
        cap_inds, cap_rows = self._capacities(cap, N0)
        dev = nat.Device(land.dim[0], land.dim[1], land.n_lyrs, L=L, n_traits=n_traits,
                         cap_inds=cap_inds, cap_rows=cap_rows, seed=self._seed,
                         device=self._device_ordinal)
        dev.upload_rasters(land._stack())
        dev.set_species_params(self._species_params_struct(land))
        # Offspring ids (reference structs/species.py:614-619: max_ind_idx + 1 ... in the order of
        # the mating pairs, which there is a Python set's - unspecified): virtual tile by virtual
        # tile of a fixed 8 x 8 blocking of the landscape wherever its dimensions allow it, the
        # order a run over several GPUs can hand out without telling every pair to every rank
        # (csrc/gnx_comm.hip) - so that the same model script gives the same individuals, id by
        # id, on one GPU and on eight.  GNX_ID_ORDER=0: the (hash cell, focal id) order of the
        # whole landscape everywhere.
        if self._tile_major_ids(land):
            dev.set_id_order(1)
        self._land_ref = land
        self._dev = dev
        self._upload_gen_arch()
        return dev

    def _tile_major_ids(self, land):
        if os.environ.get('GNX_ID_ORDER', '1') == '0' or self._pv.mating_radius is None:
            return False
        return land.dim[0] % 8 == 0 and land.dim[1] % 8 == 0

    def _capacities(self, cap, N0):
        """(individual slots, genome rows) of the device state"""
        return cap, cap

    def _upload_gen_arch(self):
        ga = self.gen_arch
        if ga is None:
            return
        dev = self._dev
        dev.set_recomb_paths(ga.recombinations._paths)
        if ga.traits is not None:
            for t, trt in ga.traits.items():
                dev.set_trait(t, trt.loci, trt.alpha, trt.lyr_num, trt.phi, trt.gamma,
                              trt.univ_adv)
        dev.set_dominance(ga.dom if ga._use_dom else None)
        dev.set_deleterious(ga.delet_loci, ga.delet_loci_s)

    # -- dict-like read API ------------------------------------------------------
    def __len__(self):
        return int(self._dev.N) if self._dev is not None else 0

    def _field(self, field):
        """one per-individual array of the device state (a tiled species gathers the
        tiles' arrays: structs/tiled.py)"""
        return self._dev.download(field)

    def _locus_counts(self):
        """per-locus counts of 1-alleles and of heterozygotes (sim/stats.py)"""
        return self._dev.stats_locus_counts()

    def _after_init_population(self, N):
        """hook between init_population and the first spatial snapshot"""

    def _ids_sorted(self):
        ids = self._field(nat.F_ID)
        order = np.argsort(ids, kind='stable')
        return ids, order

    def __iter__(self):
        ids, order = self._ids_sorted()
        return iter(ids[order].tolist())

    def keys(self):
        return [*self]

    def __contains__(self, idx):
        return idx in set(self.keys())

    def values(self):
        return [*self._get_individs(np.array(self.keys(), dtype=np.int64)).values()]

    def items(self):
        return [*self._get_individs(np.array(self.keys(), dtype=np.int64)).items()]

    def _set_coords_and_cells(self):
        """write the coordinates assigned through Individual objects to the device and
        refresh e (reference structs/species.py:937-939 caches coords and cells)"""
        if not self._pending_xy:
            return
        ids = self._dev.download(nat.F_ID)
        x = self._dev.download(nat.F_X).copy()
        y = self._dev.download(nat.F_Y).copy()
        pos = {int(i): k for k, i in enumerate(ids)}
        for i, (nx, ny) in self._pending_xy.items():
            if i in pos:
                if nx is not None:
                    x[pos[i]] = nx
                if ny is not None:
                    y[pos[i]] = ny
        self._pending_xy = {}
        self._dev.set_positions(x, y)

    def __getitem__(self, idx):
        ids = self._field(nat.F_ID)
        w = np.nonzero(ids == idx)[0]
        if w.size == 0:
            raise KeyError(idx)
        s = int(w[0])
        d = self._dev
        g = None
        if self.gen_arch is not None and self.burned and d.L > 0:
            g = self._unpack(d.download_genomes([s]))[0]
        e = self._field(nat.F_E)[:, s].astype(np.float64).tolist()
        z = self._field(nat.F_Z)[:, s].astype(np.float64).tolist() if d.n_traits else []
        return Individual(int(idx), float(self._field(nat.F_X)[s]),
                          float(self._field(nat.F_Y)[s]), int(self._field(nat.F_AGE)[s]),
                          int(self._field(nat.F_SEX)[s]), e, z,
                          float(self._field(nat.F_FIT)[s]), g, spp=self)

    def _get_individs(self, ids):
        """{id: Individual} for the listed ids (ascending), one download per field;
        genomes are left out (the writers fetch the sample's genotypes separately)"""
        ids = np.asarray(ids, dtype=np.int64)
        d = self._dev
        all_ids, order = self._ids_sorted()
        sorted_ids = all_ids[order]
        pos = np.searchsorted(sorted_ids, ids)
        assert (pos < len(sorted_ids)).all() and (sorted_ids[pos] == ids).all(), (
            'some requested individuals are not alive')
        slots = order[pos]
        x, y = self._field(nat.F_X)[slots], self._field(nat.F_Y)[slots]
        age, sex = self._field(nat.F_AGE)[slots], self._field(nat.F_SEX)[slots]
        fit = self._field(nat.F_FIT)[slots]
        e = self._field(nat.F_E)[:, slots].astype(np.float64).T
        z = (self._field(nat.F_Z)[:, slots].astype(np.float64).T if d.n_traits
             else np.zeros((len(ids), 0)))
        return {int(i): Individual(int(i), float(x[k]), float(y[k]), int(age[k]), int(sex[k]),
                                   e[k].tolist(), z[k].tolist(), float(fit[k]), None, spp=self)
                for k, i in enumerate(ids)}

    def __str__(self):
        return "%s\n%i Individuals (on MI355X, slots %i)\n" % (str(type(self)), len(self),
                                                               self._cap)

    __repr__ = __str__

    # -- small setters used by the Model's function queue -------------------------
    def _set_K(self, land):
        """reference structs/species.py:545-546: K = K-layer raster * K_factor (this
        also drops whatever a demographic change had scaled K to, as there)"""
        self._K = land[self.K_layer].rast * self.K_factor
        if self._dev is not None:
            self._dev.upload_layer(self.K_layer, land[self.K_layer].rast)
            self._dev.set_K_raster(None)
            self._K_explicit = False

    def _set_N(self, N):
        self._N_cache = N

    @property
    def N(self):
        """Current density raster (reference attribute Species.N, set by
        _calc_density(set_N=True) inside _do_pop_dynamics)."""
        try:
            return self._dev.download_raster(nat.R_N)
        except nat.GnxError:
            return None

    def _set_Nt(self):
        self.Nt.append(len(self))

    def _set_t(self):
        self.t += 1

    def _reset_t(self):
        self.t = -1

    def _check_extinct(self):
        return len(self) == 0

    # -- the hot path ---------------------------------------------------------------
    def _set_age_stage(self):
        """reference structs/species.py:567-569"""
        self._dev.age()

    def _do_movement(self, land=None):
        """reference structs/species.py:582-585 (+ _set_e, _set_coords_and_cells)"""
        self._dev.move()

    def _do_pop_dynamics(self, land=None):
        """reference structs/species.py:822-833 -> ops/demography.py:183-330"""
        with_selection = self.selection and self.burned
        burn = not self.burned
        dev = self._dev
        n_before = dev.N
        for attempt in range(6):
            try:
                dev.pop_dynamics_mate(burn)
                break
            except nat.GnxError as e:
                # the reference's population is a dict that simply grows; here the slots
                # are preallocated, so make room and repeat the call (it failed before any
                # offspring was written and every draw is keyed by id and step: the
                # repeated call takes the same decisions)
                if 'capacity exceeded' not in str(e) or attempt == 5 or \
                        getattr(self, '_comm', None) is not None:
                    raise
                dev = self._grow_device()
        n_after, births, _ = dev.counts()
        self.n_births.append(int(births))
        if births:
            self.max_ind_idx += int(births)
        if self._tt is not None and not burn and births > 0:
            child, par, keys, starts, xy = dev.last_births()
            self._tt.add_births(self.t, child, par, keys, starts, xy)
        if self.mutate and not burn and births > 0:
            self._do_mutation(n_before, int(births))
        dev.pop_dynamics_die(burn, with_selection)
        _, _, deaths = dev.counts()
        self.n_deaths.append(int(deaths))
        dev.step_index = dev.step_index + 1
        if self._check_extinct():
            self.extinct = True

    def _can_walk_on_device(self):
        """nothing in this Species' time step needs the host between two steps: no mutation
        (host draws), no pedigree recording, no scheduled parameter changes"""
        return (self.burned and not self.extinct and not self.mutate and self._tt is None and
                self._changer is None and getattr(self, '_comm', None) is None)

    def _walk_on_device(self, T):
        """T main time steps in ONE call into the library (gnx_walk: _set_age_stage,
        _do_movement, _do_pop_dynamics T times, reference structs/species.py:567-585, 822-833):
        Nt, n_births and n_deaths of every step come back afterwards in one piece.  Returns
        the number of steps taken (fewer than T if the Species went extinct)."""
        dev = self._dev
        with_selection = self.selection and self.burned
        done = 0
        while done < T and not self.extinct:
            # the device-driven steps cannot grow the device state (the queue's steps can,
            # _grow_device): they run in pieces, and only while the population leaves a third
            # of the capacity free - a piece whose births would not fit ends in an error
            n_now = len(self)
            if n_now > 0.66 * self._cap:
                break
            # (the fuller the device, the shorter the piece: a population can grow by about R per
            # step - a piece must not be able to outgrow the free third)
            room = max(self._cap - n_now, 1)
            grow = max(0.02, float(getattr(self._pv, 'R', 0.5)) * 0.25) * max(n_now, 1)
            chunk = int(max(1, min(T - done, 256, 0.5 * room / grow)))
            try:
                dev.walk(chunk, False, with_selection)
            except nat.GnxError as e:
                # A walk that ran out of slots or genome rows is NOT recoverable: the step
                # that did not fit dropped births (or let offspring share genome rows) and the
                # steps enqueued behind it started from that state; a host-driven step fails
                # half-way (aged and moved, not yet born).  The library keeps such steps out
                # of gnx_walk_history; here the run ends loudly, as the reference would on an
                # inconsistent population - the pieces above are sized so that it does not
                # happen (0.66 of the capacity, half the free room per piece).
                if 'capacity exceeded' in str(e):
                    raise nat.GnxError(
                        '%s - inside Model.walk the device-driven steps cannot move the '
                        'population to a larger device state; start with more headroom '
                        '(GNX_CAP_FACTOR, now %.2f x N0)' % (e, self._cap / float(self.start_N or 1))
                    ) from e
                raise
            n0, births, deaths = dev.walk_history(chunk)
            for a, b, d in zip(n0.tolist(), births.tolist(), deaths.tolist()):
                if a == 0:
                    self.extinct = True
                    break
                self.n_births.append(int(b))
                self.n_deaths.append(int(d))
                self.max_ind_idx += int(b)
                self.t += 1
                done += 1
                if a + b - d == 0:
                    # (as in the queue: an extinct Species' step appends no Nt, sim/model.py:776-787)
                    self.extinct = True
                    break
                self.Nt.append(int(a + b - d))
        return done

    def _grow_device(self, factor=2.0):
        """a larger device state with the same population (capacity is an implementation
        detail of the build: GNX_CAP_FACTOR sets the initial headroom)"""
        d = self._dev
        has_geno = bool(d.L > 0 and self.gen_arch is not None and self.burned and
                        self.__dict__.get('_genomes_assigned', False))
        keep = dict(x=d.download(nat.F_X), y=d.download(nat.F_Y), age=d.download(nat.F_AGE),
                    sex=d.download(nat.F_SEX), id=d.download(nat.F_ID), step=d.step_index,
                    geno=d.download(nat.F_GENO) if has_geno else None)
        d.close()
        self._make_device(self._land_ref, len(keep['x']), cap=int(self._cap * factor) + 1024)
        nd = self._dev
        nd.upload_population(keep['x'], keep['y'], keep['age'], keep['sex'], keep['id'])
        if keep['geno'] is not None:
            nd.upload_genomes(keep['geno'])
        nd.step_index = keep['step']
        nd.set_max_id(self.max_ind_idx)
        if self.__dict__.get('_K_explicit', False):
            nd.set_K_raster(np.asarray(self._K, dtype=np.float64))
        return nd

    def _do_mutation(self, first_slot, n_offspring):
        """ops/mutation.py:169-206 on the new offspring (slots
        [first_slot, first_slot + n_offspring)).  Genotypes are tracked in full, so
        every mutation sets allele 1 at a never-mutated locus of one homologue of a
        random offspring (infinite sites); trait / deleterious mutations also extend
        the trait's locus table (structs/genome.py:753-788) and refresh the mutant's
        phenotype (ops/mutation.py:121-123)."""
        ga = self.gen_arch
        rng = self._rng
        n_muts = rng.binomial(n=n_offspring * ga.L, p=ga._mu_tot)
        if n_muts == 0 or not ga._mutables:
            return
        n_muts = min(n_muts, len(ga._mutables))
        kinds = ga._draw_mut_types(n_muts)
        slots = first_slot + rng.randint(0, n_offspring, n_muts)
        homs = rng.binomial(1, 0.5, n_muts)
        loci = [ga._mutables.pop() for _ in range(n_muts)]
        dirty = False
        for kind, locus in zip(kinds, loci):
            if kind == 'neut':
                continue
            if kind == 'delet':
                ga._add_nonneut_locus(locus, delet_s=ga._draw_delet_s())
            else:
                ga._add_nonneut_locus(locus, trait_nums=[int(kind[1:])])
            dirty = True
        if dirty:
            self._upload_gen_arch()
        self._dev.mutate(slots, loci, homs)
        if self._tt is not None:
            self._tt.add_mutations(self._dev.download(nat.F_ID)[slots], loci, homs)
        if dirty and ga.traits is not None:
            for s in set(int(v) for v, k in zip(slots, kinds) if k != 'neut'):
                self._dev.set_z_range(s, 1)
        if self.mut_log:
            ids = self._dev.download(nat.F_ID)
            with open(self.mut_log, 'a') as f:
                for kind, s, locus in zip(kinds, slots, loci):
                    f.write('MUTATION: %s\n\t INDIVIDUAL %i,  LOCUS %i\n\t timestep %i\n\n'
                            % (kind, ids[s], locus, self.t))

    def _set_genomes_and_tables(self, burn_T, T):
        """reference structs/species.py:956-967,1080-1094 (no-tskit branch) +
        structs/genome.py:1108-1157."""
        ga = self.gen_arch
        n_births_tail = self.n_births[-int(burn_T):] if self.n_births else [0]
        est_tot_muts = float(np.mean(n_births_tail)) * ga.L * (ga._mu_tot or 0) * T
        _genome._check_mutation_rates(ga, est_tot_muts, burn_T, T)
        n = _genome._starting_mutation_counts(len(self), ga.p)
        self._dev.assign_genomes(n)
        self._genomes_assigned = True
        self._start_pedigree()

    # founders of the tree-sequence tables: the population at genome assignment
    _PEDIGREE_MAX_BITS = 2e8

    def _start_pedigree(self):
        self._tt = None
        if not self._record_pedigree:
            return
        ga = self.gen_arch
        if len(self) * ga.L * 2 > self._PEDIGREE_MAX_BITS:
            import warnings
            warnings.warn("'use_tskit': True - the spatial pedigree is recorded as plain "
                          "tree-sequence tables on the host for models up to %.0e genotype "
                          "bits; this one is larger: not recorded. "
                          "Genotypes are tracked in full on the device either way."
                          % self._PEDIGREE_MAX_BITS)
            return
        from .pedigree import TreeTables
        off, loci = ga.recombinations._breakpoints()
        self._tt = TreeTables(ga.L, off, loci)
        ids, order = self._ids_sorted()
        self._tt.add_founders(ids[order], self._get_coords(), self._get_genotypes())

    def _set_z(self):
        self._dev.set_z()

    def _remove_individuals(self, individs=None, n=None, n_left=None, keep_sites_tab=False,
                            check_extinct=False, verbose=False):
        """remove listed or randomly drawn individuals (reference
        structs/species.py:1559-1640); their genome rows go back on the free stack"""
        given = [p is not None for p in (individs, n, n_left)]
        assert sum(given) == 1, ("One of 'individs', 'n', and 'n_left' must be provided, the "
                                 "other two must be None.")
        ids = self._dev.download(nat.F_ID)
        if individs is None:
            if n_left is not None:
                assert 0 <= n_left <= len(self), (
                    "'n_left' must be a number between 0 and the current size of the "
                    "population (%i)." % len(self))
                n = len(self) - n_left
            assert isinstance(n, (int, np.integer)) and n >= 0, (
                "'n' must either be a non-negative int or None.")
            assert n <= len(self), ("Cannot remove more Individuals than currently exist "
                                    "(current population size is %i)." % len(self))
            individs = self._rng.choice(np.sort(ids), n, replace=False)
        individs = np.asarray(individs, dtype=np.int64)
        dead = np.isin(ids, individs)
        assert dead.sum() == individs.size, 'some of the listed Individuals do not exist'
        n_b4 = len(self)
        self._dev.op_mortality(dead.astype(np.uint8))
        assert n_b4 - len(self) == individs.size
        if verbose:
            print('\n%i Individuals successfully removed.\n' % individs.size)
        if check_extinct and self._check_extinct():
            self.extinct = True

    # -- burn-in spatial test (reference sim/burnin.py:21-91) -------------------------
    def _spatial_update(self):
        m, s = self._dev.spatial_diff_stats()
        self._burnin_spat_stats['mean'].append(m)
        self._burnin_spat_stats['std'].append(s)

    def _do_spatial_burnin_test(self, num_timesteps_back):
        self._spatial_update()
        return _burnin.spatial_test(self._burnin_spat_stats, num_timesteps_back)

    # -- accessors (reference structs/species.py:1347-1543) ---------------------------
    def _sorted(self, field):
        arr = self._field(field)
        ids, order = self._ids_sorted()
        return ids, order, arr

    def _select(self, vals, ids_sorted, individs):
        if individs is None:
            return vals
        pos = {int(i): k for k, i in enumerate(ids_sorted)}
        return vals[[pos[int(i)] for i in individs]]

    def _get_coords(self, individs=None, as_float=True):
        ids, order = self._ids_sorted()
        x = self._field(nat.F_X)[order].astype(np.float64)
        y = self._field(nat.F_Y)[order].astype(np.float64)
        coords = self._select(np.stack([x, y], axis=1), ids[order], individs)
        if not as_float:
            coords = np.int32(np.floor(coords))
        return np.atleast_2d(coords)

    def _get_cells(self, individs=None):
        return self._get_coords(individs=individs, as_float=False)

    def _get_x(self, individs=None):
        return self._get_coords(individs=individs)[:, 0]

    def _get_y(self, individs=None):
        return self._get_coords(individs=individs)[:, 1]

    def _get_e(self, lyr_num=None, individs=None):
        ids, order = self._ids_sorted()
        e = self._field(nat.F_E)[:, order].T.astype(np.float64)
        e = self._select(e, ids[order], individs)
        return e if lyr_num is None else e[:, lyr_num]

    def _get_z(self, trait_num=None, individs=None):
        ids, order = self._ids_sorted()
        z = self._field(nat.F_Z)[:, order].T.astype(np.float64)
        z = self._select(z, ids[order], individs)
        return z if trait_num is None else np.atleast_2d(z)[:, trait_num]

    def _get_fit(self, individs=None):
        ids, order = self._ids_sorted()
        return self._select(self._field(nat.F_FIT)[order].astype(np.float64),
                            ids[order], individs)

    def _get_age(self, individs=None):
        ids, order = self._ids_sorted()
        return self._select(self._field(nat.F_AGE)[order], ids[order], individs)

    def _get_sex(self, individs=None):
        ids, order = self._ids_sorted()
        return self._select(self._field(nat.F_SEX)[order], ids[order], individs)

    def _unpack(self, packed):
        L = self.gen_arch.L
        by = np.ascontiguousarray(packed).view(np.uint8).reshape(packed.shape[0], 2, -1)
        bits = np.unpackbits(by, axis=2, bitorder='little')[:, :, :L]
        return np.transpose(bits, (0, 2, 1)).astype(np.int8)

    def _packed_genomes(self, ids):
        """bit-packed genomes [n][2][W64] of the listed (ascending, living) ids"""
        all_ids = self._dev.download(nat.F_ID)
        order = np.argsort(all_ids, kind='stable')
        pos = np.searchsorted(all_ids[order], ids)
        return self._dev.download_genomes(order[pos])

    def _get_genotypes(self, loci=None, individs=None, biallelic=True, as_dict=False):
        """N x L x 2 int8 (or N x L means if biallelic=False), sorted by id
        (reference structs/species.py:1364-1448)."""
        if individs is None:
            ids, order = self._ids_sorted()
            out_ids = ids[order]
        else:
            out_ids = np.sort(np.asarray(individs, dtype=np.int64))
        gts = self._unpack(self._packed_genomes(out_ids))
        if loci is not None:
            gts = gts[:, np.asarray(loci), :]
        if not biallelic:
            gts = gts.mean(axis=2)
        if as_dict:
            return {int(i): g for i, g in zip(out_ids, gts)}
        return gts

    def _calc_fitness(self, trait_num=None, set_fit=True):
        """reference ops/selection.py:51-112.  Overall fitness (trait_num None) is what
        the death-probability kernel of the last _do_pop_dynamics stored; the fitness of
        one trait is recomputed here from the downloaded e and z with the same formula,
        w = max(1 - phi |e^(not univ_adv) - z|^gamma, 0.001)."""
        if trait_num is None or self.gen_arch is None or self.gen_arch.traits is None:
            return self._get_fit()
        trt = self.gen_arch.traits[trait_num]
        e = self._get_e()[:, trt.lyr_num]
        z = self._get_z()[:, trt.idx]
        phi = trt.phi if np.isscalar(trt.phi) else np.asarray(trt.phi)[
            tuple(self._get_cells()[:, ::-1].T)]
        w = 1 - phi * np.abs((e ** (not trt.univ_adv)) - z) ** trt.gamma
        return np.clip(w, a_min=0.001, a_max=None)

    def _calc_density(self, normalize=False, as_layer=False, set_N=False):
        """reference structs/species.py:845-882"""
        x = self._field(nat.F_X)
        y = self._field(nat.F_Y)
        _, dens = self._dev.op_density(x, y)
        if normalize:
            dens = (dens - dens.min()) / (dens.max() - dens.min())
        if set_N:
            return None
        return dens

    # -- snapshot / restore (deepcopy semantics of Model iterations) -------------------
    def _snapshot(self):
        d = self._dev
        snap = dict(x=d.download(nat.F_X), y=d.download(nat.F_Y), age=d.download(nat.F_AGE),
                    sex=d.download(nat.F_SEX), id=d.download(nat.F_ID),
                    step=d.step_index, Nt=list(self.Nt), n_births=list(self.n_births),
                    n_deaths=list(self.n_deaths), burned=self.burned, t=self.t,
                    max_ind_idx=self.max_ind_idx, extinct=self.extinct,
                    spat=copy.deepcopy(self._burnin_spat_stats), geno=None,
                    K=None if self._K is None else np.array(self._K, copy=True),
                    K_explicit=bool(self.__dict__.get('_K_explicit', False)),
                    # only mutation changes the genomic architecture within an iteration
                    gen_arch=(copy.deepcopy(self.gen_arch)
                              if self.gen_arch is not None and getattr(self, 'mutate', False)
                              else None),
                    pv=copy.deepcopy(self._pv.__dict__))
        if self.gen_arch is not None and self.burned and d.L > 0:
            snap['geno'] = d.download(nat.F_GENO)
        return snap

    def _restore(self, snap):
        """back to the snapshot in everything an iteration can change (the reference
        deep-copies the whole community, sim/model.py:386-399): population, genomes, the
        carrying capacity a demographic change has scaled (ops/change.py:633-651), the
        genomic architecture mutation has extended (structs/genome.py:753-788), the
        life-history parameters and the event schedules"""
        d = self._dev
        if snap.get('gen_arch') is not None:
            self.gen_arch = copy.deepcopy(snap['gen_arch'])
            self._upload_gen_arch()
        if snap['K_explicit'] and snap['K'] is not None:
            self.K = np.array(snap['K'], copy=True)
        else:
            self._K = None if snap['K'] is None else np.array(snap['K'], copy=True)
            d.set_K_raster(None)
            self._K_explicit = False
        d.upload_population(snap['x'], snap['y'], snap['age'], snap['sex'], snap['id'])
        if snap['geno'] is not None:
            d.upload_genomes(snap['geno'])
        self._genomes_assigned = snap['geno'] is not None
        d.step_index = snap['step']
        self._tt = None
        self.Nt = list(snap['Nt'])
        self.n_births = list(snap['n_births'])
        self.n_deaths = list(snap['n_deaths'])
        self.burned = snap['burned']
        self.t = snap['t']
        self.max_ind_idx = snap['max_ind_idx']
        self.extinct = snap['extinct']
        self._burnin_spat_stats = copy.deepcopy(snap['spat'])
        if snap['geno'] is not None:         # a new iteration starts a new pedigree
            self._start_pedigree()
        if self._changer is not None:        # events start over with the iteration
            self._changer = copy.deepcopy(self._changer_orig)
            self._pv.__dict__.update(copy.deepcopy(snap['pv']))
            self._dev.set_species_params(self._species_params_struct(self._land_ref))


def _make_K(spp, land, K_layer, K_factor):
    """reference structs/species.py:3258-3273"""
    lyrs = [lyr for lyr in land.values() if lyr.name == K_layer]
    assert len(lyrs) == 1, ("The K_layer parameter should point to a single Layer, "
                            "but instead %i Layers were found.") % len(lyrs)
    spp.K_layer = lyrs[0].idx
    spp.K_factor = K_factor
    lyrs[0]._is_K.append(spp.idx)
    spp.K = land[spp.K_layer].rast * spp.K_factor


def _make_species(land, name, idx, spp_params, burn=False, verbose=False, seed=0,
                  device=0, rng=None, comm=None):
    """reference structs/species.py:3276-3397"""
    rng = np.random if rng is None else rng
    init_params = copy.deepcopy(dict(spp_params.init))
    if verbose:
        print('\t\tMAKING SPECIES %s...\n' % name, flush=True)
    gen_arch = None
    if 'gen_arch' in spp_params.keys():
        if verbose:
            print('\t\t\tmaking genomic architecture...\n', flush=True)
        gen_arch = _genome._make_genomic_architecture(spp_params=spp_params, land=land,
                                                      rng=rng)
    if 'msprime' in init_params:
        raise NotImplementedError('msprime-seeded populations are outside the hot path '
                                  '(SURVEY section 2).')
    N = init_params.pop('N')
    cls, extra = Species, {}
    if comm is not None and comm.world > 1:
        from .tiled import TiledSpecies
        cls, extra = TiledSpecies, {'comm': comm}
    spp = cls(name=name, idx=idx, land=land, spp_params=spp_params,
              genomic_architecture=gen_arch, seed=seed, device=device, rng=rng, **extra)
    _make_K(spp, land, **init_params)
    if verbose:
        print('\t\t\tmaking individuals...\n', flush=True)
    spp._make_device(land, N)
    spp._dev.init_population(N)
    spp.start_N = N
    spp.max_ind_idx = N - 1
    spp._after_init_population(N)
    # the burn-in spatial tester takes its first count at creation
    # (reference sim/burnin.py:36-37)
    spp._spatial_update()
    # change events (reference structs/species.py:3376-3395); movement / dispersal
    # surfaces follow their Layer on the device, so only parameterised changes
    # need a changer
    if 'change' in spp_params.keys():
        from ..ops.change import _SpeciesChanger
        spp._changer = _SpeciesChanger(spp, spp_params.change, land=land, rng=rng)
        spp._changer_orig = copy.deepcopy(spp._changer)
    return spp
