"""ctypes binding of libgnxhip.so (include/gnx_hip.h).

The product has no CPU fallback: loading fails loudly if the shared library is
missing, and creating a device state fails if there is no HIP device.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GNX_LIB: another build of the same library - A/B runs of two builds on one box, tools/)
LIB_PATH = os.environ.get('GNX_LIB') or os.path.join(_HERE, 'libgnxhip.so')

# enums (include/gnx_hip.h)
DIST = {'lognormal': 0, 'wald': 1, 'levy': 2}
MATE_UNIFORM, MATE_NEAREST, MATE_INVERSE = 0, 1, 2
SURF_NONE, SURF_MIXTURE, SURF_UNIMODAL = 0, 1, 2
(F_X, F_Y, F_AGE, F_SEX, F_ID, F_E, F_Z, F_FIT, F_GROW, F_GENO) = range(10)
(R_N, R_NPAIRS, R_K, R_D, R_COUNTS) = range(5)
KERNELS = ['move', 'sort', 'permute', 'find_mates', 'pairs', 'offspring',
           'crossover', 'phenotype', 'density', 'death', 'compact', 'crossover_tail']

EXPORTS = [
    'gnx_create', 'gnx_destroy', 'gnx_last_error', 'gnx_words_per_hom', 'gnx_blocks_per_hom',
    'gnx_set_stream', 'gnx_synchronize', 'gnx_upload_rasters',
    'gnx_upload_layer', 'gnx_set_species_params', 'gnx_upload_population',
    'gnx_init_population', 'gnx_set_recomb_paths', 'gnx_set_trait',
    'gnx_set_dominance', 'gnx_set_deleterious', 'gnx_upload_genomes',
    'gnx_assign_genomes', 'gnx_set_z', 'gnx_age', 'gnx_move',
    'gnx_pop_dynamics', 'gnx_pop_dynamics_mate', 'gnx_pop_dynamics_die',
    'gnx_set_z_range', 'gnx_step', 'gnx_counts', 'gnx_step_index',
    'gnx_set_step_index', 'gnx_mutate', 'gnx_download', 'gnx_download_genomes',
    'gnx_download_raster', 'gnx_spatial_diff_stats', 'gnx_op_move',
    'gnx_op_move_draws', 'gnx_op_find_pairs', 'gnx_op_crossover',
    'gnx_op_dispersal', 'gnx_op_density', 'gnx_density_lattice_dims', 'gnx_density_nmax',
    'gnx_op_death_probs', 'gnx_op_mortality', 'gnx_profiling',
    'gnx_kernel_time', 'gnx_tile_set', 'gnx_tile_export_migrants',
    'gnx_tile_export_halo', 'gnx_tile_get_staged', 'gnx_tile_import',
    'gnx_tile_import_ghosts', 'gnx_tile_pairs', 'gnx_tile_pair_info',
    'gnx_density_bin_count', 'gnx_get_bins', 'gnx_set_bins',
    'gnx_tile_offspring', 'gnx_tile_get_requests', 'gnx_tile_serve_gametes',
    'gnx_tile_put_gametes', 'gnx_tile_finish_births', 'gnx_tile_die',
    'gnx_set_max_id', 'gnx_stats_locus_counts', 'gnx_stats_ld',
    'gnx_tile_export_migrants_dev', 'gnx_tile_export_halo_dev', 'gnx_tile_staged_ptrs',
    'gnx_tile_import_dev', 'gnx_tile_import_ghosts_dev', 'gnx_tile_pair_ptrs',
    'gnx_tile_offspring_dev', 'gnx_tile_group_requests', 'gnx_tile_serve_gametes_dev',
    'gnx_tile_put_gametes_dev', 'gnx_tile_bins_ptr', 'gnx_set_k_raster', 'gnx_last_births', 'gnx_set_positions', 'gnx_n_slots', 'gnx_stats_ld_counts',
    'gnx_set_defer_crossover', 'gnx_last_crossover_births', 'gnx_set_crossover_overlap', 'gnx_set_crossover_split', 'gnx_debug_halves', 'gnx_spatial_diff_sums', 'gnx_last_crossover_jobs',
    'gnx_genome_info', 'gnx_measure_copy', 'gnx_totals', 'gnx_reset_totals',
    'gnx_step_begin', 'gnx_step_mid', 'gnx_step_end', 'gnx_step_many',
    'gnx_walk', 'gnx_walk_many', 'gnx_walk_history',
    'gnx_comm_unique_id', 'gnx_comm_init_rccl', 'gnx_comm_init_single', 'gnx_comm_local_create',
    'gnx_comm_local_join', 'gnx_comm_local_abort', 'gnx_comm_local_destroy', 'gnx_comm_free',
    'gnx_comm_bytes_sent', 'gnx_comm_selftest', 'gnx_tile_step', 'gnx_set_id_order',
    'gnx_tile2_route_begin', 'gnx_tile2_route_finish', 'gnx_tile2_requests_dev',
    'gnx_tile2_set_requests',
    'gnx_stream_ptr', 'gnx_tile2_move_route', 'gnx_tile2_route_ptrs', 'gnx_tile2_import',
    'gnx_tile2_pairs', 'gnx_tile2_offspring', 'gnx_tile2_serve', 'gnx_tile2_put',
    'gnx_tile2_finish_births', 'gnx_tile2_die', 'gnx_tile_pair_ptrs_nosync',
    'gnx_tile_step_begin', 'gnx_tile_step_births', 'gnx_tile_step_end', 'gnx_comm_probe',
    'gnx_tile2_pairs_mode', 'gnx_tile2_pairs_settle', 'gnx_tile2_settle_births',
    'gnx_tile2_vt_counts', 'gnx_tile2_vt_bases', 'gnx_tile_step_abort', 'gnx_comm_info', 'gnx_tile_walk',
]


class Config(C.Structure):
    _fields_ = [('W', C.c_int32), ('H', C.c_int32), ('n_layers', C.c_int32),
                ('L', C.c_int32), ('n_traits', C.c_int32),
                ('cap_inds', C.c_int64), ('cap_rows', C.c_int64),
                ('seed', C.c_uint64), ('device', C.c_int32),
                ('reserved', C.c_int32)]


class SpeciesParams(C.Structure):
    _fields_ = [
        ('b', C.c_double), ('R', C.c_double), ('n_births_lambda', C.c_double),
        ('n_births_fixed', C.c_int32), ('sexed', C.c_int32),
        ('p_male', C.c_double), ('mating_radius', C.c_double),
        ('mate_mode', C.c_int32), ('repro_age', C.c_int32 * 2),
        ('max_age', C.c_int32), ('d_min', C.c_double), ('d_max', C.c_double),
        ('window_width', C.c_double), ('move', C.c_int32),
        ('dir_mu', C.c_double), ('dir_kappa', C.c_double),
        ('move_distr', C.c_int32), ('move_p1', C.c_double),
        ('move_p2', C.c_double), ('disp_distr', C.c_int32),
        ('disp_p1', C.c_double), ('disp_p2', C.c_double),
        ('move_surf', C.c_int32), ('move_surf_layer', C.c_int32),
        ('move_surf_kappa', C.c_double), ('disp_surf', C.c_int32),
        ('disp_surf_layer', C.c_int32), ('disp_surf_kappa', C.c_double),
        ('res_ratio', C.c_double * 2), ('K_layer', C.c_int32),
        ('pad0', C.c_int32), ('K_factor', C.c_double)]


# gnx_ind_rec (include/gnx_hip.h) as a numpy record dtype (32 bytes)
IND_REC = np.dtype([('x', np.float32), ('y', np.float32), ('age', np.int32),
                    ('sex', np.int32), ('id', np.int64), ('fit', np.float32),
                    ('nbr_mask', np.int32)], align=True)
assert IND_REC.itemsize == 32

_lib = None


def load():
    """Load libgnxhip.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'geonomics_amd: %s is missing - build it with '
            '`python -m geonomics_amd.build` (hipcc, gfx950). There is no CPU '
            'fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.gnx_last_error.restype = C.c_char_p
    lib.gnx_step_index.restype = C.c_int64
    lib.gnx_n_slots.restype = C.c_int64
    lib.gnx_last_crossover_births.restype = C.c_int64
    lib.gnx_walk_history.restype = C.c_int64
    lib.gnx_comm_bytes_sent.restype = C.c_int64
    lib.gnx_destroy.restype = None
    _lib = lib
    return lib


def _ptr(a, ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class GnxError(RuntimeError):
    pass


class Device:
    """One device-resident Species state (opaque gnx_state handle)."""

    def __init__(self, W, H, n_layers, L=0, n_traits=0, cap_inds=1024,
                 cap_rows=None, seed=0, device=0):
        self.lib = load()
        self.cfg = Config(W=W, H=H, n_layers=n_layers, L=L, n_traits=n_traits,
                          cap_inds=int(cap_inds),
                          cap_rows=int(cap_inds if cap_rows is None else cap_rows),
                          seed=int(seed) & 0xFFFFFFFFFFFFFFFF, device=device)
        self.W, self.H, self.n_layers, self.L = W, H, n_layers, L
        self.n_traits = n_traits
        self.W64 = self.lib.gnx_words_per_hom(L) if L > 0 else 0
        h = C.c_void_p()
        rc = self.lib.gnx_create(C.byref(self.cfg), C.byref(h))
        if rc:
            raise GnxError(self.lib.gnx_last_error().decode())
        self.h = h

    @property
    def blocks_per_hom(self):
        """blocks a homologue is stored in (the crossover copies or shares whole blocks)"""
        return int(self.lib.gnx_blocks_per_hom(self.h))

    @property
    def h(self):
        """the opaque handle; a Device that was closed raises instead of handing the
        library a null pointer"""
        h = self.__dict__.get('_h')
        if h is None or not h:
            raise GnxError('this Device was closed (gnx_destroy): no device state behind it')
        return h

    @h.setter
    def h(self, v):
        self.__dict__['_h'] = v

    def close(self):
        h = self.__dict__.get('_h')
        if h is not None and h:
            self.lib.gnx_destroy(h)
            self.__dict__['_h'] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise GnxError(self.lib.gnx_last_error().decode())

    # -- setup ---------------------------------------------------------------
    def upload_rasters(self, rasts):
        r = _arr(rasts, np.float32)
        assert r.shape == (self.n_layers, self.H, self.W), r.shape
        self._chk(self.lib.gnx_upload_rasters(self.h, _ptr(r, C.c_float)))

    def upload_layer(self, layer, rast):
        r = _arr(rast, np.float32)
        assert r.shape == (self.H, self.W)
        self._chk(self.lib.gnx_upload_layer(self.h, int(layer), _ptr(r, C.c_float)))

    def set_K_raster(self, K):
        """explicit K raster (float64 [H][W]) or None for rast[K_layer] * K_factor"""
        if K is None:
            self._chk(self.lib.gnx_set_k_raster(self.h, None))
            return
        k = _arr(K, np.float64)
        assert k.shape == (self.H, self.W), k.shape
        self._chk(self.lib.gnx_set_k_raster(self.h, _ptr(k, C.c_double)))

    @property
    def births_fixed_lambda(self):
        """births per pair when n_births_fixed (structs/species.py:604-609), else 0"""
        sp = getattr(self, '_sp', None)
        return int(sp.n_births_lambda) if (sp is not None and sp.n_births_fixed) else 0

    def set_species_params(self, sp):
        self._sp = sp
        self.sp = sp
        self._chk(self.lib.gnx_set_species_params(self.h, C.byref(sp)))

    def upload_population(self, x, y, age, sex, ids):
        x = _arr(x, np.float32)
        n = x.size
        y = _arr(y, np.float32)
        age = _arr(age, np.int32)
        sex = _arr(sex, np.uint8)
        ids = _arr(ids, np.int64)
        assert y.size == n and age.size == n and sex.size == n and ids.size == n
        self._chk(self.lib.gnx_upload_population(
            self.h, C.c_int64(n), _ptr(x, C.c_float), _ptr(y, C.c_float),
            _ptr(age, C.c_int32), _ptr(sex, C.c_uint8), _ptr(ids, C.c_int64)))

    def set_positions(self, x, y):
        x, y = _arr(x, np.float32), _arr(y, np.float32)
        assert x.size == self.N and y.size == self.N
        self._chk(self.lib.gnx_set_positions(self.h, _ptr(x, C.c_float), _ptr(y, C.c_float)))

    def init_population(self, n):
        self._chk(self.lib.gnx_init_population(self.h, C.c_int64(n)))

    def set_recomb_paths(self, paths_packed):
        p = _arr(paths_packed, np.uint64)
        assert p.ndim == 2 and p.shape[1] == self.W64, (p.shape, self.W64)
        self._chk(self.lib.gnx_set_recomb_paths(self.h, p.shape[0],
                                                _ptr(p, C.c_uint64)))

    def set_trait(self, t, loci, alpha, layer, phi, gamma, univ_adv):
        loci = _arr(loci, np.int32)
        alpha = _arr(alpha, np.float64)
        assert loci.size == alpha.size
        phi_rast = None
        phi_s = 0.0
        if np.ndim(phi) == 2:
            phi_rast = _arr(phi, np.float32)
            assert phi_rast.shape == (self.H, self.W)
        else:
            phi_s = float(phi)
        self._chk(self.lib.gnx_set_trait(
            self.h, int(t), int(loci.size), _ptr(loci, C.c_int32),
            _ptr(alpha, C.c_double), int(layer), C.c_double(phi_s),
            _ptr(phi_rast, C.c_float), C.c_double(float(gamma)),
            int(bool(univ_adv))))

    def set_dominance(self, dom):
        d = None if dom is None else _arr(dom, np.uint8)
        self._chk(self.lib.gnx_set_dominance(self.h, _ptr(d, C.c_uint8)))

    def set_deleterious(self, loci, s):
        loci = _arr(loci, np.int32)
        s = _arr(s, np.float64)
        self._chk(self.lib.gnx_set_deleterious(self.h, int(loci.size),
                                               _ptr(loci, C.c_int32),
                                               _ptr(s, C.c_double)))

    def upload_genomes(self, geno):
        g = _arr(geno, np.uint64)
        assert g.shape == (self.N, 2, self.W64), (g.shape, self.N, self.W64)
        self._chk(self.lib.gnx_upload_genomes(self.h, _ptr(g, C.c_uint64)))

    def assign_genomes(self, n_per_site):
        n = _arr(n_per_site, np.int32)
        assert n.size == self.L
        self._chk(self.lib.gnx_assign_genomes(self.h, _ptr(n, C.c_int32)))

    def set_z(self):
        self._chk(self.lib.gnx_set_z(self.h))

    # -- stepping ------------------------------------------------------------
    def age(self):
        self._chk(self.lib.gnx_age(self.h))

    def move(self):
        self._chk(self.lib.gnx_move(self.h))

    def pop_dynamics(self, burn, with_selection):
        self._chk(self.lib.gnx_pop_dynamics(self.h, int(bool(burn)),
                                            int(bool(with_selection))))

    def pop_dynamics_mate(self, burn):
        self._chk(self.lib.gnx_pop_dynamics_mate(self.h, int(bool(burn))))

    def pop_dynamics_die(self, burn, with_selection):
        self._chk(self.lib.gnx_pop_dynamics_die(self.h, int(bool(burn)),
                                                int(bool(with_selection))))

    def set_z_range(self, first, n):
        self._chk(self.lib.gnx_set_z_range(self.h, C.c_int64(first), C.c_int64(n)))

    def step(self, burn, with_selection):
        self._chk(self.lib.gnx_step(self.h, int(bool(burn)),
                                    int(bool(with_selection))))

    def walk(self, T, burn, with_selection):
        """T time steps in one call, counts kept on the device (gnx_walk): no read-back and
        one graph launch per step"""
        self._chk(self.lib.gnx_walk(self.h, C.c_int64(int(T)), int(bool(burn)),
                                    int(bool(with_selection))))

    def walk_history(self, max_steps=1 << 16):
        """(N at the start, births, deaths) of the last walk's steps, int64 arrays"""
        n = np.zeros(max_steps, np.int64)
        b = np.zeros(max_steps, np.int64)
        d = np.zeros(max_steps, np.int64)
        k = self.lib.gnx_walk_history(self.h, C.c_int64(max_steps), _ptr(n, C.c_int64),
                                      _ptr(b, C.c_int64), _ptr(d, C.c_int64))
        return n[:k], b[:k], d[:k]

    # -- one tiled step per call, exchanges issued by the library (csrc/gnx_comm.hip) -------
    def comm_init_rccl(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self.lib.gnx_comm_init_rccl(self.h, buf, int(rank), int(world)))

    def comm_init_single(self):
        self._chk(self.lib.gnx_comm_init_single(self.h))

    def comm_local_join(self, group, rank):
        self._chk(self.lib.gnx_comm_local_join(self.h, group, int(rank)))

    def set_id_order(self, mode):
        """0: offspring ids in (hash cell, focal id) order (default); 1: virtual-tile-major"""
        self._chk(self.lib.gnx_set_id_order(self.h, int(mode)))
        self._id_order = int(mode)

    @property
    def id_order(self):
        return getattr(self, '_id_order', 0)

    def comm_selftest(self):
        """known words through every operation of the transport (collective); raises GnxError"""
        self._chk(self.lib.gnx_comm_selftest(self.h))

    def comm_free(self):
        self._chk(self.lib.gnx_comm_free(self.h))

    @property
    def comm_bytes_sent(self):
        return int(self.lib.gnx_comm_bytes_sent(self.h))

    def comm_info(self):
        """what the handle's tile communicator says about itself (gnx_comm_info): transport, rank,
        world, ncclCommCount / UserRank / CuDevice, device ordinal, steps, per-phase host ms per step"""
        out = np.zeros(16, np.int64)
        self._chk(self.lib.gnx_comm_info(self.h, _ptr(out, C.c_int64)))
        steps = max(int(out[7]), 1)
        names = ('route_and_count_exchange', 'migrant_ghost_exchange_import',
                 'sort_pairs_second_count_exchange', 'births_gamete_service',
                 'allreduce_deaths')
        return {'transport': ('single', 'rccl', 'local')[int(out[0])], 'rank': int(out[1]),
                'world': int(out[2]), 'nccl_comm_count': int(out[3]),
                'nccl_comm_user_rank': int(out[4]), 'nccl_comm_device': int(out[5]),
                'hip_device': int(out[6]), 'tile_steps': int(out[7]),
                'phase_ms_per_step': {n: float(out[8 + k]) / 1e3 / steps
                                      for k, n in enumerate(names)},
                'bytes_sent': int(out[13]), 'gc_runs': int(out[14])}

    def tile_step_abort(self):
        self._chk(self.lib.gnx_tile_step_abort(self.h))

    def tile_step(self, burn, with_selection, exact=True):
        """one time step of this tile and, through its communicator, of the whole tiled
        landscape (gnx_tile_step) -> (N, births, deaths), see include/gnx_hip.h"""
        out = np.zeros(3, np.int64)
        self._chk(self.lib.gnx_tile_step(self.h, int(bool(burn)), int(bool(with_selection)),
                                         int(bool(exact)), _ptr(out, C.c_int64)))
        self._id_order = 1          # (the library numbers a tiled step's offspring tile-major)
        return int(out[0]), int(out[1]), int(out[2])

    def tile_walk(self, T, burn, with_selection, exact=True):
        """T tiled steps in one call, no compaction between them (gnx_tile_walk) -> ((N, births,
        deaths) of the last step as tile_step reports them, sum of the global N at the start of
        every step, sum of the global births)"""
        out = np.zeros(5, np.int64)
        self._chk(self.lib.gnx_tile_walk(self.h, C.c_int64(int(T)), int(bool(burn)),
                                         int(bool(with_selection)), int(bool(exact)),
                                         _ptr(out, C.c_int64)))
        self._id_order = 1
        return (int(out[0]), int(out[1]), int(out[2])), int(out[3]), int(out[4])

    def tile_step_begin(self, burn):
        """the tiled step up to and including the births (gnx_tile_step_begin) -> (first id,
        number) of the step's offspring over ALL tiles; the host's work on the newborns
        (mutations, pedigree rows) goes between this and tile_step_end"""
        self._chk(self.lib.gnx_tile_step_begin(self.h, int(bool(burn))))
        self._id_order = 1
        a, b = C.c_int64(), C.c_int64()
        self._chk(self.lib.gnx_tile_step_births(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def tile_step_end(self, burn, with_selection, exact=True):
        out = np.zeros(3, np.int64)
        self._chk(self.lib.gnx_tile_step_end(self.h, int(bool(burn)), int(bool(with_selection)),
                                             int(bool(exact)), _ptr(out, C.c_int64)))
        return int(out[0]), int(out[1]), int(out[2])

    def tile2_vt_counts(self):
        """tile-major offspring ids: this tile's births per virtual tile, int64 [64]"""
        cnt = np.zeros(64, np.int64)
        self._chk(self.lib.gnx_tile2_vt_counts(self.h, _ptr(cnt, C.c_int64)))
        return cnt

    def tile2_vt_bases(self, bases):
        b = _arr(bases, np.int64)
        assert b.size == 64
        self._chk(self.lib.gnx_tile2_vt_bases(self.h, _ptr(b, C.c_int64)))

    def step_begin(self, burn):
        self._chk(self.lib.gnx_step_begin(self.h, int(bool(burn))))

    def step_mid(self, burn, with_selection):
        self._chk(self.lib.gnx_step_mid(self.h, int(bool(burn)), int(bool(with_selection))))

    def step_end(self, burn):
        self._chk(self.lib.gnx_step_end(self.h, int(bool(burn))))

    def counts(self):
        n, b, d = C.c_int64(), C.c_int64(), C.c_int64()
        self._chk(self.lib.gnx_counts(self.h, C.byref(n), C.byref(b), C.byref(d)))
        return n.value, b.value, d.value

    def totals(self):
        """dict(steps, ind_steps, births, deaths, xo_births, dd_steps) summed over the gnx_step calls
        since reset_totals() - host-side bookkeeping of the library, no device access"""
        out = np.zeros(6, np.int64)
        self._chk(self.lib.gnx_totals(self.h, _ptr(out, C.c_int64)))
        return dict(zip(('steps', 'ind_steps', 'births', 'deaths', 'xo_births', 'dd_steps'),
                        (int(v) for v in out)))

    def reset_totals(self):
        self._chk(self.lib.gnx_reset_totals(self.h))

    @property
    def N(self):
        return self.counts()[0]

    @property
    def step_index(self):
        return self.lib.gnx_step_index(self.h)

    @step_index.setter
    def step_index(self, v):
        self._chk(self.lib.gnx_set_step_index(self.h, C.c_int64(int(v))))

    def synchronize(self):
        self._chk(self.lib.gnx_synchronize(self.h))

    def set_defer_crossover(self, on):
        """cut the offspring's genomes after the death draws, survivors only (default)"""
        self._chk(self.lib.gnx_set_defer_crossover(self.h, int(bool(on))))

    def set_crossover_overlap(self, mode):
        """0 / False (default): full-width crossover beside the compaction and the next
        movement, the next cell sort waits for it; 1 / True: a narrow crossover beside the whole
        next step; 2: nothing runs beside the crossover"""
        self._chk(self.lib.gnx_set_crossover_overlap(self.h, int(mode)))

    def debug_halves(self):
        """(logical blocks in use / 2, broken references, collections so far, physical blocks
        in use, free blocks, blocks in all) of the shared genome blocks, after a collection"""
        out = np.zeros(6, np.int64)
        self._chk(self.lib.gnx_debug_halves(self.h, _ptr(out, C.c_int64)))
        return out

    def genome_info(self):
        """host-side view of the genome blocks, no device access: dict(NB, BW, gc_runs,
        row_spread, sparse, free_blocks_est, free_rows, deferred)"""
        out = np.zeros(8, np.int64)
        self._chk(self.lib.gnx_genome_info(self.h, _ptr(out, C.c_int64)))
        return dict(zip(('NB', 'BW', 'gc_runs', 'row_spread', 'sparse', 'free_blocks_est',
                         'free_rows', 'deferred'), (int(v) for v in out)))

    def set_crossover_split(self, wide_per_1024):
        """share (/1024) of a deferred crossover's jobs that runs at full width before the next
        cell sort; the rest runs narrow beside the sort and the kernels after it"""
        self._chk(self.lib.gnx_set_crossover_split(self.h, int(wide_per_1024)))

    def last_crossover_jobs(self):
        """int32 [n, 4]: the parent's two physical blocks, the block written,
        (path * 2 + start homologue) | block index << 24"""
        n = C.c_int64()
        self._chk(self.lib.gnx_last_crossover_jobs(self.h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 4), np.int32)
        if n.value:
            self._chk(self.lib.gnx_last_crossover_jobs(self.h, _ptr(out, C.c_int32), n.value,
                                                       C.byref(n)))
        return out

    @property
    def last_crossover_births(self):
        return self.lib.gnx_last_crossover_births(self.h)

    def set_stream(self, stream_ptr):
        self._chk(self.lib.gnx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def mutate(self, slots, loci, homs):
        slots = _arr(slots, np.int64)
        loci = _arr(loci, np.int32)
        homs = _arr(homs, np.uint8)
        self._chk(self.lib.gnx_mutate(self.h, int(slots.size),
                                      _ptr(slots, C.c_int64),
                                      _ptr(loci, C.c_int32),
                                      _ptr(homs, C.c_uint8)))

    # -- read-back -----------------------------------------------------------
    _FIELD_DT = {F_X: np.float32, F_Y: np.float32, F_AGE: np.int32,
                 F_SEX: np.uint8, F_ID: np.int64, F_FIT: np.float32,
                 F_GROW: np.int32}

    def download(self, field):
        n = int(self.lib.gnx_n_slots(self.h))     # = N, plus the ghosts while a tiled step runs
        if field == F_E:
            out = np.empty((self.n_layers, n), dtype=np.float32)
        elif field == F_Z:
            out = np.empty((self.n_traits, n), dtype=np.float32)
        elif field == F_GENO:
            out = np.empty((n, 2, self.W64), dtype=np.uint64)
        else:
            out = np.empty(n, dtype=self._FIELD_DT[field])
        if out.size:
            self._chk(self.lib.gnx_download(self.h, field,
                                            out.ctypes.data_as(C.c_void_p),
                                            C.c_int64(out.nbytes)))
        return out

    def download_genomes(self, slots):
        slots = _arr(slots, np.int64)
        out = np.empty((slots.size, 2, self.W64), dtype=np.uint64)
        if slots.size:
            self._chk(self.lib.gnx_download_genomes(
                self.h, C.c_int64(slots.size), _ptr(slots, C.c_int64),
                _ptr(out, C.c_uint64)))
        return out

    def download_raster(self, which):
        out = np.empty((self.H, self.W), dtype=np.float64)
        self._chk(self.lib.gnx_download_raster(self.h, which, _ptr(out, C.c_double)))
        return out

    def spatial_diff_stats(self):
        m, s = C.c_double(), C.c_double()
        self._chk(self.lib.gnx_spatial_diff_stats(self.h, C.byref(m), C.byref(s)))
        return m.value, s.value

    def spatial_diff_sums(self):
        a, b = C.c_double(), C.c_double()
        self._chk(self.lib.gnx_spatial_diff_sums(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- operator-level (tests) ----------------------------------------------
    def op_move(self, theta, dist):
        t = _arr(theta, np.float32)
        d = _arr(dist, np.float32)
        assert t.size == self.N and d.size == self.N
        self._chk(self.lib.gnx_op_move(self.h, _ptr(t, C.c_float), _ptr(d, C.c_float)))

    def op_move_draws(self):
        n = self.N
        t = np.empty(n, np.float32)
        d = np.empty(n, np.float32)
        self._chk(self.lib.gnx_op_move_draws(self.h, _ptr(t, C.c_float),
                                             _ptr(d, C.c_float)))
        return t, d

    def op_find_pairs(self, keep=None):
        n = self.N
        k = None if keep is None else _arr(keep, np.uint8)
        mate = np.empty(n, np.int32)
        pairs = np.empty((max(n, 1), 2), np.int32)
        npairs = C.c_int64()
        self._chk(self.lib.gnx_op_find_pairs(self.h, _ptr(k, C.c_uint8),
                                             _ptr(mate, C.c_int32),
                                             _ptr(pairs, C.c_int32),
                                             C.byref(npairs)))
        return mate, pairs[:npairs.value].copy()

    def op_crossover(self, parent_slots, keys, start_homs):
        p = _arr(parent_slots, np.int32)
        k = _arr(keys, np.int32)
        s = _arr(start_homs, np.uint8)
        B = p.shape[0]
        assert p.shape == (B, 2) and k.shape == (B, 2) and s.shape == (B, 2)
        self._chk(self.lib.gnx_op_crossover(self.h, C.c_int64(B),
                                            _ptr(p, C.c_int32),
                                            _ptr(k, C.c_int32),
                                            _ptr(s, C.c_uint8)))

    def op_dispersal(self, mid_x, mid_y, theta, dist):
        mx = _arr(mid_x, np.float32)
        my = _arr(mid_y, np.float32)
        th = _arr(theta, np.float32)
        ds = _arr(dist, np.float32)
        A, B = th.shape
        ox = np.empty(B, np.float32)
        oy = np.empty(B, np.float32)
        used = np.empty(B, np.int32)
        self._chk(self.lib.gnx_op_dispersal(
            self.h, C.c_int64(B), int(A), _ptr(mx, C.c_float),
            _ptr(my, C.c_float), _ptr(th, C.c_float), _ptr(ds, C.c_float),
            _ptr(ox, C.c_float), _ptr(oy, C.c_float), _ptr(used, C.c_int32)))
        return ox, oy, used

    def lattice_dims(self):
        jx, jy = C.c_int32(), C.c_int32()
        self._chk(self.lib.gnx_density_lattice_dims(self.h, C.byref(jx), C.byref(jy)))
        return jx.value, jy.value

    def density_nmax(self):
        """N.max() of the last death probabilities' density raster (ops/demography.py:116)"""
        v = C.c_double()
        self._chk(self.lib.gnx_density_nmax(self.h, C.byref(v)))
        return v.value

    def op_density(self, x, y, want_raster=True):
        x = _arr(x, np.float32)
        y = _arr(y, np.float32)
        jx, jy = self.lattice_dims()
        nodes = np.empty((jy, jx), np.float64)
        rast = np.empty((self.H, self.W), np.float64) if want_raster else None
        self._chk(self.lib.gnx_op_density(self.h, C.c_int64(x.size),
                                          _ptr(x, C.c_float), _ptr(y, C.c_float),
                                          _ptr(nodes, C.c_double),
                                          _ptr(rast, C.c_double)))
        return nodes, rast

    def op_death_probs(self, with_selection, nodes_N, nodes_pairs=None):
        nN = _arr(nodes_N, np.float64)
        nP = None if nodes_pairs is None else _arr(nodes_pairs, np.float64)
        n = self.N
        p = np.empty(n, np.float64)
        d = np.empty(n, np.float64)
        self._chk(self.lib.gnx_op_death_probs(self.h, int(bool(with_selection)),
                                              _ptr(nN, C.c_double),
                                              _ptr(nP, C.c_double),
                                              _ptr(p, C.c_double),
                                              _ptr(d, C.c_double)))
        return p, d

    def op_mortality(self, dead):
        d = _arr(dead, np.uint8)
        assert d.size == self.N
        self._chk(self.lib.gnx_op_mortality(self.h, _ptr(d, C.c_uint8)))

    # -- spatial tiling (geonomics_amd/parallel.py) ---------------------------------
    def tile_set(self, R, C, r, c):
        self._chk(self.lib.gnx_tile_set(self.h, int(R), int(C), int(r), int(c)))
        self._n_tiles = int(R) * int(C)

    def _get_staged(self, n, with_z, with_geno):
        rec = np.zeros(n, dtype=IND_REC)
        z = np.zeros((n, self.n_traits), np.float32) if (with_z and self.n_traits) else None
        geno = np.zeros((n, 2, self.W64), np.uint64) if with_geno else None
        if n:
            self._chk(self.lib.gnx_tile_get_staged(
                self.h, rec.ctypes.data_as(C.c_void_p), _ptr(z, C.c_float),
                _ptr(geno, C.c_uint64)))
        return rec, z, geno

    def tile_export_migrants(self, with_geno):
        n = C.c_int64()
        self._chk(self.lib.gnx_tile_export_migrants(self.h, C.byref(n)))
        return self._get_staged(n.value, True, with_geno and self.L > 0)

    def tile_export_halo(self):
        n = C.c_int64()
        self._chk(self.lib.gnx_tile_export_halo(self.h, C.byref(n)))
        return self._get_staged(n.value, False, False)[0]

    def tile_import(self, rec, z=None, geno=None):
        rec = np.ascontiguousarray(rec, dtype=IND_REC)
        z = None if z is None else _arr(z, np.float32)
        geno = None if geno is None else _arr(geno, np.uint64)
        self._chk(self.lib.gnx_tile_import(self.h, C.c_int64(rec.size),
                                           rec.ctypes.data_as(C.c_void_p),
                                           _ptr(z, C.c_float), _ptr(geno, C.c_uint64)))

    def tile_import_ghosts(self, rec):
        rec = np.ascontiguousarray(rec, dtype=IND_REC)
        self._chk(self.lib.gnx_tile_import_ghosts(self.h, C.c_int64(rec.size),
                                                  rec.ctypes.data_as(C.c_void_p)))

    def tile_pairs(self, burn):
        p, b = C.c_int64(), C.c_int64()
        self._chk(self.lib.gnx_tile_pairs(self.h, int(bool(burn)), C.byref(p), C.byref(b)))
        self._n_pairs = p.value
        return p.value, b.value

    def tile_pair_info(self):
        P = self._n_pairs
        ids = np.zeros(P, np.int64)
        nb = np.zeros(P, np.int32)
        if P:
            self._chk(self.lib.gnx_tile_pair_info(self.h, _ptr(ids, C.c_int64),
                                                  _ptr(nb, C.c_int32)))
        return ids, nb

    def get_bins(self, which):
        out = np.zeros(self.lib.gnx_density_bin_count(self.h), np.int32)
        self._chk(self.lib.gnx_get_bins(self.h, int(which), _ptr(out, C.c_int32)))
        return out

    def set_bins(self, which, bins):
        b = _arr(bins, np.int32)
        assert b.size == self.lib.gnx_density_bin_count(self.h)
        self._chk(self.lib.gnx_set_bins(self.h, int(which), _ptr(b, C.c_int32)))

    def tile_offspring(self, burn, id_base, pair_goff):
        """pair_goff None: tile-major offspring ids (tile2_vt_counts / tile2_vt_bases came first)"""
        g = None if pair_goff is None else _arr(pair_goff, np.int64)
        assert g is None or g.size == self._n_pairs
        n = C.c_int64()
        self._chk(self.lib.gnx_tile_offspring(self.h, int(bool(burn)), C.c_int64(int(id_base)),
                                              _ptr(g, C.c_int64), C.byref(n)))
        self._n_req = n.value
        return n.value

    def tile_get_requests(self):
        n = self._n_req
        pid = np.zeros(n, np.int64)
        ck = np.zeros(n, np.int32)
        key = np.zeros(n, np.int32)
        st = np.zeros(n, np.uint8)
        px = np.zeros(n, np.float32)
        py = np.zeros(n, np.float32)
        if n:
            self._chk(self.lib.gnx_tile_get_requests(
                self.h, _ptr(pid, C.c_int64), _ptr(ck, C.c_int32), _ptr(key, C.c_int32),
                _ptr(st, C.c_uint8), _ptr(px, C.c_float), _ptr(py, C.c_float)))
        return pid, ck, key, st, px, py

    def tile_serve_gametes(self, pids, keys, starts):
        pids = _arr(pids, np.int64)
        keys = _arr(keys, np.int32)
        starts = _arr(starts, np.uint8)
        out = np.zeros((pids.size, self.W64), np.uint64)
        if pids.size:
            self._chk(self.lib.gnx_tile_serve_gametes(
                self.h, C.c_int64(pids.size), _ptr(pids, C.c_int64), _ptr(keys, C.c_int32),
                _ptr(starts, C.c_uint8), _ptr(out, C.c_uint64)))
        return out

    def tile_put_gametes(self, child_k, data):
        ck = _arr(child_k, np.int32)
        d = _arr(data, np.uint64)
        if ck.size:
            assert d.shape == (ck.size, self.W64)
            self._chk(self.lib.gnx_tile_put_gametes(self.h, C.c_int64(ck.size),
                                                    _ptr(ck, C.c_int32), _ptr(d, C.c_uint64)))

    def tile_finish_births(self, burn):
        self._chk(self.lib.gnx_tile_finish_births(self.h, int(bool(burn))))

    def tile_die(self, burn, with_selection, have_pairs):
        self._chk(self.lib.gnx_tile_die(self.h, int(bool(burn)), int(bool(with_selection)),
                                        int(bool(have_pairs))))

    def set_max_id(self, v):
        self._chk(self.lib.gnx_set_max_id(self.h, C.c_int64(int(v))))

    # -- device-resident transport: addresses are plain ints (device memory) -----------
    def _tile_counts(self):
        return np.zeros(max(getattr(self, '_n_tiles', 1), 1), np.int64)

    def tile_export_migrants_dev(self):
        """-> counts[R*C], (rec, z, geno) device addresses (0 when absent)"""
        cnt = self._tile_counts()
        self._chk(self.lib.gnx_tile_export_migrants_dev(self.h, _ptr(cnt, C.c_int64)))
        return cnt, self._staged_ptrs()

    def tile_export_halo_dev(self):
        cnt = self._tile_counts()
        self._chk(self.lib.gnx_tile_export_halo_dev(self.h, _ptr(cnt, C.c_int64)))
        return cnt, self._staged_ptrs()[0]

    def _staged_ptrs(self):
        r, z, g = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.lib.gnx_tile_staged_ptrs(self.h, C.byref(r), C.byref(z), C.byref(g)))
        return r.value or 0, z.value or 0, g.value or 0

    def tile_import_dev(self, n, rec, z=0, geno=0):
        self._chk(self.lib.gnx_tile_import_dev(self.h, C.c_int64(int(n)), C.c_void_p(rec),
                                               C.c_void_p(z or None), C.c_void_p(geno or None)))

    def tile_import_ghosts_dev(self, n, rec):
        self._chk(self.lib.gnx_tile_import_ghosts_dev(self.h, C.c_int64(int(n)),
                                                      C.c_void_p(rec)))

    def tile_pair_ptrs(self):
        """-> P, order keys address (int64[P], ascending), n_births address (int32[P]) or 0"""
        n, a, b = C.c_int64(), C.c_void_p(), C.c_void_p()
        self._chk(self.lib.gnx_tile_pair_ptrs(self.h, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, a.value or 0, b.value or 0

    def tile_pair_ptrs_nosync(self):
        """the same addresses without waiting for the stream (tile2: the consumer is ordered
        behind the library's stream)"""
        n, a, b = C.c_int64(), C.c_void_p(), C.c_void_p()
        self._chk(self.lib.gnx_tile_pair_ptrs_nosync(self.h, C.byref(n), C.byref(a), C.byref(b)))
        return n.value, a.value or 0, b.value or 0

    def tile_offspring_dev(self, burn, id_base, goff_ptr):
        n = C.c_int64()
        self._chk(self.lib.gnx_tile_offspring_dev(self.h, int(bool(burn)),
                                                  C.c_int64(int(id_base)),
                                                  C.c_void_p(goff_ptr or None), C.byref(n)))
        self._n_req = n.value
        return n.value

    def tile_group_requests(self):
        cnt = self._tile_counts()
        a = C.c_void_p()
        self._chk(self.lib.gnx_tile_group_requests(self.h, _ptr(cnt, C.c_int64), C.byref(a)))
        return cnt, a.value or 0

    def tile_serve_gametes_dev(self, n, req_ptr):
        a = C.c_void_p()
        self._chk(self.lib.gnx_tile_serve_gametes_dev(self.h, C.c_int64(int(n)),
                                                      C.c_void_p(req_ptr or None), C.byref(a)))
        return a.value or 0

    def tile_put_gametes_dev(self, n, data_ptr):
        self._chk(self.lib.gnx_tile_put_gametes_dev(self.h, C.c_int64(int(n)),
                                                    C.c_void_p(data_ptr or None)))

    def tile_bins_ptr(self):
        a, n = C.c_void_p(), C.c_int64()
        self._chk(self.lib.gnx_tile_bins_ptr(self.h, C.byref(a), C.byref(n)))
        return a.value, n.value

    # -- tile2: the device-driven protocol (include/gnx_hip.h) -------------------------
    def stream_ptr(self):
        a = C.c_void_p()
        self._chk(self.lib.gnx_stream_ptr(self.h, C.byref(a)))
        return a.value or 0

    def tile2_move_route(self, move):
        """-> counts int64 [2][R*C]: migrants per rank, ghosts per rank (one host wait)"""
        T = max(getattr(self, '_n_tiles', 1), 1)
        cnt = np.zeros(2 * T, np.int64)
        self._chk(self.lib.gnx_tile2_move_route(self.h, int(bool(move)), _ptr(cnt, C.c_int64)))
        return cnt.reshape(2, T)

    def tile2_route_ptrs(self):
        a, b, c, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self.lib.gnx_tile2_route_ptrs(self.h, C.byref(a), C.byref(b), C.byref(c),
                                                C.byref(d)))
        return a.value or 0, b.value or 0, c.value or 0, d.value or 0

    def tile2_import(self, n_mig, rec, z, geno, n_ghost, ghost_rec):
        self._chk(self.lib.gnx_tile2_import(
            self.h, C.c_int64(int(n_mig)), C.c_void_p(rec or None), C.c_void_p(z or None),
            C.c_void_p(geno or None), C.c_int64(int(n_ghost)), C.c_void_p(ghost_rec or None)))

    def tile2_pairs(self, burn):
        """-> P, births, gamete requests per owning rank int64 [R*C]"""
        T = max(getattr(self, '_n_tiles', 1), 1)
        cnt = np.zeros(2 + T, np.int64)
        self._chk(self.lib.gnx_tile2_pairs(self.h, int(bool(burn)), _ptr(cnt, C.c_int64)))
        self._n_pairs = int(cnt[0])
        return int(cnt[0]), int(cnt[1]), cnt[2:]

    def tile2_offspring(self, burn, id_base, goff_ptr):
        a = C.c_void_p()
        self._chk(self.lib.gnx_tile2_offspring(self.h, int(bool(burn)), C.c_int64(int(id_base)),
                                               C.c_void_p(goff_ptr or None), C.byref(a)))
        return a.value or 0

    def tile2_serve(self, n, req_ptr):
        a = C.c_void_p()
        self._chk(self.lib.gnx_tile2_serve(self.h, C.c_int64(int(n)), C.c_void_p(req_ptr or None),
                                           C.byref(a)))
        return a.value or 0

    def tile2_put(self, n, data_ptr):
        self._chk(self.lib.gnx_tile2_put(self.h, C.c_int64(int(n)), C.c_void_p(data_ptr or None)))

    def tile2_finish_births(self, burn):
        a, n = C.c_void_p(), C.c_int64()
        self._chk(self.lib.gnx_tile2_finish_births(self.h, int(bool(burn)), C.byref(a),
                                                   C.byref(n)))
        return a.value or 0, n.value

    def tile2_die(self, burn, with_selection, have_pairs):
        """-> the all-reduced (N before this step's deaths, births, deaths of the previous step)"""
        tot = np.zeros(3, np.int64)
        self._chk(self.lib.gnx_tile2_die(self.h, int(bool(burn)), int(bool(with_selection)),
                                         int(bool(have_pairs)), _ptr(tot, C.c_int64)))
        return int(tot[0]), int(tot[1]), int(tot[2])

    def last_births(self, with_gametes=True):
        """offspring of the last pop_dynamics_mate (call before pop_dynamics_die):
        child ids [B], parent ids [B,2], path keys [B,2], start homologues [B,2], xy [B,2]"""
        B = self.counts()[1]
        child = np.zeros(B, np.int64)
        par = np.zeros((B, 2), np.int64)
        keys = np.zeros((B, 2), np.int32)
        starts = np.zeros((B, 2), np.uint8)
        xy = np.zeros((B, 2), np.float32)
        if B:
            self._chk(self.lib.gnx_last_births(
                self.h, _ptr(child, C.c_int64), _ptr(par, C.c_int64),
                _ptr(keys, C.c_int32) if with_gametes else None,
                _ptr(starts, C.c_uint8) if with_gametes else None, _ptr(xy, C.c_float)))
        return child, par, keys, starts, xy

    # -- statistics ------------------------------------------------------------------
    def stats_locus_counts(self):
        c1 = np.zeros(self.L, np.int32)
        ch = np.zeros(self.L, np.int32)
        self._chk(self.lib.gnx_stats_locus_counts(self.h, _ptr(c1, C.c_int32),
                                                  _ptr(ch, C.c_int32)))
        return c1, ch

    def stats_ld(self, loci):
        loci = _arr(loci, np.int32)
        out = np.zeros((loci.size, loci.size), np.float64)
        self._chk(self.lib.gnx_stats_ld(self.h, int(loci.size), _ptr(loci, C.c_int32),
                                        _ptr(out, C.c_double)))
        return out

    def stats_ld_counts(self, loci):
        loci = _arr(loci, np.int32)
        c = np.zeros(loci.size, np.int64)
        cc = np.zeros((loci.size, loci.size), np.int64)
        self._chk(self.lib.gnx_stats_ld_counts(self.h, int(loci.size), _ptr(loci, C.c_int32),
                                               _ptr(c, C.c_int64), _ptr(cc, C.c_int64)))
        return c, cc

    # -- measurement ---------------------------------------------------------
    def profiling(self, on):
        """0 off, 1 all kernel families, 2 only the dominant kernel (crossover)"""
        self._chk(self.lib.gnx_profiling(self.h, int(on)))

    def kernel_times(self):
        out = {}
        for k, name in enumerate(KERNELS):
            ms, n, by = C.c_double(), C.c_int64(), C.c_double()
            self._chk(self.lib.gnx_kernel_time(self.h, k, C.byref(ms), C.byref(n),
                                               C.byref(by)))
            out[name] = dict(ms=ms.value, launches=n.value, bytes=by.value)
        return out


def measure_copy(nbytes=8 << 30, reps=5):
    """GB/s (read + written) of the library's own 16-byte-per-lane copy kernel on the current
    device: the box's streaming rate beside the 8 TB/s of the data sheet"""
    lib = load()
    out = C.c_double()
    if lib.gnx_measure_copy(C.c_int64(int(nbytes)), int(reps), C.byref(out)):
        raise GnxError(lib.gnx_last_error().decode())
    return out.value


def step_many(devs, burn, with_selection):
    """one time step of several independent Devices (the iterations of one model), their
    kernels side by side on the handles' own streams (gnx_step_many)"""
    if not devs:
        return
    arr = (C.c_void_p * len(devs))(*[d.h for d in devs])
    devs[0]._chk(devs[0].lib.gnx_step_many(arr, len(devs), int(bool(burn)),
                                           int(bool(with_selection))))


def comm_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 makes them, every rank joins with them)"""
    lib = load()
    buf = (C.c_uint8 * 128)()
    if lib.gnx_comm_unique_id(buf):
        raise GnxError(lib.gnx_last_error().decode())
    return bytes(buf)


def comm_probe():
    """librccl can be loaded and has every entry point the tile transport uses (raises GnxError
    otherwise): what the ranks agree on BEFORE any of them enters Device.comm_init_rccl"""
    lib = load()
    if lib.gnx_comm_probe():
        raise GnxError(lib.gnx_last_error().decode())


def comm_local_create(world):
    """meeting point of `world` tiles that live in this process (the one-GPU tests)"""
    lib = load()
    g = C.c_void_p()
    if lib.gnx_comm_local_create(int(world), C.byref(g)):
        raise GnxError(lib.gnx_last_error().decode())
    return g


def comm_local_abort(group):
    load().gnx_comm_local_abort(group)


def comm_local_destroy(group):
    load().gnx_comm_local_destroy(group)


def walk_many(devs, T, burn, with_selection):
    """T time steps of several independent Devices side by side (gnx_walk_many)"""
    if not devs:
        return
    arr = (C.c_void_p * len(devs))(*[d.h for d in devs])
    devs[0]._chk(devs[0].lib.gnx_walk_many(arr, len(devs), C.c_int64(int(T)), int(bool(burn)),
                                            int(bool(with_selection))))


def default_species_params(**kw):
    """SpeciesParams with the parameters-file template defaults
    (sim/params.py SPP_PARAMS)."""
    sp = SpeciesParams()
    sp.b = 0.2
    sp.R = 0.5
    sp.n_births_lambda = 1
    sp.n_births_fixed = 1
    sp.sexed = 0
    sp.p_male = 0.5
    sp.mating_radius = 10
    sp.mate_mode = MATE_UNIFORM
    sp.repro_age[0] = 0
    sp.repro_age[1] = 0
    sp.max_age = -1
    sp.d_min = 0
    sp.d_max = 1
    sp.window_width = -1
    sp.move = 1
    sp.dir_mu = 0
    sp.dir_kappa = 0
    sp.move_distr = DIST['lognormal']
    sp.move_p1 = 0.01
    sp.move_p2 = 0.5
    sp.disp_distr = DIST['lognormal']
    sp.disp_p1 = -1
    sp.disp_p2 = 0.05
    sp.move_surf = SURF_NONE
    sp.move_surf_layer = 0
    sp.move_surf_kappa = 12
    sp.disp_surf = SURF_NONE
    sp.disp_surf_layer = 0
    sp.disp_surf_kappa = 12
    sp.res_ratio[0] = 1
    sp.res_ratio[1] = 1
    sp.K_layer = 0
    sp.K_factor = 1
    for k, v in kw.items():
        if k == 'repro_age':
            sp.repro_age[0], sp.repro_age[1] = v
        elif k == 'res_ratio':
            sp.res_ratio[0], sp.res_ratio[1] = v
        else:
            if not hasattr(sp, k):
                raise AttributeError(k)
            setattr(sp, k, v)
    return sp
