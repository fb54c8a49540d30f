"""A Species spread over the GPUs of a node: the same object API as
structs/species.py, one process per GPU (launch the unchanged model script with
`python -m torch.distributed.run --nproc-per-node N script.py`), every rank runs
the same host code and owns one spatial tile of the individuals.

What changes with respect to the single-GPU Species:
  * the hot path of a step (age, movement, pop dynamics) is ONE call of
    geonomics_amd.parallel.TiledStepper.step, issued from `_do_pop_dynamics`;
    `_set_age_stage` / `_do_movement` are the stepper's first phase and do nothing
    when the queue calls them;
  * counters (`Nt`, `n_births`, `n_deaths`, `len(spp)`) are global;
  * the burn-in spatial test combines the tiles' sums (the count rasters of
    disjoint tiles add);
  * starting genomes: the exact global count of 1-alleles per site
    (structs/genome.py:1124-1130) is split over the tiles by a multivariate
    hypergeometric draw from a generator seeded alike on every rank, and each tile
    places its share (the union is a uniform choice of n of the 2N homologues);
  * accessors and statistics return the GLOBAL population on every rank (tiles
    gathered); files are written by rank 0.
Mutations: every rank draws the same list from the host generator and applies those
whose offspring it owns (_mutate_tiled); the pedigree tables are kept whole on every
rank (birth records gathered each step); linkage r^2 from counts summed over the tiles
(sim/stats.py).  Panmixia (mating_radius None): every tile holds everybody's record, the
Python-driven protocol (parallel.py) - correct, not fast.
"""
import numpy as np

from .. import _native as nat
from ..parallel import DeviceShard, TiledStepper
from .species import Species
from . import genome as _genome
from ..sim import burnin as _burnin


class TiledSpecies(Species):
    def __init__(self, *args, comm=None, **kw):
        super().__init__(*args, **kw)
        self._comm = comm
        self._stepper = None
        self._glob_N = 0

    # -- construction -----------------------------------------------------------------
    def _capacities(self, cap, N0):
        """Every rank draws the whole initial population before keeping its tile, so the
        individual slots are sized for the global N0; genome rows (the big table) only
        for this tile's share, with a margin for uneven tiles (GNX_TILE_ROW_MARGIN,
        default 2: clumped populations do not spread evenly over tiles)."""
        import os
        margin = float(os.environ.get('GNX_TILE_ROW_MARGIN', '2.0'))
        rows = int(cap / self._comm.world * margin) + 1024
        if self.mating_radius is None:
            # panmixia: every tile holds everybody's record (its own individuals and all the
            # others as ghosts) - slots for the whole population; genome rows for its own share
            return cap, rows
        return max(int(N0 * 1.02) + 1024, rows), rows


    def _after_init_population(self, N):
        W, H = self._land_dim
        self._shard = DeviceShard(self._dev)
        from ..parallel import tile_grid
        R, C = tile_grid(self._comm.world)
        # (tile-major offspring ids need a tile grid the 8 x 8 virtual tiles nest in: 1, 2, 4 or
        # 8 tiles per axis; any other world keeps the (hash cell, focal id) order)
        if self._dev.id_order == 1 and (8 % R or 8 % C):
            self._dev.set_id_order(0)
        # The library's own protocol whenever it can be joined (RCCL, or the in-process transport
        # of the one-GPU tests): one or two C calls per step, the host's work on the newborns
        # (mutations, pedigree rows) between them; else the Python-driven protocol over
        # torch.distributed (gloo rehearsals), with the same offspring ids either way.
        self._stepper = TiledStepper(
            self._shard, self._comm, W, H,
            None if self.mating_radius is None else float(self.mating_radius), move=self._move,
            max_id=N - 1,
            fixed_births=int(self.n_births_distr_lambda) if self.n_births_fixed else 0)
        self._shard.export_migrants()       # every rank drew all N; keep this tile's
        self._glob_N = int(N)

    # -- counters -----------------------------------------------------------------------
    def __len__(self):
        return int(self._glob_N)

    def _check_extinct(self):
        return self._glob_N == 0

    # -- the hot path -------------------------------------------------------------------
    def _set_age_stage(self):
        pass

    def _do_movement(self, land=None):
        pass

    def _do_pop_dynamics(self, land=None):
        burn = not self.burned
        hook = self._after_births if (not burn and (self.mutate or self._tt is not None)) \
            else None
        n, births, deaths = self._stepper.step(burn, self.selection and self.burned,
                                               after_births=hook)
        self._glob_N = int(n)
        self.n_births.append(int(births))
        self.n_deaths.append(int(deaths))
        self.max_ind_idx = self._stepper.max_id
        if self._check_extinct():
            self.extinct = True

    def _after_births(self, first_id, n_offspring):
        if self._tt is not None:
            # every rank keeps the whole pedigree: gather the tiles' birth records
            rec = self._dev.last_births()
            parts = [np.concatenate([b.view(a.dtype).reshape((-1,) + a.shape[1:])
                                     for b in self._gather_bytes(a)]) for a in rec]
            self._tt.add_births(self.t, *parts)
        if self.mutate:
            self._mutate_tiled(first_id, n_offspring)

    def _mutate_tiled(self, first_id, n_offspring):
        """ops/mutation.py:169-206 over the step's offspring of ALL tiles: every rank
        draws the same mutations from the host generator (same seed, same call
        sequence) - how many, their kinds, the offspring by its position in the step's
        id order, the homologue, the never-mutated locus - and applies those whose
        offspring it owns.  A single-GPU run draws the same list (structs/species.py
        _do_mutation: slot k of the newborn block has id first_id + k)."""
        ga = self.gen_arch
        rng = self._rng
        n_muts = rng.binomial(n=n_offspring * ga.L, p=ga._mu_tot)
        if n_muts == 0 or not ga._mutables:
            return
        n_muts = min(n_muts, len(ga._mutables))
        kinds = ga._draw_mut_types(n_muts)
        who = first_id + rng.randint(0, n_offspring, n_muts)
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
        if self._tt is not None:
            self._tt.add_mutations(who, loci, homs)
        ids = self._dev.download(nat.F_ID)
        order = np.argsort(ids, kind='stable')
        pos = np.minimum(np.searchsorted(ids[order], who), max(ids.size - 1, 0))
        mine = (ids[order][pos] == who) if ids.size else np.zeros(n_muts, bool)
        slots = order[pos[mine]]
        if slots.size:
            self._dev.mutate(slots, np.asarray(loci)[mine], homs[mine])
            if dirty and ga.traits is not None:
                for s_, k in zip(slots, np.asarray(kinds, dtype=object)[mine]):
                    if k != 'neut':
                        self._dev.set_z_range(int(s_), 1)

    # -- burn-in ---------------------------------------------------------------------------
    def _spatial_update(self):
        # offspring may have dispersed across a tile border since the step's migration: hand
        # them over first, so that a tile's count raster is non-zero inside its own region
        # only; the integer sums then add exactly over the tiles and mean and std are the
        # single-GPU run's to the last bit (the burn-in tests compare p-values with
        # thresholds: a last-bit difference can end the burn-in a step earlier or later)
        self._stepper._migrate()
        a, b = self._dev.spatial_diff_sums()
        cells = float(self._land_dim[0] * self._land_dim[1])
        tot = self._comm.allreduce_sum(np.array([a, b]))
        mean = tot[0] / cells
        var = tot[1] / cells - mean * mean
        self._burnin_spat_stats['mean'].append(float(mean))
        self._burnin_spat_stats['std'].append(float(np.sqrt(var)) if var > 0 else 0.0)

    def _set_genomes_and_tables(self, burn_T, T):
        ga = self.gen_arch
        n_births_tail = self.n_births[-int(burn_T):] if self.n_births else [0]
        est_tot_muts = float(np.mean(n_births_tail)) * ga.L * (ga._mu_tot or 0) * T
        _genome._check_mutation_rates(ga, est_tot_muts, burn_T, T)
        n_glob = _genome._starting_mutation_counts(len(self), ga.p).astype(np.int64)
        # homologues per tile, known to everybody
        mine = np.zeros(self._comm.world, np.int64)
        mine[self._comm.rank] = 2 * int(self._dev.N)
        homs = self._comm.allreduce_sum(mine)
        # sequential hypergeometric splits = one multivariate hypergeometric draw per
        # site; the generator is seeded alike on every rank
        gen = np.random.default_rng([self._seed & 0x7fffffff, 0x67656e6f])
        left_n, left_h = n_glob.copy(), int(homs.sum())
        share = None
        for r in range(self._comm.world):
            h_r = int(homs[r])
            if r == self._comm.world - 1 or left_h - h_r == 0:
                k = left_n.copy()
            elif h_r == 0:
                k = np.zeros_like(left_n)
            else:
                k = gen.hypergeometric(h_r, left_h - h_r, left_n)
            if r == self._comm.rank:
                share = k
            left_n = left_n - k
            left_h -= h_r
        self._dev.assign_genomes(share.astype(np.int32))
        self._shard.has_genomes = True
        self._genomes_assigned = True
        self._start_pedigree()

    # -- iterations (Model snapshots) ---------------------------------------------------
    def _snapshot(self):
        snap = super()._snapshot()
        snap['glob_N'] = self._glob_N
        snap['stepper_max_id'] = self._stepper.max_id
        snap['has_genomes'] = self._shard.has_genomes
        return snap

    def _restore(self, snap):
        super()._restore(snap)
        self._glob_N = snap['glob_N']
        self._stepper.max_id = snap['stepper_max_id']
        self._shard.set_max_id(self._stepper.max_id)
        self._shard.has_genomes = snap['has_genomes']

    # -- global views --------------------------------------------------------------------
    def _field(self, field):
        arr = np.ascontiguousarray(self._dev.download(field))
        parts = self._gather_bytes(arr)
        tail_first = arr.ndim == 2          # e / z planes: [k][N]
        out = []
        for b in parts:
            a = b.view(arr.dtype)
            out.append(a.reshape(arr.shape[0], -1) if tail_first else a)
        return np.concatenate(out, axis=1 if tail_first else 0)

    def _gather_bytes(self, arr):
        """every rank's array (as bytes) on every rank"""
        raw = np.ascontiguousarray(arr).view(np.uint8).ravel()
        pad = (-raw.size) % 8
        buf = np.concatenate([raw, np.zeros(pad, np.uint8)]).view(np.int64)
        sizes = self._comm.allgather_i64(np.array([raw.size], np.int64))
        parts = self._comm.allgather_i64(buf)
        return [p.view(np.uint8)[:int(n[0])] for p, n in zip(parts, sizes)]

    def _locus_counts(self):
        c1, ch = self._dev.stats_locus_counts()
        tot = self._comm.allreduce_sum(np.concatenate([c1, ch]).astype(np.int64))
        return tot[:c1.size], tot[c1.size:]

    def _packed_genomes(self, ids):
        ids = np.asarray(ids, dtype=np.int64)
        own = self._dev.download(nat.F_ID)
        order = np.argsort(own, kind='stable')
        pos = np.searchsorted(own[order], ids)
        pos = np.minimum(pos, max(own.size - 1, 0))
        have = (own[order][pos] == ids) if own.size else np.zeros(ids.size, bool)
        g = self._dev.download_genomes(order[pos[have]]) if have.any() else \
            np.zeros((0, 2, self._dev.W64), np.uint64)
        id_parts = self._comm.allgather_i64(ids[have])
        g_parts = self._gather_bytes(g)
        W64 = self._dev.W64
        all_ids = np.concatenate(id_parts)
        all_g = np.concatenate([b.view(np.uint64).reshape(-1, 2, W64) for b in g_parts])
        o = np.argsort(all_ids, kind='stable')
        assert all_ids.size == ids.size and (all_ids[o] == ids).all(), (
            'some requested individuals are not alive')
        return all_g[o]

    def _remove_individuals(self, individs=None, n=None, n_left=None, keep_sites_tab=False,
                            check_extinct=False, verbose=False):
        """reference structs/species.py:1559-1640 over the tiles: the same individuals are
        drawn on every rank (shared generator, gathered ids), each tile removes the ones it
        owns, the counts are global"""
        given = [p is not None for p in (individs, n, n_left)]
        assert sum(given) == 1, ("One of 'individs', 'n', and 'n_left' must be provided, the "
                                 "other two must be None.")
        all_ids = np.sort(self._field(nat.F_ID))
        n_glob = all_ids.size
        if individs is None:
            if n_left is not None:
                assert 0 <= n_left <= n_glob, (
                    "'n_left' must be a number between 0 and the current size of the "
                    "population (%i)." % n_glob)
                n = n_glob - n_left
            assert isinstance(n, (int, np.integer)) and 0 <= n <= n_glob, (
                "'n' must be a non-negative int no larger than the population (%i)." % n_glob)
            individs = self._rng.choice(all_ids, n, replace=False)
        individs = np.asarray(individs, dtype=np.int64)
        assert np.isin(individs, all_ids).all(), 'some of the listed Individuals do not exist'
        mine = self._dev.download(nat.F_ID)
        dead = np.isin(mine, individs)
        self._dev.op_mortality(dead.astype(np.uint8))
        removed = int(self._comm_sum(np.array([dead.sum()], np.int64))[0])
        assert removed == individs.size
        self._glob_N = n_glob - removed
        if verbose:
            print('\n%i Individuals successfully removed.\n' % individs.size)
        if check_extinct and self._check_extinct():
            self.extinct = True

    def _comm_sum(self, a):
        return self._stepper.comm.allreduce_sum(a)

    def _calc_density(self, normalize=False, as_layer=False, set_N=False):
        """reference structs/species.py:845-882, over the whole landscape: every rank gathers
        everybody's positions (Species._field does) and evaluates the same density raster on
        its own device - an observer's call, not part of a step (the step's own density comes
        from the all-reduced bins)"""
        return Species._calc_density(self, normalize=normalize, as_layer=as_layer, set_N=set_N)
