"""Model: the driver API kept from the reference (geonomics/sim/model.py:47-1182
and the accessors at :2787-3147).  It builds the Landscape and Community, the
burn-in and main function queues, and runs them; the queue entries call the
device-backed Species methods (geonomics_amd/structs/species.py), which call
libgnxhip.so.

Kept: make_model -> Model.walk(T, mode) / Model.run() / Model.burn(); attributes
t, burn_t, it, T, burn_T, comm, land, its; the queue order _set_t, _set_comm_t,
_set_spp_t, _set_age_stage, _do_movement, _do_pop_dynamics, _set_Nt
(sim/model.py:603-667); ValueError when walking 'main' before burn-in (:1087);
exceptions inside run() are caught per iteration (:938-953).
Fixed: queue lambdas bind their own species; both
params.model.seed.num and params.model.num seed the model.
Statistics (params.model.stats), data sampling / writers (params.model.data) and
change events run from the main queue after the hot path (sim/stats.py,
sim/data.py, ops/change.py).  Out of scope here: plotting.
"""
import copy
import os
import random
import sys
import traceback

import numpy as np

from ..structs.landscape import _make_landscape
from ..structs.community import _make_community
from ..structs import genome as _genome
from .stats import _StatsCollector
from .data import (_DataCollector, _get_adhoc_sample, _format_vcf, _format_fasta, _write_csv,
                   _write_file)


import threading as _threading

# per-thread communicator override for the in-process rehearsals of a run over several ranks
_rehearsal = _threading.local()


class Model:
    def __init__(self, name, params, verbose=False, device=None):
        self.params = copy.deepcopy(params)
        m_params = self.params.model
        self.name = 'unnamed_model' if name is None else name
        self._pid = os.getpid()
        self._verbose = verbose
        self.__term_width__ = 80
        self.__tab_len__ = len('\t'.expandtabs())
        if verbose:
            print('\nMAKING MODEL...\n', flush=True)
        # seeds: params.model.seed.num (code) or params.model.num (documented)
        self.seed = None
        if 'seed' in [*m_params] and m_params.seed is not None:
            sd = m_params.seed
            self.seed = sd.num if hasattr(sd, 'keys') else sd
        if self.seed is None and m_params.get('num', None) is not None:
            self.seed = m_params.num
        self._set_seeds()
        self.burn_T = m_params.burn_T
        self.burn_t = -1
        self.T = m_params.T
        self.t = -1
        self.n_its = m_params.its.n_its
        self.its = [*range(self.n_its)][::-1]
        self.it = -1
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', '0'))
        self._device = device
        self._comm = self._make_comm()
        self._rank = 0 if self._comm is None else self._comm.rank
        self.land = self._make_landscape(self._verbose)
        self.comm = self._make_community(self._verbose)
        self._never_been_run = True
        self._data_collector = None
        self._stats_collector = None
        if 'stats' in [*m_params]:
            self._stats_collector = self._make_stats_collector()
        if 'data' in [*m_params]:
            self._data_collector = self._make_data_collector()
        self.reassign_genomes = None
        self.rand_genarch = m_params.its.rand_genarch
        self.rand_landscape = m_params.its.rand_landscape
        self.rand_comm = m_params.its.rand_comm
        self.repeat_burn = m_params.its.repeat_burn
        # snapshots replace the reference's deepcopy(comm); only taken when a
        # later iteration can need them
        self.orig_land = None if self.rand_landscape else self.land
        # a changing landscape is restored from a copy at every new iteration
        self._orig_land_copy = (copy.deepcopy(self.land)
                                if (self.land._changer is not None and not self.rand_landscape)
                                else None)
        self._orig_comm_snap = None
        if not self.rand_comm and self.n_its > 1:
            self._orig_comm_snap = self._snapshot_comm()
        self.burn_fn_queue = None
        self.main_fn_queue = None
        self._lanes_active = False
        self.iteration_log = {}
        self.iteration_times = {}       # it -> wall-clock (start, end) of its main phase

    def __str__(self):
        return ('%s\nModel name: %s\nLayers: %s\nSpecies: %s\nNumber of iterations: %i\n'
                'Number of burn-in timesteps (minimum): %i\nNumber of main timesteps: %i'
                % (str(type(self)), self.name,
                   ', '.join("%i: '%s'" % (i, l.name) for i, l in self.land.items()),
                   ', '.join("%i: '%s'" % (i, s.name) for i, s in self.comm.items()),
                   self.n_its, self.burn_T, self.T))

    __repr__ = __str__

    # -- helpers --------------------------------------------------------------
    def _get_lyr_num(self, lyr_id):
        return self.land._get_lyr_num(lyr_id)

    def _get_spp_num(self, spp_id):
        if isinstance(spp_id, int):
            assert spp_id in self.comm.keys(), (
                'A Species with numeric spp_id %s does not exist.' % str(spp_id))
            return spp_id
        if isinstance(spp_id, str):
            nums = [k for k, spp in self.comm.items() if spp.name == spp_id]
            assert len(nums) == 1, ("Expected to find a single Species with a name "
                                    "matching the name provided (%s). Instead found %i."
                                    % (spp_id, len(nums)))
            return nums[0]
        raise ValueError("The Species identifier must be either a str or an int. "
                         "Instead, a %s was provided." % str(type(spp_id)))

    def _get_trt_num(self, spp, trt_id):
        if isinstance(trt_id, int) or trt_id is None:
            return trt_id
        nums = [k for k, trt in spp.gen_arch.traits.items() if trt.name == trt_id]
        assert len(nums) == 1
        return nums[0]

    def _set_seeds(self):
        """reference sim/model.py:364-366; the device streams are keyed by the
        same seed (one 64-bit Philox key)."""
        self._rng = np.random.RandomState(self.seed if self.seed is not None else None)
        if self.seed is not None:
            random.seed(self.seed)
            np.random.seed(self.seed)
            self._dev_seed = int(self.seed)
        else:
            self._dev_seed = int(np.random.randint(0, 2 ** 31 - 1))

    def _set_it(self):
        # (a lane of a concurrent run has already taken its iteration off the shared list)
        nxt = self.__dict__.pop('_next_it', None)
        self.it = self.its.pop() if nxt is None else nxt

    def _set_t(self):
        self.t += 1

    def _reset_t(self):
        self.t = -1

    def _set_burn_t(self):
        self.burn_t += 1

    def _reset_burn_t(self):
        self.burn_t = -1

    def _set_reassign_genomes(self):
        self.reassign_genomes = bool(np.any([spp.gen_arch is not None
                                             for spp in self.comm.values()]))

    def _make_comm(self):
        """Several GPUs: when the script runs under torch.distributed
        (WORLD_SIZE > 1; `python -m torch.distributed.run --nproc-per-node N script.py`)
        every rank builds the same Model and each Species is tiled over the ranks
        (structs/tiled.py).  Backend: RCCL ("nccl") unless GNX_DIST_BACKEND says
        otherwise (CPU rehearsals use "gloo")."""
        # (rehearsals on a one-GPU box: the ranks are threads of one process, each building its
        # Model with the communicator it was handed - tests/_local_comm.py - instead of a
        # torch.distributed group; RCCL refuses two ranks on one device)
        injected = getattr(_rehearsal, 'comm', None)
        if injected is not None:
            if injected.rank != 0:
                self._verbose = False
            return injected
        if int(os.environ.get('WORLD_SIZE', '1')) <= 1:
            return None
        import torch
        import torch.distributed as dist
        from ..parallel import Comm
        if not dist.is_initialized():
            backend = os.environ.get('GNX_DIST_BACKEND', 'nccl')
            if backend == 'nccl':
                torch.cuda.set_device(self._device)
                dist.init_process_group('nccl', device_id=torch.device('cuda', self._device))
            else:
                dist.init_process_group(backend)
        comm = Comm(dist)
        if self.seed is None:            # one device seed for all ranks
            self._dev_seed = int(comm.allgather_i64(np.array([self._dev_seed]))[0][0])
            self._rng = np.random.RandomState(self._dev_seed)
        # every rank prints nothing but rank 0; files are written by rank 0
        if comm.rank != 0:
            self._verbose = False
        return comm

    def _make_landscape(self, verbose=False):
        return _make_landscape(mod=self, params=self.params, verbose=verbose)

    def _make_community(self, verbose=False):
        comm = _make_community(self.land, self.params, burn=True, verbose=verbose,
                               seed=self._dev_seed, device=self._device, rng=self._rng,
                               comm=self._comm)
        return comm

    def _make_land_change(self):
        """reference sim/model.py:397-398, plus the device mirrors of the changed layers"""
        self.land._make_change(t=self.t, verbose=self._verbose)
        self._sync_changed_layers()

    def _sync_changed_layers(self):
        for lyr_num in sorted(self.land._changed_lyrs):
            for spp in self.comm.values():
                spp._dev.upload_layer(lyr_num, self.land[lyr_num].rast)
        self.land._changed_lyrs.clear()

    def _make_data_collector(self):
        """reference sim/model.py:515-523"""
        dc = _DataCollector(self.name, self.params, rng=self._rng)
        dc.writer = self._rank == 0
        return dc

    def _write_data(self):
        """reference sim/model.py:1185-1186"""
        self._data_collector._write_data(self.comm, self.land, self.it)

    def remove_individuals(self, spp=0, n=None, n_left=None, individs=None):
        """remove n random individuals, all but n_left, or the listed ids
        (reference sim/model.py:3179-3225)"""
        spp = self.comm[self._get_spp_num(spp)]
        spp._remove_individuals(individs=individs, n=n, n_left=n_left,
                                verbose=self._rank == 0)

    def write_tskit_table_collection(self, file_basename, spp=0, sep=','):
        """the spatial pedigree as <basename>_{NODES,EDGES,SITES,MUTATIONS,INDIVIDUALS}.csv
        (reference sim/model.py:3449-3486) and, beside them, tskit's text format
        (<basename>.nodes.txt ..., readable with tskit.load_text)"""
        spp = self.comm[self._get_spp_num(spp)]
        if spp._tt is None:
            raise ValueError("no pedigree was recorded for this Species ('use_tskit' False, "
                             "not yet burned in, or the model is too large)")
        spp._tt.write_csv(file_basename, sep=sep)
        spp._tt.write_text(file_basename)

    def get_tree_sequence(self, spp=0):
        """tskit.TreeSequence of the recorded pedigree (needs tskit on this machine;
        reference sim/model.py get_tree_sequence)"""
        import io
        import tempfile
        try:
            import tskit
        except ImportError:
            raise ImportError('get_tree_sequence needs the tskit package; '
                              'write_tskit_table_collection writes the tables without it')
        spp = self.comm[self._get_spp_num(spp)]
        with tempfile.TemporaryDirectory() as d:
            base = os.path.join(d, 'ts')
            spp._tt.write_text(base)
            args = {k: io.StringIO(open('%s.%s.txt' % (base, k)).read())
                    for k in ('nodes', 'edges', 'sites', 'mutations', 'individuals')}
            return tskit.load_text(sequence_length=spp.gen_arch.L, strict=False, **args)

    def write_gendata(self, filepath, spp=0, n=None, include_fixed_sites=True):
        """VCF / FASTA (by extension) of all or n random individuals
        (reference sim/model.py:3342-3396)"""
        extension = filepath.split('.')[-1].lower()
        assert extension in ['vcf', 'fasta'], ('Must provide valid file extension. Valid '
                                               'extensions include ".vcf" and ".fasta".')
        spp = self.comm[self._get_spp_num(spp)]
        sample = _get_adhoc_sample(spp, n, rng=self._rng)
        gts = spp._get_genotypes(individs=[*sample], as_dict=True)
        if extension == 'vcf':
            text = _format_vcf(sample, gts, spp.gen_arch,
                               include_fixed_sites=include_fixed_sites)
        else:
            text = _format_fasta(sample, gts)
        _write_file(filepath, text)

    def write_geodata(self, filepath, spp=0, n=None):
        """CSV of all or n random individuals (reference sim/model.py:3399-3446;
        shapefile / GeoJSON need geopandas and are not written by this build)"""
        extension = filepath.split('.')[-1].lower()
        assert extension in ['csv', 'shp', 'json'], (
            'Must provide valid file extension. Valid extensions include ".csv", ".shp", '
            'and ".json".')
        if extension != 'csv':
            raise NotImplementedError('shapefile / GeoJSON output needs geopandas')
        spp = self.comm[self._get_spp_num(spp)]
        _write_csv(filepath, _get_adhoc_sample(spp, n, rng=self._rng))

    def _make_stats_collector(self):
        """reference sim/model.py:527-535"""
        sc = _StatsCollector(self.name, self.params)
        sc.writer = self._rank == 0
        return sc

    def calc_stats(self):
        """reference sim/model.py:1190-1191"""
        self._stats_collector._calc_stats(self.comm, self.t, self.it)

    def _snapshot_comm(self):
        return {k: spp._snapshot() for k, spp in self.comm.items()}

    def _restore_comm(self, snap):
        for k, spp in self.comm.items():
            spp._restore(snap[k])
        self.comm.burned = bool(np.all([spp.burned for spp in self.comm.values()]))

    def _reset_community(self, rand_comm=True):
        """reference sim/model.py:457-512"""
        if not rand_comm and self._orig_comm_snap is not None:
            if self._verbose:
                print('Copying the original community for iteration %i...\n\n' % self.it,
                      flush=True)
            self._restore_comm(self._orig_comm_snap)
            if self.rand_genarch:
                for spp in self.comm.values():
                    if spp.gen_arch is not None:
                        spp.gen_arch = _genome._make_genomic_architecture(
                            spp_params=self.params['comm']['species'][spp.name],
                            land=self.land, rng=self._rng)
                        spp._upload_gen_arch()
                        if not self.repeat_burn and spp.burned:
                            spp._set_genomes_and_tables(self.burn_T, self.T)
        else:
            if self._verbose:
                print('Creating new community for iteration %i...\n\n' % self.it,
                      flush=True)
            for spp in self.comm.values():
                spp._dev.close()
            self.comm = self._make_community(self._verbose)

    def _reset(self, rand_landscape=None, rand_comm=None, rand_genarch=None,
               repeat_burn=None):
        """reference sim/model.py:540-593"""
        rand_landscape = self.rand_landscape if rand_landscape is None else rand_landscape
        rand_comm = self.rand_comm if rand_comm is None else rand_comm
        repeat_burn = self.repeat_burn if repeat_burn is None else repeat_burn
        if not self._never_been_run:
            if rand_landscape:
                self.land = self._make_landscape()
            elif self._orig_land_copy is not None:
                self.land = copy.deepcopy(self._orig_land_copy)
                for spp in self.comm.values():
                    spp._land_ref = self.land
                    spp._dev.upload_rasters(self.land._stack())
                    spp._set_K(self.land)
            self._reset_community(rand_comm)
        else:
            self._never_been_run = False
        self._reset_t()
        if self._stats_collector is not None:
            self._stats_collector = self._make_stats_collector()
        if self._data_collector is not None:
            self._data_collector = self._make_data_collector()
        if repeat_burn:
            self._reset_burn_t()
        self.comm._reset_t()
        for spp in self.comm.values():
            spp._reset_t()
        self._set_reassign_genomes()
        # (a Community made anew for this iteration needs a queue bound to ITS Species:
        # the reference's condition - repeat_burn or the first iteration - leaves the old
        # queue in place, whose entries here would call into a closed device)
        if repeat_burn or self.it <= 0 or rand_comm or self.burn_fn_queue is None:
            self.burn_fn_queue = self._make_fn_queue(burn=True)
        self.main_fn_queue = self._make_fn_queue(burn=False)

    # -- function queue (reference sim/model.py:603-667) ----------------------------
    def _make_fn_queue(self, burn=False):
        queue = []
        if burn:
            queue.append(self._set_burn_t)
        else:
            queue.append(self._set_t)
            queue.append(self.comm._set_t)
            for spp in self.comm.values():
                queue.append(spp._set_t)
        for spp in self.comm.values():
            queue.append(spp._set_age_stage)
        for spp in self.comm.values():
            if spp._move:
                queue.append(lambda spp=spp: spp._do_movement(self.land))
        for spp in self.comm.values():
            queue.append(lambda spp=spp: spp._do_pop_dynamics(self.land))
        for spp in self.comm.values():
            queue.append(spp._set_Nt)
        if not burn:
            # change events, then statistics (reference sim/model.py:644-662)
            if self.land._changer is not None:
                queue.append(self._make_land_change)
                for spp in self.comm.values():
                    queue.append(lambda spp=spp: spp._set_K(self.land))
            for spp in self.comm.values():
                if spp._changer is not None:
                    queue.append(lambda spp=spp: spp._make_change(verbose=self._verbose))
            if self._data_collector is not None:
                queue.append(self._write_data)
            if self._stats_collector is not None:
                queue.append(self.calc_stats)
        if burn:
            queue.append(self._check_comm_burned)
        return queue

    def _check_comm_burned(self):
        self.comm._check_burned(burn_T=self.burn_T, params=self.params)

    def _print_timestep_info(self, mode):
        msg = '%s:\tit=%i:\tt=%i\n' % (mode, self.it,
                                       self.burn_t if mode == 'burn' else self.t)
        for spp in self.comm.values():
            Nt = spp.Nt[-1] if spp.Nt else np.nan
            nb = spp.n_births[-1] if spp.n_births else np.nan
            nd = spp.n_deaths[-1] if spp.n_deaths else np.nan
            msg += '\tspecies: %s%sN=%s\t(births=%s\tdeaths=%s)\n' % (
                spp.name, ' ' * (30 - len(spp.name)), Nt, nb, nd)
        print(msg)
        print('\t' + '.' * (self.__term_width__ - self.__tab_len__), flush=True)

    def _do_timestep(self, mode):
        """reference sim/model.py:699-787"""
        queue = self.burn_fn_queue if mode == 'burn' else self.main_fn_queue
        for fn in queue:
            if True not in [spp.extinct for spp in self.comm.values()]:
                fn()
            else:
                break
        if self._verbose:
            self._print_timestep_info(mode)
        if mode == 'burn' and np.all([spp.burned for spp in self.comm.values()]):
            if self.reassign_genomes:
                for spp in self.comm.values():
                    if spp.gen_arch is not None:
                        if self._verbose:
                            print('\nAssigning genomes for species "%s"...\n\n' % spp.name,
                                  flush=True)
                        spp._set_genomes_and_tables(self.burn_T, self.T)
                self.reassign_genomes = False
            self.comm.burned = True
            if self._verbose:
                print('Burn-in complete.\n\n', flush=True)
        extinct = bool(np.any([spp.extinct for spp in self.comm.values()]))
        if extinct and self._verbose:
            print('XXXX     Species %s went extinct. Iteration %i aborting.\n\n' % (
                ' & '.join('"' + spp.name + '"' for spp in self.comm.values()
                           if spp.extinct), self.it), flush=True)
        return extinct

    def _iter_seed(self, it):
        """the host-side random stream of iteration it >= 1 (community / genomic
        architecture / mutation / sampling draws).  The reference draws every iteration from
        one global stream in turn (sim/model.py:364-366), which chains them; here each
        iteration starts its own stream, so iterations can run in any order - and side by
        side (GNX_CONCURRENT_ITS=K) - with the results of a sequential run."""
        return (int(self._dev_seed) * 1000003 + 7919 * int(it) + 12345) % (2 ** 32)

    def _set_next_iteration(self):
        self._set_it()
        if self.it > 0:
            self._rng.seed(self._iter_seed(self.it))
        if self._verbose:
            print('~' * self.__term_width__ + '\n\n')
            print('Setting up iteration %i...\n\n' % self.it, flush=True)
        self._reset(rand_landscape=self.rand_landscape, rand_comm=self.rand_comm,
                    rand_genarch=self.rand_genarch, repeat_burn=self.repeat_burn)

    def _do_next_iteration(self):
        """reference sim/model.py:808-858"""
        self._iteration_burn()
        self._iteration_main()

    def _iteration_burn(self):
        """set the next iteration up and burn it in (when this iteration has a burn-in of
        its own: reference sim/model.py:808-840)"""
        self._set_next_iteration()
        if self.rand_comm or (not self.rand_comm and self.repeat_burn) or self.it == 0:
            if self._verbose:
                print('Running burn-in, iteration %i...\n\n' % self.it, flush=True)
            while not np.all([spp.burned for spp in self.comm.values()]):
                if self._do_timestep(mode='burn'):
                    break
            if not self.rand_comm and not self.repeat_burn and self.n_its > 1:
                self._orig_comm_snap = self._snapshot_comm()

    def _walk_main_on_device(self, T):
        """T main timesteps without the function queue, when nothing in it needs the host
        between two steps (one Species that neither mutates nor records a pedigree; no change
        events, no data or statistics collection, no per-step printing): the queue's
        _set_t / _set_age_stage / _do_movement / _do_pop_dynamics / _set_Nt entries
        (reference sim/model.py:603-667) T times in one call into the library.  Returns the
        number of timesteps taken, or None when the queue has to be walked."""
        if os.environ.get('GNX_MODEL_WALK', '1') == '0' or self._verbose or T < 2:
            return None
        # (lanes of a concurrent run share the process: while one host thread captures a step's
        # graph, HIP refuses the other threads' plain hipMemcpy / hipMemset calls)
        if getattr(self, '_is_lane', False) and self._lanes_active:
            return None
        if len(self.comm) != 1 or self.land._changer is not None:
            return None
        if self._data_collector is not None or self._stats_collector is not None:
            return None
        spp = self.comm[0]
        if not (self.comm.burned and spp._move and spp._can_walk_on_device()):
            return None
        done = spp._walk_on_device(T)
        self.t += done
        self.comm.t += done
        return done

    def _iteration_main(self):
        """the T main timesteps of the iteration (reference sim/model.py:841-858)"""
        if self._verbose:
            print('Running main model, iteration %i...\n\n' % self.it, flush=True)
        if np.any([spp.extinct for spp in self.comm.values()]):
            if self._verbose:
                print("WARNING: At least one Species went extinct during the burn-in. "
                      "Cannot run main phase for iteration %i.\n\n" % self.it, flush=True)
            return
        import time
        t0 = time.perf_counter()
        done = self._walk_main_on_device(self.T) or 0
        for _ in range(self.T - done):
            if np.any([spp.extinct for spp in self.comm.values()]) or self._do_timestep('main'):
                break
        self.iteration_times[self.it] = (t0, time.perf_counter())
        self._log_iteration()

    def _log_iteration(self):
        """per-iteration record (shared by the lanes of a concurrent run): population
        sizes, births and deaths of every Species over the whole iteration"""
        self.iteration_log[self.it] = {
            spp.name: {'Nt': list(spp.Nt), 'n_births': list(spp.n_births),
                       'n_deaths': list(spp.n_deaths), 'extinct': bool(spp.extinct)}
            for spp in self.comm.values()}
        hook = getattr(self, '_on_iteration_end', None)
        if hook is not None:
            hook(self)

    # -- iterations side by side on one GPU ---------------------------------------------
    # The reference walks its n_its iterations one after another and notes that they could
    # be farmed out (sim/model.py:866-953, TODO at :924-925).  A model of 10^5 individuals
    # fills a tenth of an MI355X: K iterations run as K "lanes" - clones of this Model, each
    # with its own Community (device handle, three HIP streams of its own), landscape
    # copy where the landscape changes, collectors and random stream - on K host threads;
    # every call into libgnxhip.so releases the interpreter lock, so while one lane waits for
    # a count its siblings enqueue, and the lanes' kernels share the chip.
    def _concurrency(self):
        """lanes of run(): GNX_CONCURRENT_ITS, else 1.  An experiment kept behind the environment,
        not part of run()'s signature (which is the reference's): on ROCm 7.2 kernels of different
        handles barely overlap and the lanes' runtime calls contend - measured 0.4-1.1 x the
        sequential rate, tools/its_bench.py, DESIGN 4.4."""
        k = max(1, min(int(os.environ.get('GNX_CONCURRENT_ITS', '1')), len(self.its)))
        if self._comm is not None or self._device != 0:
            k = 1           # tiles: one Species already spans the GPUs
        return k

    def _spawn_lane(self, template_comm, share_comm=False):
        lane = copy.copy(self)
        lane._is_lane = True
        lane._verbose = False
        lane._never_been_run = self._never_been_run if share_comm else False
        if share_comm:
            return lane           # the first iteration runs on the Community of make_model
        # a random stream of its own (seeded per iteration, _set_next_iteration); everything
        # the lane builds draws from it
        lane._rng = np.random.RandomState(0)
        if self._orig_land_copy is not None:
            lane.land = copy.deepcopy(self._orig_land_copy)
        if self.rand_comm:
            lane.comm = {}        # _reset_community makes the iteration's Community
        else:
            # a Community of its own to restore the snapshot into; the genomic architecture
            # persists across iterations (unless rand_genarch), so it is the template's
            lane.comm = lane._make_community(False)
            for k, spp in lane.comm.items():
                src = template_comm[k]
                if src.gen_arch is not None:
                    spp.gen_arch = copy.deepcopy(src.gen_arch)
                    spp._upload_gen_arch()
        return lane

    def _lane_loop(self, first_main_only=False):
        """what run() does, on this lane, until the shared list of iterations is empty"""
        first = first_main_only
        while first or self.__dict__.get('_next_it') is not None or len(self.its) > 0:
            try:
                if first:
                    first = False
                    self._iteration_main()
                else:
                    if self.__dict__.get('_next_it') is None:    # (else: handed one by _run_concurrent)
                        try:
                            self._next_it = self.its.pop()
                        except IndexError:          # a sibling took the last iteration
                            break
                    self._iteration_burn()
                    self._iteration_main()
            except Exception as e:
                msg = ('XXXX\tAn error occurred during iteration %i, timestep %i.\n'
                       % (self.it, self.t if getattr(self.comm, 'burned', False)
                          else self.burn_t))
                print(msg)
                print('Error message:\n\t%s\n\n' % e)
                traceback.print_exc(file=sys.stdout)

    def _run_concurrent(self, k):
        import threading
        last_it = max(self.its)
        lanes = [self._spawn_lane(self.comm, share_comm=True)]
        # the burned-in Community every later iteration starts from is a product of the
        # first iteration's burn-in: that part runs before the siblings exist
        first_main_only = False
        if 0 in self.its and not self.rand_comm and not self.repeat_burn:
            lanes[0]._iteration_burn()
            self._orig_comm_snap = lanes[0]._orig_comm_snap
            self._never_been_run = False
            first_main_only = True
        elif self.its and self.its[-1] == 0:
            # iteration 0 is the one that is not reseeded: it belongs to the lane that carries the
            # model's own random stream and original Community, whichever thread runs first
            lanes[0]._next_it = self.its.pop()
        for _ in range(k - 1):
            lanes.append(self._spawn_lane(lanes[0].comm))
        for lane in lanes:
            lane._lanes_active = True
        threads = [threading.Thread(target=lane._lane_loop,
                                    args=(first_main_only and n == 0,))
                   for n, lane in enumerate(lanes)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        # the model is left as the sequential run leaves it: in the state of the last iteration
        keep = [lane for lane in lanes if lane.it == last_it]
        keep = keep[0] if keep else lanes[0]
        for lane in lanes:
            if lane is not keep:
                for spp in (lane.comm.values() if lane.comm else []):
                    if not any(spp is s for s in keep.comm.values()):
                        spp._dev.close()
        keep._lanes_active = False
        for name in ('comm', 'land', 't', 'burn_t', 'it', '_rng', '_stats_collector',
                     '_data_collector', 'reassign_genomes', '_orig_comm_snap',
                     '_orig_land_copy', 'orig_land'):
            setattr(self, name, getattr(keep, name))
        self._never_been_run = False
        self.burn_fn_queue = self._make_fn_queue(burn=True)
        self.main_fn_queue = self._make_fn_queue(burn=False)

    # -- public API ---------------------------------------------------------------------
    def run(self, verbose=False):
        """Run all iterations: burn-in then T main timesteps each
        (reference sim/model.py:866-961)."""
        self._verbose = verbose and self._rank == 0
        if self._verbose:
            print('\n\n' + '#' * self.__term_width__ + '\n\n')
            print('Running model "%s"...\n\n' % self.name, flush=True)
        k = self._concurrency()
        if k > 1:
            self._run_concurrent(k)
        while len(self.its) > 0:
            try:
                self._do_next_iteration()
            except Exception as e:
                msg = ('XXXX\tAn error occurred during iteration %i, timestep %i.\n'
                       % (self.it, self.t if self.comm.burned else self.burn_t))
                print(msg)
                print('Error message:\n\t%s\n\n' % e)
                traceback.print_exc(file=sys.stdout)
        if self._verbose:
            print('\n\nModel "%s" is complete.\n' % self.name, flush=True)
        self._verbose = False

    def walk(self, T=1, mode='main', verbose=True, animate=False):
        """Run T timesteps in 'burn' or 'main' mode
        (reference sim/model.py:966-1161)."""
        assert isinstance(T, (int, float)), "'T' must be a numeric data type."
        T = int(T)
        if mode not in ('burn', 'main'):
            raise ValueError("mode must be 'burn' or 'main'")
        if mode == 'main' and not self.comm.burned:
            raise ValueError("The Model.walk method cannot be run in 'main' mode if the "
                             "Model's Community has not yet been burned in (i.e. if "
                             "Model.comm.burned is False).")
        if animate not in (False, None):
            raise NotImplementedError('plotting is outside the GPU hot path (SURVEY 2)')
        old_verbose = self._verbose
        self._verbose = verbose and self._rank == 0
        if self._verbose:
            print('\n')
        if mode == 'main' and self.main_fn_queue is not None:
            T -= self._walk_main_on_device(T) or 0
            if np.any([spp.extinct for spp in self.comm.values()]):
                T = 0
        for _ in range(T):
            if mode == 'burn' and self.comm.burned:
                break
            if self.burn_fn_queue is None:
                if self._verbose:
                    print('No mod.burn_fn_queue was found. Running mod.reset()...\n\n',
                          flush=True)
                self._reset()
            if self._do_timestep(mode=mode):
                break
        self._verbose = old_verbose

    def burn(self):
        """reference sim/model.py:1166-1181"""
        if self.comm.burned:
            print('\nModel has already been burned in.\n')
        else:
            while not self.comm.burned:
                self.walk(10, 'burn')
                if np.any([spp.extinct for spp in self.comm.values()]):
                    break
            print('\nModel has now been burned in.\n')

    # -- accessors (reference sim/model.py:2787-3147): rows sorted by individual id -----
    def get_coords(self, spp=0, individs=None):
        spp = self.comm[self._get_spp_num(spp)]
        return spp._get_coords(individs=None if individs is None else np.sort(individs))

    def get_x(self, spp=0, individs=None):
        spp = self.comm[self._get_spp_num(spp)]
        return spp._get_x(individs=None if individs is None else np.sort(individs))

    def get_y(self, spp=0, individs=None):
        spp = self.comm[self._get_spp_num(spp)]
        return spp._get_y(individs=None if individs is None else np.sort(individs))

    def get_cells(self, spp=0, individs=None):
        spp = self.comm[self._get_spp_num(spp)]
        return spp._get_cells(individs=None if individs is None else np.sort(individs))

    def get_random_individs(self, n, spp=0):
        spp = self.comm[self._get_spp_num(spp)]
        ids = np.array([*spp])
        return self._rng.choice(ids, n, replace=False)

    def get_e(self, spp=0, lyr_num=None, individs=None):
        spp = self.comm[self._get_spp_num(spp)]
        return spp._get_e(lyr_num=lyr_num,
                          individs=None if individs is None else np.sort(individs))

    def get_z(self, spp=0, trt=None, individs=None):
        spp = self.comm[self._get_spp_num(spp)]
        trt = self._get_trt_num(spp, trt)
        return spp._get_z(trait_num=trt,
                          individs=None if individs is None else np.sort(individs))

    def get_fitness(self, spp=0, trt=None, individs=None):
        """overall fitness, or that of one trait (reference sim/model.py get_fitness)"""
        spp = self.comm[self._get_spp_num(spp)]
        if trt is None:
            return spp._get_fit(individs=None if individs is None else np.sort(individs))
        w = spp._calc_fitness(trait_num=self._get_trt_num(spp, trt))
        if individs is None:
            return w
        ids = np.array([*spp])
        return w[np.searchsorted(ids, np.sort(individs))]

    def get_genotypes(self, spp=0, loci=None, individs=None, biallelic=False):
        spp = self.comm[self._get_spp_num(spp)]
        return spp._get_genotypes(loci=loci, individs=individs, biallelic=biallelic)
