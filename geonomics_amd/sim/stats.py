"""Statistics collection (reference geonomics/sim/stats.py).  The calculators
read the bit-packed genomes where they live: het / maf come from one
popcount pass over the genotype matrix (gnx_stats_locus_counts), ld from
homologue-major bitsets (gnx_stats_ld); nothing N x L is downloaded.

Kept from the reference: the stat names and params block ('Nt', 'ld', 'het',
'maf', 'mean_fit' with calc/freq[/mean]), freq == 0 meaning first and last
timestep (sim/stats.py:111-114), every stat forced at t == T-1 (:127-129), the
output tree GNX_mod-<name>/it-<i>/spp-<name>/mod-<name>_it-<i>_spp-<name>_
{HET.csv,MAF.csv,LD.txt,OTHER_STATS.csv} (:151-170), row-per-timestep CSVs with
a 't' column, the LD matrix stack in %0.5f text, and OTHER_STATS written once
at the last timestep with 5-decimal floats (utils/io.py:126-168).
Not kept: _plot_stat (plotting is outside the hot path).
"""
import csv
import os

import numpy as np


# -- calculators (reference sim/stats.py:353-435) ---------------------------------
def _calc_Nt(spp):
    return spp.Nt[-1]


def _calc_het(spp, mean=False):
    """fraction of heterozygous individuals per locus (sim/stats.py:394-405)"""
    N = len(spp)
    _, cnt_het = spp._locus_counts()
    with np.errstate(divide='ignore', invalid='ignore'):
        het = cnt_het / N
    if mean:
        het = np.mean(het)
    return het


def _calc_maf(spp):
    """minor-allele frequency per locus (sim/stats.py:408-421)"""
    two_N = 2 * len(spp)
    cnt1, _ = spp._locus_counts()
    with np.errstate(divide='ignore', invalid='ignore'):
        f1 = cnt1 / two_N
    return np.where(f1 > 0.5, 1 - f1, f1)


def _calc_ld(spp, plot=False, loci=None):
    """L x L matrix of r^2, NaN on the diagonal (sim/stats.py:359-390).  `loci`
    restricts the matrix to a subset (the full matrix is capped at 8192 loci
    per call by the C-ABI)."""
    if loci is None:
        loci = np.arange(spp.gen_arch.L)
    loci = np.asarray(loci, dtype=np.int32)
    comm = getattr(spp, '_comm', None)
    if comm is None:
        return spp._dev.stats_ld(loci)
    # tiles: the chromosome counts add; r^2 from the global counts, as the kernel does
    c, cc = spp._dev.stats_ld_counts(loci)
    tot = comm.allreduce_sum(np.concatenate([c, cc.ravel()]))
    c, cc = tot[:c.size].astype(np.float64), tot[c.size:].reshape(cc.shape).astype(np.float64)
    two_n = 2.0 * len(spp)
    f = c / two_n
    D = cc / two_n - (f[:, None] * f[None, :])
    with np.errstate(divide='ignore', invalid='ignore'):
        r2 = (D * D) / ((f * (1 - f))[:, None] * (f * (1 - f))[None, :])
    r2[np.arange(loci.size), np.arange(loci.size)] = np.nan
    return r2


def _calc_mean_fitness(spp):
    """mean fitness of the living individuals, NaN without traits
    (sim/stats.py:424-432)"""
    if spp.gen_arch is not None and spp.gen_arch.traits is not None:
        return float(np.mean(spp._calc_fitness()))
    return np.nan


_OTHER = 'OTHER_STATS.csv'


def _fmt(v):
    if v is None:
        return ''
    if isinstance(v, (int, np.integer)):
        return '%i' % v
    return '' if np.isnan(v) else '%0.5f' % v


class _StatsCollector:
    calc_fn_dict = {'Nt': _calc_Nt, 'ld': _calc_ld, 'het': _calc_het, 'maf': _calc_maf,
                    'mean_fit': _calc_mean_fitness}
    file_suffix_dict = {'Nt': _OTHER, 'ld': 'LD.txt', 'het': 'HET.csv', 'maf': 'MAF.csv',
                        'mean_fit': _OTHER}
    _needs_genome = ('ld', 'het', 'maf', 'mean_fit')

    def __init__(self, model_name, params):
        self.model_name = model_name
        self.T = params.model.T
        self.writer = True      # several GPUs: every rank calculates, rank 0 writes
        stats_params = params.model.stats
        self.stats = {}
        for spp_name, spp_params in params.comm.species.items():
            has_genome = 'gen_arch' in spp_params.keys()
            sub = self.stats[str(spp_name)] = {}
            for stat, sp in stats_params.items():
                if stat not in self.calc_fn_dict:
                    raise ValueError("unknown statistic '%s'; valid: %s"
                                     % (stat, ', '.join(self.calc_fn_dict)))
                if not has_genome and stat in self._needs_genome:
                    # the reference stops at the first genome-dependent stat
                    # for a species without a genome (sim/stats.py:84-86)
                    break
                if sp.calc:
                    freq = sp.freq if sp.freq != 0 else self.T - 1
                    sub[stat] = {'vals': [np.nan] * self.T, 'freq': max(int(freq), 1),
                                 'filepath': None,
                                 'other_params': {k: v for k, v in sp.items()
                                                  if k not in ('calc', 'freq')}}

    def _calc_stats(self, community, t, iteration):
        if t == 0:
            self._set_filepaths(iteration)
        for spp in community.values():
            sub = self.stats[spp.name]
            if t == self.T - 1:
                todo = [*sub]
            else:
                todo = [k for k, v in sub.items() if t % v['freq'] == 0]
            for stat in todo:
                val = self.calc_fn_dict[stat](spp, **sub[stat]['other_params'])
                vals = sub[stat]['vals']
                if t >= len(vals):       # Model.walk() past T (sim/stats.py:141-147)
                    vals.extend([np.nan] * (t + 1 - len(vals)))
                vals[t] = val
        self._write_stats(t)

    def _set_filepaths(self, iteration):
        dirname = os.path.join('GNX_mod-%s' % self.model_name, 'it-%i' % iteration)
        for spp_name, sub in self.stats.items():
            subdir = os.path.join(dirname, 'spp-%s' % spp_name)
            if self.writer:
                os.makedirs(subdir, exist_ok=True)
            for stat in sub:
                sub[stat]['filepath'] = os.path.join(
                    subdir, 'mod-%s_it-%i_spp-%s_%s' % (self.model_name, iteration, spp_name,
                                                       self.file_suffix_dict[stat]))

    @staticmethod
    def _write_row_to_csv(filepath, row, t):
        row = np.atleast_1d(np.asarray(row))
        new = not os.path.exists(filepath)
        with open(filepath, 'a', newline='') as f:
            w = csv.writer(f)
            if new:
                w.writerow(['t'] + [*range(row.size)])
            w.writerow([t] + row.tolist())

    @staticmethod
    def _write_array_to_stack(filepath, arr, t):
        with open(filepath, 'a') as f:
            np.savetxt(f, arr, fmt='%0.5f')

    def _write_other_stats(self):
        if not self.writer:
            return
        for sub in self.stats.values():
            cols = {k: v['vals'] for k, v in sub.items() if _OTHER in v['filepath']}
            if not cols:
                continue
            path = next(v['filepath'] for v in sub.values() if _OTHER in v['filepath'])
            n = max(len(v) for v in cols.values())
            # a column with any missing timestep is a float column (as a
            # DataFrame would make it); a complete integer column stays integer
            as_int = {k: all(isinstance(x, (int, np.integer)) for x in v)
                      for k, v in cols.items()}
            with open(path, 'w', newline='') as f:
                w = csv.writer(f, lineterminator='\n')
                w.writerow(['t'] + [*cols])
                for t in range(n):
                    row = [t]
                    for k, v in cols.items():
                        x = v[t] if t < len(v) else np.nan
                        if x is not None and not as_int[k]:
                            x = float(x)
                        row.append(_fmt(x))
                    w.writerow(row)

    def _write_stats(self, t):
        for sub in self.stats.values():
            for stat, sd in sub.items():
                if _OTHER in sd['filepath'] or t % sd["freq"] != 0:
                    continue
                vals = sd['vals']
                if t >= len(vals) or vals[t] is None or vals[t] is np.nan:
                    continue
                writer = (self._write_array_to_stack if stat == 'ld'
                          else self._write_row_to_csv)
                if self.writer:
                    writer(sd['filepath'], vals[t], t)
                # keep only the latest sample in memory (sim/stats.py:214-222)
                for k in range(len(vals)):
                    if k != t and vals[k] is not None and vals[k] is not np.nan:
                        vals[k] = None
        if t == self.T - 1:
            self._write_other_stats()

    def _plot_stat(self, stat, spp_name=None):
        raise NotImplementedError('plotting is outside the device hot path')
