"""Genomic architecture on the host (reference: geonomics/structs/genome.py
Recombinations:47, Trait:284, GenomicArchitecture:440,
_make_genomic_architecture:870).

Everything here is one-off setup; the per-generation work (crossover,
phenotype, fitness) runs on the GPU from the tables this module uploads:
bit-packed recombination paths, trait loci / effect sizes, dominance,
deleterious loci.  Genotypes are tracked in full on the device (the
reference's `use_tskit=False` mode); with `use_tskit=True` the pedigree is
also recorded as tree-sequence tables on the host (structs/pedigree.py).
"""
import bisect
import warnings

import numpy as np

from .. import _native


class MutationRateError(Exception):
    pass


def _set_bit_range(row, lo, hi):
    if hi <= lo:
        return
    w0, w1 = lo >> 6, (hi - 1) >> 6
    full = np.uint64(0xFFFFFFFFFFFFFFFF)
    m0 = full << np.uint64(lo & 63)
    m1 = full >> np.uint64(63 - ((hi - 1) & 63))
    if w0 == w1:
        row[w0] |= m0 & m1
    else:
        row[w0] |= m0
        row[w0 + 1:w1] = full
        row[w1] |= m1


class Recombinations:
    """Per-locus recombination rates and the cache of `n` pre-drawn
    recombination paths (reference structs/genome.py:47-230).  The reference
    keeps each path as a 2L-bit bitarray 'subsetter'; here a path is L bits
    (bit l = homologue the path is on at locus l), packed into u64 words."""

    def __init__(self, L, positions, n, r_distr_alpha, r_distr_beta, recomb_rates,
                 jitter_breakpoints=False, rng=None):
        self._rng = np.random if rng is None else rng
        self._L = L
        if positions is None:
            positions = np.arange(L)
        self._positions = np.sort(np.asarray(positions))
        self._n = int(n)
        self._r_distr_alpha = r_distr_alpha
        self._r_distr_beta = r_distr_beta
        self._jitter_breakpoints = jitter_breakpoints
        if recomb_rates is not None:
            assert len(recomb_rates) == len(self._positions), (
                "Lengths of provided recombination rates and recombination "
                "positions don't match!")
            assert recomb_rates[0] == 0, (
                "The first recombination rate (i.e. index 0) must be 0.")
            self._rates = np.asarray(recomb_rates, dtype=float)
        else:
            self._rates = self._draw_recombination_rates()
        self._paths = None          # uint64 [n, W64]

    def _draw_recombination_rates(self):
        """reference structs/genome.py:164-184"""
        n = len(self._positions)
        if self._r_distr_alpha is not None and self._r_distr_beta is not None:
            rates = np.clip(self._rng.beta(a=self._r_distr_alpha, b=self._r_distr_beta,
                                           size=n), a_min=0, a_max=0.5)
        elif self._r_distr_alpha is not None:
            rates = np.ones(n) * self._r_distr_alpha
        else:
            rates = np.ones(n) * (1 / self._L)
        rates[0] = 0
        return rates

    def _set_events(self, *args, **kwargs):
        """Draw the n cached paths: c_l ~ Bernoulli(r_l), path_l = cumsum(c) mod 2
        (reference structs/genome.py:188-230)."""
        L, n = self._L, self._n
        W64 = _native.load().gnx_words_per_hom(L)
        out = np.zeros((n, W64), dtype=np.uint64)
        rates = np.zeros(L)
        rates[self._positions] = self._rates
        expected = rates.sum()
        uniform = np.allclose(rates[1:], rates[1]) if L > 1 else True
        if uniform and expected <= 32 and L > 1:
            # equal small rates: #switches ~ Binomial(L-1, r) at distinct loci; the path
            # words are the prefix parity of the switch bits (word-parallel)
            ks = self._rng.binomial(L - 1, rates[1], n)
            rows = np.repeat(np.arange(n), ks)
            bps = self._rng.randint(1, L, rows.size)
            while True:                      # redraw the (rare) repeats inside a path
                key = rows * L + bps
                _, first = np.unique(key, return_index=True)
                dup = np.ones(key.size, bool)
                dup[first] = False
                if not dup.any():
                    break
                bps[dup] = self._rng.randint(1, L, int(dup.sum()))
            T = np.zeros((n, W64), dtype=np.uint64)
            np.bitwise_xor.at(T, (rows, bps >> 6), np.uint64(1) << (bps & 63).astype(np.uint64))
            out = T.copy()
            for sh in (1, 2, 4, 8, 16, 32):
                out ^= out << np.uint64(sh)
            # parity of each word's switch count = top bit of its prefix XOR (no
            # np.bitwise_count: that needs NumPy >= 2.0)
            par = (out >> np.uint64(63)).astype(np.int64)
            carry = (np.cumsum(par, axis=1) - par) & 1
            out ^= np.where(carry == 1, np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0))
            # bits beyond L stay clear
            last = L >> 6
            if last < W64:
                out[:, last] &= (np.uint64(1) << np.uint64(L & 63)) - np.uint64(1)
                out[:, last + 1:] = 0
        else:
            chunk = max(1, int(2e7 // max(L, 1)))
            for a in range(0, n, chunk):
                m = min(chunk, n - a)
                c = self._rng.random_sample((m, L)) < rates
                bits = np.zeros((m, W64 * 64), dtype=np.uint8)
                bits[:, :L] = np.cumsum(c, axis=1) & 1
                out[a:a + m] = np.packbits(bits, axis=1, bitorder='little').view('<u8')
        self._paths = out

    def _breakpoints(self):
        """CSR list of the loci where each cached path switches homologue
        (path[l] != path[l-1]; a path starts on homologue 0)"""
        x = self._paths
        prev_top = np.concatenate([np.zeros((x.shape[0], 1), np.uint64),
                                   x[:, :-1] >> np.uint64(63)], axis=1)
        sw = x ^ ((x << np.uint64(1)) | prev_top)          # bit l set: path[l] != path[l-1]
        last = self._L >> 6
        if last < sw.shape[1]:
            sw[:, last] &= (np.uint64(1) << np.uint64(self._L & 63)) - np.uint64(1)
            sw[:, last + 1:] = 0
        r, w = np.nonzero(sw)
        rows, loci = [], []
        for b in range(64):                # at most a few set bits per non-zero word
            hit = (sw[r, w] >> np.uint64(b)) & np.uint64(1) == 1
            rows.append(r[hit])
            loci.append(w[hit] * 64 + b)
        rows, loci = np.concatenate(rows), np.concatenate(loci)
        o = np.lexsort((loci, rows))
        counts = np.bincount(rows, minlength=x.shape[0])
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        return off, loci[o].astype(np.int64)

    def _get_path_bits(self, key):
        by = self._paths[key].view(np.uint8)
        return np.unpackbits(by, bitorder='little')[:self._L]


class Trait:
    """reference structs/genome.py:284-438"""

    def __init__(self, idx, name, phi, n_loci, mu, layer, alpha_distr_mu,
                 alpha_distr_sigma, max_alpha_mag, gamma, univ_adv):
        self.idx = idx
        self.name = name
        self.phi = phi
        self.n_loci = n_loci
        self.mu = 0 if mu is None else mu
        self.lyr_num = layer
        self.alpha_distr_mu = alpha_distr_mu
        self.alpha_distr_sigma = alpha_distr_sigma
        self.max_alpha_mag = max_alpha_mag
        self.gamma = gamma
        self.univ_adv = univ_adv
        self.loci = np.int64([])
        self.loci_idxs = None
        self.alpha = np.array([])

    def _get_phi(self, spp):
        if type(self.phi) in (float, int):
            return np.array([self.phi] * len(spp))
        cells = spp._get_cells()
        return np.asarray(self.phi)[cells[:, 1], cells[:, 0]]

    def _set_loci(self, loci):
        self.loci = np.sort(np.hstack((self.loci, np.array([*loci])))).astype(np.int64)
        self.n_loci = self.loci.size

    def _add_locus(self, locus, alpha):
        k = bisect.bisect_left(self.loci, locus)
        self.loci = np.hstack((self.loci[:k], locus, self.loci[k:])).astype(np.int64)
        self.alpha = np.hstack((self.alpha[:k], alpha, self.alpha[k:]))
        self.n_loci += 1


class GenomicArchitecture:
    """reference structs/genome.py:440-810"""

    def __init__(self, dom, g_params, land, recomb_rates=None, recomb_positions=None,
                 rng=None):
        self._rng = np.random if rng is None else rng
        self.x = 2
        self.L = g_params.L
        self.p = None
        self.pleiotropy = g_params.pleiotropy
        self.dom = np.asarray(dom)
        self._use_dom = bool(np.any(self.dom))
        self.sex = g_params.sex
        # genotypes are tracked in full on the device either way; True additionally
        # records the spatial pedigree as tree-sequence tables (structs/pedigree.py)
        self.use_tskit = bool(g_params.get('use_tskit', False))
        self.tskit_simp_interval = g_params.get('tskit_simp_interval', None)
        self.mu_neut = g_params.mu_neut or 0
        self.neut_loci = np.arange(self.L)
        self.nonneut_loci = np.array([], dtype=np.int64)
        self.mu_delet = g_params.mu_delet or 0
        self.delet_alpha_distr_shape = g_params.delet_alpha_distr_shape
        self.delet_alpha_distr_scale = g_params.delet_alpha_distr_scale
        self.delet_loci = np.int64([])
        self.delet_loci_idxs = None
        self.delet_loci_s = np.array([])
        self.traits = None
        if 'traits' in [*g_params]:
            self.traits = _make_traits(g_params.traits, land)
        mus = [self.mu_neut, self.mu_delet]
        if self.traits is not None:
            mus = mus + [trt.mu for trt in self.traits.values()]
        self._mu_tot = sum(mus)
        self._mu_nonneut = self._mu_tot - self.mu_neut
        self._mutables = None
        self._planned_muts = None
        self.recombinations = Recombinations(
            self.L, recomb_positions, g_params.n_recomb_sims, g_params.r_distr_alpha,
            g_params.r_distr_beta, recomb_rates,
            g_params.get('jitter_breakpoints', False), rng=self._rng)

    def _draw_mut_types(self, num):
        """reference structs/genome.py:650-663"""
        type_dict = {'neut': self.mu_neut, 'delet': self.mu_delet}
        if self.traits is not None:
            type_dict.update({'t%i' % k: v.mu for k, v in self.traits.items()})
        types = [*type_dict]
        probs = np.array([type_dict[k] for k in types], dtype=float)
        return self._rng.choice(types, p=probs / probs.sum(), size=num, replace=True)

    def _draw_trait_alpha(self, trait_num, n=1):
        """reference structs/genome.py:666-687"""
        trt = self.traits[trait_num]
        if trt.alpha_distr_sigma == 0:
            alpha = trt.alpha_distr_mu * np.array([1 - (i % 2) * 2 for i in range(n)])
        else:
            alpha = self._rng.normal(trt.alpha_distr_mu, trt.alpha_distr_sigma, n)
            if trt.max_alpha_mag is not None:
                alpha = np.clip(alpha, -trt.max_alpha_mag, trt.max_alpha_mag)
        if trt.n_loci == 1:
            alpha = np.abs(alpha)
        return alpha

    def _draw_delet_s(self):
        return min(self._rng.gamma(self.delet_alpha_distr_shape,
                                   self.delet_alpha_distr_scale), 1)

    def _set_trait_loci(self, trait_num, mutational=False, loci=None, alpha=None):
        """reference structs/genome.py:696-748"""
        trt = self.traits[trait_num]
        n = 1 if mutational else trt.n_loci
        assert n <= self.L, ("The number of loci parameterized for trait number %i "
                             "('n_loci') is greater than the length of the genome!"
                             % trait_num)
        if loci is not None:
            loci = [loci] if not np.iterable(loci) else [*loci]
            assert len(set(loci)) == len(loci), 'Some trait loci appear repeated.'
        elif not self.pleiotropy:
            loci = [*self._rng.choice(self.neut_loci, size=n, replace=False)]
        else:
            loci = [*self._rng.choice(self.L, size=n, replace=False)]
        if alpha is not None:
            effects = np.atleast_1d(np.asarray(alpha, dtype=float))
        else:
            # the reference sorts the drawn loci and assigns the drawn effects to
            # them positionally (structs/genome.py:398-403,747-748)
            loci = sorted(loci)
            effects = self._draw_trait_alpha(trait_num, n)
        if not mutational and n == 1:
            effects = np.array([0.5])
        assert len(loci) == len(effects)
        # keep (locus, alpha) pairs together while sorting by locus
        all_loci = np.hstack((trt.loci, np.array(loci, dtype=np.int64)))
        all_alpha = np.hstack((trt.alpha, effects))
        order = np.argsort(all_loci, kind='stable')
        trt.loci = all_loci[order].astype(np.int64)
        trt.alpha = all_alpha[order]
        trt.n_loci = trt.loci.size
        self.nonneut_loci = np.array(sorted([*self.nonneut_loci] + loci), dtype=np.int64)
        self.neut_loci = np.array(sorted(set(self.neut_loci) - set(self.nonneut_loci)),
                                  dtype=np.int64)

    def _add_nonneut_locus(self, locus, trait_nums=None, delet_s=None):
        """reference structs/genome.py:753-788 (full genotypes are tracked, so no
        genotype-array index bookkeeping is needed)"""
        self.neut_loci = self.neut_loci[self.neut_loci != locus]
        k = bisect.bisect_left(self.nonneut_loci, locus)
        self.nonneut_loci = np.hstack((self.nonneut_loci[:k], locus,
                                       self.nonneut_loci[k:])).astype(np.int64)
        if trait_nums is not None and delet_s is None:
            for n in trait_nums:
                self.traits[n]._add_locus(locus, self._draw_trait_alpha(n)[0])
        elif delet_s is not None and trait_nums is None:
            j = bisect.bisect_left(self.delet_loci, locus)
            self.delet_loci = np.hstack((self.delet_loci[:j], locus,
                                         self.delet_loci[j:])).astype(np.int64)
            self.delet_loci_s = np.hstack((self.delet_loci_s[:j], delet_s,
                                           self.delet_loci_s[j:]))
        return k


def _make_traits(traits_params, land):
    """reference structs/genome.py:824-866"""
    traits = {}
    for n, (name, v) in enumerate(traits_params.items()):
        v = dict(v)
        if isinstance(v['layer'], str):
            lyr_num = [num for num, lyr in land.items() if lyr.name == v['layer']]
        else:
            lyr_num = [num for num, lyr in land.items() if lyr.idx == v['layer']]
        assert len(lyr_num) == 1, ("Expected to find a single Layer with the Layer name "
                                   "indicated for Trait %s, but instead found %i."
                                   % (name, len(lyr_num)))
        v['layer'] = lyr_num[0]
        traits[n] = Trait(n, name, **v)
    for n, trt in traits.items():
        if trt.n_loci == 1 and trt.mu != 0:
            warnings.warn("Coercing Trait %i ('%s') to a 0 mutation rate because it "
                          "is monogenic." % (n, trt.name))
            trt.mu = 0
    return traits


def _make_genomic_architecture(spp_params, land, rng=None):
    """reference structs/genome.py:870-1062"""
    rng = np.random if rng is None else rng
    g_params = spp_params.gen_arch
    gen_arch_file = None
    if g_params.get('gen_arch_file', None) is not None:
        import pandas as pd
        gen_arch_file = pd.read_csv(g_params.gen_arch_file)
        assert len(gen_arch_file) == g_params.L, (
            "The length of the custom genomic architecture file must match the "
            "genome length 'L' in the parameters file.")
    g_params['sex'] = spp_params.mating.sex
    recomb_rates = recomb_positions = None
    if gen_arch_file is not None:
        recomb_rates = gen_arch_file['r'].values
        recomb_positions = gen_arch_file['locus'].values
        dom = gen_arch_file['dom'].values
    else:
        dom = np.array([int(g_params.dom)] * g_params.L)
    ga = GenomicArchitecture(dom, g_params, land, recomb_rates, recomb_positions, rng=rng)
    if gen_arch_file is not None and ga.traits is not None:
        names = {trt.name: num for num, trt in ga.traits.items()}
        trait_col = [[names[v.strip()] for v in str(row).split(',') if v.strip() in names]
                     for row in gen_arch_file['trait']]
        alpha_col = [[float(a) for a in str(row).split(',') if a.strip() not in ('', 'nan')]
                     for row in gen_arch_file['alpha']]
        for tnum in ga.traits:
            loci, alphas = [], []
            for l, (ts, als) in enumerate(zip(trait_col, alpha_col)):
                for t, a in zip(ts, als):
                    if t == tnum:
                        loci.append(int(gen_arch_file['locus'][l]))
                        alphas.append(a)
            assert len(loci) == ga.traits[tnum].n_loci, (
                "The number of times a Trait appears in the custom genomic architecture "
                "file must equal its 'n_loci'.")
            ga.traits[tnum].n_loci = len(loci)
            ga._set_trait_loci(tnum, mutational=False, loci=loci, alpha=alphas)
    elif ga.traits is not None:
        for tnum in ga.traits:
            ga._set_trait_loci(tnum, mutational=False)
    if gen_arch_file is None:
        spf = g_params.start_p_fixed
        if spf is not None and isinstance(spf, bool):
            ga.p = np.array([0.5] * g_params.L) if spf else rng.beta(1, 1, g_params.L)
        elif spf is not None:
            assert 0 <= spf <= 1, ("If a starting allele frequency value is provided "
                                   "then it must be between 0 and 1.")
            ga.p = np.array([float(spf)] * g_params.L)
        else:
            ga.p = rng.beta(1, 1, g_params.L)
        if g_params.start_neut_zero and len(ga.neut_loci) > 0:
            ga.p[ga.neut_loci] = 0
    else:
        ga.p = gen_arch_file['p'].values.astype(float)
    ga.recombinations._set_events()
    return ga


def _starting_mutation_counts(N, p):
    """n_l = round(2N p_l) kept inside [1, 2N-1] unless p_l is exactly 0 or 1
    (reference structs/genome.py:1124-1130)."""
    n = np.array([int(round(2 * N * f, 0)) for f in p], dtype=np.int64)
    p = np.asarray(p, dtype=float)
    n = np.where((n == 2 * N) & (p < 1), n - 1, n)
    n = np.where((n == 0) & (p > 0), 1, n)
    return n.astype(np.int32)


def _check_mutation_rates(gen_arch, est_tot_muts, burn_T, T):
    """reference structs/genome.py:1066-1105"""
    n_free = gen_arch.L - len(gen_arch.nonneut_loci)
    if est_tot_muts > 0.75 * n_free:
        raise MutationRateError(
            "This species has been parameterized with too few neutral loci to "
            "accommodate the expected number of mutations. (Geonomics only uses an "
            "infinite sites model.) Please tweak some combination of the genome "
            "length, model run time, or mutation rates.")
    if len(gen_arch.neut_loci) == 0 and gen_arch._mu_tot > 0:
        warnings.warn("This species has been parameterized with non-zero mutation "
                      "rates but without any neutral loci, leaving no target for "
                      "mutations.")
        gen_arch.mu_neut = 0
        gen_arch.mu_delet = 0
        for trt in (gen_arch.traits or {}).values():
            trt.mu = 0
        gen_arch._mu_tot = 0
    elif gen_arch._mu_tot == 0:
        pass
    else:
        mutables = np.array(sorted(set(range(gen_arch.L)) - set(gen_arch.nonneut_loci)))
        gen_arch._rng.shuffle(mutables)
        gen_arch._mutables = [*mutables]
