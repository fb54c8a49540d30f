"""Host-side logic of the drop-in layer (no GPU): parameters-file format,
landscape construction, genomic architecture, burn-in statistics."""
import os

import numpy as np
import pytest

import gnx_oracle as O


def test_params_template_and_reader(tmp_path):
    import geonomics_amd as gnx
    from geonomics_amd.sim import params as P
    f = str(tmp_path / 'GNX_params_x.py')
    gnx.make_parameters_file(f, layers=2, species=[{'genomes': True, 'n_traits': 2,
                                                    'movement_surface': True}],
                             data=True, stats=True)
    p = gnx.read_parameters_file(f)
    assert p.model.name == 'GNX_params_x'
    assert p.landscape.main.dim == (20, 20)
    s = p.comm.species.spp_0
    assert s.mating.b == 0.2 and s.mating.mating_radius == 10
    assert s.movement.move_surf.vm_distr_kappa == 12
    assert s.gen_arch.n_recomb_sims == 10000 and s.gen_arch.r_distr_alpha == 0.5
    assert [*s.gen_arch.traits] == ['trait_0', 'trait_1']
    assert p.model.its.n_its == 1 and p.model.T == 100 and p.model.burn_T == 30
    # dot access and item access agree; deepcopy keeps the type
    import copy
    q = copy.deepcopy(p)
    q.model.T = 5
    assert p['model']['T'] == 100 and type(q) is P.ParametersDict
    with pytest.raises(ValueError):
        P.ParametersDict({'model': {'keys': 1}})
    # duplicate names are rejected (reference main.py:336-396)
    txt = open(f).read().replace("'lyr_1'", "'lyr_0'")
    open(f, 'w').write(txt)
    with pytest.raises(ValueError):
        gnx.read_parameters_file(f)


def test_defined_layer_needs_np_in_namespace(tmp_path):
    import geonomics_amd as gnx
    f = str(tmp_path / 'GNX_params_np.py')
    gnx.make_parameters_file(f, layers=[{'type': 'defined'}])
    p = gnx.read_parameters_file(f)
    assert p.landscape.layers.lyr_0.init.defined.rast.shape == (20, 20)


def test_landscape_construction():
    from geonomics_amd.sim import params as P
    from geonomics_amd.structs.landscape import _make_landscape
    d = P.default_params_dict(layers=[{'type': 'defined'}, {'type': 'random'}])
    d['landscape']['main']['dim'] = (12, 9)
    d['landscape']['layers']['lyr_0']['init']['defined']['rast'] = np.full((9, 12), 0.5)
    p = P.ParametersDict(d)
    np.random.seed(0)
    land = _make_landscape(None, p)
    assert land.dim == (12, 9) and land.n_lyrs == 2
    assert land[1].rast.shape == (9, 12)
    assert land[1].rast.min() >= 0 and land[1].rast.max() <= 1
    assert land._stack().shape == (2, 9, 12) and land._stack().dtype == np.float32
    assert land._get_lyr_num('lyr_1') == 1
    d['landscape']['layers']['lyr_0']['init']['defined']['rast'] = np.full((9, 12), 1.5)
    with pytest.raises(AssertionError):
        _make_landscape(None, P.ParametersDict(d))


def _spp_params(L=300, alpha=0.01, beta=None, n=40, traits=True):
    from geonomics_amd.sim import params as P
    d = P.default_params_dict(layers=[{'type': 'defined'}, {'type': 'defined'}],
                              species=[{'genomes': True, 'n_traits': 2 if traits else 0}])
    s = d['comm']['species']['spp_0']
    s['gen_arch'].update({'L': L, 'r_distr_alpha': alpha, 'r_distr_beta': beta,
                          'n_recomb_sims': n, 'use_tskit': False})
    if traits:
        s['gen_arch']['traits']['trait_0'].update({'n_loci': 5, 'layer': 'lyr_1'})
    return P.ParametersDict(d)


@pytest.mark.parametrize('alpha,beta', [(0.01, None), (0.5, None), (None, None),
                                        (2.0, 30.0)])
def test_recombination_paths(alpha, beta):
    from geonomics_amd.structs.genome import _make_genomic_architecture
    from geonomics_amd.structs.landscape import _make_landscape
    p = _spp_params(alpha=alpha, beta=beta, n=400)
    land = _make_landscape(None, p)
    rng = np.random.RandomState(1)
    ga = _make_genomic_architecture(p.comm.species.spp_0, land, rng=rng)
    rec = ga.recombinations
    assert rec._rates[0] == 0 and (rec._rates <= 0.5).all()
    paths = O.unpack_bits(rec._paths, ga.L)
    assert paths.shape == (400, ga.L)
    assert (paths[:, 0] == 0).all()               # r_0 = 0: every path starts on hom 0
    sw = (paths[:, 1:] != paths[:, :-1])
    # breakpoint density ~ per-locus rate (reference tests/validation/recomb)
    exp = rec._rates[1:].sum()
    got = sw.sum(axis=1).mean()
    assert abs(got - exp) < 5 * np.sqrt(max(exp, 0.5) / 400) + 0.02 * exp
    # padding bits are zero
    assert (O.unpack_bits(rec._paths, rec._paths.shape[1] * 64)[:, ga.L:] == 0).all()
    # traits: loci sorted, distinct across traits (no pleiotropy), monogenic alpha 0.5
    t0, t1 = ga.traits[0], ga.traits[1]
    assert (np.diff(t0.loci) > 0).all() and t0.n_loci == 5 and t1.n_loci == 1
    assert not set(t0.loci) & set(t1.loci)
    np.testing.assert_allclose(t0.alpha, [0.1, -0.1, 0.1, -0.1, 0.1])
    np.testing.assert_allclose(t1.alpha, [0.5])
    assert len(ga.neut_loci) == ga.L - 6 and (ga.p == 0.5).all()


def test_starting_counts_match_oracle():
    from geonomics_amd.structs.genome import _starting_mutation_counts
    rng = np.random.RandomState(2)
    p = rng.rand(200)
    p[:4] = [0, 1, 1e-9, 1 - 1e-9]
    np.testing.assert_array_equal(_starting_mutation_counts(77, p),
                                  O.starting_mutation_counts(77, p))


def test_adfuller_reproduces_statsmodels_known_answers():
    """The burn-in gate's ADF test pinned to the known answers of statsmodels' own
    test suite - Stata's values on the US macro series (statsmodels 0.12.2,
    tsa/tests/test_stattools.py:66-146, tsa/tests/test_adfuller_lag.py:13-43; the series are
    committed as data, tests/golden/make_adf_fixture.py) - to the 5 decimals that suite asserts."""
    from geonomics_amd.sim.burnin import adfuller
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden',
                             'adf_macrodata.npz'))
    for series, key in (('realgdp', 'constant_realgdp'), ('infl', 'constant_infl')):
        maxlag, stat, p = g[key]
        got = adfuller(g[series], maxlag=int(maxlag), autolag=None)
        assert abs(got[0] - stat) < 1.5e-5, (series, got, stat)
        assert abs(got[1] - p) < 1.5e-5, (series, got, p)
        assert got[2] == int(maxlag)
    # autolag='AIC' with the default maxlag: 16 candidate lag lengths on 203 observations, the
    # chosen one is 2, and the result is that of a fixed lag of 2
    x = np.log(g['realgdp'])
    n_cand, usedlag = (int(v) for v in g['autolag_log_realgdp'])
    assert int(np.ceil(12.0 * (x.size / 100.0) ** 0.25)) + 1 == n_cand
    auto = adfuller(x)
    assert auto[2] == usedlag
    fixed = adfuller(x, maxlag=usedlag, autolag=None)
    assert abs(auto[0] - fixed[0]) < 1e-12 and abs(auto[1] - fixed[1]) < 1e-12
    with pytest.raises(ValueError):
        adfuller(x, maxlag=150)


def test_adfuller_behaviour():
    from geonomics_amd.sim.burnin import adfuller, mackinnonp
    rng = np.random.RandomState(0)
    # MacKinnon (1994) asymptotic critical values for 'c', N=1: -2.86 (5%), -3.43 (1%)
    assert abs(mackinnonp(-2.86) - 0.05) < 0.003
    assert abs(mackinnonp(-3.43) - 0.01) < 0.002
    assert abs(mackinnonp(-2.57) - 0.10) < 0.005
    assert mackinnonp(3.0) == 1.0 and mackinnonp(-20) == 0.0
    rej_stat = rej_rw = 0
    for k in range(40):
        e = rng.randn(120)
        x = np.zeros(120)
        for t in range(1, 120):
            x[t] = 0.3 * x[t - 1] + e[t]
        rej_stat += adfuller(x + 300)[1] < 0.05
        rej_rw += adfuller(np.cumsum(e) + 300)[1] < 0.05
    assert rej_stat >= 36          # stationary AR(1): unit root rejected
    assert rej_rw <= 8             # random walk: ~5 % false rejections
    with pytest.raises(ValueError):
        adfuller(np.ones(30))
    with pytest.raises(ValueError):
        adfuller(np.arange(3.0))


def test_spatial_test_needs_enough_samples():
    from geonomics_amd.sim.burnin import spatial_test
    rng = np.random.RandomState(1)
    short = {'mean': [0.0, 0.1], 'std': [1.0, 1.1]}
    assert spatial_test(short, 30) is False
    ok = {'mean': list(rng.randn(60) * 0.01), 'std': list(1 + rng.randn(60) * 0.01)}
    assert spatial_test(ok, 30) in (True, False)


# ---- statistics collector: file formats against the reference's writers ---------------
def test_stats_writers_match_reference_text(tmp_path):
    from conftest import load_golden
    from geonomics_amd.sim.stats import _StatsCollector
    d = load_golden('g12_stats')
    row = d['row_in']
    p1 = str(tmp_path / 'a.csv')
    _StatsCollector._write_row_to_csv(p1, row, 0)
    _StatsCollector._write_row_to_csv(p1, row[::-1], 5)
    _StatsCollector._write_row_to_csv(p1, 0.125, 10)
    assert open(p1).read() == str(d['row_txt'])
    p2 = str(tmp_path / 'b.txt')
    m = d['ld'][:4, :4]
    _StatsCollector._write_array_to_stack(p2, m, 0)
    _StatsCollector._write_array_to_stack(p2, m * 2, 1)
    assert open(p2).read() == str(d['stack_txt'])


def _collector(T, stats, genome=True):
    import geonomics_amd as gnx
    from geonomics_amd.sim.params import default_params_dict
    from geonomics_amd.sim.stats import _StatsCollector
    pd = default_params_dict(1, 1, stats=True)
    pd['model']['T'] = T
    pd['model']['stats'] = stats
    if not genome:
        del pd['comm']['species']['spp_0']['gen_arch']
    return _StatsCollector('m', gnx.make_params_dict(pd, 'm'))


def test_stats_collector_schedule_and_other_stats(tmp_path, monkeypatch):
    from conftest import load_golden
    from geonomics_amd.sim import stats as S
    d = load_golden('g12_stats')
    monkeypatch.chdir(tmp_path)
    sc = _collector(4, {'Nt': {'calc': True, 'freq': 1},
                        'mean_fit': {'calc': True, 'freq': 0},
                        'het': {'calc': True, 'freq': 2, 'mean': False},
                        'ld': {'calc': False, 'freq': 1}})
    assert [*sc.stats['spp_0']] == ['Nt', 'mean_fit', 'het']
    assert sc.stats['spp_0']['mean_fit']['freq'] == 3      # freq 0 -> first and last
    Nts = [100, 101, 104, 99]
    fits = [0.9912345, None, None, 1.0]
    calls = []

    class Spp:
        name = 'spp_0'
        Nt = []
    spp = Spp()
    monkeypatch.setitem(sc.calc_fn_dict, 'mean_fit', lambda s: fits[len(s.Nt) - 1])
    monkeypatch.setitem(sc.calc_fn_dict, 'het',
                        lambda s, mean=False: calls.append(len(s.Nt) - 1) or np.array([0.5, 0.25]))
    for t in range(4):
        spp.Nt.append(Nts[t])
        sc._calc_stats({0: spp}, t, 0)
    assert calls == [0, 2, 3]                  # every 2nd step + forced at T-1
    base = tmp_path / 'GNX_mod-m' / 'it-0' / 'spp-spp_0'
    other = (base / 'mod-m_it-0_spp-spp_0_OTHER_STATS.csv').read_text()
    # mean_fit sampled at t=0 and t=3 only; Nt complete -> integer column
    assert other == 't,Nt,mean_fit\n0,100,0.99123\n1,101,\n2,104,\n3,99,1.00000\n'
    het = (base / 'mod-m_it-0_spp-spp_0_HET.csv').read_text()
    assert het == 't,0,1\n0,0.5,0.25\n2,0.5,0.25\n'      # t=3 collected, not written (ref)
    # the reference's float/int column rendering, from its own writer
    sc2 = _collector(4, {'Nt': {'calc': True, 'freq': 1}, 'mean_fit': {'calc': True, 'freq': 1}})
    sc2._set_filepaths(1)
    sc2.stats['spp_0']['Nt']['vals'] = [100, np.nan, 104, 99]
    sc2.stats['spp_0']['mean_fit']['vals'] = [0.9912345, np.nan, np.nan, 1.0]
    sc2._write_other_stats()
    path = sc2.stats['spp_0']['Nt']['filepath']
    assert open(path).read() == str(d['other_float_txt'])
    sc2.stats['spp_0'].pop('mean_fit')
    sc2.stats['spp_0']['Nt']['vals'] = [100, 101, 104, 99]
    sc2._write_other_stats()
    assert open(path).read() == str(d['other_int_txt'])


def test_stats_collector_species_without_genome():
    sc = _collector(5, {'Nt': {'calc': True, 'freq': 1}, 'het': {'calc': True, 'freq': 1},
                        'maf': {'calc': True, 'freq': 1}}, genome=False)
    assert [*sc.stats['spp_0']] == ['Nt']


# ---- change events (reference ops/change.py) -------------------------------------------
def test_layer_change_series_matches_reference():
    from conftest import load_golden
    from geonomics_amd.ops import change as CH
    d = load_golden('g13_change')

    class Lyr:
        rast = d['lyr_start']
        coord_prec = 0
    series = CH._make_lyr_series(Lyr(), d['lyr_end'], start_t=3, end_t=11, n_steps=4)
    assert [t for t, _ in series] == d['lyr_t'].tolist()
    np.testing.assert_array_equal(np.stack([r for _, r in series]), d['lyr_rasts'])
    np.testing.assert_array_equal(series[-1][1], d['lyr_end'])


def _run_dem(T, rng=None, **kw):
    from geonomics_amd.ops import change as CH

    class Spp:
        pass

    class Ch:
        base_K = None

        def _set_base_K(self, spp):
            self.base_K = spp.K
    spp, ch = Spp(), Ch()
    spp.K = np.full((2, 2), 2.0)
    fns = CH._get_dem_change_fns(spp, rng=rng, **kw)
    Ks = []
    for t in range(T):
        spp.t = t
        for tt, fn in fns:
            if tt == t:
                fn(ch, spp)
        Ks.append(spp.K[0, 0])
    return np.array(Ks), np.array([t for t, _ in fns])


def test_demographic_change_sizes_match_reference():
    from conftest import load_golden
    d = load_golden('g13_change')
    cases = {
        'mono': (20, dict(kind='monotonic', start_t=5, end_t=12, rate=0.98), None),
        'cyc': (60, dict(kind='cyclical', start_t=5, end_t=45, n_cycles=4,
                         size_range=(0.5, 1.5)), None),
        'cyc2': (40, dict(kind='cyclical', start_t=2, end_t=30, n_cycles=3, min_size=0.25,
                          max_size=2.0, increase_first=False), None),
        'cust': (30, dict(kind='custom', timesteps=[4, 9, 20], sizes=[2, 5, 0.5]), None),
        'stoch_u': (30, dict(kind='stochastic', start_t=3, end_t=23, interval=4,
                             size_range=(0.5, 1.5), distr='uniform'), 3),
        'stoch_n': (30, dict(kind='stochastic', start_t=3, end_t=23, interval=None,
                             size_range=(0.5, 1.5), distr='normal'), 3),
    }
    for tag, (T, kw, seed) in cases.items():
        rng = np.random.RandomState(seed) if seed is not None else None
        K, ts = _run_dem(T, rng=rng, **kw)
        np.testing.assert_array_equal(ts, d[tag + '_t'], err_msg=tag)
        np.testing.assert_array_equal(K, d[tag + '_K'], err_msg=tag)


def test_changer_fires_on_equal_timestep_only():
    from geonomics_amd.ops.change import _Changer
    ch = _Changer({})
    log = []
    ch._set_changes_list([(2, lambda changer: log.append('a')),
                          (2, lambda changer: log.append('b')),
                          (5, lambda changer: log.append('c'))])
    for t in range(4):
        ch._make_change(t, {})
    assert log == ['a', 'b'] and ch.next_change[0] == 5
    ch._add_change((4, lambda changer: log.append('x')))
    ch._make_change(4, {})
    ch._make_change(6, {})            # 5 was skipped: it blocks (reference :56-86)
    assert log == ['a', 'b', 'x'] and ch.next_change[0] == 5


# ---- data sampling / writers (reference sim/data.py, utils/io.py) ---------------------------
def _golden_sample():
    from conftest import load_golden
    from geonomics_amd.structs.species import Individual
    d = load_golden('g14_data')
    ids = d['ids'].tolist()
    sample = {i: Individual(i, float(d['x'][k]), float(d['y'][k]), int(d['age'][k]),
                            int(d['sex'][k]), d['e'][k].tolist(), d['z'][k].tolist(), 1.0, None)
              for k, i in enumerate(ids)}
    genotypes = {i: d['g'][k] for k, i in enumerate(ids)}
    return d, sample, genotypes


def test_vcf_and_fasta_text_match_reference():
    import re
    from geonomics_amd.sim import data as D
    d, sample, genotypes = _golden_sample()

    class GA:
        L = d['g'].shape[1]
    for key, fixed in (('vcf', False), ('vcf_fixed', True)):
        mine = re.sub(r'##fileDate=\d+', '##fileDate=DATE',
                      D._format_vcf(sample, genotypes, GA, include_fixed_sites=fixed))
        assert mine == str(d[key])
    assert 'FIX' in str(d['vcf_fixed']) and 'FIX' not in str(d['vcf'])
    # FASTA: sequences exact; headers the same numbers (the reference's text carries numpy's
    # scalar repr, 'np.float64(0.55)', under the numpy it ran with)
    ref = re.sub(r'np\.float64\(([^)]*)\)', r'\1', str(d['fasta'])).splitlines()
    mine = D._format_fasta(sample, genotypes).splitlines()
    assert len(ref) == len(mine) == 4 * len(sample)
    for a, b in zip(ref, mine):
        if a.startswith('>'):
            assert a.split(';')[0] == b.split(';')[0]            # >idx:hap
            fa = [float(v) for part in a.split(';')[1:] for v in part.split('|')]
            fb = [float(v) for part in b.split(';')[1:] for v in part.split('|')]
            assert fa == fb
        else:
            assert a == b


def test_data_schedule_transect_and_buffers():
    from conftest import load_golden
    import geonomics_amd as gnx
    from geonomics_amd.sim import data as D
    from geonomics_amd.sim.params import default_params_dict
    d = load_golden('g14_data')
    for tag, when in (('w0', 0), ('wNone', None), ('w6', 6), ('wlist', [2, 5, 19]),
                      ('wlist2', [2, 5])):
        pd = default_params_dict(1, 1, data=True)
        pd['model']['T'] = 20
        pd['model']['data']['sampling'].update({'scheme': 'all', 'when': when})
        pd['model']['data']['format'].update({'geo_vect_format': 'csv',
                                              'geo_rast_format': 'txt'})
        dc = D._DataCollector('m', gnx.make_params_dict(pd, 'm'))
        assert [dc.next_t] + list(dc.when) == d['when_' + tag].tolist(), tag
    np.testing.assert_array_equal(
        np.array(D._get_transect_points([(1.5, 2.0), (9.0, 11.0)], 5)), d['transect'])
    # the buffer is shapely's 64-gon: vertices on the circle at multiples of 2*pi/64
    R, w = 2.0, 2 * np.pi / 64
    apo = R * np.cos(w / 2)
    at = lambda r, ang: (5 + r * np.cos(ang), 7 + r * np.sin(ang))     # noqa: E731
    assert D._in_buffer(*at(0.999 * R, 3 * w), 5, 7, R)                # towards a vertex
    assert not D._in_buffer(*at(0.5 * (apo + R), 3.5 * w), 5, 7, R)    # beyond an edge midpoint
    assert D._in_buffer(*at(0.999 * apo, 3.5 * w), 5, 7, R)
    assert not D._in_buffer(*at(1.001 * R, 0.0), 5, 7, R)


def test_csv_writer_columns(tmp_path):
    import csv
    from geonomics_amd.sim import data as D
    d, sample, _ = _golden_sample()
    path = str(tmp_path / 'geo')
    D._write_csv(path, sample)
    rows = list(csv.DictReader(open(path + '.csv')))
    assert [*rows[0]] == ['idx', 'z', 'e', 'age', 'sex', 'x', 'y']     # io.py:173-193
    assert [int(r['idx']) for r in rows] == d['ids'].tolist()
    for k, r in enumerate(rows):
        assert eval(r['z']) == d['z'][k].tolist() and eval(r['e']) == d['e'][k].tolist()
        assert float(r['x']) == d['x'][k] and float(r['y']) == d['y'][k]
    with pytest.raises(ValueError):
        D._write_csv(str(tmp_path / 'geo.vcf'), sample)


# ---- spatial pedigree as tree-sequence tables -------------------------------------------------
def test_tree_tables_segments_and_genotypes(tmp_path):
    """edges per path segment with the reference's half-locus breakpoints
    (structs/genome.py:234-281), parent homologues alternating from the start homologue;
    genotypes read back through the edges equal direct bit selection (the oracle's
    crossover)"""
    from geonomics_amd.structs.pedigree import TreeTables
    rng = np.random.RandomState(3)
    L, F, n_paths = 37, 6, 9
    cross = (rng.rand(n_paths, L) < 0.15).astype(np.uint8)
    cross[:, 0] = 0
    paths = O.recomb_paths(cross)
    off, loci = O.breakpoints_from_paths(paths)
    g = rng.randint(0, 2, (F, L, 2)).astype(np.int8)
    tt = TreeTables(L, off, loci)
    ids = np.array([3, 4, 8, 11, 12, 20])
    tt.add_founders(ids, rng.rand(F, 2), g)
    geno = {int(i): g[k] for k, i in enumerate(ids)}
    next_id = 21
    for t in range(4):
        alive = np.array(sorted(geno))
        B = 5
        par = alive[rng.randint(0, alive.size, (B, 2))]
        keys = rng.randint(0, n_paths, (B, 2))
        starts = rng.randint(0, 2, (B, 2))
        child = np.arange(next_id, next_id + B)
        next_id += B
        # hand the births over in scrambled order, as the device might
        o = rng.permutation(B)
        tt.add_births(t, child[o], par[o], keys[o], starts[o], rng.rand(B, 2))
        for k in range(B):
            kid = np.zeros((L, 2), np.int8)
            for h in range(2):
                sel = paths[keys[k, h]] ^ starts[k, h]
                kid[:, h] = geno[int(par[k, h])][np.arange(L), sel]
            geno[int(child[k])] = kid
    tt.add_mutations([int(child[0])], [5], [1])
    geno[int(child[0])][5, 1] = 1
    all_ids = np.array(sorted(geno))
    np.testing.assert_array_equal(tt.genotypes_of(all_ids), np.stack([geno[int(i)] for i in all_ids]))
    tab = tt.tables()
    e, n = tab['edges'], tab['nodes']
    assert (n['time'][e['parent']] > n['time'][e['child']]).all()      # tskit's requirement
    # every non-founder node is covered exactly once over [0, L)
    for node in range(2 * F, n['time'].size):
        seg = sorted(zip(e['left'][e['child'] == node], e['right'][e['child'] == node]))
        assert seg[0][0] == 0 and seg[-1][1] == L
        assert all(a[1] == b[0] for a, b in zip(seg, seg[1:]))
        assert all(l == 0 or l % 1 == 0.5 for l, _ in seg)
    tt.write_csv(str(tmp_path / 'ped'))
    tt.write_text(str(tmp_path / 'ped'))
    head = open(tmp_path / 'ped_EDGES.csv').readline().strip()
    assert head == 'left,right,parent,child'
    assert open(tmp_path / 'ped.nodes.txt').readline().split() == [
        'is_sample', 'time', 'population', 'individual']
    assert len(open(tmp_path / 'ped.edges.txt').read().splitlines()) == 1 + e['left'].size


# ---- the reference's own parameter files (read in place; this container only) ------------------
REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference is mounted in the build container only')
@pytest.mark.parametrize('rel', ['tests/runtime/runtime_params.py',
                                 'tests/runtime/runtime_params_selection.py',
                                 'tests/validation/bottleneck/bottleneck_params.py',
                                 'tests/validation/wf/wf_params.py'])
def test_reference_parameter_files_load_unchanged(rel):
    """drop-in for the parameters-file format: the files the reference's own runtime and
    validation tests use are read as they are, and the host-side structures they describe
    (landscape, genomic architecture, change events, data / stats schedules) build"""
    import geonomics_amd as gnx
    from geonomics_amd.structs.landscape import _make_landscape
    from geonomics_amd.structs import genome as G
    from geonomics_amd.sim.stats import _StatsCollector
    from geonomics_amd.sim.data import _DataCollector
    from geonomics_amd.ops.change import _SpeciesChanger
    p = gnx.read_parameters_file(os.path.join(REF, rel))
    land = _make_landscape(None, p)
    assert tuple(land.dim) == tuple(p.landscape.main.dim) and len(land) == len(p.landscape.layers)
    for name, sp in p.comm.species.items():
        assert {'init', 'mating', 'mortality', 'movement'} <= set(sp.keys())
        if 'gen_arch' in sp:
            ga = G._make_genomic_architecture(sp, land, rng=np.random.RandomState(1))
            assert ga.L == sp.gen_arch.L
            assert ga.recombinations._rates[0] == 0
            n_traits = len(sp.gen_arch.traits) if 'traits' in sp.gen_arch else 0
            assert (ga.traits is None and n_traits == 0) or len(ga.traits) == n_traits
        if 'change' in sp:
            class Spp:
                K = np.ones((land.dim[1], land.dim[0]))
                t = 0
            ch = _SpeciesChanger(Spp(), sp.change, land=land, rng=np.random.RandomState(1))
            assert len(ch._list) > 0 and ch.next_change is not None
    if 'data' in p.model:
        dc = _DataCollector('m', p)
        assert dc._when[-1] == p.model.T - 1
    if 'stats' in p.model:
        assert _StatsCollector('m', p).stats
