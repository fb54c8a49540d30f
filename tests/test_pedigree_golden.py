"""The spatial pedigree's edge rows (geonomics_amd/structs/pedigree.py) against the reference's
own segment arithmetic: Recombinations._set_seg_info / _get_seg_info
(structs/genome.py:209-281) as ops/mating.py:141-148 calls them, captured in fixture G17
(tests/golden/make_golden.py g17; tskit only stores these rows, structs/species.py:731-736)."""
import numpy as np
import pytest

from conftest import load_golden
from geonomics_amd.structs.pedigree import TreeTables


def _births_from_fixture(g, tag):
    """two consecutive fixture gametes = the two homologues of one offspring"""
    keys, starts, node0 = g[tag + '_keys'], g[tag + '_starts'], g[tag + '_parent_node0']
    assert keys.size % 2 == 0
    B = keys.size // 2
    parents = (node0 // 2).reshape(B, 2)
    return B, parents, keys.reshape(B, 2), starts.reshape(B, 2)


@pytest.mark.parametrize('tag', ['sparse', 'homog', 'free'])
def test_edge_rows_equal_reference_segments(tag):
    g = load_golden('g17_pedigree_segments')
    L = int(g[tag + '_L'][0])
    tt = TreeTables(L, g[tag + '_bp_off'], g[tag + '_bp_loci'])
    n_f = 500
    tt.add_founders(np.arange(n_f), np.zeros((n_f, 2)))
    B, parents, keys, starts = _births_from_fixture(g, tag)
    child = n_f + np.arange(B)
    tt.add_births(3, child, parents, keys, starts, np.zeros((B, 2)))
    e = tt.tables()['edges']
    # edges of child node c, in table order = segment order (left to right)
    seg_n = g[tag + '_seg_n']
    off = np.concatenate([[0], np.cumsum(seg_n)])
    assert e['left'].size == off[-1]
    for q in range(keys.size):                     # gamete q -> child node 2 * (n_f + q // 2) + q % 2
        node = 2 * (n_f + q // 2) + q % 2
        rows = np.nonzero(e['child'] == node)[0]
        assert rows.size == seg_n[q]
        sl = slice(off[q], off[q + 1])
        np.testing.assert_array_equal(e['parent'][rows], g[tag + '_seg_node'][sl])
        np.testing.assert_array_equal(e['left'][rows], g[tag + '_seg_left'][sl])      # exact: k - 0.5
        np.testing.assert_array_equal(e['right'][rows], g[tag + '_seg_right'][sl])
    # every gamete covers [0, L) without gaps (the reference asserts right > left)
    assert (e['right'] > e['left']).all()
    nodes = tt.tables()['nodes']
    assert (nodes['time'][2 * n_f:] == -3.0).all() and (nodes['time'][:2 * n_f] == 1.0).all()


@pytest.mark.gpu
def test_device_births_give_reference_segments():
    """gnx_last_births -> TreeTables: the rows built from the DEVICE's births (path keys,
    start homologues, parents) are the rows the reference's segment arithmetic gives for the
    same (start homologue, key, parent nodes), and they reproduce the device's genotypes"""
    import gnx_oracle as O
    from test_gpu_parity import make_dev, native
    nat = native()
    g = load_golden('g17_pedigree_segments')
    tag = 'sparse'
    L = int(g[tag + '_L'][0])
    bp_off, bp_loci = g[tag + '_bp_off'], g[tag + '_bp_loci']
    n_paths = bp_off.size - 1
    cross = np.zeros((n_paths, L), np.uint8)
    for k in range(n_paths):
        cross[k, bp_loci[bp_off[k]:bp_off[k + 1]]] = 1
    dev = make_dev(40, 40, L=L, cap=8192, seed=9, mating_radius=3.0, K_factor=1.0)
    dev.set_recomb_paths(O.pack_bits(O.recomb_paths(cross)))
    dev.init_population(1500)
    for _ in range(3):
        dev.step(True, False)
    dev.assign_genomes(O.starting_mutation_counts(dev.N, np.full(L, 0.5)))
    ids0 = dev.download(nat.F_ID)
    o = np.argsort(ids0)
    tt = TreeTables(L, bp_off, bp_loci)
    tt.add_founders(ids0[o], np.stack([dev.download(nat.F_X)[o], dev.download(nat.F_Y)[o]], 1),
                    O.unpack_genomes(dev.download(nat.F_GENO)[o], L))
    seg = {}
    for q in range(g[tag + '_keys'].size):        # reference rows by (key, start), node ids 0 / 1
        sl = slice(int(np.sum(g[tag + '_seg_n'][:q])), int(np.sum(g[tag + '_seg_n'][:q + 1])))
        seg[(int(g[tag + '_keys'][q]), int(g[tag + '_starts'][q]))] = (
            g[tag + '_seg_node'][sl] - g[tag + '_parent_node0'][q], g[tag + '_seg_left'][sl],
            g[tag + '_seg_right'][sl])
    checked = 0
    for t in range(4):
        dev.age()
        dev.move()
        dev.pop_dynamics_mate(False)
        child, par, keys, starts, xy = dev.last_births()
        n_before = tt.ids.size
        tt.add_births(t, child, par, keys, starts, xy)
        e = tt.tables()['edges']
        known = tt.ids
        order = np.argsort(child, kind='stable')
        for r, k in enumerate(order[:200]):                     # rows of 200 offspring per step
            for hh in range(2):
                node = 2 * (n_before + r) + hh
                rows = np.nonzero(e['child'] == node)[0]
                hom, left, right = seg[(int(keys[k, hh]), int(starts[k, hh]))]
                pnode0 = 2 * int(np.searchsorted(known, par[k, hh]))
                np.testing.assert_array_equal(e['parent'][rows], pnode0 + hom)
                np.testing.assert_array_equal(e['left'][rows], left)
                np.testing.assert_array_equal(e['right'][rows], right)
                checked += 1
        dev.pop_dynamics_die(False, False)
        dev.step_index = dev.step_index + 1
    assert checked > 1000
    # and the tables encode the genotypes the device holds
    ids = dev.download(nat.F_ID)
    pick = np.sort(np.random.RandomState(0).choice(ids.size, 40, replace=False))
    got = O.unpack_genomes(dev.download_genomes(pick), L)
    np.testing.assert_array_equal(tt.genotypes_of(ids[pick]), got)
    dev.close()
