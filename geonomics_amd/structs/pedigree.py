"""Spatial pedigree as tree-sequence tables (reference: the tskit.TableCollection
kept by structs/species.py:442,692-736,956-1094 when `use_tskit` is True).

The device reports each step's births (gnx_last_births: child and parent ids, the
recombination path and start homologue of both gametes, the birth position); this
module turns them into the rows the reference adds to its tables:

  individuals : one row per individual, location (x, y) at birth, metadata = its id
  nodes       : two per individual (homologue 0, 1), flags 1, time = 1 for the founders
                (the individuals alive at genome assignment) and -t for offspring born
                in main timestep t (the reference's convention: :717-726)
  edges       : per offspring homologue h, one edge per segment of the path of the
                gamete from parent h: [0, bp1 - 0.5), [bp1 - 0.5, bp2 - 0.5), ...,
                [.., L), parent node alternating between the parent's two homologues
                starting at the gamete's start homologue (structs/genome.py:234-281)
  sites, mutations : the founders' genotypes (site per locus with a derived allele in
                at least one founder; one mutation per founder node carrying it)

tskit itself is not needed to build or to write the tables (`write_text` emits
tskit's own text format, `tskit.load_text` reads it; `write_csv` the per-table CSVs of
Model.write_tskit_table_collection, sim/model.py:3449-3486).  Not kept: the coalescent
history msprime simulates for the founders (:956-1094) - founders are roots here -
and periodic simplification (the tables keep every individual that ever lived).
"""
import numpy as np


class TreeTables:
    def __init__(self, L, bp_off, bp_loci):
        self.L = int(L)
        self._bp_off = np.asarray(bp_off, dtype=np.int64)
        self._bp_loci = np.asarray(bp_loci, dtype=np.int64)
        self._ind_id = [np.zeros(0, np.int64)]       # chunks; ids ascend over the table
        self._ind_xy = [np.zeros((0, 2), np.float64)]
        self._ind_time = [np.zeros(0, np.float64)]
        self._edges = [np.zeros((0, 4), np.float64)]  # left, right, parent node, child node
        self._founder_g = None                        # int8 [n_founders, L, 2]
        self.n_founders = 0
        self._new_muts = []                           # (locus, individual id, homologue)

    # -- building ---------------------------------------------------------------------
    def _flush(self):
        for name in ('_ind_id', '_ind_xy', '_ind_time', '_edges'):
            chunks = getattr(self, name)
            if len(chunks) > 1:
                setattr(self, name, [np.concatenate(chunks)])

    @property
    def ids(self):
        self._flush()
        return self._ind_id[0]

    def add_founders(self, ids, xy, genotypes=None):
        ids = np.asarray(ids, dtype=np.int64)
        assert self.ids.size == 0, 'founders are added once, first'
        assert (np.diff(ids) > 0).all(), 'founders must come in ascending id order'
        self._ind_id.append(ids)
        self._ind_xy.append(np.asarray(xy, dtype=np.float64))
        self._ind_time.append(np.full(ids.size, 1.0))
        self.n_founders = ids.size
        self._founder_g = None if genotypes is None else np.asarray(genotypes, dtype=np.int8)

    def add_births(self, t, child, parents, keys, starts, xy):
        """one main timestep's offspring (arrays as gnx_last_births returns them)"""
        child = np.asarray(child, dtype=np.int64)
        if child.size == 0:
            return
        order = np.argsort(child, kind='stable')
        child, parents, keys = child[order], np.asarray(parents)[order], np.asarray(keys)[order]
        starts, xy = np.asarray(starts)[order], np.asarray(xy)[order]
        known = self.ids
        assert known.size == 0 or child[0] > known[-1], 'offspring ids must ascend'
        first_row = known.size
        prow = np.searchsorted(known, parents)              # parents were recorded earlier
        assert (known[np.minimum(prow, known.size - 1)] == parents).all(), (
            'a parent is missing from the individuals table')
        self._ind_id.append(child)
        self._ind_xy.append(xy.astype(np.float64))
        self._ind_time.append(np.full(child.size, -float(t)))
        # one gamete per (offspring, homologue): segments from the path's switch points
        B = child.size
        key = keys.reshape(-1).astype(np.int64)             # [2B], (k, h) -> 2k + h
        nbp = self._bp_off[key + 1] - self._bp_off[key]
        nseg = nbp + 1
        g_of = np.repeat(np.arange(2 * B), nseg)             # gamete of each segment
        seg_start = np.concatenate([[0], np.cumsum(nseg)[:-1]])
        j = np.arange(nseg.sum()) - seg_start[g_of]           # segment number inside its gamete
        bp_idx = self._bp_off[key][g_of] + j                 # index of the segment's right switch
        right = np.where(j < nbp[g_of], self._bp_loci[np.minimum(bp_idx, self._bp_loci.size - 1)]
                         - 0.5, float(self.L)) if self._bp_loci.size else np.full(
                             g_of.size, float(self.L))
        left = np.where(j > 0, self._bp_loci[np.maximum(bp_idx - 1, 0)] - 0.5, 0.0) \
            if self._bp_loci.size else np.zeros(g_of.size)
        hom = (j + starts.reshape(-1).astype(np.int64)[g_of]) % 2
        parent_node = 2 * prow.reshape(-1)[g_of] + hom
        child_node = 2 * (first_row + g_of // 2) + (g_of % 2)
        self._edges.append(np.stack([left, right, parent_node.astype(np.float64),
                                     child_node.astype(np.float64)], axis=1))

    def add_mutations(self, ind_ids, loci, homs):
        """new mutations on this step's offspring (ops/mutation.py:62-131)"""
        for i, l, h in zip(ind_ids, loci, homs):
            self._new_muts.append((int(l), int(i), int(h)))

    # -- tables ------------------------------------------------------------------------
    def tables(self):
        self._flush()
        ids, xy, tm = self._ind_id[0], self._ind_xy[0], self._ind_time[0]
        n = ids.size
        nodes = dict(flags=np.ones(2 * n, np.int64), time=np.repeat(tm, 2),
                     population=np.zeros(2 * n, np.int64),
                     individual=np.repeat(np.arange(n), 2))
        e = self._edges[0]
        edges = dict(left=e[:, 0], right=e[:, 1], parent=e[:, 2].astype(np.int64),
                     child=e[:, 3].astype(np.int64))
        individuals = dict(flags=np.zeros(n, np.int64), x=xy[:, 0], y=xy[:, 1], gnx_id=ids)
        sites = dict(position=np.zeros(0), ancestral_state=np.zeros(0, 'U1'))
        muts = dict(site=np.zeros(0, np.int64), node=np.zeros(0, np.int64),
                    derived_state=np.zeros(0, 'U1'))
        m_loc = np.array([m[0] for m in self._new_muts], dtype=np.int64)
        m_node = (2 * np.searchsorted(ids, np.array([m[1] for m in self._new_muts],
                                                     dtype=np.int64))
                  + np.array([m[2] for m in self._new_muts], dtype=np.int64))
        f_loc = f_node = np.zeros(0, np.int64)
        if self._founder_g is not None and self.n_founders:
            f, f_loc, h = np.nonzero(self._founder_g)              # [F, L, 2]
            f_node = 2 * f + h
        loc = np.concatenate([f_loc, m_loc]).astype(np.int64)
        node = np.concatenate([f_node, m_node]).astype(np.int64)
        if loc.size:
            pos, site_idx = np.unique(loc, return_inverse=True)
            sites = dict(position=pos.astype(np.float64), ancestral_state=np.full(pos.size, '0'))
            o = np.lexsort((node, site_idx))
            muts = dict(site=site_idx[o].astype(np.int64), node=node[o],
                        derived_state=np.full(o.size, '1'))
        return dict(nodes=nodes, edges=edges, individuals=individuals, sites=sites,
                    mutations=muts)

    def write_csv(self, file_basename, sep=','):
        """<basename>_{NODES,EDGES,SITES,MUTATIONS,INDIVIDUALS}.csv
        (reference sim/model.py:3449-3486)"""
        for name, tab in self.tables().items():
            cols = [*tab]
            with open('%s_%s.csv' % (file_basename, name.upper()), 'w') as f:
                f.write(sep.join(cols) + '\n')
                for row in zip(*[tab[c] for c in cols]):
                    f.write(sep.join(repr(v.item()) if hasattr(v, 'item') and not isinstance(
                        v, np.str_) else str(v) for v in row) + '\n')

    def write_text(self, file_basename):
        """tskit's text format (tskit.load_text): <basename>.{nodes,edges,sites,mutations,
        individuals}.txt"""
        t = self.tables()
        n, e, i, s, m = t['nodes'], t['edges'], t['individuals'], t['sites'], t['mutations']
        with open(file_basename + '.nodes.txt', 'w') as f:
            f.write('is_sample\ttime\tpopulation\tindividual\n')
            for a, b, c, d in zip(n['flags'], n['time'], n['population'], n['individual']):
                f.write('%i\t%r\t%i\t%i\n' % (a, float(b), c, d))
        with open(file_basename + '.edges.txt', 'w') as f:
            f.write('left\tright\tparent\tchild\n')
            for a, b, c, d in zip(e['left'], e['right'], e['parent'], e['child']):
                f.write('%r\t%r\t%i\t%i\n' % (float(a), float(b), c, d))
        with open(file_basename + '.individuals.txt', 'w') as f:
            f.write('flags\tlocation\tparents\tmetadata\n')
            for a, x, y, g in zip(i['flags'], i['x'], i['y'], i['gnx_id']):
                f.write('%i\t%r,%r\t\t%i\n' % (a, float(x), float(y), g))
        with open(file_basename + '.sites.txt', 'w') as f:
            f.write('position\tancestral_state\n')
            for a, b in zip(s['position'], s['ancestral_state']):
                f.write('%r\t%s\n' % (float(a), b))
        with open(file_basename + '.mutations.txt', 'w') as f:
            f.write('site\tnode\tderived_state\n')
            for a, b, c in zip(m['site'], m['node'], m['derived_state']):
                f.write('%i\t%i\t%s\n' % (a, b, c))

    # -- check -----------------------------------------------------------------------------
    def genotypes_of(self, ids):
        """genotypes [n, L, 2] of the listed individuals, read back through the edges to
        the founders' genotypes - what the tree sequence encodes"""
        assert self._founder_g is not None, 'founder genotypes were not kept'
        t = self.tables()
        e = t['edges']
        n_nodes = t['nodes']['time'].size
        order = np.argsort(e['child'], kind='stable')
        child_sorted = e['child'][order]
        start = np.searchsorted(child_sorted, np.arange(n_nodes + 1))
        cache = {}
        L = self.L
        own = {}
        for l, i, h in self._new_muts:
            own.setdefault(2 * int(np.searchsorted(self.ids, i)) + h, []).append(l)

        def node_geno(node):
            if node in cache:
                return cache[node]
            if node < 2 * self.n_founders:
                g = self._founder_g[node // 2, :, node % 2]
            else:
                g = np.zeros(L, np.int8)
                for k in order[start[node]:start[node + 1]]:
                    lo = int(np.ceil(e['left'][k]))
                    hi = int(np.ceil(e['right'][k])) if e['right'][k] < L else L
                    g[lo:hi] = node_geno(int(e['parent'][k]))[lo:hi]
                for l in own.get(node, ()):
                    g[l] = 1
            cache[node] = g
            return g
        import sys
        sys.setrecursionlimit(max(sys.getrecursionlimit(), 100000))
        rows = np.searchsorted(self.ids, np.asarray(ids, dtype=np.int64))
        return np.stack([np.stack([node_geno(2 * int(r)), node_geno(2 * int(r) + 1)], axis=1)
                         for r in rows])
