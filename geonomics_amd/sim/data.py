"""Data sampling and writers (reference geonomics/sim/data.py, utils/io.py:120-260).

The sample is chosen on the host from the ids (and, for point / transect
schemes, the coordinates) downloaded from the device; only the sampled
individuals' genomes are gathered and downloaded (gnx_download_genomes of the
listed slots), never the whole N x L/4-byte table.

Kept from the reference: parameters (`scheme` all / random / point / transect,
`n`, `points`, `transect_endpoints`, `n_transect_points`, `radius`, `when`,
`include_landscape`, `include_fixed_sites`; formats vcf / fasta, csv, txt
rasters, `nonneut_loc_format`); the schedule (`when` 0/None -> last step only,
int -> every `when` steps, list as given, the last step always added:
data.py:113-139); the output tree GNX_mod-<name>/it-<i>/spp-<name>/
mod-<name>_it-<i>_t-<t>_spp-<name>.<ext>, the `_ZERO_SAMPLE` placeholder, the
`_NONNEUTS.csv` table; VCF 4.2 text with CHROM 0, POS = locus, REF A / ALT T,
INFO SEG|FIX, phased `a|b` genotypes of the sampled individuals in ascending
id order (data.py:460-544); FASTA with one record per homologue and the header
`>idx:hap;x;y;age;sex;z|..;e|..` (data.py:427-457); CSV columns
idx,z,e,age,sex,x,y (io.py:173-193); point buffers are the 64-gon that
shapely's `Point.buffer(radius)` produces.
Not kept: shapefile / GeoJSON / GeoTIFF writers (geopandas / rasterio);
numbers in headers are printed as plain floats (the reference's text carries
whatever `str()` of numpy scalars gives under the installed numpy).
"""
import csv
import datetime
import os

import numpy as np

FILE_EXT = {'vcf': 'vcf', 'fasta': 'fasta', 'csv': 'csv', 'shapefile': 'shp',
            'geojson': 'json', 'geotiff': 'tif', 'txt': 'txt'}


def _write_file(filepath, data):
    with open(filepath, 'w') as f:
        f.write(data)


def _set_extension(filepath, ft_ext):
    """reference utils/io.py:261-288"""
    if isinstance(ft_ext, str):
        ft_ext = [ft_ext]
    ext = os.path.splitext(filepath)[1]
    if ext and ext.lower().lstrip('.') not in ft_ext:
        raise ValueError("File name already contains an extension ('%s'), but it is "
                         "incompatible with the filetype to be written (which requires one "
                         "of the following extensions: %s)." % (
                             ext, ','.join("'.%s'" % e for e in ft_ext)))
    return filepath if ext else '.'.join([filepath, ft_ext[0]])


def _get_transect_points(endpoints, n):
    x = np.linspace(endpoints[0][0], endpoints[1][0], n)
    y = np.linspace(endpoints[0][1], endpoints[1][1], n)
    return list(zip(x, y))


_NSEG = 64          # shapely Point.buffer(): 16 segments per quarter circle


def _in_buffer(px, py, cx, cy, radius):
    """inside the regular 64-gon inscribed in the circle, vertex at angle 0"""
    dx, dy = np.asarray(px) - cx, np.asarray(py) - cy
    r = np.hypot(dx, dy)
    w = 2 * np.pi / _NSEG
    phi = np.mod(np.arctan2(dy, dx), w) - w / 2
    return r * np.cos(phi) < radius * np.cos(w / 2)


def _fmt_num(v):
    return repr(float(v))


def _fmt_list(vals, sep):
    return sep.join(_fmt_num(v) for v in np.atleast_1d(vals))


def _format_fasta(sample, genotypes):
    """reference sim/data.py:427-457"""
    assert [*sample] == [*genotypes], "'sample' and 'genotypes' do not have identical orders!"
    chunks = []
    for ind, g in zip(sample.values(), genotypes.values()):
        g = np.asarray(g)
        tail = ';'.join([_fmt_num(ind.x), _fmt_num(ind.y), str(int(ind.age)),
                         str(int(ind.sex)), _fmt_list(ind.z, '|') if len(ind.z) else '',
                         _fmt_list(ind.e, '|')])
        for hap in range(2):
            seq = (g[:, hap].astype(np.uint8) + ord('0')).tobytes().decode('ascii')
            chunks.append('>%i:%i;%s\n%s\n' % (ind.idx, hap, tail, seq))
    return ''.join(chunks)


_GT = np.array(['0|0', '0|1', '1|0', '1|1'])


def _format_vcf(sample, genotypes, gen_arch, include_fixed_sites=False):
    """reference sim/data.py:460-544"""
    assert [*sample] == [*genotypes], "'sample' and 'genotypes' do not have identical orders!"
    inds = [*sample.keys()]
    now = datetime.datetime.now()
    head = ('##fileformat=VCFv4.2\n##fileDate=%d%s%s\n##source=Geonomics\n'
            % (now.year, str(now.month).zfill(2), str(now.day).zfill(2)))
    cols = ('#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n'
            % '\t'.join(str(i) for i in inds))
    samplome = np.array([np.asarray(genotypes[i]) for i in inds])     # [n, L, 2]
    tot = samplome.sum(axis=2).sum(axis=0) if len(inds) else np.zeros(gen_arch.L)
    seg = (tot > 0) & (tot < 2 * len(sample))
    loci = np.arange(gen_arch.L) if include_fixed_sites else np.nonzero(seg)[0]
    rows = []
    if len(loci):
        code = _GT[2 * samplome[:, loci, 0].astype(np.int64) + samplome[:, loci, 1]]  # [n, nl]
        for k, locus in enumerate(loci):
            rows.append('0\t%i\t.\tA\tT\t1000\tPASS\t%s\tGT\t%s\n' % (
                locus, 'SEG' if seg[locus] else 'FIX', '\t'.join(code[:, k])))
    return ''.join([head, cols] + rows)


def _write_csv(filepath, individuals):
    """idx,z,e,age,sex,x,y per sampled individual (reference utils/io.py:173-212)"""
    filepath = _set_extension(filepath, 'csv')
    with open(filepath, 'w', newline='') as f:
        w = csv.writer(f, lineterminator='\n')
        w.writerow(['idx', 'z', 'e', 'age', 'sex', 'x', 'y'])
        for ind in individuals.values():
            w.writerow([ind.idx, '[%s]' % _fmt_list(ind.z, ', ') if len(ind.z) else '[]',
                        '[%s]' % _fmt_list(ind.e, ', '), int(ind.age), int(ind.sex),
                        _fmt_num(ind.x), _fmt_num(ind.y)])


def _write_txt_array(filepath, lyr):
    np.savetxt(_set_extension(filepath, 'txt'), lyr.rast, fmt='%0.5f')


def _get_adhoc_sample(spp, n, rng=None):
    """all individuals (n None) or a random n of them, ascending ids
    (reference sim/data.py:408-424)"""
    rng = np.random if rng is None else rng
    ids = np.array([*spp], dtype=np.int64)
    if n is not None and len(ids) > n:
        ids = rng.choice(ids, size=n, replace=False)
    return spp._get_individs(np.sort(ids))


class _DataCollector:
    def __init__(self, model_name, params, rng=None):
        self.model_name = model_name
        self.T = params.model.T
        self._rng = np.random if rng is None else rng
        self.writer = True      # several GPUs: every rank samples and formats, rank 0 writes
        sp = params.model.data.sampling
        fp = params.model.data.format
        self.scheme = sp.scheme
        assert self.scheme in ['all', 'random', 'point', 'transect'], (
            "The sampling scheme provided in the parameters must be one of the following "
            "values: 'all', 'random', 'point', or 'transect'.")
        self.n = None
        if self.scheme != 'all':
            assert 'n' in sp.keys() and type(sp.n) is int, (
                "If the sampling scheme is not 'all' then the integer 'n' parameter must be "
                "defined: the number of individuals to be sampled each time.")
            self.n = sp.n
        self.pts = None
        if self.scheme == 'point':
            self.pts = sp.points
        elif self.scheme == 'transect':
            self.pts = _get_transect_points(sp.transect_endpoints, sp.n_transect_points)
        self.radius = sp.get('radius', None)
        if self.scheme in ('point', 'transect'):
            assert self.pts is not None and self.radius is not None, (
                "'point' and 'transect' sampling need points and a radius")
        self.include_landscape = bool(sp.get('include_landscape', False) is True)
        self.include_fixed_sites = bool(sp.get('include_fixed_sites', False) is True)
        when = sp.when
        if isinstance(when, list):
            assert all(n < self.T for n in when), (
                'Values provided for sampling times must be less than total model run-time.')
            when = list(when)
        else:
            assert when is None or when < self.T, (
                'Values provided for sampling times must be less than total model run-time.')
            when = [] if when in (0, None) else [*range(0, self.T, int(when))]
        if len(when) == 0 or when[-1] != self.T - 1:
            when.append(self.T - 1)
        self._when = when
        self._pos = 0
        self.gen_formats = [fp.gen_format] if isinstance(fp.gen_format, str) else list(
            fp.gen_format)
        for fmt in self.gen_formats:
            assert fmt in ('vcf', 'fasta'), "gen_format must be 'vcf' and/or 'fasta'"
        self.geo_formats = [fp.geo_vect_format]
        for fmt in self.geo_formats:
            if fmt != 'csv':
                raise NotImplementedError("geo_vect_format '%s' needs geopandas; this build "
                                          "writes 'csv'" % fmt)
        self.rast_format = None
        if self.include_landscape and 'geo_rast_format' in fp.keys():
            self.rast_format = fp.geo_rast_format
            if self.rast_format not in ('txt', None):
                raise NotImplementedError("geo_rast_format '%s' needs rasterio; this build "
                                          "writes 'txt'" % self.rast_format)
        self.nonneut_loc_format = fp.get('nonneut_loc_format', None)
        assert self.nonneut_loc_format in ['csv', None], (
            "the 'nonneut_loc_format' parameter must be either 'csv' or None.")

    @property
    def next_t(self):
        return self._when[self._pos] if self._pos < len(self._when) else None

    @property
    def when(self):
        return iter(self._when[self._pos + 1:])

    def _set_next_t(self):
        self._pos += 1

    def _make_filenames(self, iteration, spp_name):
        return [['mod-%s_it-%i_t-%i_spp-%s.%s' % (self.model_name, iteration, self.next_t,
                                                 spp_name, FILE_EXT[fmt])
                 for fmt in getattr(self, att)] for att in ('gen_formats', 'geo_formats')]

    # -- sampling (reference sim/data.py:285-321) -------------------------------------------
    def _get_random_sample(self, ids):
        if len(ids) > self.n:
            return self._rng.choice(ids, size=self.n, replace=False)
        return ids

    def _get_sample(self, spp):
        ids = np.array([*spp], dtype=np.int64)
        if self.scheme == 'all':
            chosen = ids
        elif self.scheme == 'random':
            chosen = np.asarray(self._get_random_sample(ids))
        else:
            xy = spp._get_coords()
            parts = []
            for cx, cy in self.pts:
                inside = ids[_in_buffer(xy[:, 0], xy[:, 1], cx, cy, self.radius)]
                parts.append(np.asarray(self._get_random_sample(inside)))
            chosen = np.unique(np.concatenate(parts)) if parts else ids[:0]
        return spp._get_individs(np.sort(np.unique(chosen)))

    # -- writing ---------------------------------------------------------------------------------
    def _format_gen_data(self, data_format, sample, spp):
        genotypes = spp._get_genotypes(individs=[*sample], as_dict=True)
        if data_format == 'fasta':
            return _format_fasta(sample, genotypes)
        return _format_vcf(sample, genotypes, spp.gen_arch,
                           include_fixed_sites=self.include_fixed_sites)

    def _write_nonneut_loc_file(self, spp, subdir, iteration):
        """trait names as columns, their loci down the rows (sim/data.py:347-381)"""
        import pandas as pd
        locs = {}
        if spp.gen_arch is not None and spp.gen_arch.traits is not None:
            locs = {trt.name: [*trt.loci] for trt in spp.gen_arch.traits.values()}
            nrow = max(len(v) for v in locs.values())
            for k, v in locs.items():
                if len(v) < nrow:
                    locs[k] = np.array([*v] + [np.nan] * (nrow - len(v)))
        path = os.path.join(subdir, 'mod-%s_it-%i_t-%i_spp-%s_NONNEUTS.csv' % (
            self.model_name, iteration, self.next_t, spp.name))
        pd.DataFrame.from_dict(locs).to_csv(path, index=False)

    def _write_data(self, community, land, iteration):
        if community.t != self.next_t:
            return
        dirname = os.path.join(os.getcwd(), 'GNX_mod-%s' % self.model_name, 'it-%i' % iteration)
        for spp in community.values():
            subdir = os.path.join(dirname, 'spp-%s' % spp.name)
            if spp.t == self.next_t:
                if self.writer:
                    os.makedirs(subdir, exist_ok=True)
                gen_files, geo_files = self._make_filenames(iteration, spp.name)
                sample = self._get_sample(spp)
                if len(sample) > 0:
                    if spp.gen_arch is not None:
                        for fname, fmt in zip(gen_files, self.gen_formats):
                            text = self._format_gen_data(fmt, sample, spp)
                            if self.writer:
                                _write_file(os.path.join(subdir, fname), text)
                    for fname in geo_files:
                        if self.writer:
                            _write_csv(os.path.join(subdir, fname), sample)
                elif self.writer:
                    base = os.path.splitext((gen_files + geo_files)[0])[0]
                    _write_file(os.path.join(subdir, base + '_ZERO_SAMPLE'), '')
            if self.nonneut_loc_format is not None and self.writer:
                os.makedirs(subdir, exist_ok=True)
                self._write_nonneut_loc_file(spp, subdir, iteration)
        if self.rast_format is not None and self.writer:
            os.makedirs(dirname, exist_ok=True)
            for lyr in land.values():
                _write_txt_array(os.path.join(dirname, 'mod-%s_it-%i_t-%i_lyr-%s.txt' % (
                    self.model_name, iteration, self.next_t, lyr.name)), lyr)
        self._set_next_t()
