"""TEST INFRASTRUCTURE ONLY - CPU restatement (numpy) of Geonomics' per-generation
hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file; the product (geonomics_amd/) never does.

Every function restates one reference operator (file:line cited, paths relative
to /root/reference/geonomics/) on struct-of-arrays inputs with the random draws
passed in explicitly, so the same inputs can be fed to (i) the reference itself
(tests/golden/make_golden.py, this container only), (ii) this oracle and
(iii) the HIP kernels through the C-ABI.

Pinning: tests/golden/*.npz were produced by running the REFERENCE operators
(v1.4.9, numpy 2.2.6, scipy 1.15.3) and tests/test_oracle_golden.py checks every
function here against them.  Integer/bit results are exact; float results are
compared at the tolerance stated in each test.
"""
import numpy as np

F32 = np.float32

# --------------------------------------------------------------------------
# genotype packing.  Reference: Individual.g is an L x 2 int8 array
# (structs/individual.py:103-104).  Build layout: per individual one "row" of
# 2 homologues x W64 little-endian u64 words, bit l of homologue h =
# g[l, h]  (word l >> 6, bit l & 63).
# --------------------------------------------------------------------------

def words_per_hom(L, align_words=16):
    w = (L + 63) // 64
    return ((w + align_words - 1) // align_words) * align_words


def pack_genomes(g, align_words=16):
    """g: [N, L, 2] 0/1 -> uint64 [N, 2, W]."""
    g = np.asarray(g)
    N, L, _ = g.shape
    W = words_per_hom(L, align_words)
    out = np.zeros((N, 2, W), dtype=np.uint64)
    bits = np.zeros((N, 2, W * 64), dtype=np.uint8)
    bits[:, :, :L] = np.transpose(g, (0, 2, 1)).astype(np.uint8)
    by = np.packbits(bits, axis=2, bitorder='little')
    out[:] = by.view('<u8').reshape(N, 2, W)
    return out


def unpack_genomes(p, L):
    """uint64 [N, 2, W] -> int8 [N, L, 2]."""
    p = np.ascontiguousarray(p)
    N = p.shape[0]
    by = p.view(np.uint8).reshape(N, 2, -1)
    bits = np.unpackbits(by, axis=2, bitorder='little')[:, :, :L]
    return np.transpose(bits, (0, 2, 1)).astype(np.int8)


def pack_bits(b, align_words=16):
    """b: [n, L] 0/1 -> uint64 [n, W]."""
    b = np.asarray(b)
    n, L = b.shape
    W = words_per_hom(L, align_words)
    bits = np.zeros((n, W * 64), dtype=np.uint8)
    bits[:, :L] = b
    return np.packbits(bits, axis=1, bitorder='little').view('<u8').reshape(n, W)


def unpack_bits(p, L):
    p = np.ascontiguousarray(p)
    by = p.view(np.uint8).reshape(p.shape[0], -1)
    return np.unpackbits(by, axis=1, bitorder='little')[:, :L]


# --------------------------------------------------------------------------
# A9  Recombinations (structs/genome.py:164-230)
# --------------------------------------------------------------------------

def recomb_rates(L, r_distr_alpha, r_distr_beta, beta_draws=None):
    """structs/genome.py:164-184.  beta_draws = the np.random.beta(a, b, L)
    sample when both alpha and beta are given."""
    if r_distr_alpha is not None and r_distr_beta is not None:
        rates = np.clip(np.asarray(beta_draws, dtype=np.float64), 0, 0.5)
    elif r_distr_alpha is not None:
        rates = np.ones(L) * r_distr_alpha
    else:
        rates = np.ones(L) * (1 / L)
    rates = rates.copy()
    rates[0] = 0
    return rates


def recomb_paths(crossovers):
    """structs/genome.py:194,209-221.  crossovers: [n, L] 0/1 Bernoulli(r_l)
    indicators.  path_l = (sum_{k<=l} c_k) mod 2; the reference stores the
    path as a 2L-bit 'subsetter' ('10' where path==0, '01' where path==1);
    the build stores the path bits themselves."""
    c = np.asarray(crossovers, dtype=np.int64)
    return (np.cumsum(c, axis=1) % 2).astype(np.uint8)


def subsetter_from_path(path):
    """The reference's 2L-bit mask for one path (structs/genome.py:56,220)."""
    path = np.asarray(path, dtype=np.uint8)
    out = np.empty(2 * path.size, dtype=np.uint8)
    out[0::2] = 1 - path
    out[1::2] = path
    return out


def breakpoints_from_paths(paths):
    """CSR list of loci where each path switches homologue (path[l]!=path[l-1],
    path[-1] := 0).  Returns (offsets int32 [n+1], loci int32 [nnz])."""
    paths = np.asarray(paths, dtype=np.uint8)
    prev = np.concatenate([np.zeros((paths.shape[0], 1), np.uint8),
                           paths[:, :-1]], axis=1)
    sw = paths != prev
    counts = sw.sum(axis=1)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    loci = np.nonzero(sw)[1].astype(np.int32)
    return offs, loci


# --------------------------------------------------------------------------
# A10  crossover (ops/mating.py:130-214)
# --------------------------------------------------------------------------

def reference_key_layout(n_births, recomb_keys):
    """Map the reference's flat recomb_keys list to per-offspring (k0, k1).

    ops/mating.py:206-209 slices 2*n keys per pair; :179-181 pops two keys
    from the END of the pair's slice per offspring; the first popped goes
    with pair[0] (zip order at :167)."""
    n_births = np.asarray(n_births, dtype=np.int64)
    keys = np.asarray(recomb_keys, dtype=np.int64)
    out = []
    start = 0
    for n in n_births:
        sl = list(keys[start:start + 2 * n])
        start += 2 * n
        for _ in range(n):
            k0 = sl.pop()
            k1 = sl.pop()
            out.append((k0, k1))
    return np.asarray(out, dtype=np.int32).reshape(-1, 2)


def crossover(geno, paths_packed, parent_rows, keys, start_homs):
    """geno: uint64 [R, 2, W]; paths_packed: uint64 [n, W];
    parent_rows: int [B, 2] genome rows of (pair[0], pair[1]) per offspring;
    keys: int [B, 2]; start_homs: 0/1 [B, 2].  Returns uint64 [B, 2, W].

    gamete_p[l] = g_p[l, path_{k_p}[l] XOR s_p]  (ops/mating.py:165-168);
    child[:, 0] = gamete of pair[0], child[:, 1] = gamete of pair[1] (:169).
    Word-parallel: m = path ^ (-s);  gamete = (h0 & ~m) | (h1 & m)."""
    geno = np.asarray(geno, dtype=np.uint64)
    parent_rows = np.asarray(parent_rows, dtype=np.int64)
    keys = np.asarray(keys, dtype=np.int64)
    s = np.asarray(start_homs, dtype=np.uint64)
    B = parent_rows.shape[0]
    W = geno.shape[2]
    child = np.empty((B, 2, W), dtype=np.uint64)
    ones = np.uint64(0xFFFFFFFFFFFFFFFF)
    for p in range(2):
        m = paths_packed[keys[:, p]] ^ (s[:, p, None] * ones)
        h0 = geno[parent_rows[:, p], 0]
        h1 = geno[parent_rows[:, p], 1]
        child[:, p] = (h0 & ~m) | (h1 & m)
    return child


# --------------------------------------------------------------------------
# A12  phenotype (ops/selection.py:22-48)
# --------------------------------------------------------------------------

def phenotype(g_unpacked, loci, alpha, dom=None):
    """g_unpacked: [N, L, 2]; loci: int [n_loci]; alpha: [n_loci];
    dom: [L] 0/1 or None.  Returns float64 [N]."""
    gt = np.mean(g_unpacked[:, loci, :], axis=2)
    if dom is not None and np.any(dom):
        gt = np.clip(gt * (1 + np.asarray(dom)[loci]), a_min=None, a_max=1)
    if len(loci) > 1:
        # ops/selection.py:44 uses the builtin sum over loci, left to right
        z = np.full(gt.shape[0], 0.0)
        acc = np.zeros(gt.shape[0])
        for j in range(len(loci)):
            acc = acc + gt[:, j] * alpha[j]
        z = 0.5 + acc
    else:
        z = gt[:, 0]
    return z


def phenotype_packed(geno, rows, loci, alpha, dom=None):
    """Same, reading bit-packed rows."""
    loci = np.asarray(loci, dtype=np.int64)
    w = loci >> 6
    b = (loci & 63).astype(np.uint64)
    a0 = (geno[rows][:, 0, :][:, w] >> b) & np.uint64(1)
    a1 = (geno[rows][:, 1, :][:, w] >> b) & np.uint64(1)
    g = np.stack([a0, a1], axis=2).astype(np.float64)
    gt = g.mean(axis=2)
    if dom is not None and np.any(dom):
        gt = np.clip(gt * (1 + np.asarray(dom)[loci]), a_min=None, a_max=1)
    if len(loci) > 1:
        acc = np.zeros(gt.shape[0])
        for j in range(len(loci)):
            acc = acc + gt[:, j] * alpha[j]
        return 0.5 + acc
    return gt[:, 0]


# --------------------------------------------------------------------------
# A15  fitness and death probability (ops/selection.py:51-125)
# --------------------------------------------------------------------------

def fitness_traits(e, z, trait_lyr, phi, gamma, univ_adv):
    """e: [N, n_lyr]; z: [N, n_trt]; phi: list of scalar or [N] per trait.
    w = clip(prod_t 1 - phi_t |e^(not univ_adv) - z_t|^gamma_t, 0.001, None)."""
    N = z.shape[0]
    w = np.ones(N)
    for t in range(z.shape[1]):
        et = e[:, trait_lyr[t]] ** (not univ_adv[t])
        w = w * (1 - phi[t] * (np.abs(et - z[:, t]) ** gamma[t]))
    return np.clip(w, a_min=0.001, a_max=None)


def fitness_deleterious(g_unpacked, delet_loci, delet_s):
    """ops/selection.py:78-94."""
    deletome = np.sum(g_unpacked[:, delet_loci, :], axis=2)
    return (1 - deletome * np.asarray(delet_s)).prod(axis=1)


def prob_death(d_at_cell, w):
    """ops/selection.py:119-125."""
    return 1 - (1 - d_at_cell) * w


# --------------------------------------------------------------------------
# A2  movement transform (ops/movement.py:74-92) and dispersal (:98-141)
# --------------------------------------------------------------------------

def move_transform(x, y, theta, dist, dim, res_ratio=(1, 1), dtype=np.float64):
    """new = clip(old + (cos, sin)(theta) * dist * res_ratio, 0, dim - 0.001)."""
    x = np.asarray(x, dtype=dtype)
    y = np.asarray(y, dtype=dtype)
    theta = np.asarray(theta, dtype=dtype)
    dist = np.asarray(dist, dtype=dtype)
    dx = np.cos(theta) * dist
    dy = np.sin(theta) * dist
    if res_ratio[0] != 1:
        dx = dx * dtype(res_ratio[0])
    if res_ratio[1] != 1:
        dy = dy * dtype(res_ratio[1])
    nx = np.clip(x + dx, dtype(0), dtype(dim[0] - 0.001))
    ny = np.clip(y + dy, dtype(0), dtype(dim[1] - 0.001))
    return nx, ny


def dispersal(mid_x, mid_y, theta, dist, dim, res_ratio=(1, 1),
              dtype=np.float64):
    """ops/movement.py:98-141 with the retry loop made explicit.

    theta, dist: [A, B] = A successive attempts for each of B offspring.
    The reference clips to [0, dim-0.001] and THEN tests 0 < x < dim, so an
    attempt is rejected exactly when a clipped coordinate equals 0 (a draw
    that crossed the low edge); draws crossing the high edge are accepted
    clipped.  The first accepted attempt is used; if all A fail the last is
    kept (the build bounds the loop at A = 8).  Returns x, y, attempt_used."""
    theta = np.atleast_2d(theta)
    dist = np.atleast_2d(dist)
    A, B = theta.shape
    ox = np.zeros(B, dtype=dtype)
    oy = np.zeros(B, dtype=dtype)
    used = np.full(B, -1, dtype=np.int32)
    for a in range(A):
        nx, ny = move_transform(mid_x, mid_y, theta[a], dist[a], dim,
                                res_ratio, dtype)
        ok = (nx > 0) & (nx < dim[0]) & (ny > 0) & (ny < dim[1])
        take = (used < 0) & (ok | (a == A - 1))
        ox[take] = nx[take]
        oy[take] = ny[take]
        used[take] = a
    return ox, oy, used


def gather_e(rasts, x, y):
    """Species._set_e (structs/species.py:913-922): e[i,l]=rast_l[int(y),int(x)]."""
    cx = np.asarray(x).astype(np.int64)
    cy = np.asarray(y).astype(np.int64)
    return np.stack([r[cy, cx] for r in rasts], axis=1)


# --------------------------------------------------------------------------
# A3  conductance-surface direction distribution (utils/spatial.py:365-461)
# --------------------------------------------------------------------------

QUEEN_DIRS = np.array([-3 * np.pi / 4, -np.pi / 2, -np.pi / 4, np.pi,
                       0, 3 * np.pi / 4, np.pi / 2, np.pi / 4])
# neighbour offsets (dy, dx) in the order of QUEEN_DIRS (row-major 3x3 minus
# the centre; utils/spatial.py:434-435 with i = row = y, j = col = x)
QUEEN_OFFS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0),
              (1, 1)]


def conductance_weights(rast, cy, cx):
    """8 neighbour weights of cell (cy, cx) on the zero-embedded raster
    (utils/spatial.py:442-444,410-418): values / sum, or 1/8 each if sum==0."""
    H, W = rast.shape
    cy = np.asarray(cy)
    cx = np.asarray(cx)
    n = np.zeros((cy.size, 8))
    for k, (dy, dx) in enumerate(QUEEN_OFFS):
        yy = cy + dy
        xx = cx + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        n[ok, k] = rast[yy[ok], xx[ok]]
    s = n.sum(axis=1)
    p = np.where(s[:, None] > 0, n / np.where(s > 0, s, 1)[:, None], 0.125)
    return p


def conductance_unimodal_loc(rast, cy, cx):
    """utils/spatial.py:370-381: arithmetic mean of the bearings of the
    maximum-valued neighbours."""
    H, W = rast.shape
    cy = np.asarray(cy)
    cx = np.asarray(cx)
    n = np.zeros((cy.size, 8))
    for k, (dy, dx) in enumerate(QUEEN_OFFS):
        yy = cy + dy
        xx = cx + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        n[ok, k] = rast[yy[ok], xx[ok]]
    ismax = n == n.max(axis=1, keepdims=True)
    return (ismax * QUEEN_DIRS).sum(axis=1) / ismax.sum(axis=1)


# --------------------------------------------------------------------------
# A6/A7  mating pairs (structs/species.py:2157-2215; utils/spatial.py:191-245;
#        ops/mating.py:24-117)
# --------------------------------------------------------------------------

def neighbour_lists(x, y, radius, dtype=np.float32):
    """All j != i with dx^2 + dy^2 <= r^2 (cKDTree.query_ball_point, p=2),
    computed in `dtype` arithmetic in the order (dx*dx) + (dy*dy).  O(N^2):
    small inputs only."""
    x = np.asarray(x, dtype=dtype)
    y = np.asarray(y, dtype=dtype)
    r2 = dtype(radius) * dtype(radius)
    out = []
    for i in range(x.size):
        dx = x - x[i]
        dy = y - y[i]
        d2 = dx * dx + dy * dy
        nb = np.nonzero(d2 <= r2)[0]
        out.append(nb[nb != i])
    return out


import os as _os

def cell_div():
    """cells per mating radius of the uniform / inverse-distance mate search (csrc/gnx_api.hip:
    setup_hash_grid, GNX_CELL_DIV, read per call): 1 = cells of a radius and the 3 x 3 block around
    the focal cell (default), 2 = cells of half a radius and the 5 x 5 block"""
    return max(1, min(2, int(_os.environ.get('GNX_CELL_DIV', '1'))))


def hash_grid(dim, radius, fine=False):
    """geometry of the device's hash grid (csrc/gnx_api.hip: setup_hash_grid): cell size >= mating
    radius / cell_div() (an eighth of the radius for the nearest-mate search, fine=True), at
    most 2048 cells per axis"""
    W, H = dim
    cs = float(radius) * (1.0 + 1e-9) if radius > 0 else 8.0
    if radius > 0:
        cs /= 8.0 if fine else float(cell_div())
    cs = max(cs, max(W, H) / 2048.0)
    inv_cs = 1.0 / cs
    return inv_cs, max(1, int(np.ceil(W / cs))), max(1, int(np.ceil(H / cs)))


def cell_ref(dim, radius, fine=False):
    """cells that cover the mating radius (the device's cell_ref): the candidate block of a focal
    individual is (2 * ref + 1)^2 cells"""
    if radius <= 0:
        return 1
    inv_cs, _, _ = hash_grid(dim, radius, fine)
    return max(1, int(np.ceil(float(radius) * (1.0 + 1e-9) * inv_cs - 1e-12)))


def cell_of(x, y, inv_cs, ncx, ncy):
    """gnx_cell_of: (int)((double)x * inv_cs), clamped to the last cell"""
    cx = np.minimum(ncx - 1, (np.asarray(x, dtype=np.float32).astype(np.float64)
                              * inv_cs).astype(np.int64))
    cy = np.minimum(ncy - 1, (np.asarray(y, dtype=np.float32).astype(np.float64)
                              * inv_cs).astype(np.int64))
    return cy * ncx + cx, cx, cy


def pair_order_keys(x, y, ids, dim, radius, mate_mode='uniform'):
    """Order key of an individual as the focal one of a pair: (hash cell << 40) | id.  Pairs
    are taken in ascending key order (the device's cell-sorted slot order), which fixes the
    order offspring ids are handed out in - independent of storage order and of tiling."""
    inv_cs, ncx, ncy = hash_grid(dim, radius if radius is not None else -1.0,
                                 fine=mate_mode == 'nearest')
    cell, _, _ = cell_of(x, y, inv_cs, ncx, ncy)
    return (cell.astype(np.int64) << 40) | np.asarray(ids, dtype=np.int64)


def mate_tries(M):
    """index draws before the exact scan: 4 * clamp(M // 8, 8, 2048)"""
    return 4 * np.clip(np.asarray(M, dtype=np.int64) >> 3, 8, 2048)


def mate_blocks_weighted(M):
    """Philox blocks (2 weighted tries each) before the exact scan: clamp(M // 4, 16, 4096)"""
    return np.clip(np.asarray(M, dtype=np.int64) >> 2, 16, 4096)


def choose_mates_uniform(x, y, ids, radius, seed, step, dim, focal=None, weighted=False,
                         fallback_out=None):
    """The build's uniform mate choice (utils/spatial.py:232-242 picks
    np.random.choice among the neighbours within the radius): rejection sampling in
    index space.  Candidates of a focal individual = the individuals in the (2 ref + 1)^2
    block of hash cells around its own (3 x 3; 5 x 5 with cells of half a radius), in CANONICAL order (cell rows ascending, cells
    ascending inside a row, ids ascending inside a cell); the focal draws indices
    (w * M) >> 32 from its Philox stream (seed, id, step, OP_MATE_PICK) until the
    candidate drawn lies within the radius and is not itself; after T = mate_tries(M)
    rejections it takes the ((w_T * m) >> 32)-th of the m in-radius candidates in
    canonical order (w_T = the first word of Philox block T/4).
    weighted=True is the inverse-distance choice (utils/spatial.py:209-229: P(j) ~ r - d_ij
    over the neighbours with d > 0): two stream words per try, the index and an acceptance
    draw u - the candidate drawn is taken iff u * r < r - d (f32); after
    mate_blocks_weighted(M) blocks the exact pick: total weight W summed in canonical order
    (f32), then the first candidate whose running weight passes u01(w) * W.
    focal: optional mask of the individuals that need a mate (others get -1);
    fallback_out: optional bool array, set where the exact pick decided."""
    import philox as P
    F = np.float32
    x = np.asarray(x, dtype=F)
    y = np.asarray(y, dtype=F)
    ids = np.asarray(ids, dtype=np.int64)
    n = x.size
    mate = np.full(n, -1, dtype=np.int64)
    if n < 2:
        return mate
    inv_cs, ncx, ncy = hash_grid(dim, radius)
    cell, cx, cy = cell_of(x, y, inv_cs, ncx, ncy)
    order = np.lexsort((ids & 0xffffffff, cell))            # canonical order of the slots
    cell_sorted = cell[order]
    cell_start = np.searchsorted(cell_sorted, np.arange(ncx * ncy + 1))
    foc = np.arange(n) if focal is None else np.nonzero(np.asarray(focal, dtype=bool))[0]
    if foc.size == 0:
        return mate
    ref = cell_ref(dim, radius)
    n_rows = 2 * ref + 1
    lo, hi = np.maximum(cx[foc] - ref, 0), np.minimum(cx[foc] + ref, ncx - 1)
    st = np.zeros((foc.size, n_rows), np.int64)
    ln = np.zeros((foc.size, n_rows), np.int64)
    for q in range(n_rows):
        ry = cy[foc] - ref + q
        ok = (ry >= 0) & (ry < ncy)
        ryc = np.clip(ry, 0, ncy - 1)
        s0 = cell_start[ryc * ncx + lo]
        s1 = cell_start[ryc * ncx + hi + 1]
        st[:, q] = np.where(ok, s0, 0)
        ln[:, q] = np.where(ok, s1 - s0, 0)
    M = ln.sum(axis=1)
    r2 = F(radius) * F(radius)
    found = np.full(foc.size, -1, dtype=np.int64)
    rF = F(radius)
    per_blk = 2 if weighted else 4
    blocks = mate_blocks_weighted(M) if weighted else mate_tries(M) // 4
    fid = ids[foc].astype(np.uint64)
    active = M > 1
    blk = 0
    while active.any():
        a = np.nonzero(active & (blocks > blk))[0]
        if a.size == 0:
            break
        w4 = P.philox4x32(seed, fid[a], P.block_index(step, P.OP_MATE_PICK, blk)).astype(
            np.uint64)
        for t in range(per_blk):
            live = active[a]
            if not live.any():
                break
            aa = a[live]
            wi = w4[live, 2 * t if weighted else t]
            j = ((wi * M[aa].astype(np.uint64)) >> np.uint64(32)).astype(np.int64)
            slot = st[aa, 0] + j
            acc = ln[aa, 0].copy()
            for q in range(1, n_rows):
                slot = np.where(j >= acc, st[aa, q] + (j - acc), slot)
                acc = acc + ln[aa, q]
            c = order[slot]
            dx = x[c] - x[foc[aa]]
            dy = y[c] - y[foc[aa]]
            d2 = dx * dx + dy * dy
            ok = (c != foc[aa]) & (d2 <= r2)
            if weighted:
                u = P.u01(w4[live, 2 * t + 1].astype(np.uint32))
                ok = ok & (d2 > 0) & ((u * rF).astype(F) < (rF - np.sqrt(d2)).astype(F))
            found[aa[ok]] = c[ok]
            active[aa[ok]] = False
        blk += 1
    for k in np.nonzero(active)[0]:                 # exact fallback
        i = foc[k]
        c = np.concatenate([order[st[k, q]:st[k, q] + ln[k, q]] for q in range(n_rows)])
        dx = x[c] - x[i]
        dy = y[c] - y[i]
        d2 = dx * dx + dy * dy
        sel = (c != i) & (d2 <= r2)
        if weighted:
            sel = sel & (d2 > 0)
        inr = c[sel]
        if inr.size:
            if fallback_out is not None:
                fallback_out[i] = True
            w = int(P.philox4x32(seed, fid[k:k + 1],
                                 P.block_index(step, P.OP_MATE_PICK, int(blocks[k])))[0, 0])
            if not weighted:
                found[k] = inr[(w * inr.size) >> 32]
            else:
                wt = (rF - np.sqrt(d2[sel])).astype(F)
                run = F(0)
                for v in wt:
                    run = F(run + v)
                target = F(P.u01(np.array([w], np.uint32))[0] * run)
                acc = F(0)
                pick = inr[-1]
                for cc, v in zip(inr, wt):
                    acc = F(acc + v)
                    if acc > target:
                        pick = cc
                        break
                found[k] = pick
    mate[foc] = found
    return mate


def mate_fallbacks(x, y, ids, radius, seed, step, mode='uniform', dim=None, focal=None):
    """Who ran out of index tries and took the exact pick (test instrumentation)."""
    out = np.zeros(np.asarray(x).size, dtype=bool)
    choose_mates_uniform(x, y, ids, radius, seed, step, dim, focal=focal,
                         weighted=mode == 'inverse', fallback_out=out)
    return out


def choose_mates(x, y, ids, radius, seed, step, mode='uniform',
                 dtype=np.float32, dim=None, focal=None):
    """For each individual with >= 1 other within radius pick one mate.

    mode 'uniform'  : utils/spatial.py:232-242 picks np.random.choice(opts);
                      the build samples indices of a canonical candidate list until
                      one lies within the radius (choose_mates_uniform; needs dim).
    mode 'nearest'  : utils/spatial.py:194-203 (ties -> smaller id).
    mode 'inverse'  : utils/spatial.py:209-229, P(j) ~ (r - d_ij) over
                      candidates with d > 0; the build samples candidate indices like
                      'uniform' and accepts with probability (r - d) / r
                      (choose_mates_uniform(weighted=True); needs dim).
    Returns mate index per individual (-1 = none)."""
    from philox import pair_hash, u01
    if mode in ('uniform', 'inverse'):
        assert dim is not None, "index-sampled mate choice needs the landscape dim (W, H)"
        return choose_mates_uniform(x, y, ids, radius, seed, step, dim, focal=focal,
                                    weighted=mode == 'inverse')
    x = np.asarray(x, dtype=dtype)
    y = np.asarray(y, dtype=dtype)
    ids = np.asarray(ids, dtype=np.uint64)
    nbs = neighbour_lists(x, y, radius, dtype)
    mate = np.full(x.size, -1, dtype=np.int64)
    for i, nb in enumerate(nbs):
        if nb.size == 0:
            continue
        if mode == 'nearest':
            dx = x[nb] - x[i]
            dy = y[nb] - y[i]
            d2 = dx * dx + dy * dy
            order = np.lexsort((ids[nb], d2))
            mate[i] = nb[order[0]]
        elif mode == 'inverse':
            dx = x[nb] - x[i]
            dy = y[nb] - y[i]
            d = np.sqrt(dx * dx + dy * dy)
            ok = d > 0
            if not ok.any():
                continue
            nb2 = nb[ok]
            wgt = dtype(radius) - d[ok]
            h = pair_hash(seed, step, ids[i], ids[nb2])
            u = u01(h)
            key = -np.log(u) / wgt
            order = np.lexsort((ids[nb2], key))
            mate[i] = nb2[order[0]]
    return mate


def pairs_from_mates(mate, keep, ids=None):
    """structs/species.py:2210-2214 then ops/mating.py:62-65.

    mate: chosen mate per focal (-1 none); keep: Bernoulli(b) per focal.
    Reference: pairs = [(i, mate_i)] for focal with a mate, filtered by the
    Bernoulli draw, then de-duplicated as unordered sets.  Build rule giving
    the same SET: drop (i, m) iff (m, i) is also present and m < i.
    Returned in ascending focal order."""
    mate = np.asarray(mate)
    keep = np.asarray(keep, dtype=bool)
    has = (mate >= 0) & keep
    i = np.nonzero(has)[0]
    m = mate[i]
    if ids is None:
        recip = has[m] & (mate[m] == i) & (m < i)
    else:       # the build's rule: by id, so that every tile decides alike
        ids = np.asarray(ids)
        recip = has[m] & (mate[m] == i) & (ids[m] < ids[i])
    return np.stack([i[~recip], m[~recip]], axis=1)


def sexed_pairs(mate, keep, sex):
    """ops/mating.py:41-55: keep pairs with female (0) focal and male (1) mate."""
    mate = np.asarray(mate)
    has = (mate >= 0) & np.asarray(keep, dtype=bool)
    i = np.nonzero(has)[0]
    m = mate[i]
    ok = (sex[i] == 0) & (sex[m] == 1)
    return np.stack([i[ok], m[ok]], axis=1)


def repro_age_filter(pairs, age, repro_age, sexed):
    """ops/mating.py:79-104."""
    if repro_age is None or not np.any(np.atleast_1d(repro_age) > 0):
        return pairs
    if len(pairs) == 0:
        return pairs
    if sexed:
        ok = (age[pairs[:, 0]] >= repro_age[0]) & (age[pairs[:, 1]] >= repro_age[1])
    else:
        ok = (age[pairs[:, 0]] >= repro_age) & (age[pairs[:, 1]] >= repro_age)
    return pairs[ok]


def panmictic_pairs(draws, n_mates):
    """structs/species.py:2178-2194: 2*n_mates draws with replacement folded
    to [n_mates, 2], selfing pairs dropped."""
    p = np.asarray(draws).reshape(n_mates, 2)
    return p[p[:, 0] != p[:, 1]]


# --------------------------------------------------------------------------
# A8  births (ops/mating.py:120-126; structs/species.py:604-609)
# --------------------------------------------------------------------------

def poisson_knuth(lam, u):
    """Poisson(lam) by multiplication of uniforms (Knuth); u: f32 [n, K].
    Exact f32 products, so the HIP kernel reproduces it bit for bit."""
    thr = F32(np.exp(-np.float64(lam)))
    n, K = u.shape
    k = np.zeros(n, dtype=np.int32)
    p = np.ones(n, dtype=F32)
    alive = np.ones(n, dtype=bool)
    for j in range(K):
        p = np.where(alive, p * u[:, j], p).astype(F32)
        alive = alive & (p > thr)
        k = k + alive.astype(np.int32)
    return k


def n_births(n_pairs, lam, fixed, poisson_draws=None):
    if fixed:
        return np.full(n_pairs, int(lam), dtype=np.int32)
    return np.clip(np.asarray(poisson_draws), a_min=1, a_max=None).astype(np.int32)


# --------------------------------------------------------------------------
# A13  local density (utils/spatial.py:34-146, 270-360)
# --------------------------------------------------------------------------

class DensityLattice:
    """The union of the four window grids of _DensityGridStack is a regular
    lattice of spacing ww/2 (utils/spatial.py:282-293,354-360); the window of
    lattice node j covers [(j-1)*hww, (j+1)*hww) (:79-82,327-328), i.e. the two
    half-window bins j-1 and j.  areas = window area inside the landscape,
    0 -> 1e-4 (:313-319)."""

    def __init__(self, dim, ww=None):
        self.dim = (int(dim[0]), int(dim[1]))
        if ww is None:
            ww = round(0.1 * max(self.dim))          # utils/spatial.py:110-111
        self.ww = ww
        hww = ww / 2.
        self.hww = hww
        self.J = [self._n_nodes(self.dim[0], ww), self._n_nodes(self.dim[1], ww)]
        self.cx = np.arange(self.J[0]) * hww
        self.cy = np.arange(self.J[1]) * hww
        ax = np.clip(np.minimum(self.cx + hww, self.dim[0])
                     - np.maximum(self.cx - hww, 0), 0, None)
        ay = np.clip(np.minimum(self.cy + hww, self.dim[1])
                     - np.maximum(self.cy - hww, 0), 0, None)
        areas = ay[:, None] * ax[None, :]
        areas[areas == 0] = 0.0001
        self.areas = areas                              # [Jy, Jx]

    @staticmethod
    def _n_nodes(d, ww):
        hww = ww / 2.
        edge = np.arange(0, d + ww, ww)                 # :282
        inner = np.arange(0 + hww, d + hww, ww)         # :283
        return int(round(max(edge.max(), inner.max()) / hww)) + 1

    def bins(self, x, y):
        """Half-window bin counts [Jy, Jx] (bin b covers [b*hww, (b+1)*hww))."""
        hx = np.floor(np.asarray(x, dtype=np.float64) / self.hww).astype(np.int64)
        hy = np.floor(np.asarray(y, dtype=np.float64) / self.hww).astype(np.int64)
        hist = np.zeros((self.J[1], self.J[0]), dtype=np.int64)
        np.add.at(hist, (np.minimum(hy, self.J[1] - 1), np.minimum(hx, self.J[0] - 1)), 1)
        return hist

    def nodes_from_bins(self, hist):
        """node density = (sum of the 2x2 bins around the node) / area"""
        h = np.asarray(hist, dtype=np.int64).reshape(self.J[1], self.J[0])
        p = np.zeros((self.J[1] + 1, self.J[0] + 1), dtype=np.int64)
        p[1:, 1:] = h
        c = p[1:, 1:] + p[:-1, 1:] + p[1:, :-1] + p[:-1, :-1]
        return c / self.areas

    def counts(self, x, y):
        """Window counts at every lattice node, [Jy, Jx]."""
        hx = np.floor(np.asarray(x, dtype=np.float64) / self.hww).astype(np.int64)
        hy = np.floor(np.asarray(y, dtype=np.float64) / self.hww).astype(np.int64)
        nbx, nby = self.J[0], self.J[1]                 # bins 0..J-1 (last empty)
        hist = np.zeros((nby + 1, nbx + 1), dtype=np.int64)
        np.add.at(hist, (hy, hx), 1)
        c = np.zeros((self.J[1], self.J[0]), dtype=np.int64)
        for j in range(self.J[0]):
            for i in range(self.J[1]):
                s = 0
                for dj in (-1, 0):
                    for di in (-1, 0):
                        jj, ii = j + dj, i + di
                        if jj >= 0 and ii >= 0:
                            s += hist[ii, jj]
                c[i, j] = s
        return c

    def node_density(self, x, y):
        return self.counts(x, y) / self.areas


def natural_spline_second_derivs(v, h, axis):
    """Second derivatives of the natural cubic spline through equally spaced
    samples v along `axis` (m_0 = m_{J-1} = 0)."""
    v = np.moveaxis(np.asarray(v, dtype=np.float64), axis, 0)
    J = v.shape[0]
    m = np.zeros_like(v)
    if J > 2:
        A = np.zeros((J - 2, J - 2))
        np.fill_diagonal(A, 4.0)
        idx = np.arange(J - 3)
        A[idx, idx + 1] = 1.0
        A[idx + 1, idx] = 1.0
        rhs = 6.0 * (v[:-2] - 2 * v[1:-1] + v[2:]) / (h * h)
        m[1:-1] = np.linalg.solve(A, rhs.reshape(J - 2, -1)).reshape(rhs.shape)
    return np.moveaxis(m, 0, axis)


def spline_eval_1d(v0, v1, m0, m1, t, h):
    a = 1.0 - t
    return (a * v0 + t * v1
            + ((a * a * a - a) * m0 + (t * t * t - t) * m1) * (h * h / 6.0))


def density_raster(lat, x, y):
    """Build's estimator: natural bicubic spline through the lattice-node
    densities, evaluated at every cell centre (j+.5, i+.5), clipped >= 0
    (clip: structs/species.py:865; ops/demography.py:81).  The reference
    interpolates the same nodes with scipy griddata(method='cubic')
    (utils/spatial.py:144), a Clough-Tocher scheme on qhull's triangulation of
    this regular lattice, which is not bit-reproducible; tests state the
    tolerance between the two."""
    V = lat.node_density(x, y)
    return spline_raster(lat, V)


def spline_coeffs(lat, V):
    h = lat.hww
    Mx = natural_spline_second_derivs(V, h, axis=1)
    My = natural_spline_second_derivs(V, h, axis=0)
    Mxy = natural_spline_second_derivs(My, h, axis=1)
    return Mx, My, Mxy


def spline_at(lat, V, px, py, coeffs=None):
    """Evaluate the bicubic spline at points (px, py)."""
    h = lat.hww
    Mx, My, Mxy = coeffs if coeffs is not None else spline_coeffs(lat, V)
    fx = np.asarray(px, dtype=np.float64) / h
    fy = np.asarray(py, dtype=np.float64) / h
    j = np.clip(np.floor(fx).astype(np.int64), 0, lat.J[0] - 2)
    i = np.clip(np.floor(fy).astype(np.int64), 0, lat.J[1] - 2)
    tx = fx - j
    ty = fy - i

    def row(ii):
        v = spline_eval_1d(V[ii, j], V[ii, j + 1], Mx[ii, j], Mx[ii, j + 1], tx, h)
        m = spline_eval_1d(My[ii, j], My[ii, j + 1], Mxy[ii, j], Mxy[ii, j + 1],
                           tx, h)
        return v, m
    v0, m0 = row(i)
    v1, m1 = row(i + 1)
    return spline_eval_1d(v0, v1, m0, m1, ty, h)


def spline_raster(lat, V):
    gx, gy = np.meshgrid(np.arange(lat.dim[0]) + 0.5, np.arange(lat.dim[1]) + 0.5)
    d = spline_at(lat, V, gx.ravel(), gy.ravel()).reshape(lat.dim[1], lat.dim[0])
    return np.clip(d, 0, None)


# --------------------------------------------------------------------------
# A14  demography raster algebra (ops/demography.py:95-172)
# --------------------------------------------------------------------------

def calc_dNdt(R, N, K):
    with np.errstate(divide='ignore', invalid='ignore'):
        dNdt = R * (1 - (N / K)) * N
    nmax = N.max()
    dNdt = np.clip(dNdt, a_min=-1 * nmax, a_max=None)
    dNdt[np.isnan(dNdt)] = -1 * nmax
    dNdt[np.isinf(dNdt)] = -1 * nmax
    return dNdt


def calc_d(N, K, n_pairs, R, b, lam, d_min, d_max):
    """ops/demography.py:253-291: dNdt -> N_b -> N_d -> d."""
    dNdt = calc_dNdt(R, N, K)
    N_b = b * lam * n_pairs
    N_d = N_b - dNdt
    with np.errstate(divide='ignore', invalid='ignore'):
        d = N_d / N
    d[np.isnan(d)] = 0
    d = np.clip(d, d_min, d_max)
    return dNdt, N_b, N_d, d


# --------------------------------------------------------------------------
# G  starting genomes (structs/genome.py:1108-1157)
# --------------------------------------------------------------------------

def starting_mutation_counts(N, p):
    """n_l = round(2N p_l), forced into [1, 2N-1] unless p_l in {0,1}
    (structs/genome.py:1124-1130; python round = banker's)."""
    p = np.asarray(p, dtype=np.float64)
    n = np.array([int(round(2 * N * f, 0)) for f in p], dtype=np.int64)
    n = np.where((n == 2 * N) & (p < 1), n - 1, n)
    n = np.where((n == 0) & (p > 0), 1, n)
    return n


def starting_genomes(N, L, n_per_site, seed, align_words=16):
    """Choose exactly n_l of the 2N homologues per site, uniformly without
    replacement, by selection sampling (Knuth 3.4.2 Algorithm S) over
    homologue index q = 2*ind + hom with u = site_hash(seed, l, q):
    select iff (u * (2N - q)) >> 32 < n_l - selected_so_far.
    Reference shuffles the homologue list per site and takes the first n_l
    (structs/genome.py:1132-1133): same distribution.  Returns uint64 [N,2,W]."""
    from philox import site_hash
    n_per_site = np.asarray(n_per_site, dtype=np.int64)
    g = np.zeros((N, L, 2), dtype=np.uint8)
    sel = np.zeros(L, dtype=np.int64)
    sites = np.arange(L, dtype=np.uint64)
    for q in range(2 * N):
        u = site_hash(seed, sites, np.uint64(q)).astype(np.uint64)
        take = ((u * np.uint64(2 * N - q)) >> np.uint64(32)).astype(np.int64) \
            < (n_per_site - sel)
        g[q // 2, :, q % 2] = take
        sel += take
    return pack_genomes(g, align_words)


# ---------------------------------------------------------------------------
# statistics (reference sim/stats.py:359-421), g = int [N, L, 2]
# ---------------------------------------------------------------------------
def stats_het(g):
    """fraction of heterozygotes per locus (sim/stats.py:394-405)"""
    g = np.asarray(g)
    return (g[:, :, 0] != g[:, :, 1]).sum(axis=0) / g.shape[0]


def stats_maf(g):
    """minor-allele frequency per locus (sim/stats.py:408-421)"""
    g = np.asarray(g)
    f1 = g.sum(axis=(0, 2)) / (2 * g.shape[0])
    return np.where(f1 > 0.5, 1 - f1, f1)


def stats_ld(g):
    """r^2 over the 2N chromosomes, NaN diagonal (sim/stats.py:359-390)"""
    g = np.asarray(g)
    N, L, _ = g.shape
    chrom = np.concatenate([g[:, :, 0], g[:, :, 1]], axis=0).astype(np.float64)  # [2N, L]
    two_N = 2.0 * N
    f = chrom.sum(axis=0) / two_N
    f11 = (chrom.T @ chrom) / two_N
    D = f11 - (f[:, None] * f[None, :])
    with np.errstate(divide='ignore', invalid='ignore'):
        r2 = (D * D) / ((f * (1 - f))[:, None] * (f * (1 - f))[None, :])
    r2[np.arange(L), np.arange(L)] = np.nan
    return r2


# --------------------------------------------------------------------------
# f4  burn-in spatial tester (sim/burnin.py:44-59)
# --------------------------------------------------------------------------

def spatial_diff_stats(prev_counts, x, y, dim):
    """SpatialTester.update: per-cell counts of individuals (cell = int(x), int(y);
    counts[i, j] holds cell (x=j, y=i)), their difference to the previous counts,
    and np.mean / np.std (population) of the difference raster.
    Returns counts, mean, std."""
    W, H = dim
    cx = np.asarray(x).astype(np.int64)
    cy = np.asarray(y).astype(np.int64)
    counts = np.bincount(cy * W + cx, minlength=W * H).reshape(H, W).astype(np.float64)
    diff = counts - np.asarray(prev_counts, dtype=np.float64)
    return counts, float(np.mean(diff)), float(np.std(diff))

