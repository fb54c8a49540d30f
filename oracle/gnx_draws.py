"""TEST INFRASTRUCTURE ONLY.  The random draws of the HIP kernels restated in
numpy: same Philox streams (oracle/philox.py), same f32 formulas
(geonomics_amd/csrc/gnx_rng.h).  Integer draws are reproduced exactly; float
draws agree to transcendental-function rounding (logf/cosf/expf), which the
tests bound explicitly."""
import numpy as np

import philox as P

F = np.float32
PI = F(3.14159274101257324)


def _u(seed, ids, step, op, blk=0):
    return P.philox4x32(seed, np.asarray(ids, dtype=np.uint64),
                        P.block_index(step, op, blk))


def normal(u0, u1):
    return (np.sqrt(F(-2.0) * np.log(u0)) * np.cos(F(2.0) * PI * u1)).astype(F)


def distance(distr, p1, p2, r):
    """gnx_distance: lognormal / wald / levy from one Philox block r [n,4]."""
    p1, p2 = F(p1), F(p2)
    zn = normal(P.u01(r[:, 0]), P.u01(r[:, 1]))
    if distr == 'lognormal':
        return np.exp(p1 + p2 * zn).astype(F)
    if distr == 'wald':
        mu_2l = p1 / (F(2.0) * p2)
        Y = p1 * zn * zn
        X = p1 + mu_2l * (Y - np.sqrt(F(4.0) * p2 * Y + Y * Y))
        U = P.u01(r[:, 2])
        return np.where(U <= p1 / (p1 + X), X, p1 * p1 / X).astype(F)
    return (p1 + p2 / (zn * zn)).astype(F)


def vonmises(seed, ids, step, op, mu, kappa, first_word=0):
    """gnx_vonmises over the (id, step, op) stream starting at u32 index
    first_word; returns f32 angles."""
    ids = np.asarray(ids, dtype=np.uint64)
    n = ids.size
    nblk = 9 + (first_word // 4)
    words = np.concatenate([_u(seed, ids, step, op, b) for b in range(nblk)], axis=1)
    w = first_word
    mu, kappa = F(mu), F(kappa)
    if kappa < 1e-8:
        return (PI * (F(2.0) * P.u01(words[:, w]) - F(1.0))).astype(F)
    if kappa < 1e-5:
        sv = F(1.0) / kappa + kappa
    else:
        r = F(1.0) + np.sqrt(F(1.0) + F(4.0) * kappa * kappa)
        rho = (r - np.sqrt(F(2.0) * r)) / (F(2.0) * kappa)
        sv = (F(1.0) + rho * rho) / (F(2.0) * rho)
    Wv = np.ones(n, dtype=F)
    done = np.zeros(n, dtype=bool)
    pos = np.full(n, w, dtype=np.int64)
    rows = np.arange(n)
    for it in range(14):
        U = P.u01(words[rows, pos])
        V = P.u01(words[rows, pos + 1])
        Z = np.cos(PI * U)
        Wn = ((F(1.0) + sv * Z) / (sv + Z)).astype(F)
        Y = (kappa * (sv - Wn)).astype(F)
        with np.errstate(divide='ignore', invalid='ignore'):
            acc = (Y * (F(2.0) - Y) - V >= 0) | (np.log(Y / V) + F(1.0) - Y >= 0)
        upd = ~done
        Wv = np.where(upd, Wn, Wv)
        pos = np.where(upd, pos + 2, pos)
        done = done | acc
    U = P.u01(words[rows, pos])
    Wv = np.clip(Wv, F(-1), F(1))
    res = np.arccos(Wv).astype(F)
    res = np.where(U < F(0.5), -res, res) + mu
    neg = res < 0
    m = np.abs(res)
    m = np.fmod(m + PI, F(2.0) * PI) - PI
    return np.where(neg, -m, m).astype(F)


def move_draws(seed, ids, step, distr, p1, p2, mu, kappa):
    theta = vonmises(seed, ids, step, P.OP_MOVE_DIR, mu, kappa)
    dist = distance(distr, p1, p2, _u(seed, ids, step, P.OP_MOVE_DIST))
    return theta, dist


def keep_draws(seed, ids, step, b):
    return P.u01(_u(seed, ids, step, P.OP_PAIR_KEEP)[:, 0]) < F(b)


def panmixia_draws(seed, ids, step, N):
    r = _u(seed, ids, step, P.OP_PAIR_KEEP, 1).astype(np.uint64)
    f = ((r[:, 0] * np.uint64(N)) >> np.uint64(32)).astype(np.int64)
    m = ((r[:, 1] * np.uint64(N)) >> np.uint64(32)).astype(np.int64)
    return f, m


def births_draws(seed, focal_ids, step, lam):
    import gnx_oracle as O
    u = np.concatenate([P.u01(_u(seed, focal_ids, step, P.OP_BIRTHS, b))
                        for b in range(16)], axis=1)
    return np.maximum(O.poisson_knuth(lam, u), 1)


def offspring_draws(seed, off_ids, step, n_paths, sexed, p_male):
    r = _u(seed, off_ids, step, P.OP_OFFSPRING, 0)
    r2 = _u(seed, off_ids, step, P.OP_OFFSPRING, 1)
    start = np.stack([r[:, 0] & 1, (r[:, 0] >> 1) & 1], axis=1).astype(np.uint8)
    keys = np.stack([(r[:, 1].astype(np.uint64) * np.uint64(n_paths)) >> np.uint64(32),
                     (r[:, 2].astype(np.uint64) * np.uint64(n_paths)) >> np.uint64(32)],
                    axis=1).astype(np.int32)
    male_first = P.u01(r2[:, 0]) < F(p_male)
    coin = P.u01(r2[:, 1]) < F(0.5)
    sex = np.where(bool(sexed) & male_first, 1, coin.astype(np.int64)).astype(np.uint8)
    return start, keys, sex


def dispersal_draws(seed, off_ids, step, distr, p1, p2, attempts=8):
    th, ds = [], []
    for a in range(attempts):
        r = _u(seed, off_ids, step, P.OP_DISPERSAL, a)
        th.append((PI * (F(2.0) * P.u01(r[:, 3]) - F(1.0))).astype(F))
        ds.append(distance(distr, p1, p2, r))
    return np.stack(th), np.stack(ds)


def death_draws(seed, ids, step):
    return P.u01(_u(seed, ids, step, P.OP_DEATH)[:, 0])


def init_positions(seed, n, W, H):
    r = _u(seed, np.arange(n), 0, P.OP_INIT)
    x = np.minimum(P.u01(r[:, 0]) * F(W), F(W - 0.001))
    y = np.minimum(P.u01(r[:, 1]) * F(H), F(H - 0.001))
    sex = (P.u01(r[:, 2]) < F(0.5)).astype(np.uint8)
    return x.astype(F), y.astype(F), sex


QUEEN_DIRS_F = np.array([-2.35619449019234492885, -1.57079632679489661923,
                         -0.78539816339744830962, 3.14159265358979323846, 0.0,
                         2.35619449019234492885, 1.57079632679489661923,
                         0.78539816339744830962], dtype=F)
QUEEN_OFFS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


def surf_directions(seed, ids, step, op, rast, x, y, mixture, kappa, first_blk=0):
    """surf_sample (csrc/gnx_kernels_pop.hip): the on-the-fly restatement of the
    reference's conductance-surface LUT (utils/spatial.py:365-461)."""
    rast = np.asarray(rast, dtype=F)
    H, W = rast.shape
    cx = np.asarray(x).astype(np.int64)
    cy = np.asarray(y).astype(np.int64)
    n = np.zeros((cx.size, 8), dtype=F)
    for k, (dy, dx) in enumerate(QUEEN_OFFS):
        yy, xx = cy + dy, cx + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        n[ok, k] = rast[yy[ok], xx[ok]]
    ssum = np.zeros(cx.size, dtype=F)
    for k in range(8):
        ssum = (ssum + n[:, k]).astype(F)
    words = _u(seed, ids, step, op, first_blk)
    if mixture:
        u = P.u01(words[:, 0])
        t = (u * ssum).astype(F)
        pick = np.full(cx.size, -1)
        c = np.zeros(cx.size, dtype=F)
        last = np.zeros(cx.size, dtype=np.int64)
        for k in range(8):
            c = (c + n[:, k]).astype(F)
            last = np.where(n[:, k] > 0, k, last)
            pick = np.where((pick < 0) & (t < c), k, pick)
        pick = np.where(pick < 0, last, pick)
        uni = np.minimum(7, (u * F(8.0)).astype(np.int64))
        pick = np.where(ssum > 0, pick, uni)
        loc = QUEEN_DIRS_F[pick]
        first_word = 1
    else:
        mx = n.max(axis=1)
        ismax = n == mx[:, None]
        acc = np.zeros(cx.size, dtype=F)
        for k in range(8):
            acc = np.where(ismax[:, k], (acc + QUEEN_DIRS_F[k]).astype(F), acc)
        loc = (acc / ismax.sum(axis=1).astype(F)).astype(F)
        first_word = 0
    vm = vonmises(seed, ids, step, op, 0.0, kappa, first_word=first_word + 4 * first_blk)
    return (loc + vm).astype(F)
