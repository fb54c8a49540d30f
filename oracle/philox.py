"""TEST INFRASTRUCTURE ONLY (see oracle/README.md) - not imported by the product.

Counter-based random numbers for the oracle: Philox4x32-10 exactly as the
rocRAND device generator the HIP kernels use
(/opt/rocm/include/rocrand/rocrand_philox4x32_10.h: key = seed lo/hi,
counter = {offset/4 lo, offset/4 hi, subsequence lo, subsequence hi},
ten rounds, output {hi1^c.y^k.x, lo1, hi0^c.w^k.y, lo0}).

The reference (geonomics/sim/model.py:364-366) seeds ONE global MT19937 stream
that is consumed in data-dependent order; that cannot be reproduced on a GPU.
The build's stream layout is instead:

    subsequence = individual id   (or offspring id / focal id, see each op)
    block index = ((step * 32 + op) * 64 + blk)         [= rocRAND offset / 4]

so every (individual, step, op) owns 64 blocks of 4 x u32, independent of slot
order, launch geometry and GPU count.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = np.uint32(0x9E3779B9)
W1 = np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)

# op codes (shared with geonomics_amd/csrc/gnx_rng.h)
OP_MOVE_DIR = 0
OP_MOVE_DIST = 1
OP_PAIR_KEEP = 2
OP_BIRTHS = 3
OP_DISPERSAL = 4      # blk = attempt number
OP_OFFSPRING = 5      # sex, start homologues, recomb keys
OP_DEATH = 6
OP_INIT = 7           # initial positions / sex
OP_MOVE_SURF = 8
OP_DISP_SURF = 9      # blk = attempt number
OP_MATE_PICK = 10     # index draws of the uniform mate choice (32 words + 1 fallback word)


def block_index(step, op, blk=0):
    return (np.uint64(step) * np.uint64(32) + np.uint64(op)) * np.uint64(64) \
        + np.uint64(blk)


def philox4x32(seed, subseq, block):
    """Vectorised Philox4x32-10.

    seed: python int / uint64 scalar; subseq: uint64 array [n];
    block: uint64 scalar or array [n].  Returns uint32 array [n, 4]."""
    subseq = np.atleast_1d(np.asarray(subseq, dtype=np.uint64))
    block = np.broadcast_to(np.asarray(block, dtype=np.uint64), subseq.shape)
    seed = np.uint64(seed)
    k0 = np.uint32(seed & MASK32)
    k1 = np.uint32((seed >> np.uint64(32)) & MASK32)
    c0 = (block & MASK32).astype(np.uint32)
    c1 = (block >> np.uint64(32)).astype(np.uint32)
    c2 = (subseq & MASK32).astype(np.uint32)
    c3 = (subseq >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return np.stack([c0, c1, c2, c3], axis=1)


def u01(x):
    """u32 -> f32 in (0,1): (x >> 8) * 2^-24 + 2^-25 (exact in f32)."""
    return ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
            + np.float32(2.0 ** -25))


def mix64(z):
    """splitmix64 finaliser (uint64 array)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def lowbias32(x):
    """Wellons' 2-multiply 32-bit finaliser (uint32 array)."""
    x = np.asarray(x, dtype=np.uint32).copy()
    with np.errstate(over='ignore'):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7feb352d)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846ca68b)
        x ^= x >> np.uint32(16)
    return x


def ind_tag(seed, step, ids):
    """32-bit mate-choice tag of an individual at a step."""
    with np.errstate(over='ignore'):
        s = mix64(np.uint64(seed) + np.uint64(step) * np.uint64(0x9E3779B97F4A7C15))
        return (mix64(s ^ np.asarray(ids, dtype=np.uint64)) >> np.uint64(32)).astype(np.uint32)


def pair_hash(seed, step, id_a, id_b):
    """Random 32-bit key of the ordered pair (focal a, candidate b) at a step:
    lowbias32((tag_a * 0x9E3779B1) ^ tag_b).  Used to pick a mate uniformly and
    independently of candidate order: the candidate with the smallest key wins
    (ties -> smaller id).  For a fixed focal the candidates' keys are a
    bijection of their iid tags, hence iid uniform."""
    ta = np.atleast_1d(ind_tag(seed, step, id_a))
    tb = np.atleast_1d(ind_tag(seed, step, id_b))
    with np.errstate(over='ignore'):
        return lowbias32((ta * np.uint32(0x9E3779B1)) ^ tb)


def site_hash(seed, site, hom):
    """u32 key for (site, homologue index) used by starting-genome assignment."""
    with np.errstate(over='ignore'):
        s = mix64(np.uint64(seed) ^ np.uint64(0xA0761D6478BD642F))
        h = mix64(s + np.asarray(site, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        h = mix64(h ^ np.asarray(hom, dtype=np.uint64))
    return (h >> np.uint64(32)).astype(np.uint32)
