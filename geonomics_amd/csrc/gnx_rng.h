// Device random numbers for the Geonomics hot path.
//
// The reference draws everything from ONE global MT19937 stream in
// data-dependent order (sim/model.py:364-366); that is not reproducible on a
// GPU.  Here every draw comes from rocRAND's counter-based Philox4x32-10
// device generator, addressed by (seed, subsequence = individual id,
// block = ((step*32 + op)*64 + blk)), so a draw depends only on WHO draws, WHEN
// and FOR WHAT - not on slot order, launch geometry or GPU count.
// oracle/philox.py restates the identical generator in numpy.
#pragma once
#include <hip/hip_runtime.h>
#define ROCRAND_DETAIL_BM_NOT_IN_STATE
#include <rocrand/rocrand_philox4x32_10.h>

// op codes (shared with oracle/philox.py)
enum {
  OP_MOVE_DIR = 0, OP_MOVE_DIST = 1, OP_PAIR_KEEP = 2, OP_BIRTHS = 3,
  OP_DISPERSAL = 4, OP_OFFSPRING = 5, OP_DEATH = 6, OP_INIT = 7,
  OP_MOVE_SURF = 8, OP_DISP_SURF = 9, OP_MATE_PICK = 10
};

#define GNX_PI_F 3.14159274101257324f
#define GNX_PI_D 3.14159265358979323846

__device__ __forceinline__ unsigned long long gnx_block(long long step, int op, int blk) {
  return ((unsigned long long)step * 32ull + (unsigned long long)op) * 64ull +
         (unsigned long long)blk;
}

// one Philox block (4 x u32) of stream (subseq, step, op, blk)
__device__ __forceinline__ uint4 gnx_rand4(unsigned long long seed, unsigned long long subseq,
                                           long long step, int op, int blk) {
  rocrand_state_philox4x32_10 st;
  rocrand_init(seed, subseq, gnx_block(step, op, blk) * 4ull, &st);
  return rocrand4(&st);
}

// A small cursor over the 64 blocks of one (subseq, step, op) stream.
struct GnxStream {
  unsigned long long seed, subseq;
  long long step;
  int op;
  int blk;      // next block to fetch
  int pos;      // position inside cur (4 = empty)
  uint4 cur;
  __device__ __forceinline__ GnxStream(unsigned long long s, unsigned long long q, long long t,
                                       int o, int first_blk = 0)
      : seed(s), subseq(q), step(t), op(o), blk(first_blk), pos(4) {}
  __device__ __forceinline__ unsigned int next() {
    if (pos == 4) {
      cur = gnx_rand4(seed, subseq, step, op, blk);
      blk++;
      pos = 0;
    }
    unsigned int r = pos == 0 ? cur.x : pos == 1 ? cur.y : pos == 2 ? cur.z : cur.w;
    pos++;
    return r;
  }
};

// u32 -> f32 in (0,1): (x >> 8) * 2^-24 + 2^-25 (exact in f32; oracle: u01)
__device__ __forceinline__ float gnx_u01(unsigned int x) {
  return (float)(x >> 8) * 5.9604644775390625e-08f + 2.98023223876953125e-08f;
}

// standard normal by Box-Muller from two uniforms
__device__ __forceinline__ float gnx_normal(float u0, float u1) {
  return sqrtf(-2.0f * logf(u0)) * cosf(2.0f * GNX_PI_F * u1);
}

// von Mises(mu, kappa): numpy's legacy algorithm (Best & Fisher 1979), the
// sampler behind np.random.vonmises (ops/movement.py:55) and
// scipy.stats.vonmises.rvs (utils/spatial.py:383,421).  Consumes from `s`:
// one uniform when kappa < 1e-8, else (U,V) per rejection round, bounded at
// 14 rounds, plus one for the sign.
__device__ __forceinline__ float gnx_vonmises(GnxStream& s, float mu, float kappa) {
  if (kappa < 1e-8f) return GNX_PI_F * (2.0f * gnx_u01(s.next()) - 1.0f);
  float sv;
  if (kappa < 1e-5f) {
    sv = 1.0f / kappa + kappa;
  } else {
    float r = 1.0f + sqrtf(1.0f + 4.0f * kappa * kappa);
    float rho = (r - sqrtf(2.0f * r)) / (2.0f * kappa);
    sv = (1.0f + rho * rho) / (2.0f * rho);
  }
  float Wv = 1.0f;
  for (int it = 0; it < 14; ++it) {
    float U = gnx_u01(s.next());
    float V = gnx_u01(s.next());
    float Z = cosf(GNX_PI_F * U);
    Wv = (1.0f + sv * Z) / (sv + Z);
    float Y = kappa * (sv - Wv);
    if ((Y * (2.0f - Y) - V >= 0.0f) || (logf(Y / V) + 1.0f - Y >= 0.0f)) break;
  }
  float U = gnx_u01(s.next());
  Wv = fminf(fmaxf(Wv, -1.0f), 1.0f);
  float res = acosf(Wv);
  if (U < 0.5f) res = -res;
  res += mu;
  bool neg = res < 0.0f;
  float m = fabsf(res);
  // (fmodf(m + pi, 2 pi) - pi: m + pi < 4 pi unless |mu| is huge, and t - 2 pi is exact for t in
  // [2 pi, 4 pi) (Sterbenz), i.e. the same bits as fmodf without its division loop)
  float t = m + GNX_PI_F;
  const float two_pi = 2.0f * GNX_PI_F;
  if (t >= 2.0f * two_pi) t = fmodf(t, two_pi);
  else if (t >= two_pi) t -= two_pi;
  m = t - GNX_PI_F;
  return neg ? -m : m;
}

// distance draw: lognormal / wald / levy (ops/movement.py:61-72, 112-120)
__device__ __forceinline__ float gnx_distance(int distr, float p1, float p2, uint4 r) {
  float zn = gnx_normal(gnx_u01(r.x), gnx_u01(r.y));
  if (distr == 0) {                        // lognormal(mean=p1, sigma=p2)
    return expf(p1 + p2 * zn);
  } else if (distr == 1) {                 // wald(mean=p1, scale=p2), numpy legacy_wald
    float mu_2l = p1 / (2.0f * p2);
    float Y = p1 * zn * zn;
    float X = p1 + mu_2l * (Y - sqrtf(4.0f * p2 * Y + Y * Y));
    float U = gnx_u01(r.z);
    return (U <= p1 / (p1 + X)) ? X : p1 * p1 / X;
  } else {                                 // levy(loc=p1, scale=p2) = loc + scale / Z^2
    return p1 + p2 / (zn * zn);
  }
}

// splitmix64 finaliser and the integer hashes (oracle/philox.py: mix64,
// pair_hash, site_hash)
__device__ __host__ __forceinline__ unsigned long long gnx_mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __host__ __forceinline__ unsigned long long gnx_pair_seed(unsigned long long seed,
                                                                     long long step) {
  return gnx_mix64(seed + (unsigned long long)step * 0x9E3779B97F4A7C15ull);
}
// Mate choice keys.  Every individual gets a 32-bit tag per step,
// tag = high32(mix64(pair_seed ^ id)); the key of the ordered pair (focal f,
// candidate c) is lowbias32((tag_f * 0x9E3779B1) ^ tag_c) (lowbias32: Wellons'
// 2-multiply 32-bit finaliser).  For a fixed focal the keys of its candidates
// are a bijection of iid uniform tags, hence iid uniform: the candidate with the
// smallest key is a uniform draw, independent of candidate order.
__device__ __host__ __forceinline__ unsigned int gnx_lowbias32(unsigned int x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned int gnx_ind_tag(unsigned long long pair_seed,
                                                    unsigned long long id) {
  return (unsigned int)(gnx_mix64(pair_seed ^ id) >> 32);
}
__device__ __forceinline__ unsigned int gnx_pair_key(unsigned int tag_f_mul, unsigned int tag_c) {
  return gnx_lowbias32(tag_f_mul ^ tag_c);
}
__device__ __host__ __forceinline__ unsigned long long gnx_site_seed(unsigned long long seed) {
  return gnx_mix64(seed ^ 0xA0761D6478BD642Full);
}
__device__ __forceinline__ unsigned int gnx_site_hash(unsigned long long site_seed,
                                                      unsigned long long site,
                                                      unsigned long long hom) {
  unsigned long long h = gnx_mix64(site_seed + site * 0x9E3779B97F4A7C15ull);
  h = gnx_mix64(h ^ hom);
  return (unsigned int)(h >> 32);
}
