// soil_math.hpp — numerical contract of the hot path.
//
// The reference leans on two things no other platform can reproduce: CUDA's
// fast intrinsics __expf/__powf (erosion.cu:85,134-136,345-346,500,872;
// graph.cu:139,409-411; filter.cu:48; path.cu:134) and cuRAND XORWOW
// (erosion.cu:57-58, graph.cu:97-101,150).  This header fixes portable
// replacements, written for host and device alike so that host-side setup
// (blur weights, CPU twins) and the kernels agree to the bit:
//
//   soil::expf_  — range-reduced degree-6 polynomial, <= 1 ulp, flush below e^-87
//   soil::log2f_ — mantissa polynomial, positive normal inputs
//   soil::powf_  — exp2(y * log2 x), the definition of __powf
//   soil::Philox — Philox4x32-10, counter = {offset, subsequence}, key = seed
//
// Everything is plain IEEE fp32 evaluated in the written order; the library is
// compiled with -ffp-contract=off so no product-sum is fused behind our back.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SOIL_HD __host__ __device__ __forceinline__

namespace soil {

constexpr float kSqrt2 = 1.41421354f;  // CUDART_SQRT_TWO_F, erosion_map.cu:61
constexpr float kFltMin = 1.17549435e-38f;

SOIL_HD float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
SOIL_HD uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }
SOIL_HD float pow2i(int n) { return bits2f(static_cast<uint32_t>(n + 127) << 23); }  // n in [-126,127]

// y * 2^n for y in [0.7, 1.42].  The contract spells this (y * 2^(n/2)) * 2^(n - n/2):
// the first product is exact, the second rounds the exact value once, which is what
// ldexp does in one instruction (v_ldexp_f32), overflow to +inf included.
SOIL_HD float scale2(float y, int n) { return __builtin_ldexpf(y, n); }

SOIL_HD float exp_poly(float r) {  // ~ exp(r) on |r| <= ln2/2
  float p = 1.9875691500E-4f;
  p = p * r + 1.3981999507E-3f;
  p = p * r + 8.3334519073E-3f;
  p = p * r + 4.1665795894E-2f;
  p = p * r + 1.6666665459E-1f;
  p = p * r + 5.0000001201E-1f;
  return (p * (r * r) + r) + 1.0f;
}

SOIL_HD float expf_(float x) {
  if (x != x) return x;
  if (x > 88.72283f) return __builtin_inff();
  if (x < -87.0f) return 0.0f;
  const float n = __builtin_rintf(x * 1.44269504f);
  float r = x - n * 0.693145752f;
  r = r - n * 1.42860677e-6f;
  const float y = exp_poly(r);
  return scale2(y, static_cast<int>(n));
}
// The same values in straight-line code: the three exits become two selects (a NaN
// runs through the arithmetic as a NaN; what the saturated cases compute is
// discarded).  For call sites where saturation is the exception — three of them per
// fluvial particle step — so the exits never skip anything for a whole wave and only
// cost exec-mask bookkeeping.
__device__ __forceinline__ float expf_flat(float x) {
  const float n = __builtin_rintf(x * 1.44269504f);
  float r = x - n * 0.693145752f;
  r = r - n * 1.42860677e-6f;
  const float y = scale2(exp_poly(r), static_cast<int>(n));
  return (x > 88.72283f) ? __builtin_inff() : ((x < -87.0f) ? 0.0f : y);
}

SOIL_HD float log2f_(float x) {
  if (x != x) return x;
  if (x < 0.0f) return __builtin_nanf("");
  if (x < kFltMin) return -__builtin_inff();
  if (x == __builtin_inff()) return x;
  const uint32_t b = f2bits(x);
  int e = static_cast<int>((b >> 23) & 0xffu) - 127;
  float m = bits2f((b & 0x007fffffu) | 0x3f800000u);
  if (m > kSqrt2) {
    m = m * 0.5f;
    e = e + 1;
  }
  const float f = m - 1.0f;
  const float z = f * f;
  float p = 7.0376836292E-2f;
  p = p * f - 1.1514610310E-1f;
  p = p * f + 1.1676998740E-1f;
  p = p * f - 1.2420140846E-1f;
  p = p * f + 1.4249322787E-1f;
  p = p * f - 1.6668057665E-1f;
  p = p * f + 2.0000714765E-1f;
  p = p * f - 2.4999993993E-1f;
  p = p * f + 3.3333331174E-1f;
  float y = (f * z) * p;
  y = y - 0.5f * z;
  const float ln_m = f + y;
  return ln_m * 1.44269504f + static_cast<float>(e);
}

SOIL_HD float powf_(float x, float y) {
  const float t = y * log2f_(x);
  if (t != t) return t;
  if (t > 128.0f) return __builtin_inff();
  if (t < -126.0f) return 0.0f;
  const float n = __builtin_rintf(t);
  const float r = (t - n) * 0.693147182f;
  const float v = exp_poly(r);
  return scale2(v, static_cast<int>(n));
}

// IEEE quotients that share a denominator (device only).
//
// A particle step (erosion_particles_tiled.hip) divides nine (debris: eleven) times,
// by four (five) different numbers.
// The compiler expands every fp32 `/` into v_div_scale x2, v_rcp, a Newton step on
// the reciprocal, three residual corrections of the quotient (the last one being
// v_div_fmas) and v_div_fixup.  For operands in the plain range below the scale
// instructions are the identity, v_div_fmas is a plain fma and the fix-up passes the
// value through (ISA guide, V_DIV_SCALE/V_DIV_FMAS/V_DIV_FIXUP), so what is left is
//      r = rcp(b); r = fma(fma(-b, r, 1), r, r)
//      q = a * r;  q = fma(fma(-b, q, a), r, q);  q = fma(fma(-b, q, a), r, q)
// — the same instructions on the same values, hence the same bits — and r only
// depends on b.  recip()/quot() spell that sequence out so that r is computed once
// per denominator.  What "plain" has to guarantee (V_DIV_SCALE): b and 1/b normal,
// exponent(a) - exponent(b) < 96, a/b normal, exponent(a) > 23 (|a| >= 2^-103).
// |b| in [2^-40, 2^40] with |a| in [2^-80, 2^50] does; so does any pair whose quotient
// turns out in [2^-60, 2^90] over such a b.  a == +0 runs through the chain to the
// correctly signed zero, -0 needs quot0().  Anything else — denormals, infinities,
// zero or NaN denominators — has to take the written-out `/`.
struct Recip { float b, r; };
__device__ __forceinline__ Recip recip(float b) {
  float r = __builtin_amdgcn_rcpf(b);
  r = __builtin_fmaf(__builtin_fmaf(-b, r, 1.0f), r, r);
  return {b, r};
}
__device__ __forceinline__ float quot(float a, const Recip d) {
  float q = a * d.r;
  q = __builtin_fmaf(__builtin_fmaf(-d.b, q, a), d.r, q);
  return __builtin_fmaf(__builtin_fmaf(-d.b, q, a), d.r, q);
}
// A zero numerator of either sign: the first product already is the IEEE answer (zero,
// sign of a times sign of b); the corrections after it would turn -0 into +0.
__device__ __forceinline__ float quot0(float a, const Recip d) {
  const float q0 = a * d.r, q = quot(a, d);
  return (a == 0.0f) ? q0 : q;
}
constexpr float kDenLo = 0x1p-40f, kDenHi = 0x1p40f;  // denominators
constexpr float kNumLo = 0x1p-60f;                      // numerators (given a plain denominator)
__device__ __forceinline__ bool plain_den(float b) { return fabsf(b) >= kDenLo && fabsf(b) <= kDenHi; }
// +0 or of plain magnitude (the caller bounds it from above)
__device__ __forceinline__ bool plain_num(float a) { return f2bits(a) == 0u || fabsf(a) >= kNumLo; }


// Quotients of a stencil kernel over a denominator that is the same for every cell (a cell size,
// sqrt(2)): quot() with the reciprocal refined once per thread, and the check of the plain range
// made on the RESULTS of a whole group of cells at once — a quotient is beyond doubt when it is
// zero (from a +0 numerator: differences of equal heights; the denominators are positive) or lies
// in [2^-60, 2^90] (see above); Watch collects the extremes of the bit patterns, and a thread whose
// group has a quotient outside (a denormal difference, an infinity, a NaN that is not the
// kernels' own sentinel) redoes the group with the written-out divisions.
struct QuotWatch {
  uint32_t lo = 0xffffffffu, hi = 0u;
  bool neg_zero = false;
  __device__ __forceinline__ float operator()(float q) {
    const uint32_t u = f2bits(q) & 0x7fffffffu;
    lo = min(lo, u - 1u);  // a zero wraps to the top: never the minimum
    hi = max(hi, u);
    return q;
  }
  // For a quotient whose SIGN OF ZERO is seen by the caller (stored, not just compared): a
  // numerator -0 (the difference (-0) - (+0), half of it, ...) has the IEEE quotient -0 over these
  // positive denominators, the chain of quot() ends in +0.  Such a group is in doubt.
  __device__ __forceinline__ float operator()(float q, float numerator) {
    neg_zero = neg_zero || f2bits(numerator) == 0x80000000u;
    return (*this)(q);
  }
  __device__ __forceinline__ bool doubtful() const {
    constexpr uint32_t kLoBits = (127u - 60u) << 23, kHiBits = (127u + 90u) << 23;
    return lo < kLoBits - 1u || hi > kHiBits || neg_zero;
  }
};

// ---- device-only helpers of the particle step ----------------------------------------------

// The attenuation factors of a particle (att_m, att_w, att_v: erosion.cu:134-136; debris att_v :346)
// are __expf in the reference — the hardware's fast exponential, ex2.approx(x * log2e).  They only
// scale what a particle deposits and never feed back into where it goes, so they take gfx950's
// counterpart here: v_exp_f32(x * log2e), 1 ulp, two instructions instead of the ~22 of expf_.
// (Debris' att_d, :345, does feed back into the speed; it stays on expf_ so that trajectories remain
// bit-identical to the oracle's.)  Parity of the flux planes is a tolerance either way: fp32
// summation order (tests/test_gpu_parity.py, _flux_close).
__device__ __forceinline__ float att_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); }

// floor(f) as an integer in one instruction (v_cvt_flr_i32_f32), saturating at the int32 range
// — for a position inside the grid the same cell as the reference's float -> int truncation
// (erosion_map.cu:42-47), and for one outside of it an index an unsigned comparison against the grid
// size rejects (px in (-1, 0) floors to -1; -0.0 to 0, which `px < 0` lets pass as well).  A NaN
// comes out as INT_MAX (measured; CUDA's conversion, which the reference relies on, gives 0): the
// caller has to look at NaN positions itself, nan_cell() below.
__device__ __forceinline__ int floor_cell(float f) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(f));
  return r;
}
__device__ __forceinline__ int nan_cell(float f, int cell) { return (f != f) ? 0 : cell; }

// Correctly rounded square root for x >= 2^-96 (and +0, +inf, NaN): the compiler's own expansion of
// sqrtf — v_sqrt_f32 (1 ulp), then the neighbours one ulp down and up are tried against the exact
// residual x - y'*y — without its pre-scaling of arguments below 2^-96 and without the class test that
// hands zero and infinity through (the residual tests leave those alone by themselves: they compare
// false on NaN).  Below 2^-96 the result is merely close; the particle step only asks whether it is
// under eps = 1e-12 there.
__device__ __forceinline__ float sqrt_rn(float x) {
  const float y = __builtin_amdgcn_sqrtf(x);
  const float dn = bits2f(f2bits(y) - 1u), up = bits2f(f2bits(y) + 1u);
  const float r_dn = __builtin_fmaf(-dn, y, x), r_up = __builtin_fmaf(-up, y, x);
  float s = (r_dn <= 0.0f) ? dn : y;
  s = (r_up > 0.0f) ? up : s;
  return s;
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as
// easy as 1, 2, 3", SC'11).  Only word 0 of the block is consumed per draw.
SOIL_HD uint32_t philox4x32_10_w0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                  uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = static_cast<uint32_t>(p1);
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = static_cast<uint32_t>(p0);
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c0;
}

// ... the whole block
SOIL_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                           uint32_t out[4]) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = static_cast<uint64_t>(0xD2511F53u) * c0;
    const uint64_t p1 = static_cast<uint64_t>(0xCD9E8D57u) * c2;
    const uint32_t n0 = static_cast<uint32_t>(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = static_cast<uint32_t>(p1);
    const uint32_t n2 = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = static_cast<uint32_t>(p0);
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}
SOIL_HD float uniform_of_word(uint32_t r) { return static_cast<float>((r >> 8) + 1u) * 5.9604644775390625e-08f; }  // (0, 1]
// The draws of the four cells 4 q .. 4 q + 3 of a flow graph (random_weighted): block (offset,
// subsequence q), cell n takes word n & 3.  The reference seeds a generator state per cell,
// curand_init(seed, n, offset), for ONE uniform (graph.cu:97-101, :150); Philox makes four words a
// block, so one block serves four cells — a quarter of what was half of the kernel's instructions
// (the generator is build-defined on both sides; oracle: orc_rng_uniform_cell).
SOIL_HD void rng_uniform_quad(uint64_t seed, uint64_t q, uint64_t offset, float u[4]) {
  uint32_t w[4];
  philox4x32_10(static_cast<uint32_t>(offset), static_cast<uint32_t>(offset >> 32), static_cast<uint32_t>(q),
                static_cast<uint32_t>(q >> 32), static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), w);
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = uniform_of_word(w[i]);
}

// One draw in (0, 1] from stream (seed, subsequence) at position `offset` —
// the addressing of curand_init(seed, subsequence, offset) + curand_uniform.
SOIL_HD float rng_uniform_at(uint64_t seed, uint64_t subsequence, uint64_t offset) {
  const uint32_t r = philox4x32_10_w0(
      static_cast<uint32_t>(offset), static_cast<uint32_t>(offset >> 32),
      static_cast<uint32_t>(subsequence), static_cast<uint32_t>(subsequence >> 32),
      static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
  return static_cast<float>((r >> 8) + 1u) * 5.9604644775390625e-08f;  // 2^-24
}

}  // namespace soil
