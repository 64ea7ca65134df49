// graph.hip — flow-graph kernels: steepest / direction / random_weighted / slope
// and rake-compress upstream accumulation (graph.hpp:49-63, graph.cu).
//
// Index arithmetic is 64-bit: K*n overflows int32 at 16384^2 with D8
// (graph.cu:345 uses int).  Scratch for accumulate lives in a per-device
// workspace that is grown on demand and reused, instead of the reference's
// eight cudaMallocs per call (graph.cu:539-550).
#include <mutex>
#include <unordered_map>

#include "common.hpp"
#include "window.hpp"

namespace soil {

constexpr int kGBlock = 256;

// D4_t / D8_t neighbour tables, graph.hpp:21-46 (first four entries = D4)
__device__ __constant__ int kShiftX[8] = {-1, 0, 0, 1, -1, -1, 1, 1};
__device__ __constant__ int kShiftY[8] = {0, -1, 1, 0, -1, 1, -1, 1};

// the same tables as compile-time constants for fully unrolled loops; __length(shift)
// (graph.cu:23-25) is sqrtf(1) or sqrtf(2), correctly rounded
constexpr int kDX[8] = {-1, 0, 0, 1, -1, -1, 1, 1};
constexpr int kDY[8] = {0, -1, 1, 0, -1, 1, -1, 1};
constexpr float kShiftLen[8] = {1.0f, 1.0f, 1.0f, 1.0f, kSqrt2, kSqrt2, kSqrt2, kSqrt2};

__device__ __forceinline__ float shift_len(int k) {  // __length(shift), graph.cu:23-25
  const float dx = static_cast<float>(kShiftX[k]), dy = static_cast<float>(kShiftY[k]);
  return sqrtf(dx * dx + dy * dy);
}

// __steepest (graph.cu:27-70) / __direction (:201-243)
template <int K, bool STORE_K>
__global__ void __launch_bounds__(kGBlock)
    k_steepest(int32_t* __restrict__ out, const float* __restrict__ height, int64_t H, int64_t W) {
  const int64_t y = static_cast<int64_t>(blockIdx.x) * kGBlock + threadIdx.x;
  if (y >= W) return;
  SOIL_ROW_LOOP(x, H) {
    const int64_t n = x * W + y;
    const float hlocal = height[n];  // :40
    float smax = 0.0f;               // :42
    int32_t next = -1;               // :43
#pragma unroll
    for (int k = 0; k < K; ++k) {  // :46
      const int64_t nx = x + kDX[k], ny = y + kDY[k];
      if (nx < 0 || ny < 0 || nx >= H || ny >= W) continue;  // :51-52
      const int64_t nind = nx * W + ny;
      const float scur = (hlocal - height[nind]) / kShiftLen[k];  // :56
      if (scur > smax) {                                          // :57-60
        smax = scur;
        next = STORE_K ? static_cast<int32_t>(k) : static_cast<int32_t>(nind);
      }
    }
    out[n] = next;  // :68
  }
}

// __steepest / __direction with four cells per thread (window.hpp); the scalar kernel above
// serves widths that are not a multiple of four.
//
// steepest_group: the loop as written, the diagonal slopes as IEEE quotients over sqrt(2).
// steepest_group_fast (round 4: the kernel is bound by the issue of vector instructions — 79 per
// cell, 0.75-0.83 of the SIMDs' cycles, profiles/r04_stencils — not by memory): the same receiver with
// ONE quotient per cell instead of four.  The four straight neighbours come first in the loop and
// their slopes are the differences themselves; of the diagonal ones only the winner matters, and
// d -> RN(d / sqrt 2) is monotone: the largest quotient is that of the largest difference `dmax`,
// and the loop's choice among the diagonals is the FIRST k whose quotient equals it.  That is the
// first k with d_k == dmax unless a smaller difference rounds to the same quotient — possible only
// within 1.42 ulp of dmax (the quotient's rounding interval, times sqrt 2); a wave that holds such a
// near-tie (bit distance 1 .. 4; none on a float terrain, and equal differences are not near-ties)
// takes the loop as written, as does one whose window is not plain (window.hpp: the quotient of
// `dmax` then is the IEEE one).  The winner replaces the straight maximum iff its quotient is
// greater (:57), and the receiver is -1 iff no slope was positive (:42-43).
template <int K, bool STORE_K, class Walk>
__device__ __forceinline__ void steepest_group(int32_t oi[4], const Walk& w, const WinThread& t,
                                               int64_t x, int64_t W) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float hlocal = w.mid.v[c + 1];  // :40
    float smax = 0.0f;                    // :42
    int32_t next = -1;                    // :43
    const int64_t y = t.y0 + c;
#pragma unroll
    for (int k = 0; k < K; ++k) {  // :46
      const bool row_ok = kDX[k] < 0 ? w.has_up : (kDX[k] > 0 ? w.has_dn : true);
      const bool col_ok = kDY[k] < 0 ? (c > 0 || t.y0 > 0) : (kDY[k] > 0 ? (c < 3 || t.y0 + 4 < W) : true);
      if (!row_ok || !col_ok) continue;  // :51-52
      const Row6& r = kDX[k] < 0 ? w.up : (kDX[k] > 0 ? w.dn : w.mid);
      const float scur = (hlocal - r.v[c + 1 + kDY[k]]) / kShiftLen[k];  // :56
      if (scur > smax) {  // :57-60
        smax = scur;
        next = STORE_K ? static_cast<int32_t>(k)
                       : static_cast<int32_t>((x + kDX[k]) * W + (y + kDY[k]));
      }
    }
    oi[c] = next;  // :68
  }
}

// returns false (wave-uniform) when the wave has to take steepest_group instead
template <int K, bool STORE_K, class Walk>
__device__ __forceinline__ bool steepest_group_fast(int32_t oi[4], const Walk& w, const WinThread& t,
                                                    int64_t x, int64_t W, const Recip& rdiag) {
  const int32_t iW = static_cast<int32_t>(W);
  const int32_t n0 = static_cast<int32_t>(x * W + t.y0);
  bool tie = false;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float hlocal = w.mid.v[c + 1];
    float smax = 0.0f;
    int32_t pick = 0;  // k, or the receiver's index minus the cell's; looked at only if smax > 0
    float dmax = 0.0f;
    int32_t dpick = 0;
    // (k, row, column and "exists" are compile-time / wave-uniform per k; the differences are
    // taken again for the near-tie test instead of being kept: registers)
    auto exists = [&](int k) {
      const bool row_ok = kDX[k] < 0 ? w.has_up : (kDX[k] > 0 ? w.has_dn : true);
      const bool col_ok = kDY[k] < 0 ? (c > 0 || t.y0 > 0) : (kDY[k] > 0 ? (c < 3 || t.y0 + 4 < W) : true);
      return row_ok && col_ok;
    };
    auto diff = [&](int k) {
      const Row6& r = kDX[k] < 0 ? w.up : (kDX[k] > 0 ? w.dn : w.mid);
      return hlocal - r.v[c + 1 + kDY[k]];
    };
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int32_t id = STORE_K ? static_cast<int32_t>(k) : kDX[k] * iW + kDY[k];
      const float d = diff(k);
      if (k < 4) {
        if (exists(k) && d > smax) {
          smax = d;
          pick = id;
        }
      } else if (exists(k) && d > dmax) {
        dmax = d;
        dpick = id;
      }
    }
    if (K > 4) {
      const uint32_t below = f2bits(dmax) - 1u;  // a difference 1 .. 4 bit patterns under dmax: a near-tie
      uint32_t nearest = 0xffffffffu;
#pragma unroll
      for (int k = 4; k < K; ++k)
        if (exists(k)) nearest = min(nearest, below - f2bits(diff(k)));
      tie = tie || nearest < 4u;
      const float q = quot(dmax, rdiag);
      if (q > smax) {
        smax = q;
        pick = dpick;
      }
    }
    oi[c] = smax > 0.0f ? (STORE_K ? pick : n0 + c + pick) : -1;
    __builtin_amdgcn_sched_barrier(0);  // cell by cell: interleaving the four costs registers (8 waves: 64)
  }
  return __ballot(tie) == 0ull;
}

template <int K, bool STORE_K, class Walk>
__global__ void __launch_bounds__(kWinBlock) __attribute__((amdgpu_waves_per_eu(8, 8)))
    k_steepest4(int32_t* __restrict__ out, const float* __restrict__ height, int64_t H, int64_t W) {
  const WinThread t = Walk::thread(H, W);
  const Recip rdiag = recip(kSqrt2);
  SOIL_WIN_WALK(Walk, w, W);
  SOIL_WIN_ROWS(x, w, height, H, W, t) {
    int4 o;
    int32_t* oi = reinterpret_cast<int32_t*>(&o);
    if (!w.plain() || !steepest_group_fast<K, STORE_K>(oi, w, t, x, W, rdiag))
      steepest_group<K, STORE_K>(oi, w, t, x, W);
    if (t.live) *reinterpret_cast<int4*>(out + x * W + t.y0) = o;
  }
}

template <int K, bool STORE_K, class Walk>
void launch_steepest4_as(int32_t* out, const float* height, int64_t H, int64_t W, hipStream_t st) {
  k_steepest4<K, STORE_K, Walk><<<Walk::grid(H, W), kWinBlock, 0, st>>>(out, height, H, W);
}
template <int K, bool STORE_K>
void launch_steepest4(int32_t* out, const float* height, int64_t H, int64_t W, hipStream_t st) {
  constexpr bool kWatch = K > 4;  // the straight slopes are the differences: nothing to watch for d4
  // (steepest D8 on blocks of four rows spills its 64 registers; direction does not)
  switch (win_shape_for(STORE_K ? 4 : 0, STORE_K ? 4 : 5, H, W)) {
    case 0: return launch_steepest4_as<K, STORE_K, RowWalkReg<kWatch>>(out, height, H, W, st);
    case 1: return launch_steepest4_as<K, STORE_K, RowWalkLds<kWatch>>(out, height, H, W, st);
    case 3: return launch_steepest4_as<K, STORE_K, RowWalkTall<kWatch>>(out, height, H, W, st);
    case 4: return launch_steepest4_as<K, STORE_K, RowWalkBlock4<kWatch>>(out, height, H, W, st);
    case 5: return launch_steepest4_as<K, STORE_K, RowWalkBlock2<kWatch>>(out, height, H, W, st);
    case 6: return launch_steepest4_as<K, STORE_K, RowWalkShort<kWatch>>(out, height, H, W, st);
    case 7: return launch_steepest4_as<K, STORE_K, RowWalkStack2<kWatch>>(out, height, H, W, st);
    case 8: return launch_steepest4_as<K, STORE_K, RowWalkStack4<kWatch>>(out, height, H, W, st);
    default: return launch_steepest4_as<K, STORE_K, RowWalkFlat<kWatch>>(out, height, H, W, st);
  }
}

// __seed (graph.cu:97-101) + __random_weighted (:103-173): the per-cell
// generator state is never materialised — cell n reads its one uniform straight
// from block (seed; offset, subsequence n >> 2), word n & 3 (round 5: one Philox block per FOUR cells,
// soil_math.hpp rng_uniform_quad; rounds 1-4 spent a block per cell on its word 0).
//
// Arithmetic (round 4).  The reference computes the Gibbs weights with the fast intrinsic,
// `P = __expf(dE / T)` (graph.cu:139) = ex2.approx(dE / T * log2 e): they are a tolerance by
// construction, and so is every receiver whose draw lies within the weights' error
// of a CDF edge.  The kernel therefore takes gfx950's counterpart of that intrinsic and nothing
// dearer: one multiplication by the host-made constant log2(e) / (|shift| T) and one v_exp_f32
// per neighbour, and the inverse-CDF test `u < CDF[k] / Z` (:160) as `u Z < CDF[k]` — no division.
// (Round 3 kept the oracle's bits here: eight software exponentials and twenty IEEE divisions per
// cell, ~570 vector instructions, 0.98 ms at 8192^2 = 7 % of the HBM roofline.)  The oracle keeps the
// exact statement (expf_, IEEE divisions); the parity tests count the receivers that differ, bound
// them (a few per million) and check that each of them sits on a CDF edge
// (tests/test_gpu_parity.py::test_random_weighted_against_the_oracle).
struct RwConst {  // log2(e) / (|shift_k| T) for the straight and the diagonal neighbours
  float straight, diagonal;
};
inline RwConst rw_const(float T) {
  const double log2e = 1.4426950408889634;
  return RwConst{static_cast<float>(log2e / static_cast<double>(T)),
                 static_cast<float>(log2e / (static_cast<double>(kSqrt2) * static_cast<double>(T)))};
}
// cumulative weights of a cell (:126-143): `hn[k]`, `ok[k]`: neighbour k's height, and whether it
// lies in the grid.  A neighbour outside leaves CDF[k] at the running sum; it is never looked at.
template <int K>
__device__ __forceinline__ float rw_cdf(float CDF[K], float hlocal, const float hn[K], const bool ok[K],
                                        RwConst rc) {
  float Z = 0.0f;  // :127
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float diff = hlocal - hn[k];  // :138 (the division by |shift| is part of the constant)
    // :139 — `dE <= 0 ? 0 : exp(dE / T)`, NaN heights included (NaN <= 0 is false: the weight is NaN)
    float P = (diff <= 0.0f) ? 0.0f : __builtin_amdgcn_exp2f(diff * (k < 4 ? rc.straight : rc.diagonal));
    if (!ok[k]) P = 0.0f;
    CDF[k] = Z + P;  // :140
    Z += P;          // :141
  }
  return Z;
}
// the receiver a draw picks (:149-171): the first neighbour in the grid with u < CDF[k] / Z
template <int K>
__device__ __forceinline__ int32_t rw_pick(const float CDF[K], float Z, const bool ok[K], const int32_t to[K],
                                           float uniform) {
  const float uz = uniform * Z;  // (Z == 0: 0 < 0 is false like NaN < x; Z = inf: inf < inf is)
  int32_t next = -1;             // :149
#pragma unroll
  for (int k = K - 1; k >= 0; --k)  // the lowest k that passes wins
    if (ok[k] && uz < CDF[k]) next = to[k];
  return next;
}

constexpr int kRwBatch = 4;
struct RwBatch {  // up to kRwBatch realisations per pass over the heights (soil_multiflow): the
  int32_t* graph[kRwBatch];  // cumulative weights of a cell do not depend on the draw
  uint64_t offset[kRwBatch];
  int n;
};

// one thread per cell (any width)
template <int K>
__global__ void __launch_bounds__(kGBlock)
    k_random_weighted(RwBatch b, const float* __restrict__ height, int64_t H, int64_t W,
                      uint64_t seed, RwConst rc) {
  const int64_t y = static_cast<int64_t>(blockIdx.x) * kGBlock + threadIdx.x;
  if (y >= W) return;
  SOIL_ROW_LOOP(x, H) {
    const int64_t n = x * W + y;
    const float hlocal = height[n];  // :118
    float hn[K], CDF[K];
    bool ok[K];
    int32_t to[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {  // :129-137
      const int64_t nx = x + kDX[k], ny = y + kDY[k];
      ok[k] = !(nx < 0 || ny < 0 || nx >= H || ny >= W);
      to[k] = static_cast<int32_t>(nx * W + ny);
      hn[k] = ok[k] ? height[nx * W + ny] : 0.0f;
    }
    const float Z = rw_cdf<K>(CDF, hlocal, hn, ok, rc);
#pragma unroll
    for (int m = 0; m < kRwBatch; ++m) {
      if (m >= b.n) break;
      float u4[4];  // :100, :150 — the block of cells 4 (n >> 2) .. + 3, this cell's word (soil_math.hpp)
      rng_uniform_quad(seed, static_cast<uint64_t>(n) >> 2, b.offset[m], u4);
      const int word = static_cast<int>(n & 3);
      const float uniform = word == 0 ? u4[0] : (word == 1 ? u4[1] : (word == 2 ? u4[2] : u4[3]));
      b.graph[m][n] = rw_pick<K>(CDF, Z, ok, to, uniform);  // :171
    }
  }
}

// four cells per thread, the neighbours from the thread's three-row window (window.hpp): the same
// operations on the same values as the kernel above
template <int K, class Walk>
__global__ void __launch_bounds__(kWinBlock)
    k_random_weighted4(RwBatch b, const float* __restrict__ height, int64_t H, int64_t W, uint64_t seed,
                       RwConst rc) {
  const WinThread t = Walk::thread(H, W);
  SOIL_WIN_WALK(Walk, w, W);
  SOIL_WIN_ROWS(x, w, height, H, W, t) {
    int4 o[kRwBatch];
    // the thread's four cells are one block's four words: W and y0 are multiples of four
    float u[kRwBatch][4];
#pragma unroll
    for (int m = 0; m < kRwBatch; ++m) {
      if (m >= b.n) break;
      rng_uniform_quad(seed, static_cast<uint64_t>(x * W + t.y0) >> 2, b.offset[m], u[m]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t y = t.y0 + c;
      const int32_t n = static_cast<int32_t>(x * W + y);
      float hn[K], CDF[K];
      bool ok[K];
      int32_t to[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const bool row_ok = kDX[k] < 0 ? w.has_up : (kDX[k] > 0 ? w.has_dn : true);
        const bool col_ok = kDY[k] < 0 ? (c > 0 || t.y0 > 0) : (kDY[k] > 0 ? (c < 3 || t.y0 + 4 < W) : true);
        const Row6& r = kDX[k] < 0 ? w.up : (kDX[k] > 0 ? w.dn : w.mid);
        ok[k] = row_ok && col_ok;
        hn[k] = r.v[c + 1 + kDY[k]];
        to[k] = n + kDX[k] * static_cast<int32_t>(W) + kDY[k];
      }
      const float Z = rw_cdf<K>(CDF, w.mid.v[c + 1], hn, ok, rc);
#pragma unroll
      for (int m = 0; m < kRwBatch; ++m) {
        if (m >= b.n) break;
        reinterpret_cast<int32_t*>(&o[m])[c] = rw_pick<K>(CDF, Z, ok, to, u[m][c]);
      }
    }
    if (t.live) {
#pragma unroll
      for (int m = 0; m < kRwBatch; ++m) {
        if (m >= b.n) break;
        *reinterpret_cast<int4*>(b.graph[m] + x * W + t.y0) = o[m];
      }
    }
  }
}

template <int K>
static void launch_random_weighted(const RwBatch& b, const float* height, int64_t H, int64_t W, uint64_t seed,
                                   float T, hipStream_t st) {
  bool aligned = W % 4 == 0 && W >= 4 && (reinterpret_cast<uintptr_t>(height) & 15) == 0;
  for (int m = 0; m < b.n; ++m) aligned = aligned && (reinterpret_cast<uintptr_t>(b.graph[m]) & 15) == 0;
  if (aligned) {
    const int shape = win_shape_for(4, 4, H, W);
    if (shape == 4) k_random_weighted4<K, RowWalkBlock4<false>><<<RowWalkBlock4<false>::grid(H, W), kWinBlock, 0, st>>>(b, height, H, W, seed, rw_const(T));
    else if (shape == 5) k_random_weighted4<K, RowWalkBlock2<false>><<<RowWalkBlock2<false>::grid(H, W), kWinBlock, 0, st>>>(b, height, H, W, seed, rw_const(T));
    else if (shape == 7) k_random_weighted4<K, RowWalkStack2<false>><<<RowWalkStack2<false>::grid(H, W), kWinBlock, 0, st>>>(b, height, H, W, seed, rw_const(T));
    else if (shape == 8) k_random_weighted4<K, RowWalkStack4<false>><<<RowWalkStack4<false>::grid(H, W), kWinBlock, 0, st>>>(b, height, H, W, seed, rw_const(T));
    else k_random_weighted4<K, RowWalk><<<win_grid(H, W), kWinBlock, 0, st>>>(b, height, H, W, seed, rw_const(T));
  }
  else
    k_random_weighted<K><<<grid_rows(H, W, kGBlock), kGBlock, 0, st>>>(b, height, H, W, seed, rw_const(T));
}

// __slope, graph.cu:270-295.  Threads along the row, a work-group walks a band of rows: the
// cell's own coordinates cost nothing, and the receiver's — `next / W`, `next % W`, 64-bit
// divisions that were most of this kernel's instructions — come from the index difference when the
// receiver is one of the eight neighbours (every graph steepest / direction / random_weighted
// make); any other index takes the divisions.
// 32-bit index arithmetic throughout (the flow graph holds int32 cell indices, so H * W < 2^31):
// with int64 rows and columns the kernel was ~150 instructions per cell, most of them the two
// halves of 64-bit adds, multiplies and conversions — instruction-bound at 3.5 TB/s.
__global__ void __launch_bounds__(kGBlock)
    k_slope(float* __restrict__ slope, const float* __restrict__ tensor,
            const int32_t* __restrict__ flow, int32_t H, int32_t W, Scale2 s) {
  const int32_t y = static_cast<int32_t>(blockIdx.x) * kGBlock + static_cast<int32_t>(threadIdx.x);
  if (y >= W) return;
  const float iy = static_cast<float>(y);
  for (int32_t band = static_cast<int32_t>(blockIdx.y); band * kRowBand < H; band += static_cast<int32_t>(gridDim.y)) {
    const int32_t x_end = (band * kRowBand + kRowBand < H) ? band * kRowBand + kRowBand : H;
    for (int32_t x = band * kRowBand; x < x_end; ++x) {
      const int32_t n = x * W + y;
      const int32_t next = flow[n];  // :282
      if (next < 0 || next == n) {   // :283-286
        slope[n] = 0.0f;
        continue;
      }
      // row and column of the receiver relative to the cell: d = rd * W + cd with rd, cd in -1..1
      const int32_t d = next - n;
      const int32_t rd = d > 1 ? 1 : (d < -1 ? -1 : 0);
      const int32_t cd = d - rd * W;
      int32_t qx = x + rd, qy = y + cd;
      if (cd < -1 || cd > 1 || qy < 0 || qy >= W) {  // not a neighbour (or W < 3): the general case
        qx = next / W;
        qy = next % W;
      }
      const float ix = static_cast<float>(x);                                // :288
      const float nx = static_cast<float>(qx), ny = static_cast<float>(qy);  // :289
      const float ival = tensor[n];                                          // :291
      const float nval = tensor[next];                                       // :292
      const float dx = s.x * (nx - ix), dy = s.y * (ny - iy);
      slope[n] = (nval - ival) / sqrtf(dx * dx + dy * dy);  // :293
    }
  }
}

// __slope with four cells per thread (window.hpp): the receiver of a cell is one of its eight
// neighbours in every graph this library makes, and then its value is already in the thread's
// three-row window — no gather; any other index takes the scalar kernel's general case.  16-byte
// loads of the flow graph and the tensor, 16-byte stores.
template <class Walk>
__global__ void __launch_bounds__(kWinBlock)
    k_slope4(float* __restrict__ slope, const float* __restrict__ tensor,
             const int32_t* __restrict__ flow, int64_t H, int64_t W, Scale2 s) {
  const WinThread t = Walk::thread(H, W);
  const int32_t iW = static_cast<int32_t>(W), iH = static_cast<int32_t>(H);
  (void)iH;
  SOIL_WIN_WALK(Walk, w, W);
  SOIL_WIN_ROWS(x, w, tensor, H, W, t) {
    const int32_t n0 = static_cast<int32_t>(x * W + t.y0);
    const int4 f = *reinterpret_cast<const int4*>(flow + n0);  // :282
    const int32_t fi[4] = {f.x, f.y, f.z, f.w};
    float4 o;
    float* of = reinterpret_cast<float*>(&o);
    const float ix = static_cast<float>(static_cast<int32_t>(x));  // :288
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int32_t n = n0 + c, next = fi[c], y = static_cast<int32_t>(t.y0) + c;
      const float ival = w.mid.v[c + 1];  // :291
      float res = 0.0f;                   // :283-286
      if (!(next < 0 || next == n)) {
        const int32_t d = next - n;
        const int32_t rd = d > 1 ? 1 : (d < -1 ? -1 : 0);
        const int32_t cd = d - rd * iW;
        int32_t qx = static_cast<int32_t>(x) + rd, qy = y + cd;
        float nval;
        if (cd < -1 || cd > 1 || qy < 0 || qy >= iW) {  // not a neighbour (or W < 3): the general case
          qx = next / iW;
          qy = next % iW;
          nval = tensor[next];  // :292
        } else {                // column c + 1 + cd of row rd of the window
          // (values picked one by one: a row picked as a whole lives in scratch memory)
          const float u = cd < 0 ? w.up.v[c] : (cd > 0 ? w.up.v[c + 2] : w.up.v[c + 1]);
          const float m = cd < 0 ? w.mid.v[c] : (cd > 0 ? w.mid.v[c + 2] : w.mid.v[c + 1]);
          const float dn = cd < 0 ? w.dn.v[c] : (cd > 0 ? w.dn.v[c + 2] : w.dn.v[c + 1]);
          nval = rd < 0 ? u : (rd > 0 ? dn : m);
        }
        const float iy = static_cast<float>(y);
        const float nx = static_cast<float>(qx), ny = static_cast<float>(qy);  // :289
        const float dx = s.x * (nx - ix), dy = s.y * (ny - iy);
        res = (nval - ival) / sqrtf(dx * dx + dy * dy);  // :293
      }
      of[c] = res;
    }
    if (t.live) *reinterpret_cast<float4*>(slope + x * W + t.y0) = o;
  }
}

// ---- accumulate --------------------------------------------------------------

// donor / decay slot k of cell n lives at [k * elem + n] (the reference interleaves the
// K slots of a cell, graph.cu:448-520): neighbouring threads then touch neighbouring
// words, and a cell with one pending donor moves 4 bytes per array instead of a
// 32-byte sector
struct Acc {  // acc_t, graph.cu:422-427
  int32_t* donor;
  int32_t* count;
  float* value;
  float* decay;
};

// __donor (graph.cu:321-348) + __count (:350-380) + my_decay (:382-420) in one pass
// from the receiver's side.  The reference lets every donor n write itself into
// slot k of its receiver, k = the direction with n + shift[k] == receiver, then
// compacts the K slots in that order and assigns the per-edge decay (diagonal
// exponent by COMPACTED slot index k >= 4).  Here cell n
// asks its K neighbours n - shift[k] whether they drain into it: same slots, same
// order, no -1 fill of the K*elem slot array, no scatter, and value = source rides
// along.
template <int K, bool TENSOR_DECAY>
__global__ void __launch_bounds__(kGBlock)
    k_donors(int32_t* __restrict__ count, int32_t* __restrict__ donor, float* __restrict__ decay,
             float* __restrict__ value, const int32_t* __restrict__ graph,
             const float* __restrict__ source, const float* __restrict__ decayIn, int64_t H,
             int64_t W) {
  const int64_t y = static_cast<int64_t>(blockIdx.x) * kGBlock + threadIdx.x;
  const int64_t elem = H * W;
  if (y >= W) return;
  SOIL_ROW_LOOP(x, H) {
    const int64_t n = x * W + y;
    int c = 0;
    int32_t dn[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int64_t dx = x - kDX[k], dy = y - kDY[k];  // the cell whose k-th neighbour is n
      if (dx < 0 || dy < 0 || dx >= H || dy >= W) continue;
      const int64_t d = dx * W + dy;
      if (graph[d] == n) dn[c++] = static_cast<int32_t>(d);  // :333-345
    }
    count[n] = c;
    value[n] = source[n];  // silt::set(value, source), :553
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k < c) {
        donor[k * elem + n] = dn[k];
        if (TENSOR_DECAY) {  // (without a decay tensor every edge's decay is 1: no decay array at all, k_rake_compress)
          const float D = decayIn[dn[k]];
          decay[k * elem + n] = (k < 4) ? D : powf_(D, 1.414f);
        }
      }
    }
  }
}

// The same with four cells per thread (window.hpp): the graph's three rows in registers — the
// eight neighbours a cell asks are compile-time positions of the window, no gathers and no 64-bit
// index arithmetic per neighbour —, count and value as 16-byte stores.  (The window moves the
// int32 indices as float bit patterns: loads, shuffles and selects do not touch them.)
template <int K, bool TENSOR_DECAY, class Walk>
__global__ void __launch_bounds__(kWinBlock)
    k_donors4(int32_t* __restrict__ count, int32_t* __restrict__ donor, float* __restrict__ decay,
              float* __restrict__ value, const int32_t* __restrict__ graph,
              const float* __restrict__ source, const float* __restrict__ decayIn, int64_t H,
              int64_t W) {
  const WinThread t = Walk::thread(H, W);
  const int64_t elem = H * W;
  const int32_t iW = static_cast<int32_t>(W);
  SOIL_WIN_WALK(Walk, w, W);
  SOIL_WIN_ROWS(x, w, reinterpret_cast<const float*>(graph), H, W, t) {
    const int32_t n0 = static_cast<int32_t>(x * W + t.y0);
    int4 cnt;
    int32_t* ci = reinterpret_cast<int32_t*>(&cnt);
    int32_t dn[4][K];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int32_t n = n0 + c;
      int m = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) dn[c][k] = -1;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        // the cell whose k-th neighbour is n: row x - kDX[k], column y - kDY[k]
        const bool row_ok = kDX[k] > 0 ? w.has_up : (kDX[k] < 0 ? w.has_dn : true);
        const bool col_ok = kDY[k] > 0 ? (c > 0 || t.y0 > 0) : (kDY[k] < 0 ? (c < 3 || t.y0 + 4 < W) : true);
        const Row6& r = kDX[k] > 0 ? w.up : (kDX[k] < 0 ? w.dn : w.mid);
        const int32_t g = static_cast<int32_t>(f2bits(r.v[c + 1 - kDY[k]]));
        const bool drains = row_ok && col_ok && g == n;  // :333-345
        const int32_t d = n - kDX[k] * iW - kDY[k];
        // slot m takes it (compacted in the order of k); written out as selects, no indexed array
#pragma unroll
        for (int j = 0; j < K; ++j) dn[c][j] = (drains && m == j) ? d : dn[c][j];
        m += drains ? 1 : 0;
      }
      ci[c] = m;
    }
    if (t.live) {
      *reinterpret_cast<int4*>(count + n0) = cnt;
      *reinterpret_cast<float4*>(value + n0) = *reinterpret_cast<const float4*>(source + n0);  // :553
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (k < ci[c]) {
            donor[k * elem + n0 + c] = dn[c][k];
            if (TENSOR_DECAY) {
              const float D = decayIn[dn[c][k]];
              decay[k * elem + n0 + c] = (k < 4) ? D : powf_(D, 1.414f);
            }
          }
        }
      }
    }
  }
}

// __rake_compress, graph.cu:429-522: one synchronous round, in -> out.
//
// The arithmetic and its order are the reference's; two things keep finished work
// out of HBM.  (1) `count` carries one more state: > 0 donors pending, 0 final but
// the other buffer still holds the stale cell, -1 final in both buffers — such a
// cell costs one 4-byte read per round instead of 16 bytes read + written (on a
// 4096^2 DEM most cells are final after a few of the 26 rounds).  (2) A round whose
// predecessor left every cell at -1 has nothing to do and returns at once
// (`flags`: [round % 3] = "work left", set by the previous round).
//
// DECAY = false (round 5): `accumulate` without a decay tensor (BASELINE config 3's call,
// example/dem_multiflow.py:48).  The reference then runs with a decay of 1 on every edge (:382-420
// with no tensor: D = 1, and powf(1, 1.414) = 1), 1.0f * v is v bit for bit and so is a product of
// ones: the decay arrays are neither made (k_donors), read, gathered nor written — a third of the
// bytes a pending cell moves per round — and the sums are the same floats added in the same order.
// A word at a 32-bit BYTE offset from a uniform base: the form the compiler turns into a load with a scalar
// base and one offset register per lane (`global_load_dword v, v, s[..]`) instead of a 64-bit address per lane.
template <typename T, typename IDX>
__device__ __forceinline__ T& word_at(T* base, IDX i) {
  if constexpr (sizeof(IDX) == 4)
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + static_cast<uint32_t>(i * 4u));
  else
    return base[i];
}
template <typename T, typename IDX>
__device__ __forceinline__ const T& word_at(const T* base, IDX i) {
  if constexpr (sizeof(IDX) == 4)
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + static_cast<uint32_t>(i * 4u));
  else
    return base[i];
}
// One cell of one round (graph.cu:438-520), `in` -> `out`.  Returns whether the cell has to be looked at
// again in the next round (it still has donors, or it became final only in `out`).  The cell's count on
// entry is passed in: the dense rounds read it for every cell, the listed rounds only for their cells.
template <int K, bool DECAY, typename IDX>
__device__ __forceinline__ bool rake_cell(const Acc& out, const Acc& in, IDX elem, IDX n, int count) {
  float value = word_at(in.value, n);  // :440
  int32_t donors[K];
  float decays[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {  // :448-468
    if (k < count) {
      donors[k] = word_at(in.donor, k * elem + n);
      decays[k] = DECAY ? word_at(in.decay, k * elem + n) : 1.0f;
    }
  }
  const bool was_final = count == 0;
  for (int k = 0; k < count; ++k) {  // :471
    const IDX donor = static_cast<IDX>(donors[k]);
    const float decay = decays[k];
    const int dcount = word_at(in.count, donor);  // :476
    if (dcount <= 0) {                   // :479-487
      value += DECAY ? decay * word_at(in.value, donor) : word_at(in.value, donor);
      donors[k] = donors[count - 1];
      decays[k] = decays[count - 1];
      donors[count - 1] = -1;
      decays[count - 1] = 0.0f;
      count -= 1;
      k -= 1;
    } else if (dcount == 1) {  // :490-494
      value += DECAY ? decay * word_at(in.value, donor) : word_at(in.value, donor);
      donors[k] = word_at(in.donor, donor);  // slot 0 of the donor
      if (DECAY) decays[k] = decay * word_at(in.decay, donor);
    }
  }
  word_at(out.value, n) = value;  // :498
  if (was_final) {       // both buffers hold the final value from here on
    word_at(out.count, n) = -1;
    word_at(in.count, n) = -1;
    return false;
  }
  word_at(out.count, n) = count;  // :499
#pragma unroll
  for (int k = 0; k < K; ++k) {  // :500-520
    if (k < count) {
      word_at(out.donor, k * elem + n) = donors[k];
      if (DECAY) word_at(out.decay, k * elem + n) = decays[k];
    }
  }
  return true;  // still has donors, or became final only in `out`
}

// The lists of the listed rounds (round 6) are kept per work-group: work-group b of a round writes the
// cells it wants looked at again into segment b of the list (`seg` entries: its share of the cells in the
// dense round that makes the first list, never more afterwards) and their number into fill[b]; work-group
// b of the next round reads that segment.  ONE counter for the whole list was tried first: an atomic per
// wave on one address — 262 144 of them in the round that makes the list at 4096^2 — serialises at the
// memory side: 7.1 ms per accumulation instead of 1.8 (profiles/r06_accumulate/experiments.txt).  The
// counter of a segment lives in LDS; the entries of a wave stand side by side in the order of its lanes,
// so the cells of neighbouring lanes stay neighbours in memory.
template <typename IDX>
__device__ __forceinline__ void rake_append(bool again, IDX n, uint32_t* __restrict__ segment, uint32_t* s_fill) {
  const uint64_t m = __ballot(again);
  if (m == 0) return;
  const int lane = static_cast<int>(threadIdx.x & 63u), leader = __ffsll(static_cast<long long>(m)) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(s_fill, static_cast<uint32_t>(__popcll(m)));
  base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(base), leader));
  if (again) segment[base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)))] = static_cast<uint32_t>(n);
}

// Control words of the dense rounds of one accumulation (workspace): flags[0..2] = "work left"
constexpr int kRakeFlags = 4;
__global__ void k_rake_init(int* __restrict__ flags) {
  if (threadIdx.x < kRakeFlags) flags[threadIdx.x] = threadIdx.x == 0 ? 1 : 0;
}

// IDX: uint32_t where K * elem words are under 4 GiB (a uniform base and a 32-bit offset per lane instead
// of a 64-bit address per lane), int64_t otherwise.
// `list_out` (round 6): the dense round in front of the listed rounds writes down the cells the next round
// has to look at (k_rake_list).
template <int K, bool DECAY, typename IDX>
__global__ void __launch_bounds__(kGBlock)
    k_rake_compress(Acc out, const Acc in, int64_t elem64, int* __restrict__ flags, int round,
                    uint32_t* __restrict__ list_out, uint32_t* __restrict__ fill_out, uint32_t seg) {
  const IDX elem = static_cast<IDX>(elem64);
  __shared__ uint32_t s_fill;
  // the word round + 2 will read is cleared either way: a round that returns at once must not
  // leave its predecessor's "work left" standing for the round three launches on
  if (blockIdx.x == 0 && threadIdx.x == 0) flags[(round + 2) % 3] = 0;
  if (flags[round % 3] == 0) {
    if (list_out && threadIdx.x == 0) fill_out[blockIdx.x] = 0;  // nothing pending anywhere: empty lists
    return;
  }
  if (list_out) {
    if (threadIdx.x == 0) s_fill = 0;
    __syncthreads();
  }
  uint32_t* const segment = list_out ? list_out + static_cast<size_t>(blockIdx.x) * seg : nullptr;
  bool pending = false;
  // a grid of a few work-groups per CU strides over the cells: a round that returns at once (11 of
  // the 26 at 4096^2) costs a few us instead of the 15 us it takes to hand out 65 536 work-groups
  for (IDX n = static_cast<IDX>(blockIdx.x) * kGBlock + threadIdx.x; n < elem;
       n += static_cast<IDX>(gridDim.x) * kGBlock) {
    const int count = word_at(in.count, n);  // :441
    bool again = false;
    if (count >= 0) again = rake_cell<K, DECAY, IDX>(out, in, elem, n, count);
    pending = pending || again;
    if (list_out) rake_append<IDX>(again, n, segment, &s_fill);
  }
  if (__any(pending) && (threadIdx.x & 63) == 0) flags[(round + 1) % 3] = 1;
  if (list_out) {
    __syncthreads();
    if (threadIdx.x == 0) fill_out[blockIdx.x] = s_fill;
  }
}

// A round over the LISTS of the cells that still change (round 6).  After a handful of rounds a few per
// cent of the cells are pending — 9 % going into round 6 of a 4096^2 D8 realisation, 0.8 % into round 10
// (tools/count_rake_bytes.py) — and a dense round still reads the count of every cell: rounds 6-15 cost
// 0.45 ms of the 1.61 ms of the 26 rounds for 4 bytes per cell each.  A listed round touches its cells
// only and makes the lists of the next round; a work-group whose segment is empty returns at once.
template <int K, bool DECAY, typename IDX>
__global__ void __launch_bounds__(kGBlock)
    k_rake_list(Acc out, const Acc in, int64_t elem64, const uint32_t* __restrict__ list_in,
                const uint32_t* __restrict__ fill_in, uint32_t* __restrict__ list_out, uint32_t* __restrict__ fill_out,
                uint32_t seg) {
  const IDX elem = static_cast<IDX>(elem64);
  const uint32_t n_in = fill_in[blockIdx.x];
  if (n_in == 0) {
    if (threadIdx.x == 0) fill_out[blockIdx.x] = 0;
    return;
  }
  __shared__ uint32_t s_fill;
  if (threadIdx.x == 0) s_fill = 0;
  __syncthreads();
  const uint32_t* const mine = list_in + static_cast<size_t>(blockIdx.x) * seg;
  uint32_t* const segment = list_out + static_cast<size_t>(blockIdx.x) * seg;
  for (uint32_t i = threadIdx.x; i < n_in; i += kGBlock) {
    const IDX n = static_cast<IDX>(mine[i]);
    const int count = word_at(in.count, n);  // (>= 0: it is on the list)
    const bool again = rake_cell<K, DECAY, IDX>(out, in, elem, n, count);
    rake_append<IDX>(again, n, segment, &s_fill);
  }
  __syncthreads();
  if (threadIdx.x == 0) fill_out[blockIdx.x] = s_fill;
}

// Workspace slots of an accumulation: soil_multiflow keeps two of them going side by side (kAccLanes)
constexpr int kAccLanes = 2;
constexpr int kAccSlot[kAccLanes] = {0, 10};

template <int K>
static int accumulate_impl(float* out, const int32_t* graph, const float* source,
                           const float* decayIn, int64_t H, int64_t W, hipStream_t st, bool sync = true, int lane = 0) {
  const int64_t elem = H * W;
  auto align = [](size_t b) { return (b + 255) & ~static_cast<size_t>(255); };
  const size_t b1 = align(sizeof(float) * elem), bK = align(sizeof(float) * elem * K);
  void* base = nullptr;
  static const unsigned rake_groups = [] { const char* e = std::getenv("SOIL_RAKE_GROUPS"); return e && std::atoi(e) > 0 ? static_cast<unsigned>(std::atoi(e)) : 256u * 32u; }();  // 1024 .. 65536 groups: 3.16 3.02 2.68 2.54 2.61 2.99 ms per realisation
  const unsigned nb = std::min(blocks_for(elem, kGBlock), rake_groups);
  // a work-group's segment of the lists of the listed rounds: its share of the cells, in whole waves
  const size_t seg = ((static_cast<size_t>(elem) + nb - 1) / nb + 255) / 256 * 256;
  const size_t bL = align(sizeof(uint32_t) * seg * nb), bF = align(sizeof(uint32_t) * nb);
  int rc = workspace_get(kAccSlot[lane], 3 * b1 + 4 * bK + 2 * bL + 2 * bF + 256, &base);
  if (rc != SOIL_OK) return rc;
  char* p = static_cast<char*>(base);
  Acc A, B;
  A.count = reinterpret_cast<int32_t*>(p); p += b1;
  A.value = out;  // the even rounds write here: the result needs no copy
  B.count = reinterpret_cast<int32_t*>(p); p += b1;
  B.value = reinterpret_cast<float*>(p);   p += b1;
  A.donor = reinterpret_cast<int32_t*>(p); p += bK;
  A.decay = reinterpret_cast<float*>(p);   p += bK;
  B.donor = reinterpret_cast<int32_t*>(p); p += bK;
  B.decay = reinterpret_cast<float*>(p);   p += bK;
  uint32_t* const list[2] = {reinterpret_cast<uint32_t*>(p), reinterpret_cast<uint32_t*>(p + bL)};
  p += 2 * bL;
  uint32_t* const fill[2] = {reinterpret_cast<uint32_t*>(p), reinterpret_cast<uint32_t*>(p + bF)};
  p += 2 * bF;
  int* flags = reinterpret_cast<int*>(p);

  const bool wide = W % 4 == 0 && W >= 4 &&
                    ((reinterpret_cast<uintptr_t>(graph) | reinterpret_cast<uintptr_t>(source) |
                      reinterpret_cast<uintptr_t>(A.value) | reinterpret_cast<uintptr_t>(A.count)) & 15) == 0;
  if (wide) {  // :552-556
    // (round 5: blocks of rows on grids whose band walk is under 2048 work-groups, window.hpp win_shape_for —
    // 4096^2, BASELINE config 3: band walk | blocks of four | two rows 143 | 118 | 116 us per realisation)
    const int shape = win_shape_for(0, 5, H, W);
    auto go = [&](auto kern, dim3 grid) {
      kern<<<grid, kWinBlock, 0, st>>>(A.count, A.donor, A.decay, A.value, graph, source, decayIn, H, W);
    };
    if (shape == 4) {
      if (decayIn) go(k_donors4<K, true, RowWalkBlock4<false>>, RowWalkBlock4<false>::grid(H, W));
      else go(k_donors4<K, false, RowWalkBlock4<false>>, RowWalkBlock4<false>::grid(H, W));
    } else if (shape == 5) {
      if (decayIn) go(k_donors4<K, true, RowWalkBlock2<false>>, RowWalkBlock2<false>::grid(H, W));
      else go(k_donors4<K, false, RowWalkBlock2<false>>, RowWalkBlock2<false>::grid(H, W));
    } else {
      if (decayIn) go(k_donors4<K, true, RowWalk>, win_grid(H, W));
      else go(k_donors4<K, false, RowWalk>, win_grid(H, W));
    }
  }
  else if (decayIn)
    k_donors<K, true><<<grid_rows(H, W, kGBlock), kGBlock, 0, st>>>(A.count, A.donor, A.decay, A.value, graph, source,
                                              decayIn, H, W);
  else
    k_donors<K, false><<<grid_rows(H, W, kGBlock), kGBlock, 0, st>>>(A.count, A.donor, A.decay, A.value, graph, source,
                                               nullptr, H, W);
  SOIL_LAUNCH_CHECK();

  const int64_t iter =
      static_cast<int64_t>(std::ceil(std::log2(static_cast<float>(elem)) / 2.0f));  // :559
  k_rake_init<<<1, 64, 0, st>>>(flags);  // flags = {1, 0, 0} (a kernel, not a copy from the host's stack: stream-ordered)
  // (Round 4: a variant with two / four / eight cells in flight per thread and the donors' words asked
  // for in batches — three round trips per group of cells instead of four to six per cell — ran at
  // 2.23-2.32 / 2.38-2.59 / 3.1 ms per 4096^2 realisation against 2.27-2.36 for this kernel: the
  // rounds are not bound by the length of a thread's chain of loads.  profiles/r04_accumulate has
  // their bytes: round 0 moves 51 B/cell at 2.4 TB/s, the 26 rounds 279 B/cell in 2.03 ms.  Nor by the
  // number of their memory instructions or of the sectors their gathers touch: with a cell's count,
  // value, first slot and first decay in ONE 16-byte record — one gather per donor instead of up to four,
  // five memory instructions per pending cell instead of ten to twelve, the same bytes — the realisation
  // took the same 2.28 ms; with the record kept as a mirror beside the arrays (16 bytes more written per
  // pending cell and round) 3.08 ms.  Bytes are what a round costs, at ~2.4 TB/s.
  // Round 5: not the bytes either.  Without the decay arrays (DECAY = false) the rounds move 216 instead of
  // 279 B/cell and take 1.64 instead of 2.03 ms, still at 2.2 TB/s; slots written as whole 32-byte sectors (the
  // eight lanes of a sector all write slot k as soon as one has a donor there: more bytes, no partly written
  // sectors): 1.87 | 1.88 ms; the arrays and slot planes a non-power-of-two apart (64 elements ... 1 M between
  // them instead of 64 MiB exactly): 1.99 | 1.99 ... 1.94 on one box.  tools/pmc_rake.sh: round 0 issues 242
  // vector, 80 scalar, 15 load and 6 store instructions per wave and 64 cells — 0.36 of the vector issue slots.)
  // (Round 5: the gathers of a cell asked for together — count -> value and slots -> every donor's count and
  // value -> the first slots of the donors that hold one -> stores: four round trips whatever the number of
  // donors, the list walked in registers as the two pointers the reference's loop amounts to, no branch
  // between the loads (with one per slot the compiler waits for each load in turn: vmcnt(0) everywhere) —
  // bit-identical and SLOWER: 1.97-2.02 against 1.79-1.80 ms per realisation.  The rounds are not bound by
  // the length of a wave's chain of loads but by the number of its memory requests.)
  const bool idx32 = static_cast<uint64_t>(elem) * K * sizeof(float) < (1ull << 32);
  // Rounds from `list_from` on run over the lists of the cells that still change (k_rake_list); the dense
  // round in front of them makes the first lists.  SOIL_RAKE_LIST_FROM: that round (0 or beyond the last
  // round: dense rounds throughout).  4096^2 D8, ms per accumulation: dense throughout 1.84-1.87; from round
  // 1 / 2 / 3 / 4 / 5 / 6 / 8: 1.52-1.55 / 1.50-1.54 / 1.50-1.54 / 1.52 / 1.55 / 1.60 / 1.70.
  // The round count is the reference's: 2 (ceil(log2(HW) / 2) + 1).
  static const int list_from_env = [] { const char* e = std::getenv("SOIL_RAKE_LIST_FROM"); return e ? std::atoi(e) : 2; }();
  const int rounds = static_cast<int>(2 * (iter + 1));
  const int list_from = (list_from_env >= 1 && list_from_env < rounds) ? list_from_env : rounds;
  const uint32_t seg32 = static_cast<uint32_t>(seg);
  for (int r = 0; r < rounds; ++r) {                                                // :560-563
    const Acc& o = (r & 1) ? A : B;
    const Acc& in = (r & 1) ? B : A;
    if (r >= list_from) {  // work-group b reads segment b of the lists round r - 1 made, and makes segment b of the next
      const uint32_t *li = list[r & 1], *fi = fill[r & 1];
      uint32_t *lo = list[(r + 1) & 1], *fo = fill[(r + 1) & 1];
      if (idx32) {
        if (decayIn) k_rake_list<K, true, uint32_t><<<nb, kGBlock, 0, st>>>(o, in, elem, li, fi, lo, fo, seg32);
        else k_rake_list<K, false, uint32_t><<<nb, kGBlock, 0, st>>>(o, in, elem, li, fi, lo, fo, seg32);
      } else {
        if (decayIn) k_rake_list<K, true, int64_t><<<nb, kGBlock, 0, st>>>(o, in, elem, li, fi, lo, fo, seg32);
        else k_rake_list<K, false, int64_t><<<nb, kGBlock, 0, st>>>(o, in, elem, li, fi, lo, fo, seg32);
      }
      continue;
    }
    uint32_t* lo = r + 1 == list_from ? list[(r + 1) & 1] : nullptr;
    uint32_t* fo = fill[(r + 1) & 1];
    if (idx32) {
      if (decayIn) k_rake_compress<K, true, uint32_t><<<nb, kGBlock, 0, st>>>(o, in, elem, flags, r, lo, fo, seg32);
      else k_rake_compress<K, false, uint32_t><<<nb, kGBlock, 0, st>>>(o, in, elem, flags, r, lo, fo, seg32);
    } else {
      if (decayIn) k_rake_compress<K, true, int64_t><<<nb, kGBlock, 0, st>>>(o, in, elem, flags, r, lo, fo, seg32);
      else k_rake_compress<K, false, int64_t><<<nb, kGBlock, 0, st>>>(o, in, elem, flags, r, lo, fo, seg32);
    }
  }
  SOIL_LAUNCH_CHECK();
  if (sync) SOIL_HIP(hipStreamSynchronize(st));  // cudaDeviceSynchronize, graph.cu:564
  return SOIL_OK;
}

// sum += double(float(acc / K)): the term of example/dem_multiflow.py:49
// (`multiflow += accumulation.cpu().numpy() / float(K)`, a float32 quotient
// added into a float64 array), kept on the device
__global__ void __launch_bounds__(kGBlock)
    k_mean_add(double* __restrict__ sum, const float* __restrict__ acc, float Kf, int64_t elem) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kGBlock + threadIdx.x;
  if (n >= elem) return;
  sum[n] += static_cast<double>(acc[n] / Kf);
}

}  // namespace soil

using namespace soil;

extern "C" {

int soil_direction(int32_t* direction, const float* height, int64_t H, int64_t W, int edge,
                   void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(direction && height, "direction: null tensor");
  SOIL_REQUIRE(H > 0 && W > 0, "direction: empty grid");
  const bool wide = W % 4 == 0 && W >= 4;  // four cells per thread (window.hpp)
  hipStream_t st = as_stream(stream);
  switch (edge) {
    case SOIL_D4:
      if (wide) launch_steepest4<4, true>(direction, height, H, W, st);
      else k_steepest<4, true><<<grid_rows(H, W, kGBlock), kGBlock, 0, st>>>(direction, height, H, W);
      break;
    case SOIL_D8:
      if (wide) launch_steepest4<8, true>(direction, height, H, W, st);
      else k_steepest<8, true><<<grid_rows(H, W, kGBlock), kGBlock, 0, st>>>(direction, height, H, W);
      break;
    default: return fail(SOIL_ERR_INVALID_ARGUMENT, "invalid edge enumerator");  // graph.cu:262
  }
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_steepest(int32_t* graph, const float* height, int64_t H, int64_t W, int edge,
                  void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(graph && height, "steepest: null tensor");
  SOIL_REQUIRE(H > 0 && W > 0 && H * W <= INT32_MAX, "steepest: grid must have 1..2^31-1 cells");
  const bool wide = W % 4 == 0 && W >= 4;
  hipStream_t st = as_stream(stream);
  switch (edge) {
    case SOIL_D4:
      if (wide) launch_steepest4<4, false>(graph, height, H, W, st);
      else k_steepest<4, false><<<grid_rows(H, W, kGBlock), kGBlock, 0, st>>>(graph, height, H, W);
      break;
    case SOIL_D8:
      if (wide) launch_steepest4<8, false>(graph, height, H, W, st);
      else k_steepest<8, false><<<grid_rows(H, W, kGBlock), kGBlock, 0, st>>>(graph, height, H, W);
      break;
    default: return fail(SOIL_ERR_INVALID_ARGUMENT, "invalid edge enumerator");  // graph.cu:88
  }
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_random_weighted(int32_t* graph, const float* height, int64_t H, int64_t W, int edge,
                         uint64_t seed, uint64_t offset, float T, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(graph && height, "random_weighted: null tensor");
  SOIL_REQUIRE(H > 0 && W > 0 && H * W <= INT32_MAX,
               "random_weighted: grid must have 1..2^31-1 cells");
  RwBatch b{};
  b.graph[0] = graph, b.offset[0] = offset, b.n = 1;
  switch (edge) {
    case SOIL_D4: launch_random_weighted<4>(b, height, H, W, seed, T, as_stream(stream)); break;
    case SOIL_D8: launch_random_weighted<8>(b, height, H, W, seed, T, as_stream(stream)); break;
    default: return fail(SOIL_ERR_INVALID_ARGUMENT, "invalid edge enumerator");  // graph.cu:192
  }
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_slope(float* slope, const float* tensor, const int32_t* flow, int64_t H, int64_t W,
               const float scale[2], void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(slope && tensor && flow && scale, "slope: null argument");
  SOIL_REQUIRE(H > 0 && W > 0 && H * W <= INT32_MAX, "slope: grid must have 1..2^31-1 cells (int32 flow graph)");
  if (W % 4 == 0 && W >= 4 && (reinterpret_cast<uintptr_t>(flow) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(slope) & 15) == 0 && (reinterpret_cast<uintptr_t>(tensor) & 15) == 0) {
    // (round 5, ms at 8192^2 by shape — band walk | blocks of four rows | of two: see DESIGN.md 3.3)
    const int shape = win_shape_for(5, 5, H, W);
    auto k = shape == 4 ? k_slope4<RowWalkBlock4<false>> : (shape == 5 ? k_slope4<RowWalkBlock2<false>> : (shape == 7 ? k_slope4<RowWalkStack2<false>> : (shape == 8 ? k_slope4<RowWalkStack4<false>> : k_slope4<RowWalk>)));
    const dim3 grid = shape == 4 ? RowWalkBlock4<false>::grid(H, W) : (shape == 5 ? RowWalkBlock2<false>::grid(H, W) : (shape == 7 ? RowWalkStack2<false>::grid(H, W) : (shape == 8 ? RowWalkStack4<false>::grid(H, W) : win_grid(H, W))));
    k<<<grid, kWinBlock, 0, as_stream(stream)>>>(slope, tensor, flow, H, W, Scale2{scale[0], scale[1]});
  } else
    k_slope<<<grid_rows(H, W, kGBlock), kGBlock, 0, as_stream(stream)>>>(
        slope, tensor, flow, static_cast<int32_t>(H), static_cast<int32_t>(W), Scale2{scale[0], scale[1]});
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_accumulate(float* out, const int32_t* graph, const float* source, const float* decay,
                    int64_t H, int64_t W, int edge, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && graph && source, "accumulate: null tensor");
  SOIL_REQUIRE(H > 0 && W > 0 && H * W <= INT32_MAX, "accumulate: grid must have 1..2^31-1 cells");
  switch (edge) {
    case SOIL_D4: return accumulate_impl<4>(out, graph, source, decay, H, W, as_stream(stream));
    case SOIL_D8: return accumulate_impl<8>(out, graph, source, decay, H, W, as_stream(stream));
    default: return fail(SOIL_ERR_INVALID_ARGUMENT, "invalid edge enumerator");  // graph.cu:573
  }
}

// soil_multiflow keeps kAccLanes accumulations in flight (round 6).  The realisations are independent
// (example/dem_multiflow.py:43-49 loops over them) and the later rounds of an accumulation are short
// kernels over a few thousand cells that leave the chip idle — 0.3 of the 1.5 ms at 4096^2 — so
// realisation i runs on lane i % kAccLanes, a stream and a workspace of its own, while the caller's
// stream makes the graphs (a batch ahead) and adds the results into `sum` IN THE ORDER OF k: the
// float64 sum is the one the script's loop makes.  SOIL_FLOW_LANES=1: one after the other on the
// caller's stream (A/B).
struct FlowLanes {
  int device = -1;
  hipStream_t lane[kAccLanes] = {};
  hipEvent_t start = nullptr, graphs[2] = {}, done[kAccLanes] = {}, used[kAccLanes] = {};
  int ensure() {
    int dev = 0;
    SOIL_HIP(hipGetDevice(&dev));
    if (device == dev) return SOIL_OK;
    SOIL_REQUIRE(device < 0, "multiflow: a host thread's lanes belong to the device of its first call");
    for (int j = 0; j < kAccLanes; ++j) {
      SOIL_HIP(hipStreamCreateWithFlags(&lane[j], hipStreamNonBlocking));
      SOIL_HIP(hipEventCreateWithFlags(&done[j], hipEventDisableTiming));
      SOIL_HIP(hipEventCreateWithFlags(&used[j], hipEventDisableTiming));
    }
    SOIL_HIP(hipEventCreateWithFlags(&start, hipEventDisableTiming));
    for (int j = 0; j < 2; ++j) SOIL_HIP(hipEventCreateWithFlags(&graphs[j], hipEventDisableTiming));
    device = dev;
    return SOIL_OK;
  }
};
static thread_local FlowLanes t_flow;

int soil_multiflow(double* sum, const float* height, const float* source, int64_t H, int64_t W,
                   int edge, uint64_t seed, uint64_t k_first, uint64_t k_stride, uint64_t k_end,
                   uint64_t K, float T, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(sum && height && source, "multiflow: null tensor");
  SOIL_REQUIRE(k_stride > 0 && K > 0, "multiflow: stride and realisation count must be positive");
  const int64_t elem = H * W;
  auto align = [](size_t b) { return (b + 255) & ~static_cast<size_t>(255); };
  void* base = nullptr;
  const size_t b_graph = align(sizeof(int32_t) * elem), b_acc = align(sizeof(float) * elem);
  // two batches of graphs (the next one is made while the accumulations of this one run) and a result plane per lane
  if (int rc = workspace_get(3, 2 * kRwBatch * b_graph + kAccLanes * b_acc, &base); rc != SOIL_OK)
    return rc;
  char* ws = static_cast<char*>(base);
  float* acc[kAccLanes];
  for (int j = 0; j < kAccLanes; ++j) acc[j] = reinterpret_cast<float*>(ws + 2 * kRwBatch * b_graph + j * b_acc);
  hipStream_t st = as_stream(stream);
  SOIL_REQUIRE(H > 0 && W > 0 && elem <= INT32_MAX, "multiflow: grid must have 1..2^31-1 cells");
  SOIL_REQUIRE(edge == SOIL_D4 || edge == SOIL_D8, "invalid edge enumerator");
  static const int lanes_env = [] { const char* e = std::getenv("SOIL_FLOW_LANES"); return e ? std::atoi(e) : kAccLanes; }();
  const bool side_by_side = lanes_env >= 2;
  if (side_by_side) {
    if (int rc = t_flow.ensure(); rc != SOIL_OK) return rc;
    SOIL_HIP(hipEventRecord(t_flow.start, st));  // the lanes start behind what the caller has queued (`sum` zeroed, the DEM made)
    for (int j = 0; j < kAccLanes; ++j) SOIL_HIP(hipStreamWaitEvent(t_flow.lane[j], t_flow.start, 0));
  }
  // kRwBatch realisations' graphs per pass over the heights (the weights of a cell are the same for
  // every draw), then one accumulation each
  struct Batch { RwBatch b; int made; };
  auto make_batch = [&](uint64_t& k, int set) {
    Batch q{};
    int m = 0;
    for (; m < kRwBatch && k < k_end; ++m, k += k_stride) {
      q.b.graph[m] = reinterpret_cast<int32_t*>(ws + (static_cast<size_t>(set) * kRwBatch + m) * b_graph);
      q.b.offset[m] = k;
    }
    q.made = m, q.b.n = m;
    if (m > 0) {
      if (edge == SOIL_D4)
        launch_random_weighted<4>(q.b, height, H, W, seed, T, st);
      else
        launch_random_weighted<8>(q.b, height, H, W, seed, T, st);
    }
    return q;
  };
  uint64_t k = k_first;
  int set = 0;
  uint64_t index = 0;  // realisations so far: lane = index % kAccLanes
  bool lane_used[kAccLanes] = {};
  Batch cur = make_batch(k, set);
  SOIL_LAUNCH_CHECK();
  if (side_by_side && cur.made > 0) SOIL_HIP(hipEventRecord(t_flow.graphs[set], st));
  while (cur.made > 0) {
    // the graphs of the next batch, into the other set: its last readers — the accumulations of the batch
    // before this one — have been waited for by the caller's stream (their results are in `sum`)
    Batch next = make_batch(k, set ^ 1);
    SOIL_LAUNCH_CHECK();
    if (side_by_side && next.made > 0) SOIL_HIP(hipEventRecord(t_flow.graphs[set ^ 1], st));
    for (int j = 0; j < cur.made; ++j, ++index) {
      const int lane = side_by_side ? static_cast<int>(index % kAccLanes) : 0;
      hipStream_t ls = side_by_side ? t_flow.lane[lane] : st;
      if (side_by_side) {
        SOIL_HIP(hipStreamWaitEvent(ls, t_flow.graphs[set], 0));
        if (lane_used[lane]) SOIL_HIP(hipStreamWaitEvent(ls, t_flow.used[lane], 0));  // its result plane has been added
      }
      // (stream-ordered: the host does not wait for every realisation as soil_accumulate does for its caller)
      const int rc = edge == SOIL_D4 ? accumulate_impl<4>(acc[lane], cur.b.graph[j], source, nullptr, H, W, ls, false, lane)
                                     : accumulate_impl<8>(acc[lane], cur.b.graph[j], source, nullptr, H, W, ls, false, lane);
      if (rc != SOIL_OK) return rc;
      if (side_by_side) {
        SOIL_HIP(hipEventRecord(t_flow.done[lane], ls));
        SOIL_HIP(hipStreamWaitEvent(st, t_flow.done[lane], 0));
      }
      k_mean_add<<<blocks_for(elem, kGBlock), kGBlock, 0, st>>>(sum, acc[lane], static_cast<float>(K), elem);
      SOIL_LAUNCH_CHECK();
      if (side_by_side) {
        SOIL_HIP(hipEventRecord(t_flow.used[lane], st));
        lane_used[lane] = true;
      }
    }
    cur = next;
    set ^= 1;
  }
  SOIL_HIP(hipStreamSynchronize(st));  // like soil_accumulate (graph.cu:564), once: every lane's work is behind it
  return SOIL_OK;
}

int soil_workspace_release(void) {
  SOIL_DEVICE();
  return workspace_release_all();
}

}  // extern "C"
