// window.hpp — the launch shape of the 3x3 stencils on a scalar plane (steepest / direction,
// gradient, negslope, laplacian D = 1): a thread owns FOUR consecutive cells of a row and walks a band
// of rows with the three rows it needs in registers.
//
// Per row a thread issues one 16-byte load; the two columns next to its four cells come from the
// neighbouring lanes' registers (wave shuffles; lanes 0 and 63 reload one float), and going down
// a row re-uses two of the three rows.  Against one thread per cell with nine L1 gathers each
// (stencil.hip's first shape) that is 0.3 load instructions per cell instead of 9, and the stores
// are 16 bytes wide.  The arithmetic per cell is the reference's, statement for statement — the
// kernels that use this header stay bit-identical to the scalar ones (which still serve widths
// that are not a multiple of four).
#pragma once

#include "common.hpp"

namespace soil {

constexpr int kWinBlock = 256;  // threads per work-group: 1024 columns
constexpr int kWinBand = 32;    // rows a work-group walks

inline dim3 win_grid(int64_t H, int64_t W) {
  const int64_t bands = (H + kWinBand - 1) / kWinBand;
  return dim3(static_cast<unsigned>((W / 4 + kWinBlock - 1) / kWinBlock),
              static_cast<unsigned>(bands < 65535 ? bands : 65535));
}

// columns y0 - 1 .. y0 + 4 of one row (y0 = the thread's first cell)
struct Row6 {
  float v[6];
};

// Every lane of the wave must call this (shuffles); `row_ok`: the row lies inside the grid.
// Columns outside the grid (and rows that are) come back as 0 — callers test existence first.
__device__ __forceinline__ Row6 load_row6(const float* __restrict__ in, int64_t x, int64_t W,
                                          int64_t y0, bool row_ok) {
  const int lane = static_cast<int>(threadIdx.x & 63u);
  float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const float* row = in + x * W;
  if (row_ok) c = *reinterpret_cast<const float4*>(row + y0);
  float l = __shfl_up(c.w, 1, 64), r = __shfl_down(c.x, 1, 64);
  if (lane == 0) l = (row_ok && y0 > 0) ? row[y0 - 1] : 0.0f;
  if (lane == 63) r = (row_ok && y0 + 4 < W) ? row[y0 + 4] : 0.0f;
  return Row6{{l, c.x, c.y, c.z, c.w, r}};
}

// Rows x - 1, x, x + 1 as a thread walks down its band: start() loads all three, next() moves one
// row down re-using two.  Every lane of the wave must make the calls (they shuffle).
struct RowWalk {
  Row6 up, mid, dn;
  bool has_up, has_dn;
  __device__ __forceinline__ void start(const float* __restrict__ in, int64_t x, int64_t H,
                                        int64_t W, int64_t y0) {
    has_up = x > 0;
    has_dn = x + 1 < H;
    up = load_row6(in, x - 1, W, y0, has_up);
    mid = load_row6(in, x, W, y0, true);
    dn = load_row6(in, x + 1, W, y0, has_dn);
  }
  __device__ __forceinline__ void next(const float* __restrict__ in, int64_t x, int64_t H,
                                       int64_t W, int64_t y0) {  // x: the row moved onto
    up = mid;
    mid = dn;
    has_up = true;
    has_dn = x + 1 < H;
    dn = load_row6(in, x + 1, W, y0, has_dn);
  }
};

// the thread's first column; threads past the end of the row keep running on the last group
// (their lanes feed the shuffles) and store nothing
struct WinThread {
  int64_t y0;
  bool live;
};
__device__ __forceinline__ WinThread win_thread(int64_t W) {
  const int64_t y = (static_cast<int64_t>(blockIdx.x) * kWinBlock + threadIdx.x) * 4;
  return WinThread{y < W ? y : W - 4, y < W};
}

// A lane that has 32 contiguous bytes to store (two float4: four cells of a two-channel plane) would
// issue two 16-byte stores at a 32-byte stride — every store instruction of the wave then covers
// half of each 32-byte sector, and the kernel runs at 49 % of the HBM roofline instead of 61 %
// (k_gradient4, measured).  Through a wave-private 2 KiB of LDS the two instructions become 1 KiB
// of consecutive bytes each.  `tile`: 128 float4 of LDS owned by this wave; every lane calls it;
// `live`: the lane's data is to be stored (live lanes are a prefix of the wave).
__device__ __forceinline__ void store_pair_contiguous(float4* __restrict__ wave_dst, float4 a,
                                                      float4 b, float4* tile, bool live) {
  const int lane = static_cast<int>(threadIdx.x & 63u);
  const int n = 2 * __popcll(__ballot(live));  // float4s of the wave that are real
  tile[2 * lane] = a;
  tile[2 * lane + 1] = b;
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own writes have landed
  __builtin_amdgcn_wave_barrier();
  const float4 p = tile[lane], q = tile[64 + lane];
  if (lane < n) wave_dst[lane] = p;
  if (64 + lane < n) wave_dst[64 + lane] = q;
  __builtin_amdgcn_wave_barrier();
}

// for (x over the rows of this work-group's bands) with `w` holding rows x - 1 .. x + 1
#define SOIL_WIN_ROWS(x, w, in, H, W, y0)                                                        \
  for (int64_t x##_band = blockIdx.y; x##_band * ::soil::kWinBand < (H); x##_band += gridDim.y)  \
    for (int64_t x = x##_band * ::soil::kWinBand,                                                \
                 x##_end = (x + ::soil::kWinBand < (H)) ? x + ::soil::kWinBand : (H),            \
                 x##_go = ((w).start(in, x, H, W, y0), 1);                                       \
         x < x##_end && x##_go; ++x, (x < x##_end ? (w).next(in, x, H, W, y0) : (void)0))

}  // namespace soil
