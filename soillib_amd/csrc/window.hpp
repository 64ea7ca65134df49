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

#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "soil_math.hpp"

namespace soil {

constexpr int kWinBlock = 256;  // threads per work-group: 1024 columns
constexpr int kWinBand = 32;    // rows a work-group walks

inline dim3 win_grid(int64_t H, int64_t W) {
  const int64_t bands = (H + kWinBand - 1) / kWinBand;
  return dim3(static_cast<unsigned>((W / 4 + kWinBlock - 1) / kWinBlock),
              static_cast<unsigned>(bands < 65535 ? bands : 65535));
}

// the thread's first column; threads past the end of the row keep running on the last group
// (their lanes feed the shuffles) and store nothing
struct WinThread {
  int64_t y0;
  bool live;
  int64_t group;  // the work-group's 1024 columns: group * 1024 ..
  int64_t row;    // the flat shape's row (-1: none); unused by the band shape
  int64_t wave_y0 = (group * kWinBlock + (threadIdx.x & ~63u)) * 4;  // first column of the wave's 256 (lanes past the row's end included)
};
__device__ __forceinline__ WinThread win_thread(int64_t W) {
  const int64_t y = (static_cast<int64_t>(blockIdx.x) * kWinBlock + threadIdx.x) * 4;
  return WinThread{y < W ? y : W - 4, y < W, static_cast<int64_t>(blockIdx.x), -1};
}

// ---- the flat shape (round 4) ------------------------------------------------------------------
//
// tools/microbench/copy_shapes.hip, 8192^2, read 4 + write 4 bytes per cell, no arithmetic:
//   every thread one float4, blocks in address order          87 us  6.15 TB/s
//   the band walk above (32 rows)                             112 us  4.80 TB/s   (8 .. 128 rows: 101 .. 120)
//   every thread its float4 and the ones above and below       90 us  5.94 TB/s
// — the band walk itself costs a quarter of the streaming rate (2048 work-groups each on its own
// DRAM rows, in step), which is what the window kernels without much arithmetic ran at (laplacian
// D = 1: 108 us).  In the flat shape a work-group owns ONE row of 1024 columns and loads its three
// rows; the rows above and below come out of the L2 that the neighbouring work-groups filled.
// For that they have to run on the same XCD: work-groups are dealt to the eight XCDs round-robin
// by their linear index, so the index is taken apart as (XCD k, i-th of that XCD) and XCD k sweeps
// rows k H/8 .. (k + 1) H/8 in address order, whatever the width.
constexpr int kWinXcd = 8;
inline int64_t win_groups(int64_t W) { return (W / 4 + kWinBlock - 1) / kWinBlock; }
inline dim3 win_grid_flat(int64_t H, int64_t W) {
  static const bool natural = [] { const char* e = std::getenv("SOIL_WIN_FLAT_ORDER"); return e && e[0] == '0'; }();
  if (natural && H > 1 && H < 65536) return dim3(static_cast<unsigned>(win_groups(W)), static_cast<unsigned>(H));
  int64_t pad = 1;  // the column groups padded to a power of two: the kernel takes the index apart by shifts
  while (pad < win_groups(W)) pad *= 2;
  return dim3(static_cast<unsigned>(kWinXcd * ((H + kWinXcd - 1) / kWinXcd) * pad));
}
__device__ __forceinline__ WinThread win_thread_flat(int64_t H, int64_t W) {
  int64_t g, row;
  if (gridDim.y > 1) {  // rows in address order (SOIL_WIN_FLAT_ORDER=0)
    g = blockIdx.x;
    row = blockIdx.y;
  } else {  // (shifts, no division: a work-group lives for one row)
    const uint32_t G = static_cast<uint32_t>((W / 4 + kWinBlock - 1) / kWinBlock);
    const uint32_t shift = G > 1 ? 32u - static_cast<uint32_t>(__builtin_clz(G - 1u)) : 0u;
    const uint32_t k = blockIdx.x & (kWinXcd - 1), i = blockIdx.x >> 3;
    g = i & ((1u << shift) - 1u);
    row = static_cast<int64_t>(k) * ((H + kWinXcd - 1) / kWinXcd) + (i >> shift);
    if (g >= G || (i >> shift) >= (H + kWinXcd - 1) / kWinXcd) row = H;
  }
  const int64_t y = (g * kWinBlock + threadIdx.x) * 4;
  return WinThread{y < W ? y : W - 4, y < W, g, row < H ? row : -1};
}

// columns y0 - 1 .. y0 + 4 of one row (y0 = the thread's first cell)
struct Row6 {
  float v[6];
};

// Every lane of the wave must call this (shuffles); `row_ok`: the row lies inside the grid.
// Columns outside the grid (and rows that are) come back as 0 — callers test existence first.
// (One masked load for both halo lanes, issued together with the 16-byte loads of all three rows
// before the first shuffle waits, was measured: the band walk 0.110 -> 0.135 ms on laplacian D = 1.)
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

// the same values with the halo columns loaded by every lane itself (they are in the cache lines
// the wave's 16-byte load brings): three independent loads, nothing to wait for in between
__device__ __forceinline__ Row6 load_row6_direct(const float* __restrict__ in, int64_t x, int64_t W,
                                                 int64_t y0, bool row_ok) {
  const float* row = in + x * W + y0;
  float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float l = 0.0f, r = 0.0f;
  if (row_ok) {
    c = *reinterpret_cast<const float4*>(row);
    l = row[y0 > 0 ? -1 : 0];
    r = row[y0 + 4 < W ? 4 : 3];
  }
  return Row6{{y0 > 0 ? l : 0.0f, c.x, c.y, c.z, c.w, y0 + 4 < W ? r : 0.0f}};
}

// "Plain" rows (round 4).  The shared-reciprocal quotients of soil_math.hpp are the IEEE ones for
// numerators that are +-0 (where the sign of a zero quotient is not looked at) or of magnitude
// 2^-80 .. 2^50.  A kernel whose numerators are DIFFERENCES OF TWO VALUES OF THE WINDOW gets that from
// the values: if every one of them is +-0 or of magnitude 2^-54 .. 2^48, a difference is +-0 or at
// least an ulp of the smaller one (>= 2^-77) and at most 2^49.  That is one test per value LOADED — a
// value takes part in up to nine cells' stencils and several quotients in each — instead of one per
// quotient (QuotWatch: 4 of the 12 instructions of a diagonal neighbour of k_steepest4), and it is
// wave-uniform: the fallback is a scalar branch.  `edge`: lane 0 passes its left, lane 63 its right
// halo column (the other lanes' halo columns are their neighbours' own cells).
__device__ __forceinline__ bool win_row_plain(const float4& c, float edge) {
  constexpr uint32_t kLo = (127u - 54u) << 24, kHi = (127u + 48u) << 24;  // exponents, sign shifted out
  const uint32_t u0 = f2bits(c.x) << 1, u1 = f2bits(c.y) << 1, u2 = f2bits(c.z) << 1, u3 = f2bits(c.w) << 1,
                 u4 = f2bits(edge) << 1;
  const uint32_t lo = min(min(min(u0 - 1u, u1 - 1u), min(u2 - 1u, u3 - 1u)), u4 - 1u);  // a zero wraps to the top
  const uint32_t hi = max(max(max(u0, u1), max(u2, u3)), u4);
  return __ballot(lo < kLo - 1u || hi > kHi) == 0ull;
}

// Rows x - 1, x, x + 1 as a thread walks down its band: start() loads all three, next() moves one
// row down re-using two.  Every lane of the wave must make the calls (they shuffle).
// WATCH: plain() says whether all three rows are plain for the whole wave.
template <int BAND>
struct WinBandRows {  // a work-group walks bands of BAND rows: grid.y bands at a time
  static constexpr int kBand = BAND;
  static dim3 grid(int64_t H, int64_t W) {
    const int64_t bands = (H + BAND - 1) / BAND;
    return dim3(static_cast<unsigned>((W / 4 + kWinBlock - 1) / kWinBlock), static_cast<unsigned>(bands < 65535 ? bands : 65535));
  }
  static __device__ __forceinline__ WinThread thread(int64_t, int64_t W) { return win_thread(W); }
  static __device__ __forceinline__ int64_t band_first(const WinThread&) { return blockIdx.y; }
  static __device__ __forceinline__ bool band_ok(int64_t band, int64_t H) { return band * kBand < H; }
  static __device__ __forceinline__ int64_t band_next(int64_t band) { return band + gridDim.y; }
};
using WinBandShape = WinBandRows<kWinBand>;
struct WinFlatShape {  // a work-group owns one row
  static constexpr int kBand = 1;
  static dim3 grid(int64_t H, int64_t W) { return win_grid_flat(H, W); }
  static __device__ __forceinline__ WinThread thread(int64_t H, int64_t W) { return win_thread_flat(H, W); }
  static __device__ __forceinline__ int64_t band_first(const WinThread& t) { return t.row; }
  static __device__ __forceinline__ bool band_ok(int64_t band, int64_t) { return band >= 0; }
  static __device__ __forceinline__ int64_t band_next(int64_t) { return -1; }
};

template <bool WATCH, class SHAPE = WinBandShape>
struct RowWalkReg : SHAPE {
  static constexpr int kLdsFloats = 4;
  Row6 up, mid, dn;
  bool has_up, has_dn;
  bool p_up = true, p_mid = true, p_dn = true;
  __device__ __forceinline__ bool plain() const { return p_up && p_mid && p_dn; }
  __device__ __forceinline__ bool watch(const Row6& r) const {
    if (!WATCH) return true;
    const int lane = static_cast<int>(threadIdx.x & 63u);
    return win_row_plain(make_float4(r.v[1], r.v[2], r.v[3], r.v[4]), lane == 0 ? r.v[0] : r.v[5]);
  }
  __device__ __forceinline__ void start(const float* __restrict__ in, int64_t x, int64_t H,
                                        int64_t W, int64_t y0) {
    has_up = x > 0;
    has_dn = x + 1 < H;
    if (SHAPE::kBand == 1) {  // a wave that lives for one row: nine independent loads, no shuffle to wait at
      up = load_row6_direct(in, x - 1, W, y0, has_up);
      mid = load_row6_direct(in, x, W, y0, true);
      dn = load_row6_direct(in, x + 1, W, y0, has_dn);
    } else {
      up = load_row6(in, x - 1, W, y0, has_up);
      mid = load_row6(in, x, W, y0, true);
      dn = load_row6(in, x + 1, W, y0, has_dn);
    }
    p_up = watch(up);
    p_mid = watch(mid);
    p_dn = watch(dn);
  }
  __device__ __forceinline__ void next(const float* __restrict__ in, int64_t x, int64_t H,
                                       int64_t W, int64_t y0) {  // x: the row moved onto
    up = mid;
    mid = dn;
    has_up = true;
    has_dn = x + 1 < H;
    dn = load_row6(in, x + 1, W, y0, has_dn);
    p_up = p_mid;
    p_mid = p_dn;
    p_dn = watch(dn);
  }
};
using RowWalk = RowWalkReg<false>;
template <bool WATCH>
using RowWalkFlat = RowWalkReg<WATCH, WinFlatShape>;
template <bool WATCH>
using RowWalkTall = RowWalkReg<WATCH, WinBandRows<kWinBand / 2>>;  // bands of 16 rows
template <bool WATCH>
using RowWalkShort = RowWalkReg<WATCH, WinBandRows<8>>;  // bands of 8 rows

// ---- blocks of a few rows, asked for at once (round 5) -------------------------------------------
//
// tools/microbench/copy_shapes.hip again, with what round 4 had not tried (8192^2, 8 B per cell, us):
//   flat 87.8 | rows3 87.9 | block 2 87.9 | block 4 92.6 | block 8 100.3 | block 16 108.6 | band 8 101.8 |
//   band 16 107.4 | band 32 111.1 | band 64 104.6 | band 128 122.7 | band 32 with two rows in flight 111.3 |
//   band 32, an XCD sweeping its own eighth of the rows 112.5
// The fewer rows a work-group lives for, the faster the plane streams: short-lived work-groups handed
// out in address order keep the chip's accesses on a narrow front of DRAM pages, a band walk has 2048
// fronts a megabyte apart whatever is done about its loads in flight.  A block of R rows asks for its
// R + 2 rows at once (independent 16-byte loads, the two extra rows out of the L2 the vertical
// neighbours fill — work-groups b G + g and (b +- 1) G + g run on the same XCD when G = W / 1024 is a
// multiple of 8), walks them out of registers and is gone.  Round 4's flat shape is the block of ONE
// row: it pays the per-thread set-up (reciprocals, halo columns, index arithmetic) once per row and
// lost to the band walk on every kernel; at R = 4 that cost is a quarter and the plane still streams
// at 5.8 TB/s where the band walk gets 4.8.
// The halo columns: a lane's left / right neighbour cells are its neighbour lanes' (wave shifts on DPP,
// no LDS crossbar); lanes 0 and 63 take the cell beside the wave's 256 from one more load that every
// lane issues (theirs the edge columns, everybody else's a harmless repeat) — no divergent reload.
template <int CTRL>
__device__ __forceinline__ float dpp_wave_shift(float v) {  // 0x138: wave_shr:1 (lane i reads lane i - 1), 0x130: wave_shl:1
  return bits2f(static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(f2bits(v)), CTRL, 0xf, 0xf, false)));
}
// `x` is a row of the grid (the caller clamps); `row_ok`: it is the row asked for (else zeros)
__device__ __forceinline__ Row6 load_row6_dpp(const float* __restrict__ in, int64_t x, int64_t W, int64_t y0,
                                              bool row_ok) {
  const int lane = static_cast<int>(threadIdx.x & 63u);
  const float* row = in + x * W;
  const int64_t ec = lane == 63 ? (y0 + 4 < W ? y0 + 4 : W - 1) : (y0 > 0 ? y0 - 1 : 0);
  float4 c = *reinterpret_cast<const float4*>(row + y0);
  float e = row[ec];
  if (!row_ok) {
    c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    e = 0.0f;
  }
  float l = dpp_wave_shift<0x138>(c.w), r = dpp_wave_shift<0x130>(c.x);
  if (lane == 0) l = y0 > 0 ? e : 0.0f;
  if (lane == 63) r = y0 + 4 < W ? e : 0.0f;
  return Row6{{l, c.x, c.y, c.z, c.w, r}};
}
inline int64_t win_groups_of(int64_t W) { return (W / 4 + kWinBlock - 1) / kWinBlock; }
template <int R>
struct WinBlockRows {  // a work-group owns R rows of 1024 columns; work-groups in address order
  static constexpr int kBand = R;
  static dim3 grid(int64_t H, int64_t W) {
    return dim3(static_cast<unsigned>(win_groups_of(W) * ((H + R - 1) / R)));
  }
  static __device__ __forceinline__ WinThread thread(int64_t, int64_t W) {
    const uint32_t G = static_cast<uint32_t>((W / 4 + kWinBlock - 1) / kWinBlock);
    const uint32_t b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int64_t y = (static_cast<int64_t>(g) * kWinBlock + threadIdx.x) * 4;
    return WinThread{y < W ? y : W - 4, y < W, static_cast<int64_t>(g), static_cast<int64_t>(b)};
  }
  static __device__ __forceinline__ int64_t band_first(const WinThread& t) { return t.row; }
  static __device__ __forceinline__ bool band_ok(int64_t band, int64_t H) { return band >= 0 && band * R < H; }
  static __device__ __forceinline__ int64_t band_next(int64_t) { return -1; }
};
// Round 5, second half: the same blocks with the work-group's four waves STACKED — wave w owns rows
// (4 b + w) R .. + R of one 256-column strip instead of a quarter of a 1024-column row piece.  The two extra
// rows a wave asks for are then its neighbour waves' own rows, on their way into the same CU's vector
// cache, instead of rows of work-groups on other CUs (out of L2).  SOIL_WIN_SHAPE 7 / 8: R = 2 / 4.
template <int R>
struct WinStackRows {
  static constexpr int kBand = R;
  static constexpr int kWaves = kWinBlock / 64;
  static int64_t strips(int64_t W) { return (W / 4 + 63) / 64; }
  static dim3 grid(int64_t H, int64_t W) {
    return dim3(static_cast<unsigned>(strips(W) * ((H + R * kWaves - 1) / (R * kWaves))));
  }
  static __device__ __forceinline__ WinThread thread(int64_t, int64_t W) {
    const uint32_t G = static_cast<uint32_t>((W / 4 + 63) / 64);
    const uint32_t b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int64_t y = (static_cast<int64_t>(g) * 64 + (threadIdx.x & 63u)) * 4;
    WinThread t{y < W ? y : W - 4, y < W, static_cast<int64_t>(g), static_cast<int64_t>(b) * kWaves + (threadIdx.x >> 6)};
    t.wave_y0 = static_cast<int64_t>(g) * 256;
    return t;
  }
  static __device__ __forceinline__ int64_t band_first(const WinThread& t) { return t.row; }
  static __device__ __forceinline__ bool band_ok(int64_t band, int64_t H) { return band >= 0 && band * R < H; }
  static __device__ __forceinline__ int64_t band_next(int64_t) { return -1; }
};
template <bool WATCH, int R, class SHAPE = WinBlockRows<R>>
struct RowBlockReg : SHAPE {
  static constexpr int kLdsFloats = 4;
  Row6 up, mid, dn;
  Row6 ahead[R > 1 ? R - 1 : 1];  // rows x + 2 .. x + R of the block's first row x, taken by next()
  bool has_up, has_dn;
  bool p_up = true, p_mid = true, p_dn = true, p_ahead[R > 1 ? R - 1 : 1];
  __device__ __forceinline__ bool plain() const { return p_up && p_mid && p_dn; }
  __device__ __forceinline__ bool watch(const Row6& r) const {
    if (!WATCH) return true;
    const int lane = static_cast<int>(threadIdx.x & 63u);
    return win_row_plain(make_float4(r.v[1], r.v[2], r.v[3], r.v[4]), lane == 0 ? r.v[0] : r.v[5]);
  }
  __device__ __forceinline__ void start(const float* __restrict__ in, int64_t x, int64_t H, int64_t W, int64_t y0) {
    has_up = x > 0;
    has_dn = x + 1 < H;
    auto row = [&](int64_t xr) { return load_row6_dpp(in, xr < 0 ? 0 : (xr < H ? xr : H - 1), W, y0, xr >= 0 && xr < H); };
    up = row(x - 1);
    mid = row(x);
    dn = row(x + 1);
#pragma unroll
    for (int i = 0; i < R - 1; ++i) ahead[i] = row(x + 2 + i);
    p_up = watch(up);
    p_mid = watch(mid);
    p_dn = watch(dn);
#pragma unroll
    for (int i = 0; i < R - 1; ++i) p_ahead[i] = watch(ahead[i]);
  }
  __device__ __forceinline__ void next(const float* __restrict__, int64_t x, int64_t H, int64_t, int64_t) {
    up = mid;
    mid = dn;
    dn = ahead[0];
    p_up = p_mid;
    p_mid = p_dn;
    p_dn = p_ahead[0];
#pragma unroll
    for (int i = 0; i + 1 < R - 1; ++i) {
      ahead[i] = ahead[i + 1];
      p_ahead[i] = p_ahead[i + 1];
    }
    has_up = true;
    has_dn = x + 1 < H;
  }
};
template <bool WATCH>
using RowWalkBlock4 = RowBlockReg<WATCH, 4>;
template <bool WATCH>
using RowWalkBlock2 = RowBlockReg<WATCH, 2>;
template <bool WATCH>
using RowWalkStack2 = RowBlockReg<WATCH, 2, WinStackRows<2>>;
template <bool WATCH>
using RowWalkStack4 = RowBlockReg<WATCH, 4, WinStackRows<4>>;

// ---- the same walk with the rows landing in LDS (round 4) ---------------------------------------
//
// RowWalk has ONE 16-byte load per wave in flight and waits for it before the shuffles: a launch at
// 8192^2 is one generation of 8192 waves, 8 MiB in flight, which at the loaded latency of HBM is
// 4-5 TB/s (the 41-52 % rows of profiles/r04_final/bench_stencils.txt).  More rows in flight through
// registers cost the waves that hide the latency (round 3).  RowWalkDma keeps kWinDmaDepth rows in
// flight per wave WITHOUT registers: `global_load_lds_dwordx4` writes a wave's 1 KiB of a row
// straight into a wave-private ring in LDS (one more 4-byte DMA brings the two columns beside the
// wave's 256), and the walk reads its six values per row from there — no shuffles, no reloads by
// lanes 0 and 63.  The ring is private to the wave: no barrier, only the wave's own vmcnt.
//
// hipcc counts neither the DMA nor its completion (the statement is inline asm): the walk waits
// itself.  vmcnt retires in issue order and the kernel's stores share the counter; the wait names
// only the DMAs queued after the row it needs (two per row), which is exact when no store was
// issued since and merely early otherwise (a store after a DMA only makes the wait cover more).
constexpr int kWinDmaDepth = 4;            // rows queued ahead per wave
constexpr int kWinDmaRow = 256 + 4;        // floats per ring slot: the row piece, the two edge columns, pad
constexpr int kWinDmaLds = (kWinBlock / 64) * kWinDmaDepth * kWinDmaRow;  // floats per work-group

template <int N>
__device__ __forceinline__ void win_dma_wait() {  // vmcnt <= N (gfx9 encoding: [3:0] and [15:14])
  __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}

template <bool WATCH>
struct RowWalkLds : WinBandShape {
  static constexpr int kLdsFloats = kWinDmaLds;
  Row6 up, mid, dn;
  bool has_up, has_dn;
  bool p_up = true, p_mid = true, p_dn = true, p_take = true;
  __device__ __forceinline__ bool plain() const { return p_up && p_mid && p_dn; }
  float* ring;          // this wave's kWinDmaDepth slots
  uint32_t ring_lds;    // the same as an LDS byte address (wave-uniform)
  int64_t edge_col;     // lane 0: the column left of the wave's piece, lane 1: right of it (clamped)
  int64_t x_queue, x_last;  // next row to queue; last row the band needs
  int slot_queue, slot_read, queued;  // ring positions; rows queued and not yet read

  __device__ __forceinline__ void bind(float* lds, int64_t W) {
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    ring = lds + wave * (kWinDmaDepth * kWinDmaRow);
    ring_lds = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ring)));
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const int64_t first = (static_cast<int64_t>(blockIdx.x) * kWinBlock + (threadIdx.x & ~63u)) * 4;
    const int64_t y_first = first < W ? first : W - 4;
    const int64_t last = first + 63 * 4;
    const int64_t y_last = last < W ? last : W - 4;
    edge_col = lane == 0 ? (y_first > 0 ? y_first - 1 : 0) : (y_last + 4 < W ? y_last + 4 : W - 1);
  }
  __device__ __forceinline__ void queue(const float* __restrict__ in, int64_t H, int64_t W, int64_t y0) {
    const int64_t xr = x_queue < 0 ? 0 : (x_queue < H ? x_queue : H - 1);  // a row outside: any row, never looked at
    const float* row = in + xr * W;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(
        ring_lds + static_cast<uint32_t>(slot_queue) * static_cast<uint32_t>(kWinDmaRow * sizeof(float)));
    uint32_t keep;
    const float* src = row + y0;
    // lgkmcnt(0): the slot's previous row has been read out before the DMA may overwrite it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    const float* esrc = row + edge_col;
    const uint32_t edst = dst + 256u * sizeof(float);
    if ((threadIdx.x & 63u) < 2u)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(esrc), "s"(edst) : "memory");
    ++x_queue;
    slot_queue = slot_queue + 1 == kWinDmaDepth ? 0 : slot_queue + 1;
    ++queued;
  }
  // the oldest queued row, once it has landed; queues the next row the band needs into its slot
  __device__ __forceinline__ Row6 take(const float* __restrict__ in, int64_t H, int64_t W, int64_t y0) {
    const int after = queued - 1;  // rows queued behind the one wanted: two DMAs each
    if (after >= 3) win_dma_wait<6>();
    else if (after == 2) win_dma_wait<4>();
    else if (after == 1) win_dma_wait<2>();
    else win_dma_wait<0>();
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const float* s = ring + slot_read * kWinDmaRow;
    const float4 c = *reinterpret_cast<const float4*>(s + 4 * lane);
    const float l = s[lane == 0 ? 256 : 4 * lane - 1];
    const float r = s[lane == 63 ? 257 : 4 * lane + 4];
    slot_read = slot_read + 1 == kWinDmaDepth ? 0 : slot_read + 1;
    --queued;
    if (x_queue <= x_last) queue(in, H, W, y0);
    if (WATCH) p_take = win_row_plain(c, lane == 0 ? l : r);
    return Row6{{l, c.x, c.y, c.z, c.w, r}};
  }
  __device__ __forceinline__ void start(const float* __restrict__ in, int64_t x, int64_t H,
                                        int64_t W, int64_t y0) {
    has_up = x > 0;
    has_dn = x + 1 < H;
    x_queue = x - 1;
    const int64_t band_end = (x + kWinBand < H) ? x + kWinBand : H;  // SOIL_WIN_ROWS' x_end
    x_last = band_end;                                             // row x_end - 1 looks at row x_end
    slot_queue = slot_read = queued = 0;
    static_assert(kWinDmaDepth == 4, "take() spells the waits of a depth of four out");
    for (int i = 0; i < kWinDmaDepth; ++i)  // every row queued is taken: no DMA outlives the wave's LDS
      if (x_queue <= x_last) queue(in, H, W, y0);
    up = take(in, H, W, y0);
    p_up = p_take;
    mid = take(in, H, W, y0);
    p_mid = p_take;
    dn = take(in, H, W, y0);
    p_dn = p_take;
  }
  __device__ __forceinline__ void next(const float* __restrict__ in, int64_t x, int64_t H,
                                       int64_t W, int64_t y0) {
    up = mid;
    mid = dn;
    has_up = true;
    has_dn = x + 1 < H;
    dn = take(in, H, W, y0);
    p_up = p_mid;
    p_mid = p_dn;
    p_dn = p_take;
  }
};
using RowWalkDma = RowWalkLds<false>;

// `SOIL_WIN_WALK(Walk, w, W);` declares walk `w` of either kind with the LDS it needs
template <bool WATCH, class SHAPE>
__device__ __forceinline__ void win_bind(RowWalkReg<WATCH, SHAPE>&, float*, int64_t) {}
template <bool WATCH, int R, class SHAPE>
__device__ __forceinline__ void win_bind(RowBlockReg<WATCH, R, SHAPE>&, float*, int64_t) {}
template <bool WATCH>
__device__ __forceinline__ void win_bind(RowWalkLds<WATCH>& w, float* lds, int64_t W) { w.bind(lds, W); }
template <class Walk>
constexpr int win_lds_floats() {
  return Walk::kLdsFloats;
}
#define SOIL_WIN_WALK(Walk, w, W)                                               \
  __shared__ __attribute__((aligned(16))) float w##_lds[::soil::win_lds_floats<Walk>()]; \
  Walk w;                                                                       \
  ::soil::win_bind(w, w##_lds, W)
// SOIL_WIN_SHAPE (A/B): 0 the band walk through registers, 1 the band walk through LDS, 2 the flat
// shape, 3 bands of 16 rows through registers (measured: as 32; 64 rows, half the waves: 20-30 % slower),
// 4 / 5 blocks of four / two rows asked for at once (round 5); unset: the kernel's own default
inline int win_shape(int dflt) {
  static const int env = [] {
    const char* e = std::getenv("SOIL_WIN_SHAPE");
    return (e && e[0] >= '0' && e[0] <= '8') ? e[0] - '0' : -1;
  }();
  return env >= 0 ? env : dflt;
}
// A kernel's own choice by grid (round 5; ms at 8192^2, band walk | blocks of four rows | of two, one box,
// tools/bench_stencils.py: gradient 0.162 | 0.154 | 0.140, slope 0.165 | 0.178 | 0.152, negslope 0.119 |
// 0.110 | 0.119, direction 0.140 | 0.131 | 0.139, random_weighted 0.260 | 0.223 | 0.230, laplacian D = 1
// 0.109 | 0.113 | 0.122, steepest D8 0.134 | 0.318 (its 64 registers spill) | 0.170): `large`.  A grid
// whose band walk would be fewer than 2048 work-groups — 4096^2 has 512, half a generation of the chip —
// takes `small` instead: at 4096^2 gradient 0.062 | 0.042 | 0.042, slope 0.070 | 0.045 | 0.045, negslope
// 0.048 | 0.033 | 0.034, direction 0.056 | 0.039 | 0.038.
inline int win_shape_for(int large, int small, int64_t H, int64_t W) {
  const int64_t band_groups = ((W / 4 + kWinBlock - 1) / kWinBlock) * ((H + kWinBand - 1) / kWinBand);
  return win_shape(band_groups < 2048 ? small : large);
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

// for (x over the rows of this work-group) with `w` holding rows x - 1 .. x + 1; `t`: w.thread(H, W)
#define SOIL_WIN_ROWS(x, w, in, H, W, t)                                                         \
  for (int64_t x##_band = (w).band_first(t); (w).band_ok(x##_band, H);                           \
       x##_band = (w).band_next(x##_band))                                                       \
    for (int64_t x = x##_band * (w).kBand, x##_end = (x + (w).kBand < (H)) ? x + (w).kBand : (H), \
                 x##_go = ((w).start(in, x, H, W, (t).y0), 1);                                   \
         x < x##_end && x##_go; ++x, (x < x##_end ? (w).next(in, x, H, W, (t).y0) : (void)0))

}  // namespace soil
