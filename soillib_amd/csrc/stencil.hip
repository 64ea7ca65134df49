// stencil.hip — gradient / negslope / laplacian (grad.cu), separable Gaussian
// blur (filter.cu) and the surface-normal map (op/normal.hpp).
#include "common.hpp"
#include "window.hpp"

namespace soil {

constexpr int kSBlock = 256;

// __gradient, grad.cu:22-87
__global__ void __launch_bounds__(kSBlock)
    k_gradient(float2* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t W,
               Scale2 s) {
  const int64_t y = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x;
  if (y >= W) return;
  const float nan = __builtin_nanf("");
  SOIL_ROW_LOOP(x, H) {
    const int64_t n = x * W + y;
    const float h = in[n];
    const float hn0 = (x - 1 < 0) ? nan : in[n - W];  // :35-38
    const float hp0 = (x + 1 >= H) ? nan : in[n + W];
    const float h0n = (y - 1 < 0) ? nan : in[n - 1];
    const float h0p = (y + 1 >= W) ? nan : in[n + 1];
    const float gxn = (h - hn0) / s.x;    // :46
    const float gyn = (h - h0n) / s.y;    // :50
    const float gxp = (hp0 - h) / s.x;    // :54
    const float gyp = (h0p - h) / s.y;    // :58
    float gx = 0.5f * (hp0 - hn0) / s.x;  // :62
    float gy = 0.5f * (h0p - h0n) / s.y;  // :63
    if (gx != gx) gx = gxn;               // :65-67
    if (gx != gx) gx = gxp;
    if (gx != gx) gx = 0.0f;
    if (gy != gy) gy = gyn;  // :69-71
    if (gy != gy) gy = gyp;
    if (gy != gy) gy = 0.0f;
    out[n] = make_float2(gx, gy);  // :84-85
  }
}

// __negslope, grad.cu:101-131
__global__ void __launch_bounds__(kSBlock)
    k_negslope(float* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t W,
               Scale2 s) {
  const int64_t y = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x;
  if (y >= W) return;
  SOIL_ROW_LOOP(x, H) {
    const int64_t n = x * W + y;
    const float h = in[n];
    float gx = 0.0f;  // :120-122, glm::max(a, b) = (a < b) ? b : a
    if (x - 1 >= 0) {
      const float c = (h - in[n - W]) / s.x;
      gx = (gx < c) ? c : gx;
    }
    if (x + 1 < H) {
      const float c = (h - in[n + W]) / s.x;
      gx = (gx < c) ? c : gx;
    }
    float gy = 0.0f;  // :124-126
    if (y - 1 >= 0) {
      const float c = (h - in[n - 1]) / s.y;
      gy = (gy < c) ? c : gy;
    }
    if (y + 1 < W) {
      const float c = (h - in[n + 1]) / s.y;
      gy = (gy < c) ? c : gy;
    }
    out[n] = sqrtf(gx * gx + gy * gy);  // :129
  }
}

// __laplacian<D>, grad.cu:147-183 — one thread per (cell, channel)
template <int D>
__global__ void __launch_bounds__(kSBlock)
    k_laplacian(float* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t W,
                Scale2 s) {
  const int64_t f = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x;  // float of the row
  if (f >= W * D) return;
  const int64_t y = f / D;
  const int c = static_cast<int>(f % D);
  const float hx = (1.0f / s.x / s.x);  // :175
  const float hy = (1.0f / s.y / s.y);  // :176
  SOIL_ROW_LOOP(x, H) {
    const int64_t n = x * W + y;
    auto at = [&](int dx, int dy) -> float {  // clamp-to-self, :166-173
      const int64_t nx = x + dx, ny = y + dy;
      if (nx < 0 || nx >= H || ny < 0 || ny >= W) return in[D * n + c];
      return in[D * (nx * W + ny) + c];
    };
    const float v00 = in[D * n + c];
    const float vn0 = at(-1, 0), vp0 = at(1, 0), v0n = at(0, -1), v0p = at(0, 1);
    const float vnn = at(-1, -1), vpp = at(1, 1), vpn = at(1, -1), vnp = at(-1, 1);
    const float LH = (vn0 - v00) * hx + (vp0 - v00) * hx + (v0n - v00) * hy + (v0p - v00) * hy;  // :178
    const float LD = 0.5f * (vnn - v00) * hx + 0.5f * (vpp - v00) * hx + 0.5f * (vpn - v00) * hy +
                     0.5f * (vnp - v00) * hy;   // :179
    out[D * n + c] = 0.5f * LH + 0.5f * LD;    // :181
  }
}

// ---- the same three stencils, four cells per thread (window.hpp) ------------------------------

// The divisions of these kernels are by the cell size: with FAST they are quot() over a reciprocal
// refined once per thread, checked per group of four cells (QuotWatch, soil_math.hpp), and redone
// as written when in doubt — same bits either way.  `div(a, b, rb)`: a / b.
struct DivWritten {
  __device__ __forceinline__ float operator()(float a, float b, const Recip&) const { return a / b; }
};
// (for the differences of two values of a plain window, window.hpp: nothing to check per quotient;
// the sign of a zero quotient must not be seen)
struct DivPlain {
  __device__ __forceinline__ float operator()(float a, float, const Recip& rb) const { return quot(a, rb); }
};
// a scale the shared-reciprocal quotient may divide by (positive, 2^-40 .. 2^40)
static bool plain_scale(float v) { return v >= 0x1p-40f && v <= 0x1p40f; }

// one of a window kernel's builds: the flat shape, the band walk through LDS or through registers
// (window.hpp; SOIL_WIN_SHAPE picks, `dflt` is the kernel's own choice), and the written-out
// divisions for scales out of the plain range
template <class Walk, class KF, class... A>
static void win_launch_as(KF kernel, int64_t H, int64_t W, hipStream_t st, A... a) {
  kernel<<<Walk::grid(H, W), kWinBlock, 0, st>>>(a...);
}
#define SOIL_WIN_LAUNCH(large, small, fast, K_FAST, K_WRITTEN, WATCH, H, W, st, ...)                 \
  do {                                                                                              \
    if (!(fast)) win_launch_as<RowWalk>(K_WRITTEN<RowWalk>, H, W, st, __VA_ARGS__);                 \
    else switch (win_shape_for(large, small, H, W)) {                                               \
      case 0: win_launch_as<RowWalkReg<WATCH>>(K_FAST<RowWalkReg<WATCH>>, H, W, st, __VA_ARGS__); break;   \
      case 1: win_launch_as<RowWalkLds<WATCH>>(K_FAST<RowWalkLds<WATCH>>, H, W, st, __VA_ARGS__); break;   \
      case 3: win_launch_as<RowWalkTall<WATCH>>(K_FAST<RowWalkTall<WATCH>>, H, W, st, __VA_ARGS__); break; \
      case 4: win_launch_as<RowWalkBlock4<WATCH>>(K_FAST<RowWalkBlock4<WATCH>>, H, W, st, __VA_ARGS__); break; \
      case 5: win_launch_as<RowWalkBlock2<WATCH>>(K_FAST<RowWalkBlock2<WATCH>>, H, W, st, __VA_ARGS__); break; \
      case 6: win_launch_as<RowWalkShort<WATCH>>(K_FAST<RowWalkShort<WATCH>>, H, W, st, __VA_ARGS__); break; \
      case 7: win_launch_as<RowWalkStack2<WATCH>>(K_FAST<RowWalkStack2<WATCH>>, H, W, st, __VA_ARGS__); break; \
      case 8: win_launch_as<RowWalkStack4<WATCH>>(K_FAST<RowWalkStack4<WATCH>>, H, W, st, __VA_ARGS__); break; \
      default: win_launch_as<RowWalkFlat<WATCH>>(K_FAST<RowWalkFlat<WATCH>>, H, W, st, __VA_ARGS__); break; \
    }                                                                                               \
  } while (0)

// __gradient, grad.cu:22-87, for the four cells of a thread
template <typename DIV, class Walk>
__device__ __forceinline__ void gradient_group(float of[8], const Walk& w, const WinThread& t,
                                               int64_t W, Scale2 s, const Recip& rx, const Recip& ry,
                                               DIV div) {
  const float nan = __builtin_nanf("");
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float h = w.mid.v[k + 1];
    const float hn0 = w.has_up ? w.up.v[k + 1] : nan;  // :35-38
    const float hp0 = w.has_dn ? w.dn.v[k + 1] : nan;
    const float h0n = (k > 0 || t.y0 > 0) ? w.mid.v[k] : nan;
    const float h0p = (k < 3 || t.y0 + 4 < W) ? w.mid.v[k + 2] : nan;
    const float gxn = div(h - hn0, s.x, rx);            // :46
    const float gyn = div(h - h0n, s.y, ry);            // :50
    const float gxp = div(hp0 - h, s.x, rx);            // :54
    const float gyp = div(h0p - h, s.y, ry);            // :58
    float gx = div(0.5f * (hp0 - hn0), s.x, rx);        // :62
    float gy = div(0.5f * (h0p - h0n), s.y, ry);        // :63
    if (gx != gx) gx = gxn;                             // :65-67
    if (gx != gx) gx = gxp;
    if (gx != gx) gx = 0.0f;
    if (gy != gy) gy = gyn;  // :69-71
    if (gy != gy) gy = gyp;
    if (gy != gy) gy = 0.0f;
    of[2 * k] = gx;  // :84-85
    of[2 * k + 1] = gy;
  }
}

template <bool FAST, class Walk>
__global__ void __launch_bounds__(kWinBlock)
    k_gradient4(float2* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t W, Scale2 s) {
  __shared__ float4 s_tile[kWinBlock / 64][128];
  const WinThread t = Walk::thread(H, W);
  const Recip rx = recip(s.x), ry = recip(s.y);
  SOIL_WIN_WALK(Walk, w, W);
  SOIL_WIN_ROWS(x, w, in, H, W, t) {
    float4 o[2];
    float* of = reinterpret_cast<float*>(o);
    // at the grid's edge the NaN sentinels of :35-38 run through the quotients: written-out there
    bool redo = !FAST || !w.has_up || !w.has_dn || t.y0 == 0 || t.y0 + 4 >= W;
    if (!redo) {
      // inside the grid all four neighbours exist: the gradient is the central difference (:62-63)
      // and the one-sided ones (:46-58) are only its stand-ins for a NaN — which the check below
      // sends to the written-out path together with everything else out of the plain range
      QuotWatch watch;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float ax = 0.5f * (w.dn.v[k + 1] - w.up.v[k + 1]), ay = 0.5f * (w.mid.v[k + 2] - w.mid.v[k]);
        of[2 * k] = watch(quot(ax, rx), ax);  // (the sign of a zero gradient is stored: see QuotWatch)
        of[2 * k + 1] = watch(quot(ay, ry), ay);
      }
      redo = watch.doubtful();
    }
    if (redo) gradient_group(of, w, t, W, s, rx, ry, DivWritten{});
    // the wave's 256 cells start at column y0 - 4 * lane (lanes past the row's end sit on the last group)
    store_pair_contiguous(reinterpret_cast<float4*>(out + x * W + t.wave_y0), o[0], o[1],
                          s_tile[threadIdx.x >> 6], t.live);
  }
}

// __negslope, grad.cu:101-131
template <typename DIV, class Walk>
__device__ __forceinline__ void negslope_group(float of[4], const Walk& w, const WinThread& t,
                                               int64_t W, Scale2 s, const Recip& rx, const Recip& ry,
                                               DIV div) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float h = w.mid.v[k + 1];
    float gx = 0.0f;  // :120-122, glm::max(a, b) = (a < b) ? b : a
    if (w.has_up) {
      const float c = div(h - w.up.v[k + 1], s.x, rx);
      gx = (gx < c) ? c : gx;
    }
    if (w.has_dn) {
      const float c = div(h - w.dn.v[k + 1], s.x, rx);
      gx = (gx < c) ? c : gx;
    }
    float gy = 0.0f;  // :124-126
    if (k > 0 || t.y0 > 0) {
      const float c = div(h - w.mid.v[k], s.y, ry);
      gy = (gy < c) ? c : gy;
    }
    if (k < 3 || t.y0 + 4 < W) {
      const float c = div(h - w.mid.v[k + 2], s.y, ry);
      gy = (gy < c) ? c : gy;
    }
    of[k] = sqrtf(gx * gx + gy * gy);  // :129
  }
}

template <bool FAST, class Walk>
__global__ void __launch_bounds__(kWinBlock)
    k_negslope4(float* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t W, Scale2 s) {
  const WinThread t = Walk::thread(H, W);
  const Recip rx = recip(s.x), ry = recip(s.y);
  SOIL_WIN_WALK(Walk, w, W);
  SOIL_WIN_ROWS(x, w, in, H, W, t) {
    float4 o;
    float* of = reinterpret_cast<float*>(&o);
    // the quotients are of differences of the window's values and only compared with zero and
    // squared: shared-reciprocal ones on a plain window (wave-uniform), as written otherwise
    if (FAST && w.plain()) negslope_group(of, w, t, W, s, rx, ry, DivPlain{});
    else negslope_group(of, w, t, W, s, rx, ry, DivWritten{});
    if (t.live) *reinterpret_cast<float4*>(out + x * W + t.y0) = o;
  }
}

// __laplacian<1>, grad.cu:147-183
template <class Walk>
__global__ void __launch_bounds__(kWinBlock)
    k_laplacian4(float* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t W, float hx, float hy) {
  const WinThread t = Walk::thread(H, W);  // hx = 1 / s.x / s.x, hy likewise (:175-176), divided on the host
  SOIL_WIN_WALK(Walk, w, W);
  SOIL_WIN_ROWS(x, w, in, H, W, t) {
    float4 o;
    float* of = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v00 = w.mid.v[k + 1];
      const bool l = k > 0 || t.y0 > 0, r = k < 3 || t.y0 + 4 < W;
      // clamp-to-self, :166-173: a neighbour outside the grid reads as the centre
      const float vn0 = w.has_up ? w.up.v[k + 1] : v00, vp0 = w.has_dn ? w.dn.v[k + 1] : v00;
      const float v0n = l ? w.mid.v[k] : v00, v0p = r ? w.mid.v[k + 2] : v00;
      const float vnn = (w.has_up && l) ? w.up.v[k] : v00, vpp = (w.has_dn && r) ? w.dn.v[k + 2] : v00;
      const float vpn = (w.has_dn && l) ? w.dn.v[k] : v00, vnp = (w.has_up && r) ? w.up.v[k + 2] : v00;
      const float LH = (vn0 - v00) * hx + (vp0 - v00) * hx + (v0n - v00) * hy + (v0p - v00) * hy;  // :178
      const float LD = 0.5f * (vnn - v00) * hx + 0.5f * (vpp - v00) * hx + 0.5f * (vpn - v00) * hy +
                       0.5f * (vnp - v00) * hy;  // :179
      of[k] = 0.5f * LH + 0.5f * LD;              // :181
    }
    if (t.live) *reinterpret_cast<float4*>(out + x * W + t.y0) = o;
  }
}

// __laplacian<2>, grad.cu:147-183, on a two-channel plane: a thread owns four cells = eight
// consecutive floats of the (H, 2 W) matrix; the cells left and right of them come from the
// neighbouring lanes (lanes 0 and 63 reload 8 bytes), three rows slide through registers.
struct Row12 {
  float v[12];  // [0,1] the cell left of the thread's four, [2..9] its own, [10,11] the cell right
};
__device__ __forceinline__ Row12 load_row12(const float* __restrict__ in, int64_t x, int64_t W,
                                            int64_t y0, bool row_ok) {
  const int lane = static_cast<int>(threadIdx.x & 63u);
  const float* row = in + x * W * 2;
  float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a;
  if (row_ok) {
    a = *reinterpret_cast<const float4*>(row + 2 * y0);
    b = *reinterpret_cast<const float4*>(row + 2 * y0 + 4);
  }
  float l0 = __shfl_up(b.z, 1, 64), l1 = __shfl_up(b.w, 1, 64);
  float r0 = __shfl_down(a.x, 1, 64), r1 = __shfl_down(a.y, 1, 64);
  if (lane == 0) {
    const bool ok = row_ok && y0 > 0;
    l0 = ok ? row[2 * y0 - 2] : 0.0f;
    l1 = ok ? row[2 * y0 - 1] : 0.0f;
  }
  if (lane == 63) {
    const bool ok = row_ok && y0 + 4 < W;
    r0 = ok ? row[2 * y0 + 8] : 0.0f;
    r1 = ok ? row[2 * y0 + 9] : 0.0f;
  }
  return Row12{{l0, l1, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, r0, r1}};
}

template <int BAND>
__global__ void __launch_bounds__(kWinBlock)
    k_laplacian4x2(float* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t W, Scale2 s) {
  __shared__ float4 s_tile[kWinBlock / 64][128];
  const WinThread t = win_thread(W);
  const float hx = (1.0f / s.x / s.x);  // :175
  const float hy = (1.0f / s.y / s.y);  // :176
  const int64_t wave_y0 = (static_cast<int64_t>(blockIdx.x) * kWinBlock + (threadIdx.x & ~63u)) * 4;
  for (int64_t band = blockIdx.y; band * BAND < H; band += gridDim.y) {
    int64_t x = band * BAND;
    const int64_t x_end = (x + BAND < H) ? x + BAND : H;
    bool has_up = x > 0, has_dn = x + 1 < H;
    Row12 up = load_row12(in, x - 1, W, t.y0, has_up), mid = load_row12(in, x, W, t.y0, true);
    Row12 dn = load_row12(in, x + 1, W, t.y0, has_dn);
    for (;;) {
      float4 o[2];
      float* of = reinterpret_cast<float*>(o);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool l = k > 0 || t.y0 > 0, r = k < 3 || t.y0 + 4 < W;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int i = 2 + 2 * k + c;
          const float v00 = mid.v[i];
          // clamp-to-self, :166-173: a neighbour outside the grid reads as the centre
          const float vn0 = has_up ? up.v[i] : v00, vp0 = has_dn ? dn.v[i] : v00;
          const float v0n = l ? mid.v[i - 2] : v00, v0p = r ? mid.v[i + 2] : v00;
          const float vnn = (has_up && l) ? up.v[i - 2] : v00, vpp = (has_dn && r) ? dn.v[i + 2] : v00;
          const float vpn = (has_dn && l) ? dn.v[i - 2] : v00, vnp = (has_up && r) ? up.v[i + 2] : v00;
          const float LH = (vn0 - v00) * hx + (vp0 - v00) * hx + (v0n - v00) * hy + (v0p - v00) * hy;  // :178
          const float LD = 0.5f * (vnn - v00) * hx + 0.5f * (vpp - v00) * hx + 0.5f * (vpn - v00) * hy +
                           0.5f * (vnp - v00) * hy;  // :179
          of[2 * k + c] = 0.5f * LH + 0.5f * LD;      // :181
        }
      }
      store_pair_contiguous(reinterpret_cast<float4*>(out + (x * W + wave_y0) * 2), o[0], o[1],
                            s_tile[threadIdx.x >> 6], t.live);
      if (++x >= x_end) break;
      up = mid;
      mid = dn;
      has_up = true;
      has_dn = x + 1 < H;
      dn = load_row12(in, x + 1, W, t.y0, has_dn);
    }
  }
}

// __gaussian_blur / __blur, filter.cu:24-70.  The 33 tap weights depend on
// sigma only; the host evaluates them once with the same expression as
// filter.cu:47-48 instead of once per tap per cell.
struct BlurWeights {
  float w[33];
};

// __blur<T>, filter.cu:24-70: 33 taps along one axis, index clamped to the edge,
// accumulated in the order k = -16..16 (:35-54).  `val += src * kernel` (:50) is stated as the fused
// multiply-add nvcc makes of it (its default -fmad=true contracts the statement; this build and the
// oracle compile with contraction off, so the fma is written out on both sides): one rounding per
// tap, as on the reference's hardware, and half the instructions.  Two launch shapes, both on the
// plane seen as an (H, W*C) matrix of floats:
//
//  * along axis 0 (rows): a thread owns one float column of a 32-row band, reads the
//    64 rows it needs (coalesced across the wave) into registers and produces its 32
//    outputs from them — no LDS, 33 taps from registers.  (Bands of 2 .. 8 such chunks, the next
//    chunk's 32 new rows on their way while one is computed, read less — and took as long or
//    longer: 116 / 117 / 125 us at 8192^2 for 1 / 2 / 8 chunks.)
//  * along axis 1 (the contiguous one): a work-group stages 1024-float segments of kBlurRows
//    rows plus their 16-cell aprons in LDS (clamped per cell, channels interleaved) — all of a
//    thread's 16-byte loads issued before the first is used: with one row per work-group the pass
//    had 8 MiB in flight chip-wide and ran at 2.6 TB/s (now 5.9); a thread then pulls the 4 + 32 C
//    consecutive floats its four consecutive outputs need into registers with 16-byte LDS reads
//    (9 or 17 instead of 132 four-byte ones) and stores its outputs as one 16-byte word.
//
// The one-thread-per-output form it replaces issued 33 global loads per output
// (1.4-1.6 ms per pass at 8192^2, 4.5 % of the HBM roofline).
constexpr int kBlurBand = 32;  // output rows per thread of the axis-0 pass
constexpr int kBlurRows = 8;   // rows per work-group of the axis-1 pass

__global__ void __launch_bounds__(kSBlock)
    k_blur_rows(float* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t WC,
                BlurWeights bw) {
  const int64_t col = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x;
  const int64_t x0 = static_cast<int64_t>(blockIdx.y) * kBlurBand;
  if (col >= WC) return;
  float v[kBlurBand + 32];
#pragma unroll
  for (int r = 0; r < kBlurBand + 32; ++r) {
    int64_t x = x0 + r - 16;  // :39-43
    if (x < 0) x = 0;
    if (x > H - 1) x = H - 1;
    v[r] = in[x * WC + col];
  }
#pragma unroll
  for (int r = 0; r < kBlurBand; ++r) {
    if (x0 + r >= H) break;
    float val = 0.0f;  // :35
#pragma unroll
    for (int k = 0; k < 33; ++k) val = __builtin_fmaf(v[r + k], bw.w[k], val);  // :49-50
    out[(x0 + r) * WC + col] = val;                           // :54
  }
}

// WIDE (decided per work-group from launch constants only, so that the compiler keeps the two
// forms apart instead of merging them into four-byte accesses with selected addresses): rows are
// 16-byte aligned and the whole segment lies inside the row.
template <int C, bool WIDE>
__device__ __forceinline__ void blur_cols_group(float (*seg)[4 * kSBlock + 32 * C],
                                                float* __restrict__ out,
                                                const float* __restrict__ in, int64_t H, int64_t W,
                                                int64_t row0, int64_t f0, const BlurWeights& bw) {
  constexpr int kSeg = 4 * kSBlock;
  const int64_t WC = W * C;
  const int i0 = 4 * static_cast<int>(threadIdx.x);
  // LDS float i of a row holds row float f0 - 16*C + i, its cell clamped into the row (:39-43)
  auto clamped = [&](const float* src, int64_t f) {
    const int64_t c = ((f % C) + C) % C;
    int64_t y = (f - c) / C;
    if (y < 0) y = 0;
    if (y > W - 1) y = W - 1;
    return src[y * C + c];
  };
  // (rows past the grid's last one restage the last row — no exits from these loops, so that
  // `body` stays in registers: indexed behind a break it went through scratch memory, twice the
  // cache traffic)
  auto row_of = [&](int r) { return in + (row0 + r < H ? row0 + r : H - 1) * WC; };
  float4 body[kBlurRows];
#pragma unroll
  for (int r = 0; r < kBlurRows; ++r) {
    const float* src = row_of(r);
    if constexpr (WIDE) {
      body[r] = *reinterpret_cast<const float4*>(src + f0 + i0);
    } else {
      body[r] = make_float4(clamped(src, f0 + i0), clamped(src, f0 + i0 + 1),
                            clamped(src, f0 + i0 + 2), clamped(src, f0 + i0 + 3));
    }
  }
  for (int a = threadIdx.x; a < 32 * C * kBlurRows; a += kSBlock) {
    const int r = a / (32 * C), j = a % (32 * C);
    const int i = j < 16 * C ? j : kSeg + j;
    seg[r][i] = clamped(row_of(r), f0 - 16 * C + i);
  }
#pragma unroll
  for (int r = 0; r < kBlurRows; ++r) *reinterpret_cast<float4*>(&seg[r][16 * C + i0]) = body[r];
  __syncthreads();
  // outputs f0 + 4 t .. + 3 of thread t read LDS floats 4 t .. 4 t + 3 + 32 C
  constexpr int kWin = 4 + 32 * C;
  if (!WIDE && f0 + i0 >= WC) return;
  for (int r = 0; r < kBlurRows; ++r) {
    if (row0 + r >= H) break;
    // Whole 16-byte reads, kept whole: left to itself the compiler takes the window apart into
    // ds_read2_b32 pairs — at the 16-byte stride between lanes an 8-way bank conflict each.
    typedef float f4 __attribute__((ext_vector_type(4)));
    float w[kWin];
    f4 t4[kWin / 4];
#pragma unroll
    for (int q = 0; q < kWin / 4; ++q) t4[q] = *reinterpret_cast<const f4*>(&seg[r][i0 + 4 * q]);
#pragma unroll
    for (int q = 0; q < kWin / 4; ++q) {
      asm("" : "+v"(t4[q]));  // (all reads issued before the first is looked at)
      w[4 * q] = t4[q].x, w[4 * q + 1] = t4[q].y, w[4 * q + 2] = t4[q].z, w[4 * q + 3] = t4[q].w;
    }
    float val[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      val[j] = 0.0f;  // :35
#pragma unroll
      for (int k = 0; k < 33; ++k) val[j] = __builtin_fmaf(w[j + k * C], bw.w[k], val[j]);  // :49-50
    }
    float* dst = out + (row0 + r) * WC + f0 + i0;  // :54
    if constexpr (WIDE) {
      *reinterpret_cast<float4*>(dst) = make_float4(val[0], val[1], val[2], val[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (f0 + i0 + j < WC) dst[j] = val[j];
    }
  }
}

template <int C>
__global__ void __launch_bounds__(kSBlock)
    k_blur_cols(float* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t W,
                int wide_rows, BlurWeights bw) {
  constexpr int kSeg = 4 * kSBlock;    // output floats per row and work-group
  constexpr int kLds = kSeg + 32 * C;  // LDS floats per row: the segment and its two aprons
  __shared__ __attribute__((aligned(16))) float seg[kBlurRows][kLds];
  // consecutive work-groups take consecutive segments of the same rows: what is in flight at a
  // time is whole rows, not one 4 KiB column of many rows
  const int64_t row0 = static_cast<int64_t>(blockIdx.y) * kBlurRows;
  const int64_t f0 = static_cast<int64_t>(blockIdx.x) * kSeg;  // first output float of the segment
  if (wide_rows && f0 + kSeg <= W * C) blur_cols_group<C, true>(seg, out, in, H, W, row0, f0, bw);
  else blur_cols_group<C, false>(seg, out, in, H, W, row0, f0, bw);
}


// lerp5 gradient along one axis: the build's definition of silt's lerp5_t::grad
// (un-vendored): 4th-order central difference when all five
// samples exist and are finite, else 2nd-order central, else one-sided, else 0.
SOIL_HD float lerp5_axis(const float* in, int64_t n, int64_t stride, int64_t i, int64_t len) {
  const float nan = __builtin_nanf("");
  const float f0 = in[n];
  const float fm2 = (i - 2 >= 0) ? in[n - 2 * stride] : nan;
  const float fm1 = (i - 1 >= 0) ? in[n - stride] : nan;
  const float fp1 = (i + 1 < len) ? in[n + stride] : nan;
  const float fp2 = (i + 2 < len) ? in[n + 2 * stride] : nan;
  auto fin = [](float v) { return (v - v) == 0.0f; };  // finite: not NaN, not +-inf
  if (fin(fm2) && fin(fm1) && fin(fp1) && fin(fp2))
    return ((fm2 - 8.0f * fm1) + (8.0f * fp1 - fp2)) / 12.0f;
  if (fin(fm1) && fin(fp1)) return 0.5f * (fp1 - fm1);
  if (fin(fp1) && fin(f0)) return fp1 - f0;
  if (fin(fm1) && fin(f0)) return f0 - fm1;
  return 0.0f;
}

// soil::op::normal, normal.hpp:29-35, for one cell
SOIL_HD void normal_cell(float* out, const float* in, int64_t x, int64_t y, int64_t H, int64_t W,
                         Scale3 s) {
  const int64_t n = x * W + y;
  const float gx = lerp5_axis(in, n, W, x, H) * s.z / s.x;  // :31-32
  const float gy = lerp5_axis(in, n, 1, y, W) * s.z / s.y;
  const float vx = -gx, vy = -gy, vz = 1.0f;  // :33
  const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
  out[3 * n] = vx * inv;
  out[3 * n + 1] = vy * inv;
  out[3 * n + 2] = vz * inv;
}

__global__ void __launch_bounds__(kSBlock)
    k_normal(float* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t W,
             Scale3 s) {
  const int64_t y = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x;
  if (y >= W) return;
  SOIL_ROW_LOOP(x, H) normal_cell(out, in, x, y, H, W, s);
}

// The same with the window shape (window.hpp): a thread owns four consecutive cells of a row and
// walks a band of rows with rows x - 2 .. x + 2 of its four columns in registers; the two columns
// either side of its cells in row x come from the neighbouring lanes.  One 16-byte load per row
// instead of nine gathers per cell, and the wave's 3 KiB of normals leave through LDS as three
// store instructions of 1 KiB of consecutive bytes each (stored as they are, a lane's 48 bytes
// would make every instruction cover a third of each sector).
struct RowN {
  float4 c;          // the thread's four cells
  float l2, l1, r1, r2;  // columns y0 - 2, y0 - 1, y0 + 4, y0 + 5 (NaN outside the grid)
};
__device__ __forceinline__ RowN load_row_n(const float* __restrict__ in, int64_t x, int64_t H,
                                           int64_t W, int64_t y0, bool halo) {
  const float nan = __builtin_nanf("");
  const int lane = static_cast<int>(threadIdx.x & 63u);
  const bool row_ok = x >= 0 && x < H;
  RowN r;
  r.c = make_float4(nan, nan, nan, nan);
  const float* row = in + x * W;
  if (row_ok) r.c = *reinterpret_cast<const float4*>(row + y0);
  r.l2 = r.l1 = r.r1 = r.r2 = nan;
  if (halo) {  // uniform: every lane of the wave shuffles
    r.l2 = __shfl_up(r.c.z, 1, 64);
    r.l1 = __shfl_up(r.c.w, 1, 64);
    r.r1 = __shfl_down(r.c.x, 1, 64);
    r.r2 = __shfl_down(r.c.y, 1, 64);
    if (lane == 0) {
      r.l2 = (row_ok && y0 >= 2) ? row[y0 - 2] : nan;
      r.l1 = (row_ok && y0 >= 1) ? row[y0 - 1] : nan;
    }
    if (lane == 63) {
      r.r1 = (row_ok && y0 + 4 < W) ? row[y0 + 4] : nan;
      r.r2 = (row_ok && y0 + 5 < W) ? row[y0 + 5] : nan;
    }
  }
  return r;
}
// lerp5_axis on five samples in registers (NaN: outside the grid)
__device__ __forceinline__ float lerp5_regs(float fm2, float fm1, float f0, float fp1, float fp2) {
  auto fin = [](float v) { return (v - v) == 0.0f; };
  if (fin(fm2) && fin(fm1) && fin(fp1) && fin(fp2))
    return ((fm2 - 8.0f * fm1) + (8.0f * fp1 - fp2)) / 12.0f;
  if (fin(fm1) && fin(fp1)) return 0.5f * (fp1 - fm1);
  if (fin(fp1) && fin(f0)) return fp1 - f0;
  if (fin(fm1) && fin(f0)) return f0 - fm1;
  return 0.0f;
}
__device__ __forceinline__ float at4(const float4& v, int k) {
  return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
}

// (round 5: the kernel took 93 registers = 5 waves per SIMD; held to 80 = 6 waves 0.318 -> 0.289 ms at 8192^2,
// to 64 = 8 waves it spills, 0.65 ms; bands of 32 | 16 | 8 | 4 rows 0.348 | 0.324 | 0.318 | 0.319 ms)
#ifndef SOIL_NORMAL_WAVES
#define SOIL_NORMAL_WAVES 6
#endif
template <int BAND>
__global__ void __launch_bounds__(kWinBlock) __attribute__((amdgpu_waves_per_eu(SOIL_NORMAL_WAVES, 8)))
    k_normal4(float* __restrict__ out, const float* __restrict__ in, int64_t H, int64_t W, Scale3 s, bool fast) {
  __shared__ float4 s_tile[kWinBlock / 64][192];
  const WinThread t = win_thread(W);
  const Recip r12 = recip(12.0f), rx = recip(s.x), ry = recip(s.y);
  const int lane = static_cast<int>(threadIdx.x & 63u);
  float4* tile = s_tile[threadIdx.x >> 6];
  const int64_t wave_y0 = (static_cast<int64_t>(blockIdx.x) * kWinBlock + (threadIdx.x & ~63u)) * 4;
  for (int64_t band = blockIdx.y; band * BAND < H; band += gridDim.y) {
    int64_t x = band * BAND;
    const int64_t x_end = (x + BAND < H) ? x + BAND : H;
    // rows x - 2 .. x + 2; the halo columns are wanted for the row a cell's y-derivative is taken
    // in, the centre one: loaded with every row that will get there (not the two above the band)
    float4 m2 = load_row_n(in, x - 2, H, W, t.y0, false).c, m1 = load_row_n(in, x - 1, H, W, t.y0, false).c;
    RowN c0 = load_row_n(in, x, H, W, t.y0, true), p1 = load_row_n(in, x + 1, H, W, t.y0, true);
    RowN p2 = load_row_n(in, x + 2, H, W, t.y0, true);
    for (; x < x_end; ++x) {
      float o[12];
      // the row's six columns either side included: y0 - 2 .. y0 + 5
      const float rowv[8] = {c0.l2, c0.l1, c0.c.x, c0.c.y, c0.c.z, c0.c.w, c0.r1, c0.r2};
      // a thread past the end of the row sits on the last group (t.y0 = W - 4): its columns exist
      const float nan = __builtin_nanf("");
      // Round 5: the cell's five IEEE divisions (two by 12, by the two cell sizes, 1 / length) and its
      // square root were ~60 of its ~115 vector instructions.  Away from the grid's edge, with every
      // sample finite, lerp5 is its fourth-order formula and the quotients are shared-reciprocal ones
      // (soil_math.hpp: the IEEE quotient's own instruction sequence on a reciprocal refined once per
      // thread / per cell) whose results are watched; a group in doubt — a non-finite sample shows as a
      // NaN or an infinity in the results, a denormal difference, a -0 numerator, the sentinels at the
      // edge — is redone as written.  Same bits either way (the parity tests run both).
      bool redo = !fast || x < 2 || x + 2 >= H || t.y0 < 2 || t.y0 + 5 >= W;
      if (!redo) {
        QuotWatch watch;
        bool len_ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float nx = (at4(m2, k) - 8.0f * at4(m1, k)) + (8.0f * at4(p1.c, k) - at4(p2.c, k));
          const float ny = (rowv[k] - 8.0f * rowv[k + 1]) + (8.0f * rowv[k + 3] - rowv[k + 4]);
          const float dx = watch(quot(nx, r12), nx), dy = watch(quot(ny, r12), ny);
          const float ax = dx * s.z, ay = dy * s.z;
          const float gx = watch(quot(ax, rx), ax), gy = watch(quot(ay, ry), ay);  // :31-32
          const float vx = -gx, vy = -gy, vz = 1.0f;                               // :33
          const float len = sqrt_rn(vx * vx + vy * vy + vz * vz);                  // (>= 1: no scaling wanted)
          len_ok = len_ok && len <= kDenHi;
          const float inv = quot(1.0f, recip(len));
          o[3 * k] = vx * inv;
          o[3 * k + 1] = vy * inv;
          o[3 * k + 2] = vz * inv;
        }
        redo = watch.doubtful() || !len_ok;
      }
      if (redo) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t y = t.y0 + k;
        const float f0 = rowv[k + 2];
        const float dx = lerp5_regs(at4(m2, k), at4(m1, k), f0, at4(p1.c, k), at4(p2.c, k));
        // columns outside the grid read NaN: the neighbours' registers hold them for lanes 0 / 63,
        // inside the wave the shuffled values are real cells, and y - 2 >= 0, y + 2 < W decide
        const float ym2 = (y - 2 >= 0) ? rowv[k] : nan, ym1 = (y - 1 >= 0) ? rowv[k + 1] : nan;
        const float yp1 = (y + 1 < W) ? rowv[k + 3] : nan, yp2 = (y + 2 < W) ? rowv[k + 4] : nan;
        const float dy = lerp5_regs(ym2, ym1, f0, yp1, yp2);
        const float gx = dx * s.z / s.x;  // :31-32
        const float gy = dy * s.z / s.y;
        const float vx = -gx, vy = -gy, vz = 1.0f;  // :33
        const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
        o[3 * k] = vx * inv;
        o[3 * k + 1] = vy * inv;
        o[3 * k + 2] = vz * inv;
      }
      }
      {  // the wave's 64 x 48 bytes, contiguous in memory, as three instructions of 1 KiB each
        const int n = 3 * __popcll(__ballot(t.live));  // float4s of the wave that are real
        tile[3 * lane] = make_float4(o[0], o[1], o[2], o[3]);
        tile[3 * lane + 1] = make_float4(o[4], o[5], o[6], o[7]);
        tile[3 * lane + 2] = make_float4(o[8], o[9], o[10], o[11]);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own writes have landed
        __builtin_amdgcn_wave_barrier();
        const float4 a = tile[lane], b = tile[64 + lane], c = tile[128 + lane];
        float4* dst = reinterpret_cast<float4*>(out + 3 * (x * W + wave_y0));
        if (lane < n) dst[lane] = a;
        if (64 + lane < n) dst[64 + lane] = b;
        if (128 + lane < n) dst[128 + lane] = c;
        __builtin_amdgcn_wave_barrier();
      }
      m2 = m1;
      m1 = c0.c;
      c0 = p1;
      p1 = p2;
      if (x + 1 < x_end) p2 = load_row_n(in, x + 3, H, W, t.y0, true);
    }
  }
}

// soil.resize of the multiscale driver (example/erosion_gpu_multiscale.py:104-141).
// The reference snapshot holds no definition of it; defined here as
// bilinear resampling at corner-aligned positions: new cell (i, j) samples the old
// grid at (i*(Ho-1)/(Hn-1), j*(Wo-1)/(Wn-1)), with the weights written as in the
// reference's sampler, (1 - t)*a + t*b (sample.hpp:48-60) — exact at t = 0 and 1, so
// equal resolutions give the identity and the corners are kept.  (That sampler itself
// stops interpolating in the last cell of each axis, sample.hpp:172-173; a resize
// must not.)
__device__ __forceinline__ float resize_pos(int64_t i, int64_t n_new, int64_t n_old) {
  if (n_new <= 1) return 0.0f;
  const float step = static_cast<float>(n_old - 1) / static_cast<float>(n_new - 1);
  return fminf(static_cast<float>(i) * step, static_cast<float>(n_old - 1));
}

template <int D>
__global__ void __launch_bounds__(kSBlock)
    k_resize(float* __restrict__ dst, const float* __restrict__ src, int64_t Hn, int64_t Wn,
             int64_t Ho, int64_t Wo) {
  const int64_t y = static_cast<int64_t>(blockIdx.x) * kSBlock + threadIdx.x;
  if (y >= Wn) return;
  const float py = resize_pos(y, Wn, Wo);
  int64_t iy = static_cast<int64_t>(py);
  if (iy > Wo - 2) iy = Wo - 2;
  if (iy < 0) iy = 0;  // Wo == 1
  const int64_t jy = (Wo > 1) ? iy + 1 : iy;
  const float wy = py - static_cast<float>(iy);
  const float ay = 1.0f + -1.0f * wy, by = 0.0f + 1.0f * wy;
  SOIL_ROW_LOOP(x, Hn) {
    const float px = resize_pos(x, Hn, Ho);
    int64_t ix = static_cast<int64_t>(px);
    if (ix > Ho - 2) ix = Ho - 2;
    if (ix < 0) ix = 0;
    const int64_t jx = (Ho > 1) ? ix + 1 : ix;
    const float wx = px - static_cast<float>(ix);
    const float ax = 1.0f + -1.0f * wx, bx = 0.0f + 1.0f * wx;
    const int64_t i00 = ix * Wo + iy, i01 = ix * Wo + jy, i10 = jx * Wo + iy, i11 = jx * Wo + jy;
    const int64_t n = x * Wn + y;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const float l0 = ay * src[D * i00 + c] + by * src[D * i01 + c];
      const float l1 = ay * src[D * i10 + c] + by * src[D * i11 + c];
      dst[D * n + c] = ax * l0 + bx * l1;
    }
  }
}

}  // namespace soil

using namespace soil;

// (the launch macro names a kernel by a template over the walk alone)
template <class Walk> constexpr auto kGradientFast = k_gradient4<true, Walk>;
template <class Walk> constexpr auto kGradientWritten = k_gradient4<false, Walk>;
template <class Walk> constexpr auto kNegslopeFast = k_negslope4<true, Walk>;
template <class Walk> constexpr auto kNegslopeWritten = k_negslope4<false, Walk>;
template <class Walk> constexpr auto kLaplacian1 = k_laplacian4<Walk>;

extern "C" {

int soil_resize(float* dst, const float* src, int64_t Hn, int64_t Wn, int64_t Ho, int64_t Wo, int D,
                void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(dst && src, "resize: null argument");
  SOIL_REQUIRE(Hn > 0 && Wn > 0 && Ho > 0 && Wo > 0, "resize: empty grid");
  SOIL_REQUIRE(D >= 1 && D <= 3, "resize: 1, 2 or 3 channels");
  const dim3 grid = grid_rows(Hn, Wn, kSBlock);
  hipStream_t st = as_stream(stream);
  if (D == 1) k_resize<1><<<grid, kSBlock, 0, st>>>(dst, src, Hn, Wn, Ho, Wo);
  else if (D == 2) k_resize<2><<<grid, kSBlock, 0, st>>>(dst, src, Hn, Wn, Ho, Wo);
  else k_resize<3><<<grid, kSBlock, 0, st>>>(dst, src, Hn, Wn, Ho, Wo);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_gradient(float* out, const float* in, int64_t H, int64_t W, const float scale[2],
                  void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && in && scale, "gradient: null argument");
  SOIL_REQUIRE(H > 0 && W > 0, "gradient: empty grid");
  const bool fast = plain_scale(scale[0]) && plain_scale(scale[1]);
  if (W % 4 == 0 && W >= 4)
    SOIL_WIN_LAUNCH(5, 5, fast, kGradientFast, kGradientWritten, false, H, W, as_stream(stream),
                    reinterpret_cast<float2*>(out), in, H, W, Scale2{scale[0], scale[1]});
  else
    k_gradient<<<grid_rows(H, W, kSBlock), kSBlock, 0, as_stream(stream)>>>(
        reinterpret_cast<float2*>(out), in, H, W, Scale2{scale[0], scale[1]});
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_negslope(float* out, const float* in, int64_t H, int64_t W, const float scale[2],
                  void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && in && scale, "negslope: null argument");
  SOIL_REQUIRE(H > 0 && W > 0, "negslope: empty grid");
  const bool fast = plain_scale(scale[0]) && plain_scale(scale[1]);
  if (W % 4 == 0 && W >= 4)
    SOIL_WIN_LAUNCH(8, 4, fast, kNegslopeFast, kNegslopeWritten, true, H, W, as_stream(stream), out, in, H, W,
                    Scale2{scale[0], scale[1]});
  else
    k_negslope<<<grid_rows(H, W, kSBlock), kSBlock, 0, as_stream(stream)>>>(
        out, in, H, W, Scale2{scale[0], scale[1]});
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_laplacian(float* out, const float* in, int64_t H, int64_t W, int D, const float scale[2],
                   void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && in && scale, "laplacian: null argument");
  SOIL_REQUIRE(H > 0 && W > 0, "laplacian: empty grid");
  const Scale2 s{scale[0], scale[1]};
  if (D == 1 && W % 4 == 0 && W >= 4)  // grad.cu:196-198
    SOIL_WIN_LAUNCH(0, 4, true, kLaplacian1, kLaplacian1, false, H, W, as_stream(stream), out, in, H, W,
                    1.0f / s.x / s.x, 1.0f / s.y / s.y);  // (IEEE fp32 on the host: the same bits as on the device)
  else if (D == 1)
    k_laplacian<1><<<grid_rows(H, W, kSBlock), kSBlock, 0, as_stream(stream)>>>(out, in, H, W, s);
  else if (D == 2 && W % 4 == 0 && W >= 4)  // grad.cu:200-202
  {
    // (round 5, 8192^2: bands of 32 | 16 | 8 | 4 rows 0.256 | 0.220 | 0.241 | 0.235 ms)
    static const int band = [] { const char* e = std::getenv("SOIL_LAP2_BAND"); return e ? std::atoi(e) : 16; }();
    auto grid = [&](int b) {
      const int64_t bands = (H + b - 1) / b;
      return dim3(static_cast<unsigned>((W / 4 + kWinBlock - 1) / kWinBlock), static_cast<unsigned>(bands < 65535 ? bands : 65535));
    };
    if (band == 4) k_laplacian4x2<4><<<grid(4), kWinBlock, 0, as_stream(stream)>>>(out, in, H, W, s);
    else if (band == 8) k_laplacian4x2<8><<<grid(8), kWinBlock, 0, as_stream(stream)>>>(out, in, H, W, s);
    else if (band == 16) k_laplacian4x2<16><<<grid(16), kWinBlock, 0, as_stream(stream)>>>(out, in, H, W, s);
    else k_laplacian4x2<32><<<grid(32), kWinBlock, 0, as_stream(stream)>>>(out, in, H, W, s);
  }
  else if (D == 2)
    k_laplacian<2><<<grid_rows(H, 2 * W, kSBlock), kSBlock, 0, as_stream(stream)>>>(out, in, H, W, s);
  else
    return fail(SOIL_ERR_INVALID_ARGUMENT, "laplacian: channel count must be 1 or 2");
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_gaussian_blur(float* tensor, float* scratch, int64_t H, int64_t W, int C, float sigma,
                       void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(tensor && scratch, "gaussian_blur: null tensor");
  SOIL_REQUIRE(H > 0 && W > 0, "gaussian_blur: empty grid");
  SOIL_REQUIRE(C == 1 || C == 2, "gaussian_blur: channel count must be 1 or 2");
  BlurWeights bw;
  for (int k = -16; k <= 16; ++k) {
    const float Z = sqrtf(2.0f * 3.14159265f) * sigma;                                // filter.cu:47
    bw.w[k + 16] = expf_(-0.5f * (static_cast<float>(k) / sigma) * (static_cast<float>(k) / sigma)) / Z;  // :48
  }
  const int64_t WC = W * C;
  hipStream_t st = as_stream(stream);
  const dim3 grid_rows(blocks_for(WC, kSBlock), static_cast<unsigned>((H + kBlurBand - 1) / kBlurBand));
  SOIL_REQUIRE(grid_rows.y <= 65535u && WC <= 65535 * 4 * static_cast<int64_t>(kSBlock) &&
                   (H + kBlurRows - 1) / kBlurRows <= 65535,
               "gaussian_blur: grid too large for one launch");
  k_blur_rows<<<grid_rows, kSBlock, 0, st>>>(scratch, tensor, H, WC, bw);  // :81 / :86
  const dim3 grid_cols(blocks_for(WC, 4 * kSBlock), static_cast<unsigned>((H + kBlurRows - 1) / kBlurRows));
  const int wide_rows = (WC & 3) == 0 && ((reinterpret_cast<uintptr_t>(tensor) | reinterpret_cast<uintptr_t>(scratch)) & 15) == 0;
  if (C == 1) k_blur_cols<1><<<grid_cols, kSBlock, 0, st>>>(tensor, scratch, H, W, wide_rows, bw);  // :82 / :87
  else k_blur_cols<2><<<grid_cols, kSBlock, 0, st>>>(tensor, scratch, H, W, wide_rows, bw);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_normal(float* out, const float* in, int64_t H, int64_t W, const float scale[3],
                void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && in && scale, "normal: null argument");
  SOIL_REQUIRE(H > 0 && W > 0, "normal: empty grid");
  if (W % 4 == 0 && W >= 4)
  {
    static const int band = [] { const char* e = std::getenv("SOIL_NORMAL_BAND"); return e ? std::atoi(e) : 8; }();
    const bool fast = plain_scale(scale[0]) && plain_scale(scale[1]);
    const Scale3 s3{scale[0], scale[1], scale[2]};
    auto grid = [&](int b) {
      const int64_t bands = (H + b - 1) / b;
      return dim3(static_cast<unsigned>((W / 4 + kWinBlock - 1) / kWinBlock), static_cast<unsigned>(bands < 65535 ? bands : 65535));
    };
    if (band == 4) k_normal4<4><<<grid(4), kWinBlock, 0, as_stream(stream)>>>(out, in, H, W, s3, fast);
    else if (band == 8) k_normal4<8><<<grid(8), kWinBlock, 0, as_stream(stream)>>>(out, in, H, W, s3, fast);
    else if (band == 16) k_normal4<16><<<grid(16), kWinBlock, 0, as_stream(stream)>>>(out, in, H, W, s3, fast);
    else k_normal4<32><<<grid(32), kWinBlock, 0, as_stream(stream)>>>(out, in, H, W, s3, fast);
  }
  else
    k_normal<<<grid_rows(H, W, kSBlock), kSBlock, 0, as_stream(stream)>>>(
        out, in, H, W, Scale3{scale[0], scale[1], scale[2]});
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_normal_host(float* out, const float* in, int64_t H, int64_t W, const float scale[3]) {
  SOIL_REQUIRE(out && in && scale, "normal_host: null argument");
  SOIL_REQUIRE(H > 0 && W > 0, "normal_host: empty grid");
  const Scale3 s{scale[0], scale[1], scale[2]};
  for (int64_t x = 0; x < H; ++x)
    for (int64_t y = 0; y < W; ++y) normal_cell(out, in, x, y, H, W, s);  // normal.hpp:29-35
  return SOIL_OK;
}

}  // extern "C"
