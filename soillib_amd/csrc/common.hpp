// common.hpp — shared host/device plumbing of libsoil_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <string>

#include "../../include/soil_hip.h"
#include "soil_math.hpp"

namespace soil {

// ---- error channel ---------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
// Makes sure a HIP device is usable; SOIL_ERR_NO_DEVICE otherwise.  There is no
// CPU fallback anywhere in this library.
int require_device();

#define SOIL_HIP(expr)                                                     \
  do {                                                                     \
    hipError_t soil_e_ = (expr);                                           \
    if (soil_e_ != hipSuccess) return ::soil::hip_fail(soil_e_, #expr, __FILE__, __LINE__); \
  } while (0)

#define SOIL_REQUIRE(cond, msg)                                            \
  do {                                                                     \
    if (!(cond)) return ::soil::fail(SOIL_ERR_INVALID_ARGUMENT, msg);      \
  } while (0)

#define SOIL_REQUIRE_IO(cond, msg)                            \
  do {                                                        \
    if (!(cond)) return ::soil::fail(SOIL_ERR_IO, msg);       \
  } while (0)

#define SOIL_DEVICE()                                  \
  do {                                                 \
    int soil_rc_ = ::soil::require_device();           \
    if (soil_rc_ != SOIL_OK) return soil_rc_;          \
  } while (0)

// Reports launch-configuration errors of the kernel launched just before.
#define SOIL_LAUNCH_CHECK() SOIL_HIP(hipGetLastError())

// Per-device scratch, grown on demand and reused across calls (the reference
// cudaMallocs its scratch on every call, graph.cu:539-550, 182-183).  Slot 0:
// accumulate; slot 1: particle staging.  Calls that share a slot must be
// stream-ordered with respect to each other.
int workspace_get(int slot, size_t bytes, void** out);
int workspace_release_all();

// Device counter of particle steps (iterations that pass the loop head and its
// slab check, i.e. what the oracle counts), accumulated by every particle launch
// on this device; read and reset through soil_particle_steps().
int step_counter(unsigned long long** out);

// Launch shape of the per-cell kernels: threads along the contiguous axis, and a
// work-group walks a band of kRowBand consecutive rows (SOIL_ROW_LOOP).  A 64-bit
// n / W, n % W per cell costs more than most of these kernels' arithmetic, and with
// consecutive rows in one work-group the rows x-1, x of a 3x3 stencil come out of
// that CU's L1 instead of being fetched again by a work-group on another XCD.
constexpr int kRowBand = 16;
// grid.y is capped at 65535: a work-group of a taller grid (> 1 M rows) walks several bands
inline dim3 grid_rows(int64_t H, int64_t W, int threads) {
  const int64_t bands = (H + kRowBand - 1) / kRowBand;
  return dim3(static_cast<unsigned>((W + threads - 1) / threads),
              static_cast<unsigned>(bands < 65535 ? bands : 65535));
}
#define SOIL_ROW_LOOP(x, H)                                                          \
  for (int64_t x##_band = blockIdx.y; x##_band * ::soil::kRowBand < (H);             \
       x##_band += gridDim.y)                                                        \
    for (int64_t x = x##_band * ::soil::kRowBand,                                    \
                 x##_end = (x + ::soil::kRowBand < (H)) ? x + ::soil::kRowBand : (H); \
         x < x##_end; ++x)

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }
inline unsigned blocks_for(int64_t n, int threads) {
  return static_cast<unsigned>((n + threads - 1) / threads);
}

// ---- kernel-side PODs ------------------------------------------------------

struct Dom {  // soil_domain by value
  int64_t H, W, x0, rows, r0, r1;
};
inline Dom full_domain(int64_t H, int64_t W) { return Dom{H, W, 0, H, 0, H}; }
inline Dom to_dom(const soil_domain* d) { return Dom{d->H, d->W, d->x0, d->rows, d->r0, d->r1}; }
int check_domain(const Dom& d);

struct Scale3 {
  float x, y, z;
};
struct Scale2 {
  float x, y;
};

// param_t travels to the kernels by value, like in the reference.
using Param = soil_param;

// ---- device helpers shared by the erosion kernels --------------------------

// float -> cell coordinate with the semantics of the reference's device code
// (CUDA cvt.rzi: truncate toward zero, NaN -> 0).  The NaN case is live: a
// particle spawned on a pit cell with zero velocity has speed 0/sqrt(0) = NaN
// (erosion.cu:77-79), is never "out of bounds", and keeps depositing into cell
// (0,0) until maxage — restated as is.
__device__ __forceinline__ int64_t cell_of(float f) {
  return (f != f) ? 0 : static_cast<int64_t>(f);
}

__device__ __forceinline__ float length2(float x, float y) {  // erosion_map.cu:49-53
  return sqrtf(x * x + y * y);
}

// erosion_map.cu:56-78 (and its duplicate path.cu:27-49).  IEEE division by
// zero and fmaxf/fminf NaN handling are load-bearing here.
//
// The reference takes fmax of the times to both faces of the cell, (x_neg - px) / dx
// and (x_pos - px) / dx.  x_neg - px <= 0 <= x_pos - px for every finite px, and IEEE
// division is monotonic and sign-symmetric, so for dx > 0 the maximum IS the second
// quotient and for dx < 0 the first, bit for bit (signed zeros included; a NaN px or
// dx gives NaN either way): one division per axis instead of two.  A zero dx divides
// to infinities whose maximum depends on both numerators: that case keeps both.
__device__ __forceinline__ float stepsize_both(float neg, float pos, float d) {
  return fmaxf(neg / d, pos / d);
}
__device__ __forceinline__ float stepsize(float px, float py, float dx, float dy) {
  const float tmax = kSqrt2;
  const float x_neg = floorf(px);
  const float y_neg = floorf(py);
  const float x_pos = 1.0f + x_neg;
  const float y_pos = 1.0f + y_neg;
  float tx, ty;
  if (dx * dy == 0.0f) {  // a zero (or underflowing) direction component: as written
    tx = stepsize_both(x_neg - px, x_pos - px, dx);
    ty = stepsize_both(y_neg - py, y_pos - py, dy);
  } else {
    tx = ((dx > 0.0f ? x_pos : x_neg) - px) / dx;
    ty = ((dy > 0.0f ? y_pos : y_neg) - py) / dy;
  }
  tx = fminf(tx, tmax);
  ty = fminf(ty, tmax);
  return 0.5f * (tx + ty);
}

// Downhill-clamped one-sided slopes from the five heights of a cell's
// neighbourhood, erosion_map.cu:131-157.  A NaN height marks a neighbour
// outside the GLOBAL grid (the reference's sentinel, :122-125).
__device__ __forceinline__ float2 glocal_from_heights(float h, float hn0, float hp0, float h0n,
                                                      float h0p, Scale3 s, float exitSlope) {
  float gxn = (h - hn0) * s.z / s.x;
  if (gxn != gxn) gxn = exitSlope;
  else gxn = fmaxf(gxn, 0.0f);
  float gyn = (h - h0n) * s.z / s.y;
  if (gyn != gyn) gyn = exitSlope;
  else gyn = fmaxf(gyn, 0.0f);
  float gxp = (hp0 - h) * s.z / s.x;
  if (gxp != gxp) gxp = -exitSlope;
  else gxp = fminf(gxp, 0.0f);
  float gyp = (h0p - h) * s.z / s.y;
  if (gyp != gyp) gyp = -exitSlope;
  else gyp = fminf(gyp, 0.0f);

  float gx = 0.0f;
  if (fabsf(gxn) > fabsf(gx)) gx = gxn;
  if (fabsf(gxp) > fabsf(gx)) gx = gxp;
  float gy = 0.0f;
  if (fabsf(gyn) > fabsf(gy)) gy = gyn;
  if (fabsf(gyp) > fabsf(gy)) gy = gyp;
  return make_float2(gx, gy);
}

// __glocal, erosion_map.cu:107-159, for global cell (gx, y) of a slab-local
// (rows, W, 2) layer plane.
__device__ __forceinline__ float2 glocal(const float2* __restrict__ layers, const Dom& d, Scale3 s,
                                         int64_t gx, int64_t y, float exitSlope) {
  const int64_t i = (gx - d.x0) * d.W + y;
  const float2 c = layers[i];
  const float h = c.x + c.y;
  const float nan = __builtin_nanf("");
  float hn0 = nan, hp0 = nan, h0n = nan, h0p = nan;
  if (gx - 1 >= 0) {
    const float2 v = layers[i - d.W];
    hn0 = v.x + v.y;
  }
  if (gx + 1 < d.H) {
    const float2 v = layers[i + d.W];
    hp0 = v.x + v.y;
  }
  if (y - 1 >= 0) {
    const float2 v = layers[i - 1];
    h0n = v.x + v.y;
  }
  if (y + 1 < d.W) {
    const float2 v = layers[i + 1];
    h0p = v.x + v.y;
  }
  return glocal_from_heights(h, hn0, hp0, h0n, h0p, s, exitSlope);
}

}  // namespace soil
