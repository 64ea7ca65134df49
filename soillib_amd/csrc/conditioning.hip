// conditioning.hip — depression filling of a DEM before flow routing.
//
// The reference has no such code: its example conditions DEMs with the
// third-party pysheds (example/dem_condition.py:35-41, grid.fill_pits /
// fill_depressions), yet BASELINE config 3 asks for a pit-filled
// 4096^2 DEM.  Build-defined, parity unpinned by the reference; the CPU oracle
// is Barnes' priority-flood (oracle/soil_oracle.c: orc_fill_depressions).
//
// Definition: w(c) = min over all paths from c to an outlet of the highest z on
// the path, where an outlet is a step off the grid or onto a NaN (NoData) cell.
// w is the least surface >= z without depressions; it is the greatest fixed
// point below the start of
//        w(c) <- max(z(c), min over the neighbours n of w(n))
// started from w = z on cells next to an outlet and +inf elsewhere.  Only max
// and min are involved, so the fixed point is reached exactly in fp32 whatever
// the update order: the kernel relaxes 64x64 tiles in LDS until they stop
// changing (chaotic Gauss-Seidel inside a tile, Jacobi across tiles per launch)
// and the host repeats launches until no tile changed.
#include <utility>

#include "common.hpp"

namespace soil {

constexpr int kFT = 64;             // tile edge
constexpr int kFH = kFT + 2;        // with its one-cell apron
constexpr int kFBlock = 256;
constexpr int kFPer = kFT * kFT / kFBlock;

template <int K>
__global__ void __launch_bounds__(kFBlock)
    k_fill_relax(float* __restrict__ w, const float* __restrict__ z, int64_t H, int64_t W,
                 int tiles_w, int tiles_h, int inner_max, int* __restrict__ changed,
                 const unsigned char* __restrict__ dirty_prev,
                 unsigned char* __restrict__ dirty_next) {
  __shared__ float sw[kFH * kFH];
  __shared__ int s_flag, s_any;
  const int tid = threadIdx.x;
  {  // a tile can only move if it or one of its 8 neighbours moved in the previous launch
    const int tx = blockIdx.x / tiles_w, ty = blockIdx.x % tiles_w;
    bool live = false;
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy) {
        const int nx = tx + dx, ny = ty + dy;
        if (nx >= 0 && ny >= 0 && nx < tiles_h && ny < tiles_w)
          live = live || dirty_prev[nx * tiles_w + ny] != 0;
      }
    if (!live) return;
  }
  const int64_t row0 = static_cast<int64_t>(blockIdx.x / tiles_w) * kFT;
  const int64_t col0 = static_cast<int64_t>(blockIdx.x % tiles_w) * kFT;
  const float ninf = -__builtin_inff();
  // tile + apron; off-grid and NaN cells are outlets: -inf
  for (int i = tid; i < kFH * kFH; i += kFBlock) {
    const int64_t x = row0 + i / kFH - 1, y = col0 + i % kFH - 1;
    float v = ninf;
    if (x >= 0 && y >= 0 && x < H && y < W) {
      v = w[x * W + y];
      if (v != v) v = ninf;
    }
    sw[i] = v;
  }
  float zc[kFPer];
  bool in[kFPer];
#pragma unroll
  for (int j = 0; j < kFPer; ++j) {
    const int c = tid + j * kFBlock;
    const int64_t x = row0 + c / kFT, y = col0 + c % kFT;
    in[j] = x < H && y < W;
    zc[j] = in[j] ? z[x * W + y] : 0.0f;
    if (zc[j] != zc[j]) in[j] = false;  // NaN cells stay NaN and act as outlets
  }
  if (tid == 0) s_any = 0;
  __syncthreads();
  for (int it = 0; it < inner_max; ++it) {
    if (tid == 0) s_flag = 0;
    __syncthreads();
    bool moved = false;
#pragma unroll
    for (int j = 0; j < kFPer; ++j) {
      if (!in[j]) continue;
      const int c = tid + j * kFBlock;
      const int p = (c / kFT + 1) * kFH + (c % kFT + 1);
      float m = fminf(fminf(sw[p - kFH], sw[p + kFH]), fminf(sw[p - 1], sw[p + 1]));
      if (K == 8)
        m = fminf(m, fminf(fminf(sw[p - kFH - 1], sw[p - kFH + 1]),
                           fminf(sw[p + kFH - 1], sw[p + kFH + 1])));
      const float v = fmaxf(zc[j], m);
      if (v < sw[p]) {
        sw[p] = v;
        moved = true;
      }
    }
    if (moved) s_flag = 1;
    __syncthreads();
    if (s_flag == 0) break;
    if (tid == 0) s_any = 1;
    __syncthreads();
  }
  __syncthreads();
  if (s_any) {
#pragma unroll
    for (int j = 0; j < kFPer; ++j) {
      if (!in[j]) continue;
      const int c = tid + j * kFBlock;
      const int64_t x = row0 + c / kFT, y = col0 + c % kFT;
      w[x * W + y] = sw[(c / kFT + 1) * kFH + (c % kFT + 1)];
    }
    if (tid == 0) {
      *changed = 1;
      dirty_next[blockIdx.x] = 1;
    }
  }
}

// w = z where a neighbour is an outlet (or z is NaN), +inf elsewhere
template <int K>
__global__ void __launch_bounds__(kFBlock)
    k_fill_init(float* __restrict__ w, const float* __restrict__ z, int64_t H, int64_t W) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kFBlock + threadIdx.x;
  if (n >= H * W) return;
  const int64_t x = n / W, y = n % W;
  const float zv = z[n];
  bool outlet = zv != zv;
  const int dx[8] = {-1, 0, 0, 1, -1, -1, 1, 1}, dy[8] = {0, -1, 1, 0, -1, 1, -1, 1};
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int64_t nx = x + dx[k], ny = y + dy[k];
    if (nx < 0 || ny < 0 || nx >= H || ny >= W) {
      outlet = true;
    } else {
      const float nv = z[nx * W + ny];
      if (nv != nv) outlet = true;
    }
  }
  w[n] = outlet ? zv : __builtin_inff();
}

template <int K>
static int fill_impl(float* out, const float* height, int64_t H, int64_t W, hipStream_t st) {
  const int tiles_w = static_cast<int>((W + kFT - 1) / kFT);
  const int tiles_h = static_cast<int>((H + kFT - 1) / kFT);
  const size_t ntiles = static_cast<size_t>(tiles_w) * tiles_h, b_dirty = (ntiles + 255) & ~size_t{255};
  void* base = nullptr;
  if (int rc = workspace_get(4, 256 + 2 * b_dirty, &base); rc != SOIL_OK) return rc;
  // "some tile moved in this launch": a pinned, device-mapped word the tiles write straight
  // into (a device-to-host copy is a 25-50 us blit kernel on this stack, per launch)
  static thread_local int *t_flag = nullptr, *t_flag_dev = nullptr;
  if (!t_flag) {
    SOIL_HIP(hipHostMalloc(reinterpret_cast<void**>(&t_flag), sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    SOIL_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&t_flag_dev), t_flag, 0));
  }
  int* changed = t_flag_dev;
  unsigned char* dirty_prev = static_cast<unsigned char*>(base) + 256;
  unsigned char* dirty_next = dirty_prev + b_dirty;
  SOIL_HIP(hipMemsetAsync(dirty_prev, 1, ntiles, st));  // first launch: every tile
  k_fill_init<K><<<blocks_for(H * W, kFBlock), kFBlock, 0, st>>>(out, height, H, W);
  SOIL_LAUNCH_CHECK();
  // a launch moves information at least one tile further; H*W launches is a bound
  // no terrain reaches, typical counts are a few times the number of tiles per side
  const int64_t max_launches = 4 * (static_cast<int64_t>(tiles_w) + tiles_h) * kFT + 16;
  for (int64_t launch = 0; launch < max_launches; ++launch) {
    *t_flag = 0;  // the stream is idle here: the previous launch was waited for
    SOIL_HIP(hipMemsetAsync(dirty_next, 0, ntiles, st));
    k_fill_relax<K><<<static_cast<unsigned>(ntiles), kFBlock, 0, st>>>(
        out, height, H, W, tiles_w, tiles_h, 4 * kFT, changed, dirty_prev, dirty_next);
    std::swap(dirty_prev, dirty_next);
    SOIL_LAUNCH_CHECK();
    SOIL_HIP(hipStreamSynchronize(st));
    if (!__atomic_load_n(t_flag, __ATOMIC_ACQUIRE)) return SOIL_OK;
  }
  return fail(SOIL_ERR_HIP, "fill_depressions: did not converge");
}

}  // namespace soil

using namespace soil;

extern "C" {

int soil_fill_depressions(float* out, const float* height, int64_t H, int64_t W, int edge,
                          void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && height && out != height, "fill_depressions: needs distinct in and out tensors");
  SOIL_REQUIRE(H > 0 && W > 0, "fill_depressions: empty grid");
  switch (edge) {
    case SOIL_D4: return fill_impl<4>(out, height, H, W, as_stream(stream));
    case SOIL_D8: return fill_impl<8>(out, height, H, W, as_stream(stream));
    default: return fail(SOIL_ERR_INVALID_ARGUMENT, "invalid edge enumerator");
  }
}

}  // extern "C"
