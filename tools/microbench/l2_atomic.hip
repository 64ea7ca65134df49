// Microbenchmark: global_atomic_add_f32 into a per-work-group 64x64-cell window of 4 planes
// (the access pattern the tiled particle kernel would have if it deposited to L2 instead of LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int NPL>
__global__ void __launch_bounds__(512) k(float* planes, int64_t plane_stride, int W, int iters, int tiles_w) {
  const int tile = blockIdx.x;
  const int row0 = (tile / tiles_w) * 64, col0 = (tile % tiles_w) * 64;
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  float v = 1.0f + threadIdx.x * 1e-3f;
  // a walking cell like a particle: +-1 per step
  int r = (s >> 3) & 63, c = (s >> 11) & 63;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    r = (r + ((s >> 9) & 1)) & 63;
    c = (c + ((s >> 17) & 1)) & 63;
    const int64_t l = static_cast<int64_t>(row0 + r) * W + col0 + c;
#pragma unroll
    for (int p = 0; p < NPL; ++p) atomicAdd(&planes[p * plane_stride + l], v);
    v = v * 1.0000001f;
  }
}

int main() {
  const int W = 8192, H = 8192, tiles_w = W / 64;
  const int64_t stride = static_cast<int64_t>(W) * H;
  float* planes; hipMalloc(&planes, sizeof(float) * stride * 4); hipMemset(planes, 0, sizeof(float) * stride * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int blocks : {256, 2048, 16384}) {
    const int iters = blocks == 16384 ? 32 : 512;
    k<4><<<blocks, 512>>>(planes, stride, W, 4, tiles_w);
    hipEventRecord(a);
    k<4><<<blocks, 512>>>(planes, stride, W, iters, tiles_w);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = 4.0 * blocks * 512.0 * iters;
    printf("L2 float atomics x4 planes, %5d blocks x %3d steps: %8.3f ms -> %.1f G atomics/s\n", blocks, iters, ms, n / ms / 1e6);
    k<2><<<blocks, 512>>>(planes, stride, W, 4, tiles_w);
    hipEventRecord(a);
    k<2><<<blocks, 512>>>(planes, stride, W, iters, tiles_w);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("L2 float atomics x2 planes, %5d blocks x %3d steps: %8.3f ms -> %.1f G atomics/s\n", blocks, iters, ms, 2.0 * blocks * 512.0 * iters / ms / 1e6);
  }
  return 0;
}
