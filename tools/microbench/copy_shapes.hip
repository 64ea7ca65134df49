// Microbenchmark: what the launch shape of the window stencils (window.hpp) costs against plain
// streaming, on an 8192^2 fp32 plane (read 4 + write 4 B per cell unless noted).
//   hipcc --offload-arch=gfx950 -O3 copy_shapes.hip -o copy_shapes && ./copy_shapes
//
//   flat        every thread one float4, blocks in address order
//   band R      256 threads own 1024 columns and walk R rows, one 16-byte load per row (RowWalk's
//               traffic without its arithmetic; R = 32 is kWinBand)
//   rows3       every thread one float4 of row x plus the float4s above and below (L2 re-use
//               instead of a register window), blocks in address order
//   set / memcpy  write only / hipMemcpyAsync device to device
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_flat(float4* __restrict__ o, const float4* __restrict__ in, int64_t n4) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n4) o[i] = in[i];
}
__global__ void __launch_bounds__(256) k_set(float4* __restrict__ o, int64_t n4) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n4) o[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
template <int R>
__global__ void __launch_bounds__(256) k_band(float* __restrict__ o, const float* __restrict__ in, int64_t H, int64_t W) {
  const int64_t y0 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  const int64_t x0 = static_cast<int64_t>(blockIdx.y) * R;
  if (y0 >= W) return;
  float4 up = *reinterpret_cast<const float4*>(in + (x0 > 0 ? x0 - 1 : 0) * W + y0);
  float4 mid = *reinterpret_cast<const float4*>(in + x0 * W + y0);
  for (int64_t x = x0; x < x0 + R && x < H; ++x) {
    const float4 dn = *reinterpret_cast<const float4*>(in + (x + 1 < H ? x + 1 : x) * W + y0);
    *reinterpret_cast<float4*>(o + x * W + y0) =
        make_float4(up.x + mid.x + dn.x, up.y + mid.y + dn.y, up.z + mid.z + dn.z, up.w + mid.w + dn.w);
    up = mid;
    mid = dn;
  }
}
__global__ void __launch_bounds__(256) k_rows3(float* __restrict__ o, const float* __restrict__ in, int64_t H, int64_t W) {
  const int64_t i = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (i >= H * W) return;
  const int64_t x = i / W;
  const float4 mid = *reinterpret_cast<const float4*>(in + i);
  const float4 up = *reinterpret_cast<const float4*>(in + (x > 0 ? i - W : i));
  const float4 dn = *reinterpret_cast<const float4*>(in + (x + 1 < H ? i + W : i));
  *reinterpret_cast<float4*>(o + i) =
      make_float4(up.x + mid.x + dn.x, up.y + mid.y + dn.y, up.z + mid.z + dn.z, up.w + mid.w + dn.w);
}

// block R (round 5): 256 threads own 1024 columns of R rows; every thread asks for its R + 2 rows at once
// (independent 16-byte loads), then writes its R outputs: a short-lived work-group like `flat`, in
// address order, with (R + 2) / R of the reads (the two extra rows out of the L2 the vertical
// neighbours fill: work-groups b * G + g and (b +- 1) * G + g share an XCD when G = W / 1024 is 8)
template <int R>
__global__ void __launch_bounds__(256) k_block(float* __restrict__ o, const float* __restrict__ in, int64_t H, int64_t W) {
  const int64_t G = W / 1024;
  const int64_t g = blockIdx.x % G, b = blockIdx.x / G;
  const int64_t y0 = (g * 256 + threadIdx.x) * 4;
  const int64_t x0 = b * R;
  float4 r[R + 2];
#pragma unroll
  for (int i = 0; i < R + 2; ++i) {
    int64_t x = x0 - 1 + i;
    x = x < 0 ? 0 : (x >= H ? H - 1 : x);
    r[i] = *reinterpret_cast<const float4*>(in + x * W + y0);
  }
#pragma unroll
  for (int i = 0; i < R; ++i)
    if (x0 + i < H)
      *reinterpret_cast<float4*>(o + (x0 + i) * W + y0) =
          make_float4(r[i].x + r[i + 1].x + r[i + 2].x, r[i].y + r[i + 1].y + r[i + 2].y,
                      r[i].z + r[i + 1].z + r[i + 2].z, r[i].w + r[i + 1].w + r[i + 2].w);
}
// band R with the row after next asked for before the current one is used (two loads in flight)
template <int R>
__global__ void __launch_bounds__(256) k_band2(float* __restrict__ o, const float* __restrict__ in, int64_t H, int64_t W) {
  const int64_t y0 = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  const int64_t x0 = static_cast<int64_t>(blockIdx.y) * R;
  auto row = [&](int64_t x) { x = x < 0 ? 0 : (x >= H ? H - 1 : x); return *reinterpret_cast<const float4*>(in + x * W + y0); };
  float4 up = row(x0 - 1), mid = row(x0), dn = row(x0 + 1);
  for (int64_t x = x0; x < x0 + R && x < H; ++x) {
    const float4 nx = row(x + 2);
    *reinterpret_cast<float4*>(o + x * W + y0) =
        make_float4(up.x + mid.x + dn.x, up.y + mid.y + dn.y, up.z + mid.z + dn.z, up.w + mid.w + dn.w);
    up = mid;
    mid = dn;
    dn = nx;
  }
}
// band R, work-groups numbered so that an XCD sweeps its own eighth of the rows top to bottom
template <int R>
__global__ void __launch_bounds__(256) k_band_xcd(float* __restrict__ o, const float* __restrict__ in, int64_t H, int64_t W) {
  const int64_t G = W / 1024, bands = H / R;
  const int64_t k = blockIdx.x & 7, i = blockIdx.x >> 3;      // XCD k, its i-th work-group
  const int64_t g = i % G, b = k * (bands / 8) + i / G;
  const int64_t y0 = (g * 256 + threadIdx.x) * 4;
  const int64_t x0 = b * R;
  float4 up = *reinterpret_cast<const float4*>(in + (x0 > 0 ? x0 - 1 : 0) * W + y0);
  float4 mid = *reinterpret_cast<const float4*>(in + x0 * W + y0);
  for (int64_t x = x0; x < x0 + R && x < H; ++x) {
    const float4 dn = *reinterpret_cast<const float4*>(in + (x + 1 < H ? x + 1 : x) * W + y0);
    *reinterpret_cast<float4*>(o + x * W + y0) =
        make_float4(up.x + mid.x + dn.x, up.y + mid.y + dn.y, up.z + mid.z + dn.z, up.w + mid.w + dn.w);
    up = mid;
    mid = dn;
  }
}

int main() {
  const int64_t H = 8192, W = 8192, n = H * W;
  float *a, *b;
  CK(hipMalloc(&a, n * 4));
  CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 0, n * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timed = [&](const char* name, double bytes, auto&& launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0, nullptr);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-12s %8.1f us  %6.0f GB/s\n", name, ms * 1e3, bytes / ms / 1e6);
  };
  const unsigned flat_blocks = static_cast<unsigned>(n / 4 / 256);
  timed("flat", 8.0 * n, [&] { k_flat<<<flat_blocks, 256>>>(reinterpret_cast<float4*>(b), reinterpret_cast<const float4*>(a), n / 4); });
  timed("set", 4.0 * n, [&] { k_set<<<flat_blocks, 256>>>(reinterpret_cast<float4*>(b), n / 4); });
  timed("memcpy", 8.0 * n, [&] { hipMemcpyAsync(b, a, n * 4, hipMemcpyDeviceToDevice, nullptr); });
  timed("band 8", 8.0 * n, [&] { k_band<8><<<dim3(W / 1024, H / 8), 256>>>(b, a, H, W); });
  timed("band 16", 8.0 * n, [&] { k_band<16><<<dim3(W / 1024, H / 16), 256>>>(b, a, H, W); });
  timed("band 32", 8.0 * n, [&] { k_band<32><<<dim3(W / 1024, H / 32), 256>>>(b, a, H, W); });
  timed("band 64", 8.0 * n, [&] { k_band<64><<<dim3(W / 1024, H / 64), 256>>>(b, a, H, W); });
  timed("band 128", 8.0 * n, [&] { k_band<128><<<dim3(W / 1024, H / 128), 256>>>(b, a, H, W); });
  timed("rows3", 8.0 * n, [&] { k_rows3<<<flat_blocks, 256>>>(b, a, H, W); });
  timed("block 2", 8.0 * n, [&] { k_block<2><<<static_cast<unsigned>(W / 1024 * (H / 2)), 256>>>(b, a, H, W); });
  timed("block 4", 8.0 * n, [&] { k_block<4><<<static_cast<unsigned>(W / 1024 * (H / 4)), 256>>>(b, a, H, W); });
  timed("block 8", 8.0 * n, [&] { k_block<8><<<static_cast<unsigned>(W / 1024 * (H / 8)), 256>>>(b, a, H, W); });
  timed("block 16", 8.0 * n, [&] { k_block<16><<<static_cast<unsigned>(W / 1024 * (H / 16)), 256>>>(b, a, H, W); });
  timed("band2 16", 8.0 * n, [&] { k_band2<16><<<dim3(W / 1024, H / 16), 256>>>(b, a, H, W); });
  timed("band2 32", 8.0 * n, [&] { k_band2<32><<<dim3(W / 1024, H / 32), 256>>>(b, a, H, W); });
  timed("band2 64", 8.0 * n, [&] { k_band2<64><<<dim3(W / 1024, H / 64), 256>>>(b, a, H, W); });
  timed("bandx 16", 8.0 * n, [&] { k_band_xcd<16><<<static_cast<unsigned>(W / 1024 * (H / 16)), 256>>>(b, a, H, W); });
  timed("bandx 32", 8.0 * n, [&] { k_band_xcd<32><<<static_cast<unsigned>(W / 1024 * (H / 32)), 256>>>(b, a, H, W); });
  timed("bandx 64", 8.0 * n, [&] { k_band_xcd<64><<<static_cast<unsigned>(W / 1024 * (H / 64)), 256>>>(b, a, H, W); });
  return 0;
}
