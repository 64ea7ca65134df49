// Microbenchmark: throughput of LDS float atomics vs integer atomics vs plain stores on gfx950.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic && ./lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE, int ACTIVE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  __shared__ float f[16384];
  __shared__ uint32_t u[4096];
  for (int i = threadIdx.x; i < 16384; i += 512) f[i] = 0.f;
  for (int i = threadIdx.x; i < 4096; i += 512) u[i] = 0;
  __syncthreads();
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  const bool act = (threadIdx.x & 63) < ACTIVE;
  float v = 1.0f + threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const int c = (s >> 8) & 4095;
    if (act) {
      if (MODE == 0) { atomicAdd(&f[c], v); atomicAdd(&f[4096 + c], v); atomicAdd(&f[8192 + c], v); atomicAdd(&f[12288 + c], v); }
      if (MODE == 1) { atomicAdd(&u[c], 1u); atomicAdd(&u[(c + 1) & 4095], 2u); atomicAdd(&u[(c + 2) & 4095], 3u); atomicAdd(&u[(c + 3) & 4095], 4u); }
      if (MODE == 2) { f[c] = v; f[4096 + c] = v; f[8192 + c] = v; f[12288 + c] = v; }
      if (MODE == 3) { float a = f[c], b = f[4096 + c], d = f[8192 + c], e = f[12288 + c]; v += a + b + d + e; }
      if (MODE == 4) { atomicAdd(&f[c * 4], v); atomicAdd(&f[c * 4 + 1], v); atomicAdd(&f[c * 4 + 2], v); atomicAdd(&f[c * 4 + 3], v); }
      if (MODE == 5) { atomicAdd(&f[c], v); }
      if (MODE == 6) {  // 4 float adds through integer compare-and-swap loops
        uint32_t* p0 = reinterpret_cast<uint32_t*>(&f[c]);
        uint32_t* p1 = p0 + 4096; uint32_t* p2 = p0 + 8192; uint32_t* p3 = p0 + 12288;
        uint32_t a0 = *p0, a1 = *p1, a2 = *p2, a3 = *p3;
        bool d0 = false, d1 = false, d2 = false, d3 = false;
        for (;;) {
          uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
          if (!d0) r0 = atomicCAS(p0, a0, __float_as_uint(__uint_as_float(a0) + v));
          if (!d1) r1 = atomicCAS(p1, a1, __float_as_uint(__uint_as_float(a1) + v));
          if (!d2) r2 = atomicCAS(p2, a2, __float_as_uint(__uint_as_float(a2) + v));
          if (!d3) r3 = atomicCAS(p3, a3, __float_as_uint(__uint_as_float(a3) + v));
          if (!d0) { d0 = r0 == a0; a0 = r0; }
          if (!d1) { d1 = r1 == a1; a1 = r1; }
          if (!d2) { d2 = r2 == a2; a2 = r2; }
          if (!d3) { d3 = r3 == a3; a3 = r3; }
          if (d0 && d1 && d2 && d3) break;
        }
      }
    }
    v = v * 1.0000001f;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = f[0] + u[0] + v;
}

template <int MODE, int ACTIVE>
void run(const char* name) {
  float* out; hipMalloc(&out, 4096 * 4);
  const int iters = 2000, blocks = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, ACTIVE><<<blocks, 512>>>(out, 10);
  hipEventRecord(a);
  k<MODE, ACTIVE><<<blocks, 512>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per CU: 8 waves x iters x (4 LDS instrs, 1 for MODE 5)
  const double instr = 8.0 * iters * (MODE == 5 ? 1 : 4);
  printf("%-44s active lanes %2d: %8.3f ms  -> %.1f cycles(2.1GHz) per wave-instr per CU\n", name, ACTIVE, ms,
         ms * 1e-3 * 2.1e9 / instr);
  hipFree(out);
}

int main() {
  run<0, 64>("ds_add_f32 x4 (SoA planes, random cell)");
  run<0, 32>("ds_add_f32 x4 (SoA planes, random cell)");
  run<0, 8>("ds_add_f32 x4 (SoA planes, random cell)");
  run<0, 2>("ds_add_f32 x4 (SoA planes, random cell)");
  run<0, 1>("ds_add_f32 x4 (SoA planes, random cell)");
  run<5, 16>("ds_add_f32 x1");
  run<5, 4>("ds_add_f32 x1");
  run<5, 1>("ds_add_f32 x1");
  run<6, 2>("float add via ds_cmpst CAS loops x4");
  run<4, 64>("ds_add_f32 x4 (AoS float4, random cell)");
  run<5, 64>("ds_add_f32 x1");
  run<6, 64>("float add via ds_cmpst CAS loops x4");
  run<6, 32>("float add via ds_cmpst CAS loops x4");
  run<1, 64>("ds_add_u32 x4 (random)");
  run<2, 64>("ds_write_b32 x4 (random)");
  run<3, 64>("ds_read_b32 x4 (random)");
  return 0;
}
