// Microbenchmark: the round kernel's 16-byte record gather (erosion_particles_tiled.hip: `p4[lcell]`) by itself.
// A work-group of 768 lanes owns a 64x64-cell tile of a 8192^2 plane of float4 records (64 KiB of the
// plane), two work-groups per CU (LDS held as the round kernel holds it), every lane walks its cell by
// one step per iteration in a direction of its own and loads the cell's record; the loaded value feeds
// the next iteration's direction choice only weakly (no address dependence: what is measured is the
// throughput of 64-line gathers through the vector cache, as in the round kernel where the other
// waves cover the latency).
//   policy 0 plain | 1 nontemporal (nt) | 2 sc1 | 3 sc0 sc1
//   layout 0 row-major (a 128-byte line = 8 cells of a row) | 1 blocks of 2 rows x 4 columns per line |
//          2 blocks of 4 x 2
//   bytes  16 (dwordx4) | 8 (dwordx2: what a record of two floats would cost)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int POLICY>
__device__ __forceinline__ v4f load16(const v4f* p) {
  v4f r;
  if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory");
  if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
  if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r) : "v"(p) : "memory");
  return r;
}
template <int POLICY>
__device__ __forceinline__ v2f load8(const v2f* p) {
  v2f r;
  if (POLICY == 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  if (POLICY == 1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory");
  if (POLICY == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
  if (POLICY == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(r) : "v"(p) : "memory");
  return r;
}

template <int LAYOUT>
__device__ __forceinline__ uint32_t cell_index(uint32_t row, uint32_t col, uint32_t W) {
  if (LAYOUT == 0) return row * W + col;
  if (LAYOUT == 1) return ((row >> 1) * (W >> 2) + (col >> 2)) * 8u + ((row & 1u) << 2) + (col & 3u);
  return ((row >> 2) * (W >> 1) + (col >> 1)) * 8u + ((row & 3u) << 1) + (col & 1u);
}

template <int POLICY, int LAYOUT, int BYTES, int DEPTH>
__global__ void __launch_bounds__(768) k_gather(float* __restrict__ sink, const void* __restrict__ plane, uint32_t W,
                                               int iters, int tiles_w, int lds_words) {
  extern __shared__ float s_hold[];  // occupancy: what the round kernel's accumulators take
  if (threadIdx.x < 4) s_hold[threadIdx.x * (lds_words / 4)] = 0.0f;
  const uint32_t tile = blockIdx.x;
  const uint32_t row0 = (tile / tiles_w) * 64u, col0 = (tile % tiles_w) * 64u;
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  // a walker: position in 1/256 cells, a direction that turns slowly (channels: persistent directions)
  int pr = (s >> 3) & 0x3fff, pc = (s >> 17) & 0x3fff;
  int dr = static_cast<int>((s >> 5) & 511) - 256, dc = static_cast<int>((s >> 14) & 511) - 256;
  float acc = 0.0f;
  v4f q[DEPTH];
  for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      s = s * 1664525u + 1013904223u;
      dr += static_cast<int>((s >> 9) & 63) - 32;
      dc += static_cast<int>((s >> 19) & 63) - 32;
      dr = dr > 256 ? 256 : (dr < -256 ? -256 : dr);
      dc = dc > 256 ? 256 : (dc < -256 ? -256 : dc);
      pr = (pr + dr) & 0x3fff;
      pc = (pc + dc) & 0x3fff;
      const uint32_t l = cell_index<LAYOUT>(row0 + (static_cast<uint32_t>(pr) >> 8), col0 + (static_cast<uint32_t>(pc) >> 8), W);
      if (BYTES == 16) {
        q[j] = load16<POLICY>(static_cast<const v4f*>(plane) + l);
      } else {
        const v2f h = load8<POLICY>(static_cast<const v2f*>(plane) + 2 * l);
        q[j] = v4f{h.x, h.y, 0.0f, 0.0f};
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      asm volatile("" : "+v"(q[j]));
      acc += q[j].x + q[j].y;
    }
  }
  if (acc == 123.456f) sink[blockIdx.x * 768 + threadIdx.x] = acc + s_hold[0];
}

template <int POLICY, int LAYOUT, int BYTES, int DEPTH>
static void run(const char* what, float* sink, const void* plane, int blocks, int iters) {
  const int lds = 78 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_gather<POLICY, LAYOUT, BYTES, DEPTH>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k_gather<POLICY, LAYOUT, BYTES, DEPTH><<<blocks, 768, lds>>>(sink, plane, 8192u, 8, 128, lds / 4);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    k_gather<POLICY, LAYOUT, BYTES, DEPTH><<<blocks, 768, lds>>>(sink, plane, 8192u, iters, 128, lds / 4);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  const double n = static_cast<double>(blocks) * 768.0 * iters;
  printf("%-58s %8.3f ms  %7.1f G gathers/s\n", what, best, n / best / 1e6);
  fflush(stdout);
}

int main() {
  const size_t bytes = static_cast<size_t>(8192) * 8192 * 16;
  void* plane;
  float* sink;
  hipMalloc(&plane, bytes);
  hipMemset(plane, 0, bytes);
  hipMalloc(&sink, sizeof(float) * 16384 * 768);
  const int blocks = 16384, iters = 64;  // every tile of the 8192^2 plane, a round's worth of steps
  printf("16-byte gathers of a tile's records, 16384 tiles x 768 lanes x 64 steps (two work-groups per CU)\n");
  run<0, 0, 16, 1>("plain, row-major lines", sink, plane, blocks, iters);
  run<1, 0, 16, 1>("nt, row-major lines", sink, plane, blocks, iters);
  run<2, 0, 16, 1>("sc1, row-major lines", sink, plane, blocks, iters);
  run<3, 0, 16, 1>("sc0 sc1, row-major lines", sink, plane, blocks, iters);
  run<0, 1, 16, 1>("plain, lines of 2 x 4 cells", sink, plane, blocks, iters);
  run<0, 2, 16, 1>("plain, lines of 4 x 2 cells", sink, plane, blocks, iters);
  run<1, 1, 16, 1>("nt, lines of 2 x 4 cells", sink, plane, blocks, iters);
  run<2, 1, 16, 1>("sc1, lines of 2 x 4 cells", sink, plane, blocks, iters);
  run<0, 0, 8, 1>("plain, 8-byte records, row-major (16 cells a line)", sink, plane, blocks, iters);
  run<0, 0, 16, 2>("plain, row-major, two gathers in flight per lane", sink, plane, blocks, iters);
  run<0, 0, 16, 4>("plain, row-major, four gathers in flight per lane", sink, plane, blocks, iters);
  run<2, 0, 16, 4>("sc1, row-major, four gathers in flight per lane", sink, plane, blocks, iters);
  run<0, 1, 16, 4>("plain, 2 x 4 lines, four gathers in flight per lane", sink, plane, blocks, iters);
  return 0;
}
