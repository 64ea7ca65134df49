// Microbenchmark: returning 32-bit atomic adds on random counters (the rank draw of the spawn and of
// a round's survivors: 8.4 M walkers onto 53 k queue-section counters), device scope against
// work-group scope on a private copy of the counters per XCD (the atomic is then carried out in the
// XCD's own L2; s_getreg XCC_ID picks the copy).
//   hipcc --offload-arch=gfx950 -O3 atomic_scope.hip -o atomic_scope && ./atomic_scope
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <numeric>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

template <int MODE>  // 0 device scope, 1 work-group scope on the XCD's copy, 2 device scope, not returning
__global__ void __launch_bounds__(256) k_rank(uint32_t* __restrict__ rank, uint32_t* __restrict__ count, uint32_t keys, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t key = hash(static_cast<uint32_t>(i)) % keys;
  if (MODE == 0) {
    rank[i] = atomicAdd(&count[key], 1u);
  } else if (MODE == 1) {
    const uint32_t x = xcc_id();
    rank[i] = __hip_atomic_fetch_add(&count[x * keys + key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) | (x << 28);
  } else {
    atomicAdd(&count[key], 1u);
    rank[i] = key;
  }
}

int main() {
  const int64_t n = 8388608;
  const uint32_t keys = 53772;
  uint32_t *rank, *count;
  CK(hipMalloc(&rank, n * 4));
  CK(hipMalloc(&count, keys * 8 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timed = [&](const char* name, auto&& launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemsetAsync(count, 0, keys * 8 * 4, nullptr);
      hipEventRecord(e0, nullptr);
      launch();
      hipEventRecord(e1, nullptr);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("%-46s %8.1f us  %6.1f G atomics/s\n", name, best * 1e3, n / best / 1e6);
  };
  const unsigned blocks = static_cast<unsigned>(n / 256);
  timed("device scope, returning", [&] { k_rank<0><<<blocks, 256>>>(rank, count, keys, n); });
  timed("device scope, not returning", [&] { k_rank<2><<<blocks, 256>>>(rank, count, keys, n); });
  timed("work-group scope on the XCD's copy, returning", [&] { k_rank<1><<<blocks, 256>>>(rank, count, keys, n); });
  // every walker drew a distinct (xcd, key, rank): the per-XCD counts add up and the ranks of a copy are 0 .. count-1
  std::vector<uint32_t> hc(keys * 8), hr(n);
  CK(hipMemcpy(hc.data(), count, keys * 8 * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hr.data(), rank, n * 4, hipMemcpyDeviceToHost));
  uint64_t total = std::accumulate(hc.begin(), hc.end(), uint64_t(0));
  uint64_t bad = 0, per_xcd[8] = {0};
  for (int64_t i = 0; i < n; ++i) {
    uint32_t x = hr[i] >> 28, r = hr[i] & 0x0fffffffu;
    uint32_t h = static_cast<uint32_t>(i); h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    if (r >= hc[x * keys + h % keys]) ++bad;
    ++per_xcd[x];
  }
  printf("work-group scope: counts add up to %llu of %lld, %llu ranks out of range; walkers per XCD:", (unsigned long long)total, (long long)n, (unsigned long long)bad);
  for (int x = 0; x < 8; ++x) printf(" %llu", (unsigned long long)per_xcd[x]);
  printf("\n");
  // uniqueness of (xcd, key, rank)
  std::vector<uint8_t> seen;
  std::vector<uint64_t> base(keys * 8 + 1, 0);
  for (size_t k = 0; k < hc.size(); ++k) base[k + 1] = base[k] + hc[k];
  seen.assign(total, 0);
  uint64_t dup = 0;
  for (int64_t i = 0; i < n; ++i) {
    uint32_t x = hr[i] >> 28, r = hr[i] & 0x0fffffffu;
    uint32_t h = static_cast<uint32_t>(i); h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    const size_t k = x * keys + h % keys;
    if (r < hc[k]) { uint64_t s = base[k] + r; if (seen[s]) ++dup; seen[s] = 1; }
  }
  printf("duplicate (xcd, key, rank): %llu\n", (unsigned long long)dup);
  return 0;
}
