// quot_check.hip — brute-force check of the shared-reciprocal quotient (soil_math.hpp:
// recip/quot) against the compiler's IEEE `/` on random plain operands, on the device.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I soillib_amd/csrc \
//         tools/microbench/quot_check.hip -o /tmp/quot_check && /tmp/quot_check [log2 pairs]
//
// Operands: |b| in [2^-40, 2^40), |a| in [2^-80, 2^50), random signs and mantissas;
// every 4th pair has a within 4 ulp of b (quotients around 1, where a wrong rounding
// would show first); every 4th takes a from all normal numbers and zeros of both
// signs and is checked when the quotient comes out in [2^-60, 2^90] (the rule for
// debris' decay_d in erosion_particles_tiled.hip).  Prints the number of pairs and of mismatching bit patterns.
#include <cstdio>
#include <cstdlib>

#include "soil_math.hpp"

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void k_check(unsigned long long* bad, unsigned long long* first, uint64_t per_thread) {
  const uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  unsigned long long mine = 0;
  for (uint64_t j = 0; j < per_thread; ++j) {
    const uint64_t h = mix(t * per_thread + j), g = mix(h);
    const uint32_t eb = 127 - 40 + static_cast<uint32_t>(h % 80), ea = 127 - 80 + static_cast<uint32_t>((h >> 8) % 130);
    uint32_t ub = (static_cast<uint32_t>(g) & 0x807fffffu) | (eb << 23);
    uint32_t ua = (static_cast<uint32_t>(g >> 32) & 0x807fffffu) | (ea << 23);
    if ((j & 3) == 3) ua = ub + static_cast<uint32_t>(h >> 60) - 4u;
    if ((j & 3) == 2) {  // any normal numerator: accepted by the size of the quotient
      ua = (ua & 0x807fffffu) | ((1u + static_cast<uint32_t>((h >> 16) % 254)) << 23);
      if ((h >> 40) % 64 == 0) ua &= 0x80000000u;  // and zeros of both signs
    }
    const float a = soil::bits2f(ua), b = soil::bits2f(ub);
    const float want = a / b, got = soil::quot0(a, soil::recip(b));
    if ((j & 3) == 2 && a != 0.0f && !(fabsf(got) >= soil::kNumLo && fabsf(got) <= 0x1p90f)) continue;
    if (soil::f2bits(want) != soil::f2bits(got)) {
      ++mine;
      atomicMin(first, (static_cast<unsigned long long>(ua) << 32) | ub);
    }
  }
  if (mine) atomicAdd(bad, mine);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? std::atoi(argv[1]) : 34;
  const uint64_t threads = 256ull * 8192ull, per = (1ull << lg) / threads;
  unsigned long long *bad, *first, h_bad = 0, h_first = ~0ull;
  hipMalloc(&bad, 8);
  hipMalloc(&first, 8);
  hipMemcpy(bad, &h_bad, 8, hipMemcpyHostToDevice);
  hipMemcpy(first, &h_first, 8, hipMemcpyHostToDevice);
  k_check<<<8192, 256>>>(bad, first, per);
  hipDeviceSynchronize();
  hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&h_first, first, 8, hipMemcpyDeviceToHost);
  std::printf("pairs %llu mismatches %llu", static_cast<unsigned long long>(threads * per), h_bad);
  if (h_bad) std::printf(" first a=%08x b=%08x", static_cast<unsigned>(h_first >> 32), static_cast<unsigned>(h_first));
  std::printf("\n");
  return h_bad != 0;
}
