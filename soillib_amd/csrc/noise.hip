// noise.hip — soil::noise (op/noise.hpp:14-56): 3-D OpenSimplex2 noise summed
// as FBm, sampled at (x/ext0, y/ext1, seed).  The reference evaluates this on
// the host through the third-party FastNoiseLite it vendors; here one routine
// serves a device kernel (bench-sized heightmaps are generated straight into
// HBM) and a host twin, bit-identical to each other and to the reference's
// generator (pinned by tests/golden/noise_*.npy).
//
// OpenSimplex2 in 3-D (K.jpg): the point is rotated so that the main diagonal
// becomes the z axis, then evaluated on two cubic lattices offset by half a
// cell; on each lattice the nearest vertex and its neighbour along the
// dominant axis contribute a radially attenuated gradient ramp.
#include "common.hpp"

namespace soil {

constexpr int kNBlock = 256;
constexpr int32_t kPrime[3] = {501125321, 1136930381, 1720413743};

SOIL_HD int32_t wrap_mul(int32_t a, int32_t b) {
  return static_cast<int32_t>(static_cast<uint32_t>(a) * static_cast<uint32_t>(b));
}

// gradient ramp of one lattice vertex: hash -> one of 64 directions (the 12
// cube-edge vectors x5, plus 4 fill-ins), dotted with the offset
SOIL_HD float vertex_ramp(int32_t seed, const int32_t v[3], const float d[3]) {
  uint32_t h = static_cast<uint32_t>(seed ^ v[0] ^ v[1] ^ v[2]) * 0x27d4eb2du;
  int32_t hs = static_cast<int32_t>(h);
  hs ^= hs >> 15;
  const int idx = (hs >> 2) & 63;
  float g[3];
  if (idx < 60) {
    const int e = idx % 12, axis = e >> 2;
    const float a = (e & 1) ? -1.0f : 1.0f, b = (e & 2) ? -1.0f : 1.0f;
    g[0] = (axis == 0) ? 0.0f : a;
    g[1] = (axis == 0) ? a : ((axis == 1) ? 0.0f : b);
    g[2] = (axis == 2) ? 0.0f : b;
  } else {
    const int f = idx - 60;  // (1,1,0) (0,-1,1) (-1,1,0) (0,-1,-1)
    g[0] = (f == 0) ? 1.0f : ((f == 2) ? -1.0f : 0.0f);
    g[1] = (f & 1) ? -1.0f : 1.0f;
    g[2] = (f == 1) ? 1.0f : ((f == 3) ? -1.0f : 0.0f);
  }
  return d[0] * g[0] + d[1] * g[1] + d[2] * g[2];
}

SOIL_HD float simplex3(int32_t seed, float x, float y, float z) {
  const float p[3] = {x, y, z};
  int32_t cell[3], sgn[3];
  float d[3], ad[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int32_t r = p[c] >= 0 ? static_cast<int32_t>(p[c] + 0.5f) : static_cast<int32_t>(p[c] - 0.5f);
    d[c] = p[c] - static_cast<float>(r);
    sgn[c] = static_cast<int32_t>(-1.0f - d[c]) | 1;
    ad[c] = static_cast<float>(sgn[c]) * -d[c];
    cell[c] = wrap_mul(r, kPrime[c]);
  }
  float value = 0.0f;
  float a = (0.6f - d[0] * d[0]) - (d[1] * d[1] + d[2] * d[2]);
#pragma unroll
  for (int lattice = 0; lattice < 2; ++lattice) {
    if (a > 0) value += (a * a) * (a * a) * vertex_ramp(seed, cell, d);
    // neighbour along the axis with the largest |offset| (ties: x, then y)
    const int ax = (ad[0] >= ad[1] && ad[0] >= ad[2]) ? 0 : ((ad[1] > ad[0] && ad[1] >= ad[2]) ? 1 : 2);
    float d1[3] = {d[0], d[1], d[2]};
    int32_t c1[3] = {cell[0], cell[1], cell[2]};
    float b = a + 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (c == ax) {
        d1[c] += static_cast<float>(sgn[c]);
        b -= static_cast<float>(sgn[c] * 2) * d1[c];
        c1[c] -= wrap_mul(sgn[c], kPrime[c]);
      }
    }
    if (b > 0) value += (b * b) * (b * b) * vertex_ramp(seed, c1, d1);
    if (lattice == 1) break;
    // step to the second lattice (offset by half a cell on every axis)
#pragma unroll
    for (int c = 0; c < 3; ++c) ad[c] = 0.5f - ad[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = static_cast<float>(sgn[c]) * ad[c];
    a += (0.75f - ad[0]) - (ad[1] + ad[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      cell[c] += (sgn[c] >> 1) & kPrime[c];
      sgn[c] = -sgn[c];
    }
    seed = ~seed;
  }
  return value * 32.69428253173828125f;
}

struct NoiseSetup {
  float frequency, gain, lacunarity, seed, ext0, ext1, bounding;
  int octaves;
};

SOIL_HD float noise_cell(int64_t n, int64_t W, const NoiseSetup& q) {
  const int32_t px = static_cast<int32_t>(n / W), py = static_cast<int32_t>(n % W);
  float x = static_cast<float>(px) / q.ext0, y = static_cast<float>(py) / q.ext1, z = q.seed;  // noise.hpp:38
  x *= q.frequency;
  y *= q.frequency;
  z *= q.frequency;
  const float R3 = static_cast<float>(2.0 / 3.0);
  const float r = (x + y + z) * R3;  // rotation onto the main diagonal
  x = r - x;
  y = r - y;
  z = r - z;
  int32_t seed = 1337;  // the generator's default integer seed; noise_param_t never changes it
  float sum = 0.0f, amp = q.bounding;
  for (int o = 0; o < q.octaves; ++o) {
    const float v = simplex3(seed++, x, y, z);
    sum += v * amp;
    amp *= 1.0f + 0.0f * ((v + 1) * 0.5f - 1.0f);  // weighted strength 0 (generator default)
    x *= q.lacunarity;
    y *= q.lacunarity;
    z *= q.lacunarity;
    amp *= q.gain;
  }
  return sum;
}

__global__ void __launch_bounds__(kNBlock)
    k_noise(float* __restrict__ out, int64_t cells, int64_t W, int64_t first_cell, NoiseSetup q) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kNBlock + threadIdx.x;
  if (n < cells) out[n] = noise_cell(first_cell + n, W, q);
}

static NoiseSetup make_setup(const soil_noise_param* p) {
  NoiseSetup q;
  q.frequency = p->frequency;
  q.gain = p->gain;
  q.lacunarity = p->lacunarity;
  q.seed = p->seed;
  q.ext0 = p->ext[0];
  q.ext1 = p->ext[1];
  q.octaves = p->octaves;
  // amplitude normalisation of the octave sum: 1 / (1 + g + g^2 + ... + g^(octaves-1))
  const float g = p->gain < 0 ? -p->gain : p->gain;
  float amp = g, total = 1.0f;
  for (int o = 1; o < p->octaves; ++o) {
    total += amp;
    amp *= g;
  }
  q.bounding = 1 / total;
  return q;
}

}  // namespace soil

using namespace soil;

extern "C" {

void soil_noise_param_default(soil_noise_param* p) {  // noise.hpp:29-34
  p->frequency = 1.0f;
  p->octaves = 8;
  p->gain = 0.6f;
  p->lacunarity = 2.0f;
  p->seed = 0.0f;
  p->ext[0] = 512.0f;
  p->ext[1] = 512.0f;
}

int soil_noise(float* out, int64_t H, int64_t W, const soil_noise_param* p, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && p, "noise: null argument");
  SOIL_REQUIRE(H > 0 && W > 0, "noise: empty grid");  // noise.hpp:44-45 rejects non-2D shapes
  k_noise<<<blocks_for(H * W, kNBlock), kNBlock, 0, as_stream(stream)>>>(out, H * W, W, 0,
                                                                         make_setup(p));
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_noise_window(float* out, int64_t rows, int64_t W, int64_t x0,
                      const soil_noise_param* p, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && p, "noise_window: null argument");
  SOIL_REQUIRE(rows > 0 && W > 0 && x0 >= 0, "noise_window: bad window");
  k_noise<<<blocks_for(rows * W, kNBlock), kNBlock, 0, as_stream(stream)>>>(out, rows * W, W,
                                                                            x0 * W, make_setup(p));
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_noise_host(float* out, int64_t H, int64_t W, const soil_noise_param* p) {
  SOIL_REQUIRE(out && p, "noise_host: null argument");
  SOIL_REQUIRE(H > 0 && W > 0, "noise_host: empty grid");
  const NoiseSetup q = make_setup(p);
  for (int64_t n = 0; n < H * W; ++n) out[n] = noise_cell(n, W, q);  // noise.hpp:49-52
  return SOIL_OK;
}

}  // extern "C"
