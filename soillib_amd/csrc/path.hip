// path.hip — soil::solve_uniform (path.hpp:30-37, path.cu:180-219): generic
// Monte-Carlo streamline estimator over a given velocity field.
//   __solve_uniform<K> path.cu:52-139, __normalize<K> :142-170,
//   bilinear gather sample.hpp:154-186 (view overload, y-fast indexing).
#include "common.hpp"

namespace soil {

constexpr int kUBlock = 256;

// sample_t<vec2,2,1>::gather(view) followed by val(): sample.hpp:154-186, :92-94, :48-50
__device__ __forceinline__ float2 bilinear(const float2* __restrict__ flow, int64_t H, int64_t W,
                                           float px, float py) {
  const float rx = static_cast<float>(H), ry = static_cast<float>(W);
  const float nan = __builtin_nanf("");
  if (px < 0 || py < 0 || px > rx - 1 || py > ry - 1) return make_float2(nan, nan);  // :167-170
  const int64_t ix = cell_of(px), iy = cell_of(py);        // :156-159
  float wx = px - floorf(px), wy = py - floorf(py);                                  // :160
  int64_t i00 = ix * W + iy, i01 = ix * W + (iy + 1);                                // :162-165
  int64_t i10 = (ix + 1) * W + iy, i11 = (ix + 1) * W + (iy + 1);
  if (px + 1 > rx - 1) {  // :172
    wx = 0;
    i10 = 0;
    i11 = 0;
  }
  if (py + 1 > ry - 1) {  // :173
    wy = 0;
    i01 = 0;
    i11 = 0;
  }
  const float2 h00 = flow[i00], h01 = flow[i01], h10 = flow[i10], h11 = flow[i11];
  const float ay = 1.0f + -1.0f * wy, by = 0.0f + 1.0f * wy;  // M()*vec2(1,t), :55-60
  const float ax = 1.0f + -1.0f * wx, bx = 0.0f + 1.0f * wx;
  const float l0x = ay * h00.x + by * h01.x, l0y = ay * h00.y + by * h01.y;  // :48-50
  const float l1x = ay * h10.x + by * h11.x, l1y = ay * h10.y + by * h11.y;
  return make_float2(ax * l0x + bx * l1x, ax * l0y + bx * l1y);  // :92-94
}

__device__ __forceinline__ bool oob_hw(int64_t H, int64_t W, float px, float py) {
  return px < 0 || py < 0 || px >= static_cast<float>(H) || py >= static_cast<float>(W);
}

template <int K>
__global__ void __launch_bounds__(kUBlock)
    k_solve_uniform(float* __restrict__ flux, const float2* __restrict__ flow,
                    const float* __restrict__ source, const float* __restrict__ decay,
                    soil_rng* __restrict__ rng, int64_t N, int64_t H, int64_t W, Scale2 s,
                    float epsilon, float maxstep) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kUBlock + threadIdx.x;
  if (n >= N) return;
  float att = 1.0f;  // :78
  soil_rng st = rng[n];
  float px = rng_uniform_at(st.seed, static_cast<uint64_t>(n), st.offset) * static_cast<float>(H);      // :81
  float py = rng_uniform_at(st.seed, static_cast<uint64_t>(n), st.offset + 1) * static_cast<float>(W);  // :82
  st.offset += 2;
  rng[n] = st;
  // a draw of exactly 1 lands on the far edge; the reference then reads out of
  // bounds at :90 (undefined) — the sample is dropped instead
  if (oob_hw(H, W, px, py)) return;
  int64_t ind = cell_of(px) * W + cell_of(py);  // :84
  const float L = sqrtf(s.x * s.x + s.y * s.y);                           // :87
  const float A = s.x * s.y;                                              // :88
  const float P = 1.0f / (A * static_cast<float>(H * W));                 // :89
  float S[K];
#pragma unroll
  for (int c = 0; c < K; ++c) S[c] = source[K * ind + c] / P;  // :90
  const float Slen = (K == 1) ? sqrtf(S[0] * S[0]) : sqrtf(S[0] * S[0] + S[K - 1] * S[K - 1]);
  if (Slen < epsilon) return;  // :91-92
  float2 v = bilinear(flow, H, W, px, py);  // :99-100
  int step = 0;
  while (!oob_hw(H, W, px, py) && epsilon < fabsf(att) &&
         static_cast<float>(++step) < maxstep) {  // :104
    const int64_t nind = cell_of(px) * W + cell_of(py);  // :107
    if (nind != ind) {                                                             // :108-116
      ind = nind;
#pragma unroll
      for (int c = 0; c < K; ++c) atomicAdd(&flux[K * ind + c], S[c] * att);
    }
    v = bilinear(flow, H, W, px, py);                   // :119-120
    const float v_len = sqrtf(v.x * v.x + v.y * v.y);   // :123
    if (v_len < epsilon) break;                         // :124-125
    const float ux = v.x / v_len, uy = v.y / v_len;     // :128
    const float stp = stepsize(px, py, ux, uy);         // :129
    px += stp * ux;                                     // :130
    py += stp * uy;
    const float dlambda = stp * L / v_len;       // :133
    att *= expf_(-dlambda * decay[ind]);         // :134
  }
}

// __normalize<K>, path.cu:142-170
template <int K>
__global__ void __launch_bounds__(kUBlock)
    k_path_normalize(float* __restrict__ flux, const float2* __restrict__ flow,
                     const float* __restrict__ source, int64_t cells, Scale2 s, float count) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kUBlock + threadIdx.x;
  if (n >= cells) return;
  const float2 v = flow[n];                                   // :160
  const float A = s.x * s.y;                                  // :161
  const float norm = fabsf(v.x * s.y) + fabsf(v.y * s.x);     // :162
#pragma unroll
  for (int c = 0; c < K; ++c)
    flux[K * n + c] = (source[K * n + c] * A + flux[K * n + c] / count) / norm;  // :168
}

template <int K>
static int solve_impl(float* flux, const float* flow, const float* source, const float* decay,
                      soil_rng* rng, int64_t N, int64_t H, int64_t W, Scale2 s, uint64_t count,
                      hipStream_t st) {
  SOIL_HIP(hipMemsetAsync(flux, 0, sizeof(float) * H * W * K, st));  // silt::set(flux, 0), :196
  const float epsilon = 1E-16f;                                       // :199
  const float maxstep = static_cast<float>(H + W);                    // :200
  if (N > 0)
    k_solve_uniform<K><<<blocks_for(N, kUBlock), kUBlock, 0, st>>>(
        flux, reinterpret_cast<const float2*>(flow), source, decay, rng, N, H, W, s, epsilon,
        maxstep);
  k_path_normalize<K><<<blocks_for(H * W, kUBlock), kUBlock, 0, st>>>(
      flux, reinterpret_cast<const float2*>(flow), source, H * W, s, static_cast<float>(count));
  SOIL_LAUNCH_CHECK();
  SOIL_HIP(hipStreamSynchronize(st));  // cudaDeviceSynchronize, :216
  return SOIL_OK;
}

}  // namespace soil

using namespace soil;

extern "C" int soil_solve_uniform(float* flux, const float* flow, const float* source,
                                  const float* decay, soil_rng* rng, int64_t N, int64_t H,
                                  int64_t W, int K, const float scale[2], uint64_t count,
                                  void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(flux && flow && source && decay && scale, "solve_uniform: null argument");
  SOIL_REQUIRE(H > 0 && W > 0 && N >= 0 && (N == 0 || rng), "solve_uniform: bad sizes");
  const Scale2 s{scale[0], scale[1]};
  if (K == 1) return solve_impl<1>(flux, flow, source, decay, rng, N, H, W, s, count, as_stream(stream));
  if (K == 2) return solve_impl<2>(flux, flow, source, decay, rng, N, H, W, s, count, as_stream(stream));
  // the reference silently returns zeros for other K (path.cu:212-213)
  return fail(SOIL_ERR_INVALID_ARGUMENT, "solve_uniform: source must have 1 or 2 channels");
}
