// erosion_particles.hip — Monte-Carlo transport half of the erosion model:
//   __transport_fluvial erosion.cu:29-141, __transport_debris :245-351, and the
//   host wrappers soil::transport_fluvial :189-239 / soil::transport_debris :395-436.
//
// One lane integrates one streamline.  The flux planes are accumulated with
// hardware fp32 atomics (global_atomic_add_f32; built with -munsafe-fp-atomics).
#include "cell_math.hpp"

namespace soil {

constexpr int kPBlock = 256;

int launch_normalize_fluvial(const float* waterFlux, const float* massFlux,
                             const float* velocityFlux, float* albedoFlux, const float* layers,
                             const float* waterSource, float* waterHeight, float* mass,
                             float* velocity, const float* albedoSource, const Dom& d, Scale3 s,
                             const Param& p, hipStream_t st);
int launch_normalize_debris(const float* massFlux, const float* velocityFlux, float* albedoFlux,
                            const float* layers, float* mass, float* velocity,
                            const float* albedoSource, const Dom& d, Scale3 s, const Param& p,
                            hipStream_t st);

__device__ __forceinline__ bool oob(const Dom& d, float px, float py) {  // erosion_map.cu:29-40
  if (px < 0) return true;
  if (py < 0) return true;
  if (px >= static_cast<float>(d.H)) return true;
  if (py >= static_cast<float>(d.W)) return true;
  return false;
}

// A slab traces a particle only while the cell's 5-point stencil lies inside
// the rows it holds (see soil_hip.h, soil_particles_*_slab).
__device__ __forceinline__ bool slab_escape(const Dom& d, int64_t gx) {
  const int64_t lx = gx - d.x0;
  const int64_t lo = (d.x0 == 0) ? 0 : 1;
  const int64_t hi = (d.x0 + d.rows == d.H) ? d.rows - 1 : d.rows - 2;
  return lx < lo || lx > hi;
}

__global__ void __launch_bounds__(kPBlock)
    k_particles_fluvial(float* __restrict__ waterFlux, float* __restrict__ massFlux,
                        float* __restrict__ velocityFlux, float* __restrict__ albedoFlux,
                        soil_rng* __restrict__ rng, int64_t N, const float2* __restrict__ layers,
                        const float* __restrict__ waterSource,
                        const float* __restrict__ waterHeight, const float2* __restrict__ velocity,
                        const float* __restrict__ albedoSource, Dom d, Scale3 s, Param param) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kPBlock + threadIdx.x;
  if (n >= N) return;

  const float A = s.x * s.y;                                       // :50
  const float Lx = s.x, Ly = s.y;                                  // :51
  const float P = 1.0f / (A * static_cast<float>(d.H * d.W));      // :53
  const float Q = 1.0f / (P * static_cast<float>(N));              // :54
  const float eps = 1E-12f;                                        // :55

  soil_rng st = rng[n];
  const float u1 = rng_uniform_at(st.seed, static_cast<uint64_t>(n), st.offset);      // :57
  const float u2 = rng_uniform_at(st.seed, static_cast<uint64_t>(n), st.offset + 1);  // :58
  st.offset += 2;
  rng[n] = st;  // the state persists in the tensor, like curandState
  float px = 0.5f + u1 * static_cast<float>(d.H - 1);
  float py = 0.5f + u2 * static_cast<float>(d.W - 1);
  {
    const int64_t sx = static_cast<int64_t>(px) - d.x0;  // spawn-row ownership
    if (sx < d.r0 || sx >= d.r1) return;
  }
  const int64_t W = d.W;
  const int64_t base = d.x0 * W;
  int64_t ind = static_cast<int64_t>(px) * W + static_cast<int64_t>(py);  // :60

  const float rho_w = param.densityWater;                 // :63
  const float tau = param.bedShearWater;                  // :65
  const float nu = param.viscosityWater;                  // :66
  const float g = param.gravity;                          // :67
  const float ks = param.suspensionRateFluvial / 64.0f;   // :68
  const float kd = param.depositionRateFluvial * 1.33f;   // :69
  const float fD = param.frictionFactor / 8.0f;           // :70
  const float alpha = param.fluvialExponent;              // :71
  const float R = param.rainfall;                         // :72

  const float2 vel = velocity[ind - base];  // :75
  float2 grad = glocal(layers, d, s, static_cast<int64_t>(px), static_cast<int64_t>(py),
                       param.exitSlope);  // :76
  float spx = -(g * grad.x) + nu * vel.x + param.force[0];  // :77
  float spy = -(g * grad.y) + nu * vel.y + param.force[1];
  {
    const float den = sqrtf(length2(Lx * spx, Ly * spy));  // :78
    spx = spx / den;
    spy = spy / den;
  }
  if (length2(spx, spy) < eps) return;  // :79-80

  const float v = length2(vel.x, vel.y);                                // :83
  const float shear = 0.125f * fD * rho_w * v * v;                      // :84
  const float power = powf_(shear * length2(grad.x, grad.y), alpha);    // :85
  const float source_m = Q * ks * power;                                // :88
  const float source_w = Q * R * waterSource[ind - base];               // :89
  const float source_vx = Q * (-(g * grad.x) + nu * vel.x);             // :90
  const float source_vy = Q * (-(g * grad.y) + nu * vel.y);
  float source_a[3] = {0.0f, 0.0f, 0.0f};
  if (albedoSource)  // :91
    for (int c = 0; c < 3; ++c) source_a[c] = source_m * albedoSource[3 * (ind - base) + c];

  float att_w = 1.0f, att_m = 1.0f, att_v = 1.0f;  // :94-96
  const float lenL = length2(Lx, Ly);
  uint64_t iter = 0;
  while (!oob(d, px, py) && ++iter < param.maxage) {  // :100
    const int64_t cx = static_cast<int64_t>(px), cy = static_cast<int64_t>(py);
    if (slab_escape(d, cx)) break;
    const int64_t nind = cx * W + cy;  // :103
    if (nind != ind) {                 // :104-113
      ind = nind;
      const int64_t l = ind - base;
      atomicAdd(&waterFlux[l], att_w * source_w);
      atomicAdd(&massFlux[l], att_m * source_m);
      atomicAdd(&velocityFlux[2 * l], att_v * source_vx);
      atomicAdd(&velocityFlux[2 * l + 1], att_v * source_vy);
      if (albedoFlux)
        for (int c = 0; c < 3; ++c) atomicAdd(&albedoFlux[3 * l + c], att_m * source_a[c]);
    }
    const float v_norm = length2(spx, spy);             // :116
    const float ux = spx / v_norm, uy = spy / v_norm;   // :117
    const float v_step = stepsize(px, py, ux, uy);      // :118
    const float dL = v_step * lenL;                     // :119
    const float ds = dL / v_norm;                       // :120
    if (v_norm < eps) break;                            // :121-122

    grad = glocal(layers, d, s, cx, cy, param.exitSlope);  // :125
    const int64_t l = ind - base;
    const float2 vc = velocity[l];
    const float ax = -(g * grad.x) + nu * vc.x + param.force[0];  // :126
    const float ay = -(g * grad.y) + nu * vc.y + param.force[1];
    const float w0 = 1.0f / (1.0f + dL * (tau + nu));  // :127
    const float w1 = dL / (1.0f + dL * (tau + nu));
    spx = w0 * spx + w1 * ax;
    spy = w0 * spy + w1 * ay;

    const float decay_m = kd;                                    // :130
    const float decay_w = param.evapRate;                        // :131
    const float decay_v = 0.125f * fD / (eps + waterHeight[l]);  // :132
    att_m = att_m * expf_(-ds * decay_m);                        // :134
    att_w = att_w * expf_(-ds * decay_w);                        // :135
    att_v = att_v * expf_(-dL * decay_v);                        // :136
    px += v_step * ux;                                           // :137
    py += v_step * uy;
  }
}

__global__ void __launch_bounds__(kPBlock)
    k_particles_debris(float* __restrict__ massFlux, float* __restrict__ velocityFlux,
                       float* __restrict__ albedoFlux, soil_rng* __restrict__ rng, int64_t N,
                       const float2* __restrict__ layers, const float2* __restrict__ velocity,
                       const float* __restrict__ albedoSource, Dom d, Scale3 s, Param param) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kPBlock + threadIdx.x;
  if (n >= N) return;

  const float A = s.x * s.y;                                   // :263
  const float Lx = s.x, Ly = s.y;                              // :264
  const float P = 1.0f / (A * static_cast<float>(d.H * d.W));  // :266
  const float Q = 1.0f / (P * static_cast<float>(N));          // :267
  const float eps = 1E-12f;                                    // :268

  soil_rng st = rng[n];
  const float u1 = rng_uniform_at(st.seed, static_cast<uint64_t>(n), st.offset);      // :270
  const float u2 = rng_uniform_at(st.seed, static_cast<uint64_t>(n), st.offset + 1);  // :271
  st.offset += 2;
  rng[n] = st;
  float px = 0.5f + u1 * static_cast<float>(d.H - 1);
  float py = 0.5f + u2 * static_cast<float>(d.W - 1);
  {
    const int64_t sx = static_cast<int64_t>(px) - d.x0;
    if (sx < d.r0 || sx >= d.r1) return;
  }
  const int64_t W = d.W;
  const int64_t base = d.x0 * W;
  int64_t ind = static_cast<int64_t>(px) * W + static_cast<int64_t>(py);  // :273

  const float theta = param.critSlopeBedrock;    // :276
  const float nu = param.viscosityDebris;        // :277
  const float tau = param.bedShearDebris;        // :278
  const float g = param.gravity;                 // :279
  const float kl = param.landslideRateDebris;    // :280
  const float kdd = param.depositionRateDebris;  // :281
  const float kds = param.suspensionRateDebris;  // :282
  const float tau_y = param.yieldStress;         // :283

  const float2 vel = velocity[ind - base];  // :286
  float2 grad = glocal(layers, d, s, static_cast<int64_t>(px), static_cast<int64_t>(py),
                       param.exitSlope);  // :287
  float spx = -(g * grad.x) + nu * vel.x;  // :288
  float spy = -(g * grad.y) + nu * vel.y;
  {
    const float den = sqrtf(length2(Lx * spx, Ly * spy));  // :289
    spx = spx / den;
    spy = spy / den;
  }
  if (length2(spx, spy) < eps) return;  // :290-291

  const float excessSlope0 = length2(grad.x, grad.y) - theta;  // :294
  const float suspend = fmaxf(0.0f, kl * excessSlope0);        // :295
  const float source_d = Q * suspend;                          // :297
  const float source_vx = Q * (-g * grad.x + nu * vel.x);      // :298
  const float source_vy = Q * (-g * grad.y + nu * vel.y);
  float source_a[3] = {0.0f, 0.0f, 0.0f};
  if (albedoSource)  // :299
    for (int c = 0; c < 3; ++c) source_a[c] = source_d * albedoSource[3 * (ind - base) + c];

  float att_d = 1.0f, att_v = 1.0f;  // :301-302
  const float lenL = length2(Lx, Ly);
  uint64_t iter = 0;
  while (!oob(d, px, py) && ++iter < param.maxage) {  // :306
    const int64_t cx = static_cast<int64_t>(px), cy = static_cast<int64_t>(py);
    if (slab_escape(d, cx)) break;
    const int64_t nind = cx * W + cy;  // :309
    if (nind != ind) {                 // :310-318
      ind = nind;
      const int64_t l = ind - base;
      atomicAdd(&massFlux[l], att_d * source_d);
      atomicAdd(&velocityFlux[2 * l], att_v * source_vx);
      atomicAdd(&velocityFlux[2 * l + 1], att_v * source_vy);
      if (albedoFlux)
        for (int c = 0; c < 3; ++c) atomicAdd(&albedoFlux[3 * l + c], att_d * source_a[c]);
    }
    const float v_norm = length2(spx, spy);            // :321
    const float ux = spx / v_norm, uy = spy / v_norm;  // :322
    const float v_step = stepsize(px, py, ux, uy);     // :323
    const float dL = v_step * lenL;                    // :324
    const float ds = dL / v_norm;                      // :325
    if (v_norm < eps) break;                           // :326-327

    grad = glocal(layers, d, s, cx, cy, param.exitSlope);  // :330
    const int64_t l = ind - base;
    const float2 vc = velocity[l];
    const float debrisHeight = eps + att_d * source_d;  // :331
    const float ax = -(g * grad.x) + nu * vc.x;         // :332
    const float ay = -(g * grad.y) + nu * vc.y;
    const float decay = nu + tau / debrisHeight;        // :333
    const float w = 1.0f / (1.0f + dL * decay);         // :334
    spx = w * spx + w * dL * ax;                        // :335
    spy = w * spy + w * dL * ay;

    const float excessSlope = length2(grad.x, grad.y) - theta;            // :339
    const float excessStress = g * (excessSlope - tau_y / debrisHeight);  // :340
    const float shearRate = (excessStress < 0.0f) ? kdd : kds;            // :341
    const float decay_d = ds * shearRate * excessStress / v_norm;         // :342
    const float decay_v = nu + tau / debrisHeight;                        // :343
    att_d = att_d * expf_(decay_d);                                       // :345
    att_v = att_v * expf_(-dL * decay_v);                                 // :346
    px += v_step * ux;                                                    // :347
    py += v_step * uy;
  }
}

static Scale3 s3p(const float* s) { return Scale3{s[0], s[1], s[2]}; }

static int launch_particles_fluvial(float* waterFlux, float* massFlux, float* velocityFlux,
                                    float* albedoFlux, soil_rng* rng, int64_t N,
                                    const float* layers, const float* waterSource,
                                    const float* waterHeight, const float* velocity,
                                    const float* albedoSource, const Dom& d, Scale3 s,
                                    const Param& p, hipStream_t st) {
  if (N <= 0) return SOIL_OK;
  k_particles_fluvial<<<blocks_for(N, kPBlock), kPBlock, 0, st>>>(
      waterFlux, massFlux, velocityFlux, albedoFlux, rng, N,
      reinterpret_cast<const float2*>(layers), waterSource, waterHeight,
      reinterpret_cast<const float2*>(velocity), albedoSource, d, s, p);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

static int launch_particles_debris(float* massFlux, float* velocityFlux, float* albedoFlux,
                                   soil_rng* rng, int64_t N, const float* layers,
                                   const float* velocity, const float* albedoSource, const Dom& d,
                                   Scale3 s, const Param& p, hipStream_t st) {
  if (N <= 0) return SOIL_OK;
  k_particles_debris<<<blocks_for(N, kPBlock), kPBlock, 0, st>>>(
      massFlux, velocityFlux, albedoFlux, rng, N, reinterpret_cast<const float2*>(layers),
      reinterpret_cast<const float2*>(velocity), albedoSource, d, s, p);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

}  // namespace soil

using namespace soil;

extern "C" {

int64_t soil_ghost_rows(const soil_param* param) {
  const double travel = 1.41421356237309515 * static_cast<double>(param ? param->maxage : 512);
  return static_cast<int64_t>(std::ceil(travel)) + 2;
}

int soil_transport_fluvial(const float* layers, const float* rainfall, float* waterHeight,
                           float* waterFlux, float* mass, float* massFlux, float* velocity,
                           float* velocityFlux, const float* albedo_bedrock, float* albedoFlux,
                           const float* albedoSource, soil_rng* rng, int64_t N, int64_t H,
                           int64_t W, const float scale[3], const soil_param* param,
                           void* stream) {
  (void)albedo_bedrock;  // accepted and unused, erosion.cu:198
  SOIL_DEVICE();
  SOIL_REQUIRE(layers && rainfall && waterHeight && waterFlux && mass && massFlux && velocity &&
                   velocityFlux && scale && param,
               "transport_fluvial: null tensor");
  SOIL_REQUIRE((albedoFlux == nullptr) == (albedoSource == nullptr),
               "transport_fluvial: pass both albedoFlux and albedoSource or neither");
  SOIL_REQUIRE(H > 0 && W > 0 && N >= 0 && (N == 0 || rng), "transport_fluvial: bad sizes");
  const Dom d = full_domain(H, W);
  const Scale3 s = s3p(scale);
  int rc = launch_particles_fluvial(waterFlux, massFlux, velocityFlux, albedoFlux, rng, N, layers,
                                    rainfall, waterHeight, velocity, albedoSource, d, s, *param,
                                    as_stream(stream));  // erosion.cu:209
  if (rc != SOIL_OK) return rc;
  return launch_normalize_fluvial(waterFlux, massFlux, velocityFlux, albedoFlux, layers, rainfall,
                                  waterHeight, mass, velocity, albedoSource, d, s, *param,
                                  as_stream(stream));  // erosion.cu:224
}

int soil_transport_debris(const float* layers, float* velocity, float* velocityFlux, float* mass,
                          float* massFlux, const float* albedo_bedrock, float* albedoFlux,
                          const float* albedoSource, soil_rng* rng, int64_t N, int64_t H,
                          int64_t W, const float scale[3], const soil_param* param,
                          void* stream) {
  (void)albedo_bedrock;  // accepted and unused, erosion.cu:401
  SOIL_DEVICE();
  SOIL_REQUIRE(layers && velocity && velocityFlux && mass && massFlux && scale && param,
               "transport_debris: null tensor");
  SOIL_REQUIRE((albedoFlux == nullptr) == (albedoSource == nullptr),
               "transport_debris: pass both albedoFlux and albedoSource or neither");
  SOIL_REQUIRE(H > 0 && W > 0 && N >= 0 && (N == 0 || rng), "transport_debris: bad sizes");
  const Dom d = full_domain(H, W);
  const Scale3 s = s3p(scale);
  int rc = launch_particles_debris(massFlux, velocityFlux, albedoFlux, rng, N, layers, velocity,
                                   albedoSource, d, s, *param, as_stream(stream));  // :412
  if (rc != SOIL_OK) return rc;
  return launch_normalize_debris(massFlux, velocityFlux, albedoFlux, layers, mass, velocity,
                                 albedoSource, d, s, *param, as_stream(stream));  // :424
}

int soil_particles_fluvial_slab(float* waterFlux, float* massFlux, float* velocityFlux,
                                float* albedoFlux, soil_rng* rng, int64_t N, const float* layers,
                                const float* rainfall, const float* waterHeight,
                                const float* velocity, const float* albedoSource,
                                const soil_domain* dom, const float scale[3],
                                const soil_param* param, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(waterFlux && massFlux && velocityFlux && layers && rainfall && waterHeight &&
                   velocity && dom && scale && param,
               "particles_fluvial_slab: null argument");
  SOIL_REQUIRE((albedoFlux == nullptr) == (albedoSource == nullptr),
               "particles_fluvial_slab: pass both albedo planes or neither");
  SOIL_REQUIRE(N >= 0 && (N == 0 || rng), "particles_fluvial_slab: bad particle count");
  const Dom d = to_dom(dom);
  int rc = check_domain(d);
  if (rc != SOIL_OK) return rc;
  return launch_particles_fluvial(waterFlux, massFlux, velocityFlux, albedoFlux, rng, N, layers,
                                  rainfall, waterHeight, velocity, albedoSource, d, s3p(scale),
                                  *param, as_stream(stream));
}

int soil_particles_debris_slab(float* massFlux, float* velocityFlux, float* albedoFlux,
                               soil_rng* rng, int64_t N, const float* layers,
                               const float* velocity, const float* albedoSource,
                               const soil_domain* dom, const float scale[3],
                               const soil_param* param, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(massFlux && velocityFlux && layers && velocity && dom && scale && param,
               "particles_debris_slab: null argument");
  SOIL_REQUIRE((albedoFlux == nullptr) == (albedoSource == nullptr),
               "particles_debris_slab: pass both albedo planes or neither");
  SOIL_REQUIRE(N >= 0 && (N == 0 || rng), "particles_debris_slab: bad particle count");
  const Dom d = to_dom(dom);
  int rc = check_domain(d);
  if (rc != SOIL_OK) return rc;
  return launch_particles_debris(massFlux, velocityFlux, albedoFlux, rng, N, layers, velocity,
                                 albedoSource, d, s3p(scale), *param, as_stream(stream));
}

}  // extern "C"
