// erosion_particles.hip — Monte-Carlo transport half of the erosion model:
//   __transport_fluvial erosion.cu:29-141, __transport_debris :245-351, and the
//   host wrappers soil::transport_fluvial :189-239 / soil::transport_debris :395-436.
//
// One lane integrates one streamline; the flux planes are accumulated with
// hardware fp32 atomics (global_atomic_add_f32; built with -munsafe-fp-atomics).
//
// The reference launches thread n on particle n and lets every step gather a
// 5-point float2 stencil at a random place: on MI355X that is one HBM sector
// per gather (measured, profiles/r01_first).  Two MI355X-side changes, neither
// of which alters a single trajectory or deposit:
//   * STAGED GATHERS — a streaming pre-pass evaluates __glocal once per cell and
//     packs {grad.x, grad.y, vel.x, vel.y} into one float4 plane, so a step
//     costs one 16-byte gather (+4 bytes of waterHeight for fluvial) instead of
//     seven scattered loads;
//   * SPATIAL ORDER — particles are bucketed by the 16x16-cell tile of their
//     spawn point (count / scan / scatter) and traced in tile order, and the
//     work-groups of one XCD walk a contiguous band of tiles, so the lanes of a
//     wave and the waves of an XCD share cache lines for most of their walk.
// The reference's launch shape (thread n = particle n, direct stencil gathers)
// is kept as `direct` mode for ablation (soil_set_particle_mode).
#include "particles_common.hpp"

namespace soil {

constexpr int kPBlock = 256;
constexpr int kTile = 16;  // spawn-order bucket edge, in cells

int launch_normalize_fluvial(const float* waterFlux, const float* massFlux,
                             const float* velocityFlux, float* albedoFlux, const float* layers,
                             const float* waterSource, float* waterHeight, float* mass,
                             float* velocity, const float* albedoSource, const Dom& d, Scale3 s,
                             const Param& p, hipStream_t st);
int launch_normalize_debris(const float* massFlux, const float* velocityFlux, float* albedoFlux,
                            const float* layers, float* mass, float* velocity,
                            const float* albedoSource, const Dom& d, Scale3 s, const Param& p,
                            hipStream_t st);

static int g_particle_mode = 0;  // 0 auto, 1 direct, 2 staged, 3 tiled
// Arithmetic of the particle step in the tiled shape: 0 exact (IEEE quotients and square root: the
// oracle's walks step for step), 1 fast (v_rcp_f32 / v_sqrt_f32, erosion_particles_tiled.hip:
// step_geom_fast; statistical parity).  SOIL_PARTICLE_DIV=fast in the environment sets the default.
static int g_particle_arith = [] {
  const char* e = std::getenv("SOIL_PARTICLE_DIV");
  return (e && (e[0] == 'f' || e[0] == 'F' || e[0] == '1')) ? 1 : 0;
}();
bool particle_arith_fast() { return g_particle_arith == 1; }
// Spent debris walkers (erosion_particles_tiled.hip: debris_spent): 1 (default) they end their walks, 0 every
// walker is walked to the end as in the reference, 2 they are marked, walked on and watched (tests).
// SOIL_DEBRIS_RETIRE in the environment sets the default of the process.
static int g_debris_retire = [] {
  const char* e = std::getenv("SOIL_DEBRIS_RETIRE");
  const int v = e ? std::atoi(e) : 1;
  return (v >= 0 && v <= 2) ? v : 1;
}();
int debris_retire_mode() { return g_debris_retire; }

// ---- where a step gets grad(cell) and velocity(cell) from -------------------

struct DirectFields {  // the reference's access pattern
  const float2* __restrict__ layers;
  const float2* __restrict__ velocity;
  Dom d;
  Scale3 s;
  float exitSlope;
  __device__ __forceinline__ void at(int64_t cx, int64_t cy, int64_t l, float2& grad,
                                     float2& vel) const {
    grad = glocal(layers, d, s, cx, cy, exitSlope);
    vel = velocity[l];
  }
};

struct PackedFields {  // one 16-byte gather per step
  const float4* __restrict__ p4;
  __device__ __forceinline__ void at(int64_t, int64_t, int64_t l, float2& grad,
                                     float2& vel) const {
    const float4 v = p4[l];
    grad = make_float2(v.x, v.y);
    vel = make_float2(v.z, v.w);
  }
};

struct FluvialPlanes {
  float* __restrict__ waterFlux;
  float* __restrict__ massFlux;
  float* __restrict__ velocityFlux;
  float* __restrict__ albedoFlux;
  const float* __restrict__ waterSource;
  const float* __restrict__ waterHeight;
  const float* __restrict__ albedoSource;
  float* __restrict__ remote0;
  unsigned long long* __restrict__ steps;  // step_counter() of the device
};

// __transport_fluvial, erosion.cu:49-139, from the spawn position on
template <class Fields>
__device__ __forceinline__ void trace_fluvial(const Fields& F, const FluvialPlanes& P, float px,
                                              float py, int64_t N, const Dom& d, Scale3 s,
                                              const Param& param) {
  const float A = s.x * s.y;                                   // :50
  const float Lx = s.x, Ly = s.y;                              // :51
  const float Pr = 1.0f / (A * static_cast<float>(d.H * d.W)); // :53
  const float Q = 1.0f / (Pr * static_cast<float>(N));         // :54
  const float eps = 1E-12f;                                    // :55
  const int64_t W = d.W;
  const int64_t base = d.x0 * W;
  int64_t ind = cell_of(px) * W + cell_of(py);  // :60

  const float rho_w = param.densityWater;                 // :63
  const float tau = param.bedShearWater;                  // :65
  const float nu = param.viscosityWater;                  // :66
  const float g = param.gravity;                          // :67
  const float ks = param.suspensionRateFluvial / 64.0f;   // :68
  const float kd = param.depositionRateFluvial * 1.33f;   // :69
  const float fD = param.frictionFactor / 8.0f;           // :70
  const float alpha = param.fluvialExponent;              // :71
  const float R = param.rainfall;                         // :72

  float2 vel, grad;
  F.at(cell_of(px), cell_of(py), ind - base, grad, vel);    // :75-76
  float spx = -(g * grad.x) + nu * vel.x + param.force[0];  // :77
  float spy = -(g * grad.y) + nu * vel.y + param.force[1];
  {
    const float den = sqrtf(length2(Lx * spx, Ly * spy));  // :78
    spx = spx / den;
    spy = spy / den;
  }
  if (length2(spx, spy) < eps) return;  // :79-80

  const float v = length2(vel.x, vel.y);                              // :83
  const float shear = 0.125f * fD * rho_w * v * v;                    // :84
  const float power = powf_(shear * length2(grad.x, grad.y), alpha);  // :85
  const float source_m = Q * ks * power;                              // :88
  const float source_w = Q * R * P.waterSource[ind - base];           // :89
  const float source_vx = Q * (-(g * grad.x) + nu * vel.x);           // :90
  const float source_vy = Q * (-(g * grad.y) + nu * vel.y);
  float source_a[3] = {0.0f, 0.0f, 0.0f};
  if (P.albedoSource)  // :91
    for (int c = 0; c < 3; ++c) source_a[c] = source_m * P.albedoSource[3 * (ind - base) + c];

  float att_w = 1.0f, att_m = 1.0f, att_v = 1.0f;  // :94-96
  const float lenL = length2(Lx, Ly);
  uint64_t iter = 0;
  uint32_t nsteps = 0;
  while (!oob(d, px, py) && ++iter < param.maxage) {  // :100
    const int64_t cx = cell_of(px), cy = cell_of(py);
    if (slab_escape(d, cx)) {
      // a NaN walker's single deposit belongs to global cell (0,0); when that
      // cell lives on another rank it is parked in `remote0` for its owner
      if (px != px && P.remote0 && ind != 0) {
        atomicAdd(&P.remote0[0], att_w * source_w);
        atomicAdd(&P.remote0[1], att_m * source_m);
        atomicAdd(&P.remote0[2], att_v * source_vx);
        atomicAdd(&P.remote0[3], att_v * source_vy);
      }
      break;
    }
    ++nsteps;
    const int64_t nind = cx * W + cy;  // :103
    if (nind != ind) {                 // :104-113
      ind = nind;
      const int64_t l = ind - base;
      atomicAdd(&P.waterFlux[l], att_w * source_w);
      atomicAdd(&P.massFlux[l], att_m * source_m);
      atomicAdd(&P.velocityFlux[2 * l], att_v * source_vx);
      atomicAdd(&P.velocityFlux[2 * l + 1], att_v * source_vy);
      if (P.albedoFlux)
        for (int c = 0; c < 3; ++c) atomicAdd(&P.albedoFlux[3 * l + c], att_m * source_a[c]);
    }
    const float v_norm = length2(spx, spy);            // :116
    const float ux = spx / v_norm, uy = spy / v_norm;  // :117
    const float v_step = stepsize(px, py, ux, uy);     // :118
    const float dL = v_step * lenL;                    // :119
    const float ds = dL / v_norm;                      // :120
    if (v_norm < eps) break;                           // :121-122

    const int64_t l = ind - base;
    float2 vc;
    F.at(cx, cy, l, grad, vc);                                    // :125
    const float ax = -(g * grad.x) + nu * vc.x + param.force[0];  // :126
    const float ay = -(g * grad.y) + nu * vc.y + param.force[1];
    const float w0 = 1.0f / (1.0f + dL * (tau + nu));  // :127
    const float w1 = dL / (1.0f + dL * (tau + nu));
    spx = w0 * spx + w1 * ax;
    spy = w0 * spy + w1 * ay;

    const float decay_m = kd;                                      // :130
    const float decay_w = param.evapRate;                          // :131
    const float decay_v = 0.125f * fD / (eps + P.waterHeight[l]);  // :132
    att_m = att_m * att_exp(-ds * decay_m);                          // :134
    att_w = att_w * att_exp(-ds * decay_w);                          // :135
    att_v = att_v * att_exp(-dL * decay_v);                        // :136
    px += v_step * ux;                                             // :137
    py += v_step * uy;
  }
  atomicAdd(P.steps, static_cast<unsigned long long>(nsteps));  // one atomic per wave
}

struct DebrisPlanes {
  float* __restrict__ massFlux;
  float* __restrict__ velocityFlux;
  float* __restrict__ albedoFlux;
  const float* __restrict__ albedoSource;
  float* __restrict__ remote0;
  unsigned long long* __restrict__ steps;  // step_counter() of the device
};

// __transport_debris, erosion.cu:262-349, from the spawn position on
template <class Fields>
__device__ __forceinline__ void trace_debris(const Fields& F, const DebrisPlanes& P, float px,
                                             float py, int64_t N, const Dom& d, Scale3 s,
                                             const Param& param) {
  const float A = s.x * s.y;                                    // :263
  const float Lx = s.x, Ly = s.y;                               // :264
  const float Pr = 1.0f / (A * static_cast<float>(d.H * d.W));  // :266
  const float Q = 1.0f / (Pr * static_cast<float>(N));          // :267
  const float eps = 1E-12f;                                     // :268
  const int64_t W = d.W;
  const int64_t base = d.x0 * W;
  int64_t ind = cell_of(px) * W + cell_of(py);  // :273

  const float theta = param.critSlopeBedrock;    // :276
  const float nu = param.viscosityDebris;        // :277
  const float tau = param.bedShearDebris;        // :278
  const float g = param.gravity;                 // :279
  const float kl = param.landslideRateDebris;    // :280
  const float kdd = param.depositionRateDebris;  // :281
  const float kds = param.suspensionRateDebris;  // :282
  const float tau_y = param.yieldStress;         // :283

  float2 vel, grad;
  F.at(cell_of(px), cell_of(py), ind - base, grad, vel);  // :286-287
  float spx = -(g * grad.x) + nu * vel.x;                 // :288
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
  if (P.albedoSource)  // :299
    for (int c = 0; c < 3; ++c) source_a[c] = source_d * P.albedoSource[3 * (ind - base) + c];

  float att_d = 1.0f, att_v = 1.0f;  // :301-302
  const float lenL = length2(Lx, Ly);
  uint64_t iter = 0;
  uint32_t nsteps = 0;
  while (!oob(d, px, py) && ++iter < param.maxage) {  // :306
    const int64_t cx = cell_of(px), cy = cell_of(py);
    if (slab_escape(d, cx)) {
      if (px != px && P.remote0 && ind != 0) {  // NaN walker: see trace_fluvial
        atomicAdd(&P.remote0[4], att_d * source_d);
        atomicAdd(&P.remote0[5], att_v * source_vx);
        atomicAdd(&P.remote0[6], att_v * source_vy);
      }
      break;
    }
    ++nsteps;
    const int64_t nind = cx * W + cy;  // :309
    if (nind != ind) {                 // :310-318
      ind = nind;
      const int64_t l = ind - base;
      atomicAdd(&P.massFlux[l], att_d * source_d);
      atomicAdd(&P.velocityFlux[2 * l], att_v * source_vx);
      atomicAdd(&P.velocityFlux[2 * l + 1], att_v * source_vy);
      if (P.albedoFlux)
        for (int c = 0; c < 3; ++c) atomicAdd(&P.albedoFlux[3 * l + c], att_d * source_a[c]);
    }
    const float v_norm = length2(spx, spy);            // :321
    const float ux = spx / v_norm, uy = spy / v_norm;  // :322
    const float v_step = stepsize(px, py, ux, uy);     // :323
    const float dL = v_step * lenL;                    // :324
    const float ds = dL / v_norm;                      // :325
    if (v_norm < eps) break;                           // :326-327

    const int64_t l = ind - base;
    float2 vc;
    F.at(cx, cy, l, grad, vc);                          // :330
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
    att_v = att_v * att_exp(-dL * decay_v);                               // :346
    px += v_step * ux;                                                    // :347
    py += v_step * uy;
  }
  atomicAdd(P.steps, static_cast<unsigned long long>(nsteps));
}

// ---- direct mode: thread n = particle n ---------------------------------------

__global__ void __launch_bounds__(kPBlock)
    k_fluvial_direct(FluvialPlanes P, soil_rng* __restrict__ rng, int64_t N, DirectFields F,
                     Param param) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kPBlock + threadIdx.x;
  if (n >= N) return;
  const float2 pos = spawn_position(rng, n, F.d);
  if (!owns_spawn(F.d, pos.x)) return;
  trace_fluvial(F, P, pos.x, pos.y, N, F.d, F.s, param);
}

__global__ void __launch_bounds__(kPBlock)
    k_debris_direct(DebrisPlanes P, soil_rng* __restrict__ rng, int64_t N, DirectFields F,
                    Param param) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kPBlock + threadIdx.x;
  if (n >= N) return;
  const float2 pos = spawn_position(rng, n, F.d);
  if (!owns_spawn(F.d, pos.x)) return;
  trace_debris(F, P, pos.x, pos.y, N, F.d, F.s, param);
}

// ---- staged mode ------------------------------------------------------------------

// pre-pass: p4[cell] = {__glocal(cell), velocity[cell]} for every row with a full stencil
__global__ void __launch_bounds__(kPBlock)
    k_pack_fields(float4* __restrict__ p4, const float2* __restrict__ layers,
                  const float2* __restrict__ velocity, Dom d, Scale3 s, float exitSlope,
                  int64_t row_lo, int64_t cells) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kPBlock + threadIdx.x;
  if (t >= cells) return;
  const int64_t lx = row_lo + t / d.W, y = t % d.W;
  const int64_t l = lx * d.W + y;
  const float2 g = glocal(layers, d, s, d.x0 + lx, y, exitSlope);
  const float2 v = velocity[l];
  p4[l] = make_float4(g.x, g.y, v.x, v.y);
}

__device__ __forceinline__ int64_t tile_of(const Dom& d, float px, float py, int64_t tiles_w) {
  const int64_t lx = cell_of(px) - d.x0, cy = cell_of(py);
  return (lx / kTile) * tiles_w + cy / kTile;
}

// pass 1: draw the spawn points (advancing every particle's stream) and count per tile
__global__ void __launch_bounds__(kPBlock)
    k_spawn_count(float2* __restrict__ spawn, uint32_t* __restrict__ count,
                  soil_rng* __restrict__ rng, int64_t N, Dom d, int64_t tiles_w) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kPBlock + threadIdx.x;
  if (n >= N) return;
  const float2 pos = spawn_position(rng, n, d);
  spawn[n] = pos;
  if (owns_spawn(d, pos.x)) atomicAdd(&count[tile_of(d, pos.x, pos.y, tiles_w)], 1u);
}

// pass 2: exclusive scan of the tile counts (one work-group; tiles <= a few 1e5)
__global__ void __launch_bounds__(1024)
    k_tile_scan(uint32_t* start, const uint32_t* count, int64_t tiles) {
  __shared__ uint32_t part[1024];
  const int tid = threadIdx.x;
  const int64_t chunk = (tiles + 1023) / 1024;
  const int64_t b = tid * chunk, e = (b + chunk < tiles) ? b + chunk : tiles;
  uint32_t sum = 0;
  for (int64_t i = b; i < e; ++i) sum += count[i];
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
    const uint32_t v = (tid >= off) ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum;
  for (int64_t i = b; i < e; ++i) {
    start[i] = run;
    run += count[i];
  }
  if (tid == 1023) start[tiles] = part[1023];  // total number of owned particles
}

// pass 3: drop every owned spawn point into its tile's range
__global__ void __launch_bounds__(kPBlock)
    k_spawn_scatter(float2* __restrict__ sorted, uint32_t* __restrict__ fill,
                    const uint32_t* __restrict__ start, const float2* __restrict__ spawn,
                    int64_t N, Dom d, int64_t tiles_w) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kPBlock + threadIdx.x;
  if (n >= N) return;
  const float2 pos = spawn[n];
  if (!owns_spawn(d, pos.x)) return;
  const int64_t tile = tile_of(d, pos.x, pos.y, tiles_w);
  sorted[start[tile] + atomicAdd(&fill[tile], 1u)] = pos;
}

// work-group -> slot in the sorted order: block b runs on XCD b % 8; give every
// XCD one contiguous eighth of the tile sequence
__device__ __forceinline__ int64_t sorted_slot(const uint32_t* __restrict__ total_ptr) {
  const int64_t total = *total_ptr;
  const int64_t nb = (total + kPBlock - 1) / kPBlock;
  const int64_t per = (nb + 7) / 8;
  const int64_t b = blockIdx.x;
  const int64_t slot = b / 8;
  if (slot >= per) return -1;
  const int64_t blk = (b % 8) * per + slot;
  const int64_t t = blk * kPBlock + threadIdx.x;
  return (blk < nb && t < total) ? t : -1;
}

__global__ void __launch_bounds__(kPBlock)
    k_fluvial_sorted(FluvialPlanes P, const float2* __restrict__ sorted,
                     const uint32_t* __restrict__ total, int64_t N, PackedFields F, Dom d,
                     Scale3 s, Param param) {
  const int64_t t = sorted_slot(total);
  if (t < 0) return;
  const float2 pos = sorted[t];
  trace_fluvial(F, P, pos.x, pos.y, N, d, s, param);
}

__global__ void __launch_bounds__(kPBlock)
    k_debris_sorted(DebrisPlanes P, const float2* __restrict__ sorted,
                    const uint32_t* __restrict__ total, int64_t N, PackedFields F, Dom d, Scale3 s,
                    Param param) {
  const int64_t t = sorted_slot(total);
  if (t < 0) return;
  const float2 pos = sorted[t];
  trace_debris(F, P, pos.x, pos.y, N, d, s, param);
}

static Scale3 s3p(const float* s) { return Scale3{s[0], s[1], s[2]}; }

static bool use_staged(int64_t N) {
  if (g_particle_mode == 1) return false;
  if (g_particle_mode == 2) return true;
  return N >= 1024;
}
// the tiled shape keeps the last cell index in 32 bits (and the slot indices of its
// queues): grids and launches beyond 2^31 stay on the staged shape
static bool use_tiled(int64_t N, const Dom& d) {
  if (d.H * d.W > 0x7fffffffll || N > 0x7fffffffll) return false;
  if (d.H >= (1 << 24) || d.W >= (1 << 24)) return false;  // its cell index is a 24-bit multiply-add
  if (g_particle_mode == 3) return true;
  // measured crossover against the staged shape: 512^2 (N = 32768) 2.15 vs 2.69 ms per step,
  // 640^2 (N = 51200) 2.90 vs 2.76
  return g_particle_mode == 0 && N >= 45000;
}

bool use_tiled_launch(int64_t N, const Dom& d) { return use_tiled(N, d); }

// The slab launches on explicit streams (particles_common.hpp, Streams): the tiled shape takes uniform
// streams as they are; the small-N shapes read a tensor, which is seeded here when the streams are
// uniform (the caller hands one over for that).
static int materialise(const Streams& st, int64_t N, hipStream_t s) {
  if (!st.uniform) return SOIL_OK;
  if (!st.rng) return fail(SOIL_ERR_INVALID_ARGUMENT, "uniform streams on a small launch need a tensor to seed");
  return soil_rng_seed(st.rng, N, st.seed, st.offset, s);
}
int particles_fluvial_streams(const soil_erosion_planes& P, Streams rng, int64_t N, float* remote0,
                              const Dom& d, Scale3 s, const Param& p, hipStream_t st) {
  if (N <= 0) return SOIL_OK;
  if (use_tiled(N, d))
    return launch_fluvial_tiled(P.waterFlux, P.massFlux, P.velocityFlux, nullptr, rng, N, P.layers, P.rainfall,
                                P.waterHeight, P.velocity, nullptr, remote0, d, s, p, st);
  if (int rc = materialise(rng, N, st); rc != SOIL_OK) return rc;
  return soil_particles_fluvial_slab(P.waterFlux, P.massFlux, P.velocityFlux, nullptr, rng.rng, N, P.layers,
                                     P.rainfall, P.waterHeight, P.velocity, nullptr, remote0,
                                     reinterpret_cast<const soil_domain*>(&d), &s.x, &p, st);
}
int particles_debris_streams(const soil_erosion_planes& P, Streams rng, int64_t N, float* remote0,
                             const Dom& d, Scale3 s, const Param& p, hipStream_t st) {
  if (N <= 0) return SOIL_OK;
  if (use_tiled(N, d))
    return launch_debris_tiled(P.debrisFlux, P.debrisVelocityFlux, nullptr, rng, N, P.layers, P.debrisVelocity,
                               nullptr, remote0, d, s, p, st);
  if (int rc = materialise(rng, N, st); rc != SOIL_OK) return rc;
  return soil_particles_debris_slab(P.debrisFlux, P.debrisVelocityFlux, nullptr, rng.rng, N, P.layers,
                                    P.debrisVelocity, nullptr, remote0, reinterpret_cast<const soil_domain*>(&d),
                                    &s.x, &p, st);
}
int particles_pair_streams(const soil_erosion_planes& P, Streams rf, Streams rd, int64_t N, float* remote0,
                           const Dom& d, Scale3 s, const Param& p, hipStream_t st, bool overwrite) {
  if (N > 0 && use_tiled(N, d)) return launch_pair_tiled(P, rf, rd, N, remote0, d, s, p, st, overwrite);
  if (overwrite) {  // a launch that cannot store its first round clears the planes it adds to
    const size_t b = sizeof(float) * static_cast<size_t>(d.rows) * static_cast<size_t>(d.W);
    SOIL_HIP(hipMemsetAsync(P.waterFlux, 0, b, st));
    SOIL_HIP(hipMemsetAsync(P.massFlux, 0, b, st));
    SOIL_HIP(hipMemsetAsync(P.velocityFlux, 0, 2 * b, st));
    SOIL_HIP(hipMemsetAsync(P.debrisFlux, 0, b, st));
    SOIL_HIP(hipMemsetAsync(P.debrisVelocityFlux, 0, 2 * b, st));
  }
  if (N <= 0) return SOIL_OK;
  if (int rc = materialise(rf, N, st); rc != SOIL_OK) return rc;
  if (int rc = materialise(rd, N, st); rc != SOIL_OK) return rc;
  return soil_particles_pair_slab(&P, rf.rng, rd.rng, N, remote0, reinterpret_cast<const soil_domain*>(&d), &s.x,
                                  &p, st);
}

// Shared staging: pack the fields, bucket the spawn points.  Returns device
// pointers into the per-device workspace (valid until the next staged call).
struct Staged {
  float4* p4;
  float2* sorted;
  uint32_t* total;
};

static int stage(Staged* out, soil_rng* rng, int64_t N, const float* layers,
                 const float* velocity, const Dom& d, Scale3 s, const Param& p, hipStream_t st) {
  const int64_t tiles_w = (d.W + kTile - 1) / kTile, tiles_h = (d.rows + kTile - 1) / kTile;
  const int64_t tiles = tiles_w * tiles_h;
  auto align = [](size_t b) { return (b + 255) & ~static_cast<size_t>(255); };
  const size_t b_p4 = align(sizeof(float4) * d.rows * d.W);
  const size_t b_pos = align(sizeof(float2) * N);
  const size_t b_cnt = align(sizeof(uint32_t) * (tiles + 1));
  void* base = nullptr;
  int rc = workspace_get(1, b_p4 + 2 * b_pos + 3 * b_cnt, &base);
  if (rc != SOIL_OK) return rc;
  char* w = static_cast<char*>(base);
  out->p4 = reinterpret_cast<float4*>(w);      w += b_p4;
  float2* spawn = reinterpret_cast<float2*>(w); w += b_pos;
  out->sorted = reinterpret_cast<float2*>(w);  w += b_pos;
  uint32_t* count = reinterpret_cast<uint32_t*>(w); w += b_cnt;
  uint32_t* fill = reinterpret_cast<uint32_t*>(w);  w += b_cnt;
  uint32_t* start = reinterpret_cast<uint32_t*>(w);
  out->total = start + tiles;

  const int64_t lo = stencil_lo(d), hi = stencil_hi(d);
  const int64_t cells = (hi - lo + 1) * d.W;
  if (cells > 0)
    k_pack_fields<<<blocks_for(cells, kPBlock), kPBlock, 0, st>>>(
        out->p4, reinterpret_cast<const float2*>(layers),
        reinterpret_cast<const float2*>(velocity), d, s, p.exitSlope, lo, cells);
  SOIL_HIP(hipMemsetAsync(count, 0, 2 * b_cnt, st));  // count and fill are adjacent
  k_spawn_count<<<blocks_for(N, kPBlock), kPBlock, 0, st>>>(spawn, count, rng, N, d, tiles_w);
  k_tile_scan<<<1, 1024, 0, st>>>(start, count, tiles);
  k_spawn_scatter<<<blocks_for(N, kPBlock), kPBlock, 0, st>>>(out->sorted, fill, start, spawn, N,
                                                              d, tiles_w);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

static int launch_particles_fluvial(float* waterFlux, float* massFlux, float* velocityFlux,
                                    float* albedoFlux, soil_rng* rng, int64_t N,
                                    const float* layers, const float* waterSource,
                                    const float* waterHeight, const float* velocity,
                                    const float* albedoSource, float* remote0, const Dom& d,
                                    Scale3 s, const Param& p, hipStream_t st) {
  if (N <= 0) return SOIL_OK;
  if (use_tiled(N, d))
    return launch_fluvial_tiled(waterFlux, massFlux, velocityFlux, albedoFlux, streams_of(rng), N, layers,
                                waterSource, waterHeight, velocity, albedoSource, remote0, d, s, p,
                                st);
  unsigned long long* steps = nullptr;
  if (int rc = step_counter(&steps); rc != SOIL_OK) return rc;
  const FluvialPlanes P{waterFlux,   massFlux,    velocityFlux, albedoFlux, waterSource,
                        waterHeight, albedoSource, remote0,      steps};
  if (use_staged(N)) {
    Staged sg;
    int rc = stage(&sg, rng, N, layers, velocity, d, s, p, st);
    if (rc != SOIL_OK) return rc;
    k_fluvial_sorted<<<blocks_for(N, kPBlock) + 8, kPBlock, 0, st>>>(
        P, sg.sorted, sg.total, N, PackedFields{sg.p4}, d, s, p);
  } else {
    const DirectFields F{reinterpret_cast<const float2*>(layers),
                         reinterpret_cast<const float2*>(velocity), d, s, p.exitSlope};
    k_fluvial_direct<<<blocks_for(N, kPBlock), kPBlock, 0, st>>>(P, rng, N, F, p);
  }
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

static int launch_particles_debris(float* massFlux, float* velocityFlux, float* albedoFlux,
                                   soil_rng* rng, int64_t N, const float* layers,
                                   const float* velocity, const float* albedoSource,
                                   float* remote0, const Dom& d, Scale3 s, const Param& p,
                                   hipStream_t st) {
  if (N <= 0) return SOIL_OK;
  if (use_tiled(N, d))
    return launch_debris_tiled(massFlux, velocityFlux, albedoFlux, streams_of(rng), N, layers, velocity,
                               albedoSource, remote0, d, s, p, st);
  unsigned long long* steps = nullptr;
  if (int rc = step_counter(&steps); rc != SOIL_OK) return rc;
  const DebrisPlanes P{massFlux, velocityFlux, albedoFlux, albedoSource, remote0, steps};
  if (use_staged(N)) {
    Staged sg;
    int rc = stage(&sg, rng, N, layers, velocity, d, s, p, st);
    if (rc != SOIL_OK) return rc;
    k_debris_sorted<<<blocks_for(N, kPBlock) + 8, kPBlock, 0, st>>>(
        P, sg.sorted, sg.total, N, PackedFields{sg.p4}, d, s, p);
  } else {
    const DirectFields F{reinterpret_cast<const float2*>(layers),
                         reinterpret_cast<const float2*>(velocity), d, s, p.exitSlope};
    k_debris_direct<<<blocks_for(N, kPBlock), kPBlock, 0, st>>>(P, rng, N, F, p);
  }
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

}  // namespace soil

using namespace soil;

extern "C" {

int soil_set_particle_mode(int mode) {
  SOIL_REQUIRE(mode >= 0 && mode <= 3, "particle mode: 0 auto, 1 direct, 2 staged, 3 tiled");
  g_particle_mode = mode;
  return SOIL_OK;
}

int soil_set_particle_arith(int mode) {
  SOIL_REQUIRE(mode == 0 || mode == 1, "particle arithmetic: 0 exact, 1 fast");
  g_particle_arith = mode;
  return SOIL_OK;
}
int soil_get_particle_arith(void) { return g_particle_arith; }

int soil_set_debris_retire(int mode) {
  SOIL_REQUIRE(mode >= 0 && mode <= 2, "debris retirement: 0 off, 1 on, 2 watched");
  g_debris_retire = mode;
  return SOIL_OK;
}
int soil_get_debris_retire(void) { return g_debris_retire; }

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
                                    rainfall, waterHeight, velocity, albedoSource, nullptr, d, s,
                                    *param, as_stream(stream));  // erosion.cu:209
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
                                   albedoSource, nullptr, d, s, *param,
                                   as_stream(stream));  // :412
  if (rc != SOIL_OK) return rc;
  return launch_normalize_debris(massFlux, velocityFlux, albedoFlux, layers, mass, velocity,
                                 albedoSource, d, s, *param, as_stream(stream));  // :424
}

int soil_particles_fluvial_slab(float* waterFlux, float* massFlux, float* velocityFlux,
                                float* albedoFlux, soil_rng* rng, int64_t N, const float* layers,
                                const float* rainfall, const float* waterHeight,
                                const float* velocity, const float* albedoSource, float* remote0,
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
                                  rainfall, waterHeight, velocity, albedoSource, remote0, d,
                                  s3p(scale), *param, as_stream(stream));
}

int soil_particles_debris_slab(float* massFlux, float* velocityFlux, float* albedoFlux,
                               soil_rng* rng, int64_t N, const float* layers,
                               const float* velocity, const float* albedoSource, float* remote0,
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
                                 albedoSource, remote0, d, s3p(scale), *param,
                                 as_stream(stream));
}

int soil_particles_pair_slab(const soil_erosion_planes* planes, soil_rng* rng_fluvial,
                             soil_rng* rng_debris, int64_t N, float* remote0,
                             const soil_domain* dom, const float scale[3], const soil_param* param,
                             void* stream) {
  return soil_particles_pair_slab_ex(planes, rng_fluvial, rng_debris, N, remote0, dom, scale, param, 0, stream);
}

int soil_particles_pair_slab_ex(const soil_erosion_planes* planes, soil_rng* rng_fluvial,
                                soil_rng* rng_debris, int64_t N, float* remote0,
                                const soil_domain* dom, const float scale[3], const soil_param* param,
                                int flags, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(planes && dom && scale && param, "particles_pair_slab: null argument");
  const soil_erosion_planes& P = *planes;
  SOIL_REQUIRE(P.layers && P.rainfall && P.waterHeight && P.waterFlux && P.massFlux && P.velocity &&
                   P.velocityFlux && P.debrisFlux && P.debrisVelocity && P.debrisVelocityFlux,
               "particles_pair_slab: null plane");
  SOIL_REQUIRE(N >= 0 && (N == 0 || (rng_fluvial && rng_debris && rng_fluvial != rng_debris)),
               "particles_pair_slab: needs two distinct rng tensors");
  const Dom d = to_dom(dom);
  int rc = check_domain(d);
  if (rc != SOIL_OK) return rc;
  hipStream_t st = as_stream(stream);
  const bool overwrite = (flags & SOIL_FLUX_OVERWRITE) != 0;
  auto clear_flux = [&]() -> int {  // what a launch that cannot store its first round does instead
    const size_t b = sizeof(float) * static_cast<size_t>(d.rows) * static_cast<size_t>(d.W);
    SOIL_HIP(hipMemsetAsync(P.waterFlux, 0, b, st));
    SOIL_HIP(hipMemsetAsync(P.massFlux, 0, b, st));
    SOIL_HIP(hipMemsetAsync(P.velocityFlux, 0, 2 * b, st));
    SOIL_HIP(hipMemsetAsync(P.debrisFlux, 0, b, st));
    SOIL_HIP(hipMemsetAsync(P.debrisVelocityFlux, 0, 2 * b, st));
    return SOIL_OK;
  };
  if (N <= 0) return overwrite ? clear_flux() : SOIL_OK;
  const Scale3 s = s3p(scale);
  if (use_tiled(N, d))
    return launch_pair_tiled(P, streams_of(rng_fluvial), streams_of(rng_debris), N, remote0, d, s, *param, st, overwrite);
  if (overwrite)
    if (int rc2 = clear_flux(); rc2 != SOIL_OK) return rc2;
  rc = launch_particles_fluvial(P.waterFlux, P.massFlux, P.velocityFlux, nullptr, rng_fluvial, N,
                                P.layers, P.rainfall, P.waterHeight, P.velocity, nullptr, remote0, d,
                                s, *param, st);
  if (rc != SOIL_OK) return rc;
  return launch_particles_debris(P.debrisFlux, P.debrisVelocityFlux, nullptr, rng_debris, N, P.layers,
                                 P.debrisVelocity, nullptr, remote0, d, s, *param, st);
}

int soil_particle_steps(uint64_t* total, int reset, void* stream) {
  SOIL_REQUIRE(total != nullptr, "soil_particle_steps: null output");
  unsigned long long* counter = nullptr;
  if (int rc = step_counter(&counter); rc != SOIL_OK) return rc;
  unsigned long long v = 0;
  hipStream_t st = as_stream(stream);
  SOIL_HIP(hipMemcpyAsync(&v, counter, sizeof(v), hipMemcpyDeviceToHost, st));
  if (reset) SOIL_HIP(hipMemsetAsync(counter, 0, sizeof(v), st));
  SOIL_HIP(hipStreamSynchronize(st));
  *total = v;
  return SOIL_OK;
}

}  // extern "C"
