// erosion_cells.hip — per-cell (bandwidth-bound) half of the erosion model:
// the stand-alone ops of the reference API and the fused step kernel.
//
//   __normalize_fluvial erosion.cu:143-187   __normalize_debris :353-393
//   __transfer :453-574   __mass_creep :633-710   __layer_merge :733-745
//   __albedo_layer :759-791   __albedo_stratum :794-826   __albedo_discharge :857-875
#include <cstdlib>

#include "cell_math.hpp"
#include "window.hpp"

namespace soil {

constexpr int kBlock = 256;

// five-point neighbourhood of a layer plane for global cell (gx, y); neighbours
// outside the GLOBAL grid come back flagged
struct Nbhd {
  float2 l00, ln0, lp0, l0n, l0p;
  bool has_n0, has_p0, has_0n, has_0p;
};

__device__ __forceinline__ Nbhd load_nbhd(const float2* __restrict__ layers, const Dom& d,
                                          int64_t lx, int64_t y) {
  const int64_t gx = d.x0 + lx;
  const int64_t n = lx * d.W + y;
  Nbhd nb;
  nb.l00 = layers[n];
  nb.has_n0 = gx - 1 >= 0;
  nb.has_p0 = gx + 1 < d.H;
  nb.has_0n = y - 1 >= 0;
  nb.has_0p = y + 1 < d.W;
  nb.ln0 = nb.has_n0 ? layers[n - d.W] : nb.l00;
  nb.lp0 = nb.has_p0 ? layers[n + d.W] : nb.l00;
  nb.l0n = nb.has_0n ? layers[n - 1] : nb.l00;
  nb.l0p = nb.has_0p ? layers[n + 1] : nb.l00;
  return nb;
}

__device__ __forceinline__ float2 grad_of(const Nbhd& nb, Scale3 s, float exitSlope) {
  const float nan = __builtin_nanf("");
  const float h = nb.l00.x + nb.l00.y;
  const float hn0 = nb.has_n0 ? nb.ln0.x + nb.ln0.y : nan;
  const float hp0 = nb.has_p0 ? nb.lp0.x + nb.lp0.y : nan;
  const float h0n = nb.has_0n ? nb.l0n.x + nb.l0n.y : nan;
  const float h0p = nb.has_0p ? nb.l0p.x + nb.l0p.y : nan;
  return glocal_from_heights(h, hn0, hp0, h0n, h0p, s, exitSlope);
}

// ---- stand-alone kernels (one thread per cell of rows [r0, r1)) -------------

__global__ void __launch_bounds__(kBlock)
    k_normalize_fluvial(const float* __restrict__ waterFlux, const float* __restrict__ massFlux,
                        const float2* __restrict__ velocityFlux, float* __restrict__ albedoFlux,
                        const float2* __restrict__ layers, const float* __restrict__ waterSource,
                        float* __restrict__ waterHeight, float* __restrict__ mass,
                        float2* __restrict__ velocity, const float* __restrict__ albedoSource,
                        Dom d, Scale3 s, Param p) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= (d.r1 - d.r0) * d.W) return;
  const int64_t lx = d.r0 + t / d.W, y = t % d.W;
  const int64_t n = lx * d.W + y;
  const float2 grad = glocal(layers, d, s, d.x0 + lx, y, p.exitSlope);  // :168
  const float m = massFlux[n];
  const FluvialOut o =
      normalize_fluvial_cell(grad, waterFlux[n], m, velocityFlux[n], waterSource[n], s, p);
  waterHeight[n] = o.waterHeight;
  mass[n] = o.mass;
  velocity[n] = o.velocity;
  if (albedoFlux) normalize_albedo_cell(albedoFlux, albedoSource, n, m);
}

__global__ void __launch_bounds__(kBlock)
    k_normalize_debris(const float* __restrict__ massFlux, const float2* __restrict__ velocityFlux,
                       float* __restrict__ albedoFlux, const float2* __restrict__ layers,
                       float* __restrict__ mass, float2* __restrict__ velocity,
                       const float* __restrict__ albedoSource, Dom d, Scale3 s, Param p) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= (d.r1 - d.r0) * d.W) return;
  const int64_t lx = d.r0 + t / d.W, y = t % d.W;
  const int64_t n = lx * d.W + y;
  const float2 grad = glocal(layers, d, s, d.x0 + lx, y, p.exitSlope);  // :375
  const float m = massFlux[n];
  const DebrisOut o = normalize_debris_cell(grad, m, velocityFlux[n], s, p);
  mass[n] = o.mass;
  velocity[n] = o.velocity;
  if (albedoFlux) normalize_albedo_cell(albedoFlux, albedoSource, n, m);
}

__global__ void __launch_bounds__(kBlock)
    k_transfer(float2* __restrict__ deltas, const float2* __restrict__ layers,
               const float* __restrict__ upliftBase, const float* __restrict__ mass,
               const float2* __restrict__ velocityFluvial, const float* __restrict__ debris,
               const float* __restrict__ albedo_bedrock, const float* __restrict__ albedoFluvial,
               const float* __restrict__ albedoDebris, float* __restrict__ albedo_surface, Dom d,
               Scale3 s, Param p) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= (d.r1 - d.r0) * d.W) return;
  const int64_t lx = d.r0 + t / d.W, y = t % d.W;
  const int64_t n = lx * d.W + y;
  const float2 grad = glocal(layers, d, s, d.x0 + lx, y, p.exitSlope);  // :492
  const float2 layer = layers[n];                                      // :530
  float2 delta = deltas[n];                                            // :531
  const float massHeight = mass[n];
  const float debrisHeight = debris[n];
  const float transfer = transfer_cell(delta, layer, grad, upliftBase[n], massHeight,
                                       velocityFluvial[n], debrisHeight, s, p);
  deltas[n] = delta;  // :547

  if (albedo_surface) {  // :553-572
    const float eps = 1E-12f;
    const float totalHeight = massHeight + debrisHeight;
    const float mixDepth = 1.0f;
    if (layer.y == 0.0f) {
      for (int c = 0; c < 3; ++c) albedo_surface[3 * n + c] = albedo_bedrock[3 * n + c];
    } else if (totalHeight > 0.0f && transfer > eps) {
      const float wMass = fminf(massHeight / totalHeight, 1.0f);
      const float wSurf = fminf(mixDepth, layer.y * s.z);
      const float wTrsp = fmaxf(eps, transfer);
      const float w = fminf(wTrsp / (wTrsp + wSurf), 1.0f);
      for (int c = 0; c < 3; ++c) {
        const float colorTransport = fminf(
            wMass * albedoFluvial[3 * n + c] + (1.0f - wMass) * albedoDebris[3 * n + c], 1.0f);
        const float colorSurface = fminf(albedo_surface[3 * n + c], 1.0f);
        albedo_surface[3 * n + c] = w * colorTransport + (1.0f - w) * colorSurface;
      }
    }
  }
}

__global__ void __launch_bounds__(kBlock)
    k_mass_creep(float2* __restrict__ delta, const float2* __restrict__ layers, Dom d, Scale3 s,
                 Param p) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= (d.r1 - d.r0) * d.W) return;
  const int64_t lx = d.r0 + t / d.W, y = t % d.W;
  const int64_t n = lx * d.W + y;
  const Nbhd nb = load_nbhd(layers, d, lx, y);  // :654-658
  delta[n].y += creep_cell(nb.l00, nb.ln0, nb.lp0, nb.l0n, nb.l0p, s, p.critSlopeSediment);
}

__global__ void __launch_bounds__(kBlock)
    k_layer_merge(float* __restrict__ height, const float2* __restrict__ layers, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const float2 l = layers[i];
  height[i] = l.x + l.y;  // :743
}

__global__ void __launch_bounds__(kBlock)
    k_layers_from_planes(float2* __restrict__ layers, const float* __restrict__ bedrock,
                         const float* __restrict__ sediment, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  layers[i] = make_float2(bedrock[i], sediment ? sediment[i] : 0.0f);
}

__global__ void __launch_bounds__(kBlock)
    k_layers_to_planes(float* __restrict__ bedrock, float* __restrict__ sediment,
                       const float2* __restrict__ layers, int64_t n) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const float2 l = layers[i];
  bedrock[i] = l.x;
  if (sediment) sediment[i] = l.y;
}

__global__ void __launch_bounds__(kBlock)
    k_albedo_stratum(float* __restrict__ albedoBedrock, const float* __restrict__ uplift,
                     const float2* __restrict__ layers, int64_t n, Scale3 s, float ku, float3 cA,
                     float3 cB, float age, float freq) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const float shift = age * ku * uplift[i];                       // :812
  const float depth = fmaxf(shift - layers[i].x * s.z, 0.0f);     // :814
  const int index = static_cast<int>(floorf(depth / freq));       // :819
  const float3 c = (index % 2 == 0) ? cA : cB;                    // :820-824
  albedoBedrock[3 * i] = c.x;
  albedoBedrock[3 * i + 1] = c.y;
  albedoBedrock[3 * i + 2] = c.z;
}

__global__ void __launch_bounds__(kBlock)
    k_albedo_layer(float* __restrict__ albedo, const float* __restrict__ albedoBedrock,
                   const float* __restrict__ albedoSediment, const float2* __restrict__ layers,
                   int64_t n, float scaleSediment, float3 shiftSediment) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const float blend = 1.0f / (1.0f + scaleSediment * layers[i].y);  // :777
  const float sh[3] = {shiftSediment.x, shiftSediment.y, shiftSediment.z};
  for (int k = 0; k < 3; ++k) {
    const float colorSediment = fminf(albedoSediment[3 * i + k] + sh[k], 1.0f);             // :775
    albedo[3 * i + k] = blend * albedoBedrock[3 * i + k] + (1.0f - blend) * colorSediment;  // :778
  }
}

__global__ void __launch_bounds__(kBlock)
    k_albedo_discharge(float* __restrict__ albedo, const float* __restrict__ discharge, int64_t n,
                       float3 color, float extinction, float scale) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const float value = fmaxf(0.0f, discharge[i]);                        // :871
  const float blend = scale * (1.0f - expf_(-extinction * value));      // :872
  const float col[3] = {color.x, color.y, color.z};
  for (int k = 0; k < 3; ++k)
    albedo[3 * i + k] = blend * col[k] + (1.0f - blend) * albedo[3 * i + k];  // :873
}

// ---- fused step kernel -------------------------------------------------------
//
// One thread owns VEC=4 consecutive cells of one row.  Every plane is touched
// exactly once with 16-byte accesses (vec2 planes: two per thread), the layer
// rows above/below are read with the same aligned accesses, and the two
// y-neighbours that live in other lanes' registers are fetched with wave
// shuffles; only the first/last lane of a wave falls back to an 8-byte load.
// Work-groups are mapped so that each XCD sweeps a contiguous band of rows and
// the rows shared between neighbouring tiles stay in that XCD's L2.

constexpr int kVec = 4;

struct Planes {  // soil_erosion_planes by value, typed
  const float2* layers;
  float2* layers_next;
  float* height;
  const float* uplift;
  const float* rainfall;
  float* waterHeight;
  float* waterFlux;
  float* mass;
  float* massFlux;
  float2* velocity;
  float2* velocityFlux;
  float* debris;
  float* debrisFlux;
  float2* debrisVelocity;
  float2* debrisVelocityFlux;
};

struct Row4 {  // four consecutive float2
  float2 v[kVec];
};
typedef float v4f __attribute__((ext_vector_type(4)));
// streaming accesses: planes that are read or written exactly once per step
// bypass the caches' retention policy (nontemporal), the layer rows that the
// neighbouring rows re-read do not
template <bool NT>
__device__ __forceinline__ v4f ldv(const float* __restrict__ p) {
  const v4f* q = reinterpret_cast<const v4f*>(p);
  return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT>
__device__ __forceinline__ void stv(float* __restrict__ p, v4f v) {
  v4f* q = reinterpret_cast<v4f*>(p);
  if (NT) __builtin_nontemporal_store(v, q);
  else *q = v;
}
template <bool NT = false>
__device__ __forceinline__ Row4 load_row4(const float2* __restrict__ p) {
  const float* f = reinterpret_cast<const float*>(p);
  const v4f a = ldv<NT>(f), b = ldv<NT>(f + 4);
  Row4 r;
  r.v[0] = make_float2(a.x, a.y);
  r.v[1] = make_float2(a.z, a.w);
  r.v[2] = make_float2(b.x, b.y);
  r.v[3] = make_float2(b.z, b.w);
  return r;
}
template <bool NT = false>
__device__ __forceinline__ void store_row4(float2* __restrict__ p, const Row4& r) {
  float* f = reinterpret_cast<float*>(p);
  stv<NT>(f, v4f{r.v[0].x, r.v[0].y, r.v[1].x, r.v[1].y});
  stv<NT>(f + 4, v4f{r.v[2].x, r.v[2].y, r.v[3].x, r.v[3].y});
}
template <bool NT = false>
__device__ __forceinline__ void load4(const float* __restrict__ p, float o[kVec]) {
  const v4f a = ldv<NT>(p);
  o[0] = a.x;
  o[1] = a.y;
  o[2] = a.z;
  o[3] = a.w;
}
template <bool NT = false>
__device__ __forceinline__ void store4(float* __restrict__ p, const float o[kVec]) {
  stv<NT>(p, v4f{o[0], o[1], o[2], o[3]});
}

// The arithmetic of one cell of the fused step; shared by the vector and the
// scalar-tail paths.  Order of operations = reference order (see soil_hip.h).
struct CellResult {
  float2 layers_next;
  float height;
  FluvialOut fl;
  DebrisOut db;
};
__device__ __forceinline__ CellResult fused_cell(const Nbhd& nb, float uplift, float rainfall,
                                                 float waterFlux, float massFlux, float2 velFlux,
                                                 float debrisFlux, float2 debrisVelFlux, Scale3 s,
                                                 const Param& p) {
  const float2 grad = grad_of(nb, s, p.exitSlope);
  CellResult r;
  r.fl = normalize_fluvial_cell(grad, waterFlux, massFlux, velFlux, rainfall, s, p);
  r.db = normalize_debris_cell(grad, debrisFlux, debrisVelFlux, s, p);
  float2 delta = make_float2(0.0f, 0.0f);  // silt.set(delta, 0)
  (void)transfer_cell(delta, nb.l00, grad, uplift, r.fl.mass, r.fl.velocity, r.db.mass, s, p);
  delta.y += creep_cell(nb.l00, nb.ln0, nb.lp0, nb.l0n, nb.l0p, s, p.critSlopeSediment);
  r.layers_next = make_float2(nb.l00.x + delta.x, nb.l00.y + delta.y);  // silt.add(layers, delta)
  r.height = r.layers_next.x + r.layers_next.y;                         // __layer_merge
  return r;
}

// DIRECT: every lane stores its own 32 bytes of a two-channel plane (round 1's stores) instead of
// swapping them through LDS into 1 KiB-contiguous instructions — kept for A/B inside bench.py
// (SOIL_CELLS_VARIANT=4)
// REZERO = false: the five flux planes are left as they are (SOIL_CELLS_KEEP_FLUX: the next particle
// launch overwrites them, soil_erode_step's lazy mode) — 84 instead of 112 bytes per cell.
template <bool XCD_REMAP, bool NT, int BLOCK = kBlock, bool DIRECT = false, bool REZERO = true>
__global__ void __launch_bounds__(BLOCK)
    k_erode_cells_fused(Planes P, Dom d, Scale3 s, Param p, int64_t groups_per_row,
                        int64_t total_groups) {
  // group = kVec consecutive cells of one row; one thread per group
  int64_t blk = blockIdx.x;
  if (XCD_REMAP) {
    // blocks are dealt round-robin to the 8 XCDs (block b -> XCD b % 8); give
    // XCD k the k-th contiguous eighth of the tile sequence (gridDim.x % 8 == 0)
    const int64_t per = gridDim.x / 8;
    blk = (blk % 8) * per + blk / 8;
  }
  const int64_t g = blk * BLOCK + threadIdx.x;
  const bool active = g < total_groups;
  const int64_t gsafe = active ? g : total_groups - 1;
  const int64_t lx = d.r0 + gsafe / groups_per_row;
  const int64_t y0 = (gsafe % groups_per_row) * kVec;
  const int64_t gx = d.x0 + lx;
  const int64_t n0 = lx * d.W + y0;
  const int lane = threadIdx.x & 63;

  // every streaming load is issued before anything waits on one of them (the
  // shuffles below need `c`): 13 x 16-byte loads in flight per lane
  const bool has_n0 = gx - 1 >= 0, has_p0 = gx + 1 < d.H;
  const Row4 c = load_row4(P.layers + n0);
  const Row4 up = has_n0 ? load_row4(P.layers + n0 - d.W) : c;
  const Row4 dn = has_p0 ? load_row4(P.layers + n0 + d.W) : c;
  float uplift[kVec], rain[kVec], wflux[kVec], mflux[kVec], dflux[kVec];
  load4<NT>(P.uplift + n0, uplift);
  load4<NT>(P.rainfall + n0, rain);
  load4<NT>(P.waterFlux + n0, wflux);
  load4<NT>(P.massFlux + n0, mflux);
  load4<NT>(P.debrisFlux + n0, dflux);
  const Row4 vflux = load_row4<NT>(P.velocityFlux + n0);
  const Row4 dvflux = load_row4<NT>(P.debrisVelocityFlux + n0);
  const bool has_left = y0 - 1 >= 0, has_right = y0 + kVec < d.W;
  // lanes at a wave edge, or whose shuffle partner sits on another row, reload
  const bool left_ok = lane != 0 && (gsafe % groups_per_row) != 0;
  const bool right_ok = lane != 63 && (gsafe % groups_per_row) != groups_per_row - 1;
  float2 left = make_float2(0.0f, 0.0f), right = make_float2(0.0f, 0.0f);
  if (has_left && !left_ok) left = P.layers[n0 - 1];
  if (has_right && !right_ok) right = P.layers[n0 + kVec];

  // y-neighbours across the group boundary: previous lane's v[3], next lane's v[0]
  {
    const float lx_ = __shfl_up(c.v[3].x, 1, 64), ly_ = __shfl_up(c.v[3].y, 1, 64);
    const float rx_ = __shfl_down(c.v[0].x, 1, 64), ry_ = __shfl_down(c.v[0].y, 1, 64);
    if (left_ok) left = make_float2(lx_, ly_);
    if (right_ok) right = make_float2(rx_, ry_);
  }

  __shared__ float4 s_tile[DIRECT ? 1 : BLOCK / 64][DIRECT ? 1 : 128];
  Row4 o_layers, o_vel, o_dvel;
  float o_h[kVec], o_wh[kVec], o_m[kVec], o_d[kVec];
#pragma unroll
  for (int k = 0; k < kVec; ++k) {
    Nbhd nb;
    nb.l00 = c.v[k];
    nb.has_n0 = has_n0;
    nb.has_p0 = has_p0;
    nb.has_0n = (k > 0) || has_left;
    nb.has_0p = (k < kVec - 1) || has_right;
    nb.ln0 = up.v[k];
    nb.lp0 = dn.v[k];
    nb.l0n = (k > 0) ? c.v[k - 1] : (has_left ? left : c.v[k]);
    nb.l0p = (k < kVec - 1) ? c.v[k + 1] : (has_right ? right : c.v[k]);
    const CellResult r = fused_cell(nb, uplift[k], rain[k], wflux[k], mflux[k], vflux.v[k],
                                    dflux[k], dvflux.v[k], s, p);
    o_layers.v[k] = r.layers_next;
    o_h[k] = r.height;
    o_wh[k] = r.fl.waterHeight;
    o_m[k] = r.fl.mass;
    o_vel.v[k] = r.fl.velocity;
    o_d[k] = r.db.mass;
    o_dvel.v[k] = r.db.velocity;
  }

  if constexpr (DIRECT) {
    if (active) {
      store_row4<NT>(P.layers_next + n0, o_layers);
      store_row4<NT>(P.velocity + n0, o_vel);
      store_row4<NT>(P.debrisVelocity + n0, o_dvel);
      Row4 z2;
#pragma unroll
      for (int k = 0; k < kVec; ++k) z2.v[k] = make_float2(0.0f, 0.0f);
      if (REZERO) {
        store_row4<NT>(P.velocityFlux + n0, z2);
        store_row4<NT>(P.debrisVelocityFlux + n0, z2);
      }
    }
  } else {
    // A lane's four cells of a two-channel plane are 32 contiguous bytes: stored as they are, every
    // store instruction of the wave would cover half of each 32-byte sector.  The wave's groups are
    // consecutive in memory (n0 = 4 g, also across a row's end), so its 2 KiB go out as two
    // instructions of 1 KiB of consecutive bytes each, swapped through LDS (window.hpp).
    const int64_t wave_n0 = d.r0 * d.W + (g - lane) * kVec;  // n0 of the wave's lane 0
    auto pair = [&](float2* plane, const Row4& r) {
      store_pair_contiguous(reinterpret_cast<float4*>(plane + wave_n0),
                            make_float4(r.v[0].x, r.v[0].y, r.v[1].x, r.v[1].y),
                            make_float4(r.v[2].x, r.v[2].y, r.v[3].x, r.v[3].y),
                            s_tile[threadIdx.x >> 6], active);
    };
    pair(P.layers_next, o_layers);
    pair(P.velocity, o_vel);
    pair(P.debrisVelocity, o_dvel);
    if (REZERO) {  // re-zero the two-channel flux planes (zeros need no swapping: lane i takes the i-th and the
       // (64 + i)-th 16 bytes of the wave's stretch — every lane of the wave, its idle ones included)
      const int n_real = 2 * __popcll(__ballot(active));
      v4f* zv = reinterpret_cast<v4f*>(P.velocityFlux + wave_n0);
      v4f* zd = reinterpret_cast<v4f*>(P.debrisVelocityFlux + wave_n0);
      const v4f zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
      if (lane < n_real) {
        zv[lane] = zero4;
        zd[lane] = zero4;
      }
      if (64 + lane < n_real) {
        zv[64 + lane] = zero4;
        zd[64 + lane] = zero4;
      }
    }
  }
  if (!active) return;
  if (P.height) store4<NT>(P.height + n0, o_h);
  store4<NT>(P.waterHeight + n0, o_wh);
  store4<NT>(P.mass + n0, o_m);
  store4<NT>(P.debris + n0, o_d);
  if (REZERO) {  // re-zero the scalar flux planes for the next step's atomics
    const float z[kVec] = {0.0f, 0.0f, 0.0f, 0.0f};
    store4<NT>(P.waterFlux + n0, z);
    store4<NT>(P.massFlux + n0, z);
    store4<NT>(P.debrisFlux + n0, z);
  }
}

// The re-zero of the five flux planes as a pass of its own (28 bytes per cell, write only): what the
// eager path runs behind the 84-byte kernel when the two together beat the 112-byte kernel
// (SOIL_CELLS_SPLIT, soil_erode_cells_fused_ex).  One thread per 4 cells of the owned rows: 7 stores of 16 bytes.
__global__ void __launch_bounds__(kBlock)
    k_zero_flux(Planes P, int64_t n_first, int64_t groups) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (g >= groups) return;
  const int64_t n0 = n_first + g * kVec;
  const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
  *reinterpret_cast<v4f*>(P.waterFlux + n0) = z;
  *reinterpret_cast<v4f*>(P.massFlux + n0) = z;
  *reinterpret_cast<v4f*>(P.debrisFlux + n0) = z;
  v4f* zv = reinterpret_cast<v4f*>(P.velocityFlux + n0);
  v4f* zd = reinterpret_cast<v4f*>(P.debrisVelocityFlux + n0);
  zv[0] = z, zv[1] = z, zd[0] = z, zd[1] = z;
}

// scalar path for W % 4 != 0 (ragged widths): one thread per cell
__global__ void __launch_bounds__(kBlock)
    k_erode_cells_fused_scalar(Planes P, Dom d, Scale3 s, Param p, bool rezero) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= (d.r1 - d.r0) * d.W) return;
  const int64_t lx = d.r0 + t / d.W, y = t % d.W;
  const int64_t n = lx * d.W + y;
  const Nbhd nb = load_nbhd(P.layers, d, lx, y);
  const CellResult r = fused_cell(nb, P.uplift[n], P.rainfall[n], P.waterFlux[n], P.massFlux[n],
                                  P.velocityFlux[n], P.debrisFlux[n], P.debrisVelocityFlux[n], s, p);
  P.layers_next[n] = r.layers_next;
  if (P.height) P.height[n] = r.height;
  P.waterHeight[n] = r.fl.waterHeight;
  P.mass[n] = r.fl.mass;
  P.velocity[n] = r.fl.velocity;
  P.debris[n] = r.db.mass;
  P.debrisVelocity[n] = r.db.velocity;
  if (!rezero) return;
  P.waterFlux[n] = 0.0f;
  P.massFlux[n] = 0.0f;
  P.debrisFlux[n] = 0.0f;
  P.velocityFlux[n] = make_float2(0.0f, 0.0f);
  P.debrisVelocityFlux[n] = make_float2(0.0f, 0.0f);
}

static Scale3 s3(const float* s) { return Scale3{s[0], s[1], s[2]}; }
static float3 f3(const float* c) { return make_float3(c[0], c[1], c[2]); }

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// cell-phase entry points on an explicit domain (used by the slab ABI and by
// transport_* in erosion_particles.hip)
int launch_normalize_fluvial(const float* waterFlux, const float* massFlux,
                             const float* velocityFlux, float* albedoFlux, const float* layers,
                             const float* waterSource, float* waterHeight, float* mass,
                             float* velocity, const float* albedoSource, const Dom& d, Scale3 s,
                             const Param& p, hipStream_t st) {
  const int64_t cells = (d.r1 - d.r0) * d.W;
  if (cells <= 0) return SOIL_OK;
  k_normalize_fluvial<<<blocks_for(cells, kBlock), kBlock, 0, st>>>(
      waterFlux, massFlux, reinterpret_cast<const float2*>(velocityFlux), albedoFlux,
      reinterpret_cast<const float2*>(layers), waterSource, waterHeight, mass,
      reinterpret_cast<float2*>(velocity), albedoSource, d, s, p);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int launch_normalize_debris(const float* massFlux, const float* velocityFlux, float* albedoFlux,
                            const float* layers, float* mass, float* velocity,
                            const float* albedoSource, const Dom& d, Scale3 s, const Param& p,
                            hipStream_t st) {
  const int64_t cells = (d.r1 - d.r0) * d.W;
  if (cells <= 0) return SOIL_OK;
  k_normalize_debris<<<blocks_for(cells, kBlock), kBlock, 0, st>>>(
      massFlux, reinterpret_cast<const float2*>(velocityFlux), albedoFlux,
      reinterpret_cast<const float2*>(layers), mass, reinterpret_cast<float2*>(velocity),
      albedoSource, d, s, p);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

}  // namespace soil

using namespace soil;

extern "C" {

int soil_mass_transfer(float* delta, const float* layers, const float* uplift,
                       const float* waterHeight, const float* mass, const float* velocityFluvial,
                       const float* debris, const float* momentumDebris,
                       const float* albedo_bedrock, const float* albedoFluxFluvial,
                       const float* albedoFluxDebris, float* albedo_surface, int64_t H, int64_t W,
                       const float scale[3], const soil_param* param, void* stream) {
  (void)waterHeight;     // accepted and unread, erosion.cu:457
  (void)momentumDebris;  // accepted and unread, erosion.cu:461
  SOIL_DEVICE();
  SOIL_REQUIRE(delta && layers && uplift && mass && velocityFluvial && debris && scale && param,
               "mass_transfer: null tensor");
  SOIL_REQUIRE(H > 0 && W > 0, "mass_transfer: empty grid");
  const bool any_albedo = albedo_bedrock || albedoFluxFluvial || albedoFluxDebris || albedo_surface;
  const bool all_albedo = albedo_bedrock && albedoFluxFluvial && albedoFluxDebris && albedo_surface;
  SOIL_REQUIRE(!any_albedo || all_albedo, "mass_transfer: pass all four albedo planes or none");
  const Dom d = full_domain(H, W);
  k_transfer<<<blocks_for(H * W, kBlock), kBlock, 0, as_stream(stream)>>>(
      reinterpret_cast<float2*>(delta), reinterpret_cast<const float2*>(layers), uplift, mass,
      reinterpret_cast<const float2*>(velocityFluvial), debris, albedo_bedrock, albedoFluxFluvial,
      albedoFluxDebris, albedo_surface, d, s3(scale), *param);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_mass_creep(float* delta, const float* layers, int64_t H, int64_t W, const float scale[3],
                    const soil_param* param, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(delta && layers && scale && param, "mass_creep: null tensor");
  SOIL_REQUIRE(H > 0 && W > 0, "mass_creep: empty grid");
  const Dom d = full_domain(H, W);
  k_mass_creep<<<blocks_for(H * W, kBlock), kBlock, 0, as_stream(stream)>>>(
      reinterpret_cast<float2*>(delta), reinterpret_cast<const float2*>(layers), d, s3(scale),
      *param);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_layer_merge(float* height, const float* layers, int64_t n, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(height && layers, "layer_merge: null tensor");
  if (n <= 0) return SOIL_OK;
  k_layer_merge<<<blocks_for(n, kBlock), kBlock, 0, as_stream(stream)>>>(
      height, reinterpret_cast<const float2*>(layers), n);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_layers_from_planes(float* layers, const float* bedrock, const float* sediment, int64_t n,
                            void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(layers && bedrock, "layers_from_planes: null tensor");
  if (n <= 0) return SOIL_OK;
  k_layers_from_planes<<<blocks_for(n, kBlock), kBlock, 0, as_stream(stream)>>>(
      reinterpret_cast<float2*>(layers), bedrock, sediment, n);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_layers_to_planes(float* bedrock, float* sediment, const float* layers, int64_t n,
                          void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(layers && bedrock, "layers_to_planes: null tensor");
  if (n <= 0) return SOIL_OK;
  k_layers_to_planes<<<blocks_for(n, kBlock), kBlock, 0, as_stream(stream)>>>(
      bedrock, sediment, reinterpret_cast<const float2*>(layers), n);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_albedo_stratum(float* albedoBedrock, const float* uplift, const float* layers, int64_t n,
                        const float scale[3], const soil_param* param, const float colorA[3],
                        const float colorB[3], float age, float freq, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(albedoBedrock && uplift && layers && scale && param && colorA && colorB,
               "albedo_stratum: null argument");
  if (n <= 0) return SOIL_OK;
  k_albedo_stratum<<<blocks_for(n, kBlock), kBlock, 0, as_stream(stream)>>>(
      albedoBedrock, uplift, reinterpret_cast<const float2*>(layers), n, s3(scale), param->uplift,
      f3(colorA), f3(colorB), age, freq);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_albedo_layer(float* albedo, const float* albedoBedrock, const float* albedoSediment,
                      const float* layers, int64_t n, float scaleSediment,
                      const float shiftSediment[3], void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(albedo && albedoBedrock && albedoSediment && layers && shiftSediment,
               "albedo_layer: null argument");
  if (n <= 0) return SOIL_OK;
  k_albedo_layer<<<blocks_for(n, kBlock), kBlock, 0, as_stream(stream)>>>(
      albedo, albedoBedrock, albedoSediment, reinterpret_cast<const float2*>(layers), n,
      scaleSediment, f3(shiftSediment));
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_albedo_discharge(float* albedo, const float* discharge, int64_t n,
                          const float colorDischarge[3], float extinction, float scale,
                          void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(albedo && discharge && colorDischarge, "albedo_discharge: null argument");
  if (n <= 0) return SOIL_OK;
  k_albedo_discharge<<<blocks_for(n, kBlock), kBlock, 0, as_stream(stream)>>>(
      albedo, discharge, n, f3(colorDischarge), extinction, scale);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_erode_cells_fused(const soil_erosion_planes* pl, const soil_domain* dom,
                           const float scale[3], const soil_param* param, void* stream) {
  return soil_erode_cells_fused_ex(pl, dom, scale, param, 0, stream);
}

int soil_erode_cells_fused_ex(const soil_erosion_planes* pl, const soil_domain* dom,
                              const float scale[3], const soil_param* param, int flags, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(pl && dom && scale && param, "erode_cells_fused: null argument");
  SOIL_REQUIRE(pl->layers && pl->layers_next && pl->uplift && pl->rainfall && pl->waterHeight &&
                   pl->waterFlux && pl->mass && pl->massFlux && pl->velocity &&
                   pl->velocityFlux && pl->debris && pl->debrisFlux && pl->debrisVelocity &&
                   pl->debrisVelocityFlux,
               "erode_cells_fused: null plane (only `height` is optional)");
  SOIL_REQUIRE(pl->layers != pl->layers_next,
               "erode_cells_fused: layers and layers_next must be distinct buffers");
  const Dom d = to_dom(dom);
  int rc = check_domain(d);
  if (rc != SOIL_OK) return rc;
  const int64_t cells = (d.r1 - d.r0) * d.W;
  if (cells <= 0) return SOIL_OK;

  Planes P;
  P.layers = reinterpret_cast<const float2*>(pl->layers);
  P.layers_next = reinterpret_cast<float2*>(pl->layers_next);
  P.height = pl->height;
  P.uplift = pl->uplift;
  P.rainfall = pl->rainfall;
  P.waterHeight = pl->waterHeight;
  P.waterFlux = pl->waterFlux;
  P.mass = pl->mass;
  P.massFlux = pl->massFlux;
  P.velocity = reinterpret_cast<float2*>(pl->velocity);
  P.velocityFlux = reinterpret_cast<float2*>(pl->velocityFlux);
  P.debris = pl->debris;
  P.debrisFlux = pl->debrisFlux;
  P.debrisVelocity = reinterpret_cast<float2*>(pl->debrisVelocity);
  P.debrisVelocityFlux = reinterpret_cast<float2*>(pl->debrisVelocityFlux);

  const bool vec_ok = (d.W % kVec == 0) && aligned16(pl->layers) && aligned16(pl->layers_next) &&
                      (!pl->height || aligned16(pl->height)) && aligned16(pl->uplift) &&
                      aligned16(pl->rainfall) && aligned16(pl->waterHeight) &&
                      aligned16(pl->waterFlux) && aligned16(pl->mass) && aligned16(pl->massFlux) &&
                      aligned16(pl->velocity) && aligned16(pl->velocityFlux) &&
                      aligned16(pl->debris) && aligned16(pl->debrisFlux) &&
                      aligned16(pl->debrisVelocity) && aligned16(pl->debrisVelocityFlux);
  hipStream_t st = as_stream(stream);
  if (vec_ok) {
    const int64_t groups_per_row = d.W / kVec;
    const int64_t total = (d.r1 - d.r0) * groups_per_row;
    const unsigned nblk = blocks_for(total, kBlock);
    static const bool nt = [] { const char* e = std::getenv("SOIL_CELLS_NT"); return e && e[0] == '1'; }();  // measured slower than plain accesses; kept for A/B
    const int variant = [] { const char* e = std::getenv("SOIL_CELLS_VARIANT"); return e ? std::atoi(e) : 0; }();  // read per call: bench.py alternates variants in one process
    const bool remap = nblk % 8 == 0 && nblk >= 64 && variant != 2;
    // split: the eager call as the 84-byte kernel + the zeroing pass (variant 5 / SOIL_CELLS_SPLIT=1)
    // Measured back to back at 8192^2 on one box (tools/bench_cells.py, two processes each): 1.30 / 1.27 ms
    // split against 1.38 / 1.29 ms for the one kernel that moves all 112 bytes — the kernel with seven
    // store streams less plus a write-only pass at 7 TB/s is never the slower one.  SOIL_CELLS_SPLIT=0:
    // the one kernel (variant 0) again.
    static const bool split_env = [] { const char* e = std::getenv("SOIL_CELLS_SPLIT"); return !(e && e[0] == '0'); }();
    const bool split = (flags & SOIL_CELLS_KEEP_FLUX) == 0 && (variant == 5 || (split_env && variant == 0));
    const bool keep = (flags & SOIL_CELLS_KEEP_FLUX) != 0 || split;
    if (keep && remap)
      k_erode_cells_fused<true, false, kBlock, false, false><<<nblk, kBlock, 0, st>>>(P, d, s3(scale), *param, groups_per_row, total);
    else if (keep)
      k_erode_cells_fused<false, false, kBlock, false, false><<<nblk, kBlock, 0, st>>>(P, d, s3(scale), *param, groups_per_row, total);
    else if (variant == 1 && (total % 512) == 0 && ((total / 512) % 8) == 0)
      k_erode_cells_fused<true, false, 512><<<static_cast<unsigned>(total / 512), 512, 0, st>>>(P, d, s3(scale), *param, groups_per_row, total);
    else if (variant == 3 && (total % 128) == 0 && ((total / 128) % 8) == 0)
      k_erode_cells_fused<true, false, 128><<<static_cast<unsigned>(total / 128), 128, 0, st>>>(P, d, s3(scale), *param, groups_per_row, total);
    else if (variant == 4 && remap)
      k_erode_cells_fused<true, false, kBlock, true><<<nblk, kBlock, 0, st>>>(P, d, s3(scale), *param, groups_per_row, total);
    else if (remap && nt)
      k_erode_cells_fused<true, true><<<nblk, kBlock, 0, st>>>(P, d, s3(scale), *param, groups_per_row, total);
    else if (remap)
      k_erode_cells_fused<true, false><<<nblk, kBlock, 0, st>>>(P, d, s3(scale), *param, groups_per_row, total);
    else if (nt)
      k_erode_cells_fused<false, true><<<nblk, kBlock, 0, st>>>(P, d, s3(scale), *param, groups_per_row, total);
    else
      k_erode_cells_fused<false, false><<<nblk, kBlock, 0, st>>>(P, d, s3(scale), *param, groups_per_row, total);
    if (split) k_zero_flux<<<nblk, kBlock, 0, st>>>(P, d.r0 * d.W, total);
  } else {
    k_erode_cells_fused_scalar<<<blocks_for(cells, kBlock), kBlock, 0, st>>>(
        P, d, s3(scale), *param, (flags & SOIL_CELLS_KEEP_FLUX) == 0);
  }
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

}  // extern "C"
