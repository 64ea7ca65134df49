// cell_math.hpp — per-cell arithmetic of the erosion cell phase, written once
// and shared by the stand-alone kernels (one per reference op) and the fused
// step kernel, so that both evaluate the very same fp32 expression trees.
#pragma once

#include "common.hpp"

namespace soil {

struct FluvialOut {
  float waterHeight, mass;
  float2 velocity;
};

// __normalize_fluvial, erosion.cu:163-179 (physics planes)
__device__ __forceinline__ FluvialOut normalize_fluvial_cell(float2 grad, float waterFlux,
                                                             float massFlux, float2 velocityFlux,
                                                             float waterSource, Scale3 s,
                                                             const Param& p) {
  const float A = s.x * s.y;                                 // :163
  const float norm = fabsf(1.0f * s.y) + fabsf(0.0f * s.x);  // :165-166 (velocity fixed to (1,0))
  const float source_w = p.rainfall * waterSource;           // :173
  const float svx = -p.gravity * grad.x + p.force[0];        // :174
  const float svy = -p.gravity * grad.y + p.force[1];
  const float source_m = 0.0f;                               // :175
  FluvialOut o;
  o.waterHeight = (A * source_w + waterFlux) / norm;         // :177
  o.mass = (A * source_m + massFlux) / norm;                 // :178
  o.velocity.x = (A * svx + velocityFlux.x) / norm;          // :179
  o.velocity.y = (A * svy + velocityFlux.y) / norm;
  return o;
}

struct DebrisOut {
  float mass;
  float2 velocity;
};

// __normalize_debris, erosion.cu:370-385
__device__ __forceinline__ DebrisOut normalize_debris_cell(float2 grad, float massFlux,
                                                           float2 velocityFlux, Scale3 s,
                                                           const Param& p) {
  const float A = s.x * s.y;
  const float norm = fabsf(1.0f * s.y) + fabsf(0.0f * s.x);
  const float svx = -p.gravity * grad.x;  // :380
  const float svy = -p.gravity * grad.y;
  const float source_d = 0.0f;            // :381
  DebrisOut o;
  o.mass = (A * source_d + massFlux) / norm;         // :384
  o.velocity.x = (A * svx + velocityFlux.x) / norm;  // :385
  o.velocity.y = (A * svy + velocityFlux.y) / norm;
  return o;
}

// albedo tail of both normalise kernels, erosion.cu:181-185 / :387-391
// (3-component norm: SURVEY.md Appendix A4)
__device__ __forceinline__ void normalize_albedo_cell(float* __restrict__ albedoFlux,
                                                      const float* __restrict__ albedoSource,
                                                      int64_t n, float m) {
  const float a0 = albedoFlux[3 * n], a1 = albedoFlux[3 * n + 1], a2 = albedoFlux[3 * n + 2];
  if (m > 0.0f && sqrtf(a0 * a0 + a1 * a1 + a2 * a2) > 0.0f) {
    albedoFlux[3 * n] = a0 / m;
    albedoFlux[3 * n + 1] = a1 / m;
    albedoFlux[3 * n + 2] = a2 / m;
  } else {
    albedoFlux[3 * n] = albedoSource[3 * n];
    albedoFlux[3 * n + 1] = albedoSource[3 * n + 1];
    albedoFlux[3 * n + 2] = albedoSource[3 * n + 2];
  }
}

// Physics half of __transfer, erosion.cu:476-547.  Updates `delta` in place and
// returns the post-update `transfer` the albedo block needs (:558-572).
__device__ __forceinline__ float transfer_cell(float2& delta, float2 layer, float2 grad,
                                               float upliftBase, float massHeight, float2 speed,
                                               float debrisHeight, Scale3 s, const Param& p) {
  // rate constants of the launch, derived as erosion.cu:476-487 derives them
  const float step = p.timeStep;
  const float pick_rate = p.suspensionRateFluvial / 64.0f;    // :478
  const float settle_rate = p.depositionRateFluvial * 1.33f;  // :479
  const float drag = p.frictionFactor / 8.0f;                 // :480

  const float cell_diag = length2(s.x, s.y);                  // :493
  const float steepness = length2(grad.x, grad.y);            // :494

  // water: stream power picks sediment up, suspended mass settles (:498-506)
  const float flow_speed = length2(speed.x, speed.y);
  const float bed_shear = 0.125f * drag * p.densityWater * flow_speed * flow_speed;
  const float stream_power = powf_(bed_shear * steepness, p.fluvialExponent);
  const float picked_up = pick_rate * stream_power;
  const float settled = settle_rate * massHeight;
  const float raised = p.uplift * upliftBase;

  // debris: landslides above the critical slope, yield stress either way (:510-514)
  const float over_critical = steepness - p.critSlopeBedrock;
  const float slide = fmaxf(0.0f, p.landslideRateDebris * over_critical);
  const float yield_excess = p.gravity * (debrisHeight * over_critical - p.yieldStress);
  const float debris_up = slide + p.suspensionRateDebris * fmaxf(0.0f, yield_excess);
  const float debris_down = fminf(debrisHeight, fmaxf(0.0f, -p.depositionRateDebris * yield_excess));

  // what the cell gains (+) or loses (-) this step, limited to a quarter of the slope's drop (:526-528)
  float transfer = step * (settled - picked_up + debris_down - debris_up);
  transfer = fmaxf(transfer, -0.25f * cell_diag * steepness);
  transfer = fminf(transfer, 0.25f * cell_diag * 0.3f);

  delta.x += step * raised / s.z;            // :532  bedrock rises
  delta.y += fmaxf(0.0f, transfer / s.z);    // :533  a gain goes to the sediment
  if (transfer < 0.0f) {                     // :535-545  a loss takes sediment first, bedrock for the rest
    const float from_sediment = fmaxf(-layer.y * s.z, transfer);
    delta.y += from_sediment / s.z;
    transfer -= from_sediment;
    delta.x += transfer / s.z;
  }
  return transfer;
}

// the lambda at erosion.cu:675-680
__device__ __forceinline__ float creep_T(float2 lb, float2 lt, float dx, float sz,
                                         float critSlope) {
  const float hb = (lb.x + lb.y) * sz;
  const float ht = (lt.x + lt.y) * sz;
  const float tmax = 0.5f * ((ht - hb) - critSlope * dx);
  return fmaxf(0.0f, fminf(lt.y * sz, tmax));
}

// __mass_creep, erosion.cu:660-708: returns the increment of delta.y.  The
// neighbour layers are already "self" where the neighbour is outside the grid.
__device__ __forceinline__ float creep_cell(float2 l00, float2 ln0, float2 lp0, float2 l0n,
                                            float2 l0p, Scale3 s, float critSlope) {
  const float h00 = (l00.x + l00.y) * s.z;
  const float hn0 = (ln0.x + ln0.y) * s.z;
  const float hp0 = (lp0.x + lp0.y) * s.z;
  const float h0n = (l0n.x + l0n.y) * s.z;
  const float h0p = (l0p.x + l0p.y) * s.z;
  float t = 0.0f;
  if (hp0 > h00) t += creep_T(l00, lp0, s.x, s.z, critSlope);
  else t -= creep_T(lp0, l00, s.x, s.z, critSlope);
  if (hn0 > h00) t += creep_T(l00, ln0, s.x, s.z, critSlope);
  else t -= creep_T(ln0, l00, s.x, s.z, critSlope);
  if (h0p > h00) t += creep_T(l00, l0p, s.y, s.z, critSlope);
  else t -= creep_T(l0p, l00, s.y, s.z, critSlope);
  if (h0n > h00) t += creep_T(l00, l0n, s.y, s.z, critSlope);
  else t -= creep_T(l0n, l00, s.y, s.z, critSlope);
  return 0.25f * t / s.z;  // :708
}

}  // namespace soil
