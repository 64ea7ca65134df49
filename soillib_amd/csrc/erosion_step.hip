// erosion_step.hip — the step driver behind the C ABI: one whole erosion step, and the legacy
// composite soil::erode, composed in C++ from the library's own entry points.
//
// The reference snapshot no longer holds soil::erode (only its commented-out binding,
// python/source/model.cpp:142, and the call in example/erosion_gpu.py:102-106); what remains are the
// kernels it was made of.  SURVEY.md 3.1 fixes the composition of one step:
//     silt.seed(rng, seed, step * N)                        example/dem_process.py:81
//     transport_fluvial, transport_debris (particle halves) erosion.cu:29-141, :245-351
//     normalise x2, delta = 0, mass_transfer, mass_creep,   erosion.cu:143-187, :353-393, :453-574,
//     layers += delta, layer_merge, track.* = 0             :633-710, dem_process.py:47, erosion.cu:733-745
// — two particle launches and ONE fused cell kernel here (erosion_cells.hip).  The host language
// above this file only forwards pointers: a C++ program gets the step loop, the re-seeding and the
// buffer swap from the library (include/soil.hpp, soil::erode), exactly as the Python module does.
#include <cstdlib>

#include "common.hpp"

using namespace soil;

namespace soil {

// How far into a slab's ghost rows did this step's deposits get?  One work-group per ghost row
// looks for a value that is not zero; depth[0] = rows above the owned range [r0, r1) that hold one
// (counted from the boundary), depth[1] = rows below.  Accumulates with max: the caller clears
// `depth` and may call this for several planes.
__global__ void __launch_bounds__(256)
    k_ghost_extent(int32_t* __restrict__ depth, const float* __restrict__ plane, int64_t rows,
                   int64_t row_floats, int64_t r0, int64_t r1) {
  const int64_t ghost = static_cast<int64_t>(blockIdx.x);        // 0 .. r0 + (rows - r1) - 1
  const int64_t lx = ghost < r0 ? ghost : r1 + (ghost - r0);
  const float* row = plane + lx * row_floats;
  bool hit = false;
  // any bit set counts: NaN, and -0.0 too (what a dead debris walker deposits: 0 times a negative
  // source) — such a row is shipped and re-zeroed like in the untrimmed exchange
  for (int64_t i = threadIdx.x; i < row_floats && !hit; i += 256) hit = f2bits(row[i]) != 0u;
  if (__syncthreads_or(hit) && threadIdx.x == 0) {
    if (lx < r0) atomicMax(&depth[0], static_cast<int32_t>(r0 - lx));
    else atomicMax(&depth[1], static_cast<int32_t>(lx - r1 + 1));
  }
}

}  // namespace soil

extern "C" {

int soil_ghost_extent(int32_t* depth, const float* plane, int64_t rows, int64_t row_floats,
                      int64_t r0, int64_t r1, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(depth && plane, "ghost_extent: null argument");
  SOIL_REQUIRE(rows > 0 && row_floats > 0 && 0 <= r0 && r0 <= r1 && r1 <= rows,
               "ghost_extent: bad row ranges");
  const int64_t ghost = r0 + (rows - r1);
  if (ghost == 0) return SOIL_OK;
  k_ghost_extent<<<static_cast<unsigned>(ghost), 256, 0, as_stream(stream)>>>(depth, plane, rows,
                                                                              row_floats, r0, r1);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_erode_step(const soil_erosion_planes* planes, soil_rng* rng, int64_t N, uint64_t seed,
                    uint64_t step_index, int64_t H, int64_t W, const float scale[3],
                    const soil_param* param, void* stream) {
  return soil_erode_step_ex(planes, rng, N, seed, step_index, H, W, scale, param, 0, stream);
}

int soil_erode_step_ex(const soil_erosion_planes* planes, soil_rng* rng, int64_t N, uint64_t seed,
                       uint64_t step_index, int64_t H, int64_t W, const float scale[3],
                       const soil_param* param, int flags, void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(planes && rng && scale && param, "erode_step: null argument");
  SOIL_REQUIRE(H > 0 && W > 0 && N > 0, "erode_step: empty grid or no particles");
  const soil_erosion_planes& P = *planes;
  SOIL_REQUIRE(P.layers && P.layers_next && P.uplift && P.rainfall && P.waterHeight && P.waterFlux &&
                   P.mass && P.massFlux && P.velocity && P.velocityFlux && P.debris && P.debrisFlux &&
                   P.debrisVelocity && P.debrisVelocityFlux,
               "erode_step: every plane but `height` is required");
  const soil_domain dom{H, W, 0, H, 0, H};
  // One stream of draws per particle and step: (seed, subsequence n, offset step * N).  The
  // fluvial launch takes draws 0 and 1 of every stream, the debris launch draws 2 and 3.
  const uint64_t offset = step_index * static_cast<uint64_t>(N);
  // read on every call (one getenv per step): a host may switch it between steps
  const char* pair_env = std::getenv("SOIL_STEP_PAIR");
  const bool sequential = pair_env && pair_env[0] == '0';
  // Lazy flux planes (SOIL_STEP_FLUX_IN_DIRTY / _OUT_DIRTY): a chain of steps need not write 28 bytes
  // of zeros per cell and step — the first round of the next step's particle launches overwrites
  // the planes (SOIL_FLUX_OVERWRITE) — only the last step of a chain re-zeroes them.
  const bool in_dirty = (flags & SOIL_STEP_FLUX_IN_DIRTY) != 0, out_dirty = (flags & SOIL_STEP_FLUX_OUT_DIRTY) != 0;
  const int cell_flags = out_dirty ? SOIL_CELLS_KEEP_FLUX : 0;
  if (sequential) {  // one launch after the other on the caller's streams (diagnostics: phase timings)
    if (in_dirty) {
      const size_t b = sizeof(float) * static_cast<size_t>(H) * static_cast<size_t>(W);
      for (float* t : {P.waterFlux, P.massFlux, P.debrisFlux}) SOIL_HIP(hipMemsetAsync(t, 0, b, as_stream(stream)));
      for (float* t : {P.velocityFlux, P.debrisVelocityFlux}) SOIL_HIP(hipMemsetAsync(t, 0, 2 * b, as_stream(stream)));
    }
    if (int rc = soil_rng_seed(rng, N, seed, offset, stream); rc != SOIL_OK) return rc;
    if (int rc = soil_particles_fluvial_slab(P.waterFlux, P.massFlux, P.velocityFlux, nullptr, rng, N,
                                             P.layers, P.rainfall, P.waterHeight, P.velocity, nullptr,
                                             nullptr, &dom, scale, param, stream);
        rc != SOIL_OK)
      return rc;
    if (int rc = soil_particles_debris_slab(P.debrisFlux, P.debrisVelocityFlux, nullptr, rng, N, P.layers,
                                            P.debrisVelocity, nullptr, nullptr, &dom, scale, param,
                                            stream);
        rc != SOIL_OK)
      return rc;
    return soil_erode_cells_fused_ex(planes, &dom, scale, param, cell_flags, stream);
  }
  // The two launches do not depend on each other (they add to different flux planes and read the
  // same fields), so they run overlapped: the sparse late rounds and the finishing launch of
  // one fill with the dense rounds of the other — 3.0 -> 2.0 ms per step at 1024^2, 5.7 -> 4.8 at
  // 2048^2, 12.1 -> 11.7 at 4096^2, 37.3 -> 37.0 at 8192^2.  The fluvial launch draws from a scratch
  // tensor, the debris launch from the caller's, seeded two draws on: results and the state `rng`
  // is left in are those of the sequential order.
  void* scratch = nullptr;
  if (int rc = workspace_get(7, sizeof(soil_rng) * static_cast<size_t>(N), &scratch); rc != SOIL_OK) return rc;
  soil_rng* rng_fluvial = static_cast<soil_rng*>(scratch);
  if (int rc = soil_rng_seed(rng_fluvial, N, seed, offset, stream); rc != SOIL_OK) return rc;
  if (int rc = soil_rng_seed(rng, N, seed, offset + 2, stream); rc != SOIL_OK) return rc;
  if (int rc = soil_particles_pair_slab_ex(planes, rng_fluvial, rng, N, nullptr, &dom, scale, param,
                                           in_dirty ? SOIL_FLUX_OVERWRITE : 0, stream);
      rc != SOIL_OK)
    return rc;
  return soil_erode_cells_fused_ex(planes, &dom, scale, param, cell_flags, stream);
}

int soil_erode(const soil_erode_model* model, int64_t H, int64_t W, int64_t N, uint64_t seed,
               uint64_t first_step, int steps, const float scale[3], const soil_param* param,
               void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(model && scale && param, "erode: null argument");
  SOIL_REQUIRE(H > 0 && W > 0 && N > 0 && steps >= 0, "erode: empty grid, no particles or negative steps");
  const soil_erode_model& M = *model;
  SOIL_REQUIRE(M.height && M.sediment && M.uplift && M.rainfall && M.discharge && M.mass &&
                   M.momentum && M.debris && M.debris_momentum && M.discharge_track &&
                   M.mass_track && M.momentum_track && M.debris_track && M.debris_momentum_track,
               "erode: null plane");
  if (steps == 0) return SOIL_OK;
  const int64_t n = H * W;
  // what the library owns during the call: the double-buffered (H,W,2) layer plane and the
  // particle streams (scratch like graph.cu:539-550 allocates per call; cached here)
  auto align = [](size_t b) { return (b + 255) & ~static_cast<size_t>(255); };
  const size_t b_layers = align(sizeof(float) * 2 * n), b_rng = align(sizeof(soil_rng) * N);
  void* base = nullptr;
  if (int rc = workspace_get(6, 2 * b_layers + b_rng, &base); rc != SOIL_OK) return rc;
  char* w = static_cast<char*>(base);
  float* layers = reinterpret_cast<float*>(w);
  float* layers_next = reinterpret_cast<float*>(w + b_layers);
  soil_rng* rng = reinterpret_cast<soil_rng*>(w + 2 * b_layers);
  if (int rc = soil_layers_from_planes(layers, M.height, M.sediment, n, stream); rc != SOIL_OK) return rc;
  // silt.set(track.*, 0): the particle kernels only ever add to the flux planes
  for (float* t : {M.discharge_track, M.mass_track, M.debris_track})
    if (int rc = soil_set_f32(t, 0.0f, n, stream); rc != SOIL_OK) return rc;
  for (float* t : {M.momentum_track, M.debris_momentum_track})
    if (int rc = soil_set_f32(t, 0.0f, 2 * n, stream); rc != SOIL_OK) return rc;
  soil_erosion_planes P{};
  P.height = nullptr;  // model.height is the bedrock plane here; it is split back out below
  P.uplift = M.uplift;
  P.rainfall = M.rainfall;
  P.waterHeight = M.discharge;
  P.waterFlux = M.discharge_track;
  P.mass = M.mass;
  P.massFlux = M.mass_track;
  P.velocity = M.momentum;
  P.velocityFlux = M.momentum_track;
  P.debris = M.debris;
  P.debrisFlux = M.debris_track;
  P.debrisVelocity = M.debris_momentum;
  P.debrisVelocityFlux = M.debris_momentum_track;
  for (int s = 0; s < steps; ++s) {
    P.layers = layers;
    P.layers_next = layers_next;
    // the track planes were zeroed above and are left zeroed by the last step; in between nobody
    // looks at them
    const int flags = (s > 0 ? SOIL_STEP_FLUX_IN_DIRTY : 0) | (s + 1 < steps ? SOIL_STEP_FLUX_OUT_DIRTY : 0);
    if (int rc = soil_erode_step_ex(&P, rng, N, seed, first_step + static_cast<uint64_t>(s), H, W, scale,
                                    param, flags, stream);
        rc != SOIL_OK)
      return rc;
    float* t = layers;
    layers = layers_next;
    layers_next = t;
  }
  return soil_layers_to_planes(M.height, M.sediment, layers, n, stream);
}

}  // extern "C"
