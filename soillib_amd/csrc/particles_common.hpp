// particles_common.hpp — pieces shared by the particle-transport launch shapes
// (erosion_particles.hip: direct / staged; erosion_particles_tiled.hip: tiled).
#pragma once

#include "cell_math.hpp"

namespace soil {

__device__ __forceinline__ bool oob(const Dom& d, float px, float py) {  // erosion_map.cu:29-40
  if (px < 0) return true;
  if (py < 0) return true;
  if (px >= static_cast<float>(d.H)) return true;
  if (py >= static_cast<float>(d.W)) return true;
  return false;
}

// local rows whose 5-point stencil lies inside the rows this slab holds
__device__ __host__ __forceinline__ int64_t stencil_lo(const Dom& d) { return (d.x0 == 0) ? 0 : 1; }
__device__ __host__ __forceinline__ int64_t stencil_hi(const Dom& d) {  // inclusive
  return (d.x0 + d.rows == d.H) ? d.rows - 1 : d.rows - 2;
}
// A slab traces a particle only while its cell's stencil is available
// (soil_hip.h, soil_particles_*_slab).
__device__ __forceinline__ bool slab_escape(const Dom& d, int64_t gx) {
  const int64_t lx = gx - d.x0;
  return lx < stencil_lo(d) || lx > stencil_hi(d);
}

__device__ __forceinline__ bool owns_spawn(const Dom& d, float px) {
  const int64_t sx = cell_of(px) - d.x0;
  return sx >= d.r0 && sx < d.r1;
}
// first two draws of particle n: spawn position (erosion.cu:56-59 / :269-272).  The row
// decides who traces the particle: a slab that does not own it (7 of 8 streams on an
// 8-GPU run) leaves the second draw out — the stream still moves on by two.
__device__ __forceinline__ float2 spawn_position(soil_rng* __restrict__ rng, int64_t n,
                                                 const Dom& d) {
  soil_rng st = rng[n];
  const float u1 = rng_uniform_at(st.seed, static_cast<uint64_t>(n), st.offset);
  const float x = 0.5f + u1 * static_cast<float>(d.H - 1);
  float y = 0.0f;
  if (owns_spawn(d, x))
    y = 0.5f + rng_uniform_at(st.seed, static_cast<uint64_t>(n), st.offset + 1) * static_cast<float>(d.W - 1);
  st.offset += 2;
  rng[n] = st;  // the state persists in the tensor, like curandState
  return make_float2(x, y);
}

// Where a launch's streams of draws come from.  The reference keeps one generator state per particle
// (silt::rng = curandState, erosion.cu:57); a launch reads every state, draws, and writes it back.
// `uniform`: every stream's state is {seed, offset} — a tensor that was seeded for this very launch
// and is seeded again before the next, which is what the library's own step drivers do.  Nothing is
// read or written then, and only the particles this slab owns get a record slot: the replay of the
// other ranks' streams (7 of 8 on an 8-GPU run) costs one Philox draw per stream and no memory
// traffic, where the tensor form moves 36 bytes per stream (csrc/slab_runner.hip, HipOps).
struct Streams {
  soil_rng* rng;
  bool uniform;
  uint64_t seed, offset;
};
inline Streams streams_of(soil_rng* rng) { return Streams{rng, false, 0, 0}; }
// spawn position of particle n from the two first draws of its stream (see spawn_position)
__device__ __forceinline__ float2 spawn_position(const Streams& st, int64_t n, const Dom& d) {
  if (!st.uniform) return spawn_position(st.rng, n, d);
  const float u1 = rng_uniform_at(st.seed, static_cast<uint64_t>(n), st.offset);
  const float x = 0.5f + u1 * static_cast<float>(d.H - 1);
  float y = 0.0f;
  if (owns_spawn(d, x))
    y = 0.5f + rng_uniform_at(st.seed, static_cast<uint64_t>(n), st.offset + 1) * static_cast<float>(d.W - 1);
  return make_float2(x, y);
}
// Walkers handed over at the slab's edge (SURVEY.md 8e option B; the slab runner's `migrate` mode,
// round 5): a walker that steps off the rows this launch may walk on, inside the grid and with life
// left, is written — state untouched, at the top of an iteration, as at a tile edge — into the box of
// the side it left through instead of being dropped (the deep-halo runner never lets one get there).
// 64-byte records (erosion_particles_tiled.hip: PRec), global coordinates: the neighbour injects them
// into its own queues as they are.  count[0] / count[1]: records written up / down (may exceed `cap`:
// the caller checks).
struct MigrateBox {
  void* up = nullptr;
  void* down = nullptr;
  uint32_t* count = nullptr;
  uint32_t cap = 0;
};
// one launch of `kind` (0 fluvial, 1 debris) in the tiled shape whatever N: from the streams' spawns
// (`inbox` null) or from `n_in` handed-over records; deposits into P's flux planes of that kind
int launch_pass_tiled(int kind, const soil_erosion_planes& P, Streams rng, int64_t N, float* remote0, const Dom& d,
                      Scale3 s, const Param& p, hipStream_t st, const void* inbox, uint32_t n_in, MigrateBox box);

// the launch shape a launch of N particles on domain d gets (erosion_particles.hip)
bool use_tiled_launch(int64_t N, const Dom& d);
int debris_retire_mode();    // soil_set_debris_retire / SOIL_DEBRIS_RETIRE (erosion_particles.hip): 0 off, 1 on, 2 watched
bool particle_arith_fast();  // soil_set_particle_arith(1) / SOIL_PARTICLE_DIV=fast (erosion_particles.hip)

// exclusive scan of per-tile counts, start[tiles] = total (one 1024-thread group;
// defined in erosion_particles.hip)
__global__ void __launch_bounds__(1024)
    k_tile_scan(uint32_t* start, const uint32_t* count, int64_t tiles);

// tiled launch shape (erosion_particles_tiled.hip)
int launch_fluvial_tiled(float* waterFlux, float* massFlux, float* velocityFlux, float* albedoFlux,
                         Streams rng, int64_t N, const float* layers, const float* waterSource,
                         const float* waterHeight, const float* velocity,
                         const float* albedoSource, float* remote0, const Dom& d, Scale3 s,
                         const Param& p, hipStream_t st);
int launch_debris_tiled(float* massFlux, float* velocityFlux, float* albedoFlux, Streams rng,
                        int64_t N, const float* layers, const float* velocity,
                        const float* albedoSource, float* remote0, const Dom& d, Scale3 s,
                        const Param& p, hipStream_t st);
// both launches of a step overlapped on two internal streams forked from / joined into `st`
// (`overwrite`: the flux planes hold stale values — SOIL_FLUX_OVERWRITE, soil_hip.h)
int launch_pair_tiled(const soil_erosion_planes& P, Streams rng_fluvial, Streams rng_debris,
                      int64_t N, float* remote0, const Dom& d, Scale3 s, const Param& p,
                      hipStream_t st, bool overwrite, MigrateBox box_fluvial = MigrateBox{},
                      MigrateBox box_debris = MigrateBox{}, const void* inbox_fluvial = nullptr,
                      uint32_t n_fluvial = 0, const void* inbox_debris = nullptr, uint32_t n_debris = 0);
// (inboxes: both launches start from handed-over records instead of the streams' spawns — the immigrants of
// both kinds walked on side by side, slab runner's migrate mode; the pack pass of the step's spawn launches
// stands)
// the slab entry points of soil_hip.h on explicit streams (the slab runner's HIP back-end)
int particles_fluvial_streams(const soil_erosion_planes& P, Streams rng, int64_t N, float* remote0,
                              const Dom& d, Scale3 s, const Param& p, hipStream_t st);
int particles_debris_streams(const soil_erosion_planes& P, Streams rng, int64_t N, float* remote0,
                             const Dom& d, Scale3 s, const Param& p, hipStream_t st);
int particles_pair_streams(const soil_erosion_planes& P, Streams rng_fluvial, Streams rng_debris, int64_t N,
                           float* remote0, const Dom& d, Scale3 s, const Param& p, hipStream_t st, bool overwrite = false);

}  // namespace soil
