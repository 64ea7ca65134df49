/*
 * soil_hip.h — C ABI of the MI355X-native grid-erosion hot path.
 *
 * Every entry point below replaces one free function of the reference's
 * `namespace soil` operator API (the functions the nanobind module
 * python/source/model.cpp binds).  The reference file:line each one stands in
 * for is cited next to the declaration.  Paths are relative to the reference
 * repository root:
 *
 *   erosion.hpp / erosion.cu / erosion_map.cu = source/soillib/model/path/...
 *   graph.hpp / graph.cu                      = source/soillib/model/graph/...
 *   grad.hpp / grad.cu                        = source/soillib/model/grad/...
 *   filter.hpp / filter.cu                    = source/soillib/model/filter/...
 *   path.hpp / path.cu / sample.hpp           = source/soillib/model/path/...
 *   normal.hpp / noise.hpp                    = source/soillib/op/...
 *   model.cpp                                 = python/source/model.cpp
 *
 * Conventions
 *  - Plain pointers and sizes only.  All tensor pointers are DEVICE pointers
 *    (HBM) to dense row-major fp32 / int32 arrays unless the name ends in
 *    `_host`.  A grid has shape (H, W): axis 0 (x, H rows) is the slow axis,
 *    axis 1 (y, W columns) is contiguous; a trailing channel axis is fastest
 *    ((H,W,2) "vec2" planes are float pairs, (H,W,3) "vec3" planes triples).
 *  - `scale` = {sx, sy, sz}: cell size along axis 0 / axis 1 and metres per
 *    height unit (erosion.cu:50-51).  2-component scales are {sx, sy}.
 *  - `stream` is a hipStream_t passed as void*; NULL = the null stream.  All
 *    functions are asynchronous on that stream unless stated otherwise.
 *  - Return value: 0 (SOIL_OK) or a negative soil_status; the message of the
 *    last failure on the calling thread is available from soil_last_error().
 *  - No entry point has a CPU fallback.  Without a usable HIP device every
 *    compute call returns SOIL_ERR_NO_DEVICE.
 */
#ifndef SOIL_HIP_H
#define SOIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOIL_HIP_ABI_VERSION 1

/* ------------------------------------------------------------------ status */

typedef enum soil_status {
  SOIL_OK = 0,
  SOIL_ERR_INVALID_ARGUMENT = -1, /* std::invalid_argument in the reference (graph.cu:88) */
  SOIL_ERR_NO_DEVICE = -2,        /* no HIP device / runtime failure at init            */
  SOIL_ERR_HIP = -3,              /* a HIP runtime call failed (message has the detail) */
  SOIL_ERR_OUT_OF_MEMORY = -4,
  SOIL_ERR_IO = -5, /* silt::error::missing_file / an unreadable or unsupported file (tiff.hpp:73) */
  SOIL_ERR_COMM = -6 /* the wire between ranks failed or timed out (soil_slab.h); no reference counterpart */
} soil_status;

/* graph.hpp:11-14  enum edge_t { D4 = 0, D8 = 1 } */
typedef enum soil_edge { SOIL_D4 = 0, SOIL_D8 = 1 } soil_edge;

/* ------------------------------------------------------------------- types */

/* erosion.hpp:17-58  soil::param_t — same fields, same order, same defaults
 * (see soil_param_default).  112 bytes, passed by const pointer, copied into
 * kernel arguments. */
typedef struct soil_param {
  uint64_t maxage;              /* erosion.hpp:20 */
  float lrate;                  /* :21 (read by no kernel) */
  float timeStep;               /* :22 */
  float exitSlope;              /* :25 */
  float uplift;                 /* :26 */
  float rainfall;               /* :27 */
  float gravity;                /* :28 */
  float evapRate;               /* :29 */
  float frictionFactor;         /* :32 */
  float fluvialExponent;        /* :33 */
  float suspensionRateFluvial;  /* :35 */
  float depositionRateFluvial;  /* :36 */
  float suspensionRateDebris;   /* :38 */
  float depositionRateDebris;   /* :39 */
  float landslideRateDebris;    /* :40 */
  float critSlopeBedrock;       /* :43 */
  float critSlopeSediment;      /* :44 */
  float yieldStress;            /* :45 */
  float viscosityWater;         /* :47 */
  float bedShearWater;          /* :48 */
  float densityWater;           /* :49 */
  float viscosityDebris;        /* :51 */
  float bedShearDebris;         /* :52 */
  float densityDebris;          /* :53 */
  float force[2];               /* :56 */
  float _pad;
} soil_param;

/* One element of a `silt::rng` tensor (erosion.hpp:6 uses curandState).
 * The reference's generator is cuRAND XORWOW, closed source and not
 * reproducible off NVIDIA hardware; this ABI fixes a counter-based generator
 * instead: Philox4x32-10 with key = seed, counter =
 * {offset, subsequence}, subsequence = element index — i.e. the same
 * (seed, subsequence = n, offset) addressing as curand_init(seed, n, offset)
 * at graph.cu:100.  One draw advances `offset` by one. */
typedef struct soil_rng {
  uint64_t seed;
  uint64_t offset;
} soil_rng;

/* A row slab of a global (H, W) grid held by one GPU.  Every tensor handed to
 * a *_slab entry point covers local rows [0, rows) == global rows
 * [x0, x0+rows); the kernel writes only local rows [r0, r1) and evaluates
 * boundary conditions (exitSlope, clamp-to-self) against the GLOBAL border.
 * Single-GPU calls use {H, W, 0, H, 0, H}. */
typedef struct soil_domain {
  int64_t H, W;   /* global grid shape                              */
  int64_t x0;     /* global row index of local row 0                */
  int64_t rows;   /* local rows held by every buffer (owned+ghost)  */
  int64_t r0, r1; /* local row range this call computes / owns      */
} soil_domain;

/* ---------------------------------------------------------------- runtime */

int soil_abi_version(void);
const char* soil_last_error(void);
/* Number of visible HIP devices (0 if none); never fails. */
int soil_device_count(void);
int soil_set_device(int device);
/* Writes the gcnArchName of the current device ("gfx950...") into buf. */
int soil_device_name(char* buf, size_t len);
void soil_param_default(soil_param* p);

/* silt tensor storage (un-vendored silt: tensor_t(shape, GPU) allocations,
 * .cpu()/.gpu() copies — call sites graph.cu:80, example/dem_multiflow.py:25). */
int soil_malloc(void** ptr, size_t bytes);
int soil_free(void* ptr);
int soil_memcpy_h2d(void* dst, const void* src_host, size_t bytes, void* stream);
int soil_memcpy_d2h(void* dst_host, const void* src, size_t bytes, void* stream);
int soil_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream);
int soil_stream_synchronize(void* stream);
int soil_device_synchronize(void);

/* HIP-event timing on the stream kernels are launched on (bench.py). */
int soil_event_create(void** event);
int soil_event_destroy(void* event);
int soil_event_record(void* event, void* stream);
int soil_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on stop */

/* silt element-wise ops used by the scripts (silt.set/add/multiply/seed:
 * example/erosion_gpu.py:19, example/dem_process.py:46-47,81; graph.cu:552-553). */
int soil_set_f32(float* t, float value, int64_t n, void* stream);
int soil_set_i32(int32_t* t, int32_t value, int64_t n, void* stream);
int soil_add_f32(float* t, const float* other, int64_t n, void* stream);       /* t += other */
int soil_multiply_f32(float* t, float value, int64_t n, void* stream);         /* t *= value */
int soil_rng_seed(soil_rng* rng, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* Evaluates the library's numerical primitives on device arrays so that tests
 * can compare them with the oracle bit for bit:
 *   op 0: out[i] = expf_(a[i])        stand-in for __expf
 *   op 1: out[i] = log2f_(a[i])
 *   op 2: out[i] = powf_(a[i], b[i])  stand-in for __powf
 *   op 3: out[i] = uniform in (0,1] of stream (seed = bits of a[i],
 *                  subsequence = i, offset = bits of b[i])  — Philox4x32-10
 *   op 4: out[i] = a[i] * b[i] (plain product; exposes denormal flushing)
 *   op 5: out[i] = a[i] / b[i] (the compiler's IEEE division)
 *   op 6: out[i] = quot0(a[i], recip(b[i])), the shared-reciprocal quotient of the
 *                  particle step (soil_math.hpp); equals op 5 on plain operands
 *   op 7: out[i] = expf_flat(a[i]), the branch-free twin of op 0
 *   op 8: out[i] = att_exp(a[i]) = v_exp_f32(a[i] * log2e), the particle attenuations'
 *                  exponential (the reference's __expf, erosion.cu:134-136,346)
 *   op 9: bits of out[i] = floor_cell(a[i]): floor as int32, saturating, NaN -> INT_MAX
 *   op 10: out[i] = sqrt_rn(a[i]), the particle step's square root; equals op 11 (sqrtf) for
 *                  a[i] >= 2^-96, +0, +inf and NaN */
int soil_selftest_math(float* out, const float* a, const float* b, int64_t n, int op,
                       void* stream);

/* --------------------------------------------------- erosion: particle ops */

/* soil::transport_fluvial — erosion.hpp:69-84, erosion.cu:189-239
 * (= __transport_fluvial :29-141 + __normalize_fluvial :143-187), bound at
 * model.cpp:237-268.  Argument names follow erosion.cu:189-204.
 *   layers (H,W,2)  rainfall (H,W)  waterHeight (H,W) out  waterFlux (H,W) inout
 *   mass (H,W) out  massFlux (H,W) inout  velocity (H,W,2) inout
 *   velocityFlux (H,W,2) inout  albedoFlux (H,W,3) inout  albedoSource (H,W,3)
 *   rng [N] inout.
 * The flux planes are only ever added to (atomics) and must be zeroed by the
 * caller.  `albedo_bedrock` is accepted and unused, as in the reference.
 * albedoFlux/albedoSource may both be NULL: the colour channels are then
 * skipped (physics planes are unaffected). */
int soil_transport_fluvial(const float* layers, const float* rainfall, float* waterHeight,
                           float* waterFlux, float* mass, float* massFlux, float* velocity,
                           float* velocityFlux, const float* albedo_bedrock, float* albedoFlux,
                           const float* albedoSource, soil_rng* rng, int64_t N, int64_t H,
                           int64_t W, const float scale[3], const soil_param* param,
                           void* stream);

/* soil::transport_debris — erosion.hpp:86-98, erosion.cu:395-436
 * (= __transport_debris :245-351 + __normalize_debris :353-393), model.cpp:270-295. */
int soil_transport_debris(const float* layers, float* velocity, float* velocityFlux, float* mass,
                          float* massFlux, const float* albedo_bedrock, float* albedoFlux,
                          const float* albedoSource, soil_rng* rng, int64_t N, int64_t H,
                          int64_t W, const float scale[3], const soil_param* param,
                          void* stream);

/* ------------------------------------------------------- erosion: cell ops */

/* soil::mass_transfer — erosion.hpp:104-119, erosion.cu:576-611 (__transfer
 * :453-574), model.cpp:297-328.  delta (H,W,2) inout, layers (H,W,2),
 * uplift/waterHeight/mass/debris (H,W), velocityFluvial/momentumDebris
 * (H,W,2); waterHeight and momentumDebris are accepted and unread, as in the
 * reference.  The four albedo planes (H,W,3) may all be NULL (colour mixing
 * skipped). */
int soil_mass_transfer(float* delta, const float* layers, const float* uplift,
                       const float* waterHeight, const float* mass, const float* velocityFluvial,
                       const float* debris, const float* momentumDebris,
                       const float* albedo_bedrock, const float* albedoFluxFluvial,
                       const float* albedoFluxDebris, float* albedo_surface, int64_t H, int64_t W,
                       const float scale[3], const soil_param* param, void* stream);

/* soil::mass_creep — erosion.hpp:121-126, erosion.cu:712-727 (__mass_creep
 * :633-710), model.cpp:330-341. */
int soil_mass_creep(float* delta, const float* layers, int64_t H, int64_t W,
                    const float scale[3], const soil_param* param, void* stream);

/* soil::layer_merge — erosion.hpp:130-133, erosion.cu:747-757 (__layer_merge
 * :733-745), model.cpp:343-351.  n = H*W cells. */
int soil_layer_merge(float* height, const float* layers, int64_t n, void* stream);

/* Interleave / split the (n,2) layer plane and its two (n) component planes:
 * the legacy map_t kept `height` (bedrock) and `sediment` apart
 * (model.cpp:67-97, commented out) while the live kernels take `layers`
 * (layer_t = vec2, erosion.hpp:60).  `sediment` == NULL reads as zeros in
 * from_planes and is skipped in to_planes. */
int soil_layers_from_planes(float* layers, const float* bedrock, const float* sediment, int64_t n,
                            void* stream);
int soil_layers_to_planes(float* bedrock, float* sediment, const float* layers, int64_t n,
                          void* stream);

/* soil::albedo_stratum / albedo_layer / albedo_discharge — erosion.hpp:139-166,
 * erosion.cu:828-854 / :877-898 / :900-919, model.cpp:353-407. */
int soil_albedo_stratum(float* albedoBedrock, const float* uplift, const float* layers,
                        int64_t n, const float scale[3], const soil_param* param,
                        const float colorA[3], const float colorB[3], float age, float freq,
                        void* stream);
int soil_albedo_layer(float* albedo, const float* albedoBedrock, const float* albedoSediment,
                      const float* layers, int64_t n, float scaleSediment,
                      const float shiftSediment[3], void* stream);
int soil_albedo_discharge(float* albedo, const float* discharge, int64_t n,
                          const float colorDischarge[3], float extinction, float scale,
                          void* stream);

/* ---------------------------------------------- erosion: fused step (slab) */

/* The planes of one erosion model, for the fused step.  All device pointers.
 * `layers`/`layers_next` are the double buffer of the (rows,W,2) layer plane:
 * the step reads `layers`, writes `layers_next`; the caller swaps them. */
typedef struct soil_erosion_planes {
  const float* layers;      /* (rows,W,2) in   bedrock, sediment                         */
  float* layers_next;       /* (rows,W,2) out  layers + delta                            */
  float* height;            /* (rows,W)   out  layer_merge of layers_next (may be NULL)  */
  const float* uplift;      /* (rows,W)   in                                             */
  const float* rainfall;    /* (rows,W)   in   waterSource                               */
  float* waterHeight;       /* (rows,W)   out  "discharge"                               */
  float* waterFlux;         /* (rows,W)   in → re-zeroed  "discharge_track"              */
  float* mass;              /* (rows,W)   out  fluvial suspended mass                    */
  float* massFlux;          /* (rows,W)   in → re-zeroed                                 */
  float* velocity;          /* (rows,W,2) out  fluvial "momentum"                        */
  float* velocityFlux;      /* (rows,W,2) in → re-zeroed                                 */
  float* debris;            /* (rows,W)   out  debris mass                               */
  float* debrisFlux;        /* (rows,W)   in → re-zeroed                                 */
  float* debrisVelocity;    /* (rows,W,2) out                                            */
  float* debrisVelocityFlux;/* (rows,W,2) in → re-zeroed                                 */
} soil_erosion_planes;

/* Fused cell phase of one erosion step: for every owned cell, in one pass,
 *   __normalize_fluvial (erosion.cu:143-187) + __normalize_debris (:353-393)
 *   + [delta = 0] + __transfer (:453-574) + __mass_creep (:633-710)
 *   + layers_next = layers + delta (silt.add, example/dem_process.py:47)
 *   + __layer_merge (:733-745) + re-zero of the five flux planes,
 * bit-identical to running those reference steps one after another (same
 * operation order per cell).  Physics planes only (no albedo).  This is the
 * HBM-roofline kernel: 112 algorithmic bytes per cell. */
int soil_erode_cells_fused(const soil_erosion_planes* planes, const soil_domain* dom,
                           const float scale[3], const soil_param* param, void* stream);
/* The same with flags.  SOIL_CELLS_KEEP_FLUX: the five flux planes are read and left as they are
 * (84 bytes per cell instead of 112); whoever adds to them next must overwrite them first —
 * SOIL_FLUX_OVERWRITE of the particle launches does. */
#define SOIL_CELLS_KEEP_FLUX 1
int soil_erode_cells_fused_ex(const soil_erosion_planes* planes, const soil_domain* dom,
                              const float scale[3], const soil_param* param, int flags, void* stream);

/* Particle halves of transport_fluvial / transport_debris alone (no
 * normalise), on a slab: __transport_fluvial erosion.cu:29-141,
 * __transport_debris :245-351.  Thread n draws its spawn position in the
 * GLOBAL (dom->H, dom->W) grid from rng[n]; only particles whose spawn row
 * lies in global rows [x0+r0, x0+r1) are traced, the rest only advance their
 * rng state, so N ranks that each own one slab trace every particle exactly
 * once.  A trajectory that leaves local rows [0, rows) through an interior
 * (non-global) slab edge is a caller error (size the ghost zone with
 * soil_ghost_rows).
 * `remote0` (device float[8], may be NULL) collects what the reference's "NaN
 * walkers"  deposit into GLOBAL cell (0,0) when
 * that cell is not held by this slab: [0..3] = water, mass, velocity.x/.y flux
 * (fluvial), [4..6] = mass, velocity.x/.y flux (debris).  The owner of global
 * row 0 adds the all-reduced sums to its cell (0,0). */
int soil_particles_fluvial_slab(float* waterFlux, float* massFlux, float* velocityFlux,
                                float* albedoFlux, soil_rng* rng, int64_t N,
                                const float* layers, const float* rainfall,
                                const float* waterHeight, const float* velocity,
                                const float* albedoSource, float* remote0,
                                const soil_domain* dom, const float scale[3],
                                const soil_param* param, void* stream);
int soil_particles_debris_slab(float* massFlux, float* velocityFlux, float* albedoFlux,
                               soil_rng* rng, int64_t N, const float* layers,
                               const float* velocity, const float* albedoSource, float* remote0,
                               const soil_domain* dom, const float scale[3],
                               const soil_param* param, void* stream);
/* Both particle launches of one step (the two calls above) issued together so that
 * they overlap: the debris launch fills the SIMD slots the fluvial launch leaves idle
 * in its sparse late rounds (two internal streams forked from and joined into
 * `stream`).  The reference runs them back to back on ONE rng tensor, each launch
 * consuming two draws per particle; give the debris launch its own tensor seeded two
 * draws further — soil_rng_seed(rng_debris, N, seed, offset + 2) — and every
 * trajectory is the same as in the sequential order.  Planes as in
 * soil_erode_cells_fused (the cell-phase outputs are not touched). */
int soil_particles_pair_slab(const soil_erosion_planes* planes, soil_rng* rng_fluvial,
                             soil_rng* rng_debris, int64_t N, float* remote0,
                             const soil_domain* dom, const float scale[3], const soil_param* param,
                             void* stream);
/* The same with flags.  SOIL_FLUX_OVERWRITE: the flux planes hold stale values on entry (the cell
 * phase ran with SOIL_CELLS_KEEP_FLUX) and hold exactly this call's deposits on return.  The tiled
 * launch shape gets there without a clearing pass: the first round of each launch, whose tiles
 * partition the plane, flushes its LDS accumulators with plain stores — zeros included — instead
 * of read-modify-writes; where that cannot be done (an empty tile, a tile shared by several
 * work-groups, the small-N launch shapes) the planes are cleared first. */
#define SOIL_FLUX_OVERWRITE 1
int soil_particles_pair_slab_ex(const soil_erosion_planes* planes, soil_rng* rng_fluvial,
                                soil_rng* rng_debris, int64_t N, float* remote0,
                                const soil_domain* dom, const float scale[3],
                                const soil_param* param, int flags, void* stream);
/* ------------------------------------------------ erosion: whole steps */

/* One whole erosion step on one device: re-seed the particle streams at
 * (seed, subsequence n, offset step_index * N) — the `silt.seed(rng, seed, step * N)` of
 * example/dem_process.py:81 —, both particle launches, the fused cell phase.  Planes as in
 * soil_erode_cells_fused, single-device shapes (H, W[, 2]); `rng` holds N elements.  Reads
 * planes->layers, writes planes->layers_next: the caller swaps the two handles afterwards.
 * The two particle launches are issued overlapped (as soil_particles_pair_slab does: two
 * internal streams forked from `stream` and joined back into it; the fluvial launch draws
 * from a scratch tensor of the library's workspace, the debris launch from `rng` seeded two
 * draws on), with the results and the final state of `rng` of the sequential order;
 * SOIL_STEP_PAIR=0 in the environment issues them one after the other on `stream`.  The host
 * returns once the step is queued; it does wait, between the rounds of a particle launch, for
 * the word that tells it how many work-groups the next round needs. */
int soil_erode_step(const soil_erosion_planes* planes, soil_rng* rng, int64_t N, uint64_t seed,
                    uint64_t step_index, int64_t H, int64_t W, const float scale[3],
                    const soil_param* param, void* stream);
/* A step inside a chain of steps.  The reference zeroes its track planes between steps
 * (example/dem_process.py: silt.set(track.*, 0)) — 28 bytes of stores per cell that nobody reads.
 * SOIL_STEP_FLUX_OUT_DIRTY: this step leaves the five flux planes holding its accumulated flux
 * (SOIL_CELLS_KEEP_FLUX); SOIL_STEP_FLUX_IN_DIRTY: the previous step did so, this step's particle
 * launches overwrite them (SOIL_FLUX_OVERWRITE).  flags == 0 is soil_erode_step. */
#define SOIL_STEP_FLUX_IN_DIRTY 1
#define SOIL_STEP_FLUX_OUT_DIRTY 2
int soil_erode_step_ex(const soil_erosion_planes* planes, soil_rng* rng, int64_t N, uint64_t seed,
                       uint64_t step_index, int64_t H, int64_t W, const float scale[3],
                       const soil_param* param, int flags, void* stream);

/* The containers of the legacy API (example/erosion_gpu.py:44-71): model_t, the `data` and the
 * `track` buffers.  All float32 device planes of H*W cells ((H,W,2) for the momenta). */
typedef struct soil_erode_model {
  float* height;                /* inout  bedrock surface        (erosion_gpu.py:44-48)  */
  float* sediment;              /* inout  sediment on top of it                           */
  const float* uplift;          /* in                                                     */
  const float* rainfall;        /* in                                                     */
  float* discharge;             /* out    data.discharge = waterHeight (:59-63)           */
  float* mass;                  /* out    data.mass                                       */
  float* momentum;              /* out    data.momentum (H,W,2) = velocity                */
  float* debris;                /* out    data.debris                                     */
  float* debris_momentum;       /* out    data.debris_momentum (H,W,2)                    */
  float* discharge_track;       /* scratch track.* (:65-71): zeroed on entry and on exit  */
  float* mass_track;
  float* momentum_track;        /* (H,W,2) */
  float* debris_track;
  float* debris_momentum_track; /* (H,W,2) */
} soil_erode_model;

/* soil::erode(model, data, track, param[, steps]) — the legacy composite the acceptance script
 * calls (example/erosion_gpu.py:102-106; its binding survives only as a comment,
 * python/source/model.cpp:142): `steps` erosion steps numbered first_step, first_step + 1, ...
 * on the model's planes, in place.  The (H,W,2) layer double buffer and the N particle streams
 * live in the library's workspace for the duration of the call (and stay cached for the next). */
int soil_erode(const soil_erode_model* model, int64_t H, int64_t W, int64_t N, uint64_t seed,
               uint64_t first_step, int steps, const float scale[3], const soil_param* param,
               void* stream);

/* Launch shape of the particle kernels: 0 = auto, 1 = direct (the reference's:
 * thread n = particle n, 5-point stencil gathers), 2 = staged (packed field
 * plane + tile-ordered particles), 3 = tiled (per-tile particle queues, one
 * gather of pre-digested cell terms per step, flux tiles in LDS).  Auto picks
 * tiled for N >= 45000 (grids up to 2^31 cells), staged for N >= 1024, else direct.  All
 * shapes produce the same trajectories and deposits; only the order of the
 * fp32 additions into a cell differs.  For ablation and tests. */
int soil_set_particle_mode(int mode);
/* Arithmetic of the particle step (the loops of erosion.cu:100-139 / :306-349) in the tiled shape:
 * 0 = exact (default): every `/` of the reference's step a correctly rounded IEEE quotient, sqrt
 *     correctly rounded — the walks of the oracle, step for step (the parity tests' contract);
 * 1 = fast: quotients as numerator x v_rcp_f32(denominator), v_sqrt_f32, debris' mass attenuation on the
 *     hardware exponential (what nvcc -use_fast_math makes of the same statements: __fdividef,
 *     sqrt.approx, __expf).  Walks are chaotic in the last bit, so results agree with the exact
 *     mode statistically: plane sums within 2e-3, visited cells within 0.5 %, step counts within
 *     0.5 % (tests/test_fast_particles.py).  9 % less time per 8192^2 step.
 * Launches that carry colour planes and the direct / staged shapes always run exact.
 * SOIL_PARTICLE_DIV=fast in the environment makes 1 the default of the process. */
int soil_set_particle_arith(int mode);
int soil_get_particle_arith(void);
/* Spent debris walkers (tiled shape; the loop of erosion.cu:306-349).  With the reference's example
 * parameters (example/erosion_gpu.py:75-100) a debris walker's two attenuations underflow to exact zeros
 * within two steps (decay_d ~ -1e18, decay_v ~ 1e10) and it walks the rest of its 256 steps adding +-0 to the
 * flux planes.  A walker for which that is certain — att_v == 0, att_d * source_d == 0, its state finite, and
 * every cell of the slab checked by the step's pack pass (excessStress finite and negative at debrisHeight =
 * eps, record finite), launch constants in range: csrc/erosion_particles_tiled.hip, debris_spent — is
 *   1 = retired (default): its walk ends there.  The flux planes hold the same bits as if it had been
 *       walked to the end (x + (+-0) = x); soil_particle_steps counts the steps actually walked.
 *   0 = walked to the end, as the reference does.
 *   2 = watched: marked, walked on, and every deposit of a marked walker that is not an exact zero (and every
 *       marked walker that stops qualifying) counted — soil_debris_retire_violations; the tests want 0.
 * Off in the slab runner's migrate mode (the walker's later cells lie on other ranks) and with colour planes.
 * SOIL_DEBRIS_RETIRE in the environment sets the default of the process. */
int soil_set_debris_retire(int mode);
int soil_get_debris_retire(void);
int soil_debris_retire_violations(uint64_t* total, int reset, void* stream);
/* Ghost rows a slab needs on each interior side so that no trajectory can
 * leave it: ceil(sqrt(2) * maxage) + 2 (one __stepsize step moves a particle
 * by at most sqrt(2) cells, erosion_map.cu:61-76). */
int64_t soil_ghost_rows(const soil_param* param);
/* What of a slab's ghost zone this step's deposits reached: depth[0] = number of rows above the
 * owned local rows [r0, r1) — counted from the boundary — down to the farthest one holding a
 * value other than zero in `plane` ((rows, row_floats) floats), depth[1] = likewise below.
 * Accumulates with max (clear `depth`, two device int32, first; call once per flux plane).  A
 * slab only has to ship that many rows of flux to its neighbour, and next step's particles
 * need the fields refreshed about that deep (soillib_amd/parallel.py).  No counterpart in the
 * single-GPU reference. */
int soil_ghost_extent(int32_t* depth, const float* plane, int64_t rows, int64_t row_floats,
                      int64_t r0, int64_t r1, void* stream);
/* Particle steps (loop iterations of erosion.cu:100 / :306 that pass the loop
 * head) executed by all particle launches on the current device since the last
 * reset; synchronises `stream`.  The reference has no counterpart: it is the
 * numerator of the Mparticle-steps/s that SURVEY.md 8d asks to report, and an
 * exact integer the parity tests compare with the oracle's count. */
int soil_particle_steps(uint64_t* total, int reset, void* stream);

/* ------------------------------------------------------------- flow graphs */

/* soil::direction — graph.hpp:49, graph.cu:246-264 (__direction :201-243), model.cpp:157-159. */
int soil_direction(int32_t* direction, const float* height, int64_t H, int64_t W, int edge,
                   void* stream);
/* soil::steepest — graph.hpp:52, graph.cu:73-91 (__steepest :27-70), model.cpp:169-171. */
int soil_steepest(int32_t* graph, const float* height, int64_t H, int64_t W, int edge,
                  void* stream);
/* soil::random_weighted — graph.hpp:54, graph.cu:175-195 (__seed :97-101,
 * __random_weighted :103-173), model.cpp:173-175.  Stateless: cell n draws its one uniform in (0, 1]
 * from the Philox4x32-10 block (key seed; counter {offset, n >> 2}), word n & 3 — the reference's
 * curand_init(seed, n, offset) + one curand_uniform per cell, with one block serving four cells. */
int soil_random_weighted(int32_t* graph, const float* height, int64_t H, int64_t W, int edge,
                         uint64_t seed, uint64_t offset, float T, void* stream);
/* soil::slope — graph.hpp:63, graph.cu:297-311 (__slope :270-295), model.cpp:161-163. */
int soil_slope(float* slope, const float* tensor, const int32_t* flow, int64_t H, int64_t W,
               const float scale[2], void* stream);
/* soil::accumulate / accumulate_decay — graph.hpp:57-60, graph.cu:578-593
 * (__accumulate :526-576: __donor :321-348, __count :350-380, my_decay
 * :382-420, __rake_compress :429-522), model.cpp:181-187.  `decay` == NULL
 * selects accumulate (scalar decay 1).  Scratch comes from a cached
 * per-device workspace, not from per-call allocations.  Synchronises the
 * stream before returning, like the reference (graph.cu:564). */
int soil_accumulate(float* out, const int32_t* graph, const float* source, const float* decay,
                    int64_t H, int64_t W, int edge, void* stream);
/* The realisation loop of example/dem_multiflow.py:43-49 as one call, without the
 * per-realisation trip through host memory: for k = k_first, k_first+k_stride, ... < k_end
 *   sum += double(float(accumulate(random_weighted(height, edge, seed, k, T), source) / K))
 * `sum` (H*W doubles) is accumulated into — zero it first.  A rank of an N-GPU run
 * passes k_first = rank, k_stride = N and all-reduces `sum` afterwards: accumulation
 * does not shard (pointer jumps span the grid), the realisations do. */
int soil_multiflow(double* sum, const float* height, const float* source, int64_t H, int64_t W,
                   int edge, uint64_t seed, uint64_t k_first, uint64_t k_stride, uint64_t k_end,
                   uint64_t K, float T, void* stream);
/* Depression filling before flow routing (BASELINE config 3).  The reference has
 * none — example/dem_condition.py:35-41 calls the third-party pysheds — so this is
 * build-defined, parity unpinned: out(c) = the lowest level at which
 * cell c can drain to an outlet (a step off the grid or onto a NaN cell) along `edge`
 * connectivity, i.e. the priority-flood surface; NaN cells stay NaN.  Exact in fp32
 * (only min/max): tile relaxation in LDS, started from the recursively filled 4x coarser
 * level (csrc/conditioning.hip); synchronises the stream. */
int soil_fill_depressions(float* out, const float* height, int64_t H, int64_t W, int edge,
                          void* stream);
/* Scratch memory.  The reference allocates its scratch per call (graph.cu:539-550, :182-183;
 * path.cu:195; filter.cu:77); this library keeps one cached block per host thread, device and
 * purpose (accumulate, the particle launches of either kind, fill_depressions, soil_erode) and
 * grows it on demand.  Several host threads may drive one device at the same time, each on its
 * own stream: they share no scratch (nor streams, events or pinned words of the launches, which
 * are per thread too).  Calls of ONE thread that use the same block must not overlap in time —
 * they do not, since a thread's calls are ordered on the streams it passes.  A call that finds its
 * block too small synchronises the device before replacing it.
 * soil_workspace_release frees all cached blocks of the current device, of every thread: call it
 * while no other thread is inside the library. */
int soil_workspace_release(void);

/* ---------------------------------------------------------------- stencils */

/* soil::gradient — grad.hpp:11, grad.cu:89-97 (__gradient :22-87), model.cpp:193-195.  out (H,W,2). */
int soil_gradient(float* out, const float* in, int64_t H, int64_t W, const float scale[2],
                  void* stream);
/* soil::negslope — grad.hpp:17, grad.cu:133-141 (__negslope :101-131), model.cpp:201-203. */
int soil_negslope(float* out, const float* in, int64_t H, int64_t W, const float scale[2],
                  void* stream);
/* soil::laplacian — grad.hpp:14, grad.cu:186-206 (__laplacian<D> :147-183), model.cpp:197-199.
 * in/out (H,W,D), D in {1,2}. */
int soil_laplacian(float* out, const float* in, int64_t H, int64_t W, int D,
                   const float scale[2], void* stream);
/* soil::gaussian_blur — filter.hpp:11, filter.cu:72-91 (__blur :59-70,
 * __gaussian_blur :24-56), model.cpp:189-191.  tensor (H,W,C), C in {1,2}, is
 * blurred IN PLACE (the reference returns its input handle, filter.cu:90);
 * scratch (H,W,C) is the intermediate of the axis-0 pass. */
int soil_gaussian_blur(float* tensor, float* scratch, int64_t H, int64_t W, int C, float sigma,
                       void* stream);
/* soil::op::normal — normal.hpp:19-39 (CPU loop in the reference; unbound in
 * its module, used by example/tiff_normal.py:14).  out (H,W,3). */
int soil_normal(float* out, const float* in, int64_t H, int64_t W, const float scale[3],
                void* stream);
/* Host twin of soil_normal for CPU tensors (the reference's only placement). */
int soil_normal_host(float* out_host, const float* in_host, int64_t H, int64_t W,
                     const float scale[3]);

/* -------------------------------------------------------- path-integral MC */

/* soil::solve_uniform — path.hpp:30-37, path.cu:180-219 (__solve_uniform<K>
 * :52-139, __normalize<K> :142-170; bilinear gather sample.hpp:154-186),
 * model.cpp:209-227.  flow (H,W,2), source/flux (H,W,K), K in {1,2}, decay
 * (H,W), rng [N].  flux is zeroed, filled and normalised; synchronises. */
int soil_solve_uniform(float* flux, const float* flow, const float* source, const float* decay,
                       soil_rng* rng, int64_t N, int64_t H, int64_t W, int K,
                       const float scale[2], uint64_t count, void* stream);

/* ------------------------------------------------------------------- noise */

/* soil::noise_param_t / soil::noise — noise.hpp:14-56, model.cpp:413-421:
 * OpenSimplex2 FBm, 3-D sample at (x/ext0, y/ext1, seed). */
typedef struct soil_noise_param {
  float frequency;  /* noise.hpp:29, default 1    */
  int32_t octaves;  /* :30, default 8             */
  float gain;       /* :31, default 0.6           */
  float lacunarity; /* :32, default 2             */
  float seed;       /* :33, default 0 (z coord)   */
  float ext[2];     /* :34, default {512, 512}    */
} soil_noise_param;
void soil_noise_param_default(soil_noise_param* p);
/* Device generator (bench inputs at 8192^2+) and host twin (the reference's
 * placement, noise.hpp:49-52); both produce identical bits. */
int soil_noise(float* out, int64_t H, int64_t W, const soil_noise_param* p, void* stream);
int soil_noise_host(float* out_host, int64_t H, int64_t W, const soil_noise_param* p);
/* Rows [x0, x0+rows) of the same heightmap (the slab a rank owns). */
int soil_noise_window(float* out, int64_t rows, int64_t W, int64_t x0, const soil_noise_param* p,
                      void* stream);

/* ------------------------------------------------------ multiscale driver */
/* soil.resize(dst, src, newres, oldres) of example/erosion_gpu_multiscale.py:104-141
 * (SURVEY.md 8f row 4).  The reference snapshot has no definition of it; this
 * one is bilinear resampling at corner-aligned positions (equal resolutions give
 * the identity, corners are kept), for planes of D = 1..3 interleaved channels.
 * Parity unpinned. */
int soil_resize(float* dst, const float* src, int64_t Hn, int64_t Wn, int64_t Ho, int64_t Wo, int D,
                void* stream);

/* ------------------------------------------------------- TIFF / GeoTIFF IO */
/* Host-side file IO of the callers either side of the path:
 * soil::io::tiff (io/tiff.hpp:20-241) and soil::io::geotiff (io/geotiff.hpp:63-318),
 * bound in python/source/io.cpp:20-100.  The reference delegates the format to
 * libtiff (third party, not in its tree); this is a codec of its own for
 * single-band rasters: classic + BigTIFF, either byte order, strips or tiles,
 * compression none / LZW / Deflate / PackBits, predictors 1-3.  No GPU needed. */
#define SOIL_TIFFTAG_GEOPIXELSCALE 33550   /* geotiff.hpp:13-21 */
#define SOIL_TIFFTAG_GEOTIEPOINTS 33922
#define SOIL_TIFFTAG_GEOKEYDIRECTORY 34735
#define SOIL_TIFFTAG_GEODOUBLEPARAMS 34736
#define SOIL_TIFFTAG_GEOASCIIPARAMS 34737
#define SOIL_TIFFTAG_GDAL_METADATA 42112
#define SOIL_TIFFTAG_GDAL_NODATA 42113

typedef struct soil_tiff_info { /* what tiff::peek / geotiff::peek learn (tiff.hpp:69-99) */
  uint32_t width, height;       /* ImageWidth, ImageLength                          */
  uint32_t bits;                /* BitsPerSample                                    */
  uint32_t sample_format;       /* 1 unsigned, 2 signed, 3 IEEE float               */
  uint32_t samples;             /* SamplesPerPixel                                  */
  uint32_t tiled, tile_width, tile_height;
  uint32_t compression, predictor;
  /* element counts of the GeoTIFF / GDAL tags present (0 = absent), for soil_tiff_tag */
  uint32_t n_scale, n_tiepoints, n_params, n_keydir, n_ascii, n_metadata, n_nodata;
} soil_tiff_info;

typedef struct soil_geotiff_tags { /* geotiff::meta_t as written by geotiff::write (:199-213) */
  const double* scale;     uint32_t n_scale;      /* GeoPixelScale   */
  const double* tiepoints; uint32_t n_tiepoints;  /* GeoTiePoints    */
  const double* params;    uint32_t n_params;     /* GeoDoubleParams */
  const int16_t* keydir;   uint32_t n_keydir;     /* GeoKeyDirectory */
  const char* ascii;       /* GeoAsciiParams, NUL-terminated or NULL */
  const char* metadata;    /* GDAL_METADATA                          */
  const char* nodata;      /* GDAL_NODATA                            */
} soil_geotiff_tags;

/* tiff::peek + geotiff::peek: SOIL_ERR_IO when the file is missing (the
 * reference throws silt::error::missing_file) or is not a TIFF. */
int soil_tiff_peek(const char* filename, soil_tiff_info* info);
/* Payload of one tag: doubles (8 B each), shorts (2 B) or the raw bytes of an
 * ASCII tag including its NUL.  *written_bytes = 0 when the tag is absent. */
int soil_tiff_tag(const char* filename, int tag, void* dst, uint64_t capacity_bytes,
                  uint64_t* written_bytes);
/* tiff::read (tiff.hpp:102-213): width*height samples in scanline order into
 * `dst`: float64 when the file holds 64-bit samples, float32 otherwise
 * (tiff.hpp:116-124).  Integer and half-float samples are converted to
 * float32 (the reference leaves the 16-bit case unfilled). */
int soil_tiff_read(const char* filename, void* dst, uint64_t dst_bytes);
/* tiff::write / geotiff::write (tiff.hpp:215-241, geotiff.hpp:183-226):
 * uncompressed little-endian IEEE-float strips, ROWSPERSTRIP = width (what
 * TIFFDefaultStripSize(tif, width) returns); `geo` may be NULL.  Images beyond
 * 4 GiB are written as BigTIFF (libtiff would fail). */
int soil_tiff_write(const char* filename, const void* data, uint32_t width, uint32_t height,
                    uint32_t bits, const soil_geotiff_tags* geo);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* SOIL_HIP_H */
