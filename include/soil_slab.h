/* soil_slab.h — the sharded erosion step behind the C ABI.
 *
 * The reference runs on one GPU only (one default-stream launch per kernel,
 * /root/reference/source/soillib/model/path/erosion.cu:209, :413); BASELINE.json configs[4] asks
 * for the 16384^2 grid cut into row slabs across the 8 GPUs of a node.  This header is that step
 * as a library object: one `soil_slab` per rank (= per GPU) owns a slab of rows plus its ghost
 * rows, `soil_slab_step` advances the GLOBAL grid by one erosion step — same trajectories and
 * deposits as soil_erode_step on the whole grid (soil_hip.h), only the fp32 summation order of the
 * flux differs.
 *
 * Two small tables of function pointers keep the step's host logic independent of where it runs:
 *   soil_comm      the wire between ranks.  soil_comm_rccl_* (below) = RCCL over xGMI
 *                  (ncclGroupStart .. ncclSend/ncclRecv .. ncclGroupEnd on the runner's streams);
 *                  tests plug in gloo or an in-process mailbox.
 *   soil_slab_ops  the compute back-end.  NULL = this library's HIP kernels on the current device;
 *                  the CPU tests plug in the oracle, so that the very same C++ exchange schedule is
 *                  exercised by world-size-2/3 gloo runs without a GPU.
 * Schedule, halo trimming by measured reach and the repeat-launch fallback: DESIGN.md 5.
 */
#ifndef SOIL_SLAB_H
#define SOIL_SLAB_H

#include "soil_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ the wire */

typedef struct soil_xfer {
  void* ptr;     /* device (or back-end) memory */
  int64_t bytes;
  int32_t peer;  /* rank */
} soil_xfer;

#define SOIL_COMM_HOST_ORDERED 1 /* the wire does not follow streams: the runner synchronises its
                                    back-end before every call and the call blocks until done */

typedef struct soil_comm {
  void* ctx;
  int32_t rank, world;
  int32_t flags;
  /* All sends and receives of one call form one group (ncclGroupStart/End).  Stream-ordered on
   * `stream` (one of the runner's two streams) unless SOIL_COMM_HOST_ORDERED. */
  int (*exchange)(void* ctx, const soil_xfer* sends, int32_t n_sends, const soil_xfer* recvs,
                  int32_t n_recvs, void* stream);
  /* in-place sum over all ranks of `n` floats in back-end memory */
  int (*all_reduce_sum_f32)(void* ctx, float* buf, int64_t n, void* stream);
  int (*barrier)(void* ctx);
  /* May be NULL.  SOIL_OK, or the failure of a wire that has given up meanwhile (a stream-ordered
   * transfer that did not complete within its timeout and was aborted): the runner asks after every
   * wait on its streams, so that a step never returns planes a dead wire left half-filled. */
  int (*status)(void* ctx);
} soil_comm;

/* RCCL communicator owned by the library.  The 128-byte id is made by one rank
 * (soil_comm_rccl_unique_id) and handed to the others by whatever launched the job (bench.py:
 * torch.distributed's store; an MPI program: MPI_Bcast); every rank then calls
 * soil_comm_rccl_create with its rank on the device it has made current (soil_set_device).
 * librccl is looked up at run time (an already loaded copy first — PyTorch ships one —, then
 * librccl.so.1, or the path in SOIL_RCCL_LIB): a single-GPU user never needs it. */
int soil_comm_rccl_unique_id(uint8_t id[128]);
/* binds librccl and reports ncclGetVersion: no device touched, no thread or socket started — what a
 * rank calls to prove that it can enter the collective soil_comm_rccl_create at all */
int soil_comm_rccl_probe(int32_t* version);
int soil_comm_rccl_create(soil_comm** out, const uint8_t id[128], int32_t rank, int32_t world);
int soil_comm_rccl_destroy(soil_comm* comm);
/* what RCCL reports for the communicator: ncclCommCount, ncclCommUserRank, ncclCommCuDevice */
int soil_comm_rccl_info(const soil_comm* comm, int32_t* count, int32_t* rank, int32_t* device);
/* which librccl the library bound (the file the symbols came from, NUL-terminated into path[capacity])
 * and its ncclGetVersion code (major * 10000 + minor * 100 + patch); no device touched */
int soil_comm_rccl_library(char* path, int32_t capacity, int32_t* version);
/* Bounded waits.  Every RCCL communicator has a watchdog thread: a library call that does not return,
 * or a transfer on a stream that does not complete, within SOIL_RCCL_TIMEOUT_S seconds (default 30;
 * ncclCommInitRank: SOIL_RCCL_INIT_TIMEOUT_S, default 120) makes it call ncclCommAbort — the blocked
 * call and the stream come back — and every later call on the communicator, soil_slab_step and
 * soil_slab_sync return SOIL_ERR_COMM; soil_last_error() names the operation, the first peer, the
 * byte count, the rank and the librccl file.  A hung wire fails; it does not hang the host. */
/* a wire whose every operation blocks like a transfer whose peer never shows up, under the same
 * watchdog (timeout_s): what the tests use to prove that the runner surfaces the failure */
int soil_comm_wedged_create(soil_comm** out, int32_t rank, int32_t world, double timeout_s);
int soil_comm_wedged_destroy(soil_comm* comm);
/* a one-rank world that needs no library at all (exchange with oneself: device copies) */
int soil_comm_self_create(soil_comm** out);
int soil_comm_self_destroy(soil_comm* comm);

/* --------------------------------------------------------- compute back-end */

typedef struct soil_slab_ops {
  void* ctx;
  int (*alloc)(void* ctx, void** out, int64_t bytes); /* zero-filled */
  int (*release)(void* ctx, void* p);
  /* lane 0: the step's main stream, lane 1: its communication stream */
  int (*fill_f32)(void* ctx, float* dst, float value, int64_t n, int32_t lane);
  int (*add_f32)(void* ctx, float* dst, const float* src, int64_t n, int32_t lane);
  int (*rng_seed)(void* ctx, soil_rng* rng, int64_t N, uint64_t seed, uint64_t offset);
  int (*particles_fluvial)(void* ctx, const soil_erosion_planes* planes, soil_rng* rng, int64_t N,
                           float* remote0, const soil_domain* dom, const float scale[3],
                           const soil_param* param);
  int (*particles_debris)(void* ctx, const soil_erosion_planes* planes, soil_rng* rng, int64_t N,
                          float* remote0, const soil_domain* dom, const float scale[3],
                          const soil_param* param);
  /* both launches overlapped (soil_particles_pair_slab); NULL: the back-end has none */
  int (*particles_pair)(void* ctx, const soil_erosion_planes* planes, soil_rng* rng_fluvial,
                        soil_rng* rng_debris, int64_t N, float* remote0, const soil_domain* dom,
                        const float scale[3], const soil_param* param);
  /* fused cell phase on the local rows [dom->r0, dom->r1) */
  int (*cells)(void* ctx, const soil_erosion_planes* planes, const soil_domain* dom,
               const float scale[3], const soil_param* param);
  /* soil_ghost_extent of one plane, max-accumulated into the two host ints; blocks (lane 0) */
  int (*ghost_extent)(void* ctx, const float* plane, int64_t rows, int64_t row_floats, int64_t r0,
                      int64_t r1, int32_t depth[2]);
  int (*noise_rows)(void* ctx, float* out, int64_t rows, int64_t W, int64_t x0,
                    const soil_noise_param* p);
  int (*layers_from_bedrock)(void* ctx, float* layers, const float* bedrock, int64_t n);
  int (*to_host)(void* ctx, void* dst_host, const void* src, int64_t bytes);   /* blocks, lane 0 */
  int (*from_host)(void* ctx, void* dst, const void* src_host, int64_t bytes); /* lane 0 */
  int (*fork)(void* ctx);  /* lane 1 waits for everything queued on lane 0 so far */
  int (*join)(void* ctx);  /* lane 0 waits for lane 1 */
  int (*sync)(void* ctx);  /* the host waits for both lanes */
  void* (*stream)(void* ctx, int32_t lane);
  /* SOIL_SLAB_MIGRATE (below): one launch of `kind` (0 fluvial, 1 debris) that ADDS to the kind's flux
   * planes — from the streams' spawns (`inbox` NULL) or from `n_in` walkers handed over by the neighbours
   * (64-byte records, taken as they are; at most N of a kind per launch).  A walker that steps off the
   * rows the launch was given — the owned rows plus the shallow halo either side, i.e. off local rows
   * [0, dom->rows) of a slab with neighbours there —, in the grid and with life left, is written to
   * `out_up` / `out_down` (room for `cap` records each; a record beyond `cap` is counted, not written) at the
   * top of that iteration, state untouched; out_count[0..1] (back-end memory, zeroed by the caller)
   * count them.  kind 2: both kinds' spawn launches overlapped (soil_particles_pair_slab; the debris
   * launch draws from `rng_debris`); fluvial records go to the first cap / 2 slots of the boxes, debris
   * records to the second half, out_count[0..3] = fluvial up, down, debris up, down.  kind 2 with an
   * `inbox`: both kinds' handed-over walkers walked on side by side — the inbox holds the fluvial records
   * first, then the debris ones, `n_in` = fluvial count | debris count << 32 (both > 0); boxes and counts
   * as for kind 2.
   * NULL: the back-end has no such launch and the mode is refused. */
  int (*particles_pass)(void* ctx, int32_t kind, const soil_erosion_planes* planes, soil_rng* rng,
                        soil_rng* rng_debris, int64_t N, float* remote0, const soil_domain* dom, const float scale[3],
                        const soil_param* param, const void* inbox, int64_t n_in, void* out_up,
                        void* out_down, uint32_t* out_count, int64_t cap);
} soil_slab_ops;

/* ------------------------------------------------------------ the slab runner */

typedef struct soil_slab soil_slab; /* opaque */

/* How a walk that crosses a slab's edge is served (soil_slab_config.mode).
 * SOIL_SLAB_DEEP_HALO  every rank carries ceil(sqrt(2) maxage) + 2 ghost rows a side and walks its own
 *                      walkers to their end; rows of flux and fields travel, trimmed to the measured
 *                      reach (the default; SURVEY.md 8e's option A made exact by the halo's depth).
 * SOIL_SLAB_MIGRATE    a shallow halo (64 ghost rows a side, SOIL_MIGRATE_HALO); a walker that reaches its far
 *                      end is handed over as the 64-byte record the tiled transport parks it as anyway, the
 *                      neighbour walks it on in a further launch of the same step (SURVEY.md 8e's option
 *                      B).  What travels: the walkers that cross (a few per cent, 64 B each) and 64 rows of
 *                      flux and fields; 6 % ghost rows at 2048-row slabs.  What it
 *                      costs: a host look at two counters and an all-reduce per pass, one or two short
 *                      launches per kind and step for the immigrants.  Same walks either way. */
#define SOIL_SLAB_DEEP_HALO 0
#define SOIL_SLAB_MIGRATE 1

typedef struct soil_slab_config {
  int64_t rows_per_rank; /* S: rows every rank owns; the global grid is (world * S) x W */
  int64_t W;
  int64_t particles_div; /* N = H * W / particles_div particles of the GLOBAL grid per launch */
  uint64_t seed;
  float scale[3];        /* all zero: (20 / H, 20 / W, 4) */
  float noise_seed;      /* initial bedrock = soil.noise rows of the global heightmap (init != 0) */
  int64_t noise_rows;    /* row extent the noise is normalised by; 0: H */
  int32_t init;          /* 0: planes stay zero (the caller fills them through soil_slab_plane) */
  int32_t trim;          /* halos trimmed to the measured reach: 1 / 0; -1: on for world > 1 unless
                            SOIL_HALO_FULL=1 */
  int32_t pair;          /* particle launches overlapped: 1 / 0; -1: on unless SOIL_STEP_PAIR=0 (as
                            soil_erode_step) */
  int32_t halo_need;     /* > 0: ghost rows to refresh whatever the history says (tests: a
                            prediction that is too small on purpose); 0: predicted; also SOIL_HALO_NEED.
                            Must be the same on every rank (checked by soil_slab_create) */
  int32_t mode;          /* SOIL_SLAB_DEEP_HALO / SOIL_SLAB_MIGRATE; -1: SOIL_SLAB_MODE=migrate in the
                            environment, else deep halo */
} soil_slab_config;

typedef struct soil_slab_info {
  int64_t H, W, S, G;          /* global rows, columns, owned rows, ghost rows per interior side */
  int64_t x0, rows, r0, r1;    /* global row of local row 0, local rows, owned local rows [r0, r1) */
  int64_t N;
  uint64_t step_index;
  int32_t rank, world, trim, pair;
  int64_t rows_flux, rows_field, rows_full; /* plane-rows shipped so far, and what the bound asks */
  int64_t repeated_launches;
  int32_t reach_hist[4];       /* max reach over all ranks, last steps (0: none yet) */
  int32_t n_reach;
  int64_t rows_window, rows_window_full; /* ghost rows the particle launches were given so far (the rows with
                                            fresh fields + 2, SOIL_HALO_WINDOW=0: all), and the bound's */
  int64_t passes, walkers_handed;        /* migrate mode: launches of either kind so far (>= 2 per step), walkers
                                            this rank handed to its neighbours so far */
  int32_t mode, reserved;
} soil_slab_info;

/* marks of one step for a timing harness (bench.py records a HIP event per mark on lane 0):
 * 0 seeded | 1 fluvial launch queued | 2 both launches queued | 4 flux halo complete |
 * 5 interior cell rows queued | 3 step queued */
typedef void (*soil_slab_mark_fn)(void* ctx, int32_t mark);

/* `comm` and `ops` (if given) must outlive the runner; `ops` == NULL: HIP on the current device. */
int soil_slab_create(soil_slab** out, const soil_slab_config* cfg, const soil_param* param,
                     const soil_comm* comm, const soil_slab_ops* ops);
int soil_slab_step(soil_slab* slab, soil_slab_mark_fn mark, void* mark_ctx);
/* A plane of the slab by the names of soil_erosion_planes ("layers" is the current one): local
 * rows incl. ghost rows, `channels` floats per cell.  The five flux planes are scratch of a step:
 * with the HIP back-end they are not re-zeroed behind the cell phase (the next step's launches
 * overwrite them; SOIL_SLAB_LAZY=0 restores the zeros), so between two steps their OWNED rows hold
 * the flux the last one consumed; their ghost rows have no stated content. */
int soil_slab_plane(soil_slab* slab, const char* name, float** data, int64_t* rows,
                    int64_t* channels);
int soil_slab_get_info(const soil_slab* slab, soil_slab_info* info);
int soil_slab_sync(soil_slab* slab);
/* the back-end's stream of a lane (0 main, 1 communication) as a hipStream_t; NULL for a back-end
 * without streams */
int soil_slab_stream(soil_slab* slab, int32_t lane, void** stream);
int soil_slab_destroy(soil_slab* slab);
/* the HIP back-end by itself (what `ops` == NULL uses), for callers that wrap it */
int soil_slab_ops_hip_create(soil_slab_ops** out);
int soil_slab_ops_hip_destroy(soil_slab_ops* ops);
/* rows a rank holds: (x0, rows, r0, r1) of rank `rank` */
void soil_slab_layout(int32_t rank, int32_t world, int64_t S, int64_t G, int64_t out[4]);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* SOIL_SLAB_H */
