// erosion_particles_tiled.hip — the MI355X launch shape of the particle
// transport (__transport_fluvial erosion.cu:29-141, __transport_debris :245-351).
//
// Why: with one lane per streamline and global gathers/atomics every step, the
// phase is bound by fp32 atomics into L2 (measured 22.7 G atomics/s,
// tools/microbench/l2_atomic.hip; 8.6 G of them per fluvial launch at 8192^2)
// and by random 64-byte HBM traffic with no reuse (profiles/r01_first: ~1.1 TB
// per fluvial launch).  Every step only needs data of the cell the particle
// stands on, and every cell is visited ~30 times per launch, so the work is
// re-organised around cells:
//
//   * a streaming pre-pass evaluates every cell-only sub-expression of the loop
//     body once per cell and packs it into one float4 plane (k_tiled_pack), so a
//     step gathers 16 bytes, through L1/L2;
//   * the grid is cut into 64x64-cell tiles; a tile's flux accumulators (4
//     planes fluvial / 3 debris, 64 / 48 KiB) live in the LDS of one work-group,
//     which leaves room for 2 / 3 work-groups per CU (24 waves);
//   * particles are kept in per-tile queues of 64-byte records.  A work-group
//     advances the particles of its tile step by step (deposits = split-phase
//     compare-and-swap on LDS, see CasDeposit) until they die, step onto
//     another tile or use up the round's step budget; survivors are written
//     back into the slots their queue occupied together with the queue section
//     they are bound for (tile x expected residence time, queue_key) and their
//     rank in it, the 4-byte slot indices are sorted by section without atomics,
//     and the next round resumes them;
//   * between two rounds k_queue_prepare scans the section counts, orders the
//     tiles longest queue first, lists the work-groups of the round (none for an
//     empty tile, several for a queue longer than the chip's share) and hands the
//     host the queue total and the step counter through pinned memory;
//   * at the end of a round the tile's flux is added to the global planes with
//     non-atomic read-modify-writes of the cells that changed (atomic adds when
//     several work-groups share the tile);
//   * once rounds advance particles more slowly than the finishing launch would
//     (measured rate of the last round), one last launch walks what is left to
//     the end against global memory.
//
// A particle executes exactly the instruction sequence of the reference loop —
// state is only ever parked at the top of an iteration, before `++iter` — so
// every trajectory and every deposit is bit-identical to the direct launch
// shape; only the order of the fp32 additions into a cell differs.
//
// Measured at 8192^2, N = 8.4 M, maxage 256 (ms per launch, fluvial / debris):
// direct 418 / 115; fields+flux in LDS with ds_add_f32 84 / 36; this file 23.9 / 11.0.
// A fluvial step is ~160 vector + ~110 scalar and branch instructions (seven IEEE quotients over
// four denominators, a square root, three hardware exponentials, see step_geom / step_apply), mostly
// one depending on the other; the vector pipes are issuing on ~40 % of all SIMD cycles
// (profiles/particle_roofline.json).  What keeps them from more: a wave issues a dependent
// instruction every ~11 cycles, 16 B of LDS per cell cap a CU at ~1280 walkers in flight, a
// work-group holds its tile until its longest walker has taken the round's steps.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <thread>
#include <vector>

#include "particles_common.hpp"
#include "window.hpp"

namespace soil {

// A tile is TR rows x TC columns of cells (TC a power of two; TR is whatever fills the LDS of
// a CU with a whole number of tiles).  The acceptance workload spawns one particle per 8 cells
// (SURVEY 8d): a 64x64 tile starts with ~512 of them.
// TR, log2(TC) and the origin of the tile grid: tile (0,0) starts at local cell
// (-off_r, -off_c).  Odd rounds shift the grid by half a tile in both directions, so
// a walker that zig-zags along a tile edge — parked after a step or two, round after
// round — finds itself in the middle of a tile every other round.
struct TileShape { int tr, shift_c, off_r, off_c; };
constexpr uint32_t kNoTile = 0xffffffffu;     // dest[] of an empty record slot

enum Kind { FLUVIAL = 0, DEBRIS = 1 };

struct alignas(16) PRec {  // parked particle, 64 bytes
  float px, py, spx, spy;
  float a0, a1, a2, s0;  // fluvial: att_w att_m att_v source_w | debris: att_d att_v - source_d
  float s1, svx, svy;    // fluvial: source_m, source_v        | debris: -, source_v
  int32_t iter;
  uint32_t ind;          // global flat index of the last cell deposited into (erosion.cu:60,105)
  float sa0, sa1, sa2;   // colour source = source_m | source_d times albedoSource[spawn cell] (:91 / :299)
};
static_assert(sizeof(PRec) == 64, "PRec must be one 64-byte line");

struct TiledHostWord {  // pinned, device-mapped
  uint32_t live;    // particles queued for the round
  uint32_t blocks;  // work-groups the round needs (entries of the block list)
  unsigned long long steps;
  uint32_t seq;     // written last: the number of the k_queue_prepare launch that filled the word
  uint32_t whole;   // 1: one work-group per tile and no tile empty (the round's flush may store, see `store_all`)
  uint32_t mode;    // TiledCtl::mode after this scan
  uint32_t stop_round;  // the round at which mode left 0 (valid when mode != 0)
  // `live` of the scan with sequence number q, at [q % kLiveRing].  The host runs up to `depth` rounds
  // ahead of the words it has read, so by the time it looks at the word of scan r the device may have
  // written the words of scans r + 1 and r + 2 over it — and `live` of a LATER scan is too small a
  // bound for the grids of the rounds the host queues next (a slot sort or a round kernel launched with
  // fewer work-groups than the round has: walkers dropped).  Found by the eight-process run of
  // BASELINE config 5 on one GPU, where the hosts lag behind a time-sliced device (round 4; the single
  // `live` above was read since round 3's run-ahead).  One scan in kLiveRing writes a slot.
  static constexpr uint32_t kLiveRing = 16;
  uint32_t live_ring[kLiveRing];
};

// What the DEVICE decides between two rounds, and what the kernels of the following rounds look at
// before they do anything.  The host queues rounds ahead of the words it has seen (TiledRun::advance);
// whether such a round still has work is settled here: the scan of round r counts the queues, decides
// "another round" / "the finishing launch takes over" / "nothing left" — by the number of live
// particles and by the rate of the round just done (steps per second between two scans, the
// constant-rate realtime counter) — and every scan, slot sort and round kernel queued behind a
// decision to stop returns at once.
struct TiledCtl {
  uint32_t mode;    // 0: rounds go on, 1: the finishing launch takes over, 2: nothing left
  uint32_t blocks;  // work-groups of the round the last scan prepared
  uint32_t slots;   // record slots the round before (or the spawn) filled: what the slot sort and the finishing launch look at
  uint32_t live;    // particles the last scan queued
  unsigned long long steps_prev, t_prev;  // step counter and realtime clock at the last scan
  uint32_t stop_round;
  uint32_t done;        // work-groups of the running round that have finished (its last one scans)
  uint32_t started;     // ... that have started (the last one to start opens the gate of the other launch, PairGate)
  // blocks of round r in slot r & 1: the scan at the tail of round r writes the other slot, which no
  // work-group of round r reads — stragglers beyond the round's count may still be arriving then
  uint32_t blocks_of[2];
  // ... and of the sparse tiles of round r (k_tiled_round<..., SPARSE>): the entries of the block list
  // behind the blocks_of[r & 1] dense ones
  uint32_t sparse_of[2];
  // ... and how many of those — the last ones — hold fewer than 16 walkers: the sparse kernel takes
  // them four to a wave (0: one tile per wave throughout)
  uint32_t sparse_tiny[2];
  uint32_t sparse_pair[2];  // ... and in front of those, the ones of 16 .. 31 walkers: two to a wave
  uint32_t retire;          // debris: spent walkers end their walk (1) / are watched (2), see debris_spent; written by the spawn
};
struct ScanRule {  // when the rounds stop (TiledRun::setup)
  uint32_t round, tail, max_round;
  float ticks_per_step_max;  // a round slower than this many realtime ticks per particle step hands over
};
__device__ __forceinline__ unsigned long long realtime_ticks() { return wall_clock64(); }
// thread 0 of a scan: the decision for round `rule.round`, given its `total` queued particles
__device__ __forceinline__ uint32_t scan_decide(TiledCtl* ctl, const ScanRule& rule, uint32_t total,
                                                unsigned long long steps_now) {
  const unsigned long long now = realtime_ticks();
  uint32_t mode = 0;
  if (total == 0) {
    mode = 2;
  } else if (rule.round > 0) {
    const unsigned long long ds = steps_now - ctl->steps_prev, dt = now - ctl->t_prev;
    const bool slow = static_cast<float>(dt) > rule.ticks_per_step_max * static_cast<float>(ds);
    if (total <= rule.tail || slow || rule.round >= rule.max_round) mode = 1;
  }
  ctl->slots = ctl->live;  // what the round before left in the record array
  ctl->live = total;
  ctl->steps_prev = steps_now;
  ctl->t_prev = now;
  ctl->mode = mode;
  if (mode != 0) ctl->stop_round = rule.round;
  return mode;
}

// float -> cell coordinate, 32-bit flavour of cell_of (positions are < 2^31)
__device__ __forceinline__ int cell32(float f) { return (f != f) ? 0 : static_cast<int>(f); }


// Diagnostics build (-DSOIL_ABLATE): parts of the round kernel switched off by a bit mask (timing
// experiments only — the results are wrong by construction).  SOIL_ABLATE=<mask> in the environment:
// 1 no queue sections, 2 no deposits, 4 deposits as plain LDS stores, 8 no gather of the cell record, 16 lost swaps dropped
#ifdef SOIL_ABLATE
__device__ int soil_ablate = 0;
#define ABLATED(bit) ((soil_ablate & (bit)) != 0)
#else
#define ABLATED(bit) false
#endif

// A tile's queue is kept in kNB sections by how long a walker is expected to stay:
// the steps until it leaves the tile along its present direction (about one cell
// per step) or runs out of life, against the round's step budget K — >= K, >= K/2,
// >= K/4, less.  Lanes of a wave take consecutive queue entries, so walkers that
// stop early share waves and those waves retire early, instead of every wave
// carrying a few idle lanes to the end of the round.  Only an ordering hint: any
// section is a valid place for any particle.
constexpr int kNB = 4;

// sparse tiles (k_tiled_round<..., SPARSE>, below)
#ifndef SOIL_SPARSE_BUCKETS
#define SOIL_SPARSE_BUCKETS 4
#endif
// (tiles of up to 31 / 63 / 127 / 191 walkers on the one-wave kernel: 31.4 / 31.0 / 31.9 / 34.6 ms per 8192^2 step —
// beyond one wave's worth the wave refills from the queue, its chain doubles and its table overflows)
constexpr int kSparseBuckets = SOIL_SPARSE_BUCKETS;   // the lowest buckets (16 walkers each) of the scan's histogram
constexpr int kSparseMax = 16 * kSparseBuckets - 1;   // walkers
static_assert(kSparseBuckets >= 2 && kSparseMax < 64,
              "a sparse tile's walkers fit the wave's lanes (the kernel no longer refills: the tiles of under 16 / 32 walkers "
              "are packed four / two to a wave) and the two packed classes are buckets of their own");
#ifndef SOIL_SPARSE_TAB_BITS
#define SOIL_SPARSE_TAB_BITS 8  // 1024 / 512 / 256 / 128 / 64 entries: 32.13 32.21 31.92 | 31.35 31.53 31.68 ms per 8192^2 step (two boxes)
#endif
constexpr int kSparseTabBits = SOIL_SPARSE_TAB_BITS, kSparseTab = 1 << kSparseTabBits, kSparseProbe = 16;
constexpr uint32_t kSparseEmpty = 0xffffffffu;
constexpr int kSparseLanes = 64;

__device__ __forceinline__ uint32_t queue_key(int x0, float px, float py, float spx, float spy,
                                              uint32_t life, int tiles_w, TileShape ts, int K) {
  const int lx = cell32(px) - x0, cy = cell32(py);
  // lx + off_r >= 0 (parked particles stand on owned rows): an unsigned division
  const int trow = static_cast<int>(static_cast<uint32_t>(lx + ts.off_r) / static_cast<uint32_t>(ts.tr));
  const int tcol = (cy + ts.off_c) >> ts.shift_c;
  const float x_lo = static_cast<float>(trow * ts.tr - ts.off_r + x0);
  const float y_lo = static_cast<float>((tcol << ts.shift_c) - ts.off_c);
  const float x_hi = x_lo + static_cast<float>(ts.tr), y_hi = y_lo + static_cast<float>(1 << ts.shift_c);
  const float inv = __builtin_amdgcn_rsqf(spx * spx + spy * spy);
  const float ux = spx * inv, uy = spy * inv;
  const float big = 1.0e9f;
  // (an ordering hint: the hardware's approximate reciprocal will do — two IEEE divisions were a
  // third of this function's instructions, on the path of a work-group's last wave)
  const float tx = ux != 0.0f ? ((ux > 0.0f ? x_hi : x_lo) - px) * __builtin_amdgcn_rcpf(ux) : big;
  const float ty = uy != 0.0f ? ((uy > 0.0f ? y_hi : y_lo) - py) * __builtin_amdgcn_rcpf(uy) : big;
  const float t = fminf(fminf(tx, ty), static_cast<float>(life));
  const float k = static_cast<float>(K);
  uint32_t section = t >= k ? 0u : (t >= 0.5f * k ? 1u : (t >= 0.25f * k ? 2u : 3u));
  if (ABLATED(1)) section = 0u;
  return static_cast<uint32_t>(trow * tiles_w + tcol) * kNB + section;
}

// per-launch constants of the step (erosion.cu:63-72 / :276-283), hoisted
struct StepConst {
  float Hf, Wf, eps, g, lenL, nu, tau, kd, fD, evap, theta, kdd, kds, tau_y, fx, fy;
  int x0, lo, hi, W;  // slab origin, rows with a full stencil (local, inclusive), width
  uint32_t maxage, Wu, base;  // base = x0 * W: local -> global cell index
  bool plain;                 // launch constants in the plain range of quot()
};

template <int KIND>
__device__ __forceinline__ StepConst make_const(const Dom& d, Scale3 s, const Param& p) {
  StepConst k;
  k.Hf = static_cast<float>(d.H);
  k.Wf = static_cast<float>(d.W);
  k.eps = 1E-12f;
  k.g = p.gravity;
  k.lenL = length2(s.x, s.y);
  k.nu = (KIND == FLUVIAL) ? p.viscosityWater : p.viscosityDebris;
  k.tau = (KIND == FLUVIAL) ? p.bedShearWater : p.bedShearDebris;
  k.kd = p.depositionRateFluvial * 1.33f;
  k.fD = p.frictionFactor / 8.0f;
  k.evap = p.evapRate;
  k.theta = p.critSlopeBedrock;
  k.kdd = p.depositionRateDebris;
  k.kds = p.suspensionRateDebris;
  k.tau_y = p.yieldStress;
  k.fx = p.force[0];
  k.fy = p.force[1];
  k.x0 = static_cast<int>(d.x0);
  k.lo = static_cast<int>(stencil_lo(d));
  k.hi = static_cast<int>(stencil_hi(d));
  k.W = static_cast<int>(d.W);
  k.Wu = static_cast<uint32_t>(d.W);
  // lenL and tau + nu as operands of quot(): see advance()
  k.plain = k.lenL >= 0x1p-30f && k.lenL <= 0x1p30f && k.tau + k.nu >= 0.0f && k.tau + k.nu <= 0x1p8f;
  k.base = static_cast<uint32_t>(d.x0 * d.W);
  k.maxage = p.maxage > 0x7fffffffull ? 0x7fffffffu : static_cast<uint32_t>(p.maxage);
  return k;
}

// ---- spent debris walkers (round 6) ------------------------------------------------------------
//
// A debris walker deposits att_v * source_v and att_d * source_d (:310-318).  With the reference's own
// example parameters (example/erosion_gpu.py: yieldStress 2e6, a debris height of Q * kl * excess ~ 1e-7)
// the first step's decay_d is ~ -1e18: att_d underflows to zero on the spot, debrisHeight is eps from
// then on, decay_v = nu + tau / eps ~ 1e10 takes att_v to zero on the next step — and the walker goes
// on for the rest of its 256 steps adding exact zeros (8192^2: 0.85 G of the launch's steps; the debris
// launch was a third of the particle phase).  Such a walker is SPENT: nothing it will ever add can change
// a bit of a flux plane (x + (+-0) = x; the planes start from +0), and it ends its walk here.
//
// When is that certain?  A walker at the top of an iteration with
//     att_v == 0,   att_d * source_d == 0   (so debrisHeight = eps + (+-0) = eps exactly, :331),
//     position, speed and sources finite
// keeps all of it on every later step provided that, for every cell it may stand on,
//     the cell's record is finite and  excessStress = g * (excessSlope - tau_y / eps)  is finite and < 0  (:340)
// and nu, tau, kdd >= 0 with nu + tau / eps finite:  then shearRate = kdd (:341), decay_d = ds * kdd *
// excessStress / v_norm <= 0 or -0 (ds >= 0 or -0, v_norm >= eps > 0: no 0 x inf, no NaN), att_d *= exp(decay_d)
// in [0, 1] stays finite and att_d * source_d stays +-0; decay_v = nu + tau / eps >= 0, att_v *= exp(-dL *
// decay_v) = 0 x [0, 1] = 0 (:346); w = 1 / (1 + dL * decay) in (0, 1], the speed and the position stay finite
// (:335, :347).  The per-cell condition is checked by the pack pass on every cell of the slab with the
// step's own expression (`debris_cell_bad`; one word, set when a cell fails: then nobody retires in this
// launch), the launch constants on the host (TiledRun::setup), the walker's own state by `debris_spent`.
// A NaN walker (reference quirks) never qualifies.  soil_set_debris_retire(0) walks every walker
// to the end as the reference does (same planes, bit for bit up to the order of the additions);
// mode 2 marks spent walkers (PRec::a2), walks them on and counts every deposit that is not an exact zero
// and every marked walker that stops qualifying (soil_debris_retire_violations: the tests want 0).
__device__ unsigned long long soil_retire_violations_dev = 0;
#ifdef SOIL_RETIRE_DEBUG
__device__ float soil_retire_dbg[16];
__device__ int soil_retire_dbg_taken = 0;
__device__ __forceinline__ void retire_dbg(const PRec& r, float kind) {
  if (atomicAdd(&soil_retire_dbg_taken, 1) == 0) {
    float* o = soil_retire_dbg;
    o[0] = kind, o[1] = r.px, o[2] = r.py, o[3] = r.spx, o[4] = r.spy, o[5] = r.a0, o[6] = r.a1, o[7] = r.a2, o[8] = r.s0;
    o[9] = r.svx, o[10] = r.svy, o[11] = static_cast<float>(r.iter);
  }
}
extern "C" int soil_retire_dbg_read(float* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(soil_retire_dbg), sizeof(float) * 16) == hipSuccess ? 0 : 1;
}
#define RETIRE_DBG(r, k) retire_dbg(r, k)
#else
#define RETIRE_DBG(r, k)
#endif
__device__ __forceinline__ bool debris_cell_bad(const float4 qd, const Param& param) {
  const float es = param.gravity * (qd.z - param.yieldStress / 1E-12f);  // :340 with debrisHeight = eps
  const float t = (qd.x - qd.x) + (qd.y - qd.y) + (es - es);             // 0 iff all three are finite
  return !(es < 0.0f) || !(t == 0.0f);
}
__device__ __forceinline__ bool debris_spent(const PRec& r) {
  if (!(r.a1 == 0.0f && r.a0 * r.s0 == 0.0f)) return false;
  const float t = ((r.px - r.px) + (r.py - r.py)) + ((r.spx - r.spx) + (r.spy - r.spy)) +
                  ((r.svx - r.svx) + (r.svy - r.svy)) + ((r.s0 - r.s0) + (r.a0 - r.a0));
  return t == 0.0f && r.a0 >= 0.0f;
}
static bool debris_params_allow_retire(const Param& p) {
  auto ok = [](float v, float hi) { return v >= 0.0f && v <= hi; };
  return ok(p.viscosityDebris, 1.0e30f) && ok(p.bedShearDebris, 1.0e25f) && ok(p.depositionRateDebris, 1.0e30f) &&
         (p.gravity - p.gravity) == 0.0f && (p.yieldStress - p.yieldStress) == 0.0f;
}

// The body of one loop iteration AFTER the bookkeeping at its top (oob, ++iter,
// escape) and the deposit: erosion.cu:116-137 / :321-347.  `q` is the cell's record
// of k_tiled_pack.  Returns false when the walk ends.  Written with `/` throughout:
// the definition, and the path of every lane whose operands are not plain.
template <int KIND>
__device__ __forceinline__ bool advance_slow(PRec& r, const float4 q, const StepConst& k,
                                          const float v_norm) {
  const float ux = r.spx / v_norm, uy = r.spy / v_norm;  // :117 / :322
  const float v_step = stepsize(r.px, r.py, ux, uy);     // :118 / :323
  const float dL = v_step * k.lenL;                      // :119 / :324
  const float ds = dL / v_norm;                          // :120 / :325
  if (KIND == FLUVIAL) {
    const float ax = q.x + k.fx;  // :126
    const float ay = q.y + k.fy;
    const float w0 = 1.0f / (1.0f + dL * (k.tau + k.nu));  // :127
    const float w1 = dL / (1.0f + dL * (k.tau + k.nu));
    r.spx = w0 * r.spx + w1 * ax;
    r.spy = w0 * r.spy + w1 * ay;
    const float decay_v = q.z;  // :132
    r.a1 = r.a1 * att_exp(-ds * k.kd);                   // att_m :134 (__expf: soil_math.hpp, att_exp)
    r.a0 = r.a0 * att_exp(-ds * k.evap);                 // att_w :135
    r.a2 = r.a2 * att_exp(-dL * decay_v);                // att_v :136
  } else {
    const float debrisHeight = k.eps + r.a0 * r.s0;  // :331
    const float ax = q.x;  // :332
    const float ay = q.y;
    const float decay = k.nu + k.tau / debrisHeight;  // :333
    const float w = 1.0f / (1.0f + dL * decay);       // :334
    r.spx = w * r.spx + w * dL * ax;                  // :335
    r.spy = w * r.spy + w * dL * ay;
    const float excessSlope = q.z;                                            // :339
    const float excessStress = k.g * (excessSlope - k.tau_y / debrisHeight);  // :340
    const float shearRate = (excessStress < 0.0f) ? k.kdd : k.kds;            // :341
    const float decay_d = ds * shearRate * excessStress / v_norm;             // :342
    const float decay_v = k.nu + k.tau / debrisHeight;                        // :343
    r.a0 = r.a0 * expf_(decay_d);                                             // :345
    r.a1 = r.a1 * att_exp(-dL * decay_v);                                     // :346 att_v: deposits only
  }
  r.px += v_step * ux;  // :137 / :347
  r.py += v_step * uy;
  return true;
}

// (Round 4: three of the debris step's thirteen quotients have operands the launch constants bound —
// tau / debrisHeight, tau_y / debrisHeight on one reciprocal, 1 / D — and were moved onto quot() with one
// per-lane comparison and no fallback of the whole step: same bits, 12 vector instructions fewer per
// step, and the debris launch 9.96 -> 10.09 ms.  The step is not bound by its instruction count.  Left
// as written.)
//
// The fluvial iteration with the shared-reciprocal quotients of soil_math.hpp.  The
// `v_norm < eps` exit (:121-122) is taken first: nothing the reference computes
// before it has an effect when it fires.  Everything is computed on the assumption
// that the operands are plain; `ok` collects the evidence, and a lane without it
// redoes the step with advance_slow (ok is tested in the positive, so a NaN anywhere
// lands there too).  What sends a lane there at 8192^2: a direction component of
// exactly zero (0.04 % of the steps) and NaN walkers (0.02 %) — 2 % of the wave-steps
// run both paths.  34.2 -> 33.4 ms per launch.
//
// Debris stays on advance_slow: on bare rock (debrisHeight = eps) its velocity
// shrinks by ~1e-8 per step, so direction components underflow, decay_d runs to
// 1e25 and beyond (the scaled regime of the division) and after special-casing all
// of that 1.3 % of the lanes — more than half of the waves — still needed the slow
// path on top of the fast one (14.9 vs 13.4 ms).
// The first half of the iteration needs neither the cell record nor anything from LDS — the norm
// of the speed, the direction, the step length, the reciprocal of the implicit-Euler denominator
// (:116-120 and the D of :127) — and is written apart (StepGeom) so that the round kernel can run it
// while the gather of the record and the LDS reads of the deposit are in flight.
struct StepGeom {
  float v_norm, ux, uy, v_step, dL, ds;
  Recip rd;
  bool alive, ok;  // alive: past the `v_norm < eps` exit; ok: every operand plain (fluvial)
  // the exit itself as one plain comparison of v_norm (for wave ballots): !alive
};
// ---- the step in FAST arithmetic (round 5; soil_set_particle_arith(1), SOIL_PARTICLE_DIV=fast) ----
//
// The reference's step divides nine (debris: thirteen) times with correctly rounded IEEE quotients
// and takes a correctly rounded square root; the exact step above reproduces every bit of that
// (its trajectories equal the oracle's step for step).  nvcc's own fast build of the same source
// (-use_fast_math: __fdividef, sqrt.approx) would not, and neither does this mode: every quotient is
// numerator x v_rcp_f32(denominator) (<= 1.5 ulp), the norm v_sqrt_f32 (<= 1 ulp), debris' att_d the
// hardware exponential the other attenuations already use.  A walk is chaotic in the last bit of its
// speed (a contracted multiply-add already moves walkers across cell edges), so parity
// in this mode is the statistical kind SURVEY.md 8 a5 asks of the transport — plane sums, visited
// sets, step counts against the oracle within the bounds of tests/test_fast_particles.py — not equal
// walks.  What it buys: 120 instead of 172 vector instructions per fluvial step, 13 IEEE divisions
// fewer per debris step; 8192^2 step 30.4 -> 27.6 ms (A/B on one box, three alternations).
__device__ __forceinline__ float fast_quot(float a, float r) { return a * r; }
template <int KIND>
__device__ __forceinline__ StepGeom step_geom_fast(const PRec& r, const StepConst& k) {
  StepGeom g;
  g.v_norm = __builtin_amdgcn_sqrtf(r.spx * r.spx + r.spy * r.spy);  // :116 / :321
  g.alive = !(g.v_norm < k.eps);
  g.ok = true;
  const float rn = __builtin_amdgcn_rcpf(g.v_norm);
  g.ux = fast_quot(r.spx, rn);  // :117 / :322
  g.uy = fast_quot(r.spy, rn);
  // stepsize (erosion_map.cu:56-78) with the quotient over the face the direction points at
  const float x_neg = floorf(r.px), y_neg = floorf(r.py);
  const float nx = ((g.ux > 0.0f) ? 1.0f + x_neg : x_neg) - r.px;
  const float ny = ((g.uy > 0.0f) ? 1.0f + y_neg : y_neg) - r.py;
  // (nx and ux have the same sign whenever ux is not zero, so the face time is |nx| / |ux|; written
  // that way a direction component of exactly zero — every walker that runs down an axis-aligned
  // slope from rest has one — gives |nx| x inf = +inf (or NaN for nx = 0), which the minimum turns into
  // sqrt(2) as the reference's fmax / fmin pair does, where nx x rcp(+-0) would be -inf)
  const float tx = fminf(fast_quot(fabsf(nx), __builtin_amdgcn_rcpf(fabsf(g.ux))), kSqrt2);
  const float ty = fminf(fast_quot(fabsf(ny), __builtin_amdgcn_rcpf(fabsf(g.uy))), kSqrt2);
  g.v_step = 0.5f * (tx + ty);
  g.dL = g.v_step * k.lenL;       // :119 / :324
  g.ds = fast_quot(g.dL, rn);     // :120 / :325
  g.rd = Recip{g.v_norm, rn};     // (carries 1 / v_norm for debris' decay_d)
  return g;
}
template <int KIND>
__device__ __forceinline__ bool step_apply_fast(PRec& r, const float4 q, const StepConst& k, const StepGeom& g) {
  if (KIND == FLUVIAL) {
    const float ax = q.x + k.fx, ay = q.y + k.fy;                          // :126
    const float w0 = __builtin_amdgcn_rcpf(1.0f + g.dL * (k.tau + k.nu));  // :127
    const float w1 = g.dL * w0;
    r.spx = w0 * r.spx + w1 * ax;
    r.spy = w0 * r.spy + w1 * ay;
    r.a1 = r.a1 * att_exp(-g.ds * k.kd);    // att_m :134
    r.a0 = r.a0 * att_exp(-g.ds * k.evap);  // att_w :135
    r.a2 = r.a2 * att_exp(-g.dL * q.z);     // att_v :136
  } else {
    const float debrisHeight = k.eps + r.a0 * r.s0;                        // :331
    const float rh = __builtin_amdgcn_rcpf(debrisHeight);
    const float decay = k.nu + fast_quot(k.tau, rh);                       // :333 (= decay_v, :343)
    const float w = __builtin_amdgcn_rcpf(1.0f + g.dL * decay);            // :334
    r.spx = w * r.spx + w * g.dL * q.x;                                    // :335
    r.spy = w * r.spy + w * g.dL * q.y;
    const float excessStress = k.g * (q.z - fast_quot(k.tau_y, rh));       // :340
    const float shearRate = (excessStress < 0.0f) ? k.kdd : k.kds;         // :341
    const float decay_d = fast_quot(g.ds * shearRate * excessStress, g.rd.r);  // :342
    r.a0 = r.a0 * att_exp(decay_d);                                        // :345
    r.a1 = r.a1 * att_exp(-g.dL * decay);                                  // :346
  }
  r.px += g.v_step * g.ux;  // :137 / :347
  r.py += g.v_step * g.uy;
  return g.alive;
}

template <int KIND>
__device__ __forceinline__ StepGeom step_geom(const PRec& r, const StepConst& k) {
  StepGeom g;
  // :116 / :321.  sqrt_rn is sqrtf for everything that is not under eps = 1e-12 anyway
  g.v_norm = sqrt_rn(r.spx * r.spx + r.spy * r.spy);
  g.alive = !(g.v_norm < k.eps);  // eps = 1e-12 > 2^-40: v_norm is plain from below
  g.ok = false;
  g.ux = g.uy = g.v_step = g.dL = g.ds = 0.0f;
  g.rd = Recip{1.0f, 1.0f};
  if (KIND == DEBRIS) return g;
  bool ok = k.plain && g.v_norm <= kDenHi;
  const Recip rn = recip(g.v_norm);
  // a direction is only ever a denominator for numerators <= 1: plain down to 2^-60
  // (and then |spx| = |ux| v_norm >= 2^-100, a plain numerator in hindsight)
  constexpr float kDirLo = 0x1p-60f;
  g.ux = quot(r.spx, rn);  // :117
  g.uy = quot(r.spy, rn);
  ok = ok && fminf(fabsf(g.ux), fabsf(g.uy)) >= kDirLo;
  // stepsize() with one quotient per axis (see there), :118
  const float x_neg = floorf(r.px), y_neg = floorf(r.py);
  const float nx = ((g.ux > 0.0f) ? 1.0f + x_neg : x_neg) - r.px;
  const float ny = ((g.uy > 0.0f) ? 1.0f + y_neg : y_neg) - r.py;
  const float tx = fminf(quot(nx, recip(g.ux)), kSqrt2);
  const float ty = fminf(quot(ny, recip(g.uy)), kSqrt2);
  g.v_step = 0.5f * (tx + ty);
  g.dL = g.v_step * k.lenL;  // :119
  // dL = -0 is what a walker stuck on a cell corner computes until it dies of age
  // (both face times -0; 1 % of all steps): quot0
  g.ds = quot0(g.dL, rn);  // :120
  // nx, ny: +0 or not tiny; dL: a zero or not tiny.  One min3 decides the common case
  if (fminf(fminf(fabsf(nx), fabsf(ny)), fabsf(g.dL)) < kNumLo)
    ok = ok && plain_num(nx) && plain_num(ny) && (g.dL == 0.0f || fabsf(g.dL) >= kNumLo);
  // D = 1 + dL * (tau + nu) lies in [1, 2^40] by k.plain (dL in [0, sqrt2 * lenL])
  g.rd = recip(1.0f + g.dL * (k.tau + k.nu));
  g.ok = ok;
  return g;
}
// ... and the second half: the speed update, the attenuations, the move (:125-137 / :331-347)
template <int KIND>
__device__ __forceinline__ bool step_apply(PRec& r, const float4 q, const StepConst& k, const StepGeom& g) {
  // A lane whose walk has just ended (v_norm < eps, :121-122 / :326-327) computes on regardless: its
  // record is dropped by the caller, nothing here touches memory, and a branch around the rest costs
  // every iteration of every wave three scalar instructions.  It takes the fast path whatever its
  // operands look like (its results are not used).
  if (KIND == DEBRIS || (!g.ok && g.alive)) {
    (void)advance_slow<KIND>(r, q, k, g.v_norm);
    return g.alive;
  }
  const float ax = q.x + k.fx, ay = q.y + k.fy;                 // :126
  const float w0 = quot(1.0f, g.rd), w1 = quot0(g.dL, g.rd);    // :127
  r.spx = w0 * r.spx + w1 * ax;
  r.spy = w0 * r.spy + w1 * ay;
  r.a1 = r.a1 * att_exp(-g.ds * k.kd);    // att_m :134
  r.a0 = r.a0 * att_exp(-g.ds * k.evap);  // att_w :135
  r.a2 = r.a2 * att_exp(-g.dL * q.z);     // att_v :136
  r.px += g.v_step * g.ux;                // :137
  r.py += g.v_step * g.uy;
  return g.alive;
}
template <int KIND, bool FAST = false>
__device__ __forceinline__ bool advance(PRec& r, const float4 q, const StepConst& k) {
  if constexpr (FAST) return step_apply_fast<KIND>(r, q, k, step_geom_fast<KIND>(r, k));
  else return step_apply<KIND>(r, q, k, step_geom<KIND>(r, k));
}

// a NaN walker's deposit for global cell (0,0) held by another rank (soil_hip.h)
template <int KIND>
__device__ __forceinline__ void park_remote(const PRec& r, float* __restrict__ remote0) {
  if (!(r.px != r.px) || !remote0 || r.ind == 0) return;
  if (KIND == FLUVIAL) {
    atomicAdd(&remote0[0], r.a0 * r.s0);
    atomicAdd(&remote0[1], r.a1 * r.s1);
    atomicAdd(&remote0[2], r.a2 * r.svx);
    atomicAdd(&remote0[3], r.a2 * r.svy);
  } else {
    atomicAdd(&remote0[4], r.a0 * r.s0);
    atomicAdd(&remote0[5], r.a1 * r.svx);
    atomicAdd(&remote0[6], r.a1 * r.svy);
  }
}

// a walker that left the launch's rows through their upper (lx < lo) or lower edge: into the box
__device__ __forceinline__ void migrate_out(const PRec& r, const MigrateBox& box, bool up) {
  const uint32_t slot = atomicAdd(&box.count[up ? 0 : 1], 1u);
  if (slot < box.cap) static_cast<PRec*>(up ? box.up : box.down)[slot] = r;
}

// ---- pre-pass: everything a step needs from the cell it stands on ------------------
//
// All cell-only sub-expressions of the loop body are evaluated once per cell
// (instead of once per visit, ~30x) with the operations and order of the
// reference, so a step is one 16-byte gather:
//   fluvial {a0x, a0y, decay_v, power}: a0 = -(g*grad) + nu*vel  (:77,:90,:126 before
//            `+ force`), decay_v = 0.125*fD/(eps + waterHeight) (:132),
//            power = pow(shear*|grad|, alpha) (:83-85, used at spawn)
//   debris  {ax, ay, excessSlope, 0}:   a = -(g*grad) + nu*vel (:288,:298,:332),
//            excessSlope = |grad| - critSlopeBedrock (:294,:339)

template <int KIND>
__global__ void __launch_bounds__(256)
    k_tiled_pack(float4* __restrict__ q4, const float2* __restrict__ layers,
                 const float2* __restrict__ velocity, const float* __restrict__ waterHeight,
                 Dom d, Scale3 s, Param param, int64_t row_lo, int64_t row_end, uint32_t* __restrict__ debris_bad) {
  // threads along the row, a work-group walks a band of rows (common.hpp: grid_rows)
  const int64_t y = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (y >= d.W) return;
  bool bad = false;
  SOIL_ROW_LOOP(band, row_end - row_lo) {
  const int64_t lx = row_lo + band;
  const int64_t l = lx * d.W + y;
  const float2 grad = glocal(layers, d, s, d.x0 + lx, y, param.exitSlope);
  const float2 vel = velocity[l];
  const float g = param.gravity;
  if (KIND == FLUVIAL) {
    const float nu = param.viscosityWater;
    const float fD = param.frictionFactor / 8.0f;  // :70
    const float eps = 1E-12f;
    const float v = length2(vel.x, vel.y);                                              // :83
    const float shear = 0.125f * fD * param.densityWater * v * v;                       // :84
    const float power = powf_(shear * length2(grad.x, grad.y), param.fluvialExponent);  // :85
    q4[l] = make_float4(-(g * grad.x) + nu * vel.x, -(g * grad.y) + nu * vel.y,
                        0.125f * fD / (eps + waterHeight[l]), power);
  } else {
    const float nu = param.viscosityDebris;
    const float4 qd = make_float4(-(g * grad.x) + nu * vel.x, -(g * grad.y) + nu * vel.y,
                                  length2(grad.x, grad.y) - param.critSlopeBedrock, 0.0f);
    q4[l] = qd;
    bad = bad || debris_cell_bad(qd, param);
  }
  }
  if (KIND == DEBRIS && debris_bad && bad) atomicOr(debris_bad, 1u);
}

// Both launches of a step walk the same layers: one pass evaluates __glocal once per cell and
// writes the two records (soil_particles_pair_slab; 0.68 + 0.55 ms -> one kernel at 8192^2).
__global__ void __launch_bounds__(256)
    k_tiled_pack_pair(float4* __restrict__ q_fluvial, float4* __restrict__ q_debris,
                      const float2* __restrict__ layers, const float2* __restrict__ velocity,
                      const float* __restrict__ waterHeight, const float2* __restrict__ debrisVelocity,
                      Dom d, Scale3 s, Param param, int64_t row_lo, int64_t row_end, uint32_t* __restrict__ debris_bad) {
  const int64_t y = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (y >= d.W) return;
  bool bad = false;
  SOIL_ROW_LOOP(band, row_end - row_lo) {
  const int64_t lx = row_lo + band;
  const int64_t l = lx * d.W + y;
  const float2 grad = glocal(layers, d, s, d.x0 + lx, y, param.exitSlope);
  const float2 vel = velocity[l], dvel = debrisVelocity[l];
  const float g = param.gravity;
  const float glen = length2(grad.x, grad.y);
  {  // the fluvial record, as k_tiled_pack<FLUVIAL>
    const float nu = param.viscosityWater;
    const float fD = param.frictionFactor / 8.0f;  // :70
    const float eps = 1E-12f;
    const float v = length2(vel.x, vel.y);                                     // :83
    const float shear = 0.125f * fD * param.densityWater * v * v;              // :84
    const float power = powf_(shear * glen, param.fluvialExponent);            // :85
    q_fluvial[l] = make_float4(-(g * grad.x) + nu * vel.x, -(g * grad.y) + nu * vel.y,
                               0.125f * fD / (eps + waterHeight[l]), power);
  }
  {  // the debris record, as k_tiled_pack<DEBRIS>
    const float nu = param.viscosityDebris;
    const float4 qd = make_float4(-(g * grad.x) + nu * dvel.x, -(g * grad.y) + nu * dvel.y,
                                  glen - param.critSlopeBedrock, 0.0f);
    q_debris[l] = qd;
    bad = bad || debris_cell_bad(qd, param);
  }
  }
  if (debris_bad && bad) atomicOr(debris_bad, 1u);
}

// The pair pass with four cells per thread (round 4; widths that are a multiple of four, 16-byte
// aligned planes).  __glocal only ever looks at the SUM of a cell's two layers: the thread keeps the
// heights of three rows of its four cells (+ one column either side, from the neighbouring lanes) in
// registers, walks down a band of rows and evaluates glocal_from_heights on them — the operations
// and the order of the pass above, so the same records bit for bit — with 16-byte loads of every
// plane and the two records of its four cells (64 bytes each) handed through a wave-private 4 KiB of
// LDS so that a store instruction of the wave covers 1 KiB of consecutive bytes.  The pass above
// issues five 8-byte gathers of the layer plane per cell and ran at 4.5 TB/s (0.9 ms at 8192^2, in
// front of everything else of the step).
struct HRow6 {
  float v[6];  // heights of columns y0 - 1 .. y0 + 4 of one row; NaN: outside the grid (the reference's sentinel)
};
__device__ __forceinline__ HRow6 load_hrow6(const float2* __restrict__ layers, int64_t lx, bool row_ok, int64_t W,
                                            int64_t y0) {
  const int lane = static_cast<int>(threadIdx.x & 63u);
  const float nan = __builtin_nanf("");
  float c[4] = {nan, nan, nan, nan};
  const float2* row = layers + lx * W;
  if (row_ok) {
    const float4 a = *reinterpret_cast<const float4*>(row + y0), b = *reinterpret_cast<const float4*>(row + y0 + 2);
    c[0] = a.x + a.y, c[1] = a.z + a.w, c[2] = b.x + b.y, c[3] = b.z + b.w;
  }
  float l = __shfl_up(c[3], 1, 64), r = __shfl_down(c[0], 1, 64);
  if (lane == 0 && row_ok && y0 > 0) {
    const float2 v = row[y0 - 1];
    l = v.x + v.y;
  }
  if (lane == 63 && row_ok && y0 + 4 < W) {
    const float2 v = row[y0 + 4];
    r = v.x + v.y;
  }
  if (y0 == 0) l = nan;       // (also what a lane past the row's end must not hand to its neighbour)
  if (y0 + 4 >= W) r = nan;
  return HRow6{{l, c[0], c[1], c[2], c[3], r}};
}
__global__ void __launch_bounds__(kWinBlock)
    k_tiled_pack_pair4(float4* __restrict__ q_fluvial, float4* __restrict__ q_debris,
                       const float2* __restrict__ layers, const float2* __restrict__ velocity,
                       const float* __restrict__ waterHeight, const float2* __restrict__ debrisVelocity,
                       Dom d, Scale3 s, Param param, int64_t row_lo, int64_t row_end, int band_rows,
                       uint32_t* __restrict__ debris_bad) {
  __shared__ float4 s_tile[kWinBlock / 64][256];
  bool bad = false;
  const WinThread t = win_thread(d.W);
  const int lane = static_cast<int>(threadIdx.x & 63u);
  float4* const tile = s_tile[threadIdx.x >> 6];
  const float g = param.gravity;
  auto row_in_grid = [&](int64_t lx) { return d.x0 + lx >= 0 && d.x0 + lx < d.H; };
  for (int64_t band = blockIdx.y; row_lo + band * band_rows < row_end; band += gridDim.y) {
    const int64_t x_first = row_lo + band * band_rows;
    const int64_t x_last = x_first + band_rows < row_end ? x_first + band_rows : row_end;
    HRow6 up = load_hrow6(layers, x_first - 1, row_in_grid(x_first - 1), d.W, t.y0);
    HRow6 mid = load_hrow6(layers, x_first, true, d.W, t.y0);
    for (int64_t lx = x_first; lx < x_last; ++lx) {
      const HRow6 dn = load_hrow6(layers, lx + 1, row_in_grid(lx + 1), d.W, t.y0);
      const int64_t n0 = lx * d.W + t.y0;
      const float4 va = *reinterpret_cast<const float4*>(velocity + n0), vb = *reinterpret_cast<const float4*>(velocity + n0 + 2);
      const float4 da = *reinterpret_cast<const float4*>(debrisVelocity + n0), db = *reinterpret_cast<const float4*>(debrisVelocity + n0 + 2);
      const float4 wh4 = *reinterpret_cast<const float4*>(waterHeight + n0);
      const float velx[4] = {va.x, va.z, vb.x, vb.z}, vely[4] = {va.y, va.w, vb.y, vb.w};
      const float dvx[4] = {da.x, da.z, db.x, db.z}, dvy[4] = {da.y, da.w, db.y, db.w};
      const float wh[4] = {wh4.x, wh4.y, wh4.z, wh4.w};
      float4 qf[4], qd[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float2 grad = glocal_from_heights(mid.v[c + 1], up.v[c + 1], dn.v[c + 1], mid.v[c], mid.v[c + 2], s,
                                                param.exitSlope);
        const float glen = length2(grad.x, grad.y);
        {  // the fluvial record, as k_tiled_pack<FLUVIAL>
          const float nu = param.viscosityWater;
          const float fD = param.frictionFactor / 8.0f;  // :70
          const float eps = 1E-12f;
          const float v = length2(velx[c], vely[c]);                       // :83
          const float shear = 0.125f * fD * param.densityWater * v * v;    // :84
          const float power = powf_(shear * glen, param.fluvialExponent);  // :85
          qf[c] = make_float4(-(g * grad.x) + nu * velx[c], -(g * grad.y) + nu * vely[c], 0.125f * fD / (eps + wh[c]), power);
        }
        {  // the debris record, as k_tiled_pack<DEBRIS>
          const float nu = param.viscosityDebris;
          qd[c] = make_float4(-(g * grad.x) + nu * dvx[c], -(g * grad.y) + nu * dvy[c], glen - param.critSlopeBedrock, 0.0f);
          bad = bad || debris_cell_bad(qd[c], param);  // (a lane past the row's end looks at real cells, its clamped group's)
        }
      }
      // 64 bytes per lane and plane -> 1 KiB-contiguous stores through the wave's tile
      const int n_real = 4 * __popcll(__ballot(t.live));  // float4s of the wave that are real (live lanes are a prefix)
      // the first cell of the wave's lane 0 — from the thread index, not from the lane's own n0: a lane
      // past the row's end has its y0 clamped to W - 4 and still stores its share of the live lanes'
      // records below (widths that are a multiple of 4 but not of 256; advisor finding of round 4)
      const int64_t wave_n0 = lx * d.W + (static_cast<int64_t>(blockIdx.x) * kWinBlock + (threadIdx.x & ~63u)) * 4;
      auto put = [&](float4* plane, const float4* q) {
#pragma unroll
        for (int c = 0; c < 4; ++c) tile[4 * lane + c] = q[c];
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own writes have landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = tile[j * 64 + lane];
          if (j * 64 + lane < n_real) plane[wave_n0 + j * 64 + lane] = v;
        }
        __builtin_amdgcn_wave_barrier();
      };
      put(q_fluvial, qf);
      put(q_debris, qd);
      up = mid;
      mid = dn;
    }
  }
  if (debris_bad && bad) atomicOr(debris_bad, 1u);
}

// ---- spawn: draws, ownership, trajectory initialisation (erosion.cu:49-96 / :262-302)

template <int KIND>
__global__ void __launch_bounds__(256)
    k_tiled_spawn(PRec* __restrict__ recs, uint32_t* __restrict__ dest, uint32_t* __restrict__ rank,
                  uint32_t* __restrict__ count, Streams rng, int64_t N, const float4* __restrict__ p4,
                  const float* __restrict__ waterSource, const float* __restrict__ albedoSource,
                  Dom d, Scale3 s, Param param,
                  int tiles_w, TileShape ts, int steps_per_round, TiledCtl* __restrict__ ctl,
                  const uint32_t* __restrict__ retire_bad, uint32_t retire_mode, bool fast,
                  unsigned long long* __restrict__ steps) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const uint32_t retire = (retire_mode != 0u && retire_bad && *retire_bad == 0u) ? retire_mode : 0u;  // (the pack pass is over)
  if (n == 0) {
    ctl->live = static_cast<uint32_t>(N);  // the spawn fills slots 0 .. N-1 (scan 0 turns it into `slots`)
    ctl->retire = retire;
  }
  __shared__ uint32_t s_head_steps;
  if (KIND == DEBRIS && retire == 1u) {  // (uniform)
    if (threadIdx.x == 0) s_head_steps = 0;
    __syncthreads();
  }
  uint32_t head_steps = 0;
  const bool in_range = n < N;
  PRec r;
  uint32_t tile = kNoTile;
  const float2 pos = in_range ? spawn_position(rng, n, d) : make_float2(-1.0f, -1.0f);
  if (in_range && owns_spawn(d, pos.x)) {
    const float A = s.x * s.y;
    const float Pr = 1.0f / (A * static_cast<float>(d.H * d.W));
    const float Q = 1.0f / (Pr * static_cast<float>(N));
    const float eps = 1E-12f;
    const int64_t cx = cell_of(pos.x), cy = cell_of(pos.y);
    const int64_t ind = cx * d.W + cy;
    const int64_t l = ind - d.x0 * d.W;
    // one record of 16 bytes at a random cell: a streaming load (no line of 128 bytes pulled into L2
    // for the 16 that are used)
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const v4f_ qv = __builtin_nontemporal_load(reinterpret_cast<const v4f_*>(p4 + l));
    const float4 q = make_float4(qv.x, qv.y, qv.z, qv.w);
    float spx, spy;
    if (KIND == FLUVIAL) {
      spx = q.x + param.force[0];  // :77
      spy = q.y + param.force[1];
    } else {
      spx = q.x;  // :288
      spy = q.y;
    }
    const float den = sqrtf(length2(s.x * spx, s.y * spy));  // :78 / :289
    spx = spx / den;
    spy = spy / den;
    if (!(length2(spx, spy) < eps)) {  // :79-80 / :290-291 (a NaN speed walks on)
      r.px = pos.x;
      r.py = pos.y;
      r.spx = spx;
      r.spy = spy;
      r.ind = static_cast<uint32_t>(ind);
      r.iter = 0;
      if (KIND == FLUVIAL) {
        const float ks = param.suspensionRateFluvial / 64.0f;  // :68
        r.a0 = 1.0f;                                  // att_w
        r.a1 = 1.0f;                                  // att_m
        r.a2 = 1.0f;                                  // att_v
        r.s0 = Q * param.rainfall * __builtin_nontemporal_load(waterSource + l);   // source_w :89
        r.s1 = Q * ks * q.w;                          // source_m :88
        r.svx = Q * q.x;                              // :90
        r.svy = Q * q.y;
      } else {
        const float suspend = fmaxf(0.0f, param.landslideRateDebris * q.z);  // :295
        r.a0 = 1.0f;                              // att_d
        r.a1 = 1.0f;                              // att_v
        r.a2 = 0.0f;
        r.s0 = Q * suspend;                       // source_d :297
        r.s1 = 0.0f;
        r.svx = Q * q.x;                          // :298
        r.svy = Q * q.y;
      }
      const float source_mass = (KIND == FLUVIAL) ? r.s1 : r.s0;
      r.sa0 = albedoSource ? source_mass * albedoSource[3 * l] : 0.0f;  // :91 / :299
      r.sa1 = albedoSource ? source_mass * albedoSource[3 * l + 1] : 0.0f;
      r.sa2 = albedoSource ? source_mass * albedoSource[3 * l + 2] : 0.0f;
      const uint32_t maxage = param.maxage > 0x7fffffffull ? 0x7fffffffu : static_cast<uint32_t>(param.maxage);
      bool walks_on = true;
      if (KIND == DEBRIS && retire == 1u && maxage > 1u && (spx - spx) + (spy - spy) == 0.0f) {  // (a NaN speed walks on as ever)
        // The walker's FIRST step, here (debris_spent): with the example's parameters it is the last one that can
        // matter — att_d and att_v underflow to exact zeros in it — and a walker that is spent after it is never
        // made: no record, no place in a queue, no slot in round 0.  The iteration is the round kernel's: the
        // walker stands on its spawn cell (no deposit: nind == ind, :309), `++iter < maxage` passes, the record of
        // the cell is the one the spawn has just read.  Whoever is not spent starts round 0 with iter = 1.
        const StepConst k = make_const<DEBRIS>(d, s, param);
        r.iter = 1;
        head_steps = 1;
        walks_on = fast ? advance<DEBRIS, true>(r, q, k) : advance<DEBRIS, false>(r, q, k);  // false: v_norm < eps, :326-327
        walks_on = walks_on && !debris_spent(r);
        // ... and where the step took it: off the grid the walk is over (:306), off the slab's rows it is another
        // rank's (a finite walker leaves nothing behind: park_remote), as the round kernel sorts a stop out
        const int ix = floor_cell(r.px), iy = floor_cell(r.py);
        const bool oob = static_cast<uint32_t>(ix) >= static_cast<uint32_t>(d.H) || static_cast<uint32_t>(iy) >= k.Wu;
        const int lx = ix - k.x0;
        walks_on = walks_on && !oob && !(lx < k.lo || lx > k.hi) && r.px == r.px && r.py == r.py;
      }
      if (walks_on)
        tile = queue_key(static_cast<int>(d.x0), r.px, r.py, r.spx, r.spy, maxage - static_cast<uint32_t>(r.iter),
                         tiles_w, ts, steps_per_round);
    }
  }
  if (KIND == DEBRIS && retire == 1u) {  // the steps taken here, one atomic per work-group
    const uint32_t in_wave = static_cast<uint32_t>(__popcll(__builtin_amdgcn_ballot_w64(head_steps != 0u)));
    if ((threadIdx.x & 63u) == 0 && in_wave) atomicAdd(&s_head_steps, in_wave);
    __syncthreads();
    if (threadIdx.x == 0 && s_head_steps) atomicAdd(steps, static_cast<unsigned long long>(s_head_steps));
  }
  if (!in_range) return;
  // (Handing record slots only to the walkers that are made — so that a slab owning an eighth of the
  // streams sorts an eighth of the slots — was tried with uniform streams: one counter for the whole
  // launch, drawn from once per wave, is a million same-address atomics at 67 M streams: 61 instead
  // of 39 ms per step for a rank of an 8-GPU world.  The slots stay one per stream.)
  const bool made = tile != kNoTile;
  if (made) {
    rank[n] = atomicAdd(&count[tile], 1u);  // its place in that queue section
    recs[n] = r;
  }
  dest[n] = tile;
}

// ---- a launch that starts from handed-over walkers instead of spawns (MigrateBox) ---------------
// record i of the inbox takes slot i: its queue section and its rank there, as the spawn gives them
template <int KIND>
__global__ void __launch_bounds__(256)
    k_tiled_inject(PRec* __restrict__ recs, uint32_t* __restrict__ dest, uint32_t* __restrict__ rank,
                   uint32_t* __restrict__ count, const PRec* __restrict__ inbox, uint32_t n_in, Dom d, Param param,
                   int tiles_w, TileShape ts, int steps_per_round, TiledCtl* __restrict__ ctl,
                   const uint32_t* __restrict__ retire_bad, uint32_t retire_mode) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i == 0) {
    ctl->live = n_in;
    ctl->retire = (retire_mode != 0u && retire_bad && *retire_bad == 0u) ? retire_mode : 0u;
  }
  if (i >= n_in) return;
  const PRec r = inbox[i];
  const uint32_t maxage = param.maxage > 0x7fffffffull ? 0x7fffffffu : static_cast<uint32_t>(param.maxage);
  const uint32_t key = queue_key(static_cast<int>(d.x0), r.px, r.py, r.spx, r.spy, maxage - static_cast<uint32_t>(r.iter),
                                 tiles_w, ts, steps_per_round);
  rank[i] = atomicAdd(&count[key], 1u);
  recs[i] = r;
  dest[i] = key;
}

// Convergent per-key aggregation of atomicAdd(&counter[key], 1): lanes of a wave
// that share a key are served by one atomic.  Queue order makes neighbouring
// lanes share their tile most of the time, so this removes ~95 % of the atomics
// (and their same-address serialisation in L2).  Every lane of the wave must
// call it; returns the lane's slot (old value + rank among its key group).
// The groups are worked out first (ballots and lane reads only), then the leaders of
// all groups issue their atomics in ONE instruction and hand the result to their
// group: one memory round trip per wave, not one per distinct key (a wave's survivors
// head for up to a dozen queue sections).
__device__ __forceinline__ uint32_t wave_key_append(uint32_t* __restrict__ counter, bool valid,
                                                    uint32_t key) {
  const int lane = threadIdx.x & 63;
  uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
  int my_leader = lane;
  uint32_t my_rank = 0, my_count = 0;
  while (todo) {
    // the leader is the same lane for the whole wave: its key comes through v_readlane (a
    // shuffle is an LDS round trip, a dozen of them in a row where a wave's survivors head for a
    // dozen sections — on the path of the work-group's last wave)
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const uint32_t k0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(key), leader));
    const uint64_t group = __builtin_amdgcn_ballot_w64(valid && key == k0);
    if (valid && key == k0) {
      my_leader = leader;
      my_rank = static_cast<uint32_t>(__popcll(group & ((1ull << lane) - 1ull)));
      my_count = static_cast<uint32_t>(__popcll(group));
    }
    todo &= ~group;
  }
  uint32_t base = 0;
  if (valid && lane == my_leader) base = atomicAdd(&counter[key], my_count);
  base = __shfl(base, my_leader, 64);
  return base + my_rank;
}

// convergent wave-aggregated slot allocation: every lane calls it
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool pred) {
  const uint64_t mask = __ballot(pred);
  if (mask == 0) return 0;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll(static_cast<long long>(mask)) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, static_cast<uint32_t>(__popcll(mask)));
  base = __shfl(base, leader, 64);
  return base + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
}

// ---- counting sort of the record slots by queue section ---------------------------------
//
// dest[i] is the queue section record slot i is bound for (kNoTile: empty slot) and
// rank[i] its place in that section — the value the atomic that counted it returned —
// both written by whoever filled the slot.  Only the 4-byte slot indices are moved;
// the 64-byte records stay where they are and are gathered by the round kernel.

__global__ void __launch_bounds__(256)
    k_tiled_scatter(uint32_t* __restrict__ order, const uint32_t* __restrict__ start,
                    const uint32_t* __restrict__ dest, const uint32_t* __restrict__ rank,
                    const TiledCtl* __restrict__ ctl, uint32_t* __restrict__ clear, int64_t n_clear) {
  if (ctl->mode != 0) return;  // the scan before decided that no round follows
  const int64_t n_src = ctl->slots;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  // the section counters of the round that follows, zeroed on the side (saves a fill
  // launch per round)
  for (int64_t c = i; c < n_clear; c += static_cast<int64_t>(gridDim.x) * 256) clear[c] = 0u;
  if (i >= n_src) return;
  const uint32_t key = dest[i];
  if (key != kNoTile) order[start[key] + rank[i]] = static_cast<uint32_t>(i);
}

// ---- between two rounds: queue offsets, dispatch order, word for the host -----------------
//
// One work-group does the three small jobs a round needs done first:
//  * exclusive scan of the kNB section counts of every tile (a thread owns a run of
//    whole tiles and moves them as uint4);
//  * heaviest tiles first: once the particles sit in channels a few tiles hold ten
//    times the average queue.  Work-groups are dispatched in blockIdx order, so the
//    round kernel looks its tile up in a list sorted by queue length (256 buckets of
//    16 particles, longest first): the long queues start at once and the short ones
//    fill in behind them, instead of a long queue starting last and the rest of the
//    chip idling until it is done;
//  * the queue total and the launch's step counter go straight into pinned host
//    memory (a device-to-host copy is a 23 us blit kernel each on this stack).
static_assert(kNB == 4, "k_queue_prepare moves the sections of a tile as one uint4");

#ifdef SOIL_PROF
__device__ unsigned long long soil_prof_prepare[8];  // cycles between the stamps of k_queue_prepare, summed
#define PREP_AT(i) if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&soil_prof_prepare[i], now_ - prep_last_); prep_last_ = now_; }
#define PREP_DECL unsigned long long prep_last_ = __builtin_readcyclecounter()
#else
#define PREP_AT(i)
#define PREP_DECL
#endif

// A work-group's job in a round: the tile and its share of the tile's queue.  A queue longer than
// the chip's share is served by `groups` work-groups (k_queue_prepare), each with accumulators of
// its own that it adds to the global planes atomically (w = 1).
__device__ __forceinline__ uint4 queue_share(uint32_t tile, uint32_t q_first, uint32_t q_cnt, uint32_t groups,
                                             uint32_t q) {
  const uint32_t per = (q_cnt + groups - 1) / groups;
  const uint32_t off = q * per < q_cnt ? q * per : q_cnt;
  const uint32_t cnt = q_cnt - off < per ? q_cnt - off : per;
  return make_uint4(tile, q_first + off, cnt, groups > 1 ? 1u : 0u);
}


// inclusive prefix sum over the lanes of a wave / the 1024 threads of k_queue_prepare's
// work-group (two barriers; `wsum`: 16 words of LDS)
__device__ __forceinline__ uint32_t wave_scan(uint32_t v) {
  const int lane = static_cast<int>(threadIdx.x & 63u);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* wsum) {
  const int wave = static_cast<int>(threadIdx.x >> 6), lane = static_cast<int>(threadIdx.x & 63u);
  v = wave_scan(v);
  __syncthreads();  // wsum may still be read from the previous call
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) before += (w < wave) ? wsum[w] : 0u;
  return v + before;
}

// inclusive prefix sum over the NT threads of a work-group (two barriers; `wsum`: NT / 64 words of LDS)
template <int NT>
__device__ __forceinline__ uint32_t block_scan_nt(uint32_t v, uint32_t* wsum) {
  const int wave = static_cast<int>(threadIdx.x >> 6), lane = static_cast<int>(threadIdx.x & 63u);
  v = wave_scan(v);
  __syncthreads();  // wsum may still be read from the previous call
  if (lane == 63) wsum[wave] = v;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) before += (w < wave) ? wsum[w] : 0u;
  return v + before;
}

// What a scan is asked to do, by value: it runs as a kernel of its own once per launch (the queues
// the spawn filled) and at the tail of every round kernel (QueueScan of the round that follows).
struct QueueScan {
  uint32_t* start;       // out: kNB offsets per tile (+ the total)
  uint32_t* tile_order;  // scratch: tiles, longest queue first
  uint4* block_list;     // out: the work-groups of the round
  const uint4* count4;   // in: section counts per tile
  int64_t tiles;
  int lanes, slots;
  const unsigned long long* steps_run;
  TiledHostWord* host;
  uint32_t seq;
  TiledCtl* ctl;
  ScanRule rule;
  // a round that stores its tiles (SOIL_FLUX_OVERWRITE) wants a work-group for an EMPTY tile too —
  // it stores the zeros (a slab's ghost rows hold whole tiles nobody spawns on)
  uint32_t include_empty;
  // the round may hand its sparse tiles (fewer than 64 walkers: the last non-empty buckets of the
  // dispatch order) to the one-wave kernel (k_tiled_round<..., SPARSE>); the scan decides whether it does
  uint32_t sparse_ok;
  uint32_t sparse_min, sparse_pct;  // ... when there are at least so many of them, and so many per cent of the non-empty tiles
  uint32_t sparse_pack;             // tiles of fewer than 16 walkers four to a wave (SOIL_TILED_SPARSE_PACK=2: off)
};
constexpr int scan_lds_words(int nt) { return 16 * nt + 256 + 256 + 8 + 16; }

// The scan with its panels in LDS scratch handed in by the caller (scan_lds_words(NT) words): any
// number of tiles, NT threads, all of which must call it.  Global traffic is coalesced (thread t
// takes tiles t, t + NT, ...); the scan wants each thread on a run of consecutive tiles, so the
// per-tile totals go through LDS.
template <int NT>
__device__ void queue_scan_dev(const QueueScan& q, uint32_t* lds) {
  constexpr int kP = 16 * NT, kRun = 16;  // tiles per panel, consecutive tiles per thread within it
  uint32_t* tot = lds;
  uint32_t* hist = lds + kP;
  uint32_t* base = hist + 256;
  uint32_t* misc = base + 256;  // 0 carry, 1 batches, 2 longest, 3 chunk capacity, 4 the panel's total
  uint32_t* wsum = misc + 8;
  uint32_t* start = q.start;
  uint32_t* tile_order = q.tile_order;
  uint4* block_list = q.block_list;
  const uint4* count4 = q.count4;
  const int64_t tiles = q.tiles;
  const int lanes = q.lanes, slots = q.slots;
  TiledHostWord* host = q.host;
  TiledCtl* ctl = q.ctl;
  const int tid = threadIdx.x;
  // the section counts of a tile, read where the atomics that made them were carried out (device
  // scope: past this XCD's L2 — at the tail of a round kernel other XCDs are still adding to them
  // until the last ticket is drawn)
  auto counts_of = [&](int64_t i) {
    const uint32_t* c = reinterpret_cast<const uint32_t*>(count4 + i);
    return make_uint4(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(c + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                      __hip_atomic_load(c + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  };
  // the host spins on host->seq (TiledRun::wait_word): everything it reads is stored, and
  // fenced out to system scope, before the number
  // `blocks`: entries of the block list; the last `n_sparse` of them are the sparse kernel's
  auto publish = [&](uint32_t blocks, uint32_t n_sparse) {
    ctl->blocks = blocks;
    ctl->blocks_of[q.rule.round & 1u] = blocks - n_sparse;
    ctl->sparse_of[q.rule.round & 1u] = n_sparse;
    ctl->sparse_tiny[q.rule.round & 1u] = (n_sparse != 0u && q.sparse_pack != 0u) ? hist[254] : 0u;  // bucket 254: 1 .. 15 walkers
    ctl->sparse_pair[q.rule.round & 1u] = (n_sparse != 0u && q.sparse_pack != 0u && kSparseBuckets >= 2) ? hist[253] : 0u;
    host->blocks = blocks;
    // words of later scans overwrite this one while the host may still be reading it: the verdict
    // is a single word, and what goes with it is out before it
    host->stop_round = ctl->stop_round;
    __threadfence_system();
    __atomic_store_n(&host->mode, ctl->mode, __ATOMIC_RELEASE);
    __threadfence_system();
    __atomic_store_n(&host->seq, q.seq, __ATOMIC_RELEASE);
  };
  // bucket 255: empty tiles; 254..0: 1-15, 16-31, ... particles (longest first)
  auto bucket = [](uint32_t c) { return c == 0 ? 255u : 254u - (c >> 4 > 254u ? 254u : c >> 4); };
  __syncthreads();  // the caller may have used the scratch for something else
  for (int i = tid; i < 256; i += NT) hist[i] = 0;
  if (tid == 0) misc[0] = misc[1] = misc[2] = 0;
  __syncthreads();
  uint4* out = reinterpret_cast<uint4*>(start);
  uint32_t my_batches = 0, my_longest = 0;
  for (int64_t p0 = 0; p0 < tiles; p0 += kP) {
    const int n = static_cast<int>(tiles - p0 < kP ? tiles - p0 : kP);
    for (int i = tid; i < kP; i += NT) {
      uint32_t t = 0;
      if (i < n) {
        const uint4 c = counts_of(p0 + i);
        t = c.x + c.y + c.z + c.w;
        atomicAdd(&hist[bucket(t)], 1u);
        my_batches += (t + lanes - 1) / lanes;
        my_longest = t > my_longest ? t : my_longest;
      }
      tot[i] = t;
    }
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < kRun; ++j) sum += tot[tid * kRun + j];
    const uint32_t incl = block_scan_nt<NT>(sum, wsum);
    if (tid == NT - 1) misc[4] = incl;  // the panel's total
    uint32_t run = misc[0] + incl - sum;
#pragma unroll
    for (int j = 0; j < kRun; ++j) {  // tot[] becomes the exclusive prefix
      const uint32_t t = tot[tid * kRun + j];
      tot[tid * kRun + j] = run;
      run += t;
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
      const uint4 c = counts_of(p0 + i);
      const uint32_t r = tot[i];
      out[p0 + i] = make_uint4(r, r + c.x, r + c.x + c.y, r + c.x + c.y + c.z);
    }
    __syncthreads();
    if (tid == NT - 1) misc[0] += misc[4];
    __syncthreads();
  }
  {  // one atomic per wave on the two scalars
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      my_batches += __shfl_xor(my_batches, off, 64);
      const uint32_t o = __shfl_xor(my_longest, off, 64);
      my_longest = o > my_longest ? o : my_longest;
    }
    if ((tid & 63) == 0) {
      atomicAdd(&misc[1], my_batches);
      atomicMax(&misc[2], my_longest);
    }
  }
  __syncthreads();
  // The block list: walking the tiles longest queue first, every non-empty tile gets a
  // work-group.  A work-group serves its queue in batches of `lanes` particles, so a
  // round takes at least (batches of the longest queue) x (time of a batch), while the
  // chip as a whole needs (all batches / resident work-groups) batch times: a queue
  // longer than that share is cut into chunks of that many batches, each with a
  // work-group of its own (they share the tile, keep flux accumulators of their own
  // and add them to the global planes atomically).  On an 8192^2 grid the share is
  // ~40 batches and nothing is cut; on 1024^2 .. 2048^2 it is 1-3 and the handful
  // of channel tiles would otherwise be the critical path of the whole round.
  if (tid == 0) {
    const uint32_t total = misc[0];
    start[tiles * kNB] = total;  // particles queued in total
    host->live = total;
    host->live_ring[q.seq % TiledHostWord::kLiveRing] = total;
    host->steps = *q.steps_run;
    scan_decide(ctl, q.rule, total, *q.steps_run);
    const uint32_t share = (misc[1] + slots - 1) / slots;
    misc[3] = (share > 0 ? share : 1u) * static_cast<uint32_t>(lanes);  // chunk capacity
    misc[0] = 0;
  }
  {  // first slot of every bucket: exclusive scan of the histogram
    uint32_t h = 0, h2 = 0;  // NT >= 256 except for ... every shape has NT >= 512
    if (tid < 256) h = hist[tid];
    const uint32_t incl = block_scan_nt<NT>(h, wsum);
    if (tid < 256) base[tid] = incl - h;
    (void)h2;
  }
  __syncthreads();
  const uint32_t chunk_cap = misc[3];
  const bool cut = misc[2] > chunk_cap;  // some queue needs more than one work-group
  const uint32_t empty = hist[255];
  // Sparse tiles (1 .. kSparseMax walkers) are the last four non-empty buckets of the order.  They go
  // to the one-wave kernel when they are at least a quarter of the round's tiles (in the dense early
  // rounds the handful there is does not pay for a launch that stands in front of the dense one).
  uint32_t n_sparse = 0;
  if (q.sparse_ok != 0u) {
#pragma unroll
    for (int b = 0; b < kSparseBuckets; ++b) n_sparse += hist[254 - b];  // buckets of 16 walkers: 254 holds 1 .. 15
    const uint32_t nonempty = static_cast<uint32_t>(tiles) - empty;
    if (n_sparse * 100u < nonempty * q.sparse_pct || n_sparse < q.sparse_min) n_sparse = 0;
  }
  for (int64_t i = tid; i < tiles; i += NT) {
    const uint4 c = counts_of(i);
    const uint32_t t = c.x + c.y + c.z + c.w;
    const uint32_t pos = atomicAdd(&base[bucket(t)], 1u);
    tile_order[pos] = static_cast<uint32_t>(i);
    if (!cut && (t > 0 || q.include_empty != 0u))  // (empty tiles are the last in the order)
      block_list[pos] = make_uint4(static_cast<uint32_t>(i), start[i * kNB], t, 0u);
  }
  if (!cut) {  // the common case on large grids: one work-group per non-empty tile
    if (tid == 0) {
      host->whole = (empty == 0 || q.include_empty != 0u) ? 1u : 0u;
      publish(static_cast<uint32_t>(tiles) - (q.include_empty != 0u ? 0u : empty), q.include_empty != 0u ? 0u : n_sparse);
    }
    return;
  }
  if (tid == 0) host->whole = 0u;
  __syncthreads();
  auto groups = [&](int64_t pos) {
    const uint32_t t = tile_order[pos];
    const uint32_t c = start[(t + 1) * kNB] - start[t * kNB];
    return (c + chunk_cap - 1) / chunk_cap;
  };
  for (int64_t p0 = 0; p0 < tiles; p0 += kP) {
    const int n = static_cast<int>(tiles - p0 < kP ? tiles - p0 : kP);
    for (int i = tid; i < kP; i += NT) tot[i] = i < n ? groups(p0 + i) : 0u;
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < kRun; ++j) sum += tot[tid * kRun + j];
    const uint32_t incl = block_scan_nt<NT>(sum, wsum);
    if (tid == NT - 1) misc[4] = incl;  // the panel's total
    uint32_t run = misc[0] + incl - sum;
#pragma unroll
    for (int j = 0; j < kRun; ++j) {
      const int i = tid * kRun + j;
      const uint32_t g = tot[i];
      if (i < n) {
        const uint32_t tile = tile_order[p0 + i];
        const uint32_t q_first = start[tile * kNB], c = start[(tile + 1) * kNB] - q_first;
        for (uint32_t k = 0; k < g; ++k) block_list[run + k] = queue_share(tile, q_first, c, g, k);
      }
      run += g;
    }
    __syncthreads();
    if (tid == NT - 1) misc[0] += misc[4];
    __syncthreads();
  }
  if (tid == 0) publish(misc[0], n_sparse);
}

// the word the host waits for when the scan it belongs to does not take place (the rounds were
// stopped by an earlier scan); one thread
__device__ __forceinline__ void scan_skipped(const QueueScan& q) {
  q.host->live = q.ctl->live;
  q.host->live_ring[q.seq % TiledHostWord::kLiveRing] = q.ctl->live;
  q.host->steps = *q.steps_run;
  q.host->whole = 0u;
  q.host->blocks = 0u;
  q.host->stop_round = q.ctl->stop_round;
  __threadfence_system();
  __atomic_store_n(&q.host->mode, q.ctl->mode, __ATOMIC_RELEASE);
  __threadfence_system();
  __atomic_store_n(&q.host->seq, q.seq, __ATOMIC_RELEASE);
}

// the scan as a kernel of its own: the queues the spawn filled
__global__ void __launch_bounds__(1024) k_queue_scan(QueueScan q) {
  __shared__ uint32_t lds[scan_lds_words(1024)];
  if (q.ctl->mode != 0) {
    if (threadIdx.x == 0) scan_skipped(q);
    return;
  }
  queue_scan_dev<1024>(q, lds);
}

// ---- taking turns: the two launches of a step on a grid either of them fills by itself -------------
//
// Mixed freely, the two round kernels wasted LDS (with the 68-row debris tiles this was measured
// with, a CU held two fluvial tiles or three debris ones,
// one of each leaves 28 KiB unused); run one after the other, each leaves the chip half empty at the
// end of every round (the last generation of work-groups, the slot sort, the scan).  So they take
// turns: a launch's round may begin once the other launch is NOT in the dense part of a round — its
// work-groups all handed out (what is still running is the tail), or between two rounds, or done —
// and the work-groups of the new round fill the slots the other's tail leaves.  `dense[k]` is launch
// k's state; a one-thread gate kernel in front of every round kernel waits (device-scope loads,
// s_sleep, bounded: after 50 ms it lets the round through) while the other's is set and sets its
// own; the last work-group of a round to START clears it.  Both gates waiting at once means neither
// launch is in a round: both flags are clear and both pass.
struct PairGate {
  uint32_t dense[2];
};
__global__ void k_pair_gate(PairGate* gate, int me, const TiledCtl* __restrict__ ctl,
                            unsigned long long ticks_max, uint32_t free_below) {
  if (ctl->mode != 0) return;  // no round follows
  // A round of few walkers does not fill the chip: it neither waits nor makes the other launch wait
  // (SOIL_PAIR_FREE: its launch's chain of late rounds then runs under the other's dense rounds
  // instead of in step with them, and is over before the step's end is all tails)
  if (ctl->live < free_below) return;
  const unsigned long long t0 = realtime_ticks();
  while (__hip_atomic_load(&gate->dense[1 - me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    __builtin_amdgcn_s_sleep(32);
    if (realtime_ticks() - t0 > ticks_max) break;
  }
  __hip_atomic_store(&gate->dense[me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- one round: advance the particles of one tile against LDS ---------------------

// Float adds on LDS words by compare-and-swap, split in two halves so that the
// swaps' round trip hides under the step arithmetic: begin() reads the old words
// and issues one swap per plane, finish() (after the step) looks at the results
// and a lane that lost a race falls back to the native atomic.  ds_add_f32 needs no such care but
// occupies the LDS pipe ~170 cycles per wave instruction on gfx950 (2.6 per lane;
// tools/microbench/lds_atomic.hip), a ds_cmpst ~6.
// Sum over the 64 lanes of a wave, delivered in lane 63; all lanes must be active.
// Row-wise Hillis-Steele over DPP shifts (zeros shift in), then lane 15 of a row is
// broadcast into the next row and lane 31 into the upper half.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + bits2f(static_cast<uint32_t>(__builtin_amdgcn_update_dpp(
                 0, static_cast<int>(f2bits(v)), CTRL, ROW_MASK, 0xf, true)));
}
__device__ __forceinline__ float wave_sum63(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8   lane 15 of each row: the row's sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return v;
}

// The first 2 * NPAIRS planes come in pairs that share an 8-byte LDS word (p[2i + 1] == p[2i] + 1:
// water | mass, velocity x | y): one 64-bit read and one 64-bit swap serve both — half the LDS
// operations of a deposit; a lost swap of a pair is made up for plane by plane like any other.
template <int NP, int NPAIRS = 0>
struct CasDeposit {
  float* p[NP];
  float v[NP];
  uint32_t o[NP], g[NP];
  static __device__ __forceinline__ uint32_t swap(float* q, uint32_t expect, float add) {
    return atomicCAS(reinterpret_cast<uint32_t*>(q), expect, f2bits(bits2f(expect) + add));
  }
  __device__ __forceinline__ void load() {  // the old words: issued early, used by swap_all()
#pragma unroll
    for (int i = 0; i < NPAIRS; ++i) {
      const uint2 w = *reinterpret_cast<const uint2*>(p[2 * i]);
      o[2 * i] = w.x;
      o[2 * i + 1] = w.y;
    }
#pragma unroll
    for (int j = 2 * NPAIRS; j < NP; ++j) o[j] = f2bits(*p[j]);
  }
  __device__ __forceinline__ void swap_all() {
#pragma unroll
    for (int i = 0; i < NPAIRS; ++i) {
      const unsigned long long expect = static_cast<unsigned long long>(o[2 * i]) |
                                        (static_cast<unsigned long long>(o[2 * i + 1]) << 32);
      const unsigned long long want =
          static_cast<unsigned long long>(f2bits(bits2f(o[2 * i]) + v[2 * i])) |
          (static_cast<unsigned long long>(f2bits(bits2f(o[2 * i + 1]) + v[2 * i + 1])) << 32);
      const unsigned long long got = atomicCAS(reinterpret_cast<unsigned long long*>(p[2 * i]), expect, want);
      g[2 * i] = static_cast<uint32_t>(got);  // a pair is lost as a whole, see lost_plane()
      g[2 * i + 1] = static_cast<uint32_t>(got >> 32);
    }
#pragma unroll
    for (int j = 2 * NPAIRS; j < NP; ++j) g[j] = swap(p[j], o[j], v[j]);
  }
  // Nonzero when one of the swaps was lost.  Looking at the answers waits for them: call it after
  // the work their round trip is to hide under.
  __device__ __forceinline__ uint32_t lost_bits() const {
    uint32_t d = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) d |= g[j] ^ o[j];
    return d;
  }
  __device__ __forceinline__ bool lost_plane(int j) const {
    if (j < 2 * NPAIRS) return ((g[j & ~1] ^ o[j & ~1]) | (g[j | 1] ^ o[j | 1])) != 0u;
    return g[j] != o[j];
  }
  __device__ __forceinline__ void begin() {
    load();
    swap_all();
  }
  // Convergent: every lane of the wave calls it, once per iteration, with lost_bits() of its
  // deposit (0: did not deposit, or won every swap).
  // A lane that lost its race (another walker hit the same cell in between — common
  // once the particles share channels) does not retry: nothing was written by the
  // failed swap, and a k-way collision would cost k round trips.
  //  * few losers in the wave: each hands its adds to the native atomic (no answer
  //    needed; k lane-slots of the LDS pipe);
  //  * many (a wave walking a channel: most lanes stand on a handful of cells): the
  //    losers of one cell add their values up in registers (a masked wave sum over DPP)
  //    and one lane issues the atomic — on the hot tiles that set the length of a
  //    round on small grids, ds_add_f32 at 2.6 cycles per lane was all the LDS pipe did.
  // `cell` is the lane's cell index in the tile (any value for a lane that lost nothing).
  template <bool TOGETHER = false>
  __device__ __forceinline__ void finish(uint32_t lost_bits, int cell, int agg_min, int agg_groups, int retries) {
    uint64_t todo = __builtin_amdgcn_ballot_w64(lost_bits != 0u);
    if (todo == 0) return;  // the common case
    bool lost = lost_bits != 0u;
    const int lane = static_cast<int>(threadIdx.x & 63u);
    if (__popcll(todo) >= agg_min) {
      bool lostj[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) lostj[j] = lost && lost_plane(j);
      for (int it = 0; todo != 0 && it < agg_groups; ++it) {
        const int leader = __ffsll(static_cast<long long>(todo)) - 1;
        const int c0 = __builtin_amdgcn_readlane(cell, leader);
        const bool in = lost && cell == c0;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const float sum = wave_sum63((in && lostj[j]) ? v[j] : 0.0f);
          const float total = bits2f(static_cast<uint32_t>(
              __builtin_amdgcn_readlane(static_cast<int>(f2bits(sum)), 63)));
          if (lane == leader && total != 0.0f) atomicAdd(p[j], total);
        }
        todo &= ~__ballot(in);
      }
      lost = lost && ((todo >> lane) & 1ull) != 0;  // cells beyond the budget: one by one
    }
    if (TOGETHER && lost && NPAIRS == 2 && NP == 4) {
      // Round 5 (the fast step, whose registers allow it): both pairs' repeated swaps issued together —
      // one LDS round trip per attempt instead of one per pair and attempt.
      bool open0 = lost_plane(0), open1 = lost_plane(2);
      unsigned long long e0 = static_cast<unsigned long long>(g[0]) | (static_cast<unsigned long long>(g[1]) << 32);
      unsigned long long e1 = static_cast<unsigned long long>(g[2]) | (static_cast<unsigned long long>(g[3]) << 32);
      auto sum2 = [](unsigned long long e, float a, float b) {
        return static_cast<unsigned long long>(f2bits(bits2f(static_cast<uint32_t>(e)) + a)) |
               (static_cast<unsigned long long>(f2bits(bits2f(static_cast<uint32_t>(e >> 32)) + b)) << 32);
      };
#pragma unroll
      for (int attempt = 0; attempt < kRetries; ++attempt) {
        if (attempt < retries && (open0 || open1)) {
          unsigned long long g0 = e0, g1 = e1;
          if (open0) g0 = atomicCAS(reinterpret_cast<unsigned long long*>(p[0]), e0, sum2(e0, v[0], v[1]));
          if (open1) g1 = atomicCAS(reinterpret_cast<unsigned long long*>(p[2]), e1, sum2(e1, v[2], v[3]));
          open0 = open0 && g0 != e0;
          open1 = open1 && g1 != e1;
          e0 = g0;
          e1 = g1;
        }
      }
      if (open0) {
        atomicAdd(p[0], v[0]);
        atomicAdd(p[1], v[1]);
      }
      if (open1) {
        atomicAdd(p[2], v[2]);
        atomicAdd(p[3], v[3]);
      }
      return;
    }
    if (lost) {
      // The failed swap answered with what the word holds now: swap again against that — twice at
      // most — before falling back to the native add.  (Rounds 2-4 read the microbenchmark as "a
      // ds_add_f32 keeps the LDS pipe ~170 cycles per instruction however few lanes take part"; measured
      // by active lanes in round 5 — 64: 169, 32: 86, 8: 22, 2: 6.5, 1: 4.3 cycles — it is 2.65 cycles per
      // ACTIVE lane, so a few losers' native adds are cheap on the pipe; what the repeated swaps buy is the
      // retry without the float pipe's serialisation when many lanes lost.)  A ds_cmpst ~6: once walkers share
      // channels (every round after the first) a quarter of all wave-iterations have a loser,
      // and with four native adds each the pipe was 0.7 busy against 0.46 in the first round.
#pragma unroll
      for (int i = 0; i < NPAIRS; ++i) {
        bool open = lost_plane(2 * i);
        unsigned long long expect = static_cast<unsigned long long>(g[2 * i]) |
                                    (static_cast<unsigned long long>(g[2 * i + 1]) << 32);
#pragma unroll
        for (int attempt = 0; attempt < kRetries; ++attempt) {
          if (open && attempt < retries) {
            const unsigned long long want =
                static_cast<unsigned long long>(f2bits(bits2f(static_cast<uint32_t>(expect)) + v[2 * i])) |
                (static_cast<unsigned long long>(f2bits(bits2f(static_cast<uint32_t>(expect >> 32)) + v[2 * i + 1])) << 32);
            const unsigned long long got = atomicCAS(reinterpret_cast<unsigned long long*>(p[2 * i]), expect, want);
            open = got != expect;
            expect = got;
          }
        }
        if (open) {
          atomicAdd(p[2 * i], v[2 * i]);
          atomicAdd(p[2 * i + 1], v[2 * i + 1]);
        }
      }
#pragma unroll
      for (int j = 2 * NPAIRS; j < NP; ++j) {
        bool open = lost_plane(j);
        uint32_t expect = g[j];
#pragma unroll
        for (int attempt = 0; attempt < kRetries; ++attempt) {
          if (open && attempt < retries) {
            const uint32_t got = swap(p[j], expect, v[j]);
            open = got != expect;
            expect = got;
          }
        }
        if (open) atomicAdd(p[j], v[j]);
      }
    }
  }
  // (Round 4, tried: the repeated swaps of all planes issued together, one LDS round trip per attempt
  // instead of one per plane — 8192^2 step 31.6 -> 32.8 ms, 1024^2 1.78 -> 1.82: the arrays the attempt
  // loop carries cost the stepping loop registers (3 more spilled) and that is worth more than the
  // round trips of the rare path.)
  static constexpr int kRetries = 2;  // at most
};

// Diagnostics build (-DSOIL_PROF): s_memtime stamps at the seams of an iteration, summed per
// wave into soil_prof[] (tools/prof_round.py reads them).  Not part of the product build.
#ifdef SOIL_PROF
__device__ unsigned long long soil_prof[2][16];
#define PROF_DECL unsigned long long pt_last = __builtin_readcyclecounter(), pt_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned pt_iters = 0
#define PROF_AT(i) { const unsigned long long pt_now = __builtin_readcyclecounter(); pt_acc[i] += pt_now - pt_last; pt_last = pt_now; }
#define PROF_FLUSH(kind) { PROF_AT(9); __shared__ unsigned long long pt_sh[12]; if (threadIdx.x < 12) pt_sh[threadIdx.x] = 0; __syncthreads(); \
    if ((threadIdx.x & 63u) == 0) { for (int i = 0; i < 10; ++i) atomicAdd(&pt_sh[i], pt_acc[i]); atomicAdd(&pt_sh[10], static_cast<unsigned long long>(pt_iters)); atomicAdd(&pt_sh[11], 1ull); } \
    __syncthreads(); if (threadIdx.x < 12) atomicAdd(&soil_prof[kind][(threadIdx.x + blockIdx.x) % 12], pt_sh[(threadIdx.x + blockIdx.x) % 12]); }
extern "C" int soil_prof_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(soil_prof), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(soil_prof), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
extern "C" int soil_prof_read_prepare(unsigned long long* out, int reset) {  // 8 words
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(soil_prof_prepare), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(soil_prof_prepare), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#else
#define PROF_DECL
#define PROF_AT(i)
#define PROF_FLUSH(kind)
#endif

// Diagnostics build (-DSOIL_STATS, tools/stats_round.py): what the deposits of the dense round kernel meet,
// counted per wave in scalar registers and added up at the end of the work-group.  Not part of the product build.
//  [0] wave-iterations  [1] lanes stepping  [2] lanes depositing  [3] -
//  [4] wave-iterations with a lost swap  [5] lanes that lost  [6] losers that share their cell with another depositing
//  lane of the wave  [7] depositing lanes that do  [8] distinct cells among the losers, summed  [9] wave-iterations
//  whose losers all stand on ONE cell  [10] lanes still open after the repeats (native add)
#ifdef SOIL_STATS
__device__ unsigned long long soil_stats[2][24];
#define STATS_DECL unsigned long long st_acc[24] = {0}
#define STATS_ADD(i, n) st_acc[i] += static_cast<unsigned long long>(n)
#define STATS_FLUSH(kind) { if ((threadIdx.x & 63u) == 0) for (int i = 0; i < 24; ++i) if (st_acc[i]) atomicAdd(&soil_stats[kind][i], st_acc[i]); }
extern "C" int soil_stats_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(soil_stats), sizeof(unsigned long long) * 48) != hipSuccess) return 1;
  if (reset) { unsigned long long z[48] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(soil_stats), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#else
#define STATS_DECL
#define STATS_ADD(i, n)
#define STATS_FLUSH(kind)
#endif

// Waves per SIMD the launch is meant to run with (bounds the register allocation): two
// work-groups of 768 or three of 512 per CU are 6; the colour shape (1024, one per CU) 4.
// LDS of one work-group of the round kernel: the flux accumulators of its tile (+ a few words)
constexpr int kLdsPerCU = 160 * 1024;
constexpr int round_lds_bytes(int kind, int tr, int tc, bool alb) {
  return tr * tc * 4 * ((kind == FLUVIAL ? 4 : 3) + (alb ? 3 : 0)) + 16;
}
// ... work-groups of it a CU holds (LDS, 32 wave slots) and the waves per SIMD that makes — what the
// register allocation is held to
constexpr int round_groups_per_cu(int kind, int tr, int tc, int nt, bool alb) {
  const int by_lds = kLdsPerCU / round_lds_bytes(kind, tr, tc, alb), by_waves = 32 / (nt / 64);
  return by_lds < by_waves ? by_lds : by_waves;
}
// (a work-group's waves are dealt out over the four SIMDs: the fullest one holds ceil(waves / 4) of
// each work-group — 640 or 896 lanes need 6 / 8 waves per SIMD where 768 need 6; measured as
// "occupancy cliffs", 36 against 24 ms, when the bound said 5 / 7)
constexpr int round_waves_per_simd(int kind, int tr, int tc, int nt, bool alb) {
  return round_groups_per_cu(kind, tr, tc, nt, alb) * ((nt / 64 + 3) / 4);
}
// The deposit's LDS round trips are placed around the step's arithmetic on purpose (reads asked for, the
// geometry, swaps issued, the speed update, answers looked at).  With the short arithmetic of the fast
// step the compiler merges the deposit's two conditional blocks and the waits land right behind the
// requests; an empty volatile statement that names the phase's results keeps them apart (measured: no
// difference at 8192^2 — the other waves of the SIMD cover the round trips either way; kept because
// the listing then reads like the source).
__device__ __forceinline__ void pin3(float& a, float& b, float& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c) : : "memory"); }
#ifndef SOIL_RETRY_TOGETHER
#define SOIL_RETRY_TOGETHER 1
#endif

__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
// A value the compiler may not look through.  A ballot wants to be the ballot of a comparison of
// register values (v_cmp writes the mask); given a predicate that was merged over divergent
// branches the compiler keeps it as a lane mask and pays a v_cndmask + v_cmp pair per ballot to
// turn it back into one.  Merging the compared VALUE instead and hiding it behind this keeps the
// comparison at the ballot.
__device__ __forceinline__ float opaque(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
  asm volatile("" : "+v"(x));
  return x;
}

// SPARSE (round 4): the work-group of a tile that holds fewer than 64 walkers.  In the late rounds of a
// launch most tiles do — the walkers sit in channels — and each of them used to hold 78 KiB of LDS
// and a dozen wave slots for the 30-40 us its longest walker needs: the chip was full of waiting waves
// (profiles/r04_stalls: 83-92 % of the wave slots taken in every round, the vector pipes issuing on
// one cycle in 13-19).  A sparse tile's walkers touch a few hundred cells, so its accumulators are a
// hash table keyed by the cell (kSparseTab = 256 entries: 5 KiB fluvial / 4 KiB debris) in the LDS of a
// ONE-wave work-group: thirty of them per CU where two tiles fitted.  Same walks, same deposits per
// cell; a cell that finds no slot within kSparseProbe probes adds straight to the planes.  The scan
// decides per round whether the sparse tiles get this kernel (QueueScan::sparse_ok, k sparse tiles in
// four non-empty ones); it runs in front of the dense kernel of the round, on the same stream.

template <int KIND, int DEP, int TR, int TC, int NT, bool ALB, bool SPARSE = false, bool FAST = false>
__global__ void __launch_bounds__(NT, SPARSE ? 2 : round_waves_per_simd(KIND, TR, TC, NT, ALB))
    k_tiled_round(PRec* __restrict__ out, uint32_t* __restrict__ dest, uint32_t* __restrict__ rank,
                  uint32_t* count_next, const PRec* __restrict__ in,
                  const uint32_t* __restrict__ order, const uint4* block_list,
                  float* __restrict__ flux0,
                  float* __restrict__ flux1, float2* __restrict__ fluxV,
                  float* __restrict__ fluxA, const float4* __restrict__ p4,
                  float* __restrict__ remote0, unsigned long long* __restrict__ steps, Dom d,
                  Scale3 s, Param param, int tiles_w, int off_r, int off_c, int steps_per_round,
                  TileShape ts_next,
                  int tiles_w_next, int agg_min, int agg_groups, int retries, int store_all,
                  TiledCtl* __restrict__ ctl, uint32_t round, QueueScan next, uint32_t* my_dense, uint32_t gate_early,
                  MigrateBox box) {
  constexpr int kCells = TR * TC, kBlock = NT, kPer = (kCells + NT - 1) / NT;
  // Queued ahead of the scan's verdict: no round at all (the word the host waits for at the end of
  // this round still goes out), or fewer work-groups than the launch has.
  if (ctl->mode != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && next.host) scan_skipped(next);
    return;
  }
  static_assert(!SPARSE || (NT == kSparseLanes && !ALB && DEP == 1), "a sparse tile: one wave, no colour planes");
  const uint32_t n_dense = ctl->blocks_of[round & 1u];
  const uint32_t retire = (KIND == DEBRIS) ? ctl->retire : 0u;  // debris_spent: 1 spent walkers end here, 2 they are watched
  // SPARSE: the last `n_tiny` of the round's sparse tiles hold fewer than 16 walkers and go four to a
  // wave, a quarter of the wave each (`sub`): a wave of 6-14 walkers issues every instruction for 64
  // lanes, and thirty such waves per CU share the vector pipes — the late rounds' sparse kernels were
  // bound by that, not by their walkers' chains.  Everything that derives from the job — the tile's
  // origin, its bounds, the queue — is a per-lane value then; the table is shared (keyed by the cell's
  // index in the slab, not in the tile).
  const uint32_t n_sparse_tiles = SPARSE ? ctl->sparse_of[round & 1u] : 0u;
  const uint32_t n_tiny = SPARSE ? ctl->sparse_tiny[round & 1u] : 0u;
  const uint32_t n_pair = SPARSE ? ctl->sparse_pair[round & 1u] : 0u;  // (16 .. 31 walkers: two to a wave, the same way)
  const uint32_t n_single = n_sparse_tiles - n_tiny - n_pair;
  const uint32_t g_pair = (n_pair + 1u) / 2u, g_tiny = (n_tiny + 3u) / 4u;
  const uint32_t n_groups = SPARSE ? n_single + g_pair + g_tiny : n_dense;
  // (uniform) tiles per wave of this work-group: 1, 2 or 4
  const uint32_t pack_shift = !SPARSE || blockIdx.x < n_single ? 0u : (blockIdx.x < n_single + g_pair ? 1u : 2u);
  const bool packed = pack_shift != 0u;
  const uint32_t sub_lanes = static_cast<uint32_t>(NT) >> pack_shift;
  const uint32_t sub = packed ? threadIdx.x >> (6u - pack_shift) : 0u;  // (kSparseLanes = 64)
  const uint32_t lane_sub = packed ? threadIdx.x & (sub_lanes - 1u) : threadIdx.x;  // the lane's place in its tile's share of the wave
  // (no dense tile at all — every tile of the round went to the sparse kernel, which has run: it stands
  // in front of this one on the stream — : the first work-group is the round's last, see below)
  const bool all_sparse = !SPARSE && n_groups == 0u;
  if (blockIdx.x >= n_groups && !(all_sparse && blockIdx.x == 0)) return;
  // taking turns with the other launch of the step (PairGate): the last work-group of the round to
  // start — every one has been handed out, what follows is the round's tail — lets the other's next
  // round in
  // The ticket is a returning device-scope atomic (carried out at the memory side, ~2.5 us): drawn
  // where it stood — in front of the job's three dependent loads — it held wave 0, and with it the
  // barrier behind the zeroing of the tile, up for that long in every work-group of the overlapped
  // step.  It is drawn behind the record loads now and looked at behind that barrier.
  const bool gate_lane = my_dense != nullptr && threadIdx.x == 0;
  const uint32_t gate_target =
      n_groups - 1u - static_cast<uint32_t>((static_cast<uint64_t>(n_groups - 1u) * gate_early) / 100u);
  auto gate_draw = [&]() {
    return __hip_atomic_fetch_add(&ctl->started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto gate_look = [&](uint32_t ticket) {
    if (ticket == gate_target) __hip_atomic_store(my_dense, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  PROF_DECL;
  STATS_DECL;
  // this work-group's share of its tile's queue (the scan's block list)
  uint4 job;
  if (packed) {
    // this share's tile: the pairs stand behind the single tiles in the list, the quads behind the pairs
    const bool quad = pack_shift == 2u;
    const uint32_t t = quad ? (blockIdx.x - n_single - g_pair) * 4u + sub : (blockIdx.x - n_single) * 2u + sub;
    const uint32_t n_class = quad ? n_tiny : n_pair, at = n_dense + n_single + (quad ? n_pair : 0u);
    job = block_list[at + (t < n_class ? t : n_class - 1u)];
    if (t >= n_class) job.z = 0u;  // (the last wave's shares beyond the list: nothing queued)
  } else {
    job = block_list[(SPARSE ? n_dense : 0u) + blockIdx.x];
  }
  const int tile = static_cast<int>(job.x);
  const uint32_t first = job.y, cnt = job.z;
  const bool shared_tile = job.w != 0;  // other work-groups deposit into the same cells
  // local row, column of the tile's first cell (negative on the rim of a shifted grid)
  const int row0 = (tile / tiles_w) * TR - off_r, col0 = (tile % tiles_w) * TC - off_c;
  const int tid = threadIdx.x;

  // flux accumulators as separate planes: lane addresses c map to 32 distinct
  // banks (an AoS float4 would put every lane of a deposit on 8 banks)
  // fluvial: water | mass interleaved (an 8-byte word per cell: CasDeposit's pairs); debris: mass
  // one block of LDS: the accumulators, and — for the work-group that finishes the round — the
  // scratch of the scan of the round that follows (queue_scan_dev)
  constexpr int kA = (KIND == FLUVIAL) ? 2 : 1;  // floats of s_a per cell
  constexpr int kAcc = kA * kCells + 2 * kCells + (ALB ? 3 * kCells : 0);
  constexpr int kFluxPlanesL = (KIND == FLUVIAL) ? 4 : 3;
  constexpr int kWords = SPARSE ? kSparseTab * (1 + kFluxPlanesL)
                                : (kAcc > scan_lds_words(NT) ? kAcc : scan_lds_words(NT));
  __shared__ __attribute__((aligned(16))) float s_mem[kWords];
  // SPARSE: the table — kSparseTab keys (the cell's index in the slab; kSparseEmpty: free), then one
  // array of kSparseTab sums per flux plane, in deposit_terms' order
  uint32_t* const t_key = reinterpret_cast<uint32_t*>(s_mem);
  float* const t_val = s_mem + kSparseTab;
  if constexpr (!SPARSE) {
    if (all_sparse) {  // uniform: nothing to walk, the scan of the round that follows is this work-group's
      if (my_dense && threadIdx.x == 0) __hip_atomic_store(my_dense, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (next.host) queue_scan_dev<NT>(next, reinterpret_cast<uint32_t*>(s_mem));
      return;
    }
  }
  float* const s_a = s_mem;                                  // water | mass (fluvial), mass (debris)
  float* const s_v = s_mem + kA * kCells;                    // velocity flux x | y interleaved
  float* const s_c0 = s_v + 2 * kCells;                      // colour (ALB)
  float* const s_c1 = s_c0 + (ALB ? kCells : 0);
  float* const s_c2 = s_c1 + (ALB ? kCells : 0);
  __shared__ uint32_t s_next, s_out, s_steps, s_last;
  __shared__ uint32_t s_out_sub[4];  // SPARSE, packed: slots each quarter's tile has filled
  // The ticket of the round's end (the work-group that draws the last one scans the queues of the
  // round that follows): drawn by thread 0 as soon as this work-group's survivors are counted — behind
  // the barrier that ends the stepping — and looked at behind the flush, which its round trip hides under.
  uint32_t done_ticket = 0;
  bool done_drawn = false;
  auto done_draw = [&]() {
    done_drawn = true;
    if (next.host && tid == 0)
      done_ticket = __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  do {  // the round's work proper (a chunk of a cut queue may be empty)
  if (!packed && cnt == 0 && !store_all) {  // (storing: an empty tile's zeros go out like any other's)
    if (gate_lane) gate_look(gate_draw());
    break;
  }
  if (tid == 0) {
    s_next = kBlock;  // the first kBlock queue entries go to the lanes directly, see below
    s_out = 0;
    s_steps = 0;
  }
  if (SPARSE && tid < 4) s_out_sub[tid] = 0;
  const StepConst k = make_const<KIND>(d, s, param);
  // Lane t starts on queue entry t — no counter involved, and the record's two dependent
  // loads (slot index, then the 64-byte record) travel while the flux tile is zeroed.
  // Only a queue longer than the work-group is handed out through s_next.
  PRec r;
  r.iter = -1;
  // (Round 5, tried: queue entry e to wave e % waves, so that walkers parked one after the other — they travel
  // together down a channel and lose each other's swaps by construction when they share a wave — stand in
  // different waves: 8192^2 27.7 -> 28.5 ms, the waves no longer retire by residence class; 1024^2 1.245 ->
  // 1.227 ms.  Not kept.)
  bool have = lane_sub < cnt;
  if (have) r = in[order[first + lane_sub]];
  uint32_t gate_ticket = 0;
  if (gate_lane) gate_ticket = gate_draw();
  if constexpr (SPARSE) {
    uint4* const t4 = reinterpret_cast<uint4*>(s_mem);
#pragma unroll
    for (int j = 0; j < (kWords / 4 + NT - 1) / NT; ++j) {
      const int i = tid + j * NT;
      const uint32_t w = i < kSparseTab / 4 ? kSparseEmpty : 0u;
      if (i < kWords / 4) t4[i] = make_uint4(w, w, w, w);
    }
    static_assert(kWords % 4 == 0 && kSparseTab % 4 == 0, "the table is cleared 16 bytes at a time");
  } else {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int c = tid + j * kBlock;
      if (kCells % NT != 0 && c >= kCells) break;
      s_a[kA * c] = 0.0f;
      if (KIND == FLUVIAL) s_a[2 * c + 1] = 0.0f;
      s_v[2 * c] = 0.0f;
      s_v[2 * c + 1] = 0.0f;
      if (ALB) s_c0[c] = s_c1[c] = s_c2[c] = 0.0f;
    }
  }
  __syncthreads();
  if (gate_lane) gate_look(gate_ticket);

  // Cells of this tile a particle may take a step on: the tile, cut down to the rows whose
  // stencil this slab holds (the whole grid on a single device) and to the grid's columns.
  // One unsigned comparison per axis then answers "out of the grid?", "escaped the slab?"
  // and "on another tile?" at once for a walker that carries on — the usual case; what
  // exactly ended a walk is sorted out once, when it ends.
  int r_lo = row0 > k.lo ? row0 : k.lo, r_hi = row0 + TR - 1 < k.hi ? row0 + TR - 1 : k.hi;
  int c_lo = col0 > 0 ? col0 : 0, c_hi = col0 + TC - 1 < k.W - 1 ? col0 + TC - 1 : k.W - 1;
  if (r_hi < r_lo || c_hi < c_lo) r_lo = r_hi = c_lo = c_hi = 0x40000000;  // nothing to step on
  const uint32_t r_span = static_cast<uint32_t>(r_hi - r_lo), c_span = static_cast<uint32_t>(c_hi - c_lo);
  const int row_org = k.x0 + r_lo;                // global row of local row r_lo
  // LDS cell of tile row `tr`, column `tc`.  A row of the tile is TC * 8 bytes of each pair plane — a
  // whole number of bank rounds — so the cells of one column would share their banks, and walkers that
  // follow a channel down a column would queue up on them.  Row r is rotated by kSkew * r columns
  // (kSkew a multiple of 4: the flush reads groups of four cells) : neighbours in any direction lie on
  // different banks.  SOIL_LDS_SKEW=0 at build time: rows as they are (A/B).
#ifndef SOIL_LDS_SKEW
#define SOIL_LDS_SKEW 4
#endif
  constexpr int kSkew = SOIL_LDS_SKEW;
  static_assert(kSkew % 4 == 0 && (TC & (TC - 1)) == 0, "tile columns: a power of two; skew: whole groups");
  auto tile_cell = [](int tr, int tc) {
    return (kSkew == 0 || SPARSE) ? tr * TC + tc : tr * TC + ((tc + kSkew * tr) & (TC - 1));
  };
  const int tr_org = r_lo - row0, tc_org = c_lo - col0;  // tile row / column of (r_lo, c_lo)
  const uint32_t l_org = static_cast<uint32_t>(r_lo) * k.Wu + static_cast<uint32_t>(c_lo);  // its local cell index
  const int last_iter = static_cast<int>(k.maxage) - 1;  // `++iter < maxage` passes while iter < maxage - 1

  // A lane that has to park its particle keeps the record in registers and idles;
  // the record is written out only when the lane takes another particle or the
  // loop is over: a tile starts a round with about one particle per lane, so
  // refills are the exception.
  // `more`: the queue holds entries beyond the ones handed out so far (uniform per wave)
  bool more = !SPARSE && cnt > static_cast<uint32_t>(kBlock), parked = false;  // (a sparse tile fits its lanes)
  constexpr int kRefillLanes = 16;
  auto write_out = [&]() {
    const uint32_t slot = atomicAdd(&s_out, 1u);  // slots this tile's queue occupied
    const uint32_t to = queue_key(k.x0, r.px, r.py, r.spx, r.spy, k.maxage - static_cast<uint32_t>(r.iter),
                                  tiles_w_next, ts_next, steps_per_round);
    out[first + slot] = r;
    dest[first + slot] = to;
    rank[first + slot] = atomicAdd(&count_next[to], 1u);
    parked = false;
  };
  // r.iter may grow up to `limit` in this round: the round's step budget, or the particle's age
  int limit = r.iter + steps_per_round < last_iter ? r.iter + steps_per_round : last_iter;
  uint32_t nsteps = 0;
  constexpr int kFluxPlanes = (KIND == FLUVIAL) ? 4 : 3;
  // what a particle adds to the cell it stands on, and where (LDS cell c): :104-113 / :310-318
  auto deposit_terms = [&](int c, float* v, float** pp) {
    if (KIND == FLUVIAL) {
      v[0] = r.a0 * r.s0, v[1] = r.a1 * r.s1, v[2] = r.a2 * r.svx, v[3] = r.a2 * r.svy;
      pp[0] = &s_a[2 * c], pp[1] = &s_a[2 * c + 1], pp[2] = &s_v[2 * c], pp[3] = &s_v[2 * c + 1];
    } else {
      // the pair (velocity x | y) first, the single plane after it: CasDeposit's order
      v[0] = r.a1 * r.svx, v[1] = r.a1 * r.svy, v[2] = r.a0 * r.s0;
      pp[0] = &s_v[2 * c], pp[1] = &s_v[2 * c + 1], pp[2] = &s_a[c];
    }
    if (ALB) {  // colour rides on the mass attenuation, :110-112 / :315-317
      const float att = (KIND == FLUVIAL) ? r.a1 : r.a0;
      v[kFluxPlanes] = att * r.sa0, v[kFluxPlanes + 1] = att * r.sa1, v[kFluxPlanes + 2] = att * r.sa2;
      pp[kFluxPlanes] = &s_c0[c], pp[kFluxPlanes + 1] = &s_c1[c], pp[kFluxPlanes + 2] = &s_c2[c];
    }
  };
  // SPARSE: what the walker adds, into the table slot of tile cell `c` (found or taken by a
  // compare-and-swap on the key; linear probing); `lcell`: the cell's index in the planes, for the
  // deposit that finds the table full around its hash
  auto sparse_deposit = [&](int c, uint32_t lcell, const float* v) {
    (void)c;  // keyed by the cell's index in the slab: four tiles may share the table (packed)
    uint32_t slot = (lcell * 2654435761u) >> (32 - kSparseTabBits);
    bool placed = false;
    for (int t = 0; t < retries; ++t) {  // (SPARSE: `retries` carries the probe count, kSparseProbe unless a test says otherwise)
      const uint32_t old = atomicCAS(&t_key[slot], kSparseEmpty, lcell);
      if (old == kSparseEmpty || old == lcell) {
        placed = true;
        break;
      }
      slot = (slot + 1u) & static_cast<uint32_t>(kSparseTab - 1);
    }
    if (placed) {
#pragma unroll
      for (int j = 0; j < kFluxPlanes; ++j) atomicAdd(&t_val[j * kSparseTab + slot], v[j]);
    } else if (KIND == FLUVIAL) {
      atomicAdd(&flux0[lcell], v[0]);
      atomicAdd(&flux1[lcell], v[1]);
      atomicAdd(&fluxV[lcell].x, v[2]);
      atomicAdd(&fluxV[lcell].y, v[3]);
    } else {
      atomicAdd(&fluxV[lcell].x, v[0]);
      atomicAdd(&fluxV[lcell].y, v[1]);
      atomicAdd(&flux0[lcell], v[2]);
    }
  };
  PROF_AT(8);  // prologue: flux tile zeroed, first records loaded
  for (;;) {
    if (more && !have) {  // take the next particle of this tile's queue
      const uint32_t i = atomicAdd(&s_next, 1u);
      if (i < cnt) {
        if (parked) write_out();
        r = in[order[first + i]];
        have = true;
        limit = r.iter + steps_per_round < last_iter ? r.iter + steps_per_round : last_iter;
      }
    }
    if (more) more = __builtin_amdgcn_readfirstlane(static_cast<int>(*const_cast<volatile uint32_t*>(&s_next))) < static_cast<int>(cnt);
    if (!any_lane(have)) break;
    PROF_AT(1);  // refill

    // ---- the stepping loop.  Top of the reference loop: while(!__oob(pos) && ++iter < maxage)
    // (:100 / :306), the slab check, "still on my tile, still within the round's budget?" — one
    // unsigned comparison per axis (see r_span above) and the step budget.  A lane whose walker
    // cannot step on just drops out (`run`); what stopped it is sorted out once, after the loop,
    // for all lanes of the wave together — inside the loop every wave would walk through that
    // rare path on almost every iteration (64 lanes, one stop per ~20 steps each).  NaN
    // coordinates (reference quirks) read as "out of range" here: floor_cell turns
    // them into INT_MAX.
    // The wave-level tests are ballots of plain comparisons combined with scalar mask
    // arithmetic (`runm`: lanes still stepping), the per-lane predicate comes back from the mask
    // (inverse ballot: free): the ballot of a combined predicate costs a v_cndmask + v_cmp pair.
    uint64_t runm = __builtin_amdgcn_ballot_w64(have);
    uint32_t ended = 0u;
#ifdef SOIL_STATS
    int st_owner = static_cast<int>(threadIdx.x & 63u);
#endif
    for (;;) {
#ifdef SOIL_PROF
      ++pt_iters;
#endif
      const uint32_t dr = static_cast<uint32_t>(floor_cell(r.px)) - static_cast<uint32_t>(row_org);  // row, column counted
      const uint32_t dc = static_cast<uint32_t>(floor_cell(r.py)) - static_cast<uint32_t>(c_lo);     // from (r_lo, c_lo); unsigned: floor_cell saturates, the wrap-around is meant
      const uint64_t stepm = __builtin_amdgcn_ballot_w64(dr <= r_span) & __builtin_amdgcn_ballot_w64(dc <= c_span) &
                             __builtin_amdgcn_ballot_w64(r.iter < limit) & runm;
      runm = stepm;
      // Nobody left to step — or, with a queue longer than the work-group (`more`), enough lanes of
      // the wave free to take new particles (enough: leaving the loop and coming back costs about as
      // much as two iterations): one population count against a bound that is 0 unless `more`.
      if (__popcll(stepm) <= (more ? 64 - kRefillLanes : 0)) break;
      PROF_AT(2);  // head
      CasDeposit<kFluxPlanes + (ALB ? 3 : 0), (KIND == FLUVIAL) ? 2 : 1> dep;
      uint32_t lost_bits = 0;
      const int c = tile_cell(tr_org + static_cast<int>(dr), tc_org + static_cast<int>(dc));  // LDS cell (any value when idle)
      // rows, W < 2^24 and H*W < 2^31 (use_tiled): one v_mad_u32_u24 per index
      const uint32_t lcell = l_org + __umul24(dr, k.Wu) + dc;
      const uint32_t nind = lcell + k.base;  // global cell: cx * W + cy, :103 / :309
      float v_norm = 1.0f;
#ifdef SOIL_STATS
      bool st_dep = false, st_stuck = false;
      int st_zero = 0;
#endif
      if (__builtin_amdgcn_inverse_ballot_w64(stepm)) {
        ++r.iter;
        ++nsteps;
        // the cell's record comes from the packed plane through L1/L2 (the tile's
        // 64 KiB are touched ~4x per round); issued first, the gather's latency
        // hides under the deposit and the other waves of the SIMD.  (Round 5, tried: the record of the
        // NEXT cell asked for as soon as the step's geometry gives it, one iteration ahead, checked and
        // asked for again when the guess was wrong — with the record switched off the fluvial launch
        // takes 16 % less, tools/ablate_fast.sh — : 8192^2 step 28.0 -> 29.3 ms fast, 30.7 -> 32.8 exact.
        // What the gather costs is its 64 lines per wave instruction through the vector cache, not
        // their latency; a second request in flight and the guesses that miss only add to that.)
        const float4 q = p4[ABLATED(8) ? l_org : lcell];
        bool deposit = false;
        if (nind != r.ind && !ABLATED(2)) {    // :104-113 / :310-318
          r.ind = nind;
#ifdef SOIL_STATS
          st_dep = true;
          st_zero = KIND == FLUVIAL ? ((r.a2 * r.svx == 0.0f && r.a2 * r.svy == 0.0f ? 1 : 0) | (r.a1 * r.s1 == 0.0f ? 2 : 0) |
                                       (r.a0 * r.s0 == 0.0f ? 4 : 0))
                                    : 0;
#endif
          // DEP 0: native ds_add_f32, fire and forget; DEP 1: CasDeposit
          float v[kFluxPlanes + 3];
          float* pp[kFluxPlanes + 3];
          deposit_terms(c, v, pp);
          if (KIND == DEBRIS && retire == 2u && r.a2 != 0.0f && !(v[0] == 0.0f && v[1] == 0.0f && v[2] == 0.0f)) {
            atomicAdd(&soil_retire_violations_dev, 1ull);  // a marked walker added something: must not happen
            RETIRE_DBG(r, 3.0f);
          }
          if constexpr (SPARSE) sparse_deposit(c, lcell, v);
#pragma unroll
          for (int j = 0; j < kFluxPlanes + (ALB ? 3 : 0); ++j) {
            if (SPARSE) break;
            if (ABLATED(4)) {
              *pp[j] = v[j];
            } else if (DEP == 1) {
              dep.p[j] = pp[j];
              dep.v[j] = v[j];
            } else {
              atomicAdd(pp[j], v[j]);
            }
          }
          if (DEP == 1 && !ABLATED(4) && !SPARSE) {
            if (KIND == FLUVIAL) {
              dep.load();  // the old words travel while the geometry of the step is worked out
            } else {
              dep.begin();
            }
            deposit = true;
          }
        }
        PROF_AT(3);  // gather issued, deposit begun
        StepGeom geom;                                           // needs neither q nor LDS
        if constexpr (FAST) {
          geom = step_geom_fast<KIND>(r, k);
          pin3(geom.v_step, geom.ds, geom.dL);                   // (the swaps go out behind the geometry)
        } else {
          geom = step_geom<KIND>(r, k);
        }
        // (round 3: swapping right behind the load, as the debris launch does, times the same — 24.07-24.14 ms
        // either way at 8192^2; round 6: nor does issuing the swaps in the middle of the geometry, once the
        // direction is known — 30.40-30.64 against 30.58-30.91 ms per step, A/B on one box)
        if (KIND == FLUVIAL && deposit) dep.swap_all();          // the swaps' round trip hides under step_apply
        v_norm = geom.v_norm;
#ifdef SOIL_STATS
        st_stuck = geom.dL == 0.0f && !(geom.v_norm < k.eps);  // a step of length zero: the walker stays where it is (step_geom: "stuck on a cell corner")
#endif
        // :121-122 / :326-327: a walk that is over shows in `ended` (a value, not a lane mask merged
        // through the branches of the iteration: `have` stays what it was when the loop began)
        if constexpr (FAST) {
          ended |= step_apply_fast<KIND>(r, q, k, geom) ? 0u : 1u;
          pin3(r.spx, r.spy, r.px);                              // (... and their answers are looked at behind the update)
        } else {
          ended |= step_apply<KIND>(r, q, k, geom) ? 0u : 1u;
        }
        if (deposit) lost_bits = dep.lost_bits();                // the swaps' answers, only now
        if (KIND == DEBRIS && retire != 0u) {
          // a spent walker (debris_spent) has nothing left to add: its walk ends at the top of the next
          // iteration — like the `v_norm < eps` exit, it just is not taken up again
          // (a walk that has just ended — v_norm < eps — computed on regardless, step_apply: its record may
          // hold anything and is dropped)
          const bool walks_on = !(v_norm < k.eps);
          const bool maybe = walks_on && r.a1 == 0.0f && r.a0 * r.s0 == 0.0f;
          if (any_lane(maybe)) {
            const bool spent = maybe && debris_spent(r);
            if (retire == 1u) {
              if (spent) {
                ended |= 1u;
                v_norm = 0.0f;  // (leaves `runm` with the lanes whose speed ran out, below)
              }
            } else {  // watched
              if (walks_on && r.a2 != 0.0f && !spent) { atomicAdd(&soil_retire_violations_dev, 1ull); RETIRE_DBG(r, 1.0f); }
              if (spent) r.a2 = 1.0f;
            }
          } else if (retire == 2u && walks_on && r.a2 != 0.0f) {
            atomicAdd(&soil_retire_violations_dev, 1ull);
            RETIRE_DBG(r, 2.0f);
          }
        }
        PROF_AT(4);  // the step's arithmetic
      }
      runm &= ~__builtin_amdgcn_ballot_w64(opaque(v_norm) < k.eps);  // ... the same exit, for the wave
#ifdef SOIL_STATS
      if (DEP == 1 && !SPARSE) {
        const bool dep_lane = st_dep;
        const int skey = dep_lane ? c : ~static_cast<int>(threadIdx.x & 63u);
        int same = 0;
        for (int l = 0; l < 64; ++l) same += __builtin_amdgcn_readlane(skey, l) == skey ? 1 : 0;
        const bool shares = dep_lane && same > 1, lostl = lost_bits != 0u;
        const uint64_t lm = __builtin_amdgcn_ballot_w64(lostl);
        STATS_ADD(0, 1);
        STATS_ADD(1, __popcll(stepm));
        STATS_ADD(2, __popcll(__builtin_amdgcn_ballot_w64(dep_lane)));
        STATS_ADD(4, lm != 0 ? 1 : 0);
        STATS_ADD(5, __popcll(lm));
        STATS_ADD(6, __popcll(__builtin_amdgcn_ballot_w64(lostl && shares)));
        STATS_ADD(7, __popcll(__builtin_amdgcn_ballot_w64(shares)));
        int distinct = 0;
        for (uint64_t t = lm; t != 0;) {
          const int c0 = __builtin_amdgcn_readlane(skey, __ffsll(static_cast<long long>(t)) - 1);
          t &= ~__builtin_amdgcn_ballot_w64(skey == c0);
          ++distinct;
        }
        {  // deposits whose velocity pair | mass | water term is an exact zero; iterations where every deposit's pair is
          const uint64_t dm = __builtin_amdgcn_ballot_w64(dep_lane), zv = __builtin_amdgcn_ballot_w64(dep_lane && (st_zero & 1));
          (void)dm; (void)zv;  // (round 6: 0.0 % in either launch; the slots count walkers that do not move instead)
          const uint64_t sm = __builtin_amdgcn_ballot_w64(st_stuck);
          STATS_ADD(10, __popcll(sm));                                  // lanes whose step has length zero
          STATS_ADD(11, sm != 0 && sm == stepm ? 1 : 0);                 // wave-iterations in which nobody else steps
          STATS_ADD(23, sm != 0 && __popcll(stepm & ~sm) <= 8 ? 1 : 0);  // ... in which at most 8 others do
          STATS_ADD(3, __popcll(stepm) <= 8 ? 1 : 0);                    // wave-iterations of at most 8 lanes
        }
        STATS_ADD(8, distinct);
        STATS_ADD(9, distinct == 1 ? 1 : 0);
        // distance to the nearest lower lane that deposits on the same cell: 1 | 2 | 3 | 4-7 | 8-15 | 16+ -> [16..21]
        int near = 0;
        const int me = static_cast<int>(threadIdx.x & 63u);
        for (int l = 0; l < 64; ++l) {
          const int kl = __builtin_amdgcn_readlane(skey, l);
          if (l < me && kl == skey) near = me - l;  // the last (highest) such lane below wins
        }
        const bool has = dep_lane && near > 0;
        STATS_ADD(12, __popcll(__builtin_amdgcn_ballot_w64(has && near == 1)));
        STATS_ADD(13, __popcll(__builtin_amdgcn_ballot_w64(has && near == 2)));
        STATS_ADD(14, __popcll(__builtin_amdgcn_ballot_w64(has && near == 3)));
        STATS_ADD(15, __popcll(__builtin_amdgcn_ballot_w64(has && near >= 4 && near < 8)));
        STATS_ADD(16, __popcll(__builtin_amdgcn_ballot_w64(has && near >= 8 && near < 16)));
        STATS_ADD(17, __popcll(__builtin_amdgcn_ballot_w64(has && near >= 16)));
        {  // the star of the iteration before (st_owner: the highest lane that deposited on my cell then) as a
           // prediction: a lane whose owner deposits on its cell again follows; who of the others still collides?
          const int ok = __builtin_amdgcn_ds_bpermute(st_owner << 2, skey);
          const bool st_follow = dep_lane && st_owner != me && ok == skey;
          const int rkey = (dep_lane && !st_follow) ? c : ~me;  // keys of the lanes that would still swap
          int rsame = 0, high = me;
          for (int l = 0; l < 64; ++l) {
            rsame += __builtin_amdgcn_readlane(rkey, l) == rkey ? 1 : 0;
            if (__builtin_amdgcn_readlane(skey, l) == skey) high = l;
          }
          st_owner = dep_lane ? high : me;
          const uint64_t still = __builtin_amdgcn_ballot_w64(rsame > 1);
          STATS_ADD(20, __popcll(__builtin_amdgcn_ballot_w64(st_follow)));   // predicted followers
          STATS_ADD(21, __popcll(still));                                      // lanes that still share a cell among the swappers
          STATS_ADD(22, still != 0 ? 1 : 0);                                   // iterations with such lanes
        }
        STATS_ADD(18, __popcll(__builtin_amdgcn_ballot_w64(has && near == 1 && (me & 1))));      // ... in its aligned pair
        STATS_ADD(19, __popcll(__builtin_amdgcn_ballot_w64(has && near <= (me & 3))));           // ... in its aligned quad
      }
#endif
#ifdef SOIL_ABLATE
      // (32, with 16: the swaps' answers waited for and counted, the losers dropped — the wait apart from the making up)
      if (DEP == 1 && !SPARSE && ABLATED(32)) nsteps += __popcll(__builtin_amdgcn_ballot_w64(opaque(lost_bits) == 0x12345u));
#endif
      if (DEP == 1 && !SPARSE && !ABLATED(16))  // (16: lost swaps dropped — what the lost path costs)
        dep.template finish<FAST && SOIL_RETRY_TOGETHER>(opaque(lost_bits), c, agg_min, agg_groups, retries);
      PROF_AT(5);  // deposit finished
    }
    const bool run = __builtin_amdgcn_inverse_ballot_w64(runm);
    have = have && opaque(ended) == 0u;

    // ---- once per particle and round: what stopped it?
    if (have && !run) {
      int ix = floor_cell(r.px), iy = floor_cell(r.py);
      if (r.px != r.px || r.py != r.py) {
        // A NaN walker: a NaN coordinate stands for cell 0 (the reference's float -> int
        // conversion) and is never out of bounds.  On the tile that holds that cell it walks
        // on, here, in the reference's own terms (0.02 % of all steps): it adds its sources to
        // the cell once and then idles there until it dies of age.
        for (;;) {
          ix = nan_cell(r.px, ix);
          iy = nan_cell(r.py, iy);
          const uint32_t dr = static_cast<uint32_t>(ix) - static_cast<uint32_t>(row_org), dc = static_cast<uint32_t>(iy) - static_cast<uint32_t>(c_lo);
          if (!(dr <= r_span && dc <= c_span && r.iter < limit)) break;
          ++r.iter;
          ++nsteps;
          const uint32_t lcell = l_org + __umul24(dr, k.Wu) + dc;
          const float4 q = p4[lcell];
          const uint32_t nind = lcell + k.base;
          if (nind != r.ind) {
            r.ind = nind;
            float v[kFluxPlanes + 3];
            float* pp[kFluxPlanes + 3];
            const int cn = tile_cell(tr_org + static_cast<int>(dr), tc_org + static_cast<int>(dc));
            deposit_terms(cn, v, pp);
            if constexpr (SPARSE) {
              sparse_deposit(cn, lcell, v);
            } else {
#pragma unroll
              for (int j = 0; j < kFluxPlanes + (ALB ? 3 : 0); ++j) atomicAdd(pp[j], v[j]);
            }
          }
          if (!advance<KIND, FAST>(r, q, k)) {  // :121-122 / :326-327
            have = false;
            break;
          }
          ix = floor_cell(r.px);
          iy = floor_cell(r.py);
        }
      }
      if (have) {
        have = false;
        // erosion_map.cu:29-40 on the floored coordinates (floor_cell, soil_math.hpp)
        const bool oob = static_cast<uint32_t>(ix) >= static_cast<uint32_t>(d.H) ||
                         static_cast<uint32_t>(iy) >= k.Wu;
        const int lx = ix - k.x0;
        const bool esc = lx < k.lo || lx > k.hi;  // slab_escape
        const bool aged = r.iter >= last_iter;
        // in the grid and with life left: its walk continues elsewhere — on another
        // rank (only a NaN walker's deposit travels, park_remote) or, state untouched,
        // in this slab's next round
        if (!oob && !aged) {
          if (esc) {
            // (a NaN walker stands for cell (0, 0) and never moves: only its deposit travels)
            if (box.count && !(r.px != r.px || r.py != r.py)) migrate_out(r, box, lx < k.lo);
            else park_remote<KIND>(r, remote0);
          } else {
            parked = true;
          }
        }
      }
    }
    PROF_AT(0);  // stops sorted out
  }
  {  // everything still parked goes out together (convergent: aggregate the counters)
    uint32_t slot;
    if (packed) slot = parked ? atomicAdd(&s_out_sub[sub], 1u) : 0u;  // (a handful of lanes)
    else slot = wave_append(&s_out, parked);
    const uint32_t dest_tile =
        parked ? queue_key(k.x0, r.px, r.py, r.spx, r.spy, k.maxage - static_cast<uint32_t>(r.iter),
                           tiles_w_next, ts_next, steps_per_round)
               : 0u;
    const uint32_t place = wave_key_append(count_next, parked, dest_tile);
    if (parked) {
      out[first + slot] = r;
      dest[first + slot] = dest_tile;
      rank[first + slot] = place;
    }
  }
  atomicAdd(&s_steps, nsteps);
  PROF_AT(6);  // survivors written out
  __syncthreads();
  PROF_AT(7);  // waiting for the slowest wave of the work-group
  done_draw();
  if (tid == 0) atomicAdd(steps, static_cast<unsigned long long>(s_steps));
  for (uint32_t j = (packed ? s_out_sub[sub] : s_out) + lane_sub; j < cnt; j += sub_lanes) dest[first + j] = kNoTile;  // unused slots

  // flush the tile's flux into the global planes: with one work-group per tile per
  // round plain read-modify-writes suffice; the groups of a split tile add atomically.  Only cells that received a
  // deposit are touched (late rounds: the particles sit in channels and most of
  // the tile is still zero).  All of a thread's cells are in flight at once: the old words of every
  // cell that changed are asked for before the first one is looked at (the stepping loop's
  // registers are free by now) — a work-group holds its tile's LDS for one round trip to the
  // planes, not for one per pair of cells (four of them: ~6 us of the ~40 a sparse tile's
  // work-group lives).
  auto flush_cell = [&](int j, int& c, int64_t& l) {  // -> inside the tile, the slab's rows and the grid's columns
    const int cc = tid + j * kBlock;                  // c: where the cell's accumulators live in LDS
    const int lx = row0 + cc / TC, y = col0 + cc % TC;
    c = tile_cell(cc / TC, cc % TC);
    l = static_cast<int64_t>(lx) * k.W + y;
    return cc < kCells && lx >= 0 && y >= 0 && lx < static_cast<int>(d.rows) && y < k.W;
  };
  // Four consecutive cells of a row per thread where the planes allow 16-byte accesses (the width a
  // multiple of four — tile columns and the half-tile shift are — and the planes aligned): the
  // flush is a few hundred instructions per thread cell by cell, issued at a lone wave's pace while the
  // work-group holds its tile; by groups of four it is a third of that.  A group none of whose
  // cells took a deposit is left alone; in a group that is read, a cell is added to exactly when the
  // cell-by-cell flush adds to it, so the planes hold the same bits either way.
  if constexpr (SPARSE) {
    // The table's cells go out one by one: the old words of a thread's occupied slots asked for together
    // (device-scope loads: a deposit that found the table full has added to the planes behind this
    // CU's vector cache), then added to and stored.  Nobody else touches this tile's cells in this
    // round, and the round's dense kernel starts when this one is over.
    constexpr int kTPer = (kSparseTab + NT - 1) / NT;
    float g[kTPer][kFluxPlanes];
    uint32_t key[kTPer];
#pragma unroll
    for (int j = 0; j < kTPer; ++j) {
      const int i = tid + j * NT;
      key[j] = i < kSparseTab ? t_key[i] : kSparseEmpty;
#pragma unroll
      for (int q = 0; q < kFluxPlanes; ++q) g[j][q] = 0.0f;
      if (key[j] == kSparseEmpty) continue;
      const int64_t l = static_cast<int64_t>(key[j]);  // the cell's index in the slab
      float* const fv = reinterpret_cast<float*>(fluxV + l);
      if (KIND == FLUVIAL) {
        g[j][0] = __hip_atomic_load(flux0 + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g[j][1] = __hip_atomic_load(flux1 + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g[j][2] = __hip_atomic_load(fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g[j][3] = __hip_atomic_load(fv + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        g[j][0] = __hip_atomic_load(fv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g[j][1] = __hip_atomic_load(fv + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g[j][2] = __hip_atomic_load(flux0 + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#pragma unroll
    for (int j = 0; j < kTPer; ++j) {
      const int i = tid + j * NT;
      if (key[j] == kSparseEmpty) continue;
      const int64_t l = static_cast<int64_t>(key[j]);
      float a[kFluxPlanes];
#pragma unroll
      for (int q = 0; q < kFluxPlanes; ++q) a[q] = t_val[q * kSparseTab + i];
      // (the dense flush's conditions, cell for cell: a plane is added to where its sum is not zero,
      // the velocity pair where either half is)
      if (KIND == FLUVIAL) {
        if (a[0] != 0.0f) flux0[l] = g[j][0] + a[0];
        if (a[1] != 0.0f) flux1[l] = g[j][1] + a[1];
        if (a[2] != 0.0f || a[3] != 0.0f) fluxV[l] = make_float2(g[j][2] + a[2], g[j][3] + a[3]);
      } else {
        if (a[2] != 0.0f) flux0[l] = g[j][2] + a[2];
        if (a[0] != 0.0f || a[1] != 0.0f) fluxV[l] = make_float2(g[j][0] + a[0], g[j][1] + a[1]);
      }
    }
  }
  const bool flush_vec = !SPARSE && (k.W & 3) == 0 && !shared_tile && !ALB &&
                         ((reinterpret_cast<uintptr_t>(flux0) | reinterpret_cast<uintptr_t>(flux1) |
                           reinterpret_cast<uintptr_t>(fluxV)) & 15u) == 0;
  if (flush_vec) {
    constexpr int kGroups = kCells / 4, kGPer = (kGroups + NT - 1) / NT;
    static_assert(kCells % 4 == 0 && TC % 4 == 0, "a group of four cells lies in one row of the tile");
    const float4* const sa4 = reinterpret_cast<const float4*>(s_a);
    const float4* const sv4 = reinterpret_cast<const float4*>(s_v);
    auto group_at = [&](int j, int& gi, int64_t& l) {  // gi: the group's place in LDS (rows rotated, tile_cell)
      const int gg = tid + j * kBlock;
      const int c = 4 * gg;
      const int lx = row0 + c / TC, y = col0 + c % TC;
      gi = tile_cell(c / TC, c % TC) >> 2;
      l = static_cast<int64_t>(lx) * k.W + y;
      return gg < kGroups && lx >= 0 && y >= 0 && lx < static_cast<int>(d.rows) && y < k.W;
    };
    auto nz = [](float4 v) { return ((f2bits(v.x) | f2bits(v.y) | f2bits(v.z) | f2bits(v.w)) & 0x7fffffffu) != 0u; };
    if (store_all) {  // (see below)
#pragma unroll
      for (int j = 0; j < kGPer; ++j) {
        int gi;
        int64_t l;
        if (!group_at(j, gi, l)) continue;
        if (KIND == FLUVIAL) {
          const float4 a = sa4[2 * gi], b = sa4[2 * gi + 1];  // water | mass of cells 0 1, 2 3
          *reinterpret_cast<float4*>(flux0 + l) = make_float4(a.x, a.z, b.x, b.z);
          *reinterpret_cast<float4*>(flux1 + l) = make_float4(a.y, a.w, b.y, b.w);
        } else {
          *reinterpret_cast<float4*>(flux0 + l) = sa4[gi];
        }
        *reinterpret_cast<float4*>(fluxV + l) = sv4[2 * gi];
        *reinterpret_cast<float4*>(fluxV + l + 2) = sv4[2 * gi + 1];
      }
    } else {
      float4 g0[kGPer], g1[kGPer], gva[kGPer], gvb[kGPer];
      bool hit[kGPer];
#pragma unroll
      for (int j = 0; j < kGPer; ++j) {  // every old word asked for before the first one is looked at
        int gi;
        int64_t l;
        hit[j] = group_at(j, gi, l);
        if (hit[j])
          hit[j] = (KIND == FLUVIAL ? (nz(sa4[2 * gi]) || nz(sa4[2 * gi + 1])) : nz(sa4[gi])) ||
                   nz(sv4[2 * gi]) || nz(sv4[2 * gi + 1]);
        g0[j] = g1[j] = gva[j] = gvb[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (hit[j]) {
          g0[j] = *reinterpret_cast<const float4*>(flux0 + l);
          if (KIND == FLUVIAL) g1[j] = *reinterpret_cast<const float4*>(flux1 + l);
          gva[j] = *reinterpret_cast<const float4*>(fluxV + l);
          gvb[j] = *reinterpret_cast<const float4*>(fluxV + l + 2);
        }
      }
#pragma unroll
      for (int j = 0; j < kGPer; ++j) {
        int gi;
        int64_t l;
        (void)group_at(j, gi, l);
        if (!hit[j]) continue;
        auto add = [](float g, float a) { return a != 0.0f ? g + a : g; };
        auto add2 = [](float g, float a, float other) { return (a != 0.0f || other != 0.0f) ? g + a : g; };
        if (KIND == FLUVIAL) {
          const float4 a = sa4[2 * gi], b = sa4[2 * gi + 1];
          *reinterpret_cast<float4*>(flux0 + l) =
              make_float4(add(g0[j].x, a.x), add(g0[j].y, a.z), add(g0[j].z, b.x), add(g0[j].w, b.z));
          *reinterpret_cast<float4*>(flux1 + l) =
              make_float4(add(g1[j].x, a.y), add(g1[j].y, a.w), add(g1[j].z, b.y), add(g1[j].w, b.w));
        } else {
          const float4 a = sa4[gi];
          *reinterpret_cast<float4*>(flux0 + l) =
              make_float4(add(g0[j].x, a.x), add(g0[j].y, a.y), add(g0[j].z, a.z), add(g0[j].w, a.w));
        }
        const float4 va = sv4[2 * gi], vb = sv4[2 * gi + 1];  // x | y of cells 0 1, 2 3
        *reinterpret_cast<float4*>(fluxV + l) = make_float4(add2(gva[j].x, va.x, va.y), add2(gva[j].y, va.y, va.x),
                                                            add2(gva[j].z, va.z, va.w), add2(gva[j].w, va.w, va.z));
        *reinterpret_cast<float4*>(fluxV + l + 2) = make_float4(add2(gvb[j].x, vb.x, vb.y), add2(gvb[j].y, vb.y, vb.x),
                                                                add2(gvb[j].z, vb.z, vb.w), add2(gvb[j].w, vb.w, vb.z));
      }
    }
  } else if (SPARSE) {
    // (done above)
  } else if (store_all) {
    // The first round of a launch that was told to OVERWRITE the flux planes (soil_erode_step's
    // lazy mode: the cell phase did not re-zero them): every tile has exactly one work-group, the
    // tiles partition the plane, so plain stores of the tile's accumulators — zeros included —
    // leave the planes holding this round's deposits and nothing else, without a read.
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      int c;
      int64_t l;
      if (!flush_cell(j, c, l)) continue;
      flux0[l] = s_a[kA * c];
      if (KIND == FLUVIAL) flux1[l] = s_a[2 * c + 1];
      fluxV[l] = make_float2(s_v[2 * c], s_v[2 * c + 1]);
    }
  } else if (shared_tile) {  // uniform per work-group
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      int c;
      int64_t l;
      if (!flush_cell(j, c, l)) continue;
      const float a0 = s_a[kA * c], a1 = (KIND == FLUVIAL) ? s_a[2 * c + 1] : 0.0f;
      const float ax = s_v[2 * c], ay = s_v[2 * c + 1];
      if (a0 != 0.0f) atomicAdd(&flux0[l], a0);
      if (KIND == FLUVIAL && a1 != 0.0f) atomicAdd(&flux1[l], a1);
      if (ax != 0.0f) atomicAdd(&fluxV[l].x, ax);
      if (ay != 0.0f) atomicAdd(&fluxV[l].y, ay);
    }
  } else {
    float g0[kPer], g1[kPer];
    float2 gv[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      int c;
      int64_t l;
      const bool ok = flush_cell(j, c, l);
      g0[j] = g1[j] = 0.0f;
      gv[j] = make_float2(0.0f, 0.0f);
      if (!ok) continue;
      if (s_a[kA * c] != 0.0f) g0[j] = flux0[l];
      if (KIND == FLUVIAL && s_a[2 * c + 1] != 0.0f) g1[j] = flux1[l];
      if (s_v[2 * c] != 0.0f || s_v[2 * c + 1] != 0.0f) gv[j] = fluxV[l];
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      int c;
      int64_t l;
      if (!flush_cell(j, c, l)) continue;
      const float a0 = s_a[kA * c], a1 = (KIND == FLUVIAL) ? s_a[2 * c + 1] : 0.0f;
      const float ax = s_v[2 * c], ay = s_v[2 * c + 1];
      if (a0 != 0.0f) flux0[l] = g0[j] + a0;
      if (KIND == FLUVIAL && a1 != 0.0f) flux1[l] = g1[j] + a1;
      if (ax != 0.0f || ay != 0.0f) fluxV[l] = make_float2(gv[j].x + ax, gv[j].y + ay);
    }
  }
  if (ALB) {  // the three colour planes, AoS (vec3) in global memory
    for (int cc = tid; cc < kCells; cc += kBlock) {
      const int lx = row0 + cc / TC, y = col0 + cc % TC;
      if (lx < 0 || y < 0 || lx >= static_cast<int>(d.rows) || y >= k.W) continue;
      const int64_t l3 = 3 * (static_cast<int64_t>(lx) * k.W + y);
      const int sc = tile_cell(cc / TC, cc % TC);
      const float c0 = s_c0[sc], c1 = s_c1[sc], c2 = s_c2[sc];
      if (c0 == 0.0f && c1 == 0.0f && c2 == 0.0f) continue;
      if (shared_tile) {
        atomicAdd(&fluxA[l3], c0);
        atomicAdd(&fluxA[l3 + 1], c1);
        atomicAdd(&fluxA[l3 + 2], c2);
      } else {
        fluxA[l3] += c0;
        fluxA[l3 + 1] += c1;
        fluxA[l3 + 2] += c2;
      }
    }
  }
  PROF_FLUSH(KIND);  // [9]: flush of the tile's flux
  if (!SPARSE) STATS_FLUSH(KIND);
  } while (false);

  // The work-group that finishes last scans the queues of the round that follows: no launch of its
  // own for the scan (it used to wait for a CU with 139 KiB of free LDS behind the other launch's
  // round kernel: 0.24 ms per scan in the overlapped step), no launch gap on either side of it.
  // No fences: a device-scope release / acquire writes back and invalidates the L2 of the XCD (the
  // eight L2s are not coherent with each other) — per work-group that cost 20 % of the launch.  None
  // is needed: all the scan reads of this round are the section counts, and those are only ever
  // touched by device-scope atomics, which are carried out at the memory side; every one of them is
  // a returning atomic whose answer this work-group has stored by now (barrier above), so it is
  // done before the ticket is drawn; the scan reads the counts with device-scope loads.  Everything
  // else the round writes (records, ranks, flux) is for later kernels.
  if constexpr (SPARSE) return;  // (the round's ticket and scan belong to the dense kernel behind this one)
  if (!next.host) return;  // uniform: the scan is a launch of its own (SOIL_TILED_TAILSCAN=2)
  if (!done_drawn) done_draw();  // (an empty chunk of a cut queue)
  if (tid == 0) s_last = done_ticket == n_groups - 1u ? 1u : 0u;
  __syncthreads();  // ... and every thread is done with the tile: the scan's scratch is the tile's LDS
  if (s_last != 0u) {
    if (tid == 0) {
      __hip_atomic_store(&ctl->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctl->started, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    queue_scan_dev<NT>(next, reinterpret_cast<uint32_t*>(s_mem));
  }
}

// ---- the last launch: walk the remaining particles to the end against HBM ----------

template <int KIND, bool FAST = false>
__global__ void __launch_bounds__(256)
    k_tiled_finish(MigrateBox box, const PRec* __restrict__ recs, const uint32_t* __restrict__ dest,
                   const TiledCtl* __restrict__ ctl, float* __restrict__ flux0,
                   float* __restrict__ flux1, float* __restrict__ fluxV, float* __restrict__ fluxA,
                   const float4* __restrict__ p4, float* __restrict__ remote0,
                   unsigned long long* __restrict__ steps, Dom d, Scale3 s, Param param) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  if (ctl->mode != 1 || i >= static_cast<int64_t>(ctl->slots)) return;
  if (dest[i] == kNoTile) return;
  PRec r = recs[i];
  uint32_t nsteps = 0;
  const StepConst k = make_const<KIND>(d, s, param);
  const int64_t base = static_cast<int64_t>(k.x0) * k.W;
  const uint32_t retire = (KIND == DEBRIS) ? ctl->retire : 0u;
  // (Round 4: this launch is as long as the longest walk left — 0.74 / 0.80 ms at the end of every
  // 8192^2 step, ~3 us per step of a straggler with 250 steps to go.  Asking for the neighbouring rows'
  // records an iteration ahead, with streaming or with plain loads, changed nothing.  Round 5: nor is it
  // the returnless atomics, as round 4 supposed.  The listing has a vmcnt(0) at the loop's head (the record's
  // loads are first looked at there) and one in front of the cell record's use (the deposits stand under a
  // branch), so every step did wait for its atomics — but with the record waited for before the loop and
  // the deposits issued on every step behind the gather (adding -0.0f where the walker has not left its
  // cell: `vmcnt(4)` in front of the record's use) the two launches took 835 / 764 us against 834 / 715, and
  // with everything waited for BEFORE the step's arithmetic 899 / 881: the atomics are acknowledged within
  // the arithmetic's time either way.  A step is its gather's latency — a sector of a 1 GiB plane nobody
  // else has touched, ~2.5 us with the translation — plus ~0.6 us of arithmetic; exact or fast arithmetic
  // changes the launch by 5 %.  Removed again.)
  for (;;) {
    if (r.px < 0 || r.py < 0 || r.px >= k.Hf || r.py >= k.Wf) break;
    const int cx = cell32(r.px), cy = cell32(r.py);
    const int lx = cx - k.x0;
    const bool esc = lx < k.lo || lx > k.hi;
    if (static_cast<uint32_t>(++r.iter) >= k.maxage) break;
    if (esc) {
      if (box.count && !(r.px != r.px || r.py != r.py)) {
        --r.iter;  // handed over as it stood at the top of this iteration
        migrate_out(r, box, lx < k.lo);
      } else {
        park_remote<KIND>(r, remote0);
      }
      break;
    }
    ++nsteps;
    const int64_t nind = static_cast<int64_t>(cx) * k.W + cy;
    const int64_t l = nind - base;
    const float4 q = p4[l];
    if (static_cast<uint32_t>(nind) != r.ind) {
      r.ind = static_cast<uint32_t>(nind);
      if (KIND == FLUVIAL) {
        atomicAdd(&flux0[l], r.a0 * r.s0);
        atomicAdd(&flux1[l], r.a1 * r.s1);
        atomicAdd(&fluxV[2 * l], r.a2 * r.svx);
        atomicAdd(&fluxV[2 * l + 1], r.a2 * r.svy);
      } else {
        if (retire == 2u && r.a2 != 0.0f && !(r.a0 * r.s0 == 0.0f && r.a1 * r.svx == 0.0f && r.a1 * r.svy == 0.0f))
          atomicAdd(&soil_retire_violations_dev, 1ull);
        atomicAdd(&flux0[l], r.a0 * r.s0);
        atomicAdd(&fluxV[2 * l], r.a1 * r.svx);
        atomicAdd(&fluxV[2 * l + 1], r.a1 * r.svy);
      }
      if (fluxA) {
        const float att = (KIND == FLUVIAL) ? r.a1 : r.a0;
        atomicAdd(&fluxA[3 * l], att * r.sa0);
        atomicAdd(&fluxA[3 * l + 1], att * r.sa1);
        atomicAdd(&fluxA[3 * l + 2], att * r.sa2);
      }
    }
    if (!advance<KIND, FAST>(r, q, k)) break;
    if (KIND == DEBRIS && retire != 0u) {  // (the round kernel's rule)
      const bool spent = debris_spent(r);
      if (retire == 1u) {
        if (spent) break;
      } else {
        if (r.a2 != 0.0f && !spent) { atomicAdd(&soil_retire_violations_dev, 1ull); RETIRE_DBG(r, 4.0f); }
        if (spent) r.a2 = 1.0f;
      }
    }
  }
  atomicAdd(steps, static_cast<unsigned long long>(nsteps));  // one atomic per wave
}

static int env_int(const char* name, int fallback) {
  const char* e = std::getenv(name);
  const int v = e ? std::atoi(e) : fallback;
  return v > 0 ? v : fallback;
}
// NAME_F / NAME_D (fluvial / debris launches only) take precedence over NAME
static int env_kind(const char* name, int kind, int fallback) {
  char buf[96];
  std::snprintf(buf, sizeof(buf), "%s_%c", name, kind == FLUVIAL ? 'F' : 'D');
  return env_int(buf, env_int(name, fallback));
}

__global__ void k_fold_steps(unsigned long long* total, const unsigned long long* part) {
  atomicAdd(total, *part);
}

// Work-group shapes a round can use: tile rows x columns and threads.
// Every round re-sorts the particles by tile, so the shape may change from round to
// round (SOIL_TILED_LATE / SOIL_TILED_SWITCH).  Smaller tiles (32x64, 32x32) were
// measured too: more exits per step and no better occupancy.  What bounds the kernel is the
// number of walkers a CU has in flight — one per 8 cells of LDS-resident tile — so the tile
// heights are the ones that fill the 160 KiB of a CU with a whole number of tiles:
//   fluvial (16 B/cell): 2 x 78 rows (79 still fit);  debris (12 B/cell): 3 x 68 (69 fit; with 70
//   the CU takes only two work-groups although 3 x 53 772 B < 160 KiB: LDS is handed out in blocks
//   of 1280 B).  Measured at 8192^2 against 64 rows: fluvial 25.9 -> 25.0 ms, debris 11.2 -> 11.0 ms
//   (a quarter more walkers in flight buys 4 %: the round kernel is not bound by concurrency alone,
//   DESIGN.md 3.2).  Also measured, and not kept: 3 x 52 rows x 512 lanes (fluvial 25.5), 4 x 52 x 384
//   (debris 13.3), and small tiles for the sparse late rounds (32x32 x 64 lanes, 32x64 x 128: debris
//   11.9-14.3 ms, fluvial 26.0-27.2 whatever the round they take over from).
//   Round 3: debris on 104 rows x 768 lanes, two per CU like the fluvial tiles and the same
//   79 872 B of LDS each (any mix of the two kernels fills a CU): with 40-step rounds 11.05 -> 10.78 ms
//   per launch, the overlapped 8192^2 step 34.1-34.4 -> 33.9-34.0 ms on one box (fewer tile edges to
//   park at; the walkers in flight are the same 1664 per CU as with three 68-row tiles).
struct RoundShape { int tr, tc, nt; };
constexpr int kShapeColour = 2;  // serves the launches that carry colour: 7 / 6 LDS planes, one work-group per CU
constexpr int kShapeFull = 3;    // the LDS-filling tiles, two per CU of either kind: fluvial 78 rows, debris 104 rows, 768 lanes
constexpr int kNumShapes = 4;
template <int KIND>
struct Shapes {
  static constexpr RoundShape v[kNumShapes] = {
      {64, 64, 512}, {64, 64, 768}, {64, 64, 1024},
      KIND == FLUVIAL ? RoundShape{78, 64, 768} : RoundShape{104, 64, 768}};
};

// `fast`: the step in fast arithmetic (step_geom_fast; soil_set_particle_arith).  Instantiated for the
// compare-and-swap deposits without colour planes — what every launch of N >= 45 000 particles runs
// unless it carries colour; the other variants stay on the exact step.
template <int KIND, int DEP, int SH, bool ALB, typename... A>
static void launch_shape(bool fast, unsigned grid, hipStream_t st, A... a) {
  constexpr RoundShape S = Shapes<KIND>::v[SH];
  static_assert(round_lds_bytes(KIND, S.tr, S.tc, ALB) <= kLdsPerCU, "tile does not fit the LDS");
  if constexpr (DEP == 1 && !ALB) {
    if (fast) {
      k_tiled_round<KIND, DEP, S.tr, S.tc, S.nt, ALB, false, true><<<grid, S.nt, 0, st>>>(a...);
      return;
    }
  }
  k_tiled_round<KIND, DEP, S.tr, S.tc, S.nt, ALB><<<grid, S.nt, 0, st>>>(a...);
}
// the one-wave kernel of the sparse tiles of a round (same tile geometry as the round's dense shape)
template <int KIND, int SH, typename... A>
static void launch_sparse_shape(bool fast, unsigned grid, hipStream_t st, A... a) {
  constexpr RoundShape S = Shapes<KIND>::v[SH];
  if (fast) k_tiled_round<KIND, 1, S.tr, S.tc, kSparseLanes, false, true, true><<<grid, kSparseLanes, 0, st>>>(a...);
  else k_tiled_round<KIND, 1, S.tr, S.tc, kSparseLanes, false, true><<<grid, kSparseLanes, 0, st>>>(a...);
}
template <int KIND, typename... A>
static void launch_sparse(bool fast, int shape, unsigned grid, hipStream_t st, A... a) {
  if (shape == kShapeFull) launch_sparse_shape<KIND, kShapeFull>(fast, grid, st, a...);
  else launch_sparse_shape<KIND, 0>(fast, grid, st, a...);  // shapes 0 and 1: 64 x 64 tiles
}
// work-groups of a shape's kernel one CU holds, as the runtime sees it
template <int KIND, int SH, bool ALB>
static int shape_occupancy() {
  constexpr RoundShape S = Shapes<KIND>::v[SH];
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_tiled_round<KIND, 1, S.tr, S.tc, S.nt, ALB>, S.nt, 0) != hipSuccess) {
    (void)hipGetLastError();
    n = round_groups_per_cu(KIND, S.tr, S.tc, S.nt, ALB);
  }
  return n;
}
template <int KIND>
static int occupancy_of(int shape) {
  switch (shape) {
    case 1: return shape_occupancy<KIND, 1, false>();
    case kShapeColour: return shape_occupancy<KIND, kShapeColour, true>();
    case kShapeFull: return shape_occupancy<KIND, kShapeFull, false>();
    default: return shape_occupancy<KIND, 0, false>();
  }
}
template <int KIND, int DEP, typename... A>
static void launch_round(bool fast, int shape, unsigned grid, hipStream_t st, A... a) {
  switch (shape) {
    case 1: launch_shape<KIND, DEP, 1, false>(fast, grid, st, a...); break;
    case kShapeColour:
      if constexpr (DEP == 1)  // colour only with the compare-and-swap deposits
        launch_shape<KIND, 1, kShapeColour, true>(false, grid, st, a...);
      break;
    case kShapeFull: if constexpr (DEP == 1) launch_shape<KIND, 1, kShapeFull, false>(fast, grid, st, a...); break;
    default: launch_shape<KIND, DEP, 0, false>(fast, grid, st, a...); break;
  }
}

// One tiled launch as a resumable object: begin() queues the pre-pass, the spawn and
// the first queue scan; every advance() waits for the scan of the round that is about
// to start (its queue length and the step counter travel to pinned host memory),
// decides between another round and the finishing launch, and queues that work plus
// the next scan.  Between two advance() calls the stream stays busy, so two runs on
// two streams (fluvial and debris of one step, run_pair below) overlap: while the
// host waits for one, the other's kernels fill the SIMD slots the first leaves idle
// in its sparse late rounds.
template <int KIND>
struct TiledRun {
  // arguments
  float *flux0, *flux1, *fluxV;
  float* fluxA = nullptr;                // colour flux (vec3), optional
  const float* albedoSource = nullptr;   // colour of the cell a particle starts on
  Streams rng;
  int64_t N;
  const float *layers, *waterSource, *waterHeight, *velocity;
  float* remote0;
  Dom d;
  Scale3 s;
  Param p;
  hipStream_t st;
  // tuning (see the comments at their definitions in setup())
  int steps_per_round = 32, deposit = 0, shape_early = 0, shape_late = 0, switch_round = 1 << 30;
  int64_t tail = 0;
  double finish_rate = 4.0e9;
  bool verbose = false;
  // workspace
  PRec *cur = nullptr, *next = nullptr;
  uint32_t *dest = nullptr, *rank = nullptr, *order = nullptr, *count = nullptr,
           *count_next = nullptr, *start = nullptr, *tile_order = nullptr;
  uint4* block_list = nullptr;  // (tile, first, count, shared) of every work-group of the round
  float4* p4 = nullptr;
  unsigned long long *steps_global = nullptr, *steps_run = nullptr;
  size_t b_cnt = 0;
  TiledHostWord *host = nullptr, *host_dev = nullptr;  // the same pinned word, host / device view
  TiledCtl* ctl = nullptr;      // the device's own state of the chain of rounds
  // progress.  The host runs ahead of the device: `scans` scans (each with its slot sort) and `rounds`
  // round kernels are queued, `seen` scan words have been read; round r is only ever queued behind
  // scan r, and what a queued round finds to do is the device's decision (TiledCtl).
  uint64_t round = 0;           // = rounds (kept under its old name: the pair driver reads it)
  uint64_t scans = 0, seen = 0;
  int depth = 2;                // rounds queued beyond the last word seen (SOIL_TILED_AHEAD)
  bool tail_scan = true;        // the scan of a round at the tail of the round kernel before it (SOIL_TILED_TAILSCAN=2: a launch of its own)
  PairGate* gate = nullptr;     // taking turns with the other launch of the step (launch_pair_tiled); me = KIND
  uint32_t* seq_ctr = nullptr;  // number of the last k_queue_prepare launch (TiledHostWord::seq)
  uint32_t seq_first = 0;       // ... of this run's scan 0
  int64_t live_known = 0;       // an upper bound of the record slots in use: the last live count seen
  double ticks_per_second = 1.0e8;
  bool done = false;
  bool ready = false, skip_pack = false;  // setup() done; p4 filled by k_tiled_pack_pair
  // The flux planes hold stale values on entry (the cell phase left them as they were): the launch
  // must leave them holding its deposits only.  Round 0 stores instead of adding where it can (one
  // work-group per tile, none empty, no colour planes); otherwise the planes are cleared first.
  bool overwrite = false;
  int resident_groups[2] = {512, 512};  // work-groups of a round kernel the chip holds at once (early, late shape)

  int shape_of(uint64_t r) const { return r >= static_cast<uint64_t>(switch_round) ? shape_late : shape_early; }
  // steps a walker may take in round r: `steps_late` from round `steps_late_from` on (SOIL_TILED_STEPS_LATE,
  // SOIL_TILED_LATE_FROM; off by default)
  int steps_late = 0, steps_late_from = 1 << 30;
  int steps_of(uint64_t r) const { return (steps_late > 0 && r >= static_cast<uint64_t>(steps_late_from)) ? steps_late : steps_per_round; }
  // the tile grid of round r: shifted by half a tile on odd rounds (TileShape)
  bool stagger = true;
  bool sparse_ok = false;  // rounds >= 1 may hand their sparse tiles to the one-wave kernel (SOIL_TILED_SPARSE=2: off)
  int sparse_min = 64, sparse_pct = 25;  // SOIL_TILED_SPARSE_MIN, _PCT: see QueueScan
  bool sparse_pack = true;               // SOIL_TILED_SPARSE_PACK=2: one tile per wave whatever it holds
  int sparse_probe = kSparseProbe;       // SOIL_TILED_SPARSE_PROBE: slots a deposit tries before it adds to the planes directly
  uint32_t pair_free_below = 0;          // SOIL_PAIR_FREE (per cent of N): rounds of fewer walkers pass the gate (k_pair_gate)
  int host_lag_us = 0;                   // SOIL_TILED_HOST_LAG_US (tests): the host sleeps that long before every look at a word
  int agg_min = 48, agg_groups = 4, retries = 2;
  uint32_t* retire_bad = nullptr;  // debris: set by the pack pass when a cell rules retirement out (debris_cell_bad)
  uint32_t retire_mode = 0;        // debris: soil_set_debris_retire, if the launch constants allow it
  bool fast = false;  // the step in fast arithmetic (soil_set_particle_arith; not with colour planes or native adds)
  MigrateBox box{};                 // where walkers that leave the launch's rows go (null count: dropped, as ever)
  const PRec* inbox = nullptr;      // the launch starts from these records instead of the streams' spawns
  uint32_t n_in = 0;
  PRec* recs_of(uint64_t r) const { return (r & 1) ? next : cur; }               // records round r reads
  uint32_t* count_of(uint64_t r) const { return (r & 1) ? count_next : count; }   // section counts round r's scan reads
  TileShape ts_of(int sh, uint64_t r) const {
    const bool odd = stagger && (r & 1);
    return TileShape{Shapes<KIND>::v[sh].tr, __builtin_ctz(Shapes<KIND>::v[sh].tc),
                     odd ? Shapes<KIND>::v[sh].tr / 2 : 0, odd ? Shapes<KIND>::v[sh].tc / 2 : 0};
  }
  int tiles_w_of(int sh, uint64_t r) const {
    return static_cast<int>((d.W + ts_of(sh, r).off_c + Shapes<KIND>::v[sh].tc - 1) / Shapes<KIND>::v[sh].tc);
  }
  int64_t tiles_of(int sh, uint64_t r) const {
    return static_cast<int64_t>(tiles_w_of(sh, r)) *
           ((d.rows + ts_of(sh, r).off_r + Shapes<KIND>::v[sh].tr - 1) / Shapes<KIND>::v[sh].tr);
  }

  static int64_t resident_groups_hint() {  // work-groups of the LDS-filling round kernel the chip holds (2 per CU)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      (void)hipGetLastError();
      cus = 256;
    }
    return 2 * static_cast<int64_t>(cus);
  }
  int setup() {
    // steps a particle may take per round: bounds the time a work-group waits for
    // its longest walker
    // (fluvial walkers live out their 256 steps and stay on a 78-row tile longer than debris
    // walkers do: 8192^2, ms per launch at 24 / 28 / 32 / 36 / 40 / 48 / 64 steps: fluvial 27.6 26.5 24.9
    // 24.1 23.8 23.8 23.7, debris 11.6 11.1 10.9 10.95 11.0 11.05 11.3)
    // the finishing launch pays ~4 L2 atomics per step (22.7 G/s), a round a fixed
    // cost that grows with the number of tiles: N/40 within [4096, 200000] is where
    // they cross for 512^2 .. 8192^2 grids with N = cells/8
    const int tail_env = env_int("SOIL_TILED_TAIL", 0);
    tail = tail_env > 0 ? tail_env : std::min<int64_t>(200000, std::max<int64_t>(4096, N / 40));
    deposit = env_int("SOIL_TILED_DEP", 0);
    // measured at 8192^2 (N = cells/8): 768 threads on a 64x64 tile serve the fluvial
    // queues (about half of them hold 513..700 particles) in one batch, 45 vs 48 ms;
    // the debris kernel keeps 3 work-groups of 512 per CU instead, 17 vs 20 ms.  The
    // LDS-filling tiles (kShapeFull) where the grid has tiles enough to fill the chip twice
    // over with them; smaller grids want more, smaller work-groups.
    int cus = 256;
    {
      int dev = 0;
      SOIL_HIP(hipGetDevice(&dev));
      SOIL_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    }
    {
      constexpr RoundShape F = Shapes<KIND>::v[kShapeFull];
      const int64_t full_slots = static_cast<int64_t>(cus) * round_groups_per_cu(KIND, F.tr, F.tc, F.nt, false);
      const int by_default = (deposit == 0 && tiles_of(kShapeFull, 0) >= 2 * full_slots) ? kShapeFull
                                                                                          : (KIND == FLUVIAL ? 1 : 0);
      shape_early = ((std::getenv("SOIL_TILED_SHAPE") || std::getenv(KIND == FLUVIAL ? "SOIL_TILED_SHAPE_F" : "SOIL_TILED_SHAPE_D"))
                         ? env_kind("SOIL_TILED_SHAPE", KIND, 0)
                         : by_default) % kNumShapes;
    }
    shape_late = env_kind("SOIL_TILED_LATE", KIND, shape_early) % kNumShapes;
    if (deposit == 1) {  // the native-add variant exists for the 64-row shapes only
      if (shape_early == kShapeFull) shape_early = KIND == FLUVIAL ? 1 : 0;
      if (shape_late == kShapeFull) shape_late = KIND == FLUVIAL ? 1 : 0;
    }
    if (shape_early == kShapeColour) shape_early = 1;  // that shape goes with the colour planes only
    if (shape_late == kShapeColour) shape_late = 1;
    switch_round = env_kind("SOIL_TILED_SWITCH", KIND, 1 << 30);
    if (fluxA) {
      shape_early = shape_late = kShapeColour;
      deposit = 0;  // compare-and-swap deposits
    }
    // (debris on its 104-row tiles, round 3: 32 / 40 / 48 steps 11.05 / 10.78 / 10.77 ms per launch)
    // (fluvial, in the overlapped 8192^2 step at the end of round 3: 40 / 44 / 48 steps 33.30 / 32.90 / 33.06 ms
    // per step, four runs each on one box; by itself the launch does not tell them apart)
    // (round 4, with the cheaper epilogue and the sparse tiles' kernel: fluvial 36 / 40 / 44 / 48 / 52 / 56 / 60 /
    // 64 / 68 / 72 steps 32.2 32.0 31.5 31.6 31.3 31.4 31.5 31.1 31.3 31.4 ms per overlapped 8192^2 step, one box;
    // 4096^2 9.14 -> 8.99 at 64; 2048^2 4.00 -> 4.18 and 1024^2 1.65 -> 1.76: the 64-row tiles keep 44.
    // Debris 32 / 40 / 48 / 56: 31.8 31.5 31.5 31.8)
    // (round 5, fast arithmetic, 64-row tiles: 1024^2 — every tile's work-group resident at once, a round is its
    // hottest tile's chain — 44 | 36 | 32 | 28 | 24 steps: 1.51 | 1.49 | 1.49 ms per step early in a run, 1.39 | 1.28 |
    // 1.29 | 1.29 | 1.30 late (steps 6000-7500, three runs each); 2048^2 2.76 | 2.81 | 2.82: keeps 44)
    const bool all_resident = tiles_of(shape_early, 0) <= resident_groups_hint();
    steps_per_round = env_kind("SOIL_TILED_STEPS", KIND, KIND == FLUVIAL ? (shape_early == kShapeFull ? 64 : (all_resident ? 36 : 44))
                                                                         : (shape_early == kShapeFull ? 40 : (all_resident ? 48 : 32)));
    // (round 5, debris where every tile's work-group is resident at once: its chain of 14 rounds had become the
    // longer of the two at 1024^2 — 24 | 32 | 40 | 48 | 56 | 64 steps: 1.32 | 1.25 | 1.17 | 1.15 | 1.14 | 1.15 ms per
    // step late in a run (steps 3000-4500, three runs each, fast arithmetic), 1.585 | 1.575 early)
    // Worth its launch in front of every round only where a round is many generations of work-groups:
    // measured on one box, ms per step with | without: 1024^2 1.48 | 1.45, 2048^2 4.44 | 4.17, 4096^2
    // 9.11 | 8.99, 8192^2 31.97 | 32.23 — on from 16 tiles per resident work-group slot (SOIL_TILED_SPARSE=1
    // forces it on, 2 off; the parity tests force it on small grids)
    {
      const int sparse_env = env_kind("SOIL_TILED_SPARSE", KIND, 0);
      const bool by_size = tiles_of(shape_early, 0) >= 16 * static_cast<int64_t>(resident_groups_hint());
      sparse_ok = deposit == 0 && !fluxA && (sparse_env == 1 || (sparse_env == 0 && by_size));
    }
    host_lag_us = env_int("SOIL_TILED_HOST_LAG_US", 0);
    sparse_probe = env_kind("SOIL_TILED_SPARSE_PROBE", KIND, kSparseProbe);
    pair_free_below = static_cast<uint32_t>(std::min<int64_t>(0x7fffffff, N * env_kind("SOIL_PAIR_FREE", KIND, 20) / 100));
    // (measured, ms per overlapped step at 0 | 6 | 12 | 25 | 50 per cent: 8192^2 30.93 | 30.88 | 30.83 | 30.83 | 30.99; debris
    // alone at 25 | 60: 30.79 | 31.06; 4096^2 10.01 | 9.98 at 20, 2048^2 4.81 | 4.82: the late rounds' chain is bound by
    // its own latencies, free or in step)
    sparse_min = env_kind("SOIL_TILED_SPARSE_MIN", KIND, 64);
    sparse_pct = env_kind("SOIL_TILED_SPARSE_PCT", KIND, 25);
    sparse_pack = env_kind("SOIL_TILED_SPARSE_PACK", KIND, 1) != 2;  // (2: off — env_int reads 0 as "unset")
    steps_late = env_kind("SOIL_TILED_STEPS_LATE", KIND, 0);
    steps_late_from = env_kind("SOIL_TILED_LATE_FROM", KIND, 1 << 30);
    // A round is worth its fixed cost while it advances particles faster than the
    // finishing launch would (4 L2 atomics per step at 22.7 G/s = 5.7 G steps/s).
    // Particles that zig-zag along a tile edge get a handful of steps per round; on
    // small grids they are most of what is left after maxage/steps_per_round rounds.
    // The rate of the round just done (step counter / HIP event time) decides.
    finish_rate = env_int("SOIL_TILED_FINISH_MRATE", 4000) * 1e6;
    verbose = std::getenv("SOIL_TILED_VERBOSE") != nullptr;
#ifdef SOIL_ABLATE
    {
      // SOIL_ABLATE_AFTER=n: only from the n-th launch of this kind on (the terrain of the
      // warm-up steps is then the real one)
      static int launches = 0;
      const int after = std::getenv("SOIL_ABLATE_AFTER") ? std::atoi(std::getenv("SOIL_ABLATE_AFTER")) : 0;
      const int mask = (std::getenv("SOIL_ABLATE") && launches++ >= after) ? std::atoi(std::getenv("SOIL_ABLATE")) : 0;
      SOIL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(soil_ablate), &mask, sizeof(int)));
    }
#endif
    {  // LDS decides how many work-groups a CU holds
      for (int which = 0; which < 2; ++which) {
        const int sh = which ? shape_late : shape_early;
        const int per_cu = occupancy_of<KIND>(sh);
        if (verbose) std::fprintf(stderr, "[tiled kind %d] shape %d: %d work-groups per CU\n", KIND, sh, per_cu);
        resident_groups[which] = env_int("SOIL_TILED_SLOTS", cus * (per_cu > 0 ? per_cu : 1));
      }
    }

    // measured (1024^2 .. 8192^2): fluvial 1-6 % faster; debris, whose walks are short, 2 % slower
    // wave-aggregated adds for the losers of a compare-and-swap round (CasDeposit::finish):
    // from how many losers per wave on, and for how many distinct cells.  Swept at
    // 1024^2 / 2048^2 / 8192^2 (tools/sweep_agg.sh): 48 / 4 gives 3.7 / 6.5 / 46.2 ms per
    // step against 3.8 / 7.2 / 46.1 without; from 24 losers on the VALU work it adds
    // costs the large grids more than the LDS pipe gains (46.9), from 8 on 72 ms.
    agg_min = env_int("SOIL_TILED_AGG_MIN", 48);
    agg_groups = env_int("SOIL_TILED_AGG_GROUPS", 4);
    // swaps a loser repeats against the answer of the failed one before it falls back to the native
    // add (CasDeposit::finish).  Measured against none: 1024^2 1.89 -> 1.82 ms per step, 2048^2 (late
    // steps) 4.06 -> 3.78, 4096^2 11.3 -> 10.9 (with one try: 11.3), 8192^2 within the noise between
    // runs (36.3 -> 36.1; one try 35.3 in one run, 36.3 in another); four tries cost 1-2 % everywhere
    retries = 2;
    if (const char* e = std::getenv("SOIL_TILED_RETRIES")) retries = std::atoi(e) > 0 ? std::atoi(e) : 0;  // 0: none
    stagger = env_int("SOIL_TILED_STAGGER", KIND == FLUVIAL ? 1 : 2) == 1;
    fast = particle_arith_fast() && deposit == 0 && !fluxA;
    // (round 5, the fast step: one repeated swap — both pairs' issued together, CasDeposit::finish<true> —
    // before the native add: 8192^2 27.75 -> 27.3 ms per step on one box, two alternations; two swaps one
    // after the other per pair, the exact step's way: 27.8; in exact arithmetic one or two: the same)
    if (fast && !std::getenv("SOIL_TILED_RETRIES")) retries = 1;
    const int64_t max_tiles = std::max(std::max(tiles_of(shape_early, 0), tiles_of(shape_late, 0)),
                                       std::max(tiles_of(shape_early, 1), tiles_of(shape_late, 1)));
    auto align = [](size_t b) { return (b + 255) & ~static_cast<size_t>(255); };
    const size_t b_rec = align(sizeof(PRec) * N);
    const size_t b_p4 = align(sizeof(float4) * d.rows * d.W), b_idx = align(sizeof(uint32_t) * N);
    b_cnt = align(sizeof(uint32_t) * (max_tiles * kNB + 1));
    void* base = nullptr;
    // one workspace per kind: the two launches of a step may be in flight together
    const size_t b_blk = align(sizeof(uint4) * (max_tiles + N / 128 + 1));
    int rc = workspace_get(KIND == FLUVIAL ? 2 : 5, 2 * b_rec + 3 * b_idx + 4 * b_cnt + b_blk + b_p4 + 256, &base);
    if (rc != SOIL_OK) return rc;
    char* w = static_cast<char*>(base);
    cur = reinterpret_cast<PRec*>(w);    w += b_rec;   // records of this round (any order)
    next = reinterpret_cast<PRec*>(w);   w += b_rec;   // survivors, grouped by the tile they left
    dest = reinterpret_cast<uint32_t*>(w);   w += b_idx;  // queue section each slot is bound for
    rank = reinterpret_cast<uint32_t*>(w);   w += b_idx;  // ... and its place in that section
    order = reinterpret_cast<uint32_t*>(w);  w += b_idx;  // slots sorted by queue section
    p4 = reinterpret_cast<float4*>(w);  w += b_p4;
    count = reinterpret_cast<uint32_t*>(w);       w += b_cnt;
    count_next = reinterpret_cast<uint32_t*>(w);  w += b_cnt;
    start = reinterpret_cast<uint32_t*>(w);       w += b_cnt;
    tile_order = reinterpret_cast<uint32_t*>(w);  w += b_cnt;
    block_list = reinterpret_cast<uint4*>(w);     w += b_blk;
    steps_run = reinterpret_cast<unsigned long long*>(w);
    ctl = reinterpret_cast<TiledCtl*>(w + 64);
    static_assert(64 + sizeof(TiledCtl) <= 224, "the pack pass's word stands behind the block begin() clears");
    retire_bad = reinterpret_cast<uint32_t*>(w + 224);
    // spent debris walkers end their walks (debris_spent).  Not in migrate mode: the walker's later cells lie on
    // other ranks, whose pack passes this rank's word knows nothing about.  Not with colour planes (never measured).
    retire_mode = (KIND == DEBRIS && !box.count && !fluxA && debris_params_allow_retire(p))
                      ? static_cast<uint32_t>(debris_retire_mode()) : 0u;
    rc = step_counter(&steps_global);
    if (rc != SOIL_OK) return rc;
    // pinned word + events, one set per (thread, device, kind): a host thread that moves on to
    // another device (soil_set_device) must not poll a word or record events of the first one
    struct HostSide {
      TiledHostWord *host = nullptr, *host_dev = nullptr;
      uint32_t seq = 0;  // numbers the launches that fill `host`, across runs
      double ticks_per_second = 1.0e8;
    };
    static thread_local std::map<int, HostSide> t_side;
    int dev = 0;
    SOIL_HIP(hipGetDevice(&dev));
    HostSide& hs = t_side[dev];
    if (!hs.host) {
      SOIL_HIP(hipHostMalloc(reinterpret_cast<void**>(&hs.host), sizeof(TiledHostWord), hipHostMallocMapped | hipHostMallocCoherent));
      SOIL_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&hs.host_dev), hs.host, 0));
      hs.host->seq = 0;
      int khz = 0;  // rate of the realtime counter the scans time the rounds with
      if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0)
        hs.ticks_per_second = 1.0e3 * khz;
      else
        (void)hipGetLastError();
    }
    host = hs.host;
    host_dev = hs.host_dev;
    seq_ctr = &hs.seq;
    ticks_per_second = hs.ticks_per_second;
    tail_scan = env_int("SOIL_TILED_TAILSCAN", 1) == 1;
    depth = verbose ? 0 : env_int("SOIL_TILED_AHEAD", 2);
    if (std::getenv("SOIL_TILED_AHEAD") && std::atoi(std::getenv("SOIL_TILED_AHEAD")) == 0) depth = 0;
    depth = std::min(depth, static_cast<int>(TiledHostWord::kLiveRing) / 2);  // (a ring slot is read before its scan + 16 writes it)
    ready = true;
    return SOIL_OK;
  }

  // What the scan of the queues round r starts from is asked to do (QueueScan): offsets, dispatch
  // order and work-group list of that round, the device's decision whether the round takes place at
  // all (scan_decide) and the word for the host.
  QueueScan make_scan(uint64_t r) {
    QueueScan q;
    q.start = start;
    q.tile_order = tile_order;
    q.block_list = block_list;
    q.count4 = reinterpret_cast<const uint4*>(count_of(r));
    q.tiles = tiles_of(shape_of(r), r);
    q.lanes = Shapes<KIND>::v[shape_of(r)].nt;
    q.slots = resident_groups[r >= static_cast<uint64_t>(switch_round) ? 1 : 0];
    q.steps_run = steps_run;
    q.host = host_dev;
    q.seq = ++*seq_ctr;
    q.ctl = ctl;
    // every live particle advances >= 1 step per round: maxage + 2 rounds always suffice
    const uint64_t max_round = p.maxage + 2;
    q.rule.round = static_cast<uint32_t>(r);
    q.rule.tail = static_cast<uint32_t>(tail);
    q.rule.max_round = max_round > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(max_round);
    q.rule.ticks_per_step_max = static_cast<float>(ticks_per_second / finish_rate);
    q.include_empty = (r == 0 && overwrite && !fluxA && deposit == 0) ? 1u : 0u;
    q.sparse_ok = (r >= 1 && sparse_ok && shape_of(r) != kShapeColour) ? 1u : 0u;
    q.sparse_min = static_cast<uint32_t>(sparse_min);
    q.sparse_pct = static_cast<uint32_t>(sparse_pct);
    q.sparse_pack = sparse_pack ? 1u : 0u;
    return q;
  }
  // the scan of round 0 (the queues the spawn filled) is a kernel of its own; every later one runs at
  // the tail of the round kernel before it
  int queue_scan() {
    const QueueScan q = make_scan(0);
    seq_first = q.seq;
    k_queue_scan<<<1, 1024, 0, st>>>(q);
    SOIL_LAUNCH_CHECK();
    scans = 1;
    return SOIL_OK;
  }

  // Wait for the word of scan `seen`.  Polling the pinned word sees it a few microseconds after the
  // kernel stored it; hipStreamSynchronize adds the runtime's completion handling on top.  Falls back
  // to the runtime's wait (and its error reporting) when the word does not show up soon.  Words of
  // later scans may have overwritten it by then: they carry the same verdict or a later one, and
  // sequence numbers only grow.
  int wait_word() {
    const uint32_t want = seq_first + static_cast<uint32_t>(seen);
    auto arrived = [&]() { return static_cast<int32_t>(__atomic_load_n(&host->seq, __ATOMIC_ACQUIRE) - want) >= 0; };
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
      if (arrived()) return SOIL_OK;
      if ((spins & 1023u) == 1023u &&
          std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
        break;
    }
    SOIL_HIP(hipStreamSynchronize(st));
    if (!arrived()) return fail(SOIL_ERR_HIP, "tiled transport: the queue word of the round never arrived");
    return SOIL_OK;
  }

  int begin() {
    if (!ready)
      if (int rc = setup(); rc != SOIL_OK) return rc;
    const int64_t lo = stencil_lo(d), hi = stencil_hi(d);
    if (hi >= lo && !skip_pack) {
      if (KIND == DEBRIS) SOIL_HIP(hipMemsetAsync(retire_bad, 0, sizeof(uint32_t), st));
      k_tiled_pack<KIND><<<grid_rows(hi - lo + 1, d.W, 256), 256, 0, st>>>(
          p4, reinterpret_cast<const float2*>(layers), reinterpret_cast<const float2*>(velocity),
          waterHeight, d, s, p, lo, hi + 1, KIND == DEBRIS ? retire_bad : nullptr);
    }
    SOIL_HIP(hipMemsetAsync(count, 0, b_cnt, st));
    // the step counter of the run and, behind it, the device's control block: mode 0, the N spawn
    // slots as what "the round before" left
    SOIL_HIP(hipMemsetAsync(steps_run, 0, 64 + sizeof(TiledCtl), st));
    if (inbox) {
      k_tiled_inject<KIND><<<blocks_for(std::max<int64_t>(n_in, 1), 256), 256, 0, st>>>(
          cur, dest, rank, count, inbox, n_in, d, p, tiles_w_of(shape_of(0), 0), ts_of(shape_of(0), 0), steps_per_round, ctl,
          retire_bad, retire_mode);
    } else {
      k_tiled_spawn<KIND><<<blocks_for(N, 256), 256, 0, st>>>(
          cur, dest, rank, count, rng, N, p4, waterSource, albedoSource, d, s, p, tiles_w_of(shape_of(0), 0),
          ts_of(shape_of(0), 0), steps_per_round, ctl, retire_bad, retire_mode, fast, steps_run);
    }
    SOIL_LAUNCH_CHECK();
    live_known = inbox ? static_cast<int64_t>(n_in) : N;  // slots of the record array to look at (spawn output, then survivor slots)
    round = scans = seen = 0;
    return queue_scan();
  }

  int finish_steps() {  // fold this launch's steps into the device-wide counter
    k_fold_steps<<<1, 1, 0, st>>>(steps_global, steps_run);
    SOIL_LAUNCH_CHECK();
    done = true;
    return SOIL_OK;
  }

  // Round `round`: the slot sort (it needs the scan's offsets and the slots the round before filled,
  // both on the device) and the round kernel, whose last work-group scans the queues of the round
  // after it.  Queued ahead of the host's look at the scan's word: what the round finds to do is
  // the device's decision.
  // `slots_bound`: an upper bound of the record slots the round before filled (what the slot sort has
  // to look at): the live count of the last word seen BEFORE the word of this round's own scan
  int queue_round(int store_all, int64_t slots_bound) {
    const uint64_t r = round;
    const int sh = shape_of(r), sh_next = shape_of(r + 1);
    const int64_t tiles = tiles_of(sh, r);
    const int tiles_w = tiles_w_of(sh, r);
    const TileShape ts_cur = ts_of(sh, r);
    k_tiled_scatter<<<blocks_for(std::max<int64_t>(slots_bound, 1), 256), 256, 0, st>>>(
        order, start, dest, rank, ctl, count_of(r + 1), static_cast<int64_t>(b_cnt / sizeof(uint32_t)));
    SOIL_LAUNCH_CHECK();
    // as many work-groups as a round can have (a tile each, plus the chunks long queues are cut into);
    // those beyond the scan's count return at once
    const int slots = resident_groups[r >= static_cast<uint64_t>(switch_round) ? 1 : 0];
    // (a round that stores its tiles has a work-group for every tile, the empty ones included,
    // however few walkers there are: advisor finding of round 3)
    const unsigned grid = (r == 0 && overwrite)
                              ? static_cast<unsigned>(tiles + slots)
                              : static_cast<unsigned>(std::min<int64_t>(tiles + slots, std::max<int64_t>(live_known, 1)));
    PRec* in = recs_of(r);
    PRec* out = recs_of(r + 1);
    uint32_t* my_dense = nullptr;
    static const uint32_t gate_early = static_cast<uint32_t>(std::min(100, std::max(0, env_int("SOIL_PAIR_EARLY", 20))));
    QueueScan no_scan{};  // (the sparse kernel leaves the round's ticket and scan to the dense one)
    no_scan.host = nullptr;
    if (r >= 1 && sparse_ok && sh != kShapeColour) {
      // the sparse tiles of the round, if its scan made any (ctl->sparse_of): in front of the gate —
      // eight one-wave work-groups fit a CU beside whatever the other launch has there
      const unsigned grid_sparse = static_cast<unsigned>(std::min<int64_t>(tiles, std::max<int64_t>(live_known, 1)));
      launch_sparse<KIND>(fast, sh, grid_sparse, st, out, dest, rank, count_of(r + 1), static_cast<const PRec*>(in),
                          static_cast<const uint32_t*>(order), static_cast<const uint4*>(block_list), flux0, flux1,
                          reinterpret_cast<float2*>(fluxV), fluxA, static_cast<const float4*>(p4), remote0, steps_run,
                          d, s, p, tiles_w, ts_cur.off_r, ts_cur.off_c, steps_of(r), ts_of(sh_next, r + 1),
                          tiles_w_of(sh_next, r + 1), agg_min, agg_groups, sparse_probe, 0, ctl, static_cast<uint32_t>(r),
                          no_scan, static_cast<uint32_t*>(nullptr), 0u, box);
      SOIL_LAUNCH_CHECK();
    }
    if (gate && tail_scan) {  // (the `started` ticket is reset by the tail scan's work-group)
      k_pair_gate<<<1, 1, 0, st>>>(gate, KIND, ctl, static_cast<unsigned long long>(0.05 * ticks_per_second), pair_free_below);
      SOIL_LAUNCH_CHECK();
      my_dense = &gate->dense[KIND];
    }
    QueueScan next_scan = make_scan(r + 1);
    const QueueScan standalone = next_scan;
    if (!tail_scan) next_scan.host = nullptr;  // the round kernel leaves the scan to a launch of its own

    if (deposit == 1)
      launch_round<KIND, 0>(false, sh, grid, st, out, dest, rank, count_of(r + 1),
                            static_cast<const PRec*>(in), static_cast<const uint32_t*>(order),
                            static_cast<const uint4*>(block_list), flux0, flux1,
                            reinterpret_cast<float2*>(fluxV), fluxA, static_cast<const float4*>(p4),
                            remote0, steps_run, d, s, p, tiles_w, ts_cur.off_r, ts_cur.off_c,
                            steps_of(r), ts_of(sh_next, r + 1),
                            tiles_w_of(sh_next, r + 1), agg_min, agg_groups, retries, store_all,
                            ctl, static_cast<uint32_t>(r), next_scan, my_dense, gate_early, box);
    else
      launch_round<KIND, 1>(fast, sh, grid, st, out, dest, rank, count_of(r + 1),
                            static_cast<const PRec*>(in), static_cast<const uint32_t*>(order),
                            static_cast<const uint4*>(block_list), flux0, flux1,
                            reinterpret_cast<float2*>(fluxV), fluxA, static_cast<const float4*>(p4),
                            remote0, steps_run, d, s, p, tiles_w, ts_cur.off_r, ts_cur.off_c,
                            steps_of(r), ts_of(sh_next, r + 1),
                            tiles_w_of(sh_next, r + 1), agg_min, agg_groups, retries, store_all,
                            ctl, static_cast<uint32_t>(r), next_scan, my_dense, gate_early, box);
    SOIL_LAUNCH_CHECK();
    if (!tail_scan) {
      k_queue_scan<<<1, 1024, 0, st>>>(standalone);
      SOIL_LAUNCH_CHECK();
    }
    ++round;
    ++scans;
    return SOIL_OK;
  }

  // One look at the device's progress: wait for the next scan's word; if the rounds go on, make sure
  // the round it belongs to is queued and queue `depth` rounds beyond it; if the device has stopped
  // them, queue the finishing launch for the records the last executed round left.
  int advance() {
    if (done) return SOIL_OK;
    // (tests: a host that falls behind the device by so many microseconds before every look at a word —
    // the rounds queued ahead run on, later scans write their words over the one awaited)
    if (host_lag_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(host_lag_us));
    if (int rc = wait_word(); rc != SOIL_OK) return rc;
    const uint64_t r = seen++;  // the word of scan r (or of a later one carrying the same verdict)
    const uint32_t mode = __atomic_load_n(&host->mode, __ATOMIC_ACQUIRE);
    const int64_t slots_before = live_known;  // >= the slots round r's sort and the finishing launch look at
    // (the count of scan r itself, not of whatever scan wrote the word last: see TiledHostWord::live_ring)
    const uint32_t live_r = host->live_ring[(seq_first + static_cast<uint32_t>(r)) % TiledHostWord::kLiveRing];
    if (mode == 0) live_known = std::min<int64_t>(live_known, static_cast<int64_t>(live_r));
#ifdef SOIL_PROF
    if (verbose) {  // where the waves of the round just done spent their cycles (depth 0: round r - 1 is over)
      static const char* seg[10] = {"stops", "refill", "head", "gather+dep begin", "advance", "dep finish",
                                    "survivors out", "barrier wait", "prologue", "flush"};
      unsigned long long v[32];
      SOIL_HIP(hipStreamSynchronize(st));
      if (soil_prof_read(v, 1) == 0) {
        const unsigned long long* w = v + 16 * KIND;
        unsigned long long tot = 0;
        for (int i = 0; i < 10; ++i) tot += w[i];
        if (tot > 0) {
          std::fprintf(stderr, "[prof kind %d] before scan %llu: %llu waves, %llu wave-iterations, %.0f ticks per wave;",
                       KIND, static_cast<unsigned long long>(r), w[11], w[10], static_cast<double>(tot) / std::max<unsigned long long>(w[11], 1));
          for (int i = 0; i < 10; ++i) std::fprintf(stderr, " %s %.1f%%", seg[i], 100.0 * w[i] / tot);
          std::fprintf(stderr, "\n");
        }
      }
    }
#endif
    if (verbose) {  // queue-length statistics of the round (diagnostics only; depth 0: the word is scan r's)
      const int sh = shape_of(r);
      const int64_t tiles = tiles_of(sh, r);
      std::vector<uint32_t> pre(static_cast<size_t>(tiles * kNB + 1)), h(static_cast<size_t>(tiles));
      SOIL_HIP(hipMemcpy(pre.data(), start, sizeof(uint32_t) * pre.size(), hipMemcpyDeviceToHost));
      for (int64_t t = 0; t < tiles; ++t) h[t] = pre[(t + 1) * kNB] - pre[t * kNB];
      std::sort(h.begin(), h.end());
      const int lanes = Shapes<KIND>::v[sh].nt;
      uint64_t batches = 0, empty = 0, sparse = 0;
      for (uint32_t c : h) {
        batches += (c + lanes - 1) / lanes;
        empty += c == 0;
        sparse += c > 0 && c < static_cast<uint32_t>(lanes) / 4;
      }
      std::fprintf(stderr,
                   "[tiled kind %d] round %llu: %u live (mode %u); tiles %lld empty %llu sparse(<1/4) %llu "
                   "median %u p90 %u p99 %u max %u batches %llu; %llu steps so far\n",
                   KIND, static_cast<unsigned long long>(r), host->live, mode, static_cast<long long>(tiles),
                   static_cast<unsigned long long>(empty), static_cast<unsigned long long>(sparse),
                   h[h.size() / 2], h[h.size() * 9 / 10], h[h.size() * 99 / 100], h.back(),
                   static_cast<unsigned long long>(batches), static_cast<unsigned long long>(host->steps));
    }
    int store_all = 0;
    if (overwrite && r == 0) {  // round 0 is never queued ahead of its word in this mode (round == 0 here)
      if (mode == 0 && host->whole == 1u && !fluxA && deposit == 0) {
        store_all = 1;
      } else {  // cannot: clear the planes the launch adds to
        const size_t cells_b = sizeof(float) * static_cast<size_t>(d.rows) * static_cast<size_t>(d.W);
        SOIL_HIP(hipMemsetAsync(flux0, 0, cells_b, st));
        if (flux1) SOIL_HIP(hipMemsetAsync(flux1, 0, cells_b, st));
        SOIL_HIP(hipMemsetAsync(fluxV, 0, 2 * cells_b, st));
      }
    }
    if (mode != 0) {
      if (mode == 1) {  // the records round `stop_round` would have read, and the slots in use among them
        const uint64_t stop = host->stop_round;
        if (fast)
          k_tiled_finish<KIND, true><<<blocks_for(std::max<int64_t>(slots_before, 1), 256), 256, 0, st>>>(
              box, recs_of(stop), dest, ctl, flux0, flux1, fluxV, fluxA, p4, remote0, steps_run, d, s, p);
        else
          k_tiled_finish<KIND><<<blocks_for(std::max<int64_t>(slots_before, 1), 256), 256, 0, st>>>(
              box, recs_of(stop), dest, ctl, flux0, flux1, fluxV, fluxA, p4, remote0, steps_run, d, s, p);
        SOIL_LAUNCH_CHECK();
      }
      return finish_steps();
    }
    if (round == r)
      if (int rc = queue_round(store_all, slots_before); rc != SOIL_OK) return rc;
    while (round < seen + static_cast<uint64_t>(depth) && round < p.maxage + 3)
      if (int rc = queue_round(0, live_known); rc != SOIL_OK) return rc;
    return SOIL_OK;
  }
};

template <int KIND>
static TiledRun<KIND> make_run(float* flux0, float* flux1, float* fluxV, float* fluxA,
                               const float* albedoSource, Streams rng, int64_t N,
                               const float* layers, const float* waterSource,
                               const float* waterHeight, const float* velocity, float* remote0,
                               const Dom& d, Scale3 s, const Param& p, hipStream_t st) {
  TiledRun<KIND> r;
  r.flux0 = flux0, r.flux1 = flux1, r.fluxV = fluxV, r.rng = rng, r.N = N;
  r.fluxA = fluxA, r.albedoSource = fluxA ? albedoSource : nullptr;
  r.layers = layers, r.waterSource = waterSource, r.waterHeight = waterHeight, r.velocity = velocity;
  r.remote0 = remote0, r.d = d, r.s = s, r.p = p, r.st = st;
  return r;
}

template <int KIND>
static int run_tiled(float* flux0, float* flux1, float* fluxV, float* fluxA,
                     const float* albedoSource, Streams rng, int64_t N, const float* layers, const float* waterSource, const float* waterHeight,
                     const float* velocity, float* remote0, const Dom& d, Scale3 s, const Param& p,
                     hipStream_t st) {
  TiledRun<KIND> r = make_run<KIND>(flux0, flux1, fluxV, fluxA, albedoSource, rng, N, layers,
                                    waterSource, waterHeight, velocity, remote0, d, s, p, st);
  if (int rc = r.begin(); rc != SOIL_OK) return rc;
  while (!r.done)
    if (int rc = r.advance(); rc != SOIL_OK) return rc;
  return SOIL_OK;
}

// Both launches of a step, overlapped: two internal streams forked from `st` and joined
// back into it; the host alternates between the two runs' decisions.
int launch_pair_tiled(const soil_erosion_planes& P, Streams rng_fluvial, Streams rng_debris, int64_t N,
                      float* remote0, const Dom& d, Scale3 s, const Param& p, hipStream_t st, bool overwrite,
                      MigrateBox box_fluvial, MigrateBox box_debris, const void* inbox_fluvial, uint32_t n_fluvial,
                      const void* inbox_debris, uint32_t n_debris) {
  // forked streams and their events, one set per (thread, device)
  struct Fork {
    hipStream_t sA = nullptr, sB = nullptr;
    hipEvent_t fork = nullptr, joinA = nullptr, joinB = nullptr;
  };
  static thread_local std::map<int, Fork> t_fork;
  int dev = 0;
  SOIL_HIP(hipGetDevice(&dev));
  Fork& f = t_fork[dev];
  if (!f.sA) {
    SOIL_HIP(hipStreamCreateWithFlags(&f.sA, hipStreamNonBlocking));
    SOIL_HIP(hipStreamCreateWithFlags(&f.sB, hipStreamNonBlocking));
    SOIL_HIP(hipEventCreateWithFlags(&f.fork, hipEventDisableTiming));
    SOIL_HIP(hipEventCreateWithFlags(&f.joinA, hipEventDisableTiming));
    SOIL_HIP(hipEventCreateWithFlags(&f.joinB, hipEventDisableTiming));
  }
  const hipStream_t sA = f.sA, sB = f.sB;
  TiledRun<FLUVIAL> A = make_run<FLUVIAL>(P.waterFlux, P.massFlux, P.velocityFlux, nullptr, nullptr,
                                          rng_fluvial, N,
                                          P.layers, P.rainfall, P.waterHeight, P.velocity, remote0,
                                          d, s, p, sA);
  TiledRun<DEBRIS> B = make_run<DEBRIS>(P.debrisFlux, nullptr, P.debrisVelocityFlux, nullptr, nullptr,
                                        rng_debris, N,
                                        P.layers, nullptr, nullptr, P.debrisVelocity, remote0, d, s,
                                        p, sB);
  // When the second launch starts: from the fluvial launch's round `delay` on.  Measured
  // (tools/ab_pair_sizes.sh; ms per step, sequential | delay 1, 2, 4, 8): 1024^2 3.00 | 2.13 2.03 2.29
  // 2.54; 2048^2 5.70 | 4.99 4.92 4.79 4.94; 4096^2 12.14 | 11.72 11.75 11.98 11.78; 8192^2 37.3 | 37.0
  // (2) 37.2 (6) 37.3 (8): small grids are bound by the latency of each launch's chain of rounds, and two
  // chains interleave; at 8192^2 either launch fills the chip by itself.
  static const int delay_env = env_int("SOIL_PAIR_DELAY", 0);
  // How the two launches share the chip (SOIL_PAIR_MODE; default: by size).  1: overlapped freely;
  // 2: overlapped, taking turns round by round behind the device-side gate (PairGate); 3: the debris
  // launch after the fluvial one.  All three share the one pack pass.  Mixing the two round kernels
  // freely loses where either fills the chip by itself (a CU's LDS holds two fluvial tiles or three
  // debris ones; one of each leaves 28 KiB unused and no third); one after the other, every round
  // ends with the chip half empty.  Measured on one box, ms per step, 1 | 2 | 3: 1024^2 1.58 | 1.63 | 2.10,
  // 2048^2 3.98 | 3.90 | 4.44, 4096^2 10.89 | 10.20 | 10.59, 8192^2 38.27 | 35.94 | 36.86.  (Round 2's
  // overlapped 8192^2 step took turns by accident: its scan kernels asked for 139 KiB of LDS and so
  // waited for a CU the other launch's round had drained.  Stream priorities do not change the mix
  // of mode 1 (38.4 with the fluvial stream at the highest priority); opening the gate 128 / 256 / 512
  // work-groups before the round's last one has started: 35.66 / 36.25 / 38.21 against 35.66.)
  const int pair_mode = env_int("SOIL_PAIR_MODE", 0);
  const bool turns = pair_mode == 2 || (pair_mode == 0 && N >= 500000);
  // (immigrants' launches side by side, slab runner's migrate mode: taking turns as the streams' count says — 2.33 ms
  // for 140 k fluvial + 62 k debris walkers of a 4-way split of 16384^2; mixed freely 2.5-2.7; one after the other 2.75)
  const bool serial_pair = pair_mode == 3;
  if (turns) {
    void* g = nullptr;
    if (int rc = workspace_get(9, 256, &g); rc != SOIL_OK) return rc;
    SOIL_HIP(hipMemsetAsync(g, 0, sizeof(PairGate), st));  // ahead of the fork
    A.gate = B.gate = static_cast<PairGate*>(g);
  }
  // (round 3, with rounds queued ahead of the host: counted in scans the host has seen; 1024^2 1.59 / 1.62 /
  // 1.67 ms per step at 1 / 2 / 3, 2048^2 3.98 / 3.86 / 3.75, 4096^2 10.28 / 10.36 / 10.48)
  // (taking turns the debris launch begins at once — its spawn beside the fluvial one's, its rounds behind
  // the gate: 8192^2 34.72 / 34.78 / 34.88 / 34.95 / 35.18 ms per step at 0 / 1 / 2 / 3 / 5, 2048^2 3.75 at 0, 3.79 at 1)
  const uint64_t delay = delay_env > 0 ? static_cast<uint64_t>(delay_env) : (turns ? 0 : (N <= 300000 ? 1 : 2));
  // Whatever happens in between, `st` is joined with both streams before this returns: rounds may
  // still be in flight on the workspace the next call reuses.
  A.overwrite = B.overwrite = overwrite;
  A.box = box_fluvial, B.box = box_debris;
  const bool immigrants = inbox_fluvial != nullptr || inbox_debris != nullptr;
  if (immigrants) {  // both kinds' handed-over walkers, walked on side by side: the step's pack pass stands
    A.inbox = static_cast<const PRec*>(inbox_fluvial), A.n_in = n_fluvial;
    B.inbox = static_cast<const PRec*>(inbox_debris), B.n_in = n_debris;
    A.skip_pack = B.skip_pack = true;
  }
  auto run = [&]() -> int {
    if (int rc = A.setup(); rc != SOIL_OK) return rc;
    if (int rc = B.setup(); rc != SOIL_OK) return rc;
    static const bool fused_pack = env_int("SOIL_PACK_PAIR", 1) == 1;   // 2: off (A/B)
    if (fused_pack && !immigrants) {
      const int64_t lo = stencil_lo(d), hi = stencil_hi(d);
      static const bool pack4 = env_int("SOIL_PACK_WINDOW", 1) == 1;  // 2: the one-cell-per-thread pass (A/B)
      const bool wide = pack4 && d.W % 4 == 0 && d.W >= 4 &&
                        ((reinterpret_cast<uintptr_t>(P.layers) | reinterpret_cast<uintptr_t>(P.velocity) |
                          reinterpret_cast<uintptr_t>(P.waterHeight) | reinterpret_cast<uintptr_t>(P.debrisVelocity) |
                          reinterpret_cast<uintptr_t>(A.p4) | reinterpret_cast<uintptr_t>(B.p4)) & 15u) == 0;
      if (hi >= lo) SOIL_HIP(hipMemsetAsync(B.retire_bad, 0, sizeof(uint32_t), st));
      if (hi >= lo && wide) {
        // rows per work-group: the window shape's 32 where that still makes four work-groups per CU, fewer
        // on small grids (1024^2 is ONE column group: 32 work-groups of 256 threads for the whole chip,
        // 107 us per step of BASELINE config 2's 1.4 ms; round 5)
        const int64_t rows = hi - lo + 1, groups = (d.W / 4 + kWinBlock - 1) / kWinBlock;
        static const int band_max = env_int("SOIL_PACK_BAND", kWinBand);
        const int band_rows = static_cast<int>(std::max<int64_t>(2, std::min<int64_t>(band_max, rows * groups / 1024)));
        const int64_t bands = (rows + band_rows - 1) / band_rows;
        k_tiled_pack_pair4<<<dim3(static_cast<unsigned>(groups), static_cast<unsigned>(std::min<int64_t>(bands, 65535))), kWinBlock, 0, st>>>(
            A.p4, B.p4, reinterpret_cast<const float2*>(P.layers), reinterpret_cast<const float2*>(P.velocity),
            P.waterHeight, reinterpret_cast<const float2*>(P.debrisVelocity), d, s, p, lo, hi + 1, band_rows, B.retire_bad);
      }
      else if (hi >= lo)
        k_tiled_pack_pair<<<grid_rows(hi - lo + 1, d.W, 256), 256, 0, st>>>(
            A.p4, B.p4, reinterpret_cast<const float2*>(P.layers), reinterpret_cast<const float2*>(P.velocity),
            P.waterHeight, reinterpret_cast<const float2*>(P.debrisVelocity), d, s, p, lo, hi + 1, B.retire_bad);
      SOIL_LAUNCH_CHECK();
      A.skip_pack = B.skip_pack = true;
    }
    // The fork, behind the one pack pass on the caller's stream.  (With the pass on the fluvial
    // stream and the debris stream waiting for a second event recorded there, the debris launch's
    // first kernel started when the fluvial launch's round 0 ended, 3 ms late, on every step of the
    // rocprof trace.  Why is not known: by itself such a wait resolves within 12 us of the kernel
    // it stands behind, tools/microbench/event_wait.hip.)
    SOIL_HIP(hipEventRecord(f.fork, st));
    SOIL_HIP(hipStreamWaitEvent(sA, f.fork, 0));
    SOIL_HIP(hipStreamWaitEvent(sB, f.fork, 0));
    if (int rc = A.begin(); rc != SOIL_OK) return rc;
    bool b_started = false;
    while (!A.done || !B.done) {
      if (!b_started && (A.done || (!serial_pair && A.seen >= delay))) {  // `delay` scans of the fluvial launch seen
        if (int rc = B.begin(); rc != SOIL_OK) return rc;
        b_started = true;
      }
      if (int rc = A.advance(); rc != SOIL_OK) return rc;
      if (b_started)
        if (int rc = B.advance(); rc != SOIL_OK) return rc;
    }
    return SOIL_OK;
  };
  const int rc = run();
  if (rc != SOIL_OK) {  // keep the first error's message; the joins below are best effort
    if (hipEventRecord(f.joinA, sA) == hipSuccess) (void)hipStreamWaitEvent(st, f.joinA, 0);
    if (hipEventRecord(f.joinB, sB) == hipSuccess) (void)hipStreamWaitEvent(st, f.joinB, 0);
    (void)hipGetLastError();
    return rc;
  }
  SOIL_HIP(hipEventRecord(f.joinA, sA));
  SOIL_HIP(hipEventRecord(f.joinB, sB));
  SOIL_HIP(hipStreamWaitEvent(st, f.joinA, 0));
  SOIL_HIP(hipStreamWaitEvent(st, f.joinB, 0));
  return SOIL_OK;
}

int launch_fluvial_tiled(float* waterFlux, float* massFlux, float* velocityFlux, float* albedoFlux,
                         Streams rng, int64_t N, const float* layers, const float* waterSource,
                         const float* waterHeight, const float* velocity,
                         const float* albedoSource, float* remote0, const Dom& d, Scale3 s,
                         const Param& p, hipStream_t st) {
  return run_tiled<FLUVIAL>(waterFlux, massFlux, velocityFlux, albedoFlux, albedoSource, rng, N,
                            layers, waterSource, waterHeight, velocity, remote0, d, s, p, st);
}

int launch_debris_tiled(float* massFlux, float* velocityFlux, float* albedoFlux, Streams rng,
                        int64_t N, const float* layers, const float* velocity,
                        const float* albedoSource, float* remote0, const Dom& d, Scale3 s,
                        const Param& p, hipStream_t st) {
  return run_tiled<DEBRIS>(massFlux, nullptr, velocityFlux, albedoFlux, albedoSource, rng, N, layers,
                           nullptr, nullptr, velocity, remote0, d, s, p, st);
}

// One launch of one kind for the slab runner's migrate mode: spawns (inbox null) or handed-over records,
// leavers into `box`.  Later passes of a step find the kind's cell records where the first one packed them.
int launch_pass_tiled(int kind, const soil_erosion_planes& P, Streams rng, int64_t N, float* remote0, const Dom& d,
                      Scale3 s, const Param& p, hipStream_t st, const void* inbox, uint32_t n_in, MigrateBox box) {
  auto run = [&](auto r) -> int {
    r.box = box;
    r.inbox = static_cast<const PRec*>(inbox);
    r.n_in = n_in;
    r.skip_pack = inbox != nullptr;
    if (int rc = r.begin(); rc != SOIL_OK) return rc;
    while (!r.done)
      if (int rc = r.advance(); rc != SOIL_OK) return rc;
    return SOIL_OK;
  };
  if (kind == FLUVIAL)
    return run(make_run<FLUVIAL>(P.waterFlux, P.massFlux, P.velocityFlux, nullptr, nullptr, rng, N, P.layers, P.rainfall,
                                 P.waterHeight, P.velocity, remote0, d, s, p, st));
  return run(make_run<DEBRIS>(P.debrisFlux, nullptr, P.debrisVelocityFlux, nullptr, nullptr, rng, N, P.layers, nullptr,
                              nullptr, P.debrisVelocity, remote0, d, s, p, st));
}

}  // namespace soil

// soil_hip.h: what the watched mode of the debris retirement (soil_set_debris_retire(2)) has counted
extern "C" int soil_debris_retire_violations(uint64_t* total, int reset, void* stream) {
  using namespace soil;
  SOIL_REQUIRE(total, "soil_debris_retire_violations: null pointer");
  SOIL_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  unsigned long long v = 0;
  SOIL_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(soil_retire_violations_dev), sizeof(v)));
  if (reset) {
    const unsigned long long z = 0;
    SOIL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(soil_retire_violations_dev), &z, sizeof(z)));
  }
  *total = v;
  return SOIL_OK;
}
