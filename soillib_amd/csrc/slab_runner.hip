// slab_runner.hip — the sharded erosion step (include/soil_slab.h): row slabs, deep halos trimmed to
// the measured reach of the walks, RCCL on the runner's own streams.
//
// The reference has no multi-GPU path at all (one default-stream launch per kernel,
// erosion.cu:209, :413); BASELINE.json configs[4] defines the job.  Three parts:
//   1. SlabRunner — the host logic of one rank.  It only talks to the two function tables of
//      soil_slab.h, so the same code runs on HIP + RCCL (the product), on HIP + an in-process wire
//      (several slabs on one GPU: tests) and on the CPU oracle + gloo (tests without a GPU).
//   2. HipOps     — soil_slab_ops over this library's kernels: a main and a communication stream.
//   3. RcclComm   — soil_comm over librccl, resolved at run time (dlopen).
//
// Why deep halos: one __stepsize step moves a walker by at most sqrt(2) cells
// (erosion_map.cu:61-76), so with G = ceil(sqrt(2) maxage) + 2 ghost rows per interior side every
// walker born in the owned rows ends inside the slab; what travels between ranks are rows of planes,
// nearest neighbours only, and how many rows is decided by measurement (soil_ghost_extent): see
// step() below.  DESIGN.md 5 has the cost model next to the alternative (walkers handed over at the
// slab edge).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <map>

#include "../../include/soil_slab.h"
#include "common.hpp"
#include "particles_common.hpp"

namespace soil {

// ------------------------------------------------------------------------------------------------
// 1. the runner
// ------------------------------------------------------------------------------------------------

namespace {

struct Layout { int64_t x0, rows, r0, r1; };

Layout slab_layout(int rank, int world, int64_t S, int64_t G) {
  const int64_t H = world * S, o0 = rank * S, o1 = (rank + 1) * S;
  const int64_t x0 = std::max<int64_t>(0, o0 - G), x1 = std::min<int64_t>(H, o1 + G);
  return Layout{x0, x1 - x0, o0 - x0, o1 - x0};
}

enum Plane {
  kLayers, kLayersNext, kHeight, kUplift, kRainfall, kWaterHeight, kWaterFlux, kMass, kMassFlux,
  kVelocity, kVelocityFlux, kDebris, kDebrisFlux, kDebrisVelocity, kDebrisVelocityFlux, kPlanes
};
const char* const kPlaneName[kPlanes] = {
    "layers", "layers_next", "height", "uplift", "rainfall", "waterHeight", "waterFlux", "mass",
    "massFlux", "velocity", "velocityFlux", "debris", "debrisFlux", "debrisVelocity",
    "debrisVelocityFlux"};
const int kPlaneCh[kPlanes] = {2, 2, 1, 1, 1, 1, 1, 1, 1, 2, 2, 1, 1, 2, 2};
const Plane kField[4] = {kLayers, kVelocity, kWaterHeight, kDebrisVelocity};  // what particles read
const Plane kFluxFluvial[3] = {kWaterFlux, kMassFlux, kVelocityFlux};         // final after the fluvial launch
const Plane kFluxDebris[2] = {kDebrisFlux, kDebrisVelocityFlux};              // final after the debris launch

}  // namespace

}  // namespace soil

using soil::fail;

// what soillib_amd/_abi.py mirrors with ctypes
static_assert(sizeof(soil_xfer) == 24 && sizeof(soil_comm) == 56 && sizeof(soil_slab_ops) == 20 * 8 &&
                  sizeof(soil_slab_config) == 80 && sizeof(soil_slab_info) == 192,
              "soil_slab.h struct layout changed: update soillib_amd/_abi.py");

struct soil_slab {
  const soil_comm* comm = nullptr;
  const soil_slab_ops* ops = nullptr;
  soil_slab_ops* own_ops = nullptr;  // HIP back-end made by soil_slab_create
  soil_param param{};
  int rank = 0, world = 1;
  int64_t S = 0, W = 0, H = 0, G = 0, N = 0;
  soil::Layout lay{};
  float scale[3] = {0, 0, 0};
  uint64_t seed = 0, step_index = 0;
  bool trim = false, pair = false, host_ordered = false;
  int halo_need = 0;
  float* P[soil::kPlanes] = {};
  float* stage[soil::kPlanes][2] = {};  // [plane][0: from the neighbour above, 1: from below]
  soil_rng *rng = nullptr, *rng_debris = nullptr;
  float* remote0 = nullptr;
  float* ints = nullptr;  // all_ints scratch (world * 4 floats)
  int up = -1, down = -1;
  int64_t gu = 0, gd = 0;  // my ghost rows above / below
  // per rank: ghost rows (above, below) with fresh fields — those every walker reads (layers) and
  // the fluvial walkers' (velocity, waterHeight) in `fresh_all`, the debris walkers' own field
  // (debrisVelocity) in `fresh_debris`: debris walks end within a few cells at the example's
  // parameters, so that plane's halo is a few rows where the others' are hundreds
  std::vector<std::pair<int64_t, int64_t>> fresh_all, fresh_debris;
  std::vector<int> reach_hist, reach_hist_debris;  // max over ranks and both kinds; debris alone
  int64_t fallbacks = 0, rows_flux = 0, rows_field = 0, rows_full = 0;
  // The rows a particle launch is given (round 4): the owned rows and, either side, the ghost rows
  // with fresh fields plus kWindowMargin — not all G of them.  A walk that gets to a ghost row whose
  // fields are stale makes the step repeat its launches anyway (too_deep()); the rows beyond can only
  // be reached through such a row, so nothing valid ever happens there, and the pack pass, the tiles,
  // the first round's stores and the reach scan need not cover them.  [w0, w1): local rows.
  static constexpr int64_t kWindowMargin = 2;
  bool window = true;  // SOIL_HALO_WINDOW=0: every launch on all the ghost rows (A/B)
  int64_t w0 = 0, w1 = 0;
  int64_t rows_window = 0, rows_window_full = 0;  // ghost rows the launches were given so far / the bound
  // SOIL_SLAB_MIGRATE: walkers handed over at the far end of a shallow halo (G = SOIL_MIGRATE_HALO = 64
  // ghost rows a side, soil_slab_create below).  The boxes hold 64-byte walker records: out[0] / out[1] what this rank hands up / down, `inbox` what the
  // neighbours handed it ([from above | from below]); two counters on the device.
  int mode = SOIL_SLAB_DEEP_HALO;
  static constexpr int64_t kRecBytes = 64;
  void* out_box[2] = {nullptr, nullptr};
  void* inbox = nullptr;
  uint32_t* out_count = nullptr;
  int64_t box_cap = 0;
  int64_t passes = 0, walkers_handed = 0;

  // ---- helpers --------------------------------------------------------------------------------
  int64_t row_floats(int p) const { return W * soil::kPlaneCh[p]; }
  float* rowp(int p, int64_t local_row) const { return P[p] + local_row * row_floats(p); }
  int64_t peer_ghost(int peer) const {  // ghost rows `peer` holds on the side facing this rank
    if (peer < 0) return 0;
    const soil::Layout l = soil::slab_layout(peer, world, S, G);
    return peer > rank ? l.r0 : l.rows - l.r1;
  }
  soil_domain domain(int64_t r0, int64_t r1) const { return soil_domain{H, W, lay.x0, lay.rows, r0, r1}; }
  // the planes from local row `first` on (0: whole)
  soil_erosion_planes planes(int64_t first = 0) const {
    soil_erosion_planes q{};
    auto at = [&](int p) { return rowp(p, first); };
    q.layers = at(soil::kLayers), q.layers_next = at(soil::kLayersNext), q.height = at(soil::kHeight);
    q.uplift = at(soil::kUplift), q.rainfall = at(soil::kRainfall), q.waterHeight = at(soil::kWaterHeight);
    q.waterFlux = at(soil::kWaterFlux), q.mass = at(soil::kMass), q.massFlux = at(soil::kMassFlux);
    q.velocity = at(soil::kVelocity), q.velocityFlux = at(soil::kVelocityFlux), q.debris = at(soil::kDebris);
    q.debrisFlux = at(soil::kDebrisFlux), q.debrisVelocity = at(soil::kDebrisVelocity);
    q.debrisVelocityFlux = at(soil::kDebrisVelocityFlux);
    return q;
  }
  // the launch window of this step, from the ghost rows that hold fresh fields right now
  void set_window() {
    w0 = 0, w1 = lay.rows;
    if (!window || !trim || fresh_all.empty()) return;
    const auto &fa = fresh_all[static_cast<size_t>(rank)], &fd = fresh_debris[static_cast<size_t>(rank)];
    w0 = lay.r0 - std::min<int64_t>(lay.r0, std::max(fa.first, fd.first) + kWindowMargin);
    w1 = lay.r1 + std::min<int64_t>(lay.rows - lay.r1, std::max(fa.second, fd.second) + kWindowMargin);
  }
  soil_domain window_domain() const { return soil_domain{H, W, lay.x0 + w0, w1 - w0, lay.r0 - w0, lay.r1 - w0}; }
  void note_window() {
    rows_window += (lay.r0 - w0) + (w1 - lay.r1);
    rows_window_full += gu + gd;
  }
  void* stream(int lane) const { return ops->stream ? ops->stream(ops->ctx, lane) : nullptr; }
  int wire_status() const { return comm->status ? comm->status(comm->ctx) : SOIL_OK; }

#define SLAB_TRY(expr)                     \
  do {                                     \
    const int slab_rc_ = (expr);           \
    if (slab_rc_ != SOIL_OK) return slab_rc_; \
  } while (0)

  // ---- the wire ---------------------------------------------------------------------------------
  int exchange(const std::vector<soil_xfer>& sends, const std::vector<soil_xfer>& recvs, int lane) {
    if (sends.empty() && recvs.empty()) return SOIL_OK;
    if (host_ordered) SLAB_TRY(ops->sync(ops->ctx));
    return comm->exchange(comm->ctx, sends.data(), static_cast<int32_t>(sends.size()), recvs.data(),
                          static_cast<int32_t>(recvs.size()), stream(lane));
  }
  // every rank's k small non-negative ints, as out[rank * k + i]: one all-reduce of a zero-padded
  // vector (any wire that can sum will do); blocks until the host has them
  // `with_remote0`: the 8 sums of remote0 (the NaN walkers' deposits for global cell (0,0)) ride behind
  // the ints in the same all-reduce and stay on the device, at remote_sum()
  float* remote_sum() const { return ints + static_cast<int64_t>(world) * 4; }
  int all_ints(const int* mine, int k, std::vector<int>& out, bool with_remote0 = false) {
    const int64_t n = static_cast<int64_t>(world) * k;
    std::vector<float> h(static_cast<size_t>(n), 0.0f);
    for (int i = 0; i < k; ++i) h[static_cast<size_t>(rank) * k + i] = static_cast<float>(mine[i]);
    SLAB_TRY(ops->from_host(ops->ctx, ints, h.data(), n * 4));
    int64_t total = n;
    if (with_remote0) {  // (k == 4: the ints fill the block in front of remote_sum())
      SLAB_TRY(ops->fill_f32(ops->ctx, remote_sum(), 0.0f, 8, 0));
      SLAB_TRY(ops->add_f32(ops->ctx, remote_sum(), remote0, 8, 0));
      total = static_cast<int64_t>(world) * 4 + 8;
    }
    if (host_ordered) SLAB_TRY(ops->sync(ops->ctx));
    SLAB_TRY(comm->all_reduce_sum_f32(comm->ctx, ints, total, stream(0)));
    SLAB_TRY(ops->to_host(ops->ctx, h.data(), ints, n * 4));
    SLAB_TRY(wire_status());  // the copy waited for the stream: a wire that gave up meanwhile left garbage
    out.resize(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; ++i) out[static_cast<size_t>(i)] = static_cast<int>(std::lround(h[static_cast<size_t>(i)]));
    return SOIL_OK;
  }

  struct Counts { int64_t send_up, send_down, recv_up, recv_down; };

  // Ship the flux deposited into my ghost rows to their owners (rows nearest the boundary first),
  // add what the neighbours deposited for me, clear what I shipped.
  int flux_exchange(const soil::Plane* planes_, int n, const Counts* counts, int lane) {
    Counts c = counts ? *counts : Counts{gu, gd, peer_ghost(up), peer_ghost(down)};
    std::vector<soil_xfer> sends, recvs;
    for (int i = 0; i < n; ++i) {
      const int p = planes_[i];
      const int64_t rb = row_floats(p) * 4;
      if (up >= 0) {
        if (c.send_up) sends.push_back({rowp(p, lay.r0 - c.send_up), c.send_up * rb, up});
        if (c.recv_up) recvs.push_back({stage[p][0], c.recv_up * rb, up});
      }
      if (down >= 0) {
        if (c.send_down) sends.push_back({rowp(p, lay.r1), c.send_down * rb, down});
        if (c.recv_down) recvs.push_back({stage[p][1], c.recv_down * rb, down});
      }
    }
    rows_flux += (c.send_up + c.send_down) * n;
    rows_full += (gu + gd) * n;
    SLAB_TRY(exchange(sends, recvs, lane));
    for (int i = 0; i < n; ++i) {
      const int p = planes_[i];
      const int64_t rf = row_floats(p);
      // the up neighbour's lower ghost rows are my first owned rows
      if (up >= 0 && c.recv_up) SLAB_TRY(ops->add_f32(ops->ctx, rowp(p, lay.r0), stage[p][0], c.recv_up * rf, lane));
      if (down >= 0 && c.recv_down)
        SLAB_TRY(ops->add_f32(ops->ctx, rowp(p, lay.r1 - c.recv_down), stage[p][1], c.recv_down * rf, lane));
      if (up >= 0 && c.send_up) SLAB_TRY(ops->fill_f32(ops->ctx, rowp(p, lay.r0 - c.send_up), 0.0f, c.send_up * rf, lane));
      if (down >= 0 && c.send_down) SLAB_TRY(ops->fill_f32(ops->ctx, rowp(p, lay.r1), 0.0f, c.send_down * rf, lane));
    }
    return SOIL_OK;
  }

  // Refresh the ghost rows of the fields the next launches read.  need_*: rows I want from each
  // neighbour, give_*: rows I owe them (nearest the boundary first).  `layers_plane`: which of the
  // two layer buffers is the current one at this point of the step.
  int field_exchange(int layers_plane, const Counts* counts, int lane, const Counts* counts_debris = nullptr) {
    // Counts reused as (need_up, need_down, give_up, give_down); `counts_debris`: the same for
    // debrisVelocity alone (NULL: as the others)
    const Counts all{gu, gd, peer_ghost(up), peer_ghost(down)};
    const Counts ca = counts ? *counts : all;
    const Counts cdv = counts_debris ? *counts_debris : ca;
    std::vector<soil_xfer> sends, recvs;
    for (int i = 0; i < 4; ++i) {
      const int p = soil::kField[i] == soil::kLayers ? layers_plane : soil::kField[i];
      const Counts& c = soil::kField[i] == soil::kDebrisVelocity ? cdv : ca;
      const int64_t need_up = c.send_up, need_down = c.send_down, give_up = c.recv_up, give_down = c.recv_down;
      const int64_t rb = row_floats(p) * 4;
      if (up >= 0) {
        if (give_up) sends.push_back({rowp(p, lay.r0), give_up * rb, up});
        if (need_up) recvs.push_back({rowp(p, lay.r0 - need_up), need_up * rb, up});
      }
      if (down >= 0) {
        if (give_down) sends.push_back({rowp(p, lay.r1 - give_down), give_down * rb, down});
        if (need_down) recvs.push_back({rowp(p, lay.r1), need_down * rb, down});
      }
      rows_field += give_up + give_down;
    }
    rows_full += (peer_ghost(up) + peer_ghost(down)) * 4;
    return exchange(sends, recvs, lane);
  }

  // ---- measured reach ---------------------------------------------------------------------------
  // [rank] -> (rows above, rows below) its owned rows the deposits in `planes_` got to
  int reach(const soil::Plane* planes_, int n, std::vector<int>& all) {
    int32_t depth[2] = {0, 0};
    for (int i = 0; i < n; ++i)
      SLAB_TRY(ops->ghost_extent(ops->ctx, rowp(planes_[i], w0), w1 - w0, row_floats(planes_[i]), lay.r0 - w0, lay.r1 - w0, depth));
    const int mine[2] = {depth[0], depth[1]};
    return all_ints(mine, 2, all);
  }
  // ... of both launches of the overlapped pair in ONE exchange (round 4: it was two, each with its
  // blocking read-back), the NaN walkers' sums riding along: all[4 k .. 4 k + 3] = rank k's fluvial
  // (above, below), debris (above, below)
  int reach_pair(std::vector<int>& rf, std::vector<int>& rd) {
    int32_t df[2] = {0, 0}, dd[2] = {0, 0};
    for (int i = 0; i < 3; ++i)
      SLAB_TRY(ops->ghost_extent(ops->ctx, rowp(soil::kFluxFluvial[i], w0), w1 - w0, row_floats(soil::kFluxFluvial[i]), lay.r0 - w0, lay.r1 - w0, df));
    for (int i = 0; i < 2; ++i)
      SLAB_TRY(ops->ghost_extent(ops->ctx, rowp(soil::kFluxDebris[i], w0), w1 - w0, row_floats(soil::kFluxDebris[i]), lay.r0 - w0, lay.r1 - w0, dd));
    const int mine[4] = {df[0], df[1], dd[0], dd[1]};
    std::vector<int> all;
    SLAB_TRY(all_ints(mine, 4, all, true));
    rf.assign(static_cast<size_t>(2 * world), 0), rd.assign(static_cast<size_t>(2 * world), 0);
    for (int k = 0; k < world; ++k) {
      rf[static_cast<size_t>(2 * k)] = all[static_cast<size_t>(4 * k)], rf[static_cast<size_t>(2 * k + 1)] = all[static_cast<size_t>(4 * k + 1)];
      rd[static_cast<size_t>(2 * k)] = all[static_cast<size_t>(4 * k + 2)], rd[static_cast<size_t>(2 * k + 1)] = all[static_cast<size_t>(4 * k + 3)];
    }
    return SOIL_OK;
  }
  // Did a launch, on any rank, get within a row of ghost rows that were not refreshed?  (The cell
  // record of ghost row d is made of rows d - 1 .. d + 1.)  The same answer on every rank.
  bool too_deep(const std::vector<int>& r, bool debris = false) const {
    const auto& fresh = debris ? fresh_debris : fresh_all;
    for (int k = 0; k < world; ++k) {
      const soil::Layout l = soil::slab_layout(k, world, S, G);
      const int64_t f_up = fresh[static_cast<size_t>(k)].first, f_down = fresh[static_cast<size_t>(k)].second;
      const int64_t u = r[static_cast<size_t>(2 * k)], d = r[static_cast<size_t>(2 * k + 1)];
      if ((u >= f_up && f_up < l.r0) || (d >= f_down && f_down < l.rows - l.r1)) return true;
    }
    return false;
  }
  void fresh_everything() {
    fresh_all.clear();
    for (int k = 0; k < world; ++k) {
      const soil::Layout l = soil::slab_layout(k, world, S, G);
      fresh_all.emplace_back(l.r0, l.rows - l.r1);
    }
    fresh_debris = fresh_all;
  }
  // The prediction was too small: fetch the whole ghost zones of the fields as they stand (the cell
  // phase of this step has not touched them yet).
  int refresh_all() {
    ++fallbacks;
    SLAB_TRY(field_exchange(soil::kLayers, nullptr, 0));
    fresh_everything();
    return SOIL_OK;
  }
  // Ghost rows to refresh for the next step: as deep as the walks of the last steps got anywhere, a
  // tenth more and ten rows on top (the reach moves by a row or two from step to step); everything
  // while there is no history.
  void predict_need(const std::vector<int>& hist, int64_t& nu, int64_t& nd) const {
    if (hist.empty()) {
      nu = gu, nd = gd;
      return;
    }
    int64_t want = static_cast<int64_t>(1.1 * *std::max_element(hist.begin(), hist.end())) + 10;
    if (halo_need > 0) want = halo_need;
    nu = std::min(gu, want), nd = std::min(gd, want);
  }
  Counts counts_of(const std::vector<int>& r) const {
    return Counts{r[static_cast<size_t>(2 * rank)], r[static_cast<size_t>(2 * rank + 1)],
                  up >= 0 ? r[static_cast<size_t>(2 * up + 1)] : 0, down >= 0 ? r[static_cast<size_t>(2 * down)] : 0};
  }
  void note_reach(const std::vector<int>& a, const std::vector<int>& b) {
    int m = 0, md = 0;
    for (int v : a) m = std::max(m, v);
    for (int v : b) md = std::max(md, v);
    reach_hist.push_back(std::max(m, md));
    if (reach_hist.size() > 4) reach_hist.erase(reach_hist.begin());
    reach_hist_debris.push_back(md);
    if (reach_hist_debris.size() > 4) reach_hist_debris.erase(reach_hist_debris.begin());
  }
  int zero_planes(const soil::Plane* planes_, int n) {
    for (int i = 0; i < n; ++i) SLAB_TRY(ops->fill_f32(ops->ctx, P[planes_[i]], 0.0f, lay.rows * row_floats(planes_[i]), 0));
    return SOIL_OK;
  }

  // ---- SOIL_SLAB_MIGRATE: the particle phase ------------------------------------------------------
  // Per kind: a launch from the streams' spawns; the walkers that stepped onto a neighbour's row are
  // counted (two ints to the host), every rank learns every rank's counts (one all-reduce: the same
  // exchange that says whether anybody has anything left), the records travel to the neighbours, who
  // walk them on in a launch of their own; until no rank handed anything over.  A walker crosses at
  // most reach / S slab edges, so two or three passes; a walker may come back (it is handed over again).
  // Deposits land on owned rows and the shallow halo (soil_slab_create): the flux halo of a few rows goes
  // home once, after the last pass of both kinds.  Same walks as the single-domain launch: a walker is
  // handed over at the top of an iteration with its state untouched, and the neighbour's record of the
  // cell it stands on is made of the same fields.
  // counts[0..1] walkers this rank has just handed up / down (already in out_box); exchanges them, returns
  // the walkers that arrived (in `inbox`) and whether anybody anywhere handed anything over
  // Failures of the hand-over are decided on numbers every rank holds: a rank whose box overflowed sends
  // kOverflow in place of its count, and the arrivals of EVERY rank are checked from the gathered counts —
  // so all ranks leave the step together with the same error instead of one returning while its peers wait
  // in the next collective for ever.  An inbox holds 2 * box_cap records, but the launch that walks them on
  // has room for N (the record, destination and rank arrays of the tiled transport): more than N arrivals
  // of one kind are refused (unpaired launches let up to 2 N through).
  static constexpr int kOverflow = -1;
  int gathered_failure(const std::vector<int>& all, int k) const {
    for (int v : all)
      if (v == kOverflow)
        return fail(SOIL_ERR_OUT_OF_MEMORY, "slab step (migrate): more walkers left a slab in one pass than its boxes hold");
    for (int q = 0; q < world; ++q)
      for (int kind = 0; kind < k / 2; ++kind) {
        const int64_t from_up = q > 0 ? all[static_cast<size_t>(k * (q - 1) + 2 * kind + 1)] : 0;
        const int64_t from_down = q + 1 < world ? all[static_cast<size_t>(k * (q + 1) + 2 * kind)] : 0;
        if (from_up + from_down > std::min<int64_t>(N, 2 * box_cap))
          return fail(SOIL_ERR_OUT_OF_MEMORY, "slab step (migrate): more walkers arrive at rank " + std::to_string(q) +
                                                  " in one pass than its launch has room for (N = " + std::to_string(N) +
                                                  "): slabs this small against walks this long want the deep-halo mode");
      }
    return SOIL_OK;
  }
  int hand_over(const uint32_t counts[2], const void* const src[2], int64_t cap, int64_t& n_in, bool& any) {
    // (no neighbour on a side: the grid ends there, such walkers are out of bounds and never get here)
    const bool over = counts[0] > cap || counts[1] > cap;
    const int mine[2] = {over ? kOverflow : (up >= 0 ? static_cast<int>(counts[0]) : 0),
                         over ? kOverflow : (down >= 0 ? static_cast<int>(counts[1]) : 0)};
    std::vector<int> all;
    SLAB_TRY(all_ints(mine, 2, all));
    SLAB_TRY(gathered_failure(all, 2));
    int64_t total = 0;
    for (int v : all) total += v;
    any = total > 0;
    n_in = 0;
    if (!any) return SOIL_OK;
    walkers_handed += mine[0] + mine[1];
    const int64_t from_up = up >= 0 ? all[static_cast<size_t>(2 * up + 1)] : 0;
    const int64_t from_down = down >= 0 ? all[static_cast<size_t>(2 * down)] : 0;
    std::vector<soil_xfer> sends, recvs;
    if (mine[0]) sends.push_back({const_cast<void*>(src[0]), mine[0] * kRecBytes, up});
    if (mine[1]) sends.push_back({const_cast<void*>(src[1]), mine[1] * kRecBytes, down});
    if (from_up) recvs.push_back({inbox, from_up * kRecBytes, up});
    if (from_down) recvs.push_back({static_cast<char*>(inbox) + from_up * kRecBytes, from_down * kRecBytes, down});
    SLAB_TRY(exchange(sends, recvs, 0));
    n_in = from_up + from_down;
    return SOIL_OK;
  }
  // the launches that walk handed-over walkers of `kind` on, until nobody hands any over
  // (`first`: what the kind's spawn launch handed this rank's neighbours, already counted on the host)
  int migrate_on(int kind, const soil_erosion_planes& pl, const soil_domain& dom, const uint32_t first[2],
                 const void* const first_src[2], int64_t cap) {
    // (every pass walks a handed-over walker at least one step further: maxage + 2 passes always suffice)
    const int64_t max_pass = static_cast<int64_t>(std::min<uint64_t>(param.maxage, 1u << 20)) + 2;
    uint32_t counts[2] = {first[0], first[1]};
    const void* src[2] = {first_src[0], first_src[1]};
    for (int64_t pass = 0; pass <= max_pass; ++pass) {
      int64_t n_in = 0;
      bool any = false;
      SLAB_TRY(hand_over(counts, src, cap, n_in, any));
      if (!any) return SOIL_OK;
      SLAB_TRY(ops->fill_f32(ops->ctx, reinterpret_cast<float*>(out_count), 0.0f, 4, 0));  // (all-zero bits)
      if (n_in > 0) {
        static const bool verbose = std::getenv("SOIL_SLAB_VERBOSE") != nullptr;
        std::chrono::steady_clock::time_point t0;
        if (verbose) {
          SLAB_TRY(ops->sync(ops->ctx));
          t0 = std::chrono::steady_clock::now();
        }
        SLAB_TRY(ops->particles_pass(ops->ctx, kind, &pl, rng, nullptr, N, remote0, &dom, scale, &param, inbox, n_in,
                                     out_box[0], out_box[1], out_count, cap));
        ++passes;
        if (verbose) {
          SLAB_TRY(ops->sync(ops->ctx));
          std::fprintf(stderr, "[slab rank %d] kind %d pass %lld: %lld immigrants walked on in %.3f ms\n", rank, kind,
                       static_cast<long long>(pass + 1), static_cast<long long>(n_in),
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
      }
      SLAB_TRY(ops->to_host(ops->ctx, counts, out_count, 8));  // (an overflow travels with the next hand-over)
      src[0] = out_box[0], src[1] = out_box[1];
    }
    return fail(SOIL_ERR_HIP, "slab step (migrate): walkers still crossing after maxage + 2 passes");  // (the same pass on every rank)
  }
  // Both kinds' immigrants in one pass each round of hand-overs (the paired step): one all-reduce of the four
  // counts, one exchange of up to four transfers per neighbour pair (fluvial before debris on every link),
  // and — where a rank received walkers of both kinds — their two launches side by side like the step's
  // spawn launches (chains of a dozen thin rounds each: latency, not work).  The boxes stay halves: fluvial
  // leavers in the first half of out_up / out_down, debris in the second; the inbox holds the fluvial
  // arrivals (from above, from below), then the debris ones.  SOIL_MIGRATE_PAIR=0: kind by kind (migrate_on).
  int migrate_on_both(const soil_erosion_planes& pl, const soil_domain& dom, const uint32_t first[4], int64_t half) {
    const int64_t max_pass = static_cast<int64_t>(std::min<uint64_t>(param.maxage, 1u << 20)) + 2;
    uint32_t counts[4] = {first[0], first[1], first[2], first[3]};
    char* const up_d = static_cast<char*>(out_box[0]) + half * kRecBytes;
    char* const down_d = static_cast<char*>(out_box[1]) + half * kRecBytes;
    for (int64_t pass = 0; pass <= max_pass; ++pass) {
      bool over = false;
      for (int j = 0; j < 4; ++j) over = over || counts[j] > half;
      int mine[4] = {up >= 0 ? static_cast<int>(counts[0]) : 0, down >= 0 ? static_cast<int>(counts[1]) : 0,
                     up >= 0 ? static_cast<int>(counts[2]) : 0, down >= 0 ? static_cast<int>(counts[3]) : 0};
      if (over) mine[0] = mine[1] = mine[2] = mine[3] = kOverflow;  // every rank fails together (gathered_failure)
      std::vector<int> all;
      SLAB_TRY(all_ints(mine, 4, all));
      SLAB_TRY(gathered_failure(all, 4));
      int64_t total = 0;
      for (int v : all) total += v;
      if (total == 0) return SOIL_OK;
      walkers_handed += mine[0] + mine[1] + mine[2] + mine[3];
      // what the neighbour above handed DOWN is mine, and what the one below handed UP
      const int64_t in_f_up = up >= 0 ? all[static_cast<size_t>(4 * up + 1)] : 0, in_f_down = down >= 0 ? all[static_cast<size_t>(4 * down)] : 0;
      const int64_t in_d_up = up >= 0 ? all[static_cast<size_t>(4 * up + 3)] : 0, in_d_down = down >= 0 ? all[static_cast<size_t>(4 * down + 2)] : 0;
      const int64_t n_f = in_f_up + in_f_down, n_d = in_d_up + in_d_down;
      if (n_f + n_d > 2 * box_cap)  // (cannot happen: each kind's arrivals fit a half box a side)
        return fail(SOIL_ERR_OUT_OF_MEMORY, "slab step (migrate): more walkers arrive than the inbox holds");
      char* const in = static_cast<char*>(inbox);
      std::vector<soil_xfer> sends, recvs;
      if (mine[0]) sends.push_back({out_box[0], mine[0] * kRecBytes, up});
      if (mine[1]) sends.push_back({out_box[1], mine[1] * kRecBytes, down});
      if (mine[2]) sends.push_back({up_d, mine[2] * kRecBytes, up});
      if (mine[3]) sends.push_back({down_d, mine[3] * kRecBytes, down});
      if (in_f_up) recvs.push_back({in, in_f_up * kRecBytes, up});
      if (in_f_down) recvs.push_back({in + in_f_up * kRecBytes, in_f_down * kRecBytes, down});
      if (in_d_up) recvs.push_back({in + n_f * kRecBytes, in_d_up * kRecBytes, up});
      if (in_d_down) recvs.push_back({in + (n_f + in_d_up) * kRecBytes, in_d_down * kRecBytes, down});
      SLAB_TRY(exchange(sends, recvs, 0));
      SLAB_TRY(ops->fill_f32(ops->ctx, reinterpret_cast<float*>(out_count), 0.0f, 4, 0));  // (all-zero bits)
      static const bool verbose = std::getenv("SOIL_SLAB_VERBOSE") != nullptr;
      std::chrono::steady_clock::time_point t0;
      if (verbose && n_f + n_d > 0) {
        SLAB_TRY(ops->sync(ops->ctx));
        t0 = std::chrono::steady_clock::now();
      }
      if (n_f > 0 && n_d > 0) {
        SLAB_TRY(ops->particles_pass(ops->ctx, 2, &pl, rng, rng_debris, N, remote0, &dom, scale, &param, inbox, n_f | (n_d << 32),
                                     out_box[0], out_box[1], out_count, 2 * half));
        passes += 2;
      } else if (n_f > 0) {
        SLAB_TRY(ops->particles_pass(ops->ctx, 0, &pl, rng, nullptr, N, remote0, &dom, scale, &param, inbox, n_f, out_box[0],
                                     out_box[1], out_count, half));
        ++passes;
      } else if (n_d > 0) {
        SLAB_TRY(ops->particles_pass(ops->ctx, 1, &pl, rng, nullptr, N, remote0, &dom, scale, &param, inbox, n_d, up_d, down_d,
                                     out_count + 2, half));
        ++passes;
      }
      if (verbose && n_f + n_d > 0) {
        SLAB_TRY(ops->sync(ops->ctx));
        std::fprintf(stderr, "[slab rank %d] pass %lld: %lld fluvial + %lld debris immigrants walked on in %.3f ms\n", rank,
                     static_cast<long long>(pass + 1), static_cast<long long>(n_f), static_cast<long long>(n_d),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
      SLAB_TRY(ops->to_host(ops->ctx, counts, out_count, 16));
    }
    return fail(SOIL_ERR_HIP, "slab step (migrate): walkers still crossing after maxage + 2 passes");
  }
  // Per step: the spawn launches of both kinds (overlapped like soil_erode_step's when the runner pairs
  // its launches, else one after the other), then per kind the immigrants' launches.
  int migrate_particles(const soil_erosion_planes& pl, const soil_domain& dom, uint64_t off, bool paired,
                        soil_slab_mark_fn mark, void* mctx) {
    uint32_t c[4] = {0, 0, 0, 0};
    SLAB_TRY(ops->fill_f32(ops->ctx, reinterpret_cast<float*>(out_count), 0.0f, 4, 0));
    if (paired) {
      SLAB_TRY(ops->rng_seed(ops->ctx, rng_debris, N, seed, off + 2));
      SLAB_TRY(ops->particles_pass(ops->ctx, 2, &pl, rng, rng_debris, N, remote0, &dom, scale, &param, nullptr, 0,
                                   out_box[0], out_box[1], out_count, box_cap));
      passes += 2;
      SLAB_TRY(ops->to_host(ops->ctx, c, out_count, 16));
      if (mark) mark(mctx, 1);
      const int64_t half = box_cap / 2;  // (an overflow of a half box is reported by all ranks together: migrate_on*)
      // The fluvial leavers lie in the boxes' first halves, the debris ones in the second.  The launches
      // that walk immigrants on write THEIR leavers into the first halves only (cap = half): the debris
      // records stay where they are until their turn.
      const void* src_f[2] = {out_box[0], out_box[1]};
      const void* src_d[2] = {static_cast<char*>(out_box[0]) + half * kRecBytes, static_cast<char*>(out_box[1]) + half * kRecBytes};
      static const bool side_by_side = !(std::getenv("SOIL_MIGRATE_PAIR") && std::atoi(std::getenv("SOIL_MIGRATE_PAIR")) == 0);
      if (side_by_side) return migrate_on_both(pl, dom, c, half);
      SLAB_TRY(migrate_on(0, pl, dom, c, src_f, half));
      SLAB_TRY(migrate_on(1, pl, dom, c + 2, src_d, half));
      return SOIL_OK;
    }
    for (int kind = 0; kind < 2; ++kind) {
      SLAB_TRY(ops->fill_f32(ops->ctx, reinterpret_cast<float*>(out_count), 0.0f, 4, 0));
      SLAB_TRY(ops->particles_pass(ops->ctx, kind, &pl, rng, nullptr, N, remote0, &dom, scale, &param, nullptr, 0, out_box[0],
                                   out_box[1], out_count, box_cap));
      ++passes;
      SLAB_TRY(ops->to_host(ops->ctx, c, out_count, 8));
      if (kind == 0 && mark) mark(mctx, 1);
      const void* src[2] = {out_box[0], out_box[1]};
      SLAB_TRY(migrate_on(kind, pl, dom, c, src, box_cap));
    }
    return SOIL_OK;
  }
  // ---- one step ----------------------------------------------------------------------------------
  //   1 fluvial particles            -
  //   2 debris particles             overlapped: flux halo-accumulate of the fluvial planes
  //   3 flux halo, debris planes     exposed (the bands below need it)
  //   4 cell phase, bands next to    -
  //     the neighbours
  //   5 cell phase, interior rows    overlapped: field halo of the NEW layers, velocity,
  //                                  waterHeight, debrisVelocity -> the neighbours' ghost rows
  // With `pair` the two launches run overlapped on the back-end's own streams instead and all five
  // flux planes travel in step 3.
  int step(soil_slab_mark_fn mark, void* mctx) {
    auto mk = [&](int i) { if (mark) mark(mctx, i); };
    const soil_erosion_planes pl = planes();
    const soil_domain dom = domain(lay.r0, lay.r1);
    set_window();
    note_window();
    soil_erosion_planes plw = planes(w0);  // what the particle launches get
    soil_domain domw = window_domain();
    auto rewindow = [&]() {  // after refresh_all(): every ghost row is fresh, the window is all of them
      set_window();
      plw = planes(w0), domw = window_domain();
    };
    const uint64_t off = step_index * static_cast<uint64_t>(N);
    SLAB_TRY(ops->rng_seed(ops->ctx, rng, N, seed, off));
    SLAB_TRY(ops->fill_f32(ops->ctx, remote0, 0.0f, 8, 0));
    bool early = false;  // the fluvial planes' halo went out before the debris launch ended
    bool remote_summed = false;  // the NaN walkers' sums rode with the reach exchange (remote_sum())
    Counts cf{}, cd{};
    std::vector<int> rf, rd;
    mk(0);
    const bool paired = pair && ops->particles_pair && rng_debris;
    if (mode == SOIL_SLAB_MIGRATE) {
      SLAB_TRY(migrate_particles(pl, dom, off, paired, mark, mctx));
    } else if (paired) {
      // the debris launch draws from a tensor of its own, seeded where the fluvial launch leaves
      // the shared one in the sequential order
      SLAB_TRY(ops->rng_seed(ops->ctx, rng_debris, N, seed, off + 2));
      SLAB_TRY(ops->particles_pair(ops->ctx, &plw, rng, rng_debris, N, remote0, &domw, scale, &param));
      if (trim) {
        SLAB_TRY(reach_pair(rf, rd));
        remote_summed = true;
        // a fluvial walk reads layers, velocity, waterHeight; a debris walk layers and debrisVelocity
        if (too_deep(rf) || too_deep(rd) || too_deep(rd, true)) {  // rare: both launches again, on complete fields
          SLAB_TRY(refresh_all());
          rewindow();
          SLAB_TRY(zero_planes(soil::kFluxFluvial, 3));
          SLAB_TRY(zero_planes(soil::kFluxDebris, 2));
          SLAB_TRY(ops->fill_f32(ops->ctx, remote0, 0.0f, 8, 0));
          SLAB_TRY(ops->rng_seed(ops->ctx, rng, N, seed, off));
          SLAB_TRY(ops->rng_seed(ops->ctx, rng_debris, N, seed, off + 2));
          SLAB_TRY(ops->particles_pair(ops->ctx, &plw, rng, rng_debris, N, remote0, &domw, scale, &param));
          SLAB_TRY(reach_pair(rf, rd));
        }
        cf = counts_of(rf), cd = counts_of(rd);
        note_reach(rf, rd);
      }
      mk(1);
    } else {
      SLAB_TRY(ops->particles_fluvial(ops->ctx, &plw, rng, N, remote0, &domw, scale, &param));
      if (trim) {
        SLAB_TRY(reach(soil::kFluxFluvial, 3, rf));
        if (too_deep(rf)) {  // rare: repeat the launch on complete fields
          SLAB_TRY(refresh_all());
          rewindow();
          SLAB_TRY(zero_planes(soil::kFluxFluvial, 3));
          SLAB_TRY(ops->fill_f32(ops->ctx, remote0, 0.0f, 8, 0));
          SLAB_TRY(ops->rng_seed(ops->ctx, rng, N, seed, off));
          SLAB_TRY(ops->particles_fluvial(ops->ctx, &plw, rng, N, remote0, &domw, scale, &param));
          SLAB_TRY(reach(soil::kFluxFluvial, 3, rf));
        }
        cf = counts_of(rf);
      }
      mk(1);
      if (world > 1) {
        // the fluvial flux is final: its halo travels, and is added, while the debris launch runs
        SLAB_TRY(ops->fork(ops->ctx));
        SLAB_TRY(flux_exchange(soil::kFluxFluvial, 3, trim ? &cf : nullptr, 1));
        early = true;
      }
      SLAB_TRY(ops->particles_debris(ops->ctx, &plw, rng, N, remote0, &domw, scale, &param));
      if (trim) {
        SLAB_TRY(reach(soil::kFluxDebris, 2, rd));
        if (too_deep(rd) || too_deep(rd, true)) {
          SLAB_TRY(refresh_all());
          rewindow();
          SLAB_TRY(zero_planes(soil::kFluxDebris, 2));
          // the NaN walkers' debris deposits are entries 4..6 of remote0; the launch draws where
          // the fluvial one left the streams (two draws per particle on)
          SLAB_TRY(ops->fill_f32(ops->ctx, remote0 + 4, 0.0f, 4, 0));
          SLAB_TRY(ops->rng_seed(ops->ctx, rng, N, seed, off + 2));
          SLAB_TRY(ops->particles_debris(ops->ctx, &plw, rng, N, remote0, &domw, scale, &param));
          SLAB_TRY(reach(soil::kFluxDebris, 2, rd));
        }
        cd = counts_of(rd);
        note_reach(rf, rd);
      }
    }
    mk(2);
    if (world == 1) {
      SLAB_TRY(ops->cells(ops->ctx, &pl, &dom, scale, &param));
    } else {
      // NaN walkers of the other ranks -> global cell (0,0) (8 floats, latency only)
      const float* sums = remote0;
      if (remote_summed) {
        sums = remote_sum();
      } else {
        if (host_ordered) SLAB_TRY(ops->sync(ops->ctx));
        SLAB_TRY(comm->all_reduce_sum_f32(comm->ctx, remote0, 8, stream(0)));
      }
      if (rank == 0) {
        SLAB_TRY(ops->add_f32(ops->ctx, P[soil::kWaterFlux], sums + 0, 1, 0));
        SLAB_TRY(ops->add_f32(ops->ctx, P[soil::kMassFlux], sums + 1, 1, 0));
        SLAB_TRY(ops->add_f32(ops->ctx, P[soil::kVelocityFlux], sums + 2, 2, 0));
        SLAB_TRY(ops->add_f32(ops->ctx, P[soil::kDebrisFlux], sums + 4, 1, 0));
        SLAB_TRY(ops->add_f32(ops->ctx, P[soil::kDebrisVelocityFlux], sums + 5, 2, 0));
      }
      // rows whose flux is complete without the neighbours' contribution
      const int64_t i0 = std::min(lay.r1, lay.r0 + (up >= 0 ? peer_ghost(up) : 0));
      const int64_t i1 = std::max(i0, lay.r1 - (down >= 0 ? peer_ghost(down) : 0));
      // 1. the rest of the flux halo (exposed: the bands below need it)
      if (early) {
        SLAB_TRY(flux_exchange(soil::kFluxDebris, 2, trim ? &cd : nullptr, 0));
      } else {
        SLAB_TRY(flux_exchange(soil::kFluxFluvial, 3, trim ? &cf : nullptr, 0));
        SLAB_TRY(flux_exchange(soil::kFluxDebris, 2, trim ? &cd : nullptr, 0));
      }
      SLAB_TRY(ops->join(ops->ctx));  // ... and the part that travelled early
      mk(4);                          // 2 -> 4: flux halo not hidden by the debris launch
      // 2. the bands next to the neighbours first: they are what the neighbours' ghost rows get
      if (i0 > lay.r0) {
        const soil_domain b = domain(lay.r0, i0);
        SLAB_TRY(ops->cells(ops->ctx, &pl, &b, scale, &param));
      }
      if (lay.r1 > i1) {
        const soil_domain b = domain(i1, lay.r1);
        SLAB_TRY(ops->cells(ops->ctx, &pl, &b, scale, &param));
      }
      // 3. the field halo travels while the interior rows are computed (they are G rows away from
      //    anything the exchange reads or writes)
      Counts fc{}, fd{};
      const Counts *fcp = nullptr, *fdp = nullptr;
      if (trim) {  // as deep as next step's walks are expected to get; everybody says what it wants
        int64_t nu, nd, du, dd;
        predict_need(reach_hist, nu, nd);
        predict_need(reach_hist_debris, du, dd);
        // What the others want needs no exchange: the history holds the reach over ALL ranks (the same
        // numbers everywhere), and a rank's ghost depths follow from its place in the world (round 4:
        // this was a third blocking all-reduce per step).
        auto of = [&](int k, int i) {
          const soil::Layout l = soil::slab_layout(k, world, S, G);
          const int64_t ghost = (i & 1) ? l.rows - l.r1 : l.r0;
          const std::vector<int>& hist = i < 2 ? reach_hist : reach_hist_debris;
          if (hist.empty()) return ghost;
          int64_t want = static_cast<int64_t>(1.1 * *std::max_element(hist.begin(), hist.end())) + 10;
          if (halo_need > 0) want = halo_need;
          return std::min(ghost, want);
        };
        fc = Counts{nu, nd, up >= 0 ? of(up, 1) : 0, down >= 0 ? of(down, 0) : 0};
        fd = Counts{du, dd, up >= 0 ? of(up, 3) : 0, down >= 0 ? of(down, 2) : 0};
        fcp = &fc, fdp = &fd;
        fresh_all.clear(), fresh_debris.clear();
        for (int k = 0; k < world; ++k) {
          fresh_all.emplace_back(of(k, 0), of(k, 1));
          fresh_debris.emplace_back(of(k, 2), of(k, 3));
        }
      }
      SLAB_TRY(ops->fork(ops->ctx));
      SLAB_TRY(field_exchange(soil::kLayersNext, fcp, 1, fdp));
      if (i1 > i0) {
        const soil_domain b = domain(i0, i1);
        SLAB_TRY(ops->cells(ops->ctx, &pl, &b, scale, &param));
      }
      mk(5);  // 5 -> 3: field halo not hidden by the interior rows
      SLAB_TRY(ops->join(ops->ctx));
    }
    mk(3);
    std::swap(P[soil::kLayers], P[soil::kLayersNext]);
    ++step_index;
    return SOIL_OK;
  }
};

namespace soil {

// ------------------------------------------------------------------------------------------------
// 2. the HIP back-end
// ------------------------------------------------------------------------------------------------

namespace {

struct HipOps {
  hipStream_t main = nullptr, comm = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int32_t* depth_dev = nullptr;
  int device = 0;
  // The runner seeds a stream tensor right before every launch that draws from it, so the state
  // of every stream is known without the tensor: {seed, offset}, two draws further after a launch.
  // `rng_seed` only notes it down here; the launches take it as uniform streams (particles_common.hpp:
  // nothing read or written per stream, record slots only for the particles this slab owns — the
  // replay of the other ranks' streams was 36 bytes of traffic per stream, 7 of 8 streams on 8 GPUs).
  // SOIL_SLAB_UNIFORM=0: seed and read the tensors as the stand-alone entry points do.
  struct Noted { uint64_t seed, offset; };
  std::map<const soil_rng*, Noted> noted;
  bool uniform = true;
  Streams streams(soil_rng* rng) const {
    const auto it = noted.find(rng);
    if (!uniform || it == noted.end()) return streams_of(rng);
    return Streams{rng, true, it->second.seed, it->second.offset};
  }
  void drew(soil_rng* rng) {  // a launch took its two draws per stream
    const auto it = noted.find(rng);
    if (it != noted.end()) it->second.offset += 2;
  }
  // Flux planes left as they are by the cell phase (SOIL_CELLS_KEEP_FLUX: 84 instead of 112 B per
  // cell) whenever the launches that follow can overwrite them — the overlapped pair launch, whose
  // first rounds store whole tiles, empty ones included (SOIL_FLUX_OVERWRITE), as soil_erode_step
  // does on one GPU.  `flux_stale`: the planes `stale_planes` describes hold a finished step's
  // flux; a launch that only adds clears them first.  Someone who reads a flux plane through
  // soil_slab_plane between two steps sees the consumed flux of the last one in the owned rows.  The
  // GHOST rows of the flux planes are scratch without a stated content: inside the step's launch window
  // (set_window) they were stored by the first rounds and zeroed again behind the exchange, outside of
  // it — rows no walk can reach without the step repeating its launches on all rows — they keep whatever
  // an earlier, wider window left there.  Nothing reads them: the reach scan, the flux exchange and the
  // stores all stop at the window, and a fallback zeroes every row first (advisor finding of round 4).
  // SOIL_SLAB_LAZY=0 switches it off.
  bool lazy = true, last_pair = false, flux_stale = false;
  soil_erosion_planes stale_planes{};
  int64_t stale_cells = 0;
  int clear_stale() {
    if (!flux_stale) return SOIL_OK;
    const size_t b = sizeof(float) * static_cast<size_t>(stale_cells);
    SOIL_HIP(hipMemsetAsync(stale_planes.waterFlux, 0, b, main));
    SOIL_HIP(hipMemsetAsync(stale_planes.massFlux, 0, b, main));
    SOIL_HIP(hipMemsetAsync(stale_planes.velocityFlux, 0, 2 * b, main));
    SOIL_HIP(hipMemsetAsync(stale_planes.debrisFlux, 0, b, main));
    SOIL_HIP(hipMemsetAsync(stale_planes.debrisVelocityFlux, 0, 2 * b, main));
    flux_stale = false;
    return SOIL_OK;
  }
};

#define HIP_OPS(c) HipOps& o = *static_cast<HipOps*>(c)

int hip_alloc(void* c, void** out, int64_t bytes) {
  HIP_OPS(c);
  SOIL_HIP(hipMalloc(out, static_cast<size_t>(bytes > 0 ? bytes : 4)));
  // cleared ON THE MAIN LANE: a hipMemset would run on the null stream, which the two (non-blocking)
  // lanes are not ordered with — it could land after the first kernels that fill the block
  SOIL_HIP(hipMemsetAsync(*out, 0, static_cast<size_t>(bytes > 0 ? bytes : 4), o.main));
  return SOIL_OK;
}
int hip_release(void*, void* p) {
  if (p) SOIL_HIP(hipFree(p));
  return SOIL_OK;
}
int hip_fill(void* c, float* dst, float v, int64_t n, int32_t lane) {
  HIP_OPS(c);
  return soil_set_f32(dst, v, n, lane ? o.comm : o.main);
}
int hip_add(void* c, float* dst, const float* src, int64_t n, int32_t lane) {
  HIP_OPS(c);
  return soil_add_f32(dst, src, n, lane ? o.comm : o.main);
}
int hip_seed(void* c, soil_rng* rng, int64_t N, uint64_t seed, uint64_t offset) {
  HIP_OPS(c);
  if (o.uniform) {
    o.noted[rng] = HipOps::Noted{seed, offset};
    return SOIL_OK;
  }
  return soil_rng_seed(rng, N, seed, offset, o.main);
}
int hip_fluvial(void* c, const soil_erosion_planes* p, soil_rng* rng, int64_t N, float* remote0,
                const soil_domain* dom, const float scale[3], const soil_param* param) {
  HIP_OPS(c);
  const Dom d = to_dom(dom);
  if (int rc = check_domain(d); rc != SOIL_OK) return rc;
  if (int rc = o.clear_stale(); rc != SOIL_OK) return rc;
  o.last_pair = false;
  const int rc = particles_fluvial_streams(*p, o.streams(rng), N, remote0, d, Scale3{scale[0], scale[1], scale[2]},
                                           *param, o.main);
  o.drew(rng);
  return rc;
}
int hip_debris(void* c, const soil_erosion_planes* p, soil_rng* rng, int64_t N, float* remote0,
               const soil_domain* dom, const float scale[3], const soil_param* param) {
  HIP_OPS(c);
  const Dom d = to_dom(dom);
  if (int rc = check_domain(d); rc != SOIL_OK) return rc;
  if (int rc = o.clear_stale(); rc != SOIL_OK) return rc;
  o.last_pair = false;
  const int rc = particles_debris_streams(*p, o.streams(rng), N, remote0, d, Scale3{scale[0], scale[1], scale[2]},
                                          *param, o.main);
  o.drew(rng);
  return rc;
}
int hip_pair(void* c, const soil_erosion_planes* p, soil_rng* rf, soil_rng* rd, int64_t N, float* remote0,
             const soil_domain* dom, const float scale[3], const soil_param* param) {
  HIP_OPS(c);
  const Dom d = to_dom(dom);
  if (int rc = check_domain(d); rc != SOIL_OK) return rc;
  const int rc = particles_pair_streams(*p, o.streams(rf), o.streams(rd), N, remote0, d,
                                        Scale3{scale[0], scale[1], scale[2]}, *param, o.main, o.flux_stale);
  o.flux_stale = false;
  o.last_pair = true;
  o.drew(rf);
  o.drew(rd);
  return rc;
}
// kind 0 / 1: one launch of that kind (soil_slab.h).  kind 2 with an inbox: both kinds' immigrants walked on
// side by side (n_in = fluvial | debris << 32, the fluvial records first).  kind 2: both kinds' SPAWN launches overlapped, as
// hip_pair runs them (`rng`: the fluvial streams; the debris launch draws from `rng_debris`, two draws on):
// the boxes are halves — fluvial records in the first `cap / 2` slots of out_up / out_down, debris in the
// second, out_count[0..3] = fluvial up, down, debris up, down.
int hip_pass(void* c, int32_t kind, const soil_erosion_planes* p, soil_rng* rng, soil_rng* rng_debris, int64_t N, float* remote0,
             const soil_domain* dom, const float scale[3], const soil_param* param, const void* inbox, int64_t n_in,
             void* out_up, void* out_down, uint32_t* out_count, int64_t cap) {
  HIP_OPS(c);
  const Dom d = to_dom(dom);
  if (int rc = check_domain(d); rc != SOIL_OK) return rc;
  SOIL_REQUIRE(kind >= 0 && kind <= 2, "particles_pass: kind 0 (fluvial), 1 (debris) or 2 (both spawn launches)");
  SOIL_REQUIRE(N > 0 && N <= 0x7fffffffll && d.H * d.W <= 0x7fffffffll && d.H < (1 << 24) && d.W < (1 << 24),
               "particles_pass: the tiled launch shape needs 1 .. 2^31 - 1 particles and cells, rows and columns below 2^24");
  SOIL_REQUIRE(n_in >= 0 && (kind == 2 || n_in <= 0xffffffffll) && cap >= 0 && cap <= 0xffffffffll, "particles_pass: bad record counts");
  const Scale3 s3{scale[0], scale[1], scale[2]};
  if (kind == 2) {
    SOIL_REQUIRE(rng_debris, "particles_pass: the overlapped launches need both kinds' streams");
    const uint32_t half = static_cast<uint32_t>(cap / 2);
    MigrateBox bf, bd;
    bf.up = out_up, bf.down = out_down, bf.count = out_count, bf.cap = half;
    bd.up = static_cast<char*>(out_up) + static_cast<size_t>(half) * 64, bd.down = static_cast<char*>(out_down) + static_cast<size_t>(half) * 64;
    bd.count = out_count + 2, bd.cap = half;
    if (inbox) {  // both kinds' immigrants: n_in = fluvial count | debris count << 32, fluvial records first
      const uint32_t n_f = static_cast<uint32_t>(n_in & 0xffffffffll), n_d = static_cast<uint32_t>(n_in >> 32);
      SOIL_REQUIRE(n_f > 0 && n_d > 0, "particles_pass: the overlapped immigrants' launches want walkers of both kinds");
      if (int rc = o.clear_stale(); rc != SOIL_OK) return rc;
      return launch_pair_tiled(*p, o.streams(rng), o.streams(rng_debris), N, remote0, d, s3, *param, o.main, false, bf, bd,
                               inbox, n_f, static_cast<const char*>(inbox) + static_cast<size_t>(n_f) * 64, n_d);
    }
    const int rc = launch_pair_tiled(*p, o.streams(rng), o.streams(rng_debris), N, remote0, d, s3, *param, o.main,
                                     o.flux_stale, bf, bd);
    o.flux_stale = false;
    o.last_pair = true;
    o.drew(rng);
    o.drew(rng_debris);
    return rc;
  }
  if (int rc = o.clear_stale(); rc != SOIL_OK) return rc;
  if (!inbox) o.last_pair = false;  // (a launch of immigrants behind the overlapped pair leaves the step a paired one)
  MigrateBox box;
  box.up = out_up, box.down = out_down, box.count = out_count, box.cap = static_cast<uint32_t>(cap);
  const int rc = launch_pass_tiled(kind, *p, o.streams(rng), N, remote0, d, s3, *param, o.main, inbox,
                                   static_cast<uint32_t>(n_in), box);
  if (!inbox) o.drew(rng);
  return rc;
}
int hip_cells(void* c, const soil_erosion_planes* p, const soil_domain* dom, const float scale[3],
              const soil_param* param) {
  HIP_OPS(c);
  if (dom->r1 <= dom->r0) return SOIL_OK;
  const bool keep = o.lazy && o.last_pair;
  if (keep) {
    o.flux_stale = true;
    o.stale_planes = *p;
    o.stale_cells = dom->rows * dom->W;
  }
  return soil_erode_cells_fused_ex(p, dom, scale, param, keep ? SOIL_CELLS_KEEP_FLUX : 0, o.main);
}
int hip_extent(void* c, const float* plane, int64_t rows, int64_t row_floats, int64_t r0, int64_t r1,
               int32_t depth[2]) {
  HIP_OPS(c);
  SOIL_HIP(hipMemsetAsync(o.depth_dev, 0, 8, o.main));
  if (int rc = soil_ghost_extent(o.depth_dev, plane, rows, row_floats, r0, r1, o.main); rc != SOIL_OK) return rc;
  int32_t h[2] = {0, 0};
  SOIL_HIP(hipMemcpyAsync(h, o.depth_dev, 8, hipMemcpyDeviceToHost, o.main));
  SOIL_HIP(hipStreamSynchronize(o.main));
  depth[0] = std::max(depth[0], h[0]);
  depth[1] = std::max(depth[1], h[1]);
  return SOIL_OK;
}
int hip_noise(void* c, float* out, int64_t rows, int64_t W, int64_t x0, const soil_noise_param* p) {
  HIP_OPS(c);
  return soil_noise_window(out, rows, W, x0, p, o.main);
}
int hip_layers(void* c, float* layers, const float* bed, int64_t n) {
  HIP_OPS(c);
  return soil_layers_from_planes(layers, bed, nullptr, n, o.main);
}
int hip_to_host(void* c, void* dst, const void* src, int64_t bytes) {
  HIP_OPS(c);
  SOIL_HIP(hipMemcpyAsync(dst, src, static_cast<size_t>(bytes), hipMemcpyDeviceToHost, o.main));
  SOIL_HIP(hipStreamSynchronize(o.main));
  return SOIL_OK;
}
int hip_from_host(void* c, void* dst, const void* src, int64_t bytes) {
  HIP_OPS(c);
  SOIL_HIP(hipMemcpyAsync(dst, src, static_cast<size_t>(bytes), hipMemcpyHostToDevice, o.main));
  SOIL_HIP(hipStreamSynchronize(o.main));  // `src` is the caller's stack
  return SOIL_OK;
}
int hip_fork(void* c) {
  HIP_OPS(c);
  SOIL_HIP(hipEventRecord(o.ev_fork, o.main));
  SOIL_HIP(hipStreamWaitEvent(o.comm, o.ev_fork, 0));
  return SOIL_OK;
}
int hip_join(void* c) {
  HIP_OPS(c);
  SOIL_HIP(hipEventRecord(o.ev_join, o.comm));
  SOIL_HIP(hipStreamWaitEvent(o.main, o.ev_join, 0));
  return SOIL_OK;
}
int hip_sync(void* c) {
  HIP_OPS(c);
  SOIL_HIP(hipStreamSynchronize(o.main));
  SOIL_HIP(hipStreamSynchronize(o.comm));
  return SOIL_OK;
}
void* hip_stream(void* c, int32_t lane) {
  HIP_OPS(c);
  return lane ? o.comm : o.main;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// 3. RCCL
// ------------------------------------------------------------------------------------------------

namespace {

// ---- a bounded wait for a wire ---------------------------------------------------------------------
// A transfer that never completes must become an error, not a process that sits until somebody kills
// it (the driver's one execution of the RCCL point-to-point path hung for 300 s and
// took the whole GPU test session with it).  WireWatch is a small watchdog thread per communicator:
// the wire tells it when the calling thread enters / leaves a library call and hands it a completion
// token (a HIP event recorded behind the operation) for everything it has put on a stream; when a call
// or a token is older than the timeout the watchdog calls the wire's abort hook (ncclCommAbort: the
// blocked call returns, the device kernel sees the abort flag and exits, the host's stream waits come
// back) and every later call on the communicator returns SOIL_ERR_COMM with the description.
struct WireOp {
  const char* op = "";
  int32_t n_sends = 0, n_recvs = 0, peer = -1;
  int64_t bytes = 0;
  std::string describe() const {
    char b[160];
    std::snprintf(b, sizeof b, "%s (%d sends, %d receives, first peer %d, %lld bytes)", op, n_sends, n_recvs, peer,
                  static_cast<long long>(bytes));
    return b;
  }
};

class WireWatch {
 public:
  using Clock = std::chrono::steady_clock;
  struct Hooks {
    std::function<void()> thread_start;        // e.g. hipSetDevice on the watchdog thread
    std::function<int(void*)> token_state;     // 1 complete, 0 pending, < 0 broken
    std::function<void(void*)> token_release;  // back to the pool
    std::function<std::string()> async_error;  // "" while the library is content
    std::function<void()> abort;               // must make blocked calls and stream work return
    std::string where;                         // "librccl /path (2.26.6), rank 3 of 8"
  };
  WireWatch(double timeout_s, Hooks h) : timeout_(timeout_s), hooks_(std::move(h)) {
    thread_ = std::thread([this] { run(); });
  }
  ~WireWatch() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_all();
    if (thread_.joinable()) thread_.join();
    for (auto& it : pending_) hooks_.token_release(it.token);
  }
  // brackets a library call on the calling thread
  void enter(const WireOp& op) {
    std::lock_guard<std::mutex> l(m_);
    in_call_ = true, call_ = op, call_t0_ = Clock::now();
  }
  void leave() {
    std::lock_guard<std::mutex> l(m_);
    in_call_ = false;
  }
  void track(void* token, const WireOp& op) {
    std::lock_guard<std::mutex> l(m_);
    pending_.push_back(Item{token, op, Clock::now()});
  }
  bool dead() {
    std::lock_guard<std::mutex> l(m_);
    return dead_;
  }
  std::string error() {
    std::lock_guard<std::mutex> l(m_);
    return error_;
  }
  // blocks until everything tracked has completed or the watchdog has given up
  void drain() {
    std::unique_lock<std::mutex> l(m_);
    idle_.wait(l, [this] { return dead_ || pending_.empty(); });
  }
  double timeout() const { return timeout_; }

 private:
  struct Item {
    void* token;
    WireOp op;
    Clock::time_point t0;
  };
  void give_up(std::unique_lock<std::mutex>& l, const std::string& why) {
    dead_ = true;
    error_ = why + " on " + hooks_.where + "; the communicator was aborted";
    l.unlock();
    hooks_.abort();  // not under the lock: the blocked caller takes it in leave()
    l.lock();
    idle_.notify_all();
  }
  void run() {
    if (hooks_.thread_start) hooks_.thread_start();
    std::unique_lock<std::mutex> l(m_);
    while (!stop_) {
      cv_.wait_for(l, std::chrono::milliseconds(pending_.empty() && !in_call_ ? 100 : 10));
      if (stop_ || dead_) continue;
      const auto now = Clock::now();
      auto age = [&](Clock::time_point t) { return std::chrono::duration<double>(now - t).count(); };
      while (!pending_.empty()) {
        const int st = hooks_.token_state(pending_.front().token);
        if (st == 0) break;
        if (st < 0) {
          give_up(l, "the device reported a failure behind " + pending_.front().op.describe());
          break;
        }
        hooks_.token_release(pending_.front().token);
        pending_.pop_front();
      }
      if (dead_) continue;
      if (pending_.empty()) idle_.notify_all();
      if (hooks_.async_error) {
        const std::string e = hooks_.async_error();
        if (!e.empty()) {
          give_up(l, "asynchronous error " + e);
          continue;
        }
      }
      char sec[32];
      std::snprintf(sec, sizeof sec, "%.1f s", timeout_);
      if (in_call_ && age(call_t0_) > timeout_)
        give_up(l, std::string("no return from ") + call_.describe() + " within " + sec + " (SOIL_RCCL_TIMEOUT_S)");
      else if (!pending_.empty() && age(pending_.front().t0) > timeout_)
        give_up(l, std::string("no completion of ") + pending_.front().op.describe() + " within " + sec +
                       " (SOIL_RCCL_TIMEOUT_S)");
    }
  }
  const double timeout_;
  Hooks hooks_;
  std::mutex m_;
  std::condition_variable cv_, idle_;
  std::deque<Item> pending_;
  bool in_call_ = false, stop_ = false, dead_ = false;
  WireOp call_;
  Clock::time_point call_t0_;
  std::string error_;
  std::thread thread_;
};

double env_seconds(const char* name, double dflt) {
  if (const char* e = std::getenv(name)) {
    char* end = nullptr;
    const double v = std::strtod(e, &end);
    if (end != e && v > 0) return v;
  }
  return dflt;
}

// The few entry points of rccl.h this file needs, bound at run time.  (ncclUniqueId is 128 bytes,
// ncclComm_t an opaque pointer; ncclInt8 = 0, ncclFloat32 = 7, ncclSum = 0, ncclInProgress = 7 —
// nccl.h's enums.)
struct Id128 { char bytes[128]; };
struct Rccl {
  void* lib = nullptr;
  std::string path;  // the file the symbols came from (dladdr)
  int version = 0;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, /* ncclUniqueId by value */ Id128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommAbort)(void*) = nullptr;
  int (*CommGetAsyncError)(void*, int*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  int (*CommCuDevice)(void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

// Which librccl: SOIL_RCCL_LIB if set; else a copy the process has loaded already (a Python host
// that imported torch has the wheel's: two RCCLs in one process would each bring their idea of the
// HIP runtime); else the ROCm installation's librccl.so.1 through the usual search path, which is
// what a torch-less C++ host gets.  soil_comm_rccl_library() reports the outcome.
int rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);  // several host threads may make their communicators at once
  if (g_rccl.lib) return SOIL_OK;
  void* h = nullptr;
  if (const char* e = std::getenv("SOIL_RCCL_LIB")) {
    h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(SOIL_ERR_INVALID_ARGUMENT, std::string("SOIL_RCCL_LIB: ") + dlerror());
  }
  for (const char* name : {"librccl.so", "librccl.so.1"})
    if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
    if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(SOIL_ERR_INVALID_ARGUMENT, std::string("librccl not found (set SOIL_RCCL_LIB): ") + dlerror());
  auto sym = [&](const char* n) { return dlsym(h, n); };
#define RCCL_BIND(field, name)                                                       \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(sym(name));              \
  if (!g_rccl.field) return fail(SOIL_ERR_INVALID_ARGUMENT, std::string("librccl lacks ") + name)
  RCCL_BIND(GetUniqueId, "ncclGetUniqueId");
  RCCL_BIND(CommInitRank, "ncclCommInitRank");
  RCCL_BIND(CommDestroy, "ncclCommDestroy");
  RCCL_BIND(CommAbort, "ncclCommAbort");
  RCCL_BIND(CommGetAsyncError, "ncclCommGetAsyncError");
  RCCL_BIND(GroupStart, "ncclGroupStart");
  RCCL_BIND(GroupEnd, "ncclGroupEnd");
  RCCL_BIND(Send, "ncclSend");
  RCCL_BIND(Recv, "ncclRecv");
  RCCL_BIND(AllReduce, "ncclAllReduce");
  RCCL_BIND(CommCount, "ncclCommCount");
  RCCL_BIND(CommUserRank, "ncclCommUserRank");
  RCCL_BIND(CommCuDevice, "ncclCommCuDevice");
  RCCL_BIND(GetErrorString, "ncclGetErrorString");
  RCCL_BIND(GetVersion, "ncclGetVersion");
#undef RCCL_BIND
  Dl_info info{};
  if (dladdr(reinterpret_cast<void*>(g_rccl.GetVersion), &info) && info.dli_fname) g_rccl.path = info.dli_fname;
  (void)g_rccl.GetVersion(&g_rccl.version);
  g_rccl.lib = h;
  return SOIL_OK;
}

std::string rccl_where() {
  char v[48];
  std::snprintf(v, sizeof v, " (RCCL/NCCL %d.%d.%d)", g_rccl.version / 10000, g_rccl.version / 100 % 100,
                g_rccl.version % 100);
  return "librccl " + (g_rccl.path.empty() ? std::string("?") : g_rccl.path) + v;
}

int rccl_fail(int e, const char* what) {
  return fail(SOIL_ERR_COMM, std::string("RCCL error ") + std::to_string(e) + " (" +
                                 (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?") + ") from " + what + " on " +
                                 rccl_where());
}

struct RcclCtx {
  void* comm = nullptr;
  float* scratch = nullptr;  // barrier
  int device = 0, rank = 0, world = 1;
  std::mutex events_mutex;
  std::vector<hipEvent_t> events;  // pool of completion tokens
  std::unique_ptr<WireWatch> watch;

  hipEvent_t take_event() {
    {
      std::lock_guard<std::mutex> l(events_mutex);
      if (!events.empty()) {
        hipEvent_t e = events.back();
        events.pop_back();
        return e;
      }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    return e;
  }
  void give_event(hipEvent_t e) {
    std::lock_guard<std::mutex> l(events_mutex);
    events.push_back(e);
  }
  int dead_rc() { return fail(SOIL_ERR_COMM, watch->error()); }
  // the operation just issued on `st`: a token behind it for the watchdog
  int issued(hipStream_t st, const WireOp& op) {
    if (hipEvent_t e = take_event()) {
      if (hipEventRecord(e, st) == hipSuccess) {
        watch->track(e, op);
      } else {
        (void)hipGetLastError();
        give_event(e);
      }
    }
    return SOIL_OK;
  }
};

// every RCCL call of a live communicator: bracketed for the watchdog; a failure after the watchdog has
// aborted is reported as the timeout it was
#define SOIL_RCCL_CALL(r, expr)                                        \
  do {                                                                 \
    const int rccl_e_ = (expr);                                        \
    if (rccl_e_ != 0) {                                                \
      (r).watch->leave();                                              \
      return (r).watch->dead() ? (r).dead_rc() : rccl_fail(rccl_e_, #expr); \
    }                                                                  \
  } while (0)
#define SOIL_RCCL(expr)                                  \
  do {                                                   \
    const int rccl_e_ = (expr);                          \
    if (rccl_e_ != 0) return rccl_fail(rccl_e_, #expr);  \
  } while (0)

int rccl_exchange(void* c, const soil_xfer* sends, int32_t ns, const soil_xfer* recvs, int32_t nr, void* stream) {
  RcclCtx& r = *static_cast<RcclCtx*>(c);
  if (r.watch->dead()) return r.dead_rc();
  hipStream_t st = static_cast<hipStream_t>(stream);
  WireOp op;
  op.op = "exchange [ncclGroupStart .. ncclRecv/ncclSend .. ncclGroupEnd]";
  op.n_sends = ns, op.n_recvs = nr, op.peer = ns ? sends[0].peer : (nr ? recvs[0].peer : -1);
  for (int i = 0; i < ns; ++i) op.bytes += sends[i].bytes;
  for (int i = 0; i < nr; ++i) op.bytes += recvs[i].bytes;
  r.watch->enter(op);
  SOIL_RCCL_CALL(r, g_rccl.GroupStart());
  for (int i = 0; i < nr; ++i)
    SOIL_RCCL_CALL(r, g_rccl.Recv(recvs[i].ptr, static_cast<size_t>(recvs[i].bytes), 0 /* ncclInt8 */, recvs[i].peer,
                                  r.comm, st));
  for (int i = 0; i < ns; ++i)
    SOIL_RCCL_CALL(r, g_rccl.Send(sends[i].ptr, static_cast<size_t>(sends[i].bytes), 0, sends[i].peer, r.comm, st));
  SOIL_RCCL_CALL(r, g_rccl.GroupEnd());
  r.watch->leave();
  if (r.watch->dead()) return r.dead_rc();
  return r.issued(st, op);
}
int rccl_all_reduce(void* c, float* buf, int64_t n, void* stream) {
  RcclCtx& r = *static_cast<RcclCtx*>(c);
  if (r.watch->dead()) return r.dead_rc();
  hipStream_t st = static_cast<hipStream_t>(stream);
  WireOp op;
  op.op = "all_reduce_sum_f32 [ncclAllReduce]", op.bytes = n * 4;
  r.watch->enter(op);
  SOIL_RCCL_CALL(r, g_rccl.AllReduce(buf, buf, static_cast<size_t>(n), 7 /* ncclFloat32 */, 0 /* ncclSum */, r.comm, st));
  r.watch->leave();
  if (r.watch->dead()) return r.dead_rc();
  return r.issued(st, op);
}
int rccl_barrier(void* c) {
  RcclCtx& r = *static_cast<RcclCtx*>(c);
  if (r.watch->dead()) return r.dead_rc();
  // every rank's own device work first (the runner's streams are non-blocking: the null stream does not
  // wait for them), then the collective: past the barrier all ranks' earlier work is done
  SOIL_HIP(hipDeviceSynchronize());
  WireOp op;
  op.op = "barrier [ncclAllReduce of one word]", op.bytes = 4;
  r.watch->enter(op);
  SOIL_RCCL_CALL(r, g_rccl.AllReduce(r.scratch, r.scratch, 1, 7, 0, r.comm, nullptr));
  r.watch->leave();
  r.issued(nullptr, op);
  // a bounded wait: the watchdog aborts a transfer that does not complete, which ends the kernel
  for (;;) {
    const hipError_t q = hipStreamQuery(nullptr);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery (barrier)", __FILE__, __LINE__);
    if (r.watch->dead()) return r.dead_rc();
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  if (r.watch->dead()) return r.dead_rc();
  return SOIL_OK;
}
// what the runner asks after it has waited for its streams: did the wire give up meanwhile?
int rccl_status(void* c) {
  RcclCtx& r = *static_cast<RcclCtx*>(c);
  return r.watch->dead() ? r.dead_rc() : SOIL_OK;
}

// ---- a wire that never delivers (tests) ------------------------------------------------------------
// soil_comm_wedged_create: exchange / all-reduce / barrier block like a transfer whose peer never
// shows up, under the same watchdog as the RCCL wire; its abort hook releases the blocked caller.
struct WedgedCtx {
  std::mutex m;
  std::condition_variable cv;
  bool aborted = false;
  std::unique_ptr<WireWatch> watch;
  int block(const WireOp& op) {
    if (watch->dead()) return fail(SOIL_ERR_COMM, watch->error());
    watch->enter(op);
    {
      std::unique_lock<std::mutex> l(m);
      cv.wait(l, [this] { return aborted; });
    }
    watch->leave();
    return fail(SOIL_ERR_COMM, watch->error());
  }
};
int wedged_exchange(void* c, const soil_xfer* sends, int32_t ns, const soil_xfer* recvs, int32_t nr, void*) {
  WireOp op;
  op.op = "exchange [wedged test wire]", op.n_sends = ns, op.n_recvs = nr;
  op.peer = ns ? sends[0].peer : (nr ? recvs[0].peer : -1);
  for (int i = 0; i < ns; ++i) op.bytes += sends[i].bytes;
  for (int i = 0; i < nr; ++i) op.bytes += recvs[i].bytes;
  return static_cast<WedgedCtx*>(c)->block(op);
}
int wedged_all_reduce(void* c, float*, int64_t n, void*) {
  WireOp op;
  op.op = "all_reduce_sum_f32 [wedged test wire]", op.bytes = n * 4;
  return static_cast<WedgedCtx*>(c)->block(op);
}
int wedged_barrier(void* c) {
  WireOp op;
  op.op = "barrier [wedged test wire]";
  return static_cast<WedgedCtx*>(c)->block(op);
}
int wedged_status(void* c) {
  WedgedCtx& w = *static_cast<WedgedCtx*>(c);
  return w.watch->dead() ? fail(SOIL_ERR_COMM, w.watch->error()) : SOIL_OK;
}

// a world of one: exchanges with oneself are device copies
int self_exchange(void*, const soil_xfer* sends, int32_t ns, const soil_xfer* recvs, int32_t nr, void* stream) {
  SOIL_REQUIRE(ns == nr, "self comm: sends and receives must pair up");
  for (int i = 0; i < ns; ++i) {
    SOIL_REQUIRE(sends[i].bytes == recvs[i].bytes && sends[i].peer == 0 && recvs[i].peer == 0,
                 "self comm: mismatched transfer");
    SOIL_HIP(hipMemcpyAsync(recvs[i].ptr, sends[i].ptr, static_cast<size_t>(sends[i].bytes), hipMemcpyDeviceToDevice,
                            static_cast<hipStream_t>(stream)));
  }
  return SOIL_OK;
}
int self_all_reduce(void*, float*, int64_t, void*) { return SOIL_OK; }
int self_barrier(void*) { return SOIL_OK; }

}  // namespace

}  // namespace soil

using namespace soil;

extern "C" {

void soil_slab_layout(int32_t rank, int32_t world, int64_t S, int64_t G, int64_t out[4]) {
  const Layout l = slab_layout(rank, world, S, G);
  out[0] = l.x0, out[1] = l.rows, out[2] = l.r0, out[3] = l.r1;
}

int soil_slab_ops_hip_create(soil_slab_ops** out) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out, "slab_ops_hip_create: null argument");
  HipOps* o = new HipOps;
  if (const char* e = std::getenv("SOIL_SLAB_UNIFORM")) o->uniform = e[0] != '0';
  if (const char* e = std::getenv("SOIL_SLAB_LAZY")) o->lazy = e[0] != '0';
  SOIL_HIP(hipGetDevice(&o->device));
  SOIL_HIP(hipStreamCreateWithFlags(&o->main, hipStreamNonBlocking));
  SOIL_HIP(hipStreamCreateWithFlags(&o->comm, hipStreamNonBlocking));
  SOIL_HIP(hipEventCreateWithFlags(&o->ev_fork, hipEventDisableTiming));
  SOIL_HIP(hipEventCreateWithFlags(&o->ev_join, hipEventDisableTiming));
  SOIL_HIP(hipMalloc(reinterpret_cast<void**>(&o->depth_dev), 8));
  soil_slab_ops* t = new soil_slab_ops{};
  t->ctx = o;
  t->alloc = hip_alloc, t->release = hip_release, t->fill_f32 = hip_fill, t->add_f32 = hip_add;
  t->rng_seed = hip_seed, t->particles_fluvial = hip_fluvial, t->particles_debris = hip_debris;
  t->particles_pair = hip_pair, t->cells = hip_cells, t->ghost_extent = hip_extent;
  t->noise_rows = hip_noise, t->layers_from_bedrock = hip_layers, t->to_host = hip_to_host;
  t->from_host = hip_from_host, t->fork = hip_fork, t->join = hip_join, t->sync = hip_sync;
  t->stream = hip_stream;
  t->particles_pass = hip_pass;
  *out = t;
  return SOIL_OK;
}

int soil_slab_ops_hip_destroy(soil_slab_ops* ops) {
  if (!ops) return SOIL_OK;
  HipOps* o = static_cast<HipOps*>(ops->ctx);
  if (o) {
    (void)hipStreamSynchronize(o->main);
    (void)hipStreamSynchronize(o->comm);
    (void)hipFree(o->depth_dev);
    (void)hipEventDestroy(o->ev_fork);
    (void)hipEventDestroy(o->ev_join);
    (void)hipStreamDestroy(o->main);
    (void)hipStreamDestroy(o->comm);
    delete o;
  }
  delete ops;
  return SOIL_OK;
}

int soil_slab_create(soil_slab** out, const soil_slab_config* cfg, const soil_param* param,
                     const soil_comm* comm, const soil_slab_ops* ops) {
  SOIL_REQUIRE(out && cfg && param && comm, "slab_create: null argument");
  SOIL_REQUIRE(cfg->rows_per_rank > 0 && cfg->W > 0 && cfg->particles_div > 0, "slab_create: empty slab");
  SOIL_REQUIRE(comm->world >= 1 && comm->rank >= 0 && comm->rank < comm->world && comm->exchange &&
                   comm->all_reduce_sum_f32 && comm->barrier,
               "slab_create: incomplete communicator");
  soil_slab* s = new soil_slab;
  auto bail = [&](int rc) {
    soil_slab_destroy(s);
    return rc;
  };
  if (!ops) {
    if (int rc = soil_slab_ops_hip_create(&s->own_ops); rc != SOIL_OK) {
      delete s;
      return rc;
    }
    ops = s->own_ops;
  }
  s->comm = comm, s->ops = ops, s->param = *param;
  s->rank = comm->rank, s->world = comm->world;
  s->host_ordered = (comm->flags & SOIL_COMM_HOST_ORDERED) != 0;
  s->S = cfg->rows_per_rank, s->W = cfg->W, s->H = s->world * s->S;
  s->G = soil_ghost_rows(param);
  s->mode = cfg->mode >= 0 ? cfg->mode : SOIL_SLAB_DEEP_HALO;
  if (cfg->mode < 0)
    if (const char* e = std::getenv("SOIL_SLAB_MODE")) s->mode = (e[0] == 'm' || e[0] == '1') ? SOIL_SLAB_MIGRATE : SOIL_SLAB_DEEP_HALO;
  if (s->mode != SOIL_SLAB_DEEP_HALO && s->mode != SOIL_SLAB_MIGRATE)
    return bail(fail(SOIL_ERR_INVALID_ARGUMENT, "slab_create: mode must be SOIL_SLAB_DEEP_HALO or SOIL_SLAB_MIGRATE"));
  if (s->mode == SOIL_SLAB_MIGRATE) {
    if (!ops->particles_pass)
      return bail(fail(SOIL_ERR_INVALID_ARGUMENT, "slab_create: this back-end cannot hand walkers over (no particles_pass): SOIL_SLAB_MIGRATE refused"));
    // A shallow halo instead of none: with one ghost row a walker that zig-zags along the slab's edge is
    // handed back and forth, a pass of all ranks per crossing (the first version of this mode: 48-step
    // walks on 64-row slabs were still crossing after four passes).  On kMigrateHalo ghost rows it walks
    // on as the deep-halo walkers do — its deposits there go home with the flux halo, that many rows —
    // and is handed over only at their far end, well inside the neighbour's rows: coming back takes
    // another 2 x that many steps.
    // How deep: immigrants' launches per step on an interior rank of a 4-way split of 16384^2 (fast arithmetic,
    // SOIL_SLAB_VERBOSE) — 16 rows: 230 k fluvial + 143 k debris walkers, 3.35 + 0.2 + 2.46 + 0.26 ms in two
    // passes per kind; 64 rows: 140 k + 62 k, 1.44 + 1.31 ms, one pass; 128 rows: 54 k + 8 k, 0.48 + 0.55 ms —
    // against 15 / 59 / 118 MB of flux and field rows per neighbour on the wire.  64 is where the two meet.
    int64_t halo = 64;
    if (const char* e = std::getenv("SOIL_MIGRATE_HALO")) halo = std::max(1, std::atoi(e));
    s->G = std::min<int64_t>(std::min<int64_t>(halo, s->S), soil_ghost_rows(param));
  }
  if (s->world > 1 && s->G > s->S)
    return bail(fail(SOIL_ERR_INVALID_ARGUMENT, "ghost depth " + std::to_string(s->G) + " exceeds the " +
                                                    std::to_string(s->S) + " rows a neighbour owns"));
  s->lay = slab_layout(s->rank, s->world, s->S, s->G);
  s->N = s->H * s->W / cfg->particles_div;
  s->seed = cfg->seed;
  const bool given = cfg->scale[0] != 0.0f || cfg->scale[1] != 0.0f || cfg->scale[2] != 0.0f;
  s->scale[0] = given ? cfg->scale[0] : 20.0f / static_cast<float>(s->H);
  s->scale[1] = given ? cfg->scale[1] : 20.0f / static_cast<float>(s->W);
  s->scale[2] = given ? cfg->scale[2] : 4.0f;
  auto env_is = [](const char* n, const char* v) {
    const char* e = std::getenv(n);
    return e && std::strcmp(e, v) == 0;
  };
  s->trim = cfg->trim >= 0 ? cfg->trim != 0 : (s->world > 1 && !env_is("SOIL_HALO_FULL", "1"));
  if (s->world == 1 || !ops->ghost_extent || s->mode == SOIL_SLAB_MIGRATE) s->trim = false;
  s->pair = cfg->pair >= 0 ? cfg->pair != 0 : !env_is("SOIL_STEP_PAIR", "0");  // on by default, as in soil_erode_step
  s->halo_need = cfg->halo_need;
  s->window = !env_is("SOIL_HALO_WINDOW", "0");
  if (s->halo_need <= 0)
    if (const char* e = std::getenv("SOIL_HALO_NEED")) s->halo_need = std::atoi(e);
  s->up = s->rank > 0 ? s->rank - 1 : -1;
  s->down = s->rank < s->world - 1 ? s->rank + 1 : -1;
  s->gu = s->lay.r0, s->gd = s->lay.rows - s->lay.r1;
  s->fresh_everything();
  for (int p = 0; p < kPlanes; ++p) {
    void* q = nullptr;
    if (int rc = ops->alloc(ops->ctx, &q, s->lay.rows * s->row_floats(p) * 4); rc != SOIL_OK) return bail(rc);
    s->P[p] = static_cast<float*>(q);
  }
  for (const Plane* list : {kFluxFluvial, kFluxDebris})
    for (int i = 0; i < (list == kFluxFluvial ? 3 : 2); ++i) {
      const int p = list[i];
      for (int side = 0; side < 2; ++side) {
        const int peer = side ? s->down : s->up;
        if (peer < 0) continue;
        void* q = nullptr;
        if (int rc = ops->alloc(ops->ctx, &q, s->peer_ghost(peer) * s->row_floats(p) * 4); rc != SOIL_OK) return bail(rc);
        s->stage[p][side] = static_cast<float*>(q);
      }
    }
  void* q = nullptr;
  if (int rc = ops->alloc(ops->ctx, &q, s->N * static_cast<int64_t>(sizeof(soil_rng))); rc != SOIL_OK) return bail(rc);
  s->rng = static_cast<soil_rng*>(q);
  if (ops->particles_pair) {
    if (int rc = ops->alloc(ops->ctx, &q, s->N * static_cast<int64_t>(sizeof(soil_rng))); rc != SOIL_OK) return bail(rc);
    s->rng_debris = static_cast<soil_rng*>(q);
  }
  if (int rc = ops->alloc(ops->ctx, &q, 8 * 4); rc != SOIL_OK) return bail(rc);
  s->remote0 = static_cast<float*>(q);
  // (world * 4 small ints, and behind them the 8 sums of the NaN walkers' deposits: one all-reduce carries both)
  if (int rc = ops->alloc(ops->ctx, &q, (static_cast<int64_t>(s->world) * 4 + 8) * 4); rc != SOIL_OK) return bail(rc);
  s->ints = static_cast<float*>(q);
  if (s->mode == SOIL_SLAB_MIGRATE) {
    // (the counts travel as floats in the all-reduce: exact below 2^24)
    s->box_cap = std::min<int64_t>(std::max<int64_t>(s->N, 1), (1 << 24) - 1);
    for (int side = 0; side < 2; ++side) {
      if (int rc = ops->alloc(ops->ctx, &q, s->box_cap * soil_slab::kRecBytes); rc != SOIL_OK) return bail(rc);
      s->out_box[side] = q;
    }
    if (int rc = ops->alloc(ops->ctx, &q, 2 * s->box_cap * soil_slab::kRecBytes); rc != SOIL_OK) return bail(rc);
    s->inbox = q;
    if (int rc = ops->alloc(ops->ctx, &q, 16); rc != SOIL_OK) return bail(rc);
    s->out_count = static_cast<uint32_t*>(q);
  }
  // The neighbours' refresh depths are computed, not exchanged (round 4): that is only right when every
  // rank runs with the same forced depth (cfg->halo_need / SOIL_HALO_NEED: tests).  One sum at create time.
  if (s->world > 1) {
    std::vector<int> all;
    const int mine[1] = {s->halo_need > 0 ? s->halo_need : 0};
    if (int rc = s->all_ints(mine, 1, all); rc != SOIL_OK) return bail(rc);
    for (int v : all)
      if (v != mine[0])
        return bail(fail(SOIL_ERR_INVALID_ARGUMENT, "slab_create: halo_need differs between the ranks (" +
                                                        std::to_string(mine[0]) + " here, " + std::to_string(v) + " elsewhere)"));
  }
  if (cfg->init) {
    if (int rc = ops->alloc(ops->ctx, &q, s->lay.rows * s->W * 4); rc != SOIL_OK) return bail(rc);
    float* bed = static_cast<float*>(q);
    soil_noise_param np;
    soil_noise_param_default(&np);
    np.seed = cfg->noise_seed;
    np.ext[0] = static_cast<float>(cfg->noise_rows > 0 ? cfg->noise_rows : s->H);
    np.ext[1] = static_cast<float>(s->W);
    int rc = ops->noise_rows(ops->ctx, bed, s->lay.rows, s->W, s->lay.x0, &np);
    if (rc == SOIL_OK) rc = ops->layers_from_bedrock(ops->ctx, s->P[kLayers], bed, s->lay.rows * s->W);
    if (rc == SOIL_OK) rc = ops->fill_f32(ops->ctx, s->P[kRainfall], 1.0f, s->lay.rows * s->W, 0);
    if (rc == SOIL_OK) rc = ops->sync(ops->ctx);
    ops->release(ops->ctx, bed);
    if (rc != SOIL_OK) return bail(rc);
  }
  // The back-end clears every block it hands out on its own lane (a non-blocking stream): nothing the
  // caller does next — an upload through soil_slab_plane on the null or any stream of its own — is
  // ordered against those fills unless they are done when this returns (with init == 0 nothing above
  // waited for them: a late memset wiped uploaded layers, depending on timing and grid size).
  if (int rc = ops->sync(ops->ctx); rc != SOIL_OK) return bail(rc);
  *out = s;
  return SOIL_OK;
}

int soil_slab_step(soil_slab* slab, soil_slab_mark_fn mark, void* mark_ctx) {
  SOIL_REQUIRE(slab, "slab_step: null runner");
  if (int rc = slab->step(mark, mark_ctx); rc != SOIL_OK) return rc;
  return slab->wire_status();
}

int soil_slab_plane(soil_slab* slab, const char* name, float** data, int64_t* rows, int64_t* channels) {
  SOIL_REQUIRE(slab && name && data, "slab_plane: null argument");
  for (int p = 0; p < kPlanes; ++p)
    if (std::strcmp(name, kPlaneName[p]) == 0) {
      *data = slab->P[p];
      if (rows) *rows = slab->lay.rows;
      if (channels) *channels = kPlaneCh[p];
      return SOIL_OK;
    }
  return fail(SOIL_ERR_INVALID_ARGUMENT, std::string("slab_plane: no plane called ") + name);
}

int soil_slab_get_info(const soil_slab* s, soil_slab_info* info) {
  SOIL_REQUIRE(s && info, "slab_get_info: null argument");
  *info = soil_slab_info{};
  info->H = s->H, info->W = s->W, info->S = s->S, info->G = s->G;
  info->x0 = s->lay.x0, info->rows = s->lay.rows, info->r0 = s->lay.r0, info->r1 = s->lay.r1;
  info->N = s->N, info->step_index = s->step_index;
  info->rank = s->rank, info->world = s->world, info->trim = s->trim, info->pair = s->pair;
  info->rows_flux = s->rows_flux, info->rows_field = s->rows_field, info->rows_full = s->rows_full;
  info->repeated_launches = s->fallbacks;
  info->rows_window = s->rows_window, info->rows_window_full = s->rows_window_full;
  info->passes = s->passes, info->walkers_handed = s->walkers_handed;
  info->mode = s->mode, info->reserved = 0;
  info->n_reach = static_cast<int32_t>(s->reach_hist.size());
  for (int i = 0; i < info->n_reach && i < 4; ++i) info->reach_hist[i] = s->reach_hist[static_cast<size_t>(i)];
  return SOIL_OK;
}

int soil_slab_sync(soil_slab* slab) {
  SOIL_REQUIRE(slab, "slab_sync: null runner");
  if (int rc = slab->ops->sync(slab->ops->ctx); rc != SOIL_OK) return rc;
  return slab->wire_status();
}

int soil_slab_stream(soil_slab* slab, int32_t lane, void** stream) {
  SOIL_REQUIRE(slab && stream, "slab_stream: null argument");
  *stream = slab->stream(lane);
  return SOIL_OK;
}

int soil_slab_destroy(soil_slab* s) {
  if (!s) return SOIL_OK;
  if (s->ops) {
    if (s->ops->sync) (void)s->ops->sync(s->ops->ctx);
    auto drop = [&](void* p) {
      if (p) (void)s->ops->release(s->ops->ctx, p);
    };
    for (int p = 0; p < kPlanes; ++p) {
      drop(s->P[p]);
      drop(s->stage[p][0]);
      drop(s->stage[p][1]);
    }
    drop(s->rng), drop(s->rng_debris), drop(s->remote0), drop(s->ints);
    drop(s->out_box[0]), drop(s->out_box[1]), drop(s->inbox), drop(s->out_count);
  }
  if (s->own_ops) soil_slab_ops_hip_destroy(s->own_ops);
  delete s;
  return SOIL_OK;
}

// ---- communicators --------------------------------------------------------------------------------

int soil_comm_rccl_unique_id(uint8_t id[128]) {
  SOIL_DEVICE();
  SOIL_REQUIRE(id, "comm_rccl_unique_id: null argument");
  if (int rc = rccl_load(); rc != SOIL_OK) return rc;
  SOIL_RCCL(g_rccl.GetUniqueId(id));
  return SOIL_OK;
}

int soil_comm_rccl_probe(int32_t* version) {
  // No SOIL_DEVICE(): binding the library and asking for its version touches no device and starts
  // nothing (ncclGetUniqueId, which the non-root ranks used to call as their probe, starts a
  // bootstrap thread and a listening socket per call — advisor finding of round 4).
  if (int rc = rccl_load(); rc != SOIL_OK) return rc;
  int v = 0;
  SOIL_RCCL(g_rccl.GetVersion(&v));
  if (version) *version = v;
  return SOIL_OK;
}

int soil_comm_rccl_library(char* path, int32_t capacity, int32_t* version) {
  if (int rc = rccl_load(); rc != SOIL_OK) return rc;
  if (path && capacity > 0) {
    std::strncpy(path, g_rccl.path.c_str(), static_cast<size_t>(capacity) - 1);
    path[capacity - 1] = 0;
  }
  if (version) *version = g_rccl.version;
  return SOIL_OK;
}

int soil_comm_rccl_create(soil_comm** out, const uint8_t id[128], int32_t rank, int32_t world) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && id && world >= 1 && rank >= 0 && rank < world, "comm_rccl_create: bad argument");
  if (int rc = rccl_load(); rc != SOIL_OK) return rc;
  int device = 0;
  SOIL_HIP(hipGetDevice(&device));
  char who[64];
  std::snprintf(who, sizeof who, ", rank %d of %d, device %d", rank, world, device);
  // ncclCommInitRank blocks until every rank of the world has called it; a rank that never shows up
  // must not hold this one for ever.  There is no communicator to abort yet, so the call runs on a
  // helper thread that is left behind if it does not return in time (SOIL_RCCL_INIT_TIMEOUT_S, 120 s).
  struct Init {
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    int e = 0;
    void* comm = nullptr;
  };
  auto init = std::make_shared<Init>();
  Id128 uid;
  std::memcpy(uid.bytes, id, 128);
  std::thread([init, uid, rank, world, device] {
    void* comm = nullptr;
    int e = hipSetDevice(device) == hipSuccess ? g_rccl.CommInitRank(&comm, world, uid, rank) : 1 /* ncclUnhandledCudaError */;
    std::lock_guard<std::mutex> l(init->m);
    init->done = true, init->e = e, init->comm = comm;
    init->cv.notify_all();
  }).detach();
  const double init_timeout = env_seconds("SOIL_RCCL_INIT_TIMEOUT_S", 120.0);
  {
    std::unique_lock<std::mutex> l(init->m);
    if (!init->cv.wait_for(l, std::chrono::duration<double>(init_timeout), [&] { return init->done; })) {
      char sec[32];
      std::snprintf(sec, sizeof sec, "%.1f s", init_timeout);
      return fail(SOIL_ERR_COMM, std::string("no return from ncclCommInitRank within ") + sec +
                                     " (SOIL_RCCL_INIT_TIMEOUT_S; is every rank of the world up?) on " + rccl_where() + who);
    }
  }
  if (init->e != 0) return rccl_fail(init->e, "ncclCommInitRank");
  RcclCtx* r = new RcclCtx;
  r->comm = init->comm, r->device = device, r->rank = rank, r->world = world;
  if (hipMalloc(reinterpret_cast<void**>(&r->scratch), 4) != hipSuccess || hipMemset(r->scratch, 0, 4) != hipSuccess) {
    (void)hipGetLastError();
    (void)g_rccl.CommDestroy(r->comm);
    delete r;
    return fail(SOIL_ERR_OUT_OF_MEMORY, "comm_rccl_create: no device memory for the barrier word");
  }
  WireWatch::Hooks h;
  h.where = rccl_where() + who;
  h.thread_start = [device] { (void)hipSetDevice(device); };
  h.token_state = [](void* t) {
    const hipError_t q = hipEventQuery(static_cast<hipEvent_t>(t));
    if (q == hipSuccess) return 1;
    (void)hipGetLastError();
    return q == hipErrorNotReady ? 0 : -1;
  };
  h.token_release = [r](void* t) { r->give_event(static_cast<hipEvent_t>(t)); };
  h.async_error = [r]() -> std::string {
    int e = 0;
    if (!r->comm || g_rccl.CommGetAsyncError(r->comm, &e) != 0 || e == 0 || e == 7 /* ncclInProgress */) return "";
    return std::string(g_rccl.GetErrorString(e)) + " (ncclCommGetAsyncError)";
  };
  h.abort = [r] {
    void* comm = r->comm;
    r->comm = nullptr;
    if (comm) (void)g_rccl.CommAbort(comm);
  };
  r->watch.reset(new WireWatch(env_seconds("SOIL_RCCL_TIMEOUT_S", 30.0), std::move(h)));
  soil_comm* c = new soil_comm{};
  c->ctx = r, c->rank = rank, c->world = world, c->flags = 0;
  c->exchange = rccl_exchange, c->all_reduce_sum_f32 = rccl_all_reduce, c->barrier = rccl_barrier;
  c->status = rccl_status;
  *out = c;
  return SOIL_OK;
}

int soil_comm_rccl_info(const soil_comm* comm, int32_t* count, int32_t* rank, int32_t* device) {
  SOIL_REQUIRE(comm && comm->ctx && g_rccl.lib, "comm_rccl_info: not an RCCL communicator");
  RcclCtx& r = *static_cast<RcclCtx*>(comm->ctx);
  if (r.watch->dead()) return r.dead_rc();
  int v = 0;
  if (count) {
    SOIL_RCCL(g_rccl.CommCount(r.comm, &v));
    *count = v;
  }
  if (rank) {
    SOIL_RCCL(g_rccl.CommUserRank(r.comm, &v));
    *rank = v;
  }
  if (device) {
    SOIL_RCCL(g_rccl.CommCuDevice(r.comm, &v));
    *device = v;
  }
  return SOIL_OK;
}

int soil_comm_rccl_destroy(soil_comm* comm) {
  if (!comm) return SOIL_OK;
  RcclCtx* r = static_cast<RcclCtx*>(comm->ctx);
  if (r) {
    r->watch->drain();  // bounded by the watchdog: what is still in flight completes or is aborted
    const bool dead = r->watch->dead();
    r->watch.reset();
    if (!dead) (void)hipDeviceSynchronize();
    if (!dead && r->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(r->comm);
    (void)hipFree(r->scratch);
    for (hipEvent_t e : r->events) (void)hipEventDestroy(e);
    delete r;
  }
  delete comm;
  return SOIL_OK;
}

int soil_comm_wedged_create(soil_comm** out, int32_t rank, int32_t world, double timeout_s) {
  SOIL_REQUIRE(out && world >= 1 && rank >= 0 && rank < world && timeout_s > 0, "comm_wedged_create: bad argument");
  WedgedCtx* w = new WedgedCtx;
  WireWatch::Hooks h;
  h.where = "the wedged test wire (soil_comm_wedged_create)";
  h.token_state = [](void*) { return 0; };
  h.token_release = [](void*) {};
  h.abort = [w] {
    std::lock_guard<std::mutex> l(w->m);
    w->aborted = true;
    w->cv.notify_all();
  };
  w->watch.reset(new WireWatch(timeout_s, std::move(h)));
  soil_comm* c = new soil_comm{};
  c->ctx = w, c->rank = rank, c->world = world, c->flags = SOIL_COMM_HOST_ORDERED;
  c->exchange = wedged_exchange, c->all_reduce_sum_f32 = wedged_all_reduce, c->barrier = wedged_barrier;
  c->status = wedged_status;
  *out = c;
  return SOIL_OK;
}

int soil_comm_wedged_destroy(soil_comm* comm) {
  if (!comm) return SOIL_OK;
  if (WedgedCtx* w = static_cast<WedgedCtx*>(comm->ctx)) {
    w->watch.reset();
    delete w;
  }
  delete comm;
  return SOIL_OK;
}

int soil_comm_self_create(soil_comm** out) {
  SOIL_REQUIRE(out, "comm_self_create: null argument");
  soil_comm* c = new soil_comm{};
  c->rank = 0, c->world = 1;
  c->exchange = self_exchange, c->all_reduce_sum_f32 = self_all_reduce, c->barrier = self_barrier;
  *out = c;
  return SOIL_OK;
}

int soil_comm_self_destroy(soil_comm* comm) {
  delete comm;
  return SOIL_OK;
}

}  // extern "C"
