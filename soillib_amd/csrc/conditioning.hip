// conditioning.hip — depression filling of a DEM before flow routing.
//
// The reference has no such code: its example conditions DEMs with the
// third-party pysheds (example/dem_condition.py:35-41, grid.fill_pits /
// fill_depressions), yet BASELINE config 3 asks for a pit-filled
// 4096^2 DEM.  Build-defined, parity unpinned by the reference; the CPU oracle
// is Barnes' priority-flood (oracle/soil_oracle.c: orc_fill_depressions).
//
// Definition: w(c) = min over all paths from c to an outlet of the highest z on
// the path, where an outlet is a step off the grid or onto a NaN (NoData) cell.
// w is the least surface >= z without depressions; it is the greatest fixed
// point below the start of
//        w(c) <- max(z(c), min over the neighbours n of w(n))
// started from +inf (an outlet's neighbour comes down to z in the first step).  Only max and min
// are involved, so the fixed point is reached exactly in fp32 whatever the update order: the
// kernel settles 64x64 tiles in LDS (chaotic inside a tile, Jacobi across tiles per launch) and
// the host repeats launches until no tile changed.
//
// Started from +inf, information has to travel from the grid's edge to its middle, one tile per
// launch (39 launches at 4096^2).  The iteration reaches the same fixed point from ANY surface
// that lies on or above it (it only ever lowers cells, never below w; and w is the only fixed
// point: walk the cells in the order of their w).  So the start is taken from the same problem on
// a 4x coarser grid — each coarse cell the maximum of its 4x4 block, filled recursively: a fine
// cell can always follow the coarse path through its block's neighbours without meeting anything
// higher than the blocks' maxima — and what is left is to bring every cell down from its block's
// level and to re-level the lakes from their true spill points: 16 launches at 4096^2 (and 4-10 on
// each of the three small levels); 16x coarsening leaves 22, 128x128 tiles cost more per launch
// than they save.
//
// Inside a tile a step of the plain iteration moves information by one cell: a work-group spent
// 100-300 us per launch whatever the number of tiles, 4.0 ms in all at 4096^2 (round 2).  Round 3:
// whole lines at once.  Along a row (or column), going one way, cell j takes
//        w'_j = max(z_j, min(w_j, w'_{j-1}))  =  clamp of its updated predecessor into [z_j, w_j],
// and clamps compose to clamps — the chain is a prefix scan over (lo, hi) pairs, six DPP steps for
// 64 cells (fill_line_pass).  Rows there and back, columns there and back, then one plain step
// over all K neighbours (the diagonals; and what makes "nothing moved" the fixed point of the full
// operator): a tile settles in 2-9 such rounds instead of up to 256 steps.  1024 threads per
// tile (16 waves x 4 lines per direction); every tile writes its own "moved" mark (no clearing
// pass between launches); the first launch of a level makes its start surface itself (no pass
// over the level for it): 4.0 -> 1.7 ms at 4096^2, 8.1 -> 5.6 ms at 8192^2.
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <map>
#include <utility>
#include <vector>

#include "common.hpp"

namespace soil {

constexpr int kFC = 4;              // coarsening factor per level
constexpr int kFT = 64;             // tile edge
constexpr int kFH = kFT + 2;        // with its one-cell apron
constexpr int kFBlock = 256;
constexpr int kFRelax = 1024;  // threads of a relaxing work-group: 16 waves, 4 lines each per direction
constexpr int kFCells = kFT * kFT / kFRelax;

// One directional pass along a line of 64 cells (a row or a column of the tile), all 64 at once.
// Going along the line, cell j takes  w'_j = max(z_j, min(w_j, w'_{j-1}))  — the clamp of its
// updated predecessor into [z_j, w_j].  Clamps compose to clamps,
//     (clamp into [l1,h1], then into [l2,h2]) = clamp into [clamp(l1; l2,h2), clamp(h1; l2,h2)],
// so the whole chain is an inclusive prefix scan over (lo, hi) pairs: six cross-lane steps instead
// of 64 dependent ones, med3 only (exact).  The steps are DPP moves (row_shr 1/2/4/8, row_bcast
// 15/31): through ds_bpermute shuffles a pass was a chain of 24 LDS round trips and the kernel
// slower than the cell-by-cell relaxation it replaces.  A lane without a source takes the identity
// (-inf, +inf), which leaves its pair as it is.  `x_in`: the value in front of the line (apron).
// The pass runs towards higher lanes; the caller reads the line backwards for the other direction.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void fill_scan_step(float& slo, float& shi) {
  const float ninf = -__builtin_inff(), pinf = __builtin_inff();
  const float plo = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
      __builtin_bit_cast(int, ninf), __builtin_bit_cast(int, slo), CTRL, ROW_MASK, 0xf, false));
  const float phi = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
      __builtin_bit_cast(int, pinf), __builtin_bit_cast(int, shi), CTRL, ROW_MASK, 0xf, false));
  // the earlier segment (p) first, then this one (s)
  const float nlo = __builtin_amdgcn_fmed3f(plo, slo, shi), nhi = __builtin_amdgcn_fmed3f(phi, slo, shi);
  slo = nlo;
  shi = nhi;
}
__device__ __forceinline__ float fill_line_pass(float z, float w, float x_in) {
  float slo = (z != z) ? -__builtin_inff() : z;  // cells outside the grid / NoData: w = -inf, held there
  float shi = w;
  fill_scan_step<0x111, 0xf>(slo, shi);  // row_shr:1
  fill_scan_step<0x112, 0xf>(slo, shi);  // row_shr:2
  fill_scan_step<0x114, 0xf>(slo, shi);  // row_shr:4
  fill_scan_step<0x118, 0xf>(slo, shi);  // row_shr:8
  fill_scan_step<0x142, 0xa>(slo, shi);  // row_bcast:15 into rows 1 and 3
  fill_scan_step<0x143, 0xc>(slo, shi);  // row_bcast:31 into rows 2 and 3
  return __builtin_amdgcn_fmed3f(x_in, slo, shi);
}

constexpr int kFZ = kFT + 1;  // row stride of the tile's z in LDS (column reads without conflicts)

template <int K>
__global__ void __launch_bounds__(kFRelax)
    k_fill_relax(float* __restrict__ w, const float* __restrict__ z, int64_t H, int64_t W,
                 int tiles_w, int tiles_h, int inner_max, int* __restrict__ changed,
                 const unsigned char* __restrict__ dirty_prev,
                 unsigned char* __restrict__ dirty_next, int first,
                 const float* __restrict__ wc, int64_t Wc) {
  __shared__ float sw[kFH * kFH];
  __shared__ float sz[kFT * kFZ];
  __shared__ int s_flag, s_any;
  const int tid = threadIdx.x;
  // `first`: the level's first launch — every tile takes part and makes its own start, the coarse
  // level's surface of each cell's block (`wc`) or +inf on the coarsest level, never below z: a
  // pass over the whole level for that (and one to mark every tile) is saved.  (Cells next to an
  // outlet come down to z in the first relaxation step by themselves.)
  if (!first) {  // a tile can only move if it or one of its 8 neighbours moved in the previous launch
    const int tx = blockIdx.x / tiles_w, ty = blockIdx.x % tiles_w;
    bool live = false;
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy) {
        const int nx = tx + dx, ny = ty + dy;
        if (nx >= 0 && ny >= 0 && nx < tiles_h && ny < tiles_w)
          live = live || dirty_prev[nx * tiles_w + ny] != 0;
      }
    if (!live) {
      if (tid == 0) dirty_next[blockIdx.x] = 0;  // (every tile writes its mark: no clearing pass)
      return;
    }
  }
  const int64_t row0 = static_cast<int64_t>(blockIdx.x / tiles_w) * kFT;
  const int64_t col0 = static_cast<int64_t>(blockIdx.x % tiles_w) * kFT;
  const float ninf = -__builtin_inff();
  // tile + apron; off-grid and NaN cells are outlets: -inf
  for (int i = tid; i < kFH * kFH; i += kFRelax) {
    const int64_t x = row0 + i / kFH - 1, y = col0 + i % kFH - 1;
    float v = ninf;
    if (x >= 0 && y >= 0 && x < H && y < W) {
      if (first) {
        const float zv = z[x * W + y];
        v = wc ? fmaxf(zv, wc[(x / kFC) * Wc + y / kFC]) : __builtin_inff();  // (fmaxf: a NaN z gives wc)
        if (zv != zv) v = ninf;
      } else {
        v = w[x * W + y];
        if (v != v) v = ninf;
      }
    }
    sw[i] = v;
  }
  float zc[kFCells];
  bool in[kFCells];
#pragma unroll
  for (int j = 0; j < kFCells; ++j) {
    const int c = tid + j * kFRelax;
    const int64_t x = row0 + c / kFT, y = col0 + c % kFT;
    in[j] = x < H && y < W;
    zc[j] = in[j] ? z[x * W + y] : 0.0f;
    if (zc[j] != zc[j]) in[j] = false;  // NaN cells stay NaN (zc keeps it) and act as outlets
    sz[(c / kFT) * kFZ + c % kFT] = in[j] ? zc[j] : __builtin_nanf("");
  }
  if (tid == 0) s_any = 0;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int kLines = kFT / (kFRelax / 64);  // lines per wave and direction
  for (int it = 0; it < inner_max; ++it) {
    if (tid == 0) s_flag = 0;
    __syncthreads();
    bool moved = false;
    // along the rows, there and back: a lake levels across the whole tile in one pass, not a cell
    // per iteration.  (A wave owns its lines; the way back reads what the way there wrote.  Four
    // lines at a time: their scans are independent chains the wave can interleave.)
    constexpr int kTogether = 4;
    for (int q0 = 0; q0 < kLines; q0 += kTogether) {
#pragma unroll
      for (int back = 0; back < 2; ++back) {
        const int col = back ? kFT - 1 - lane : lane;
        float wv[kTogether], zv[kTogether], xin[kTogether];
#pragma unroll
        for (int u = 0; u < kTogether; ++u) {
          const int r = wave * kLines + q0 + u;
          wv[u] = sw[(r + 1) * kFH + col + 1];
          zv[u] = sz[r * kFZ + col];
          xin[u] = sw[(r + 1) * kFH + (back ? kFT + 1 : 0)];
        }
#pragma unroll
        for (int u = 0; u < kTogether; ++u) {
          if (__ballot(wv[u] > zv[u]) == 0) continue;  // no cell of the line stands above its z
          const int r = wave * kLines + q0 + u;
          const float v = fill_line_pass(zv[u], wv[u], xin[u]);
          if (v < wv[u]) {
            sw[(r + 1) * kFH + col + 1] = v;
            moved = true;
          }
        }
      }
    }
    __syncthreads();
    // along the columns
    for (int q0 = 0; q0 < kLines; q0 += kTogether) {
#pragma unroll
      for (int back = 0; back < 2; ++back) {
        const int row = back ? kFT - 1 - lane : lane;
        float wv[kTogether], zv[kTogether], xin[kTogether];
#pragma unroll
        for (int u = 0; u < kTogether; ++u) {
          const int c = wave * kLines + q0 + u;
          wv[u] = sw[(row + 1) * kFH + c + 1];
          zv[u] = sz[row * kFZ + c];
          xin[u] = sw[(back ? kFT + 1 : 0) * kFH + c + 1];
        }
#pragma unroll
        for (int u = 0; u < kTogether; ++u) {
          if (__ballot(wv[u] > zv[u]) == 0) continue;
          const int c = wave * kLines + q0 + u;
          const float v = fill_line_pass(zv[u], wv[u], xin[u]);
          if (v < wv[u]) {
            sw[(row + 1) * kFH + c + 1] = v;
            moved = true;
          }
        }
      }
    }
    __syncthreads();
    // every neighbour, one step (the diagonals of D8; and what makes "nothing moved" the fixed
    // point of the full operator)
#pragma unroll
    for (int j = 0; j < kFCells; ++j) {
      const int c = tid + j * kFRelax;
      const int p = (c / kFT + 1) * kFH + (c % kFT + 1);
      if (__ballot(in[j] && sw[p] > zc[j]) == 0) continue;  // nothing of these 64 cells can come down
      if (!in[j]) continue;
      float m = fminf(fminf(sw[p - kFH], sw[p + kFH]), fminf(sw[p - 1], sw[p + 1]));
      if (K == 8)
        m = fminf(m, fminf(fminf(sw[p - kFH - 1], sw[p - kFH + 1]),
                           fminf(sw[p + kFH - 1], sw[p + kFH + 1])));
      const float v = fmaxf(zc[j], m);
      if (v < sw[p]) {
        sw[p] = v;
        moved = true;
      }
    }
    if (moved) s_flag = 1;
    __syncthreads();
    if (s_flag == 0) break;
    if (tid == 0) s_any = 1;
    __syncthreads();
  }
  __syncthreads();
  if (s_any || first) {
#pragma unroll
    for (int j = 0; j < kFCells; ++j) {
      const int c = tid + j * kFRelax;
      const int64_t x = row0 + c / kFT, y = col0 + c % kFT;
      if (in[j]) w[x * W + y] = sw[(c / kFT + 1) * kFH + (c % kFT + 1)];
      else if (first && x < H && y < W) w[x * W + y] = zc[j];  // NaN cells stay NaN
    }
  }
  if (tid == 0) {
    if (s_any) *changed = 1;
    dirty_next[blockIdx.x] = (s_any || first) ? 1 : 0;
  }
}


// block maxima (NaN cells skipped: leaving an outlet out only raises the start, which stays valid)
__global__ void __launch_bounds__(kFBlock)
    k_fill_coarsen(float* __restrict__ zc, const float* __restrict__ z, int64_t H, int64_t W,
                   int64_t Hc, int64_t Wc) {
  const int64_t n = static_cast<int64_t>(blockIdx.x) * kFBlock + threadIdx.x;
  if (n >= Hc * Wc) return;
  const int64_t X = n / Wc, Y = n % Wc;
  float m = -__builtin_inff();
  for (int64_t x = X * kFC; x < (X + 1) * kFC && x < H; ++x)
    for (int64_t y = Y * kFC; y < (Y + 1) * kFC && y < W; ++y) {
      const float v = z[x * W + y];
      if (v == v) m = fmaxf(m, v);
    }
  zc[n] = m;
}

// one level: `out` = fill of `height` (H x W), started from the coarse surface `wc` (or +inf)
template <int K>
static int fill_level(float* out, const float* height, int64_t H, int64_t W, const float* wc,
                      int64_t Wc, unsigned char* dirty_prev, unsigned char* dirty_next, int* flag_host,
                      int* flag_dev, hipStream_t st) {
  const int tiles_w = static_cast<int>((W + kFT - 1) / kFT);
  const int tiles_h = static_cast<int>((H + kFT - 1) / kFT);
  const size_t ntiles = static_cast<size_t>(tiles_w) * tiles_h;
  // a launch moves information at least one tile further; H*W launches is a bound
  // no terrain reaches
  const int64_t max_launches = 4 * (static_cast<int64_t>(tiles_w) + tiles_h) * kFT + 16;
  // (a value below 1 or not a number would skip the launches and return the unrelaxed start)
  static const int per_check = [] {
    const char* e = std::getenv("SOIL_FILL_PER_CHECK");
    const int v = e ? std::atoi(e) : 3;
    return v >= 1 ? v : 3;
  }();
  static const bool verbose = std::getenv("SOIL_FILL_VERBOSE") != nullptr;
  for (int64_t launch = 0; launch < max_launches; launch += per_check) {
    *flag_host = 0;  // the stream is idle here: the previous launches were waited for
    // several launches per look at the flag: a launch whose predecessor moved nothing costs a
    // few microseconds (every tile returns at once), a host round trip ~20
    for (int k = 0; k < per_check; ++k) {
      k_fill_relax<K><<<static_cast<unsigned>(ntiles), kFRelax, 0, st>>>(
          out, height, H, W, tiles_w, tiles_h, 4 * kFT, flag_dev, dirty_prev, dirty_next,
          launch + k == 0 ? 1 : 0, wc, Wc);
      std::swap(dirty_prev, dirty_next);
    }
    SOIL_LAUNCH_CHECK();
    SOIL_HIP(hipStreamSynchronize(st));
    if (verbose) std::fprintf(stderr, "[fill] %lld x %lld: launches %lld..%lld moved=%d\n", (long long)H, (long long)W, (long long)launch, (long long)launch + per_check - 1, *flag_host);
    if (!__atomic_load_n(flag_host, __ATOMIC_ACQUIRE)) return SOIL_OK;
  }
  return fail(SOIL_ERR_HIP, "fill_depressions: did not converge");
}

template <int K>
static int fill_impl(float* out, const float* height, int64_t H, int64_t W, hipStream_t st) {
  // the pyramid: level 0 is the DEM itself, level l + 1 the 4x4 (kFC) block maxima of level l
  struct Level { int64_t H, W; size_t off_z, off_w; };
  std::vector<Level> lv{{H, W, 0, 0}};
  auto align = [](size_t b) { return (b + 255) & ~size_t{255}; };
  const size_t ntiles0 = static_cast<size_t>((W + kFT - 1) / kFT) * ((H + kFT - 1) / kFT);
  const size_t b_dirty = align(ntiles0);
  size_t bytes = 256 + 2 * b_dirty;
  while (lv.back().H * lv.back().W > 4 * kFT * kFT && std::getenv("SOIL_FILL_FLAT") == nullptr) {
    const int64_t Hc = (lv.back().H + kFC - 1) / kFC, Wc = (lv.back().W + kFC - 1) / kFC;
    const size_t b = align(sizeof(float) * Hc * Wc);
    lv.push_back({Hc, Wc, bytes, bytes + b});
    bytes += 2 * b;
  }
  void* base = nullptr;
  if (int rc = workspace_get(4, bytes, &base); rc != SOIL_OK) return rc;
  char* ws = static_cast<char*>(base);
  // "some tile moved in this launch": a pinned, device-mapped word the tiles write straight
  // into (a device-to-host copy is a 25-50 us blit kernel on this stack, per launch)
  // (one per host thread and device)
  static thread_local std::map<int, std::pair<int*, int*>> t_flags;
  int dev = 0;
  SOIL_HIP(hipGetDevice(&dev));
  auto& fl = t_flags[dev];
  if (!fl.first) {
    SOIL_HIP(hipHostMalloc(reinterpret_cast<void**>(&fl.first), sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    SOIL_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&fl.second), fl.first, 0));
  }
  int *const t_flag = fl.first, *const t_flag_dev = fl.second;
  unsigned char* dirty_a = reinterpret_cast<unsigned char*>(ws) + 256;
  unsigned char* dirty_b = dirty_a + b_dirty;
  auto zc = [&](size_t l) { return l == 0 ? height : reinterpret_cast<const float*>(ws + lv[l].off_z); };
  auto wl = [&](size_t l) { return l == 0 ? out : reinterpret_cast<float*>(ws + lv[l].off_w); };
  for (size_t l = 1; l < lv.size(); ++l) {
    k_fill_coarsen<<<blocks_for(lv[l].H * lv[l].W, kFBlock), kFBlock, 0, st>>>(
        reinterpret_cast<float*>(ws + lv[l].off_z), zc(l - 1), lv[l - 1].H, lv[l - 1].W, lv[l].H, lv[l].W);
    SOIL_LAUNCH_CHECK();
  }
  for (size_t l = lv.size(); l-- > 0;) {  // coarsest first
    const bool top = l + 1 == lv.size();
    if (int rc = fill_level<K>(wl(l), zc(l), lv[l].H, lv[l].W, top ? nullptr : wl(l + 1),
                               top ? 0 : lv[l + 1].W, dirty_a, dirty_b, t_flag, t_flag_dev, st);
        rc != SOIL_OK)
      return rc;
  }
  return SOIL_OK;
}

}  // namespace soil

using namespace soil;

extern "C" {

int soil_fill_depressions(float* out, const float* height, int64_t H, int64_t W, int edge,
                          void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && height && out != height, "fill_depressions: needs distinct in and out tensors");
  SOIL_REQUIRE(H > 0 && W > 0, "fill_depressions: empty grid");
  switch (edge) {
    case SOIL_D4: return fill_impl<4>(out, height, H, W, as_stream(stream));
    case SOIL_D8: return fill_impl<8>(out, height, H, W, as_stream(stream));
    default: return fail(SOIL_ERR_INVALID_ARGUMENT, "invalid edge enumerator");
  }
}

}  // extern "C"
