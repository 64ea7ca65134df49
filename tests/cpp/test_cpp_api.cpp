// Exercises include/soil.hpp (the C++ host mirror) against the C-ABI library:
// ramp -> steepest/accumulate known answers, mass_creep conservation, noise.
#include <cmath>
#include <cstdio>
#include <soil.hpp>

using soil::F;
using silt::check;

#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
  if (soil_device_count() == 0) {  // no CPU fallback: the first allocation must throw
    try { silt::tensor_t<float> t(silt::shape(4, 4)); } catch (const std::runtime_error&) {
      std::printf("NO_DEVICE_OK\n"); return 0; }
    return 1;
  }
  const int H = 12, W = 9;
  std::vector<float> ramp(H * W);
  for (int x = 0; x < H; ++x) for (int y = 0; y < W; ++y) ramp[x * W + y] = float(x);
  auto h = silt::tensor_t<float>::from_host(ramp, silt::shape(H, W));
  auto g = soil::steepest(h, soil::D8);
  auto gv = g.to_host();
  for (int y = 0; y < W; ++y) EXPECT(gv[y] == -1);                 // lowest row: no receiver
  for (int x = 1; x < H; ++x) for (int y = 0; y < W; ++y) EXPECT(gv[x * W + y] == (x - 1) * W + y);
  silt::tensor_t<float> ones(silt::shape(H, W));
  silt::set(ones, 1.0f);
  auto acc = soil::accumulate(g, ones, soil::D8).to_host();
  for (int x = 0; x < H; ++x) EXPECT(acc[x * W + 3] == float(H - x));  // upstream cells incl. self
  try { soil::steepest(h, soil::edge_t(7)); return 1; } catch (const std::invalid_argument&) {}

  soil::noise_param_t np; np.seed = 3.0f; np.ext[0] = 64; np.ext[1] = 64;
  auto bed = soil::noise(silt::shape(64, 64), np).to_host();
  EXPECT(std::fabs(bed[0] - (-0.0f)) < 1.0f);
  std::vector<float> lay(64 * 64 * 2);
  for (int i = 0; i < 64 * 64; ++i) { lay[2 * i] = bed[i]; lay[2 * i + 1] = 0.05f * float((i * 7) % 11) / 11.0f; }
  auto layers = silt::tensor_t<float>::from_host(lay, silt::shape(64, 64, 2));
  silt::tensor_t<float> delta(silt::shape(64, 64, 2));
  silt::set(delta, 0.0f);
  soil::param_t p; p.critSlopeSediment = 0.01f;
  soil::mass_creep(delta, layers, silt::vec3{20.f / 64, 20.f / 64, 4.f}, p);
  auto d = delta.to_host();
  double sum = 0, mag = 0;
  for (int i = 0; i < 64 * 64; ++i) { EXPECT(d[2 * i] == 0.0f); sum += d[2 * i + 1]; mag += std::fabs(d[2 * i + 1]); }
  EXPECT(mag > 0 && std::fabs(sum) < 1e-6 * mag + 1e-9);               // creep conserves sediment

  // three whole steps through soil::erode (the library's step driver): the Python test runs the
  // same three steps through its own binding and compares the sums printed here
  {
    const int S = 96;
    const silt::shape sh(S, S), sh2(S, S, 2);
    soil::noise_param_t q; q.seed = 3.0f; q.ext[0] = S; q.ext[1] = S;
    soil::map_t model(sh, silt::vec3{20.f / S, 20.f / S, 4.f});
    model.height = soil::noise(sh, q);
    for (F* t : {&model.sediment, &model.uplift, &model.rainfall}) *t = F(sh);
    silt::set(model.sediment, 0.0f); silt::set(model.uplift, 0.0f); silt::set(model.rainfall, 1.0f);
    soil::data_t data(sh), track(sh);
    for (soil::data_t* d : {&data, &track}) {
      d->discharge = F(sh); d->mass = F(sh); d->debris = F(sh); d->momentum = F(sh2); d->debris_momentum = F(sh2);
      for (F* t : {&d->discharge, &d->mass, &d->debris, &d->momentum, &d->debris_momentum}) silt::set(*t, 0.0f);
    }
    soil::erode_param_t ep;
    ep.samples = S * S / 8; ep.maxage = 64; ep.timeStep = 1000.0f;
    ep.critSlopeBedrock = 0.57f; ep.suspensionRateFluvial = 0.0008f;
    soil::erode(model, data, track, ep, 2);
    soil::erode(model, data, track, ep);          // a third step, numbered 2
    EXPECT(model.steps_taken == 3);
    uint64_t steps = 0;
    check(soil_particle_steps(&steps, 0, nullptr));
    double sh_ = 0, sd = 0, st = 0;
    for (float v : model.height.to_host()) sh_ += v;
    for (float v : data.discharge.to_host()) if (v == v) sd += v;
    for (float v : track.discharge.to_host()) st += std::fabs(v);
    EXPECT(st == 0.0);                             // the flux planes are left zeroed
    std::printf("ERODE3 %llu %.9e %.9e\n", static_cast<unsigned long long>(steps), sh_, sd);
  }
  // the sharded step through the C ABI: a world of one (one-rank wire, then a one-rank RCCL
  // communicator made by the library) must walk the walks of soil::erode on the same grid
  {
    const int S = 96;
    soil::param_t sp;
    sp.maxage = 64; sp.timeStep = 1000.0f; sp.critSlopeBedrock = 0.57f; sp.suspensionRateFluvial = 0.0008f;
    double sums[2] = {0, 0};
    for (int which = 0; which < 2; ++which) {
      soil::comm wire = which == 0 ? soil::comm::self() : soil::comm::rccl(soil::comm::rccl_unique_id(), 0, 1);
      soil::slab_runner slab(soil::slab_runner::config(S, S), sp, wire);
      uint64_t before = 0, after = 0;
      check(soil_particle_steps(&before, 1, nullptr));
      for (int s = 0; s < 3; ++s) slab.step();
      slab.sync();
      check(soil_particle_steps(&after, 1, nullptr));
      EXPECT(slab.info().step_index == 3 && slab.info().world == 1 && slab.info().rows == S);
      EXPECT(after > 0);
      {  // bedrock = channel 0 of the layer plane: what soil::erode hands back as model.height
        const std::vector<float> lay = slab.owned_rows("layers");
        for (size_t i = 0; i < lay.size(); i += 2) sums[which] += lay[i];
      }
      std::printf("SLAB%d %llu %.9e\n", which, static_cast<unsigned long long>(after), sums[which]);
    }
    EXPECT(std::fabs(sums[0] - sums[1]) <= 1e-6 * std::fabs(sums[0]));
  }
  // The wire itself on the real library: a one-rank RCCL communicator exchanging with ITSELF
  // (ncclSend / ncclRecv to one's own rank inside a group are legal), so that the grouped
  // point-to-point path of csrc/slab_runner.hip (rccl_exchange) executes on RCCL on a one-GPU box:
  // one pair at BASELINE config 5's fluvial flux halo (250 rows x 16384 cells x 16 B = 65.5 MB), then
  // the grouped four-transfer pattern of a step's field exchange (two sends + two receives per side).
  {
    soil::comm wire = soil::comm::rccl(soil::comm::rccl_unique_id(), 0, 1);
    int32_t n = 0, r = -1, dev = -1;
    check(soil_comm_rccl_info(wire.get(), &n, &r, &dev));
    EXPECT(n == 1 && r == 0 && dev >= 0);
    const size_t big = size_t(250) * 16384 * 16, words = big / 4;
    float *src = nullptr, *dst = nullptr;
    check(soil_malloc(reinterpret_cast<void**>(&src), big));
    check(soil_malloc(reinterpret_cast<void**>(&dst), big));
    std::vector<float> pat(words);
    for (size_t i = 0; i < words; ++i) pat[i] = float(i % 8191) - 4000.0f;
    check(soil_memcpy_h2d(src, pat.data(), big, nullptr));
    check(soil_set_f32(dst, -1.0f, int64_t(words), nullptr));
    check(soil_device_synchronize());
    void *e0 = nullptr, *e1 = nullptr;
    check(soil_event_create(&e0));
    check(soil_event_create(&e1));
    float ms_first = 0, ms = 0;
    for (int rep = 0; rep < 4; ++rep) {  // the first call sets the channels up
      check(soil_event_record(e0, nullptr));
      wire.exchange({soil_xfer{src, int64_t(big), 0}}, {soil_xfer{dst, int64_t(big), 0}});
      check(soil_event_record(e1, nullptr));
      check(soil_event_elapsed_ms(e0, e1, rep == 0 ? &ms_first : &ms));
    }
    std::vector<float> back(words);
    check(soil_memcpy_d2h(back.data(), dst, big, nullptr));
    size_t bad = 0;
    for (size_t i = 0; i < words; ++i) bad += back[i] != pat[i];
    EXPECT(bad == 0);
    // four transfers in one group, unequal sizes, sources and destinations interleaved in one block
    const size_t q = words / 8;
    check(soil_set_f32(dst, -1.0f, int64_t(words), nullptr));
    check(soil_device_synchronize());
    std::vector<soil_xfer> sends, recvs;
    const size_t len[4] = {q, q / 2, 3 * q / 4, 1024};
    size_t so = 0, ro = 0;
    for (int k = 0; k < 4; ++k) {
      sends.push_back(soil_xfer{src + so, int64_t(len[k] * 4), 0});
      recvs.push_back(soil_xfer{dst + ro, int64_t(len[k] * 4), 0});
      so += len[k] + 64, ro += len[k] + 256;
    }
    wire.exchange(sends, recvs);
    check(soil_memcpy_d2h(back.data(), dst, big, nullptr));
    so = 0, ro = 0;
    for (int k = 0; k < 4; ++k) {  // the k-th receive holds the k-th send (matched in order), the gaps are untouched
      for (size_t i = 0; i < len[k]; ++i) bad += back[ro + i] != pat[so + i];
      for (size_t i = 0; i < 256; ++i) bad += back[ro + len[k] + i] != -1.0f;
      so += len[k] + 64, ro += len[k] + 256;
    }
    EXPECT(bad == 0);
    float one[4] = {1.5f, -2.0f, 0.25f, 8.0f};
    check(soil_memcpy_h2d(dst, one, 16, nullptr));
    wire.all_reduce_sum(dst, 4);
    check(soil_memcpy_d2h(back.data(), dst, 16, nullptr));
    EXPECT(back[0] == 1.5f && back[1] == -2.0f && back[2] == 0.25f && back[3] == 8.0f);  // a world of one sums to itself
    std::printf("RCCL_SELF ranks %d bytes %zu first_ms %.3f ms %.3f GBps %.1f\n", n, big, ms_first, ms,
                double(big) / (double(ms) * 1e6));
    check(soil_event_destroy(e0));
    check(soil_event_destroy(e1));
    check(soil_free(src));
    check(soil_free(dst));
  }
  std::printf("CPP_API_OK\n");
  return 0;
}
