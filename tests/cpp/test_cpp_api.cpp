// Exercises include/soil.hpp (the C++ host mirror) against the C-ABI library:
// ramp -> steepest/accumulate known answers, mass_creep conservation, noise.
#include <cmath>
#include <cstdio>
#include <soil.hpp>

#include "watchdog.hpp"

using soil::F;
using silt::check;

#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
  start_watchdog(60.0);  // a hang becomes exit code 3 with the last marker, not a killed child without output
  mark("device count");
  if (soil_device_count() == 0) {  // no CPU fallback: the first allocation must throw
    try { silt::tensor_t<float> t(silt::shape(4, 4)); } catch (const std::runtime_error&) {
      std::printf("NO_DEVICE_OK\n"); return 0; }
    return 1;
  }
  mark("steepest / accumulate on a ramp");
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

  mark("noise, mass_creep");
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
  mark("three steps of soil::erode");
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
  // the sharded step through the C ABI: a world of one on the one-rank wire must walk the walks of
  // soil::erode on the same grid (the same over an RCCL communicator: tests/cpp/test_cpp_rccl.cpp)
  mark("slab runner, one-rank wire");
  {
    const int S = 96;
    soil::param_t sp;
    sp.maxage = 64; sp.timeStep = 1000.0f; sp.critSlopeBedrock = 0.57f; sp.suspensionRateFluvial = 0.0008f;
    soil::slab_runner slab(soil::slab_runner::config(S, S), sp, soil::comm::self());
    uint64_t before = 0, after = 0;
    check(soil_particle_steps(&before, 1, nullptr));
    for (int s = 0; s < 3; ++s) slab.step();
    slab.sync();
    check(soil_particle_steps(&after, 1, nullptr));
    EXPECT(slab.info().step_index == 3 && slab.info().world == 1 && slab.info().rows == S);
    EXPECT(after > 0);
    double sum = 0;  // bedrock = channel 0 of the layer plane: what soil::erode hands back as model.height
    const std::vector<float> lay = slab.owned_rows("layers");
    for (size_t i = 0; i < lay.size(); i += 2) sum += lay[i];
    std::printf("SLAB0 %llu %.9e\n", static_cast<unsigned long long>(after), sum);
  }
  std::printf("CPP_API_OK\n");
  mark(kExitMarker);
  return 0;
}
