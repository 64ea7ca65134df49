// Exercises include/soil.hpp (the C++ host mirror) against the C-ABI library:
// ramp -> steepest/accumulate known answers, mass_creep conservation, noise.
#include <cmath>
#include <cstdio>
#include <soil.hpp>

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
  std::printf("CPP_API_OK\n");
  return 0;
}
