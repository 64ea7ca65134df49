// fnl_ref_shim.cpp — builds the REFERENCE's own noise generator for the oracle.
//
// Compiled only in the authoring container, against the reference's vendored
// header where it lies (-I/root/reference/source/soillib/external, see
// oracle/Makefile target `ref`); the output goes to oracle/_ref/ (git-ignored).
// No reference source is copied: this shim only drives FastNoiseLite the way
// soil::noise does (source/soillib/op/noise.hpp:14-56):
//   update()      noise.hpp:16-23  (OpenSimplex2, FBm, frequency/octaves/gain/lacunarity)
//   operator()    noise.hpp:37-39  GetNoise(pos[0]/ext[0], pos[1]/ext[1], seed)
//   soil::noise   noise.hpp:42-56  row-major loop over shape.unflatten(i)
#include <FastNoiseLite.h>

#include <cstdint>

extern "C" void fnl_ref_noise(float* out, int64_t H, int64_t W, float frequency, int octaves,
                              float gain, float lacunarity, float seed, float ext0, float ext1) {
  FastNoiseLite source;  // default-constructed, as the member at noise.hpp:28
  source.SetNoiseType(FastNoiseLite::NoiseType_OpenSimplex2);
  source.SetFractalType(FastNoiseLite::FractalType_FBm);
  source.SetFrequency(frequency);
  source.SetFractalOctaves(octaves);
  source.SetFractalGain(gain);
  source.SetFractalLacunarity(lacunarity);
  for (int64_t i = 0; i < H * W; ++i) {
    const int px = static_cast<int>(i / W), py = static_cast<int>(i % W);
    out[i] = source.GetNoise(px / ext0, py / ext1, seed);
  }
}
