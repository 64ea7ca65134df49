// soil::io::tiff / soil::io::geotiff of include/soil.hpp (io/tiff.hpp, io/geotiff.hpp):
// write -> read round trip with GeoTIFF tags and NoData -> NaN; no GPU involved.
#include <cmath>
#include <cstdio>

#include "soil.hpp"

#define EXPECT(c) do { if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/tmp/soil_cpp_io.tiff";
  const uint32_t W = 48, H = 48;
  std::vector<float> dem(W * H);
  for (uint32_t i = 0; i < W * H; ++i) dem[i] = 0.25f * float(i % 97) - 3.0f;
  dem[5 * W + 7] = -9999.0f;
  soil::io::geotiff out(dem, W, H);
  EXPECT(out._meta.coords[3] == W && out._meta.coords[4] == H);
  out._meta.scale = {30.0, -30.0, 0.0};
  out._meta.coords = {0, 0, 0, 5e5, 4.1e6, 0};
  out._meta.gdal_nodata = "-9999";
  out._meta.keydir = {1, 1, 0, 0};
  EXPECT(out.write(path));

  soil::io::geotiff in(path);
  EXPECT(in.width() == W && in.height() == H && in.bits() == 32);
  EXPECT(in.shape()[0] == W && in.shape()[1] == H);
  EXPECT(in._meta.scale.size() == 3 && in._meta.scale[1] == -30.0 && in._meta.scale[2] == 1.0);
  EXPECT(in._meta.coords[3] == 5e5 && in._meta.gdal_nodata == "-9999" && in._meta.keydir.size() == 4);
  for (uint32_t i = 0; i < W * H; ++i) {
    if (i == 5 * W + 7) EXPECT(std::isnan(in.f32[i]));
    else EXPECT(in.f32[i] == dem[i]);
  }
  std::vector<double> wide(dem.begin(), dem.end());
  soil::io::tiff t64(wide, W, H);
  EXPECT(t64.write(path));
  soil::io::tiff back(path);
  EXPECT(back.bits() == 64 && back.f64.size() == W * H && back.f64[11] == wide[11]);
  try { soil::io::tiff missing("/nonexistent/dir/x.tiff"); return 1; } catch (const std::runtime_error&) {}
  std::printf("CPP_IO_OK\n");
  return 0;
}
