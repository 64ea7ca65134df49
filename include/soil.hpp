// soil.hpp — header-only C++ host mirror of the reference's operator API over the
// C ABI (soil_hip.h).  Same function names, argument order and meaning as the
// free functions of `namespace soil` in the reference:
//   source/soillib/model/path/erosion.hpp:69-166   graph/graph.hpp:49-63
//   model/grad/grad.hpp:11-17   model/filter/filter.hpp:11   model/path/path.hpp:30-37
//   op/noise.hpp:42
// plus the sliver of `silt` those signatures need (shape, tensor_t<T>, host_t,
// rng): ref-counted device buffers that can be passed by value like the
// reference's handles.  Errors of the C ABI become C++ exceptions, as the
// reference throws std::invalid_argument / silt::error::mismatch_host.
//
//   g++ -std=c++17 -Iinclude app.cpp -Lsoillib_amd/lib -lsoil_hip
#pragma once

#include <array>
#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "soil_hip.h"
#include "soil_slab.h"

namespace silt {

enum host_t { CPU = 0, GPU = 1 };
using rng = soil_rng;  // erosion.hpp:6 uses curandState; see DESIGN.md §4
struct vec2 { float x, y; };
struct vec3 { float x, y, z; };

namespace error {
struct mismatch_host : std::runtime_error {
  mismatch_host(host_t want, host_t got)
      : std::runtime_error(std::string("mismatch_host: expected ") + (want == GPU ? "GPU" : "CPU") +
                           ", got " + (got == GPU ? "GPU" : "CPU")) {}
};
}  // namespace error

inline void check(int rc) {
  if (rc == SOIL_OK) return;
  if (rc == SOIL_ERR_INVALID_ARGUMENT) throw std::invalid_argument(soil_last_error());
  if (rc == SOIL_ERR_OUT_OF_MEMORY) throw std::bad_alloc();
  throw std::runtime_error(soil_last_error());
}

class shape {  // dense row-major, flatten((x, y)) = x*shape[1] + y
 public:
  shape() = default;
  shape(int64_t a) : n_(1), d_{a, 1, 1, 1} {}
  shape(int64_t a, int64_t b) : n_(2), d_{a, b, 1, 1} {}
  shape(int64_t a, int64_t b, int64_t c) : n_(3), d_{a, b, c, 1} {}
  int64_t operator[](int i) const { return i < n_ ? d_[i] : 1; }
  int dim() const { return n_; }
  int64_t elem() const { return d_[0] * d_[1] * d_[2] * d_[3]; }
 private:
  int n_ = 0;
  std::array<int64_t, 4> d_{0, 1, 1, 1};
};

template <typename T>
class tensor_t {  // shared handle to a GPU buffer (host tensors: std::vector on the caller's side)
 public:
  tensor_t() = default;
  tensor_t(const shape& s, host_t host = GPU) : shape_(s) {
    if (host != GPU) throw error::mismatch_host(GPU, host);
    void* p = nullptr;
    check(soil_malloc(&p, sizeof(T) * static_cast<size_t>(s.elem())));
    mem_ = std::shared_ptr<void>(p, [](void* q) { soil_free(q); });
  }
  static tensor_t from_host(const std::vector<T>& v, const shape& s) {
    tensor_t t(s, GPU);
    check(soil_memcpy_h2d(t.data(), v.data(), sizeof(T) * v.size(), nullptr));
    return t;
  }
  std::vector<T> to_host() const {
    std::vector<T> v(static_cast<size_t>(elem()));
    check(soil_memcpy_d2h(v.data(), data(), sizeof(T) * v.size(), nullptr));
    return v;
  }
  T* data() const { return static_cast<T*>(mem_.get()); }
  const silt::shape& shape() const { return shape_; }
  int64_t elem() const { return shape_.elem(); }
  host_t host() const { return GPU; }
 private:
  std::shared_ptr<void> mem_;
  silt::shape shape_;
};

inline void set(tensor_t<float> t, float v) { check(soil_set_f32(t.data(), v, t.elem(), nullptr)); }
inline void set(tensor_t<int> t, int v) { check(soil_set_i32(t.data(), v, t.elem(), nullptr)); }
inline void add(tensor_t<float> a, tensor_t<float> b) { check(soil_add_f32(a.data(), b.data(), a.elem(), nullptr)); }
inline void multiply(tensor_t<float> a, float v) { check(soil_multiply_f32(a.data(), v, a.elem(), nullptr)); }
inline void seed(tensor_t<rng> r, uint64_t seed, uint64_t offset) {
  check(soil_rng_seed(r.data(), r.elem(), seed, offset, nullptr));
}

}  // namespace silt

namespace soil {

struct param_t : soil_param {  // erosion.hpp:17-58, defaults included
  param_t() { soil_param_default(this); }
};
enum edge_t { D4 = SOIL_D4, D8 = SOIL_D8 };  // graph.hpp:11-14
using silt::check;
using F = silt::tensor_t<float>;

namespace detail {
struct s3 { float v[3]; s3(silt::vec3 s) : v{s.x, s.y, s.z} {} };
struct s2 { float v[2]; s2(silt::vec2 s) : v{s.x, s.y} {} };
}  // namespace detail

// ---- erosion.hpp:69-133 ---------------------------------------------------------------------
inline void transport_fluvial(F layers, F rainfall, F discharge, F discharge_track, F mass,
                              F mass_track, F momentum, F momentum_track, F albedo_bedrock,
                              F albedo_transport, F albedo_surface, silt::tensor_t<silt::rng> rng,
                              const silt::vec3 scale, const param_t param) {
  const auto s = layers.shape();
  check(soil_transport_fluvial(layers.data(), rainfall.data(), discharge.data(),
                               discharge_track.data(), mass.data(), mass_track.data(),
                               momentum.data(), momentum_track.data(), albedo_bedrock.data(),
                               albedo_transport.data(), albedo_surface.data(), rng.data(),
                               rng.elem(), s[0], s[1], detail::s3(scale).v, &param, nullptr));
}
inline void transport_debris(F layers, F velocity, F velocity_track, F mass, F mass_track,
                             F albedo_bedrock, F albedo_transport, F albedo_surface,
                             silt::tensor_t<silt::rng> rng, const silt::vec3 scale,
                             const param_t param) {
  const auto s = layers.shape();
  check(soil_transport_debris(layers.data(), velocity.data(), velocity_track.data(), mass.data(),
                              mass_track.data(), albedo_bedrock.data(), albedo_transport.data(),
                              albedo_surface.data(), rng.data(), rng.elem(), s[0], s[1],
                              detail::s3(scale).v, &param, nullptr));
}
inline void mass_transfer(F delta, F layers, const F uplift, const F discharge, const F mass,
                          const F momentum, const F debris, const F momentumDebris,
                          F albedo_bedrock, F albedo_transport_fluvial, F albedo_transport_debris,
                          F albedo_surface, const silt::vec3 scale, const param_t param) {
  const auto s = uplift.shape();
  check(soil_mass_transfer(delta.data(), layers.data(), uplift.data(), discharge.data(),
                           mass.data(), momentum.data(), debris.data(), momentumDebris.data(),
                           albedo_bedrock.data(), albedo_transport_fluvial.data(),
                           albedo_transport_debris.data(), albedo_surface.data(), s[0], s[1],
                           detail::s3(scale).v, &param, nullptr));
}
inline void mass_creep(F delta, const F layers, const silt::vec3 scale, const param_t param) {
  const auto s = layers.shape();
  check(soil_mass_creep(delta.data(), layers.data(), s[0], s[1], detail::s3(scale).v, &param, nullptr));
}
inline void layer_merge(F height, const F layers) {
  check(soil_layer_merge(height.data(), layers.data(), height.elem(), nullptr));
}

// ---- the legacy step API (example/erosion_gpu.py:44-106; binding python/source/model.cpp:62-143,
// commented out in the snapshot) -----------------------------------------------------------------
struct map_t {  // model.cpp:67-97: terrain planes + pixel scale
  map_t(const silt::shape shape, const silt::vec3 scale) : shape(shape), scale(scale) {}
  silt::shape shape;
  silt::vec3 scale;
  F height, sediment, uplift, rainfall;  // height = bedrock surface
  uint64_t steps_taken = 0;              // numbers the particle streams of the next step
};
struct data_t {  // model.cpp:103-140: transported quantities, or their flux accumulators ("track")
  explicit data_t(const silt::shape shape) : shape(shape) {}
  silt::shape shape;
  F discharge, momentum, mass, debris, debris_momentum;
};
struct erode_param_t : param_t {  // `param.samples`: particles per step (erosion_gpu.py:77)
  size_t samples = 8192;
};
// soil::erode — `steps` whole erosion steps on the model's planes, in place.  The loop, the
// re-seeding of the particle streams and the layer double buffer are the library's (soil_erode).
inline void erode(map_t& model, data_t& data, data_t& track, const erode_param_t& param,
                  const int steps = 1) {
  soil_erode_model m{};
  m.height = model.height.data();
  m.sediment = model.sediment.data();
  m.uplift = model.uplift.data();
  m.rainfall = model.rainfall.data();
  m.discharge = data.discharge.data();
  m.mass = data.mass.data();
  m.momentum = data.momentum.data();
  m.debris = data.debris.data();
  m.debris_momentum = data.debris_momentum.data();
  m.discharge_track = track.discharge.data();
  m.mass_track = track.mass.data();
  m.momentum_track = track.momentum.data();
  m.debris_track = track.debris.data();
  m.debris_momentum_track = track.debris_momentum.data();
  check(soil_erode(&m, model.shape[0], model.shape[1], static_cast<int64_t>(param.samples), 0,
                   model.steps_taken, steps, detail::s3(model.scale).v, &param, nullptr));
  model.steps_taken += static_cast<uint64_t>(steps);
}

// ---- graph.hpp:49-63 ------------------------------------------------------------------------
inline silt::tensor_t<int> direction(const F height, const edge_t edge) {
  silt::tensor_t<int> out(height.shape(), silt::GPU);
  check(soil_direction(out.data(), height.data(), height.shape()[0], height.shape()[1], edge, nullptr));
  return out;
}
inline silt::tensor_t<int> steepest(const F height, const edge_t edge) {
  silt::tensor_t<int> out(height.shape(), silt::GPU);
  check(soil_steepest(out.data(), height.data(), height.shape()[0], height.shape()[1], edge, nullptr));
  return out;
}
inline silt::tensor_t<int> random_weighted(const F height, const edge_t edge, const size_t seed,
                                           const size_t offset, const float T) {
  silt::tensor_t<int> out(height.shape(), silt::GPU);
  check(soil_random_weighted(out.data(), height.data(), height.shape()[0], height.shape()[1], edge,
                             seed, offset, T, nullptr));
  return out;
}
inline F accumulate(const silt::tensor_t<int> graph, const F source, const edge_t edge) {
  F out(graph.shape(), silt::GPU);
  check(soil_accumulate(out.data(), graph.data(), source.data(), nullptr, graph.shape()[0],
                        graph.shape()[1], edge, nullptr));
  return out;
}
inline F accumulate_decay(const silt::tensor_t<int> graph, const F source, const F decay,
                          const edge_t edge) {
  F out(graph.shape(), silt::GPU);
  check(soil_accumulate(out.data(), graph.data(), source.data(), decay.data(), graph.shape()[0],
                        graph.shape()[1], edge, nullptr));
  return out;
}
inline F fill_depressions(const F height, const edge_t edge) {  // build-defined
  F out(height.shape(), silt::GPU);
  check(soil_fill_depressions(out.data(), height.data(), height.shape()[0], height.shape()[1], edge,
                              nullptr));
  return out;
}
inline F slope(const F tensor, const silt::tensor_t<int> flow, const silt::vec2 scale) {
  F out(tensor.shape(), silt::GPU);
  check(soil_slope(out.data(), tensor.data(), flow.data(), tensor.shape()[0], tensor.shape()[1],
                   detail::s2(scale).v, nullptr));
  return out;
}

// ---- grad.hpp:11-17, filter.hpp:11 ------------------------------------------------------------
inline F gradient(const F& tensor, const silt::vec2 scale) {
  F out(silt::shape(tensor.shape()[0], tensor.shape()[1], 2), silt::GPU);
  check(soil_gradient(out.data(), tensor.data(), tensor.shape()[0], tensor.shape()[1],
                      detail::s2(scale).v, nullptr));
  return out;
}
inline F negslope(const F& tensor, const silt::vec2 scale) {
  F out(silt::shape(tensor.shape()[0], tensor.shape()[1]), silt::GPU);
  check(soil_negslope(out.data(), tensor.data(), tensor.shape()[0], tensor.shape()[1],
                      detail::s2(scale).v, nullptr));
  return out;
}
inline F laplacian(const F& tensor, const silt::vec2 scale) {
  F out(tensor.shape(), silt::GPU);
  check(soil_laplacian(out.data(), tensor.data(), tensor.shape()[0], tensor.shape()[1],
                       static_cast<int>(tensor.shape()[2]), detail::s2(scale).v, nullptr));
  return out;
}
inline F gaussian_blur(F tensor, const float sigma) {  // blurs in place, returns its input (filter.cu:90)
  F scratch(tensor.shape(), silt::GPU);
  check(soil_gaussian_blur(tensor.data(), scratch.data(), tensor.shape()[0], tensor.shape()[1],
                           static_cast<int>(tensor.shape()[2]), sigma, nullptr));
  return tensor;
}

// ---- path.hpp:30-37, noise.hpp:14-56 -----------------------------------------------------------
inline F solve_uniform(const F flow, const F source, const F decay, silt::tensor_t<silt::rng> rng,
                       const silt::vec2 scale, const size_t count) {
  F flux(source.shape(), silt::GPU);
  check(soil_solve_uniform(flux.data(), flow.data(), source.data(), decay.data(), rng.data(),
                           rng.elem(), source.shape()[0], source.shape()[1],
                           static_cast<int>(source.shape()[2]), detail::s2(scale).v, count, nullptr));
  return flux;
}
// ---- the sharded step (soil_slab.h): one slab_runner per rank / GPU -------------------------------
//
// The reference has no multi-GPU path; BASELINE configs[4] cuts the 16384^2 grid into row slabs.
// A rank makes its device current (soil_set_device), builds a communicator — rccl(id, rank, world)
// with the 128-byte id of rccl_unique_id() handed round by whatever launched the job (MPI_Bcast, a
// file, ...) — and steps:
//     soil::comm wire = soil::comm::rccl(id, rank, world);
//     soil::slab_runner slab(cfg, param, wire);
//     for (int s = 0; s < steps; ++s) slab.step();
class comm {
 public:
  static std::array<uint8_t, 128> rccl_unique_id() {
    std::array<uint8_t, 128> id{};
    silt::check(soil_comm_rccl_unique_id(id.data()));
    return id;
  }
  static comm rccl(const std::array<uint8_t, 128>& id, int rank, int world) {
    soil_comm* c = nullptr;
    silt::check(soil_comm_rccl_create(&c, id.data(), rank, world));
    return comm(c, &soil_comm_rccl_destroy);
  }
  static comm self() {  // a world of one
    soil_comm* c = nullptr;
    silt::check(soil_comm_self_create(&c));
    return comm(c, &soil_comm_self_destroy);
  }
  const soil_comm* get() const { return c_.get(); }
  // one grouped point-to-point exchange (ncclGroupStart .. ncclRecv / ncclSend .. ncclGroupEnd on an
  // RCCL communicator), stream-ordered on `stream` — what soil_slab_step issues for its halos
  void exchange(const std::vector<soil_xfer>& sends, const std::vector<soil_xfer>& recvs, void* stream = nullptr) const {
    silt::check(c_->exchange(c_->ctx, sends.data(), static_cast<int32_t>(sends.size()), recvs.data(),
                             static_cast<int32_t>(recvs.size()), stream));
  }
  void all_reduce_sum(float* buf, int64_t n, void* stream = nullptr) const {
    silt::check(c_->all_reduce_sum_f32(c_->ctx, buf, n, stream));
  }
  int rank() const { return c_->rank; }
  int world() const { return c_->world; }
 private:
  comm(soil_comm* c, int (*drop)(soil_comm*)) : c_(c, [drop](soil_comm* q) { drop(q); }) {}
  std::shared_ptr<soil_comm> c_;
};

class slab_runner {
 public:
  // rows_per_rank x W cells owned by every rank; N = H * W / particles_div particles per launch
  static soil_slab_config config(int64_t rows_per_rank, int64_t W, int64_t particles_div = 8, uint64_t seed = 0) {
    soil_slab_config c{};
    c.rows_per_rank = rows_per_rank, c.W = W, c.particles_div = particles_div, c.seed = seed;
    c.noise_seed = 3.0f, c.init = 1, c.trim = -1, c.pair = -1, c.mode = -1;
    return c;
  }
  slab_runner(const soil_slab_config& cfg, const param_t& param, comm wire) : wire_(std::move(wire)) {
    soil_slab* s = nullptr;
    silt::check(soil_slab_create(&s, &cfg, &param, wire_.get(), nullptr));
    s_.reset(s, [](soil_slab* q) { soil_slab_destroy(q); });
  }
  void step() { silt::check(soil_slab_step(s_.get(), nullptr, nullptr)); }
  void sync() { silt::check(soil_slab_sync(s_.get())); }
  soil_slab_info info() const {
    soil_slab_info i{};
    silt::check(soil_slab_get_info(s_.get(), &i));
    return i;
  }
  // the owned rows of a plane ("layers", "height", "waterHeight", ...) copied to the host
  std::vector<float> owned_rows(const char* name) {
    float* p = nullptr;
    int64_t rows = 0, ch = 0;
    silt::check(soil_slab_plane(s_.get(), name, &p, &rows, &ch));
    const soil_slab_info i = info();
    sync();
    std::vector<float> out(static_cast<size_t>((i.r1 - i.r0) * i.W * ch));
    silt::check(soil_memcpy_d2h(out.data(), p + i.r0 * i.W * ch, out.size() * sizeof(float), nullptr));
    return out;
  }
 private:
  comm wire_;
  std::shared_ptr<soil_slab> s_;
};

struct noise_param_t : soil_noise_param {
  noise_param_t() { soil_noise_param_default(this); }
};
inline F noise(const silt::shape shape, noise_param_t param) {  // generated straight into HBM
  if (shape.dim() != 2)
    throw std::invalid_argument("can't extract a full noise buffer from a non-2D index");
  F out(shape, silt::GPU);
  check(soil_noise(out.data(), shape[0], shape[1], &param, nullptr));
  return out;
}

// ---- io/tiff.hpp:20-241, io/geotiff.hpp:63-318 -------------------------------------------------
// Host-side raster IO.  The samples live in a std::vector (the reference keeps a CPU
// silt::tensor): `f32` unless the file holds 64-bit samples (tiff.hpp:116-124), in
// scanline order; shape() is (width, height) as in the reference.
namespace io {

struct tiff {
  tiff() = default;
  explicit tiff(const char* filename) { read(filename); }
  tiff(const std::vector<float>& data, uint32_t width, uint32_t height)
      : f32(data), _width(width), _height(height), _bits(32) {}
  tiff(const std::vector<double>& data, uint32_t width, uint32_t height)
      : f64(data), _width(width), _height(height), _bits(64) {}

  bool peek(const char* filename) {
    check_io(soil_tiff_peek(filename, &info));
    _width = info.width, _height = info.height, _bits = info.bits;
    meta_loaded = true;
    return true;
  }
  bool read(const char* filename) {
    if (!meta_loaded) peek(filename);
    const size_t n = static_cast<size_t>(_width) * _height;
    if (_bits == 64) {
      f64.assign(n, 0.0);
      check_io(soil_tiff_read(filename, f64.data(), n * sizeof(double)));
    } else {
      f32.assign(n, 0.0f);
      check_io(soil_tiff_read(filename, f32.data(), n * sizeof(float)));
    }
    return true;
  }
  bool write(const char* filename) { return write_tags(filename, nullptr); }

  uint32_t bits() const { return _bits; }
  uint32_t width() const { return _width; }
  uint32_t height() const { return _height; }
  silt::shape shape() const { return silt::shape(_width, _height); }

  std::vector<float> f32;
  std::vector<double> f64;

 protected:
  static void check_io(int rc) {
    if (rc == SOIL_ERR_IO) throw std::runtime_error(soil_last_error());  // silt::error::missing_file
    silt::check(rc);
  }
  bool write_tags(const char* filename, const soil_geotiff_tags* geo) {
    const void* data = _bits == 64 ? static_cast<const void*>(f64.data()) : f32.data();
    check_io(soil_tiff_write(filename, data, _width, _height, _bits, geo));
    return true;
  }
  bool meta_loaded = false;
  soil_tiff_info info{};
  uint32_t _width = 0, _height = 0, _bits = 0;
};

struct geotiff : tiff {
  struct meta_t {  // geotiff.hpp:83-101
    std::string filename;
    size_t width = 0, height = 0, bits = 0;
    std::string gdal_nodata, gdal_metadata, geoasciiparams;
    std::vector<double> scale = {1.0, 1.0, 1.0};
    std::vector<double> coords = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    std::vector<double> params;
    std::vector<short> keydir;
  };

  geotiff() = default;
  explicit geotiff(const char* filename) { read(filename); }
  geotiff(const std::vector<float>& data, uint32_t width, uint32_t height) : tiff(data, width, height) {
    _meta.coords[3] = width;  // geotiff.hpp:72-73
    _meta.coords[4] = height;
  }

  bool peek(const char* filename) {  // geotiff.hpp:131-173
    tiff::peek(filename);
    _meta.filename = filename;
    _meta.width = _width, _meta.height = _height, _meta.bits = _bits;
    auto text = [&](int tag, uint32_t n, std::string& out) {
      if (!n) return;
      std::vector<char> buf(n + 1, 0);
      uint64_t got = 0;
      check_io(soil_tiff_tag(filename, tag, buf.data(), n, &got));
      out = std::string(buf.data());
    };
    auto reals = [&](int tag, uint32_t n, std::vector<double>& out) {
      if (!n) return;
      out.assign(n, 0.0);
      uint64_t got = 0;
      check_io(soil_tiff_tag(filename, tag, out.data(), n * sizeof(double), &got));
    };
    text(SOIL_TIFFTAG_GDAL_NODATA, info.n_nodata, _meta.gdal_nodata);
    text(SOIL_TIFFTAG_GDAL_METADATA, info.n_metadata, _meta.gdal_metadata);
    text(SOIL_TIFFTAG_GEOASCIIPARAMS, info.n_ascii, _meta.geoasciiparams);
    reals(SOIL_TIFFTAG_GEOPIXELSCALE, info.n_scale, _meta.scale);
    if (_meta.scale.size() > 2 && _meta.scale[2] == 0.0) _meta.scale[2] = 1.0;  // :160-161
    reals(SOIL_TIFFTAG_GEOTIEPOINTS, info.n_tiepoints, _meta.coords);
    reals(SOIL_TIFFTAG_GEODOUBLEPARAMS, info.n_params, _meta.params);
    if (info.n_keydir) {
      _meta.keydir.assign(info.n_keydir, 0);
      uint64_t got = 0;
      check_io(soil_tiff_tag(filename, SOIL_TIFFTAG_GEOKEYDIRECTORY, _meta.keydir.data(),
                             info.n_keydir * sizeof(short), &got));
    }
    return true;
  }
  bool read(const char* filename) {  // geotiff.hpp:175-182, NoData -> NaN :228-263
    peek(filename);
    tiff::read(filename);
    if (!_meta.gdal_nodata.empty()) {
      if (_bits == 64) {
        const double nodata = std::stod(_meta.gdal_nodata);
        for (double& v : f64) if (v == nodata) v = std::numeric_limits<double>::quiet_NaN();
      } else {
        const float nodata = std::stof(_meta.gdal_nodata);
        for (float& v : f32) if (v == nodata) v = std::numeric_limits<float>::quiet_NaN();
      }
    }
    return true;
  }
  bool write(const char* filename) {  // geotiff.hpp:183-226
    soil_geotiff_tags g{};
    g.scale = _meta.scale.data(), g.n_scale = static_cast<uint32_t>(_meta.scale.size());
    g.tiepoints = _meta.coords.data(), g.n_tiepoints = static_cast<uint32_t>(_meta.coords.size());
    g.params = _meta.params.data(), g.n_params = static_cast<uint32_t>(_meta.params.size());
    g.keydir = _meta.keydir.data(), g.n_keydir = static_cast<uint32_t>(_meta.keydir.size());
    g.ascii = _meta.geoasciiparams.empty() ? nullptr : _meta.geoasciiparams.c_str();
    g.metadata = _meta.gdal_metadata.empty() ? nullptr : _meta.gdal_metadata.c_str();
    g.nodata = _meta.gdal_nodata.empty() ? nullptr : _meta.gdal_nodata.c_str();
    return write_tags(filename, &g);
  }
  silt::vec2 scale() const { return {static_cast<float>(_meta.scale[0]), static_cast<float>(_meta.scale[1])}; }
  meta_t _meta;
};

}  // namespace io

}  // namespace soil
