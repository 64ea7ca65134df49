// io_tiff.hip — TIFF / GeoTIFF reading and writing behind the C ABI (host code
// only; it lives in libsoil_hip.so so that C++ and Python callers share it).
//
// The reference reads and writes DEMs through libtiff (io/tiff.hpp:69-241,
// io/geotiff.hpp:131-226); libtiff is a third-party dependency that is not part
// of the reference tree, so this file is a codec of its own for the subset of
// TIFF 6.0 / BigTIFF that single-band raster DEMs use:
//   read : classic and BigTIFF, both byte orders, strips or tiles, compression
//          none / LZW / Deflate / PackBits, predictors 1, 2 and 3 (floating point),
//          IEEE float 16/32/64 and 8/16/32-bit integers (converted to fp32)
//   write: little-endian, uncompressed strips, IEEE float 32/64 — what
//          tiff::write / geotiff::write emit (tiff.hpp:215-241, geotiff.hpp:183-226),
//          including ROWSPERSTRIP = TIFFDefaultStripSize(tif, width) = width.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "common.hpp"

namespace soil {
namespace {

enum : int {
  T_WIDTH = 256, T_LENGTH = 257, T_BITS = 258, T_COMPRESSION = 259, T_PHOTOMETRIC = 262,
  T_STRIPOFFSETS = 273, T_ORIENTATION = 274, T_SAMPLES = 277, T_ROWSPERSTRIP = 278,
  T_STRIPBYTECOUNTS = 279, T_PLANAR = 284, T_PREDICTOR = 317, T_TILEWIDTH = 322,
  T_TILELENGTH = 323, T_TILEOFFSETS = 324, T_TILEBYTECOUNTS = 325, T_SAMPLEFORMAT = 339,
};

struct Entry {
  uint16_t tag = 0, type = 0;
  uint64_t count = 0;
  const uint8_t* data = nullptr;  // payload inside the file image
};

struct File {
  std::vector<uint8_t> bytes;
  bool big_endian = false, bigtiff = false;
  std::vector<Entry> entries;

  uint16_t u16(const uint8_t* p) const {
    return big_endian ? static_cast<uint16_t>(p[0] << 8 | p[1]) : static_cast<uint16_t>(p[1] << 8 | p[0]);
  }
  uint32_t u32(const uint8_t* p) const {
    return big_endian ? (uint32_t(p[0]) << 24 | uint32_t(p[1]) << 16 | uint32_t(p[2]) << 8 | p[3])
                      : (uint32_t(p[3]) << 24 | uint32_t(p[2]) << 16 | uint32_t(p[1]) << 8 | p[0]);
  }
  uint64_t u64(const uint8_t* p) const {
    return big_endian ? (uint64_t(u32(p)) << 32 | u32(p + 4)) : (uint64_t(u32(p + 4)) << 32 | u32(p));
  }
  const Entry* find(int tag) const {
    for (const Entry& e : entries)
      if (e.tag == tag) return &e;
    return nullptr;
  }
};

size_t type_size(uint16_t type) {
  switch (type) {
    case 1: case 2: case 6: case 7: return 1;  // BYTE ASCII SBYTE UNDEFINED
    case 3: case 8: return 2;                  // SHORT SSHORT
    case 4: case 9: case 11: case 13: return 4;  // LONG SLONG FLOAT IFD
    case 5: case 10: case 12: case 16: case 17: case 18: return 8;  // RATIONALs DOUBLE LONG8s
    default: return 0;
  }
}

int open_file(const char* filename, File& f) {
  SOIL_REQUIRE(filename != nullptr, "tiff: null filename");
  std::FILE* fp = std::fopen(filename, "rb");
  if (!fp) {
    set_error(std::string("tiff: missing file: ") + filename);  // silt::error::missing_file, tiff.hpp:73
    return SOIL_ERR_IO;
  }
  std::fseek(fp, 0, SEEK_END);
  const long size = std::ftell(fp);
  std::fseek(fp, 0, SEEK_SET);
  f.bytes.resize(size > 0 ? static_cast<size_t>(size) : 0);
  const size_t got = f.bytes.empty() ? 0 : std::fread(f.bytes.data(), 1, f.bytes.size(), fp);
  std::fclose(fp);
  SOIL_REQUIRE_IO(got == f.bytes.size() && got >= 8, "tiff: short read");
  const uint8_t* b = f.bytes.data();
  SOIL_REQUIRE_IO((b[0] == 'I' && b[1] == 'I') || (b[0] == 'M' && b[1] == 'M'), "tiff: bad byte-order mark");
  f.big_endian = b[0] == 'M';
  const uint16_t magic = f.u16(b + 2);
  SOIL_REQUIRE_IO(magic == 42 || magic == 43, "tiff: bad magic number");
  f.bigtiff = magic == 43;
  const uint64_t ifd = f.bigtiff ? f.u64(b + 8) : f.u32(b + 4);
  const uint64_t n = f.bytes.size();
  // Every bound below is written so that it cannot wrap: offsets and counts come straight
  // from the file (libtiff rejects such headers with an error; so does this).
  const uint64_t esz = f.bigtiff ? 20 : 12, head = f.bigtiff ? 8 : 2;
  SOIL_REQUIRE_IO(f.bigtiff ? n >= 16 : true, "tiff: short BigTIFF header");
  SOIL_REQUIRE_IO(ifd <= n && head <= n - ifd, "tiff: IFD offset outside the file");
  const uint64_t count = f.bigtiff ? f.u64(b + ifd) : f.u16(b + ifd);
  SOIL_REQUIRE_IO(count <= (n - ifd - head) / esz, "tiff: IFD runs past the end of the file");
  for (uint64_t i = 0; i < count; ++i) {
    const uint8_t* e = b + ifd + head + i * esz;
    Entry en;
    en.tag = f.u16(e);
    en.type = f.u16(e + 2);
    en.count = f.bigtiff ? f.u64(e + 4) : f.u32(e + 4);
    const uint64_t tsz = type_size(en.type);
    // an unknown field type or a count the file cannot hold: the tag is skipped, like
    // libtiff warns and goes on (a payload is never larger than the file it sits in)
    if (tsz == 0 || en.count > n / tsz) continue;
    const uint64_t bytes = tsz * en.count;
    const uint64_t inl = f.bigtiff ? 8 : 4;
    const uint8_t* v = e + (f.bigtiff ? 12 : 8);
    if (bytes <= inl) {
      en.data = v;
    } else {
      const uint64_t off = f.bigtiff ? f.u64(v) : f.u32(v);
      if (off > n || bytes > n - off) continue;  // a damaged tag is skipped as well
      en.data = b + off;
    }
    f.entries.push_back(en);
  }
  return SOIL_OK;
}

std::vector<uint64_t> ints(const File& f, const Entry* e) {
  std::vector<uint64_t> v;
  if (!e) return v;
  v.reserve(e->count);
  for (uint64_t i = 0; i < e->count; ++i) {
    const uint8_t* p = e->data + i * type_size(e->type);
    switch (e->type) {
      case 1: case 6: case 7: v.push_back(p[0]); break;
      case 3: case 8: v.push_back(f.u16(p)); break;
      case 4: case 9: case 13: v.push_back(f.u32(p)); break;
      case 16: case 17: case 18: v.push_back(f.u64(p)); break;
      default: v.push_back(0);
    }
  }
  return v;
}
uint64_t int1(const File& f, int tag, uint64_t fallback) {
  const auto v = ints(f, f.find(tag));
  return v.empty() ? fallback : v[0];
}
std::vector<double> doubles(const File& f, const Entry* e) {
  std::vector<double> v;
  if (!e || e->type != 12) return v;
  for (uint64_t i = 0; i < e->count; ++i) {
    const uint64_t bits = f.u64(e->data + 8 * i);
    double d;
    std::memcpy(&d, &bits, 8);
    v.push_back(d);
  }
  return v;
}

// ---- decompression -----------------------------------------------------------------

bool lzw_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
  // TIFF 6.0 section 13: MSB-first codes of 9..12 bits, ClearCode 256, EOI 257,
  // code width grows one code early ("early change")
  struct Node { int32_t prev; uint8_t ch; uint16_t len; };
  static thread_local std::vector<Node> table(4096);
  for (int i = 0; i < 256; ++i) table[i] = {-1, static_cast<uint8_t>(i), 1};
  out.clear();
  out.reserve(expect);
  uint32_t acc = 0;
  int nbits = 0, width = 9, next = 258;
  int32_t prev = -1;
  size_t pos = 0;
  std::vector<uint8_t> tmp;
  for (;;) {
    while (nbits < width && pos < n) {
      acc = (acc << 8) | src[pos++];
      nbits += 8;
    }
    if (nbits < width) break;
    const int code = static_cast<int>((acc >> (nbits - width)) & ((1u << width) - 1u));
    nbits -= width;
    if (code == 257) break;
    if (code == 256) {
      width = 9;
      next = 258;
      prev = -1;
      continue;
    }
    if (prev < 0) {
      if (code >= 256) return false;
      out.push_back(static_cast<uint8_t>(code));
      prev = code;
      continue;
    }
    int32_t emit = code;
    uint8_t first;
    if (code < next) {
      int32_t c = code;
      while (table[c].prev >= 0) c = table[c].prev;
      first = table[c].ch;
    } else if (code == next) {  // KwKwK
      int32_t c = prev;
      while (table[c].prev >= 0) c = table[c].prev;
      first = table[c].ch;
      emit = -1;
    } else {
      return false;
    }
    if (next < 4096) {
      table[next] = {prev, first, static_cast<uint16_t>(table[prev].len + 1)};
      if (emit < 0) emit = next;
      ++next;
    } else if (emit < 0) {
      return false;
    }
    const size_t len = table[emit].len, at = out.size();
    out.resize(at + len);
    for (int32_t c = emit, i = static_cast<int32_t>(len) - 1; c >= 0; c = table[c].prev, --i)
      out[at + i] = table[c].ch;
    prev = code < next ? code : next - 1;
    if (next + 1 >= (1 << width) && width < 12) ++width;
    if (out.size() >= expect) break;
  }
  return true;
}

bool packbits_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
  out.clear();
  out.reserve(expect);
  size_t i = 0;
  while (i < n && out.size() < expect) {
    const int8_t h = static_cast<int8_t>(src[i++]);
    if (h >= 0) {
      const size_t len = static_cast<size_t>(h) + 1;
      if (i + len > n) return false;
      out.insert(out.end(), src + i, src + i + len);
      i += len;
    } else if (h != -128) {
      if (i >= n) return false;
      out.insert(out.end(), static_cast<size_t>(1 - h), src[i++]);
    }
  }
  return true;
}

bool inflate_chunk(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
  out.assign(expect, 0);
  uLongf len = static_cast<uLongf>(expect);
  const int rc = uncompress(out.data(), &len, src, static_cast<uLong>(n));
  return rc == Z_OK || rc == Z_BUF_ERROR;
}

// undo predictor 2 (horizontal differencing) / 3 (floating-point) on one row
void unpredict_row(uint8_t* row, size_t width, size_t bps, int predictor, bool big_endian,
                   std::vector<uint8_t>& scratch) {
  if (predictor == 2) {
    auto acc = [&](auto tag) {
      using T = decltype(tag);
      for (size_t i = 1; i < width; ++i) {
        T a, b;
        std::memcpy(&a, row + (i - 1) * sizeof(T), sizeof(T));
        std::memcpy(&b, row + i * sizeof(T), sizeof(T));
        if (big_endian && sizeof(T) > 1) {  // sums are taken on the values, not the file bytes
          a = sizeof(T) == 2 ? static_cast<T>(__builtin_bswap16(static_cast<uint16_t>(a)))
              : sizeof(T) == 4 ? static_cast<T>(__builtin_bswap32(static_cast<uint32_t>(a)))
                               : static_cast<T>(__builtin_bswap64(static_cast<uint64_t>(a)));
          b = sizeof(T) == 2 ? static_cast<T>(__builtin_bswap16(static_cast<uint16_t>(b)))
              : sizeof(T) == 4 ? static_cast<T>(__builtin_bswap32(static_cast<uint32_t>(b)))
                               : static_cast<T>(__builtin_bswap64(static_cast<uint64_t>(b)));
        }
        T s = static_cast<T>(a + b);
        if (big_endian && sizeof(T) > 1)
          s = sizeof(T) == 2 ? static_cast<T>(__builtin_bswap16(static_cast<uint16_t>(s)))
              : sizeof(T) == 4 ? static_cast<T>(__builtin_bswap32(static_cast<uint32_t>(s)))
                               : static_cast<T>(__builtin_bswap64(static_cast<uint64_t>(s)));
        std::memcpy(row + i * sizeof(T), &s, sizeof(T));
      }
    };
    if (bps == 1) acc(uint8_t{});
    else if (bps == 2) acc(uint16_t{});
    else if (bps == 4) acc(uint32_t{});
    else if (bps == 8) acc(uint64_t{});
  } else if (predictor == 3) {
    // TIFF Technical Note 3: bytes of a row are differenced, then stored as bps
    // planes, most significant byte plane first
    const size_t nbytes = width * bps;
    for (size_t i = 1; i < nbytes; ++i) row[i] = static_cast<uint8_t>(row[i] + row[i - 1]);
    scratch.assign(row, row + nbytes);
    for (size_t i = 0; i < width; ++i)
      for (size_t b = 0; b < bps; ++b) {
        const uint8_t v = scratch[b * width + i];  // plane b = byte b counted from the MSB
        row[i * bps + (big_endian ? b : bps - 1 - b)] = v;
      }
  }
}

float half_to_float(uint16_t h) {
  const uint32_t sign = (h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal half: renormalise
      int e = -1;
      uint32_t m = man;
      do { ++e; m <<= 1; } while (!(m & 0x400u));
      bits = sign | ((127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

struct Layout {
  uint32_t width = 0, height = 0, bits = 0, format = 1, samples = 1, compression = 1, predictor = 1;
  bool tiled = false;
  uint32_t tw = 0, th = 0, rows_per_strip = 0;
};

int read_layout(const File& f, Layout& L) {
  SOIL_REQUIRE_IO(f.find(T_WIDTH) && f.find(T_LENGTH), "tiff: ImageWidth / ImageLength missing");
  L.width = static_cast<uint32_t>(int1(f, T_WIDTH, 0));
  L.height = static_cast<uint32_t>(int1(f, T_LENGTH, 0));
  L.bits = static_cast<uint32_t>(int1(f, T_BITS, 1));
  L.format = static_cast<uint32_t>(int1(f, T_SAMPLEFORMAT, 1));
  L.samples = static_cast<uint32_t>(int1(f, T_SAMPLES, 1));
  L.compression = static_cast<uint32_t>(int1(f, T_COMPRESSION, 1));
  L.predictor = static_cast<uint32_t>(int1(f, T_PREDICTOR, 1));
  // libtiff hooks the predictor into its LZW and Deflate codecs only; for any
  // other compression scheme the tag has no effect
  if (L.compression != 5 && L.compression != 8 && L.compression != 32946) L.predictor = 1;
  L.tiled = f.find(T_TILEWIDTH) || f.find(T_TILELENGTH);  // tiff.hpp:88-92
  L.tw = static_cast<uint32_t>(int1(f, T_TILEWIDTH, 0));
  L.th = static_cast<uint32_t>(int1(f, T_TILELENGTH, 0));
  L.rows_per_strip = static_cast<uint32_t>(int1(f, T_ROWSPERSTRIP, 0xffffffffu));
  return SOIL_OK;
}

template <typename D>
void convert_row(D* dst, const uint8_t* src, size_t n, const Layout& L, bool swap) {
  for (size_t i = 0; i < n; ++i) {
    const uint8_t* p = src + i * (L.bits / 8);
    uint64_t raw = 0;
    switch (L.bits) {
      case 8: raw = p[0]; break;
      case 16: { uint16_t v; std::memcpy(&v, p, 2); raw = swap ? __builtin_bswap16(v) : v; break; }
      case 32: { uint32_t v; std::memcpy(&v, p, 4); raw = swap ? __builtin_bswap32(v) : v; break; }
      case 64: { uint64_t v; std::memcpy(&v, p, 8); raw = swap ? __builtin_bswap64(v) : v; break; }
    }
    D out;
    if (L.format == 3) {
      if (L.bits == 16) out = static_cast<D>(half_to_float(static_cast<uint16_t>(raw)));
      else if (L.bits == 32) { uint32_t b = static_cast<uint32_t>(raw); float v; std::memcpy(&v, &b, 4); out = static_cast<D>(v); }
      else { double v; std::memcpy(&v, &raw, 8); out = static_cast<D>(v); }
    } else if (L.format == 2) {
      const int64_t v = L.bits == 8 ? static_cast<int8_t>(raw) : L.bits == 16 ? static_cast<int16_t>(raw)
                      : L.bits == 32 ? static_cast<int32_t>(raw) : static_cast<int64_t>(raw);
      out = static_cast<D>(v);
    } else {
      out = static_cast<D>(raw);
    }
    dst[i] = out;
  }
}

bool host_is_big_endian() {
  const uint16_t one = 1;
  return *reinterpret_cast<const uint8_t*>(&one) == 0;
}

template <typename D>
int decode_image(const File& f, const Layout& L, D* dst) {
  const size_t bps = L.bits / 8;
  const bool swap = f.big_endian != host_is_big_endian();
  const auto offsets = ints(f, f.find(L.tiled ? T_TILEOFFSETS : T_STRIPOFFSETS));
  const auto counts = ints(f, f.find(L.tiled ? T_TILEBYTECOUNTS : T_STRIPBYTECOUNTS));
  SOIL_REQUIRE_IO(!offsets.empty(), "tiff: no strip or tile offsets");
  std::vector<uint8_t> chunk, scratch;
  auto fetch = [&](size_t idx, size_t expect) -> const uint8_t* {
    if (idx >= offsets.size()) return nullptr;
    const uint64_t off = offsets[idx];
    uint64_t cnt = idx < counts.size() ? counts[idx] : 0;
    if (off >= f.bytes.size()) return nullptr;
    if (cnt == 0 || off + cnt > f.bytes.size()) cnt = f.bytes.size() - off;
    const uint8_t* src = f.bytes.data() + off;
    bool ok = true;
    switch (L.compression) {
      case 1:
        if (cnt < expect) return nullptr;
        chunk.assign(src, src + expect);
        break;
      case 5: ok = lzw_decode(src, cnt, chunk, expect); break;
      case 8: case 32946: ok = inflate_chunk(src, cnt, chunk, expect); break;
      case 32773: ok = packbits_decode(src, cnt, chunk, expect); break;
      default: return nullptr;
    }
    if (!ok || chunk.size() < expect) return nullptr;
    return chunk.data();
  };
  if (!L.tiled) {
    const uint32_t rps = L.rows_per_strip == 0 ? L.height : L.rows_per_strip;
    size_t strip = 0;
    for (uint32_t row0 = 0; row0 < L.height; row0 += rps, ++strip) {
      const uint32_t rows = (L.height - row0 < rps) ? L.height - row0 : rps;
      const size_t row_bytes = static_cast<size_t>(L.width) * bps;
      uint8_t* data = const_cast<uint8_t*>(fetch(strip, rows * row_bytes));
      SOIL_REQUIRE_IO(data != nullptr, "tiff: a strip could not be decoded");
      for (uint32_t r = 0; r < rows; ++r) {
        uint8_t* row = data + r * row_bytes;
        unpredict_row(row, L.width, bps, static_cast<int>(L.predictor), f.big_endian, scratch);
        convert_row(dst + static_cast<size_t>(row0 + r) * L.width, row, L.width, L, swap);
      }
    }
  } else {
    SOIL_REQUIRE_IO(L.tw > 0 && L.th > 0, "tiff: tiled image without tile size");
    SOIL_REQUIRE_IO(static_cast<uint64_t>(L.tw) * L.th * bps <= (1ull << 30), "tiff: unreasonable tile size");
    const size_t nx = (L.width + L.tw - 1) / L.tw, ny = (L.height + L.th - 1) / L.th;
    const size_t tile_row = static_cast<size_t>(L.tw) * bps;
    for (size_t ty = 0; ty < ny; ++ty)
      for (size_t tx = 0; tx < nx; ++tx) {
        uint8_t* data = const_cast<uint8_t*>(fetch(ty * nx + tx, tile_row * L.th));
        if (!data) continue;  // tiff.hpp:183-185: a tile that cannot be read is skipped
        for (uint32_t r = 0; r < L.th; ++r) {
          const size_t y = ty * L.th + r;
          if (y >= L.height) break;
          uint8_t* row = data + r * tile_row;
          unpredict_row(row, L.tw, bps, static_cast<int>(L.predictor), f.big_endian, scratch);
          const size_t x0 = tx * L.tw;
          const size_t n = (x0 + L.tw <= L.width) ? L.tw : L.width - x0;
          convert_row(dst + y * L.width + x0, row, n, L, swap);
        }
      }
  }
  return SOIL_OK;
}

// ---- writing ---------------------------------------------------------------------

struct OutTag {
  uint16_t tag, type;
  uint64_t count;
  std::vector<uint8_t> payload;
};

template <typename T>
void put(std::vector<uint8_t>& v, T x) {
  const uint8_t* p = reinterpret_cast<const uint8_t*>(&x);  // little-endian hosts only (x86-64)
  v.insert(v.end(), p, p + sizeof(T));
}
OutTag tag_int(uint16_t tag, uint16_t type, uint64_t value) {
  OutTag t{tag, type, 1, {}};
  if (type == 3) put<uint16_t>(t.payload, static_cast<uint16_t>(value));
  else if (type == 4) put<uint32_t>(t.payload, static_cast<uint32_t>(value));
  else put<uint64_t>(t.payload, value);
  return t;
}

// No exception crosses the C boundary: a damaged file that drives an allocation or a
// container past its limits ends as SOIL_ERR_IO with a message, like libtiff's error return.
template <typename F>
int guarded(const char* what, F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    set_error(std::string(what) + ": out of memory (damaged file?)");
  } catch (const std::exception& e) {
    set_error(std::string(what) + ": " + e.what());
  } catch (...) {
    set_error(std::string(what) + ": unknown failure");
  }
  return SOIL_ERR_IO;
}

}  // namespace
}  // namespace soil

using namespace soil;

extern "C" {

int soil_tiff_peek(const char* filename, soil_tiff_info* info) {
  return guarded("soil_tiff_peek", [&]() -> int {
  SOIL_REQUIRE(info != nullptr, "soil_tiff_peek: null info");
  File f;
  if (int rc = open_file(filename, f); rc != SOIL_OK) return rc;
  Layout L;
  if (int rc = read_layout(f, L); rc != SOIL_OK) return rc;
  std::memset(info, 0, sizeof(*info));
  info->width = L.width;
  info->height = L.height;
  info->bits = L.bits;
  info->sample_format = L.format;
  info->samples = L.samples;
  info->tiled = L.tiled ? 1 : 0;
  info->tile_width = L.tw;
  info->tile_height = L.th;
  info->compression = L.compression;
  info->predictor = L.predictor;
  auto count_of = [&](int tag) {
    const Entry* e = f.find(tag);
    return e ? static_cast<uint32_t>(e->count) : 0u;
  };
  info->n_scale = count_of(SOIL_TIFFTAG_GEOPIXELSCALE);
  info->n_tiepoints = count_of(SOIL_TIFFTAG_GEOTIEPOINTS);
  info->n_params = count_of(SOIL_TIFFTAG_GEODOUBLEPARAMS);
  info->n_keydir = count_of(SOIL_TIFFTAG_GEOKEYDIRECTORY);
  info->n_ascii = count_of(SOIL_TIFFTAG_GEOASCIIPARAMS);
  info->n_metadata = count_of(SOIL_TIFFTAG_GDAL_METADATA);
  info->n_nodata = count_of(SOIL_TIFFTAG_GDAL_NODATA);
  return SOIL_OK;
  });
}

int soil_tiff_tag(const char* filename, int tag, void* dst, uint64_t capacity_bytes,
                  uint64_t* written_bytes) {
  return guarded("soil_tiff_tag", [&]() -> int {
  SOIL_REQUIRE(dst != nullptr && written_bytes != nullptr, "soil_tiff_tag: null output");
  File f;
  if (int rc = open_file(filename, f); rc != SOIL_OK) return rc;
  *written_bytes = 0;
  const Entry* e = f.find(tag);
  if (!e) return SOIL_OK;  // absent tag: zero bytes, as TIFFGetField returning 0
  if (e->type == 12) {
    const auto v = doubles(f, e);
    SOIL_REQUIRE(v.size() * 8 <= capacity_bytes, "soil_tiff_tag: buffer too small");
    std::memcpy(dst, v.data(), v.size() * 8);
    *written_bytes = v.size() * 8;
  } else if (e->type == 3 || e->type == 8) {
    SOIL_REQUIRE(e->count * 2 <= capacity_bytes, "soil_tiff_tag: buffer too small");
    uint16_t* out = static_cast<uint16_t*>(dst);
    for (uint64_t i = 0; i < e->count; ++i) out[i] = f.u16(e->data + 2 * i);
    *written_bytes = e->count * 2;
  } else {
    SOIL_REQUIRE(e->count * type_size(e->type) <= capacity_bytes, "soil_tiff_tag: buffer too small");
    std::memcpy(dst, e->data, e->count * type_size(e->type));
    *written_bytes = e->count * type_size(e->type);
  }
  return SOIL_OK;
  });
}

int soil_tiff_read(const char* filename, void* dst, uint64_t dst_bytes) {
  return guarded("soil_tiff_read", [&]() -> int {
  SOIL_REQUIRE(dst != nullptr, "soil_tiff_read: null destination");
  File f;
  if (int rc = open_file(filename, f); rc != SOIL_OK) return rc;
  Layout L;
  if (int rc = read_layout(f, L); rc != SOIL_OK) return rc;
  SOIL_REQUIRE_IO(L.samples == 1, "tiff: only single-band images are supported (tiff.hpp reads one sample per pixel)");
  SOIL_REQUIRE_IO(L.bits == 8 || L.bits == 16 || L.bits == 32 || L.bits == 64, "tiff: unsupported BitsPerSample");
  SOIL_REQUIRE_IO(L.format >= 1 && L.format <= 3, "tiff: unsupported SampleFormat");
  SOIL_REQUIRE_IO(L.predictor >= 1 && L.predictor <= 3, "tiff: unsupported Predictor");
  SOIL_REQUIRE_IO(L.width > 0 && L.height > 0, "tiff: empty image");
  SOIL_REQUIRE_IO(L.compression == 1 || L.compression == 5 || L.compression == 8 ||
                      L.compression == 32946 || L.compression == 32773,
                  "tiff: unsupported Compression (none, LZW, Deflate and PackBits are)");
  const bool wide = L.bits == 64;  // tiff.hpp:116-124: 64-bit files give FLOAT64, all others FLOAT32
  const uint64_t need = static_cast<uint64_t>(L.width) * L.height * (wide ? 8 : 4);
  SOIL_REQUIRE(dst_bytes >= need, "soil_tiff_read: destination too small");
  return wide ? decode_image(f, L, static_cast<double*>(dst)) : decode_image(f, L, static_cast<float*>(dst));
  });
}

int soil_tiff_write(const char* filename, const void* data, uint32_t width, uint32_t height,
                    uint32_t bits, const soil_geotiff_tags* geo) {
  return guarded("soil_tiff_write", [&]() -> int {
  SOIL_REQUIRE(filename != nullptr && data != nullptr, "soil_tiff_write: null argument");
  SOIL_REQUIRE(bits == 32 || bits == 64, "soil_tiff_write: bits must be 32 or 64 (IEEE float)");
  SOIL_REQUIRE(width > 0 && height > 0, "soil_tiff_write: empty image");
  const uint64_t row_bytes = static_cast<uint64_t>(width) * (bits / 8);
  const uint64_t image_bytes = row_bytes * height;
  // TIFFDefaultStripSize(tif, width) hands a request >= 1 straight back (tiff.hpp:225)
  const uint32_t rps = width;
  const uint32_t nstrips = (height + rps - 1) / rps;
  const bool big = image_bytes + (1u << 16) > 0xffffffffull;  // libtiff would fail here; BigTIFF instead

  std::vector<OutTag> tags;
  tags.push_back(tag_int(T_WIDTH, 4, width));
  tags.push_back(tag_int(T_LENGTH, 4, height));
  tags.push_back(tag_int(T_BITS, 3, bits));
  tags.push_back(tag_int(T_COMPRESSION, 3, 1));
  tags.push_back(tag_int(T_PHOTOMETRIC, 3, 1));  // MINISBLACK
  {
    OutTag t{T_STRIPOFFSETS, static_cast<uint16_t>(big ? 16 : 4), nstrips, {}};
    const uint64_t base = big ? 16 : 8;
    for (uint32_t s = 0; s < nstrips; ++s) {
      const uint64_t off = base + static_cast<uint64_t>(s) * rps * row_bytes;
      if (big) put<uint64_t>(t.payload, off); else put<uint32_t>(t.payload, static_cast<uint32_t>(off));
    }
    tags.push_back(t);
  }
  tags.push_back(tag_int(T_ORIENTATION, 3, 1));  // TOPLEFT
  tags.push_back(tag_int(T_SAMPLES, 3, 1));
  tags.push_back(tag_int(T_ROWSPERSTRIP, 4, rps));
  {
    OutTag t{T_STRIPBYTECOUNTS, static_cast<uint16_t>(big ? 16 : 4), nstrips, {}};
    for (uint32_t s = 0; s < nstrips; ++s) {
      const uint32_t rows = (height - s * rps < rps) ? height - s * rps : rps;
      const uint64_t cnt = rows * row_bytes;
      if (big) put<uint64_t>(t.payload, cnt); else put<uint32_t>(t.payload, static_cast<uint32_t>(cnt));
    }
    tags.push_back(t);
  }
  tags.push_back(tag_int(T_PLANAR, 3, 1));        // CONTIG
  tags.push_back(tag_int(T_SAMPLEFORMAT, 3, 3));  // IEEEFP
  if (geo) {  // geotiff.hpp:199-213
    auto add_doubles = [&](uint16_t tag, const double* v, uint32_t n) {
      if (!v || n == 0) return;
      OutTag t{tag, 12, n, {}};
      for (uint32_t i = 0; i < n; ++i) put<double>(t.payload, v[i]);
      tags.push_back(t);
    };
    auto add_ascii = [&](uint16_t tag, const char* s) {
      if (!s || !s[0]) return;
      OutTag t{tag, 2, std::strlen(s) + 1, {}};
      t.payload.assign(s, s + std::strlen(s) + 1);
      tags.push_back(t);
    };
    add_doubles(SOIL_TIFFTAG_GEOPIXELSCALE, geo->scale, geo->n_scale);
    add_doubles(SOIL_TIFFTAG_GEOTIEPOINTS, geo->tiepoints, geo->n_tiepoints);
    if (geo->keydir && geo->n_keydir) {
      OutTag t{SOIL_TIFFTAG_GEOKEYDIRECTORY, 3, geo->n_keydir, {}};
      for (uint32_t i = 0; i < geo->n_keydir; ++i) put<int16_t>(t.payload, geo->keydir[i]);
      tags.push_back(t);
    }
    add_doubles(SOIL_TIFFTAG_GEODOUBLEPARAMS, geo->params, geo->n_params);
    add_ascii(SOIL_TIFFTAG_GEOASCIIPARAMS, geo->ascii);
    add_ascii(SOIL_TIFFTAG_GDAL_METADATA, geo->metadata);
    add_ascii(SOIL_TIFFTAG_GDAL_NODATA, geo->nodata);
  }
  for (size_t i = 1; i < tags.size(); ++i)  // ascending tag order (TIFF 6.0 section 2)
    for (size_t j = i; j > 0 && tags[j].tag < tags[j - 1].tag; --j) std::swap(tags[j], tags[j - 1]);

  std::FILE* fp = std::fopen(filename, "wb");
  if (!fp) {
    set_error(std::string("tiff: cannot open for writing: ") + filename);
    return SOIL_ERR_IO;
  }
  std::vector<uint8_t> head;
  head.push_back('I');
  head.push_back('I');
  uint64_t ifd_off = (big ? 16 : 8) + image_bytes;
  if (ifd_off & 1) ++ifd_off;  // IFDs start on a word boundary
  if (big) {
    put<uint16_t>(head, 43);
    put<uint16_t>(head, 8);
    put<uint16_t>(head, 0);
    put<uint64_t>(head, ifd_off);
  } else {
    put<uint16_t>(head, 42);
    put<uint32_t>(head, static_cast<uint32_t>(ifd_off));
  }
  bool ok = std::fwrite(head.data(), 1, head.size(), fp) == head.size();
  ok = ok && std::fwrite(data, 1, image_bytes, fp) == image_bytes;
  if (ok && ((big ? 16 : 8) + image_bytes) & 1) ok = std::fputc(0, fp) != EOF;

  std::vector<uint8_t> ifd, extra;
  const size_t esz = big ? 20 : 12, inl = big ? 8 : 4;
  const uint64_t extra_base = ifd_off + (big ? 8 : 2) + tags.size() * esz + (big ? 8 : 4);
  if (big) put<uint64_t>(ifd, tags.size()); else put<uint16_t>(ifd, static_cast<uint16_t>(tags.size()));
  for (const OutTag& t : tags) {
    put<uint16_t>(ifd, t.tag);
    put<uint16_t>(ifd, t.type);
    if (big) put<uint64_t>(ifd, t.count); else put<uint32_t>(ifd, static_cast<uint32_t>(t.count));
    if (t.payload.size() <= inl) {
      std::vector<uint8_t> v = t.payload;
      v.resize(inl, 0);
      ifd.insert(ifd.end(), v.begin(), v.end());
    } else {
      if (extra.size() & 1) extra.push_back(0);
      const uint64_t off = extra_base + extra.size();
      if (big) put<uint64_t>(ifd, off); else put<uint32_t>(ifd, static_cast<uint32_t>(off));
      extra.insert(extra.end(), t.payload.begin(), t.payload.end());
    }
  }
  if (big) put<uint64_t>(ifd, 0); else put<uint32_t>(ifd, 0);  // no further IFD
  ok = ok && std::fwrite(ifd.data(), 1, ifd.size(), fp) == ifd.size();
  ok = ok && (extra.empty() || std::fwrite(extra.data(), 1, extra.size(), fp) == extra.size());
  ok = (std::fclose(fp) == 0) && ok;
  if (!ok) {
    set_error(std::string("tiff: write failed: ") + filename);
    return SOIL_ERR_IO;
  }
  return SOIL_OK;
  });
}

}  // extern "C"
