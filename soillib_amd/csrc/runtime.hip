// runtime.hip — error channel, device/memory/event plumbing and the silt
// element-wise ops of the C ABI (include/soil_hip.h §runtime).
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <utility>

#include "common.hpp"

namespace soil {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  std::snprintf(buf, sizeof(buf), "HIP error %d (%s) from `%s` at %s:%d", static_cast<int>(e),
                hipGetErrorString(e), what, file, line);
  g_last_error = buf;
  return e == hipErrorOutOfMemory ? SOIL_ERR_OUT_OF_MEMORY : SOIL_ERR_HIP;
}

int require_device() {
  static thread_local int ok = 0;
  if (ok) return SOIL_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(SOIL_ERR_NO_DEVICE,
                "no usable HIP device: libsoil_hip has no CPU fallback (hipGetDeviceCount -> " +
                    std::string(hipGetErrorString(e)) + ", count " + std::to_string(n) + ")");
  }
  ok = 1;
  return SOIL_OK;
}

int check_domain(const Dom& d) {
  SOIL_REQUIRE(d.H > 0 && d.W > 0, "domain: H and W must be positive");
  SOIL_REQUIRE(d.rows > 0 && d.x0 >= 0 && d.x0 + d.rows <= d.H,
               "domain: local rows must lie inside the global grid");
  SOIL_REQUIRE(0 <= d.r0 && d.r0 <= d.r1 && d.r1 <= d.rows, "domain: bad compute row range");
  // a computed row needs its x-neighbours unless they are outside the global grid
  SOIL_REQUIRE(d.r0 == d.r1 || d.r0 > 0 || d.x0 == 0,
               "domain: first computed row has no ghost row above it");
  SOIL_REQUIRE(d.r0 == d.r1 || d.r1 < d.rows || d.x0 + d.rows == d.H,
               "domain: last computed row has no ghost row below it");
  return SOIL_OK;
}

// ---- per-device workspace ------------------------------------------------------

struct WsBlock {
  void* base = nullptr;
  size_t bytes = 0;
};
static std::mutex g_ws_mutex;
// (host thread, device, slot) -> block.  A block per host thread: two threads driving one device
// (each on its own stream) do not share scratch, so their calls may overlap in time; the streams,
// events and pinned words of the launches are per thread as well (thread_local).
using WsKey = std::tuple<std::thread::id, int, int>;
static std::map<WsKey, WsBlock> g_ws;

// A host thread's blocks go when the thread does (a pool that makes a thread per request would
// otherwise leave a full set of scratch behind each time — hundreds of MB for the tiled particle
// launches).  Not for the thread that loaded the library: its thread_local destructors run at process
// exit, next to the HIP runtime's own teardown.
static const std::thread::id g_loader_thread = std::this_thread::get_id();
struct WsThreadGuard {
  ~WsThreadGuard() {
    const std::thread::id me = std::this_thread::get_id();
    if (me == g_loader_thread) return;
    int before = 0;
    if (hipGetDevice(&before) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (auto it = g_ws.begin(); it != g_ws.end();) {
      if (std::get<0>(it->first) != me) {
        ++it;
        continue;
      }
      if (it->second.base && hipSetDevice(std::get<1>(it->first)) == hipSuccess) {
        (void)hipDeviceSynchronize();  // the thread's launches may still be in flight on its streams
        (void)hipFree(it->second.base);
      }
      it = g_ws.erase(it);
    }
    (void)hipSetDevice(before);
    (void)hipGetLastError();
  }
};

int workspace_get(int slot, size_t bytes, void** out) {
  int dev = 0;
  SOIL_HIP(hipGetDevice(&dev));
  static thread_local WsThreadGuard guard;
  (void)guard;
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  WsBlock& w = g_ws[WsKey{std::this_thread::get_id(), dev, slot}];
  if (w.bytes < bytes) {
    if (w.base) {
      SOIL_HIP(hipDeviceSynchronize());  // nobody may still be reading the old block
      SOIL_HIP(hipFree(w.base));
    }
    w.base = nullptr;
    w.bytes = 0;
    SOIL_HIP(hipMalloc(&w.base, bytes));
    w.bytes = bytes;
  }
  *out = w.base;
  return SOIL_OK;
}

static std::map<int, unsigned long long*> g_step_counter;  // device -> counter

int step_counter(unsigned long long** out) {
  int dev = 0;
  SOIL_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  unsigned long long*& c = g_step_counter[dev];
  if (!c) {
    SOIL_HIP(hipMalloc(&c, sizeof(unsigned long long)));
    SOIL_HIP(hipMemset(c, 0, sizeof(unsigned long long)));
    SOIL_HIP(hipDeviceSynchronize());  // once: callers add to it from non-blocking streams
  }
  *out = c;
  return SOIL_OK;
}

int workspace_release_all() {
  int dev = 0;
  SOIL_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  for (auto it = g_ws.begin(); it != g_ws.end();) {
    if (std::get<1>(it->first) == dev) {
      if (it->second.base) {
        SOIL_HIP(hipDeviceSynchronize());
        SOIL_HIP(hipFree(it->second.base));
      }
      it = g_ws.erase(it);
    } else {
      ++it;
    }
  }
  return SOIL_OK;
}

// ---- element-wise kernels ---------------------------------------------------

template <typename T>
__global__ void k_set(T* __restrict__ t, T v, int64_t n) {
  int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n && (reinterpret_cast<uintptr_t>(t) & 15) == 0) {
    struct alignas(16) V4 {
      T a, b, c, d;
    };
    *reinterpret_cast<V4*>(t + i) = V4{v, v, v, v};
  } else {
    for (int k = 0; k < 4 && i + k < n; ++k) t[i + k] = v;
  }
}

__global__ void k_add(float* __restrict__ t, const float* __restrict__ o, int64_t n) {
  int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n && ((reinterpret_cast<uintptr_t>(t) | reinterpret_cast<uintptr_t>(o)) & 15) == 0) {
    float4 a = *reinterpret_cast<float4*>(t + i);
    const float4 b = *reinterpret_cast<const float4*>(o + i);
    a.x += b.x;
    a.y += b.y;
    a.z += b.z;
    a.w += b.w;
    *reinterpret_cast<float4*>(t + i) = a;
  } else {
    for (int k = 0; k < 4 && i + k < n; ++k) t[i + k] += o[i + k];
  }
}

__global__ void k_mul(float* __restrict__ t, float v, int64_t n) {
  int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n && (reinterpret_cast<uintptr_t>(t) & 15) == 0) {
    float4 a = *reinterpret_cast<float4*>(t + i);
    a.x *= v;
    a.y *= v;
    a.z *= v;
    a.w *= v;
    *reinterpret_cast<float4*>(t + i) = a;
  } else {
    for (int k = 0; k < 4 && i + k < n; ++k) t[i + k] *= v;
  }
}

__global__ void k_rng_seed(soil_rng* __restrict__ r, int64_t n, uint64_t seed, uint64_t offset) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) r[i] = soil_rng{seed, offset};
}

__global__ void k_selftest_math(float* __restrict__ out, const float* __restrict__ a,
                                const float* __restrict__ b, int64_t n, int op) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r = 0.0f;
  switch (op) {
    case 0: r = expf_(a[i]); break;
    case 1: r = log2f_(a[i]); break;
    case 2: r = powf_(a[i], b[i]); break;
    case 3: r = rng_uniform_at(f2bits(a[i]), static_cast<uint64_t>(i), f2bits(b[i])); break;
    case 5: r = a[i] / b[i]; break;
    case 6: r = quot0(a[i], recip(b[i])); break;
    case 7: r = expf_flat(a[i]); break;
    case 8: r = att_exp(a[i]); break;
    case 9: r = bits2f(static_cast<uint32_t>(floor_cell(a[i]))); break;
    case 10: r = sqrt_rn(a[i]); break;
    case 11: r = sqrtf(a[i]); break;
    default: r = a[i] * b[i]; break;
  }
  out[i] = r;
}

}  // namespace soil

using namespace soil;

extern "C" {

int soil_abi_version(void) { return SOIL_HIP_ABI_VERSION; }
const char* soil_last_error(void) { return g_last_error.c_str(); }

int soil_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int soil_set_device(int device) {
  SOIL_DEVICE();
  SOIL_HIP(hipSetDevice(device));
  return SOIL_OK;
}

int soil_device_name(char* buf, size_t len) {
  SOIL_DEVICE();
  SOIL_REQUIRE(buf && len > 0, "device_name: empty buffer");
  int dev = 0;
  SOIL_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  SOIL_HIP(hipGetDeviceProperties(&prop, dev));
  std::strncpy(buf, prop.gcnArchName, len - 1);
  buf[len - 1] = 0;
  return SOIL_OK;
}

void soil_param_default(soil_param* p) {  // erosion.hpp:20-56
  std::memset(p, 0, sizeof(*p));
  p->maxage = 512;
  p->lrate = 1.0f;
  p->timeStep = 250.0f;
  p->exitSlope = 0.02f;
  p->uplift = 0.001f;
  p->rainfall = 1.0f;
  p->gravity = 9.81f;
  p->evapRate = 0.0002f;
  p->frictionFactor = 0.06f;
  p->fluvialExponent = 2.0f;
  p->suspensionRateFluvial = 4.5E-8f;
  p->depositionRateFluvial = 0.04f;
  p->suspensionRateDebris = 0.001f;
  p->depositionRateDebris = 0.01f;
  p->landslideRateDebris = 0.003f;
  p->critSlopeBedrock = 0.57f;
  p->critSlopeSediment = 0.3f;
  p->yieldStress = 0.001f;
  p->viscosityWater = 1E-6f;
  p->bedShearWater = 0.0075f;
  p->densityWater = 1.0f;
  p->viscosityDebris = 0.0f;
  p->bedShearDebris = 0.99f;
  p->densityDebris = 2.0f;
}

int soil_malloc(void** ptr, size_t bytes) {
  SOIL_DEVICE();
  SOIL_REQUIRE(ptr, "malloc: null out pointer");
  *ptr = nullptr;
  if (bytes == 0) return SOIL_OK;
  SOIL_HIP(hipMalloc(ptr, bytes));
  return SOIL_OK;
}
int soil_free(void* ptr) {
  if (!ptr) return SOIL_OK;
  SOIL_DEVICE();
  SOIL_HIP(hipFree(ptr));
  return SOIL_OK;
}
int soil_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  SOIL_DEVICE();
  SOIL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  SOIL_HIP(hipStreamSynchronize(as_stream(stream)));  // pageable host memory: keep it simple
  return SOIL_OK;
}
int soil_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  SOIL_DEVICE();
  SOIL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  SOIL_HIP(hipStreamSynchronize(as_stream(stream)));
  return SOIL_OK;
}
int soil_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  SOIL_DEVICE();
  SOIL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
  return SOIL_OK;
}
int soil_stream_synchronize(void* stream) {
  SOIL_DEVICE();
  SOIL_HIP(hipStreamSynchronize(as_stream(stream)));
  return SOIL_OK;
}
int soil_device_synchronize(void) {
  SOIL_DEVICE();
  SOIL_HIP(hipDeviceSynchronize());
  return SOIL_OK;
}

int soil_event_create(void** event) {
  SOIL_DEVICE();
  SOIL_REQUIRE(event, "event_create: null out pointer");
  hipEvent_t e;
  SOIL_HIP(hipEventCreate(&e));
  *event = e;
  return SOIL_OK;
}
int soil_event_destroy(void* event) {
  if (!event) return SOIL_OK;
  SOIL_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return SOIL_OK;
}
int soil_event_record(void* event, void* stream) {
  SOIL_HIP(hipEventRecord(static_cast<hipEvent_t>(event), as_stream(stream)));
  return SOIL_OK;
}
int soil_event_elapsed_ms(void* start, void* stop, float* ms) {
  SOIL_HIP(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
  SOIL_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return SOIL_OK;
}

int soil_set_f32(float* t, float value, int64_t n, void* stream) {
  SOIL_DEVICE();
  if (n <= 0) return SOIL_OK;
  k_set<float><<<blocks_for((n + 3) / 4, 256), 256, 0, as_stream(stream)>>>(t, value, n);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}
int soil_set_i32(int32_t* t, int32_t value, int64_t n, void* stream) {
  SOIL_DEVICE();
  if (n <= 0) return SOIL_OK;
  k_set<int32_t><<<blocks_for((n + 3) / 4, 256), 256, 0, as_stream(stream)>>>(t, value, n);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}
int soil_add_f32(float* t, const float* other, int64_t n, void* stream) {
  SOIL_DEVICE();
  if (n <= 0) return SOIL_OK;
  k_add<<<blocks_for((n + 3) / 4, 256), 256, 0, as_stream(stream)>>>(t, other, n);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}
int soil_multiply_f32(float* t, float value, int64_t n, void* stream) {
  SOIL_DEVICE();
  if (n <= 0) return SOIL_OK;
  k_mul<<<blocks_for((n + 3) / 4, 256), 256, 0, as_stream(stream)>>>(t, value, n);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}
int soil_selftest_math(float* out, const float* a, const float* b, int64_t n, int op,
                       void* stream) {
  SOIL_DEVICE();
  SOIL_REQUIRE(out && a && b, "selftest_math: null argument");
  SOIL_REQUIRE(op >= 0 && op <= 11, "selftest_math: unknown op");
  if (n <= 0) return SOIL_OK;
  k_selftest_math<<<blocks_for(n, 256), 256, 0, as_stream(stream)>>>(out, a, b, n, op);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

int soil_rng_seed(soil_rng* rng, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  SOIL_DEVICE();
  if (n <= 0) return SOIL_OK;
  k_rng_seed<<<blocks_for(n, 256), 256, 0, as_stream(stream)>>>(rng, n, seed, offset);
  SOIL_LAUNCH_CHECK();
  return SOIL_OK;
}

}  // extern "C"
