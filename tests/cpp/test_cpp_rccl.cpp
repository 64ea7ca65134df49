// The library's RCCL wire (soil_comm_rccl_*, csrc/slab_runner.hip 3) from a bare C++ process — no
// Python, no torch: the configuration INTEGRATION.md 5 gives a C++ host.  One rank on one GPU:
//   library   which librccl got bound, and its version
//   slab      three steps of the slab runner over a one-rank RCCL communicator (all-reduces, barrier)
//   self      a second communicator in the same process exchanging with ITSELF (ncclSend / ncclRecv to
//             one's own rank inside a group are legal): one pair at BASELINE config 5's fluvial flux
//             halo (250 rows x 16384 cells x 16 B = 65.5 MB), bytes checked
//   group     four transfers of unequal sizes in one group, gaps between the destinations untouched
//   reduce    ncclAllReduce of a world of one
//   absent    (only when named) a world of two whose second rank never shows up: soil_comm_rccl_create
//             must come back with SOIL_ERR_COMM after SOIL_RCCL_INIT_TIMEOUT_S instead of waiting for ever
//   stall     (only when named) a transfer that cannot complete — the stream is held by a
//             hipStreamWaitValue32 in front of it — must be aborted by the library's watchdog after
//             SOIL_RCCL_TIMEOUT_S: status / the next call return SOIL_ERR_COMM naming the operation
// Stages named on the command line run alone (diagnosis); none = all but the last two.  Every stage prints a marker
// first; the watchdog of watchdog.hpp ends a hung binary with the marker it hung in.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <soil.hpp>
#include <string>

#include <hip/hip_runtime_api.h>  // the stall stage holds a stream with hipStreamWaitValue32

#include "watchdog.hpp"

using silt::check;

#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
  start_watchdog(75.0);
  auto want = [&](const char* stage) {
    if (argc < 2) return true;
    for (int i = 1; i < argc; ++i) if (std::strcmp(argv[i], stage) == 0) return true;
    return false;
  };
  mark("device count");
  if (soil_device_count() == 0) { std::printf("NO_DEVICE_OK\n"); return 0; }

  mark("library");
  {
    char path[512] = "";
    int32_t v = 0;
    check(soil_comm_rccl_library(path, sizeof path, &v));
    std::printf("RCCL_LIB %s version %d.%d.%d\n", path, v / 10000, v / 100 % 100, v % 100);
  }
  auto named = [&](const char* stage) { return argc >= 2 && want(stage); };
  if (named("absent")) {
    mark("absent: ncclCommInitRank of a world of two, alone");
    const auto id = soil::comm::rccl_unique_id();
    soil_comm* c = nullptr;
    const double t0 = since_start();
    const int rc = soil_comm_rccl_create(&c, id.data(), 0, 2);
    std::printf("RCCL_ABSENT rc %d after %.1f s: %s\n", rc, since_start() - t0, soil_last_error());
    EXPECT(rc == SOIL_ERR_COMM && c == nullptr);
    EXPECT(std::strstr(soil_last_error(), "ncclCommInitRank") && std::strstr(soil_last_error(), "librccl"));
    std::printf("CPP_RCCL_OK\n");
    std::fflush(stdout);
    _exit(0);  // (a helper thread is still inside ncclCommInitRank: no orderly exit from here)
  }
  if (named("stall")) {
    mark("stall: communicator");
    soil::comm wire = soil::comm::rccl(soil::comm::rccl_unique_id(), 0, 1);
    const soil_comm* c = wire.get();
    const size_t bytes = 1 << 20;
    float *src = nullptr, *dst = nullptr;
    uint32_t* flag = nullptr;
    check(soil_malloc(reinterpret_cast<void**>(&src), bytes));
    check(soil_malloc(reinterpret_cast<void**>(&dst), bytes));
    check(soil_malloc(reinterpret_cast<void**>(&flag), 64));
    check(soil_set_f32(reinterpret_cast<float*>(flag), 0.0f, 16, nullptr));
    check(soil_device_synchronize());
    hipStream_t held = nullptr, side = nullptr;
    EXPECT(hipStreamCreateWithFlags(&held, hipStreamNonBlocking) == hipSuccess);
    EXPECT(hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess);
    mark("stall: hipStreamWaitValue32 holds the stream");
    if (hipStreamWaitValue32(held, flag, 1, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) {
      std::printf("RCCL_STALL skipped: hipStreamWaitValue32 is not available here\nCPP_RCCL_OK\n");
      std::fflush(stdout);
      _exit(0);
    }
    mark("stall: exchange behind the held stream");
    const soil_xfer s{src, int64_t(bytes), 0}, r{dst, int64_t(bytes), 0};
    EXPECT(c->exchange(c->ctx, &s, 1, &r, 1, held) == SOIL_OK);  // stream-ordered: returns at once
    const double t0 = since_start();
    mark("stall: waiting for the watchdog");
    int st = SOIL_OK;
    while ((st = c->status(c->ctx)) == SOIL_OK && since_start() - t0 < 20.0)
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    const double took = since_start() - t0;
    std::printf("RCCL_STALL status %d after %.2f s: %s\n", st, took, soil_last_error());
    EXPECT(st == SOIL_ERR_COMM);
    EXPECT(std::strstr(soil_last_error(), "no completion of exchange") && std::strstr(soil_last_error(), "1 sends") &&
           std::strstr(soil_last_error(), "librccl") && std::strstr(soil_last_error(), "aborted"));
    mark("stall: the communicator stays failed");
    EXPECT(c->exchange(c->ctx, &s, 1, &r, 1, held) == SOIL_ERR_COMM);
    EXPECT(c->all_reduce_sum_f32(c->ctx, dst, 4, held) == SOIL_ERR_COMM);
    mark("stall: release the stream");
    const uint32_t one = 1;
    EXPECT(hipMemcpyAsync(flag, &one, 4, hipMemcpyHostToDevice, side) == hipSuccess);
    EXPECT(hipStreamSynchronize(side) == hipSuccess);
    EXPECT(hipStreamSynchronize(held) == hipSuccess);
    std::printf("CPP_RCCL_OK\n");
    std::fflush(stdout);
    _exit(0);  // (the communicator was aborted: leave without the libraries' exit handlers)
  }
  if (want("slab")) {
    mark("slab: unique id");
    const auto id = soil::comm::rccl_unique_id();
    mark("slab: ncclCommInitRank (world of one)");
    soil::comm wire = soil::comm::rccl(id, 0, 1);
    mark("slab: runner create");
    const int S = 96;
    soil::param_t sp;
    sp.maxage = 64; sp.timeStep = 1000.0f; sp.critSlopeBedrock = 0.57f; sp.suspensionRateFluvial = 0.0008f;
    soil::slab_runner slab(soil::slab_runner::config(S, S), sp, wire);
    uint64_t before = 0, after = 0;
    check(soil_particle_steps(&before, 1, nullptr));
    mark("slab: three steps");
    for (int s = 0; s < 3; ++s) slab.step();
    slab.sync();
    check(soil_particle_steps(&after, 1, nullptr));
    EXPECT(slab.info().step_index == 3 && slab.info().world == 1 && slab.info().rows == S);
    double sum = 0;
    const std::vector<float> lay = slab.owned_rows("layers");
    for (size_t i = 0; i < lay.size(); i += 2) sum += lay[i];
    std::printf("SLAB1 %llu %.9e\n", static_cast<unsigned long long>(after), sum);
    mark("slab: destroy");
  }
  if (want("self") || want("group") || want("reduce")) {
    mark("wire: unique id");
    const auto id = soil::comm::rccl_unique_id();
    mark("wire: ncclCommInitRank (world of one)");
    soil::comm wire = soil::comm::rccl(id, 0, 1);
    int32_t n = 0, r = -1, dev = -1;
    check(soil_comm_rccl_info(wire.get(), &n, &r, &dev));
    EXPECT(n == 1 && r == 0 && dev >= 0);
    const size_t big = size_t(250) * 16384 * 16, words = big / 4;
    float *src = nullptr, *dst = nullptr;
    mark("wire: buffers");
    check(soil_malloc(reinterpret_cast<void**>(&src), big));
    check(soil_malloc(reinterpret_cast<void**>(&dst), big));
    std::vector<float> pat(words), back(words);
    for (size_t i = 0; i < words; ++i) pat[i] = float(i % 8191) - 4000.0f;
    check(soil_memcpy_h2d(src, pat.data(), big, nullptr));
    check(soil_set_f32(dst, -1.0f, int64_t(words), nullptr));
    check(soil_device_synchronize());
    size_t bad = 0;
    if (want("self")) {
      void *e0 = nullptr, *e1 = nullptr;
      check(soil_event_create(&e0));
      check(soil_event_create(&e1));
      float ms_first = 0, ms = 0;
      for (int rep = 0; rep < 4; ++rep) {  // the first call sets the channels up
        mark(rep == 0 ? "self: first 65.5 MB exchange with oneself" : "self: exchange again");
        check(soil_event_record(e0, nullptr));
        wire.exchange({soil_xfer{src, int64_t(big), 0}}, {soil_xfer{dst, int64_t(big), 0}});
        check(soil_event_record(e1, nullptr));
        mark("self: waiting for the stream");
        check(soil_event_elapsed_ms(e0, e1, rep == 0 ? &ms_first : &ms));
      }
      mark("self: copy back");
      check(soil_memcpy_d2h(back.data(), dst, big, nullptr));
      for (size_t i = 0; i < words; ++i) bad += back[i] != pat[i];
      EXPECT(bad == 0);
      std::printf("RCCL_SELF ranks %d bytes %zu first_ms %.3f ms %.3f GBps %.1f\n", n, big, ms_first, ms,
                  double(big) / (double(ms) * 1e6));
      check(soil_event_destroy(e0));
      check(soil_event_destroy(e1));
    }
    if (want("group")) {
      // four transfers in one group, unequal sizes, sources and destinations interleaved in one block
      mark("group: four transfers in one group");
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
      mark("group: copy back");
      check(soil_memcpy_d2h(back.data(), dst, big, nullptr));
      so = 0, ro = 0;
      for (int k = 0; k < 4; ++k) {  // the k-th receive holds the k-th send (matched in order), the gaps are untouched
        for (size_t i = 0; i < len[k]; ++i) bad += back[ro + i] != pat[so + i];
        for (size_t i = 0; i < 256; ++i) bad += back[ro + len[k] + i] != -1.0f;
        so += len[k] + 64, ro += len[k] + 256;
      }
      EXPECT(bad == 0);
      std::printf("RCCL_GROUP ok\n");
    }
    if (want("reduce")) {
      mark("reduce: ncclAllReduce of four floats");
      float one[4] = {1.5f, -2.0f, 0.25f, 8.0f};
      check(soil_memcpy_h2d(dst, one, 16, nullptr));
      wire.all_reduce_sum(dst, 4);
      check(soil_memcpy_d2h(back.data(), dst, 16, nullptr));
      EXPECT(back[0] == 1.5f && back[1] == -2.0f && back[2] == 0.25f && back[3] == 8.0f);  // a world of one sums to itself
      std::printf("RCCL_REDUCE ok\n");
    }
    mark("wire: release");
    check(soil_free(src));
    check(soil_free(dst));
  }
  std::printf("CPP_RCCL_OK\n");
  mark(kExitMarker);
  return 0;
}
