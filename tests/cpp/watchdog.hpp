// Markers and a watchdog for the C++ test binaries: stdout is line-buffered (a killed child still
// leaves what it printed in the pipe), every block announces itself, and a binary that sits for
// longer than its budget prints the last marker and ends itself with exit code 3 — so that a hang
// names its place instead of costing the session its subprocess timeout.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>

namespace {
const char* const kExitMarker = "process exit (static destructors of the ROCm libraries)";
std::atomic<const char*> g_marker{"start"};
std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();

inline double since_start() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - g_t0).count(); }
inline void mark(const char* what) {
  g_marker.store(what);
  std::printf("MARK %.3f %s\n", since_start(), what);
  std::fflush(stdout);
}
inline void start_watchdog(double seconds) {
  std::setvbuf(stdout, nullptr, _IOLBF, 0);
  if (const char* e = std::getenv("SOIL_TEST_WATCHDOG_S")) seconds = std::atof(e);
  std::thread([seconds] {
    while (since_start() < seconds) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    const char* at = g_marker.load();
    std::printf("WATCHDOG %.1f s: still in \"%s\"\n", seconds, at);
    std::fflush(stdout);
    // every check has passed and printed its OK line by then: a process that cannot get through the ROCm
    // libraries' exit handlers is reported (EXIT_HUNG), not counted against the code under test
    if (std::strcmp(at, kExitMarker) == 0) {
      std::printf("EXIT_HUNG\n");
      std::fflush(stdout);
      _exit(0);
    }
    _exit(3);
  }).detach();
}
}  // namespace
