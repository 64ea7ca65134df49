// When does a kernel behind hipStreamWaitEvent start?  Stream A: K1 (spins ~1 ms), event recorded,
// K2 (spins ~3 ms) queued right behind it.  Stream B waits for the event, then runs a stamp kernel.
// Prints the stamp's start relative to the end of K1 / K2, with and without a hipStreamQuery(A)
// after the record, with a timing and a timing-disabled event.
//   hipcc --offload-arch=gfx950 -O2 -o event_wait event_wait.hip && ./event_wait
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, unsigned long long* stamp) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  stamp[0] = t0;
  stamp[1] = wall_clock64();
}
__global__ void stamp_only(unsigned long long* stamp) { stamp[0] = wall_clock64(); }
int main() {
  int rate_khz = 0;
  CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  const double tick_us = 1e3 / rate_khz;
  unsigned long long* st;
  CK(hipHostMalloc(reinterpret_cast<void**>(&st), 64 * sizeof(unsigned long long), hipHostMallocDefault));
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  hipStream_t C;
  CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
  for (int reuse = 0; reuse < 2; ++reuse)
  for (int timing = 0; timing < 1; ++timing)
    for (int flush = 0; flush < 2; ++flush) {
      hipEvent_t ev;
      CK(hipEventCreateWithFlags(&ev, timing ? hipEventDefault : hipEventDisableTiming));
      for (int rep = 0; rep < 3; ++rep) {
        if (reuse) {  // the same event object first forks A and B off a third stream
          CK(hipEventRecord(ev, C));
          CK(hipStreamWaitEvent(A, ev, 0));
          CK(hipStreamWaitEvent(B, ev, 0));
        }
        spin<<<1, 1, 0, A>>>(static_cast<unsigned long long>(1000 / tick_us), st);
        CK(hipEventRecord(ev, A));
        if (flush == 1) (void)hipStreamQuery(A);
        if (flush == 2) (void)hipEventQuery(ev);
        CK(hipStreamWaitEvent(B, ev, 0));
        stamp_only<<<1, 1, 0, B>>>(st + 8);
        spin<<<1, 1, 0, A>>>(static_cast<unsigned long long>(3000 / tick_us), st + 4);
        CK(hipDeviceSynchronize());
        printf("%s%s event, %s: stamp starts %+8.1f us after K1 ended, %+8.1f us relative to K2's end\n",
               reuse ? "reused " : "", timing ? "timing" : "no-timing", flush == 0 ? "no flush      " : flush == 1 ? "hipStreamQuery" : "hipEventQuery ",
               (static_cast<double>(st[8]) - static_cast<double>(st[1])) * tick_us,
               (static_cast<double>(st[8]) - static_cast<double>(st[5])) * tick_us);
      }
      CK(hipEventDestroy(ev));
    }
  return 0;
}
