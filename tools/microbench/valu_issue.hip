// Microbenchmark: issue cost of the vector instructions the particle step is made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue && ./valu_issue
//
// Every kernel runs `iters` iterations of a block of 64 independent instructions of one kind
// (8 accumulator chains, so dependent latency never shows) in W waves per SIMD on every CU and
// reports shader cycles (s_memtime) per wave-instruction per SIMD: elapsed / (W * instructions).
// That number is the denominator of the VALU-issue roofline of k_tiled_round.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x

// one block = 8 x 8 instructions; operands %0..%7 accumulators, %8/%9 inputs
#define OP8(op)                                   \
  op " %0, %8, %9, %0\n" op " %1, %8, %9, %1\n"   \
  op " %2, %8, %9, %2\n" op " %3, %8, %9, %3\n"   \
  op " %4, %8, %9, %4\n" op " %5, %8, %9, %5\n"   \
  op " %6, %8, %9, %6\n" op " %7, %8, %9, %7\n"
#define OP8_2(op)                           \
  op " %0, %8, %0\n" op " %1, %8, %1\n"     \
  op " %2, %8, %2\n" op " %3, %8, %3\n"     \
  op " %4, %8, %4\n" op " %5, %8, %5\n"     \
  op " %6, %8, %6\n" op " %7, %8, %7\n"
#define OP8_1(op)                     \
  op " %0, %0\n" op " %1, %1\n"       \
  op " %2, %2\n" op " %3, %3\n"       \
  op " %4, %4\n" op " %5, %5\n"       \
  op " %6, %6\n" op " %7, %7\n"

enum Mode {
  FMA, MUL, ADD, PKFMA, PKMUL, PKADD, RCP, SQRT, EXP, LOG, RSQ, LDEXP, FLOOR, CVT_I32, MAD_U24, CNDMASK, CMP,
  DIV_SCALE, DIV_FMAS, DIV_FIXUP, MAX, MED3, MOV, DPP_ADD, READLANE, AND, MODE_COUNT
};
static const char* kNames[] = {
  "v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32 (2 fp32 per lane)", "v_pk_mul_f32", "v_pk_add_f32",
  "v_rcp_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_rsq_f32", "v_ldexp_f32", "v_floor_f32",
  "v_cvt_i32_f32", "v_mad_u32_u24", "v_cndmask_b32 (vcc)", "v_cmp_lt_f32 (vcc)", "v_div_scale_f32",
  "v_div_fmas_f32", "v_div_fixup_f32", "v_max_f32", "v_med3_f32", "v_mov_b32", "v_add_f32 dpp row_shr:1",
  "v_readlane_b32", "v_and_b32"};

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* cyc, int iters, float x, float y) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
        a6 = a0 + 6, a7 = a0 + 7;
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2},
          p6 = {a5, a4}, p7 = {a7, a6}, px = {x, y}, py = {y, x};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == FMA) asm volatile(REP8(OP8("v_fma_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if constexpr (MODE == MED3) asm volatile(REP8(OP8("v_med3_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if constexpr (MODE == DIV_FIXUP) asm volatile(REP8(OP8("v_div_fixup_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if constexpr (MODE == DIV_FMAS) asm volatile(REP8(OP8("v_div_fmas_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc");
    if constexpr (MODE == MAD_U24) asm volatile(REP8(OP8("v_mad_u32_u24")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
    if constexpr (MODE == PKFMA) asm volatile(REP8(OP8("v_pk_fma_f32")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(px), "v"(py));
    if constexpr (MODE == PKMUL) asm volatile(REP8(OP8_2("v_pk_mul_f32")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(px));
    if constexpr (MODE == PKADD) asm volatile(REP8(OP8_2("v_pk_add_f32")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(px));
    if constexpr (MODE == MUL) asm volatile(REP8(OP8_2("v_mul_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
    if constexpr (MODE == ADD) asm volatile(REP8(OP8_2("v_add_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
    if constexpr (MODE == MAX) asm volatile(REP8(OP8_2("v_max_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
    if constexpr (MODE == AND) asm volatile(REP8(OP8_2("v_and_b32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
    if constexpr (MODE == LDEXP) asm volatile(REP8(OP8_2("v_ldexp_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0));
    if constexpr (MODE == CNDMASK) asm volatile(REP8("v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n v_cndmask_b32 %3, %8, %3, vcc\n v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x) : "vcc");
    if constexpr (MODE == CMP) asm volatile(REP8("v_cmp_lt_f32 vcc, %8, %0\n v_cmp_lt_f32 vcc, %8, %1\n v_cmp_lt_f32 vcc, %8, %2\n v_cmp_lt_f32 vcc, %8, %3\n v_cmp_lt_f32 vcc, %8, %4\n v_cmp_lt_f32 vcc, %8, %5\n v_cmp_lt_f32 vcc, %8, %6\n v_cmp_lt_f32 vcc, %8, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x) : "vcc");
    if constexpr (MODE == DIV_SCALE) asm volatile(REP8("v_div_scale_f32 %0, vcc, %8, %8, %0\n v_div_scale_f32 %1, vcc, %8, %8, %1\n v_div_scale_f32 %2, vcc, %8, %8, %2\n v_div_scale_f32 %3, vcc, %8, %8, %3\n v_div_scale_f32 %4, vcc, %8, %8, %4\n v_div_scale_f32 %5, vcc, %8, %8, %5\n v_div_scale_f32 %6, vcc, %8, %8, %6\n v_div_scale_f32 %7, vcc, %8, %8, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x) : "vcc");
    if constexpr (MODE == DPP_ADD) asm volatile(REP8("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (MODE == READLANE) {
      int s;
      asm volatile(REP8("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %2, 5\n v_readlane_b32 %0, %3, 7\n v_readlane_b32 %0, %4, 9\n v_readlane_b32 %0, %1, 11\n v_readlane_b32 %0, %2, 13\n v_readlane_b32 %0, %3, 15\n v_readlane_b32 %0, %4, 17\n") : "=s"(s) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
      a0 += s * 1e-30f;
    }
    if constexpr (MODE == RCP) asm volatile(REP8(OP8_1("v_rcp_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (MODE == RSQ) asm volatile(REP8(OP8_1("v_rsq_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (MODE == SQRT) asm volatile(REP8(OP8_1("v_sqrt_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (MODE == EXP) asm volatile(REP8(OP8_1("v_exp_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (MODE == LOG) asm volatile(REP8(OP8_1("v_log_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (MODE == FLOOR) asm volatile(REP8(OP8_1("v_floor_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (MODE == CVT_I32) asm volatile(REP8(OP8_1("v_cvt_i32_f32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    if constexpr (MODE == MOV) asm volatile(REP8(OP8_1("v_mov_b32")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if constexpr (MODE == PKFMA || MODE == PKMUL || MODE == PKADD)
    a0 = p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// s_memtime ticks at a constant 100 MHz on some parts and at the shader clock on others: the wall time
// of the launch and the tick count together say which, and give the effective shader clock.
template <int MODE>
void run(int waves_per_simd) {
  const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
  const int blocks_per_cu = 256 * waves_per_simd / threads;
  const int blocks = 256 * blocks_per_cu, iters = 4000;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<MODE><<<blocks, threads>>>(out, cyc, 10, 1.0000001f, 1e-9f);
  hipEventRecord(a);
  k<MODE><<<blocks, threads>>>(out, cyc, iters, 1.0000001f, 1e-9f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double ticks = 0;
  for (auto t : h) ticks += t;
  ticks /= blocks;
  const double instr_per_wave = 64.0 * iters;
  // wall-clock view: ns per wave-instruction per SIMD
  const double ns = ms * 1e6 / (instr_per_wave * waves_per_simd);
  printf("%-34s W=%d  %8.3f ms  %6.3f ns/instr/SIMD  (= %5.2f cycles @2.4GHz, %5.2f @2.1GHz)  ticks/instr %6.3f\n",
         kNames[MODE], waves_per_simd, ms, ns, ns * 2.4, ns * 2.1, ticks / (instr_per_wave * waves_per_simd));
  hipFree(out);
  hipFree(cyc);
}

template <int M>
void all() {
  if constexpr (M < MODE_COUNT) {
    run<M>(4);
    if (M == FMA || M == PKFMA || M == RCP || M == MUL) { run<M>(1); run<M>(2); run<M>(8); }
    all<M + 1>();
  }
}

int main() {
  all<0>();
  return 0;
}
