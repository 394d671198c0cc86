// Does VALU work overlap with MFMAs on one SIMD of gfx950?  One workgroup of 8 waves (waves w and w + 4 share a SIMD).
//   cross<AK, BK>: waves 0-3 run stream AK (0 nothing, 1 back-to-back 32x32x16 MFMAs, 2 back-to-back 16x16x32 MFMAs),
//                  waves 4-7 run stream BK (0 nothing, 1 32 independent v_fma_f32, 2 32 independent v_exp_f32)
//   mixed<NV, W2>: a wave runs 4 x (one 32x32x16 MFMA + NV independent v_fma_f32); W2: both waves of the SIMD run it
//   hipcc --offload-arch=gfx950 -O2 tools/probes/overlap_probe.hip -o overlap_probe && ./overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int ITER = 2000;
#define MF(c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define MF16(c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define FMA8 asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]))
#define EXP8 asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]))
#define FMA1(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]))

#define PROLOGUE                                                                                   \
  const int wid = threadIdx.x >> 6;                                                                \
  bf16x8 a, b;                                                                                     \
  for (int j = 0; j < 8; j++) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)1.f; }     \
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};                                                   \
  f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};                                                    \
  float x[8];                                                                                      \
  for (int j = 0; j < 8; j++) x[j] = threadIdx.x * 1e-3f + j;                                      \
  __syncthreads();                                                                                 \
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define EPILOGUE                                                                                   \
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
  if ((threadIdx.x & 63) == 0) out[wid] = t1 - t0;                                                 \
  float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[1] + d2[2] + d3[3];                         \
  for (int j = 0; j < 8; j++) s += x[j];                                                           \
  if (s == 12345.678f) sink[threadIdx.x] = s;

template <int AK, int BK, int PRIO>
__global__ __launch_bounds__(512) void cross(unsigned long long* out, float* sink) {
  PROLOGUE
  if (wid < 4) {
    if (AK == 1) for (int i = 0; i < ITER; i++) { MF(c0); MF(c1); MF(c2); MF(c3); }
    if (AK == 2) for (int i = 0; i < ITER; i++) { MF16(d0); MF16(d1); MF16(d2); MF16(d3); MF16(d0); MF16(d1); MF16(d2); MF16(d3); }
  } else {
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    if (BK == 1) for (int i = 0; i < ITER; i++) { FMA8; FMA8; FMA8; FMA8; }
    if (BK == 2) for (int i = 0; i < ITER; i++) { EXP8; EXP8; EXP8; EXP8; }
  }
  EPILOGUE
}

template <int NV, int W2>
__global__ __launch_bounds__(512) void mixed(unsigned long long* out, float* sink) {
  PROLOGUE
  if (wid < 4 || W2) {
    for (int i = 0; i < ITER; i++) {
#define STEP(c) MF(c); _Pragma("unroll") for (int r = 0; r < NV; r++) FMA1(r & 7);
      STEP(c0) STEP(c1) STEP(c2) STEP(c3)
    }
  }
  EPILOGUE
}

static unsigned long long* out;
static float* sink;
template <class F> static void report(const char* what, F launch) {
  unsigned long long h[8];
  launch(); launch();
  (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
  printf("%-62s wave0 %.1f  wave4 %.1f cycles/iter\n", what, (double)h[0] / ITER, (double)h[4] / ITER);
}
#define CROSS(AK, BK, PR, what) report(what, [] { hipLaunchKernelGGL((cross<AK, BK, PR>), dim3(1), dim3(512), 0, 0, out, sink); })
#define MIXED(NV, W2, what) report(what, [] { hipLaunchKernelGGL((mixed<NV, W2>), dim3(1), dim3(512), 0, 0, out, sink); })

int main() {
  (void)hipMalloc(&out, 64);
  (void)hipMalloc(&sink, 4096);
  CROSS(1, 0, 0, "A: 4 MFMA32 alone");
  CROSS(2, 0, 0, "A: 8 MFMA16 alone");
  CROSS(0, 1, 0, "B: 32 v_fma alone");
  CROSS(0, 2, 0, "B: 32 v_exp alone");
  CROSS(1, 1, 0, "A: 4 MFMA32 | B: 32 v_fma");
  CROSS(1, 1, 1, "A: 4 MFMA32 | B: 32 v_fma at priority 3");
  CROSS(1, 2, 0, "A: 4 MFMA32 | B: 32 v_exp");
  CROSS(2, 1, 0, "A: 8 MFMA16 | B: 32 v_fma");
  CROSS(2, 2, 0, "A: 8 MFMA16 | B: 32 v_exp");
  MIXED(0, 0, "one wave: 4 x (MFMA32)");
  MIXED(2, 0, "one wave: 4 x (MFMA32 + 2 v_fma)");
  MIXED(4, 0, "one wave: 4 x (MFMA32 + 4 v_fma)");
  MIXED(6, 0, "one wave: 4 x (MFMA32 + 6 v_fma)");
  MIXED(7, 0, "one wave: 4 x (MFMA32 + 7 v_fma)");
  MIXED(8, 0, "one wave: 4 x (MFMA32 + 8 v_fma)");
  MIXED(12, 0, "one wave: 4 x (MFMA32 + 12 v_fma)");
  MIXED(4, 1, "two waves: 4 x (MFMA32 + 4 v_fma) each");
  MIXED(8, 1, "two waves: 4 x (MFMA32 + 8 v_fma) each");
  MIXED(12, 1, "two waves: 4 x (MFMA32 + 12 v_fma) each");
  return 0;
}
