// Issue-rate probe for gfx950: how many cycles a wave's VALU / transcendental / MFMA instruction takes, and whether the
// MFMAs of one wave overlap with the VALU work of another wave on the same SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/issue_probe.hip -o issue_probe && ./issue_probe
// One workgroup of 8 waves: waves w and w + 4 share a SIMD.  Mode bits: 1 = waves 0-3 run the MFMA loop, 2 = waves 4-7
// run the VALU loop `kind`.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int ITER = 2000;

template <int kind>
__global__ __launch_bounds__(512) void probe(int mode, unsigned long long* out, float* sink) {
  const int wid = threadIdx.x >> 6;
  const bool do_mfma = (mode & 1) && wid < 4, do_valu = (mode & 2) && wid >= 4;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  if (do_mfma) {
    bf16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)1.f; }
    if (kind < 100) {
      f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      for (int i = 0; i < ITER; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
      r = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
      f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
      for (int i = 0; i < ITER; i++) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
      }
      r = c0[0] + c1[1] + c2[2] + c3[3];
    }
  }
  if (do_valu) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    constexpr int k = kind % 100;
    for (int i = 0; i < ITER; i++) {
#pragma unroll
      for (int rep = 0; rep < 4; rep++)
      if (k == 0) {   // 8 x v_exp_f32
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (k == 1) {   // 8 x v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %4\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %0"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (k == 2) {   // 4 x v_pk_fma_f32 (8 values)
        f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
        asm volatile("v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %1, %1, %1, %2\n v_pk_fma_f32 %2, %2, %2, %3\n v_pk_fma_f32 %3, %3, %3, %0\n"
                     "v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %1, %1, %1, %2\n v_pk_fma_f32 %2, %2, %2, %3\n v_pk_fma_f32 %3, %3, %3, %0"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        x0 = p0[0]; x1 = p0[1]; x2 = p1[0]; x3 = p1[1]; x4 = p2[0]; x5 = p2[1]; x6 = p3[0]; x7 = p3[1];
      } else if (k == 3) {   // 8 x v_max3_f32
        asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (k == 4) {   // 8 x v_cvt_pk_bf16_f32
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      } else if (k == 5) {   // 8 x v_pk_mul_f32 (16 values)
        f32x2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
        asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0\n"
                     "v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        x0 = p0[0]; x1 = p0[1]; x2 = p1[0]; x3 = p1[1]; x4 = p2[0]; x5 = p2[1]; x6 = p3[0]; x7 = p3[1];
      } else {   // 8 x v_rcp_f32
        asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
      }
    }
    r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[wid] = t1 - t0;
  if (r == 12345.678f) sink[threadIdx.x] = r;
}

template <int K> static void launch1(int mode, unsigned long long* out, float* sink) { hipLaunchKernelGGL(probe<K>, dim3(1), dim3(512), 0, 0, mode, out, sink); }
static void launch(int kind, int mode, unsigned long long* out, float* sink) {
  switch (kind) {
#define C(K) case K: launch1<K>(mode, out, sink); break;
    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(100) C(101) C(102) C(103) C(104) C(105) C(106)
#undef C
  }
}
int main() {
  unsigned long long* out;
  float* sink;
  hipMalloc(&out, 64);
  hipMalloc(&sink, 4096);
  const char* names[] = {"v_exp_f32 x8", "v_fma_f32 x8", "v_pk_fma_f32 x8", "v_max3_f32 x8", "v_cvt_pk_bf16_f32 x8", "v_pk_mul_f32 x8", "v_rcp_f32 x8"};
  // s_memtime counts at a fixed 100 MHz; convert with the shader clock measured from a known-latency loop instead:
  // report raw ticks per iteration and the ratio between modes.
  for (int mf = 0; mf < 2; mf++)
    for (int kind = 0; kind < 7; kind++) {
      unsigned long long h[3][8];
      for (int mode = 1; mode <= 3; mode++) {
        launch(kind + 100 * mf, mode, out, sink);
        launch(kind + 100 * mf, mode, out, sink);
        hipMemcpy(h[mode - 1], out, 64, hipMemcpyDeviceToHost);
      }
      printf("%s  %-22s  ticks/iter: mfma-only(4 mfma) %.3f  valu-only(32 ops) %.3f  both: mfma-wave %.3f valu-wave %.3f\n",
             mf ? "16x16x32" : "32x32x16", names[kind], (double)h[0][0] / ITER, (double)h[1][4] / ITER, (double)h[2][0] / ITER, (double)h[2][4] / ITER);
    }
  return 0;
}
