// Shared device/host helpers for libvitk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/vitk.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- error reporting (thread local; never abort) --------------------------------------------
int vitk_set_error(const char* fmt, ...);
#define VITK_REQUIRE(cond, ...)                         \
  do {                                                  \
    if (!(cond)) return vitk_set_error(__VA_ARGS__);    \
  } while (0)
#define VITK_LAUNCH_CHECK(name)                                                           \
  do {                                                                                    \
    hipError_t e_ = hipGetLastError();                                                    \
    if (e_ != hipSuccess) return vitk_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
  } while (0)

// ---- scalar conversions ------------------------------------------------------------------------
template <class T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return (float)v; }
template <class T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32: RNE, NaN-safe
__device__ __forceinline__ float round_bf16(float v) { return (float)((bf16)v); }
// One output element of a row normalisation, y = (x - mean) * rstd * w (+ b), with the products and the sum rounded one
// by one, and the sum of squares of a 4-element group as an fma chain: the row kernels and the fused per-head
// normalisation of the QKV projection's epilogue use these, in the same order, so that both produce the same bits.
__device__ __forceinline__ float norm_apply(float c, float rstd, float w, float b, bool has_b) {
  const float t = __fmul_rn(__fmul_rn(c, rstd), w);
  return has_b ? __fadd_rn(t, b) : t;
}
__device__ __forceinline__ float sumsq4(float s, float c0, float c1, float c2, float c3) {
  s = __fmaf_rn(c0, c0, s);
  s = __fmaf_rn(c1, c1, s);
  s = __fmaf_rn(c2, c2, s);
  return __fmaf_rn(c3, c3, s);
}
// One output element of apply_rope (vit/rope.py:148-163), x * cos + rotate_half(x) * sin, the way eager elementwise ops
// compute it: two rounded products and a rounded sum, never contracted into an fma, so that every kernel that rotates
// (the stand-alone pass, the projection epilogue, the attention backward stores) produces the same bits.
// emu: the products are rounded to bf16 before the sum (a bf16 table in the reference).
__device__ __forceinline__ float rope_mix(float x, float c, float y, float s, int emu) {
  float a = __fmul_rn(x, c), b = __fmul_rn(y, s);
  if (emu) { a = round_bf16(a); b = round_bf16(b); }
  return __fadd_rn(a, b);
}
// value as the dtype T would hold it
template <class T> __device__ __forceinline__ float round_as(float v) { return to_f<T>(from_f<T>(v)); }

// ---- 4-wide vector access ---------------------------------------------------------------------
template <class T> struct Vec4;
template <> struct Vec4<float> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) { *(float4*)v = *(const float4*)p; }
  __device__ __forceinline__ void store(float* p) const { *(float4*)p = *(const float4*)v; }
};
template <> struct Vec4<bf16> {
  float v[4];
  __device__ __forceinline__ void load(const bf16* p) {
    bf16x4 t = *(const bf16x4*)p;
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
  }
  __device__ __forceinline__ void store(bf16* p) const {
    bf16x4 t;
    t[0] = (bf16)v[0]; t[1] = (bf16)v[1]; t[2] = (bf16)v[2]; t[3] = (bf16)v[3];
    *(bf16x4*)p = t;
  }
};

// V-wide access, V = 4 or 8 (8 for bf16 only: one 16-byte access per lane, the width at which a CU streams fastest)
template <class T, int V> struct VecN;
template <class T> struct VecN<T, 4> : Vec4<T> {};
template <> struct VecN<bf16, 8> {
  float v[8];
  __device__ __forceinline__ void load(const bf16* p) {
    const bf16x8 t = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (float)t[i];
  }
  __device__ __forceinline__ void store(bf16* p) const {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = (bf16)v[i];
    *(bf16x8*)p = t;
  }
};

// ---- wave reductions (wave = 64 lanes) ---------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- counter-based dropout stream --------------------------------------------------------------
// The keep decision of element idx is a function of (seed, idx) only, so forward and backward regenerate the same
// mask.  One 32-bit hash serves the two elements of a pair (idx >> 1): element 2i takes its low half, element 2i + 1 its
// high half, each as the top 16 bits of the draw that is compared with the threshold p * 2^32 (the keep probability is
// then exact to 2^-16).  An element costs half a hash that way, and the integer multiplies of the hash are what a
// dropout epilogue is made of (they issue at a quarter of the vector rate).
// The hash input is LINEAR in the pair number, x = pair * RNG_C1 + mix(seed) in 32-bit arithmetic, so a kernel that walks
// a 2-D index space (row * stride + column) forms it from a per-row and a per-column term with one add per draw
// (rng_lin_*), with no 64-bit arithmetic and no multiply in front of the finaliser; pair numbers 2^32 apart share draws.
constexpr uint32_t RNG_C1 = 0x9E3779B1u;
__device__ __forceinline__ uint32_t rng_seedmix(uint64_t seed) {   // wave-uniform: scalar code
  return (uint32_t)seed * 0x85EBCA77u + (uint32_t)(seed >> 32) * 0xC2B2AE3Du + 0x165667B1u;
}
__device__ __forceinline__ uint32_t rng_fmix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t rng_pair(uint64_t seed, uint64_t pair) {
  return rng_fmix((uint32_t)pair * RNG_C1 + rng_seedmix(seed));
}
__device__ __forceinline__ uint32_t rng_half(uint32_t h, int odd) { return odd ? (h & 0xffff0000u) : (h << 16); }
__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t idx) { return rng_half(rng_pair(seed, idx >> 1), (int)(idx & 1)); }
// draws of the N consecutive elements base .. base + N - 1, base EVEN and N even: N / 2 hashes
template <int N>
__device__ __forceinline__ void rng_run(uint64_t seed, uint64_t base, uint32_t (&r)[N]) {
  const uint32_t x0 = (uint32_t)(base >> 1) * RNG_C1 + rng_seedmix(seed);
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    const uint32_t h = rng_fmix(x0 + (uint32_t)(j >> 1) * RNG_C1);
    r[j] = h << 16;
    r[j + 1] = h & 0xffff0000u;
  }
}
// The same draws from a precomputed hash input: x = (hash input of the pair that holds the first element); elements
// first .. first + N - 1 with `first` even.
template <int N>
__device__ __forceinline__ void rng_lin_run(uint32_t x, uint32_t (&r)[N]) {
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    const uint32_t h = rng_fmix(x + (uint32_t)(j >> 1) * RNG_C1);
    r[j] = h << 16;
    r[j + 1] = h & 0xffff0000u;
  }
}
// keep with probability 1-p:  u32 >= thr, thr = p * 2^32
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xffffffffU : (uint32_t)t;
}

// ---- activations: vit/helpers.py:9-48 ------------------------------------------------------------
// sigmoid(a*x) with one v_exp_f32 and one v_rcp_f32 (1 ulp each); exact 0 / 1 at the far ends
__device__ __forceinline__ float fast_sigmoid(float x, float a = 1.f) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * (-1.44269504088896340736f * a)));
}
__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case VITK_ACT_RELU: return fmaxf(x, 0.f);
    case VITK_ACT_SILU: return x * fast_sigmoid(x);
    case VITK_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
    case VITK_ACT_SRELU: { float r = fmaxf(x, 0.f); return r * r; }
    default: return x * fast_sigmoid(x, 1.702f);  // OPENSWIGLU: x * sigmoid(1.702 x)
  }
}
__device__ __forceinline__ float act_grad(int act, float x) {
  switch (act) {
    case VITK_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case VITK_ACT_SILU: { float s = fast_sigmoid(x); return s * (1.f + x * (1.f - s)); }
    case VITK_ACT_GELU: {
      float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
      float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case VITK_ACT_SRELU: return x > 0.f ? 2.f * x : 0.f;
    default: { float s = fast_sigmoid(x, 1.702f); return s * (1.f + 1.702f * x * (1.f - s)); }
  }
}

// compile-time activation variants (loops over register arrays unroll only when the activation is a constant)
template <int ACT> __device__ __forceinline__ float act_apply_t(float x) {
  if (ACT == VITK_ACT_RELU) return fmaxf(x, 0.f);
  if (ACT == VITK_ACT_SILU) return x * fast_sigmoid(x);
  if (ACT == VITK_ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
  if (ACT == VITK_ACT_SRELU) { const float r = fmaxf(x, 0.f); return r * r; }
  return x * fast_sigmoid(x, 1.702f);
}
template <int ACT> __device__ __forceinline__ float act_grad_t(float x) {
  if (ACT == VITK_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (ACT == VITK_ACT_SILU) { const float s = fast_sigmoid(x); return s * (1.f + x * (1.f - s)); }
  if (ACT == VITK_ACT_GELU) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
  }
  if (ACT == VITK_ACT_SRELU) return x > 0.f ? 2.f * x : 0.f;
  const float s = fast_sigmoid(x, 1.702f);
  return s * (1.f + 1.702f * x * (1.f - s));
}
#ifdef EXP_ONLY_SILU   // diagnostic build: every fused activation is SiLU (what a single compiled variant costs)
#define ACT_SWITCH(code, ACT, ...) { constexpr int ACT = VITK_ACT_SILU; __VA_ARGS__; }
#else
#define ACT_SWITCH(code, ACT, ...)                                              \
  switch (code) {                                                              \
    case VITK_ACT_RELU: { constexpr int ACT = VITK_ACT_RELU; __VA_ARGS__; } break;   \
    case VITK_ACT_SILU: { constexpr int ACT = VITK_ACT_SILU; __VA_ARGS__; } break;   \
    case VITK_ACT_GELU: { constexpr int ACT = VITK_ACT_GELU; __VA_ARGS__; } break;   \
    case VITK_ACT_SRELU: { constexpr int ACT = VITK_ACT_SRELU; __VA_ARGS__; } break; \
    default: { constexpr int ACT = VITK_ACT_OPENSWIGLU; __VA_ARGS__; } break;        \
  }
#endif

__host__ __device__ static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
