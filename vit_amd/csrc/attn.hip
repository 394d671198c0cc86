// Scaled-dot-product attention (non-causal, no mask, dropout on the probabilities), forward and backward.
//
// MFMA kernels (bf16, v_mfma_f32_32x32x16_bf16), all with the same shape: a workgroup of NW waves, each
// wave owning 32 rows of the "stationary" side in registers while 64-row tiles of the "streamed" side
// pass through LDS (global -> registers -> LDS, next tile's loads in flight during the MFMAs):
//   attn_fwd_k   stationary Q (32 queries / wave), streamed K,V.  S^T = K Q^T is computed with the key
//                index on the accumulator rows and the query on the lane, so max/sum over keys are
//                in-register reductions plus one exchange between the two 32-lane halves; the
//                probabilities are converted in place into the B operand of O^T += V^T P^T (V read
//                column-wise with ds_read_b64_tr_b16).
//   attn_dq_k    same orientation; dS^T feeds dQ^T += K^T dS^T (K read row-wise and column-wise).
//   attn_dkdv_k  stationary K,V (32 keys / wave), streamed Q,dO; S = Q K^T and dP = dO V^T have the key on
//                the lane, P and dS feed dV^T += dO^T P and dK^T += Q^T dS without leaving registers.
// Tensors are addressed through (batch, head, row) strides so the kernels read the packed projection
// [B,S,3,H,Dh] and write [B,S,H*Dh] directly (no head split / merge copies).
// Portable kernels (any dtype, Dh <= 128, fp32 math) implement the same three passes with scalar code.
#include "common.h"
#include <stdlib.h>

struct AttnP {
  const void *q, *k, *v, *dout;
  void *o, *dq, *dk, *dv;
  float *lse, *delta;
  int64_t B, H, S, Dh;
  int64_t q_sb, q_sh, q_ss, o_sb, o_sh, o_ss, g_sb, g_sh, g_ss;
  uint64_t seed;
  float scale, keep_scale;
  uint32_t thr;
};

#define LOG2E 1.4426950408889634f

// delta[b,h,s] = sum_d dO * O   (one wave per row)
template <class T>
__global__ __launch_bounds__(256) void attn_delta_k(AttnP p) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t total = p.B * p.H * p.S;
  if (row >= total) return;
  const int64_t s = row % p.S, h = (row / p.S) % p.H, b = row / (p.S * p.H);
  const T* o = (const T*)p.o + b * p.o_sb + h * p.o_sh + s * p.o_ss;
  const T* g = (const T*)p.dout + b * p.o_sb + h * p.o_sh + s * p.o_ss;
  float acc = 0.f;
  for (int d = lane; d < p.Dh; d += 64) acc += to_f<T>(o[d]) * to_f<T>(g[d]);
  acc = wave_sum(acc);
  if (lane == 0) p.delta[row] = acc;
}

// ================================================================================================
// portable kernels: 64 stationary rows per block, 4 threads per row (each owns d = part + 4 i)
// ================================================================================================
constexpr int PMAXI = 32;  // Dh <= 128
constexpr int PT = 32;     // streamed rows per LDS tile

template <class T>
__device__ __forceinline__ void pload_tile(float* dst, const T* src, int64_t ss, int64_t row0, int64_t nrows, int Dh, int tid) {
  for (int i = tid; i < PT * Dh; i += 256) {
    const int r = i / Dh, d = i % Dh;
    dst[i] = (row0 + r < nrows) ? to_f<T>(src[(row0 + r) * ss + d]) : 0.f;
  }
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

template <class T>
__global__ __launch_bounds__(256) void attn_fwd_simple_k(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Dh = (int)p.Dh, tid = threadIdx.x, part = tid & 3;
  const int64_t b = blockIdx.z, h = blockIdx.y, qrow = (int64_t)blockIdx.x * 64 + (tid >> 2);
  float* Ks = sm;
  float* Vs = sm + PT * Dh;
  const T* qp = (const T*)p.q + b * p.q_sb + h * p.q_sh;
  const T* kp = (const T*)p.k + b * p.q_sb + h * p.q_sh;
  const T* vp = (const T*)p.v + b * p.q_sb + h * p.q_sh;
  const bool qv = qrow < p.S;
  float q[PMAXI], o[PMAXI];
#pragma unroll
  for (int i = 0; i < PMAXI; i++) {
    const int d = part + 4 * i;
    q[i] = (qv && d < Dh) ? to_f<T>(qp[qrow * p.q_ss + d]) : 0.f;
    o[i] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  const uint64_t rbase = ((uint64_t)(b * p.H + h) * p.S + (uint64_t)qrow) * p.S;
  for (int64_t k0 = 0; k0 < p.S; k0 += PT) {
    __syncthreads();
    pload_tile<T>(Ks, kp, p.q_ss, k0, p.S, Dh, tid);
    pload_tile<T>(Vs, vp, p.q_ss, k0, p.S, Dh, tid);
    __syncthreads();
    const int nk = (int)(p.S - k0 < PT ? p.S - k0 : PT);
    for (int kk = 0; kk < nk; kk++) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < PMAXI; i++) {
        const int d = part + 4 * i;
        if (d < Dh) s = fmaf(q[i], Ks[kk * Dh + d], s);
      }
      s = quad_sum(s) * p.scale;
      const float mn = fmaxf(m, s);
      const float a = expf(m - mn), pr = expf(s - mn);
      l = l * a + pr;
      m = mn;
      float pd = pr;
      if (p.thr) pd = rng_u32(p.seed, rbase + (uint64_t)(k0 + kk)) >= p.thr ? pr * p.keep_scale : 0.f;
#pragma unroll
      for (int i = 0; i < PMAXI; i++) {
        const int d = part + 4 * i;
        if (d < Dh) o[i] = fmaf(pd, Vs[kk * Dh + d], o[i] * a);
      }
    }
  }
  if (!qv) return;
  T* op = (T*)p.o + b * p.o_sb + h * p.o_sh + qrow * p.o_ss;
  const float inv = 1.f / l;
#pragma unroll
  for (int i = 0; i < PMAXI; i++) {
    const int d = part + 4 * i;
    if (d < Dh) op[d] = from_f<T>(o[i] * inv);
  }
  if (part == 0) p.lse[(b * p.H + h) * p.S + qrow] = m + logf(l);
}

template <class T>
__global__ __launch_bounds__(256) void attn_dq_simple_k(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Dh = (int)p.Dh, tid = threadIdx.x, part = tid & 3;
  const int64_t b = blockIdx.z, h = blockIdx.y, qrow = (int64_t)blockIdx.x * 64 + (tid >> 2);
  float* Ks = sm;
  float* Vs = sm + PT * Dh;
  const T* qp = (const T*)p.q + b * p.q_sb + h * p.q_sh;
  const T* kp = (const T*)p.k + b * p.q_sb + h * p.q_sh;
  const T* vp = (const T*)p.v + b * p.q_sb + h * p.q_sh;
  const T* gp = (const T*)p.dout + b * p.o_sb + h * p.o_sh;
  const bool qv = qrow < p.S;
  float q[PMAXI], g[PMAXI], dq[PMAXI];
#pragma unroll
  for (int i = 0; i < PMAXI; i++) {
    const int d = part + 4 * i;
    q[i] = (qv && d < Dh) ? to_f<T>(qp[qrow * p.q_ss + d]) : 0.f;
    g[i] = (qv && d < Dh) ? to_f<T>(gp[qrow * p.o_ss + d]) : 0.f;
    dq[i] = 0.f;
  }
  const int64_t srow = (b * p.H + h) * p.S + qrow;
  const float lse = qv ? p.lse[srow] : 0.f, delta = qv ? p.delta[srow] : 0.f;
  const uint64_t rbase = (uint64_t)srow * p.S;
  for (int64_t k0 = 0; k0 < p.S; k0 += PT) {
    __syncthreads();
    pload_tile<T>(Ks, kp, p.q_ss, k0, p.S, Dh, tid);
    pload_tile<T>(Vs, vp, p.q_ss, k0, p.S, Dh, tid);
    __syncthreads();
    const int nk = (int)(p.S - k0 < PT ? p.S - k0 : PT);
    for (int kk = 0; kk < nk; kk++) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int i = 0; i < PMAXI; i++) {
        const int d = part + 4 * i;
        if (d < Dh) { s = fmaf(q[i], Ks[kk * Dh + d], s); dp = fmaf(g[i], Vs[kk * Dh + d], dp); }
      }
      s = quad_sum(s) * p.scale;
      dp = quad_sum(dp);
      const float pr = expf(s - lse);
      if (p.thr) dp = rng_u32(p.seed, rbase + (uint64_t)(k0 + kk)) >= p.thr ? dp * p.keep_scale : 0.f;
      const float ds = pr * (dp - delta);
#pragma unroll
      for (int i = 0; i < PMAXI; i++) {
        const int d = part + 4 * i;
        if (d < Dh) dq[i] = fmaf(ds, Ks[kk * Dh + d], dq[i]);
      }
    }
  }
  if (!qv) return;
  T* op = (T*)p.dq + b * p.g_sb + h * p.g_sh + qrow * p.g_ss;
#pragma unroll
  for (int i = 0; i < PMAXI; i++) {
    const int d = part + 4 * i;
    if (d < Dh) op[d] = from_f<T>(dq[i] * p.scale);
  }
}

template <class T>
__global__ __launch_bounds__(256) void attn_dkdv_simple_k(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Dh = (int)p.Dh, tid = threadIdx.x, part = tid & 3;
  const int64_t b = blockIdx.z, h = blockIdx.y, krow = (int64_t)blockIdx.x * 64 + (tid >> 2);
  float* Qs = sm;
  float* Gs = sm + PT * Dh;
  float* Ls = sm + 2 * PT * Dh;  // lse[PT], delta[PT]
  const T* qp = (const T*)p.q + b * p.q_sb + h * p.q_sh;
  const T* kp = (const T*)p.k + b * p.q_sb + h * p.q_sh;
  const T* vp = (const T*)p.v + b * p.q_sb + h * p.q_sh;
  const T* gp = (const T*)p.dout + b * p.o_sb + h * p.o_sh;
  const bool kv = krow < p.S;
  float k[PMAXI], v[PMAXI], dk[PMAXI], dv[PMAXI];
#pragma unroll
  for (int i = 0; i < PMAXI; i++) {
    const int d = part + 4 * i;
    k[i] = (kv && d < Dh) ? to_f<T>(kp[krow * p.q_ss + d]) : 0.f;
    v[i] = (kv && d < Dh) ? to_f<T>(vp[krow * p.q_ss + d]) : 0.f;
    dk[i] = 0.f; dv[i] = 0.f;
  }
  const int64_t sbase = (b * p.H + h) * p.S;
  for (int64_t q0 = 0; q0 < p.S; q0 += PT) {
    __syncthreads();
    pload_tile<T>(Qs, qp, p.q_ss, q0, p.S, Dh, tid);
    pload_tile<T>(Gs, gp, p.o_ss, q0, p.S, Dh, tid);
    if (tid < PT) {
      const bool ok = q0 + tid < p.S;
      Ls[tid] = ok ? p.lse[sbase + q0 + tid] : 0.f;
      Ls[PT + tid] = ok ? p.delta[sbase + q0 + tid] : 0.f;
    }
    __syncthreads();
    const int nq = (int)(p.S - q0 < PT ? p.S - q0 : PT);
    for (int qq = 0; qq < nq; qq++) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int i = 0; i < PMAXI; i++) {
        const int d = part + 4 * i;
        if (d < Dh) { s = fmaf(k[i], Qs[qq * Dh + d], s); dp = fmaf(v[i], Gs[qq * Dh + d], dp); }
      }
      s = quad_sum(s) * p.scale;
      dp = quad_sum(dp);
      const float pr = expf(s - Ls[qq]);
      float pd = pr;
      if (p.thr) {
        const bool keep = rng_u32(p.seed, (uint64_t)(sbase + q0 + qq) * p.S + (uint64_t)krow) >= p.thr;
        pd = keep ? pr * p.keep_scale : 0.f;
        dp = keep ? dp * p.keep_scale : 0.f;
      }
      const float ds = pr * (dp - Ls[PT + qq]);
#pragma unroll
      for (int i = 0; i < PMAXI; i++) {
        const int d = part + 4 * i;
        if (d < Dh) { dv[i] = fmaf(pd, Gs[qq * Dh + d], dv[i]); dk[i] = fmaf(ds, Qs[qq * Dh + d], dk[i]); }
      }
    }
  }
  if (!kv) return;
  T* okp = (T*)p.dk + b * p.g_sb + h * p.g_sh + krow * p.g_ss;
  T* ovp = (T*)p.dv + b * p.g_sb + h * p.g_sh + krow * p.g_ss;
#pragma unroll
  for (int i = 0; i < PMAXI; i++) {
    const int d = part + 4 * i;
    if (d < Dh) { okp[d] = from_f<T>(dk[i] * p.scale); ovp[d] = from_f<T>(dv[i]); }
  }
}

// ================================================================================================
// MFMA kernels
// ================================================================================================
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// LDS row strides (bytes) for a [64][DHP] bf16 tile.
//   ROWS: ds_read_b128 of 32 different rows at one column is conflict-free when stride/4 = 4 (mod 8) dwords
//   COLS: ds_read_b64_tr_b16 blocks of 4 rows are conflict-free when stride/4 = 16 or 48 (mod 64) dwords
template <int DHP> struct Strides {
  static constexpr int ROWS = DHP * 2 + 16;
  static constexpr int COLS = (DHP == 96) ? 192 : DHP * 2 + 64;
};

// global [rows x Dh] (row stride ss) -> registers, zero filled outside (row < nrows, d < Dh); NCH chunks/thread
template <int DHP, int NT, int NCH>
__device__ __forceinline__ void stage_load(uint4 (&r)[NCH], const bf16* __restrict__ g, int64_t ss, int64_t row0,
                                           int64_t nrows, int Dh, int tid) {
  constexpr int CPR = DHP / 8;
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const int c = tid + i * NT;
    const int row = c / CPR, dc = c % CPR;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (c < 64 * CPR && row0 + row < nrows && dc * 8 < Dh) v = *(const uint4*)(g + (row0 + row) * ss + dc * 8);
    r[i] = v;
  }
}
template <int DHP, int NT, int NCH, int STRIDE>
__device__ __forceinline__ void stage_store(const uint4 (&r)[NCH], char* s, int tid) {
  constexpr int CPR = DHP / 8;
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const int c = tid + i * NT;
    if (c < 64 * CPR) *(uint4*)(s + (c / CPR) * STRIDE + (c % CPR) * 16) = r[i];
  }
}

// row-wise operand fragment: element j = tile[r0 + (lane&31)][ks*16 + 8*(lane>>5) + j]
template <int STRIDE>
__device__ __forceinline__ bf16x8 frag_rows(const char* s, int r0, int ks, int lane) {
  return *(const bf16x8*)(s + (r0 + (lane & 31)) * STRIDE + (ks * 16 + 8 * (lane >> 5)) * 2);
}
// column-wise operand fragment for a product whose other operand is an accumulator tile converted in
// place: element j = tile[r0 + 16*s2 + 8*(j>>2) + 4*(lane>>5) + (j&3)][c0 + (lane&31)]
template <int STRIDE>
__device__ __forceinline__ bf16x8 frag_cols(const char* s, int r0, int s2, int c0, int lane) {
  const int row = r0 + 16 * s2 + 4 * (lane >> 5) + ((lane & 15) >> 2);
  const int col = c0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const char* p = s + row * STRIDE + col * 2;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 8 * STRIDE));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
// accumulator registers 8*s2 .. 8*s2+7 as the B operand of the next product (k-step s2)
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int s2) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; j++) f[j] = (bf16)a[8 * s2 + j];
  return f;
}
// stationary-side fragments straight from global: element j = g[row][ks*16 + 8*(lane>>5) + j]
template <int DHP>
__device__ __forceinline__ void load_stationary(bf16x8 (&f)[DHP / 16], const bf16* __restrict__ g, int64_t ss, int64_t row,
                                                bool valid, int Dh, int lane) {
#pragma unroll
  for (int ks = 0; ks < DHP / 16; ks++) {
    const int d = ks * 16 + 8 * (lane >> 5);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (valid && d < Dh) v = *(const uint4*)(g + row * ss + d);
    f[ks] = __builtin_bit_cast(bf16x8, v);
  }
}
// row index of accumulator register r inside a 32x32 tile
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// store a transposed accumulator set X^T[d][row] (row on the lane) as row-major bf16 rows, optional scale
template <int DHP>
__device__ __forceinline__ void store_rows(bf16* __restrict__ dst, const f32x16 (&acc)[DHP / 32], float mul, int Dh, int lane) {
#pragma unroll
  for (int dt = 0; dt < DHP / 32; dt++)
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int d = dt * 32 + 8 * g + 4 * (lane >> 5);
      if (d < Dh) {
        bf16x4 t;
#pragma unroll
        for (int j = 0; j < 4; j++) t[j] = (bf16)(acc[dt][4 * g + j] * mul);
        *(bf16x4*)(dst + d) = t;
      }
    }
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // one v_exp_f32

#define ZERO16 {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}

template <int DHP, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_fwd_k(AttnP p) {
  constexpr int NT = NW * 64, CPR = DHP / 8, NCH = (64 * CPR + NT - 1) / NT;
  constexpr int KS = Strides<DHP>::ROWS, VS = Strides<DHP>::COLS;
  __shared__ __attribute__((aligned(16))) char smem[64 * KS + 64 * VS];
  char* sK = smem;
  char* sV = smem + 64 * KS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, Dh = (int)p.Dh, S = (int)p.S;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int qrow = (int)blockIdx.x * (NW * 32) + wid * 32 + (lane & 31);
  const bool qv = qrow < S;
  const bool wave_on = (int)blockIdx.x * (NW * 32) + wid * 32 < S;
  const bf16* qp = (const bf16*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16* kp = (const bf16*)p.k + b * p.q_sb + h * p.q_sh;
  const bf16* vp = (const bf16*)p.v + b * p.q_sb + h * p.q_sh;

  bf16x8 qf[DHP / 16];
  load_stationary<DHP>(qf, qp, p.q_ss, qrow, qv, Dh, lane);
  const f32x16 zero = ZERO16;
  f32x16 oacc[DHP / 32];
#pragma unroll
  for (int dt = 0; dt < DHP / 32; dt++) oacc[dt] = zero;
  float m = -INFINITY, l = 0.f;
  const float c = p.scale * LOG2E;
  const uint64_t rbase = ((uint64_t)(b * p.H + h) * p.S + (uint64_t)qrow) * p.S;
  const int rowoff = 4 * (lane >> 5);   // accumulator register r holds key row (r&3) + 8*(r>>2) + rowoff

  uint4 rk[NCH], rv[NCH];
  stage_load<DHP, NT, NCH>(rk, kp, p.q_ss, 0, p.S, Dh, tid);
  stage_load<DHP, NT, NCH>(rv, vp, p.q_ss, 0, p.S, Dh, tid);
  for (int k0 = 0; k0 < S; k0 += 64) {
    __syncthreads();
    stage_store<DHP, NT, NCH, KS>(rk, sK, tid);
    stage_store<DHP, NT, NCH, VS>(rv, sV, tid);
    __syncthreads();
    if (k0 + 64 < S) {
      stage_load<DHP, NT, NCH>(rk, kp, p.q_ss, k0 + 64, p.S, Dh, tid);
      stage_load<DHP, NT, NCH>(rv, vp, p.q_ss, k0 + 64, p.S, Dh, tid);
    }
    if (!wave_on) continue;                    // this wave's 32 queries are all beyond S: it only helps staging
    const bool two = k0 + 32 < S;              // the second 32-key half of the tile holds valid keys (wave-uniform)
    // scores: sacc[st][r] = S^T[key = k0 + st*32 + row(r)][query = lane&31]
    f32x16 sacc[2];
#pragma unroll
    for (int st = 0; st < 2; st++) {
      sacc[st] = zero;
      if (st == 0 || two) {
#pragma unroll
        for (int ks = 0; ks < DHP / 16; ks++)
          sacc[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<KS>(sK, st * 32, ks, lane), qf[ks], sacc[st], 0, 0, 0);
      }
    }
    if (k0 + 64 > S) {                         // last, partial tile: keys beyond S get probability 0
#pragma unroll
      for (int st = 0; st < 2; st++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (k0 + st * 32 + (r & 3) + 8 * (r >> 2) + rowoff >= S) sacc[st][r] = -INFINITY;
    }
    float mx = m;
#pragma unroll
    for (int st = 0; st < 2; st++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, sacc[st][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float nmc = -mx * c;
    float psum = 0.f;
#pragma unroll
    for (int st = 0; st < 2; st++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float pr = fast_exp2(fmaf(sacc[st][r], c, nmc));
        psum += pr;
        if (DROP) {
          const uint64_t key = (uint64_t)(k0 + st * 32 + (r & 3) + 8 * (r >> 2) + rowoff);
          pr = rng_u32(p.seed, rbase + key) >= p.thr ? pr * p.keep_scale : 0.f;
        }
        sacc[st][r] = pr;
      }
    if (__any(mx != m)) {                      // running max moved for some query of this wave: rescale O and l
      const float alpha = fast_exp2((m - mx) * c);
      l *= alpha;
#pragma unroll
      for (int dt = 0; dt < DHP / 32; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[dt][r] *= alpha;
      m = mx;
    }
    l += psum;
    // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
    for (int st = 0; st < 2; st++)
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        if (st == 1 && !two) continue;
        const bf16x8 pf = acc_to_frag(sacc[st], s2);
#pragma unroll
        for (int dt = 0; dt < DHP / 32; dt++)
          oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<VS>(sV, st * 32, s2, dt * 32, lane), pf, oacc[dt], 0, 0, 0);
      }
  }
  l += __shfl_xor(l, 32, 64);
  if (!qv) return;
  store_rows<DHP>((bf16*)p.o + b * p.o_sb + h * p.o_sh + (int64_t)qrow * p.o_ss, oacc, 1.f / l, Dh, lane);
  if (lane < 32) p.lse[(b * p.H + h) * p.S + qrow] = m * p.scale + logf(l);
}

template <int DHP, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_dq_k(AttnP p) {
  constexpr int NT = NW * 64, CPR = DHP / 8, NCH = (64 * CPR + NT - 1) / NT;
  constexpr int KS = Strides<DHP>::COLS, VS = Strides<DHP>::ROWS;  // K is read both ways, V row-wise only
  __shared__ __attribute__((aligned(16))) char smem[64 * KS + 64 * VS];
  char* sK = smem;
  char* sV = smem + 64 * KS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, Dh = (int)p.Dh, S = (int)p.S;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int qrow = (int)blockIdx.x * (NW * 32) + wid * 32 + (lane & 31);
  const bool qv = qrow < S;
  const bool wave_on = (int)blockIdx.x * (NW * 32) + wid * 32 < S;
  const bf16* qp = (const bf16*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16* kp = (const bf16*)p.k + b * p.q_sb + h * p.q_sh;
  const bf16* vp = (const bf16*)p.v + b * p.q_sb + h * p.q_sh;
  const bf16* gp = (const bf16*)p.dout + b * p.o_sb + h * p.o_sh;

  bf16x8 qf[DHP / 16], gf[DHP / 16];
  load_stationary<DHP>(qf, qp, p.q_ss, qrow, qv, Dh, lane);
  load_stationary<DHP>(gf, gp, p.o_ss, qrow, qv, Dh, lane);
  const int64_t srow = (b * p.H + h) * p.S + qrow;
  const float nlse2 = qv ? -p.lse[srow] * LOG2E : 0.f;
  // delta = rowsum(dO * O), computed here (each half-wave holds half of the row) and published for attn_dkdv_k,
  // which is launched after this kernel on the same stream
  float delta = 0.f;
  {
    bf16x8 of[DHP / 16];
    load_stationary<DHP>(of, (const bf16*)p.o + b * p.o_sb + h * p.o_sh, p.o_ss, qrow, qv, Dh, lane);
#pragma unroll
    for (int ks = 0; ks < DHP / 16; ks++)
#pragma unroll
      for (int j = 0; j < 8; j++) delta += (float)of[ks][j] * (float)gf[ks][j];
    delta += __shfl_xor(delta, 32, 64);
    if (qv && lane < 32) p.delta[srow] = delta;
  }
  const float c = p.scale * LOG2E;
  const uint64_t rbase = (uint64_t)srow * p.S;
  const int rowoff = 4 * (lane >> 5);
  const f32x16 zero = ZERO16;
  f32x16 dqacc[DHP / 32];
#pragma unroll
  for (int dt = 0; dt < DHP / 32; dt++) dqacc[dt] = zero;

  uint4 rk[NCH], rv[NCH];
  stage_load<DHP, NT, NCH>(rk, kp, p.q_ss, 0, p.S, Dh, tid);
  stage_load<DHP, NT, NCH>(rv, vp, p.q_ss, 0, p.S, Dh, tid);
  for (int k0 = 0; k0 < S; k0 += 64) {
    __syncthreads();
    stage_store<DHP, NT, NCH, KS>(rk, sK, tid);
    stage_store<DHP, NT, NCH, VS>(rv, sV, tid);
    __syncthreads();
    if (k0 + 64 < S) {
      stage_load<DHP, NT, NCH>(rk, kp, p.q_ss, k0 + 64, p.S, Dh, tid);
      stage_load<DHP, NT, NCH>(rv, vp, p.q_ss, k0 + 64, p.S, Dh, tid);
    }
    if (!wave_on) continue;
#pragma unroll
    for (int st = 0; st < 2; st++) {
      if (st == 1 && k0 + 32 >= S) continue;   // no valid key in the second half of the last tile
      f32x16 sacc = zero, pacc = zero;
#pragma unroll
      for (int ks = 0; ks < DHP / 16; ks++) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<KS>(sK, st * 32, ks, lane), qf[ks], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<VS>(sV, st * 32, ks, lane), gf[ks], pacc, 0, 0, 0);
      }
      // dS^T[key][q] = P (dP - delta); keys beyond S contribute nothing
      const bool partial = k0 + st * 32 + 32 > S;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int key = k0 + st * 32 + (r & 3) + 8 * (r >> 2) + rowoff;
        float pr = fast_exp2(fmaf(sacc[r], c, nlse2));
        if (partial) pr = key < S ? pr : 0.f;
        float dp = pacc[r];
        if (DROP) dp = rng_u32(p.seed, rbase + (uint64_t)key) >= p.thr ? dp * p.keep_scale : 0.f;
        sacc[r] = pr * (dp - delta);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const bf16x8 df = acc_to_frag(sacc, s2);
#pragma unroll
        for (int dt = 0; dt < DHP / 32; dt++)
          dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<KS>(sK, st * 32, s2, dt * 32, lane), df, dqacc[dt], 0, 0, 0);
      }
    }
  }
  if (!qv) return;
  store_rows<DHP>((bf16*)p.dq + b * p.g_sb + h * p.g_sh + (int64_t)qrow * p.g_ss, dqacc, p.scale, Dh, lane);
}

template <int DHP, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64) void attn_dkdv_k(AttnP p) {
  constexpr int NT = NW * 64, CPR = DHP / 8, NCH = (64 * CPR + NT - 1) / NT;
  constexpr int QS = Strides<DHP>::COLS;  // Q and dO tiles are read both ways
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * QS + 512];
  char* sQ = smem;
  char* sG = smem + 64 * QS;
  float* sL = (float*)(smem + 2 * 64 * QS);  // -lse*log2e [64], delta[64]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, Dh = (int)p.Dh, S = (int)p.S;
  const int64_t b = blockIdx.z, h = blockIdx.y;
  const int krow = (int)blockIdx.x * (NW * 32) + wid * 32 + (lane & 31);
  const bool kvalid = krow < S;
  const bool wave_on = (int)blockIdx.x * (NW * 32) + wid * 32 < S;
  const bf16* qp = (const bf16*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16* kp = (const bf16*)p.k + b * p.q_sb + h * p.q_sh;
  const bf16* vp = (const bf16*)p.v + b * p.q_sb + h * p.q_sh;
  const bf16* gp = (const bf16*)p.dout + b * p.o_sb + h * p.o_sh;

  bf16x8 kf[DHP / 16], vf[DHP / 16];
  load_stationary<DHP>(kf, kp, p.q_ss, krow, kvalid, Dh, lane);
  load_stationary<DHP>(vf, vp, p.q_ss, krow, kvalid, Dh, lane);
  const int64_t sbase = (b * p.H + h) * p.S;
  const float c = p.scale * LOG2E;
  const int rowoff = 4 * (lane >> 5);
  const f32x16 zero = ZERO16;
  f32x16 dkacc[DHP / 32], dvacc[DHP / 32];
#pragma unroll
  for (int dt = 0; dt < DHP / 32; dt++) { dkacc[dt] = zero; dvacc[dt] = zero; }

  uint4 rq[NCH], rg[NCH];
  float rl = 0.f;
  // rows beyond S: -lse = -inf makes their probability exactly 0
  auto load_row_consts = [&](int q0) {
    if (tid < 128) {
      const int qq = q0 + (tid & 63);
      rl = qq < S ? (tid < 64 ? -p.lse[sbase + qq] * LOG2E : p.delta[sbase + qq]) : (tid < 64 ? -INFINITY : 0.f);
    }
  };
  stage_load<DHP, NT, NCH>(rq, qp, p.q_ss, 0, p.S, Dh, tid);
  stage_load<DHP, NT, NCH>(rg, gp, p.o_ss, 0, p.S, Dh, tid);
  load_row_consts(0);
  for (int q0 = 0; q0 < S; q0 += 64) {
    __syncthreads();
    stage_store<DHP, NT, NCH, QS>(rq, sQ, tid);
    stage_store<DHP, NT, NCH, QS>(rg, sG, tid);
    if (tid < 128) sL[tid] = rl;
    __syncthreads();
    if (q0 + 64 < S) {
      stage_load<DHP, NT, NCH>(rq, qp, p.q_ss, q0 + 64, p.S, Dh, tid);
      stage_load<DHP, NT, NCH>(rg, gp, p.o_ss, q0 + 64, p.S, Dh, tid);
      load_row_consts(q0 + 64);
    }
    if (!wave_on) continue;
#pragma unroll
    for (int st = 0; st < 2; st++) {
      if (st == 1 && q0 + 32 >= S) continue;   // no valid query in the second half of the last tile
      // sacc[r] = S[q = q0 + st*32 + row(r)][key = lane&31], pacc likewise for dP
      f32x16 sacc = zero, pacc = zero;
#pragma unroll
      for (int ks = 0; ks < DHP / 16; ks++) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<QS>(sQ, st * 32, ks, lane), kf[ks], sacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows<QS>(sG, st * 32, ks, lane), vf[ks], pacc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int ql = st * 32 + 8 * g + rowoff;
        const f32x4 l4 = *(const f32x4*)(sL + ql);
        const f32x4 d4 = *(const f32x4*)(sL + 64 + ql);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int r = 4 * g + j;
          const float pr = fast_exp2(fmaf(sacc[r], c, l4[j]));
          float pd = pr, dp = pacc[r];
          if (DROP) {
            const bool keep = rng_u32(p.seed, (uint64_t)(sbase + q0 + ql + j) * p.S + (uint64_t)krow) >= p.thr;
            pd = keep ? pr * p.keep_scale : 0.f;
            dp = keep ? dp * p.keep_scale : 0.f;
          }
          sacc[r] = pd;                  // P (dropped) for dV
          pacc[r] = pr * (dp - d4[j]);   // dS for dK
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const bf16x8 pf = acc_to_frag(sacc, s2);
        const bf16x8 df = acc_to_frag(pacc, s2);
#pragma unroll
        for (int dt = 0; dt < DHP / 32; dt++) {
          dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<QS>(sG, st * 32, s2, dt * 32, lane), pf, dvacc[dt], 0, 0, 0);
          dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols<QS>(sQ, st * 32, s2, dt * 32, lane), df, dkacc[dt], 0, 0, 0);
        }
      }
    }
  }
  if (!kvalid) return;
  store_rows<DHP>((bf16*)p.dk + b * p.g_sb + h * p.g_sh + (int64_t)krow * p.g_ss, dkacc, p.scale, Dh, lane);
  store_rows<DHP>((bf16*)p.dv + b * p.g_sb + h * p.g_sh + (int64_t)krow * p.g_ss, dvacc, 1.f, Dh, lane);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static AttnP make_p(const VitkAttn* a) {
  AttnP p;
  p.q = a->q; p.k = a->k; p.v = a->v; p.dout = a->dout; p.o = a->o; p.dq = a->dq; p.dk = a->dk; p.dv = a->dv;
  p.lse = a->lse; p.delta = a->delta; p.B = a->B; p.H = a->H; p.S = a->S; p.Dh = a->Dh;
  p.q_sb = a->q_sb; p.q_sh = a->q_sh; p.q_ss = a->q_ss; p.o_sb = a->o_sb; p.o_sh = a->o_sh; p.o_ss = a->o_ss;
  p.g_sb = a->dq_sb; p.g_sh = a->dq_sh; p.g_ss = a->dq_ss;
  p.seed = a->seed; p.scale = a->scale;
  p.thr = a->drop_p > 0.f ? drop_threshold(a->drop_p) : 0;
  p.keep_scale = 1.f / (1.f - a->drop_p);
  return p;
}
static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static bool attn_mfma_ok(const VitkAttn* a, bool bwd) {
  if (a->dtype != VITK_BF16 || a->Dh % 8 || a->Dh > 128 || a->Dh < 8) return false;
  const int64_t st[] = {a->q_sb, a->q_sh, a->q_ss, a->o_sb, a->o_sh, a->o_ss};
  for (int64_t s : st) if (s % 8) return false;
  if (!al16(a->q) || !al16(a->k) || !al16(a->v)) return false;
  if (((uintptr_t)a->o & 7)) return false;
  if (bwd) {
    if (a->dq_sb % 4 || a->dq_sh % 4 || a->dq_ss % 4) return false;
    if (!al16(a->dout) || ((uintptr_t)a->dq & 7) || ((uintptr_t)a->dk & 7) || ((uintptr_t)a->dv & 7)) return false;
  }
  return true;
}
static int attn_validate(const VitkAttn* a, bool bwd) {
  VITK_REQUIRE(a && a->q && a->k && a->v && a->o && a->lse, "attn: null pointer");
  VITK_REQUIRE(a->B >= 0 && a->H > 0 && a->S > 0 && a->Dh > 0 && a->Dh <= 128, "attn: bad shape B=%lld H=%lld S=%lld Dh=%lld",
               (long long)a->B, (long long)a->H, (long long)a->S, (long long)a->Dh);
  VITK_REQUIRE(a->H <= 65535 && a->B <= 65535, "attn: B and H must be <= 65535");
  VITK_REQUIRE(a->drop_p >= 0.f && a->drop_p < 1.f, "attn: dropout p=%f out of range", a->drop_p);
  VITK_REQUIRE(a->dtype == VITK_F32 || a->dtype == VITK_BF16, "attn: bad dtype");
  if (bwd) VITK_REQUIRE(a->dout && a->dq && a->dk && a->dv && a->delta, "attn_bwd: null pointer");
  return 0;
}
// 4 waves (128 rows) per workgroup when that wastes little of the last block, else 2 waves (64 rows)
static int pick_nw(int64_t S, int64_t Dh) {
  if (const char* f = getenv("VITK_ATTN_NW")) return atoi(f) == 2 ? 2 : 4;
  return (Dh > 96 || S % 128 == 0 || S % 128 > 64 || S >= 2048) ? 4 : 2;
}

#define ATTN_DISPATCH(KERNEL, DHP, NW, grid, p, st)                                             \
  do {                                                                                          \
    if (p.thr) {                                                                                \
      if (NW == 4) hipLaunchKernelGGL((KERNEL<DHP, 4, true>), grid, dim3(256), 0, st, p);       \
      else hipLaunchKernelGGL((KERNEL<DHP, 2, true>), grid, dim3(128), 0, st, p);               \
    } else {                                                                                    \
      if (NW == 4) hipLaunchKernelGGL((KERNEL<DHP, 4, false>), grid, dim3(256), 0, st, p);      \
      else hipLaunchKernelGGL((KERNEL<DHP, 2, false>), grid, dim3(128), 0, st, p);              \
    }                                                                                           \
  } while (0)
#define ATTN_DISPATCH_DH(KERNEL, Dh, NW, grid, p, st)                  \
  do {                                                                 \
    if (Dh <= 64) ATTN_DISPATCH(KERNEL, 64, NW, grid, p, st);          \
    else if (Dh <= 96) ATTN_DISPATCH(KERNEL, 96, NW, grid, p, st);     \
    else ATTN_DISPATCH(KERNEL, 128, NW, grid, p, st);                  \
  } while (0)

extern "C" int vitk_attn_fwd(const VitkAttn* a, void* stream) {
  if (int rc = attn_validate(a, false)) return rc;
  if (a->B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const AttnP p = make_p(a);
  const bool can = attn_mfma_ok(a, false);
  VITK_REQUIRE(a->algo != 2 || can, "attn_fwd: MFMA kernel cannot take this problem (dtype/alignment/head_dim)");
  if (can && a->algo != 1) {
    const int nw = pick_nw(a->S, a->Dh);
    dim3 grid((unsigned)ceil_div(a->S, nw * 32), (unsigned)a->H, (unsigned)a->B);
    ATTN_DISPATCH_DH(attn_fwd_k, a->Dh, nw, grid, p, st);
  } else {
    dim3 grid((unsigned)ceil_div(a->S, 64), (unsigned)a->H, (unsigned)a->B);
    const size_t lds = (size_t)2 * PT * a->Dh * sizeof(float);
    if (a->dtype == VITK_F32) hipLaunchKernelGGL((attn_fwd_simple_k<float>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((attn_fwd_simple_k<bf16>), grid, dim3(256), lds, st, p);
  }
  VITK_LAUNCH_CHECK("attn_fwd");
  return 0;
}

extern "C" int vitk_attn_bwd(const VitkAttn* a, void* stream) {
  if (int rc = attn_validate(a, true)) return rc;
  if (a->B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const AttnP p = make_p(a);
  const int64_t rows = a->B * a->H * a->S;
  const bool can = attn_mfma_ok(a, true);
  VITK_REQUIRE(a->algo != 2 || can, "attn_bwd: MFMA kernel cannot take this problem (dtype/alignment/head_dim)");
  if (can && a->algo != 1) {
    const int nw = pick_nw(a->S, a->Dh);
    dim3 grid((unsigned)ceil_div(a->S, nw * 32), (unsigned)a->H, (unsigned)a->B);
    ATTN_DISPATCH_DH(attn_dq_k, a->Dh, nw, grid, p, st);    // also writes delta
    ATTN_DISPATCH_DH(attn_dkdv_k, a->Dh, nw, grid, p, st);  // reads delta
  } else {
    if (a->dtype == VITK_F32) hipLaunchKernelGGL((attn_delta_k<float>), dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_delta_k<bf16>), dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, p);
    dim3 grid((unsigned)ceil_div(a->S, 64), (unsigned)a->H, (unsigned)a->B);
    const size_t lds = (size_t)2 * PT * a->Dh * sizeof(float);
    if (a->dtype == VITK_F32) {
      hipLaunchKernelGGL((attn_dq_simple_k<float>), grid, dim3(256), lds, st, p);
      hipLaunchKernelGGL((attn_dkdv_simple_k<float>), grid, dim3(256), lds + 512, st, p);
    } else {
      hipLaunchKernelGGL((attn_dq_simple_k<bf16>), grid, dim3(256), lds, st, p);
      hipLaunchKernelGGL((attn_dkdv_simple_k<bf16>), grid, dim3(256), lds + 512, st, p);
    }
  }
  VITK_LAUNCH_CHECK("attn_bwd");
  return 0;
}
