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
  // cross-attention: Sk keys per sample, strides of k/v and of dk/dv.  Self-attention: Sk = S and the strides repeat
  // those of q and dq.
  int64_t Sk, k_sb, k_sh, k_ss, gk_sb, gk_sh, gk_ss;
  uint64_t seed;
  float scale, keep_scale;
  uint32_t thr;
  const float* rope;   // backward, head_dim 64: dq and dk leave through the transposed rotation (table [2, S - prefix, 64])
  int rope_prefix, rope_emu;
};

#define LOG2E 1.4426950408889634f
// waves per SIMD the dropout instances (head_dim <= 64, 4-wave workgroups) are compiled for; measured at S = 272, B = 256:
// forward 0.263 ms at 2 and 0.218 at 3 (one spilled register); dQ is slower at 3 (16 spilled registers); dK/dV fits two
// (251 registers, no spill) since its draws come from 32-bit terms: backward 0.77 ms at 1, 0.62 at 2
#ifndef DROP_OCC_FWD
#define DROP_OCC_FWD 3
#endif
#ifndef DROP_OCC_DQ
#define DROP_OCC_DQ 2
#endif
#ifndef DROP_OCC_DKDV
#define DROP_OCC_DKDV 2
#endif
#define drop_xrow(p, row, lane) drop_xrow_raw((p).seed, drop_ld((p).Sk), (row), (lane))
// Dropout counter of probability (query row, key): row * drop_ld(Sk) + key, with the row stride rounded up to even so
// that keys 2i and 2i + 1 of every row share one hash (rng_pair)
__device__ __forceinline__ uint64_t drop_ld(int64_t Sk) { return (uint64_t)((Sk + 1) & ~(int64_t)1); }
// Hash input (rng_lin_run) of the pair that holds (row, key = 4 * (lane >> 5)): the MFMA tiles put keys k and k + 4 on the
// two halves of a wave, so the half's offset is folded into the per-lane term once.
__device__ __forceinline__ uint32_t drop_xrow_raw(uint64_t seed, uint64_t ld, uint64_t row, int lane) {
  return (uint32_t)(row * (ld >> 1)) * RNG_C1 + (uint32_t)(2 * (lane >> 5)) * RNG_C1 + rng_seedmix(seed);
}

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
  const T* kp = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* vp = (const T*)p.v + b * p.k_sb + h * p.k_sh;
  const bool qv = qrow < p.S;
  float q[PMAXI], o[PMAXI];
#pragma unroll
  for (int i = 0; i < PMAXI; i++) {
    const int d = part + 4 * i;
    q[i] = (qv && d < Dh) ? to_f<T>(qp[qrow * p.q_ss + d]) : 0.f;
    o[i] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  const uint64_t rbase = ((uint64_t)(b * p.H + h) * p.S + (uint64_t)qrow) * drop_ld(p.Sk);
  for (int64_t k0 = 0; k0 < p.Sk; k0 += PT) {
    __syncthreads();
    pload_tile<T>(Ks, kp, p.k_ss, k0, p.Sk, Dh, tid);
    pload_tile<T>(Vs, vp, p.k_ss, k0, p.Sk, Dh, tid);
    __syncthreads();
    const int nk = (int)(p.Sk - k0 < PT ? p.Sk - k0 : PT);
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
  const T* kp = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* vp = (const T*)p.v + b * p.k_sb + h * p.k_sh;
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
  const uint64_t rbase = (uint64_t)srow * drop_ld(p.Sk);
  for (int64_t k0 = 0; k0 < p.Sk; k0 += PT) {
    __syncthreads();
    pload_tile<T>(Ks, kp, p.k_ss, k0, p.Sk, Dh, tid);
    pload_tile<T>(Vs, vp, p.k_ss, k0, p.Sk, Dh, tid);
    __syncthreads();
    const int nk = (int)(p.Sk - k0 < PT ? p.Sk - k0 : PT);
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
  const T* kp = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  const T* vp = (const T*)p.v + b * p.k_sb + h * p.k_sh;
  const T* gp = (const T*)p.dout + b * p.o_sb + h * p.o_sh;
  const bool kv = krow < p.Sk;
  float k[PMAXI], v[PMAXI], dk[PMAXI], dv[PMAXI];
#pragma unroll
  for (int i = 0; i < PMAXI; i++) {
    const int d = part + 4 * i;
    k[i] = (kv && d < Dh) ? to_f<T>(kp[krow * p.k_ss + d]) : 0.f;
    v[i] = (kv && d < Dh) ? to_f<T>(vp[krow * p.k_ss + d]) : 0.f;
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
        const bool keep = rng_u32(p.seed, (uint64_t)(sbase + q0 + qq) * drop_ld(p.Sk) + (uint64_t)krow) >= p.thr;
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
  T* okp = (T*)p.dk + b * p.gk_sb + h * p.gk_sh + krow * p.gk_ss;
  T* ovp = (T*)p.dv + b * p.gk_sb + h * p.gk_sh + krow * p.gk_ss;
#pragma unroll
  for (int i = 0; i < PMAXI; i++) {
    const int d = part + 4 * i;
    if (d < Dh) { okp[d] = from_f<T>(dk[i] * p.scale); ovp[d] = from_f<T>(dv[i]); }
  }
}

// attention probabilities softmax(scale * q k^T) written out as [B, H, S, Sk] fp32 (diagnostics: the reference's
// forward_weights).  One wave per query row; lane l owns keys l, l + 64, ...: raw scores go to the output row first, then
// the same lanes normalise their own entries in place.
template <class T>
__global__ __launch_bounds__(256) void attn_weights_k(AttnP p, float* __restrict__ w) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Dh = (int)p.Dh, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wv;
  const int64_t total = p.B * p.H * p.S;
  const bool valid = row < total;
  const int64_t r = valid ? row : 0;
  const int64_t s = r % p.S, h = (r / p.S) % p.H, b = r / (p.S * p.H);
  float* qs = sm + wv * Dh;
  const T* qp = (const T*)p.q + b * p.q_sb + h * p.q_sh + s * p.q_ss;
  for (int d = lane; d < Dh; d += 64) qs[d] = to_f<T>(qp[d]);
  __syncthreads();
  if (!valid) return;
  const T* kp = (const T*)p.k + b * p.k_sb + h * p.k_sh;
  float* out = w + r * p.Sk;
  float m = -INFINITY;
  for (int64_t j = lane; j < p.Sk; j += 64) {
    float acc = 0.f;
    for (int d = 0; d < Dh; d++) acc = fmaf(qs[d], to_f<T>(kp[j * p.k_ss + d]), acc);
    acc *= p.scale;
    out[j] = acc;
    m = fmaxf(m, acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float l = 0.f;
  for (int64_t j = lane; j < p.Sk; j += 64) {
    const float e = expf(out[j] - m);
    out[j] = e;
    l += e;
  }
  l = wave_sum(l);
  const float inv = 1.f / l;
  for (int64_t j = lane; j < p.Sk; j += 64) out[j] *= inv;
}

// ================================================================================================
// MFMA kernels
// ================================================================================================
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// LDS row strides (bytes) for a [64][DHP] bf16 tile.
//   ROWS: ds_read_b128 of 32 different rows at one column is conflict-free when stride/4 = 4 (mod 8) dwords
//   COLS: ds_read_b64_tr_b16 blocks of 4 rows are conflict-free when stride/4 = 16 or 48 (mod 64) dwords.
// A tile that is read both ways uses the COLS stride and stores 16-byte chunk c of row r at chunk position
// c ^ ((r >> 2) & 3): the transposing reads still find 4 rows x 64 contiguous bytes on distinct banks (the 4 rows of a
// read share (r >> 2) & 3), and the 16 rows one ds_read_b128 lane group covers now land on 16 different bank quads.
template <int DHP> struct Strides {
  static constexpr int ROWS = DHP * 2 + 16;
  static constexpr int COLS = (DHP == 96) ? 192 : DHP * 2 + 64;
};

// Operand rows come through buffer loads: a row at or beyond S starts past the descriptor's size and reads as zeros,
// a 16-byte chunk beyond the head dimension is given an offset past any size (the host keeps S * stride below 2^30 B).
constexpr uint32_t OOB_OFF = 0xC0000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const bf16* g, int64_t ss, int S, int Dh) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, (int)(((int64_t)(S - 1) * ss + Dh) * 2), 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// global [64 rows x Dh] tiles (row stride ss) -> registers; NCH 16-byte chunks per thread
template <int NCH> struct TileSrc {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t off[NCH];     // this thread's chunks inside tile 0
  uint32_t tile_bytes;   // 64 rows
};
template <int DHP, int NT, int NCH>
__device__ __forceinline__ TileSrc<NCH> make_tilesrc(const bf16* g, int64_t ss, int S, int Dh, int tid) {
  constexpr int CPR = DHP / 8;
  TileSrc<NCH> t;
  t.rsrc = rows_rsrc(g, ss, S, Dh);
  t.tile_bytes = (uint32_t)(64 * ss * 2);
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const int c = tid + i * NT;
    const int row = c / CPR, dc = c % CPR;
    t.off[i] = (c < 64 * CPR && dc * 8 < Dh) ? (uint32_t)((row * ss + dc * 8) * 2) : OOB_OFF;
  }
  return t;
}
template <int NCH>
__device__ __forceinline__ void stage_load(uint4 (&r)[NCH], const TileSrc<NCH>& t, int tile) {
  const uint32_t base = (uint32_t)tile * t.tile_bytes;
#pragma unroll
  for (int i = 0; i < NCH; i++) r[i] = buf_load16(t.rsrc, t.off[i] + base);
}
// chunk swizzle of a staged tile that is read both ways (see Strides): 16-byte chunk c of row r sits at chunk position
// c ^ swz_of<1>(r); swz_of<0> leaves the chunks in place
template <int SWZ>
__device__ __forceinline__ int swz_of(int row) {
  return SWZ == 1 ? (row >> 2) & 3 : 0;
}

template <int DHP, int NT, int NCH, int STRIDE, int SWZ>
__device__ __forceinline__ void stage_store(const uint4 (&r)[NCH], char* s, int tid) {
  constexpr int CPR = DHP / 8;
#pragma unroll
  for (int i = 0; i < NCH; i++) {
    const int c = tid + i * NT;
    const int row = c / CPR, dc = c % CPR;
    const int pc = dc ^ swz_of<SWZ>(row);
    if (c < 64 * CPR) *(uint4*)(s + row * STRIDE + pc * 16) = r[i];
  }
}

// Both fragment readers form their address as  tile + (a term of the lane alone) + (a compile-time constant): with r0 a
// multiple of 32 the swizzle of a row depends on the lane only and touches the two low bits of the chunk index, so one
// tile needs two lane terms per reader (k-step parity / row half) and everything else sits in the instruction's offset
// field.  Spelled out this way the loops hold a handful of address registers instead of one per fragment.
// row-wise operand fragment: element j = tile[r0 + (lane&31)][ks*16 + 8*(lane>>5) + j]   (r0 a multiple of 32)
template <int STRIDE, int SWZ>
__device__ __forceinline__ bf16x8 frag_rows(const char* s, int r0, int ks, int lane) {
  const int rl = lane & 31, hi = lane >> 5;
  const int w = swz_of<SWZ>(rl);
  const int low = (((ks * 2) & 3) | hi) ^ w;          // chunk index, low two bits: one lane term per parity of ks
  const int lane_off = rl * STRIDE + low * 16;
  const int imm = r0 * STRIDE + ((ks * 2) & ~3) * 16;
  return *(const bf16x8*)(s + lane_off + imm);
}
// column-wise operand fragment for a product whose other operand is an accumulator tile converted in
// place: element j = tile[r0 + 16*s2 + 8*(j>>2) + 4*(lane>>5) + (j&3)][c0 + (lane&31)]   (r0, c0 multiples of 32)
template <int STRIDE, int SWZ>
__device__ __forceinline__ bf16x8 frag_cols(const char* s, int r0, int s2, int c0, int lane) {
  const int hi = lane >> 5, l3 = lane & 3;
  const int rowl = 4 * hi + ((lane & 15) >> 2);       // the lane's row inside a 16-row group; its partner row is 8 below
  const int cl = 2 * ((lane >> 4) & 1) + (l3 >> 1);   // the lane's chunk inside a 32-column group
  const int sub = (l3 & 1) * 8;
  const int w_lo = SWZ == 1 ? hi : 0, w_hi = SWZ == 1 ? (hi | 2) : 0;   // swz_of(row), swz_of(row + 8)
  const int lane_lo = rowl * STRIDE + (cl ^ w_lo) * 16 + sub;
  const int lane_hi = (rowl + 8) * STRIDE + (cl ^ w_hi) * 16 + sub;
  const int imm = (r0 + 16 * s2) * STRIDE + (c0 >> 3) * 16;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(s + lane_lo + imm));
  bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(s + lane_hi + imm));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi4[0]; f[5] = hi4[1]; f[6] = hi4[2]; f[7] = hi4[3];
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
__device__ __forceinline__ void load_stationary(bf16x8 (&f)[DHP / 16], const bf16* __restrict__ g, int64_t ss, int row,
                                                int S, int Dh, int lane) {
  const __amdgpu_buffer_rsrc_t rs = rows_rsrc(g, ss, S, Dh);
#pragma unroll
  for (int ks = 0; ks < DHP / 16; ks++) {
    const int d = ks * 16 + 8 * (lane >> 5);
    f[ks] = __builtin_bit_cast(bf16x8, buf_load16(rs, (row < S && d < Dh) ? (uint32_t)((row * ss + d) * 2) : OOB_OFF));
  }
}
// row index of accumulator register r inside a 32x32 tile (plus 4 for the upper 32 lanes)
__device__ __forceinline__ constexpr int acc_row(int r) { return (r & 3) + 8 * (r >> 2); }

// lanes l and l+32 exchange: returns {lower halves of a and b side by side, upper halves side by side}
__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
__device__ __forceinline__ float max_xor32(float x) {
  uint32_t a = __builtin_bit_cast(uint32_t, x), b = a;
  swap32(a, b);
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) bf16 bf16x2_t;
  bf16x2_t t;
  t[0] = (bf16)lo;
  t[1] = (bf16)hi;
  return __builtin_bit_cast(uint32_t, t);
}

// store a transposed accumulator set X^T[d][row] (row on the lane) as row-major bf16 rows, optional scale.  A lane holds
// 4 consecutive d per register quad and its partner lane (+-32) the next 4; quads of neighbouring groups are traded so
// that every lane writes 8 consecutive d with one 16-byte store.
template <int DHP>
__device__ __forceinline__ void store_rows(bf16* __restrict__ dst, const f32x16 (&acc)[DHP / 32], float mul, int Dh, int lane) {
#pragma unroll
  for (int dt = 0; dt < DHP / 32; dt++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t a01 = pack_bf16(acc[dt][8 * h + 0] * mul, acc[dt][8 * h + 1] * mul);
      uint32_t a23 = pack_bf16(acc[dt][8 * h + 2] * mul, acc[dt][8 * h + 3] * mul);
      uint32_t b01 = pack_bf16(acc[dt][8 * h + 4] * mul, acc[dt][8 * h + 5] * mul);
      uint32_t b23 = pack_bf16(acc[dt][8 * h + 6] * mul, acc[dt][8 * h + 7] * mul);
      swap32(a01, b01);
      swap32(a23, b23);
      const int d = dt * 32 + 16 * h + 8 * (lane >> 5);
      if (d < Dh) *(uint4*)(dst + d) = make_uint4(a01, a23, b01, b23);
    }
}

// store_rows for head_dim 64 followed by the transpose of the rotation that was applied to the rows in forward
// (vit/rope.py:148-163; the same arithmetic as vitk_rope_qk with inverse = 1 on the stored bf16 values):
//   g[d] = y[d] cos[d] + y[d + 32] sin[d + 32]   (d < 32),   g[d] = y[d] cos[d] - y[d - 32] sin[d - 32]   (d >= 32)
// with y the gradient rounded to bf16.  A lane holds d .. d + 7 of both halves of its row, so the partner is in
// registers.  `tok` = the row's index behind the prefix; negative (a prefix row) stores y unrotated.
__device__ __forceinline__ void store_rows_rope64(bf16* __restrict__ dst, const f32x16 (&acc)[2], float mul, int lane,
                                                  const float* __restrict__ table, int tok, int ntok, int emu) {
#pragma unroll
  for (int h = 0; h < 2; h++) {
    uint32_t w[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; dt++) {
      uint32_t a01 = pack_bf16(acc[dt][8 * h + 0] * mul, acc[dt][8 * h + 1] * mul);
      uint32_t a23 = pack_bf16(acc[dt][8 * h + 2] * mul, acc[dt][8 * h + 3] * mul);
      uint32_t b01 = pack_bf16(acc[dt][8 * h + 4] * mul, acc[dt][8 * h + 5] * mul);
      uint32_t b23 = pack_bf16(acc[dt][8 * h + 6] * mul, acc[dt][8 * h + 7] * mul);
      swap32(a01, b01);
      swap32(a23, b23);
      w[dt][0] = a01; w[dt][1] = a23; w[dt][2] = b01; w[dt][3] = b23;
    }
    const int d = 16 * h + 8 * (lane >> 5);   // first-half columns d .. d + 7, second-half columns d + 32 ..
    if (tok >= 0) {
      const float* sn = table + (int64_t)tok * 64 + d;
      const float* cs = sn + (int64_t)ntok * 64;
      float s1[8], s2[8], c1[8], c2[8];
      *(float4*)s1 = *(const float4*)sn;        *(float4*)(s1 + 4) = *(const float4*)(sn + 4);
      *(float4*)s2 = *(const float4*)(sn + 32); *(float4*)(s2 + 4) = *(const float4*)(sn + 36);
      *(float4*)c1 = *(const float4*)cs;        *(float4*)(c1 + 4) = *(const float4*)(cs + 4);
      *(float4*)c2 = *(const float4*)(cs + 32); *(float4*)(c2 + 4) = *(const float4*)(cs + 36);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float g1[2], g2[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; e2++) {
          const int j = 2 * q + e2;
          const float y1 = __builtin_bit_cast(float, e2 ? (w[0][q] & 0xffff0000u) : (w[0][q] << 16));
          const float y2 = __builtin_bit_cast(float, e2 ? (w[1][q] & 0xffff0000u) : (w[1][q] << 16));
          g1[e2] = rope_mix(y1, c1[j], y2, s2[j], emu);
          g2[e2] = rope_mix(y2, c2[j], -y1, s1[j], emu);
        }
        w[0][q] = pack_bf16(g1[0], g1[1]);
        w[1][q] = pack_bf16(g2[0], g2[1]);
      }
    }
    *(uint4*)(dst + d) = make_uint4(w[0][0], w[0][1], w[0][2], w[0][3]);
    *(uint4*)(dst + d + 32) = make_uint4(w[1][0], w[1][1], w[1][2], w[1][3]);
  }
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // one v_exp_f32

#define ZERO16 {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}
#define MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16

// Workgroup -> (batch, head, block of stationary rows) for the streaming kernels, from a 1-D grid.  Workgroups are dealt
// to the 8 XCDs round-robin in launch order and every XCD has its own L2, so the blocks that share one head's streamed
// operands are given launch ids 8 apart: they run on the same XCD at about the same time, and the operands come from
// memory once per head instead of once per block.
struct HeadBlock {
  int64_t b, h;
  int xb;
  bool valid;
};
__device__ __forceinline__ HeadBlock head_block(const AttnP& p, int64_t nrows, int rows_per_block) {
  const int nxb = (int)((nrows + rows_per_block - 1) / rows_per_block);
  const int64_t per = 8 * (int64_t)nxb, L = blockIdx.x;
  const int64_t g = L / per, r = L - g * per;
  const int64_t hh = g * 8 + (r & 7);
  HeadBlock o;
  o.xb = (int)(r >> 3);
  o.valid = hh < p.B * p.H;
  o.b = hh / p.H;
  o.h = hh - o.b * p.H;
  return o;
}
static unsigned head_block_grid(int64_t B, int64_t H, int64_t S, int rows_per_block) {
  return (unsigned)(ceil_div(B * H, 8) * 8 * ceil_div(S, rows_per_block));
}

// ---- forward ------------------------------------------------------------------------------------
// One 64-key tile (HALVES = 2) or its first 32 keys (HALVES = 1) against a wave's 32 queries.  Scores are kept
// transposed, S^T[key][query], so a lane owns one query: row statistics need one exchange with the partner lane, and the
// exponentiated tile is already laid out as the B operand of the P V product.  The row sum is taken by the matrix
// unit (a ones operand), on the same bf16-rounded probabilities that multiply V.
template <int DHP, int HALVES, bool PARTIAL, bool DROP, int KS, int VS, int KW = 0, int VW = 0>
__device__ __forceinline__ void fwd_tile(const AttnP& p, const char* sK, const char* sV, const bf16x8 (&qf)[DHP / 16],
                                         f32x16 (&oacc)[DHP / 32], f32x16& lacc, float& m, float c, int k0, int S,
                                         uint32_t xrow, int lane) {
  const f32x16 zero = ZERO16;
  const int rowoff = 4 * (lane >> 5);
  f32x16 sacc[HALVES];
#pragma unroll
  for (int st = 0; st < HALVES; st++) {
    sacc[st] = MFMA(frag_rows<KS, KW>(sK, st * 32, 0, lane), qf[0], zero, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < DHP / 16; ks++) sacc[st] = MFMA(frag_rows<KS, KW>(sK, st * 32, ks, lane), qf[ks], sacc[st], 0, 0, 0);
  }
  if (PARTIAL) {   // keys at or beyond S (only in the last half of the last tile) get probability 0
#pragma unroll
    for (int r = 0; r < 16; r++)
      if (k0 + (HALVES - 1) * 32 + acc_row(r) + rowoff >= S) sacc[HALVES - 1][r] = -INFINITY;
  }
  float mx = sacc[0][0];
#pragma unroll
  for (int st = 0; st < HALVES; st++)
#pragma unroll
    for (int r = 0; r < 16; r++) mx = fmaxf(mx, sacc[st][r]);
  mx = max_xor32(mx);
  // The reference maximum m only has to keep the exponentials in range, so it is moved (and O, l rescaled) only when
  // some query's tile maximum exceeds it by more than 2^8 in the exponent; all probabilities of a row share one m.
  if (__any(mx * c > m * c + 8.f)) {
    const float mnew = fmaxf(m, mx);
    const float alpha = fast_exp2((m - mnew) * c);
    lacc[0] *= alpha;     // only element 0 of the sum tile is ever read
#pragma unroll
    for (int dt = 0; dt < DHP / 32; dt++)
#pragma unroll
      for (int r = 0; r < 16; r++) oacc[dt][r] *= alpha;
    m = mnew;
  }
  const float nmc = -m * c;
#pragma unroll
  for (int st = 0; st < HALVES; st++)
#pragma unroll
    for (int r = 0; r < 16; r++) sacc[st][r] = fast_exp2(fmaf(sacc[st][r], c, nmc));
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; j++) ones[j] = (bf16)1.f;
  // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
  for (int st = 0; st < HALVES; st++)
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
      bf16x8 pf = acc_to_frag(sacc[st], s2);
      lacc = MFMA(ones, pf, lacc, 0, 0, 0);
      if (DROP) {   // registers 8 s2 + 0..3 and + 4..7 are two runs of four consecutive keys, each from a multiple of 4
#pragma unroll
        for (int g = 0; g < 2; g++) {
          uint32_t dr[4];   // keys k0 + st*32 + acc_row(.) + rowoff ..: xrow already holds the row and rowoff / 2
          rng_lin_run<4>(xrow + (uint32_t)((k0 + st * 32 + acc_row(8 * s2 + 4 * g)) >> 1) * RNG_C1, dr);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int r = 8 * s2 + 4 * g + j;
            sacc[st][r] = dr[j] < p.thr ? 0.f : sacc[st][r] * p.keep_scale;
          }
        }
        pf = acc_to_frag(sacc[st], s2);
      }
#pragma unroll
      for (int dt = 0; dt < DHP / 32; dt++)
        oacc[dt] = MFMA(frag_cols<VS, VW>(sV, st * 32, s2, dt * 32, lane), pf, oacc[dt], 0, 0, 0);
    }
}

template <int DHP, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, ((DHP <= 64 && NW == 4 && (!DROP || DROP_OCC_FWD == 3)) ? 3 : DHP <= 96 ? 2 : 1)) void attn_fwd_k(AttnP p) {
#ifdef EXP_STAMP   // diagnostic build: cycle stamps of wave 0 per block go to the lse buffer (tools/stamp_attn.py reads them)
  const uint64_t st0 = __builtin_amdgcn_s_memtime();
#endif
  constexpr int NT = NW * 64, CPR = DHP / 8, NCH = (64 * CPR + NT - 1) / NT;
  constexpr int KS = Strides<DHP>::ROWS, VS = Strides<DHP>::COLS, TILE = 64 * KS + 64 * VS;
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE];   // double buffer: one barrier per key tile
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), Dh = (int)p.Dh, S = (int)p.S, Sk = (int)p.Sk;
  const HeadBlock hb = head_block(p, p.S, NW * 32);
  if (!hb.valid) return;
  const int64_t b = hb.b, h = hb.h;
  const int qrow = hb.xb * (NW * 32) + wid * 32 + (lane & 31);
  const bool wave_on = hb.xb * (NW * 32) + wid * 32 < S;   // else this wave only helps staging
  const bf16* qp = (const bf16*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16* kp = (const bf16*)p.k + b * p.k_sb + h * p.k_sh;
  const bf16* vp = (const bf16*)p.v + b * p.k_sb + h * p.k_sh;

  const TileSrc<NCH> ksrc = make_tilesrc<DHP, NT, NCH>(kp, p.k_ss, Sk, Dh, tid);
  const TileSrc<NCH> vsrc = make_tilesrc<DHP, NT, NCH>(vp, p.k_ss, Sk, Dh, tid);
  // two register sets: tile t+2 is requested while tile t is multiplied, so a load has two tile periods to arrive
  const int ntiles = (Sk + 63) >> 6, nfull = Sk >> 6;
  // The stationary fragments are requested before the first tile: loads return in order, so the wait that precedes
  // staging tile 0 also retires them, and nothing inside the loop waits on a load the loop itself has just issued.
  bf16x8 qf[DHP / 16];
  load_stationary<DHP>(qf, qp, p.q_ss, qrow, S, Dh, lane);
  uint4 rkA[NCH], rvA[NCH], rkB[NCH], rvB[NCH];
  stage_load(rkA, ksrc, 0);
  stage_load(rvA, vsrc, 0);
  stage_load(rkB, ksrc, 1);   // tiles beyond the sequence read as zeros, so prefetches need no condition
  stage_load(rvB, vsrc, 1);
  const f32x16 zero = ZERO16;
  f32x16 oacc[DHP / 32], lacc = zero;
#pragma unroll
  for (int dt = 0; dt < DHP / 32; dt++) oacc[dt] = zero;
  float m = -INFINITY;
  const float c = p.scale * LOG2E;
  // hash input of (this lane's query row, key pair rowoff / 2): row * (ld / 2) * C1 + (rowoff / 2) * C1 + mix(seed)
  const uint32_t rbase = drop_xrow(p, (uint64_t)(b * p.H + h) * p.S + (uint64_t)qrow, lane);
  stage_store<DHP, NT, NCH, KS, false>(rkA, smem, tid);
  stage_store<DHP, NT, NCH, VS, false>(rvA, smem + 64 * KS, tid);
  __syncthreads();
#ifdef EXP_STAMP
  const uint64_t st1 = __builtin_amdgcn_s_memtime();
#endif

  // full 64-key tiles, then the tail tile (staged by the last step, or above when there are fewer than 64 keys).
  // step(t): nk/nv hold tile t+1, fk/fv are free for tile t+2.
  auto step = [&](int t, uint4 (&nk)[NCH], uint4 (&nv)[NCH], uint4 (&fk)[NCH], uint4 (&fv)[NCH]) {
    const char* sK = smem + (t & 1) * TILE;
    const char* sV = sK + 64 * KS;
    stage_load(fk, ksrc, t + 2);   // unconditional: a conditional load would make every later wait a full drain
    stage_load(fv, vsrc, t + 2);
    if (wave_on) fwd_tile<DHP, 2, false, DROP, KS, VS>(p, sK, sV, qf, oacc, lacc, m, c, t * 64, Sk, rbase, lane);
    char* nK = smem + ((t + 1) & 1) * TILE;
    stage_store<DHP, NT, NCH, KS, false>(nk, nK, tid);
    stage_store<DHP, NT, NCH, VS, false>(nv, nK + 64 * KS, tid);
    __syncthreads();
  };
  for (int t = 0; t < nfull; t += 2) {
    step(t, rkB, rvB, rkA, rvA);
    if (t + 1 < nfull) step(t + 1, rkA, rvA, rkB, rvB);
  }
#ifdef EXP_STAMP
  const uint64_t st2 = __builtin_amdgcn_s_memtime();
#endif
  if (nfull < ntiles && wave_on) {
    const char* sK = smem + (nfull & 1) * TILE;
    const char* sV = sK + 64 * KS;
    if (Sk - nfull * 64 > 32) fwd_tile<DHP, 2, true, DROP, KS, VS>(p, sK, sV, qf, oacc, lacc, m, c, nfull * 64, Sk, rbase, lane);
    else fwd_tile<DHP, 1, true, DROP, KS, VS>(p, sK, sV, qf, oacc, lacc, m, c, nfull * 64, Sk, rbase, lane);
  }
#ifdef EXP_STAMP
  const uint64_t st3 = __builtin_amdgcn_s_memtime();
#endif
  if (qrow >= S) return;
  const float l = lacc[0];
  store_rows<DHP>((bf16*)p.o + b * p.o_sb + h * p.o_sh + (int64_t)qrow * p.o_ss, oacc, 1.f / l, Dh, lane);
#ifndef EXP_STAMP
  if (lane < 32) p.lse[(b * p.H + h) * p.S + qrow] = m * p.scale + __logf(l);
#else
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t st4 = __builtin_amdgcn_s_memtime();
  if (tid == 0) {
    uint32_t* dbg = (uint32_t*)p.lse + (int64_t)blockIdx.x * 8;
    dbg[0] = (uint32_t)st0; dbg[1] = (uint32_t)(st0 >> 32); dbg[2] = (uint32_t)(st1 - st0); dbg[3] = (uint32_t)(st2 - st1);
    dbg[4] = (uint32_t)(st3 - st2); dbg[5] = (uint32_t)(st4 - st3);
    dbg[6] = __builtin_amdgcn_s_getreg(63492); dbg[7] = __builtin_amdgcn_s_getreg(6164);
  }
#endif
}

// ---- backward: dQ -------------------------------------------------------------------------------
template <int DHP, int HALVES, bool PARTIAL, bool DROP, int KS, int VS, int KW = 1, int VW = 0>
__device__ __forceinline__ void dq_tile(const AttnP& p, const char* sK, const char* sV, const bf16x8 (&qf)[DHP / 16],
                                        const bf16x8 (&gf)[DHP / 16], f32x16 (&dqacc)[DHP / 32], float nlse2, float delta,
                                        float c, int k0, int S, uint32_t xrow, int lane) {
  const f32x16 zero = ZERO16;
  const int rowoff = 4 * (lane >> 5);
#pragma unroll
  for (int st = 0; st < HALVES; st++) {
    f32x16 sacc = MFMA(frag_rows<KS, KW>(sK, st * 32, 0, lane), qf[0], zero, 0, 0, 0);
    f32x16 pacc = MFMA(frag_rows<VS, VW>(sV, st * 32, 0, lane), gf[0], zero, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < DHP / 16; ks++) {
      sacc = MFMA(frag_rows<KS, KW>(sK, st * 32, ks, lane), qf[ks], sacc, 0, 0, 0);
      pacc = MFMA(frag_rows<VS, VW>(sV, st * 32, ks, lane), gf[ks], pacc, 0, 0, 0);
    }
    // dS^T[key][q] = P (dP - delta); keys beyond S contribute nothing
#pragma unroll
    for (int g = 0; g < 4; g++) {   // registers 4 g .. 4 g + 3: four consecutive keys from a multiple of 4
      uint32_t dr[4] = {0u, 0u, 0u, 0u};
      if (DROP) rng_lin_run<4>(xrow + (uint32_t)((k0 + st * 32 + acc_row(4 * g)) >> 1) * RNG_C1, dr);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = 4 * g + j;
        const int key = k0 + st * 32 + acc_row(r) + rowoff;
        float pr = fast_exp2(fmaf(sacc[r], c, nlse2));
        if (PARTIAL && st == HALVES - 1) pr = key < S ? pr : 0.f;
        float dp = pacc[r];
        if (DROP) dp = dr[j] >= p.thr ? dp * p.keep_scale : 0.f;
        sacc[r] = pr * (dp - delta);
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
      const bf16x8 df = acc_to_frag(sacc, s2);
#pragma unroll
      for (int dt = 0; dt < DHP / 32; dt++)
        dqacc[dt] = MFMA(frag_cols<KS, KW>(sK, st * 32, s2, dt * 32, lane), df, dqacc[dt], 0, 0, 0);
    }
  }
}

template <int DHP, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, ((DHP <= 64 && NW == 4 && (!DROP || DROP_OCC_DQ == 3)) ? 3 : DHP <= 96 ? 2 : 1)) void attn_dq_k(AttnP p) {
  constexpr int NT = NW * 64, CPR = DHP / 8, NCH = (64 * CPR + NT - 1) / NT;
  constexpr int KS = Strides<DHP>::COLS, VS = Strides<DHP>::ROWS;  // K is read both ways (swizzled), V row-wise only
  constexpr int TILE = 64 * KS + 64 * VS;
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), Dh = (int)p.Dh, S = (int)p.S, Sk = (int)p.Sk;
  const HeadBlock hb = head_block(p, p.S, NW * 32);
  if (!hb.valid) return;
  const int64_t b = hb.b, h = hb.h;
  const int qrow = hb.xb * (NW * 32) + wid * 32 + (lane & 31);
  const bool qv = qrow < S;
  const bool wave_on = hb.xb * (NW * 32) + wid * 32 < S;
  const bf16* qp = (const bf16*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16* kp = (const bf16*)p.k + b * p.k_sb + h * p.k_sh;
  const bf16* vp = (const bf16*)p.v + b * p.k_sb + h * p.k_sh;
  const bf16* gp = (const bf16*)p.dout + b * p.o_sb + h * p.o_sh;

  const TileSrc<NCH> ksrc = make_tilesrc<DHP, NT, NCH>(kp, p.k_ss, Sk, Dh, tid);
  const TileSrc<NCH> vsrc = make_tilesrc<DHP, NT, NCH>(vp, p.k_ss, Sk, Dh, tid);
  const int ntiles = (Sk + 63) >> 6, nfull = Sk >> 6;
  // stationary operands first (see attn_fwd_k)
  bf16x8 qf[DHP / 16], gf[DHP / 16], of[DHP / 16];
  load_stationary<DHP>(qf, qp, p.q_ss, qrow, S, Dh, lane);
  load_stationary<DHP>(gf, gp, p.o_ss, qrow, S, Dh, lane);
  load_stationary<DHP>(of, (const bf16*)p.o + b * p.o_sb + h * p.o_sh, p.o_ss, qrow, S, Dh, lane);
  const int64_t srow = (b * p.H + h) * p.S + qrow;
  const float nlse2 = qv ? -p.lse[srow] * LOG2E : 0.f;
  uint4 rkA[NCH], rvA[NCH], rkB[NCH], rvB[NCH];
  stage_load(rkA, ksrc, 0);
  stage_load(rvA, vsrc, 0);
  stage_load(rkB, ksrc, 1);   // tiles beyond the sequence read as zeros, so prefetches need no condition
  stage_load(rvB, vsrc, 1);
  // delta = rowsum(dO * O), computed here (each half-wave holds half of the row) and published for attn_dkdv_k,
  // which is launched after this kernel on the same stream
  float delta = 0.f;
  {
#pragma unroll
    for (int ks = 0; ks < DHP / 16; ks++)
#pragma unroll
      for (int j = 0; j < 8; j++) delta += (float)of[ks][j] * (float)gf[ks][j];
    delta += __shfl_xor(delta, 32, 64);
    if (qv && lane < 32) p.delta[srow] = delta;
  }
  const float c = p.scale * LOG2E;
  const uint32_t rbase = drop_xrow(p, (uint64_t)srow, lane);
  const f32x16 zero = ZERO16;
  f32x16 dqacc[DHP / 32];
#pragma unroll
  for (int dt = 0; dt < DHP / 32; dt++) dqacc[dt] = zero;
  stage_store<DHP, NT, NCH, KS, true>(rkA, smem, tid);
  stage_store<DHP, NT, NCH, VS, false>(rvA, smem + 64 * KS, tid);
  __syncthreads();

  auto step = [&](int t, uint4 (&nk)[NCH], uint4 (&nv)[NCH], uint4 (&fk)[NCH], uint4 (&fv)[NCH]) {
    const char* sK = smem + (t & 1) * TILE;
    const char* sV = sK + 64 * KS;
    stage_load(fk, ksrc, t + 2);   // unconditional: a conditional load would make every later wait a full drain
    stage_load(fv, vsrc, t + 2);
    if (wave_on) dq_tile<DHP, 2, false, DROP, KS, VS>(p, sK, sV, qf, gf, dqacc, nlse2, delta, c, t * 64, Sk, rbase, lane);
    char* nK = smem + ((t + 1) & 1) * TILE;
    stage_store<DHP, NT, NCH, KS, true>(nk, nK, tid);
    stage_store<DHP, NT, NCH, VS, false>(nv, nK + 64 * KS, tid);
    __syncthreads();
  };
  for (int t = 0; t < nfull; t += 2) {
    step(t, rkB, rvB, rkA, rvA);
    if (t + 1 < nfull) step(t + 1, rkA, rvA, rkB, rvB);
  }
  if (nfull < ntiles && wave_on) {
    const char* sK = smem + (nfull & 1) * TILE;
    const char* sV = sK + 64 * KS;
    if (Sk - nfull * 64 > 32) dq_tile<DHP, 2, true, DROP, KS, VS>(p, sK, sV, qf, gf, dqacc, nlse2, delta, c, nfull * 64, Sk, rbase, lane);
    else dq_tile<DHP, 1, true, DROP, KS, VS>(p, sK, sV, qf, gf, dqacc, nlse2, delta, c, nfull * 64, Sk, rbase, lane);
  }
  if (!qv) return;
  bf16* dst = (bf16*)p.dq + b * p.g_sb + h * p.g_sh + (int64_t)qrow * p.g_ss;
  if constexpr (DHP == 64) {
    if (p.rope) {
      store_rows_rope64(dst, dqacc, p.scale, lane, p.rope, qrow - p.rope_prefix, S - p.rope_prefix, p.rope_emu);
      return;
    }
  }
  store_rows<DHP>(dst, dqacc, p.scale, Dh, lane);
}

// ---- backward: dK, dV ---------------------------------------------------------------------------
// Query rows at or beyond S carry -lse = -inf, so their probability (and with it dS) is exactly 0.
// Without dropout the per-query constants ride in as the INITIAL accumulators of the two score products: the stationary
// K fragments are pre-multiplied by scale * log2(e), so S' = (c K) Q^T - lse * log2(e) leaves its MFMA chain ready for
// exp2, and dP' = dO V^T - delta leaves its chain ready to be multiplied by P.  That takes the fused multiply-add and the
// subtraction out of every element's vector work (the constants are accumulator rows here, so they cost no registers:
// the four LDS reads that used to feed the arithmetic feed the chains instead).  sL holds -lse * log2(e) [64] and,
// without dropout, -delta [64] (otherwise +delta, and the arithmetic stays per element).  Head sizes above 64 keep the
// per-element form as well: their instances are at the register limit, and two more live accumulator tiles at the head
// of the chains spill (head_dim 80, S = 256: 0.19 against 0.165 ms).
template <int DHP, bool DROP> constexpr bool dkdv_init_consts() { return !DROP && DHP <= 64; }
template <int DHP, int HALVES, bool DROP, int QS, int QW = 1>
__device__ __forceinline__ void dkdv_tile(const AttnP& p, const char* sQ, const char* sG, const float* sL,
                                          const bf16x8 (&kf)[DHP / 16], const bf16x8 (&vf)[DHP / 16],
                                          f32x16 (&dkacc)[DHP / 32], f32x16 (&dvacc)[DHP / 32], float c, int q0,
                                          int64_t sbase, int krow, int lane) {
  const f32x16 zero = ZERO16;
  const int rowoff = 4 * (lane >> 5);
  constexpr bool INIT = dkdv_init_consts<DHP, DROP>();
  // dropout draws: element (query, key) is pair query * (ld / 2) + key / 2, half key & 1
  const uint32_t xstep = (uint32_t)(drop_ld(p.Sk) >> 1) * RNG_C1;
  const bool kodd = krow & 1;
  const uint32_t xkey = (uint32_t)((uint64_t)(sbase + q0 + rowoff + (kodd ? 1 : 0)) * (drop_ld(p.Sk) >> 1)) * RNG_C1 +
                        (uint32_t)(krow >> 1) * RNG_C1 + rng_seedmix(p.seed);
  const int khalf = kodd ? 0 : 16;
#pragma unroll
  for (int st = 0; st < HALVES; st++) {
    // sacc[r] = S[q = q0 + st*32 + row(r)][key = lane&31], pacc likewise for dP
    f32x16 sinit = zero, pinit = zero;
    if (INIT) {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int ql = st * 32 + 8 * g + rowoff;
        const f32x4 l4 = *(const f32x4*)(sL + ql);
        const f32x4 d4 = *(const f32x4*)(sL + 64 + ql);
#pragma unroll
        for (int j = 0; j < 4; j++) { sinit[4 * g + j] = l4[j]; pinit[4 * g + j] = d4[j]; }
      }
    }
    f32x16 sacc = MFMA(frag_rows<QS, QW>(sQ, st * 32, 0, lane), kf[0], sinit, 0, 0, 0);
    f32x16 pacc = MFMA(frag_rows<QS, QW>(sG, st * 32, 0, lane), vf[0], pinit, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < DHP / 16; ks++) {
      sacc = MFMA(frag_rows<QS, QW>(sQ, st * 32, ks, lane), kf[ks], sacc, 0, 0, 0);
      pacc = MFMA(frag_rows<QS, QW>(sG, st * 32, ks, lane), vf[ks], pacc, 0, 0, 0);
    }
    if (INIT) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float pr = fast_exp2(sacc[r]);
        sacc[r] = pr;                  // P for dV
        pacc[r] = pr * pacc[r];        // dS for dK
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int ql = st * 32 + 8 * g + rowoff;
        const f32x4 l4 = *(const f32x4*)(sL + ql);
        const f32x4 d4 = *(const f32x4*)(sL + 64 + ql);
        uint32_t drops[4] = {0u, 0u, 0u, 0u};
        if (DROP) {
          // The lanes of keys 2i and 2i + 1 draw from the same hash of (query row, key pair i): the even lane computes the
          // hashes of this group's rows 0 and 2, the odd lane those of rows 1 and 3, and a neighbour swap hands each lane
          // the other two.  xkey holds the lane's own first row already (its parity folded in).
#pragma unroll
          for (int jj = 0; jj < 2; jj++) {
            const uint32_t mine = rng_fmix(xkey + (uint32_t)(st * 32 + 8 * g + 2 * jj) * xstep);
            const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
            const uint32_t h0 = kodd ? other : mine, h1 = kodd ? mine : other;   // hashes of rows 2 jj and 2 jj + 1
            drops[2 * jj] = (h0 << khalf) & 0xffff0000u;
            drops[2 * jj + 1] = (h1 << khalf) & 0xffff0000u;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int r = 4 * g + j;
          const float pr = fast_exp2(fmaf(sacc[r], c, l4[j]));
          float pd = pr, dp = pacc[r];
          if (DROP) {
            const bool keep = drops[j] >= p.thr;
            pd = keep ? pr * p.keep_scale : 0.f;
            dp = keep ? dp * p.keep_scale : 0.f;
          }
          sacc[r] = pd;                  // P (dropped) for dV
          pacc[r] = pr * (dp - d4[j]);   // dS for dK
        }
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
      const bf16x8 pf = acc_to_frag(sacc, s2);
      const bf16x8 df = acc_to_frag(pacc, s2);
#pragma unroll
      for (int dt = 0; dt < DHP / 32; dt++) {
        dvacc[dt] = MFMA(frag_cols<QS, QW>(sG, st * 32, s2, dt * 32, lane), pf, dvacc[dt], 0, 0, 0);
        dkacc[dt] = MFMA(frag_cols<QS, QW>(sQ, st * 32, s2, dt * 32, lane), df, dkacc[dt], 0, 0, 0);
      }
    }
  }
}

// AHEAD = query tiles in flight from global memory to LDS.  Two (one being written to LDS while the next is on its way)
// keep a short sequence's few tiles fed; one needs half the staging registers, which is what lets the head_dim 65-96
// instance keep two waves per SIMD without spilling, and a tile's arithmetic there is long enough to cover a load.
template <int DHP, int NW, bool DROP>
constexpr int dkdv_ahead() { return (DHP > 64 && DHP <= 96 && !DROP && NW == 4) ? 1 : 2; }
template <int DHP, int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, (((DHP <= 64 && (!DROP || (DROP_OCC_DKDV == 2 && NW == 4))) || (DHP <= 96 && NW == 4 && !DROP)) ? 2 : 1)) void attn_dkdv_k(AttnP p) {
  constexpr int AHEAD = dkdv_ahead<DHP, NW, DROP>();
  constexpr int NT = NW * 64, CPR = DHP / 8, NCH = (64 * CPR + NT - 1) / NT;
  constexpr int QS = Strides<DHP>::COLS;   // Q and dO tiles are read both ways (swizzled)
  constexpr int TILE = 2 * 64 * QS + 512;  // Q, dO, then -lse*log2e [64] and delta [64]
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), Dh = (int)p.Dh, S = (int)p.S, Sk = (int)p.Sk;
  const HeadBlock hb = head_block(p, p.Sk, NW * 32);
  if (!hb.valid) return;
  const int64_t b = hb.b, h = hb.h;
  const int krow = hb.xb * (NW * 32) + wid * 32 + (lane & 31);
  const bool wave_on = hb.xb * (NW * 32) + wid * 32 < Sk;
  const bf16* qp = (const bf16*)p.q + b * p.q_sb + h * p.q_sh;
  const bf16* kp = (const bf16*)p.k + b * p.k_sb + h * p.k_sh;
  const bf16* vp = (const bf16*)p.v + b * p.k_sb + h * p.k_sh;
  const bf16* gp = (const bf16*)p.dout + b * p.o_sb + h * p.o_sh;
  const int64_t sbase = (b * p.H + h) * p.S;

  const TileSrc<NCH> qsrc = make_tilesrc<DHP, NT, NCH>(qp, p.q_ss, S, Dh, tid);
  const TileSrc<NCH> gsrc = make_tilesrc<DHP, NT, NCH>(gp, p.o_ss, S, Dh, tid);
  const int ntiles = (S + 63) >> 6, nfull = S >> 6;
  uint4 rqA[NCH], rgA[NCH], rqB[AHEAD == 2 ? NCH : 1], rgB[AHEAD == 2 ? NCH : 1];
  float rlA = 0.f, rlB = 0.f;
  // per-row constants of a query tile: wave 0 fetches -lse*log2e, wave 1 delta (branch-free buffer loads).
  // Rows beyond S get -lse = -inf, which makes their probability exactly 0.
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lse + sbase), 0, (int)(S * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.delta + sbase), 0, (int)(S * 4), 0x00020000);
  auto load_row_consts = [&](float& rl, int q0) {
    const int qq = q0 + (tid & 63);
    const uint32_t off = wid < 2 ? (uint32_t)qq * 4u : OOB_OFF;   // wid is wave-uniform: scalar descriptor select
    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wid == 0 ? lrs : drs, off, 0, 0));
    rl = wid == 0 ? (qq < S ? -v * LOG2E : -INFINITY) : (dkdv_init_consts<DHP, DROP>() ? -v : v);   // see dkdv_tile
  };
  bf16x8 kf[DHP / 16], vf[DHP / 16];   // stationary operands first (see attn_fwd_k)
  load_stationary<DHP>(kf, kp, p.k_ss, krow, Sk, Dh, lane);
  load_stationary<DHP>(vf, vp, p.k_ss, krow, Sk, Dh, lane);
  stage_load(rqA, qsrc, 0);
  stage_load(rgA, gsrc, 0);
  load_row_consts(rlA, 0);
  if constexpr (AHEAD == 2) {
    stage_load(rqB, qsrc, 1);
    stage_load(rgB, gsrc, 1);
    load_row_consts(rlB, 64);
  }
  const float c = p.scale * LOG2E;
  if (dkdv_init_consts<DHP, DROP>()) {   // the score product then comes out in exp2 units (dkdv_tile)
#pragma unroll
    for (int ks = 0; ks < DHP / 16; ks++)
#pragma unroll
      for (int j = 0; j < 8; j++) kf[ks][j] = (bf16)((float)kf[ks][j] * c);
  }
  const f32x16 zero = ZERO16;
  f32x16 dkacc[DHP / 32], dvacc[DHP / 32];
#pragma unroll
  for (int dt = 0; dt < DHP / 32; dt++) { dkacc[dt] = zero; dvacc[dt] = zero; }
  stage_store<DHP, NT, NCH, QS, true>(rqA, smem, tid);
  stage_store<DHP, NT, NCH, QS, true>(rgA, smem + 64 * QS, tid);
  if (tid < 128) ((float*)(smem + 2 * 64 * QS))[tid] = rlA;
  __syncthreads();

  if constexpr (AHEAD == 2) {
    auto step = [&](int t, uint4 (&nq)[NCH], uint4 (&ng)[NCH], float& nl, uint4 (&fq)[NCH], uint4 (&fg)[NCH], float& fl) {
      const char* sQ = smem + (t & 1) * TILE;
      const char* sG = sQ + 64 * QS;
      const float* sL = (const float*)(sQ + 2 * 64 * QS);
      stage_load(fq, qsrc, t + 2);
      stage_load(fg, gsrc, t + 2);
      load_row_consts(fl, (t + 2) * 64);
      if (wave_on) dkdv_tile<DHP, 2, DROP, QS>(p, sQ, sG, sL, kf, vf, dkacc, dvacc, c, t * 64, sbase, krow, lane);
      char* nQ = smem + ((t + 1) & 1) * TILE;
      stage_store<DHP, NT, NCH, QS, true>(nq, nQ, tid);
      stage_store<DHP, NT, NCH, QS, true>(ng, nQ + 64 * QS, tid);
      if (tid < 128) ((float*)(nQ + 2 * 64 * QS))[tid] = nl;
      __syncthreads();
    };
    for (int t = 0; t < nfull; t += 2) {
      step(t, rqB, rgB, rlB, rqA, rgA, rlA);
      if (t + 1 < nfull) step(t + 1, rqA, rgA, rlA, rqB, rgB, rlB);
    }
  } else {
    for (int t = 0; t < nfull; t++) {   // tile t + 1 travels while tile t is multiplied
      const char* sQ = smem + (t & 1) * TILE;
      const char* sG = sQ + 64 * QS;
      const float* sL = (const float*)(sQ + 2 * 64 * QS);
      stage_load(rqA, qsrc, t + 1);   // tiles beyond the sequence read as zeros
      stage_load(rgA, gsrc, t + 1);
      load_row_consts(rlA, (t + 1) * 64);
      if (wave_on) dkdv_tile<DHP, 2, DROP, QS>(p, sQ, sG, sL, kf, vf, dkacc, dvacc, c, t * 64, sbase, krow, lane);
      char* nQ = smem + ((t + 1) & 1) * TILE;
      stage_store<DHP, NT, NCH, QS, true>(rqA, nQ, tid);
      stage_store<DHP, NT, NCH, QS, true>(rgA, nQ + 64 * QS, tid);
      if (tid < 128) ((float*)(nQ + 2 * 64 * QS))[tid] = rlA;
      __syncthreads();
    }
  }
  if (nfull < ntiles && wave_on) {
    const char* sQ = smem + (nfull & 1) * TILE;
    const char* sG = sQ + 64 * QS;
    const float* sL = (const float*)(sQ + 2 * 64 * QS);
    if (S - nfull * 64 > 32) dkdv_tile<DHP, 2, DROP, QS>(p, sQ, sG, sL, kf, vf, dkacc, dvacc, c, nfull * 64, sbase, krow, lane);
    else dkdv_tile<DHP, 1, DROP, QS>(p, sQ, sG, sL, kf, vf, dkacc, dvacc, c, nfull * 64, sbase, krow, lane);
  }
  if (krow >= Sk) return;
  store_rows<DHP>((bf16*)p.dv + b * p.gk_sb + h * p.gk_sh + (int64_t)krow * p.gk_ss, dvacc, 1.f, Dh, lane);
  bf16* dst = (bf16*)p.dk + b * p.gk_sb + h * p.gk_sh + (int64_t)krow * p.gk_ss;
  if constexpr (DHP == 64) {
    if (p.rope) {
      store_rows_rope64(dst, dkacc, p.scale, lane, p.rope, krow - p.rope_prefix, Sk - p.rope_prefix, p.rope_emu);
      return;
    }
  }
  store_rows<DHP>(dst, dkacc, p.scale, Dh, lane);
}

#undef MFMA

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static AttnP make_p(const VitkAttn* a) {
  AttnP p;
  p.q = a->q; p.k = a->k; p.v = a->v; p.dout = a->dout; p.o = a->o; p.dq = a->dq; p.dk = a->dk; p.dv = a->dv;
  p.lse = a->lse; p.delta = a->delta; p.B = a->B; p.H = a->H; p.S = a->S; p.Dh = a->Dh;
  p.q_sb = a->q_sb; p.q_sh = a->q_sh; p.q_ss = a->q_ss; p.o_sb = a->o_sb; p.o_sh = a->o_sh; p.o_ss = a->o_ss;
  p.g_sb = a->dq_sb; p.g_sh = a->dq_sh; p.g_ss = a->dq_ss;
  const bool cross = a->Sk > 0;
  p.Sk = cross ? a->Sk : a->S;
  p.k_sb = cross ? a->kv_sb : a->q_sb; p.k_sh = cross ? a->kv_sh : a->q_sh; p.k_ss = cross ? a->kv_ss : a->q_ss;
  p.gk_sb = cross ? a->dkv_sb : a->dq_sb; p.gk_sh = cross ? a->dkv_sh : a->dq_sh; p.gk_ss = cross ? a->dkv_ss : a->dq_ss;
  p.seed = a->seed; p.scale = a->scale;
  p.thr = a->drop_p > 0.f ? drop_threshold(a->drop_p) : 0;
  p.keep_scale = 1.f / (1.f - a->drop_p);
  p.rope = a->rope; p.rope_prefix = (int)a->rope_prefix; p.rope_emu = a->rope_emulate_bf16;
  return p;
}
static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static bool attn_mfma_ok(const VitkAttn* a, bool bwd) {
  if (a->dtype != VITK_BF16 || a->Dh % 8 || a->Dh > 128 || a->Dh < 8) return false;
  const int64_t st[] = {a->q_sb, a->q_sh, a->q_ss, a->o_sb, a->o_sh, a->o_ss};
  for (int64_t s : st) if (s % 8) return false;
  if (!al16(a->q) || !al16(a->k) || !al16(a->v)) return false;
  if (!al16(a->o)) return false;
  // operand rows are addressed with 32-bit byte offsets inside one (batch, head) slice
  if (a->S * a->q_ss * 2 >= (1ll << 30) || a->S * a->o_ss * 2 >= (1ll << 30)) return false;
  if (a->Sk > 0) {   // cross-attention: keys and values with a row count and strides of their own
    if (a->kv_sb % 8 || a->kv_sh % 8 || a->kv_ss % 8 || a->Sk * a->kv_ss * 2 >= (1ll << 30)) return false;
    if (bwd && (a->dkv_sb % 8 || a->dkv_sh % 8 || a->dkv_ss % 8)) return false;
  }
  if (bwd) {
    if (a->dq_sb % 8 || a->dq_sh % 8 || a->dq_ss % 8) return false;
    if (!al16(a->dout) || !al16(a->dq) || !al16(a->dk) || !al16(a->dv)) return false;
  }
  return true;
}
static int attn_validate(const VitkAttn* a, bool bwd) {
  VITK_REQUIRE(a && a->q && a->k && a->v && a->o && a->lse, "attn: null pointer");
  VITK_REQUIRE(a->B >= 0 && a->H > 0 && a->S > 0 && a->Dh > 0 && a->Dh <= 128, "attn: bad shape B=%lld H=%lld S=%lld Dh=%lld",
               (long long)a->B, (long long)a->H, (long long)a->S, (long long)a->Dh);
  VITK_REQUIRE(a->H <= 65535 && a->B <= 65535, "attn: B and H must be <= 65535");
  VITK_REQUIRE(a->Sk >= 0, "attn: Sk=%lld must be 0 (self-attention) or the number of keys", (long long)a->Sk);
  VITK_REQUIRE(a->drop_p >= 0.f && a->drop_p < 1.f, "attn: dropout p=%f out of range", a->drop_p);
  VITK_REQUIRE(a->dtype == VITK_F32 || a->dtype == VITK_BF16, "attn: bad dtype");
  if (bwd) VITK_REQUIRE(a->dout && a->dq && a->dk && a->dv && a->delta, "attn_bwd: null pointer");
  return 0;
}
// waves (32 rows each) per workgroup: 4 unless the sequence is short.  The 4-wave instances are the ones tuned for
// occupancy (three waves per SIMD for head_dim <= 64 forward and dQ, two for the head_dim 65-96 dK/dV kernel), and sharing
// each staged tile among more waves outweighs the idle waves of a padded last block (measured at S = 272, head_dim 64:
// forward 0.150 ms with 4 waves, 0.178 ms with 3; backward 0.51 against 0.545; head_dim 80: backward 0.24 against 0.32).
// At S <= 128 the count in {4, 3, 2} that pads the sequence least.
static int pick_nw(int64_t S, int64_t Dh, bool fwd) {
  if (const char* f = getenv("VITK_ATTN_NW")) { const int v = atoi(f); if (v >= 2 && v <= 4) return v; }
  (void)fwd;
  if (Dh > 96 || S > 128) return 4;
  int best = 4;
  int64_t best_pad = ceil_div(S, 128) * 128 - S;
  for (int nw = 3; nw >= 2; nw--) {
    const int64_t pad = ceil_div(S, nw * 32) * nw * 32 - S;
    if (pad < best_pad) { best = nw; best_pad = pad; }
  }
  return best;
}

#define ATTN_DISPATCH(KERNEL, DHP, NW, grid, p, st)                                             \
  do {                                                                                          \
    if (p.thr) {                                                                                \
      if (NW == 4) hipLaunchKernelGGL((KERNEL<DHP, 4, true>), grid, dim3(256), 0, st, p);       \
      else if (NW == 3) hipLaunchKernelGGL((KERNEL<DHP, 3, true>), grid, dim3(192), 0, st, p);  \
      else hipLaunchKernelGGL((KERNEL<DHP, 2, true>), grid, dim3(128), 0, st, p);               \
    } else {                                                                                    \
      if (NW == 4) hipLaunchKernelGGL((KERNEL<DHP, 4, false>), grid, dim3(256), 0, st, p);      \
      else if (NW == 3) hipLaunchKernelGGL((KERNEL<DHP, 3, false>), grid, dim3(192), 0, st, p); \
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
    const int nw = pick_nw(a->S, a->Dh, true);
    dim3 grid(head_block_grid(a->B, a->H, a->S, nw * 32));
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
  if (a->rope)
    VITK_REQUIRE(can && a->algo != 1 && a->Dh == 64 && a->Sk == 0 && a->rope_prefix >= 0 && a->rope_prefix < a->S &&
                     ((uintptr_t)a->rope & 15) == 0,
                 "attn_bwd: the fused inverse rotation needs the MFMA kernels, self-attention, head_dim 64 and 0 <= rope_prefix < S");
  if (can && a->algo != 1) {
    const int nw = pick_nw(a->S, a->Dh, false), nwk = pick_nw(p.Sk, a->Dh, false);
    dim3 grid(head_block_grid(a->B, a->H, a->S, nw * 32)), gridk(head_block_grid(a->B, a->H, p.Sk, nwk * 32));
    ATTN_DISPATCH_DH(attn_dq_k, a->Dh, nw, grid, p, st);       // also writes delta
    ATTN_DISPATCH_DH(attn_dkdv_k, a->Dh, nwk, gridk, p, st);   // reads delta
  } else {
    if (a->dtype == VITK_F32) hipLaunchKernelGGL((attn_delta_k<float>), dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_delta_k<bf16>), dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, st, p);
    dim3 grid((unsigned)ceil_div(a->S, 64), (unsigned)a->H, (unsigned)a->B);
    dim3 gridk((unsigned)ceil_div(p.Sk, 64), (unsigned)a->H, (unsigned)a->B);
    const size_t lds = (size_t)2 * PT * a->Dh * sizeof(float);
    if (a->dtype == VITK_F32) {
      hipLaunchKernelGGL((attn_dq_simple_k<float>), grid, dim3(256), lds, st, p);
      hipLaunchKernelGGL((attn_dkdv_simple_k<float>), gridk, dim3(256), lds + 512, st, p);
    } else {
      hipLaunchKernelGGL((attn_dq_simple_k<bf16>), grid, dim3(256), lds, st, p);
      hipLaunchKernelGGL((attn_dkdv_simple_k<bf16>), gridk, dim3(256), lds + 512, st, p);
    }
  }
  VITK_LAUNCH_CHECK("attn_bwd");
  return 0;
}

extern "C" int vitk_attn_weights(const VitkAttn* a, float* weights, void* stream) {
  VITK_REQUIRE(a && a->q && a->k && weights, "attn_weights: null pointer");
  VITK_REQUIRE(a->B >= 0 && a->H > 0 && a->S > 0 && a->Dh > 0 && a->Sk >= 0, "attn_weights: bad shape");
  VITK_REQUIRE(a->dtype == VITK_F32 || a->dtype == VITK_BF16, "attn_weights: bad dtype");
  if (a->B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const AttnP p = make_p(a);
  const int64_t rows = a->B * a->H * a->S;
  const size_t lds = (size_t)4 * a->Dh * sizeof(float);
  if (a->dtype == VITK_F32) hipLaunchKernelGGL((attn_weights_k<float>), dim3((unsigned)ceil_div(rows, 4)), dim3(256), lds, st, p, weights);
  else hipLaunchKernelGGL((attn_weights_k<bf16>), dim3((unsigned)ceil_div(rows, 4)), dim3(256), lds, st, p, weights);
  VITK_LAUNCH_CHECK("attn_weights");
  return 0;
}
