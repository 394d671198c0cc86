// Shared pieces of the MFMA GEMM kernels: the epilogue description and functors, the tile walk, and the LDS-DMA
// staging helpers.  Included by gemm.hip (128x128 and 256x256 tiles) and gemm2.hip (256x128 tiles, two blocks per CU).
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((address_space(3))) char lds_char_t;
// A 16-byte store of write-once output.  A plain store leaves its line in the XCD's 4 MiB L2, where a tile's 128-512 KiB of
// output push out the operand panels that the XCD's other tiles are about to re-read; with `wt` the store carries sc1
// (write-through, the line is not kept).  Issued from inline assembly: the compiler does not count it in its vmcnt
// bookkeeping, so its own waits for loads can only come out stricter than needed, never laxer.
__device__ __forceinline__ void st16(void* p, u32x4_t v, int wt) {
  if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *(u32x4_t*)p = v;
}

struct Epi {
  void* out;
  const void* bias;
  const void* res;
  const void* gamma;
  const int64_t* row_idx;
  void* aux;
  const void* pos;
  int64_t ldo, N, tokens, seq, prefix;
  uint64_t seed;
  float alpha, keep_scale;
  uint32_t thr;
  int act, use_limit, use_extra;
  float limit, extra;
  void* asum;            // A^T-layout products: asum[m] = sum over k of A[k, m] (dtype of out), or NULL
  float* asum_partial;   // split-K: per-split partial sums [splits][M]
  int64_t m_off;         // row of the whole problem that this launch's row 0 is (dropout stream index)
  int pos_f32;           // PATCH: the position table holds fp32 values and is added without rounding (output fp32)
  const float* rope;     // ROPE: [2, seq - prefix, 64] sin then cos
  int64_t rope_width;    // ROPE: columns per section (q | k | v); sections 0 and 1 are rotated, heads of 64 columns
  int rope_emu;          // ROPE: round each product to bf16 before the sum (a bf16 table in the reference)
  // ROPE mode, per-head normalisation of q and k before the rotation (vit/attention.py:25-41): weights [2][64] (q then
  // k), biases [2][64] or null, rstd / mean [2][rows_total * heads], and aux = the un-normalised q and k [2][rows_total][width]
  const float* qk_w;
  const float* qk_b;
  float* qk_rstd;
  float* qk_mean;        // LayerNorm only
  float qk_eps;
  int qk_ln;
  int64_t rows_total;    // rows of the whole problem (a launch on the rows after a cut sees fewer)
  int wt;                // write the epilogue's 16-byte outputs through the L2 (sc1): see st16
  int panel_m;           // row tiles per panel of the tile walk: see tile_coords
  int stream;            // whole tiles of the 256x256 kernel take their epilogue operand by LDS-DMA (epilogue_stream16)
  int ntl;               // ask for the epilogue's read-once rows (saved pre-activations, residual) with the streaming policy
};

// sum of a fragment's 8 bf16 values added to s (4 x v_dot2c_f32_bf16 against ones)
__device__ __forceinline__ float frag_sum(const bf16x8& f, float s) {
  typedef __attribute__((ext_vector_type(2))) bf16 bf16x2_t;
  bf16x2_t ones;
  ones[0] = (bf16)1.f;
  ones[1] = (bf16)1.f;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    bf16x2_t t;
    t[0] = f[2 * j];
    t[1] = f[2 * j + 1];
    s = __builtin_amdgcn_fdot2_f32_bf16(t, ones, s, false);
  }
  return s;
}
// rows of the A operand summed over the block's K range, written by the waves that own the block's first column slice:
// lane l and l+32 hold the two k-halves of row (l & 31) of each 32-row group
template <int NI, int MODE, class TO>
__device__ __forceinline__ void asum_store(const Epi& e, const float (&rs)[NI], int64_t mrow0, int64_t M, int64_t split, int lane) {
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const float s = rs[i] + __shfl_xor(rs[i], 32, 64);
    const int64_t m = mrow0 + i * 32 + (lane & 31);
    if (lane < 32 && m < M) {
      if (MODE == -1) e.asum_partial[split * M + m] = s;
      else ((TO*)e.asum)[m] = from_f<TO>(s);
    }
  }
}


// The activation of a fused epilogue as a compile-time constant (ACTC >= 0: the caller has already switched on the
// code, once per tile) or switched on here (ACTC < 0).  A switch inside every row step lays all five activations'
// code out sixteen times per tile; with the switch hoisted the hot path of a tile is one activation's code.
#define ACT_DISPATCH(ACTC, code, ACT, ...)                  \
  if constexpr (ACTC >= 0) {                                \
    constexpr int ACT = ACTC;                               \
    __VA_ARGS__;                                            \
  } else {                                                  \
    ACT_SWITCH(code, ACT, __VA_ARGS__)                      \
  }

constexpr int EPI_ACT = VITK_EPI_ACT, EPI_GLU = VITK_EPI_GLU, EPI_DACT = VITK_EPI_DACT, EPI_DGLU = VITK_EPI_DGLU;
constexpr int EPI_ROPE = VITK_EPI_ROPE;

// Vector form used by the MFMA kernel: 4 consecutive columns n..n+3 of row m (n % 4 == 0).
//   STORE / RESIDUAL / PATCH as EpiOp above;
//   ACT   p = round(acc + bias) -> aux[m, n];  out[m, n] = dropout(act(p))                      (fc1, plain MLP)
//   DACT  out[m, n] = dropmask * acc * act'(aux[m, n])                                          (fc2 dgrad, plain MLP)
//   DGLU  acc = dh[m, f]; lin = aux[m, f], glu = aux[m, F + f];  out[m, f] = d lin, out[m, F + f] = d glu   (F = e.N)
// (GLU forward pairs two column groups and has its own routine below.)
// Per-row operands an epilogue reads from memory (residual row, position row, saved pre-activations).  They are
// fetched for several rows before any of them is used, so the loads overlap instead of serialising on s_waitcnt.
// Raw4<T>: 4 consecutive elements as loaded (no conversion until they are used, so a prefetched row costs 2 registers
// for bf16 and 4 for fp32).
template <class T> struct Raw4;
template <> struct Raw4<float> {
  float4 q;
  __device__ __forceinline__ void load(const float* p) { q = *(const float4*)p; }
  __device__ __forceinline__ void load(const float* p, int nt) {
    typedef __attribute__((ext_vector_type(4))) float f32x4n_t;
    if (nt) q = __builtin_bit_cast(float4, __builtin_nontemporal_load((const f32x4n_t*)p));
    else q = *(const float4*)p;
  }
  __device__ __forceinline__ float get(int j) const { return j == 0 ? q.x : j == 1 ? q.y : j == 2 ? q.z : q.w; }
};
template <> struct Raw4<bf16> {
  uint2 q;
  __device__ __forceinline__ void load(const bf16* p) { q = *(const uint2*)p; }
  __device__ __forceinline__ void load(const bf16* p, int) { q = *(const uint2*)p; }
  __device__ __forceinline__ float get(int j) const {
    const uint32_t w = j < 2 ? q.x : q.y;
    return __builtin_bit_cast(float, (j & 1) ? (w & 0xffff0000u) : (w << 16));
  }
};

template <int MODE, class TO>
struct EpiPre {
  Raw4<bf16> a, b;   // PATCH: position row; DACT: pre-activations; DGLU: linear and gate pre-activations
  Raw4<TO> r;        // RESIDUAL: the residual row
  int64_t row;
};

template <int MODE, class TO>
struct EpiVec {
  typedef bf16 TI;
  static __device__ __forceinline__ void prefetch(const Epi& e, int64_t m, int64_t n, EpiPre<MODE, TO>& p) {
    if (MODE == VITK_EPI_RESIDUAL) {
      p.row = e.row_idx ? e.row_idx[m] : m;
      p.r.load((const TO*)e.res + p.row * e.ldo + n, e.ntl);
    } else if (MODE == VITK_EPI_PATCH) {
      if (e.pos) {
        if (e.pos_f32) p.r.load((const TO*)e.pos + (m % e.tokens) * e.N + n);   // output type is float then
        else p.a.load((const TI*)e.pos + (m % e.tokens) * e.N + n);
      }
    } else if (MODE == EPI_DACT) {
      p.a.load((const TI*)e.aux + m * e.N + n);
    } else if (MODE == EPI_DGLU) {
      p.a.load((const TI*)e.aux + m * 2 * e.N + n);
      p.b.load((const TI*)e.aux + m * 2 * e.N + e.N + n);
    }
  }

  // v = accumulators of columns n..n+3 of row m; bias/gamma = this lane's 4 columns (loaded once per tile)
  static __device__ __forceinline__ void apply(const Epi& e, int64_t m, int64_t n, float (&v)[4], const float (&bias)[4],
                                               const float (&gamma)[4], const EpiPre<MODE, TO>& p) {
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] += bias[j];
    if (MODE == VITK_EPI_STORE) {
      Vec4<TO> o;
#pragma unroll
      for (int j = 0; j < 4; j++) o.v[j] = v[j];
      o.store((TO*)e.out + m * e.ldo + n);
    } else if (MODE == VITK_EPI_RESIDUAL) {
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = round_as<TI>(v[j]);
      if (e.thr) {   // one uniform branch for the vector, not one per element (a taken scalar branch costs a refetch)
        uint32_t dr[4];
        rng_run<4>(e.seed, (uint64_t)((m + e.m_off) * e.N + n), dr);   // n and N are multiples of 4
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = dr[j] >= e.thr ? round_as<TI>(v[j] * e.keep_scale) : 0.f;
      }
      if (e.aux) {
        Vec4<TI> a;
#pragma unroll
        for (int j = 0; j < 4; j++) a.v[j] = v[j];
        a.store((TI*)e.aux + m * e.N + n);
      }
      Vec4<TO> r;
      if (e.gamma) {
#pragma unroll
        for (int j = 0; j < 4; j++) r.v[j] = p.r.get(j) + e.alpha * round_as<TO>(v[j] * gamma[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) r.v[j] = p.r.get(j) + e.alpha * v[j];
      }
      if constexpr (sizeof(TO) == 4) st16((TO*)e.out + p.row * e.ldo + n, __builtin_bit_cast(u32x4_t, *(const f32x4*)r.v), 0);
      else r.store((TO*)e.out + p.row * e.ldo + n);
    } else if (MODE == VITK_EPI_PATCH) {
      const int64_t b = m / e.tokens, t = m % e.tokens;
      Vec4<TO> o;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        o.v[j] = round_as<TI>(v[j]);
        if (e.pos) o.v[j] = e.pos_f32 ? o.v[j] + p.r.get(j) : round_as<TI>(o.v[j] + p.a.get(j));
      }
      o.store((TO*)e.out + (b * e.seq + e.prefix + t) * e.ldo + n);
    } else if (MODE == EPI_ACT) {
      Vec4<TI> pa;
#pragma unroll
      for (int j = 0; j < 4; j++) pa.v[j] = v[j];
      pa.store((TI*)e.aux + m * e.N + n);
      Vec4<TO> o;
      ACT_SWITCH(e.act, ACT, {
        _Pragma("unroll")
        for (int j = 0; j < 4; j++) o.v[j] = act_apply_t<ACT>(round_as<TI>(v[j]));
      });
      if (e.thr) {
#pragma unroll
        for (int j = 0; j < 4; j++)
          o.v[j] = rng_u32(e.seed, (uint64_t)((m + e.m_off) * e.N + n + j)) >= e.thr ? round_as<TI>(o.v[j]) * e.keep_scale : 0.f;
      }
      o.store((TO*)e.out + m * e.ldo + n);
    } else if (MODE == EPI_DACT) {
      Vec4<TO> o;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        v[j] = round_as<TI>(v[j]);
        if (e.thr) v[j] = rng_u32(e.seed, (uint64_t)((m + e.m_off) * e.N + n + j)) >= e.thr ? v[j] * e.keep_scale : 0.f;
      }
      ACT_SWITCH(e.act, ACT, {
        _Pragma("unroll")
        for (int j = 0; j < 4; j++) o.v[j] = v[j] * act_grad_t<ACT>(p.a.get(j));
      });
      o.store((TO*)e.out + m * e.ldo + n);
    } else {  // EPI_DGLU: p.a = linear pre-activation, p.b = gate pre-activation
      const int64_t F = e.N;
      Vec4<TO> dl, dg;
      float pl[4], pg[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        pl[j] = p.a.get(j);
        pg[j] = p.b.get(j);
        v[j] = round_as<TI>(v[j]);
        if (e.thr) v[j] = rng_u32(e.seed, (uint64_t)((m + e.m_off) * F + n + j)) >= e.thr ? v[j] * e.keep_scale : 0.f;
        dl.v[j] = v[j];
        dg.v[j] = v[j];
        if (e.use_limit) {
          const float l = pl[j], gl = pg[j];
          dl.v[j] = (l >= -e.limit && l <= e.limit) ? dl.v[j] : 0.f;
          dg.v[j] = (gl <= e.limit) ? dg.v[j] : 0.f;
          pl[j] = fminf(fmaxf(l, -e.limit), e.limit);
          pg[j] = fminf(gl, e.limit);
        }
        if (e.use_extra) pl[j] = round_as<TI>(pl[j] + e.extra);
      }
      ACT_SWITCH(e.act, ACT, {
        _Pragma("unroll")
        for (int j = 0; j < 4; j++) {
          dl.v[j] *= round_as<TI>(act_apply_t<ACT>(pg[j]));
          dg.v[j] *= pl[j] * act_grad_t<ACT>(pg[j]);
        }
      });
      dl.store((TO*)e.out + m * e.ldo + n);
      dg.store((TO*)e.out + m * e.ldo + F + n);
    }
  }
};

// GLU forward on a pair of column groups: lin[4] = fc1 columns f..f+3, glu[4] = columns F+f..F+f+3 of row m.
//   aux[m, f] = lin + b, aux[m, F + f] = glu + b  (pre-activations, kept for backward);  out[m, f] = dropout(act(glu) * lin)
template <class TO>
__device__ __forceinline__ void epi_glu(const Epi& e, int64_t m, int64_t f, float (&lin)[4], float (&glu)[4],
                                        const float (&bl)[4], const float (&bg)[4]) {
  typedef bf16 TI;
  const int64_t F = e.N / 2;
#pragma unroll
  for (int j = 0; j < 4; j++) { lin[j] += bl[j]; glu[j] += bg[j]; }
  Vec4<TI> pl, pg;
#pragma unroll
  for (int j = 0; j < 4; j++) { pl.v[j] = lin[j]; pg.v[j] = glu[j]; }
  pl.store((TI*)e.aux + m * e.N + f);
  pg.store((TI*)e.aux + m * e.N + F + f);
  Vec4<TO> o;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    lin[j] = round_as<TI>(lin[j]);
    glu[j] = round_as<TI>(glu[j]);
    if (e.use_limit) { lin[j] = fminf(fmaxf(lin[j], -e.limit), e.limit); glu[j] = fminf(glu[j], e.limit); }
    if (e.use_extra) lin[j] = round_as<TI>(lin[j] + e.extra);
  }
  ACT_SWITCH(e.act, ACT, {
_Pragma("unroll")
    for (int j = 0; j < 4; j++) o.v[j] = round_as<TI>(act_apply_t<ACT>(glu[j])) * lin[j];
  });
  if (e.thr) {
#pragma unroll
    for (int j = 0; j < 4; j++)
      o.v[j] = rng_u32(e.seed, (uint64_t)((m + e.m_off) * F + f + j)) >= e.thr ? round_as<TI>(o.v[j]) * e.keep_scale : 0.f;
  }
  o.store((TO*)e.out + m * e.ldo + f);
}

template <int MODE, class TI, class TO>
struct EpiOp {
  static __device__ __forceinline__ void apply(const Epi& e, int64_t m, int64_t n, float acc) {
    float v = acc;
    if (e.bias) v += to_f<TI>(((const TI*)e.bias)[n]);
    if (MODE == VITK_EPI_STORE) {
      ((TO*)e.out)[m * e.ldo + n] = from_f<TO>(v);
    } else if (MODE == VITK_EPI_RESIDUAL) {
      v = round_as<TI>(v);  // the linear's output as the compute dtype holds it
      if (e.thr) v = rng_u32(e.seed, (uint64_t)((m + e.m_off) * e.N + n)) >= e.thr ? round_as<TI>(v * e.keep_scale) : 0.f;
      if (e.aux) ((TI*)e.aux)[m * e.N + n] = from_f<TI>(v);
      const int64_t row = e.row_idx ? e.row_idx[m] : m;
      if (e.gamma) v = round_as<TO>(v * to_f<TO>(((const TO*)e.gamma)[n]));
      const float r = to_f<TO>(((const TO*)e.res)[row * e.ldo + n]);
      ((TO*)e.out)[row * e.ldo + n] = from_f<TO>(r + e.alpha * v);
    } else {  // PATCH
      const int64_t b = m / e.tokens, t = m % e.tokens;
      v = round_as<TI>(v);
      if (e.pos) v = e.pos_f32 ? v + ((const float*)e.pos)[t * e.N + n] : round_as<TI>(v + to_f<TI>(((const TI*)e.pos)[t * e.N + n]));
      ((TO*)e.out)[(b * e.seq + e.prefix + t) * e.ldo + n] = from_f<TO>(v);
    }
  }
};


constexpr int BK = 64;
constexpr int MODE_PARTIAL = -1;       // split-K: raw fp32 accumulators to the workspace
constexpr int STG_LD = 68;             // epilogue staging: 32 rows x 64 floats per wave, rows padded to 68 floats

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// global -> registers for one operand tile. TR=false: rows x [k0,k0+64); TR=true: [k0,k0+64) x cols.
// GLU forward: the block's 128 "virtual" fc1 columns are, per 64-column wave slice, 32 linear columns followed by
// their 32 gate partners (weight rows F + f), so that a lane finds both halves of a pair in its own wave's tile.
__device__ __forceinline__ int64_t glu_row(int64_t v, int64_t F) {
  const int64_t f = (v >> 6) * 32 + (v & 31);
  return (v & 32) ? F + f : f;
}

// Operand tiles are fetched with buffer loads: rows/columns outside the matrix are given an out-of-range offset and
// come back as zeros, with no branch around the load.  voff = this thread's 4 chunk offsets (bytes) at the first
// K-step, or OOB when the chunk's row (K-contiguous operand) / column (contraction-major operand) is outside.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr uint32_t OOB = 0xffffffffu;
__device__ __forceinline__ uint32_t clip32(int64_t v) { return v > 0xfffffff0ll ? 0xfffffff0u : (uint32_t)v; }

// ------------------------------------------------------------------------------------------------
// Epilogue shared by the MFMA kernels.  Accumulator element r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column
// l&31 of a 32x32 tile; a wave transposes 32 rows x 64 columns at a time through its own LDS slab (row stride 68
// floats) so that every lane then owns 4 consecutive columns: 16-byte loads of bias/residual, 8/16-byte stores,
// whole 128-B lines per row.
// ------------------------------------------------------------------------------------------------
// ---- 8-column epilogues --------------------------------------------------------------------------
// bf16 outputs move 8 consecutive columns per lane: 16-byte loads and stores, half as many memory instructions per
// slab (the epilogues are paced by the issue of their stores).  Used by STORE with bf16 output, GLU and DGLU; their
// arithmetic (and rounding points) is that of the 4-column forms above.
template <int MODE, class TO>
constexpr bool epi_wide() {
  return MODE == EPI_GLU || MODE == EPI_DGLU || MODE == EPI_ACT || MODE == EPI_DACT || MODE == EPI_ROPE ||
         ((MODE == VITK_EPI_STORE || MODE == VITK_EPI_RESIDUAL) && sizeof(TO) == 2);
}

struct Raw8 {   // 8 bf16 as loaded
  uint4 q;
  __device__ __forceinline__ void load(const bf16* p) { q = *(const uint4*)p; }
  // nt: a row that is read once (saved pre-activations, residual rows) is asked for with the streaming policy
  __device__ __forceinline__ void load(const bf16* p, int nt) {
    if (nt) q = __builtin_bit_cast(uint4, __builtin_nontemporal_load((const u32x4_t*)p));
    else q = *(const uint4*)p;
  }
  __device__ __forceinline__ float get(int j) const {
    const uint32_t w = (j >> 1) == 0 ? q.x : (j >> 1) == 1 ? q.y : (j >> 1) == 2 ? q.z : q.w;
    return __builtin_bit_cast(float, (j & 1) ? (w & 0xffff0000u) : (w << 16));
  }
};
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8], int wt = 0) {
  typedef float f32x8_t __attribute__((ext_vector_type(8)));
  const f32x8_t f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  st16(p, __builtin_bit_cast(u32x4_t, __builtin_convertvector(f, bf16x8)), wt);   // four v_cvt_pk_bf16_f32
}

// Element offsets of a lane's 8 columns in one row of the output and of the saved pre-activations.  slab_finish forms them
// once per slab (one 64-bit multiply each) and moves them on by whole rows with scalar arithmetic; a 64-bit multiply per
// row and per pointer is several quarter-rate vector instructions in epilogues that are bound by vector issue.
struct RowOff {
  int64_t out, aux;
};

// ---- two values per instruction --------------------------------------------------------------------------------
// The 8-column GLU epilogues are bound by vector-instruction issue (a 256 x 256 DGLU tile costs 40 k cycles of
// arithmetic with its loads and stores compiled out), and gfx950 multiplies, adds and fmas two fp32 values in one
// v_pk_* instruction for the issue cost of one.  The sigmoid-gated activations are therefore written on pairs; every
// operation is the scalar one of act_apply_t / act_grad_t on each half, so the values are the same.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 unpack_bf16x2(uint32_t w) {
  f32x2 r;
  r.x = __builtin_bit_cast(float, w << 16);
  r.y = __builtin_bit_cast(float, w & 0xffff0000u);
  return r;
}
__device__ __forceinline__ uint32_t pack_bf16x2(f32x2 v) {   // one v_cvt_pk_bf16_f32 (element-wise casts give one per value)
  typedef __attribute__((ext_vector_type(2))) bf16 bf16x2_t;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ f32x2 round_bf16x2(f32x2 v) { return unpack_bf16x2(pack_bf16x2(v)); }
template <int ACT> constexpr bool act_is_sigmoid_gate() { return ACT == VITK_ACT_SILU || ACT == VITK_ACT_OPENSWIGLU; }
template <int ACT> __device__ __forceinline__ f32x2 sigmoid2(f32x2 x) {   // fast_sigmoid(x, a) on both halves
  constexpr float a = ACT == VITK_ACT_SILU ? 1.f : 1.702f;
  const f32x2 t = x * (-1.44269504088896340736f * a);
  f32x2 d;
  d.x = __builtin_amdgcn_exp2f(t.x);
  d.y = __builtin_amdgcn_exp2f(t.y);
  d = 1.f + d;
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x);
  r.y = __builtin_amdgcn_rcpf(d.y);
  return r;
}

// sigmoid2 on four pairs, stage by stage (all scalings, then all exponentials, the increments, all reciprocals).  Written
// pair by pair the compiler emits each pair's chain on its own, every instruction waiting for the one before it (an
// exponential's result is not ready for several cycles); the register fences between the stages keep four independent
// chains in flight.  Same operations per element, so the same values.
__device__ __forceinline__ void stage_fence(f32x2 (&t)[4]) {
  asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
}
template <int ACT> __device__ __forceinline__ void sigmoid8(const float (&x)[8], f32x2 (&r)[4]) {
  constexpr float a = ACT == VITK_ACT_SILU ? 1.f : 1.702f;
  f32x2 t[4];
#pragma unroll
  for (int p = 0; p < 4; p++) t[p] = f32x2{x[2 * p], x[2 * p + 1]} * (-1.44269504088896340736f * a);
  stage_fence(t);
#pragma unroll
  for (int p = 0; p < 4; p++) {
    t[p].x = __builtin_amdgcn_exp2f(t[p].x);
    t[p].y = __builtin_amdgcn_exp2f(t[p].y);
  }
  stage_fence(t);
#pragma unroll
  for (int p = 0; p < 4; p++) t[p] = 1.f + t[p];
  stage_fence(t);
#pragma unroll
  for (int p = 0; p < 4; p++) {
    r[p].x = __builtin_amdgcn_rcpf(t[p].x);
    r[p].y = __builtin_amdgcn_rcpf(t[p].y);
  }
  stage_fence(r);
}

// GLU forward, 8 pairs: lin[8] = fc1 columns f..f+7, glu[8] = columns F+f..F+f+7 of row m (see epi_glu)
template <int ACTC = -1>
__device__ __forceinline__ void epi_glu8(const Epi& e, int64_t m, int64_t f, const RowOff& ro, float (&lin)[8], float (&glu)[8],
                                         const float (&bl)[8], const float (&bg)[8]) {
  typedef bf16 TI;
  const int64_t F = e.N / 2;
#pragma unroll
  for (int j = 0; j < 8; j++) { lin[j] += bl[j]; glu[j] += bg[j]; }
  store8((TI*)e.aux + ro.aux, lin, 0);
  store8((TI*)e.aux + ro.aux + F, glu, 0);
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2 rl = round_bf16x2(f32x2{lin[j], lin[j + 1]}), rg = round_bf16x2(f32x2{glu[j], glu[j + 1]});
    lin[j] = rl.x; lin[j + 1] = rl.y;
    glu[j] = rg.x; glu[j + 1] = rg.y;
  }
  if (e.use_limit || e.use_extra) {   // off in the usual configuration: a real (uniform) branch, not selects per element
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (e.use_limit) { lin[j] = fminf(fmaxf(lin[j], -e.limit), e.limit); glu[j] = fminf(glu[j], e.limit); }
      if (e.use_extra) lin[j] = round_as<TI>(lin[j] + e.extra);
    }
  }
  ACT_DISPATCH(ACTC, e.act, ACT, {
    if constexpr (act_is_sigmoid_gate<ACT>()) {
      f32x2 sg[4];
      sigmoid8<ACT>(glu, sg);
      _Pragma("unroll")
      for (int j = 0; j < 8; j += 2) {
        const f32x2 g = {glu[j], glu[j + 1]}, l = {lin[j], lin[j + 1]};
        const f32x2 r = round_bf16x2(g * sg[j >> 1]) * l;
        o[j] = r.x;
        o[j + 1] = r.y;
      }
    } else {
      _Pragma("unroll")
      for (int j = 0; j < 8; j++) o[j] = round_as<TI>(act_apply_t<ACT>(glu[j])) * lin[j];
    }
  });
  if (e.thr) {
    uint32_t dr[8];
    rng_run<8>(e.seed, (uint64_t)((m + e.m_off) * F + f), dr);   // f and F are multiples of 8
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = dr[j] >= e.thr ? round_as<TI>(o[j]) * e.keep_scale : 0.f;
  }
  store8((TI*)e.out + ro.out, o, 0);
}

// DGLU, 8 columns: v = dh[m, n..n+7]; pa / pb = saved linear / gate pre-activations (see EpiVec, EPI_DGLU)
template <int ACTC = -1>
__device__ __forceinline__ void epi_dglu8(const Epi& e, int64_t m, int64_t n, const RowOff& ro, float (&v)[8], const Raw8& pa,
                                          const Raw8& pb) {
  typedef bf16 TI;
  const int64_t F = e.N;
  float dl[8], dg[8], pl[8], pg[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    pl[j] = pa.get(j);
    pg[j] = pb.get(j);
  }
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2 r = round_bf16x2(f32x2{v[j], v[j + 1]});
    v[j] = r.x;
    v[j + 1] = r.y;
  }
  if (e.thr) {   // one uniform branch for the vector, not one per element
    uint32_t dr[8];
    rng_run<8>(e.seed, (uint64_t)((m + e.m_off) * F + n), dr);
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = dr[j] >= e.thr ? v[j] * e.keep_scale : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 8; j++) {
    dl[j] = v[j];
    dg[j] = v[j];
  }
  if (e.use_limit || e.use_extra) {   // a real (uniform) branch: see epi_glu8
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (e.use_limit) {
        const float l = pl[j], gl = pg[j];
        dl[j] = (l >= -e.limit && l <= e.limit) ? dl[j] : 0.f;
        dg[j] = (gl <= e.limit) ? dg[j] : 0.f;
        pl[j] = fminf(fmaxf(l, -e.limit), e.limit);
        pg[j] = fminf(gl, e.limit);
      }
      if (e.use_extra) pl[j] = round_as<TI>(pl[j] + e.extra);
    }
  }
  ACT_DISPATCH(ACTC, e.act, ACT, {
    if constexpr (act_is_sigmoid_gate<ACT>()) {
      constexpr float a = ACT == VITK_ACT_SILU ? 1.f : 1.702f;
      f32x2 sg4[4];
      sigmoid8<ACT>(pg, sg4);
      _Pragma("unroll")
      for (int j = 0; j < 8; j += 2) {
        const f32x2 g = {pg[j], pg[j + 1]}, l = {pl[j], pl[j + 1]};
        f32x2 vl = {dl[j], dl[j + 1]}, vg = {dg[j], dg[j + 1]};
        const f32x2 sg = sg4[j >> 1];
        vl *= round_bf16x2(g * sg);                              // act_apply_t
        const f32x2 one_minus = sg * -1.f + 1.f;                 // 1 - s as one packed fma (exact: a single rounding)
        vg *= l * (sg * (1.f + (a * g) * one_minus));            // act_grad_t
        dl[j] = vl.x; dl[j + 1] = vl.y;
        dg[j] = vg.x; dg[j + 1] = vg.y;
      }
    } else {
      _Pragma("unroll")
      for (int j = 0; j < 8; j++) {
        dl[j] *= round_as<TI>(act_apply_t<ACT>(pg[j]));
        dg[j] *= pl[j] * act_grad_t<ACT>(pg[j]);
      }
    }
  });
  store8((TI*)e.out + ro.out, dl, e.wt);
  store8((TI*)e.out + ro.out + F, dg, e.wt);
}

// RESIDUAL on a bf16 stream, 8 columns: v = linear output (bias added) of row m; res = the residual row `row`, columns
// n..n+7 as loaded; gam = layer scale (ones without one).  Arithmetic and rounding points of EpiVec's RESIDUAL form.
__device__ __forceinline__ void epi_res8(const Epi& e, int64_t m, int64_t row, int64_t n, float (&v)[8], const float (&gam)[8],
                                         const Raw8& res) {
  typedef bf16 T;
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = round_as<T>(v[j]);
  if (e.thr) {
    uint32_t dr[8];
    rng_run<8>(e.seed, (uint64_t)((m + e.m_off) * e.N + n), dr);
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = dr[j] >= e.thr ? round_as<T>(v[j] * e.keep_scale) : 0.f;
  }
  if (e.aux) store8((T*)e.aux + m * e.N + n, v);
  float r[8];
  if (e.gamma) {
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = res.get(j) + e.alpha * round_as<T>(v[j] * gam[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = res.get(j) + e.alpha * v[j];
  }
  store8((T*)e.out + row * e.ldo + n, r, 0);
}

// ACT, 8 columns: v = fc1 output (bias added) of row m, columns n..n+7 (see EpiVec, EPI_ACT)
template <int ACTC = -1>
__device__ __forceinline__ void epi_act8(const Epi& e, int64_t m, int64_t n, const RowOff& ro, float (&v)[8]) {
  typedef bf16 TI;
  store8((TI*)e.aux + ro.aux, v, 0);
  float o[8];
  ACT_DISPATCH(ACTC, e.act, ACT, {
    _Pragma("unroll")
    for (int j = 0; j < 8; j++) o[j] = act_apply_t<ACT>(round_as<TI>(v[j]));
  });
  if (e.thr) {
    uint32_t dr[8];
    rng_run<8>(e.seed, (uint64_t)((m + e.m_off) * e.N + n), dr);
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = dr[j] >= e.thr ? round_as<TI>(o[j]) * e.keep_scale : 0.f;
  }
  store8((TI*)e.out + ro.out, o, 0);
}

// DACT, 8 columns: v = dh[m, n..n+7]; pa = saved pre-activations (see EpiVec, EPI_DACT)
template <int ACTC = -1>
__device__ __forceinline__ void epi_dact8(const Epi& e, int64_t m, int64_t n, const RowOff& ro, float (&v)[8], const Raw8& pa) {
  typedef bf16 TI;
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = round_as<TI>(v[j]);
  if (e.thr) {
    uint32_t dr[8];
    rng_run<8>(e.seed, (uint64_t)((m + e.m_off) * e.N + n), dr);
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = dr[j] >= e.thr ? v[j] * e.keep_scale : 0.f;
  }
  float o[8];
  ACT_DISPATCH(ACTC, e.act, ACT, {
    _Pragma("unroll")
    for (int j = 0; j < 8; j++) o[j] = v[j] * act_grad_t<ACT>(pa.get(j));
  });
  store8((TI*)e.out + ro.out, o, e.wt);
}

template <int MODE, class TO>
struct EpiCols {  // per-lane column constants, loaded once per tile
  float bias0[4], bias1[4], gam[4];
  float wb0[8], wb1[8];   // 8-column forms
  int64_t ncol;   // first of this lane's 4 (or 8) output columns (GLU: the linear column f)
  __device__ __forceinline__ void init(const Epi& e, int64_t nwave, int64_t N, int lane) {
#pragma unroll
    for (int j = 0; j < 4; j++) { bias0[j] = 0.f; bias1[j] = 0.f; gam[j] = 1.f; }
    if (epi_wide<MODE, TO>()) {
#pragma unroll
      for (int j = 0; j < 8; j++) { wb0[j] = 0.f; wb1[j] = 0.f; }
      if (MODE == EPI_GLU) {
        const int cv = lane & 3;
        ncol = (nwave >> 6) * 32 + 8 * cv;
        if (e.bias && nwave + 8 * cv < N) {
          Raw8 bl, bg;
          bl.load((const bf16*)e.bias + ncol);
          bg.load((const bf16*)e.bias + (N >> 1) + ncol);
#pragma unroll
          for (int j = 0; j < 8; j++) { wb0[j] = bl.get(j); wb1[j] = bg.get(j); }
        }
      } else {
        ncol = nwave + 8 * (lane & 7);
        if (e.bias && ncol < N) {
          Raw8 b;
          b.load((const bf16*)e.bias + ncol);
#pragma unroll
          for (int j = 0; j < 8; j++) wb0[j] = b.get(j);
          if (MODE == EPI_ROPE) {   // the rotation partner's columns: the other half of the same 64-column head
            b.load((const bf16*)e.bias + (ncol ^ 32));
#pragma unroll
            for (int j = 0; j < 8; j++) wb1[j] = b.get(j);
          }
        }
        if (MODE == VITK_EPI_RESIDUAL) {   // layer scale of this lane's 8 columns (bf16 stream: gamma is bf16 too)
#pragma unroll
          for (int j = 0; j < 8; j++) wb1[j] = 1.f;
          if (e.gamma && ncol < N) {
            Raw8 g;
            g.load((const bf16*)e.gamma + ncol);
#pragma unroll
            for (int j = 0; j < 8; j++) wb1[j] = g.get(j);
          }
        }
      }
    } else if (MODE == EPI_GLU) {
      const int cv = lane & 7;
      ncol = (nwave >> 6) * 32 + 4 * cv;
      if (e.bias && nwave + 4 * cv < N) {
        Vec4<bf16> bl, bg;
        bl.load((const bf16*)e.bias + ncol);
        bg.load((const bf16*)e.bias + (N >> 1) + ncol);
#pragma unroll
        for (int j = 0; j < 4; j++) { bias0[j] = bl.v[j]; bias1[j] = bg.v[j]; }
      }
    } else {
      ncol = nwave + 4 * (lane & 15);
      if (MODE != MODE_PARTIAL && ncol < N) {
        if (e.bias) {
          Vec4<bf16> b; b.load((const bf16*)e.bias + ncol);
#pragma unroll
          for (int j = 0; j < 4; j++) bias0[j] = b.v[j];
        }
        if (MODE == VITK_EPI_RESIDUAL && e.gamma) {
          Vec4<TO> g; g.load((const TO*)e.gamma + ncol);
#pragma unroll
          for (int j = 0; j < 4; j++) gam[j] = g.v[j];
        }
      }
    }
  }
};

// One 32-row x 64-column slab of a wave's tile, in three steps so that a kernel can issue the memory reads of a slab
// (residual / position / saved pre-activation rows) well before it needs them:
//   slab_prefetch  loads of the slab's 8 rows per lane;  slab_stage  accumulators -> the wave's own LDS slab;
//   slab_finish    read back 4 consecutive columns per lane, apply the epilogue, store.
// The staging slab belongs to one wave and LDS operations of a wave complete in order, so no barrier is needed
// between the steps (the caller has one block barrier between the last main-loop LDS read and the first stage).
template <int MODE, class TO>
struct SlabPre {
  EpiPre<(MODE < 0 ? 0 : MODE), TO> r[8];
  Raw8 wa[2], wb[2];   // 8-column DGLU: linear and gate pre-activations of the row being finished and of the next one
  Raw8 wr[4];          // 8-column RESIDUAL (bf16 stream): the slab's four residual rows of this lane
  int64_t wrow[4];
};

template <int MODE>
constexpr bool epi_reads_memory() { return MODE == VITK_EPI_RESIDUAL || MODE == VITK_EPI_PATCH || MODE == EPI_DACT || MODE == EPI_DGLU; }

template <int MODE, class TO>
__device__ __forceinline__ void slab_prefetch(const Epi& e, const EpiCols<MODE, TO>& cols, SlabPre<MODE, TO>& pre,
                                              int64_t mbase, int64_t M, int64_t N, int lane) {
  if (!epi_reads_memory<MODE>()) return;
  constexpr int EMODE = MODE < 0 ? 0 : MODE;
  if (epi_wide<MODE, TO>()) {   // DGLU / DACT: the slab's first row per lane; slab_finish requests the others one ahead of their use
    if (MODE == VITK_EPI_RESIDUAL) {
#pragma unroll
      for (int it = 0; it < 4; it++) {
        const int64_t mr = mbase + it * 8 + (lane >> 3);
        if (mr < M && cols.ncol < N) {
          pre.wrow[it] = e.row_idx ? e.row_idx[mr] : mr;
          pre.wr[it].load((const bf16*)e.res + pre.wrow[it] * e.ldo + cols.ncol, e.ntl);
        }
      }
      return;
    }
    const int64_t m = mbase + (lane >> 3);
    if (m < M && cols.ncol < N) {
      if (MODE == EPI_DACT) {
        pre.wa[0].load((const bf16*)e.aux + m * e.N + cols.ncol, e.ntl);
      } else {
        pre.wa[0].load((const bf16*)e.aux + m * 2 * e.N + cols.ncol, e.ntl);
        pre.wb[0].load((const bf16*)e.aux + m * 2 * e.N + e.N + cols.ncol, e.ntl);
      }
    }
    return;
  }
  const int rr = lane >> 4;
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int64_t m = mbase + it * 4 + rr;
    if (m < M && cols.ncol < N) EpiVec<EMODE, TO>::prefetch(e, m, cols.ncol, pre.r[it]);
  }
}

__device__ __forceinline__ void slab_stage(float* stg, const f32x16& a0, const f32x16& a1, int lane) {
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    stg[row * STG_LD + (lane & 31)] = a0[r];
    stg[row * STG_LD + 32 + (lane & 31)] = a1[r];
  }
}

template <int MODE, class TO, int ACTC = -1>
__device__ __forceinline__ void slab_finish(const Epi& e, const EpiCols<MODE, TO>& cols, SlabPre<MODE, TO>& pre,
                                            const float* stg, int64_t mbase, int64_t nwave, int64_t M, int64_t N,
                                            int64_t split, float* __restrict__ partial, int lane) {
  constexpr int EMODE = MODE < 0 ? 0 : MODE;
  const int64_t ncol = cols.ncol;
  if (epi_wide<MODE, TO>() && MODE == EPI_GLU) {
    const int cv = lane & 3, rr = lane >> 2;
    const int64_t out_first = (mbase + rr) * e.ldo + ncol, aux_first = (mbase + rr) * e.N + ncol;
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int row = it * 16 + rr;
      const int64_t m = mbase + row;
      const RowOff ro = {out_first + (int64_t)(it * 16) * e.ldo, aux_first + (int64_t)(it * 16) * e.N};
      float lin[8], glu[8];
      *(float4*)lin = *(const float4*)(stg + row * STG_LD + 8 * cv);
      *(float4*)(lin + 4) = *(const float4*)(stg + row * STG_LD + 8 * cv + 4);
      *(float4*)glu = *(const float4*)(stg + row * STG_LD + 32 + 8 * cv);
      *(float4*)(glu + 4) = *(const float4*)(stg + row * STG_LD + 32 + 8 * cv + 4);
      if (m < M && nwave + 8 * cv < N) epi_glu8<ACTC>(e, m, ncol, ro, lin, glu, cols.wb0, cols.wb1);
    }
  } else if (epi_wide<MODE, TO>()) {
    const int cv = lane & 7, rr = lane >> 3;
    const int rope_t0 = MODE == EPI_ROPE ? (int)((mbase + e.m_off) % e.seq) : 0;
    const int64_t aux_ld = MODE == EPI_DGLU ? 2 * e.N : e.N;   // row length of the saved pre-activations
    const int64_t out_first = (mbase + rr) * e.ldo + ncol, aux_first = (mbase + rr) * aux_ld + ncol;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int row = it * 8 + rr;
      const int64_t m = mbase + row;
      const RowOff ro = {out_first + (int64_t)(it * 8) * e.ldo, aux_first + (int64_t)(it * 8) * aux_ld};
      float v[8];
      *(float4*)v = *(const float4*)(stg + row * STG_LD + 8 * cv);
      *(float4*)(v + 4) = *(const float4*)(stg + row * STG_LD + 8 * cv + 4);
      if (MODE == EPI_DGLU && it < 3 && m + 8 < M && ncol < N) {   // the next row's pre-activations, one step ahead
        pre.wa[(it + 1) & 1].load((const bf16*)e.aux + ro.aux + 8 * aux_ld, e.ntl);
        pre.wb[(it + 1) & 1].load((const bf16*)e.aux + ro.aux + 8 * aux_ld + e.N, e.ntl);
      }
      if (MODE == EPI_DACT && it < 3 && m + 8 < M && ncol < N) pre.wa[(it + 1) & 1].load((const bf16*)e.aux + ro.aux + 8 * aux_ld, e.ntl);
      if (m < M && ncol < N) {
        if (MODE == EPI_DGLU) epi_dglu8<ACTC>(e, m, ncol, ro, v, pre.wa[it & 1], pre.wb[it & 1]);
        else if (MODE == EPI_DACT) epi_dact8<ACTC>(e, m, ncol, ro, v, pre.wa[it & 1]);
        else if (MODE == EPI_ACT) {
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] += cols.wb0[j];
          epi_act8<ACTC>(e, m, ncol, ro, v);
        } else if (MODE == VITK_EPI_RESIDUAL) {
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] += cols.wb0[j];
          epi_res8(e, m, pre.wrow[it], ncol, v, cols.wb1, pre.wr[it]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] += cols.wb0[j];
          if (MODE == EPI_ROPE && ncol < 2 * e.rope_width) {
            // q and k sections of the fused QKV projection: x = the linear's output as bf16 holds it; then the per-head
            // normalisation (optional), then rotate_half on the rows behind the prefix (optional).  The 8 lanes that share
            // a row hold one head; a lane also reads its rotation partner's 8 columns (the other half of the head).
            float u[8];
            *(float4*)u = *(const float4*)(stg + row * STG_LD + 8 * (cv ^ 4));
            *(float4*)(u + 4) = *(const float4*)(stg + row * STG_LD + 8 * (cv ^ 4) + 4);
#pragma unroll
            for (int j = 0; j < 8; j++) {
              v[j] = round_as<bf16>(v[j]);
              u[j] = round_as<bf16>(u[j] + cols.wb1[j]);
            }
            if (e.qk_w) {
              const int sec = ncol >= e.rope_width;                       // 0: q, 1: k
              const int64_t mg = m + e.m_off;
              store8((bf16*)e.aux + ((int64_t)sec * e.rows_total + mg) * e.rope_width + (ncol - sec * e.rope_width), v);
              // row statistics in the order of the row kernel (norm_fwd_k: 4 columns per lane, butterfly 8, 4, 2, 1 over
              // its 16 lanes): this lane carries that kernel's lanes 2 cv and 2 cv + 1 as two partial sums
              float mean = 0.f;
              if (e.qk_ln) {
                float p0 = (v[0] + v[1]) + (v[2] + v[3]), p1 = (v[4] + v[5]) + (v[6] + v[7]);
#pragma unroll
                for (int o = 4; o > 0; o >>= 1) { p0 += __shfl_xor(p0, o, 64); p1 += __shfl_xor(p1, o, 64); }
                mean = (p0 + p1) * (1.f / 64.f);
              }
              float q0 = sumsq4(0.f, v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean);
              float q1 = sumsq4(0.f, v[4] - mean, v[5] - mean, v[6] - mean, v[7] - mean);
#pragma unroll
              for (int o = 4; o > 0; o >>= 1) { q0 += __shfl_xor(q0, o, 64); q1 += __shfl_xor(q1, o, 64); }
              const float rstd = rsqrtf((q0 + q1) * (1.f / 64.f) + e.qk_eps);
              if (cv == 0) {
                const int64_t hrow = mg * (e.rope_width >> 6) + ((ncol - sec * e.rope_width) >> 6);
                const int64_t per = e.rows_total * (e.rope_width >> 6);
                e.qk_rstd[sec * per + hrow] = rstd;
                if (e.qk_ln) e.qk_mean[sec * per + hrow] = mean;
              }
              const float* wq = e.qk_w + sec * 64;
              const float* bq = e.qk_b ? e.qk_b + sec * 64 : nullptr;
#pragma unroll
              for (int j = 0; j < 8; j++) {
                const int d = 8 * cv + j, dp = d ^ 32;
                v[j] = round_as<bf16>(norm_apply(v[j] - mean, rstd, wq[d], bq ? bq[d] : 0.f, bq != nullptr));
                u[j] = round_as<bf16>(norm_apply(u[j] - mean, rstd, wq[dp], bq ? bq[dp] : 0.f, bq != nullptr));
              }
            }
            // rotate_half (vit/rope.py:148-163):  o[d] = x[d] cos[d] - x[d + 32] sin[d]  (d < 32),  x[d] cos[d] + x[d - 32] sin[d]
            int tok = rope_t0 + row;              // index of the row inside its sample: one division per slab
            while (tok >= (int)e.seq) tok -= (int)e.seq;
            tok -= (int)e.prefix;
            if (e.rope && tok >= 0) {
              const float* sn = e.rope + tok * 64 + 8 * cv;
              const float* cs = sn + (e.seq - e.prefix) * 64;
              float sv[8], cw[8];
              *(float4*)sv = *(const float4*)sn;
              *(float4*)(sv + 4) = *(const float4*)(sn + 4);
              *(float4*)cw = *(const float4*)cs;
              *(float4*)(cw + 4) = *(const float4*)(cs + 4);
              const float sign = (cv & 4) ? 1.f : -1.f;
#pragma unroll
              for (int j = 0; j < 8; j++) v[j] = rope_mix(v[j], cw[j], sign * u[j], sv[j], e.rope_emu);
            }
          }
          store8((bf16*)e.out + ro.out, v, 0);
        }
      }
    }
  } else if (MODE == EPI_GLU) {
    const int cv = lane & 7, rr = lane >> 3;
#pragma unroll
    for (int it = 0; it < 4; it++) {
      const int row = it * 8 + rr;
      const int64_t m = mbase + row;
      float lin[4], glu[4];
      *(float4*)lin = *(const float4*)(stg + row * STG_LD + 4 * cv);
      *(float4*)glu = *(const float4*)(stg + row * STG_LD + 32 + 4 * cv);
      if (m < M && nwave + 4 * cv < N) epi_glu<TO>(e, m, ncol, lin, glu, cols.bias0, cols.bias1);
    }
  } else {
    const int cv = lane & 15, rr = lane >> 4;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int row = it * 4 + rr;
      const int64_t m = mbase + row;
      float v[4];
      *(float4*)v = *(const float4*)(stg + row * STG_LD + 4 * cv);
      if (m < M && ncol < N) {
        if (MODE == MODE_PARTIAL) st16(partial + (split * M + m) * N + ncol, __builtin_bit_cast(u32x4_t, *(const f32x4*)v), 0);
        else EpiVec<EMODE, TO>::apply(e, m, ncol, v, cols.bias0, cols.gam, pre.r[it]);
      }
    }
  }
}

// Tile order inside one split: panels of PANEL_M row tiles, walked column tile by column tile.  The blocks of an XCD
// run consecutive tile ids, so at any time they share a few B column tiles and one panel of A rows, and each B tile is
// fetched into that XCD's L2 once per panel instead of once per row tile.
#ifndef PANEL_M_VALUE
#define PANEL_M_VALUE 8
#endif
constexpr int PANEL_M = PANEL_M_VALUE;
// Tile indices are 32-bit: a 64-bit division is ~100 scalar instructions, and the handful of them that places a tile cost
// a tenth of a short-K tile's time when they were 64-bit.  Blocks that share an XCD (bid % 8) get consecutive tile ids, so
// one L2 sees neighbouring tiles.
__device__ __forceinline__ uint32_t xcd_remap32(uint32_t bid, uint32_t nwg) {
  const uint32_t xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__device__ __forceinline__ void tile_coords32(uint32_t wg, uint32_t tiles_m, uint32_t tiles_n, uint32_t panel_m, uint32_t& tm,
                                              uint32_t& tn, uint32_t& split) {
  const uint32_t per_split = tiles_m * tiles_n;
  split = wg / per_split;
  const uint32_t r = wg - split * per_split;
  const uint32_t per_panel = panel_m * tiles_n;
  const uint32_t panel = r / per_panel, in_panel = r - panel * per_panel;
  const uint32_t left = tiles_m - panel * panel_m, rows_here = left < panel_m ? left : panel_m;
  tn = in_panel / rows_here;
  tm = panel * panel_m + (in_panel - tn * rows_here);
}

constexpr int HS_B = 16384;        // one operand's half-stage
constexpr int SLOT_B = 2 * HS_B;   // A then B
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ int swz_rows(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }            // G[(r>>2)&3]
__device__ __forceinline__ int swz_krow(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

struct DmaSrc {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff[2];   // this lane's two 16-byte units at half-stage 0 (byte offsets), or OOB
  int32_t klim[2];    // K-contiguous operand: unit valid while kk < klim
  uint32_t hstride;   // bytes per half-stage
};

template <bool TR, bool GLUMAP>
__device__ __forceinline__ DmaSrc make_dma(const bf16* __restrict__ g, int64_t ld, int64_t row0, int64_t nrows, int64_t kbeg,
                                           int64_t kend, int64_t Ktot, int wid, int lane) {
  DmaSrc t;
  if (!TR) {
    const int64_t base_row = GLUMAP ? 0 : row0;
    const int64_t span_rows = GLUMAP ? nrows : (nrows - row0 < 256 ? nrows - row0 : 256);
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g + base_row * ld), 0, clip32(((span_rows - 1) * ld + Ktot) * 2), 0x00020000);
    t.hstride = 64;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int p = (wid * 2 + j) * 64 + lane;
      const int row = p >> 2, c = (p & 3) ^ swz_rows(row);
      const int64_t r = row0 + row;
      const int64_t rr = GLUMAP ? glu_row(r, nrows >> 1) : (int64_t)row;
      t.voff[j] = r < nrows ? (uint32_t)((rr * ld + kbeg + c * 8) * 2) : OOB;
      t.klim[j] = (int32_t)(kend - kbeg) - c * 8;
    }
  } else {
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g + kbeg * ld + row0), 0,
                                               clip32(((kend - kbeg - 1) * ld + (nrows - row0)) * 2), 0x00020000);
    t.hstride = (uint32_t)(32 * ld * 2);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int p = (wid * 2 + j) * 64 + lane;
      const int krow = p >> 5, col = ((p & 31) ^ (swz_krow(krow) << 1)) * 8;
      t.voff[j] = row0 + col < nrows ? (uint32_t)(((int64_t)krow * ld + col) * 2) : OOB;
      t.klim[j] = 0x7fffffff;
    }
  }
  return t;
}

// request half-stage h (k offset 32*h from the block's first k) of one operand into `dst`
// The compiler knows these requests write LDS and drains them (s_waitcnt vmcnt(0)) in front of the next LDS read it
// issues, whichever slot that read addresses.  Issuing them from inline assembly instead keeps three half-stages in flight
// across the fragment reads; measured on one device against this form, that was no faster on any layout and 6 % slower
// on weight gradients (both operands stream from HBM there, and workgroups that run closer together share more lines in
// L2), so the plain form stays.
template <bool TR>
__device__ __forceinline__ void dma_half(const DmaSrc& t, char* dst, int wid, int h) {
#pragma unroll
  for (int j = 0; j < 2; j++) {
    uint32_t off = t.voff[j] + (uint32_t)h * t.hstride;
    if (!TR) off = (h * 32 < t.klim[j] && t.voff[j] != OOB) ? off : OOB;
    else off = t.voff[j] != OOB ? off : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(t.rsrc, (lds_void_t*)(dst + (wid * 2 + j) * 1024), 16, off, 0, 0, 0);
  }
}

// 16x16x32 operand fragment from a DMA-staged half-stage: index r0 + (lane & 15) (r0 a multiple of 16),
// k = 8*(lane>>4) + j
template <bool TR, int PITCH = 512>   // PITCH = bytes per k-row of a contraction-major image
__device__ __forceinline__ bf16x8 read_frag16(const char* hs, int r0, int lane) {
  if (!TR) {
    const int row = r0 + (lane & 15);
    return *(const bf16x8*)(hs + row * 64 + (((lane >> 4) ^ swz_rows(row)) << 4));
  } else {
    const int i = lane & 15;
    const int krow = 8 * (lane >> 4) + (i >> 2);
    const char* p = hs + krow * PITCH + ((((r0 >> 4) ^ swz_krow(krow))) << 5) + 8 * (i & 3);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * PITCH));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
}

// barrier that leaves DMA requests in flight: only the half-stage read next has to be complete
__device__ __forceinline__ void dma_barrier(bool drain_reads) {
  __builtin_amdgcn_sched_barrier(0);
  if (drain_reads) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// ------------------------------------------------------------------------------------------------
// Epilogue operand stream (256x256 kernel, whole tiles).  The epilogues that read a memory operand per output element
// (RESIDUAL: the residual row; DACT / DGLU: the saved pre-activations) are bound by how many bytes a CU keeps in flight:
// with the rows requested into registers, one or a few steps ahead, a step waits out most of a memory latency, and
// more steps ahead do not fit beside 128 accumulator registers.  Here a wave requests its rows by LDS-DMA into a ring
// of its own in the (by then idle) operand area of LDS, STREAM_B bytes = 4 to 8 steps ahead, at no register cost.  Each
// request is one wave instruction of 64 x 16 bytes whose lane order is the order the lanes read it back in, so a lane
// only ever reads what it asked for itself and no barrier is involved.
// The requests are issued from inline assembly and waited for with counted s_waitcnt: vmcnt retires in issue order and
// counts loads, stores and LDS-DMA alike, so "all but the N youngest" with N = the operations issued after the wanted
// request (later requests and the stores of the steps in between) waits for that request alone.  That count has to be a
// compile-time constant, which is why this path runs on whole tiles only (no predicated-off store ever drops out of the
// count; optional extra stores only make the real count larger than N, which waits longer, never shorter).
// ------------------------------------------------------------------------------------------------
constexpr int STREAM_B = 8192;   // bytes of requests in flight per wave

// one request: 64 lanes x 16 bytes from rsrc[voff + soff] to LDS bytes [lds_addr, lds_addr + 1024) in lane order.  The
// statement first waits for the wave's LDS reads (the slot it refills has just been read); M0 is saved around it.
__device__ __forceinline__ void dma16_asm(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
      : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, class TO>
constexpr bool epi_streams() { return MODE == EPI_DGLU || MODE == EPI_DACT || MODE == VITK_EPI_RESIDUAL; }

template <int MODE, class TO>
struct StreamCfg {
  static constexpr bool WIDE = epi_wide<MODE, TO>();
  static constexpr int NI = MODE == EPI_DGLU ? 2 : 1;   // requests per step (DGLU: linear and gate pre-activations)
  static constexpr int NS = MODE == EPI_DGLU ? 2 : 1;   // stores per step that are always issued
  static constexpr int ROWS = WIDE ? 8 : 4;             // rows of the wave's 64-column strip per step
  static constexpr int STEPS = 128 / ROWS;              // steps per wave tile (128 rows)
  static constexpr int PER_SLAB = 32 / ROWS;
  static constexpr int SLOT = NI * 1024;
  static constexpr int D = STREAM_B / SLOT;             // requests ahead
};
// vector-memory operations a wave has issued after the request of step j at the moment it starts step j.  Order of
// issue: requests 0 .. D-1, then per step i: [request i + D, if there is one], [the step's stores].
template <class C>
constexpr int stream_younger(int j) {
  int n = 0;
  if (j < C::D) {
    n += (C::D - 1 - j) * C::NI;
    for (int i = 0; i < j; i++) n += (i + C::D < C::STEPS ? C::NI : 0) + C::NS;
  } else {
    n += C::NS;   // the stores of step j - D follow its request for step j
    for (int i = j - C::D + 1; i < j; i++) n += (i + C::D < C::STEPS ? C::NI : 0) + C::NS;
  }
  return n;
}

template <int MODE, class TO, int J, int ACTC>
__device__ __forceinline__ void stream_step(const Epi& e, const EpiCols<MODE, TO>& cols, const float* stg, const char* slots,
                                            __amdgpu_buffer_rsrc_t rsrc, uint32_t lane_off, uint32_t step_b, uint32_t lds0,
                                            int64_t mrow0, int lane) {
  typedef StreamCfg<MODE, TO> C;
  constexpr int it = J % C::PER_SLAB;
  const int cv = C::WIDE ? lane & 7 : lane & 15, rr = C::WIDE ? lane >> 3 : lane >> 4;
  const int row = it * C::ROWS + rr;               // row inside the staged slab
  const int64_t m = mrow0 + J * C::ROWS + rr;      // row of the problem
  const int64_t ncol = cols.ncol;
  wait_vmcnt<stream_younger<C>(J)>();
  const char* slot = slots + (J % C::D) * C::SLOT + lane * 16;
  const uint4 q0 = *(const uint4*)slot;
  uint4 q1 = q0;
  if (MODE == EPI_DGLU) q1 = *(const uint4*)(slot + 1024);
  if (J + C::D < C::STEPS) {   // refill the slot just read (the statement waits for the reads above first)
    const uint32_t soff = (uint32_t)(J + C::D) * step_b;
    dma16_asm(rsrc, lane_off, soff, lds0 + (J % C::D) * C::SLOT);
    if (MODE == EPI_DGLU) dma16_asm(rsrc, lane_off, soff + (uint32_t)e.N * 2u, lds0 + (J % C::D) * C::SLOT + 1024);
  }
  if constexpr (C::WIDE) {
    float v[8];
    *(float4*)v = *(const float4*)(stg + row * STG_LD + 8 * cv);
    *(float4*)(v + 4) = *(const float4*)(stg + row * STG_LD + 8 * cv + 4);
    const int64_t aux_ld = MODE == EPI_DGLU ? 2 * e.N : e.N;
    const RowOff ro = {m * e.ldo + ncol, m * aux_ld + ncol};
    Raw8 pa, pb;
    pa.q = q0;
    pb.q = q1;
    if (MODE == EPI_DGLU) epi_dglu8<ACTC>(e, m, ncol, ro, v, pa, pb);
    else if (MODE == EPI_DACT) epi_dact8<ACTC>(e, m, ncol, ro, v, pa);
    else {
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] += cols.wb0[j];
      epi_res8(e, m, m, ncol, v, cols.wb1, pa);
    }
  } else {   // RESIDUAL on an fp32 stream: 4 columns per lane
    float v[4];
    *(float4*)v = *(const float4*)(stg + row * STG_LD + 4 * cv);
    EpiPre<VITK_EPI_RESIDUAL, TO> pre;
    pre.row = m;
    if constexpr (sizeof(TO) == 4) pre.r.q = __builtin_bit_cast(float4, q0);
    EpiVec<VITK_EPI_RESIDUAL, TO>::apply(e, m, ncol, v, cols.bias0, cols.gam, pre);
  }
}

template <int MODE, class TO, int J0, int J1>
struct StreamSteps {
  template <class F>
  static __device__ __forceinline__ void run(F&& f) {
    if constexpr (J0 < J1) {
      f(std::integral_constant<int, J0>{});
      StreamSteps<MODE, TO, J0 + 1, J1>::run(f);
    }
  }
};

// 32-row x 64-column slab (row tiles 2s, 2s+1 of a wave's 8 x 4 grid of 16x16 accumulators) -> the wave's LDS slab.
// Accumulator register r of lane l is row 4*(l>>4) + r, column l&15 of its tile.
__device__ __forceinline__ void slab_stage16(float* stg, const f32x4_t (&acc)[8][4], int s, int lane) {
#pragma unroll
  for (int h = 0; h < 2; h++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        stg[(16 * h + 4 * (lane >> 4) + r) * STG_LD + 16 * j + (lane & 15)] = acc[2 * s + h][j][r];
}

// the whole epilogue of one wave tile (128 rows x 64 columns, 16x16 accumulators) on the operand stream
template <int MODE, class TO, int ACTC>
__device__ __forceinline__ void epilogue_stream16(const Epi& e, const EpiCols<MODE, TO>& cols, float* stg, char* slots,
                                                  const f32x4_t (&acc)[8][4], int64_t mrow0, int lane) {
  typedef StreamCfg<MODE, TO> C;
  const int rr = C::WIDE ? lane >> 3 : lane >> 4;
  const char* src = MODE == VITK_EPI_RESIDUAL ? (const char*)e.res : (const char*)e.aux;
  const int64_t ld = MODE == VITK_EPI_RESIDUAL ? e.ldo : (MODE == EPI_DGLU ? 2 * e.N : e.N);
  constexpr int esz = MODE == VITK_EPI_RESIDUAL ? (int)sizeof(TO) : 2;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(src + mrow0 * ld * esz), 0, clip32(128 * ld * esz), 0x00020000);
  const uint32_t lane_off = (uint32_t)((rr * ld + cols.ncol) * esz);
  const uint32_t step_b = (uint32_t)(C::ROWS * ld * esz);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_char_t*)slots;
#pragma unroll
  for (int j = 0; j < C::D; j++) {
    dma16_asm(rsrc, lane_off, (uint32_t)j * step_b, lds0 + j * C::SLOT);
    if (MODE == EPI_DGLU) dma16_asm(rsrc, lane_off, (uint32_t)j * step_b + (uint32_t)e.N * 2u, lds0 + j * C::SLOT + 1024);
  }
  StreamSteps<MODE, TO, 0, C::STEPS>::run([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    if constexpr (J % C::PER_SLAB == 0) slab_stage16(stg, acc, J / C::PER_SLAB, lane);
    stream_step<MODE, TO, J, ACTC>(e, cols, stg, slots, rsrc, lane_off, step_b, lds0, mrow0, lane);
  });
}
