// GEMM for the ViT encoder path: C[M,N] = op(A) x op(B) with fused epilogues.
//   * gemm_mfma_k   : bf16 operands, v_mfma_f32_32x32x16_bf16, 128x128x64 block tile, 4 waves (2x2),
//                     each wave a 64x64 tile = 2x2 MFMA tiles.  Operands are staged global -> registers ->
//                     LDS (the next tile's loads are issued before the current tile's MFMAs and written
//                     after the barrier).  K-contiguous operands are read back with ds_read_b128 from rows
//                     padded by 16 B; operands whose contraction index is the slow one (dgrad's weight,
//                     both wgrad operands) stay in their memory layout and are read with
//                     ds_read_b64_tr_b16 from rows padded by 64 B, so no transposed copy is ever made.
//   * gemm_simple_k : portable scalar kernel for fp32 and for shapes the MFMA kernel rejects.
// Both share the epilogue functors (bias / residual + layer-scale + dropout + drop-path scatter /
// patch-embedding placement + position add).
#include "common.h"
#include <string.h>

#include <type_traits>
#include "gemm_epi.h"

// ------------------------------------------------------------------------------------------------
// portable kernel: element (m,k) of A at A[m*sam + k*sak], element (k,n) of B at B[k*sbk + n*sbn]
// ------------------------------------------------------------------------------------------------
template <class TI, int MODE, class TO>
__global__ __launch_bounds__(256) void gemm_simple_k(const TI* __restrict__ A, const TI* __restrict__ B, int64_t M,
                                                     int64_t N, int64_t K, int64_t sam, int64_t sak, int64_t sbk,
                                                     int64_t sbn, Epi e) {
  __shared__ float As[16][65];
  __shared__ float Bs[16][65];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
  for (int64_t k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int el = tid + i * 256;
      int kk, mm;
      if (sak == 1) { kk = el & 15; mm = el >> 4; } else { mm = el & 63; kk = el >> 6; }
      As[kk][mm] = (m0 + mm < M && k0 + kk < K) ? to_f<TI>(A[(m0 + mm) * sam + (k0 + kk) * sak]) : 0.f;
      int kb, nn;
      if (sbk == 1) { kb = el & 15; nn = el >> 4; } else { nn = el & 63; kb = el >> 6; }
      Bs[kb][nn] = (n0 + nn < N && k0 + kb < K) ? to_f<TI>(B[(k0 + kb) * sbk + (n0 + nn) * sbn]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; kk++) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int64_t m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < M && n < N) EpiOp<MODE, TI, TO>::apply(e, m, n, acc[i][j]);
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA kernel
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128;
constexpr int ROW_B = BK * 2 + 16;     // 144 B: K-contiguous tile [128][64] bf16, rows padded by 16 B
constexpr int TROW_B = BM * 2 + 64;    // 320 B: contraction-major tile [64][128] bf16, rows padded by 64 B
constexpr int TILE_B = 20480;          // max(128*144, 64*320)

// fragment of the MFMA A operand (rows = M index) or B operand (cols = N index): 8 contraction values
// k = ks*16 + 8*(lane>>5) + j for index r0 + (lane & 31).
template <bool TR, int TROW = TROW_B>
__device__ __forceinline__ bf16x8 read_frag(const char* s, int r0, int ks, int lane) {
  if (!TR) {
    return *(const bf16x8*)(s + (r0 + (lane & 31)) * ROW_B + (ks * 16 + 8 * (lane >> 5)) * 2);
  } else {
    const int kbase = ks * 16 + 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int col = r0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const char* p = s + kbase * TROW + col * 2;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * TROW));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
}


struct TileSrc {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff[4];
  int32_t klim[4];   // K-contiguous operand: chunk valid while (k0 - kbeg) < klim
  uint32_t kstride;  // bytes per unit of k
};


template <bool TR, bool GLUMAP, int ROWS = 128, int NT = 256>
__device__ __forceinline__ TileSrc make_src(const bf16* __restrict__ g, int64_t ld, int64_t row0, int64_t nrows,
                                            int64_t kbeg, int64_t kend, int64_t Ktot, int tid) {
  TileSrc t;
  if (!TR) {
    const int64_t base_row = GLUMAP ? 0 : row0;
    const int64_t span_rows = GLUMAP ? nrows : (nrows - row0 < ROWS ? nrows - row0 : ROWS);
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g + base_row * ld), 0, clip32(((span_rows - 1) * ld + Ktot) * 2), 0x00020000);
    t.kstride = 2;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + i * NT;
      const int64_t r = row0 + (c >> 3);
      const int64_t rr = GLUMAP ? glu_row(r, nrows >> 1) : (int64_t)(c >> 3);
      t.voff[i] = r < nrows ? (uint32_t)((rr * ld + kbeg + (c & 7) * 8) * 2) : OOB;
      t.klim[i] = (int32_t)(kend - kbeg) - (c & 7) * 8;
    }
  } else {
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g + kbeg * ld + row0), 0,
                                               clip32(((kend - kbeg - 1) * ld + (nrows - row0)) * 2), 0x00020000);
    t.kstride = (uint32_t)(ld * 2);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + i * NT;
      constexpr int CPR = ROWS / 8;  // 16-byte chunks per k-row
      const int64_t col = row0 + (c % CPR) * 8;
      t.voff[i] = col < nrows ? (uint32_t)(((int64_t)(c / CPR) * ld + (c % CPR) * 8) * 2) : OOB;
      t.klim[i] = 0x7fffffff;
    }
  }
  return t;
}

// kk = k0 - kbeg (multiple of BK)
template <bool TR>
__device__ __forceinline__ void load_tile(uint4 (&r)[4], const TileSrc& t, int32_t kk) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t off = t.voff[i] + (uint32_t)kk * t.kstride;
    if (!TR) off = (kk < t.klim[i] && t.voff[i] != OOB) ? off : OOB;
    else off = t.voff[i] != OOB ? off : OOB;
    r[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(t.rsrc, off, 0, 0));
  }
}

template <bool TR, int ROWS = 128, int NT = 256>
__device__ __forceinline__ void store_tile(const uint4 (&r)[4], char* s, int tid) {
  constexpr int CPR = ROWS / 8, TROW = ROWS * 2 + 64;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = tid + i * NT;
    if (!TR) *(uint4*)(s + (c >> 3) * ROW_B + (c & 7) * 16) = r[i];
    else *(uint4*)(s + (c / CPR) * TROW + (c % CPR) * 16) = r[i];
  }
}


template <bool TA, bool TB, int MODE, class TO>
__global__ __launch_bounds__(256, 3) void gemm_mfma_k(const bf16* __restrict__ A, const bf16* __restrict__ B, int64_t M,
                                                   int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t tiles_m,
                                                   int64_t tiles_n, int64_t klen, Epi e, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE_B];
  char* sA = smem;
  char* sB = smem + TILE_B;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  uint32_t tm, tn, split;   // 32-bit tile arithmetic: see gemm_mfma256d_k
  tile_coords32(xcd_remap32(blockIdx.x, gridDim.x), (uint32_t)tiles_m, (uint32_t)tiles_n, (uint32_t)e.panel_m, tm, tn, split);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const bool asum_on = TA && e.asum && tn == 0 && wn == 0;   // wave-uniform
  float rs[2] = {0.f, 0.f};
  const int64_t kbeg = (int64_t)split * klen;
  const int64_t kend = kbeg + klen < K ? kbeg + klen : K;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const TileSrc srcA = make_src<TA, false>(A, lda, m0, M, kbeg, kend, K, tid);
  const TileSrc srcB = make_src<TB, MODE == EPI_GLU>(B, ldb, n0, N, kbeg, kend, K, tid);
  const int32_t klen_here = (int32_t)(kend - kbeg);
  auto multiply = [&]() {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; i++) af[i] = read_frag<TA>(sA, wm * 64 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < 2; j++) bfr[j] = read_frag<TB>(sB, wn * 64 + j * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      if (TA && asum_on) {
#pragma unroll
        for (int i = 0; i < 2; i++) rs[i] = frag_sum(af[i], rs[i]);
      }
    }
  };
  uint4 ra[4], rb[4];
  load_tile<TA>(ra, srcA, 0);
  load_tile<TB>(rb, srcB, 0);
  for (int32_t kk = 0; kk + BK < klen_here; kk += BK) {
    store_tile<TA>(ra, sA, tid);
    store_tile<TB>(rb, sB, tid);
    __syncthreads();
    load_tile<TA>(ra, srcA, kk + BK);
    load_tile<TB>(rb, srcB, kk + BK);
    multiply();
    __syncthreads();
  }
  // last K-step: the epilogue's memory operands for the first slab are requested before the multiply
  store_tile<TA>(ra, sA, tid);
  store_tile<TB>(rb, sB, tid);
  __syncthreads();
  EpiCols<MODE, TO> cols;
  cols.init(e, n0 + wn * 64, N, lane);
  SlabPre<MODE, TO> pre;
  slab_prefetch<MODE, TO>(e, cols, pre, m0 + wm * 64, M, N, lane);
  multiply();
  __syncthreads();

  if (TA && asum_on) asum_store<2, MODE, TO>(e, rs, m0 + wm * 64, M, split, lane);
  float* stg = (float*)smem + wid * (32 * STG_LD);
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int64_t mbase = m0 + wm * 64 + i * 32;
    if (i > 0) slab_prefetch<MODE, TO>(e, cols, pre, mbase, M, N, lane);
    slab_stage(stg, acc[i][0], acc[i][1], lane);
    slab_finish<MODE, TO>(e, cols, pre, stg, mbase, n0 + wn * 64, M, N, split, partial, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// 256x256x64 kernel: 8 waves (2 x 4), each a 128x64 tile (4 x 2 MFMA tiles, 128 accumulator registers), one
// block per CU, LDS double-buffered (2 x 72 KiB).  A 128^2 tile re-reads its operands from L2 twice as often per
// FLOP and spends as many LDS cycles staging and reading fragments as the MFMAs take; here every staged byte feeds
// twice the MFMAs, and the two waves that share a SIMD are kept in opposite phases: each K-step is four barrier
// intervals  R0 | M1 | R2 | M3  (R = LDS fragment reads for half a K-step + stage one operand of the next tile,
// M = 16 MFMAs), and waves 4-7 run one interval behind waves 0-3, so while one wave of a SIMD multiplies, the other
// reads.  Hazards are covered by the barriers alone: a buffer is refilled (intervals 4k..4k+3) only after its last
// read (interval 4k-1) and read again from interval 4k+4.
// ------------------------------------------------------------------------------------------------
constexpr int BM2 = 256, BN2 = 256;
constexpr int TROW2_B = BM2 * 2 + 64;   // 576 B: contraction-major tile [64][256] bf16, rows padded by 64 B
constexpr int TILE2_B = 36864;          // 256*144 == 64*576

__device__ __forceinline__ void phase_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
}

template <bool TA, bool TB, int MODE, class TO>
__global__ __launch_bounds__(512) void gemm_mfma256_k(const bf16* __restrict__ A, const bf16* __restrict__ B, int64_t M,
                                                      int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t tiles_m,
                                                      int64_t tiles_n, int64_t klen, Epi e, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE2_B];   // [buffer][A|B]
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;            // waves w and w+4 share a SIMD; grp 1 runs one interval behind grp 0
  const int wm = grp, wn = wid & 3;    // wave tile: rows [wm*128, +128), columns [wn*64, +64)
  uint32_t tm, tn, split;
  tile_coords32(xcd_remap32(blockIdx.x, gridDim.x), (uint32_t)tiles_m, (uint32_t)tiles_n, (uint32_t)e.panel_m, tm, tn, split);
  const int64_t m0 = (int64_t)tm * BM2, n0 = (int64_t)tn * BN2;
  const bool asum_on = TA && e.asum && tn == 0 && wn == 0;   // wave-uniform
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t kbeg = split * klen;
  const int64_t kend = kbeg + klen < K ? kbeg + klen : K;
  const int32_t klen_here = (int32_t)(kend - kbeg);
  const int nk = (klen_here + BK - 1) / BK;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const TileSrc srcA = make_src<TA, false, 256, 512>(A, lda, m0, M, kbeg, kend, K, tid);
  const TileSrc srcB = make_src<TB, MODE == EPI_GLU, 256, 512>(B, ldb, n0, N, kbeg, kend, K, tid);
  uint4 ra[4], rb[4];
  load_tile<TA>(ra, srcA, 0);
  load_tile<TB>(rb, srcB, 0);
  store_tile<TA, 256, 512>(ra, smem, tid);
  store_tile<TB, 256, 512>(rb, smem + TILE2_B, tid);
  if (nk > 1) {
    load_tile<TA>(ra, srcA, BK);
    load_tile<TB>(rb, srcB, BK);
  }
  phase_barrier();
  if (grp == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

  for (int k = 0; k < nk; k++) {
    const char* cA = smem + (k & 1) * 2 * TILE2_B;
    const char* cB = cA + TILE2_B;
    char* nA = smem + ((k + 1) & 1) * 2 * TILE2_B;
    char* nB = nA + TILE2_B;
    bf16x8 af[2][4], bfr[2][2];
    // ---- R0: fragments of k-substeps 0,1; stage A of tile k+1; fetch A of tile k+2
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int i = 0; i < 4; i++) af[ks][i] = read_frag<TA, TROW2_B>(cA, wm * 128 + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < 2; j++) bfr[ks][j] = read_frag<TB, TROW2_B>(cB, wn * 64 + j * 32, ks, lane);
    }
    if (k + 1 < nk) {
      store_tile<TA, 256, 512>(ra, nA, tid);
      if (k + 2 < nk) load_tile<TA>(ra, srcA, (k + 2) * BK);
    }
    phase_barrier();
    // ---- M1
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
    if (TA && asum_on) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++) rs[i] = frag_sum(af[ks][i], rs[i]);
    }
    __builtin_amdgcn_s_setprio(0);
    phase_barrier();
    // ---- R2: fragments of k-substeps 2,3; stage B of tile k+1; fetch B of tile k+2
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
#pragma unroll
      for (int i = 0; i < 4; i++) af[ks][i] = read_frag<TA, TROW2_B>(cA, wm * 128 + i * 32, ks + 2, lane);
#pragma unroll
      for (int j = 0; j < 2; j++) bfr[ks][j] = read_frag<TB, TROW2_B>(cB, wn * 64 + j * 32, ks + 2, lane);
    }
    if (k + 1 < nk) {
      store_tile<TB, 256, 512>(rb, nB, tid);
      if (k + 2 < nk) load_tile<TB>(rb, srcB, (k + 2) * BK);
    }
    phase_barrier();
    // ---- M3
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
    if (TA && asum_on) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < 4; i++) rs[i] = frag_sum(af[ks][i], rs[i]);
    }
    __builtin_amdgcn_s_setprio(0);
    phase_barrier();
  }
  if (grp == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
  __syncthreads();

  EpiCols<MODE, TO> cols;
  cols.init(e, n0 + wn * 64, N, lane);
  if (TA && asum_on) asum_store<4, MODE, TO>(e, rs, m0 + wm * 128, M, split, lane);
  float* stg = (float*)smem + wid * (32 * STG_LD);
  SlabPre<MODE, TO> pre;
  auto one_slab = [&](auto ic) {   // a compile-time slab index: see gemm_mfma256d_k
    constexpr int i = decltype(ic)::value;
    const int64_t mbase = m0 + wm * 128 + i * 32;
    slab_prefetch<MODE, TO>(e, cols, pre, mbase, M, N, lane);
    slab_stage(stg, acc[i][0], acc[i][1], lane);
    slab_finish<MODE, TO>(e, cols, pre, stg, mbase, n0 + wn * 64, M, N, split, partial, lane);
  };
  one_slab(std::integral_constant<int, 0>{});
  one_slab(std::integral_constant<int, 1>{});
  one_slab(std::integral_constant<int, 2>{});
  one_slab(std::integral_constant<int, 3>{});
}

// ------------------------------------------------------------------------------------------------
// 256x256x64 kernel with LDS-DMA staging and 16x16x32 MFMAs.  Same wave layout (8 waves, 128x64 each), phases and
// stagger as gemm_mfma256_k, but operand tiles go from global memory straight into LDS (buffer_load ... lds): no
// staging registers and no ds_write traffic.  The 16x16x32 shape does the same work per cycle as 32x32x16 at lower
// power, which is what bounds this kernel (the chip holds ~1.6 GHz under it).
// LDS is a ring of 4 slots, one per half K-step (32 k = one MFMA k-step): [A 16 KiB][B 16 KiB].  A DMA instruction
// fills 1 KiB in lane order, so rows cannot be padded; bank conflicts are avoided by storing each unit at a permuted
// position and applying the same permutation to the SOURCE address:
//   K-contiguous operand  [256 rows][4 x 16 B]:  unit c of row r at position c ^ G[(r >> 2) & 3],  G = {0, 2, 3, 1}
//   contraction-major     [32 k][16 x 32 B]:     32-byte unit u of k-row k at position u ^ ((k & 3) | ((k >> 3) & 1) << 2)
// (each ds_read_b128 lane group then meets 16 distinct bank quads; each 32-lane half of a transposing read meets 8
// distinct 32-byte bank octants).
// Schedule per K-step k (each wave group; group 1 runs one barrier behind group 0):
//   R0: fragments of half-stage 2k, request half-stage 2k+3 | M1: 32 MFMA | R2: fragments of 2k+1, request 2k+4 | M3.
// A half-stage is requested 6 barrier intervals before its first read.  Every wave issues 4 DMA instructions per
// request, so "all but the 8 youngest complete" before each barrier retires exactly the half-stage read next, for
// both groups; fragment reads are drained before the barrier that ends their phase, so a slot is never refilled
// under a pending read.
// ------------------------------------------------------------------------------------------------

// Tile counters of the persistent kernels (dynamic tile shares): 8 per-XCD draw counters in the caller's workspace,
// zeroed on the launch stream in front of the kernel.
// Diagnostic build (-DEXP_NODMA, with -DEXP_STAMP): the main loop issues no operand requests (the MFMAs multiply stale LDS
// contents), which shows what the loop costs without its L2 -> LDS traffic.
// Diagnostic build (-DEXP_STAMP): waves 0 and 4 of every block add up the shader cycles of six segments of a tile
// (0 set-up + wait for the first half-stage, 1 main loop, 2 post-loop synchronisation, 3 request of the next tile's first
// half-stages, 4 epilogue, 5 final wait + barrier) into a buffer of their own, which vitk_debug_stamps copies out.
#ifdef EXP_STAMP
__device__ unsigned long long g_stamp[512][8];
#define STAMP_DECL unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0}, st_tiles = 0; unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[i] += t_ - st_prev; st_prev = t_; if (i == 5) st_tiles++; __builtin_amdgcn_sched_barrier(0); }
#define STAMP_STORE if ((tid & 255) == 0) { unsigned long long* d_ = g_stamp[blockIdx.x * 2 + grp]; for (int i_ = 0; i_ < 6; i_++) d_[i_] = st_sum[i_]; d_[6] = st_tiles; }
extern "C" int vitk_debug_stamps(void* dst_host) { return hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_stamp), sizeof(g_stamp)) == hipSuccess ? 0 : 1; }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_STORE
#endif

template <bool TA, bool TB, int MODE, class TO>
__global__ __launch_bounds__(512) void gemm_mfma256d_k(const bf16* __restrict__ A, const bf16* __restrict__ B, int64_t M,
                                                       int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t tiles_m,
                                                       int64_t tiles_n, int64_t klen, Epi e, float* __restrict__ partial,
                                                       unsigned int* __restrict__ ctr) {
  // the operand ring; the epilogue lays its staging slabs (8 x 8704 B) and, behind them, the operand stream's rings
  // (8 x STREAM_B) over it
  constexpr int STREAM_BASE = 8 * 32 * STG_LD * 4;
  __shared__ __attribute__((aligned(1024))) char smem[STREAM_BASE + 8 * STREAM_B];
  static_assert(STREAM_BASE + 8 * STREAM_B >= 4 * SLOT_B, "the epilogue's areas cover the operand ring");
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;
  STAMP_DECL
  const int wm = grp, wn = wid & 3;
  // The host launches one block per CU and every block walks several tiles (a block launch costs a few microseconds,
  // noticeable beside a short-K tile).  Which tiles is decided as the kernel runs: the blocks of an XCD (blockIdx & 7)
  // share that XCD's run of consecutive tile ids and (when `ctr` is given) draw the next one from a counter, so a CU that is slow, or not
  // available at all because another kernel (a collective on a side stream) holds it, takes fewer tiles instead of
  // making the whole launch wait for its fixed share.  The draw for the next tile is issued before the current one
  // starts and consumed after its epilogue.
  // tile indices are 32-bit (the host checks the tile count): a 64-bit division is ~100 scalar instructions, and the
  // five of them that place a tile used to be a tenth of a short-K tile's time
  const uint32_t total_tiles = (uint32_t)tiles_m * (uint32_t)tiles_n * (uint32_t)((K + klen - 1) / klen);
  const uint32_t xcd = blockIdx.x & 7;
  const uint32_t xcd_tiles = (total_tiles >> 3) + (xcd < (total_tiles & 7) ? 1 : 0);
  const uint32_t xcd_blocks = (gridDim.x - xcd + 7) >> 3;
  const bool dynamic = ctr != nullptr;   // null: fixed shares, block b of an XCD takes that XCD's tiles b, b + blocks, ...
  __shared__ unsigned int s_next;
  for (uint32_t jt = blockIdx.x >> 3; jt < xcd_tiles;) {
  unsigned int drawn = 0;
  if (dynamic && tid == 0) drawn = __hip_atomic_fetch_add(&ctr[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t tm, tn, split;
  tile_coords32(xcd_remap32(xcd + 8 * jt, total_tiles), (uint32_t)tiles_m, (uint32_t)tiles_n, (uint32_t)e.panel_m, tm, tn, split);
  const int64_t m0 = (int64_t)tm * BM2, n0 = (int64_t)tn * BN2;
  const int64_t kbeg = (int64_t)split * klen;
  const int64_t kend = kbeg + klen < K ? kbeg + klen : K;
  const int32_t klen_here = (int32_t)(kend - kbeg);
  const int nk = (klen_here + BK - 1) / BK;
  const bool asum_on = TA && e.asum && tn == 0 && wn == 0;   // wave-uniform
  float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i][j][r] = 0.f;

  const DmaSrc sa = make_dma<TA, false>(A, lda, m0, M, kbeg, kend, K, wid, lane);
  const DmaSrc sb = make_dma<TB, MODE == EPI_GLU>(B, ldb, n0, N, kbeg, kend, K, wid, lane);
  auto request = [&](int h) {   // unconditional: half-stages past the K range read zeros and are never consumed
    char* slot = smem + (h & 3) * SLOT_B;
    dma_half<TA>(sa, slot, wid, h);
    dma_half<TB>(sb, slot + HS_B, wid, h);
  };
  bf16x8 af[8], bfr[4];
  auto read_frags = [&](const char* slot) {
#pragma unroll
    for (int j = 0; j < 4; j++) bfr[j] = read_frag16<TB>(slot + HS_B, wn * 64 + j * 16, lane);
#pragma unroll
    for (int i = 0; i < 8; i++) af[i] = read_frag16<TA>(slot, wm * 128 + i * 16, lane);
  };
  auto multiply = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if (TA && asum_on) {
#pragma unroll
      for (int i = 0; i < 8; i++) rs[i] = frag_sum(af[i], rs[i]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  request(0);
  request(1);
  request(2);
  dma_barrier(false);   // half-stage 0 complete
  STAMP(0)
  if (grp == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

  for (int k = 0; k < nk; k++) {
    read_frags(smem + ((2 * k) & 3) * SLOT_B);       // R0
#ifndef EXP_NODMA
    request(2 * k + 3);
#endif
    dma_barrier(true);
    multiply();                                      // M1
    dma_barrier(false);
    read_frags(smem + ((2 * k + 1) & 3) * SLOT_B);   // R2
#ifndef EXP_NODMA
    request(2 * k + 4);
#endif
    dma_barrier(true);
    multiply();                                      // M3
    dma_barrier(false);
  }
  STAMP(1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (empty) requests still target LDS
  if (grp == 0) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
  __syncthreads();
  STAMP(2)
  STAMP(3)
  EpiCols<MODE, TO> cols;
  cols.init(e, n0 + wn * 64, N, lane);
  if (TA && asum_on) {   // row sums: lanes l, l^16, l^32, l^48 hold the four k-quarters of row (l & 15) of each 16-row tile
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float v = rs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int64_t m = m0 + wm * 128 + i * 16 + (lane & 15);
      if (lane < 16 && m < M) {
        if (MODE == MODE_PARTIAL) e.asum_partial[split * M + m] = v;
        else ((TO*)e.asum)[m] = from_f<TO>(v);
      }
    }
  }
  float* stg = (float*)smem + wid * (32 * STG_LD);
  // the epilogue of the tile, with the activation of a fused epilogue switched on once out here (ACT_DISPATCH)
  auto run_epilogue = [&](auto actc) {
    constexpr int ACTC = decltype(actc)::value;
    if constexpr (epi_streams<MODE, TO>()) {
      if (e.stream && m0 + BM2 <= M && n0 + BN2 <= N) {   // a whole tile (block-uniform)
        epilogue_stream16<MODE, TO, ACTC>(e, cols, stg, smem + STREAM_BASE + wid * STREAM_B, acc, m0 + wm * 128, lane);
        return;
      }
    }
    SlabPre<MODE, TO> pre;
    // the four slabs, spelled out: a slab index that is not a compile-time constant would index the accumulator array
    // dynamically and send all of it to scratch memory (the unroll pragma alone is not honoured for the largest epilogues)
    auto one_slab = [&](auto slc) {
      constexpr int sl = decltype(slc)::value;
      const int64_t mbase = m0 + wm * 128 + sl * 32;
      slab_prefetch<MODE, TO>(e, cols, pre, mbase, M, N, lane);
      slab_stage16(stg, acc, sl, lane);
      slab_finish<MODE, TO, ACTC>(e, cols, pre, stg, mbase, n0 + wn * 64, M, N, split, partial, lane);
    };
    one_slab(std::integral_constant<int, 0>{});
    one_slab(std::integral_constant<int, 1>{});
    one_slab(std::integral_constant<int, 2>{});
    one_slab(std::integral_constant<int, 3>{});
  };
  if constexpr (MODE == EPI_GLU || MODE == EPI_DGLU || MODE == EPI_ACT || MODE == EPI_DACT) {
    ACT_SWITCH(e.act, ACT, run_epilogue(std::integral_constant<int, ACT>{}));
  } else {
    run_epilogue(std::integral_constant<int, -1>{});
  }
  STAMP(4)
  if (dynamic && tid == 0) s_next = drawn;
  __syncthreads();   // the staging slabs overlay the ring: done before the next tile's first requests
  STAMP(5)
  jt = dynamic ? xcd_blocks + s_next : jt + xcd_blocks;
  }
  STAMP_STORE
}

template <int MODE, class TI, class TO>
__global__ __launch_bounds__(256) void splitk_reduce_k(const float* __restrict__ partial, int64_t M, int64_t N, int splits,
                                                       Epi e) {
  const int64_t total = M * N;
  if (MODE == VITK_EPI_STORE && N % 4 == 0 && e.ldo % 4 == 0 && !e.bias) {   // 4 columns per thread, 16-byte accesses
    const int64_t nq = N / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < M * nq; q += (int64_t)gridDim.x * 256) {
      const int64_t m = q / nq, n = (q - m * nq) * 4;
      float4 s = *(const float4*)(partial + m * N + n);
      for (int k = 1; k < splits; k++) {
        const float4 t = *(const float4*)(partial + (int64_t)k * total + m * N + n);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      Vec4<TO> o;
      o.v[0] = s.x; o.v[1] = s.y; o.v[2] = s.z; o.v[3] = s.w;
      o.store((TO*)e.out + m * e.ldo + n);
    }
  } else if (MODE == VITK_EPI_RESIDUAL && N % 4 == 0 && e.ldo % 4 == 0) {   // the MFMA kernels' 4-column epilogue on the summed partials
    const int64_t nq = N / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < M * nq; q += (int64_t)gridDim.x * 256) {
      const int64_t m = q / nq, n = (q - m * nq) * 4;
      EpiPre<VITK_EPI_RESIDUAL, TO> pre;
      EpiVec<VITK_EPI_RESIDUAL, TO>::prefetch(e, m, n, pre);
      float4 s4 = *(const float4*)(partial + m * N + n);
      for (int k = 1; k < splits; k++) {
        const float4 t = *(const float4*)(partial + (int64_t)k * total + m * N + n);
        s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w;
      }
      float v[4] = {s4.x, s4.y, s4.z, s4.w}, bias[4] = {0.f, 0.f, 0.f, 0.f}, gam[4] = {1.f, 1.f, 1.f, 1.f};
      if (e.bias) {
        Vec4<bf16> b; b.load((const bf16*)e.bias + n);
#pragma unroll
        for (int j = 0; j < 4; j++) bias[j] = b.v[j];
      }
      if (e.gamma) {
        Vec4<TO> g; g.load((const TO*)e.gamma + n);
#pragma unroll
        for (int j = 0; j < 4; j++) gam[j] = g.v[j];
      }
      EpiVec<VITK_EPI_RESIDUAL, TO>::apply(e, m, n, v, bias, gam, pre);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      float s = 0.f;
      for (int k = 0; k < splits; k++) s += partial[(int64_t)k * total + i];
      EpiOp<MODE, TI, TO>::apply(e, i / N, i % N, s);
    }
  }
  if (e.asum) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
      float s = 0.f;
      for (int k = 0; k < splits; k++) s += e.asum_partial[(int64_t)k * M + i];
      ((TO*)e.asum)[i] = from_f<TO>(s);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static bool mfma_ok(const VitkGemm* g) {
  if (g->in_dtype != VITK_BF16) return false;
  if (g->trans_a && !g->trans_b) return false;  // A^T x B^T-layout never occurs on the path
  const int ep = g->epilogue;
  const bool nt = !g->trans_a && !g->trans_b, nn = !g->trans_a && g->trans_b;
  if ((ep == VITK_EPI_RESIDUAL || ep == VITK_EPI_PATCH || ep == VITK_EPI_ACT || ep == VITK_EPI_GLU || ep == VITK_EPI_ROPE) && !nt) return false;
  if (ep == VITK_EPI_ROPE && g->out_dtype != VITK_BF16) return false;
  if ((ep == VITK_EPI_DACT || ep == VITK_EPI_DGLU) && !nn) return false;
  if (ep == VITK_EPI_GLU && g->N % 64) return false;
  // the epilogue moves 4 columns per lane with vector accesses, 8 for bf16 outputs of every mode but PATCH
  if (g->N % 4 || g->ldo % 4) return false;
  if (g->out_dtype == VITK_BF16 && (ep == VITK_EPI_STORE || ep == VITK_EPI_GLU || ep == VITK_EPI_DGLU || ep == VITK_EPI_ROPE ||
                                   ep == VITK_EPI_ACT || ep == VITK_EPI_DACT || ep == VITK_EPI_RESIDUAL) && (g->N % 8 || g->ldo % 8))
    return false;
  if (!aligned16(g->out) || !aligned16(g->res) || !aligned16(g->aux) || !aligned16(g->bias) || !aligned16(g->gamma) ||
      !aligned16(g->pos))
    return false;
  if (!aligned16(g->a) || !aligned16(g->b)) return false;
  // buffer-load offsets are 32-bit: a contraction-major operand is addressed from its first K row
  const int64_t lim = 0xfffffff0ll;
  if (g->trans_a && g->K * g->lda * 2 >= lim) return false;
  if (g->trans_b && g->K * g->ldb * 2 >= lim) return false;
  if (!g->trans_b && g->N * g->ldb * 2 >= lim) return false;
  if (!g->trans_a && 128 * g->lda * 2 >= lim) return false;
  if (g->lda % 8 || g->ldb % 8) return false;
  if (!g->trans_a && g->K % 8) return false;
  if (g->trans_a && g->M % 8) return false;
  if (!g->trans_b && g->K % 8) return false;
  if (g->trans_b && g->N % 8) return false;
  // the kernels place tiles with 32-bit arithmetic: at most 2^24 tiles of 128 x 128 (times up to 64 K-splits)
  if (ceil_div(g->M, 128) * ceil_div(g->N, 128) >= (1ll << 24)) return false;
  return g->M > 0 && g->N > 0 && g->K > 0;
}

// VITK_EPI_STREAM=0 keeps the register-prefetched epilogues on whole tiles too (A/B runs)
static int epilogue_stream_default() {
  static const int v = [] {
    const char* f = getenv("VITK_EPI_STREAM");
    return f ? atoi(f) : 1;
  }();
  return v;
}

// split the contraction when the output has too few tiles to fill 256 CUs (weight gradients)
static bool use_tile256(const VitkGemm* g);
static int pick_splits(const VitkGemm* g) {
  // STORE, and the residual epilogue on the 128^2 kernel when it is asked for by name (algo 4: the rows a ragged round
  // of 256^2 tiles leaves over are few tiles with the whole contraction each; split, they fill the chip, and the
  // reduction pass applies the epilogue)
  const bool big = use_tile256(g);
  if (g->epilogue != VITK_EPI_STORE && !(g->epilogue == VITK_EPI_RESIDUAL && g->algo == 4 && !big && !g->trans_a && !g->trans_b)) return 1;
  const int64_t tiles = big ? ceil_div(g->M, BM2) * ceil_div(g->N, BN2) : ceil_div(g->M, BM) * ceil_div(g->N, BN);
  const int64_t slots = big ? 256 : 768;       // blocks one round holds: one 256^2 block or three 128^2 blocks per CU
  if (g->K < 2048) return 1;
  if (tiles >= slots) {
    // More tiles than one round holds: a last round that is mostly empty costs a whole round (288 tiles on 256 CUs take
    // two).  Row-major-A problems are cut along M instead (ragged_round_cut); for a weight gradient with a long
    // contraction, splitting K evens the rounds out at the price of the partial sums' traffic.  Model, in microseconds:
    // rounds x K-steps per tile x 1.65 us per 256^2 K-step, plus 8 bytes per partial element at 5 TB/s.
    if (!big || !g->trans_a || g->K < 8192) return 1;
    auto cost = [&](int64_t sp) {
      const double gemm = (double)ceil_div(tiles * sp, slots) * (double)ceil_div(ceil_div(g->K, sp), BK) * 1.65;
      return gemm + (sp > 1 ? (double)sp * (double)g->M * (double)g->N * 8.0 / 5e6 : 0.0);
    };
    int64_t best = 1;
    double best_cost = cost(1) * 0.9;   // a split has to win by 10 %
    for (int64_t sp = 2; sp <= 16; sp++)
      if (cost(sp) < best_cost) { best = sp; best_cost = cost(sp); }
    return (int)best;
  }
  // the fewest splits (least partial traffic) that fill whole rounds to at least 94 %, at most two rounds
  const int64_t smax = g->K / 1024 < 64 ? g->K / 1024 : 64;
  int64_t best = 1;
  double best_eff = 0.0;
  for (int64_t s = 1; s <= smax && tiles * s <= 2 * slots; s++) {
    const double eff = (double)(tiles * s) / (double)(ceil_div(tiles * s, slots) * slots);
    if (eff > best_eff + 1e-9) { best = s; best_eff = eff; }
    if (eff >= 0.94) { best = s; break; }
  }
  return (int)best;
}

constexpr size_t TILE_CTR_BYTES = 256;
static size_t partial_bytes(const VitkGemm* g, int splits) {
  return splits > 1 ? (size_t)splits * (size_t)g->M * ((size_t)g->N + (g->a_colsum ? 1 : 0)) * sizeof(float) : 0;
}
static int64_t ragged_round_cut(const VitkGemm* g);
extern "C" size_t vitk_gemm_workspace_bytes(const VitkGemm* g) {
  if (!g) return 0;
  if (!mfma_ok(g) || g->algo == 1) return g->a_colsum ? vitk_colsum_workspace_bytes(g->M) : 0;   // separate column-sum pass
  if (const int64_t m1 = ragged_round_cut(g)) {   // the rows after the cut are a problem of their own, usually split along K
    VitkGemm tail = *g;
    tail.M = g->M - m1;
    tail.algo = 4;
    const size_t part = partial_bytes(&tail, pick_splits(&tail));
    return g->dynamic_tiles ? (part + 255) / 256 * 256 + 256 : part;
  }
  const int s = pick_splits(g);
  const size_t part = s > 1 ? (size_t)s * (size_t)g->M * ((size_t)g->N + (g->a_colsum ? 1 : 0)) * sizeof(float) : 0;
  return g->dynamic_tiles && use_tile256(g) ? (part + 255) / 256 * 256 + 256 : part;   // + the tile counters
}

// Cache policy of the epilogues' 16-byte output stores (st16): VITK_WT_STORES=0/1 overrides the default (A/B runs).
static int write_through_default() {
  static const int v = [] {
    const char* f = getenv("VITK_WT_STORES");
    return f ? atoi(f) : 0;
  }();
  return v;
}

static Epi make_epi(const VitkGemm* g) {
  Epi e;
  e.out = g->out; e.bias = g->bias; e.res = g->res; e.gamma = g->gamma; e.row_idx = g->row_idx; e.aux = g->aux;
  e.pos = g->pos; e.ldo = g->ldo; e.N = g->N; e.tokens = g->tokens; e.seq = g->seq; e.prefix = g->prefix;
  e.seed = g->seed; e.alpha = g->alpha;
  e.thr = g->drop_p > 0.f ? drop_threshold(g->drop_p) : 0;
  e.keep_scale = 1.f / (1.f - g->drop_p);
  e.act = g->act; e.use_limit = g->use_limit; e.use_extra = g->use_extra; e.limit = g->limit; e.extra = g->extra_bias;
  e.asum = g->a_colsum; e.asum_partial = nullptr; e.m_off = 0;
  e.pos_f32 = g->pos_f32 || g->in_dtype == VITK_F32;
  e.rope = g->rope; e.rope_width = g->rope_width; e.rope_emu = g->rope_emulate_bf16;
  e.qk_w = g->qk_norm_weight; e.qk_b = g->qk_norm_bias; e.qk_rstd = g->qk_rstd; e.qk_mean = g->qk_mean;
  e.qk_eps = g->qk_eps; e.qk_ln = g->qk_layernorm; e.rows_total = g->M;
  e.wt = g->store_policy >= 2 || (g->store_policy == 0 && write_through_default());
  e.ntl = g->store_policy == 3;
  e.stream = !g->row_idx && epilogue_stream_default();
  // Panels of the tile walk: an output at most four 256-column tiles wide does better with short panels (consecutive
  // tiles, which run side by side on one XCD, then reach the next column of the same rows sooner): N = 768 on MI355X,
  // panel 4 against 8: fc2 + residual -2.6 %, fc1 input gradient -2.2 %, fc1 weight gradient -4 %; wider outputs keep 8
  // (N = 2304, 6144: panels of 2 and 4 cost 4-9 %)
  e.panel_m = g->panel_m > 0 ? g->panel_m : (ceil_div(g->N, BN2) <= 4 ? 4 : PANEL_M);
  return e;
}

template <class TI, class TO>
static void launch_simple(const VitkGemm* g, const Epi& e, hipStream_t st) {
  dim3 grid((unsigned)ceil_div(g->N, 64), (unsigned)ceil_div(g->M, 64)), block(256);
  const int64_t sam = g->trans_a ? 1 : g->lda, sak = g->trans_a ? g->lda : 1;
  const int64_t sbk = g->trans_b ? g->ldb : 1, sbn = g->trans_b ? 1 : g->ldb;
#define LS(MODE)                                                                                                     \
  hipLaunchKernelGGL((gemm_simple_k<TI, MODE, TO>), grid, block, 0, st, (const TI*)g->a, (const TI*)g->b, g->M, g->N, \
                     g->K, sam, sak, sbk, sbn, e)
  if (g->epilogue == VITK_EPI_STORE) LS(VITK_EPI_STORE);
  else if (g->epilogue == VITK_EPI_RESIDUAL) LS(VITK_EPI_RESIDUAL);
  else LS(VITK_EPI_PATCH);
#undef LS
}

// algo 3 / 5 force the 256^2 tile (3: register-staged kernel, 5: LDS-DMA kernel, which is also the automatic choice),
// algo 4 the 128^2 kernel; otherwise large problems take the 256^2 tile
static bool use_tile256(const VitkGemm* g) {
  if (g->algo == 3 || g->algo == 5) return true;
  if (g->algo == 4) return false;
  // measured on MI355X (ViT-B shapes): the 256^2 tile wins once the output is wide or the contraction long;
  // narrow, short problems (N = K = 768) keep more CUs busy with 128^2 tiles
  if (g->M >= 512 && g->N >= 512 && (g->N >= 1024 || g->K >= 2048)) return true;
  // the residual epilogue with its operand stream (whole tiles, no row gather) costs the large tile less than three
  // small workgroups per CU gain by overlapping their epilogues: out_proj (N = K = 768) 0.18 against 0.20 ms
  return g->epilogue == VITK_EPI_RESIDUAL && !g->row_idx && epilogue_stream_default() && g->M >= 2048 && g->N >= 512 && g->N % 256 == 0;
}

static int64_t cu_count() {
  static int cached = 0;
  if (!cached) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached = n;
  }
  return cached;
}

// blocks of a DMA-kernel launch: the CU count, or VITK_GEMM_BLOCKS (experiments)
static int64_t gemm_blocks() {
  static const int64_t n = [] {
    const char* f = getenv("VITK_GEMM_BLOCKS");
    const int64_t v = f ? atoll(f) : 0;
    return v > 0 && v <= cu_count() ? v : (int64_t)cu_count();
  }();
  return n;
}

// Tile shares of the persistent kernels: fixed by default (a little faster when nothing else runs on the GPU), drawn
// from counters when the caller sets VitkGemm::dynamic_tiles because it expects another stream's kernels to hold CUs
// during the launch (data-parallel training: the gradient all-reduce runs beside the backward pass).  The counters are
// the last TILE_CTR_BYTES of the workspace.
static unsigned int* tile_counters(const VitkGemm* g, int splits, hipStream_t st) {
  if (!g->dynamic_tiles || !g->workspace) return nullptr;
  const size_t off = (partial_bytes(g, splits) + 255) / 256 * 256;
  if ((size_t)g->workspace_bytes < off + TILE_CTR_BYTES) return nullptr;
  unsigned int* ctr = (unsigned int*)((char*)g->workspace + off);
  if (hipMemsetAsync(ctr, 0, TILE_CTR_BYTES, st) != hipSuccess) return nullptr;
  return ctr;
}

template <bool TA, bool TB, int MODE, class TO>
static void launch_mfma_one(const VitkGemm* g, const Epi& e, int splits, int64_t klen, float* partial, hipStream_t st) {
  if (use_tile256(g)) {
    const int64_t tm = ceil_div(g->M, BM2), tn = ceil_div(g->N, BN2);
    if (g->algo != 3) {   // algo 3 keeps the register-staged 32x32x16 kernel for comparisons
      int64_t nblk = tm * tn * splits;
      if (nblk > gemm_blocks()) nblk = gemm_blocks();   // at most one resident block per CU (LDS); each walks several tiles
      hipLaunchKernelGGL((gemm_mfma256d_k<TA, TB, MODE, TO>), dim3((unsigned)nblk), dim3(512), 0, st,
                         (const bf16*)g->a, (const bf16*)g->b, g->M, g->N, g->K, g->lda, g->ldb, tm, tn, klen, e, partial,
                         tile_counters(g, splits, st));
      return;
    }
    hipLaunchKernelGGL((gemm_mfma256_k<TA, TB, MODE, TO>), dim3((unsigned)(tm * tn * splits)), dim3(512), 0, st,
                       (const bf16*)g->a, (const bf16*)g->b, g->M, g->N, g->K, g->lda, g->ldb, tm, tn, klen, e, partial);
    return;
  }
  const int64_t tm = ceil_div(g->M, BM), tn = ceil_div(g->N, BN);
  dim3 grid((unsigned)(tm * tn * splits)), block(256);
  hipLaunchKernelGGL((gemm_mfma_k<TA, TB, MODE, TO>), grid, block, 0, st, (const bf16*)g->a, (const bf16*)g->b, g->M, g->N,
                     g->K, g->lda, g->ldb, tm, tn, klen, e, partial);
}

template <bool TA, bool TB, class TO>
static void launch_mfma_layout(const VitkGemm* g, const Epi& e, int splits, int64_t klen, hipStream_t st) {
  if (splits > 1) {
    Epi ep = e;
    ep.asum_partial = (float*)g->workspace + (size_t)splits * g->M * g->N;
    launch_mfma_one<TA, TB, MODE_PARTIAL, TO>(g, ep, splits, klen, (float*)g->workspace, st);
    const int64_t total = g->M * g->N;
    int64_t blocks = ceil_div(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((splitk_reduce_k<VITK_EPI_STORE, bf16, TO>), dim3((unsigned)blocks), dim3(256), 0, st,
                       (const float*)g->workspace, g->M, g->N, splits, ep);
  } else {
    launch_mfma_one<TA, TB, VITK_EPI_STORE, TO>(g, e, 1, klen, nullptr, st);
  }
}

// A 256^2-tile launch runs one block per CU, so a tile count just above a multiple of the CU count leaves most CUs idle
// for a whole extra round (816 tiles on 256 CUs: 3.2 rounds cost 4).  Row-major-A problems are then cut in two along M:
// whole rounds on the 256^2 kernel, the remaining rows on the 128^2 kernel (3 blocks per CU, a quarter of the work
// per tile).  Returns the number of rows for the first launch, or 0 for no cut.
static int64_t ragged_round_cut(const VitkGemm* g) {
  if (g->algo != 0 && g->algo != 2) return 0;
  if (g->trans_a || !use_tile256(g)) return 0;
  if (g->epilogue != VITK_EPI_STORE && g->epilogue != VITK_EPI_RESIDUAL && g->epilogue != VITK_EPI_ROPE) return 0;
  if (pick_splits(g) != 1) return 0;
  const int64_t cus = cu_count(), tm = ceil_div(g->M, BM2), tn = ceil_div(g->N, BN2);
  const int64_t tiles = tm * tn, rounds = tiles / cus, rem = tiles % cus;
  if (rounds < 1 || rem == 0 || rem * 10 > cus * 6) return 0;     // a last round that is over 60% full stays
  const int64_t tm_main = rounds * cus / tn;
  if (tm_main < 1 || tm_main >= tm) return 0;
  // The cut trades the last round of the large kernel (one tile time, whatever its fill) for a launch of the small kernel
  // over the rows behind the cut.  Measured on MI355X, in microseconds: a 256^2 tile takes 1.3 per K-step + 5 (+ 8 with
  // the residual epilogue); the small kernel 12 + its FLOPs at 0.9 PFLOP/s (+ 9 for the reduction of a split contraction).
  // Short contractions (K = 768: a 21 us tile against a 26-30 us launch) are therefore left whole.
  const double t_tile = (double)ceil_div(g->K, BK) * 1.3 + 5.0 + (g->epilogue == VITK_EPI_RESIDUAL ? 8.0 : 0.0);
  const double rows_left = (double)(g->M - tm_main * BM2);
  const double t_left = 12.0 + 2.0 * rows_left * (double)g->N * (double)g->K / 0.9e9 + (g->K >= 2048 ? 9.0 : 0.0);
  if (t_left > 0.9 * t_tile) return 0;
  return tm_main * BM2;
}

template <class TO>
static int launch_mfma(const VitkGemm* g, const Epi& e, hipStream_t st);

template <class TO>
static int launch_mfma_cut(const VitkGemm* g, const Epi& e, int64_t m1, hipStream_t st) {
  const size_t si = 2, so = g->out_dtype == VITK_F32 ? 4 : 2;
  VitkGemm a = *g, b = *g;
  a.M = m1;
  a.algo = 5;                                   // 256^2 tiles, no further cut
  b.M = g->M - m1;
  b.algo = 4;                                   // 128^2 tiles
  b.a = (const char*)g->a + (size_t)m1 * g->lda * si;
  if (g->row_idx) b.row_idx = g->row_idx + m1;  // gathered rows: out / res stay whole, the index list moves
  else {
    b.out = (char*)g->out + (size_t)m1 * g->ldo * so;
    if (g->res) b.res = (const char*)g->res + (size_t)m1 * g->ldo * so;
  }
  if (g->aux && g->epilogue != VITK_EPI_ROPE) b.aux = (char*)g->aux + (size_t)m1 * g->N * si;   // ROPE: indexed by global row
  Epi ea = make_epi(&a), eb = make_epi(&b);
  eb.m_off = m1;
  ea.rows_total = eb.rows_total = g->M;
  (void)e;
  launch_mfma<TO>(&a, ea, st);
  launch_mfma<TO>(&b, eb, st);
  return 0;
}

template <class TO>
static int launch_mfma(const VitkGemm* g, const Epi& e, hipStream_t st) {
  if (const int64_t m1 = ragged_round_cut(g)) return launch_mfma_cut<TO>(g, e, m1, st);
  int splits = pick_splits(g);
  if (splits > 1 && (!g->workspace || (size_t)g->workspace_bytes < (size_t)splits * g->M * (g->N + (g->a_colsum ? 1 : 0)) * sizeof(float)))
    splits = 1;
  int64_t klen = ceil_div(ceil_div(g->K, splits), BK) * BK;
  splits = (int)ceil_div(g->K, klen);
  if (g->epilogue == VITK_EPI_RESIDUAL) {
    if (splits > 1) {   // partial products, then one pass that sums them and applies the epilogue element by element
      launch_mfma_one<false, false, MODE_PARTIAL, TO>(g, e, splits, klen, (float*)g->workspace, st);
      const int64_t total = g->M * g->N;
      int64_t blocks = ceil_div(total, 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL((splitk_reduce_k<VITK_EPI_RESIDUAL, bf16, TO>), dim3((unsigned)blocks), dim3(256), 0, st,
                         (const float*)g->workspace, g->M, g->N, splits, e);
    } else {
      launch_mfma_one<false, false, VITK_EPI_RESIDUAL, TO>(g, e, 1, klen, nullptr, st);
    }
  }
  else if (g->epilogue == VITK_EPI_PATCH) launch_mfma_one<false, false, VITK_EPI_PATCH, TO>(g, e, 1, klen, nullptr, st);
  else if (g->epilogue >= VITK_EPI_ACT && g->epilogue <= VITK_EPI_ROPE) {   // bf16 outputs only (checked by vitk_gemm)
    if constexpr (sizeof(TO) == 2) {
      if (g->epilogue == VITK_EPI_ACT) launch_mfma_one<false, false, VITK_EPI_ACT, TO>(g, e, 1, klen, nullptr, st);
      else if (g->epilogue == VITK_EPI_GLU) launch_mfma_one<false, false, VITK_EPI_GLU, TO>(g, e, 1, klen, nullptr, st);
      else if (g->epilogue == VITK_EPI_DACT) launch_mfma_one<false, true, VITK_EPI_DACT, TO>(g, e, 1, klen, nullptr, st);
      else if (g->epilogue == VITK_EPI_DGLU) launch_mfma_one<false, true, VITK_EPI_DGLU, TO>(g, e, 1, klen, nullptr, st);
      else launch_mfma_one<false, false, VITK_EPI_ROPE, TO>(g, e, 1, klen, nullptr, st);
    }
  }
  else if (!g->trans_a && !g->trans_b) launch_mfma_layout<false, false, TO>(g, e, splits, klen, st);
  else if (!g->trans_a && g->trans_b) launch_mfma_layout<false, true, TO>(g, e, splits, klen, st);
  else launch_mfma_layout<true, true, TO>(g, e, splits, klen, st);
  return 0;
}

extern "C" int vitk_gemm(const VitkGemm* g, void* stream) {
  VITK_REQUIRE(g && g->a && g->b && g->out, "gemm: null pointer");
  VITK_REQUIRE(g->M >= 0 && g->N > 0 && g->K > 0, "gemm: bad shape M=%lld N=%lld K=%lld", (long long)g->M, (long long)g->N,
               (long long)g->K);
  VITK_REQUIRE(g->epilogue >= 0 && g->epilogue <= 7, "gemm: unknown epilogue %d", g->epilogue);
  const bool fusedact = g->epilogue >= VITK_EPI_ACT && g->epilogue <= VITK_EPI_DGLU;
  VITK_REQUIRE(g->in_dtype == VITK_F32 || g->in_dtype == VITK_BF16, "gemm: bad in_dtype %d", g->in_dtype);
  VITK_REQUIRE(g->out_dtype == VITK_F32 || g->out_dtype == VITK_BF16, "gemm: bad out_dtype %d", g->out_dtype);
  VITK_REQUIRE(!(g->in_dtype == VITK_F32 && g->out_dtype == VITK_BF16), "gemm: fp32 operands with bf16 output unsupported");
  VITK_REQUIRE(g->drop_p >= 0.f && g->drop_p < 1.f, "gemm: dropout p=%f out of range", g->drop_p);
  const int64_t out_cols = g->epilogue == VITK_EPI_GLU ? g->N / 2 : (g->epilogue == VITK_EPI_DGLU ? 2 * g->N : g->N);
  VITK_REQUIRE(g->lda >= (g->trans_a ? g->M : g->K) && g->ldb >= (g->trans_b ? g->N : g->K) && g->ldo >= out_cols,
               "gemm: leading dimension smaller than the row length");
  if (fusedact) VITK_REQUIRE(g->aux && g->in_dtype == VITK_BF16 && g->out_dtype == VITK_BF16 && g->act >= 0 && g->act <= 4,
                             "gemm: fused activation epilogues need bf16 operands/output, aux and a valid activation");
  if (g->epilogue == VITK_EPI_RESIDUAL) VITK_REQUIRE(g->res, "gemm: RESIDUAL epilogue needs res");
  if (g->epilogue == VITK_EPI_PATCH)
    VITK_REQUIRE(g->tokens > 0 && g->M % g->tokens == 0 && g->seq >= g->prefix + g->tokens,
                 "gemm: PATCH epilogue needs tokens | M and seq >= prefix + tokens");
  if (g->epilogue == VITK_EPI_ROPE)
    VITK_REQUIRE((g->rope || g->qk_norm_weight) && g->rope_width > 0 && g->rope_width % 64 == 0 && g->N % 64 == 0 && g->seq > 0 && g->prefix >= 0 &&
                     g->prefix < g->seq && g->in_dtype == VITK_BF16 && g->out_dtype == VITK_BF16 && !g->trans_a && !g->trans_b &&
                     ((uintptr_t)g->rope & 15) == 0,
                 "gemm: ROPE epilogue needs a table, bf16 operands and output, row-major A and nn.Linear-layout B, sections "
                 "of whole 64-column heads and 0 <= prefix < seq");
  if (g->qk_norm_weight)
    VITK_REQUIRE(g->epilogue == VITK_EPI_ROPE && g->aux && g->qk_rstd && (!g->qk_layernorm || g->qk_mean) &&
                     ((uintptr_t)g->qk_norm_weight & 15) == 0 && ((uintptr_t)g->qk_norm_bias & 15) == 0,
                 "gemm: per-head q/k normalisation needs the ROPE epilogue, aux for the un-normalised q and k, and rstd "
                 "(and mean for LayerNorm) outputs");
  if (g->pos_f32) VITK_REQUIRE(g->epilogue == VITK_EPI_PATCH && g->out_dtype == VITK_F32, "gemm: pos_f32 needs the PATCH epilogue and a float32 output");
  if (g->a_colsum) VITK_REQUIRE(g->trans_a && g->epilogue == VITK_EPI_STORE, "gemm: a_colsum needs trans_a = 1 and the STORE epilogue");
  if (g->M == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const Epi e = make_epi(g);
  const bool can = mfma_ok(g);
  VITK_REQUIRE(g->algo < 2 || can, "gemm: MFMA kernel cannot take this problem (dtype/alignment/layout)");
  VITK_REQUIRE(!fusedact || (can && g->algo != 1), "gemm: fused activation epilogues exist in the MFMA kernel only");
  VITK_REQUIRE(g->epilogue != VITK_EPI_ROPE || (can && g->algo != 1), "gemm: the ROPE epilogue exists in the MFMA kernel only");
  if (can && g->algo != 1) {
    if (g->out_dtype == VITK_F32) launch_mfma<float>(g, e, st);
    else launch_mfma<bf16>(g, e, st);
  } else {
    VITK_REQUIRE(ceil_div(g->M, 64) <= 65535, "gemm: M too large for the portable kernel");
    if (g->in_dtype == VITK_F32) launch_simple<float, float>(g, e, st);
    else if (g->out_dtype == VITK_F32) launch_simple<bf16, float>(g, e, st);
    else launch_simple<bf16, bf16>(g, e, st);
    if (g->a_colsum) {   // the portable kernel has no fused row sums: one pass over A ([K, M], column sums)
      VITK_REQUIRE(g->workspace && (size_t)g->workspace_bytes >= vitk_colsum_workspace_bytes(g->M), "gemm: a_colsum needs the workspace");
      if (int rc = vitk_colsum(g->a, g->in_dtype, g->lda, nullptr, nullptr, 0, 0, g->a_colsum, g->out_dtype, g->K, g->M, 1.f,
                               g->workspace, stream))
        return rc;
    }
  }
  VITK_LAUNCH_CHECK("gemm");
  return 0;
}
