// Bandwidth-bound kernels of the ViT encoder path: RMSNorm/LayerNorm fwd/bwd, row reductions,
// MLP activations (plain and GLU), RoPE on the packed projection, patch unfolding, gradient preparation.
// All of them stream rows with 4-wide vector accesses (16 B fp32 / 8 B bf16 per lane), fp32 math.
#include "common.h"
#include <mutex>
#include <type_traits>
#include <string>

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
int vitk_set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}
extern "C" const char* vitk_last_error(void) { return g_err.c_str(); }
extern "C" int vitk_version(void) { return 100; }

#define DISPATCH_DTYPE(code, T, ...)                \
  if ((code) == VITK_F32) { using T = float; __VA_ARGS__; } \
  else { using T = bf16; __VA_ARGS__; }

// ------------------------------------------------------------------------------------------------
// normalisation forward: one wave per row, rows grid-strided
// ------------------------------------------------------------------------------------------------
template <class TX, class TY, class TW, bool LN, int V = 4>
__global__ __launch_bounds__(256) void norm_fwd_k(const TX* __restrict__ x, const TW* __restrict__ w,
                                                  const TW* __restrict__ b, TY* __restrict__ y,
                                                  float* __restrict__ rstd_out, float* __restrict__ mean_out,
                                                  const int64_t* __restrict__ row_idx, int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const float invD = 1.f / (float)D;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t src = row_idx ? row_idx[r] : r;
    const TX* xr = x + src * D;
    float mean = 0.f;
    if (LN) {
      float s = 0.f;
      for (int d = lane * V; d < D; d += 64 * V) {
        VecN<TX, V> v; v.load(xr + d);
#pragma unroll
        for (int i = 0; i < V; i += 4) s += (v.v[i] + v.v[i + 1]) + (v.v[i + 2] + v.v[i + 3]);
      }
      mean = wave_sum(s) * invD;
    }
    float s2 = 0.f;
    for (int d = lane * V; d < D; d += 64 * V) {
      VecN<TX, V> v; v.load(xr + d);
#pragma unroll
      for (int i = 0; i < V; i += 4) s2 = sumsq4(s2, v.v[i] - mean, v.v[i + 1] - mean, v.v[i + 2] - mean, v.v[i + 3] - mean);
    }
    const float var = wave_sum(s2) * invD;
    const float rstd = rsqrtf(var + eps);
    if (lane == 0) {
      rstd_out[r] = rstd;
      if (LN) mean_out[r] = mean;
    }
    TY* yr = y + r * D;
    for (int d = lane * V; d < D; d += 64 * V) {
      VecN<TX, V> v; v.load(xr + d);
      VecN<TW, V> wv; wv.load(w + d);
      VecN<TY, V> o;
      if (LN && b) {
        VecN<TW, V> bv; bv.load(b + d);
#pragma unroll
        for (int i = 0; i < V; i++) o.v[i] = norm_apply(v.v[i] - mean, rstd, wv.v[i], bv.v[i], true);
      } else {
#pragma unroll
        for (int i = 0; i < V; i++) o.v[i] = norm_apply(v.v[i] - mean, rstd, wv.v[i], 0.f, false);
      }
      o.store(yr + d);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// normalisation backward.  Per row, with g = dy*w, xh = (x-mean)*rstd:
//   RMS: dx = rstd*(g - xh*mean(g*xh))        LN: dx = rstd*(g - mean(g) - xh*mean(g*xh))
// dw = sum_r dy*xh, db = sum_r dy: each wave keeps its share in registers (MAXIT*V floats),
// the 4 waves of a block combine through LDS, one partial row per block goes to the workspace.
// ------------------------------------------------------------------------------------------------
template <class TG, class TX, class TW, class TO, bool LN, int MAXIT, int V = 4>
__global__ __launch_bounds__(256) void norm_bwd_k(const TG* __restrict__ dy, const TX* __restrict__ x,
                                                  const TW* __restrict__ w, const float* __restrict__ rstd_in,
                                                  const float* __restrict__ mean_in, const TO* add, TO* dx,
                                                  const int64_t* __restrict__ row_idx, int64_t rows, int D,
                                                  float* __restrict__ partial /* [grid][2][D] */, int want_dw,
                                                  bf16* __restrict__ shadow /* bf16 copy of the dx rows, or null */) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4][2][D]
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wid;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const float invD = 1.f / (float)D;
  float aw[MAXIT][V], ab[MAXIT][V];
#pragma unroll
  for (int it = 0; it < MAXIT; it++)
#pragma unroll
    for (int i = 0; i < V; i++) { aw[it][i] = 0.f; ab[it][i] = 0.f; }

  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t src = row_idx ? row_idx[r] : r;
    const TX* xr = x + src * D;
    const TG* gr = dy + r * D;
    const float rstd = rstd_in[r];
    const float mean = LN ? mean_in[r] : 0.f;
    float c1 = 0.f, c2 = 0.f;
    float xhs[MAXIT][V], gs[MAXIT][V];   // normalised row and weighted gradient, kept for the second pass
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
      const int d = it * 64 * V + lane * V;
      if (d < D) {
        VecN<TX, V> xv; xv.load(xr + d);
        VecN<TG, V> gv; gv.load(gr + d);
        VecN<TW, V> wv; wv.load(w + d);
#pragma unroll
        for (int i = 0; i < V; i++) {
          const float xh = (xv.v[i] - mean) * rstd;
          const float g = gv.v[i] * wv.v[i];
          xhs[it][i] = xh;
          gs[it][i] = g;
          c1 += g;
          c2 += g * xh;
          aw[it][i] += gv.v[i] * xh;
          ab[it][i] += gv.v[i];
        }
      }
    }
    c2 = wave_sum(c2) * invD;
    c1 = LN ? wave_sum(c1) * invD : 0.f;
    TO* outr = dx + src * D;
    const TO* addr = add ? add + src * D : nullptr;
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
      const int d = it * 64 * V + lane * V;
      if (d < D) {
        VecN<TO, V> o;
        if (addr) o.load(addr + d);
        else {
#pragma unroll
          for (int i = 0; i < V; i++) o.v[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < V; i++) o.v[i] += rstd * (gs[it][i] - c1 - xhs[it][i] * c2);
        o.store(outr + d);
        if (shadow) {   // the next consumer multiplies this gradient in bf16: hand it the cast along with the fp32 rows
          VecN<bf16, V> sv;
#pragma unroll
          for (int i = 0; i < V; i++) sv.v[i] = round_as<TO>(o.v[i]);
          sv.store(shadow + src * D + d);
        }
      }
    }
  }
  if (!want_dw) return;
  float* mine = lds + (size_t)wid * 2 * D;
#pragma unroll
  for (int it = 0; it < MAXIT; it++) {
    const int d = it * 64 * V + lane * V;
    if (d < D) {
#pragma unroll
      for (int i = 0; i < V; i++) { mine[d + i] = aw[it][i]; mine[D + d + i] = ab[it][i]; }
    }
  }
  __syncthreads();
  float* prow = partial + (size_t)blockIdx.x * 2 * D;
  for (int i = threadIdx.x; i < 2 * D; i += 256)
    prow[i] = (lds[i] + lds[2 * D + i]) + (lds[4 * D + i] + lds[6 * D + i]);
}

// ------------------------------------------------------------------------------------------------
// column sums: out[n] = scale * sum_r a[ra(r), n] * (b ? b[r, n] : 1).  Stage 1 gives one partial row
// per block row-chunk; stage 2 (same kernel, fp32 input) folds the partials.
// ------------------------------------------------------------------------------------------------
template <class TA, class TB, class TO>
__global__ __launch_bounds__(256) void colsum_k(const TA* __restrict__ a, int64_t lda, const int64_t* __restrict__ a_idx,
                                                const TB* __restrict__ b, int64_t ldb, TO* __restrict__ out,
                                                int64_t out_ld, int64_t rows, int64_t N, float scale) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int64_t chunk = ceil_div(rows, (int64_t)gridDim.y);
  const int64_t r0 = (int64_t)blockIdx.y * chunk;
  const int64_t r1 = r0 + chunk < rows ? r0 + chunk : rows;
  float s0 = 0.f, s1 = 0.f;
  int64_t r = r0;
  for (; r + 1 < r1; r += 2) {
    const int64_t ra0 = a_idx ? a_idx[r] : r, ra1 = a_idx ? a_idx[r + 1] : r + 1;
    float v0 = to_f<TA>(a[ra0 * lda + n]), v1 = to_f<TA>(a[ra1 * lda + n]);
    if (b) { v0 *= to_f<TB>(b[r * ldb + n]); v1 *= to_f<TB>(b[(r + 1) * ldb + n]); }
    s0 += v0; s1 += v1;
  }
  if (r < r1) {
    const int64_t ra0 = a_idx ? a_idx[r] : r;
    float v0 = to_f<TA>(a[ra0 * lda + n]);
    if (b) v0 *= to_f<TB>(b[r * ldb + n]);
    s0 += v0;
  }
  out[(int64_t)blockIdx.y * out_ld + n] = from_f<TO>((s0 + s1) * scale);
}

// vectorised form (N, lda, ldb multiples of 4): a block owns 256 columns x one row chunk; each lane owns 4 columns,
// the 4 waves take interleaved rows (4 rows in flight each) and combine through LDS.
template <class TA, class TB, class TO>
__global__ __launch_bounds__(256) void colsum4_k(const TA* __restrict__ a, int64_t lda, const int64_t* __restrict__ a_idx,
                                                 const TB* __restrict__ b, int64_t ldb, TO* __restrict__ out,
                                                 int64_t out_ld, int64_t rows, int64_t N, float scale) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 256 + lane * 4;
  const int64_t chunk = ceil_div(rows, (int64_t)gridDim.y);
  const int64_t r0 = (int64_t)blockIdx.y * chunk;
  const int64_t r1 = r0 + chunk < rows ? r0 + chunk : rows;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    for (int64_t r = r0 + wid * 4; r < r1; r += 16) {
      Vec4<TA> va[4];
      Vec4<TB> vb[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (r + u < r1) {
          const int64_t ra = a_idx ? a_idx[r + u] : r + u;
          va[u].load(a + ra * lda + n);
          if (b) vb[u].load(b + (r + u) * ldb + n);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (r + u < r1) {
#pragma unroll
          for (int j = 0; j < 4; j++) acc[j] += b ? va[u].v[j] * vb[u].v[j] : va[u].v[j];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) red[wid][lane * 4 + j] = acc[j];
  __syncthreads();
  const int64_t nn = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (nn < N) {
    const int t = threadIdx.x;
    out[(int64_t)blockIdx.y * out_ld + nn] = from_f<TO>(((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) * scale);
  }
}

static const int COLSUM_CHUNKS = 256;
extern "C" size_t vitk_colsum_workspace_bytes(int64_t N) { return (size_t)COLSUM_CHUNKS * (size_t)N * sizeof(float); }

template <class TA, class TB, class TO>
static void colsum_one(bool vec, dim3 grid, hipStream_t st, const TA* a, int64_t lda, const int64_t* a_idx, const TB* b,
                       int64_t ldb, TO* out, int64_t out_ld, int64_t rows, int64_t N, float scale) {
  if (vec) hipLaunchKernelGGL((colsum4_k<TA, TB, TO>), grid, dim3(256), 0, st, a, lda, a_idx, b, ldb, out, out_ld, rows, N, scale);
  else hipLaunchKernelGGL((colsum_k<TA, TB, TO>), grid, dim3(256), 0, st, a, lda, a_idx, b, ldb, out, out_ld, rows, N, scale);
}

template <class TA, class TB>
static int colsum_launch(const void* a, int64_t lda, const int64_t* a_idx, const void* b, int64_t ldb, void* out,
                         int out_dtype, int64_t rows, int64_t N, float scale, void* ws, hipStream_t st) {
  const bool vec = N % 4 == 0 && lda % 4 == 0 && (!b || ldb % 4 == 0) && ((uintptr_t)a % (4 * sizeof(TA))) == 0 &&
                   (!b || ((uintptr_t)b % (4 * sizeof(TB))) == 0);
  const int64_t colblocks = ceil_div(N, 256);
  int64_t chunks = 2048 / colblocks;               // enough blocks to fill the chip ...
  if (chunks > ceil_div(rows, 64)) chunks = ceil_div(rows, 64);   // ... but at least 64 rows per chunk
  if (chunks > COLSUM_CHUNKS) chunks = COLSUM_CHUNKS;
  if (chunks < 1) chunks = 1;
  dim3 grid((unsigned)colblocks, (unsigned)chunks);
  if (chunks == 1) {
    if (out_dtype == VITK_F32) colsum_one<TA, TB, float>(vec, grid, st, (const TA*)a, lda, a_idx, (const TB*)b, ldb, (float*)out, N, rows, N, scale);
    else colsum_one<TA, TB, bf16>(vec, grid, st, (const TA*)a, lda, a_idx, (const TB*)b, ldb, (bf16*)out, N, rows, N, scale);
    VITK_LAUNCH_CHECK("colsum");
    return 0;
  }
  VITK_REQUIRE(ws != nullptr, "colsum: workspace required");
  colsum_one<TA, TB, float>(vec, grid, st, (const TA*)a, lda, a_idx, (const TB*)b, ldb, (float*)ws, N, rows, N, scale);
  dim3 grid2((unsigned)colblocks, 1);
  const bool vec2 = N % 4 == 0;
  if (out_dtype == VITK_F32) colsum_one<float, float, float>(vec2, grid2, st, (const float*)ws, N, nullptr, (const float*)nullptr, 0, (float*)out, N, chunks, N, 1.f);
  else colsum_one<float, float, bf16>(vec2, grid2, st, (const float*)ws, N, nullptr, (const float*)nullptr, 0, (bf16*)out, N, chunks, N, 1.f);
  VITK_LAUNCH_CHECK("colsum");
  return 0;
}

extern "C" int vitk_colsum(const void* a, int a_dtype, int64_t lda, const int64_t* a_idx, const void* b, int b_dtype,
                           int64_t ldb, void* out, int out_dtype, int64_t rows, int64_t N, float scale, void* workspace,
                           void* stream) {
  VITK_REQUIRE(a && out && rows >= 0 && N > 0, "colsum: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (rows == 0) {
    (void)hipMemsetAsync(out, 0, (size_t)N * (out_dtype == VITK_F32 ? 4 : 2), st);
    return 0;
  }
  DISPATCH_DTYPE(a_dtype, TA, {
    DISPATCH_DTYPE(b ? b_dtype : VITK_F32, TB,
                   { return colsum_launch<TA, TB>(a, lda, a_idx, b, ldb, out, out_dtype, rows, N, scale, workspace, st); });
  });
  return 0;
}

// vitk_colsum for callers that may not allocate: the per-chunk partial sums live in a file-scope device array
// instead of a caller workspace.  Every GPU holds its own copy of the array at its own address; the address is
// looked up once per device (with that device current) and only that pointer is ever handed to kernels.
constexpr int64_t COLSUM_STATIC_MAX_N = 8192;   // 256 chunks x 8192 columns x 4 B = 8 MiB per GPU
static __device__ __align__(256) float scratch[COLSUM_CHUNKS * COLSUM_STATIC_MAX_N];
constexpr int SCRATCH_MAX_DEVICES = 64;
static std::mutex g_scratch_mu;
static float* g_scratch_ptr[SCRATCH_MAX_DEVICES];   // per device: device address of `scratch`, or null before set-up

static int scratch_for_current_device(float** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SCRATCH_MAX_DEVICES)
    return vitk_set_error("colsum_static: cannot identify the current device (index %d)", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    float* p = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) return vitk_set_error("colsum_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}

extern "C" int vitk_colsum_static(const void* a, int a_dtype, int64_t lda, const int64_t* a_idx, const void* b, int b_dtype,
                                  int64_t ldb, void* out, int out_dtype, int64_t rows, int64_t N, float scale, void* stream) {
  VITK_REQUIRE(a && out && rows >= 0 && N > 0, "colsum_static: bad arguments");
  VITK_REQUIRE(N <= COLSUM_STATIC_MAX_N, "colsum_static: N=%lld exceeds the built-in scratch (at most %lld columns); use vitk_colsum "
               "with a caller workspace", (long long)N, (long long)COLSUM_STATIC_MAX_N);
  // The scratch used is the CURRENT device's copy, so the current device has to be the one that owns the operands (and
  // the stream): a kernel on another GPU would write through an address that means nothing there.
  {
    int cur = -1;
    hipPointerAttribute_t at;
    VITK_REQUIRE(hipGetDevice(&cur) == hipSuccess, "colsum_static: cannot identify the current device");
    if (hipPointerGetAttributes(&at, out) == hipSuccess) {
      VITK_REQUIRE(at.device == cur, "colsum_static: `out` lives on device %d but device %d is current; make the operands' "
                   "device current before the call", at.device, cur);
    } else {
      (void)hipGetLastError();   // not a pointer the runtime knows: the launch itself will report it
    }
  }
  float* ws = nullptr;
  if (int rc = scratch_for_current_device(&ws)) return rc;
  // every chunk row that the second pass reads is written whole by the first pass: nothing has to start at zero
  return vitk_colsum(a, a_dtype, lda, a_idx, b, b_dtype, ldb, out, out_dtype, rows, N, scale, ws, stream);
}

// ------------------------------------------------------------------------------------------------
// norm entry points
// ------------------------------------------------------------------------------------------------
static bool wide_ok(const void* p) { return ((uintptr_t)p & 15) == 0; }   // null or 16-byte aligned
static int norm_grid(int64_t rows) {
  int64_t g = ceil_div(rows, 4);
  return (int)(g < 2048 ? (g < 1 ? 1 : g) : 2048);
}

template <class TX, class TY, class TW>
static void norm_fwd_launch(const void* x, const void* w, const void* b, void* y, float* rstd, float* mean,
                            const int64_t* idx, int64_t rows, int D, float eps, int ln, hipStream_t st) {
  dim3 grid(norm_grid(rows)), block(256);
  if constexpr (std::is_same<TX, bf16>::value && std::is_same<TY, bf16>::value && std::is_same<TW, bf16>::value) {
    // an all-bf16 stream: 8 elements (16 bytes) per lane.  Rows shorter than one pass of the wave keep the 4-wide order
    // of summation, which the fused per-head normalisation reproduces bit for bit.
    if (D % 8 == 0 && D >= 512 && wide_ok(x) && wide_ok(y) && wide_ok(w) && wide_ok(b)) {
      if (ln)
        hipLaunchKernelGGL((norm_fwd_k<bf16, bf16, bf16, true, 8>), grid, block, 0, st, (const bf16*)x, (const bf16*)w,
                           (const bf16*)b, (bf16*)y, rstd, mean, idx, rows, D, eps);
      else
        hipLaunchKernelGGL((norm_fwd_k<bf16, bf16, bf16, false, 8>), grid, block, 0, st, (const bf16*)x, (const bf16*)w,
                           (const bf16*)b, (bf16*)y, rstd, mean, idx, rows, D, eps);
      return;
    }
  }
  if (ln)
    hipLaunchKernelGGL((norm_fwd_k<TX, TY, TW, true>), grid, block, 0, st, (const TX*)x, (const TW*)w, (const TW*)b,
                       (TY*)y, rstd, mean, idx, rows, D, eps);
  else
    hipLaunchKernelGGL((norm_fwd_k<TX, TY, TW, false>), grid, block, 0, st, (const TX*)x, (const TW*)w, (const TW*)b,
                       (TY*)y, rstd, mean, idx, rows, D, eps);
}

extern "C" int vitk_norm_fwd(const void* x, int x_dtype, const void* w, const void* b, int w_dtype, void* y, int y_dtype,
                             float* rstd, float* mean, const int64_t* row_idx, int64_t rows, int64_t D, float eps,
                             int is_layernorm, void* stream) {
  VITK_REQUIRE(x && w && y && rstd, "norm_fwd: null pointer");
  VITK_REQUIRE(!is_layernorm || mean, "norm_fwd: LayerNorm needs a mean buffer");
  VITK_REQUIRE(D > 0 && D % 4 == 0 && D <= (1 << 20), "norm_fwd: D=%lld must be a positive multiple of 4", (long long)D);
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DTYPE(x_dtype, TX, {
    DISPATCH_DTYPE(y_dtype, TY, {
      DISPATCH_DTYPE(w_dtype, TW,
                     { norm_fwd_launch<TX, TY, TW>(x, w, b, y, rstd, mean, row_idx, rows, (int)D, eps, is_layernorm, st); });
    });
  });
  VITK_LAUNCH_CHECK("norm_fwd");
  return 0;
}

static const int NORM_BWD_BLOCKS = 1024;  // 4 blocks per CU: measured best of 512..2048 (216 -> 178 us at M=69632, D=768)
static const int NORM_BWD_FOLD = 16;   // the per-block partials are folded in two steps: 1024 -> 16 -> 1 rows
extern "C" size_t vitk_norm_bwd_workspace_bytes(int64_t D) {
  return (size_t)(NORM_BWD_BLOCKS + NORM_BWD_FOLD) * 2 * (size_t)D * sizeof(float);
}

// fold the per-block partials [grid][2][D] (dw = first half, db = second half) of a norm backward in two steps, so that
// the first one has enough blocks in flight: [grid] -> [NORM_BWD_FOLD] rows (behind the partials in the workspace) -> 1 row
static void fold_rows(const float* ws, int grid, int64_t D, int w_dtype, void* dw, void* db, float* ws2, hipStream_t st) {
  const int fold = grid > 4 * NORM_BWD_FOLD ? NORM_BWD_FOLD : 1;
  for (int which = 0; which < 2; which++) {
    void* dst = which == 0 ? dw : db;
    if (!dst) continue;
    const float* src = ws + (size_t)which * D;
    int64_t src_ld = 2 * D, src_rows = grid;
    if (fold > 1) {
      dim3 g1((unsigned)ceil_div(D, 256), (unsigned)fold);
      colsum_one<float, float, float>(true, g1, st, src, src_ld, nullptr, (const float*)nullptr, 0, ws2 + (size_t)which * D,
                                      2 * D, src_rows, D, 1.f);
      src = ws2 + (size_t)which * D;
      src_rows = fold;
    }
    dim3 g2((unsigned)ceil_div(D, 256), 1);
    if (w_dtype == VITK_F32)
      colsum_one<float, float, float>(true, g2, st, src, src_ld, nullptr, (const float*)nullptr, 0, (float*)dst, D, src_rows, D, 1.f);
    else
      colsum_one<float, float, bf16>(true, g2, st, src, src_ld, nullptr, (const float*)nullptr, 0, (bf16*)dst, D, src_rows, D, 1.f);
  }
}
static void fold_norm_partials(float* ws, int grid, int64_t D, int w_dtype, void* dw, void* db, hipStream_t st) {
  fold_rows(ws, grid, D, w_dtype, dw, db, ws + (size_t)NORM_BWD_BLOCKS * 2 * D, st);
}

template <class TG, class TX, class TW, class TO, bool LN>
static int norm_bwd_launch(const void* dy, const void* x, const void* w, const float* rstd, const float* mean,
                           const void* add, void* dx, const int64_t* idx, int64_t rows, int D, float* ws, int want_dw,
                           int grid, hipStream_t st, void* shadow) {
  const size_t lds = want_dw ? (size_t)8 * D * sizeof(float) : 0;
#define NB_LAUNCH(MAXIT)                                                                                            \
  hipLaunchKernelGGL((norm_bwd_k<TG, TX, TW, TO, LN, MAXIT>), dim3(grid), dim3(256), lds, st, (const TG*)dy,        \
                     (const TX*)x, (const TW*)w, rstd, mean, (const TO*)add, (TO*)dx, idx, rows, D, ws, want_dw,    \
                     (bf16*)shadow)
  if constexpr (std::is_same<TG, bf16>::value && std::is_same<TX, bf16>::value && std::is_same<TW, bf16>::value &&
                std::is_same<TO, bf16>::value) {
    if (D % 8 == 0 && D >= 512 && wide_ok(dy) && wide_ok(x) && wide_ok(w) && wide_ok(add) && wide_ok(dx)) {   // see norm_fwd_launch
#define NB_LAUNCH8(MAXIT)                                                                                              \
  hipLaunchKernelGGL((norm_bwd_k<bf16, bf16, bf16, bf16, LN, MAXIT, 8>), dim3(grid), dim3(256), lds, st, (const bf16*)dy, \
                     (const bf16*)x, (const bf16*)w, rstd, mean, (const bf16*)add, (bf16*)dx, idx, rows, D, ws, want_dw,  \
                     (bf16*)shadow)
      if (D <= 512) NB_LAUNCH8(1);
      else if (D <= 1024) NB_LAUNCH8(2);
      else if (D <= 2048) NB_LAUNCH8(4);
      else return vitk_set_error("norm_bwd: D=%d > 2048 not supported", D);
#undef NB_LAUNCH8
      return 0;
    }
  }
  if (D <= 512) NB_LAUNCH(2);
  else if (D <= 1024) NB_LAUNCH(4);
  else if (D <= 2048) NB_LAUNCH(8);
  else return vitk_set_error("norm_bwd: D=%d > 2048 not supported", D);
#undef NB_LAUNCH
  return 0;
}

extern "C" int vitk_norm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const void* w, int w_dtype,
                             const float* rstd, const float* mean, const void* add, void* dx, int dx_dtype, void* dw,
                             void* db, const int64_t* row_idx, int64_t rows, int64_t D, int is_layernorm, void* workspace,
                             void* dx_bf16, void* stream) {
  VITK_REQUIRE(dy && x && w && rstd && dx, "norm_bwd: null pointer");
  VITK_REQUIRE(!is_layernorm || mean, "norm_bwd: LayerNorm needs the saved mean");
  VITK_REQUIRE(D > 0 && D % 4 == 0, "norm_bwd: D=%lld must be a positive multiple of 4", (long long)D);
  hipStream_t st = (hipStream_t)stream;
  const int want_dw = dw != nullptr;
  VITK_REQUIRE(!want_dw || workspace, "norm_bwd: workspace required for dw");
  const size_t wbytes = (size_t)D * (w_dtype == VITK_F32 ? 4 : 2);
  if (rows == 0) {
    if (dw) (void)hipMemsetAsync(dw, 0, wbytes, st);
    if (db) (void)hipMemsetAsync(db, 0, wbytes, st);
    return 0;
  }
  int grid = norm_grid(rows);
  if (grid > NORM_BWD_BLOCKS) grid = NORM_BWD_BLOCKS;
  int rc = 0;
  DISPATCH_DTYPE(dy_dtype, TG, {
    DISPATCH_DTYPE(x_dtype, TX, {
      DISPATCH_DTYPE(w_dtype, TW, {
        DISPATCH_DTYPE(dx_dtype, TO, {
          if (is_layernorm)
            rc = norm_bwd_launch<TG, TX, TW, TO, true>(dy, x, w, rstd, mean, add, dx, row_idx, rows, (int)D,
                                                       (float*)workspace, want_dw, grid, st, dx_bf16);
          else
            rc = norm_bwd_launch<TG, TX, TW, TO, false>(dy, x, w, rstd, mean, add, dx, row_idx, rows, (int)D,
                                                        (float*)workspace, want_dw, grid, st, dx_bf16);
        });
      });
    });
  });
  if (rc) return rc;
  VITK_LAUNCH_CHECK("norm_bwd");
  if (want_dw) {
    fold_norm_partials((float*)workspace, grid, D, w_dtype, dw, db, st);
    VITK_LAUNCH_CHECK("norm_bwd reduce");
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// MLP activation forward/backward (plain and GLU).  In bf16 every eager op rounds its result, so
// act(glu) and the product are rounded separately to stay close to the eager arithmetic.
// ------------------------------------------------------------------------------------------------
template <class T, bool GLU>
__global__ __launch_bounds__(256) void act_fwd_k(const T* __restrict__ p, T* __restrict__ h, int64_t rows, int64_t F,
                                                 int act, int use_limit, float limit, int use_extra, float extra,
                                                 uint32_t thr, float keep_scale, uint64_t seed) {
  const int64_t nvec = rows * (F / 4);
  const int64_t ldp = GLU ? 2 * F : F;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / (F / 4);
    const int64_t c = (i % (F / 4)) * 4;
    Vec4<T> a, o;
    if (GLU) {
      Vec4<T> lin;
      lin.load(p + r * ldp + c);
      a.load(p + r * ldp + F + c);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float l = lin.v[j], g = a.v[j];
        if (use_limit) { l = fminf(fmaxf(l, -limit), limit); g = fminf(g, limit); }
        if (use_extra) l = round_as<T>(l + extra);
        o.v[j] = round_as<T>(act_apply(act, g)) * l;
      }
    } else {
      a.load(p + r * ldp + c);
#pragma unroll
      for (int j = 0; j < 4; j++) o.v[j] = act_apply(act, a.v[j]);
    }
    if (thr) {
      uint32_t dr[4];
      rng_run<4>(seed, (uint64_t)(r * F + c), dr);   // c and the row length are multiples of 4
#pragma unroll
      for (int j = 0; j < 4; j++) o.v[j] = dr[j] >= thr ? round_as<T>(o.v[j]) * keep_scale : 0.f;
    }
    o.store(h + r * F + c);
  }
}

template <class T, bool GLU>
__global__ __launch_bounds__(256) void act_bwd_k(const T* __restrict__ dh, const T* __restrict__ p, T* __restrict__ dp,
                                                 int64_t rows, int64_t F, int act, int use_limit, float limit,
                                                 int use_extra, float extra, uint32_t thr, float keep_scale,
                                                 uint64_t seed) {
  const int64_t nvec = rows * (F / 4);
  const int64_t ldp = GLU ? 2 * F : F;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / (F / 4);
    const int64_t c = (i % (F / 4)) * 4;
    Vec4<T> g, a;
    g.load(dh + r * F + c);
    if (thr) {
      uint32_t dr[4];
      rng_run<4>(seed, (uint64_t)(r * F + c), dr);   // c and the row length are multiples of 4
#pragma unroll
      for (int j = 0; j < 4; j++) g.v[j] = dr[j] >= thr ? g.v[j] * keep_scale : 0.f;
    }
    if (GLU) {
      Vec4<T> lin, dl, dg;
      lin.load(p + r * ldp + c);
      a.load(p + r * ldp + F + c);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float l = lin.v[j], gl = a.v[j];
        float ml = 1.f, mg = 1.f;  // clamp masks: gradient passes where the clamp is inactive
        if (use_limit) {
          ml = (l >= -limit && l <= limit) ? 1.f : 0.f;
          mg = (gl <= limit) ? 1.f : 0.f;
          l = fminf(fmaxf(l, -limit), limit);
          gl = fminf(gl, limit);
        }
        if (use_extra) l = round_as<T>(l + extra);
        dl.v[j] = g.v[j] * round_as<T>(act_apply(act, gl)) * ml;
        dg.v[j] = g.v[j] * l * act_grad(act, gl) * mg;
      }
      dl.store(dp + r * ldp + c);
      dg.store(dp + r * ldp + F + c);
    } else {
      a.load(p + r * ldp + c);
      Vec4<T> o;
#pragma unroll
      for (int j = 0; j < 4; j++) o.v[j] = g.v[j] * act_grad(act, a.v[j]);
      o.store(dp + r * ldp + c);
    }
  }
}

static int ew_grid(int64_t nvec) {
  int64_t g = ceil_div(nvec, 256);
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

extern "C" int vitk_act_fwd(const void* p, void* h, int dtype, int64_t rows, int64_t F, int is_glu, int act, int use_limit,
                            float limit, int use_extra, float extra_bias, float drop_p, uint64_t seed, void* stream) {
  VITK_REQUIRE(p && h, "act_fwd: null pointer");
  VITK_REQUIRE(F > 0 && F % 4 == 0, "act_fwd: F=%lld must be a multiple of 4", (long long)F);
  VITK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "act_fwd: dropout p=%f out of range", drop_p);
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t thr = drop_p > 0.f ? drop_threshold(drop_p) : 0;
  const float ks = 1.f / (1.f - drop_p);
  dim3 grid(ew_grid(rows * (F / 4))), block(256);
  DISPATCH_DTYPE(dtype, T, {
    if (is_glu)
      hipLaunchKernelGGL((act_fwd_k<T, true>), grid, block, 0, st, (const T*)p, (T*)h, rows, F, act, use_limit, limit,
                         use_extra, extra_bias, thr, ks, seed);
    else
      hipLaunchKernelGGL((act_fwd_k<T, false>), grid, block, 0, st, (const T*)p, (T*)h, rows, F, act, use_limit, limit,
                         use_extra, extra_bias, thr, ks, seed);
  });
  VITK_LAUNCH_CHECK("act_fwd");
  return 0;
}

extern "C" int vitk_act_bwd(const void* dh, const void* p, void* dp, int dtype, int64_t rows, int64_t F, int is_glu,
                            int act, int use_limit, float limit, int use_extra, float extra_bias, float drop_p,
                            uint64_t seed, void* stream) {
  VITK_REQUIRE(dh && p && dp, "act_bwd: null pointer");
  VITK_REQUIRE(F > 0 && F % 4 == 0, "act_bwd: F=%lld must be a multiple of 4", (long long)F);
  VITK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "act_bwd: dropout p=%f out of range", drop_p);
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t thr = drop_p > 0.f ? drop_threshold(drop_p) : 0;
  const float ks = 1.f / (1.f - drop_p);
  dim3 grid(ew_grid(rows * (F / 4))), block(256);
  DISPATCH_DTYPE(dtype, T, {
    if (is_glu)
      hipLaunchKernelGGL((act_bwd_k<T, true>), grid, block, 0, st, (const T*)dh, (const T*)p, (T*)dp, rows, F, act,
                         use_limit, limit, use_extra, extra_bias, thr, ks, seed);
    else
      hipLaunchKernelGGL((act_bwd_k<T, false>), grid, block, 0, st, (const T*)dh, (const T*)p, (T*)dp, rows, F, act,
                         use_limit, limit, use_extra, extra_bias, thr, ks, seed);
  });
  VITK_LAUNCH_CHECK("act_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// RoPE in place on the q and k thirds of qkv[B*S, 3*H*Dh].  One thread per (row, q|k, head, pair of
// 2 adjacent d in the first half): reads x1 = x[d..d+1], x2 = x[d+half..], writes both.
//   forward:  o1 = x1*cos1 - x2*sin1 ; o2 = x2*cos2 + x1*sin2
//   inverse:  g1 = y1*cos1 + y2*sin2 ; g2 = y2*cos2 - y1*sin1     (transpose of the forward map)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void rope_rows_k(T* __restrict__ x, const float* __restrict__ table, int64_t B,
                                                   int64_t S, int64_t H, int64_t Dh, int64_t prefix, int64_t row_stride,
                                                   int64_t nsec, int inverse, int emu, int per_sample) {
  const int64_t half = Dh / 2;
  const int64_t np = half / 2;          // pairs per head
  const int64_t N = S - prefix;         // rotated rows per sample
  const int64_t total = B * N * nsec * H * np;
  const int64_t tbs = per_sample ? N * Dh : 0;   // table stride between samples
  const float* sinT = table;
  const float* cosT = table + (per_sample ? B : 1) * N * Dh;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i;
    const int64_t pr = t % np; t /= np;
    const int64_t h = t % H; t /= H;
    const int64_t which = t % nsec; t /= nsec;
    const int64_t n = t % N;
    const int64_t b = t / N;
    const int64_t d = pr * 2;
    T* px = x + (b * S + prefix + n) * row_stride + which * (H * Dh) + h * Dh;
    const float* ps = sinT + b * tbs + n * Dh;
    const float* pc = cosT + b * tbs + n * Dh;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const float x1 = to_f<T>(px[d + j]), x2 = to_f<T>(px[d + j + half]);
      const float s1 = ps[d + j], s2 = ps[d + j + half], c1 = pc[d + j], c2 = pc[d + j + half];
      float o1, o2;
      if (!inverse) { o1 = rope_mix(x1, c1, -x2, s1, emu); o2 = rope_mix(x2, c2, x1, s2, emu); }
      else { o1 = rope_mix(x1, c1, x2, s2, emu); o2 = rope_mix(x2, c2, -x1, s1, emu); }
      px[d + j] = from_f<T>(o1);
      px[d + j + half] = from_f<T>(o2);
    }
  }
}

// vector form: a thread rotates V consecutive pairs (d .. d+V-1 with d+half .. d+half+V-1); V = 4 for head_dim % 8 == 0,
// V = 8 (16-byte accesses, bf16) for head_dim % 16 == 0
template <class T, int V = 4>
__global__ __launch_bounds__(256) void rope_rows4_k(T* __restrict__ x, const float* __restrict__ table, int64_t B,
                                                    int64_t S, int64_t H, int64_t Dh, int64_t prefix, int64_t row_stride,
                                                    int64_t nsec, int inverse, int emu, int per_sample) {
  const int64_t half = Dh / 2;
  const int64_t nc = half / V;          // chunks of V pairs per head
  const int64_t N = S - prefix;
  const int64_t total = B * N * nsec * H * nc;
  const int64_t tbs = per_sample ? N * Dh : 0;
  const float* sinT = table;
  const float* cosT = table + (per_sample ? B : 1) * N * Dh;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i;
    const int64_t c = t % nc; t /= nc;
    const int64_t h = t % H; t /= H;
    const int64_t which = t % nsec; t /= nsec;
    const int64_t n = t % N;
    const int64_t b = t / N;
    const int64_t d = c * V;
    T* px = x + (b * S + prefix + n) * row_stride + which * (H * Dh) + h * Dh;
    const float* ps = sinT + b * tbs + n * Dh;
    const float* pc = cosT + b * tbs + n * Dh;
    VecN<T, V> x1, x2;
    x1.load(px + d);
    x2.load(px + d + half);
    float s1v[V], s2v[V], c1v[V], c2v[V];
#pragma unroll
    for (int q = 0; q < V; q += 4) {
      *(float4*)(s1v + q) = *(const float4*)(ps + d + q);
      *(float4*)(s2v + q) = *(const float4*)(ps + d + half + q);
      *(float4*)(c1v + q) = *(const float4*)(pc + d + q);
      *(float4*)(c2v + q) = *(const float4*)(pc + d + half + q);
    }
    VecN<T, V> o1, o2;
#pragma unroll
    for (int j = 0; j < V; j++) {
      const float a = x1.v[j], bb = x2.v[j];
      if (!inverse) { o1.v[j] = rope_mix(a, c1v[j], -bb, s1v[j], emu); o2.v[j] = rope_mix(bb, c2v[j], a, s2v[j], emu); }
      else { o1.v[j] = rope_mix(a, c1v[j], bb, s2v[j], emu); o2.v[j] = rope_mix(bb, c2v[j], -a, s1v[j], emu); }
    }
    o1.store(px + d);
    o2.store(px + d + half);
  }
}

extern "C" int vitk_rope_rows(void* x, int dtype, const float* table, int64_t B, int64_t S, int64_t H, int64_t Dh,
                              int64_t prefix, int64_t row_stride, int64_t nsec, int inverse, int emulate_bf16,
                              int per_sample, void* stream) {
  VITK_REQUIRE(x && table, "rope: null pointer");
  VITK_REQUIRE(Dh > 0 && Dh % 4 == 0, "rope: head_dim=%lld must be a multiple of 4", (long long)Dh);
  VITK_REQUIRE(prefix >= 0 && prefix <= S, "rope: prefix=%lld outside [0, %lld]", (long long)prefix, (long long)S);
  VITK_REQUIRE(nsec >= 1 && row_stride >= nsec * H * Dh, "rope: row_stride=%lld cannot hold %lld sections of %lld",
               (long long)row_stride, (long long)nsec, (long long)(H * Dh));
  hipStream_t st = (hipStream_t)stream;
  // 8- or 16-byte accesses when the halves of a head are 4-element aligned (rows and sections then are, too)
  const bool vec = Dh % 8 == 0 && row_stride % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)table % 16) == 0;
  const bool vec8 = vec && dtype == VITK_BF16 && Dh % 16 == 0 && row_stride % 8 == 0;
  const int64_t total = B * (S - prefix) * nsec * H * (vec8 ? Dh / 16 : vec ? Dh / 8 : Dh / 4);
  if (total == 0) return 0;
  dim3 grid(ew_grid(total)), block(256);
  if (vec8) {
    hipLaunchKernelGGL((rope_rows4_k<bf16, 8>), grid, block, 0, st, (bf16*)x, table, B, S, H, Dh, prefix, row_stride, nsec,
                       inverse, emulate_bf16, per_sample);
    VITK_LAUNCH_CHECK("rope");
    return 0;
  }
  DISPATCH_DTYPE(dtype, T, {
    if (vec)
      hipLaunchKernelGGL((rope_rows4_k<T>), grid, block, 0, st, (T*)x, table, B, S, H, Dh, prefix, row_stride, nsec, inverse,
                         emulate_bf16, per_sample);
    else
      hipLaunchKernelGGL((rope_rows_k<T>), grid, block, 0, st, (T*)x, table, B, S, H, Dh, prefix, row_stride, nsec, inverse,
                         emulate_bf16, per_sample);
  });
  VITK_LAUNCH_CHECK("rope");
  return 0;
}

extern "C" int vitk_rope_qk(void* qkv, int dtype, const float* table, int64_t B, int64_t S, int64_t H, int64_t Dh,
                            int64_t prefix, int inverse, int emulate_bf16, int per_sample, void* stream) {
  return vitk_rope_rows(qkv, dtype, table, B, S, H, Dh, prefix, 3 * H * Dh, 2, inverse, emulate_bf16, per_sample, stream);
}

// ------------------------------------------------------------------------------------------------
// Backward of the per-head q / k normalisation (vit/attention.py:25-41) in place on the packed gradient [M, 3, H, Dh]:
// sections 0 (q) and 1 (k) of every row hold H head rows of Dh values each; the gradient with respect to the normalised
// head is replaced by the gradient with respect to the un-normalised one, and the weight / bias gradients are summed.
// 16 lanes own a head row (8 values per lane, lanes past Dh / 8 idle), so a wave handles four head rows at a time and no
// head is ever copied out into a matrix of its own.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sum16(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <class T, class TW, bool LN>
__global__ __launch_bounds__(256) void qk_norm_bwd_k(T* dqkv, const T* __restrict__ pre_q, const T* __restrict__ pre_k,
                                                     const float* __restrict__ rstd_q, const float* __restrict__ rstd_k,
                                                     const float* __restrict__ mean_q, const float* __restrict__ mean_k,
                                                     const TW* __restrict__ w_q, const TW* __restrict__ w_k, int64_t M, int H,
                                                     int Dh, int64_t ld, float* __restrict__ partial /* [2][grid][2][Dh] */) {
  __shared__ float red[16][2][128];   // one slot per 16-lane group of the workgroup
  const int which = blockIdx.y;       // 0: q, 1: k
  const T* pre = which ? pre_k : pre_q;
  const float* rstd_in = which ? rstd_k : rstd_q;
  const float* mean_in = which ? mean_k : mean_q;
  const TW* w = which ? w_k : w_q;
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int d = sub * 8;
  const bool on = d < Dh;
  const float invD = 1.f / (float)Dh;
  float wv[8], aw[8], ab[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { wv[i] = on ? to_f<TW>(w[d + i]) : 0.f; aw[i] = 0.f; ab[i] = 0.f; }
  const int64_t rows = M * H;
  for (int64_t r0 = (int64_t)blockIdx.x * 16; r0 < rows; r0 += (int64_t)gridDim.x * 16) {
    const int64_t r = r0 + grp;
    const bool live = on && r < rows;
    const int64_t m = r / H;
    const int h = (int)(r - m * H);
    T* g = dqkv + m * ld + (int64_t)which * H * Dh + (int64_t)h * Dh + d;
    float gv[8], xh[8];
    float c1 = 0.f, c2 = 0.f, rstd = 0.f;
    if (live) {
      Vec4<T> g0, g1, x0, x1;
      g0.load(g); g1.load(g + 4);
      x0.load(pre + r * Dh + d); x1.load(pre + r * Dh + d + 4);
      rstd = rstd_in[r];
      const float mean = LN ? mean_in[r] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float gi = i < 4 ? g0.v[i] : g1.v[i - 4], xi = i < 4 ? x0.v[i] : x1.v[i - 4];
        xh[i] = (xi - mean) * rstd;
        gv[i] = gi * wv[i];
        c1 += gv[i];
        c2 += gv[i] * xh[i];
        aw[i] += gi * xh[i];
        ab[i] += gi;
      }
    }
    c2 = sum16(c2) * invD;
    c1 = LN ? sum16(c1) * invD : 0.f;
    if (live) {
      Vec4<T> o0, o1;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float o = rstd * (gv[i] - c1 - xh[i] * c2);
        if (i < 4) o0.v[i] = o; else o1.v[i - 4] = o;
      }
      o0.store(g); o1.store(g + 4);
    }
  }
  if (!partial) return;
#pragma unroll
  for (int i = 0; i < 8; i++) { red[grp][0][d + i] = aw[i]; red[grp][1][d + i] = ab[i]; }
  __syncthreads();
  float* prow = partial + ((size_t)which * gridDim.x + blockIdx.x) * 2 * Dh;
  for (int i = threadIdx.x; i < 2 * Dh; i += 256) {
    const int half = i / Dh, c = i - half * Dh;
    float sacc = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 16; g2++) sacc += red[g2][half][c];
    prow[i] = sacc;
  }
}

extern "C" size_t vitk_qk_norm_bwd_workspace_bytes(int64_t Dh) { return 2 * vitk_norm_bwd_workspace_bytes(Dh); }

extern "C" int vitk_qk_norm_bwd(void* dqkv, int dtype, const void* pre_q, const void* pre_k, const float* rstd_q,
                                const float* rstd_k, const float* mean_q, const float* mean_k, const void* w_q,
                                const void* w_k, int w_dtype, void* dw_q, void* db_q, void* dw_k, void* db_k, int64_t M,
                                int64_t H, int64_t Dh, int64_t ld, int is_layernorm, void* workspace, void* stream) {
  VITK_REQUIRE(dqkv && pre_q && pre_k && rstd_q && rstd_k && w_q && w_k, "qk_norm_bwd: null pointer");
  VITK_REQUIRE(!is_layernorm || (mean_q && mean_k), "qk_norm_bwd: LayerNorm needs the saved means");
  VITK_REQUIRE(M >= 0 && H > 0 && Dh > 0 && Dh % 8 == 0 && Dh <= 128, "qk_norm_bwd: head_dim=%lld must be a multiple of 8, at most 128",
               (long long)Dh);
  VITK_REQUIRE(ld >= 2 * H * Dh && ld % 4 == 0, "qk_norm_bwd: ld=%lld too small for two sections of %lld heads", (long long)ld, (long long)H);
  VITK_REQUIRE(dtype == VITK_F32 || dtype == VITK_BF16, "qk_norm_bwd: bad dtype");
  const bool want = dw_q || db_q || dw_k || db_k;
  VITK_REQUIRE(!want || workspace, "qk_norm_bwd: workspace required for the weight gradients");
  hipStream_t st = (hipStream_t)stream;
  const size_t wbytes = (size_t)Dh * (w_dtype == VITK_F32 ? 4 : 2);
  if (M == 0) {
    for (void* p : {dw_q, db_q, dw_k, db_k})
      if (p) (void)hipMemsetAsync(p, 0, wbytes, st);
    return 0;
  }
  int grid = (int)std::min<int64_t>(ceil_div(M * H, 16), NORM_BWD_BLOCKS);
  float* ws = (float*)workspace;
  float* part = want ? ws : nullptr;
  DISPATCH_DTYPE(dtype, T, {
    DISPATCH_DTYPE(w_dtype, TW, {
      if (is_layernorm)
        hipLaunchKernelGGL((qk_norm_bwd_k<T, TW, true>), dim3(grid, 2), dim3(256), 0, st, (T*)dqkv, (const T*)pre_q, (const T*)pre_k,
                           rstd_q, rstd_k, mean_q, mean_k, (const TW*)w_q, (const TW*)w_k, M, (int)H, (int)Dh, ld, part);
      else
        hipLaunchKernelGGL((qk_norm_bwd_k<T, TW, false>), dim3(grid, 2), dim3(256), 0, st, (T*)dqkv, (const T*)pre_q, (const T*)pre_k,
                           rstd_q, rstd_k, mean_q, mean_k, (const TW*)w_q, (const TW*)w_k, M, (int)H, (int)Dh, ld, part);
    });
  });
  VITK_LAUNCH_CHECK("qk_norm_bwd");
  if (want) {
    // section 1's partials follow section 0's grid rows directly; fold each with its own scratch rows
    float* s0 = ws;
    float* s1 = ws + (size_t)grid * 2 * Dh;
    float* scratch = ws + (size_t)2 * NORM_BWD_BLOCKS * 2 * Dh;   // two folds' worth of intermediate rows, behind both sections' partials
    fold_rows(s0, grid, Dh, w_dtype, dw_q, db_q, scratch, st);
    fold_rows(s1, grid, Dh, w_dtype, dw_k, db_k, scratch + (size_t)NORM_BWD_FOLD * 2 * Dh, st);
    VITK_LAUNCH_CHECK("qk_norm_bwd reduce");
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// patch unfolding / folding (stride == kernel, so it is a permutation plus zero padding)
// ------------------------------------------------------------------------------------------------
template <class TX, class TC, bool FOLD>
__global__ __launch_bounds__(256) void im2col_k(const TX* x_in, TX* x_out, const TC* c_in, TC* c_out, int64_t B, int64_t C,
                                                int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t ldc) {
  const int64_t gh = H / ph, gw = W / pw, K = C * ph * pw;
  const int64_t total = B * gh * gw * ldc;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t k = i % ldc;
    const int64_t m = i / ldc;
    if (k >= K) {
      if (!FOLD) c_out[i] = from_f<TC>(0.f);
      continue;
    }
    const int64_t v = k % pw, u = (k / pw) % ph, c = k / (pw * ph);
    const int64_t j = m % gw, ii = (m / gw) % gh, b = m / (gw * gh);
    const int64_t xi = ((b * C + c) * H + ii * ph + u) * W + j * pw + v;
    if (FOLD) x_out[xi] = from_f<TX>(to_f<TC>(c_in[i]));
    else c_out[i] = from_f<TC>(to_f<TX>(x_in[xi]));
  }
}

// Unfolding one band of patches (one image, one row of the patch grid) per workgroup: the band's ph image rows of every
// channel are read along W (whole cache lines), converted and laid out in LDS as the band's gw output rows, which are one
// contiguous range of the output and leave with 16-byte stores.  The element-indexed kernel above reads pw-element runs
// (56 bytes at pw = 14) and fetched 3x the image from memory.
template <class TX, class TC>
__global__ __launch_bounds__(256) void im2col_band_k(const TX* __restrict__ x, TC* __restrict__ cols, int64_t C, int64_t H,
                                                     int64_t W, int ph, int pw, int ldc, int gh, int gw) {
  extern __shared__ __attribute__((aligned(16))) char band_smem[];
  TC* band = (TC*)band_smem;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x / gh;
  const int ii = (int)(blockIdx.x % gh);
  const int K = (int)C * ph * pw, wc = gw * pw, n = gw * ldc;
  if (ldc > K)
    for (int i = tid; i < gw * (ldc - K); i += 256) band[(i / (ldc - K)) * ldc + K + i % (ldc - K)] = from_f<TC>(0.f);
  const int per_c = ph * wc;
  for (int i = tid; i < (int)C * per_c; i += 256) {
    const int c = i / per_c, r = i - c * per_c, u = r / wc, xx = r - u * wc;
    const int j = xx / pw, v = xx - j * pw;
    band[j * ldc + (c * ph + u) * pw + v] = from_f<TC>(to_f<TX>(x[((b * C + c) * H + (int64_t)ii * ph + u) * W + xx]));
  }
  __syncthreads();
  TC* dst = cols + (int64_t)blockIdx.x * n;
  constexpr int V = 16 / sizeof(TC);
  if ((((uintptr_t)dst) & 15) == 0 && n % V == 0) {
    for (int i = tid; i < n / V; i += 256) ((uint4*)dst)[i] = ((const uint4*)band)[i];
  } else {
    for (int i = tid; i < n; i += 256) dst[i] = band[i];
  }
}

// rows/columns of the image that no patch covers (H % ph, W % pw) get a zero gradient
template <class TX>
__global__ __launch_bounds__(256) void zero_uncovered_k(TX* dx, int64_t B, int64_t C, int64_t H, int64_t W, int64_t Hc,
                                                        int64_t Wc) {
  const int64_t total = B * C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t w = i % W, h = (i / W) % H;
    if (h >= Hc || w >= Wc) dx[i] = from_f<TX>(0.f);
  }
}

extern "C" int vitk_im2col(const void* x, int x_dtype, void* cols, int cols_dtype, int64_t B, int64_t C, int64_t H,
                           int64_t W, int64_t ph, int64_t pw, int64_t ldc, void* stream) {
  VITK_REQUIRE(x && cols && ph > 0 && pw > 0, "im2col: bad arguments");
  VITK_REQUIRE(ldc >= C * ph * pw, "im2col: ldc=%lld < C*ph*pw=%lld", (long long)ldc, (long long)(C * ph * pw));
  const int64_t total = B * (H / ph) * (W / pw) * ldc;
  if (total == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ew_grid(total)), block(256);
  const int64_t gh = H / ph, gw = W / pw;
  const size_t band_bytes = (size_t)gw * ldc * (cols_dtype == VITK_F32 ? 4 : 2);
  if (band_bytes <= 64 * 1024 && B * gh < (int64_t)1 << 31 && gw * ldc < (int64_t)1 << 30) {
    DISPATCH_DTYPE(x_dtype, TX, {
      DISPATCH_DTYPE(cols_dtype, TC, {
        hipLaunchKernelGGL((im2col_band_k<TX, TC>), dim3((unsigned)(B * gh)), block, band_bytes, st, (const TX*)x, (TC*)cols, C,
                           H, W, (int)ph, (int)pw, (int)ldc, (int)gh, (int)gw);
      });
    });
    VITK_LAUNCH_CHECK("im2col");
    return 0;
  }
  DISPATCH_DTYPE(x_dtype, TX, {
    DISPATCH_DTYPE(cols_dtype, TC, {
      hipLaunchKernelGGL((im2col_k<TX, TC, false>), grid, block, 0, st, (const TX*)x, (TX*)nullptr, (const TC*)nullptr,
                         (TC*)cols, B, C, H, W, ph, pw, ldc);
    });
  });
  VITK_LAUNCH_CHECK("im2col");
  return 0;
}

extern "C" int vitk_col2im(const void* cols, int cols_dtype, void* dx, int dx_dtype, int64_t B, int64_t C, int64_t H,
                           int64_t W, int64_t ph, int64_t pw, int64_t ldc, void* stream) {
  VITK_REQUIRE(cols && dx && ph > 0 && pw > 0, "col2im: bad arguments");
  VITK_REQUIRE(ldc >= C * ph * pw, "col2im: ldc=%lld < C*ph*pw=%lld", (long long)ldc, (long long)(C * ph * pw));
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = B * (H / ph) * (W / pw) * ldc;
  const int64_t Hc = (H / ph) * ph, Wc = (W / pw) * pw;
  DISPATCH_DTYPE(dx_dtype, TX, {
    if ((Hc != H || Wc != W) && B * C * H * W > 0)
      hipLaunchKernelGGL((zero_uncovered_k<TX>), dim3(ew_grid(B * C * H * W)), dim3(256), 0, st, (TX*)dx, B, C, H, W, Hc, Wc);
    if (total > 0) {
      DISPATCH_DTYPE(cols_dtype, TC, {
        hipLaunchKernelGGL((im2col_k<TX, TC, true>), dim3(ew_grid(total)), dim3(256), 0, st, (const TX*)nullptr, (TX*)dx,
                           (const TC*)cols, (TC*)nullptr, B, C, H, W, ph, pw, ldc);
      });
    }
  });
  VITK_LAUNCH_CHECK("col2im");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// gradient preparation: out[r,n] = mask(r,n) * alpha * gamma[n] * g[row(r), n]
// ------------------------------------------------------------------------------------------------
template <class TG, class TO>
__global__ __launch_bounds__(256) void scale_cast_k(const TG* __restrict__ g, int64_t ldg,
                                                    const int64_t* __restrict__ row_idx, const TG* __restrict__ gamma,
                                                    TO* __restrict__ out, int64_t rows, int64_t N, float alpha, uint32_t thr,
                                                    float keep_scale, uint64_t seed) {
  const int64_t nvec = rows * (N / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / (N / 4);
    const int64_t c = (i % (N / 4)) * 4;
    const int64_t src = row_idx ? row_idx[r] : r;
    Vec4<TG> v;
    v.load(g + src * ldg + c);
    Vec4<TO> o;
    if (gamma) {
      Vec4<TG> gm; gm.load(gamma + c);
#pragma unroll
      for (int j = 0; j < 4; j++) o.v[j] = v.v[j] * gm.v[j] * alpha;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) o.v[j] = v.v[j] * alpha;
    }
    if (thr) {
      uint32_t dr[4];
      rng_run<4>(seed, (uint64_t)(r * N + c), dr);   // c and the row length are multiples of 4
#pragma unroll
      for (int j = 0; j < 4; j++) o.v[j] = dr[j] >= thr ? o.v[j] * keep_scale : 0.f;
    }
    o.store(out + r * N + c);
  }
}

extern "C" int vitk_scale_cast(const void* g, int g_dtype, int64_t ldg, const int64_t* row_idx, const void* gamma, void* out,
                               int out_dtype, int64_t rows, int64_t N, float alpha, float drop_p, uint64_t seed,
                               void* stream) {
  VITK_REQUIRE(g && out, "scale_cast: null pointer");
  VITK_REQUIRE(N > 0 && N % 4 == 0 && ldg % 4 == 0, "scale_cast: N=%lld, ldg=%lld must be multiples of 4", (long long)N,
               (long long)ldg);
  VITK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "scale_cast: dropout p=%f out of range", drop_p);
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const uint32_t thr = drop_p > 0.f ? drop_threshold(drop_p) : 0;
  const float ks = 1.f / (1.f - drop_p);
  dim3 grid(ew_grid(rows * (N / 4))), block(256);
  DISPATCH_DTYPE(g_dtype, TG, {
    DISPATCH_DTYPE(out_dtype, TO, {
      hipLaunchKernelGGL((scale_cast_k<TG, TO>), grid, block, 0, st, (const TG*)g, ldg, row_idx, (const TG*)gamma, (TO*)out,
                         rows, N, alpha, thr, ks, seed);
    });
  });
  VITK_LAUNCH_CHECK("scale_cast");
  return 0;
}

// The same pass with the layer-scale gradient taken on the way: dgamma[n] = alpha * sum_r g[row(r), n] * aux[r, n]
// (aux = the branch output before the scale, as the residual epilogue saved it), so the gradient rows are read once for
// both results.  A workgroup walks rows blockIdx.x, blockIdx.x + gridDim.x, ...; thread t owns the 4-column groups t,
// t + 256, ... of every row, keeps their partial sums in registers and leaves them in partial[block][N]; the partials
// are folded like those of the norm backward.
constexpr int SCD_BLOCKS = 1024, SCD_FOLD = 16;
template <class TG, class TO, class TB, int MAXG>
__global__ __launch_bounds__(256) void scale_cast_dgamma_k(const TG* __restrict__ g, int64_t ldg, const int64_t* __restrict__ row_idx,
                                                           const TG* __restrict__ gamma, const TB* __restrict__ aux,
                                                           TO* __restrict__ out, int64_t rows, int N, float alpha, uint32_t thr,
                                                           float keep_scale, uint64_t seed, float* __restrict__ partial) {
  float gm[MAXG][4], acc[MAXG][4];
#pragma unroll
  for (int k = 0; k < MAXG; k++) {
    const int c = (threadIdx.x + k * 256) * 4;
    Vec4<TG> t;
    if (c < N) t.load(gamma + c);
#pragma unroll
    for (int i = 0; i < 4; i++) { gm[k][i] = c < N ? t.v[i] * alpha : 0.f; acc[k][i] = 0.f; }
  }
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const int64_t src = row_idx ? row_idx[r] : r;
#pragma unroll
    for (int k = 0; k < MAXG; k++) {
      const int c = (threadIdx.x + k * 256) * 4;
      if (c < N) {
        Vec4<TG> v; v.load(g + src * ldg + c);
        Vec4<TB> a; a.load(aux + r * (int64_t)N + c);
        Vec4<TO> o;
#pragma unroll
        for (int i = 0; i < 4; i++) { acc[k][i] += v.v[i] * a.v[i]; o.v[i] = v.v[i] * gm[k][i]; }
        if (thr) {
          uint32_t dr[4];
          rng_run<4>(seed, (uint64_t)(r * N + c), dr);
#pragma unroll
          for (int i = 0; i < 4; i++) o.v[i] = dr[i] >= thr ? o.v[i] * keep_scale : 0.f;
        }
        o.store(out + r * (int64_t)N + c);
      }
    }
  }
  float* prow = partial + (size_t)blockIdx.x * N;
#pragma unroll
  for (int k = 0; k < MAXG; k++) {
    const int c = (threadIdx.x + k * 256) * 4;
    if (c < N) *(float4*)(prow + c) = make_float4(acc[k][0] * alpha, acc[k][1] * alpha, acc[k][2] * alpha, acc[k][3] * alpha);
  }
}

extern "C" size_t vitk_scale_cast_dgamma_workspace_bytes(int64_t N) { return (size_t)(SCD_BLOCKS + SCD_FOLD) * (size_t)N * sizeof(float); }

extern "C" int vitk_scale_cast_dgamma(const void* g, int g_dtype, int64_t ldg, const int64_t* row_idx, const void* gamma,
                                      const void* aux, int aux_dtype, void* out, int out_dtype, void* dgamma, int64_t rows,
                                      int64_t N, float alpha, float drop_p, uint64_t seed, void* workspace, void* stream) {
  VITK_REQUIRE(g && out && gamma && aux && dgamma && workspace, "scale_cast_dgamma: null pointer");
  VITK_REQUIRE(N > 0 && N % 4 == 0 && ldg % 4 == 0 && N <= 4096, "scale_cast_dgamma: N=%lld, ldg=%lld must be multiples of 4, N at most 4096",
               (long long)N, (long long)ldg);
  VITK_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "scale_cast_dgamma: dropout p=%f out of range", drop_p);
  hipStream_t st = (hipStream_t)stream;
  const size_t gbytes = (size_t)N * (g_dtype == VITK_F32 ? 4 : 2);
  if (rows == 0) { (void)hipMemsetAsync(dgamma, 0, gbytes, st); return 0; }
  const uint32_t thr = drop_p > 0.f ? drop_threshold(drop_p) : 0;
  const float ks = 1.f / (1.f - drop_p);
  const int grid = (int)std::min<int64_t>(rows, SCD_BLOCKS);
  float* ws = (float*)workspace;
#define SCD_LAUNCH(MAXG)                                                                                                   \
  hipLaunchKernelGGL((scale_cast_dgamma_k<TG, TO, TB, MAXG>), dim3(grid), dim3(256), 0, st, (const TG*)g, ldg, row_idx,    \
                     (const TG*)gamma, (const TB*)aux, (TO*)out, rows, (int)N, alpha, thr, ks, seed, ws)
  DISPATCH_DTYPE(g_dtype, TG, {
    DISPATCH_DTYPE(out_dtype, TO, {
      DISPATCH_DTYPE(aux_dtype, TB, {
        if (N <= 1024) SCD_LAUNCH(1);
        else if (N <= 2048) SCD_LAUNCH(2);
        else SCD_LAUNCH(4);
      });
    });
  });
#undef SCD_LAUNCH
  VITK_LAUNCH_CHECK("scale_cast_dgamma");
  // fold [grid][N] -> [SCD_FOLD][N] -> [N] (dtype of gamma)
  const float* src = ws;
  int64_t src_rows = grid;
  if (grid > 4 * SCD_FOLD) {
    float* ws2 = ws + (size_t)SCD_BLOCKS * N;
    dim3 g1((unsigned)ceil_div(N, 256), (unsigned)SCD_FOLD);
    colsum_one<float, float, float>(true, g1, st, src, N, nullptr, (const float*)nullptr, 0, ws2, N, src_rows, N, 1.f);
    src = ws2;
    src_rows = SCD_FOLD;
  }
  dim3 g2((unsigned)ceil_div(N, 256), 1);
  if (g_dtype == VITK_F32) colsum_one<float, float, float>(true, g2, st, src, N, nullptr, (const float*)nullptr, 0, (float*)dgamma, N, src_rows, N, 1.f);
  else colsum_one<float, float, bf16>(true, g2, st, src, N, nullptr, (const float*)nullptr, 0, (bf16*)dgamma, N, src_rows, N, 1.f);
  VITK_LAUNCH_CHECK("scale_cast_dgamma reduce");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// prefix tokens (vit/vit.py:500-505: cat([cls, register, patches], dim=1) with the learned tokens expanded over the
// batch): out[b, r, :] = cls[r, :] for r < ncls, reg[r - ncls, :] for ncls <= r < ncls + nreg, cast to the dtype of the
// sequence.  Rows from ncls + nreg on belong to the patch tokens and are written by the patch-embedding GEMM.
// ------------------------------------------------------------------------------------------------
template <class TP, class TO>
__global__ __launch_bounds__(256) void prefix_fill_k(const TP* __restrict__ cls, const TP* __restrict__ reg,
                                                     TO* __restrict__ out, int64_t B, int64_t ncls, int64_t nreg, int64_t S,
                                                     int64_t D) {
  const int64_t npre = ncls + nreg, dv = D / 4;
  const int64_t nvec = B * npre * dv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int64_t c = (i % dv) * 4;
    const int64_t r = (i / dv) % npre;
    const int64_t b = i / (dv * npre);
    Vec4<TP> v;
    v.load(r < ncls ? cls + r * D + c : reg + (r - ncls) * D + c);
    Vec4<TO> o;
#pragma unroll
    for (int j = 0; j < 4; j++) o.v[j] = v.v[j];
    o.store(out + (b * S + r) * D + c);
  }
}

extern "C" int vitk_prefix_fill(const void* cls, int64_t ncls, const void* reg, int64_t nreg, int prefix_dtype, void* out,
                                int out_dtype, int64_t B, int64_t S, int64_t D, void* stream) {
  VITK_REQUIRE(out && ncls >= 0 && nreg >= 0 && (cls || ncls == 0) && (reg || nreg == 0), "prefix_fill: null pointer");
  VITK_REQUIRE(D > 0 && D % 4 == 0, "prefix_fill: D=%lld must be a multiple of 4", (long long)D);
  VITK_REQUIRE(S >= ncls + nreg && B >= 0, "prefix_fill: S=%lld shorter than the %lld prefix rows", (long long)S,
               (long long)(ncls + nreg));
  if (B * (ncls + nreg) == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ew_grid(B * (ncls + nreg) * (D / 4))), block(256);
  DISPATCH_DTYPE(prefix_dtype, TP, {
    DISPATCH_DTYPE(out_dtype, TO, {
      hipLaunchKernelGGL((prefix_fill_k<TP, TO>), grid, block, 0, st, (const TP*)cls, (const TP*)reg, (TO*)out, B, ncls, nreg,
                         S, D);
    });
  });
  VITK_LAUNCH_CHECK("prefix_fill");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Fourier position features (vit/pos_enc.py:137-149, the part in front of the D x D projection):
//   f[n, j] = sum_k grid[n, k] * w[k, j]  (k = 2 or 3 coordinates, fp32),  out[n, j] = sin(f) * scale,
//   out[n, J + j] = cos(f) * scale,  scale = 1 / sqrt(J)   (the reference divides by sqrt(J); torch multiplies by the
//   reciprocal).  One thread per (n, j); the backward pass recomputes f and reduces over n per block:
//   dw[k, j] = sum_n grid[n, k] * scale * (cos(f) * g[n, j] - sin(f) * g[n, J + j]).
// ------------------------------------------------------------------------------------------------
template <class TO>
__global__ __launch_bounds__(256) void fourier_fwd_k(const float* __restrict__ grid, const float* __restrict__ w,
                                                     TO* __restrict__ out, int64_t N, int Kc, int64_t J, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N * J; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / J, j = i - n * J;
    float f = 0.f;
    for (int k = 0; k < Kc; k++) f = fmaf(grid[n * Kc + k], w[k * J + j], f);
    out[n * 2 * J + j] = from_f<TO>(sinf(f) * scale);
    out[n * 2 * J + J + j] = from_f<TO>(cosf(f) * scale);
  }
}

// one block per 64 columns j: thread (jj = tid & 63, part = tid >> 6) sums rows n = part, part + 4, ...; LDS adds the 4 parts
template <class TG>
__global__ __launch_bounds__(256) void fourier_bwd_k(const float* __restrict__ grid, const float* __restrict__ w,
                                                     const TG* __restrict__ g, float* __restrict__ dw, int64_t N, int Kc,
                                                     int64_t J, float scale) {
  __shared__ float part_sum[4][3][64];
  const int jj = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + jj;
  float acc[3] = {0.f, 0.f, 0.f};
  if (j < J) {
    for (int64_t n = part; n < N; n += 4) {
      float f = 0.f;
      for (int k = 0; k < Kc; k++) f = fmaf(grid[n * Kc + k], w[k * J + j], f);
      const float df = scale * (cosf(f) * to_f<TG>(g[n * 2 * J + j]) - sinf(f) * to_f<TG>(g[n * 2 * J + J + j]));
      for (int k = 0; k < Kc; k++) acc[k] = fmaf(grid[n * Kc + k], df, acc[k]);
    }
  }
  for (int k = 0; k < 3; k++) part_sum[part][k][jj] = acc[k];
  __syncthreads();
  if (part == 0 && j < J)
    for (int k = 0; k < Kc; k++) dw[k * J + j] = (part_sum[0][k][jj] + part_sum[1][k][jj]) + (part_sum[2][k][jj] + part_sum[3][k][jj]);
}

extern "C" int vitk_fourier_features(const float* grid, const float* w, void* out, int out_dtype, int64_t N, int64_t Kc,
                                     int64_t J, void* stream) {
  VITK_REQUIRE(grid && w && out, "fourier_features: null pointer");
  VITK_REQUIRE(N >= 0 && J > 0 && Kc >= 1 && Kc <= 3, "fourier_features: bad shape N=%lld K=%lld J=%lld", (long long)N,
               (long long)Kc, (long long)J);
  VITK_REQUIRE(out_dtype == VITK_F32 || out_dtype == VITK_BF16, "fourier_features: bad dtype");
  if (N == 0) return 0;
  const float scale = (float)(1.0 / sqrt((double)J));
  DISPATCH_DTYPE(out_dtype, TO, {
    hipLaunchKernelGGL((fourier_fwd_k<TO>), dim3(ew_grid(N * J)), dim3(256), 0, (hipStream_t)stream, grid, w, (TO*)out, N,
                       (int)Kc, J, scale);
  });
  VITK_LAUNCH_CHECK("fourier_features");
  return 0;
}

extern "C" int vitk_fourier_features_bwd(const float* grid, const float* w, const void* g, int g_dtype, float* dw, int64_t N,
                                         int64_t Kc, int64_t J, void* stream) {
  VITK_REQUIRE(grid && w && g && dw, "fourier_features_bwd: null pointer");
  VITK_REQUIRE(N >= 0 && J > 0 && Kc >= 1 && Kc <= 3, "fourier_features_bwd: bad shape N=%lld K=%lld J=%lld", (long long)N,
               (long long)Kc, (long long)J);
  VITK_REQUIRE(g_dtype == VITK_F32 || g_dtype == VITK_BF16, "fourier_features_bwd: bad dtype");
  const float scale = (float)(1.0 / sqrt((double)J));
  DISPATCH_DTYPE(g_dtype, TG, {
    hipLaunchKernelGGL((fourier_bwd_k<TG>), dim3((unsigned)ceil_div(J, 64)), dim3(256), 0, (hipStream_t)stream, grid, w,
                       (const TG*)g, dw, N, (int)Kc, J, scale);
  });
  VITK_LAUNCH_CHECK("fourier_features_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// drop-path pass-through (vit/transformer.py:41-46: index_add leaves the samples outside the subset as they were):
// dst[idx[i], :] = src[idx[i], :] for the listed samples, 16 bytes per thread.  The kept samples' rows are written
// by the kernel that follows (residual epilogue / norm backward), so the merged stream is never cloned whole.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_samples_k(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                      const int64_t* __restrict__ idx, int64_t n, int64_t vec_per_sample) {
  const int64_t total = n * vec_per_sample;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t s = idx[i / vec_per_sample], off = s * vec_per_sample + i % vec_per_sample;
    dst[off] = src[off];
  }
}

extern "C" int vitk_copy_samples(const void* src, void* dst, const int64_t* idx, int64_t n, int64_t sample_bytes, void* stream) {
  VITK_REQUIRE(src && dst && (idx || n == 0) && n >= 0, "copy_samples: null pointer");
  VITK_REQUIRE(sample_bytes > 0 && sample_bytes % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0,
               "copy_samples: samples must be whole 16-byte units at 16-byte aligned addresses");
  if (n == 0) return 0;
  hipLaunchKernelGGL(copy_samples_k, dim3(ew_grid(n * (sample_bytes / 16))), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, (uint4*)dst, idx, n, sample_bytes / 16);
  VITK_LAUNCH_CHECK("copy_samples");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Diagnostics: `blocks` workgroups of one wave per SIMD that stay resident for about `cycles` shader clocks and do
// nothing -- a stand-in for a collective's kernel on a side stream, to see what a launch on the main stream loses when
// some CUs are taken (tools/bench_hog.py).  The loop is bounded by the clock AND by an iteration count.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void debug_hold_cus_k(long long cycles, unsigned int* sink) {
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  unsigned int n = 0;
  for (long long it = 0; it < (1ll << 26); it++) {
    if ((long long)__builtin_amdgcn_s_memtime() - t0 >= cycles) break;
    __builtin_amdgcn_s_sleep(32);
    n++;
  }
  if (sink && n == 0xffffffffu) *sink = n;
}

extern "C" int vitk_debug_hold_cus(int blocks, long long cycles, void* stream) {
  VITK_REQUIRE(blocks > 0 && blocks <= 1024 && cycles > 0 && cycles <= (1ll << 34), "debug_hold_cus: bad arguments");
  // 96 KiB of (unused) LDS per workgroup: no two of them fit one CU, so `blocks` CUs are held
  hipLaunchKernelGGL(debug_hold_cus_k, dim3((unsigned)blocks), dim3(256), 96 * 1024, (hipStream_t)stream, cycles,
                     (unsigned int*)nullptr);
  VITK_LAUNCH_CHECK("debug_hold_cus");
  return 0;
}
