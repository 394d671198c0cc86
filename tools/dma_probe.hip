// Probe: how fast can one workgroup per CU pull GEMM operand tiles from L2 into LDS by LDS-DMA, walking tiles the way
// the 256x256 GEMM does, with (a) 16-row x 64-byte pieces (half K-steps) and (b) 8-row x 128-byte pieces (whole 128-byte
// lines per row)?  No LDS reads, no MFMAs: DMA + counted waits + one barrier per step only.
// Build: hipcc --offload-arch=gfx950 -O3 -o exp/dma_probe tools/dma_probe.hip ; run: exp/dma_probe M N K
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

__device__ __forceinline__ u32x4_t rsrc_words(const void* base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  u32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void dma2(const u32x4_t& rsrc, uint32_t lds, uint32_t o0, uint32_t o1) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\t"
               "buffer_load_dwordx4 %2, %4, 0 offen lds\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %3, %4, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds), "v"(o0), "v"(o1), "s"(rsrc) : "memory", "scc");
}
__device__ __forceinline__ int64_t xcd_remap(int64_t bid, int64_t nwg) {
  const int64_t xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__device__ __forceinline__ void tile_coords(int64_t wg, int64_t tiles_m, int64_t tiles_n, int64_t& tm, int64_t& tn) {
  const int64_t per_panel = 8 * tiles_n;
  const int64_t panel = wg / per_panel, in_panel = wg - panel * per_panel;
  const int64_t rows_here = tiles_m - panel * 8 < 8 ? tiles_m - panel * 8 : 8;
  tm = panel * 8 + in_panel % rows_here;
  tn = in_panel / rows_here;
}

// A [M][K], B [N][K] bf16 (both K-contiguous).  WIDE = 0: half-stage (32 k) pieces of 16 rows x 64 B, ring of 4 slots,
// 3 half-stages in flight.  WIDE = 1: K-step (64 k) pieces of 8 rows x 128 B, ring of 2 slots (64 KiB each), 1 in flight.
template <int WIDE>
__global__ __launch_bounds__(512) void probe(const uint16_t* A, const uint16_t* B, int64_t M, int64_t N, int64_t K) {
  __shared__ __attribute__((aligned(1024))) char smem[131072];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tiles_m = M / 256, tiles_n = N / 256, total = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7;
  const int64_t xcd_tiles = (total >> 3) + (xcd < (total & 7) ? 1 : 0);
  const int64_t xcd_blocks = ((int64_t)gridDim.x - xcd + 7) >> 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  for (int64_t jt = blockIdx.x >> 3; jt < xcd_tiles; jt += xcd_blocks) {
    int64_t tm, tn;
    tile_coords(xcd_remap(xcd + 8 * jt, total), tiles_m, tiles_n, tm, tn);
    const u32x4_t ra = rsrc_words(A + tm * 256 * K, (uint32_t)(256 * K * 2));
    const u32x4_t rb = rsrc_words(B + tn * 256 * K, (uint32_t)(256 * K * 2));
    if (WIDE == 0) {
      uint32_t oa[2], ob[2];
      for (int j = 0; j < 2; j++) {
        const int p = (wid * 2 + j) * 64 + lane;
        const int row = p >> 2, c = p & 3;
        oa[j] = ob[j] = (uint32_t)((row * K + c * 8) * 2);
      }
      const int nh = (int)(K / 32);
      auto req = [&](int h) {
        const uint32_t base = lds0 + (h & 3) * 32768 + wid * 2048;
        dma2(ra, base, oa[0] + h * 64, oa[1] + h * 64);
        dma2(rb, base + 16384, ob[0] + h * 64, ob[1] + h * 64);
      };
      req(0); req(1); req(2);
      for (int h = 0; h < nh; h++) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        req(h + 3);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      uint32_t oa[4], ob[4];
      for (int j = 0; j < 4; j++) {
        const int p = (wid * 4 + j) * 64 + lane;
        const int row = p >> 3, c = p & 7;
        oa[j] = ob[j] = (uint32_t)((row * K + c * 8) * 2);
      }
      const int nk = (int)(K / 64);
      auto req = [&](int k) {
        const uint32_t base = lds0 + (k & 1) * 65536 + wid * 4096;
        dma2(ra, base, oa[0] + k * 128, oa[1] + k * 128);
        dma2(ra, base + 2048, oa[2] + k * 128, oa[3] + k * 128);
        dma2(rb, base + 32768, ob[0] + k * 128, ob[1] + k * 128);
        dma2(rb, base + 32768 + 2048, ob[2] + k * 128, ob[3] + k * 128);
      };
      req(0);
      for (int k = 0; k < nk; k++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        req(k + 1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
}

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 69632, N = argc > 2 ? atoll(argv[2]) : 768, K = argc > 3 ? atoll(argv[3]) : 6144;
  uint16_t *A, *B;
  hipMalloc(&A, (size_t)(M + 256) * K * 2 + 65536);
  hipMalloc(&B, (size_t)(N + 256) * K * 2 + 65536);
  hipMemset(A, 0x3c, (size_t)M * K * 2);
  hipMemset(B, 0x3c, (size_t)N * K * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)(M / 256) * (N / 256) * (K / 64) * 65536.0;
  for (int wide = 0; wide < 2; wide++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; i++) {
        if (wide) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, A, B, M, N, K);
        else hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, A, B, M, N, K);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
      if (rep) printf("M=%lld N=%lld K=%lld %s pieces: %.4f ms  %.2f TB/s into LDS  (MFMA-equivalent %.0f TF/s)\n", (long long)M, (long long)N, (long long)K,
                      wide ? "8x128B" : "16x64B", ms, bytes / ms / 1e9, 2.0 * M * N * K / ms / 1e9);
    }
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
