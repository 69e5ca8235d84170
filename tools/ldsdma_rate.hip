// Dev tool (GPU box): how fast can one CU pull data into LDS with global_load_lds_dwordx4?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsdma_rate tools/ldsdma_rate.hip && /tmp/ldsdma_rate
// 256 workgroups x 8 waves; every wave streams 1 KiB pieces into a 128 KiB LDS ring with `depth` pieces in flight
// (counted vmcnt, no barriers, nothing consumed).  Source: a per-workgroup slice that is either small (re-read from
// L2: "l2") or large (streamed from HBM: "hbm"); pieces are 2 rows x 512 B (the weight-gradient GEMM's shape) or
// 16 rows x 64 B (the 256 x 256 NT GEMM's).
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <stdio.h>
#include <stdlib.h>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
template <int DEPTH, int ROWB>   // ROWB: bytes per contiguous run (512 or 64)
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t slice_bytes, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)smem);
  const char* base = src + (size_t)blockIdx.x * slice_bytes;
  constexpr int LPR = ROWB / 16;                 // lanes per run
  const int run = lane / LPR, off = (lane % LPR) * 16;
  const size_t row_stride = ROWB == 512 ? 2048 : 2048;   // runs sit 2 KiB apart in memory (a row-major [M, 1024] bf16 matrix)
  size_t pos = (size_t)w * (64 / LPR) * row_stride;
  const size_t step = (size_t)8 * (64 / LPR) * row_stride;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    glds16(base + (pos & (slice_bytes - 1)) + (size_t)run * row_stride + off, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((i & 15) * 8 + w) * 1024u));
    pos += step;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int DEPTH, int ROWB> void run(const char* name, const char* src, size_t slice, int iters, unsigned long long* cyc) {
  hipFuncSetAttribute((const void*)k<DEPTH, ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<DEPTH, ROWB>), dim3(256), dim3(512), 128 * 1024, 0, src, slice, iters, cyc);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL((k<DEPTH, ROWB>), dim3(256), dim3(512), 128 * 1024, 0, src, slice, iters, cyc);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double bytes = 256.0 * 8 * iters * 1024;
  printf("%-34s depth %2d: %.3f ms, %.2f TB/s aggregate, %.1f GB/s per CU, %.1f B/clk/CU (block 0: %llu cycles)\n", name, DEPTH, ms,
         bytes / ms / 1e9, bytes / 256 / ms / 1e6, 8.0 * iters * 1024 / (double)h[0], h[0]);
}
int main() {
  const size_t big = (size_t)256 * (32 << 20);     // 8 GiB: 32 MiB per workgroup (HBM streaming)
  char* src; hipMalloc(&src, big); hipMemset(src, 1, big);
  unsigned long long* cyc; hipMalloc(&cyc, 256 * 8);
  const int iters = 4096;                           // 4 MiB per wave, 32 MiB per workgroup
  run<4, 512>("l2  (64 KiB slice), 2 x 512 B", src, 64 << 10, iters, cyc);
  run<12, 512>("l2  (64 KiB slice), 2 x 512 B", src, 64 << 10, iters, cyc);
  run<12, 64>("l2  (64 KiB slice), 16 x 64 B", src, 64 << 10, iters, cyc);
  run<4, 512>("hbm (32 MiB slice), 2 x 512 B", src, 32 << 20, iters, cyc);
  run<12, 512>("hbm (32 MiB slice), 2 x 512 B", src, 32 << 20, iters, cyc);
  run<12, 64>("hbm (32 MiB slice), 16 x 64 B", src, 32 << 20, iters, cyc);
  return 0;
}
