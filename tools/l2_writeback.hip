// Dev tool (GPU box): does a line that is OVERWRITTEN while it is resident in an XCD's L2 reach the memory fabric once
// (write-back) or on every store (write-through)?  The forward cluster kernel's exchange ring (2 slots per tile, 8 MB per
// launch, every slot overwritten 64 times in a 128-step sweep) shows up as 0.5 GB of WRITE_SIZE per launch
// (profiles/r04_f_pmc_hbm_per_kernel.csv); this replays the pattern with several store flavours.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_writeback tools/l2_writeback.hip
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/l2wb -- /tmp/l2_writeback
// Each of 256 workgroups owns 32 KiB and rewrites it ROUNDS times (a different value per round), waiting for the
// acknowledgements between rounds: 8 MiB resident, 512 MiB stored.  WRITE_SIZE per launch ~ 8 MiB = write-back,
// ~ 512 MiB = every store goes out.  Mode 6 also has a partner workgroup of the same XCD read the region each round
// (the exchange proper).
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

constexpr int ROUNDS = 64, WG_BYTES = 32768;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void st16(uint4* p, uint4 v4) {
  const u32x4 v = {v4.x, v4.y, v4.z, v4.w};
  if constexpr (MODE == 0) *p = v4;
  if constexpr (MODE == 1) {
    __builtin_nontemporal_store(v4.x, (unsigned*)p);
    __builtin_nontemporal_store(v4.y, (unsigned*)p + 1);
    __builtin_nontemporal_store(v4.z, (unsigned*)p + 2);
    __builtin_nontemporal_store(v4.w, (unsigned*)p + 3);
  }
  if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if constexpr (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if constexpr (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void rewrite(uint4* buf, uint4* sink) {
  uint4* mine = buf + (size_t)blockIdx.x * (WG_BYTES / 16);
  for (int r = 0; r < ROUNDS; ++r) {
    const uint4 v = make_uint4(r, blockIdx.x, threadIdx.x, 7);
#pragma unroll
    for (int k = 0; k < WG_BYTES / 16 / 256; ++k) st16<MODE>(mine + k * 256 + threadIdx.x, v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (sink == buf) sink[0] = mine[0];
}

// the exchange: workgroup b writes its region, signals, workgroup b ^ 8 (same XCD under round-robin dispatch) reads it
// with sc1 loads (as the cluster kernel does) and signals back; two flags per pair in `flags`
__global__ __launch_bounds__(256) void exchange(uint4* buf, int* flags, uint4* sink) {
  uint4* mine = buf + (size_t)blockIdx.x * (WG_BYTES / 16);
  const uint4* theirs = buf + (size_t)(blockIdx.x ^ 8) * (WG_BYTES / 16);
  int* my_flag = flags + blockIdx.x * 32;
  int* their_flag = flags + (blockIdx.x ^ 8) * 32;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int r = 0; r < ROUNDS; ++r) {
    const uint4 v = make_uint4(r, blockIdx.x, threadIdx.x, 7);
#pragma unroll
    for (int k = 0; k < WG_BYTES / 16 / 256; ++k) mine[k * 256 + threadIdx.x] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(my_flag, r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(their_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r + 1 && ++spins < (1L << 22)) {}
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < WG_BYTES / 16 / 256; ++k) {
      u32x4 q;
      asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(q) : "v"(theirs + k * 256 + threadIdx.x) : "memory");
      acc.x ^= q.x; acc.y += q.y;
    }
    __syncthreads();          // (the partner may overwrite only after this round's reads: one more flag round in a real
                              //  protocol; here the values do not matter, only the traffic)
  }
  if (acc.x == 0x12345u) sink[0] = acc;
}

int main() {
  uint4 *buf, *sink;
  int* flags;
  CK(hipMalloc(&buf, (size_t)256 * WG_BYTES));
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&flags, 256 * 128));
  CK(hipMemset(flags, 0, 256 * 128));
  hipLaunchKernelGGL(rewrite<0>, dim3(256), dim3(256), 0, 0, buf, sink);
  hipLaunchKernelGGL(rewrite<1>, dim3(256), dim3(256), 0, 0, buf, sink);
  hipLaunchKernelGGL(rewrite<2>, dim3(256), dim3(256), 0, 0, buf, sink);
  hipLaunchKernelGGL(rewrite<3>, dim3(256), dim3(256), 0, 0, buf, sink);
  hipLaunchKernelGGL(rewrite<4>, dim3(256), dim3(256), 0, 0, buf, sink);
  hipLaunchKernelGGL(rewrite<5>, dim3(256), dim3(256), 0, 0, buf, sink);
  hipLaunchKernelGGL(exchange, dim3(256), dim3(256), 0, 0, buf, flags, sink);
  CK(hipDeviceSynchronize());
  printf("done: 7 launches (rewrite<0..5> = plain, nontemporal, sc0, sc1, sc0 sc1, nt; exchange), %d rounds x 8 MiB each\n", ROUNDS);
  return 0;
}
