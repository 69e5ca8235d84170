// Dev tool (GPU box): memory / exchange pattern of a K-SPLIT clustered BPTT sweep for the H = 256 time axis, without
// any arithmetic -- the question VERDICT r2 item 4 asks before anyone builds the kernel: can 8 workgroups of one XCD,
// each owning the gate columns of 32 hidden units (its U^T rows resident in LDS, dW / dU in accumulators), run the
// backward recurrence if the only thing they exchange per step is the partial dh = dz_slice * U^T_slice ?
//
// Per step and wave (wave w of member s works on tile w of its cluster, as in lstm_fwd_cluster_kernel):
//   HBM reads   : its slices of the gate stash (4 KiB), c (2 KiB), dH (2 KiB), x and h_{t-1} (2 + 2 KiB for dW / dU)
//   HBM write   : its dz slice (32 rows x 128 gate columns, bf16: 8 KiB) -- dX = dz W^T stays a GEMM
//   exchange out: partial dh [32 rows x 256 units] as 8 pieces of 32 x 32, one per destination member
//                 (fp32: 4 KiB each = 32 KiB; bf16: 2 KiB each = 16 KiB), then one counter increment per member
//   exchange in : the 8 pieces of ITS 32 units from the 8 members (sc1 loads past the non-coherent L1)
// `idle` cycles stand in for the MFMAs that cannot overlap with the exchange (dz U^T: 64 MFMAs per wave).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bptt_ksplit tools/bptt_ksplit.hip && /tmp/bptt_ksplit
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

// PV = 16-byte vectors per lane and destination piece: 4 (fp32 partials, 4 KiB per piece) or 2 (bf16, 2 KiB)
template <int PV>
__global__ __launch_bounds__(512) void ksplit_kernel(const uint4* __restrict__ S, uint4* __restrict__ DZ, uint4* __restrict__ XS,
                                                     int* __restrict__ cnt, int steps, int idle, int exch,
                                                     unsigned long long* stamps) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, s = j & 7, cid = xcd + 8 * (j >> 3);
  int* c = cnt + cid * 32;                                  // one 128-byte line per cluster
  const long tile = 8L * cid + w;
  uint4 acc = make_uint4(lane, w, s, cid);
  for (int t = 0; t < steps; ++t) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    // ---- this member's stash slices of the step: 12 KiB per wave, unique to the wave
    const uint4* sp = S + (((tile * 8 + s) * steps + t) * 12) * 64 + lane;
    uint4 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = sp[k * 64];
#pragma unroll
    for (int k = 0; k < 12; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
    const unsigned long long t1 = __builtin_readcyclecounter();
    // ---- dz slice out (8 KiB), not waited for
    uint4* zp = DZ + (((tile * 8 + s) * steps + t) * 8) * 64 + lane;
#pragma unroll
    for (int k = 0; k < 8; ++k) zp[k * 64] = acc;
    if (idle > 0) {                                         // dz U^T
      const unsigned long long ti = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - ti < (unsigned long long)idle) __builtin_amdgcn_s_sleep(4);
    }
    unsigned long long t2 = __builtin_readcyclecounter(), t3 = t2;
    if (exch) {
      // ---- partial dh out: piece d (for member d) at [tile][parity][d][s]
      uint4* xo = XS + ((((tile * 2 + (t & 1)) * 8) * 8 + s) * PV) * 64 + lane;
#pragma unroll
      for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int k = 0; k < PV; ++k) xo[((d * 8) * PV + k) * 64] = acc;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // acknowledged by L2 before the counter moves
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t2 = __builtin_readcyclecounter();
      if (lane == 0) {
        const unsigned long long tw = __builtin_readcyclecounter();
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8 * (t + 1) &&
               __builtin_readcyclecounter() - tw < 20000000ull)
          __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_wave_barrier();
      // ---- the 8 pieces of this member's units
      const volatile uint4* xiv = XS + ((((tile * 2 + (t & 1)) * 8 + s) * 8) * PV) * 64 + lane;
      uint4 p[8 * PV];
#pragma unroll
      for (int k = 0; k < 8 * PV; ++k) p[k] = ((const uint4*)xiv)[k * 64];   // (the real kernel: sc1 loads past the non-coherent L1)
#pragma unroll
      for (int k = 0; k < 8 * PV; ++k) { acc.x ^= p[k].x; acc.y += p[k].y; acc.z ^= p[k].z; acc.w += p[k].w; }   // all four dwords: the asm load writes them
      t3 = __builtin_readcyclecounter();
    } else {
      __syncthreads();
    }
    if (blockIdx.x == 17 && threadIdx.x == 0) {
      stamps[4 * t] = t1 - t0;
      stamps[4 * t + 1] = t2 - t1;
      stamps[4 * t + 2] = t3 - t2;
      stamps[4 * t + 3] = __builtin_readcyclecounter() - t0;
    }
  }
  if (acc.x == 0x12345678u && steps < 0) DZ[0] = acc;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int steps = 128, blocks = 256;
  const size_t sbytes = (size_t)8 * blocks * steps * 12 * 1024, zbytes = (size_t)8 * blocks * steps * 8 * 1024;
  const size_t xbytes = (size_t)256 * 2 * 64 * 4096;       // fp32 pieces: 128 MiB (bf16 uses half of it)
  uint4 *S, *DZ, *XS;
  int* cnt;
  unsigned long long* st;
  CK(hipMalloc(&S, sbytes));
  CK(hipMalloc(&DZ, zbytes));
  CK(hipMalloc(&XS, xbytes));
  CK(hipMalloc(&cnt, 64 * 128));
  CK(hipMalloc(&st, 4 * steps * sizeof(unsigned long long)));
  CK(hipMemset(S, 1, sbytes));
  CK(hipMemset(XS, 0, xbytes));
  std::vector<unsigned long long> h(4 * steps);
  printf("K-split BPTT exchange replay: %d workgroups x 8 waves, %d steps; HBM per launch: %.2f GB read, %.2f GB written\n", blocks,
         steps, sbytes / 1e9, zbytes / 1e9);
  const int cfgs[][3] = {{0, 0, 4}, {0, 4000, 4}, {1, 0, 4}, {1, 4000, 4}, {1, 8000, 4}, {1, 0, 2}, {1, 4000, 2}, {1, 8000, 2}};
  for (auto& m : cfgs) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(DZ, rep, zbytes));
      CK(hipMemset(cnt, 0, 64 * 128));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (m[2] == 4)
        hipLaunchKernelGGL(ksplit_kernel<4>, dim3(blocks), dim3(512), 0, 0, S, DZ, XS, cnt, steps, m[1], m[0], st);
      else
        hipLaunchKernelGGL(ksplit_kernel<2>, dim3(blocks), dim3(512), 0, 0, S, DZ, XS, cnt, steps, m[1], m[0], st);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    CK(hipMemcpy(h.data(), st, 4 * steps * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> a, b, c, d;
    for (int t = 8; t < steps - 8; ++t) { a.push_back(h[4 * t]); b.push_back(h[4 * t + 1]); c.push_back(h[4 * t + 2]); d.push_back(h[4 * t + 3]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end()); std::sort(d.begin(), d.end());
    printf("exchange %d (%s partials, %2d KiB out + in per wave and step) idle %5d: stash wait %6llu, stores + ack + barrier %6llu, "
           "counter + pieces %6llu, step %6llu cycles; kernel %.3f ms\n",
           m[0], m[2] == 4 ? "fp32" : "bf16", m[0] ? 8 * m[2] : 0, m[1], a[a.size() / 2], b[b.size() / 2], c[c.size() / 2], d[d.size() / 2], best);
  }
  return 0;
}
