// Dev tool (GPU box): how long does a wave wait for a 16 KiB block of HBM-resident rows when every CU of the chip
// does the same once per "step", with and without a write burst per step, and with the rows shared by 8 CUs?
// (The access pattern of lstm_fwd_cluster_kernel without its arithmetic -- DESIGN.md section 5.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/memlat tools/memlat.hip && /tmp/memlat
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

// mode bit 0: write burst (14 KiB per wave and step); bit 1: the 8 blocks b, b+8, .. b+56 of a group read the SAME rows
// (8 waves x 8 tiles), else every wave reads its own; bit 3: one wave per XCD issues
// buffer_wbl2 after the burst; `work`: idle cycles between the burst and the next read
__global__ __launch_bounds__(512) void memlat_kernel(const uint4* __restrict__ X, uint4* __restrict__ Z, int steps,
                                                     int mode, int work, unsigned long long* stamps) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, s = j & 7, cid = xcd + 8 * (j >> 3);
  const long tile_r = (mode & 2) ? (8L * cid + w) : (8L * blockIdx.x + w);      // whose rows this wave reads
  const long tile_w = 8L * blockIdx.x + w;
  const long ntile_r = (mode & 2) ? gridDim.x : 8L * gridDim.x;
  uint4 acc = make_uint4(0, 0, 0, 0);
  // bit 4: the two halves of the workgroup (waves 0-3 / 4-7) run half a step apart and meet per half through LDS
  __shared__ int gcnt[2];
  if (threadIdx.x < 2) gcnt[threadIdx.x] = 0;
  __syncthreads();
  const int g = w >> 2;
  if ((mode & 16) && g == 1) {
    const unsigned long long t00 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t00 < 9000ull + work / 2) __builtin_amdgcn_s_sleep(8);
  }
  for (int t = 0; t < steps; ++t) {
    const uint4* xp = X + ((tile_r * steps + t) * 16) * 64 + lane;      // 16 KiB per (tile, step)
    const unsigned long long t0 = __builtin_readcyclecounter();
    uint4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = xp[k * 64];
    if (!(mode & 4)) {
#pragma unroll
      for (int k = 0; k < 16; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
    }
    if (!(mode & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // bit 2: the burst goes out behind the reads
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (mode & 1) {
      uint4* zp = Z + ((tile_w * steps + t) * 14) * 64 + lane;          // 14 KiB per wave and step
#pragma unroll
      for (int k = 0; k < 14; ++k) zp[k * 64] = acc;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (mode & 4) {
#pragma unroll
      for (int k = 0; k < 16; ++k) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
    }
    if ((mode & 8) && (blockIdx.x >> 3) == 0 && w == 0)      // one wave per XCD starts the L2 write-back
      asm volatile("buffer_wbl2 sc1" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (work > 0) {
      while (__builtin_readcyclecounter() - t2 < (unsigned long long)work) __builtin_amdgcn_s_sleep(8);
    }
    if (mode & 16) {
      if (lane == 0) {
        __hip_atomic_fetch_add(&gcnt[g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(&gcnt[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (t + 1))
          __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      __syncthreads();
    }
    if (blockIdx.x == 17 && threadIdx.x == 0) {
      stamps[3 * t] = t1 - t0;
      stamps[3 * t + 1] = t2 - t1;
      stamps[3 * t + 2] = __builtin_readcyclecounter() - t0;
    }
  }
  if (acc.x == 0x12345678u && ntile_r < 0) Z[0] = acc;      // keep the loads alive
}

int main() {
  const int steps = 128, blocks = 256;
  const size_t xbytes = (size_t)8 * blocks * steps * 16384, zbytes = (size_t)8 * blocks * steps * 14 * 1024;
  uint4 *X, *Z;
  unsigned long long* st;
  CK(hipMalloc(&X, xbytes));
  CK(hipMalloc(&Z, zbytes));
  CK(hipMalloc(&st, 3 * steps * sizeof(unsigned long long)));
  CK(hipMemset(X, 1, xbytes));
  CK(hipMemset(Z, 0, zbytes));
  std::vector<unsigned long long> h(3 * steps);
  const int modes[][2] = {{3, 0}, {19, 0}, {3, 8000}, {19, 8000}, {3, 12000}, {19, 12000}};
  for (auto& m : modes) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      // evict: touch Z so X is not in L2 / MALL from the last run
      CK(hipMemset(Z, rep, zbytes));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(memlat_kernel, dim3(blocks), dim3(512), 0, 0, X, Z, steps, m[0], m[1], st);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    CK(hipMemcpy(h.data(), st, 3 * steps * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> rd, wr, tot;
    for (int t = 8; t < steps - 8; ++t) { rd.push_back(h[3 * t]); wr.push_back(h[3 * t + 1]); tot.push_back(h[3 * t + 2]); }
    std::sort(rd.begin(), rd.end()); std::sort(wr.begin(), wr.end()); std::sort(tot.begin(), tot.end());
    printf("antiphase %d reads-before-burst %d wbl2 %d writes %d shared-by-8 %d idle %5d cycles: read wait median %6llu (p90 %6llu), write+ack %6llu, step %6llu cycles; kernel %.3f ms\n",
           (m[0] >> 4) & 1, (m[0] >> 2) & 1, (m[0] >> 3) & 1, m[0] & 1, (m[0] >> 1) & 1, m[1], rd[rd.size() / 2], rd[rd.size() * 9 / 10], wr[wr.size() / 2], tot[tot.size() / 2], best);
  }
  return 0;
}
