// Dev tool (GPU box): is a SECOND read of dZ served from the Infinity Cache when dX and dW of a row chunk follow each
// other inside one persistent launch?  (VERDICT r3 item 7 / DESIGN.md section 8 round 4.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dz_reuse tools/dz_reuse.hip && /tmp/dz_reuse
// Memory pattern of time layer 1's BPTT GEMMs at the BASELINE shape WITHOUT arithmetic: dZ [4 column tiles][M][256] bf16
// (2.1 GB, column-tile-major as the sweep writes it), X [M, 256], Hs [M, 256], dX [M, 256] (0.5 GB each), M = 1,048,576.
//   role A (dX = dZ W^T):  a workgroup takes 256-row tiles; a wave reads its 32 rows of all four column tiles (4 x 16 KiB)
//                          and writes 16 KiB of dX.
//   role B (dW/dU = [X | Hprev]^T dZ): workgroup = (row split of 64, column tile of 4; the four column tiles of a split on
//                          one XCD); per stage of 32 rows a wave... the WORKGROUP reads one 16 KiB block of dZ and 16 + 16
//                          KiB of X / Hs rows (the latter shared by the four column tiles: L2).
// Patterns: "separate" = A over all rows, then B over all rows (today: two launches; dZ comes from HBM twice);
//           "chunked"  = for every chunk of R rows: A on the chunk, then B on the chunk, same launch, no grid barrier
//                        (R x 2 KiB of dZ between the two reads: 128 MiB at R = 65,536, inside the 256 MiB cache).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

constexpr long M = 1L << 20;

__device__ __forceinline__ void eat(uint4& acc, const uint4& v) { acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }

// one wave: 32 rows x 256 columns of one column tile = 16 KiB contiguous
__device__ __forceinline__ void read_block(const uint4* p, int lane, uint4& acc) {
  uint4 v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = p[k * 64 + lane];
#pragma unroll
  for (int k = 0; k < 16; ++k) eat(acc, v[k]);
}

// rows [r0, r0 + nrows): role A, row tiles of 256 dealt round-robin over the workgroups
__device__ void role_a(const uint4* dZ, uint4* dX, long r0, long nrows, int lane, int w, uint4& acc) {
  for (long rt = blockIdx.x; rt < nrows / 256; rt += gridDim.x) {
    const long row = r0 + rt * 256 + w * 32;                 // this wave's 32 rows
#pragma unroll 1
    for (int ct = 0; ct < 4; ++ct) read_block(dZ + ((long)ct * M + row) * 32, lane, acc);    // 256 cols x 2 B = 32 uint4 per row
    uint4* o = dX + row * 32;
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k * 64 + lane] = acc;
  }
}
// rows [r0, r0 + nrows): role B, 64 row splits x 4 column tiles; stage = 32 rows; the 8 waves of a workgroup take stages
// w, w + 8, ... of its split (in the real kernel all waves share a stage through LDS; the bytes per workgroup are the same)
__device__ void role_b(const uint4* dZ, const uint4* X, const uint4* Hs, long r0, long nrows, int lane, int w, uint4& acc) {
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, ct = j & 3, k = j >> 2;
  const long per = nrows / 64, s0 = r0 + (long)(xcd * 8 + k) * per;
  for (long st = w; st < per / 32; st += 8) {
    const long row = s0 + st * 32;
    read_block(dZ + ((long)ct * M + row) * 32, lane, acc);
    read_block(X + row * 32, lane, acc);
    read_block(Hs + row * 32, lane, acc);
  }
}

// mode 0: separate (A all, B all)   1: chunked (A chunk, B chunk, ...)   2: A only   3: B only
__global__ __launch_bounds__(512) void replay(const uint4* dZ, const uint4* X, const uint4* Hs, uint4* dX, long chunk,
                                              int mode, uint4* sink) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint4 acc = make_uint4(1, 2, 3, 4);
  if (mode == 0 || mode == 2) role_a(dZ, dX, 0, M, lane, w, acc);
  if (mode == 0 || mode == 3) role_b(dZ, X, Hs, 0, M, lane, w, acc);
  if (mode == 1)
    for (long r0 = 0; r0 < M; r0 += chunk) {
      role_a(dZ, dX, r0, chunk, lane, w, acc);
      role_b(dZ, X, Hs, r0, chunk, lane, w, acc);
    }
  if (acc.x == 0x12345678u && chunk < 0) sink[0] = acc;
}

int main() {
  uint4 *dZ, *X, *Hs, *dX, *sink, *evict;
  const size_t zb = (size_t)4 * M * 512, xb = (size_t)M * 512, eb = (size_t)1 << 30;
  CK(hipMalloc(&dZ, zb)); CK(hipMalloc(&X, xb)); CK(hipMalloc(&Hs, xb)); CK(hipMalloc(&dX, xb)); CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&evict, eb));
  CK(hipMemset(dZ, 1, zb)); CK(hipMemset(X, 2, xb)); CK(hipMemset(Hs, 3, xb));
  struct { int mode; long chunk; const char* name; double gb; } runs[] = {
      {2, 0, "A only (dX pattern: dZ in, dX out)", (zb + xb) / 1e9},
      {3, 0, "B only (dW pattern: dZ + X + Hs in; X, Hs read by 4 column tiles)", (zb + 2 * xb) / 1e9},
      {0, 0, "separate: A over all rows, then B over all rows (one launch)", (2 * zb + 3 * xb) / 1e9},
      {1, 262144, "chunked, R = 262,144 rows (512 MiB of dZ between the two reads)", (2 * zb + 3 * xb) / 1e9},
      {1, 131072, "chunked, R = 131,072 rows (256 MiB)", (2 * zb + 3 * xb) / 1e9},
      {1, 65536, "chunked, R =  65,536 rows (128 MiB)", (2 * zb + 3 * xb) / 1e9},
      {1, 32768, "chunked, R =  32,768 rows ( 64 MiB)", (2 * zb + 3 * xb) / 1e9},
      {1, 16384, "chunked, R =  16,384 rows ( 32 MiB)", (2 * zb + 3 * xb) / 1e9},
  };
  for (auto& r : runs) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(evict, rep, eb));                      // the operands are not in L2 / MALL from the last run
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(replay, dim3(256), dim3(512), 0, 0, dZ, X, Hs, dX, r.chunk, r.mode, sink);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("%-78s %.3f ms  (%.2f GB requested by the CUs: %.2f TB/s)\n", r.name, best, r.gb, r.gb / best);
  }
  printf("today (with arithmetic): lstm_wgrad_bf16 1.16 ms + gemm_nt dX 0.85 ms = 2.01 ms for time layer 1\n");
  return 0;
}
