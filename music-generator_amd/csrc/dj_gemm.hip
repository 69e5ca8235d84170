// Dense contractions of the DeepJ hot path on the MI355X matrix cores.
//
//  dj_gemm_nt : C[M,N]  = A[M,K] * Bt[N,K]^T (+ bias)     -- input-to-hidden projections
//               x*W of every LSTM layer (reference model.py:84,122, Keras LSTM kernel) and
//               the input gradient dX = dZ * W^T of BPTT.
//  dj_gemm_tn : C[Ka,N] += A[M,Ka]^T * B[M,N]  (fp32 atomics, split over M)
//               -- weight gradients dW = X^T dZ, dU = Hprev^T dZ of BPTT (TF autodiff
//               of model.py:84,122 in the reference).
//
// 128x128 output tile per 256-thread workgroup, 2x2 waves, each wave 2x2 MFMA
// 32x32 tiles.  Operand tiles are staged global -> registers -> LDS with the next
// tile's global loads in flight during the MFMAs (register-staged pipeline).
// LDS rows hold k contiguously with a 16-byte pad: 144-byte row stride makes the
// ds_read_b128 fragment reads bank-conflict free (MI355X_MICROARCH LDS table).
#include "dj_kernels.h"

namespace {

template <typename T> struct Vec16 { uint4 v; };

template <typename T> struct GemmCfg {
  static constexpr int EPL = 16 / sizeof(T);   // elements per 16-byte lane vector
  static constexpr int BK = 8 * EPL;           // k elements per LDS tile (f32: 32, bf16: 64)
  static constexpr int LDT = BK + EPL;         // padded LDS row stride in elements (144 B)
  static constexpr int KC = 2 * EPL;           // k per MFMA chunk
  static constexpr int NCH = BK / KC;          // chunks per tile (4)
};

__device__ __forceinline__ uint4 ldg16(const void* p) { return *(const uint4*)p; }

template <typename T>
__device__ __forceinline__ void mma_tile(const T* As, const T* Bs, f32x16 (&acc)[2][2], int wr, int wc, int lane) {
  using G = GemmCfg<T>;
  using Frag = typename DjFrag<T>::type;
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int kc = 0; kc < G::NCH; ++kc) {
    Frag a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i] = dj_lds_frag(As + (wr * 64 + i * 32 + l31) * G::LDT + kc * G::KC, h);
      b[i] = dj_lds_frag(Bs + (wc * 64 + i * 32 + l31) * G::LDT + kc * G::KC, h);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) dj_mfma(acc[i][j], a[i], b[j]);
  }
}

// ------------------------------------------------------------------ NT
// Row-block stride (a_rbs / c_rbs, in 32-row blocks; 1 = dense): virtual row v lives at physical
// row ((v >> 5) * rbs) * 32 + (v & 31).  With rbs = steps and the base pointer advanced to step t
// this addresses "all sequences at step t" of a sequence-tiled activation buffer (dj_common.h
// dj_row), which is what the per-step recurrent GEMMs of the generic-H LSTM path multiply.
__device__ __forceinline__ int64_t rbs_row(int v, int rbs) { return (((int64_t)(v >> 5) * rbs) << 5) + (v & 31); }

template <typename T, typename TC>
__global__ __launch_bounds__(256) void gemm_nt_kernel(int M, int N, int K, const T* __restrict__ A, int lda,
                                                      const T* __restrict__ Bt, int ldb, TC* __restrict__ C, int ldc,
                                                      const float* __restrict__ bias, int ntn, int c_frag, int a_rbs,
                                                      int c_rbs) {
  using G = GemmCfg<T>;
  __shared__ __attribute__((aligned(16))) T As[128 * G::LDT];
  __shared__ __attribute__((aligned(16))) T Bs[128 * G::LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
  const int n0 = (blockIdx.x % ntn) * 128;
  const int m0 = (blockIdx.x / ntn) * 128;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int vid = tid + 256 * i;
      int row = vid >> 3, kv = (vid & 7) * G::EPL;
      uint4 z = make_uint4(0, 0, 0, 0);
      bool kin = (k0 + kv) < K;
      ra[i] = (kin && (m0 + row) < M) ? ldg16(A + rbs_row(m0 + row, a_rbs) * lda + k0 + kv) : z;
      rb[i] = (kin && (n0 + row) < N) ? ldg16(Bt + (int64_t)(n0 + row) * ldb + k0 + kv) : z;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int vid = tid + 256 * i;
      int row = vid >> 3, kv = (vid & 7) * G::EPL;
      *(uint4*)(As + row * G::LDT + kv) = ra[i];
      *(uint4*)(Bs + row * G::LDT + kv) = rb[i];
    }
  };

  const int nk = (K + G::BK - 1) / G::BK;
  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * G::BK);
    mma_tile<T>(As, Bs, acc, wr, wc, lane);
    __syncthreads();
    if (kt + 1 < nk) {
      lstore();
      __syncthreads();
    }
  }

  const int l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int col = n0 + wc * 64 + j * 32 + l31;
      if (col >= N) continue;
      float bv = bias ? bias[col] : 0.f;
      if (c_frag) {
        // fragment-tiled output for the recurrent kernels (dj_lstm.hip): block (rb, cb) as [lane][16]
        int rowb = m0 + wr * 64 + i * 32;
        if (rowb < M) {
          float x[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) x[r] = acc[i][j][r] + bv;
          int64_t rb = rowb >> 5, cb = (n0 + wc * 64 + j * 32) >> 5;
          store_frag(C + ((rb * (N >> 5) + cb) * 64 + lane) * 16, x);
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = m0 + wr * 64 + i * 32 + dj_crow(r, lane);
        if (row < M) C[rbs_row(row, c_rbs) * ldc + col] = dj_from_f32<TC>(acc[i][j][r] + bv);
      }
    }
}

// ------------------------------------------------------------------ TN
// A row remap for the recurrent-weight gradient: a_shift = 32 makes row m read
// row m-32 (the previous step of the same sequence tile), zeros at step 0.
__device__ __forceinline__ bool tn_a_row(int64_t m, int a_shift, int steps, int64_t& src) {
  if (a_shift == 0) {
    src = m;
    return true;
  }
  if (((m >> 5) % steps) == 0) return false;
  src = m - a_shift;
  return true;
}

// fp32: LDS tiles are k-major [BK][128] (no transpose on the way in); a fragment is
// four conflict-free ds_read_b32.
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(int64_t M, int Ka, int N, const float* __restrict__ A, int lda,
                                                          const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                          int ldc, int ntn, int ntiles, int64_t rows_per_split,
                                                          int a_shift, int steps, int ka_valid) {
  constexpr int BK = 32, LD = 128;
  __shared__ __attribute__((aligned(16))) float As[BK * LD];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int n0 = (tile % ntn) * 128, i0 = (tile / ntn) * 128;
  const int64_t ms = (int64_t)split * rows_per_split;
  int64_t me = ms + rows_per_split;
  if (me > M) me = M;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[4];
  auto gload = [&](int64_t mt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int vid = tid + 256 * i;
      int kr = vid >> 5, cv = (vid & 31) * 4;
      int64_t m = mt + kr, src;
      uint4 z = make_uint4(0, 0, 0, 0);
      bool min = m < me;
      ra[i] = (min && (i0 + cv) < Ka && tn_a_row(m, a_shift, steps, src)) ? ldg16(A + src * lda + i0 + cv) : z;
      rb[i] = (min && (n0 + cv) < N) ? ldg16(B + m * ldb + n0 + cv) : z;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int vid = tid + 256 * i;
      int kr = vid >> 5, cv = (vid & 31) * 4;
      *(uint4*)(As + kr * LD + cv) = ra[i];
      *(uint4*)(Bs + kr * LD + cv) = rb[i];
    }
  };
  const int h = lane >> 5, l31 = lane & 31;
  if (ms < me) {
    gload(ms);
    lstore();
    __syncthreads();
    for (int64_t mt = ms; mt < me; mt += BK) {
      bool more = (mt + BK) < me;
      if (more) gload(mt + BK);
#pragma unroll
      for (int kc = 0; kc < BK / 8; ++kc) {
        f32x4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int k = kc * 8 + 4 * h + e;
            a[i][e] = As[k * LD + wr * 64 + i * 32 + l31];
            b[i][e] = Bs[k * LD + wc * 64 + i * 32 + l31];
          }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) dj_mfma(acc[i][j], a[i], b[j]);
      }
      __syncthreads();
      if (more) {
        lstore();
        __syncthreads();
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int col = n0 + wc * 64 + j * 32 + l31;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = i0 + wr * 64 + i * 32 + dj_crow(r, lane);
        if (row < ka_valid) atomicAdd(C + (int64_t)row * ldc + col, acc[i][j][r]);
      }
    }
}

// bf16: each thread transposes an 8(k) x 8(col) block in registers so the LDS tiles
// are k-contiguous and fragments are single ds_read_b128.
__device__ __forceinline__ void transpose8x8_b16(const uint4 (&r)[8], uint4 (&o)[8]) {
  const uint32_t* rin = (const uint32_t*)&r[0];   // rin[row*4 + d]
  uint32_t* out = (uint32_t*)&o[0];               // out[col*4 + q]
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t lo = rin[(2 * q) * 4 + (c >> 1)], hi = rin[(2 * q + 1) * 4 + (c >> 1)];
      out[c * 4 + q] = (c & 1) ? ((lo >> 16) | (hi & 0xFFFF0000u)) : ((lo & 0xFFFFu) | (hi << 16));
    }
}

__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(int64_t M, int Ka, int N, const bf16_t* __restrict__ A,
                                                           int lda, const bf16_t* __restrict__ B, int ldb,
                                                           float* __restrict__ C, int ldc, int ntn, int ntiles,
                                                           int64_t rows_per_split, int a_shift, int steps,
                                                           int ka_valid) {
  using G = GemmCfg<bf16_t>;   // BK = 64 rows of M per tile
  __shared__ __attribute__((aligned(16))) bf16_t As[128 * G::LDT];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[128 * G::LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int n0 = (tile % ntn) * 128, i0 = (tile / ntn) * 128;
  const int64_t ms = (int64_t)split * rows_per_split;
  int64_t me = ms + rows_per_split;
  if (me > M) me = M;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // threads 0..127 stage A blocks, 128..255 stage B blocks; block = 8 k-rows x 8 cols
  const bool isA = tid < 128;
  const int q = tid & 127, kb = q >> 4, cb = q & 15;
  uint4 rg[8];
  auto gload = [&](int64_t mt) {
    const int c0 = (isA ? i0 : n0) + cb * 8;
    const bool cin = c0 < (isA ? Ka : N);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      int64_t m = mt + kb * 8 + r, src = m;
      bool ok = cin && m < me;
      if (isA) ok = ok && tn_a_row(m, a_shift, steps, src);
      const bf16_t* p = isA ? (A + src * lda + c0) : (B + m * ldb + c0);
      rg[r] = ok ? ldg16(p) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&]() {
    uint4 o[8];
    transpose8x8_b16(rg, o);
    bf16_t* dst = isA ? As : Bs;
#pragma unroll
    for (int c = 0; c < 8; ++c) *(uint4*)(dst + (cb * 8 + c) * G::LDT + kb * 8) = o[c];
  };

  if (ms < me) {
    gload(ms);
    lstore();
    __syncthreads();
    for (int64_t mt = ms; mt < me; mt += G::BK) {
      bool more = (mt + G::BK) < me;
      if (more) gload(mt + G::BK);
      mma_tile<bf16_t>(As, Bs, acc, wr, wc, lane);
      __syncthreads();
      if (more) {
        lstore();
        __syncthreads();
      }
    }
  }
  const int l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int col = n0 + wc * 64 + j * 32 + l31;
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = i0 + wr * 64 + i * 32 + dj_crow(r, lane);
        if (row < ka_valid) atomicAdd(C + (int64_t)row * ldc + col, acc[i][j][r]);
      }
    }
}


// ------------------------------------------------------------------ TN, bf16, DMA ring + transposed LDS reads
// 128 (Ka) x 256 (N) output tile, BK = 32 rows of M per stage, 3-stage LDS ring filled by
// global_load_lds (no registers, no VALU on the way in; tiles stay in their natural k-major
// [row][col] form) and consumed with ds_read_b64_tr_b16, the gfx950 transposing LDS read
// that delivers, per lane, 4 consecutive k of one column -- exactly an MFMA operand half.
// Each 16-byte chunk c of LDS row r is stored at chunk c ^ ((r&3)<<2) (applied on the DMA
// SOURCE address, cdna_hip_programming.md rule 21) so the 4-row x 64-byte footprint of a
// transposed read covers all 64 banks once.
typedef short s16x4 __attribute__((ext_vector_type(4)));
// LDS-DMA of 16 B per lane: LDS destination = lds_base (wave-uniform byte address, via M0) +
// 16*lane; the global source address is per lane.  Issued from inline asm so that hipcc does
// not drain it with vmcnt(0) before every LDS read -- completion is tracked by our own counted
// s_waitcnt + barrier (cdna_hip_programming.md 5.7: M0 written in the statement that reads it).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)p;
}
union TrFrag {
  s16x4 h[2];
  bf16x8 v;
};
constexpr int TN2_BK = 32, TN2_TA = 128, TN2_TB = 256;
constexpr int TN2_ABYTES = TN2_BK * TN2_TA * 2, TN2_BBYTES = TN2_BK * TN2_TB * 2, TN2_STAGE = TN2_ABYTES + TN2_BBYTES;
constexpr int TN2_NS = 3;

__global__ __launch_bounds__(256) void gemm_tn_bf16_dma_kernel(int64_t M, int Ka, int ka_valid, int N,
                                                               const bf16_t* __restrict__ A, int lda,
                                                               const bf16_t* __restrict__ B, int ldb,
                                                               float* __restrict__ C, int ldc, int ntn, int ntiles,
                                                               int64_t rows_per_split, int a_shift, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
  const int tile = blockIdx.x % ntiles, split = blockIdx.x / ntiles;
  const int n0 = (tile % ntn) * TN2_TB, i0 = (tile / ntn) * TN2_TA;
  const int64_t ms = (int64_t)split * rows_per_split;
  int64_t me = ms + rows_per_split;
  if (me > M) me = M;
  const int nkt = (int)((me - ms) / TN2_BK);

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- DMA: per stage 8 x 1 KiB pieces of A (4 rows each) and 16 of B (2 rows each)
  auto issue = [&](int kt) {
    const int64_t mt = ms + (int64_t)kt * TN2_BK;
    unsigned char* sa = smem + (kt % TN2_NS) * TN2_STAGE;
    unsigned char* sb = sa + TN2_ABYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = w * 2 + i, row = piece * 4 + (lane >> 4), cp = lane & 15;
      const int c = cp ^ ((row & 3) << 2);
      int64_t m = mt + row;
      if (a_shift) m = m >= a_shift ? m - a_shift : 0;
      int col = i0 + c * 8;
      if (col >= lda) col = 0;   // outside the operand: any in-bounds address (those output rows are dropped)
      glds16(A + m * lda + col, __builtin_amdgcn_readfirstlane(lds_addr(sa) + piece * 1024));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = w * 4 + i, row = piece * 2 + (lane >> 5), cp = lane & 31;
      const int c = cp ^ ((row & 3) << 2);
      glds16(B + (mt + row) * ldb + n0 + c * 8, __builtin_amdgcn_readfirstlane(lds_addr(sb) + piece * 1024));
    }
  };

  // ---- per-lane constants of the transposed reads
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, hgrp = g >> 1, colgrp = g & 1;
  const int rowl = 8 * hgrp + q;                       // + 16*ks + 4*rd
  const int chl = 2 * colgrp + (p >> 1);               // + tile column base / 8
  const int sub = (p & 1) * 8;
  auto compute = [&](int kt) {
    const unsigned char* sa = smem + (kt % TN2_NS) * TN2_STAGE;
    const unsigned char* sb = sa + TN2_ABYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      TrFrag a[2], b[4];
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        const int row = 16 * ks + 4 * rd + rowl;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int c = ((wr * 64 + mi * 32) >> 3) + chl;
          a[mi].h[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(sa + row * (TN2_TA * 2) + ((c ^ (q << 2)) << 4) + sub));
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int c = ((wc * 128 + ni * 32) >> 3) + chl;
          b[ni].h[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(sb + row * (TN2_TB * 2) + ((c ^ (q << 2)) << 4) + sub));
        }
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) dj_mfma(acc[mi][ni], a[mi].v, b[ni].v);
    }
  };

  if (nkt > 0) issue(0);
  if (nkt > 1) issue(1);
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // stage kt landed; stage kt+1 (6 DMAs) may still fly
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nkt) issue(kt + 2);
    bool skip = false;
    if (a_shift) skip = (((ms + (int64_t)kt * TN2_BK) >> 5) % steps) == 0;   // h_{-1} = 0: no contribution
    if (!skip) compute(kt);
  }

  const int l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wc * 128 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wr * 64 + i * 32 + dj_crow(r, lane);
        if (row < ka_valid) atomicAdd(C + (int64_t)row * ldc + col, acc[i][j][r]);
      }
    }
}


// ------------------------------------------------------------------ TN, bf16, small output (conv kernel gradient)
// C[Ka <= 96, N <= 64] += A[M, Ka]^T B[M, N] with M in the millions (the octave-conv weight gradient: 80 x 64 from
// 1 M rows): the whole output is ONE 96 x 64 accumulator tile per WAVE, the reduction over rows is split over all waves
// of the grid.  A wave stages 32 rows of both operands in its own LDS tile (16-byte loads, the next block's loads in
// flight under the products), reads them with ds_read_b64_tr_b16 and multiplies with 32x32x16 MFMAs; the four waves
// of a workgroup are folded through LDS, then one run of global atomics per workgroup.  (The 128 x 128 tile kernel
// above spent 0.22 ms on this 0.3 GB.)
constexpr int TNS_LDA = 104, TNS_LDB = 72;          // LDS row strides (elements): 208 / 144 bytes
__global__ __launch_bounds__(256) void gemm_tn_small_bf16_kernel(int64_t M, int Ka, int ka_valid, int N,
                                                                 const bf16_t* __restrict__ A, int lda,
                                                                 const bf16_t* __restrict__ B, int ldb,
                                                                 float* __restrict__ C, int ldc, int64_t rows_per_wave) {
  __shared__ __attribute__((aligned(16))) bf16_t tiles[4][32 * (TNS_LDA + TNS_LDB)];
  static_assert(sizeof(tiles) >= 96 * 64 * sizeof(float), "the fold buffer reuses the tiles");
  float* fold = (float*)&tiles[0][0];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31;
  bf16_t* ta = tiles[w];
  bf16_t* tb = ta + 32 * TNS_LDA;
  const int64_t gw = (int64_t)blockIdx.x * 4 + w;
  const int64_t m0 = gw * rows_per_wave;
  int64_t m1 = m0 + rows_per_wave;
  if (m1 > M) m1 = M;
  const int ca = Ka >> 3, cb = N >> 3;                 // 16-byte chunks per row
  const int na = 32 * ca, nb = 32 * cb;                // chunks per 32-row block (<= 384 / 256)
  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // zero the padding columns of the tiles once (columns Ka..95 of A, N..63 of B are read by the fragment loads)
  for (int i = lane; i < 32 * (TNS_LDA + TNS_LDB) / 8; i += 64) ((uint4*)ta)[i] = make_uint4(0, 0, 0, 0);
  __builtin_amdgcn_wave_barrier();
  uint4 ra[6], rb[4];
  auto fetch = [&](int64_t mb) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = lane + 64 * u;
      ra[u] = make_uint4(0, 0, 0, 0);
      if (i < na) {
        const int row = i / ca, c = i - row * ca;
        if (mb + row < m1) ra[u] = *(const uint4*)(A + (mb + row) * lda + c * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = lane + 64 * u;
      rb[u] = make_uint4(0, 0, 0, 0);
      if (i < nb) {
        const int row = i / cb, c = i - row * cb;
        if (mb + row < m1) rb[u] = *(const uint4*)(B + (mb + row) * ldb + c * 8);
      }
    }
  };
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pq = li & 3, hgrp = g >> 1, colgrp = g & 1;
  const int rowl = 8 * hgrp + q, coll = 16 * colgrp + 4 * pq;
  if (m0 < m1) fetch(m0);
  for (int64_t mb = m0; mb < m1; mb += 32) {
    // registers -> this wave's tile
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = lane + 64 * u;
      if (i < na) {
        const int row = i / ca, c = i - row * ca;
        *(uint4*)(ta + row * TNS_LDA + c * 8) = ra[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = lane + 64 * u;
      if (i < nb) {
        const int row = i / cb, c = i - row * cb;
        *(uint4*)(tb + row * TNS_LDB + c * 8) = rb[u];
      }
    }
    if (mb + 32 < m1) fetch(mb + 32);                  // next block: in flight under the products
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      TrFrag fa[3], fb[2];
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
        const int row = 16 * ks + 4 * rd + rowl;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          fa[i].h[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(ta + row * TNS_LDA + i * 32 + coll));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          fb[j].h[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (s16x4 __attribute__((address_space(3)))*)(tb + row * TNS_LDB + j * 32 + coll));
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dj_mfma(acc[i][j], fa[i].v, fb[j].v);
    }
    __builtin_amdgcn_wave_barrier();                   // the tile is rewritten in the next round
  }
  // fold the four waves (over the tiles, which every wave has left), then one run of atomics per workgroup
  __syncthreads();
  for (int i = tid; i < 96 * 64; i += 256) fold[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(&fold[(i * 32 + dj_crow(r, lane)) * 64 + j * 32 + l31], acc[i][j][r]);
  __syncthreads();
  for (int i = tid; i < ka_valid * N; i += 256) {
    const int row = i / N, col = i - row * N;
    atomicAdd(C + (int64_t)row * ldc + col, fold[row * 64 + col]);
  }
}

// ------------------------------------------------------------------ fused LSTM weight gradient (bf16)
// dW = X^T dZ and dU = Hprev^T dZ of one layer in ONE pass over dZ: the A operand is the
// virtual matrix [X (DP cols) | Hprev (H cols)] (Hprev = H one recurrence step earlier, zero
// rows at step 0 -> sourced from a zero line), cut into 256-column tiles.  256 x 256 output
// tile per 512-thread workgroup (8 waves as 4 x 2, each 64 x 128), BK = 32, 4-stage LDS ring
// (128 KiB) filled by LDS-DMA, operands consumed with ds_read_b64_tr_b16 (see the kernel above).
struct WgradArgs {
  int64_t M;
  const bf16_t* X; int DP, D;        // layer input [M, DP], D valid columns -> dW [D, N]
  const bf16_t* Hs; int H;           // layer output [M, H] (row-major), shifted by 32 rows -> dU [H, N]
  int steps;
  const bf16_t* dZ; int N;           // [M, N]: element (m, k) at dZ + (k >> 8) * dz_cts + m * ldz + (k & 255)
  int64_t dz_cts; int ldz;           //   row-major: (256, N); column-tile-major [N/256][M][256]: (M * 256, 256)
  float* dW; float* dU;              // fp32, += (atomics)
  const bf16_t* zeros;               // >= 16 zero bytes
  int ntn, ntiles, xcd_map;
  int64_t rows_per_split;
};
constexpr int WG_BK = 32, WG_T = 256, WG_TILE_BYTES = WG_BK * WG_T * 2, WG_STAGE = 2 * WG_TILE_BYTES, WG_NS = 4;
static_assert((WG_NS & (WG_NS - 1)) == 0, "ring slot counter wraps by masking");

__global__ __launch_bounds__(512) void lstm_wgrad_bf16_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
  // all tiles of one row split read the same rows of dZ / X / H: keep them on one XCD (blocks b, b+8, ...)
  int tile, split;
  if (a.xcd_map) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    tile = idx % a.ntiles;
    split = (idx / a.ntiles) * 8 + xcd;
  } else {
    tile = blockIdx.x % a.ntiles;
    split = blockIdx.x / a.ntiles;
  }
  const int n0 = (tile % a.ntn) * WG_T, v0 = (tile / a.ntn) * WG_T;   // v0: first virtual A column of the tile
  const int64_t ms = (int64_t)split * a.rows_per_split;
  int64_t me = ms + a.rows_per_split;
  if (me > a.M) me = a.M;
  const int nkt = (int)((me - ms) / WG_BK);
  if (nkt <= 0) return;                    // empty trailing split (uniform for the workgroup)

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this lane's DMA duties per stage: 2 pieces of A and 2 of B (1 KiB = 2 rows of 512 B each).  Which operand a
  // lane feeds (X, Hprev or the zero line) and where never changes, so the source pointers are set up once and
  // advanced by one stage (32 rows) per issue; the recurrence step of a stage is a counter.  (Recomputing them per
  // stage -- a 64-bit scalar modulo for the step among other things -- cost ~220 scalar instructions per stage and
  // wave: at one instruction per wave and 4 cycles that is as long as the stage's 16 MFMAs.)
  const bf16_t* pa[2];
  const bf16_t* pb[2];
  int64_t a_stride[2];
  bool a_is_h[2];
  unsigned piece_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = w * 2 + i, row = piece * 2 + (lane >> 5), cp = lane & 31;
    const int c = cp ^ ((row & 3) << 2);
    const int vcol = v0 + c * 8;
    const int64_t m = ms + row;
    piece_off[i] = (unsigned)piece * 1024u;
    a_is_h[i] = false;
    if (vcol < a.DP) {
      pa[i] = a.X + m * a.DP + vcol;
      a_stride[i] = (int64_t)WG_BK * a.DP;
    } else if (vcol < a.DP + a.H) {
      pa[i] = a.Hs + (m - 32) * a.H + (vcol - a.DP);
      a_stride[i] = (int64_t)WG_BK * a.H;
      a_is_h[i] = true;
    } else {
      pa[i] = a.zeros;
      a_stride[i] = 0;
    }
    pb[i] = a.dZ + (int64_t)(n0 >> 8) * a.dz_cts + m * a.ldz + c * 8;
  }
  const int64_t b_stride = (int64_t)WG_BK * a.ldz;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  int sidx = (int)((ms >> 5) % a.steps), islot = 0;      // recurrence step / ring slot of the next stage to issue
  // one A piece + one B piece of the next stage (i = 0, 1: the stage's two halves)
  auto dma_pair = [&](int i) {
    const bool step0 = sidx == 0;                        // h_{-1} = 0: the Hprev rows of this stage are the zero line
    const unsigned sa = lds0 + (unsigned)islot * WG_STAGE, sb = sa + WG_TILE_BYTES;
    glds16((a_is_h[i] && step0) ? a.zeros : pa[i], __builtin_amdgcn_readfirstlane(sa + piece_off[i]));
    glds16(pb[i], __builtin_amdgcn_readfirstlane(sb + piece_off[i]));
    pa[i] += a_stride[i];
    pb[i] += b_stride;
  };
  auto dma_advance = [&]() {
    islot = (islot + 1) & (WG_NS - 1);
    if (++sidx == a.steps) sidx = 0;
  };
  auto issue = [&](int) {
    dma_pair(0);
    dma_pair(1);
    dma_advance();
  };

  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, hgrp = g >> 1, colgrp = g & 1;
  const int rowl = 8 * hgrp + q, chl = 2 * colgrp + (p >> 1), sub = (p & 1) * 8;
  // operand fragments of one half stage (16 of the 32 k-rows): 2 A blocks + 4 B blocks, transposed LDS reads
  struct Half { TrFrag a[2], b[4]; };
  auto read_half = [&](int kt, int ks) {
    Half f;
    const unsigned char* sa = smem + (kt % WG_NS) * WG_STAGE;
    const unsigned char* sb = sa + WG_TILE_BYTES;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int row = 16 * ks + 4 * rd + rowl;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int c = ((wr * 64 + mi * 32) >> 3) + chl;
        f.a[mi].h[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(sa + row * (WG_T * 2) + ((c ^ (q << 2)) << 4) + sub));
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int c = ((wc * 128 + ni * 32) >> 3) + chl;
        f.b[ni].h[rd] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(sb + row * (WG_T * 2) + ((c ^ (q << 2)) << 4) + sub));
      }
    }
    return f;
  };
  // 8 MFMAs of one half stage with one DMA pair (an A piece + a B piece of a later stage) issued between them: a
  // global_load_lds costs its wave ~120 cycles of issue time, under the MFMAs it is free
  auto mma_half = [&](const Half& f, bool dma, int pair) {
    __builtin_amdgcn_sched_barrier(0);
    dj_mfma(acc[0][0], f.a[0].v, f.b[0].v);
    dj_mfma(acc[0][1], f.a[0].v, f.b[1].v);
    __builtin_amdgcn_sched_barrier(0);
    if (dma) dma_pair(pair);
    __builtin_amdgcn_sched_barrier(0);
    dj_mfma(acc[0][2], f.a[0].v, f.b[2].v);
    dj_mfma(acc[0][3], f.a[0].v, f.b[3].v);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) dj_mfma(acc[1][ni], f.a[1].v, f.b[ni].v);
    __builtin_amdgcn_sched_barrier(0);
  };

  // Software pipeline at half-stage granularity (cycle stamps of the plain loop -- every wave reading, then every
  // wave multiplying, behind one barrier per stage, the four DMA instructions in a block of their own -- showed the
  // MFMA pipe busy for 1.0 k of a stage's 2.2 k cycles): the LDS reads of one half stage fly under the MFMAs of the
  // previous half, and the DMA of a later stage is issued between those MFMAs.  All four ring slots are filled before
  // the loop; slot kt % 4 is refilled with stage kt+4 as soon as every wave holds its last fragment of stage kt.
  //   stage kt:  read(kt, half 1) | MFMA(half 0) + DMA pair 1 of stage kt+3 | stage kt+1 landed? (counted vmcnt) |
  //              own reads of slot kt done (lgkmcnt 0) | barrier | read(kt+1, half 0) | MFMA(half 1) + DMA pair 0 of
  //              stage kt+4
  // RAW: a stage is read only behind the counted wait that retires its DMA AND the barrier after it.  WAR: a slot is
  // refilled only behind the barrier every wave passes after its lgkmcnt(0).  The counted wait: the DMAs younger than
  // stage kt+1 are those of stages kt+2 and kt+3 (4 per stage and wave, both pairs of kt+3 issued by then).
#pragma unroll
  for (int s = 0; s < WG_NS; ++s)
    if (s < nkt) issue(s);
  if (nkt >= 4)
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (nkt == 3)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nkt == 2)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  Half f0 = read_half(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    Half f1 = read_half(kt, 1);
    const bool second = kt >= 1 && kt + 3 < nkt;       // second pair of stage kt+3 (its first went out last iteration)
    mma_half(f0, second, 1);
    if (second) dma_advance();
    const int rem = nkt - 1 - kt;                      // stages after kt
    if (rem >= 3)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (rem == 2)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the builtin, not inline asm: hipcc then knows that f1 has arrived and does not make MFMA(f1) wait for the
    // reads of the next half stage (its own lgkmcnt bookkeeping saturates at 15 outstanding operations)
    __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
    if (rem >= 1) f0 = read_half(kt + 1, 0);
    mma_half(f1, kt + 4 < nkt, 0);                     // first pair of stage kt+4 into the slot just vacated
  }

  const int l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int vr = v0 + wr * 64 + i * 32 + dj_crow(r, lane);
      float* dst = nullptr;
      if (vr < a.D)
        dst = a.dW + (int64_t)vr * a.N;
      else if (vr >= a.DP && vr < a.DP + a.H)
        dst = a.dU + (int64_t)(vr - a.DP) * a.N;
      if (!dst) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(dst + n0 + wc * 128 + j * 32 + l31, acc[i][j][r]);
    }
}


// ------------------------------------------------------------------ NT, bf16, persistent DMA ring
// C[M,N] = A[M,K] Bt[N,K]^T (+bias).  256 x 128 output tile per 512-thread workgroup (8 waves
// as 4 x 2, each 64 x 64), BK = 64, 3-stage LDS ring filled by LDS-DMA.  The K loops of this
// model are short (K = 96..1024), so every workgroup is persistent over output tiles and the
// ring runs ahead ACROSS tile boundaries: the epilogue of tile i overlaps the loads of tile i+1.
// LDS rows are 128 B (64 bf16); 16-byte chunk c of row r lives at chunk c ^ ((r>>1)&7), applied
// on the DMA source address, which makes the ds_read_b128 operand reads conflict-free.
// XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so the ntn column
// tiles of one 256-row A panel are given to `ntn` workgroups of ONE XCD at the same time -- the
// panel is fetched from HBM once and re-read from that XCD's L2 (speed only, never correctness).
__device__ uint4 dj_zero_line[4];     // 64 zero bytes: DMA source for out-of-range rows / k-tail
constexpr int NT2_BM = 256, NT2_BN = 128, NT2_BK = 64, NT2_NS = 3;
constexpr int NT2_ABYTES = NT2_BM * NT2_BK * 2, NT2_BBYTES = NT2_BN * NT2_BK * 2, NT2_STAGE = NT2_ABYTES + NT2_BBYTES;

// Row-major output (CFRAG = false) multiplies with the operands swapped, so a lane's accumulator
// block holds C^T: 4 consecutive output columns of ONE row per register quad, stored as one 8-byte
// (bf16) / 16-byte (f32) vector instead of four scalar stores.
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
  *(uint2*)p = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *(float4*)p = make_float4(a, b, c, d);
}

// the 4 values store4 would overwrite (accumulating epilogue, c_mode 3)
__device__ __forceinline__ void load4(const bf16_t* p, float* x) {
  const uint2 v = *(const uint2*)p;
  x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xFFFF0000u);
  x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xFFFF0000u);
}
__device__ __forceinline__ void load4(const float* p, float* x) {
  const float4 v = *(const float4*)p;
  x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
}

// v_permlane32_swap: lanes 32..63 of `a` trade places with lanes 0..31 of `b` (gfx950).  After it a lane of the lower half
// holds (own a, the upper partner's a) and a lane of the upper half (the lower partner's b, own b).
__device__ __forceinline__ void half_swap(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
// Row-major bf16 store of one 32 x 32 block of the C^T accumulator.  Lane (l31, h) holds row l31, columns 8 g + 4 h + e: four
// 8-byte pieces, 32 bytes apart.  A store instruction costs the memory pipe per LANE, not per byte (section 8 of DESIGN.md:
// a wave's scattered 8-byte store took as long as a coalesced 16-byte one), so the two halves of the wave first trade
// pieces -- the lower half keeps g = 0, 1 and receives the partner's, the upper half g = 2, 3 -- and every lane then owns
// columns [16 h, 16 h + 16) of its row: two 16-byte stores instead of four 8-byte ones.  All 64 lanes must call (the
// swap); `ok` guards the stores.  `p` = this lane's row + the block's first column, 16-byte aligned.
__device__ __forceinline__ void store_block_rows16(bf16_t* p, const f32x16& a, const float* __restrict__ bias_blk, int h,
                                                   bool ok) {
  uint4* q = (uint4*)(p + 16 * h);
#pragma unroll
  for (int k = 0; k < 2; ++k) {        // pieces g = k and g = k + 2 -> the lane's columns [16 h + 8 k, 16 h + 8 k + 8)
    uint2 r[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int g = k + 2 * j;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias_blk) bv = *(const float4*)(bias_blk + 8 * g + 4 * h);
      r[j] = make_uint2(pack_bf16x2(a[4 * g] + bv.x, a[4 * g + 1] + bv.y), pack_bf16x2(a[4 * g + 2] + bv.z, a[4 * g + 3] + bv.w));
    }
    half_swap(r[0].x, r[1].x);
    half_swap(r[0].y, r[1].y);
    if (ok) q[k] = make_uint4(r[0].x, r[0].y, r[1].x, r[1].y);
  }
}

// ---- forward LSTM cell as the epilogue of a step's GEMM z_t = [x_t | h_{t-1}] [W ; U] (dj_kernels.h CellEpi; scaled
// model).  One 32 x 32 block of the C^T accumulator: lane (l31, h) holds virtual row `row` = rowb + l31, registers
// 4 q + e <-> output column colb + 8 q + 4 h + e = gate q of unit 8 (colb / 32) + 4 h + e (gate-interleaved B rows).  The
// epilogue LOADS only the carry (16 bytes per lane, coalesced), and of its stores only h_t (row-major: the next step's A
// operand) is scattered: the cell-state stash and the gate stash -- the ACTIVATED gates as 8-bit codes (dj_common.h), as
// the persistent bf16 kernels keep them -- go to the accumulators' own fragment layout, one coalesced 8- / 16-byte store
// per lane and block.  The forward values themselves are not quantised.

template <bool SIGM>
__device__ __forceinline__ uint2 cell_fwd_block(const f32x16& a, const CellEpi& ce, const float* __restrict__ bias, int row,
                                               int colb, int lane, int h) {
  const int H = ce.H, G = colb >> 5, u0 = 8 * G + 4 * h;
  const int64_t pr = rbs_row(row, ce.steps);
  float z[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 bv = *(const float4*)(bias + g * H + u0);
    z[g][0] = a[4 * g] + bv.x; z[g][1] = a[4 * g + 1] + bv.y; z[g][2] = a[4 * g + 2] + bv.z; z[g][3] = a[4 * g + 3] + bv.w;
  }
  // this lane's slot in the FRAGMENT layout [row block][unit group][64 lanes]: carry (16 B), c stash (8 B), gate codes (16 B)
  const int64_t fb = ((int64_t)(row >> 5) * (H >> 3) + G) * 64 + lane;
  const int64_t fs = ((int64_t)(row >> 5) * ce.steps * (H >> 3) + G) * 64 + lane;     // the stashes hold every step
  float4* cp = (float4*)ce.carry + fb;
  float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!ce.first) cv = *cp;                                    // step 0 starts from c = 0
  float c[4] = {cv.x, cv.y, cv.z, cv.w}, hn[4];
  uint32_t code[4] = {0u, 0u, 0u, 0u};                        // word g: the four units' codes of gate g
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ig = dj_ract<SIGM>(z[0][e]), fg = dj_ract<SIGM>(z[1][e]), gg = dj_tanh(z[2][e]), og = dj_ract<SIGM>(z[3][e]);
    c[e] = fg * c[e] + ig * gg;
    hn[e] = og * dj_tanh(c[e]);
    code[0] = __builtin_amdgcn_cvt_pk_u8_f32(dj_gate_code01<SIGM>(z[0][e], ig), e, code[0]);
    code[1] = __builtin_amdgcn_cvt_pk_u8_f32(dj_gate_code01<SIGM>(z[1][e], fg), e, code[1]);
    code[2] = __builtin_amdgcn_cvt_pk_u8_f32(dj_gate_code_g(gg), e, code[2]);
    code[3] = __builtin_amdgcn_cvt_pk_u8_f32(dj_gate_code01<SIGM>(z[3][e], og), e, code[3]);
  }
  *cp = make_float4(c[0], c[1], c[2], c[3]);
  if (ce.Cs) {      // training: what BPTT reads back -- c_t (bf16) and the ACTIVATED gates as 8-bit codes, both coalesced
    *(uint2*)((bf16_t*)ce.Cs + fs * 4) = make_uint2(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3]));
    *(uint4*)((uint8_t*)ce.Z + fs * 16) = make_uint4(code[0], code[1], code[2], code[3]);
  }
  return make_uint2(pack_bf16x2(hn[0], hn[1]), pack_bf16x2(hn[2], hn[3]));     // h_t of units u0 .. u0 + 3: stored by the caller
}

template <int EPI>
__device__ __forceinline__ uint2 cell_block(const f32x16& a, const CellEpi& ce, const float* bias, int row, int colb, int lane,
                                            int h) {
  if constexpr (EPI == 1) return cell_fwd_block<false>(a, ce, bias, row, colb, lane, h);
  if constexpr (EPI == 2) return cell_fwd_block<true>(a, ce, bias, row, colb, lane, h);
  return make_uint2(0u, 0u);
}
// The cells of a wave's two column blocks (unit groups G, G + 1) of one row block, and h_t of both: a lane of block j holds
// units 4 h .. 4 h + 3 of group G + j (8 bytes); after one swap per word the lower half of the wave owns all 8 units of
// group G and the upper half those of G + 1 -- one 16-byte store per lane into the row-major h (the next step's A operand)
template <int EPI>
__device__ __forceinline__ void cell_block_pair(const f32x16& a0, const f32x16& a1, const CellEpi& ce, const float* bias,
                                                int row, int M, int colb, int N, int lane, int h) {
  const bool rok = row < M, two = colb + 32 < N;              // `two` is wave-uniform
  uint2 h0 = make_uint2(0u, 0u), h1 = h0;
  if (rok && colb < N) h0 = cell_block<EPI>(a0, ce, bias, row, colb, lane, h);
  if (rok && two) h1 = cell_block<EPI>(a1, ce, bias, row, colb + 32, lane, h);
  bf16_t* hrow = (bf16_t*)ce.Hs + rbs_row(row, ce.steps) * ce.H + 8 * (colb >> 5);
  if (two) {
    half_swap(h0.x, h1.x);
    half_swap(h0.y, h1.y);
    if (rok) *(uint4*)(hrow + 8 * h) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  } else if (rok && colb < N) {
    *(uint2*)(hrow + 4 * h) = h0;
  }
}

template <typename TC, bool CFRAG, int EPI = 0>
__global__ __launch_bounds__(512) void gemm_nt_bf16_dma_kernel(int M, int N, int K, const bf16_t* __restrict__ A,
                                                               int lda, const bf16_t* __restrict__ Bt, int ldb,
                                                               TC* __restrict__ C, int ldc,
                                                               const float* __restrict__ bias, int ntn, int ntm,
                                                               int xcd_map, int a_rbs, int c_rbs, int64_t a_cts, int c_acc,
                                                               CellEpi ce) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
  const int nk = (K + NT2_BK - 1) / NT2_BK;
  // row-major bf16 output whose rows take 16-byte stores (store_block_rows16); not for the accumulating epilogue
  const bool vec16 = !c_acc && (ldc & 7) == 0 && (((uintptr_t)C | (uintptr_t)bias) & 15) == 0;
  // tile schedule of this persistent workgroup: tile(tl) = (mt0 + tl*mstride, nt)
  int nt, mt0, mstride, my_tiles;
  if (xcd_map) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, teams = (int)(gridDim.x >> 3) / ntn;
    const int team = slot / ntn;
    nt = slot % ntn;
    mt0 = xcd + 8 * team;
    mstride = 8 * teams;
    my_tiles = (team < teams && mt0 < ntm) ? (ntm - mt0 + mstride - 1) / mstride : 0;
  } else {
    nt = blockIdx.x % ntn;
    mt0 = blockIdx.x / ntn;
    mstride = gridDim.x / ntn;      // launcher guarantees gridDim.x % ntn == 0 in this mode
    my_tiles = mt0 < ntm ? (ntm - mt0 + mstride - 1) / mstride : 0;
  }
  const int n0 = nt * NT2_BN;
  const int nstages = my_tiles * nk;
  const bf16_t* zl = (const bf16_t*)dj_zero_line;

  // DMA stream with counters and per-tile row pointers (see gemm_nt_bf16_wide_kernel): the B rows never change (one
  // column tile per workgroup), the A rows when the m-tile does
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const int rsub = lane >> 3, cp = lane & 7;
  const bf16_t* ra[4];
  const bf16_t* ra2[4];         // cell epilogue (EPI): the k range past ce.K1p comes from a second matrix (h_{t-1})
  const bf16_t* rb[2];
  int kza[4], kzb[2];
  bool va[4], vb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (w * 2 + i) * 8 + rsub;
    kzb[i] = (cp ^ ((row >> 1) & 7)) << 3;
    vb[i] = n0 + row < N;
    rb[i] = vb[i] ? Bt + (int64_t)(n0 + row) * ldb + kzb[i] : zl;
  }
  int i_tl = 0, i_kt = 0, i_slot = 0;
  auto issue = [&](int) {
    if (i_kt == 0) {
      const int m0 = (mt0 + i_tl * mstride) * NT2_BM;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + rsub;
        kza[i] = (cp ^ ((row >> 1) & 7)) << 3;
        va[i] = m0 + row < M;
        ra[i] = va[i] ? A + rbs_row(m0 + row, a_rbs) * lda + kza[i] : zl;
        if constexpr (EPI != 0)
          ra2[i] = (va[i] && ce.A2) ? (const bf16_t*)ce.A2 + rbs_row(m0 + row, a_rbs) * ce.lda2 + kza[i] : zl;
      }
    }
    const int k0 = i_kt * NT2_BK;
    // A may be column-tile-major (a_cts != 0: element (m, k) at (k >> 8) * a_cts + m * lda + (k & 255), lda = 256)
    const int64_t ka = a_cts ? (int64_t)(k0 >> 8) * a_cts + (k0 & 255) : k0;
    const unsigned sa = lds0 + (unsigned)i_slot * NT2_STAGE, sb = sa + NT2_ABYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t* src;
      if constexpr (EPI != 0)      // [x_t | h_{t-1}]: columns [0, K1) of A (zeros up to K1p), then columns of A2
        src = k0 < ce.K1p ? ((va[i] && k0 + kza[i] < ce.K1) ? ra[i] + k0 : zl)
                          : ((va[i] && k0 + kza[i] < K) ? ra2[i] + (k0 - ce.K1p) : zl);
      else
        src = (va[i] && k0 + kza[i] < K) ? ra[i] + ka : zl;
      glds16(src, __builtin_amdgcn_readfirstlane(sa + (unsigned)(w * 4 + i) * 1024u));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      glds16((vb[i] && k0 + kzb[i] < K) ? rb[i] + k0 : zl, __builtin_amdgcn_readfirstlane(sb + (unsigned)(w * 2 + i) * 1024u));
    if (++i_slot == NT2_NS) i_slot = 0;
    if (++i_kt == nk) {
      i_kt = 0;
      ++i_tl;
    }
  };

  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();

  int c_tl = 0, c_kt = 0, c_slot = 0;       // (tile, k-tile, ring slot) of the stage being multiplied
  if (nstages > 0) issue(0);
  if (nstages > 1) issue(1);
  for (int s = 0; s < nstages; ++s) {
    if (s + 1 < nstages)
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 2 < nstages) issue(s + 2);
    const unsigned char* sa = smem + c_slot * NT2_STAGE;
    const unsigned char* sb = sa + NT2_ABYTES;
    if (++c_slot == NT2_NS) c_slot = 0;
#pragma unroll
    for (int kc = 0; kc < NT2_BK / 16; ++kc) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wr * 64 + i * 32 + l31, rb = wc * 64 + i * 32 + l31;
        fa[i] = *(const bf16x8*)(sa + ra * 128 + (((2 * kc + h) ^ ((ra >> 1) & 7)) << 4));
        fb[i] = *(const bf16x8*)(sb + rb * 128 + (((2 * kc + h) ^ ((rb >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (CFRAG)
            dj_mfma(acc[i][j], fa[i], fb[j]);
          else
            dj_mfma(acc[i][j], fb[j], fa[i]);      // C^T block: lane <-> output row, registers <-> columns
        }
    }
    if (++c_kt == nk) {       // tile finished: epilogue (the next tiles' DMA is already in flight)
      const int m0 = (mt0 + c_tl * mstride) * NT2_BM;
      c_kt = 0;
      ++c_tl;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int colb = n0 + wc * 64 + j * 32, rowb = m0 + wr * 64 + i * 32;
          if constexpr (EPI != 0) {
            if (j == 0) cell_block_pair<EPI>(acc[i][0], acc[i][1], ce, bias, rowb + l31, M, colb, N, lane, h);
          } else if constexpr (CFRAG) {
            const int col = colb + l31;
            if (col < N && rowb < M) {
              const float bv = bias ? bias[col] : 0.f;
              float x[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) x[r] = acc[i][j][r] + bv;
              store_frag(C + (((int64_t)(rowb >> 5) * (N >> 5) + (colb >> 5)) * 64 + lane) * 16, x);
            }
          } else if (sizeof(TC) == 2 && vec16 && colb + 32 <= N) {     // wave-uniform
            const int row = rowb + l31;
            if (rowb < M)
              store_block_rows16((bf16_t*)C + rbs_row(row < M ? row : rowb, c_rbs) * ldc + colb, acc[i][j],
                                 bias ? bias + colb : nullptr, h, row < M);
          } else {
            const int row = rowb + l31;
            if (row < M && colb < N) {
              TC* crow = C + rbs_row(row, c_rbs) * ldc;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int col = colb + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  v[e] = acc[i][j][4 * g + e] + ((bias && col + e < N) ? bias[col + e] : 0.f);
                if (col + 4 <= N) {
                  if (c_acc) {                   // C += A Bt^T: this lane re-reads the 4 values it is about to write
                    float o[4];
                    load4(crow + col, o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += o[e];
                  }
                  store4(crow + col, v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    if (col + e < N) crow[col + e] = dj_from_f32<TC>(v[e] + (c_acc ? dj_to_f32(crow[col + e]) : 0.f));
                }
              }
            }
          }
        }
      zero_acc();
    }
  }
}


// ------------------------------------------------------------------ NT, bf16, 256 x 256 tiles
// Same contract as the kernel above for outputs at least 129 columns wide: one A panel pass covers
// 256 output columns (half the operand bytes per MFMA of the 256 x 128 tiling), 8 waves as 2 x 4,
// each 128 x 64 (8 accumulator blocks).  BK = 32, 4-stage LDS ring (4 x 32 KiB) filled by LDS-DMA
// three stages ahead.  LDS rows are 64 B (32 bf16); 16-byte chunk c of row r lives at chunk
// c ^ ((r>>2)&3), applied on the DMA source address: a ds_read_b128 wavefront phase (16 lanes =
// 16 rows) then touches every bank exactly once.  Tile order: the 32 workgroups of one XCD walk
// (m-tile, n-tile) pairs n-fastest, so the n-tiles of one A panel run side by side on ONE L2.
constexpr int NT3_BM = 256, NT3_BN = 256, NT3_BK = 32, NT3_NS = 4;
static_assert((NT3_NS & (NT3_NS - 1)) == 0, "ring slot counter wraps by masking");
constexpr int NT3_ABYTES = NT3_BM * NT3_BK * 2, NT3_BBYTES = NT3_BN * NT3_BK * 2, NT3_STAGE = NT3_ABYTES + NT3_BBYTES;


template <typename TC, bool CFRAG, int EPI = 0>
__global__ __launch_bounds__(512) void gemm_nt_bf16_wide_kernel(int M, int N, int K, const bf16_t* __restrict__ A,
                                                                int lda, const bf16_t* __restrict__ Bt, int ldb,
                                                                TC* __restrict__ C, int ldc,
                                                                const float* __restrict__ bias, int ntn, int ntm,
                                                                int a_rbs, int c_rbs, int64_t a_cts, int c_acc, CellEpi ce) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 2, wc = w & 3;
  const int nk = (K + NT3_BK - 1) / NT3_BK;
  // row-major bf16 output whose rows take 16-byte stores (store_block_rows16); not for the accumulating epilogue
  const bool vec16 = !c_acc && (ldc & 7) == 0 && (((uintptr_t)C | (uintptr_t)bias) & 15) == 0;
  // this workgroup's tiles: q = slot, slot + slots, ... over the (m-tile, n-tile) pairs of its XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = (int)(gridDim.x >> 3);
  const int mt_xcd = ntm > xcd ? (ntm - xcd + 7) / 8 : 0;       // m-tiles xcd, xcd+8, ...
  const int q_total = mt_xcd * ntn;
  const int my_tiles = slot < q_total ? (q_total - slot + slots - 1) / slots : 0;
  const int nstages = my_tiles * nk;
  const bf16_t* zl = (const bf16_t*)dj_zero_line;

  auto tile_of = [&](int tl, int& m0, int& n0) {
    const int q = slot + tl * slots;
    m0 = (xcd + 8 * (q / ntn)) * NT3_BM;
    n0 = (q % ntn) * NT3_BN;
  };
  // DMA stream: stages are issued strictly in order, so (tile, k-tile) are counters and the per-lane row pointers are
  // recomputed only when the tile changes (per stage that leaves two pointer adds and a bounds select per piece;
  // recomputing tile, row and address per stage was ~300 scalar + vector instructions per stage and wave -- as much
  // issue time as the stage's 16 MFMAs)
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const int rsub = lane >> 2, cp = lane & 3;
  const bf16_t* ra[2];
  const bf16_t* ra2[2];         // cell epilogue (EPI): the k range past ce.K1p comes from a second matrix (h_{t-1})
  const bf16_t* rb[2];
  int kz[2];                    // this lane's k offset inside a stage (swizzled 16-byte chunk)
  bool va[2], vb[2];
  int i_tl = 0, i_kt = 0, i_slot = 0;
  // one A piece + one B piece (i = 0, 1) of the stage the DMA stream stands at; dma_advance() moves it on
  auto dma_pair = [&](int i) {
    if (i_kt == 0 && i == 0) {
      int m0, n0;
      tile_of(i_tl, m0, n0);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int row = (w * 2 + ii) * 16 + rsub;
        kz[ii] = (cp ^ ((row >> 2) & 3)) << 3;
        va[ii] = m0 + row < M;
        vb[ii] = n0 + row < N;
        ra[ii] = va[ii] ? A + rbs_row(m0 + row, a_rbs) * lda + kz[ii] : zl;
        rb[ii] = vb[ii] ? Bt + (int64_t)(n0 + row) * ldb + kz[ii] : zl;
        if constexpr (EPI != 0)
          ra2[ii] = (va[ii] && ce.A2) ? (const bf16_t*)ce.A2 + rbs_row(m0 + row, a_rbs) * ce.lda2 + kz[ii] : zl;
      }
    }
    const int k0 = i_kt * NT3_BK;
    // A may be column-tile-major (a_cts != 0: element (m, k) at (k >> 8) * a_cts + m * lda + (k & 255), lda = 256)
    const int64_t ka = a_cts ? (int64_t)(k0 >> 8) * a_cts + (k0 & 255) : k0;
    const unsigned sa = lds0 + (unsigned)i_slot * NT3_STAGE, sb = sa + NT3_ABYTES;
    const bool kin = k0 + kz[i] < K;
    const unsigned po = (unsigned)(w * 2 + i) * 1024u;
    const bf16_t* asrc;
    if constexpr (EPI != 0)        // [x_t | h_{t-1}]: columns [0, K1) of A (zeros up to K1p), then columns of A2
      asrc = k0 < ce.K1p ? ((va[i] && k0 + kz[i] < ce.K1) ? ra[i] + k0 : zl) : ((va[i] && kin) ? ra2[i] + (k0 - ce.K1p) : zl);
    else
      asrc = (va[i] && kin) ? ra[i] + ka : zl;
    glds16(asrc, __builtin_amdgcn_readfirstlane(sa + po));
    glds16((vb[i] && kin) ? rb[i] + k0 : zl, __builtin_amdgcn_readfirstlane(sb + po));
  };
  auto dma_advance = [&]() {
    i_slot = (i_slot + 1) & (NT3_NS - 1);
    if (++i_kt == nk) {
      i_kt = 0;
      ++i_tl;
    }
  };
  auto issue = [&](int) {
    dma_pair(0);
    dma_pair(1);
    dma_advance();
  };

  f32x16 acc[4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();

  int c_tl = 0, c_kt = 0;       // (tile, k-tile) of the stage being multiplied
  // operand fragments of one half stage (16 of the 32 k): 4 A row blocks + 2 B row blocks
  struct Half { bf16x8 a[4], b[2]; };
  auto read_half = [&](int s, int kc) {
    Half f;
    const unsigned char* sa = smem + (s % NT3_NS) * NT3_STAGE;
    const unsigned char* sb = sa + NT3_ABYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ra_ = wr * 128 + i * 32 + l31;
      f.a[i] = *(const bf16x8*)(sa + ra_ * 64 + (((2 * kc + h) ^ ((ra_ >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rb_ = wc * 64 + j * 32 + l31;
      f.b[j] = *(const bf16x8*)(sb + rb_ * 64 + (((2 * kc + h) ^ ((rb_ >> 2) & 3)) << 4));
    }
    return f;
  };
  auto mm = [&](f32x16& c, const bf16x8& fa, const bf16x8& fb) {
    if constexpr (CFRAG)
      dj_mfma(c, fa, fb);
    else
      dj_mfma(c, fb, fa);          // C^T block: lane <-> output row, registers <-> columns
  };
  // 8 MFMAs of one half stage with one DMA pair of a later stage issued between them (lstm_wgrad_bf16_kernel)
  auto mma_half = [&](const Half& f, bool dma, int pair) {
    __builtin_amdgcn_sched_barrier(0);
    mm(acc[0][0], f.a[0], f.b[0]);
    mm(acc[0][1], f.a[0], f.b[1]);
    __builtin_amdgcn_sched_barrier(0);
    if (dma) dma_pair(pair);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 1; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mm(acc[i][j], f.a[i], f.b[j]);
    __builtin_amdgcn_sched_barrier(0);
  };
  // Half-stage software pipeline, the template of lstm_wgrad_bf16_kernel: LDS reads of one half stage under the MFMAs
  // of the previous half, DMA of later stages between the MFMAs, all four ring slots filled up front.
  //   stage s:  read(s, half 1) | MFMA(half 0) + DMA pair 1 of stage s+3 | stage s+1 landed? (counted vmcnt) | own
  //             reads done (lgkmcnt 0) | barrier | read(s+1, half 0) | MFMA(half 1) + DMA pair 0 of stage s+4 | epilogue
  //             if the tile is complete (its stores count in vmcnt behind the DMAs: the next wait over-waits, never under)
#pragma unroll
  for (int s = 0; s < NT3_NS; ++s)
    if (s < nstages) issue(s);
  if (nstages >= 4)
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (nstages == 3)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nstages == 2)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  Half f0;
  if (nstages > 0) f0 = read_half(0, 0);
  for (int s = 0; s < nstages; ++s) {
    Half f1 = read_half(s, 1);
    const bool second = s >= 1 && s + 3 < nstages;     // second pair of stage s+3 (its first went out last iteration)
    mma_half(f0, second, 1);
    if (second) dma_advance();
    const int rem = nstages - 1 - s;                   // stages after s
    if (rem >= 3)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (rem == 2)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0) as a builtin: hipcc then knows f1 has arrived
    __builtin_amdgcn_s_barrier();
    if (rem >= 1) f0 = read_half(s + 1, 0);
    mma_half(f1, s + 4 < nstages, 0);                  // first pair of stage s+4 into the slot just vacated
    if (++c_kt == nk) {       // tile finished: epilogue (the next tiles' DMA is already in flight)
      int m0, n0;
      tile_of(c_tl, m0, n0);
      c_kt = 0;
      ++c_tl;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int colb = n0 + wc * 64 + j * 32, rowb = m0 + wr * 128 + i * 32;
          if constexpr (EPI != 0) {
            if (j == 0) cell_block_pair<EPI>(acc[i][0], acc[i][1], ce, bias, rowb + l31, M, colb, N, lane, h);
          } else if constexpr (CFRAG) {
            const int col = colb + l31;
            if (col < N && rowb < M) {
              const float bv = bias ? bias[col] : 0.f;
              float x[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) x[r] = acc[i][j][r] + bv;
              store_frag(C + (((int64_t)(rowb >> 5) * (N >> 5) + (colb >> 5)) * 64 + lane) * 16, x);
            }
          } else if (sizeof(TC) == 2 && vec16 && colb + 32 <= N) {     // wave-uniform
            const int row = rowb + l31;
            if (rowb < M)
              store_block_rows16((bf16_t*)C + rbs_row(row < M ? row : rowb, c_rbs) * ldc + colb, acc[i][j],
                                 bias ? bias + colb : nullptr, h, row < M);
          } else {
            const int row = rowb + l31;
            if (row < M && colb < N) {
              TC* crow = C + rbs_row(row, c_rbs) * ldc;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int col = colb + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  v[e] = acc[i][j][4 * g + e] + ((bias && col + e < N) ? bias[col + e] : 0.f);
                if (col + 4 <= N) {
                  if (c_acc) {                   // C += A Bt^T: this lane re-reads the 4 values it is about to write
                    float o[4];
                    load4(crow + col, o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += o[e];
                  }
                  store4(crow + col, v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    if (col + e < N) crow[col + e] = dj_from_f32<TC>(v[e] + (c_acc ? dj_to_f32(crow[col + e]) : 0.f));
                }
              }
            }
          }
        }
      zero_acc();
    }
  }
}

}  // namespace

// ------------------------------------------------------------------ launchers (internal C++ API)
int dj_launch_gemm_nt(int dtype, int M, int N, int K, const void* A, int lda, const void* Bt, int ldb, void* C, int ldc,
                      int c_mode, const float* bias, hipStream_t st) {
  return dj_launch_gemm_nt_ex(dtype, M, N, K, A, lda, 1, 0, Bt, ldb, C, ldc, 1, c_mode, bias, st);
}
int dj_launch_gemm_nt_rbs(int dtype, int M, int N, int K, const void* A, int lda, int a_rbs, const void* Bt, int ldb,
                          void* C, int ldc, int c_rbs, int c_mode, const float* bias, hipStream_t st) {
  return dj_launch_gemm_nt_ex(dtype, M, N, K, A, lda, a_rbs, 0, Bt, ldb, C, ldc, c_rbs, c_mode, bias, st);
}

// a_cts != 0: A is column-tile-major, [K/256][rows][256] with a_cts elements between column tiles and lda = 256
// (the dZ layout of the bf16 persistent BPTT kernels); bf16 only
int dj_launch_gemm_nt_ex(int dtype, int M, int N, int K, const void* A, int lda, int a_rbs, int64_t a_cts, const void* Bt,
                         int ldb, void* C, int ldc, int c_rbs, int c_mode, const float* bias, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (a_cts && (dtype != DJ_BF16 || lda != 256)) return 1008;      // (composes with a row-block stride: per-step views)
  if (a_rbs < 1 || c_rbs < 1 || (c_mode == 2 && c_rbs != 1)) return 1006;
  const int epl = dtype == DJ_F32 ? 4 : 8;
  if ((K % epl) || (lda % epl) || (ldb % epl)) return 1001;
  // c_mode 3: row-major C in the operand dtype, ACCUMULATED (C += A Bt^T + bias); bf16 kernels only
  const int c_is_f32 = c_mode == 1, c_frag = c_mode == 2, c_acc = c_mode == 3;
  if (c_acc && dtype != DJ_BF16) return 1005;
  if (c_frag && ((M % 32) || (N % 32))) return 1004;
  // wide outputs take 256 x 256 tiles -- unless that leaves compute units idle: fewer than 256 tiles (one per CU) and the
  // 256 x 128 tiling has twice as many (the per-step BPTT product of the scaled model, [8192 x 4096] x [4096 x 1024]:
  // 128 tiles of 256 x 256 ran on half of the chip)
  const bool few_wide_tiles = (int64_t)((N + NT3_BN - 1) / NT3_BN) * ((M + NT3_BM - 1) / NT3_BM) < 256 &&
                              (int64_t)((N + NT2_BN - 1) / NT2_BN) * ((M + NT2_BM - 1) / NT2_BM) >=
                                  2 * (int64_t)((N + NT3_BN - 1) / NT3_BN) * ((M + NT3_BM - 1) / NT3_BM);
  if (dtype == DJ_BF16 && N > 128 && !few_wide_tiles) {
    // wide outputs: 256 x 256 tiles; a remainder of at most 128 columns goes to the 256 x 128 kernel
    const int rem = N % NT3_BN;
    if (rem > 0 && rem <= 128 && N > NT3_BN && !c_frag) {
      const int Nw = N - rem;
      const size_t cesz = c_is_f32 ? 4 : 2;
      int rc = dj_launch_gemm_nt_ex(dtype, M, Nw, K, A, lda, a_rbs, a_cts, Bt, ldb, C, ldc, c_rbs, c_mode, bias, st);
      if (rc) return rc;
      return dj_launch_gemm_nt_ex(dtype, M, rem, K, A, lda, a_rbs, a_cts, (const bf16_t*)Bt + (int64_t)Nw * ldb, ldb,
                                  (char*)C + Nw * cesz, ldc, c_rbs, c_mode, bias ? bias + Nw : nullptr, st);
    }
    const int ntn3 = (N + NT3_BN - 1) / NT3_BN, ntm3 = (M + NT3_BM - 1) / NT3_BM;
    int grid3 = 256;
    if ((int64_t)ntn3 * ntm3 < 256) grid3 = ((ntn3 * ntm3 + 7) / 8) * 8;   // whole XCD rounds (blockIdx & 7 = XCD)
    const size_t smem = (size_t)NT3_NS * NT3_STAGE;
    static bool attr3_done_dev[DJ_MAX_DEVICES] = {};
    bool& attr3_done = attr3_done_dev[dj_current_device()];
    if (!attr3_done) {
      const void* fns[3] = {(const void*)gemm_nt_bf16_wide_kernel<bf16_t, false>,
                            (const void*)gemm_nt_bf16_wide_kernel<float, false>,
                            (const void*)gemm_nt_bf16_wide_kernel<bf16_t, true>};
      for (const void* fn : fns) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
      }
      attr3_done = true;
    }
    if (ldc % 4) return 1007;
    if (c_frag)
      hipLaunchKernelGGL((gemm_nt_bf16_wide_kernel<bf16_t, true>), dim3(grid3), dim3(512), smem, st, M, N, K,
                         (const bf16_t*)A, lda, (const bf16_t*)Bt, ldb, (bf16_t*)C, ldc, bias, ntn3, ntm3, a_rbs, c_rbs, a_cts, c_acc, CellEpi{});
    else if (c_is_f32)
      hipLaunchKernelGGL((gemm_nt_bf16_wide_kernel<float, false>), dim3(grid3), dim3(512), smem, st, M, N, K,
                         (const bf16_t*)A, lda, (const bf16_t*)Bt, ldb, (float*)C, ldc, bias, ntn3, ntm3, a_rbs, c_rbs, a_cts, c_acc, CellEpi{});
    else
      hipLaunchKernelGGL((gemm_nt_bf16_wide_kernel<bf16_t, false>), dim3(grid3), dim3(512), smem, st, M, N, K,
                         (const bf16_t*)A, lda, (const bf16_t*)Bt, ldb, (bf16_t*)C, ldc, bias, ntn3, ntm3, a_rbs, c_rbs, a_cts, c_acc, CellEpi{});
    return (int)hipGetLastError();
  }
  if (dtype == DJ_BF16) {
    const int ntn2 = (N + NT2_BN - 1) / NT2_BN, ntm2 = (M + NT2_BM - 1) / NT2_BM;
    // persistent grid: one workgroup per CU.  XCD-aware schedule when there is enough work:
    // 32 slots per XCD are split into teams of ntn2 workgroups, one A panel per team at a time.
    int grid2, xcd_map = 0;
    if (ntn2 <= 32 && (ntm2 >= 64 || (32 % ntn2 == 0 && ntm2 >= 8 * (32 / ntn2)))) {   // (.. or exactly enough m-tiles for one per team)
      grid2 = 256;
      xcd_map = 1;
    } else {
      int mrows = 256 / ntn2;                       // concurrent A panels
      if (mrows < 1) mrows = 1;
      if (mrows > ntm2) mrows = ntm2;
      grid2 = mrows * ntn2;
    }
    const size_t smem = (size_t)NT2_NS * NT2_STAGE;
    static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
    if (!attr_done) {
      const void* fns[3] = {(const void*)gemm_nt_bf16_dma_kernel<bf16_t, false>,
                            (const void*)gemm_nt_bf16_dma_kernel<float, false>,
                            (const void*)gemm_nt_bf16_dma_kernel<bf16_t, true>};
      for (const void* fn : fns) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
      }
      attr_done = true;
    }
    if (ldc % 4) return 1007;
    if (c_frag)
      hipLaunchKernelGGL((gemm_nt_bf16_dma_kernel<bf16_t, true>), dim3(grid2), dim3(512), smem, st, M, N, K,
                         (const bf16_t*)A, lda, (const bf16_t*)Bt, ldb, (bf16_t*)C, ldc, bias, ntn2, ntm2, xcd_map,
                         a_rbs, c_rbs, a_cts, c_acc, CellEpi{});
    else if (c_is_f32)
      hipLaunchKernelGGL((gemm_nt_bf16_dma_kernel<float, false>), dim3(grid2), dim3(512), smem, st, M, N, K,
                         (const bf16_t*)A, lda, (const bf16_t*)Bt, ldb, (float*)C, ldc, bias, ntn2, ntm2, xcd_map,
                         a_rbs, c_rbs, a_cts, c_acc, CellEpi{});
    else
      hipLaunchKernelGGL((gemm_nt_bf16_dma_kernel<bf16_t, false>), dim3(grid2), dim3(512), smem, st, M, N, K,
                         (const bf16_t*)A, lda, (const bf16_t*)Bt, ldb, (bf16_t*)C, ldc, bias, ntn2, ntm2, xcd_map,
                         a_rbs, c_rbs, a_cts, c_acc, CellEpi{});
    return (int)hipGetLastError();
  }
  // fp32 (parity mode): register-staged 128 x 128 kernel on v_mfma_f32_32x32x2_f32
  int ntn = (N + 127) / 128, ntm = (M + 127) / 128;
  dim3 grid((unsigned)(ntn * (int64_t)ntm)), block(256);
  hipLaunchKernelGGL((gemm_nt_kernel<float, float>), grid, block, 0, st, M, N, K, (const float*)A, lda,
                     (const float*)Bt, ldb, (float*)C, ldc, bias, ntn, c_frag, a_rbs, c_rbs);
  return (int)hipGetLastError();
}

// One recurrence step of the generic-width LSTM path with the cell as the GEMM's epilogue (bf16; CellEpi in dj_kernels.h).
// Tile choice as dj_launch_gemm_nt_ex (256 x 256, or 256 x 128 when that fills more compute units); N is not split (the
// epilogue derives gate and unit from the global column), partial tiles are guarded by the kernels.
template <int EPI>
static int launch_cell(int M, int N, int K, const bf16_t* A, int lda, int a_rbs, const bf16_t* Bt, int ldb, const CellEpi& ce,
                       const float* bias, hipStream_t st) {
  const int64_t wide_tiles = (int64_t)((N + NT3_BN - 1) / NT3_BN) * ((M + NT3_BM - 1) / NT3_BM);
  const int64_t narrow_tiles = (int64_t)((N + NT2_BN - 1) / NT2_BN) * ((M + NT2_BM - 1) / NT2_BM);
  const bool few_wide_tiles = wide_tiles < 256 && narrow_tiles >= 2 * wide_tiles;
  static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_bf16_wide_kernel<bf16_t, false, EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, NT3_NS * NT3_STAGE);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)gemm_nt_bf16_dma_kernel<bf16_t, false, EPI>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, NT2_NS * NT2_STAGE);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  if (N > 128 && !few_wide_tiles) {
    const int ntn3 = (N + NT3_BN - 1) / NT3_BN, ntm3 = (M + NT3_BM - 1) / NT3_BM;
    int grid3 = 256;
    if ((int64_t)ntn3 * ntm3 < 256) grid3 = ((ntn3 * ntm3 + 7) / 8) * 8;
    hipLaunchKernelGGL((gemm_nt_bf16_wide_kernel<bf16_t, false, EPI>), dim3(grid3), dim3(512), (size_t)NT3_NS * NT3_STAGE, st, M,
                       N, K, A, lda, Bt, ldb, (bf16_t*)nullptr, 0, bias, ntn3, ntm3, a_rbs, 1, (int64_t)0, 0, ce);
    return (int)hipGetLastError();
  }
  const int ntn2 = (N + NT2_BN - 1) / NT2_BN, ntm2 = (M + NT2_BM - 1) / NT2_BM;
  int grid2, xcd_map = 0;
  if (ntn2 <= 32 && (ntm2 >= 64 || (32 % ntn2 == 0 && ntm2 >= 8 * (32 / ntn2)))) {   // (.. or exactly enough m-tiles for one per team)
    grid2 = 256;
    xcd_map = 1;
  } else {
    int mrows = 256 / ntn2;
    if (mrows < 1) mrows = 1;
    if (mrows > ntm2) mrows = ntm2;
    grid2 = mrows * ntn2;
  }
  hipLaunchKernelGGL((gemm_nt_bf16_dma_kernel<bf16_t, false, EPI>), dim3(grid2), dim3(512), (size_t)NT2_NS * NT2_STAGE, st, M, N,
                     K, A, lda, Bt, ldb, (bf16_t*)nullptr, 0, bias, ntn2, ntm2, xcd_map, a_rbs, 1, (int64_t)0, 0, ce);
  return (int)hipGetLastError();
}
int dj_launch_gemm_nt_cell(int M, const void* A, int lda, int a_rbs, const void* Bt, int ldb, const CellEpi& ce,
                           const float* bias, hipStream_t st) {
  if (M <= 0) return 0;
  const int N = 4 * ce.H, K = ce.K1p + (ce.A2 ? ce.H : 0);
  if ((M % 32) || (ce.H % 8) || (lda % 8) || (ldb % 8) || (ce.lda2 % 8) || a_rbs < 1 || !ce.carry || !ce.Hs || !bias ||
      ce.K1 > ce.K1p || (ce.K1p % 64) || ce.K1 > lda || ldb < ce.K1p + ce.H || (!ce.A2 && !ce.first) || (ce.Cs && !ce.Z))
    return 1026;
  const bf16_t *a = (const bf16_t*)A, *b = (const bf16_t*)Bt;
  return ce.sigm ? launch_cell<2>(M, N, K, a, lda, a_rbs, b, ldb, ce, bias, st)
                 : launch_cell<1>(M, N, K, a, lda, a_rbs, b, ldb, ce, bias, st);
}

int dj_launch_gemm_tn(int dtype, int64_t M, int Ka, int ka_valid, int N, const void* A, int lda, const void* B, int ldb, float* C,
                      int ldc, int a_shift, int steps, hipStream_t st) {
  if (M <= 0 || N <= 0 || Ka <= 0) return 0;
  const int epl = dtype == DJ_F32 ? 4 : 8;
  if ((Ka % epl) || (N % epl) || (lda % epl) || (ldb % epl)) return 1002;
  if (a_shift && (a_shift != 32 || steps <= 0)) return 1003;
  int ntn = (N + 127) / 128, nta = (Ka + 127) / 128, ntiles = ntn * nta;
  // split the reduction so that ~1024 workgroups are in flight; chunks are multiples of 64 rows
  int64_t target = (1024 + ntiles - 1) / ntiles;
  int64_t rps = (M + target - 1) / target;
  rps = ((rps + 63) / 64) * 64;
  int splits = (int)((M + rps - 1) / rps);
  dim3 grid((unsigned)(ntiles * splits)), block(256);
  if (dtype == DJ_BF16 && !a_shift && Ka <= 96 && N <= 64 && ka_valid <= Ka) {
    // small output (the conv kernel gradient): one accumulator tile per wave, rows split over 512 x 4 waves
    int64_t rpw = (M + 2047) / 2048;
    rpw = (rpw + 31) / 32 * 32;
    const int nwg = (int)((M + rpw * 4 - 1) / (rpw * 4));
    hipLaunchKernelGGL(gemm_tn_small_bf16_kernel, dim3(nwg), block, 0, st, M, Ka, ka_valid, N, (const bf16_t*)A, lda,
                       (const bf16_t*)B, ldb, C, ldc, rpw);
    return (int)hipGetLastError();
  }
  if (dtype == DJ_BF16 && (N % TN2_TB) == 0 && (M % TN2_BK) == 0) {
    // DMA-ring kernel: 128 x 256 tiles, ~2 resident rounds of workgroups
    int ntn2 = N / TN2_TB, nta2 = (Ka + TN2_TA - 1) / TN2_TA, nt2 = ntn2 * nta2;
    int64_t want = (512 + nt2 - 1) / nt2;
    int64_t rps2 = (M + want - 1) / want;
    rps2 = ((rps2 + TN2_BK - 1) / TN2_BK) * TN2_BK;
    int splits2 = (int)((M + rps2 - 1) / rps2);
    const size_t smem = (size_t)TN2_NS * TN2_STAGE;
    static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_bf16_dma_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    hipLaunchKernelGGL(gemm_tn_bf16_dma_kernel, dim3((unsigned)(nt2 * splits2)), block, smem, st, M, Ka, ka_valid, N,
                       (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, ntn2, nt2, rps2, a_shift, steps);
    return (int)hipGetLastError();
  }
  if (dtype == DJ_F32)
    hipLaunchKernelGGL(gemm_tn_f32_kernel, grid, block, 0, st, M, Ka, N, (const float*)A, lda, (const float*)B, ldb, C,
                       ldc, ntn, ntiles, rps, a_shift, steps, ka_valid);
  else
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, grid, block, 0, st, M, Ka, N, (const bf16_t*)A, lda, (const bf16_t*)B, ldb,
                       C, ldc, ntn, ntiles, rps, a_shift, steps, ka_valid);
  return (int)hipGetLastError();
}

// dW [D,N] += X^T dZ, dU [H,N] += Hprev^T dZ.  bf16: one fused DMA-ring kernel; fp32: two generic launches.
int dj_launch_lstm_wgrad(int dtype, int64_t M, int steps, const void* X, int DP, int D, const void* Hs, int H,
                         const void* dZ, int N, int64_t dz_cts, float* dW, float* dU, const void* zeros, hipStream_t st) {
  if (M <= 0) return 0;
  if (dz_cts && !(dtype == DJ_BF16 && (N % WG_T) == 0 && (M % 32) == 0 && (DP % 8) == 0 && (H % 8) == 0 && zeros))
    return 1009;        // the column-tile-major dZ exists for the fused bf16 kernel only
  if (dtype == DJ_BF16 && (N % WG_T) == 0 && (M % 32) == 0 && (DP % 8) == 0 && (H % 8) == 0 && zeros) {
    WgradArgs a;
    a.dz_cts = dz_cts ? dz_cts : 256;
    a.ldz = dz_cts ? 256 : N;
    a.M = M; a.X = (const bf16_t*)X; a.DP = DP; a.D = D; a.Hs = (const bf16_t*)Hs; a.H = H; a.steps = steps;
    a.dZ = (const bf16_t*)dZ; a.N = N; a.dW = dW; a.dU = dU; a.zeros = (const bf16_t*)zeros;
    a.ntn = N / WG_T;
    int nta = (DP + H + WG_T - 1) / WG_T;
    a.ntiles = a.ntn * nta;
    int64_t want = (256 + a.ntiles - 1) / a.ntiles;
    want = (want + 7) / 8 * 8;                       // multiple of 8 row splits: one XCD per split residue
    int64_t rps = (M + want - 1) / want;
    rps = ((rps + WG_BK - 1) / WG_BK) * WG_BK;
    a.rows_per_split = rps;
    int splits = (int)want;                          // trailing splits may be empty (nkt = 0)
    a.xcd_map = 1;
    const size_t smem = (size_t)WG_NS * WG_STAGE;
    static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute((const void*)lstm_wgrad_bf16_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return (int)e;
      attr_done = true;
    }
    hipLaunchKernelGGL(lstm_wgrad_bf16_kernel, dim3((unsigned)(a.ntiles * splits)), dim3(512), smem, st, a);
    return (int)hipGetLastError();
  }
  int rc = dj_launch_gemm_tn(dtype, M, DP, D, N, X, DP, dZ, N, dW, N, 0, 0, st);
  if (rc) return rc;
  return dj_launch_gemm_tn(dtype, M, H, H, N, Hs, H, dZ, N, dU, N, 32, steps, st);
}
