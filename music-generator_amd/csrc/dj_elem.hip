// Non-GEMM kernels of the DeepJ hot path: feature assembly (octave conv + positional
// features), style projections, inter-layer glue (dropout + style add + axis swap),
// play/replay/volume head with the masked loss, and the Nadam update.
// All HBM-bound; every kernel reads each activation once with coalesced rows.
#include <stdlib.h>

#include "dj_kernels.h"

namespace {

// 8 consecutive operand elements <-> 8 floats (16-byte accesses)
__device__ __forceinline__ void load8(const float* p, float (&x)[8]) {
  float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&x)[8]) {
  uint4 v = *(const uint4*)p;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[2 * i] = __uint_as_float(w[i] << 16);
    x[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
  }
}
// the same in two halves: the REQUEST (raw registers, no arithmetic on them) and the conversion -- so that a row can be
// requested one loop iteration ahead of its use (head_loss_kernel)
template <typename T> struct Raw8;
template <> struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = ((const float4*)p)[0]; b = ((const float4*)p)[1]; }
  __device__ __forceinline__ void expand(float (&x)[8]) const {
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
  }
};
template <> struct Raw8<bf16_t> {
  uint4 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *(const uint4*)p; }
  __device__ __forceinline__ void expand(float (&x)[8]) const {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[2 * i] = __uint_as_float(w[i] << 16);
      x[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
  }
};
__device__ __forceinline__ void store8(float* p, const float (&x)[8]) {
  ((float4*)p)[0] = make_float4(x[0], x[1], x[2], x[3]);
  ((float4*)p)[1] = make_float4(x[4], x[5], x[6], x[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&x)[8]) {
  *(uint4*)p = make_uint4(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]),
                          pack_bf16x2(x[6], x[7]));
}

// ------------------------------------------------------------------ small dense (fp32)
// C[m, n] = act(A[m,:K] . W[:K, n] + b[n])      (reference model.py:141-142 style
// embedding; model.py:77,110-113 per-layer style Dense followed by tanh)
__device__ __forceinline__ void dense_small_body(const float* __restrict__ A, int M, int K,
                                                 const float* __restrict__ W, const float* __restrict__ b,
                                                 float* __restrict__ C, int N, int act_tanh) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)M * N) return;
  int m = idx / N, n = idx % N;
  float s = b ? b[n] : 0.f;
  for (int k = 0; k < K; ++k) s += A[(int64_t)m * K + k] * W[(int64_t)k * N + n];
  C[idx] = act_tanh ? dj_tanh(s) : s;
}
__global__ void dense_small_kernel(const float* __restrict__ A, int M, int K, const float* __restrict__ W,
                                   const float* __restrict__ b, float* __restrict__ C, int N, int act_tanh) {
  dense_small_body(A, M, K, W, b, C, N, act_tanh);
}
// Several small dense layers that share the input A in ONE launch (blockIdx.y = layer): the per-layer style
// projections and their gradients are latency-bound kernels of a few microseconds of work each, so running
// them side by side costs the time of the longest instead of the sum.
// Tile = 32 rows x 64 columns per workgroup, A tile in LDS (one wave = one 8-row group, so its reads are
// broadcasts), each weight element loaded once per 8 outputs.  K <= 64.
__global__ __launch_bounds__(256) void dense_small_batch_kernel(DenseBatch d, int act_tanh) {
  __shared__ float As[32][65];
  const int l = blockIdx.z, N = d.N[l], K = d.K;
  const int nl = threadIdx.x & 63, rg = threadIdx.x >> 6, n = blockIdx.x * 64 + nl, m0 = blockIdx.y * 32;
  if (blockIdx.x * 64 >= N) return;                       // whole workgroup past this layer's width
  for (int i = threadIdx.x; i < 32 * K; i += 256) {
    const int r = i / K, k = i % K;
    As[r][k] = (m0 + r < d.M) ? d.A[(int64_t)(m0 + r) * K + k] : 0.f;
  }
  __syncthreads();
  if (n >= N) return;
  const float* W = d.W[l];
  const float bv = d.b[l] ? d.b[l][n] : 0.f;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = bv;
#pragma unroll 8
  for (int k = 0; k < K; ++k) {
    const float w = W[(int64_t)k * N + n];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += As[rg * 8 + i][k] * w;
  }
  float* C = d.C[l];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + rg * 8 + i;
    if (m < d.M) C[(int64_t)m * N + n] = act_tanh ? dj_tanh(acc[i]) : acc[i];
  }
}
// dA[m,k] (+)= sum_n dC[m,n] * W[k,n].  Block = DSBX_RB rows; n runs in chunks of NC columns: W^T chunk
// staged in LDS ([n][K+1]: the transposing store and the k-parallel read are conflict-free), the dC rows too;
// thread = (k, group of 8 rows).
constexpr int DSBX_NC = 192, DSBX_RB = 32;
__device__ __forceinline__ void dense_small_bwd_x_body(const float* __restrict__ dC, int M, int N,
                                                       const float* __restrict__ W, int K, float* __restrict__ dA,
                                                       int accumulate) {
  extern __shared__ float sm[];
  const int NC = N < DSBX_NC ? N : DSBX_NC;
  const int KP = K + 1;
  float* wt = sm;               // [NC][KP]
  float* dc = sm + NC * KP;     // [RB][NC]
  const int tid = threadIdx.x, m0 = blockIdx.x * DSBX_RB;
  const int k = tid % 64, rg = tid / 64;   // rows rg*8 .. rg*8+7
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
  for (int n0 = 0; n0 < N; n0 += NC) {
    const int nc = N - n0 < NC ? N - n0 : NC;
    if (n0) __syncthreads();
    for (int i = tid; i < nc * K; i += 256) {
      int kk = i / nc, n = i % nc;   // coalesced read of W[kk][n0 + n]
      wt[n * KP + kk] = W[(int64_t)kk * N + n0 + n];
    }
    for (int i = tid; i < DSBX_RB * nc; i += 256) {
      int r = i / nc, n = i % nc;
      dc[r * NC + n] = (m0 + r < M) ? dC[(int64_t)(m0 + r) * N + n0 + n] : 0.f;
    }
    __syncthreads();
    if (k < K) {
      for (int n = 0; n < nc; ++n) {
        const float w = wt[n * KP + k];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += dc[(rg * 8 + i) * NC + n] * w;
      }
    }
  }
  if (k >= K) return;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + rg * 8 + i;
    if (m >= M) continue;
    float* dst = dA + (int64_t)m * K + k;
    if (accumulate == 2)
      atomicAdd(dst, s[i]);         // several layers add into the same dA concurrently
    else
      *dst = accumulate ? *dst + s[i] : s[i];
  }
}
__global__ __launch_bounds__(256) void dense_small_bwd_x_kernel(const float* __restrict__ dC, int M, int N,
                                                                const float* __restrict__ W, int K,
                                                                float* __restrict__ dA, int accumulate) {
  dense_small_bwd_x_body(dC, M, N, W, K, dA, accumulate);
}
__global__ __launch_bounds__(256) void dense_small_bwd_x_batch_kernel(DenseBatch d, float* __restrict__ dA) {
  const int l = blockIdx.y;
  dense_small_bwd_x_body(d.dC[l], d.M, d.N[l], d.W[l], d.K, dA, 2);
}
// dW[k,n] += sum_m A[m,k] dC[m,n];  db[n] += sum_m dC[m,n].  One block per
// (row chunk, 64-col strip); thread (kq, n) owns K/4 rows of dW for its column.
__device__ __forceinline__ void dense_small_bwd_w_body(const float* __restrict__ A, int M, int K,
                                                       const float* __restrict__ dC, int N,
                                                       float* __restrict__ dW, float* __restrict__ db,
                                                       int rows_per_block) {
  extern __shared__ float sm[];   // As[rows][K]
  const int tid = threadIdx.x, nl = tid & 63, kq = tid >> 6;
  const int n = blockIdx.y * 64 + nl;
  const int m0 = blockIdx.x * rows_per_block;
  int m1 = m0 + rows_per_block;
  if (m1 > M) m1 = M;
  for (int i = tid; i < (m1 - m0) * K; i += 256) sm[i] = A[(int64_t)m0 * K + i];
  __syncthreads();
  if (n >= N) return;
  // K <= 64 in this model (style units); each thread handles k = kq, kq+4, ...
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bsum = 0.f;
#pragma unroll 4
  for (int m = m0; m < m1; ++m) {
    float d = dC[(int64_t)m * N + n];
    bsum += d;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int k = kq + 4 * i;
      if (k < K) acc[i] += sm[(m - m0) * K + k] * d;
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int k = kq + 4 * i;
    if (k < K) atomicAdd(dW + (int64_t)k * N + n, acc[i]);
  }
  if (db && kq == 0) atomicAdd(db + n, bsum);
}
__global__ __launch_bounds__(256) void dense_small_bwd_w_kernel(const float* __restrict__ A, int M, int K,
                                                                const float* __restrict__ dC, int N,
                                                                float* __restrict__ dW, float* __restrict__ db,
                                                                int rows_per_block) {
  dense_small_bwd_w_body(A, M, K, dC, N, dW, db, rows_per_block);
}
__global__ __launch_bounds__(256) void dense_small_bwd_w_batch_kernel(DenseBatch d, int rows_per_block) {
  const int l = blockIdx.z;
  dense_small_bwd_w_body(d.A, d.M, d.K, d.dC[l], d.N[l], d.dW[l], d.db[l], rows_per_block);
}

// ------------------------------------------------------------------ pitch bins (model.py:43-49)
// bins[i, b, t] = sum_k dropped_notes[b, t, i + 12k, 0]
__global__ void bins_kernel(const float* __restrict__ notes, float* __restrict__ bins, int B, int T, int N, int octave,
                            DjDrop dn) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= octave * B * T) return;
  int i = idx / (B * T), bt = idx % (B * T);
  float s = 0.f;
  for (int n = i; n < N; n += octave) {
    uint32_t r = (uint32_t)bt * N + n;
    s += notes[(int64_t)r * 3] * dj_keep(dn, dj_rowkey(dn, r), 0);
  }
  bins[idx] = s;
}

// ------------------------------------------------------------------ feature assembly

constexpr int CONV_K = 24, CONV_C = 3, CONV_O = 64, CONV_L = 11;   // 'same': pad 11 left / 12 right
constexpr int XCOL_LD = 80;                                        // 72 taps padded to a multiple of 8

// x0[b,n,t,:] = concat[pos, class, bins, drop(tanh(conv)), drop(beat)] + drop(tanh-style)  (model.py:56-82)
// The octave convolution (Conv1D(64, 24, 'same') along the notes, model.py:56-58) is a GEMM over the im2col
// view: (1) feature_xcol_kernel writes Xcol [rows, 80] (72 taps of the dropped-out notes in conv-kernel order
// k*3+c, zero padded), (2) dj_gemm_nt multiplies it with the transposed kernel (+ bias) into Y [rows, 64] on the
// MFMA units, (3) feature_asm_kernel applies tanh (stashing it in place for BPTT), dropout and the style term
// and assembles whole X rows in LDS ([FEAT_NC][FP], 16-byte stores).  Xcol doubles as the operand of the
// conv weight gradient (dWc = Xcol^T dY).
constexpr int FEAT_NC = 128;
template <typename T>
__global__ __launch_bounds__(256) void feature_xcol_kernel(FeatArgs a, T* __restrict__ Xcol) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  const int XR = (a.N + 3) / 4 * 4 + CONV_K + 3;          // padded note rows of the dropped-out input
  float* xin = (float*)fsm;                                // [XR][3]
  const int tid = threadIdx.x;
  for (int bt = blockIdx.x; bt < a.B * a.T; bt += gridDim.x) {
    const int b = bt / a.T, t = bt % a.T;
    __syncthreads();
    for (int i = tid; i < XR * 3; i += 256) {
      int n = i / 3 - CONV_L, c = i % 3;
      float v = 0.f;
      if (n >= 0 && n < a.N) {
        uint32_t r = (uint32_t)bt * a.N + n;
        v = a.notes[(int64_t)r * 3 + c] * dj_keep(a.d_notes, dj_rowkey(a.d_notes, r), c);
      }
      xin[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < a.N * (XCOL_LD / 8); i += 256) {      // XCOL_LD/8 16-byte vectors per note
      const int n = i / (XCOL_LD / 8), q0 = (i % (XCOL_LD / 8)) * 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = q0 + e;
        v[e] = q < CONV_K * CONV_C ? xin[(n + q / CONV_C) * 3 + q % CONV_C] : 0.f;
      }
      store8(Xcol + dj_row_ta(b, t, n, a.T, a.N) * XCOL_LD + q0, v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void feature_asm_kernel(FeatArgs a, T* __restrict__ X, T* __restrict__ Y,
                                                          int store_y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  uint32_t* rkc = (uint32_t*)fsm;                          // [N] conv-dropout row keys
  uint32_t* rks = rkc + a.N;                               // [N] style-dropout row keys
  T* xrow = (T*)(fsm + ((size_t)(2 * a.N) * 4 + 15) / 16 * 16);   // [FEAT_NC][FP]
  const int tid = threadIdx.x;
  const int conv_col0 = 2 + a.octave;           // 14
  const int nrest = a.FP - CONV_O, vpr = a.FP / 8;

  for (int bt = blockIdx.x; bt < a.B * a.T; bt += gridDim.x) {
    const int b = bt / a.T, t = bt % a.T;
    __syncthreads();
    for (int n = tid; n < a.N; n += 256) {
      const uint32_t r = (uint32_t)bt * a.N + n;
      rkc[n] = dj_rowkey(a.d_conv, r);
      rks[n] = dj_rowkey(a.d_style, r);
    }
    __syncthreads();
    for (int n0 = 0; n0 < a.N; n0 += FEAT_NC) {
      const int nc = a.N - n0 < FEAT_NC ? a.N - n0 : FEAT_NC;
      // conv columns: thread = (8-output chunk oc, note slot nn); 16-byte accesses to Y, 4 notes per thread in flight
      {
        const int oc = tid & 7, nn = tid >> 3;
        float sp8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sp8[e] = a.sp0[(int64_t)bt * a.F + conv_col0 + oc * 8 + e];
#pragma unroll
        for (int it = 0; it < FEAT_NC / 32; ++it) {
          const int n = n0 + it * 32 + nn;
          if (n < n0 + nc) {
            T* yp = Y + dj_row_ta(b, t, n, a.T, a.N) * CONV_O + oc * 8;
            float y[8];
            load8(yp, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = dj_tanh(y[e]);
            if (store_y) store8(yp, y);
            T* xp = xrow + (n - n0) * a.FP + conv_col0 + oc * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              xp[e] = dj_from_f32<T>(y[e] * dj_keep(a.d_conv, rkc[n], oc * 8 + e) +
                                     sp8[e] * dj_keep(a.d_style, rks[n], conv_col0 + oc * 8 + e));
          }
        }
      }
      // the remaining columns: [0, 14) and [14+64, FP)
      if (nrest == 32) {
        // fast path (FP = 96): a thread keeps ONE column q = tid & 31 for all its notes, so everything that does not
        // depend on the note is computed once per (b, t) and no index needs a division
        const int q = tid & 31, col = q < conv_col0 ? q : q + CONV_O;
        float add = 0.f;                               // note-independent part of the value
        if (col > a.octave + 1 && col < a.F) {         // beat, model.py:66
          const int j = col - conv_col0 - CONV_O;
          add = a.beat[(int64_t)bt * a.NB + j] * dj_keep(a.d_beat, dj_rowkey(a.d_beat, bt), j);
        }
        const float spc = col < a.F ? a.sp0[(int64_t)bt * a.F + col] : 0.f;
        // pitch_bins quirk (model.py:43-49): flat index f = bt N + n, value bins[((f / BT) % octave) BT + f % BT]
        const int64_t bT = (int64_t)a.Bfull * a.T, f0 = (int64_t)(a.bt0 + bt) * a.N + n0;
        const int64_t fr = f0 % bT;                    // uniform
        const int fqm = (int)((f0 / bT) % a.octave);   // uniform; both advanced per note below WITHOUT divisions:
        // the lanes of a wave hold different columns, so every branch below is walked by the whole wave in every
        // iteration -- a 64-bit modulo in the one-lane bins branch cost more than the rest of the loop
        int nm = (n0 + (tid >> 5)) % a.octave;         // n % octave of this thread's notes (n advances by 8)
        const int nstep = 8 % a.octave;
        for (int nl = tid >> 5; nl < nc; nl += 8) {
          const int n = n0 + nl;
          float v = add;
          if (col == 0) {
            v = (float)n / (float)a.N;                              // model.py:22-30
          } else if (col <= a.octave) {
            v = (nm == col - 1) ? 1.f : 0.f;                        // model.py:32-41
          } else if (col == a.octave + 1) {
            int qm = fqm;
            int64_t r2 = fr + nl;
            while (r2 >= bT) {
              r2 -= bT;
              if (++qm == a.octave) qm = 0;
            }
            v = a.bins[qm * bT + r2];
          }
          nm += nstep;
          if (nm >= a.octave) nm -= a.octave;
          if (col < a.F) v += spc * dj_keep(a.d_style, rks[n], col);
          xrow[nl * a.FP + col] = dj_from_f32<T>(v);
        }
      } else
      for (int i = tid; i < nc * nrest; i += 256) {
        const int nl = i / nrest, q = i - nl * nrest, n = n0 + nl;
        const int col = q < conv_col0 ? q : q + CONV_O;
        float v = 0.f;
        if (col == 0) {
          v = (float)n / (float)a.N;                               // model.py:22-30
        } else if (col <= a.octave) {
          v = ((n % a.octave) == col - 1) ? 1.f : 0.f;             // model.py:32-41
        } else if (col == a.octave + 1) {                          // model.py:43-49 raw-reshape quirk
          int64_t f = (int64_t)(a.bt0 + bt) * a.N + n, bT = (int64_t)a.Bfull * a.T;
          v = a.bins[((f / bT) % a.octave) * bT + (f % bT)];
        } else if (col < a.F) {                                    // beat, model.py:66
          int j = col - conv_col0 - CONV_O;
          v = a.beat[(int64_t)bt * a.NB + j] * dj_keep(a.d_beat, dj_rowkey(a.d_beat, bt), j);
        }
        if (col < a.F) v += a.sp0[(int64_t)bt * a.F + col] * dj_keep(a.d_style, rks[n], col);
        xrow[nl * a.FP + col] = dj_from_f32<T>(v);
      }
      __syncthreads();
      // rows out: FP/8 16-byte vectors per note row
      for (int i = tid; i < nc * vpr; i += 256) {
        const int nl = i / vpr, cv = (i - nl * vpr) * 8;
        const T* src = xrow + nl * a.FP + cv;
        T* dst = X + dj_row_ta(b, t, n0 + nl, a.T, a.N) * a.FP + cv;
        if constexpr (sizeof(T) == 2) {
          *(uint4*)dst = *(const uint4*)src;
        } else {
          ((uint4*)dst)[0] = ((const uint4*)src)[0];
          ((uint4*)dst)[1] = ((const uint4*)src)[1];
        }
      }
      __syncthreads();
    }
  }
}

// backward of the above.  One workgroup per (b,t):
//   dpre0[bt,d]  = (sum_n dX[b,n,t,d] * keep_style) * (1 - sp^2)          (style Dense gradient)
//   Ycol[row,o] <- dX[row,14+o] * keep_conv * (1 - y^2)   in place        (conv pre-activation grad)
//   dbc[o]     += sum of that; dWc comes from dj_gemm_tn(Xcol^T, Ycol) afterwards.
template <typename T>
__global__ __launch_bounds__(256) void feature_bwd_kernel(FeatArgs a, const T* __restrict__ dX, T* __restrict__ Ycol,
                                                          float* __restrict__ dbc, float* __restrict__ dpre0) {
  // the dX rows of one (b,t) (FEAT_NC notes at a time) are staged in LDS once and feed both reductions
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  T* dxs = (T*)fsm;                                               // [FEAT_NC][FP]
  uint32_t* rkc = (uint32_t*)(fsm + (size_t)FEAT_NC * a.FP * sizeof(T));   // [FEAT_NC] conv-dropout row keys
  uint32_t* rks = rkc + FEAT_NC;                                  // [FEAT_NC] style-dropout row keys
  float* red = (float*)(rks + FEAT_NC);                           // [2][128] style partials, [64] conv-bias sums
  const int tid = threadIdx.x;
  const int conv_col0 = 2 + a.octave, vpr = a.FP / 8;
  const int oc = tid & 7, nn = tid >> 3;                          // conv part: (8-output chunk, note slot)
  const int d = tid % 128, part = tid / 128;                      // style part: (column, note half)
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (tid < 64) red[256 + tid] = 0.f;
  for (int bt = blockIdx.x; bt < a.B * a.T; bt += gridDim.x) {
    const int b = bt / a.T, t = bt % a.T;
    float ssum = 0.f;
    for (int n0 = 0; n0 < a.N; n0 += FEAT_NC) {
      const int nc = a.N - n0 < FEAT_NC ? a.N - n0 : FEAT_NC;
      __syncthreads();
      for (int i = tid; i < nc * vpr; i += 256) {
        const int nl = i / vpr, cv = (i - nl * vpr) * 8;
        const T* src = dX + dj_row_ta(b, t, n0 + nl, a.T, a.N) * a.FP + cv;
        T* dst = dxs + nl * a.FP + cv;
        if constexpr (sizeof(T) == 2) {
          *(uint4*)dst = *(const uint4*)src;
        } else {
          ((uint4*)dst)[0] = ((const uint4*)src)[0];
          ((uint4*)dst)[1] = ((const uint4*)src)[1];
        }
      }
      for (int nl = tid; nl < nc; nl += 256) {
        const uint32_t r = (uint32_t)bt * a.N + n0 + nl;
        rkc[nl] = dj_rowkey(a.d_conv, r);
        rks[nl] = dj_rowkey(a.d_style, r);
      }
      __syncthreads();
      if (d < a.F) {                                              // style gradient: sum over the notes
        const int nh = (nc + 1) / 2;
        for (int nl = part * nh; nl < (part + 1) * nh && nl < nc; ++nl)
          ssum += dj_to_f32(dxs[nl * a.FP + d]) * dj_keep(a.d_style, rks[nl], d);
      }
#pragma unroll
      for (int it = 0; it < FEAT_NC / 32; ++it) {                 // conv pre-activation gradient, in place over Ycol
        const int nl = it * 32 + nn;
        if (nl < nc) {
          T* yp = Ycol + dj_row_ta(b, t, n0 + nl, a.T, a.N) * CONV_O + oc * 8;
          float y[8];
          load8(yp, y);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float g = dj_to_f32(dxs[nl * a.FP + conv_col0 + oc * 8 + e]) *
                            dj_keep(a.d_conv, rkc[nl], oc * 8 + e) * (1.f - y[e] * y[e]);
            y[e] = g;
            bs[e] += g;
          }
          store8(yp, y);
        }
      }
    }
    if (d < a.F) red[part * 128 + d] = ssum;
    __syncthreads();
    if (tid < a.F) {
      const float sp = a.sp0[(int64_t)bt * a.F + tid];
      dpre0[(int64_t)bt * a.F + tid] = (red[tid] + red[128 + tid]) * (1.f - sp * sp);
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) atomicAdd(&red[256 + oc * 8 + e], bs[e]);
  __syncthreads();
  if (tid < 64) atomicAdd(dbc + tid, red[256 + tid]);
}

// ------------------------------------------------------------------ inter-layer glue

// x_next = concat[drop(h), shift(drop(chosen))] + drop(tanh-style)   (model.py:85,77-82 / 101-117)
// One workgroup per (b,t), two passes: (1) the 8-column chunks that lie wholly inside h -- thread = (chunk, note group),
// notes strided by the group count: no per-element index division, 16-byte accesses, the style row read once per
// thread; (2) the remaining chunk(s) (note layer 0: the three `chosen` columns + padding), thread = (note, chunk).
// With both kinds in one loop the few tail lanes of every wave dragged the wave through their per-element path (three
// dependent loads and hashes) in every iteration: the 264-column launch ran at 2 TB/s, the others at 4.6-5.6.
template <typename T>
__global__ __launch_bounds__(256) void glue_fwd_kernel(GlueArgs a, const T* __restrict__ Hin, T* __restrict__ X) {
  const int bt = blockIdx.x, t = bt % a.T, b = bt / a.T;
  const int chunks = a.DP / 8, nwh = a.Hd / 8;            // all chunks / chunks wholly inside h
  if (nwh > 0) {
    const int groups = 256 / nwh;
    const int ch = threadIdx.x % nwh, grp = threadIdx.x / nwh;
    if (grp < groups) {
      const int d0 = ch * 8;
      float spv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) spv[e] = (a.sp && d0 + e < a.D) ? a.sp[(int64_t)bt * a.D + d0 + e] : 0.f;
      for (int n = grp; n < a.N; n += groups) {
        const uint32_t r = (uint32_t)bt * a.N + n;
        const int64_t rin = a.in_na ? dj_row_na(b, t, n, a.T, a.N) : dj_row_ta(b, t, n, a.T, a.N);
        const int64_t rout = a.out_na ? dj_row_na(b, t, n, a.T, a.N) : dj_row_ta(b, t, n, a.T, a.N);
        const uint32_t ko = dj_rowkey(a.d_out, r), ks = dj_rowkey(a.d_style, r);
        float v[8];
        load8(Hin + rin * a.Hd + d0, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= dj_keep(a.d_out, ko, d0 + e);
        if (a.sp) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (d0 + e < a.D) v[e] += spv[e] * dj_keep(a.d_style, ks, d0 + e);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (d0 + e >= a.D) v[e] = 0.f;
        store8(X + rout * a.DP + d0, v);
      }
    }
  }
  const int ntail = chunks - nwh;                          // chunks that straddle or follow the end of h
  for (int i = threadIdx.x; i < a.N * ntail; i += 256) {
    const int n = i / ntail, d0 = (nwh + i % ntail) * 8;
    const uint32_t r = (uint32_t)bt * a.N + n;
    const int64_t rin = a.in_na ? dj_row_na(b, t, n, a.T, a.N) : dj_row_ta(b, t, n, a.T, a.N);
    const int64_t rout = a.out_na ? dj_row_na(b, t, n, a.T, a.N) : dj_row_ta(b, t, n, a.T, a.N);
    const uint32_t ko = dj_rowkey(a.d_out, r), ks = dj_rowkey(a.d_style, r);
    const uint32_t kc = (a.chosen && n > 0) ? dj_rowkey(a.d_chosen, r - 1) : 0u;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = d0 + e;
      v[e] = 0.f;
      if (d < a.Hd)
        v[e] = dj_to_f32(Hin[rin * a.Hd + d]) * dj_keep(a.d_out, ko, d);
      else if (a.chosen && d < a.Hd + 3 && n > 0)
        v[e] = a.chosen[(int64_t)(r - 1) * 3 + (d - a.Hd)] * dj_keep(a.d_chosen, kc, d - a.Hd);
      if (a.sp && d < a.D) v[e] += a.sp[(int64_t)bt * a.D + d] * dj_keep(a.d_style, ks, d);
      if (d >= a.D) v[e] = 0.f;
    }
    store8(X + rout * a.DP + d0, v);
  }
}

// backward: dH = dX * keep_out ; dpre[bt,d] = (sum_n dX * keep_style) * (1 - sp^2)
// One workgroup per (b,t); thread = (8-column chunk, note group); 16-byte accesses.
template <typename T>
__global__ __launch_bounds__(256) void glue_bwd_kernel(GlueArgs a, const T* __restrict__ dX, T* __restrict__ dH,
                                                       float* __restrict__ dpre) {
  extern __shared__ float red[];                  // [groups][DP]
  const int bt = blockIdx.x, t = bt % a.T, b = bt / a.T;
  const int chunks = a.DP / 8, groups = 256 / chunks;
  const int ch = threadIdx.x % chunks, grp = threadIdx.x / chunks;
  const int d0 = ch * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (grp < groups) {
    for (int n = grp; n < a.N; n += groups) {
      const uint32_t r = (uint32_t)bt * a.N + n;
      const int64_t rin = a.in_na ? dj_row_na(b, t, n, a.T, a.N) : dj_row_ta(b, t, n, a.T, a.N);
      const int64_t rout = a.out_na ? dj_row_na(b, t, n, a.T, a.N) : dj_row_ta(b, t, n, a.T, a.N);
      float v[8];
      load8(dX + rout * a.DP + d0, v);
      if (a.sp) {
        const uint32_t ks = dj_rowkey(a.d_style, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += v[e] * dj_keep(a.d_style, ks, d0 + e);
      }
      if (d0 + 8 <= a.Hd) {
        const uint32_t ko = dj_rowkey(a.d_out, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= dj_keep(a.d_out, ko, d0 + e);
        store8(dH + rin * a.Hd + d0, v);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[grp * a.DP + d0 + e] = s[e];
  }
  if (!a.sp) return;
  __syncthreads();
  for (int d = threadIdx.x; d < a.D; d += 256) {
    float tot = 0.f;
    for (int gI = 0; gI < groups; ++gI) tot += red[gI * a.DP + d];
    const float sp = a.sp[(int64_t)bt * a.D + d];
    dpre[(int64_t)bt * a.D + d] = tot * (1.f - sp * sp);
  }
}

// ------------------------------------------------------------------ head + loss

__device__ __forceinline__ float bce_clip(float t, float p, float& pc, bool& inr) {
  // Keras/TF1 binary_crossentropy on probabilities
  const float eps = 1e-7f;
  inr = (p >= eps) && (p <= 1.f - eps);
  pc = fminf(fmaxf(p, eps), 1.f - eps);
  // hardware log / exp / reciprocal (v_log_f32, v_exp_f32, v_rcp_f32: ~1 ulp) instead of the library's logf / expf /
  // log1pf / IEEE division: those were ~250 of the ~670 vector instructions per group of four rows in a kernel that
  // is bound by instruction issue; 1 + e lies in (1, 2], so log1p loses nothing
  const float l = __logf(pc * __builtin_amdgcn_rcpf(1.f - pc));
  return fmaxf(l, 0.f) - l * t + __logf(1.f + __expf(-fabsf(l)));
}

// Per-row head math shared by both kernels: returns dl0..dl2 (already scaled) and the loss term.
__device__ __forceinline__ void head_row_loss(const HeadArgs& a, int64_t rr, float p0, float p1, float l2, float& dl0,
                                              float& dl1, float& dl2, float& Lv) {
  const float t0 = a.target[rr * 3], t1 = a.target[rr * 3 + 1], t2 = a.target[rr * 3 + 2];
  const float played = t0;
  float pc;
  bool inr;
  Lv = bce_clip(t0, p0, pc, inr);
  dl0 = inr ? (p0 - t0) : 0.f;
  const float pe = played * p1 + (1.f - played) * t1;
  Lv += bce_clip(t1, pe, pc, inr);
  dl1 = inr ? (pc - t1) * __builtin_amdgcn_rcpf(pc * (1.f - pc)) * played * p1 * (1.f - p1) : 0.f;
  const float ve = played * l2 + (1.f - played) * t2;
  const float diff = t2 - ve;
  Lv += diff * diff;
  dl2 = -2.f * diff * played;
  dl0 *= a.inv_count;
  dl1 *= a.inv_count;
  dl2 *= a.inv_count;
}

// Any Hd (multiple of 8, up to 64*8*CPL): one wave per note row, lane owns chunks lane + 64*j of 8 units.
template <typename T, int CPL>
__global__ __launch_bounds__(256) void head_loss_wide_kernel(HeadArgs a, const T* __restrict__ Hn, T* __restrict__ dH) {
  const int lane = threadIdx.x & 63, HD = a.Hd, nch = HD >> 3;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  float w0[CPL][8], w1[CPL][8], w2[CPL][8], g0[CPL][8], g1[CPL][8], g2[CPL][8];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int ch = lane + 64 * j;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = ch * 8 + e;
      w0[j][e] = ch < nch ? a.Wn[d * 2] : 0.f;
      w1[j][e] = ch < nch ? a.Wn[d * 2 + 1] : 0.f;
      w2[j][e] = ch < nch ? a.Wv[d] : 0.f;
      g0[j][e] = g1[j][e] = g2[j][e] = 0.f;
    }
  }
  const float b0 = a.bn[0], b1 = a.bn[1], b2 = a.bv[0];
  float gb0 = 0.f, gb1 = 0.f, gb2 = 0.f, lsum = 0.f;
  const int64_t rows = (int64_t)a.B * a.T * a.N;
  for (int64_t rr = wave; rr < rows; rr += nwaves) {
    const int n = rr % a.N, bt = rr / a.N, t = bt % a.T, b = bt / a.T;
    const int64_t row = dj_row_na(b, t, n, a.T, a.N);
    const uint32_t rk = dj_rowkey(a.d_out, (uint32_t)rr);
    float x[CPL][8], kp[CPL][8];
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int ch = lane + 64 * j;
      if (ch < nch) {
        load8(Hn + row * HD + ch * 8, x[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          kp[j][e] = dj_keep(a.d_out, rk, ch * 8 + e);
          x[j][e] *= kp[j][e];
          l0 += x[j][e] * w0[j][e];
          l1 += x[j][e] * w1[j][e];
          l2 += x[j][e] * w2[j][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[j][e] = kp[j][e] = 0.f;
      }
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
      l0 += __shfl_xor(l0, sft);
      l1 += __shfl_xor(l1, sft);
      l2 += __shfl_xor(l2, sft);
    }
    l0 += b0;
    l1 += b1;
    l2 += b2;
    const float p0 = dj_sigmoid(l0), p1 = dj_sigmoid(l1);
    if (a.out && lane == 0) {
      a.out[(int64_t)rr * 3] = p0;
      a.out[(int64_t)rr * 3 + 1] = p1;
      a.out[(int64_t)rr * 3 + 2] = l2;
    }
    if (!a.target) continue;
    float dl0, dl1, dl2, Lv;
    head_row_loss(a, rr, p0, p1, l2, dl0, dl1, dl2, Lv);
    if (lane == 0) {
      lsum += Lv;
      gb0 += dl0;
      gb1 += dl1;
      gb2 += dl2;
    }
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int ch = lane + 64 * j;
      float dh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        g0[j][e] += x[j][e] * dl0;
        g1[j][e] += x[j][e] * dl1;
        g2[j][e] += x[j][e] * dl2;
        dh[e] = (dl0 * w0[j][e] + dl1 * w1[j][e] + dl2 * w2[j][e]) * kp[j][e];
      }
      if (dH && ch < nch) store8(dH + row * HD + ch * 8, dh);
    }
  }
  if (!a.target) return;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int ch = lane + 64 * j;
    if (ch >= nch) continue;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = ch * 8 + e;
      atomicAdd(a.dWn + d * 2, g0[j][e]);
      atomicAdd(a.dWn + d * 2 + 1, g1[j][e]);
      atomicAdd(a.dWv + d, g2[j][e]);
    }
  }
  if (lane == 0) {
    atomicAdd(a.dbn, gb0);
    atomicAdd(a.dbn + 1, gb1);
    atomicAdd(a.dbv, gb2);
    if (a.loss) atomicAdd(a.loss, lsum * a.inv_count);
  }
}

// HD/8 lanes per note row (each lane 8 hidden units, one 16-byte load), 64/(HD/8) rows per wave.
template <typename T, int HD>
__global__ __launch_bounds__(256) void head_loss_kernel(HeadArgs a, const T* __restrict__ Hn, T* __restrict__ dH) {
  constexpr int LPR = HD / 8, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR, slot = lane / LPR, d0 = sub * 8;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  float w0[8], w1[8], w2[8], g0[8], g1[8], g2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    w0[e] = a.Wn[(d0 + e) * 2];
    w1[e] = a.Wn[(d0 + e) * 2 + 1];
    w2[e] = a.Wv[d0 + e];
    g0[e] = g1[e] = g2[e] = 0.f;
  }
  const float b0 = a.bn[0], b1 = a.bn[1], b2 = a.bv[0];
  float gb0 = 0.f, gb1 = 0.f, gb2 = 0.f, lsum = 0.f;
  // 32-bit row arithmetic (the launcher refuses more than 2^31 rows): the three 64-bit divisions per row that stood
  // here cost more instructions than the head itself and 50 registers (152 VGPRs = 3 waves per SIMD)
  const uint32_t rows = (uint32_t)a.B * (uint32_t)a.T * (uint32_t)a.N;
  const uint32_t stride = (uint32_t)(nwaves * RPW), uN = (uint32_t)a.N, uT = (uint32_t)a.T;
  // With ONE request per wave in flight -- 12 waves x 1 KiB per compute unit -- the kernel ran at 1.7 TB/s, a third of
  // what the same bytes reach when enough of them are on their way (round 4: 320 us per step at the baseline shape).
  struct Loc { uint32_t rr; bool live; int64_t row; uint32_t rk; };
  auto locate = [&](uint32_t base) {
    Loc q;
    const uint32_t r = base + slot;
    q.live = r < rows;
    q.rr = q.live ? r : rows - 1;
    const uint32_t ubt = q.rr / uN;
    const int n = (int)(q.rr - ubt * uN), b = (int)(ubt / uT), t = (int)(ubt - (uint32_t)b * uT);
    q.row = dj_row_na(b, t, n, a.T, a.N);
    q.rk = dj_rowkey(a.d_out, (uint32_t)q.rr);
    return q;
  };
  // one group of RPW rows: `raw` holds the lane's 8 elements of its row, tg the row's three targets
  auto row_step = [&](const Loc& c, const Raw8<T>& raw, const float (&tg)[3]) {
    const uint32_t rr = c.rr, rk = c.rk;
    const bool live = c.live;
    const int64_t row = c.row;
    const float tg0 = tg[0], tg1 = tg[1], tg2 = tg[2];
    float x[8], kp[8];
    raw.expand(x);
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      kp[e] = dj_keep(a.d_out, rk, d0 + e);
      x[e] *= kp[e];
      l0 += x[e] * w0[e];
      l1 += x[e] * w1[e];
      l2 += x[e] * w2[e];
    }
    if constexpr (LPR == 16) {          // a row is one DPP row
      l0 = dj_row16_sum(l0);
      l1 = dj_row16_sum(l1);
      l2 = dj_row16_sum(l2);
    } else {
#pragma unroll
      for (int sft = LPR / 2; sft > 0; sft >>= 1) {
        l0 += __shfl_xor(l0, sft);
        l1 += __shfl_xor(l1, sft);
        l2 += __shfl_xor(l2, sft);
      }
    }
    l0 += b0;
    l1 += b1;
    l2 += b2;
    const float p0 = dj_sigmoid(l0), p1 = dj_sigmoid(l1);
    if (a.out && live && sub == 0) {
      a.out[(int64_t)rr * 3] = p0;
      a.out[(int64_t)rr * 3 + 1] = p1;
      a.out[(int64_t)rr * 3 + 2] = l2;
    }
    if (!a.target || !live) return;
    const float t0 = tg0, t1 = tg1, t2 = tg2;
    const float played = t0;
    float pc;
    bool inr;
    float Lv = bce_clip(t0, p0, pc, inr);
    float dl0 = inr ? (p0 - t0) : 0.f;
    const float pe = played * p1 + (1.f - played) * t1;
    Lv += bce_clip(t1, pe, pc, inr);
    float dl1 = inr ? (pc - t1) * __builtin_amdgcn_rcpf(pc * (1.f - pc)) * played * p1 * (1.f - p1) : 0.f;
    const float ve = played * l2 + (1.f - played) * t2;
    const float diff = t2 - ve;
    Lv += diff * diff;
    float dl2 = -2.f * diff * played;
    dl0 *= a.inv_count;
    dl1 *= a.inv_count;
    dl2 *= a.inv_count;
    if (sub == 0) {
      lsum += Lv;
      gb0 += dl0;
      gb1 += dl1;
      gb2 += dl2;
    }
    float dh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      g0[e] += x[e] * dl0;
      g1[e] += x[e] * dl1;
      g2[e] += x[e] * dl2;
      dh[e] = (dl0 * w0[e] + dl1 * w1[e] + dl2 * w2[e]) * kp[e];
    }
    if (dH) store8(dH + row * HD + d0, dh);
  };
  auto request = [&](uint32_t base, Loc& c, Raw8<T>& raw, float (&tg)[3]) {
    c = locate(base);
    raw.load(Hn + c.row * HD + d0);
    if (a.target) {
      tg[0] = a.target[(int64_t)c.rr * 3]; tg[1] = a.target[(int64_t)c.rr * 3 + 1]; tg[2] = a.target[(int64_t)c.rr * 3 + 2];
    }
  };
  // NB groups of RPW rows per iteration: all their requests go out first, straight-line (rows past the end are clamped,
  // not masked: a conditional request is an exec branch and makes the compiler's waits conservative), then the groups
  // are worked off one after the other -- NB KiB per wave in flight instead of one
  constexpr int NB = 4;
#pragma unroll 1
  for (uint32_t base = (uint32_t)wave * (RPW * NB); base < rows; base += stride * NB) {
    Loc c[NB];
    Raw8<T> raw[NB];
    float tg[NB][3];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      tg[k][0] = tg[k][1] = tg[k][2] = 0.f;
      request(base + k * RPW, c[k], raw[k], tg[k]);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) row_step(c[k], raw[k], tg[k]);
  }
  if (!a.target) return;
  // fold the RPW row slots of the wave
#pragma unroll
  for (int sft = LPR; sft < 64; sft <<= 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      g0[e] += __shfl_xor(g0[e], sft);
      g1[e] += __shfl_xor(g1[e], sft);
      g2[e] += __shfl_xor(g2[e], sft);
    }
    gb0 += __shfl_xor(gb0, sft);
    gb1 += __shfl_xor(gb1, sft);
    gb2 += __shfl_xor(gb2, sft);
    lsum += __shfl_xor(lsum, sft);
  }
  // fold the block's waves in LDS, then one coalesced run of global atomics per block
  __shared__ float red[3 * HD + 4];
  for (int i = threadIdx.x; i < 3 * HD + 4; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
  if (slot == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      atomicAdd(&red[(d0 + e) * 2], g0[e]);
      atomicAdd(&red[(d0 + e) * 2 + 1], g1[e]);
      atomicAdd(&red[2 * HD + d0 + e], g2[e]);
    }
  }
  if (lane == 0) {
    atomicAdd(&red[3 * HD], gb0);
    atomicAdd(&red[3 * HD + 1], gb1);
    atomicAdd(&red[3 * HD + 2], gb2);
    atomicAdd(&red[3 * HD + 3], lsum);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * HD; i += blockDim.x) atomicAdd(a.dWn + i, red[i]);
  for (int i = threadIdx.x; i < HD; i += blockDim.x) atomicAdd(a.dWv + i, red[2 * HD + i]);
  if (threadIdx.x == 0) {
    atomicAdd(a.dbn, red[3 * HD]);
    atomicAdd(a.dbn + 1, red[3 * HD + 1]);
    atomicAdd(a.dbv, red[3 * HD + 2]);
    if (a.loss) atomicAdd(a.loss, red[3 * HD + 3] * a.inv_count);
  }
}

// ------------------------------------------------------------------ weight conversion
// out[n*ld + k] = (k < K) ? W[k*N + n] : 0     (Keras [in,out] -> k-contiguous Bt for dj_gemm_nt)
template <typename T>
__global__ void cvt_transpose_kernel(const float* __restrict__ W, int K, int N, T* __restrict__ out, int ld) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * ld) return;
  int n = idx / ld, k = idx % ld;
  out[idx] = dj_from_f32<T>(k < K ? W[(int64_t)k * N + n] : 0.f);
}
// B operand of the per-step forward GEMM whose epilogue is the cell (dj_kernels.h CellEpi): [W ; U]^T with its rows (the
// gate columns) interleaved in groups of 8 units x 4 gates and k = [input columns, zero pad to K1p, hidden units]
template <typename T>
__global__ void pack_wu_gates_kernel(const float* __restrict__ W, const float* __restrict__ U, int D, int K1p, int H,
                                     T* __restrict__ out) {
  const int ld = K1p + H;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)4 * H * ld) return;
  const int np = (int)(idx / ld), k = (int)(idx % ld);
  const int G = np >> 5, g = (np >> 3) & 3, e = np & 7;
  const int64_t col = (int64_t)g * H + 8 * G + e;
  float v = 0.f;
  if (k < D) v = W[(int64_t)k * 4 * H + col];
  else if (k >= K1p) v = U[(int64_t)(k - K1p) * 4 * H + col];
  out[idx] = dj_from_f32<T>(v);
}
template <typename T> __global__ void cvt_copy_kernel(const float* __restrict__ W, int64_t n, T* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) out[idx] = dj_from_f32<T>(W[idx]);
}
template <typename T> __global__ void cvt_to_f32_kernel(const T* __restrict__ in, int64_t n, float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) out[idx] = dj_to_f32(in[idx]);
}

// ------------------------------------------------------------------ Nadam (Keras 2.x, model.py:152)
__global__ void nadam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, NadamArgs a) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i] * a.gscale;
  float gp = gi / (1.f - a.ms_new);
  float mt = a.beta1 * m[i] + (1.f - a.beta1) * gi;
  float mp = mt / (1.f - a.ms_next);
  float vt = a.beta2 * v[i] + (1.f - a.beta2) * gi * gi;
  float vp = vt / a.bc2;
  float mbar = (1.f - a.mu_t) * gp + a.mu_t1 * mp;
  m[i] = mt;
  v[i] = vt;
  p[i] = p[i] - a.lr * mbar / (sqrtf(vp) + a.eps);
}

// scatter h from TA row order to the canonical [B,T,N,H] fp32 layout (time_model.predict output)
template <typename T>
__global__ void ta_to_canonical_kernel(const T* __restrict__ Hin, float* __restrict__ out, int B, int T_, int N,
                                       int Hd) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * T_ * N * Hd) return;
  int d = idx % Hd;
  int64_t r = idx / Hd;
  int n = r % N, bt = r / N, t = bt % T_, b = bt / T_;
  out[idx] = dj_to_f32(Hin[dj_row_ta(b, t, n, T_, N) * Hd + d]);
}
// canonical fp32 [B,T,N,Hd] features -> NA row order operand buffer (note_model.predict input)
template <typename T>
__global__ void canonical_to_na_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int T_, int N,
                                       int Hd) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * T_ * N * Hd) return;
  int d = idx % Hd;
  int64_t r = idx / Hd;
  int n = r % N, bt = r / N, t = bt % T_, b = bt / T_;
  out[dj_row_na(b, t, n, T_, N) * Hd + d] = dj_from_f32<T>(in[idx]);
}

inline unsigned nblk(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

// ------------------------------------------------------------------ launchers
#define DJ_T_DISPATCH(expr_f32, expr_bf16) \
  if (dtype == DJ_F32) { expr_f32; } else { expr_bf16; }

int dj_launch_dense_small(const float* A, int M, int K, const float* W, const float* b, float* C, int N, int act_tanh,
                          hipStream_t st) {
  hipLaunchKernelGGL(dense_small_kernel, dim3(nblk((int64_t)M * N)), dim3(256), 0, st, A, M, K, W, b, C, N, act_tanh);
  return (int)hipGetLastError();
}
int dj_launch_dense_small_bwd_x(const float* dC, int M, int N, const float* W, int K, float* dA, int accumulate,
                                hipStream_t st) {
  if (K > 64) return 1020;
  const int NC = N < DSBX_NC ? N : DSBX_NC;
  const size_t smb = ((size_t)NC * (K + 1) + DSBX_RB * (size_t)NC) * sizeof(float);
  static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)dense_small_bwd_x_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  if (smb > 96 * 1024) return 1026;
  hipLaunchKernelGGL(dense_small_bwd_x_kernel, dim3((M + DSBX_RB - 1) / DSBX_RB), dim3(256), smb, st, dC, M, N, W, K, dA,
                     accumulate);
  return (int)hipGetLastError();
}
int dj_launch_dense_small_bwd_w(const float* A, int M, int K, const float* dC, int N, float* dW, float* db,
                                hipStream_t st) {
  if (K > 64) return 1020;
  const int rpb = 32;
  dim3 grid((M + rpb - 1) / rpb, (N + 63) / 64);
  hipLaunchKernelGGL(dense_small_bwd_w_kernel, grid, dim3(256), (size_t)rpb * K * sizeof(float), st, A, M, K, dC, N, dW,
                     db, rpb);
  return (int)hipGetLastError();
}
int dj_launch_dense_small_batch(const DenseBatch* d, int act_tanh, hipStream_t st) {
  if (d->n < 1 || d->n > DJ_DENSE_BATCH_MAX) return 1028;
  int maxn = 0;
  for (int l = 0; l < d->n; ++l) maxn = d->N[l] > maxn ? d->N[l] : maxn;
  if (d->K > 64) return 1020;
  hipLaunchKernelGGL(dense_small_batch_kernel, dim3((maxn + 63) / 64, (d->M + 31) / 32, d->n), dim3(256), 0, st, *d,
                     act_tanh);
  return (int)hipGetLastError();
}
// gradients of a batch: dW_l += A^T dC_l, db_l += colsum(dC_l); dA += sum_l dC_l W_l^T (dA must be zeroed by the caller)
int dj_launch_dense_small_batch_bwd(const DenseBatch* d, float* dA, hipStream_t st) {
  if (d->n < 1 || d->n > DJ_DENSE_BATCH_MAX || d->K > 64) return 1028;
  int maxn = 0;
  for (int l = 0; l < d->n; ++l) maxn = d->N[l] > maxn ? d->N[l] : maxn;
  const int rpb = 64;      // measured: 32 -> 120 us, 64 -> 92 us, 128 -> 135 us for the four layers
  hipLaunchKernelGGL(dense_small_bwd_w_batch_kernel, dim3((d->M + rpb - 1) / rpb, (maxn + 63) / 64, d->n), dim3(256),
                     (size_t)rpb * d->K * sizeof(float), st, *d, rpb);
  const int NC = maxn < DSBX_NC ? maxn : DSBX_NC;
  const size_t smb = ((size_t)NC * (d->K + 1) + DSBX_RB * (size_t)NC) * sizeof(float);
  static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)dense_small_bwd_x_batch_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  if (smb > 96 * 1024) return 1026;
  hipLaunchKernelGGL(dense_small_bwd_x_batch_kernel, dim3((d->M + DSBX_RB - 1) / DSBX_RB, d->n), dim3(256), smb, st, *d,
                     dA);
  return (int)hipGetLastError();
}
int dj_launch_bins(const float* notes, float* bins, int B, int T, int N, int octave, DjDrop dn, hipStream_t st) {
  hipLaunchKernelGGL(bins_kernel, dim3(nblk((int64_t)octave * B * T)), dim3(256), 0, st, notes, bins, B, T, N, octave,
                     dn);
  return (int)hipGetLastError();
}
// steps (1) and (3) of the feature assembly; the caller runs the conv GEMM (2) in between
int dj_launch_feature_xcol(int dtype, const void* fa, void* Xcol, hipStream_t st) {
  const FeatArgs& a = *(const FeatArgs*)fa;
  const int XR = (a.N + 3) / 4 * 4 + CONV_K + 3;
  const size_t smem = (size_t)XR * 3 * sizeof(float);
  if (smem > 64 * 1024) return 1027;
  int grid = a.B * a.T < 4096 ? a.B * a.T : 4096;
  DJ_T_DISPATCH(hipLaunchKernelGGL(feature_xcol_kernel<float>, dim3(grid), dim3(256), smem, st, a, (float*)Xcol),
                hipLaunchKernelGGL(feature_xcol_kernel<bf16_t>, dim3(grid), dim3(256), smem, st, a, (bf16_t*)Xcol))
  return (int)hipGetLastError();
}
int dj_launch_feature_asm(int dtype, const void* fa, void* X, void* Y, int store_y, hipStream_t st) {
  const FeatArgs& a = *(const FeatArgs*)fa;
  if (a.FP - CONV_O > 64 || a.FP % 8) return 1021;
  const size_t esz = dtype == DJ_F32 ? 4 : 2;
  const size_t smem = ((size_t)(2 * a.N) * 4 + 15) / 16 * 16 + (size_t)FEAT_NC * a.FP * esz;
  if (smem > 64 * 1024) return 1027;
  int grid = a.B * a.T < 4096 ? a.B * a.T : 4096;
  DJ_T_DISPATCH(hipLaunchKernelGGL(feature_asm_kernel<float>, dim3(grid), dim3(256), smem, st, a, (float*)X, (float*)Y,
                                   store_y),
                hipLaunchKernelGGL(feature_asm_kernel<bf16_t>, dim3(grid), dim3(256), smem, st, a, (bf16_t*)X,
                                   (bf16_t*)Y, store_y))
  return (int)hipGetLastError();
}
int dj_launch_feature_bwd(int dtype, const void* fa, const void* dX, void* Ycol, float* dbc, float* dpre0,
                          hipStream_t st) {
  const FeatArgs& a = *(const FeatArgs*)fa;
  if (a.F > 128) return 1022;
  int grid = a.B * a.T < 2048 ? a.B * a.T : 2048;
  if (a.FP % 8) return 1021;
  const size_t esz = dtype == DJ_F32 ? 4 : 2;
  const size_t smem = (size_t)FEAT_NC * a.FP * esz + (size_t)2 * FEAT_NC * 4 + (size_t)(256 + 64) * 4;
  if (smem > 64 * 1024) return 1027;
  DJ_T_DISPATCH(hipLaunchKernelGGL(feature_bwd_kernel<float>, dim3(grid), dim3(256), smem, st, a, (const float*)dX,
                                   (float*)Ycol, dbc, dpre0),
                hipLaunchKernelGGL(feature_bwd_kernel<bf16_t>, dim3(grid), dim3(256), smem, st, a, (const bf16_t*)dX,
                                   (bf16_t*)Ycol, dbc, dpre0))
  return (int)hipGetLastError();
}
int dj_launch_glue_fwd(int dtype, const void* ga, const void* Hin, void* X, hipStream_t st) {
  const GlueArgs& a = *(const GlueArgs*)ga;
  if (a.DP % 8 || a.DP > 2048) return 1023;
  DJ_T_DISPATCH(hipLaunchKernelGGL(glue_fwd_kernel<float>, dim3(a.B * a.T), dim3(256), 0, st, a, (const float*)Hin,
                                   (float*)X),
                hipLaunchKernelGGL(glue_fwd_kernel<bf16_t>, dim3(a.B * a.T), dim3(256), 0, st, a, (const bf16_t*)Hin,
                                   (bf16_t*)X))
  return (int)hipGetLastError();
}
int dj_launch_glue_bwd(int dtype, const void* ga, const void* dX, void* dH, float* dpre, hipStream_t st) {
  const GlueArgs& a = *(const GlueArgs*)ga;
  if (a.DP % 8 || a.DP > 2048 || a.Hd % 8) return 1025;
  const size_t sm = (size_t)(256 / (a.DP / 8)) * a.DP * sizeof(float);
  DJ_T_DISPATCH(hipLaunchKernelGGL(glue_bwd_kernel<float>, dim3(a.B * a.T), dim3(256), sm, st, a, (const float*)dX,
                                   (float*)dH, dpre),
                hipLaunchKernelGGL(glue_bwd_kernel<bf16_t>, dim3(a.B * a.T), dim3(256), sm, st, a, (const bf16_t*)dX,
                                   (bf16_t*)dH, dpre))
  return (int)hipGetLastError();
}
int dj_launch_head(int dtype, const void* ha, const void* Hn, void* dH, hipStream_t st) {
  const HeadArgs& a = *(const HeadArgs*)ha;
  int64_t rows = (int64_t)a.B * a.T * a.N;
  if (rows >= ((int64_t)1 << 31) - (1 << 20)) return 1026;      // 32-bit row arithmetic in the kernels
  int grid = (int)((rows + 15) / 16 < 2048 ? (rows + 15) / 16 : 2048);
  if (a.Hd == 128) {
    DJ_T_DISPATCH(hipLaunchKernelGGL((head_loss_kernel<float, 128>), dim3(grid), dim3(256), 0, st, a, (const float*)Hn,
                                     (float*)dH),
                  hipLaunchKernelGGL((head_loss_kernel<bf16_t, 128>), dim3(grid), dim3(256), 0, st, a,
                                     (const bf16_t*)Hn, (bf16_t*)dH))
  } else if (a.Hd == 256) {
    DJ_T_DISPATCH(hipLaunchKernelGGL((head_loss_kernel<float, 256>), dim3(grid), dim3(256), 0, st, a, (const float*)Hn,
                                     (float*)dH),
                  hipLaunchKernelGGL((head_loss_kernel<bf16_t, 256>), dim3(grid), dim3(256), 0, st, a,
                                     (const bf16_t*)Hn, (bf16_t*)dH))
  } else {
    // generic width: one wave per row; a small persistent grid keeps the closing weight-gradient atomics cheap
    if (a.Hd % 8 || a.Hd > 2048) return 1024;
    const int g2 = (int)((rows + 3) / 4 < 512 ? (rows + 3) / 4 : 512);
#define DJ_HEAD_WIDE(CPL)                                                                                              \
  DJ_T_DISPATCH(hipLaunchKernelGGL((head_loss_wide_kernel<float, CPL>), dim3(g2), dim3(256), 0, st, a,                 \
                                   (const float*)Hn, (float*)dH),                                                      \
                hipLaunchKernelGGL((head_loss_wide_kernel<bf16_t, CPL>), dim3(g2), dim3(256), 0, st, a,                \
                                   (const bf16_t*)Hn, (bf16_t*)dH))
    if (a.Hd <= 512) {
      DJ_HEAD_WIDE(1)
    } else if (a.Hd <= 1024) {
      DJ_HEAD_WIDE(2)
    } else {
      DJ_HEAD_WIDE(4)
    }
#undef DJ_HEAD_WIDE
  }
  return (int)hipGetLastError();
}
int dj_launch_cvt_transpose(int dtype, const float* W, int K, int N, void* out, int ld, hipStream_t st) {
  int64_t n = (int64_t)N * ld;
  DJ_T_DISPATCH(hipLaunchKernelGGL(cvt_transpose_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, W, K, N, (float*)out,
                                   ld),
                hipLaunchKernelGGL(cvt_transpose_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, st, W, K, N,
                                   (bf16_t*)out, ld))
  return (int)hipGetLastError();
}
int dj_launch_pack_wu_gates(int dtype, const float* W, const float* U, int D, int K1p, int H, void* out, hipStream_t st) {
  if ((H % 8) || D > K1p) return 1025;
  int64_t n = (int64_t)4 * H * (K1p + H);
  DJ_T_DISPATCH(hipLaunchKernelGGL(pack_wu_gates_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, W, U, D, K1p, H, (float*)out),
                hipLaunchKernelGGL(pack_wu_gates_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, st, W, U, D, K1p, H,
                                   (bf16_t*)out))
  return (int)hipGetLastError();
}
int dj_launch_cvt_copy(int dtype, const float* W, int64_t n, void* out, hipStream_t st) {
  DJ_T_DISPATCH(hipLaunchKernelGGL(cvt_copy_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, W, n, (float*)out),
                hipLaunchKernelGGL(cvt_copy_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, st, W, n, (bf16_t*)out))
  return (int)hipGetLastError();
}
int dj_launch_cvt_to_f32(int dtype, const void* in, int64_t n, float* out, hipStream_t st) {
  DJ_T_DISPATCH(hipLaunchKernelGGL(cvt_to_f32_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, (const float*)in, n, out),
                hipLaunchKernelGGL(cvt_to_f32_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, st, (const bf16_t*)in, n,
                                   out))
  return (int)hipGetLastError();
}
int dj_launch_nadam(float* p, const float* g, float* m, float* v, int64_t n, const void* na, hipStream_t st) {
  hipLaunchKernelGGL(nadam_kernel, dim3(nblk(n)), dim3(256), 0, st, p, g, m, v, n, *(const NadamArgs*)na);
  return (int)hipGetLastError();
}
int dj_launch_ta_to_canonical(int dtype, const void* Hin, float* out, int B, int T, int N, int Hd, hipStream_t st) {
  int64_t n = (int64_t)B * T * N * Hd;
  DJ_T_DISPATCH(hipLaunchKernelGGL(ta_to_canonical_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, (const float*)Hin,
                                   out, B, T, N, Hd),
                hipLaunchKernelGGL(ta_to_canonical_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, st, (const bf16_t*)Hin,
                                   out, B, T, N, Hd))
  return (int)hipGetLastError();
}
int dj_launch_canonical_to_na(int dtype, const float* in, void* out, int B, int T, int N, int Hd, hipStream_t st) {
  int64_t n = (int64_t)B * T * N * Hd;
  DJ_T_DISPATCH(hipLaunchKernelGGL(canonical_to_na_kernel<float>, dim3(nblk(n)), dim3(256), 0, st, in, (float*)out, B,
                                   T, N, Hd),
                hipLaunchKernelGGL(canonical_to_na_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, st, in, (bf16_t*)out,
                                   B, T, N, Hd))
  return (int)hipGetLastError();
}
