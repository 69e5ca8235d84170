// Persistent recurrent LSTM cell kernels (forward + BPTT) for MI355X.
//
// Replaces the Keras LSTM while-loops of the reference (model.py:84 time axis,
// model.py:122 note axis; backward = TF autodiff of the same, train.py:29).
// Keras 2.x cell: z = xW + hU + b, gate column blocks i,f,c,o;
// i,f,o = hard_sigmoid (switchable to sigmoid), g = tanh; c' = f c + i g;
// h' = o tanh(c'); zero initial state.
//
// One 256-thread workgroup owns a tile of 32 independent sequences for ALL steps
// (no inter-workgroup traffic).  Wave w owns hidden units [w*H/4, (w+1)*H/4) and
// all four gate columns of those units, so the cell update is lane-local in the
// MFMA accumulator layout.  h_{t-1} lives in LDS (A operand); the recurrent
// kernel U is streamed every step from L2 as pre-packed MFMA B fragments (one
// coalesced 1 KiB wave-load per fragment).  x_t W + b arrives precomputed in Z
// (dj_gemm_nt) and is overwritten in place by the pre-activations z_t (the BPTT
// stash), which backward overwrites in place again with dz_t.
#include "dj_kernels.h"

namespace {

template <typename T, int H> struct RecCfg {
  static constexpr int EPL = 16 / sizeof(T);
  static constexpr int KC = 2 * EPL;
  static constexpr int UW = H / 4;        // units per wave
  static constexpr int NJ = UW / 32;      // 32-col tiles per gate per wave
  static constexpr int NKC = H / KC;      // k-chunks of the forward product (K = H)
  static constexpr int NKCB = 4 * H / KC; // k-chunks of the backward product (K = 4H)
  static constexpr int LDH = H + EPL;     // LDS row stride of the h tile
  static constexpr int LDZ = 4 * H + EPL; // LDS row stride of the dz tile
};

// ---------------------------------------------------------------- weight packing
// Upack[(((w*4+g)*NJ+j)*NKC + kc)*64 + lane][e] = U[kc*KC + EPL*h + e][g*H + w*UW + j*32 + l31]
template <typename T, int H>
__global__ void pack_u_fwd_kernel(const float* __restrict__ U, T* __restrict__ out) {
  using R = RecCfg<T, H>;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * 4 * H) return;
  int e = idx % R::EPL, lane = (idx / R::EPL) % 64, rest = idx / (R::EPL * 64);
  int kc = rest % R::NKC;
  rest /= R::NKC;
  int j = rest % R::NJ;
  rest /= R::NJ;
  int g = rest % 4, w = rest / 4;
  int k = kc * R::KC + R::EPL * (lane >> 5) + e;
  int col = g * H + w * R::UW + j * 32 + (lane & 31);
  out[idx] = dj_from_f32<T>(U[(int64_t)k * 4 * H + col]);
}
// UTpack[((w*NJ+j)*NKCB + kc)*64 + lane][e] = U[n = w*UW + j*32 + l31][k = kc*KC + EPL*h + e]
template <typename T, int H>
__global__ void pack_u_bwd_kernel(const float* __restrict__ U, T* __restrict__ out) {
  using R = RecCfg<T, H>;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * 4 * H) return;
  int e = idx % R::EPL, lane = (idx / R::EPL) % 64, rest = idx / (R::EPL * 64);
  int kc = rest % R::NKCB;
  rest /= R::NKCB;
  int j = rest % R::NJ, w = rest / R::NJ;
  int k = kc * R::KC + R::EPL * (lane >> 5) + e;
  int n = w * R::UW + j * 32 + (lane & 31);
  out[idx] = dj_from_f32<T>(U[(int64_t)n * 4 * H + k]);
}

// ---------------------------------------------------------------- forward
template <typename T, int H>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(T* __restrict__ Z, const T* __restrict__ Upack,
                                                       T* __restrict__ Hout, T* __restrict__ Cout, int steps, int sigm,
                                                       int store_z) {
  using R = RecCfg<T, H>;
  using Frag = typename DjFrag<T>::type;
  __shared__ __attribute__((aligned(16))) T hs[2][32 * R::LDH];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int64_t tile = blockIdx.x;

  float c[R::NJ][16];
#pragma unroll
  for (int j = 0; j < R::NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[j][r] = 0.f;

  const Frag* up = (const Frag*)Upack + (int64_t)w * 4 * R::NJ * R::NKC * 64 + lane;
  int cur = 0;
  for (int t = 0; t < steps; ++t) {
    const int64_t rowbase = (tile * steps + t) * 32;
    f32x16 acc[4][R::NJ];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < R::NJ; ++j) {
        const T* zp = Z + rowbase * (4 * H) + g * H + w * R::UW + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][j][r] = dj_to_f32(zp[(int64_t)dj_crow(r, lane) * (4 * H)]);
      }
    if (t > 0) {
      const T* hp = hs[cur] + l31 * R::LDH;
#pragma unroll 4
      for (int kc = 0; kc < R::NKC; ++kc) {
        Frag a = dj_lds_frag(hp + kc * R::KC, h);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < R::NJ; ++j) {
            Frag b = up[((g * R::NJ + j) * R::NKC + kc) * 64];
            dj_mfma(acc[g][j], a, b);
          }
      }
    }
    T* hn = hs[cur ^ 1];
#pragma unroll
    for (int j = 0; j < R::NJ; ++j) {
      const int u = w * R::UW + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = dj_crow(r, lane);
        float zi = acc[0][j][r], zf = acc[1][j][r], zg = acc[2][j][r], zo = acc[3][j][r];
        float ig = dj_ract(zi, sigm), fg = dj_ract(zf, sigm), gg = dj_tanh(zg), og = dj_ract(zo, sigm);
        float cn = fg * c[j][r] + ig * gg;
        c[j][r] = cn;
        float hv = og * dj_tanh(cn);
        hn[row * R::LDH + u] = dj_from_f32<T>(hv);
        const int64_t grow = rowbase + row;
        if (store_z) {
          T* zp = Z + grow * (4 * H) + u;
          zp[0] = dj_from_f32<T>(zi);
          zp[H] = dj_from_f32<T>(zf);
          zp[2 * H] = dj_from_f32<T>(zg);
          zp[3 * H] = dj_from_f32<T>(zo);
        }
        if (Cout) Cout[grow * H + u] = dj_from_f32<T>(cn);
      }
    }
    __syncthreads();
    // cooperative, coalesced copy of h_t (32 x H) to global
    constexpr int VPR = H / R::EPL;
#pragma unroll
    for (int v = tid; v < 32 * VPR; v += 256) {
      int row = v / VPR, cv = (v % VPR) * R::EPL;
      *(uint4*)(Hout + (rowbase + row) * H + cv) = *(const uint4*)(hn + row * R::LDH + cv);
    }
    cur ^= 1;
  }
}

// ---------------------------------------------------------------- backward (BPTT)
template <typename T, int H>
__global__ __launch_bounds__(256) void lstm_bwd_kernel(T* __restrict__ Z, const T* __restrict__ UTpack,
                                                       const T* __restrict__ C, const T* __restrict__ dH,
                                                       float* __restrict__ dbias, int steps, int sigm) {
  using R = RecCfg<T, H>;
  using Frag = typename DjFrag<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* dzs = (T*)smem_raw;   // [32][LDZ]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int64_t tile = blockIdx.x;

  float dcc[R::NJ][16];
  f32x16 acc[R::NJ];
  float dbs[4][R::NJ];
#pragma unroll
  for (int j = 0; j < R::NJ; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dcc[j][r] = 0.f;
      acc[j][r] = 0.f;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) dbs[g][j] = 0.f;
  }
  const Frag* up = (const Frag*)UTpack + (int64_t)w * R::NJ * R::NKCB * 64 + lane;

  for (int t = steps - 1; t >= 0; --t) {
    const int64_t rowbase = (tile * steps + t) * 32;
#pragma unroll
    for (int j = 0; j < R::NJ; ++j) {
      const int u = w * R::UW + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = dj_crow(r, lane);
        const int64_t grow = rowbase + row;
        const T* zp = Z + grow * (4 * H) + u;
        float zi = dj_to_f32(zp[0]), zf = dj_to_f32(zp[H]), zg = dj_to_f32(zp[2 * H]), zo = dj_to_f32(zp[3 * H]);
        float ct = dj_to_f32(C[grow * H + u]);
        float cp = (t > 0) ? dj_to_f32(C[(grow - 32) * H + u]) : 0.f;
        float dh = dj_to_f32(dH[grow * H + u]) + acc[j][r];
        float ig = dj_ract(zi, sigm), fg = dj_ract(zf, sigm), gg = dj_tanh(zg), og = dj_ract(zo, sigm);
        float tc = dj_tanh(ct);
        float dzo = dh * tc * dj_ract_grad(zo, og, sigm);
        float dc = dcc[j][r] + dh * og * (1.f - tc * tc);
        float dzi = dc * gg * dj_ract_grad(zi, ig, sigm);
        float dzf = dc * cp * dj_ract_grad(zf, fg, sigm);
        float dzg = dc * ig * (1.f - gg * gg);
        dcc[j][r] = dc * fg;
        T* dp = dzs + row * R::LDZ + u;
        dp[0] = dj_from_f32<T>(dzi);
        dp[H] = dj_from_f32<T>(dzf);
        dp[2 * H] = dj_from_f32<T>(dzg);
        dp[3 * H] = dj_from_f32<T>(dzo);
        dbs[0][j] += dzi;
        dbs[1][j] += dzf;
        dbs[2][j] += dzg;
        dbs[3][j] += dzo;
      }
    }
    __syncthreads();
    // dz_t tile -> global (in place over z_t), coalesced
    constexpr int VPR = 4 * H / R::EPL;
#pragma unroll 4
    for (int v = tid; v < 32 * VPR; v += 256) {
      int row = v / VPR, cv = (v % VPR) * R::EPL;
      *(uint4*)(Z + (rowbase + row) * (4 * H) + cv) = *(const uint4*)(dzs + row * R::LDZ + cv);
    }
    if (t > 0) {
      // dh_{t-1} (recurrent part) = dz_t [32 x 4H] * U^T [4H x H]; this wave's H/4 output units
#pragma unroll
      for (int j = 0; j < R::NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      const T* ap = dzs + l31 * R::LDZ;
#pragma unroll 8
      for (int kc = 0; kc < R::NKCB; ++kc) {
        Frag a = dj_lds_frag(ap + kc * R::KC, h);
#pragma unroll
        for (int j = 0; j < R::NJ; ++j) {
          Frag b = up[(j * R::NKCB + kc) * 64];
          dj_mfma(acc[j], a, b);
        }
      }
    }
    __syncthreads();
  }
  if (dbias) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < R::NJ; ++j) {
        float v = dbs[g][j];
        v += __shfl_xor(v, 32);
        if (h == 0) atomicAdd(dbias + g * H + w * R::UW + j * 32 + l31, v);
      }
  }
}

template <typename T, int H> int launch_pack(const float* U, void* fwd, void* bwd, hipStream_t st) {
  int n = H * 4 * H;
  dim3 grid((n + 255) / 256), block(256);
  if (fwd) hipLaunchKernelGGL((pack_u_fwd_kernel<T, H>), grid, block, 0, st, U, (T*)fwd);
  if (bwd) hipLaunchKernelGGL((pack_u_bwd_kernel<T, H>), grid, block, 0, st, U, (T*)bwd);
  return (int)hipGetLastError();
}
template <typename T, int H>
int launch_fwd(int ntiles, int steps, void* Z, const void* Upack, void* Hout, void* Cout, int sigm, int store_z,
               hipStream_t st) {
  hipLaunchKernelGGL((lstm_fwd_kernel<T, H>), dim3(ntiles), dim3(256), 0, st, (T*)Z, (const T*)Upack, (T*)Hout,
                     (T*)Cout, steps, sigm, store_z);
  return (int)hipGetLastError();
}
template <typename T, int H>
int launch_bwd(int ntiles, int steps, void* Z, const void* UTpack, const void* C, const void* dH, float* dbias,
               int sigm, hipStream_t st) {
  using R = RecCfg<T, H>;
  size_t smem = (size_t)32 * R::LDZ * sizeof(T);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_bwd_kernel<T, H>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL((lstm_bwd_kernel<T, H>), dim3(ntiles), dim3(256), smem, st, (T*)Z, (const T*)UTpack, (const T*)C,
                     (const T*)dH, dbias, steps, sigm);
  return (int)hipGetLastError();
}

}  // namespace

#define DJ_DISPATCH_TH(FN, ...)                                     \
  if (dtype == DJ_F32 && H == 256) return FN<float, 256>(__VA_ARGS__);  \
  if (dtype == DJ_F32 && H == 128) return FN<float, 128>(__VA_ARGS__);  \
  if (dtype == DJ_BF16 && H == 256) return FN<bf16_t, 256>(__VA_ARGS__); \
  if (dtype == DJ_BF16 && H == 128) return FN<bf16_t, 128>(__VA_ARGS__); \
  return 1010;

int dj_launch_lstm_pack(int dtype, int H, const float* U, void* fwd, void* bwd, hipStream_t st) {
  DJ_DISPATCH_TH(launch_pack, U, fwd, bwd, st)
}
int dj_launch_lstm_fwd(int dtype, int H, int ntiles, int steps, void* Z, const void* Upack, void* Hout, void* Cout,
                       int sigm, int store_z, hipStream_t st) {
  if (ntiles <= 0 || steps <= 0) return 0;
  DJ_DISPATCH_TH(launch_fwd, ntiles, steps, Z, Upack, Hout, Cout, sigm, store_z, st)
}
int dj_launch_lstm_bwd(int dtype, int H, int ntiles, int steps, void* Z, const void* UTpack, const void* C,
                       const void* dH, float* dbias, int sigm, hipStream_t st) {
  if (ntiles <= 0 || steps <= 0) return 0;
  DJ_DISPATCH_TH(launch_bwd, ntiles, steps, Z, UTpack, C, dH, dbias, sigm, st)
}
