// Persistent recurrent LSTM cell kernels (forward + BPTT) for MI355X.
//
// Replaces the Keras LSTM while-loops of the reference (model.py:84 time axis,
// model.py:122 note axis; backward = TF autodiff of the same, train.py:29).
// Keras 2.x cell: z = xW + hU + b, gate column blocks i,f,c,o;
// i,f,o = hard_sigmoid (switchable to sigmoid), g = tanh; c' = f c + i g;
// h' = o tanh(c'); zero initial state.
//
// One workgroup of H/32 waves owns a tile of 32 independent sequences for ALL steps
// (no inter-workgroup traffic).  Wave w owns hidden units [32w, 32w+32) and
// all four gate columns of those units, so the cell update is lane-local in the
// MFMA accumulator layout.  h_{t-1} lives in LDS (A operand); the recurrent
// kernel U is streamed every step from L2 as pre-packed MFMA B fragments (one
// coalesced 1 KiB wave-load per fragment, software-pipelined PD chunks ahead).
//
// Variants (RecCfg flags; DESIGN.md section 5 has the measurements behind each):
//  * lstm_fwd_fused_kernel: x_t W is computed inside the step from an LDS-staged X tile, W and U being one
//    continuous fragment stream (no Zx round trip through HBM);
//  * H = 128 in bf16 (one wave per SIMD, VGPR + AGPR): the wave's slices of U / U^T, of W up to 128 input
//    columns and of W^T of a <= 128-wide layer are loaded once and stay in registers (STATF / STATB / WSTAT /
//    DX), so the note-axis recurrence streams no weights; wider inputs stream W through a ring of 8 or 6
//    fragments per gate, the first ring block LDS-resident where it fits (WLDS: note layer 0);
//  * the BPTT kernel can also emit dX = dz W^T (whole, or the last 32-column block with K split over the
//    waves) from the dz tile it holds in LDS; its streamed-U^T build (H = 256) keeps the first 8 of a wave's
//    64 k-chunks in the LDS the tiles leave free (BwdUlds);
//  * lstm_fwd_cluster_kernel (bf16, H = 256, >= 64 tiles): 8 workgroups of one XCD keep the W and U slices of
//    32 hidden units each in LDS for the whole sweep and exchange h slices through L2 once per step;
//  * (two re-decompositions of the H = 256 BPTT sweep over PAIRS of workgroups were built in round 3, are correct and
//    slower than lstm_bwd_kernel: they live in tools/bwd_decompositions/, not in this library)
// The weight streams are bound by the CU's vector-memory path (64 B/clk), not by L2 or HBM: every fragment
// that can live in registers or LDS instead is time won, and every loop bound in these kernels has to be a
// compile-time constant (run-time variants of the same loops measured +0.4 ... +0.5 ms).
//
// Data layouts in HBM:
//  * Zx (x_t W + b of the unfused path), the gate stash and the cell stash C are
//    FRAGMENT-TILED: for 32-row block rb (= tile*steps + step) and 32-col block cb,
//    the 32x32 block is stored as [lane 0..63][16 accumulator registers], i.e.
//    element (s, c) of the block sits at ((rb*NCB + cb)*64 + 32*((s>>2)&1) + c)*16
//    + (s&3) + 4*(s>>3).  dj_gemm_nt writes this layout from its accumulators, and
//    both recurrent kernels read/write it with 16-byte per-lane accesses that are
//    perfectly coalesced -- no scalar loads, no partial-line writes.  The gate stash
//    holds z in fp32 mode and 8-bit activated gates in bf16 mode ("gate stash" below).
//  * h (Hout), dH and dz are ROW-MAJOR [rows, cols] (they feed the GEMMs and the
//    glue kernels) and go through LDS for wide coalesced rows.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "dj_kernels.h"

namespace {

template <typename T, int H> struct RecCfg {
  static constexpr int EPL = 16 / sizeof(T);
  static constexpr int KC = 2 * EPL;
  static constexpr int UW = 32;           // hidden units per wave (one 32-col MFMA tile per gate)
  static constexpr int NJ = 1;            // 32-col tiles per gate per wave
  static constexpr int NW = H / UW;       // waves per workgroup (H=128: 4, H=256: 8 = 2 per SIMD)
  static constexpr int NT = 64 * NW;      // threads per workgroup
  static constexpr int NKC = H / KC;      // k-chunks of the forward product (K = H)
  static constexpr int NKCB = 4 * H / KC; // k-chunks of the backward product (K = 4H)
  static constexpr int LDH = H + EPL;     // LDS row stride of the h tile
  static constexpr int LDZ = 4 * H + EPL; // LDS row stride of the dz tile
  static constexpr int NCB = 4 * H / 32;  // 32-col blocks of Z
  static constexpr int NCBH = H / 32;     // 32-col blocks of C
  static constexpr int VPT = 16 / EPL;    // 16-byte vectors per 16-register fragment (bf16: 2, f32: 4)
  // weight-fragment prefetch depth in k-chunks (H = 128 runs one wave per SIMD: registers to spare)
  static constexpr int PD = sizeof(T) == 2 ? 4 : 2;
  static constexpr int UNR = PD > 4 ? PD : 4;   // k-chunks per unrolled body (multiple of PD)
  // BPTT product: one MFMA per k-chunk and wave, so the ring must be deeper to cover L2 latency
  // (measured: PD 2 left the 64-iteration loop latency-bound at 9.3 us/step)
  static constexpr int PDB = sizeof(T) == 2 ? 8 : 4;
  static constexpr int UNRB = PDB > 8 ? PDB : 8;
  static constexpr bool HOIST = sizeof(T) == 2;   // prefetch Z fragments a step ahead (register budget)
  // H = 128 in bf16: a wave's whole U^T slice (4H x 32 units = 32 KB = 128 registers per lane) stays in
  // registers for the whole BPTT sweep -- one wave per SIMD has the register file for it -- so the
  // recurrence streams no weights at all
  static constexpr bool STATB = sizeof(T) == 2 && H == 128;
  // same for the forward kernels: U (and W when the input is at most 128 wide) as 128 registers each
  static constexpr bool STATF = sizeof(T) == 2 && H == 128;
};

// The BPTT kernel's wave geometry: RecCfg's.  (fp32 at H = 256 was also built with four waves of 64 hidden units --
// NJ = 2, one wave per SIMD, the whole VGPR + AGPR file: no spill instead of 84-164 bytes per lane, but 19.1 instead of
// 13.5 ms per training step in the sweep: the second wave per SIMD hides more latency than the spill costs.  The kernel
// stays NJ-generic -- the packed U^T stream is indexed by 32-unit block (w * NJ + j) -- and a specialisation of this
// struct with UW = 64, NJ = 2, NW = 4 selects that geometry.)
template <typename T, int H> struct BwdCfg : RecCfg<T, H> {};

// f(integral_constant<int, I>) for I = I0 .. N-1, every call inlined: a loop whose index is a compile-time constant in
// the body by construction (static register-array indices, `if constexpr` on the index)
template <int I, int N, typename Fn> __device__ __forceinline__ void dj_static_for(Fn&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    dj_static_for<I + 1, N>(f);
  }
}

// raw workgroup barrier: waits for this wave's LDS traffic only, so global prefetches
// and stores stay in flight across it (cdna_hip_programming.md "Pipelining across barriers")
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16 accumulator values <-> fragment-tiled memory
template <typename T> struct Frag16;
template <> struct Frag16<float> {
  uint4 v[4];
  __device__ __forceinline__ void load(const float* p) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ((const uint4*)p)[i];
  }
  __device__ __forceinline__ void store(float* p) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) ((uint4*)p)[i] = v[i];
  }
  __device__ __forceinline__ float get(int r) const {
    const uint4& q = v[r >> 2];
    uint32_t w = (r & 3) == 0 ? q.x : (r & 3) == 1 ? q.y : (r & 3) == 2 ? q.z : q.w;
    return __uint_as_float(w);
  }
  __device__ __forceinline__ void copy_from(const Frag16& o) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = o.v[i];
  }
};
template <> struct Frag16<bf16_t> {
  uint4 v[2];
  __device__ __forceinline__ void load(const bf16_t* p) {
    v[0] = ((const uint4*)p)[0];
    v[1] = ((const uint4*)p)[1];
  }
  __device__ __forceinline__ void store(bf16_t* p) const {
    ((uint4*)p)[0] = v[0];
    ((uint4*)p)[1] = v[1];
  }
  __device__ __forceinline__ float get(int r) const {
    const uint4& q = v[r >> 3];
    const int d = (r >> 1) & 3;
    uint32_t w = d == 0 ? q.x : d == 1 ? q.y : d == 2 ? q.z : q.w;
    return __uint_as_float((r & 1) ? (w & 0xFFFF0000u) : (w << 16));
  }
  __device__ __forceinline__ void copy_from(const Frag16& o) {
    v[0] = o.v[0];
    v[1] = o.v[1];
  }
};
// two values of one lane into two LDS cells: bf16 converts the pair with ONE v_cvt_pk_bf16_f32 (ds_write_b16 +
// ds_write_b16_d16_hi take the halves), fp32 stores them as they are
__device__ __forceinline__ void dj_lds_put2(float* p0, float* p1, float a, float b) {
  *p0 = a;
  *p1 = b;
}
__device__ __forceinline__ void dj_lds_put2(bf16_t* p0, bf16_t* p1, float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  const bf16x2_t q = __builtin_convertvector(v, bf16x2_t);
  *p0 = q[0];
  *p1 = q[1];
}

// ---------------------------------------------------------------- gate stash
// What the forward sweep leaves for BPTT, per 32x32 block (rows x hidden units) and gate, fragment-tiled like the
// MFMA accumulators ([64 lanes][16 values]; block (rb, cb) at element ((rb*NCB + cb)*64 + lane)*16):
//   fp32 mode : the pre-activations z (16 floats per lane) -- BPTT recomputes the activations, exact;
//   bf16 mode : the ACTIVATED gates as 8-bit codes (16 bytes per lane, half of a bf16 z stash: the recurrent
//               kernels are HBM-bound on exactly these bytes, and BPTT no longer recomputes 4 transcendentals per
//               element).  i, f, o in [0,1]: code = clamp(ceil(254 y), 0, 255), decoded as the interval midpoint
//               (code - 1/2)/254 clamped to [0,1] (|error| <= 1/508); codes 0 and 255 are reserved for the
//               SATURATED hard_sigmoid, so its derivative mask (0.2 inside, 0 outside) is exact.  g = tanh in
//               (-1,1): code = round(127 g) + 128 (|error| <= 1/254).  The forward values themselves are not
//               quantised -- only what BPTT reads back.
//               (The codes are produced by v_cvt_pk_u8_f32 itself -- saturating, round-to-nearest -- from 254 y + 0.49997:
//               dj_common.h dj_gate_code01; equal to the clamped ceil except in a band of 3e-5 above every integer.)
template <typename T> struct StashT { using type = float; };
template <> struct StashT<bf16_t> { using type = uint8_t; };
template <typename T> using StashElem = typename StashT<T>::type;

template <typename T, bool SIGM> struct GateEnc;
template <bool SIGM> struct GateEnc<float, SIGM> {
  static constexpr int STORES = 4;             // 16-byte stores per gate block and lane
  float z[4][16];
  __device__ __forceinline__ void put(int r, float zi, float zf, float zg, float zo, float, float, float, float) {
    z[0][r] = zi; z[1][r] = zf; z[2][r] = zg; z[3][r] = zo;
  }
  __device__ __forceinline__ void store(int g, float* p) const { store_frag(p, z[g]); }
};
template <bool SIGM> struct GateEnc<bf16_t, SIGM> {
  static constexpr int STORES = 1;
  uint32_t w[4][4] = {};
  static __device__ __forceinline__ float code01(float z, float y) { return dj_gate_code01<SIGM>(z, y); }   // dj_common.h
  __device__ __forceinline__ void put(int r, float zi, float zf, float, float zo, float ig, float fg, float gg,
                                      float og) {
    const int d = r >> 2, b = r & 3;
    w[0][d] = __builtin_amdgcn_cvt_pk_u8_f32(code01(zi, ig), b, w[0][d]);
    w[1][d] = __builtin_amdgcn_cvt_pk_u8_f32(code01(zf, fg), b, w[1][d]);
    w[2][d] = __builtin_amdgcn_cvt_pk_u8_f32(dj_gate_code_g(gg), b, w[2][d]);
    w[3][d] = __builtin_amdgcn_cvt_pk_u8_f32(code01(zo, og), b, w[3][d]);
  }
  __device__ __forceinline__ void store(int g, uint8_t* p) const { *(uint4*)p = make_uint4(w[g][0], w[g][1], w[g][2], w[g][3]); }
};

// the BPTT side: gate values and the derivative factors of the recurrent activation for accumulator register r
template <typename T, bool SIGM> struct GateDec;
template <bool SIGM> struct GateDec<float, SIGM> {
  Frag16<float> z[4];
  __device__ __forceinline__ void load(int g, const float* p) { z[g].load(p); }
  __device__ __forceinline__ void get(int r, float& ig, float& fg, float& gg, float& og, float& di, float& df,
                                      float& dO) const {
    const float zi = z[0].get(r), zf = z[1].get(r), zg = z[2].get(r), zo = z[3].get(r);
    ig = dj_ract<SIGM>(zi); fg = dj_ract<SIGM>(zf); gg = dj_tanh(zg); og = dj_ract<SIGM>(zo);
    di = dj_ract_grad<SIGM>(zi, ig); df = dj_ract_grad<SIGM>(zf, fg); dO = dj_ract_grad<SIGM>(zo, og);
  }
};
template <bool SIGM> struct GateDec<bf16_t, SIGM> {
  uint4 q[4];
  __device__ __forceinline__ void load(int g, const uint8_t* p) { q[g] = *(const uint4*)p; }
  static __device__ __forceinline__ float code(const uint4& v, int r) {
    const int d = r >> 2;
    const uint32_t w = d == 0 ? v.x : d == 1 ? v.y : d == 2 ? v.z : v.w;
    return (float)((w >> (8 * (r & 3))) & 0xFFu);                  // v_cvt_f32_ubyteN
  }
  static __device__ __forceinline__ void dec01(float cd, float& y, float& dy) {
    const float v = fmaf(cd, 1.f / 254.f, -0.5f / 254.f);
    y = __builtin_amdgcn_fmed3f(v, 0.f, 1.f);
    if constexpr (SIGM) dy = y * (1.f - y);
    else dy = (fabsf(cd - 127.5f) < 127.25f) ? 0.2f : 0.f;        // codes 1..254: inside the linear part (one compare)
  }
  __device__ __forceinline__ void get(int r, float& ig, float& fg, float& gg, float& og, float& di, float& df,
                                      float& dO) const {
    dec01(code(q[0], r), ig, di);
    dec01(code(q[1], r), fg, df);
    gg = fmaf(code(q[2], r), 1.f / 127.f, -128.f / 127.f);
    dec01(code(q[3], r), og, dO);
  }
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

// WTpack[(q*NKCB + kc)*64 + lane][e] = W[d = q*32 + l31][k = kc*KC + EPL*h + e]  (W is the Keras kernel
// [D, 4H]; rows d >= D are zero): the fragment stream of dX_t = dz_t W^T inside the BPTT kernel.
template <typename T, int H>
__global__ void pack_wt_bwd_kernel(const float* __restrict__ W, int D, int NQ, T* __restrict__ out) {
  using R = RecCfg<T, H>;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NQ * 32 * 4 * H) return;
  int e = idx % R::EPL, lane = (idx / R::EPL) % 64, rest = idx / (R::EPL * 64);
  int kc = rest % R::NKCB, q = rest / R::NKCB;
  int k = kc * R::KC + R::EPL * (lane >> 5) + e;
  int d = q * 32 + (lane & 31);
  out[idx] = dj_from_f32<T>(d < D ? W[(int64_t)d * 4 * H + k] : 0.f);
}

// ---------------------------------------------------------------- forward
// Zx: x_t W + b of every step, fragment-tiled in T (dj_gemm_nt c_mode 2); Gst: gate stash out (null = inference;
// in fp32 it may alias Zx: a lane rewrites exactly the fragments it has read).
template <typename T, int H, bool SIGM>
__global__ __launch_bounds__(2 * H) void lstm_fwd_kernel(const T* Zx, StashElem<T>* Gst,
                                                       const T* __restrict__ Upack, T* __restrict__ Hout,
                                                       T* __restrict__ Cout, int steps) {
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
  // fragment addresses of this lane: Z block (g, j) and C block j of row-block rb
  auto zoff = [&](int64_t rb, int g, int j) {
    return ((rb * R::NCB + (g * H + w * R::UW + j * 32) / 32) * 64 + lane) * 16;
  };
  auto zaddr = [&](int64_t rb, int g, int j) { return Zx + zoff(rb, g, j); };
  auto caddr = [&](int64_t rb, int j) { return Cout + ((rb * R::NCBH + (w * R::UW + j * 32) / 32) * 64 + lane) * 16; };

  Frag uf[R::STATF ? R::NKC : 1][4];
  if constexpr (R::STATF) {
    static_assert(R::NJ == 1, "stationary U assumes one column tile per gate and wave");
#pragma unroll
    for (int kc = 0; kc < R::NKC; ++kc)
#pragma unroll
      for (int q = 0; q < 4; ++q) uf[kc][q] = up[(q * R::NKC + kc) * 64];
  }
  Frag16<T> zin[4][R::NJ];
  if constexpr (R::HOIST) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < R::NJ; ++j) zin[g][j].load(zaddr(tile * steps, g, j));
  }

  int cur = 0;
  for (int t = 0; t < steps; ++t) {
    const int64_t rb = tile * steps + t;
    if constexpr (!R::HOIST) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < R::NJ; ++j) zin[g][j].load(zaddr(rb, g, j));
    }
    f32x16 acc[4][R::NJ];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < R::NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][j][r] = zin[g][j].get(r);
    // prefetch next step's x*W+b fragments (land during the MFMA phase)
    if constexpr (R::HOIST) {
      if (t + 1 < steps) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < R::NJ; ++j) zin[g][j].load(zaddr(rb + 1, g, j));
      }
    }
    if constexpr (R::STATF) {
      if (t > 0) {
        const T* hp = hs[cur] + l31 * R::LDH;
#pragma unroll
        for (int kc = 0; kc < R::NKC; ++kc) {
          Frag a = dj_lds_frag(hp + kc * R::KC, h);
#pragma unroll
          for (int q = 0; q < 4; ++q) dj_mfma(acc[q][0], a, uf[kc][q]);
        }
      }
    } else if (t > 0) {
      const T* hp = hs[cur] + l31 * R::LDH;
      Frag bq[R::PD][4 * R::NJ];
#pragma unroll
      for (int p = 0; p < R::PD; ++p)
#pragma unroll
        for (int q = 0; q < 4 * R::NJ; ++q) bq[p][q] = up[(q * R::NKC + p) * 64];
#pragma unroll 1
      for (int kc0 = 0; kc0 < R::NKC; kc0 += R::UNR) {
#pragma unroll
        for (int u = 0; u < R::UNR; ++u) {
          const int kc = kc0 + u;
          Frag a = dj_lds_frag(hp + kc * R::KC, h);
#pragma unroll
          for (int q = 0; q < 4 * R::NJ; ++q) dj_mfma(acc[q / R::NJ][q % R::NJ], a, bq[u % R::PD][q]);
          const int kn = (kc + R::PD < R::NKC) ? kc + R::PD : R::NKC - 1;   // clamped: tail reloads are harmless
#pragma unroll
          for (int q = 0; q < 4 * R::NJ; ++q) bq[u % R::PD][q] = up[(q * R::NKC + kn) * 64];
        }
      }
    }
    T* hn = hs[cur ^ 1];
#pragma unroll
    for (int j = 0; j < R::NJ; ++j) {
      const int u = w * R::UW + j * 32 + l31;
      float cv[16];
      GateEnc<T, SIGM> ge;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = dj_crow(r, lane);
        float zi = acc[0][j][r], zf = acc[1][j][r], zg = acc[2][j][r], zo = acc[3][j][r];
        float ig = dj_ract<SIGM>(zi), fg = dj_ract<SIGM>(zf), gg = dj_tanh(zg), og = dj_ract<SIGM>(zo);
        ge.put(r, zi, zf, zg, zo, ig, fg, gg, og);
        float cn = fg * c[j][r] + ig * gg;
        c[j][r] = cn;
        cv[r] = cn;
        hn[row * R::LDH + u] = dj_from_f32<T>(og * dj_tanh(cn));
      }
      if (Cout) store_frag(caddr(rb, j), cv);
      if (Gst) {
#pragma unroll
        for (int g = 0; g < 4; ++g) ge.store(g, Gst + zoff(rb, g, j));
      }
    }
    lds_barrier();
    // cooperative, coalesced copy of h_t (32 x H) to global
    constexpr int VPR = H / R::EPL;
#pragma unroll
    for (int v = tid; v < 32 * VPR; v += R::NT) {
      int row = v / VPR, cv = (v % VPR) * R::EPL;
      *(uint4*)(Hout + (rb * 32 + row) * H + cv) = *(const uint4*)(hn + row * R::LDH + cv);
    }
    cur ^= 1;
  }
}


// ---------------------------------------------------------------- forward with fused input projection
// z_t = x_t W + h_{t-1} U + b computed entirely inside the persistent kernel: the x*W GEMM launch and
// the 2 x 4H-wide Zx round trip through HBM disappear; the price is a second fragment stream (W)
// from L2 per step.  X tiles (32 x DP, row-major) are prefetched one step ahead into LDS.
// Wpack[((w*4+g)*NKX + kc)*64 + lane][e] = W[kc*KC + EPL*h + e][g*H + w*32 + l31]   (zero for k >= D)
template <typename T, int H>
__global__ void pack_w_fwd_kernel(const float* __restrict__ W, int D, int NKX, T* __restrict__ out) {
  using R = RecCfg<T, H>;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 4 * H * NKX * R::KC) return;
  int e = idx % R::EPL, lane = (idx / R::EPL) % 64, rest = idx / (R::EPL * 64);
  int kc = rest % NKX;
  rest /= NKX;
  int g = rest % 4, w = rest / 4;
  int k = kc * R::KC + R::EPL * (lane >> 5) + e;
  int col = g * H + w * R::UW + (lane & 31);
  out[idx] = dj_from_f32<T>(k < D ? W[(int64_t)k * 4 * H + col] : 0.f);
}

constexpr int FUSED_DPMAX = 288;   // widest layer input supported by the register prefetch (259 -> 264 here)

// WSTAT (stationary-weight builds only): W is held in registers as well (inputs up to H wide); otherwise W
// is streamed through an 8-deep fragment ring in front of the stationary U product.
template <typename T, int H, bool SIGM, bool WSTAT, bool WLDS>
__global__ __launch_bounds__(2 * H) void lstm_fwd_fused_kernel(const T* __restrict__ X, int DP, int NKX,
                                                               const T* __restrict__ Wpack,
                                                               const float* __restrict__ bias,
                                                               StashElem<T>* __restrict__ Zst,
                                                               const T* __restrict__ Upack, T* __restrict__ Hout,
                                                               T* __restrict__ Cout, int steps) {
  using R = RecCfg<T, H>;
  using Frag = typename DjFrag<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int LDX = NKX * R::KC + R::EPL;
  T* hs0 = (T*)smem_raw;                       // [2][32][LDH]
  T* xs0 = hs0 + 2 * 32 * R::LDH;              // [2][32][LDX]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int64_t tile = blockIdx.x;

  float c[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  float bv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bv[g] = bias[g * H + w * R::UW + l31];
  const Frag* up = (const Frag*)Upack + (int64_t)w * 4 * R::NKC * 64 + lane;
  const Frag* wp = (const Frag*)Wpack + (int64_t)w * 4 * NKX * 64 + lane;
  // WLDS (streamed-W path of the stationary build, ring of 6): the first 6 k-chunks of this wave's W slice live in
  // the LDS behind the x tiles -- the W stream is bound by the vector-memory path, LDS reads are not
  Frag* wl = (Frag*)(xs0 + 2 * 32 * LDX) + (w * 4 * 6) * 64 + lane;
  if constexpr (WLDS) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int kc = 0; kc < 6; ++kc) wl[(q * 6 + kc) * 64] = wp[(q * NKX + kc) * 64];
  }
  // stationary weights (bf16, H = 128): U always, W when its NKX k-chunks fit the same 8-chunk budget
  constexpr int NKS = R::STATF ? R::NKC : 1;      // WSTAT: the launcher guarantees NKX == NKC
  constexpr int NKW = (R::STATF && WSTAT) ? R::NKC : 1;
  Frag uf[NKS][4], wf[NKW][4];
  if constexpr (R::STATF) {
#pragma unroll
    for (int kc = 0; kc < NKS; ++kc)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uf[kc][q] = up[(q * R::NKC + kc) * 64];
        if constexpr (WSTAT) wf[kc][q] = wp[(q * NKX + kc) * 64];
      }
  }
  auto zaddr = [&](int64_t rb, int g) { return Zst + ((rb * R::NCB + (g * H + w * R::UW) / 32) * 64 + lane) * 16; };
  auto caddr = [&](int64_t rb) { return Cout + ((rb * R::NCBH + (w * R::UW) / 32) * 64 + lane) * 16; };

  // X tile staging: NVX 16-byte vectors per tile, up to NVMAX per thread
  constexpr int DPMAX = (R::STATF && WSTAT) ? H : FUSED_DPMAX;   // WSTAT builds take inputs up to H wide
  constexpr int NVMAX = (32 * DPMAX / R::EPL + R::NT - 1) / R::NT;
  const int vpr = DP / R::EPL, nvx = 32 * vpr;
  // The staged vectors travel by value (struct return / argument): as a loop-carried array written
  // through a by-reference lambda they were kept in scratch memory, and the scratch store right
  // behind the prefetch made every step wait for its HBM loads.
  struct XRegs { uint4 v[NVMAX]; };
  auto x_load = [&](int64_t rb) {
    XRegs r;
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) {
      const int v = tid + i * R::NT;
      r.v[i] = v < nvx ? *(const uint4*)(X + (rb * 32 + v / vpr) * DP + (v % vpr) * R::EPL) : make_uint4(0, 0, 0, 0);
    }
    return r;
  };
  auto x_store = [&](T* xs, const XRegs r) {
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) {
      const int v = tid + i * R::NT;
      if (v < nvx) *(uint4*)(xs + (v / vpr) * LDX + (v % vpr) * R::EPL) = r.v[i];
    }
  };
  // zero both X buffers once (k-tail columns DP..NKX*KC stay zero), then stage X[0]
  for (int i = tid; i < 2 * 32 * LDX * (int)sizeof(T) / 16; i += R::NT) ((uint4*)xs0)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();
  XRegs xr = x_load(tile * steps);
  x_store(xs0, xr);
  lds_barrier();

  int cur = 0;
  for (int t = 0; t < steps; ++t) {
    const int64_t rb = tile * steps + t;
    const T* xs = xs0 + (t & 1) * 32 * LDX;
    f32x16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = bv[g];
    if constexpr (R::STATF) {   // U (and W if WSTAT) in registers: little or no weight traffic in the recurrence
      const T* xp = xs + l31 * LDX;
      const T* hp = hs0 + cur * 32 * R::LDH + l31 * R::LDH;
      if constexpr (WSTAT) {
#pragma unroll
        for (int kc = 0; kc < NKS; ++kc) {
          Frag a = dj_lds_frag(xp + kc * R::KC, h);
#pragma unroll
          for (int q = 0; q < 4; ++q) dj_mfma(acc[q], a, wf[kc][q]);
        }
      } else {                  // wide input: W streamed, 8 (or 6) fragments per gate in flight; NKX is a multiple of
                                // the ring depth (259 inputs: 18 chunks = 3 x 6 instead of 24 = 3 x 8)
#define DJ_W_STREAM(PW)                                                                     \
  {                                                                                         \
    constexpr int k0 = (PW == 6 && WLDS) ? 6 : 0;   /* chunks 0..k0-1 come from LDS */      \
    Frag bw[PW][4];                                                                         \
    _Pragma("unroll") for (int p = 0; p < PW; ++p)                                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) bw[p][q] = wp[(q * NKX + k0 + p) * 64]; \
    if constexpr (k0 > 0) {                                                                 \
      _Pragma("unroll") for (int kc = 0; kc < 6; ++kc) {                                    \
        Frag a = dj_lds_frag(xp + kc * R::KC, h);                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) dj_mfma(acc[q], a, wl[(q * 6 + kc) * 64]); \
      }                                                                                     \
    }                                                                                       \
    _Pragma("unroll 1") for (int kc0 = k0; kc0 < NKX; kc0 += PW) {                          \
      _Pragma("unroll") for (int uu = 0; uu < PW; ++uu) {                                   \
        const int kc = kc0 + uu;                                                            \
        Frag a = dj_lds_frag(xp + kc * R::KC, h);                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) dj_mfma(acc[q], a, bw[uu][q]);        \
        const int kn = (kc + PW < NKX) ? kc + PW : NKX - 1;                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) bw[uu][q] = wp[(q * NKX + kn) * 64];  \
      }                                                                                     \
    }                                                                                       \
  }
        if constexpr (WLDS) DJ_W_STREAM(6) else if (NKX % 8 == 0) DJ_W_STREAM(8) else DJ_W_STREAM(6)
#undef DJ_W_STREAM
      }
      if (t > 0) {
#pragma unroll
        for (int kc = 0; kc < NKS; ++kc) {
          Frag a = dj_lds_frag(hp + kc * R::KC, h);
#pragma unroll
          for (int q = 0; q < 4; ++q) dj_mfma(acc[q], a, uf[kc][q]);
        }
      }
    } else {   // one continuous fragment stream: NKX chunks of x_t * W, then (t > 0) NKC chunks of h_{t-1} * U
      const T* xp = xs + l31 * LDX;
      const T* hp = hs0 + cur * 32 * R::LDH + l31 * R::LDH;
      const int ntot = NKX + (t > 0 ? R::NKC : 0);          // NKX and NKC are multiples of PD
      auto bfrag = [&](int kc, int q) {
        return kc < NKX ? wp[(q * NKX + kc) * 64] : up[(q * R::NKC + (kc - NKX)) * 64];
      };
      Frag bq[R::PD][4];
#pragma unroll
      for (int p = 0; p < R::PD; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[p][q] = bfrag(p, q);
#pragma unroll 1
      for (int kc0 = 0; kc0 < ntot; kc0 += R::PD) {
#pragma unroll
        for (int uu = 0; uu < R::PD; ++uu) {
          const int kc = kc0 + uu;
          Frag a = kc < NKX ? dj_lds_frag(xp + kc * R::KC, h) : dj_lds_frag(hp + (kc - NKX) * R::KC, h);
#pragma unroll
          for (int q = 0; q < 4; ++q) dj_mfma(acc[q], a, bq[uu][q]);
          const int kn = (kc + R::PD < ntot) ? kc + R::PD : ntot - 1;
#pragma unroll
          for (int q = 0; q < 4; ++q) bq[uu][q] = bfrag(kn, q);
        }
      }
    }
    // next X tile: issued after the weight streams of this step (in-order vmcnt), lands under the gate math
    xr = x_load(t + 1 < steps ? rb + 1 : rb);
    T* hn = hs0 + (cur ^ 1) * 32 * R::LDH;
    const int u = w * R::UW + l31;
    float cv[16];
    GateEnc<T, SIGM> ge;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float zi = acc[0][r], zf = acc[1][r], zg = acc[2][r], zo = acc[3][r];
      const float ig = dj_ract<SIGM>(zi), fg = dj_ract<SIGM>(zf), gg = dj_tanh(zg), og = dj_ract<SIGM>(zo);
      ge.put(r, zi, zf, zg, zo, ig, fg, gg, og);
      const float cn = fg * c[r] + ig * gg;
      c[r] = cn;
      cv[r] = cn;
      hn[dj_crow(r, lane) * R::LDH + u] = dj_from_f32<T>(og * dj_tanh(cn));
    }
    if (Cout) store_frag(caddr(rb), cv);
    if (Zst) {
#pragma unroll
      for (int g = 0; g < 4; ++g) ge.store(g, zaddr(rb, g));
    }
    if (t + 1 < steps) x_store(xs0 + ((t + 1) & 1) * 32 * LDX, xr);
    lds_barrier();
    constexpr int VPR = H / R::EPL;
#pragma unroll
    for (int v = tid; v < 32 * VPR; v += R::NT) {
      int row = v / VPR, cvv = (v % VPR) * R::EPL;
      *(uint4*)(Hout + (rb * 32 + row) * H + cvv) = *(const uint4*)(hn + row * R::LDH + cvv);
    }
    cur ^= 1;
  }
}

// ---------------------------------------------------------------- forward, weight-stationary cluster (bf16, H = 256)
// The per-tile kernel above streams every W and U fragment from L2 every step in every workgroup: 33.5 GB of
// L2 reads per launch of time layer 1, ~72 % of the chip's aggregate L2 bandwidth, which is what bounds it.
// Here 8 workgroups that share an XCD (blocks b, b+8, ..., b+56) form a CLUSTER that owns 8 sequence tiles:
// member s owns hidden units [32s, 32s+32) of all four gates, whose W and U slices (the fragment slices of
// "wave s" of the per-tile kernel: 4*NKX + 64 KiB) stay in LDS for the whole sweep; wave w multiplies the 32
// rows of tile w, [x_t | h_{t-1}], with them, so the cell update stays lane-local.  Per step a workgroup now
// reads 8 x (16 KiB x + 16 KiB h) instead of 1 MiB of weights.  Every member writes its 32-unit slice of
// h_t into a two-slot ring in L2.  In the training sweep (lstm_fwd_cluster_kernel) the slices ANNOUNCE THEMSELVES by a
// tag in a spare exponent bit ("TAGGED exchange" below: round 4; 2.56 -> 2.38 ms per step of the baseline shape);
// so does the cooperative body of the inference pair; in the pair's one-wave form and under DJ_KF_COUNTED_EXCHANGE a
// per-cluster counter in L2 closes the step (acknowledged
// stores, workgroup barrier, atomic, poll).  The x_t W half does not depend on the exchange and runs before the wait.
//  * x rows are fetched as full 128-byte lines (8 lanes per row) and turned into A fragments through a 4 KiB
//    LDS tile per wave, 64 columns per round (fragments read straight from the rows are 32 segments of 32 bytes
//    per instruction).
//  * h travels in the MFMA A-fragment image (the hx region of the scratch), one contiguous 1 KiB load per fragment.
//  * x_{t+1} is requested during step t: the first half right behind the h fragments (loads return in order: in
//    front of them it would hold the h fragments back), the second half once the h product has freed their
//    registers.  Only without register spills: the step must be unconditional (round 0 of the exchange carries
//    h = 0), a conditional h product costs a second set of accumulators.  (DESIGN.md section 8 has the variants.)
// Exchange state lives in a CALLER-OWNED scratch (dj_lstm_cluster_scratch_bytes; one per workspace, i.e. per
// engine / stream -- never shared between concurrent sweeps):
//     [0, 16 KiB)     per cluster two 128-byte lines: line 0 = the step counter, line 1 = the members' XCC ids
//                     (zeroed by cl_reset_kernel in front of every launch)
//     [16 KiB, +128)  fault line: [0] expired waits, [1] workgroups of clusters whose members sat on different XCDs
//                     (both sticky until the census reads them), stall census, description of the first expired
//                     wait ("bounded exchange waits" below)
//     [.., +8 MiB)    hx: [tile][step parity][k-chunk][lane] x 16 bytes
// Coherence: the members of a cluster must run on ONE XCD, whose L2 is then the coherence point for their h slices
// (counted protocol: stores acknowledged before the counter moves; both: exchange loads bypass L1 with sc1).  Round-robin dispatch puts
// blocks b and b + 8 on one XCD; HIP does not promise it, so it is VERIFIED per launch: every member publishes its
// hardware XCC id in round 0 and every wave compares the eight of its cluster.  A mismatch, like an expired wait
// (grid not co-resident), poisons the tile's cell state with NaN and is counted -- never a silent wrong answer,
// never a hung device; the host side re-runs the step with the per-tile kernel (model.py) or fails (bench.py).
// one 128-byte line per counter: counters of clusters on different XCDs must not share a line
constexpr int CL_CNT_STRIDE = 32;
constexpr int CL_M = 8;                       // members (= tiles) per cluster
constexpr int CL_CNT_INTS = 64 * 2 * CL_CNT_STRIDE;
constexpr size_t CL_OFF_FAULT = (size_t)CL_CNT_INTS * sizeof(int);
constexpr size_t CL_OFF_HX = CL_OFF_FAULT + 128;
constexpr size_t CL_BYTES = CL_OFF_HX + (size_t)256 * 2 * 16 * 64 * 16;
// h fragments of the exchange: loads past the non-coherent L1 (sc1).  A BUFFER-load builtin, not inline assembly: until
// round 4 these were assembly statements, whose destination registers the compiler considers written as soon as the
// statement is issued and whose s_waitcnt -- another assembly statement -- orders memory operations only; nothing but the
// scheduler's habits kept the uses behind the wait (an fp32 experiment of round 4 lost that bet: tag checks in front of
// the wait, late data over reassigned registers, a memory fault).  The builtin is tracked by the compiler's own vmcnt
// bookkeeping.  `rs` = the exchange region as a buffer resource (wave-uniform), `off` = byte offset of the lane's piece.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cl_hx_rsrc(const void* hxb) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)hxb, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 ld_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);      // aux 16 = sc1
  return make_uint4(v.x, v.y, v.z, v.w);
}

template <int NR> struct ClXRegs { uint4 v[NR][4]; };
// A 16-byte piece of the row-major h_t.  In the training sweep nothing in the launch reads these rows again (the exchange
// has its own copy), so they go out NON-TEMPORAL like the stashes (NT: a compile-time switch -- with a run-time one the
// four dword stores were neither merged nor all marked); in the inference pair the rows are the input of the upper
// layer, which reads them from this L2.  (tools/l2_writeback.hip: an overwritten L2-resident line costs no fabric write
// at all -- the 0.5 GB per launch the exchange ring adds to WRITE_SIZE are capacity evictions by what streams through.)
template <bool NT>
__device__ __forceinline__ void cl_store_h(bf16_t* p, const uint4& v) {
  if constexpr (!NT) {
    *(uint4*)p = v;
  } else {
    __builtin_nontemporal_store(v.x, (unsigned*)p);
    __builtin_nontemporal_store(v.y, (unsigned*)p + 1);
    __builtin_nontemporal_store(v.z, (unsigned*)p + 2);
    __builtin_nontemporal_store(v.w, (unsigned*)p + 3);
  }
}

// rows r8, r8+8, r8+16, r8+24 of a 32-row block, 16 bytes at column 64 r + 8 xc each (zero past DP), rounds R0..R1-1
template <int NR, int R0, int R1>
__device__ __forceinline__ void cl_load_x(ClXRegs<NR>& q, const bf16_t* xb, int DP, int xc) {
#pragma unroll
  for (int r = R0; r < R1; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // columns past DP: ANY finite values will do -- the packed W is zero for k >= D (pack_w_fwd_kernel) -- so the lane
      // re-reads the start of its rows instead of being masked.  (With `ok ? v : 0` the compiler put every pair of requests
      // under its own exec branch, and its vmcnt bookkeeping then had to assume the path on which all LATER requests of the
      // step are skipped: the wait for x_t at the top of a step came out as vmcnt(3) -- i.e. wait for the acknowledgement
      // of the 10 stores the previous step had just issued -- instead of vmcnt(25).)
      const int co = r * 64 + xc * 8 < DP ? r * 64 : 0;
      const uint4 v = *(const uint4*)(xb + 8 * i * DP + co);
      q.v[r][i] = make_uint4(v.x, v.y, v.z, v.w);
    }
}
// ---- bounded exchange waits.
// Fault line of a cluster scratch (32 ints at CL_OFF_FAULT; dj_lstm_cluster_fault_words):
//   [0] expired waits (waves whose bound ran out)        [1] workgroups of clusters spread over several XCDs
//   [2] test hook (DJ_KF_DEBUG_CLUSTER_FAULT)
//   [4] waits that saw two consecutive polls more than CL_GAP_STALL cycles apart   } stall census: cumulative, never
//   [5] the longest poll-to-poll gap seen, in units of 1024 shader cycles          } reset by the fault census
//   [6] polls at which the shader clock had gone BACKWARDS since the poll before (wave restored on another XCC)
//   [8] 1 = words 9..20 describe the FIRST expired wait since the host last took the census:
//       [9] who (kind << 24 | cluster << 12 | member << 8 | wave; kind 1 = bf16 sweep, 2 = cooperative body, 3 = fp32, 4 / 7 = bf16 sweep / cooperative body waiting for
//       TAGGED h fragments ([11] = fragments that had arrived, [12] = 16),
//       5 / 6 = pair / two-tile BPTT experiments (tools/); bit 28 = the wait was for the PRODUCING layer's counter),
//       [10] step, [11] counter value seen last, [12] target, [13] polls made, [14..15] shader cycles from the first poll
//       to the last (64 bit), [16] longest poll-to-poll gap in cycles (saturating), [17] hardware XCC id + 1
// THE BOUND COUNTS POLLS, NOT TIME.  Until round 4 a wait expired once 20 M (then 200 M) shader cycles had passed since
// its first poll.  That clock keeps running while the wave does not: when the whole queue is taken off the device for a
// while (wave save / restore around another process's time slice, an eviction of the process's queues by the driver)
// every waiter comes back to a poll that still shows the old count -- the late member was off the device too -- next to
// a clock that says the bound is long gone, and all of them "expire" together although nobody is late relative to anybody
// else (a 1024-step generation test lost 6 waits at once on one box that way).  A poll
// is only counted when the wave executes it, so the bound below is ~100 ms of ACTUAL polling (a poll -- a load from L2
// past the L1 plus 128 cycles of sleep -- measured 384 shader cycles: 2^19 of them are 2.0e8 cycles) whatever happens
// to the queue in between.
// STICKY: the first wave of a cluster whose bound does run out (grid not co-resident: a member never arrives) sets
// CL_POISON in the cluster's counter, which satisfies every later target at once; a wave that sees the bit -- or fails
// its placement check -- is `dead`: its tile carries NaN from there on and it does not wait again (it keeps storing and
// arriving, so no partner stalls on it).  A faulted launch therefore costs ONE bound, not one per remaining step.
constexpr int CL_POISON = 1 << 30;
constexpr unsigned CL_WAIT_POLLS = 1u << 19;
constexpr unsigned CL_WAIT_POLLS_TAGGED = 1u << 17;    // a tagged poll (up to 16 fragment loads, every wave of the cluster
                                                       // at it) measured 1,660 cycles: again ~0.1 s of polling
constexpr unsigned CL_GAP_NOTE = 1u << 17, CL_GAP_STALL = 1u << 20;   // ~60 us / ~0.5 ms at 2.1 GHz
enum { CLF_EXPIRED = 0, CLF_MISPLACED = 1, CLF_HOOK = 2, CLF_STALLS = 4, CLF_MAXGAP = 5, CLF_BACK = 6, CLF_DIAG = 8, CLF_WORDS = 32 };
constexpr int CLW_GATE = 1 << 28;
__device__ __forceinline__ int cl_who(int kind, int cid, int member, int wave) {
  return kind << 24 | (cid & 0xfff) << 12 | (member & 15) << 8 | (wave & 255);
}
// stall census of a finished wait (rare: a healthy poll returns within microseconds).  `back` = polls of this wait at
// which the shader clock read LOWER than at the poll before: s_memtime is per XCC, and a wave that was saved and restored
// on another XCC in the middle of a wait continues on that XCC's counter.  Such a difference says "the wave was moved",
// not how long it was away, so it is counted in a word of its own (CLF_BACK) and never enters the longest gap (until
// round 5 the unsigned difference turned it into a "gap" of ~2^32 cycles; a true gap beyond 2^31 cycles -- a second --
// is counted here as well).
__device__ __forceinline__ void cl_gap_step(unsigned now, unsigned& last, unsigned& maxgap, unsigned& back) {
  const int d = (int)(now - last);
  last = now;
  if (d < 0) ++back;
  else maxgap = (unsigned)d > maxgap ? (unsigned)d : maxgap;
}
__device__ __forceinline__ void cl_note_gap(int* fault, unsigned maxgap, unsigned back, int lane) {
  if ((maxgap > CL_GAP_NOTE || back) && lane == 0) {
    if (maxgap > CL_GAP_NOTE) atomicMax(fault + CLF_MAXGAP, (int)(maxgap >> 10));
    if (maxgap > CL_GAP_STALL) atomicAdd(fault + CLF_STALLS, 1);
    if (back) atomicAdd(fault + CLF_BACK, (int)back);
  }
}
// an expired wait: counted, and the first one since the last census described (words CLF_DIAG ..)
__device__ __forceinline__ void cl_note_expired(int* fault, int who, int t, int v, int target, unsigned polls,
                                                unsigned long long t0, unsigned maxgap, int lane) {
  if (lane == 0) {
    atomicAdd(fault + CLF_EXPIRED, 1);
    if (atomicCAS(fault + CLF_DIAG, 0, 1) == 0) {
      const unsigned long long el = __builtin_readcyclecounter() - t0;
      fault[CLF_DIAG + 1] = who;
      fault[CLF_DIAG + 2] = t;
      fault[CLF_DIAG + 3] = v;
      fault[CLF_DIAG + 4] = target;
      fault[CLF_DIAG + 5] = (int)polls;
      fault[CLF_DIAG + 6] = (int)(unsigned)el;
      fault[CLF_DIAG + 7] = (int)(unsigned)(el >> 32);
      fault[CLF_DIAG + 8] = (int)maxgap;
      fault[CLF_DIAG + 9] = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) + 1;
    }
  }
}
// wave-uniform bounded wait for *cnt >= target (every lane polls the same word: one request); returns the value seen
// (CL_POISON set: some wave of the cluster has given up), or -1 = this wave's bound ran out (counted, described)
__device__ __forceinline__ int cl_wait(int* cnt, int target, int* fault, int who, int t, int lane) {
  int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  if (v >= target) return v;
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned last = (unsigned)t0, maxgap = 0, polls = 0, back = 0;
#pragma nounroll
  do {
    __builtin_amdgcn_s_sleep(2);
    v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    cl_gap_step((unsigned)__builtin_readcyclecounter(), last, maxgap, back);
  } while (v < target && ++polls < CL_WAIT_POLLS);     // the last poll is always looked at before giving up
  cl_note_gap(fault, maxgap, back, lane);
  if (v >= target) return v;
  cl_note_expired(fault, who, t, v, target, polls, t0, maxgap, lane);
  return -1;
}
// One exchange wait of a wave under the sticky protocol above.  true = the wave's tile has to be poisoned NOW (its own
// bound ran out, or another wave of the cluster had given up); a dead wave returns at once.  `seen` (optional) receives
// the counter value (with CL_POISON once dead: every later target counts as met).
__device__ __forceinline__ bool cl_wait_step(int* cnt, int target, int* fault, int who, int t, int lane, bool& dead,
                                             int* seen = nullptr) {
  if (dead) return false;
  const int v = cl_wait(cnt, target, fault, who, t, lane);
  if (seen) *seen = v < 0 ? CL_POISON : v;
  if (v >= 0 && !(v & CL_POISON)) return false;
  if (v < 0 && lane == 0) __hip_atomic_fetch_or(cnt, CL_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  dead = true;
  return true;
}
// ---- TAGGED exchange (the training sweep, lstm_fwd_cluster_kernel): the h slices announce themselves.
// |h| < 1, so bit 14 of a bf16 h (the top exponent bit) is always 0; the sender puts a TAG there -- in elements 0 and 4 of
// every 16-byte piece, one per 8-byte half -- that flips every time a ring slot is rewritten: h_t carries ((t + 2) >> 1) & 1
// and lives in slot t & 1, whose previous content h_{t-2} carried the opposite.  A reader loads the 16 fragments of
// h_{t-1}, and the ones whose tags match ARE h_{t-1}: no counter, no acknowledged store, no workgroup barrier and no
// atomic between a member's last gate and its partners' products (the counted protocol spent four L2 round trips per
// step there: store acknowledgement, atomic, poll, loads).  Stale pieces are polled again (only those), under the same
// poll bound, census and description as a counted wait; a wave whose bound runs out poisons its tile and sets CL_POISON
// in the cluster's counter (for the record: nobody waits on the counter after round 0).  Why two slots are enough
// without a "done reading" signal: a member writes h_{t+1} into the slot of h_{t-1} only after it has read h_t of ALL
// members, and each of those was written after its author had finished reading h_{t-1}.  Round 0 (h_{-1} = 0 into slot
// 1, zeros with the wrong tag into slot 0, both acknowledged before the round-0 arrival) and the placement check stay
// on the counter, so a slot never shows a previous launch's bytes.  A poisoned tile's NaNs survive in the 6 untagged
// elements of every piece.
constexpr unsigned CL_TAG = 0x4000u;
__device__ __forceinline__ unsigned cl_tag_of(int t) { return (unsigned)(((t + 2) >> 1) & 1) << 14; }   // tag of h_t
__device__ __forceinline__ unsigned cl_piece_stale(const uint4& v, unsigned e) { return ((v.x ^ e) | (v.z ^ e)) & CL_TAG; }
// slow path of a tagged read: some piece of `ah` was stale.  true = the bound ran out (counted, described; `dead` set)
template <int NKC>
__device__ __forceinline__ bool cl_wait_tagged(__amdgpu_buffer_rsrc_t hxr, unsigned hoff, uint4 (&ah)[NKC], unsigned e, int* cnt,
                                               int* fault, int who, int t, int lane, bool& dead) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned last = (unsigned)t0, maxgap = 0, polls = 0, back = 0;
  int nstale;
#pragma nounroll
  do {
    __builtin_amdgcn_s_sleep(2);
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
      if (__any(cl_piece_stale(ah[kc], e) != 0)) ah[kc] = ld_sc1(hxr, hoff + kc * 1024);     // wave-uniform
    nstale = 0;
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) nstale += __any(cl_piece_stale(ah[kc], e) != 0) ? 1 : 0;
    cl_gap_step((unsigned)__builtin_readcyclecounter(), last, maxgap, back);
  } while (nstale && ++polls < CL_WAIT_POLLS_TAGGED);
  cl_note_gap(fault, maxgap, back, lane);
  if (!nstale) return false;
  cl_note_expired(fault, who, t, NKC - nstale, NKC, polls, t0, maxgap, lane);     // "counter" = fragments that did arrive
  if (lane == 0) __hip_atomic_fetch_or(cnt, CL_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  dead = true;
  return true;
}
// Wavefront pairing (inference, two stacked layers in ONE launch; lstm_fwd_cluster_pair_kernel): the blocks of the second
// half of the grid run the upper layer one to three steps behind the lower one.  The lower layer (`sp_out` set) writes,
// instead of its raw h, the upper layer's INPUT x = bf16(h + style term) -- the inter-layer glue of inference, where
// dropout is the identity -- and moves its counter once more after its last stores; the upper layer (`gate` set) asks
// for the rows of step t only once the lower cluster of the SAME tile has closed step t + 1 (stores are acknowledged in
// order, so the rows of step t have then reached the L2 both clusters share: cluster c and c + 8 sit on one XCD, which
// the upper layer verifies against the XCC ids the lower one published).  The lower layer never waits for the upper one.
struct ClPair {
  const float* sp_out;      // lower layer: [B * steps, sp_D] style term of the upper layer's input; null = plain h out
  int sp_D, n_seq, n_b;     // its row length; sequences per batch element (N) and batch elements (rows (b, n) -> b)
  int* gate;                // upper layer: counter line of the producing cluster; null = X is ready at launch
  int role;                 // 0 / 1: half of the grid (cluster ids and hx slots of the halves are disjoint)
};
template <bool SIGM, int NKX, bool TAGGED = false>
__device__ __forceinline__ void lstm_fwd_cluster_body(const bf16_t* __restrict__ X, int DP,
                                                      const bf16_t* __restrict__ Wpack, const float* __restrict__ bias,
                                                      StashElem<bf16_t>* __restrict__ Zst,
                                                      const bf16_t* __restrict__ Upack, bf16_t* __restrict__ Hout,
                                                      bf16_t* __restrict__ Cout, int steps, int* __restrict__ cl,
                                                      int ntiles, const ClPair pr, int nblocks) {
  using T = bf16_t;
  constexpr int H = 256;
  using R = RecCfg<T, H>;
  using Frag = typename DjFrag<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Frag* Bw = (Frag*)smem_raw;                              // [4][NKX][64]  W slice of this member
  Frag* Bu = Bw + 4 * NKX * 64;                            // [4][NKC][64]  U slice
  T* hto = (T*)(Bu + 4 * R::NKC * 64);                     // [8 waves] 4 KiB: x transposition rounds / h_t tile
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int bidx = (int)blockIdx.x - pr.role * nblocks;    // block index within this layer's half of the grid
  const int xcd = bidx & 7, j = bidx >> 3, s = j & (CL_M - 1);
  const int lcid = xcd + 8 * (j >> 3);                     // cluster id within the layer
  const int cid = lcid + pr.role * (nblocks >> 6) * 8;     // cluster id within the launch (counter lines)
  // wave w of every member works on tile lcid + nclusters * w: a launch with fewer tiles than wave slots (generation:
  // 5 tiles on 8 clusters) spreads them over the clusters, one wave each, instead of filling one cluster; waves
  // without a tile only keep the workgroup barriers company
  const int64_t tile = (int64_t)lcid + (int64_t)(nblocks >> 3) * w;
  const bool active = tile < ntiles;
  const int64_t hx_tile = tile + (int64_t)pr.role * 128;   // exchange slots of the two layers of a pair are disjoint
  int* cnt = cl + 2 * cid * CL_CNT_STRIDE;
  int* xccs = cnt + CL_CNT_STRIDE;
  int* fault = cl + CL_CNT_INTS;
  uint4* hxb = (uint4*)((unsigned char*)cl + CL_OFF_HX);
  const __amdgpu_buffer_rsrc_t hxr = cl_hx_rsrc(hxb);
  constexpr int ARRIVALS = CL_M;

  // stationary operand slices -> LDS (contiguous in the packed streams)
  {
    const uint4* gw = (const uint4*)((const Frag*)Wpack + (int64_t)s * 4 * NKX * 64);
    const uint4* gu = (const uint4*)((const Frag*)Upack + (int64_t)s * 4 * R::NKC * 64);
    for (int i = tid; i < 4 * NKX * 64; i += 512) ((uint4*)Bw)[i] = gw[i];
    for (int i = tid; i < 4 * R::NKC * 64; i += 512) ((uint4*)Bu)[i] = gu[i];
  }
  float c[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  float bv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bv[g] = bias[g * H + s * 32 + l31];
  const int xr8 = lane >> 3, xc = lane & 7;               // x rows: 8 lanes per row (below)
  constexpr int NR = (NKX + 3) / 4, NRA = (NR + 1) / 2;   // rounds; those requested with the h fragments
  [[maybe_unused]] ClXRegs<NR> xq0;
  // Tagged sweep: x_0 is requested HERE, in front of round 0, and retired there.  Requested right in front of the step
  // loop it is still pending at the loop header with no later request behind it on that path, and the compiler's wait
  // for x_t at the top of a step -- the merge of that path with the back edge, where the step's 10 stores follow the x
  // requests -- became vmcnt(0): every step then began by waiting for the acknowledgement of the stores the previous
  // one had just issued, one L2 round trip on the chain that the tagged exchange no longer needs.
  if constexpr (TAGGED) {
    if (active) cl_load_x<NR, 0, NR>(xq0, X + (tile * steps * 32 + xr8) * DP + xc * 8, DP, xc);
  }
  // round 0 of the exchange: h_{-1} = 0 goes into the parity-1 slots like any h_t, so that step 0 is a step like the
  // others (a conditional h product costs a second set of accumulators: 64 registers); with it every member
  // publishes the XCD it runs on
  const int my_xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) + 1;   // HW_REG_XCC_ID[3:0]
  {
    if (active) {
      uint4* hxo = hxb + ((hx_tile * 2 + 1) * 16 + 2 * s) * 64 + lane;
      hxo[0] = make_uint4(0, 0, 0, 0);
      hxo[64] = make_uint4(0, 0, 0, 0);
      if constexpr (TAGGED) {      // slot 0 must not show a previous launch's h with the tag h_0 will carry (1): zeros, tag 0
        hxo[-16 * 64] = make_uint4(0, 0, 0, 0);
        hxo[-16 * 64 + 64] = make_uint4(0, 0, 0, 0);
      }
    }
    if (tid == 0) __hip_atomic_store(xccs + s, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (TAGGED) {      // x_0 is USED here (an empty statement the compiler cannot look into): its requests stay in
                               // front of this point and its registers are not pending at the loop header
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("" : "+v"(xq0.v[r][i].x), "+v"(xq0.v[r][i].y), "+v"(xq0.v[r][i].z), "+v"(xq0.v[r][i].w));
  }
  __syncthreads();
  const int hook = __hip_atomic_load(fault + CLF_HOOK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests only
  if (tid == 0 && !((hook & 2) && s == CL_M - 1)) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // placement check: once all members have arrived, their XCC ids must be this workgroup's own
  const int who = cl_who(1, cid, s, w);
  bool dead = false;                                       // wave-uniform: the tile is poisoned, no more waits (cl_wait_step)
  {
    bool bad = cl_wait_step(cnt, ARRIVALS, fault, who, -1, lane, dead);
    if (!bad) {
      int other = my_xcc;
      if (lane < CL_M) other = __hip_atomic_load(xccs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // hook bit 0 (DEEPJ_DEBUG_CLUSTER_FAULT): the launch behaves as if its clusters were spread over XCDs
      if (!__all(other == my_xcc) || (hook & 1)) {
        if (lane == 0 && w == 0) atomicAdd(fault + CLF_MISPLACED, 1);
        dead = bad = true;
      }
    }
    if (bad) {
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = __builtin_nanf("");
    }
  }

  // upper layer of a pair: the producing cluster (same tile, lower half of the grid) must sit on this XCD too
  int* gate = pr.gate ? cl + 2 * lcid * CL_CNT_STRIDE : nullptr;
  int seen = 0;                                            // last value of the producer's counter this wave saw
  if (gate && active && !dead) {
    // rows of step 0 are out once the producer has closed step 1
    bool bad = cl_wait_step(gate, ARRIVALS * 3, fault, who | CLW_GATE, -1, lane, dead, &seen);
    if (!bad) {
      int other = my_xcc;
      if (lane < CL_M) other = __hip_atomic_load(gate + CL_CNT_STRIDE + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!__all(other == my_xcc)) {
        if (lane == 0) atomicAdd(fault + CLF_MISPLACED, 1);
        dead = bad = true;
      }
    }
    if (bad) {
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = __builtin_nanf("");
    }
    asm volatile("" ::: "memory");
  }
  // lower layer of a pair: per-lane source of the style term of its two 16-byte pieces of the row-major store
  const float* spp[2] = {nullptr, nullptr};
  if (pr.sp_out && active) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = lane + 64 * i, row = v >> 2, cq = (v & 3) * 8;
      int64_t bb = (tile * 32 + row) / pr.n_seq;
      if (bb > pr.n_b - 1) bb = pr.n_b - 1;               // padding rows of the last tile: any valid row
      spp[i] = pr.sp_out + bb * steps * pr.sp_D + s * 32 + cq;
    }
  }
  unsigned char* xs = (unsigned char*)hto + w * 4096;      // this wave's 4 KiB tile: x rounds, then the h tile
  T* ht = (T*)xs;
  // x rows are fetched as full 128-byte lines, 8 lanes per row (fragments read straight from the rows would be
  // 32 segments of 32 bytes per instruction), ONE STEP AHEAD: the reads of a step otherwise queue behind the
  // write burst of the previous one in HBM (measured: 8 k cycles until x_t arrives, a quarter of the step).
  ClXRegs<NR> xq;                  // (declared HERE: alive across round 0's statements it became a stack object)
  if constexpr (TAGGED) {
    xq = xq0;
  } else {
    if (active) cl_load_x<NR, 0, NR>(xq, X + (tile * steps * 32 + xr8) * DP + xc * 8, DP, xc);
  }
  for (int t = 0; t < steps; ++t) {
    if (!active) {                 // wave without a tile: the step barrier only (the branch is wave-uniform)
      if constexpr (!TAGGED) __syncthreads();
      continue;
    }
    const int64_t rb = tile * steps + t;
    float4 spv[2][2];                                  // style term of this step's rows: in flight for the whole step
    if (pr.sp_out) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        spv[i][0] = *(const float4*)(spp[i] + (int64_t)t * pr.sp_D);
        spv[i][1] = *(const float4*)(spp[i] + (int64_t)t * pr.sp_D + 4);
      }
    }
    f32x16 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;   // bias joins in the cell update (a hoisted splat would spill)
    // ---- x_t W: independent of the exchange, so it runs before the wait; x_t (requested during the previous step) is
    // turned into A fragments through this wave's 4 KiB LDS tile, 64 columns per round.
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(uint4*)(xs + (xr8 + 8 * i) * 128 + ((xc ^ xr8) << 4)) = xq.v[r][i];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kc = 4 * r + q;
        if (kc < NKX) {
          const Frag a = *(const Frag*)(xs + l31 * 128 + (((2 * q + h) ^ (l31 & 7)) << 4));
#pragma unroll
          for (int g = 0; g < 4; ++g) dj_mfma(acc[g], a, Bw[(g * NKX + kc) * 64 + lane]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    const T* xnext = X + ((tile * steps + (t + 1 < steps ? t + 1 : t)) * 32 + xr8) * DP + xc * 8;
    // ---- h_{t-1} U: needs the slices of all members
    {
      // a member that never arrives (grid not co-resident) is never a silent wrong answer: the wait is bounded and
      // counted (cl_wait), and the cell state is poisoned, so every later h of this tile, and the loss, is NaN
      bool bad = false;
      if constexpr (!TAGGED) {
        bad = cl_wait_step(cnt, ARRIVALS * (t + 1), fault, who, t, lane, dead);
        // upper layer of a pair: x_{t+1} (requested below) exists once the producer has closed step t + 2; the counter
        // is polled only when the last value seen does not cover it (the producer is faster and runs away)
        if (gate && t + 1 < steps && seen < ARRIVALS * (t + 4))
          bad |= cl_wait_step(gate, ARRIVALS * (t + 4), fault, who | CLW_GATE, t, lane, dead, &seen);
        if (bad) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = __builtin_nanf("");
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      const unsigned hoff = (unsigned)((((hx_tile * 2 + ((t + 1) & 1)) * 16) * 64 + lane) * 16);   // bytes into the hx region
      uint4 ah[R::NKC];
#pragma unroll
      for (int kc = 0; kc < R::NKC; ++kc) ah[kc] = ld_sc1(hxr, hoff + kc * 1024);
      asm volatile("" ::: "memory");
      // the first rounds of x_{t+1} go out right behind the h fragments, always 4*NRA requests, so the wait below
      // is a constant (tagged exchange: behind the tag check, whose slow path then has their registers to itself)
      if constexpr (!TAGGED) {
        cl_load_x<NR, 0, NRA>(xq, xnext, DP, xc);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NRA) : "memory");
      } else {                     // the fragments whose tags match are h_{t-1}; the others are asked for again
        const unsigned e = cl_tag_of(t - 1);
        unsigned stale = 0;
#pragma unroll
        for (int kc = 0; kc < R::NKC; ++kc) stale |= cl_piece_stale(ah[kc], e);
        if (!dead && __any(stale != 0))
          bad = cl_wait_tagged<R::NKC>(hxr, hoff, ah, e, cnt, fault, cl_who(4, cid, s, w), t, lane, dead);
        if (bad) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = __builtin_nanf("");
        }
#pragma unroll
        for (int kc = 0; kc < R::NKC; ++kc) {
          ah[kc].x &= ~CL_TAG;
          ah[kc].z &= ~CL_TAG;
        }
        cl_load_x<NR, 0, NRA>(xq, xnext, DP, xc);
      }
#pragma unroll
      for (int kc = 0; kc < R::NKC; ++kc) {
        Frag a;
        __builtin_memcpy(&a, &ah[kc], 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) dj_mfma(acc[g], a, Bu[(g * R::NKC + kc) * 64 + lane]);
      }
    }
    cl_load_x<NR, NRA, NR>(xq, xnext, DP, xc);   // the rest once the h fragments' registers are free
    // ---- cell update (lane-local), stash, h slice out
    float cv[16];
    GateEnc<T, SIGM> ge;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float zi = acc[0][r] + bv[0], zf = acc[1][r] + bv[1], zg = acc[2][r] + bv[2], zo = acc[3][r] + bv[3];
      const float ig = dj_ract<SIGM>(zi), fg = dj_ract<SIGM>(zf), gg = dj_tanh(zg), og = dj_ract<SIGM>(zo);
      ge.put(r, zi, zf, zg, zo, ig, fg, gg, og);
      const float cn = fg * c[r] + ig * gg;
      c[r] = cn;
      cv[r] = cn;
      ht[dj_crow(r, lane) * 32 + l31] = dj_from_f32<T>(og * dj_tanh(cn));
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the exchange copy goes out FIRST: only it has to be acknowledged before the counter moves; chunks 2s, 2s+1 of
    // this tile in fragment image, 1 KiB each
    {
      uint4* hxo = hxb + ((hx_tile * 2 + (t & 1)) * 16 + 2 * s) * 64 + lane;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        uint4 pv = *(const uint4*)(ht + l31 * 32 + 16 * jj + 8 * h);
        if constexpr (TAGGED) {
          pv.x = (pv.x & ~CL_TAG) | cl_tag_of(t);
          pv.z = (pv.z & ~CL_TAG) | cl_tag_of(t);
          if ((hook & 4) && s == CL_M - 1 && t >= 2) continue;      // test hook: this member's slices stop arriving
        }
        hxo[jj * 64] = pv;
      }
    }
    asm volatile("" ::: "memory");
    // then the row-major h slice (32 rows x 64 bytes -> 2 x 16-byte vectors per lane) and the stash
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = lane + 64 * i, row = v >> 2, cq = (v & 3) * 8;
      uint4 hv = *(const uint4*)(ht + row * 32 + cq);
      if (pr.sp_out) {                                 // the next layer's input: bf16(h + style), as glue_fwd writes it
        const float sp8[8] = {spv[i][0].x, spv[i][0].y, spv[i][0].z, spv[i][0].w,
                              spv[i][1].x, spv[i][1].y, spv[i][1].z, spv[i][1].w};
        T he[8], xo[8];
        __builtin_memcpy(he, &hv, 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) xo[e] = dj_from_f32<T>(dj_to_f32(he[e]) + sp8[e]);
        __builtin_memcpy(&hv, xo, 16);
      }
      cl_store_h<TAGGED>(Hout + (rb * 32 + row) * H + s * 32 + cq, hv);
    }
    // c and the gate stash are not read again before BPTT: non-temporal stores keep them from pushing the x rows and
    // the h exchange slots out of L2 (PMC: reads of the time-layer-1 launch 0.86 -> 0.58 GB, x alone is 0.54; the
    // 0.54 GB of exchange write-backs stay; time +-0)
    if (Cout) {
      bf16_t* cp_ = Cout + ((rb * R::NCBH + s) * 64 + lane) * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4 q = make_uint4(pack_bf16x2(cv[8 * i], cv[8 * i + 1]), pack_bf16x2(cv[8 * i + 2], cv[8 * i + 3]),
                                   pack_bf16x2(cv[8 * i + 4], cv[8 * i + 5]), pack_bf16x2(cv[8 * i + 6], cv[8 * i + 7]));
        __builtin_nontemporal_store(q.x, (unsigned*)cp_ + 4 * i);
        __builtin_nontemporal_store(q.y, (unsigned*)cp_ + 4 * i + 1);
        __builtin_nontemporal_store(q.z, (unsigned*)cp_ + 4 * i + 2);
        __builtin_nontemporal_store(q.w, (unsigned*)cp_ + 4 * i + 3);
      }
    }
    if (Zst) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        unsigned* zp_ = (unsigned*)(Zst + ((rb * R::NCB + (g * H + s * 32) / 32) * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(ge.w[g][i], zp_ + i);
      }
    }
    // stores are acknowledged in order: wait until only the ones issued after the exchange copy are outstanding
    // (2 h + 2 c + 4 gate-stash stores; they drain under the next step and at the latest at the end of the kernel)
    asm volatile("" ::: "memory");
    if constexpr (!TAGGED) {
      constexpr int NST = GateEnc<T, SIGM>::STORES;          // 16-byte stores per gate block
      if (Zst && Cout)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + 4 * NST) : "memory");
      else if (Zst)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + 4 * NST) : "memory");
      else if (Cout)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (pr.sp_out) {          // the rows of the last step are out: one more round for the consuming layer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// Cooperative form of the body above for sweeps with at most ONE tile per cluster (generation: 5 tiles) in inference:
// there a member's whole step -- 128 MFMAs and the cell update of 16 elements per lane -- sat on one wave while seven
// idled, all of it on the latency chain of the exchange.  Here waves 0-3 take one GATE each (x W and h U of that gate:
// a quarter of the products), park the pre-activations in LDS, and each then updates a quarter of the cells (accumulator
// registers 4w .. 4w+3 of all four gates, lane-local as before); wave 0 sends the h slice.  Same sums in the same
// order as the one-wave form: bit-identical results.  Two workgroup barriers per step; waves 4-7 only keep them company.
// TAGGED (default; DJ_KF_COUNTED_EXCHANGE keeps the counter): the h slices announce themselves as in the training sweep
// ("TAGGED exchange" above) -- a generated step is almost nothing BUT the exchange's latency chain.  The lower layer of
// the pair still moves its counter once per step for the upper layer, which reads ROWS, not fragments: lazily, behind a
// vmcnt(4) that only asks for the stores of the step BEFORE (long acknowledged), so closing step t promises the rows of
// step t-1 exactly as the counted protocol does, without anybody waiting for an acknowledgement.
template <bool SIGM, int NKX, bool TAGGED = false>
__device__ __forceinline__ void lstm_fwd_cluster_coop_body(const bf16_t* __restrict__ X, int DP,
                                                           const bf16_t* __restrict__ Wpack,
                                                           const float* __restrict__ bias,
                                                           const bf16_t* __restrict__ Upack, bf16_t* __restrict__ Hout,
                                                           int steps, int* __restrict__ cl, int ntiles, const ClPair pr,
                                                           int nblocks) {
  using T = bf16_t;
  constexpr int H = 256;
  using R = RecCfg<T, H>;
  using Frag = typename DjFrag<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Frag* Bw = (Frag*)smem_raw;                              // [4][NKX][64]  W slice of this member
  Frag* Bu = Bw + 4 * NKX * 64;                            // [4][NKC][64]  U slice
  unsigned char* hto = (unsigned char*)(Bu + 4 * R::NKC * 64);   // 8 x 4 KiB: x rounds of waves 0-3, h tile, gate buffer
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int bidx = (int)blockIdx.x - pr.role * nblocks;
  const int xcd = bidx & 7, j = bidx >> 3, s = j & (CL_M - 1);
  const int lcid = xcd + 8 * (j >> 3);
  const int cid = lcid + pr.role * (nblocks >> 6) * 8;
  const int64_t tile = lcid;                               // one tile per cluster
  const bool live = tile < ntiles;                         // uniform for the workgroup
  const bool gwave = w < 4;                                // gate waves; gate index = w
  const int64_t hx_tile = tile + (int64_t)pr.role * 128;
  int* cnt = cl + 2 * cid * CL_CNT_STRIDE;
  int* xccs = cnt + CL_CNT_STRIDE;
  int* fault = cl + CL_CNT_INTS;
  uint4* hxb = (uint4*)((unsigned char*)cl + CL_OFF_HX);
  const __amdgpu_buffer_rsrc_t hxr = cl_hx_rsrc(hxb);
  constexpr int ARRIVALS = CL_M;
  {
    const uint4* gw = (const uint4*)((const Frag*)Wpack + (int64_t)s * 4 * NKX * 64);
    const uint4* gu = (const uint4*)((const Frag*)Upack + (int64_t)s * 4 * R::NKC * 64);
    for (int i = tid; i < 4 * NKX * 64; i += 512) ((uint4*)Bw)[i] = gw[i];
    for (int i = tid; i < 4 * R::NKC * 64; i += 512) ((uint4*)Bu)[i] = gu[i];
  }
  float c4[4] = {0.f, 0.f, 0.f, 0.f};                      // cell state of accumulator registers 4w .. 4w+3
  const float bvg = bias[(w & 3) * H + s * 32 + l31];      // bias of this wave's gate
  const int my_xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) + 1;
  if (live && w == 0) {                                    // round 0 of the exchange: h_{-1} = 0
    uint4* hxo = hxb + ((hx_tile * 2 + 1) * 16 + 2 * s) * 64 + lane;
    hxo[0] = make_uint4(0, 0, 0, 0);
    hxo[64] = make_uint4(0, 0, 0, 0);
    if constexpr (TAGGED) {        // slot 0: zeros with the tag h_0 will NOT carry
      hxo[-16 * 64] = make_uint4(0, 0, 0, 0);
      hxo[-16 * 64 + 64] = make_uint4(0, 0, 0, 0);
    }
  }
  if (tid == 0) __hip_atomic_store(xccs + s, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int hook = __hip_atomic_load(fault + CLF_HOOK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests only
  if (tid == 0 && !((hook & 2) && s == CL_M - 1)) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int who = cl_who(2, cid, s, w);
  bool dead = false;                                       // wave-uniform: poisoned, no more waits (cl_wait_step)
  {
    bool bad = cl_wait_step(cnt, ARRIVALS, fault, who, -1, lane, dead);
    if (!bad) {
      int other = my_xcc;
      if (lane < CL_M) other = __hip_atomic_load(xccs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // hook bit 0 (DEEPJ_DEBUG_CLUSTER_FAULT): the launch behaves as if its clusters were spread over XCDs
      if (!__all(other == my_xcc) || (hook & 1)) {
        if (lane == 0 && w == 0) atomicAdd(fault + CLF_MISPLACED, 1);
        dead = bad = true;
      }
    }
    if (bad) {
#pragma unroll
      for (int e = 0; e < 4; ++e) c4[e] = __builtin_nanf("");
    }
  }
  int* gate = pr.gate ? cl + 2 * lcid * CL_CNT_STRIDE : nullptr;
  int seen = 0;
  if (gate && live && gwave && !dead) {
    bool bad = cl_wait_step(gate, ARRIVALS * 3, fault, who | CLW_GATE, -1, lane, dead, &seen);
    if (!bad) {
      int other = my_xcc;
      if (lane < CL_M) other = __hip_atomic_load(gate + CL_CNT_STRIDE + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!__all(other == my_xcc)) {
        if (lane == 0 && w == 0) atomicAdd(fault + CLF_MISPLACED, 1);
        dead = bad = true;
      }
    }
    if (bad) {
#pragma unroll
      for (int e = 0; e < 4; ++e) c4[e] = __builtin_nanf("");
    }
    asm volatile("" ::: "memory");
  }
  const float* spp[2] = {nullptr, nullptr};
  if (pr.sp_out && live && w == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = lane + 64 * i, row = v >> 2, cq = (v & 3) * 8;
      int64_t bb = (tile * 32 + row) / pr.n_seq;
      if (bb > pr.n_b - 1) bb = pr.n_b - 1;
      spp[i] = pr.sp_out + bb * steps * pr.sp_D + s * 32 + cq;
    }
  }
  unsigned char* xs = hto + (w & 3) * 4096;                // this wave's x rounds
  T* ht = (T*)hto;                                         // the h tile of the step (wave 0's slab, see the barriers)
  float* zgate = (float*)(hto + 4 * 4096);                 // [4 gates][64 lanes][16] pre-activations
  const int xr8 = lane >> 3, xc = lane & 7;
  constexpr int NR = (NKX + 3) / 4, NRA = (NR + 1) / 2;
  ClXRegs<NR> xq;
  if (live && gwave) cl_load_x<NR, 0, NR>(xq, X + (tile * steps * 32 + xr8) * DP + xc * 8, DP, xc);
  for (int t = 0; t < steps; ++t) {
    if (!live || !gwave) {         // exactly the two workgroup barriers of a gate wave's step
      __syncthreads();
      __syncthreads();
      continue;
    }
    const int64_t rb = tile * steps + t;
    float4 spv[2][2];
    if (pr.sp_out && w == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        spv[i][0] = *(const float4*)(spp[i] + (int64_t)t * pr.sp_D);
        spv[i][1] = *(const float4*)(spp[i] + (int64_t)t * pr.sp_D + 4);
      }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // ---- x_t W of gate w (before the wait)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(uint4*)(xs + (xr8 + 8 * i) * 128 + ((xc ^ xr8) << 4)) = xq.v[r][i];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kc = 4 * r + q;
        if (kc < NKX) {
          const Frag a = *(const Frag*)(xs + l31 * 128 + (((2 * q + h) ^ (l31 & 7)) << 4));
          dj_mfma(acc, a, Bw[(w * NKX + kc) * 64 + lane]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    const T* xnext = X + ((tile * steps + (t + 1 < steps ? t + 1 : t)) * 32 + xr8) * DP + xc * 8;
    // ---- h_{t-1} U of gate w
    {
      bool bad = false;
      if constexpr (!TAGGED) bad = cl_wait_step(cnt, ARRIVALS * (t + 1), fault, who, t, lane, dead);
      if (gate && t + 1 < steps && seen < ARRIVALS * (t + 4))
        bad |= cl_wait_step(gate, ARRIVALS * (t + 4), fault, who | CLW_GATE, t, lane, dead, &seen);
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      const unsigned hoff = (unsigned)((((hx_tile * 2 + ((t + 1) & 1)) * 16) * 64 + lane) * 16);   // bytes into the hx region
      uint4 ah[R::NKC];
#pragma unroll
      for (int kc = 0; kc < R::NKC; ++kc) ah[kc] = ld_sc1(hxr, hoff + kc * 1024);
      asm volatile("" ::: "memory");
      if constexpr (!TAGGED) {
        cl_load_x<NR, 0, NRA>(xq, xnext, DP, xc);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NRA) : "memory");
      } else {                     // the fragments whose tags match are h_{t-1}; the others are asked for again
        const unsigned e = cl_tag_of(t - 1);
        unsigned stale = 0;
#pragma unroll
        for (int kc = 0; kc < R::NKC; ++kc) stale |= cl_piece_stale(ah[kc], e);
        if (!dead && __any(stale != 0))
          bad |= cl_wait_tagged<R::NKC>(hxr, hoff, ah, e, cnt, fault, cl_who(7, cid, s, w), t, lane, dead);
#pragma unroll
        for (int kc = 0; kc < R::NKC; ++kc) {
          ah[kc].x &= ~CL_TAG;
          ah[kc].z &= ~CL_TAG;
        }
        cl_load_x<NR, 0, NRA>(xq, xnext, DP, xc);
      }
      if (bad) {
#pragma unroll
        for (int e = 0; e < 4; ++e) c4[e] = __builtin_nanf("");
      }
#pragma unroll
      for (int kc = 0; kc < R::NKC; ++kc) {
        Frag a;
        __builtin_memcpy(&a, &ah[kc], 16);
        dj_mfma(acc, a, Bu[(w * R::NKC + kc) * 64 + lane]);
      }
    }
    cl_load_x<NR, NRA, NR>(xq, xnext, DP, xc);
    // ---- gate w of all 16 registers -> LDS; then registers 4w .. 4w+3 of all four gates back
    {
      float* zq = zgate + (w * 64 + lane) * 16;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        *(float4*)(zq + 4 * r4) = make_float4(acc[4 * r4] + bvg, acc[4 * r4 + 1] + bvg, acc[4 * r4 + 2] + bvg, acc[4 * r4 + 3] + bvg);
    }
    __syncthreads();                                   // A: every gate is in LDS
    {
      const float4 zi4 = *(const float4*)(zgate + (0 * 64 + lane) * 16 + 4 * w);
      const float4 zf4 = *(const float4*)(zgate + (1 * 64 + lane) * 16 + 4 * w);
      const float4 zg4 = *(const float4*)(zgate + (2 * 64 + lane) * 16 + 4 * w);
      const float4 zo4 = *(const float4*)(zgate + (3 * 64 + lane) * 16 + 4 * w);
      const float zi[4] = {zi4.x, zi4.y, zi4.z, zi4.w}, zf[4] = {zf4.x, zf4.y, zf4.z, zf4.w};
      const float zg[4] = {zg4.x, zg4.y, zg4.z, zg4.w}, zo[4] = {zo4.x, zo4.y, zo4.z, zo4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * w + e;
        const float ig = dj_ract<SIGM>(zi[e]), fg = dj_ract<SIGM>(zf[e]), gg = dj_tanh(zg[e]), og = dj_ract<SIGM>(zo[e]);
        const float cn = fg * c4[e] + ig * gg;
        c4[e] = cn;
        ht[dj_crow(r, lane) * 32 + l31] = dj_from_f32<T>(og * dj_tanh(cn));
      }
    }
    __syncthreads();                                   // B: the h tile is complete (and the gate buffer is free again)
    if (w == 0) {
      // the exchange copy first (only it has to be acknowledged before the counter moves), then the row-major slice
      {
        uint4* hxo = hxb + ((hx_tile * 2 + (t & 1)) * 16 + 2 * s) * 64 + lane;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          uint4 pv = *(const uint4*)(ht + l31 * 32 + 16 * jj + 8 * h);
          if constexpr (TAGGED) {
            pv.x = (pv.x & ~CL_TAG) | cl_tag_of(t);
            pv.z = (pv.z & ~CL_TAG) | cl_tag_of(t);
            if ((hook & 4) && s == CL_M - 1 && t >= 2) continue;    // test hook: this member's slices stop arriving
          }
          hxo[jj * 64] = pv;
        }
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = lane + 64 * i, row = v >> 2, cq = (v & 3) * 8;
        uint4 hv = *(const uint4*)(ht + row * 32 + cq);
        if (pr.sp_out) {
          const float sp8[8] = {spv[i][0].x, spv[i][0].y, spv[i][0].z, spv[i][0].w,
                                spv[i][1].x, spv[i][1].y, spv[i][1].z, spv[i][1].w};
          T he[8], xo[8];
          __builtin_memcpy(he, &hv, 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) xo[e] = dj_from_f32<T>(dj_to_f32(he[e]) + sp8[e]);
          __builtin_memcpy(&hv, xo, 16);
        }
        cl_store_h<false>(Hout + (rb * 32 + row) * H + s * 32 + cq, hv);
      }
      asm volatile("" ::: "memory");
      if constexpr (!TAGGED) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (pr.sp_out) {      // lower layer of the pair: the counter the upper layer reads, one step behind (above)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (pr.sp_out) {          // the rows of the last step are out: one more round for the consuming layer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && live) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// fp32 (parity mode) sibling of the cooperative body for sweeps of at most 8 tiles in inference (generation: the
// certified sampling path runs in fp32).  The per-tile kernel streams 1 MB of fp32 U per step through ONE compute unit
// per tile (36 us per step with 5 tiles); here 8 workgroups share a tile, each with its 128 KB slice of U resident in
// LDS; x W + b comes precomputed and fragment-tiled (Zx, dj_gemm_nt c_mode 2) like for the per-tile kernel.  Waves 0-3
// take one gate each (32 k-chunks of the 32x32x2 fp32 MFMA), park the pre-activations in LDS and update a quarter of
// the cells each; wave 0 sends the h slice (fp32, A-fragment image: 4 chunks of 1 KiB per member and step).  Same sums
// in the same order as lstm_fwd_kernel<float, 256>: bit-identical results.  Exchange protocol, placement check and
// fault handling as in lstm_fwd_cluster_body.
template <bool SIGM>
__global__ __launch_bounds__(512) void lstm_fwd_cluster_f32_kernel(const float* __restrict__ Zx,
                                                                   const float* __restrict__ Upack,
                                                                   float* __restrict__ Hout, int steps,
                                                                   int* __restrict__ cl, int ntiles) {
  using T = float;
  constexpr int H = 256;
  using R = RecCfg<T, H>;
  using Frag = typename DjFrag<T>::type;                   // f32x4: 4 k per lane half, 8 k per chunk
  static_assert(R::NKC == 32 && R::NCB == 32, "fp32 H = 256 geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Frag* Bu = (Frag*)smem_raw;                              // [4][32][64]  U slice of this member: 128 KiB
  float* ht = (float*)(Bu + 4 * R::NKC * 64);              // [32 rows][32 units] h slice of the step
  float* zgate = ht + 32 * 32;                             // [4 gates][64 lanes][16] pre-activations
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, s = j & (CL_M - 1);
  const int cid = xcd + 8 * (j >> 3);
  const int64_t tile = cid;                                // one tile per cluster
  const bool live = tile < ntiles;
  const bool gwave = w < 4;
  int* cnt = cl + 2 * cid * CL_CNT_STRIDE;
  int* xccs = cnt + CL_CNT_STRIDE;
  int* fault = cl + CL_CNT_INTS;
  uint4* hxb = (uint4*)((unsigned char*)cl + CL_OFF_HX);   // [tile][parity][32 chunks][64 lanes] x 16 bytes
  const __amdgpu_buffer_rsrc_t hxr = cl_hx_rsrc(hxb);
  constexpr int ARRIVALS = CL_M;
  {
    const uint4* gu = (const uint4*)((const Frag*)Upack + (int64_t)s * 4 * R::NKC * 64);
    for (int i = tid; i < 4 * R::NKC * 64; i += 512) ((uint4*)Bu)[i] = gu[i];
  }
  float c4[4] = {0.f, 0.f, 0.f, 0.f};
  const int my_xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) + 1;
  if (live && w == 0) {                                    // round 0: h_{-1} = 0 into the parity-1 slots
    uint4* hxo = hxb + ((tile * 2 + 1) * 32 + 4 * s) * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) hxo[c * 64] = make_uint4(0, 0, 0, 0);
  }
  if (tid == 0) __hip_atomic_store(xccs + s, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int hook = __hip_atomic_load(fault + CLF_HOOK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // tests only
  if (tid == 0 && !((hook & 2) && s == CL_M - 1)) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int who = cl_who(3, cid, s, w);
  bool dead = false;                                       // wave-uniform: poisoned, no more waits (cl_wait_step)
  {
    bool bad = cl_wait_step(cnt, ARRIVALS, fault, who, -1, lane, dead);
    if (!bad) {
      int other = my_xcc;
      if (lane < CL_M) other = __hip_atomic_load(xccs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!__all(other == my_xcc) || (hook & 1)) {
        if (lane == 0 && w == 0) atomicAdd(fault + CLF_MISPLACED, 1);
        dead = bad = true;
      }
    }
    if (bad) {
#pragma unroll
      for (int e = 0; e < 4; ++e) c4[e] = __builtin_nanf("");
    }
  }
  // this gate wave's block of x W + b: fragment (gate w, units 32 s ..) of row block rb
  auto zxaddr = [&](int64_t rb) { return Zx + ((rb * R::NCB + ((w & 3) * H + s * 32) / 32) * 64 + lane) * 16; };
  float4 zx[4];
  if (live && gwave) {
#pragma unroll
    for (int i = 0; i < 4; ++i) zx[i] = ((const float4*)zxaddr(tile * steps))[i];
  }
  for (int t = 0; t < steps; ++t) {
    if (!live || !gwave) {         // exactly the two workgroup barriers of a gate wave's step
      __syncthreads();
      __syncthreads();
      continue;
    }
    const int64_t rb = tile * steps + t;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[4 * i] = zx[i].x; acc[4 * i + 1] = zx[i].y; acc[4 * i + 2] = zx[i].z; acc[4 * i + 3] = zx[i].w;
    }
    if (cl_wait_step(cnt, ARRIVALS * (t + 1), fault, who, t, lane, dead)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) c4[e] = __builtin_nanf("");
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    {
      const unsigned hoff = (unsigned)((((tile * 2 + ((t + 1) & 1)) * 32) * 64 + lane) * 16);
      uint4 ah[R::NKC];
#pragma unroll
      for (int kc = 0; kc < R::NKC; ++kc) ah[kc] = ld_sc1(hxr, hoff + kc * 1024);
      asm volatile("" ::: "memory");
      // next step's x W + b behind the h fragments (loads return in order): 4 requests, so the wait is a constant
      const float4* zn = (const float4*)zxaddr(t + 1 < steps ? rb + 1 : rb);
#pragma unroll
      for (int i = 0; i < 4; ++i) zx[i] = zn[i];
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
      for (int kc = 0; kc < R::NKC; ++kc) {
        Frag a;
        __builtin_memcpy(&a, &ah[kc], 16);
        dj_mfma(acc, a, Bu[((w & 3) * R::NKC + kc) * 64 + lane]);
      }
    }
    {
      float* zq = zgate + (w * 64 + lane) * 16;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        *(float4*)(zq + 4 * r4) = make_float4(acc[4 * r4], acc[4 * r4 + 1], acc[4 * r4 + 2], acc[4 * r4 + 3]);
    }
    __syncthreads();                                   // A: every gate is in LDS
    {
      const float4 zi4 = *(const float4*)(zgate + (0 * 64 + lane) * 16 + 4 * w);
      const float4 zf4 = *(const float4*)(zgate + (1 * 64 + lane) * 16 + 4 * w);
      const float4 zg4 = *(const float4*)(zgate + (2 * 64 + lane) * 16 + 4 * w);
      const float4 zo4 = *(const float4*)(zgate + (3 * 64 + lane) * 16 + 4 * w);
      const float zi[4] = {zi4.x, zi4.y, zi4.z, zi4.w}, zf[4] = {zf4.x, zf4.y, zf4.z, zf4.w};
      const float zg[4] = {zg4.x, zg4.y, zg4.z, zg4.w}, zo[4] = {zo4.x, zo4.y, zo4.z, zo4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * w + e;
        const float ig = dj_ract<SIGM>(zi[e]), fg = dj_ract<SIGM>(zf[e]), gg = dj_tanh(zg[e]), og = dj_ract<SIGM>(zo[e]);
        const float cn = fg * c4[e] + ig * gg;
        c4[e] = cn;
        ht[dj_crow(r, lane) * 32 + l31] = og * dj_tanh(cn);
      }
    }
    __syncthreads();                                   // B: the h tile is complete
    if (w == 0) {
      // exchange copy (A-fragment image: chunk 4 s + c holds units 8 c .. 8 c + 7: lane (l31, h) -> 4 k at 8 c + 4 h)
      {
        uint4* hxo = hxb + ((tile * 2 + (t & 1)) * 32 + 4 * s) * 64 + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) hxo[c * 64] = *(const uint4*)(ht + l31 * 32 + 8 * c + 4 * h);
      }
      asm volatile("" ::: "memory");
      // row-major slice: 32 rows x 128 bytes -> 4 x 16-byte vectors per lane
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = lane + 64 * i, row = v >> 3, cq = (v & 7) * 4;
        *(uint4*)(Hout + (rb * 32 + row) * H + s * 32 + cq) = *(const uint4*)(ht + row * 32 + cq);
      }
      asm volatile("" ::: "memory");
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // the exchange copy is acknowledged (stores in order)
      if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
template <bool SIGM, int NKX, bool TAGGED>
__global__ __launch_bounds__(512) void lstm_fwd_cluster_kernel(const bf16_t* __restrict__ X, int DP,
                                                               const bf16_t* __restrict__ Wpack,
                                                               const float* __restrict__ bias,
                                                               StashElem<bf16_t>* __restrict__ Zst,
                                                               const bf16_t* __restrict__ Upack,
                                                               bf16_t* __restrict__ Hout, bf16_t* __restrict__ Cout,
                                                               int steps, int* __restrict__ cl, int ntiles) {
  ClPair pr;
  pr.sp_out = nullptr; pr.sp_D = 0; pr.n_seq = 1; pr.n_b = 1; pr.gate = nullptr; pr.role = 0;
  lstm_fwd_cluster_body<SIGM, NKX, TAGGED>(X, DP, Wpack, bias, Zst, Upack, Hout, Cout, steps, cl, ntiles, pr, (int)gridDim.x);
}
// two stacked inference layers as a wavefront (ClPair): blocks [0, n) = the lower layer (input width 96 -> 8 k-chunks),
// blocks [n, 2n) = the upper one (256 -> 16); X1 is both the lower layer's output and the upper layer's input
struct ClPairArgs {
  const bf16_t* X0; int DP0; const bf16_t* W0; const float* b0; const bf16_t* U0;
  bf16_t* X1; const bf16_t* W1; const float* b1; const bf16_t* U1; bf16_t* H1;
  const float* sp1; int sp_D, n_seq, n_b;
};
template <bool SIGM, bool COOP, bool TAGGED = false>      // TAGGED: the cooperative body's members exchange tagged slices
__global__ __launch_bounds__(512) void lstm_fwd_cluster_pair_kernel(ClPairArgs a, int steps, int* __restrict__ cl, int ntiles) {
  const int nblocks = (int)gridDim.x >> 1;
  ClPair pr;
  if ((int)blockIdx.x < nblocks) {
    pr.sp_out = a.sp1; pr.sp_D = a.sp_D; pr.n_seq = a.n_seq; pr.n_b = a.n_b; pr.gate = nullptr; pr.role = 0;
    if (COOP)
      lstm_fwd_cluster_coop_body<SIGM, 8, TAGGED>(a.X0, a.DP0, a.W0, a.b0, a.U0, a.X1, steps, cl, ntiles, pr, nblocks);
    else
      lstm_fwd_cluster_body<SIGM, 8>(a.X0, a.DP0, a.W0, a.b0, nullptr, a.U0, a.X1, nullptr, steps, cl, ntiles, pr, nblocks);
  } else {
    pr.sp_out = nullptr; pr.sp_D = 0; pr.n_seq = 1; pr.n_b = 1; pr.gate = cl; pr.role = 1;
    if (COOP)
      lstm_fwd_cluster_coop_body<SIGM, 16, TAGGED>(a.X1, 256, a.W1, a.b1, a.U1, a.H1, steps, cl, ntiles, pr, nblocks);
    else
      lstm_fwd_cluster_body<SIGM, 16>(a.X1, 256, a.W1, a.b1, nullptr, a.U1, a.H1, nullptr, steps, cl, ntiles, pr, nblocks);
  }
}

// ---------------------------------------------------------------- backward (BPTT)
// Z: the forward's gate stash (GateDec above; read only); dZ: row-major [M,4H] output.
// DX (stationary-U^T builds only): the kernel also produces the layer's input gradient dX_t = dz_t W^T from
// the dz tile it holds in LDS -- with no U^T stream the W^T fragments are the only weight traffic of the step
// and cost less vector-memory time than a separate GEMM pass over dZ in HBM.
// DX == 2: only the LAST 32-column block of dX (columns 32*(NQ-1) ..., at most 4 of them valid: the `chosen`
// inputs of note layer 0) is produced here, its K range split over the waves (8 stationary fragments each) and
// the partial sums folded through LDS one step later; the GEMM then covers a multiple of 256 columns only.
// LDS-resident part of the streamed U^T (bf16 H = 256 only): KL k-chunks per wave behind the dz and dH tiles
template <typename T, int H> struct BwdUlds {
  using R = BwdCfg<T, H>;
  static constexpr int KL = (sizeof(T) == 2 && H == 256 && !R::STATB) ? 8 : 0;
  static constexpr size_t bytes = (size_t)R::NW * KL * 64 * 16;
  static constexpr size_t offset(int DX) {
    return (size_t)32 * (R::LDZ + (R::HOIST ? R::LDH : 0)) * sizeof(T) + (DX == 2 ? R::NW * 32 * 4 * sizeof(float) : 0);
  }
};
template <typename T, int H, bool SIGM, int DX>
__global__ __launch_bounds__((BwdCfg<T, H>::NT)) void lstm_bwd_kernel(const StashElem<T>* __restrict__ Z, const T* __restrict__ UTpack,
                                                       const T* __restrict__ C, const T* __restrict__ dH,
                                                       T* __restrict__ dZ, float* __restrict__ dbias, int steps,
                                                       const T* __restrict__ WTpack, int NQ, T* __restrict__ dX,
                                                       int DP, int64_t dz_cts, int ldz) {
  using R = BwdCfg<T, H>;
  using Frag = typename DjFrag<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* dzs = (T*)smem_raw;   // [32][LDZ]; its first 32*LDH elements double as the dH staging tile
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
  // streamed U^T (H = 256): the first KL k-chunks of every wave's slice stay in the LDS the tiles leave free
  // (KL x 8 KiB), the stream covers the rest -- the product is bound by the vector-memory path, LDS reads are not on it
  constexpr int KL = BwdUlds<T, H>::KL;
  Frag* ulw = (Frag*)(smem_raw + BwdUlds<T, H>::offset(DX)) + (w * KL) * 64 + lane;
  if constexpr (KL > 0) {
#pragma unroll
    for (int kc = 0; kc < KL; ++kc) ulw[kc * 64] = up[kc * 64];
  }
  Frag ub[R::STATB ? R::NKCB : 1];
  if constexpr (R::STATB) {
    static_assert(R::NJ == 1, "stationary U^T assumes one column tile per wave");
#pragma unroll
    for (int kc = 0; kc < R::NKCB; ++kc) ub[kc] = up[kc * 64];
  }
  constexpr int NWT = DX == 1 ? R::NKCB : DX == 2 ? R::NKCB / R::NW : 1;
  Frag wts[NWT];
  // DX == 2: [NW][32 rows][4 cols] partial sums of the last block, behind the dz tile and the dH staging tile
  float* xpart = (float*)(dzs + 32 * R::LDZ + 32 * R::LDH);
  if constexpr (DX == 1) {
    static_assert(R::STATB, "fused dX needs the stationary-U^T build");
    if (w < NQ) {
#pragma unroll
      for (int kc = 0; kc < R::NKCB; ++kc) wts[kc] = ((const Frag*)WTpack)[((int64_t)w * R::NKCB + kc) * 64 + lane];
    }
  }
  if constexpr (DX == 2) {
    static_assert(R::STATB && R::NKCB % R::NW == 0, "remainder dX needs the stationary-U^T build");
#pragma unroll
    for (int i = 0; i < NWT; ++i)       // this wave's K quarter of block NQ-1
      wts[i] = ((const Frag*)WTpack)[((int64_t)(NQ - 1) * R::NKCB + w * NWT + i) * 64 + lane];
  }
  // fold the partial sums of the step that has just been processed (row block rbp) and store 4 columns per row
  auto xpart_fold = [&](int64_t rbp) {
    if (w == 0 && h == 0) {
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v[c] = 0.f;
#pragma unroll
        for (int ww = 0; ww < R::NW; ++ww) v[c] += xpart[(ww * 32 + l31) * 4 + c];
      }
      dj_store4(dX + (rbp * 32 + l31) * DP + (NQ - 1) * 32, v[0], v[1], v[2], v[3]);
    }
  };
  auto zaddr = [&](int64_t rb, int g, int j) {
    return Z + ((rb * R::NCB + (g * H + w * R::UW + j * 32) / 32) * 64 + lane) * 16;
  };
  auto caddr = [&](int64_t rb, int j) { return C + ((rb * R::NCBH + (w * R::UW + j * 32) / 32) * 64 + lane) * 16; };

  // dH tile (32 x H, row-major): each thread stages NV (2 or 4) 16-byte vectors.  They are named
  // scalars on purpose: as a loop-carried array written through a lambda they were kept in scratch
  // memory, and the scratch store right behind the prefetch made every step wait for its HBM loads.
  constexpr int VPR = H / R::EPL, NV = 32 * VPR / R::NT;
  static_assert(NV == 2 || NV == 4 || NV == 8, "dH staging assumes 2, 4 or 8 vectors per thread");
  const uint4 z4 = make_uint4(0, 0, 0, 0);
  uint4 dh0, dh1, dh2 = z4, dh3 = z4, dh4 = z4, dh5 = z4, dh6 = z4, dh7 = z4;
  auto dh_ld = [&](int64_t rb, int i) {
    const int v = tid + R::NT * i, row = v / VPR, cv = (v % VPR) * R::EPL;
    return *(const uint4*)(dH + (rb * 32 + row) * H + cv);
  };
  // bf16: the dH staging tile has its own LDS region, which saves two of the four barriers per step;
  // f32 (LDS budget) aliases it on the first rows of dzs
  constexpr bool SPLIT = R::HOIST;
  T* dhs = SPLIT ? dzs + 32 * R::LDZ : dzs;
  auto dh_st = [&](int i, uint4 val) {
    const int v = tid + R::NT * i, row = v / VPR, cv = (v % VPR) * R::EPL;
    *(uint4*)(dhs + row * R::LDH + cv) = val;
  };
#define DJ_DH_LOAD(rbv)          \
  do {                           \
    dh0 = dh_ld((rbv), 0);       \
    dh1 = dh_ld((rbv), 1);       \
    if constexpr (NV >= 4) {     \
      dh2 = dh_ld((rbv), 2);     \
      dh3 = dh_ld((rbv), 3);     \
    }                            \
    if constexpr (NV == 8) {     \
      dh4 = dh_ld((rbv), 4);     \
      dh5 = dh_ld((rbv), 5);     \
      dh6 = dh_ld((rbv), 6);     \
      dh7 = dh_ld((rbv), 7);     \
    }                            \
  } while (0)
  // TAILPF (bf16, H = 128 with the stationary U^T): the stash of step t-1 (z_{t-1}, c_{t-2}, dH_{t-1}) is
  // requested at the start of step t's dz U^T product -- there is no weight stream it could delay, and
  // the product, the barriers and the dH staging are its head start (-0.22 ms on the note axis).  With
  // a streamed U^T (H = 256) the same requests issued after the last fragment load made the step
  // slower (+0.16 ms), so that kernel keeps requesting its stash at the top of the step.
  constexpr bool TAILPF = R::HOIST && R::STATB;
  Frag16<T> cnext[R::NJ];   // c_t of the step being processed (loaded as c_{t-1} one step earlier)
  Frag16<T> cprev[R::NJ];
  GateDec<T, SIGM> gd[R::NJ];
  DJ_DH_LOAD(tile * steps + steps - 1);
#pragma unroll
  for (int j = 0; j < R::NJ; ++j) {
    cnext[j].load(caddr(tile * steps + steps - 1, j));
    if constexpr (TAILPF) {
#pragma unroll
      for (int g = 0; g < 4; ++g) gd[j].load(g, zaddr(tile * steps + steps - 1, g, j));
      if (steps > 1) cprev[j].load(caddr(tile * steps + steps - 2, j));
    }
  }
  // stash requests of step t-1, issued from inside step t
  auto stash_prefetch = [&](int64_t rb, int t) {
#pragma unroll
    for (int j = 0; j < R::NJ; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) gd[j].load(g, zaddr(rb - 1, g, j));
      if (t > 1) cprev[j].load(caddr(rb - 2, j));
    }
  };

  for (int t = steps - 1; t >= 0; --t) {
    const int64_t rb = tile * steps + t;
    // stage dH_t into LDS, then pick it up in accumulator layout
    dh_st(0, dh0);
    dh_st(1, dh1);
    if constexpr (NV >= 4) {
      dh_st(2, dh2);
      dh_st(3, dh3);
    }
    if constexpr (NV == 8) {
      dh_st(4, dh4);
      dh_st(5, dh5);
      dh_st(6, dh6);
      dh_st(7, dh7);
    }
    if constexpr (!TAILPF) {
#pragma unroll
      for (int j = 0; j < R::NJ; ++j) {
        if constexpr (R::HOIST) {
#pragma unroll
          for (int g = 0; g < 4; ++g) gd[j].load(g, zaddr(rb, g, j));
        }
        if (t > 0) cprev[j].load(caddr(rb - 1, j));
      }
      if (t > 0) DJ_DH_LOAD(rb - 1);
    }
    lds_barrier();
    if constexpr (DX == 2) {
      if (t < steps - 1) xpart_fold(rb + 1);      // partials of step t+1: every wave wrote them before this barrier
    }
    float dhv[R::NJ][16];
#pragma unroll
    for (int j = 0; j < R::NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dhv[j][r] = dj_to_f32(dhs[dj_crow(r, lane) * R::LDH + w * R::UW + j * 32 + l31]) + acc[j][r];
    if constexpr (!SPLIT) lds_barrier();
#pragma unroll
    for (int j = 0; j < R::NJ; ++j) {
      const int u = w * R::UW + j * 32 + l31;
      if constexpr (!R::HOIST) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gd[j].load(g, zaddr(rb, g, j));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = dj_crow(r, lane);
        float ig, fg, gg, og, di, df, dO;
        gd[j].get(r, ig, fg, gg, og, di, df, dO);
        float ct = cnext[j].get(r);
        float cp = (t > 0) ? cprev[j].get(r) : 0.f;
        float dh = dhv[j][r];
        float tc = dj_tanh(ct);
        float dzo = dh * tc * dO;
        float dc = dcc[j][r] + dh * og * (1.f - tc * tc);
        float dzi = dc * gg * di;
        float dzf = dc * cp * df;
        float dzg = dc * ig * (1.f - gg * gg);
        dcc[j][r] = dc * fg;
        T* dp = dzs + row * R::LDZ + u;
        if constexpr (R::STATB) {     // the H = 128 kernels sit at 512 registers: the paired form spills there
          dp[0] = dj_from_f32<T>(dzi);
          dp[H] = dj_from_f32<T>(dzf);
          dp[2 * H] = dj_from_f32<T>(dzg);
          dp[3 * H] = dj_from_f32<T>(dzo);
        } else {
          dj_lds_put2(dp, dp + H, dzi, dzf);
          dj_lds_put2(dp + 2 * H, dp + 3 * H, dzg, dzo);
        }
        dbs[0][j] += dzi;
        dbs[1][j] += dzf;
        dbs[2][j] += dzg;
        dbs[3][j] += dzo;
      }
      if (t > 0) cnext[j].copy_from(cprev[j]);
    }
    lds_barrier();
    // dz_t tile -> global, coalesced: element (m, k) at dZ + (k >> 8) * dz_cts + m * ldz + (k & 255) -- row-major
    // (dz_cts 256, ldz 4H) or column-tile-major [4H/256][rows][256] (dz_cts rows * 256, ldz 256), where the 32 rows of
    // a step are one contiguous 16 KiB block per column tile: what the weight-gradient GEMM streams per stage
    constexpr int VPRZ = 4 * H / R::EPL;
#pragma unroll 4
    for (int v = tid; v < 32 * VPRZ; v += R::NT) {
      int row = v / VPRZ, cv = (v % VPRZ) * R::EPL;
      *(uint4*)(dZ + (int64_t)(cv >> 8) * dz_cts + (rb * 32 + row) * ldz + (cv & 255)) =
          *(const uint4*)(dzs + row * R::LDZ + cv);
    }
    if constexpr (DX == 2) {
      const T* apx = dzs + l31 * R::LDZ;
      f32x16 ax;
#pragma unroll
      for (int r = 0; r < 16; ++r) ax[r] = 0.f;
#pragma unroll
      for (int i = 0; i < NWT; ++i) {
        Frag a = dj_lds_frag(apx + (w * NWT + i) * R::KC, h);
        dj_mfma(ax, wts[i], a);
      }
      if (h == 0) *(float4*)(xpart + (w * 32 + l31) * 4) = make_float4(ax[0], ax[1], ax[2], ax[3]);   // columns 0..3
    }
    if constexpr (DX == 1) {
      // dX_t = dz_t [32 x 4H] * W^T [4H x D], D <= H: wave w owns the 32-column block q = w, whose W^T slice
      // is stationary in registers as well (wts, loaded before the sweep); operands are swapped so that a
      // lane holds 4 consecutive columns of ONE row per register quad.  (Streaming W^T instead -- wider
      // inputs -- was measured latency-bound at 16 fragments in flight per wave: +1.4 ms for -1.0 ms of GEMM.)
      if (w < NQ) {
        const T* apx = dzs + l31 * R::LDZ;
        f32x16 ax;
#pragma unroll
        for (int r = 0; r < 16; ++r) ax[r] = 0.f;
#pragma unroll
        for (int kc = 0; kc < R::NKCB; ++kc) {
          Frag a = dj_lds_frag(apx + kc * R::KC, h);
          dj_mfma(ax, wts[kc], a);
        }
        T* xrow = dX + (rb * 32 + l31) * DP;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = w * 32 + 8 * g + 4 * h;
          if (col < DP) dj_store4(xrow + col, ax[4 * g], ax[4 * g + 1], ax[4 * g + 2], ax[4 * g + 3]);
        }
      }
    }
    if (t > 0) {
      // dh_{t-1} (recurrent part) = dz_t [32 x 4H] * U^T [4H x H]; this wave's H/4 output units
#pragma unroll
      for (int j = 0; j < R::NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      const T* ap = dzs + l31 * R::LDZ;
      if constexpr (R::STATB) {
        if constexpr (TAILPF) {        // no weight stream to stay behind: the whole product is head start
          stash_prefetch(rb, t);
          DJ_DH_LOAD(rb - 1);
        }
#pragma unroll
        for (int kc = 0; kc < R::NKCB; ++kc) {
          Frag a = dj_lds_frag(ap + kc * R::KC, h);
          dj_mfma(acc[0], a, ub[kc]);
        }
      } else {
        Frag bq[R::PDB][R::NJ];
#pragma unroll
        for (int p = 0; p < R::PDB; ++p)
#pragma unroll
          for (int j = 0; j < R::NJ; ++j) bq[p][j] = up[(j * R::NKCB + KL + p) * 64];
        static_assert(R::UNRB % R::PDB == 0 && (R::NKCB - KL) % R::UNRB == 0, "ring / unroll geometry");
        static_assert(KL == 0 || R::NJ == 1, "LDS-resident chunks assume one column tile per wave");
        if constexpr (KL > 0) {          // LDS-resident chunks first: the ring's first fragments arrive meanwhile
#pragma unroll
          for (int kc = 0; kc < KL; ++kc) {
            Frag a = dj_lds_frag(ap + kc * R::KC, h);
            dj_mfma(acc[0], a, ulw[kc * 64]);
          }
        }
        // Blocks of UNRB chunks; every block but the last refills the ring PDB chunks ahead, the last one only where
        // the stream still has chunks (none past NKCB: the clamped refills of the last block used to re-request chunk
        // NKCB - 1 eight times per step and wave, 12 % of the streamed bytes).  The stream pointer and the A pointer
        // advance once per block and every chunk sits at a compile-time offset from them: with a run-time chunk
        // index hipcc computed a 64-bit address per fragment load (3-4 vector instructions per chunk) and issued the
        // block's eight loads together behind its last MFMA.
        constexpr int NBLK = (R::NKCB - KL) / R::UNRB;
        const Frag* upk = up + KL * 64;
        const T* apk = ap + KL * R::KC;
#pragma unroll 1
        for (int blk = 0; blk < NBLK - 1; ++blk) {
#pragma unroll
          for (int u = 0; u < R::UNRB; ++u) {
            Frag a = dj_lds_frag(apk + u * R::KC, h);
#pragma unroll
            for (int j = 0; j < R::NJ; ++j) dj_mfma(acc[j], a, bq[u % R::PDB][j]);
#pragma unroll
            for (int j = 0; j < R::NJ; ++j) bq[u % R::PDB][j] = upk[(j * R::NKCB + u + R::PDB) * 64];
          }
          upk += R::UNRB * 64;
          apk += R::UNRB * R::KC;
        }
#pragma unroll
        for (int u = 0; u < R::UNRB; ++u) {
          Frag a = dj_lds_frag(apk + u * R::KC, h);
#pragma unroll
          for (int j = 0; j < R::NJ; ++j) dj_mfma(acc[j], a, bq[u % R::PDB][j]);
          if (u + R::PDB < R::UNRB) {
#pragma unroll
            for (int j = 0; j < R::NJ; ++j) bq[u % R::PDB][j] = upk[(j * R::NKCB + u + R::PDB) * 64];
          }
        }
      }
    }
    // (SPLIT) the next step's first barrier already orders this step's dzs reads before its gate writes
    if constexpr (!SPLIT) lds_barrier();
  }
  if constexpr (DX == 2) {
    lds_barrier();
    xpart_fold(tile * steps);
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

#undef DJ_DH_LOAD

// ---------------------------------------------------------------- backward (BPTT), bf16 H = 256: SPLIT gate math (round 5)
// The time-axis BPTT sweep (model.py:84 under TF autodiff; the dominant kernel of the training step).  lstm_bwd_kernel above
// runs a step of a tile as a strict chain -- stash loads, gate math, dz tile, dz U^T product -- in which the product
// (46 % of the step) is bound by the compute unit's vector-memory path streaming U^T from L2 with the vector ALUs idle,
// and the gate math (28 %) is bound by vector-instruction issue with the memory path idle (phase stamps, DESIGN.md
// section 8 round 4).  Only SIX operations per cell depend on dh_t, the thing the product delivers:
//     dzo = dh A          dc = dcc + dh B       dzi = dc Ci      dzf = dc Cf      dzg = dc Cg      dcc' = dc fg
// with A = tanh(c_t) o',  B = o (1 - tanh(c_t)^2),  Ci = g i',  Cf = c_{t-1} f',  Cg = i (1 - g^2)  -- all of which come
// from the forward stash alone (gate codes, c_t, c_{t-1}).  So this kernel computes the FACTORS of step t-1 -- code
// decoding, the tanh, two thirds of the vector instructions of a step -- INSIDE the product loop of step t, one cell
// per group of k-chunks, in the issue slots the fragment stream leaves empty, and the part of a step that nothing
// overlaps shrinks to the six operations above, the dz tile and its barriers.  Order of a step t (tile = 32 sequences):
//   top       wait for the dH tile of step t (LDS-DMA, requested a step earlier); REQUEST dH_{t-1} (DMA into the other
//             dH tile) and the stash of step t-1 (gate codes, c_{t-2}) -- nothing waits for them before the product
//   barrier   (dH tile visible; every wave is done reading the dz tile of step t+1)
//   finish    dh_t = dH_t + acc;  the six operations per cell with the factors of step t (fp16 pairs, v_fma_mix_f32);
//             dz_t -> LDS, k-major: 16 ds_write_b64 per wave instead of 64 ds_write_b16;  bias sums
//   barrier   (dz tile complete)
//   product   acc = dz_t U^T (A operand out of the k-major tile by ds_read_b64_tr_b16; U^T: KL chunks per column block
//             resident in LDS, KS stationary in registers, the rest through a ring of PD), with the factors of step t-1
//             and the rows of dz_t on their way to HBM (4 rows x 256 bytes per store instruction) between its chunks
// What the measurements say about it: 2.91 -> 2.78 ms per training step; the vector work
// moved into the product is NOT hidden at two waves per SIMD (ablations: the step is the sum of MFMA + finish, U^T stream,
// factor math and row stores), and a wave's vector memory returns in order, so an HBM request in front of the ring costs
// the ring its latency.
// NJ = 32-unit column blocks per wave: 1 = eight waves (two per SIMD, 256 registers each; the product form <1, 4, 8>),
// 2 = four waves (one per SIMD, 512 registers: measured 3.19 - 3.34 ms, hipcc fills 430 - 470 registers before the first
// stationary fragment).
struct Bwd256 {
  static constexpr int H = 256, KC = 16, NKCB = 64, NCB = 32, NCBH = 8;
  static constexpr int KL = 8;                                     // k-chunks per column block resident in LDS
  // LDS: the dz tile K-MAJOR -- [1024 k][32 rows] bf16, 64 bytes per k, the 8-byte piece of rows 4c .. 4c+3 stored at
  // piece c ^ (k & 7) -- then TWO dH tiles [32][256] (filled by LDS-DMA, unpadded; step t reads tile t & 1 while the DMA of
  // step t-1 fills the other), then the resident U^T chunks: all 160 KiB of the compute unit
  static constexpr size_t off_dh = (size_t)4 * H * 64;
  static constexpr size_t dh_bytes = (size_t)32 * H * sizeof(bf16_t);
  static constexpr size_t off_u = off_dh + 2 * dh_bytes;
  static constexpr size_t smem = off_u + (size_t)8 * KL * 64 * 16;   // 65,536 + 2 x 16,384 + 65,536 = 163,840 bytes
};
// The dh-independent factors of the 16 cells a lane holds of one column block, packed in PAIRS as fp16 (48 registers
// instead of 96: the eight-wave form has 256 registers per wave and spilled 132 of them with fp32 factors).  fp16, not
// bf16: v_fma_mix_f32 takes either half of a packed register as an fp16 operand of an fp32 fma, so the finish phase
// pays nothing for the packing; 11 significant bits are 8x finer than the 8-bit gate codes the factors are made from
// (A, Ci <= 1/4, B, Cg, fg <= 1, |Cf| = |c_{t-1}| f' <= 0.25 |c|: all far inside fp16's range; a NaN stays a NaN).
typedef _Float16 dj_h2 __attribute__((ext_vector_type(2)));
struct BwdFac {
  dj_h2 ab[16], cc[16], gf[16];                 // (A, B), (Ci, Cf), (Cg, fg)
};
__device__ __forceinline__ dj_h2 dj_pack_h2(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_convertvector(v, dj_h2);     // v_cvt_pk_f16_f32 (round to nearest even)
}
// x * (fp16 half of p) [+ c] as ONE v_fma_mix_f32 (HI: the upper half).  Assembly, register-only: written as
// fmaf(x, (float)p[i], c) the SLP vectorizer paired the operations of two cells into v_pk_fma_f32 behind two
// v_cvt_f32_f16_sdwa each -- three instructions per product instead of one, and the packed zero addends were spilled and
// reloaded behind a vmcnt(0) inside the sweep.
template <bool HI> __device__ __forceinline__ float dj_mix_mul(float x, dj_h2 p) {
  float d;
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(x), "v"(p));
  else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(x), "v"(p));
  return d;
}
template <bool HI> __device__ __forceinline__ float dj_mix_fma(float x, dj_h2 p, float c) {
  float d;
  if constexpr (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(x), "v"(p), "v"(c));
  else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(d) : "v"(x), "v"(p), "v"(c));
  return d;
}
template <bool SIGM>
__device__ __forceinline__ void bwd_factors(const GateDec<bf16_t, SIGM>& gd, const Frag16<bf16_t>& ct,
                                            const Frag16<bf16_t>& cp, float cpm, int r, BwdFac& F) {
  float ig, fg, gg, og, di, df, dO;
  gd.get(r, ig, fg, gg, og, di, df, dO);
  const float tc = dj_tanh(ct.get(r));
  F.ab[r] = dj_pack_h2(tc * dO, og * (1.f - tc * tc));
  F.cc[r] = dj_pack_h2(gg * di, cp.get(r) * cpm * df);            // cpm = 0 for step 0: c_{-1} = 0
  F.gf[r] = dj_pack_h2(ig * (1.f - gg * gg), fg);
}
// LDS-DMA of 16 bytes per lane from (scalar base + 32-bit lane offset): destination = lds_base (wave-uniform, via M0) +
// 16 * lane.  Inline assembly so that hipcc neither drains it at the next barrier nor makes every later register load
// wait vmcnt(0) for it (cdna_hip_programming.md section 5, "Pipelining across barriers"): it has no register destination
// (the round-4 hazard does not apply), and an operation hipcc does not count can only make its own counted waits wait
// for MORE than they need, never for less.  Its completion is waited for by hand (the counted wait at the top of a step).
typedef short dj_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dj_glds16_s(const void* sbase, unsigned voff, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_base)
               : "memory");
}
__device__ __forceinline__ unsigned dj_lds_addr(const void* p) {
  return (unsigned)(unsigned long)(const __attribute__((address_space(3))) void*)p;
}
// ds_read_b64_tr_b16: per group of 16 lanes a 4 x 16 block of 16-bit elements, delivered column-major -- lane 4q + p of
// the group supplies the address of 4 elements (8 bytes), lane i receives element (i & 3) of the pieces of lanes
// (i >> 2), 4 + (i >> 2), 8 + (i >> 2), 12 + (i >> 2)
__device__ __forceinline__ dj_s16x4 dj_lds_tr(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((dj_s16x4 __attribute__((address_space(3)))*)p);
}
union DjTrFrag {
  dj_s16x4 h[2];
  bf16x8 v;
  uint4 q;
};
template <bool SIGM, int NJ, int KS, int PD>
__global__ __launch_bounds__(512 / NJ) void lstm_bwd256_kernel(const uint8_t* __restrict__ Z, const bf16_t* __restrict__ UTpack,
                                                               const bf16_t* __restrict__ C, const bf16_t* __restrict__ dH,
                                                               bf16_t* __restrict__ dZ, float* __restrict__ dbias, int steps,
                                                               int64_t dz_cts, int ldz) {
  using B = Bwd256;
  using Frag = bf16x8;
  constexpr int NW = 8 / NJ, NT = 64 * NW, H = B::H;
  constexpr int NS = B::NKCB - B::KL - KS;                         // streamed k-chunks per column block
  static_assert(NS >= PD && (NJ == 1 || NJ == 2), "geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* dzk = smem_raw;                                    // dz tile, k-major (Bwd256)
  const bf16_t* dhs = (const bf16_t*)(smem_raw + B::off_dh);        // [2][32][256]
  Frag* uls = (Frag*)(smem_raw + B::off_u);                         // [8 column blocks][KL][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = blockIdx.x;

  // Every global access is a BUFFER access: a descriptor per operand whose base is this tile's part of it (wave-uniform),
  // the lane's bytes as a 32-bit vector offset computed once, the step / chunk as a scalar offset -- one instruction per
  // request and no 64-bit vector address anywhere (the plain kernel spends 25 v_lshl_add_u64 per step on them, and this
  // one has no registers to keep such addresses in).
  auto rsrc = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, -1, 0x00020000); };
  const __amdgpu_buffer_rsrc_t ur = rsrc(UTpack);
  const __amdgpu_buffer_rsrc_t zr = rsrc(Z + tile * steps * (B::NCB * 1024));               // [step][32 blocks][1 KiB]
  const __amdgpu_buffer_rsrc_t cr = rsrc(C + tile * steps * (B::NCBH * 1024));              // [step][8 blocks][2 KiB]
  const __amdgpu_buffer_rsrc_t dr = rsrc(dZ + tile * steps * 32 * (int64_t)ldz);            // rows of this tile
  const bf16_t* dHt = dH + tile * steps * (32 * H);                                         // [step][32 rows][512 B]
  auto ldb = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  const int lane16 = lane * 16;
  auto ldu = [&](int j, int kc) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ur, lane16, ((w * NJ + j) * B::NKCB + kc) * 1024, 0);
    return __builtin_bit_cast(Frag, v);
  };
  // resident parts of U^T: KL chunks per column block in LDS, KS in registers
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int kc = 0; kc < B::KL; ++kc) uls[((w * NJ + j) * B::KL + kc) * 64 + lane] = ldu(j, kc);
  Frag us[NJ][KS > 0 ? KS : 1];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int kc = 0; kc < KS; ++kc) us[j][kc] = ldu(j, B::KL + kc);

  f32x16 acc[NJ];
  float dcc[NJ][16], dbs[4][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[j][r] = 0.f;
      dcc[j][r] = 0.f;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) dbs[g][j] = 0.f;
  }
  GateDec<bf16_t, SIGM> gd[NJ];
  Frag16<bf16_t> cA[NJ], cB[NJ];          // entering step t: cA = c_{t-1}; cB receives c_{t-2}
  BwdFac F[NJ];
  // stash of step `st` of this tile: gate codes and c
  auto ld_codes = [&](int st) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) gd[j].q[g] = ldb(zr, lane16, (st * B::NCB + g * 8 + w * NJ + j) * 1024);
  };
  auto ld_c = [&](Frag16<bf16_t>(&c)[NJ], int st) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int so = (st * B::NCBH + w * NJ + j) * 2048;
      c[j].v[0] = ldb(cr, lane * 32, so);
      c[j].v[1] = ldb(cr, lane * 32, so + 16);
    }
  };
  // dH tile of step `st` (32 x 256 bf16, row-major: 16 contiguous KiB) -> LDS by DMA, NI pieces of 1 KiB per wave
  constexpr int NI = 16 / NW;
  const unsigned dhs_lds = __builtin_amdgcn_readfirstlane(dj_lds_addr(dhs));
  auto dh_dma = [&](int st) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int piece = w * NI + i;
      dj_glds16_s(dHt + (int64_t)st * (32 * H) + piece * 512, lane16,
                  dhs_lds + (unsigned)(st & 1) * (unsigned)B::dh_bytes + piece * 1024);
    }
  };
  // ---- the dz tile in LDS, K-MAJOR: element (row m, column k) at byte k * 64 + (((m >> 2) ^ (k & 7)) << 3) + 2 (m & 3).
  // A lane of the finish phase holds, per gate, rows 8 rg + 4 h + (0..3) of ONE column for rg = 0..3: four 8-byte
  // stores per gate (ds_write_b64) instead of sixteen 2-byte ones into a row-major tile -- that phase was bound by its
  // 512 LDS store instructions per step and compute unit.  The product's A operand (a lane: 8 consecutive k of one row)
  // and the rows that go to HBM (16 bytes = 8 consecutive k of one row) come back out through ds_read_b64_tr_b16.
  // The XOR keeps all three patterns off each other's banks (stores: 16 consecutive k per lane group; operand reads: 4
  // consecutive k x 8 pieces per half wave -- every bank once).
  const int q4 = (lane & 15) >> 2, p4 = lane & 3, g16 = lane >> 4;
  int wadr[NJ];                              // finish-phase store address for rg = 0 and gate 0 (rg: ^ 16 rg; gate: + 16 KiB)
#pragma unroll
  for (int j = 0; j < NJ; ++j) wadr[j] = (((w * NJ + j) * 32 + l31) << 6) | ((h ^ (l31 & 7)) << 3);
  // operand reads of k-chunk kc: k = 16 kc + 8 h + 4 rd + q4, rows 16 (g16 & 1) + 4 p4 .. (piece 4 (g16 & 1) + p4)
  const int ra0 = ((8 * h + q4) << 6) | ((((4 * (g16 & 1) + p4) ^ q4) & 7) << 3);
  const int ra1 = ((8 * h + 4 + q4) << 6) | ((((4 * (g16 & 1) + p4) ^ (4 + q4)) & 7) << 3);
  // rows to HBM: instruction i of wave w covers region i + NR w of the tile's 64 regions (4 rows x 128 columns each):
  // piece c = region & 7, columns 128 (region >> 3) ...; the lane loads k = 128 kr + 32 g16 + 8 p4 + 4 rd + q4 of that
  // piece and receives row 4 c + (lane & 3), columns 128 kr + 32 g16 + 8 q4 + (0..7)
  constexpr int NR = 64 / NW;
  const int rs0 = ((32 * g16 + 8 * p4 + q4) << 6) | (q4 << 3);
  const int rs1 = ((32 * g16 + 8 * p4 + 4 + q4) << 6) | ((4 + q4) << 3);
  const int dzv_off = ((lane & 3) * ldz + (4 * g16 + q4) * 8) * 2;
  // (the two lane addresses are made opaque per step, `rs0v` / `rs1v` below: as loop invariants the compiler computed all
  // 2 NR variants before the sweep and then spilled them -- one v_xor + one v_add per read is cheaper than a reload)
  auto dz_store = [&](int t, int i, int rs0v, int rs1v) {
    const int region = w * NR + i, c = region & 7, kr = region >> 3;
    DjTrFrag v;
    v.h[0] = dj_lds_tr(dzk + (rs0v ^ (c << 3)) + kr * 8192);
    v.h[1] = dj_lds_tr(dzk + (rs1v ^ (c << 3)) + kr * 8192);
    const u32x4 o = {v.q.x, v.q.y, v.q.z, v.q.w};
    const unsigned so = (unsigned)(((kr >> 1) * dz_cts + (int64_t)(t * 32 + 4 * c) * ldz + (kr & 1) * 128) * 2);
    __builtin_amdgcn_raw_buffer_store_b128(o, dr, dzv_off, (int)so, 0);
  };

  // prologue: the factors of the last step, with nothing to hide behind
  {
    const int sl = steps - 1;
    dh_dma(sl);
    ld_codes(sl);
    ld_c(cA, sl);
    ld_c(cB, sl > 0 ? sl - 1 : sl);
    const float cpm = steps > 1 ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bwd_factors<SIGM>(gd[j], cA[j], cB[j], cpm, r, F[j]);
      cA[j].copy_from(cB[j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the first dH tile has landed (this wave's pieces)
  }

  for (int t = steps - 1; t >= 0; --t) {
    // ---- top: requests of step t-1 (clamped at the tile's first rows: step 0 re-requests its own and ignores them --
    // unconditional requests keep the compiler's vmcnt bookkeeping exact).  The dH tile of step t was requested by DMA
    // at the top of the step before; the only operations of this wave younger than it that may still be in flight are
    // the NR row stores of that step: vector memory completes in order, so all but the NR youngest done means the DMA
    // has landed (first step: the prologue's vmcnt(0)).  The DMA of step t-1 goes into the OTHER dH tile, here and not
    // behind the second barrier: requests return in order, and in front of the product's ring it held the first
    // fragments back by a whole HBM latency (measured: +3.4 k cycles per step).
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NR) : "memory");
    const int t1 = t > 0 ? t - 1 : t, t2 = t > 1 ? t - 2 : t1;
    dh_dma(t1);
    ld_codes(t1);
    ld_c(cB, t2);
    lds_barrier();
    // ---- finish step t: the six dh-dependent operations per cell
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int u = (w * NJ + j) * 32 + l31;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        float dz[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * rg + e, row = dj_crow(r, lane);
          const float dh = dj_to_f32(dhs[(t & 1) * (32 * H) + row * H + u]) + acc[j][r];
          // every factor enters as the fp16 operand of a v_fma_mix_f32, either half of its pair
          const float dc = dj_mix_fma<true>(dh, F[j].ab[r], dcc[j][r]);
          dz[0][e] = dj_mix_mul<false>(dc, F[j].cc[r]);
          dz[1][e] = dj_mix_mul<true>(dc, F[j].cc[r]);
          dz[2][e] = dj_mix_mul<false>(dc, F[j].gf[r]);
          dz[3][e] = dj_mix_mul<false>(dh, F[j].ab[r]);
          dcc[j][r] = dj_mix_mul<true>(dc, F[j].gf[r]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *(uint2*)(dzk + g * (H * 64) + (wadr[j] ^ (rg << 4))) =
              make_uint2(pack_bf16x2(dz[g][0], dz[g][1]), pack_bf16x2(dz[g][2], dz[g][3]));
          dbs[g][j] += (dz[g][0] + dz[g][1]) + (dz[g][2] + dz[g][3]);
        }
      }
    }
    lds_barrier();
    int rs0v = rs0, rs1v = rs1;
    asm volatile("" : "+v"(rs0v), "+v"(rs1v));
    // ---- product of step t with the factors of step t-1 -- and the rows of dz_t on their way to HBM -- in its gaps
    if (t > 0) {
      const float cpm = t > 1 ? 1.f : 0.f;
      // the ring's first fragments: requested here, not at the top of the step (32 registers the finish phase needs);
      // the resident chunks and the first cells of factors are their head start, and no store is in front of them
      Frag bq[PD][NJ];
#pragma unroll
      for (int p = 0; p < PD; ++p)
#pragma unroll
        for (int j = 0; j < NJ; ++j) bq[p][j] = ldu(j, B::KL + KS + p);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      constexpr int CPE = 4 / NJ;                                  // k-chunks per cell of factors (16 NJ cells, 64 chunks)
      // every index below is a compile-time constant (dj_static_for: a `#pragma unroll` of 64 such iterations was only
      // unrolled 8-fold, and the run-time chunk index put the gate codes and the factors on the stack)
      dj_static_for<0, B::NKCB>([&](auto kcc) {
        constexpr int kc = decltype(kcc)::value;
        DjTrFrag a;
        a.h[0] = dj_lds_tr(dzk + ra0 + kc * 1024);
        a.h[1] = dj_lds_tr(dzk + ra1 + kc * 1024);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if constexpr (kc < B::KL) {
            dj_mfma(acc[j], a.v, uls[((w * NJ + j) * B::KL + kc) * 64 + lane]);
          } else if constexpr (kc < B::KL + KS) {
            dj_mfma(acc[j], a.v, us[j][kc - B::KL]);
          } else {
            constexpr int sl = (kc - B::KL - KS) % PD;
            dj_mfma(acc[j], a.v, bq[sl][j]);
            if constexpr (kc + PD < B::NKCB) bq[sl][j] = ldu(j, kc + PD);
          }
        }
        if constexpr ((kc + 1) % CPE == 0) {
          constexpr int e = (kc + 1) / CPE - 1;
          bwd_factors<SIGM>(gd[e / 16], cA[e / 16], cB[e / 16], cpm, e % 16, F[e / 16]);
          // one of the NR row-store instructions of this wave per 64 / NR chunks (the tile is complete since the second
          // barrier and only read from here on).  Behind the product they were a phase of their own -- two conflicting
          // LDS reads, a wait and a store, NR times in a row: 3 k cycles of a 22 k step; here they cost issue slots
          if constexpr ((kc + 1) % (B::NKCB / NR) == 0) dz_store(t, (kc + 1) / (B::NKCB / NR) - 1, rs0v, rs1v);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
#pragma unroll
      for (int j = 0; j < NJ; ++j) cA[j].copy_from(cB[j]);
    } else {
      // ---- step 0 has no product: dz_0 -> global in one go.  Per instruction 4 rows x 256 bytes (two whole lines per
      // row); element (m, k) at dZ + (k >> 8) * dz_cts + m * ldz + (k & 255) (row-major: dz_cts 256, ldz 4H;
      // column-tile-major [4H/256][rows][256]: dz_cts rows * 256, ldz 256)
#pragma unroll
      for (int i = 0; i < NR; ++i) dz_store(t, i, rs0v, rs1v);
    }
  }
  if (dbias) {
    // the lane id afresh (mbcnt): nothing derived from threadIdx has to survive the sweep for this (it was spilled)
    const int le = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float v = dbs[g][j];
        v += __shfl_xor(v, 32);
        if (le < 32) atomicAdd(dbias + g * H + (w * NJ + j) * 32 + le, v);
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
int launch_fwd(int ntiles, int steps, const void* Zx, void* Gst, const void* Upack, void* Hout, void* Cout, int sigm,
               hipStream_t st) {
  if (sizeof(T) == 2 && Gst == Zx) return 1017;      // the 8-bit gate stash cannot overwrite the bf16 projections
  if (sigm)
    hipLaunchKernelGGL((lstm_fwd_kernel<T, H, true>), dim3(ntiles), dim3(RecCfg<T, H>::NT), 0, st, (const T*)Zx,
                       (StashElem<T>*)Gst, (const T*)Upack, (T*)Hout, (T*)Cout, steps);
  else
    hipLaunchKernelGGL((lstm_fwd_kernel<T, H, false>), dim3(ntiles), dim3(RecCfg<T, H>::NT), 0, st, (const T*)Zx,
                       (StashElem<T>*)Gst, (const T*)Upack, (T*)Hout, (T*)Cout, steps);
  return (int)hipGetLastError();
}
template <typename T, int H, int DX>
int launch_bwd_x(int ntiles, int steps, const void* Z, const void* UTpack, const void* C, const void* dH, void* dZ,
                 int64_t dz_cts_in, float* dbias, int sigm, const void* WTpack, int NQ, void* dX, int DP, hipStream_t st) {
  using R = BwdCfg<T, H>;
  const int64_t dz_cts = dz_cts_in ? dz_cts_in : 256;
  const int ldz = dz_cts_in ? 256 : 4 * H;
  if (dz_cts_in && dz_cts_in < (int64_t)ntiles * steps * 32 * 256) return 1018;
  size_t smem = BwdUlds<T, H>::offset(DX) + BwdUlds<T, H>::bytes;
  static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_bwd_kernel<T, H, false, DX>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)lstm_bwd_kernel<T, H, true, DX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  if (sigm)
    hipLaunchKernelGGL((lstm_bwd_kernel<T, H, true, DX>), dim3(ntiles), dim3(R::NT), smem, st, (const StashElem<T>*)Z,
                       (const T*)UTpack, (const T*)C, (const T*)dH, (T*)dZ, dbias, steps, (const T*)WTpack, NQ, (T*)dX, DP,
                       dz_cts, ldz);
  else
    hipLaunchKernelGGL((lstm_bwd_kernel<T, H, false, DX>), dim3(ntiles), dim3(R::NT), smem, st, (const StashElem<T>*)Z,
                       (const T*)UTpack, (const T*)C, (const T*)dH, (T*)dZ, dbias, steps, (const T*)WTpack, NQ, (T*)dX, DP,
                       dz_cts, ldz);
  return (int)hipGetLastError();
}
// bf16 H = 256: the split-gate-math sweep (lstm_bwd256_kernel)
template <int NJ, int KS, int PD>
int launch_bwd256(int ntiles, int steps, const void* Z, const void* UTpack, const void* C, const void* dH, void* dZ,
                  int64_t dz_cts_in, float* dbias, int sigm, hipStream_t st) {
  const int64_t dz_cts = dz_cts_in ? dz_cts_in : 256;
  const int ldz = dz_cts_in ? 256 : 4 * 256;
  if (dz_cts_in && dz_cts_in < (int64_t)ntiles * steps * 32 * 256) return 1018;
  static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_bwd256_kernel<false, NJ, KS, PD>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Bwd256::smem);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)lstm_bwd256_kernel<true, NJ, KS, PD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)Bwd256::smem);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  if (sigm)
    hipLaunchKernelGGL((lstm_bwd256_kernel<true, NJ, KS, PD>), dim3(ntiles), dim3(512 / NJ), Bwd256::smem, st, (const uint8_t*)Z,
                       (const bf16_t*)UTpack, (const bf16_t*)C, (const bf16_t*)dH, (bf16_t*)dZ, dbias, steps, dz_cts, ldz);
  else
    hipLaunchKernelGGL((lstm_bwd256_kernel<false, NJ, KS, PD>), dim3(ntiles), dim3(512 / NJ), Bwd256::smem, st, (const uint8_t*)Z,
                       (const bf16_t*)UTpack, (const bf16_t*)C, (const bf16_t*)dH, (bf16_t*)dZ, dbias, steps, dz_cts, ldz);
  return (int)hipGetLastError();
}
template <typename T, int H>
int launch_bwd(int ntiles, int steps, const void* Z, const void* UTpack, const void* C, const void* dH, void* dZ,
               int64_t dz_cts, float* dbias, int sigm, const void* WTpack, int NQ, void* dX, int DP, uint32_t kf,
               hipStream_t st) {
  if constexpr (sizeof(T) == 2 && H == 256) {
    // eight waves, 4 k-chunks per column block stationary in registers next to the 8 in LDS, ring of 8 (round 5 A/B in
    // one call, lstm_bwd_time per training step: <1,4,8> 2.76 ms, <1,0,8> 2.86, <1,0,12> 2.85, lstm_bwd_kernel 2.89; the
    // four-wave forms <2,0,8> 3.34 and <2,8,8> 3.19: DESIGN.md section 8 round 5 -- they instantiate from this template)
    if (!WTpack && !(kf & DJ_KF_BWD_PLAIN))
      return launch_bwd256<1, 4, 8>(ntiles, steps, Z, UTpack, C, dH, dZ, dz_cts, dbias, sigm, st);
  }
  if (WTpack) {
    if constexpr (RecCfg<T, H>::STATB) {
      if (NQ <= RecCfg<T, H>::NW)                  // one stationary 32-column block per wave: D <= H
        return launch_bwd_x<T, H, 1>(ntiles, steps, Z, UTpack, C, dH, dZ, dz_cts, dbias, sigm, WTpack, NQ, dX, DP, st);
      if (DP < (NQ - 1) * 32 + 4) return 1015;     // remainder mode stores 4 columns of the last block
      return launch_bwd_x<T, H, 2>(ntiles, steps, Z, UTpack, C, dH, dZ, dz_cts, dbias, sigm, WTpack, NQ, dX, DP, st);
    } else {
      return 1015;      // fused dX exists for the stationary-U^T build only (bf16, H = 128)
    }
  }
  return launch_bwd_x<T, H, 0>(ntiles, steps, Z, UTpack, C, dH, dZ, dz_cts, dbias, sigm, nullptr, 0, nullptr, 0, st);
}
template <typename T, int H> int launch_pack_wt(const float* W, int D, int NQ, void* out, hipStream_t st) {
  int n = NQ * 32 * 4 * H;
  hipLaunchKernelGGL((pack_wt_bwd_kernel<T, H>), dim3((n + 255) / 256), dim3(256), 0, st, W, D, NQ, (T*)out);
  return (int)hipGetLastError();
}

template <typename T, int H>
int launch_pack_w(const float* W, int D, int NKX, void* out, hipStream_t st) {
  using R = RecCfg<T, H>;
  int n = 4 * H * NKX * R::KC;
  hipLaunchKernelGGL((pack_w_fwd_kernel<T, H>), dim3((n + 255) / 256), dim3(256), 0, st, W, D, NKX, (T*)out);
  return (int)hipGetLastError();
}
template <typename T, int H, bool SIGM, bool WSTAT, bool WLDS>
int launch_fwd_fused_k(int ntiles, int steps, const void* X, int DP, int NKX, const void* Wpack, const float* bias,
                       void* Zst, const void* Upack, void* Hout, void* Cout, size_t smem, hipStream_t st) {
  using R = RecCfg<T, H>;
  static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_fwd_fused_kernel<T, H, SIGM, WSTAT, WLDS>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL((lstm_fwd_fused_kernel<T, H, SIGM, WSTAT, WLDS>), dim3(ntiles), dim3(R::NT), smem, st, (const T*)X,
                     DP, NKX, (const T*)Wpack, bias, (StashElem<T>*)Zst, (const T*)Upack, (T*)Hout, (T*)Cout, steps);
  return (int)hipGetLastError();
}
template <typename T, int H, bool SIGM, bool WSTAT>
int launch_fwd_fused_w(int ntiles, int steps, const void* X, int DP, int NKX, const void* Wpack, const float* bias,
                       void* Zst, const void* Upack, void* Hout, void* Cout, size_t smem, hipStream_t st) {
  using R = RecCfg<T, H>;
  // streamed-W path with the ring of 6: one ring block of W in LDS when it fits (note layer 0: 55 + 96 KB)
  if constexpr (R::STATF && !WSTAT) {
    const size_t extra = (size_t)R::NW * 4 * 6 * 1024;
    if (NKX % 8 && NKX % 6 == 0 && NKX > 6 && smem + extra <= 160 * 1024)
      return launch_fwd_fused_k<T, H, SIGM, WSTAT, true>(ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout,
                                                         smem + extra, st);
  }
  return launch_fwd_fused_k<T, H, SIGM, WSTAT, false>(ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, smem,
                                                      st);
}
template <typename T, int H, bool SIGM>
int launch_fwd_fused_s(int ntiles, int steps, const void* X, int DP, int NKX, const void* Wpack, const float* bias,
                       void* Zst, const void* Upack, void* Hout, void* Cout, hipStream_t st) {
  using R = RecCfg<T, H>;
  const size_t smem = ((size_t)2 * 32 * R::LDH + (size_t)2 * 32 * (NKX * R::KC + R::EPL)) * sizeof(T);
  if (smem > 160 * 1024 || DP > FUSED_DPMAX || DP % R::EPL || NKX * R::KC < DP) return 1011;
  if constexpr (R::STATF) {
    // stationary-weight build: W in registers too when the input is at most H wide, else streamed 8 deep
    if (NKX == R::NKC && DP <= H)
      return launch_fwd_fused_w<T, H, SIGM, true>(ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, smem, st);
    if (NKX % 8 && NKX % 6) return 1014;
  } else {
    if (NKX % R::PD) return 1011;       // continuous [W ; U] fragment ring
  }
  return launch_fwd_fused_w<T, H, SIGM, false>(ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, smem, st);
}
template <typename T, int H>
int launch_fwd_fused(int ntiles, int steps, const void* X, int DP, int NKX, const void* Wpack, const float* bias,
                     void* Zst, const void* Upack, void* Hout, void* Cout, int sigm, hipStream_t st) {
  return sigm ? launch_fwd_fused_s<T, H, true>(ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, st)
              : launch_fwd_fused_s<T, H, false>(ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, st);
}

template <bool SIGM, int NKX, bool TAGGED>
int launch_fwd_cluster_k(int ntiles, int steps, const void* X, int DP, const void* Wpack, const float* bias,
                         void* Zst, const void* Upack, void* Hout, void* Cout, size_t smem, void* scratch,
                         hipStream_t st) {
  static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_fwd_cluster_kernel<SIGM, NKX, TAGGED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  // whole groups of 64 blocks = 8 XCDs x 8 members; wave slots beyond ntiles stay idle
  hipLaunchKernelGGL((lstm_fwd_cluster_kernel<SIGM, NKX, TAGGED>), dim3((ntiles + 63) / 64 * 64), dim3(512), smem, st, (const bf16_t*)X, DP,
                     (const bf16_t*)Wpack, bias, (uint8_t*)Zst, (const bf16_t*)Upack, (bf16_t*)Hout, (bf16_t*)Cout, steps,
                     (int*)scratch, ntiles);
  return (int)hipGetLastError();
}
template <bool SIGM, bool TAGGED>
int launch_fwd_cluster_s(int ntiles, int steps, const void* X, int DP, int NKX, const void* Wpack, const float* bias,
                         void* Zst, const void* Upack, void* Hout, void* Cout, size_t smem, void* scratch,
                         hipStream_t st) {
  switch (NKX) {   // the input widths the model has (dj_lstm_fused_nkx): 94 -> 8 chunks, 256 -> 16
    case 8: return launch_fwd_cluster_k<SIGM, 8, TAGGED>(ntiles, steps, X, DP, Wpack, bias, Zst, Upack, Hout, Cout, smem, scratch, st);
    case 16: return launch_fwd_cluster_k<SIGM, 16, TAGGED>(ntiles, steps, X, DP, Wpack, bias, Zst, Upack, Hout, Cout, smem, scratch, st);
  }
  return 1016;
}
// compute units of the current device (one 160 KiB workgroup each): the cluster kernel needs its whole grid resident
int cluster_cus() {
  static int cus[DJ_MAX_DEVICES] = {};
  const int dev = dj_current_device();
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cus[dev] = n;
  }
  return cus[dev];
}
// Counters and XCC ids of every cluster start at zero in every launch.  A KERNEL, not hipMemsetAsync: under hipGraph
// replay a memset node followed by the cluster kernel was observed to take effect AFTER the kernel's round-0 arrivals
// in some replay histories (ROCm 7.2; every wait of that launch then expires: DESIGN.md section 8, round 3) -- kernel ->
// kernel edges do not have that problem.
// The same launch writes the TEST HOOK word of the fault line (word 2), from the caller's DJ_KF_DEBUG_* bits, so that the
// fault handling can be exercised on hardware, deterministically: DJ_KF_DEBUG_CLUSTER_FAULT (bit 0 of the word): the
// launch fails its placement check (fallback in fit, errors in predict / generation); DJ_KF_DEBUG_CLUSTER_LATE (bit 1):
// the last member of every cluster never arrives in round 0, so every other wave's bound really runs out -- once
// (poison bit, sticky), which the launch duration shows; DJ_KF_DEBUG_CLUSTER_MUTE (bit 2): the last member of every
// cluster stops publishing its h slices at step 2 of a tagged sweep, so every wave's bound runs out on the fragments of
// step 2 -- all at once, and no poisoned wave waits again.  The word is (re)written in front of EVERY cluster launch:
// no host-side memory of which scratch is armed (until round 5 a process-static table of 16 scratch addresses mirrored
// the word and could disagree with it -- after dj_workspace_init had zeroed the line, after an armed engine had died,
// under graph capture).
__global__ void cl_reset_kernel(uint4* p, int* hook, int v) {
  p[blockIdx.x * 256 + threadIdx.x] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0 && threadIdx.x == 0) *hook = v;
}
int cluster_reset(void* scratch, uint32_t kf, hipStream_t st) {
  static_assert(CL_OFF_FAULT % (256 * 16) == 0, "reset grid");
  const int want = ((kf & DJ_KF_DEBUG_CLUSTER_FAULT) ? 1 : 0) | ((kf & DJ_KF_DEBUG_CLUSTER_LATE) ? 2 : 0) |
                   ((kf & DJ_KF_DEBUG_CLUSTER_MUTE) ? 4 : 0);
  hipLaunchKernelGGL(cl_reset_kernel, dim3(CL_OFF_FAULT / (256 * 16)), dim3(256), 0, st, (uint4*)scratch,
                     (int*)((char*)scratch + CL_OFF_FAULT) + CLF_HOOK, want);
  return (int)hipGetLastError();
}
int launch_fwd_cluster(int ntiles, int steps, const void* X, int DP, int NKX, const void* Wpack, const float* bias,
                       void* Zst, const void* Upack, void* Hout, void* Cout, int sigm, void* scratch, uint32_t kf,
                       hipStream_t st) {
  using R = RecCfg<bf16_t, 256>;
  if (ntiles < 1 || ntiles > 256 || NKX * R::KC > 256 || DP > 256 || DP % 8 || !scratch || ((uintptr_t)scratch & 127))
    return 1016;
  const size_t smem = (size_t)(4 * NKX * 64 + 4 * R::NKC * 64) * 16 + (size_t)8 * 4096;
  // counters and XCC ids of every cluster start at zero in every launch (cl_reset_kernel: a kernel node under graph capture)
  if (int rc = cluster_reset(scratch, kf, st)) return rc;
  // the h slices announce themselves by their tags ("TAGGED exchange" above) unless the caller asks for the counted protocol
  if (kf & DJ_KF_COUNTED_EXCHANGE)
    return sigm ? launch_fwd_cluster_s<true, false>(ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, smem, scratch, st)
                : launch_fwd_cluster_s<false, false>(ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, smem, scratch, st);
  return sigm ? launch_fwd_cluster_s<true, true>(ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, smem, scratch, st)
              : launch_fwd_cluster_s<false, true>(ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, smem, scratch, st);
}

int launch_fwd_cluster_pair(int ntiles, int steps, const ClPairArgs& a, int sigm, void* scratch, uint32_t kf,
                            hipStream_t st) {
  using R = RecCfg<bf16_t, 256>;
  if (ntiles < 1 || ntiles > 64 || a.DP0 > 128 || a.DP0 % 8 || !scratch || ((uintptr_t)scratch & 127) || a.sp_D < 256)
    return 1016;
  if (cluster_cus() < 128) return 1017;                 // both halves of the grid must be co-resident
  const size_t smem = (size_t)(4 * 16 * 64 + 4 * R::NKC * 64) * 16 + (size_t)8 * 4096;
  static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
  if (!attr_done) {
    const void* fns[6] = {(const void*)lstm_fwd_cluster_pair_kernel<false, false>, (const void*)lstm_fwd_cluster_pair_kernel<true, false>,
                          (const void*)lstm_fwd_cluster_pair_kernel<false, true>, (const void*)lstm_fwd_cluster_pair_kernel<true, true>,
                          (const void*)lstm_fwd_cluster_pair_kernel<false, true, true>,
                          (const void*)lstm_fwd_cluster_pair_kernel<true, true, true>};
    for (const void* fn : fns) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
    }
    attr_done = true;
  }
  if (int rc = cluster_reset(scratch, kf, st)) return rc;
  // at most one tile per cluster (8 clusters per layer): the cooperative body, four waves per tile
  const bool coop_off = (kf & DJ_KF_NO_CLUSTER_COOP) != 0;
  const bool coop = ntiles <= 8 && !coop_off;
  const bool tagged = coop && !(kf & DJ_KF_COUNTED_EXCHANGE);     // the cooperative body's exchange ("TAGGED exchange")
#define DJ_PAIR_LAUNCH(S, C_, T_) \
  hipLaunchKernelGGL((lstm_fwd_cluster_pair_kernel<S, C_, T_>), dim3(128), dim3(512), smem, st, a, steps, (int*)scratch, ntiles)
  if (sigm) {
    if (tagged) DJ_PAIR_LAUNCH(true, true, true); else if (coop) DJ_PAIR_LAUNCH(true, true, false); else DJ_PAIR_LAUNCH(true, false, false);
  } else {
    if (tagged) DJ_PAIR_LAUNCH(false, true, true); else if (coop) DJ_PAIR_LAUNCH(false, true, false); else DJ_PAIR_LAUNCH(false, false, false);
  }
#undef DJ_PAIR_LAUNCH
  return (int)hipGetLastError();
}


}  // namespace

// fp32 H = 256 inference sweep of at most 8 tiles on clusters of 8 workgroups (lstm_fwd_cluster_f32_kernel).  Returns
// 1017 when the device cannot hold the grid (the caller then uses the per-tile kernel).
int dj_launch_lstm_fwd_cluster_f32(int ntiles, int steps, const void* Zx, const void* Upack, void* Hout, int sigm,
                                   void* scratch, uint32_t kf, hipStream_t st) {
  using R = RecCfg<float, 256>;
  if (ntiles < 1 || ntiles > 8 || !scratch || ((uintptr_t)scratch & 127)) return 1016;
  if (cluster_cus() < 64) return 1017;
  const size_t smem = (size_t)4 * R::NKC * 64 * 16 + (size_t)(32 * 32 + 4 * 64 * 16) * sizeof(float);
  static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_fwd_cluster_f32_kernel<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)lstm_fwd_cluster_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  if (int rc = cluster_reset(scratch, kf, st)) return rc;
  if (sigm)
    hipLaunchKernelGGL((lstm_fwd_cluster_f32_kernel<true>), dim3(64), dim3(512), smem, st, (const float*)Zx,
                       (const float*)Upack, (float*)Hout, steps, (int*)scratch, ntiles);
  else
    hipLaunchKernelGGL((lstm_fwd_cluster_f32_kernel<false>), dim3(64), dim3(512), smem, st, (const float*)Zx,
                       (const float*)Upack, (float*)Hout, steps, (int*)scratch, ntiles);
  return (int)hipGetLastError();
}

// Two stacked bf16 H = 256 inference layers (input widths <= 128 and 256) as ONE wavefront launch: the lower layer
// writes the upper layer's input X1 = bf16(h + sp1[b * steps + t]) itself (rows (b, n) of n_seq sequences per b).
// Returns 1017 when the device cannot hold both halves of the grid (the caller then runs the layers one by one).
int dj_launch_lstm_fwd_cluster_pair(int ntiles, int steps, const void* X0, int DP0, const void* W0pack, const float* b0,
                                    const void* U0pack, void* X1, const void* W1pack, const float* b1, const void* U1pack,
                                    void* H1, const float* sp1, int sp_D, int n_seq, int n_b, int sigm, void* scratch,
                                    uint32_t kf, hipStream_t st) {
  ClPairArgs a;
  a.X0 = (const bf16_t*)X0; a.DP0 = DP0; a.W0 = (const bf16_t*)W0pack; a.b0 = b0; a.U0 = (const bf16_t*)U0pack;
  a.X1 = (bf16_t*)X1; a.W1 = (const bf16_t*)W1pack; a.b1 = b1; a.U1 = (const bf16_t*)U1pack; a.H1 = (bf16_t*)H1;
  a.sp1 = sp1; a.sp_D = sp_D; a.n_seq = n_seq; a.n_b = n_b;
  return launch_fwd_cluster_pair(ntiles, steps, a, sigm, scratch, kf, st);
}

#define DJ_DISPATCH_TH(FN, ...)                                     \
  if (dtype == DJ_F32 && H == 256) return FN<float, 256>(__VA_ARGS__);  \
  if (dtype == DJ_F32 && H == 128) return FN<float, 128>(__VA_ARGS__);  \
  if (dtype == DJ_BF16 && H == 256) return FN<bf16_t, 256>(__VA_ARGS__); \
  if (dtype == DJ_BF16 && H == 128) return FN<bf16_t, 128>(__VA_ARGS__); \
  return 1010;

int dj_launch_lstm_pack(int dtype, int H, const float* U, void* fwd, void* bwd, hipStream_t st) {
  DJ_DISPATCH_TH(launch_pack, U, fwd, bwd, st)
}
int dj_launch_lstm_fwd(int dtype, int H, int ntiles, int steps, const void* Zx, void* Gst, const void* Upack, void* Hout,
                       void* Cout, int sigm, hipStream_t st) {
  if (ntiles <= 0 || steps <= 0) return 0;
  DJ_DISPATCH_TH(launch_fwd, ntiles, steps, Zx, Gst, Upack, Hout, Cout, sigm, st)
}
// bytes per row of the gate stash of a layer with H units (fragment-tiled; GateEnc above)
int64_t dj_lstm_stash_row_bytes(int dtype, int H) { return (int64_t)4 * H * (dtype == DJ_F32 ? 4 : 1); }
int64_t dj_lstm_cluster_scratch_bytes_impl() { return (int64_t)CL_BYTES; }
int dj_launch_lstm_bwd(int dtype, int H, int ntiles, int steps, const void* Z, const void* UTpack, const void* C,
                       const void* dH, void* dZ, int64_t dz_cts, float* dbias, int sigm, const void* WTpack, int D, void* dX,
                       int DP, uint32_t kf, hipStream_t st) {
  if (ntiles <= 0 || steps <= 0) return 0;
  const int NQ = (D + 31) / 32;
  if (WTpack && (!dX || D < 1 || DP < 8 || (DP % 8) || NQ * 32 < DP)) return 1013;
  DJ_DISPATCH_TH(launch_bwd, ntiles, steps, Z, UTpack, C, dH, dZ, dz_cts, dbias, sigm, WTpack, NQ, dX, DP, kf, st)
}
// does the BPTT kernel of this (dtype, H) offer the fused input gradient?
// 1: the whole dX (D <= H); 2: only the last 32-column block, for inputs whose width is 1..4 columns past a
// multiple of 256 (note layer 0: 259 = time-axis h + 3 chosen columns) -- the GEMM then covers D - D%32 columns
int dj_lstm_bwd_has_dx(int dtype, int H, int D) {
  if (!(dtype == DJ_BF16 && H == 128 && RecCfg<bf16_t, 128>::STATB)) return 0;
  if (D <= H) return 1;
  return (D % 256 >= 1 && D % 256 <= 4) ? 2 : 0;
}
// W^T fragment stream of the fused dX product: ceil(D / 32) * 32 * 4H operand elements
int dj_launch_lstm_pack_wt(int dtype, int H, const float* W, int D, void* out, hipStream_t st) {
  const int NQ = (D + 31) / 32;
  DJ_DISPATCH_TH(launch_pack_wt, W, D, NQ, out, st)
}

// k-chunks of the fused input projection for a layer input of width D: ceil(D / KC) rounded up to the ring depth
int dj_lstm_fused_nkx(int dtype, int H, int D) {
  const int kc = dtype == DJ_F32 ? 8 : 16;                                   // RecCfg::KC
  int pd = dtype == DJ_F32 ? 2 : 4;                                          // RecCfg::PD
  int n = (D + kc - 1) / kc;
  if (dtype != DJ_F32 && H == 128 && RecCfg<bf16_t, 128>::STATF)    // stationary build: W in registers (8 chunks = NKC) or
    return n <= 8 ? 8 : std::min((n + 7) / 8 * 8, (n + 5) / 6 * 6);  // streamed through a ring of 8 or 6 chunks
  return (n + pd - 1) / pd * pd;
}
int dj_launch_lstm_pack_w(int dtype, int H, const float* W, int D, int NKX, void* out, hipStream_t st) {
  DJ_DISPATCH_TH(launch_pack_w, W, D, NKX, out, st)
}
void* dj_lstm_cluster_fault_words(void* scratch) { return (char*)scratch + CL_OFF_FAULT; }
namespace {
// census taken: counts and the description of the first expired wait start again; the stall census (words 4, 5) stays
__global__ void cl_fault_clear_kernel(int* f) {
  const int i = threadIdx.x;
  if (i == CLF_EXPIRED || i == CLF_MISPLACED || (i >= CLF_DIAG && i < CLF_WORDS)) f[i] = 0;
}
}  // namespace
// the whole fault line (CLF_WORDS ints, layout at "bounded exchange waits") as it stands once `st` has drained
int dj_lstm_cluster_fault_line(void* scratch, int32_t* words_host, hipStream_t st) {
  if (!scratch || !words_host) return 1016;
  if (hipError_t e = hipMemcpyAsync(words_host, (char*)scratch + CL_OFF_FAULT, CLF_WORDS * sizeof(int32_t),
                                    hipMemcpyDeviceToHost, st); e != hipSuccess) return (int)e;
  return (int)hipStreamSynchronize(st);
}
// One blocking round trip: the fault line as it stands once `st` has drained (into words_host), and -- when it holds
// counts or a description -- a clear kernel queued behind it on the same stream (nothing waits for it: later work on
// `st` is ordered behind it anyway).  Returns the number of events, -1 on a HIP error.
int dj_lstm_cluster_faults_take(void* scratch, int32_t* words_host, hipStream_t st) {
  if (!scratch || !words_host) return -1;
  if (dj_lstm_cluster_fault_line(scratch, words_host, st)) return -1;
  const int32_t* w = words_host;
  if (w[CLF_EXPIRED] || w[CLF_MISPLACED] || w[CLF_DIAG]) {
    hipLaunchKernelGGL(cl_fault_clear_kernel, dim3(1), dim3(64), 0, st, (int*)((char*)scratch + CL_OFF_FAULT));
    if (hipGetLastError() != hipSuccess) return -1;
  }
  return w[CLF_EXPIRED] + w[CLF_MISPLACED];
}
int dj_lstm_cluster_faults_impl(void* scratch, hipStream_t st) {
  if (!scratch) return 0;
  int32_t w[CLF_WORDS];
  return dj_lstm_cluster_faults_take(scratch, w, st);
}
int dj_launch_lstm_fwd_fused(int dtype, int H, int ntiles, int steps, const void* X, int DP, int NKX,
                             const void* Wpack, const float* bias, void* Zst, const void* Upack, void* Hout,
                             void* Cout, int sigm, void* cluster_scratch, uint32_t kf, hipStream_t st) {
  if (ntiles <= 0 || steps <= 0) return 0;
  // weight-stationary cluster kernel (bf16, H = 256), in launches whose whole grid is co-resident (at most one
  // workgroup per compute unit, groups of 64 blocks = 8 clusters of 8 members); any tile count: a sweep of few tiles
  // (generation) spreads them over the clusters and still keeps every weight byte in LDS
  if (cluster_scratch && dtype == DJ_BF16 && H == 256 && (NKX == 8 || NKX == 16) && DP <= 256) {
    const int cap = cluster_cus() < 256 ? cluster_cus() / 64 * 64 : 256;
    if (cap >= 64) {
      while (ntiles > 0) {
        const int n = ntiles < cap ? ntiles : cap;
        const int rc = launch_fwd_cluster(n, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, sigm, cluster_scratch, kf, st);
        if (rc) return rc;
        const int64_t rows = (int64_t)n * steps * 32;      // all five buffers are tile-major
        X = (const bf16_t*)X + rows * DP;
        if (Zst) Zst = (uint8_t*)Zst + rows * 4 * H;      // 8-bit gate stash
        Hout = (bf16_t*)Hout + rows * H;
        if (Cout) Cout = (bf16_t*)Cout + rows * H;
        ntiles -= n;
      }
      return 0;
    }
  }
  DJ_DISPATCH_TH(launch_fwd_fused, ntiles, steps, X, DP, NKX, Wpack, bias, Zst, Upack, Hout, Cout, sigm, st)
}
