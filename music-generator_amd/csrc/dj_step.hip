// Generic-H LSTM path: one GEMM + one gate launch per recurrence step.
//
// The persistent recurrent kernels (dj_lstm.hip) keep a 32-sequence tile on one CU for the whole
// recurrence and are instantiated for H = 128 / 256 (the reference's sizes, constants.py:72-73).
// For wider layers (BASELINE "scaled model": 1024 units) the recurrent product of ONE step,
// [all sequences, H] x [H, 4H], is already a chip-filling GEMM, so the recurrence is driven from
// the host as `steps` x (GEMM + elementwise gate kernel) on the caller's stream:
//
//   forward  (Keras LSTM cell):  r_t = h_{t-1} U ; z_t = (x_t W + b) + r_t ;
//            i,f,o = recurrent_act(z) ; g = tanh(z_c) ; c_t = f c_{t-1} + i g ; h_t = o tanh(c_t)
//   backward (BPTT): dh_t = dH_t + dz_{t+1} U^T ; dz_t from (z_t, c_t, c_{t-1}, dc carry)
//
// Buffers are row-major in the sequence-tiled row order of dj_common.h (dj_row); "all sequences
// at step t" is addressed with the row-block stride of dj_launch_gemm_nt_rbs.  The cell state and
// its gradient are carried between launches in fp32 (as the persistent kernels carry them in
// registers); z, c, h stashes have the operand dtype.  bf16 FORWARD (round 4, dj_launch_lstm_step_fwd_fused): ONE launch
// per step and nothing else -- z_t = [x_t | h_{t-1}] [W ; U] + b as one product over K = DP + H whose epilogue IS the
// cell (dj_kernels.h CellEpi, dj_gemm.hip cell_fwd_block): no x W pass over the layer, no z round trip, no gate launch.
// Its training stash is the four ACTIVATED gates as 8-bit codes (dj_common.h dj_gate_code01 / dj_gate_code_g, the persistent
// kernels' quantiser) and c_t as bf16, both in the accumulators' fragment layout (coalesced 16- / 8-byte stores); BPTT's
// gate kernel for that stash is step_bwd8c_kernel.
// DJ_KF_NO_STEP_EPILOGUE keeps the round-3 form: x W as one GEMM, then per step a GEMM that ACCUMULATES h_{t-1} U into
// the stash rows of z_t in its epilogue (c_mode 3) and a 16-byte gate kernel.  bf16 BPTT: one GEMM (fp32 r out) and one
// gate kernel per step (the cell as the BPTT GEMM's epilogue was built and measured 2x slower).
#include "dj_common.h"
#include "dj_kernels.h"

namespace {

template <typename T> __device__ __forceinline__ void ld4(const T* p, float* x);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float* x) {
  const float4 v = *(const float4*)p;
  x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float* x) {
  const uint2 v = *(const uint2*)p;
  x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xFFFF0000u);
  x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xFFFF0000u);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float* x);
template <> __device__ __forceinline__ void st4<float>(float* p, const float* x) {
  *(float4*)p = make_float4(x[0], x[1], x[2], x[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float* x) {
  bf16_t t[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) t[e] = dj_from_f32<bf16_t>(x[e]);
  *(uint2*)p = *(const uint2*)t;
}

__device__ __forceinline__ int64_t step_row(int v, int steps, int t) {
  return (((int64_t)(v >> 5) * steps + t) << 5) + (v & 31);
}

// one thread = 4 consecutive units of one sequence
template <typename T, bool SIGM>
__global__ __launch_bounds__(256) void step_fwd_kernel(T* __restrict__ Z, const float* __restrict__ R,
                                                       float* __restrict__ cst, T* __restrict__ Hs,
                                                       T* __restrict__ Cs, int H, int nrows, int steps, int t) {
  // 32-bit index arithmetic (the launchers refuse nrows * H / 4 >= 2^31): a 64-bit division and modulo per thread cost
  // more instructions than the cell itself
  const uint32_t q = (uint32_t)H >> 2;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (uint32_t)nrows * q) return;
  const uint32_t vq = idx / q;
  const int v = (int)vq, u = (int)(idx - vq * q) * 4;
  const int64_t pr = step_row(v, steps, t);
  float z[4][4], c[4], hn[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) ld4<T>(Z + pr * 4 * H + g * H + u, z[g]);
  if (t > 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float r[4];
      ld4<float>(R + (int64_t)v * 4 * H + g * H + u, r);
#pragma unroll
      for (int e = 0; e < 4; ++e) z[g][e] += r[e];
    }
    ld4<float>(cst + (int64_t)v * H + u, c);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = 0.f;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float ig = dj_ract<SIGM>(z[0][e]), fg = dj_ract<SIGM>(z[1][e]), gg = dj_tanh(z[2][e]),
                og = dj_ract<SIGM>(z[3][e]);
    c[e] = fg * c[e] + ig * gg;
    hn[e] = og * dj_tanh(c[e]);
  }
  st4<float>(cst + (int64_t)v * H + u, c);
  st4<T>(Hs + pr * H + u, hn);
  if (Cs) {
    st4<T>(Cs + pr * H + u, c);
#pragma unroll
    for (int g = 0; g < 4; ++g) st4<T>(Z + pr * 4 * H + g * H + u, z[g]);   // pre-activation stash for BPTT
  }
}

template <typename T, bool SIGM>
__global__ __launch_bounds__(256) void step_bwd_kernel(const T* __restrict__ Z, const T* __restrict__ Cs,
                                                       const T* __restrict__ dH, const float* __restrict__ Rb,
                                                       float* __restrict__ dcs, T* __restrict__ dZ, int H, int nrows,
                                                       int steps, int t) {
  // 32-bit index arithmetic (the launchers refuse nrows * H / 4 >= 2^31): a 64-bit division and modulo per thread cost
  // more instructions than the cell itself
  const uint32_t q = (uint32_t)H >> 2;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (uint32_t)nrows * q) return;
  const uint32_t vq = idx / q;
  const int v = (int)vq, u = (int)(idx - vq * q) * 4;
  const int64_t pr = step_row(v, steps, t);
  float z[4][4], ct[4], cp[4], dh[4], dcc[4], dz[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) ld4<T>(Z + pr * 4 * H + g * H + u, z[g]);
  ld4<T>(Cs + pr * H + u, ct);
  ld4<T>(dH + pr * H + u, dh);
  if (t > 0) {
    ld4<T>(Cs + (pr - 32) * H + u, cp);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) cp[e] = 0.f;
  }
  if (t < steps - 1) {
    float r[4];
    ld4<float>(Rb + (int64_t)v * H + u, r);
    ld4<float>(dcs + (int64_t)v * H + u, dcc);
#pragma unroll
    for (int e = 0; e < 4; ++e) dh[e] += r[e];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) dcc[e] = 0.f;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float zi = z[0][e], zf = z[1][e], zg = z[2][e], zo = z[3][e];
    const float ig = dj_ract<SIGM>(zi), fg = dj_ract<SIGM>(zf), gg = dj_tanh(zg), og = dj_ract<SIGM>(zo);
    const float tc = dj_tanh(ct[e]);
    const float dc = dcc[e] + dh[e] * og * (1.f - tc * tc);
    dz[3][e] = dh[e] * tc * dj_ract_grad<SIGM>(zo, og);
    dz[0][e] = dc * gg * dj_ract_grad<SIGM>(zi, ig);
    dz[1][e] = dc * cp[e] * dj_ract_grad<SIGM>(zf, fg);
    dz[2][e] = dc * ig * (1.f - gg * gg);
    dcc[e] = dc * fg;
  }
  st4<float>(dcs + (int64_t)v * H + u, dcc);
#pragma unroll
  for (int g = 0; g < 4; ++g) st4<T>(dZ + pr * 4 * H + g * H + u, dz[g]);
}

// ---- bf16 forms: one thread = 8 consecutive units of one sequence (16-byte accesses; the 4-unit kernels above moved
// their bf16 operands 8 bytes at a time and ran at 2.5 TB/s in BPTT).
__device__ __forceinline__ void ld8(const bf16_t* p, float* x) {
  const uint4 v = *(const uint4*)p;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[2 * i] = __uint_as_float(w[i] << 16);
    x[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
  }
}
__device__ __forceinline__ void st8(bf16_t* p, const float* x) {
  bf16_t t[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = dj_from_f32<bf16_t>(x[e]);
  *(uint4*)p = *(const uint4*)t;
}
__device__ __forceinline__ void ld8f(const float* p, float* x) {
  ld4<float>(p, x);
  ld4<float>(p + 4, x + 4);
}
__device__ __forceinline__ void st8f(float* p, const float* x) {
  st4<float>(p, x);
  st4<float>(p + 4, x + 4);
}

// Forward cell of step t.  Z holds the FINAL pre-activations z_t = (x_t W + b) + h_{t-1} U: the recurrent product was
// accumulated into the stash in place by the GEMM's epilogue (dj_launch_gemm_nt c_mode 3), so there is no fp32 r round
// trip (2 x 16 KiB per row at H = 1024) and Z is not written again -- z is rounded to bf16 ONCE, and the value BPTT reads
// back is the value the forward activations saw.
template <bool SIGM>
__global__ __launch_bounds__(256) void step_fwd8_kernel(const bf16_t* __restrict__ Z, float* __restrict__ cst,
                                                        bf16_t* __restrict__ Hs, bf16_t* __restrict__ Cs, int H, int nrows,
                                                        int steps, int t) {
  const uint32_t q = (uint32_t)H >> 3;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (uint32_t)nrows * q) return;
  const uint32_t vq = idx / q;
  const int v = (int)vq, u = (int)(idx - vq * q) * 8;
  const int64_t pr = step_row(v, steps, t);
  float z[4][8], c[8], hn[8];
#pragma unroll
  for (int g = 0; g < 4; ++g) ld8(Z + pr * 4 * H + g * H + u, z[g]);
  if (t > 0) {
    ld8f(cst + (int64_t)v * H + u, c);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float ig = dj_ract<SIGM>(z[0][e]), fg = dj_ract<SIGM>(z[1][e]), gg = dj_tanh(z[2][e]),
                og = dj_ract<SIGM>(z[3][e]);
    c[e] = fg * c[e] + ig * gg;
    hn[e] = og * dj_tanh(c[e]);
  }
  st8f(cst + (int64_t)v * H + u, c);
  st8(Hs + pr * H + u, hn);
  if (Cs) st8(Cs + pr * H + u, c);
}

// BPTT cell of step t.  One workgroup = 32 sequences x 64 units (thread = 8 units of one sequence: a row of the block is
// one 128-byte line per gate), so the bias gradient can be summed over the block's rows through LDS and added to a
// per-row-block partial buffer dbpart [nrows / 32][4H] -- owned column by column by exactly one thread per step, no
// atomics -- instead of a second pass over all of dZ at the end of the sweep (17 GB per layer pass at the scaled shape).
template <bool SIGM>
__global__ __launch_bounds__(256) void step_bwd8_kernel(const bf16_t* __restrict__ Z, const bf16_t* __restrict__ Cs,
                                                        const bf16_t* __restrict__ dH, const float* __restrict__ Rb,
                                                        float* __restrict__ dcs, bf16_t* __restrict__ dZ,
                                                        float* __restrict__ dbpart, int H, int nrows, int steps, int t,
                                                        int64_t dz_cts) {
  __shared__ float red[32][4 * 64 + 1];
  const int tid = threadIdx.x, r = tid >> 3, cg = tid & 7;
  const int v = blockIdx.y * 32 + r, u = blockIdx.x * 64 + cg * 8;
  const bool live = v < nrows && u < H;
  float dzi[8], dzf[8], dzg[8], dzo[8];
  if (live) {
    const int64_t pr = step_row(v, steps, t);
    float z[4][8], ct[8], cp[8], dh[8], dcc[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) ld8(Z + pr * 4 * H + g * H + u, z[g]);
    ld8(Cs + pr * H + u, ct);
    ld8(dH + pr * H + u, dh);
    if (t > 0) {
      ld8(Cs + (pr - 32) * H + u, cp);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) cp[e] = 0.f;
    }
    if (t < steps - 1) {
      float rr[8];
      ld8f(Rb + (int64_t)v * H + u, rr);
      ld8f(dcs + (int64_t)v * H + u, dcc);
#pragma unroll
      for (int e = 0; e < 8; ++e) dh[e] += rr[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dcc[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float zi = z[0][e], zf = z[1][e], zg = z[2][e], zo = z[3][e];
      const float ig = dj_ract<SIGM>(zi), fg = dj_ract<SIGM>(zf), gg = dj_tanh(zg), og = dj_ract<SIGM>(zo);
      const float tc = dj_tanh(ct[e]);
      const float dc = dcc[e] + dh[e] * og * (1.f - tc * tc);
      dzo[e] = dh[e] * tc * dj_ract_grad<SIGM>(zo, og);
      dzi[e] = dc * gg * dj_ract_grad<SIGM>(zi, ig);
      dzf[e] = dc * cp[e] * dj_ract_grad<SIGM>(zf, fg);
      dzg[e] = dc * ig * (1.f - gg * gg);
      dcc[e] = dc * fg;
    }
    st8f(dcs + (int64_t)v * H + u, dcc);
    if (dz_cts) {      // column-tile-major dZ [4H/256][rows][256]: what the weight-gradient GEMM streams per stage
      st8(dZ + (int64_t)(u >> 8) * dz_cts + pr * 256 + (u & 255), dzi);
      st8(dZ + (int64_t)((H + u) >> 8) * dz_cts + pr * 256 + ((H + u) & 255), dzf);
      st8(dZ + (int64_t)((2 * H + u) >> 8) * dz_cts + pr * 256 + ((2 * H + u) & 255), dzg);
      st8(dZ + (int64_t)((3 * H + u) >> 8) * dz_cts + pr * 256 + ((3 * H + u) & 255), dzo);
    } else {
      st8(dZ + pr * 4 * H + u, dzi);
      st8(dZ + pr * 4 * H + H + u, dzf);
      st8(dZ + pr * 4 * H + 2 * H + u, dzg);
      st8(dZ + pr * 4 * H + 3 * H + u, dzo);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) dzi[e] = dzf[e] = dzg[e] = dzo[e] = 0.f;
  }
  if (!dbpart) return;                       // uniform
  // column sums over the block's 32 rows (the values as stored: rounded to bf16 like the dZ a column-sum pass would read)
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[r][0 * 64 + cg * 8 + e] = dj_to_f32(dj_from_f32<bf16_t>(dzi[e]));
    red[r][1 * 64 + cg * 8 + e] = dj_to_f32(dj_from_f32<bf16_t>(dzf[e]));
    red[r][2 * 64 + cg * 8 + e] = dj_to_f32(dj_from_f32<bf16_t>(dzg[e]));
    red[r][3 * 64 + cg * 8 + e] = dj_to_f32(dj_from_f32<bf16_t>(dzo[e]));
  }
  __syncthreads();
  const int g = tid >> 6, uu = tid & 63;     // this thread's column of the block: gate g, unit blockIdx.x * 64 + uu
  if (blockIdx.x * 64 + uu < H) {
    float sum = 0.f;
#pragma unroll 8
    for (int rr = 0; rr < 32; ++rr) sum += red[rr][g * 64 + uu];
    float* dst = dbpart + (int64_t)blockIdx.y * 4 * H + g * H + blockIdx.x * 64 + uu;
    *dst += sum;
  }
}

// The same cell for a forward sweep that ran as GEMMs with the cell epilogue (dj_gemm.hip cell_fwd_block): the gate stash
// holds the ACTIVATED gates as 8-bit codes and the cell-state stash c_t as bf16, both in the accumulators' fragment layout
// [row block of a step][unit group of 8][64 lanes] (lane = 32 * (unit >> 2 & 1) + sequence & 31; 16 / 8 bytes per lane) --
// half the stash bytes of the z form, no transcendental per gate, and every access of a thread is 8 or 16 contiguous bytes
// of a 0.5 / 1 KiB block.  Same geometry, outputs and bias-gradient partials as step_bwd8_kernel.
template <bool SIGM>
__global__ __launch_bounds__(256) void step_bwd8c_kernel(const uint8_t* __restrict__ Zc, const bf16_t* __restrict__ Cf,
                                                         const bf16_t* __restrict__ dH, const float* __restrict__ Rb,
                                                         float* __restrict__ dcs, bf16_t* __restrict__ dZ,
                                                         float* __restrict__ dbpart, int H, int nrows, int steps, int t,
                                                         int64_t dz_cts) {
  __shared__ float red[32][4 * 64 + 1];
  const int tid = threadIdx.x, r = tid >> 3, cg = tid & 7;
  const int v = blockIdx.y * 32 + r, u = blockIdx.x * 64 + cg * 8;
  const bool live = v < nrows && u < H;
  float dzi[8], dzf[8], dzg[8], dzo[8];
  if (live) {
    const int64_t pr = step_row(v, steps, t);
    // fragment slots of this thread's units u .. u + 3 (lane l31) and u + 4 .. u + 7 (lane 32 + l31) at steps t and t - 1
    const int64_t fs = (((int64_t)(v >> 5) * steps + t) * (H >> 3) + (u >> 3)) * 64 + (v & 31);
    const uint4 q0 = *(const uint4*)(Zc + fs * 16), q1 = *(const uint4*)(Zc + (fs + 32) * 16);
    const uint2 c0 = *(const uint2*)(Cf + fs * 4), c1 = *(const uint2*)(Cf + (fs + 32) * 4);
    uint2 p0 = make_uint2(0u, 0u), p1 = make_uint2(0u, 0u);
    if (t > 0) {
      const int64_t fp = fs - (int64_t)(H >> 3) * 64;
      p0 = *(const uint2*)(Cf + fp * 4);
      p1 = *(const uint2*)(Cf + (fp + 32) * 4);
    }
    const uint32_t cw[4] = {c0.x, c0.y, c1.x, c1.y}, pw[4] = {p0.x, p0.y, p1.x, p1.y};
    float ct[8], cp[8], dh[8], dcc[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ct[2 * i] = __uint_as_float(cw[i] << 16); ct[2 * i + 1] = __uint_as_float(cw[i] & 0xFFFF0000u);
      cp[2 * i] = __uint_as_float(pw[i] << 16); cp[2 * i + 1] = __uint_as_float(pw[i] & 0xFFFF0000u);
    }
    ld8(dH + pr * H + u, dh);
    if (t < steps - 1) {
      float rr[8];
      ld8f(Rb + (int64_t)v * H + u, rr);
      ld8f(dcs + (int64_t)v * H + u, dcc);
#pragma unroll
      for (int e = 0; e < 8; ++e) dh[e] += rr[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dcc[e] = 0.f;
    }
    const uint32_t gw[2][4] = {{q0.x, q0.y, q0.z, q0.w}, {q1.x, q1.y, q1.z, q1.w}};     // [half][gate]: 4 codes each
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int hh = e >> 2, sh = 8 * (e & 3);
      float ig, fg, og, di, df, dO;
      dj_gate_dec01<SIGM>((float)((gw[hh][0] >> sh) & 0xFFu), ig, di);
      dj_gate_dec01<SIGM>((float)((gw[hh][1] >> sh) & 0xFFu), fg, df);
      const float gg = dj_gate_dec_g((float)((gw[hh][2] >> sh) & 0xFFu));
      dj_gate_dec01<SIGM>((float)((gw[hh][3] >> sh) & 0xFFu), og, dO);
      const float tc = dj_tanh(ct[e]);
      const float dc = dcc[e] + dh[e] * og * (1.f - tc * tc);
      dzo[e] = dh[e] * tc * dO;
      dzi[e] = dc * gg * di;
      dzf[e] = dc * cp[e] * df;
      dzg[e] = dc * ig * (1.f - gg * gg);
      dcc[e] = dc * fg;
    }
    st8f(dcs + (int64_t)v * H + u, dcc);
    if (dz_cts) {
      st8(dZ + (int64_t)(u >> 8) * dz_cts + pr * 256 + (u & 255), dzi);
      st8(dZ + (int64_t)((H + u) >> 8) * dz_cts + pr * 256 + ((H + u) & 255), dzf);
      st8(dZ + (int64_t)((2 * H + u) >> 8) * dz_cts + pr * 256 + ((2 * H + u) & 255), dzg);
      st8(dZ + (int64_t)((3 * H + u) >> 8) * dz_cts + pr * 256 + ((3 * H + u) & 255), dzo);
    } else {
      st8(dZ + pr * 4 * H + u, dzi);
      st8(dZ + pr * 4 * H + H + u, dzf);
      st8(dZ + pr * 4 * H + 2 * H + u, dzg);
      st8(dZ + pr * 4 * H + 3 * H + u, dzo);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) dzi[e] = dzf[e] = dzg[e] = dzo[e] = 0.f;
  }
  if (!dbpart) return;                       // uniform
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[r][0 * 64 + cg * 8 + e] = dj_to_f32(dj_from_f32<bf16_t>(dzi[e]));
    red[r][1 * 64 + cg * 8 + e] = dj_to_f32(dj_from_f32<bf16_t>(dzf[e]));
    red[r][2 * 64 + cg * 8 + e] = dj_to_f32(dj_from_f32<bf16_t>(dzg[e]));
    red[r][3 * 64 + cg * 8 + e] = dj_to_f32(dj_from_f32<bf16_t>(dzo[e]));
  }
  __syncthreads();
  const int g = tid >> 6, uu = tid & 63;
  if (blockIdx.x * 64 + uu < H) {
    float sum = 0.f;
#pragma unroll 8
    for (int rr = 0; rr < 32; ++rr) sum += red[rr][g * 64 + uu];
    float* dst = dbpart + (int64_t)blockIdx.y * 4 * H + g * H + blockIdx.x * 64 + uu;
    *dst += sum;
  }
}

// dbias[c] += sum over row blocks of dbpart[rb][c]
__global__ __launch_bounds__(256) void dbpart_fold_kernel(const float* __restrict__ dbpart, int nrb, int cols,
                                                          float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int rb = 0; rb < nrb; ++rb) s += dbpart[(int64_t)rb * cols + c];
  out[c] += s;
}

// out[c] += sum over rows of A[r, c]: 256 columns per workgroup column tile, rows split over gridDim.y
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ A, int64_t rows, int cols, int64_t rps,
                                                     float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rps, r1 = r0 + rps < rows ? r0 + rps : rows;
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) s += dj_to_f32(A[r * cols + c]);
  atomicAdd(out + c, s);
}

template <typename T>
int step_fwd_t(int H, int ntiles, int steps, void* Z, const void* Ut, void* Hs, void* Cs, float* R, float* cst,
               int sigm, hipStream_t st) {
  const int dtype = sizeof(T) == 4 ? DJ_F32 : DJ_BF16;
  const int nrows = ntiles * 32;
  const int64_t n = (int64_t)nrows * (H >> 2);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if constexpr (sizeof(T) == 2) {
    // bf16: z_t += h_{t-1} U in the GEMM's epilogue, straight into the stash rows of step t (row-block-strided C view)
    const int64_t n8 = (int64_t)nrows * (H >> 3);
    const dim3 grid8((unsigned)((n8 + 255) / 256));
    for (int t = 0; t < steps; ++t) {
      if (t > 0) {
        int rc = dj_launch_gemm_nt_rbs(dtype, nrows, 4 * H, H, (const T*)Hs + (int64_t)(t - 1) * 32 * H, H, steps, Ut, H,
                                       (T*)Z + (int64_t)t * 32 * 4 * H, 4 * H, steps, 3, nullptr, st);
        if (rc) return rc;
      }
      if (sigm)
        hipLaunchKernelGGL(step_fwd8_kernel<true>, grid8, block, 0, st, (const bf16_t*)Z, cst, (bf16_t*)Hs, (bf16_t*)Cs, H,
                           nrows, steps, t);
      else
        hipLaunchKernelGGL(step_fwd8_kernel<false>, grid8, block, 0, st, (const bf16_t*)Z, cst, (bf16_t*)Hs, (bf16_t*)Cs, H,
                           nrows, steps, t);
    }
    return (int)hipGetLastError();
  }
  for (int t = 0; t < steps; ++t) {
    if (t > 0) {
      // r_t = h_{t-1} U  (Bt = U^T [4H, H], k-contiguous)
      int rc = dj_launch_gemm_nt_rbs(dtype, nrows, 4 * H, H, (const T*)Hs + (int64_t)(t - 1) * 32 * H, H, steps, Ut, H,
                                     R, 4 * H, 1, 1, nullptr, st);
      if (rc) return rc;
    }
    if (sigm)
      hipLaunchKernelGGL((step_fwd_kernel<T, true>), grid, block, 0, st, (T*)Z, R, cst, (T*)Hs, (T*)Cs, H, nrows, steps, t);
    else
      hipLaunchKernelGGL((step_fwd_kernel<T, false>), grid, block, 0, st, (T*)Z, R, cst, (T*)Hs, (T*)Cs, H, nrows, steps, t);
  }
  return (int)hipGetLastError();
}

template <typename T>
int step_bwd_t(int H, int ntiles, int steps, const void* Z, const void* Uc, const void* Cs, const void* dH, void* dZ,
               int64_t dz_cts, float* dbias, float* Rb, float* dcs, int sigm, bool codes, hipStream_t st) {
  const int dtype = sizeof(T) == 4 ? DJ_F32 : DJ_BF16;
  const int nrows = ntiles * 32;
  const int64_t n = (int64_t)nrows * (H >> 2);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  // bf16: the bias gradient is summed inside the gate kernel into per-row-block partials (the part of the fp32 scratch
  // between the r / carry areas that BPTT does not use: (nrows / 32) * 4H floats behind Rb's nrows * H)
  float* dbpart = nullptr;
  if (sizeof(T) == 2 && dbias) {
    dbpart = Rb + (int64_t)nrows * H;
    hipError_t e = hipMemsetAsync(dbpart, 0, (size_t)(nrows / 32) * 4 * H * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
  }
  for (int t = steps - 1; t >= 0; --t) {
    if (t < steps - 1) {
      // recurrent part of dh_t = dz_{t+1} U^T  (Bt = U [H, 4H], k-contiguous)
      // (dz_cts: A column-tile-major -- the rows of step t + 1 start (t + 1) * 32 rows into every 256-column tile)
      int rc = dz_cts ? dj_launch_gemm_nt_ex(dtype, nrows, H, 4 * H, (const T*)dZ + (int64_t)(t + 1) * 32 * 256, 256, steps,
                                             dz_cts, Uc, 4 * H, Rb, H, 1, 1, nullptr, st)
                      : dj_launch_gemm_nt_rbs(dtype, nrows, H, 4 * H, (const T*)dZ + (int64_t)(t + 1) * 32 * 4 * H, 4 * H,
                                              steps, Uc, 4 * H, Rb, H, 1, 1, nullptr, st);
      if (rc) return rc;
    }
    if constexpr (sizeof(T) == 2) {
      const dim3 grid8((unsigned)((H + 63) / 64), (unsigned)((nrows + 31) / 32));
      if (codes) {      // stashes of the cell-epilogue forward: 8-bit gate codes + bf16 c, fragment layout
        if (sigm)
          hipLaunchKernelGGL(step_bwd8c_kernel<true>, grid8, block, 0, st, (const uint8_t*)Z, (const bf16_t*)Cs,
                             (const bf16_t*)dH, Rb, dcs, (bf16_t*)dZ, dbpart, H, nrows, steps, t, dz_cts);
        else
          hipLaunchKernelGGL(step_bwd8c_kernel<false>, grid8, block, 0, st, (const uint8_t*)Z, (const bf16_t*)Cs,
                             (const bf16_t*)dH, Rb, dcs, (bf16_t*)dZ, dbpart, H, nrows, steps, t, dz_cts);
      } else if (sigm)
        hipLaunchKernelGGL(step_bwd8_kernel<true>, grid8, block, 0, st, (const bf16_t*)Z, (const bf16_t*)Cs,
                           (const bf16_t*)dH, Rb, dcs, (bf16_t*)dZ, dbpart, H, nrows, steps, t, dz_cts);
      else
        hipLaunchKernelGGL(step_bwd8_kernel<false>, grid8, block, 0, st, (const bf16_t*)Z, (const bf16_t*)Cs,
                           (const bf16_t*)dH, Rb, dcs, (bf16_t*)dZ, dbpart, H, nrows, steps, t, dz_cts);
    } else if (sigm)
      hipLaunchKernelGGL((step_bwd_kernel<T, true>), grid, block, 0, st, (const T*)Z, (const T*)Cs, (const T*)dH, Rb, dcs,
                         (T*)dZ, H, nrows, steps, t);
    else
      hipLaunchKernelGGL((step_bwd_kernel<T, false>), grid, block, 0, st, (const T*)Z, (const T*)Cs, (const T*)dH, Rb,
                         dcs, (T*)dZ, H, nrows, steps, t);
  }
  if (dbpart) {
    hipLaunchKernelGGL(dbpart_fold_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, st, dbpart, nrows / 32, 4 * H, dbias);
  } else if (dbias) {
    const int64_t rows = (int64_t)nrows * steps;
    int splits = (int)((rows + 511) / 512);
    if (splits > 1024) splits = 1024;
    const int64_t rps = (rows + splits - 1) / splits;
    hipLaunchKernelGGL(colsum_kernel<T>, dim3((4 * H + 255) / 256, splits), dim3(256), 0, st, (const T*)dZ, rows, 4 * H,
                       rps, dbias);
  }
  return (int)hipGetLastError();
}

}  // namespace

// float scratch the step path needs for `ntiles` sequence tiles of width H: r/rb [rows, 4H] + carry [rows, H]
int64_t dj_lstm_step_scratch_floats(int H, int64_t ntiles) { return ntiles * 32 * 5 * (int64_t)H; }

int dj_launch_lstm_step_fwd(int dtype, int H, int ntiles, int steps, void* Z, const void* Ut, void* Hs, void* Cs,
                            float* scratch, int sigm, hipStream_t st) {
  if (ntiles <= 0 || steps <= 0) return 0;
  if (H < 32 || (H % 32)) return 1012;
  if ((int64_t)ntiles * 32 * (H >> 2) >= ((int64_t)1 << 31)) return 1014;
  float* R = scratch;
  float* cst = scratch + (int64_t)ntiles * 32 * 4 * H;
  return dtype == DJ_F32 ? step_fwd_t<float>(H, ntiles, steps, Z, Ut, Hs, Cs, R, cst, sigm, st)
                         : step_fwd_t<bf16_t>(H, ntiles, steps, Z, Ut, Hs, Cs, R, cst, sigm, st);
}

int dj_step_k1p(int DP) { return (DP + 63) / 64 * 64; }

int dj_launch_lstm_step_fwd_fused(int H, int ntiles, int steps, const void* X, int DP, int D, const void* WU,
                                  const float* bias, void* Z, void* Hs, void* Cs, float* scratch, int sigm, hipStream_t st) {
  if (ntiles <= 0 || steps <= 0) return 0;
  if (H < 32 || (H % 32) || D > DP || (DP % 8)) return 1012;
  const int nrows = ntiles * 32, K1p = dj_step_k1p(DP);
  CellEpi ce{};
  ce.H = H; ce.steps = steps; ce.sigm = sigm; ce.K1 = D; ce.K1p = K1p; ce.lda2 = H;
  ce.carry = scratch + (int64_t)nrows * 4 * H;               // where the gate-launch form keeps its carry
  for (int t = 0; t < steps; ++t) {
    ce.first = t == 0;
    ce.A2 = t > 0 ? (const bf16_t*)Hs + (int64_t)(t - 1) * 32 * H : nullptr;
    ce.Z = Z ? (uint8_t*)Z + (int64_t)t * 32 * 4 * H : (uint8_t*)nullptr;      // gate codes: one byte per gate value
    ce.Hs = (bf16_t*)Hs + (int64_t)t * 32 * H;
    ce.Cs = Cs ? (bf16_t*)Cs + (int64_t)t * 32 * H : (bf16_t*)nullptr;
    const int rc = dj_launch_gemm_nt_cell(nrows, (const bf16_t*)X + (int64_t)t * 32 * DP, DP, steps, WU, K1p + H, ce, bias, st);
    if (rc) return rc;
  }
  return 0;
}

int dj_launch_lstm_step_bwd(int dtype, int H, int ntiles, int steps, const void* Z, const void* Uc, const void* Cs,
                            const void* dH, void* dZ, int64_t dz_cts, float* dbias, float* scratch, int sigm, int codes,
                            hipStream_t st) {
  if (ntiles <= 0 || steps <= 0) return 0;
  if (H < 32 || (H % 32)) return 1012;
  if ((int64_t)ntiles * 32 * (H >> 2) >= ((int64_t)1 << 31)) return 1014;
  float* Rb = scratch;
  float* dcs = scratch + (int64_t)ntiles * 32 * 4 * H;
  if (dz_cts && (dtype == DJ_F32 || (4 * H) % 256 || dz_cts < (int64_t)ntiles * 32 * steps * 256)) return 1015;
  if (codes && dtype == DJ_F32) return 1015;
  return dtype == DJ_F32 ? step_bwd_t<float>(H, ntiles, steps, Z, Uc, Cs, dH, dZ, 0, dbias, Rb, dcs, sigm, false, st)
                         : step_bwd_t<bf16_t>(H, ntiles, steps, Z, Uc, Cs, dH, dZ, dz_cts, dbias, Rb, dcs, sigm, codes != 0, st);
}
