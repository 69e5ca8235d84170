// Fused autoregressive note sampler: the inner loop of the reference's generate()
// (generate.py:112-118: for n in range(NUM_NOTES): note_model.predict(...); choose(...)).
//
// The reference re-runs the whole note-axis model for every note (O(N^2) LSTM steps per time
// step).  The note axis is causal along n (model.py:101 shifts the chosen notes by one), so
// carrying the LSTM state from note to note is exactly equivalent (SURVEY.md a-G (2)): one
// workgroup walks n = 0..N-1 once, for all G pieces, keeping h/c in LDS.  M = G (3) rows is
// far too small for MFMA: every gate column is one thread doing G dot products against the
// fp32 master weights streamed from L2 (coalesced across columns), with LDS-broadcast h.
// Bernoulli decisions are made on the device from host-drawn uniforms consumed in the
// reference's order (note-major, piece-minor; the replay draw only after a successful play
// draw, generate.py:52-58); the number consumed is returned so the host can advance NumPy's
// MT19937 stream by exactly that many draws.
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <type_traits>
#include "dj_kernels.h"

namespace {

constexpr int GEN_MAXG = 8;
// f(integral_constant<int, I>) for I = I0 .. N-1, every call inlined (compile-time loop index)
template <int I, int N, typename Fn> __device__ __forceinline__ void dj_gen_static_for(Fn&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    dj_gen_static_for<I + 1, N>(f);
  }
}
// A Bernoulli decision u <= p is reproduced by ANY implementation whose p agrees with this one's to better than
// |u - p|.  Draws closer than this band are counted (dj_gen_state.near_ties / draws_used[1]): a run with a count of
// zero is certified bit-identical to every model within 1e-5 of these probabilities (the fp32 oracle is within
// ~1e-6); a non-zero count names how many decisions depend on the last digits.
constexpr double DJ_GEN_TIE_BAND = 1e-5;
struct DjGenState;

struct GenArgs {
  int G, N, Hn, Ht, Ln, S, SU, T;
  const float* P;                 // flat parameters
  int64_t p_style_W, p_style_b, p_nd_W, p_nd_b, p_vd_W, p_vd_b;
  int64_t dW[4], db[4], W[4], U[4], b[4];   // per note layer: style Dense kernel/bias, LSTM kernel/recurrent/bias
  int D0;                         // Ht + 3
  const float* style_last;        // style of the last window step of piece g at style_last + g*style_stride
  int64_t style_stride;
  float* svec;                    // scratch: style [G,SU], sp_l [Ln][G][D0max]
  float* zx0;                     // scratch: [G, N, 4Hn]
  const double* uniforms;         // [2*N*G]  (or the whole pool when `state` is set)
  const float* temperature;       // [G]
  float* next_notes;              // [G, N, 3]
  int* draws_used;
  // device-resident generation (dj_generate_run): overrides the three fields above
  struct DjGenState* state;
  float* results;                 // [steps_cap, G, N, 3]
};

// Per-run state kept in HBM so that a generated time step needs no host round trip and the
// kernel sequence can be captured once and replayed as a hipGraph.
struct DjGenState {
  int step;                       // time steps generated so far
  int draw_off;                   // uniforms consumed so far
  int near_ties;                  // draws so far with |u - p| < DJ_GEN_TIE_BAND (precision-dependent decisions)
  int first_near_step;            // time step of the first of them, -1 = none
  double temperature[GEN_MAXG];   // MusicGeneration.temperature (float64 like the reference)
  double default_temp[GEN_MAXG];
  int silent[GEN_MAXG];           // MusicGeneration.silent_time
};

// style = style_in W_s + b_s ; sp_l = tanh(style Wd_l + bd_l)    (model.py:141-142,110-113)
// One workgroup per note layer (blockIdx.x = l); every thread first fills its own column's weights into registers
// with independent loads (the k loops below were chains of dependent ~1 us round trips: 62 us for this launch).
__global__ __launch_bounds__(512) void gen_prep_kernel(GenArgs a) {
  __shared__ float st[GEN_MAXG * 64];
  __shared__ float sin_[GEN_MAXG * 64];
  const int l = blockIdx.x;
  for (int i = threadIdx.x; i < a.G * a.S; i += blockDim.x) sin_[i] = a.style_last[(i / a.S) * a.style_stride + i % a.S];
  __syncthreads();
  for (int i = threadIdx.x; i < a.G * a.SU; i += blockDim.x) {
    const int g = i / a.SU, k = i % a.SU;
    float s = a.P[a.p_style_b + k];
#pragma unroll 8
    for (int j = 0; j < a.S; ++j) s += sin_[g * a.S + j] * a.P[a.p_style_W + (int64_t)j * a.SU + k];
    st[i] = s;
    if (l == 0) a.svec[i] = s;
  }
  __syncthreads();
  const int D = l == 0 ? a.D0 : a.Hn;
  float* sp = a.svec + GEN_MAXG * 64 + (int64_t)l * GEN_MAXG * 512;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float wcol[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) wcol[k] = k < a.SU ? a.P[a.dW[l] + (int64_t)k * D + d] : 0.f;
    const float bd = a.P[a.db[l] + d];
    for (int g = 0; g < a.G; ++g) {
      float s = bd;
#pragma unroll
      for (int k = 0; k < 64; ++k) s += st[g * a.SU + (k < a.SU ? k : 0)] * wcol[k];
      sp[g * 512 + d] = dj_tanh(s);
    }
  }
}

// zx0[g,n,col] = b0[col] + sum_{k<Ht} (feat[g,n,k] + sp0[g,k]) W0[k,col] + sum_{c<3} sp0[g,Ht+c] W0[Ht+c,col]
// feat = last window step of the time axis, read straight from the TA-ordered h buffer.  One workgroup per
// (8 notes of one piece, 256 columns): the weight column is loaded once per 8 notes, 32 loads in flight.
constexpr int ZX_NB = 8;
template <typename T>
__global__ __launch_bounds__(256) void gen_zx0_kernel(GenArgs a, const T* __restrict__ Htime) {
  __shared__ float xf[ZX_NB][512];
  const int col = blockIdx.x * blockDim.x + threadIdx.x;     // 0 .. 4Hn-1
  const int nblk = (a.N + ZX_NB - 1) / ZX_NB;
  const int g = blockIdx.y / nblk, n0 = (blockIdx.y % nblk) * ZX_NB;
  const int nn = a.N - n0 < ZX_NB ? a.N - n0 : ZX_NB;
  const float* sp0 = a.svec + GEN_MAXG * 64 + g * 512;
  const int KP = (a.Ht + 3 + 31) / 32 * 32;                  // feat + style for k < Ht, style alone for the 3 chosen rows
  for (int i = threadIdx.x; i < ZX_NB * KP; i += blockDim.x) {
    const int j = i / KP, k = i % KP;
    float v = 0.f;
    if (j < nn && k < a.Ht + 3) {
      v = sp0[k];
      if (k < a.Ht) v += dj_to_f32(Htime[dj_row_ta(g, a.T - 1, n0 + j, a.T, a.N) * a.Ht + k]);
    }
    xf[j][k] = v;
  }
  __syncthreads();
  if (col >= 4 * a.Hn) return;
  const float* W0 = a.P + a.W[0];
  const int ldw = 4 * a.Hn;
  const float b0 = a.P[a.b[0] + col];
  float s[ZX_NB];
#pragma unroll
  for (int j = 0; j < ZX_NB; ++j) s[j] = b0;
#pragma unroll 1
  for (int k0 = 0; k0 < KP; k0 += 32) {
    float wv[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) wv[i] = k0 + i < a.Ht + 3 ? W0[(int64_t)(k0 + i) * ldw + col] : 0.f;
#pragma unroll
    for (int j = 0; j < ZX_NB; ++j)
#pragma unroll
      for (int i = 0; i < 32; i += 4) {
        const float4 xv = *(const float4*)&xf[j][k0 + i];
        s[j] += xv.x * wv[i] + xv.y * wv[i + 1] + xv.z * wv[i + 2] + xv.w * wv[i + 3];
      }
  }
  for (int j = 0; j < nn; ++j) a.zx0[((int64_t)g * a.N + n0 + j) * ldw + col] = s[j];
}

// z[g] += sum_k x[g][k] * w[k * ldw]  for one gate column: the weight column is streamed from L2 with
// 32 loads in flight per thread (the chain of 48 notes x Ln layers is pure latency: at 8 in flight the
// sampler took 4.3 ms per time step), x comes from LDS as 16-byte broadcasts.  K is a multiple of 32.
__device__ __forceinline__ void gen_dot(float (&z)[GEN_MAXG], const float* __restrict__ w, int ldw,
                                        const float* __restrict__ x, int K, int G) {
#pragma unroll 1
  for (int k0 = 0; k0 < K; k0 += 32) {
    float wv[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) wv[i] = w[(int64_t)(k0 + i) * ldw];
#pragma unroll
    for (int g = 0; g < GEN_MAXG; ++g)
      if (g < G) {
        float s = z[g];
#pragma unroll
        for (int i = 0; i < 32; i += 4) {
          const float4 xv = *(const float4*)(x + g * K + k0 + i);
          s += xv.x * wv[i] + xv.y * wv[i + 1] + xv.z * wv[i + 2] + xv.w * wv[i + 3];
        }
        z[g] = s;
      }
  }
}

// one workgroup, 4*Hn threads (one per gate column).  The chain of N notes x Ln layers is pure latency, so everything
// that does not depend on the chain is taken off it: the uniforms, temperatures, style terms and head weights of the
// time step are copied to LDS up front, per-column constants live in registers, the next note's x W + b is requested
// a note ahead, and the sampled notes are parked in LDS until the end (per note that removes ~10 dependent global
// round trips of ~1 us each: the two Bernoulli draws per piece alone were six of them).
template <bool SIGM, int MAXT>
__global__ __launch_bounds__(MAXT) void gen_sample_kernel(GenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Hn = a.Hn, G = a.G, C4 = 4 * Hn, Ln = a.Ln;
  float* hs = sm;                          // [Ln][G][Hn]
  float* cs = hs + Ln * G * Hn;            // [Ln][G][Hn]
  float* zb = cs + Ln * G * Hn;            // [G][4Hn]
  float* xs = zb + G * C4;                 // [G][Hn]   input of layers >= 1 (h below + style)
  float* spl = xs + G * Hn;                // [Ln][G][Hn]  style term added to the input of layer l >= 1
  float* hw = spl + Ln * G * Hn;           // [3][Hn] head weights (play, replay, volume) + [4] biases
  float* res = hw + 3 * Hn + 4;            // [G][N][3] sampled notes of this time step
  float* chosen = res + G * a.N * 3;       // [G][4]    previous note (play, replay, volume)
  float* logit = chosen + G * 4;           // [G][4]
  float* temps = logit + G * 4;            // [G]
  double* ul = (double*)(((uintptr_t)(temps + G) + 7) & ~(uintptr_t)7);   // [2 N G] uniforms of this time step
  __shared__ int kdraw, knear;
  const int tid = threadIdx.x, col = tid;
  const int draw0 = a.state ? a.state->draw_off : 0;
  for (int i = tid; i < 2 * Ln * G * Hn; i += blockDim.x) hs[i] = 0.f;       // hs and cs
  for (int i = tid; i < G * 4; i += blockDim.x) chosen[i] = 0.f;
  for (int i = tid; i < Ln * G * Hn; i += blockDim.x) {
    const int l = i / (G * Hn), r = i % (G * Hn), g = r / Hn, u = r % Hn;
    spl[i] = l ? a.svec[GEN_MAXG * 64 + (int64_t)l * GEN_MAXG * 512 + g * 512 + u] : 0.f;
  }
  for (int i = tid; i < Hn; i += blockDim.x) {
    hw[i] = a.P[a.p_nd_W + (int64_t)i * 2];
    hw[Hn + i] = a.P[a.p_nd_W + (int64_t)i * 2 + 1];
    hw[2 * Hn + i] = a.P[a.p_vd_W + i];
  }
  if (tid < 2) hw[3 * Hn + tid] = a.P[a.p_nd_b + tid];
  if (tid == 2) hw[3 * Hn + 2] = a.P[a.p_vd_b];
  for (int i = tid; i < 2 * a.N * G; i += blockDim.x) ul[i] = a.uniforms[draw0 + i];
  if (tid < G) temps[tid] = a.state ? (float)a.state->temperature[tid] : a.temperature[tid];
  if (tid == 0) {
    kdraw = 0;                             // index into ul
    knear = 0;
  }
  // per-column constants: the three `chosen` rows of the layer-0 kernel, the biases of the upper layers
  float wch[3], bl[4];
#pragma unroll
  for (int c = 0; c < 3; ++c) wch[c] = a.P[a.W[0] + (int64_t)(a.Ht + c) * C4 + col];
#pragma unroll
  for (int l = 1; l < 4; ++l) bl[l] = l < Ln ? a.P[a.b[l] + col] : 0.f;
  float zx[GEN_MAXG];
#pragma unroll
  for (int g = 0; g < GEN_MAXG; ++g) zx[g] = g < G ? a.zx0[((int64_t)g * a.N + 0) * C4 + col] : 0.f;
  __syncthreads();

  for (int n = 0; n < a.N; ++n) {
#pragma unroll 1
    for (int l = 0; l < Ln; ++l) {
      // ---- pre-activations of column `col` for every piece
      float z[GEN_MAXG];
      if (l == 0) {
#pragma unroll
        for (int g = 0; g < GEN_MAXG; ++g)
          if (g < G) z[g] = zx[g] + chosen[g * 4] * wch[0] + chosen[g * 4 + 1] * wch[1] + chosen[g * 4 + 2] * wch[2];
        if (n + 1 < a.N) {                 // next note's x W + b: in flight for a whole note
#pragma unroll
          for (int g = 0; g < GEN_MAXG; ++g)
            if (g < G) zx[g] = a.zx0[((int64_t)g * a.N + n + 1) * C4 + col];
        }
      } else {
        const float b = l == 1 ? bl[1] : (l == 2 ? bl[2] : bl[3]);
#pragma unroll
        for (int g = 0; g < GEN_MAXG; ++g) z[g] = b;
        gen_dot(z, a.P + a.W[l] + col, C4, xs, Hn, G);
      }
      gen_dot(z, a.P + a.U[l] + col, C4, hs + l * G * Hn, Hn, G);
#pragma unroll
      for (int g = 0; g < GEN_MAXG; ++g)
        if (g < G) zb[g * C4 + col] = z[g];
      __syncthreads();
      // ---- cell update (Keras gate order i,f,c,o)
      for (int i = tid; i < G * Hn; i += blockDim.x) {
        const int g = i / Hn, u = i % Hn;
        const float* zz = zb + g * C4;
        const float ig = dj_ract<SIGM>(zz[u]), fg = dj_ract<SIGM>(zz[Hn + u]), gg = dj_tanh(zz[2 * Hn + u]),
                    og = dj_ract<SIGM>(zz[3 * Hn + u]);
        const float cn = fg * cs[(l * G + g) * Hn + u] + ig * gg;
        cs[(l * G + g) * Hn + u] = cn;
        const float hv = og * dj_tanh(cn);
        hs[(l * G + g) * Hn + u] = hv;
        if (l + 1 < Ln) xs[g * Hn + u] = hv + spl[((l + 1) * G + g) * Hn + u];
      }
      __syncthreads();
    }
    // ---- heads: (play, replay) = sigmoid(h Wn + bn), volume = h Wv + bv   (model.py:94-95)
    {
      const int wv = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
      const float* ht = hs + (Ln - 1) * G * Hn;
      for (int job = wv; job < G * 3; job += nw) {
        const int g = job / 3, o = job % 3;
        float s = 0.f;
        for (int k = lane; k < Hn; k += 64) s += ht[g * Hn + k] * hw[o * Hn + k];
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_xor(s, sft);
        if (lane == 0) logit[g * 4 + o] = s + hw[3 * Hn + o];
      }
    }
    __syncthreads();
    // ---- sampling, reference draw order (generate.py:47-58,116-118)
    if (tid == 0) {
      int k = kdraw;
      for (int g = 0; g < G; ++g) {
        float pp = dj_sigmoid(logit[g * 4]), pr = dj_sigmoid(logit[g * 4 + 1]);
        const float vol = logit[g * 4 + 2];
        const float temp = temps[g];
        if (temp != 1.0f) {                       // apply_temperature, float32 like the reference (generate.py:81-91)
          float x0 = -logf(1.0f / pp - 1.0f), x1 = -logf(1.0f / pr - 1.0f);
          pp = 1.0f / (1.0f + expf(-x0 / temp));
          pr = 1.0f / (1.0f + expf(-x1 / temp));
        }
        float play = 0.f, rep = 0.f, v = 0.f;
        const double u0 = ul[k++];
        knear += fabs(u0 - (double)pp) < DJ_GEN_TIE_BAND;
        if (u0 <= (double)pp) {
          play = 1.f;
          v = vol;
          const double u1 = ul[k++];
          knear += fabs(u1 - (double)pr) < DJ_GEN_TIE_BAND;
          if (u1 <= (double)pr) rep = 1.f;
        }
        chosen[g * 4] = play;
        chosen[g * 4 + 1] = rep;
        chosen[g * 4 + 2] = v;
        float* o = res + ((int64_t)g * a.N + n) * 3;
        o[0] = play;
        o[1] = rep;
        o[2] = v;
      }
      kdraw = k;
    }
    __syncthreads();
  }
  float* out_notes = a.state ? a.results + (int64_t)a.state->step * G * a.N * 3 : a.next_notes;
  for (int i = tid; i < G * a.N * 3; i += blockDim.x) out_notes[i] = res[i];
  if (tid == 0) {
    if (a.state) {
      a.state->draw_off = draw0 + kdraw;
      if (knear && a.state->near_ties == 0) a.state->first_near_step = a.state->step;
      a.state->near_ties += knear;
    } else {
      a.draws_used[0] = kdraw;
      a.draws_used[1] = knear;
    }
  }
}


// The same walk for G <= 4 pieces and 4 Hn <= 512 (the reference shape) with the dot products split over K: thread =
// (4 adjacent gate columns, one quarter of K).  With one thread per column every thread read the whole x vector of
// every piece from LDS -- 288 broadcast ds_read_b128 per wave and note, ~0.68 of the 0.74 ms the walk took WITHOUT
// its weight loads; here a thread reads a quarter of x, loads its weights 16 bytes at a time, and the cell-update
// thread of unit (g, u) adds the four partial sums of its four gates.
template <bool SIGM>
__global__ __launch_bounds__(512) void gen_sample_ks_kernel(GenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Hn = a.Hn, G = a.G, C4 = 4 * Hn, Ln = a.Ln, KQ = Hn / 4;
  float* hs = sm;                          // [Ln][G][Hn]
  float* cs = hs + Ln * G * Hn;            // [Ln][G][Hn]
  float* zp = cs + Ln * G * Hn;            // [4][G][4Hn] partial pre-activations per K quarter
  float* xs = zp + 4 * G * C4;             // [G][Hn]   input of layers >= 1 (h below + style)
  float* spl = xs + G * Hn;                // [Ln][G][Hn]  style term added to the input of layer l >= 1
  float* hw = spl + Ln * G * Hn;           // [3][Hn] head weights (play, replay, volume) + [4] biases
  float* res = hw + 3 * Hn + 4;            // [G][N][3] sampled notes of this time step
  float* chosen = res + G * a.N * 3;       // [G][4]    previous note (play, replay, volume)
  float* logit = chosen + G * 4;           // [G][4]
  float* temps = logit + G * 4;            // [G]
  double* ul = (double*)(((uintptr_t)(temps + G) + 7) & ~(uintptr_t)7);   // [2 N G] uniforms of this time step
  __shared__ int kdraw, knear;
  const int tid = threadIdx.x;
  const int cg = tid % Hn, ks = tid / Hn;              // columns 4 cg .. 4 cg + 3, k in [ks KQ, (ks + 1) KQ)
  const int cu_g = tid / Hn, cu_u = tid % Hn;          // cell-update role: unit (g, u), valid while tid < G Hn
  const bool cell = tid < G * Hn;
  const int draw0 = a.state ? a.state->draw_off : 0;
  for (int i = tid; i < 2 * Ln * G * Hn; i += blockDim.x) hs[i] = 0.f;       // hs and cs
  for (int i = tid; i < G * 4; i += blockDim.x) chosen[i] = 0.f;
  for (int i = tid; i < Ln * G * Hn; i += blockDim.x) {
    const int l = i / (G * Hn), r = i % (G * Hn), g = r / Hn, u = r % Hn;
    spl[i] = l ? a.svec[GEN_MAXG * 64 + (int64_t)l * GEN_MAXG * 512 + g * 512 + u] : 0.f;
  }
  for (int i = tid; i < Hn; i += blockDim.x) {
    hw[i] = a.P[a.p_nd_W + (int64_t)i * 2];
    hw[Hn + i] = a.P[a.p_nd_W + (int64_t)i * 2 + 1];
    hw[2 * Hn + i] = a.P[a.p_vd_W + i];
  }
  if (tid < 2) hw[3 * Hn + tid] = a.P[a.p_nd_b + tid];
  if (tid == 2) hw[3 * Hn + 2] = a.P[a.p_vd_b];
  for (int i = tid; i < 2 * a.N * G; i += blockDim.x) ul[i] = a.uniforms[draw0 + i];
  if (tid < G) temps[tid] = a.state ? (float)a.state->temperature[tid] : a.temperature[tid];
  if (tid == 0) {
    kdraw = 0;                             // index into ul
    knear = 0;
  }
  // cell-update constants of unit (g, u), per gate: the three `chosen` rows of the layer-0 kernel, the upper biases,
  // and this note's x W + b
  float wch[3][4], bl[4][4], zx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int c = 0; c < 3; ++c) wch[c][q] = cell ? a.P[a.W[0] + (int64_t)(a.Ht + c) * C4 + q * Hn + cu_u] : 0.f;
#pragma unroll
    for (int l = 1; l < 4; ++l) bl[l][q] = (cell && l < Ln) ? a.P[a.b[l] + q * Hn + cu_u] : 0.f;
    zx[q] = cell ? a.zx0[((int64_t)cu_g * a.N + 0) * C4 + q * Hn + cu_u] : 0.f;
  }
  __syncthreads();

  // partial[g][c] += sum over this thread's K quarter of x[g][k] W[k][4 cg + c]
  auto dot4 = [&](float (&acc)[4][4], const float* __restrict__ Wm, const float* x) {
#pragma unroll 1
    for (int k0 = 0; k0 < KQ; k0 += 8) {
      float4 wv[8];
      const float* wsrc = Wm + (int64_t)(ks * KQ + k0) * C4 + 4 * cg;
#pragma unroll
      for (int i = 0; i < 8; ++i) wv[i] = *(const float4*)(wsrc + (int64_t)i * C4);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (g < G) {
          const float4 x0 = *(const float4*)(x + g * Hn + ks * KQ + k0), x1 = *(const float4*)(x + g * Hn + ks * KQ + k0 + 4);
          const float xe[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            acc[g][0] += xe[i] * wv[i].x;
            acc[g][1] += xe[i] * wv[i].y;
            acc[g][2] += xe[i] * wv[i].z;
            acc[g][3] += xe[i] * wv[i].w;
          }
        }
    }
  };

  // barriers inside the note loop: LDS traffic only -- __syncthreads() also waits for vmcnt(0), i.e. for the next note's
  // x W + b that the cell update has just requested (one exposed L2 round trip per note)
  auto lds_barrier = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  for (int n = 0; n < a.N; ++n) {
#pragma unroll 1
    for (int l = 0; l < Ln; ++l) {
      float acc[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[g][c] = 0.f;
      if (l > 0) dot4(acc, a.P + a.W[l], xs);
      dot4(acc, a.P + a.U[l], hs + l * G * Hn);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (g < G) *(float4*)(zp + ((int64_t)(ks * G + g)) * C4 + 4 * cg) = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
      lds_barrier();
      // ---- cell update (Keras gate order i,f,c,o): unit (g, u) = (cu_g, cu_u)
      if (cell) {
        float z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float b;
          if (l == 0)
            b = zx[q] + chosen[cu_g * 4] * wch[0][q] + chosen[cu_g * 4 + 1] * wch[1][q] + chosen[cu_g * 4 + 2] * wch[2][q];
          else
            b = l == 1 ? bl[1][q] : (l == 2 ? bl[2][q] : bl[3][q]);
          const float* zq = zp + (int64_t)cu_g * C4 + q * Hn + cu_u;
          z[q] = b + ((zq[0] + zq[(int64_t)G * C4]) + (zq[(int64_t)2 * G * C4] + zq[(int64_t)3 * G * C4]));
        }
        if (l == 0 && n + 1 < a.N) {       // next note's x W + b: in flight for a whole note
#pragma unroll
          for (int q = 0; q < 4; ++q) zx[q] = a.zx0[((int64_t)cu_g * a.N + n + 1) * C4 + q * Hn + cu_u];
        }
        const float ig = dj_ract<SIGM>(z[0]), fg = dj_ract<SIGM>(z[1]), gg = dj_tanh(z[2]), og = dj_ract<SIGM>(z[3]);
        const int si = (l * G + cu_g) * Hn + cu_u;
        const float cn = fg * cs[si] + ig * gg;
        cs[si] = cn;
        const float hv = og * dj_tanh(cn);
        hs[si] = hv;
        if (l + 1 < Ln) xs[cu_g * Hn + cu_u] = hv + spl[((l + 1) * G + cu_g) * Hn + cu_u];
      }
      lds_barrier();
    }
    // ---- heads: (play, replay) = sigmoid(h Wn + bn), volume = h Wv + bv   (model.py:94-95); one 32-lane half wave
    // per (piece, output): all G x 3 sums in one round, DPP row sums + one cross-row exchange
    {
      const int half = tid >> 5, l32 = tid & 31;
      if (half < G * 3) {
        const int g = half / 3, o = half - g * 3;
        const float* ht = hs + ((Ln - 1) * G + g) * Hn;
        float s = 0.f;
        for (int k = l32; k < Hn; k += 32) s += ht[k] * hw[o * Hn + k];
        s = dj_row16_sum(s);
        s += __shfl_xor(s, 16);
        if (l32 == 0) logit[g * 4 + o] = s + hw[3 * Hn + o];
      }
    }
    lds_barrier();
    // ---- sampling, reference draw order (generate.py:47-58,116-118)
    if (tid == 0) {
      int k = kdraw;
      for (int g = 0; g < G; ++g) {
        float pp = dj_sigmoid(logit[g * 4]), pr = dj_sigmoid(logit[g * 4 + 1]);
        const float vol = logit[g * 4 + 2];
        const float temp = temps[g];
        if (temp != 1.0f) {                       // apply_temperature, float32 like the reference (generate.py:81-91)
          float x0 = -logf(1.0f / pp - 1.0f), x1 = -logf(1.0f / pr - 1.0f);
          pp = 1.0f / (1.0f + expf(-x0 / temp));
          pr = 1.0f / (1.0f + expf(-x1 / temp));
        }
        float play = 0.f, rep = 0.f, v = 0.f;
        const double u0 = ul[k++];
        knear += fabs(u0 - (double)pp) < DJ_GEN_TIE_BAND;
        if (u0 <= (double)pp) {
          play = 1.f;
          v = vol;
          const double u1 = ul[k++];
          knear += fabs(u1 - (double)pr) < DJ_GEN_TIE_BAND;
          if (u1 <= (double)pr) rep = 1.f;
        }
        chosen[g * 4] = play;
        chosen[g * 4 + 1] = rep;
        chosen[g * 4 + 2] = v;
        float* o = res + ((int64_t)g * a.N + n) * 3;
        o[0] = play;
        o[1] = rep;
        o[2] = v;
      }
      kdraw = k;
    }
    // no barrier here: the next note's first phase (h U of layer 0) needs neither `chosen` nor the logits, so the
    // other waves start it while thread 0 samples; `chosen` is read behind the next barrier (cell update of layer 0)
  }
  __syncthreads();
  float* out_notes = a.state ? a.results + (int64_t)a.state->step * G * a.N * 3 : a.next_notes;
  for (int i = tid; i < G * a.N * 3; i += blockDim.x) out_notes[i] = res[i];
  if (tid == 0) {
    if (a.state) {
      a.state->draw_off = draw0 + kdraw;
      if (knear && a.state->near_ties == 0) a.state->first_near_step = a.state->step;
      a.state->near_ties += knear;
    } else {
      a.draws_used[0] = kdraw;
      a.draws_used[1] = knear;
    }
  }
}


// ---------------------------------------------------------------- the same walk on the matrix cores (bf16 mode, round 5)
// The samplers above multiply on the vector ALUs against the fp32 master weights -- right for the fp32 mode, whose sampled
// notes are certified against the oracle (probabilities within ~1e-6), and 10 us per note: 0.48 of the 0.84 ms of a
// generated time step.  In bf16 mode the time axis already runs on bf16 operands, so the note axis may too (as it does
// in training): here every note's products are MFMAs on bf16 weight fragments -- M = 32 rows of which G are pieces,
// the rest a zero row -- with fp32 accumulation, cell state, heads and draws.  The 384 KB of fragments a note needs
// (U0, [W1 ; U1]; packed once per run by gen_pack_bf16_kernel) do not depend on the note chain, so each wave keeps a ring
// of RD fragments in flight that simply runs on from note to note: the stream costs its bytes (2.9 us per note at the
// compute unit's 64 B/clk) and no latency.  Wave w = (unit group w & 3 of 32 units, K half w >> 2) owns the four gate
// tiles of its units, so a unit's four gates meet in one thread's reach; the two K halves meet in LDS.
constexpr int GM_NT = 16, GM_KC0 = 8, GM_KC1 = 16;              // 32-column tiles of 4 Hn = 512; k-chunks of 16: layer 0 / 1
constexpr int GM_FRAGS = GM_NT * (GM_KC0 + GM_KC1);             // 384 fragments of 1 KiB
// wpk[(f * 64 + lane) * 8 + e], f = layer offset + nt * NKC + kc: B fragment (column nt * 32 + l31, k = kc * 16 + 8 h + e)
__global__ void gen_pack_bf16_kernel(const float* __restrict__ P, int64_t U0, int64_t W1, int64_t U1, bf16_t* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= GM_FRAGS * 512) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, f = idx >> 9;
  const int col_in = lane & 31, kin = 8 * (lane >> 5) + e;
  float v;
  if (f < GM_NT * GM_KC0) {
    const int nt = f / GM_KC0, kc = f % GM_KC0;
    v = P[U0 + (int64_t)(kc * 16 + kin) * 512 + nt * 32 + col_in];
  } else {
    const int f1 = f - GM_NT * GM_KC0, nt = f1 / GM_KC1, kc = f1 % GM_KC1, k = kc * 16 + kin;
    v = k < 128 ? P[W1 + (int64_t)k * 512 + nt * 32 + col_in] : P[U1 + (int64_t)(k - 128) * 512 + nt * 32 + col_in];
  }
  out[idx] = (bf16_t)v;
}
template <bool SIGM>
__global__ __launch_bounds__(512) void gen_sample_mfma_kernel(GenArgs a, const bf16_t* __restrict__ wpk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int Hn = 128, C4 = 512, RD = 16;
  const int G = a.G, N = a.N;
  float* zp = sm;                          // [2 K halves][4][512] partial pre-activations
  float* hs1 = zp + 2 * 4 * C4;            // [4][128] fp32 h of layer 1 (heads)
  float* spl = hs1 + 4 * Hn;               // [4][128] style term of layer 1
  float* hw = spl + 4 * Hn;                // [3][128] + [4]
  float* res = hw + 3 * Hn + 4;            // [4][N][3]
  float* chosen = res + 4 * N * 3;         // [4][4]
  float* logit = chosen + 16;              // [4][4]
  float* temps = logit + 16;               // [8]
  bf16_t* xb = (bf16_t*)(temps + 8);       // [3 vectors: h0, x1, h1][5 rows: 4 pieces + a zero row][128] bf16 A operand source
  double* ul = (double*)(((uintptr_t)(xb + 3 * 5 * Hn) + 7) & ~(uintptr_t)7);   // [2 N G]
  __shared__ int kdraw, knear;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), ug = w & 3, kh = w >> 2;
  const int cu_g = tid >> 7, cu_u = tid & 127;          // cell role: unit (g, u), valid while tid < 128 G
  const bool cell = tid < G * Hn;
  const int draw0 = a.state ? a.state->draw_off : 0;
  for (int i = tid; i < 3 * 5 * Hn / 2; i += 512) ((unsigned*)xb)[i] = 0u;
  for (int i = tid; i < G * Hn; i += 512) spl[i] = a.svec[GEN_MAXG * 64 + (int64_t)GEN_MAXG * 512 + (i / Hn) * 512 + i % Hn];
  for (int i = tid; i < Hn; i += 512) {
    hw[i] = a.P[a.p_nd_W + (int64_t)i * 2];
    hw[Hn + i] = a.P[a.p_nd_W + (int64_t)i * 2 + 1];
    hw[2 * Hn + i] = a.P[a.p_vd_W + i];
  }
  if (tid < 2) hw[3 * Hn + tid] = a.P[a.p_nd_b + tid];
  if (tid == 2) hw[3 * Hn + 2] = a.P[a.p_vd_b];
  for (int i = tid; i < 2 * N * G; i += 512) ul[i] = a.uniforms[draw0 + i];
  if (tid < G) temps[tid] = a.state ? (float)a.state->temperature[tid] : a.temperature[tid];
  if (tid < 16) chosen[tid] = 0.f;
  if (tid == 0) {
    kdraw = 0;
    knear = 0;
  }
  float wch[3][4], b1[4], zx[4], c0 = 0.f, c1 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int c = 0; c < 3; ++c) wch[c][q] = cell ? a.P[a.W[0] + (int64_t)(a.Ht + c) * C4 + q * Hn + cu_u] : 0.f;
    b1[q] = cell ? a.P[a.b[1] + q * Hn + cu_u] : 0.f;
    zx[q] = cell ? a.zx0[((int64_t)cu_g * N + 0) * C4 + q * Hn + cu_u] : 0.f;
  }
  // this wave's fragment stream of one note: 16 of layer 0 (k-chunk kh * 4 + i / 4, gate i % 4), then 32 of layer 1
  // (k-chunk kh * 8 + i / 4, gate i % 4); gate q of unit group ug is column tile q * 4 + ug
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wpk, 0, GM_FRAGS * 1024, 0x00020000);
  auto frag_off = [&](int i) {              // i in [0, 48)
    return i < 16 ? (((i & 3) * 4 + ug) * GM_KC0 + kh * 4 + (i >> 2)) * 1024
                  : (GM_NT * GM_KC0 + ((i & 3) * 4 + ug) * GM_KC1 + kh * 8 + ((i - 16) >> 2)) * 1024;
  };
  typedef unsigned gm_u32x4 __attribute__((ext_vector_type(4)));
  auto ldw = [&](int i) {
    const gm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, frag_off(i), 0);
    return __builtin_bit_cast(bf16x8, v);
  };
  bf16x8 rq[RD];
#pragma unroll
  for (int i = 0; i < RD; ++i) rq[i] = ldw(i);
  // A operand rows: pieces for l31 < G, the zero row (index 4) for the other lanes
  const int arow = l31 < G ? l31 : 4;
  const bf16_t* a_h0 = xb + arow * Hn + 8 * h;
  const bf16_t* a_x1 = a_h0 + 5 * Hn;
  const bf16_t* a_h1 = a_h0 + 10 * Hn;
  auto lds_barrier = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0): weight fragments and the next note's x W + b stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  __syncthreads();

  for (int n = 0; n < N; ++n) {
    // ---- both layers; every ring index below is a compile-time constant (48 % RD == 0)
    dj_gen_static_for<0, 2>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      constexpr int I0 = l == 0 ? 0 : 16, NI = l == 0 ? 16 : 32;
      f32x16 acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      dj_gen_static_for<0, NI>([&](auto ic) {
        constexpr int i = I0 + decltype(ic)::value, kcl = (i - I0) >> 2, q = i & 3;
        bf16x8 av;
        if constexpr (l == 0) {
          av = *(const bf16x8*)(a_h0 + (kh * 4 + kcl) * 16);
        } else {
          // k-chunks 0..7 of [x1 | h1] are x1, 8..15 h1: K half kh is exactly one of the two vectors
          av = *(const bf16x8*)((kh == 0 ? a_x1 : a_h1) + kcl * 16);
        }
        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, rq[i % RD], acc[q], 0, 0, 0);
        rq[i % RD] = ldw((i + RD) % 48);
      });
      // rows 0 .. G-1 of the accumulators (registers 0..2(3) of lanes 0..31) are the pieces
      if (h == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (g < G) zp[(kh * 4 + g) * C4 + q * Hn + ug * 32 + l31] = acc[q][g];
      }
      lds_barrier();
      // ---- cell update of unit (cu_g, cu_u) (Keras gate order i, f, c, o)
      if (cell) {
        float z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float b;
          if constexpr (l == 0)
            b = zx[q] + chosen[cu_g * 4] * wch[0][q] + chosen[cu_g * 4 + 1] * wch[1][q] + chosen[cu_g * 4 + 2] * wch[2][q];
          else
            b = b1[q];
          z[q] = b + (zp[cu_g * C4 + q * Hn + cu_u] + zp[(4 + cu_g) * C4 + q * Hn + cu_u]);
        }
        if (l == 0 && n + 1 < N) {         // next note's x W + b: in flight for a whole note
#pragma unroll
          for (int q = 0; q < 4; ++q) zx[q] = a.zx0[((int64_t)cu_g * N + n + 1) * C4 + q * Hn + cu_u];
        }
        const float ig = dj_ract<SIGM>(z[0]), fg = dj_ract<SIGM>(z[1]), gg = dj_tanh(z[2]), og = dj_ract<SIGM>(z[3]);
        float& cc = l == 0 ? c0 : c1;
        cc = fg * cc + ig * gg;
        const float hv = og * dj_tanh(cc);
        if constexpr (l == 0) {
          xb[cu_g * Hn + cu_u] = (bf16_t)hv;                                     // h0: next note's recurrent operand
          xb[5 * Hn + cu_g * Hn + cu_u] = (bf16_t)(hv + spl[cu_g * Hn + cu_u]);  // x1: layer 1's input
        } else {
          xb[10 * Hn + cu_g * Hn + cu_u] = (bf16_t)hv;
          hs1[cu_g * Hn + cu_u] = hv;
        }
      }
      lds_barrier();
    });
    // ---- heads: (play, replay) = sigmoid(h Wn + bn), volume = h Wv + bv   (model.py:94-95)
    {
      const int half = tid >> 5, l32 = tid & 31;
      if (half < G * 3) {
        const int g = half / 3, o = half - g * 3;
        const float* ht = hs1 + g * Hn;
        float s0 = 0.f;
        for (int k = l32; k < Hn; k += 32) s0 += ht[k] * hw[o * Hn + k];
        s0 = dj_row16_sum(s0);
        s0 += __shfl_xor(s0, 16);
        if (l32 == 0) logit[g * 4 + o] = s0 + hw[3 * Hn + o];
      }
    }
    lds_barrier();
    // ---- sampling, reference draw order (generate.py:47-58,116-118)
    if (tid == 0) {
      int k = kdraw;
      for (int g = 0; g < G; ++g) {
        float pp = dj_sigmoid(logit[g * 4]), pr = dj_sigmoid(logit[g * 4 + 1]);
        const float vol = logit[g * 4 + 2];
        const float temp = temps[g];
        if (temp != 1.0f) {                       // apply_temperature, float32 like the reference (generate.py:81-91)
          float x0 = -logf(1.0f / pp - 1.0f), x1 = -logf(1.0f / pr - 1.0f);
          pp = 1.0f / (1.0f + expf(-x0 / temp));
          pr = 1.0f / (1.0f + expf(-x1 / temp));
        }
        float play = 0.f, rep = 0.f, v = 0.f;
        const double u0 = ul[k++];
        knear += fabs(u0 - (double)pp) < DJ_GEN_TIE_BAND;
        if (u0 <= (double)pp) {
          play = 1.f;
          v = vol;
          const double u1 = ul[k++];
          knear += fabs(u1 - (double)pr) < DJ_GEN_TIE_BAND;
          if (u1 <= (double)pr) rep = 1.f;
        }
        chosen[g * 4] = play;
        chosen[g * 4 + 1] = rep;
        chosen[g * 4 + 2] = v;
        float* o = res + ((int64_t)g * N + n) * 3;
        o[0] = play;
        o[1] = rep;
        o[2] = v;
      }
      kdraw = k;
    }
    // no barrier: `chosen` is read behind the next note's first barrier (cell update of layer 0)
  }
  __syncthreads();
  float* out_notes = a.state ? a.results + (int64_t)a.state->step * G * N * 3 : a.next_notes;
  for (int i = tid; i < G * N * 3; i += 512) out_notes[i] = res[i];
  if (tid == 0) {
    if (a.state) {
      a.state->draw_off = draw0 + kdraw;
      if (knear && a.state->near_ties == 0) a.state->first_near_step = a.state->step;
      a.state->near_ties += knear;
    } else {
      a.draws_used[0] = kdraw;
      a.draws_used[1] = knear;
    }
  }
}

// end_time() of the reference on the device (generate.py:60-79): silence / temperature schedule,
// then the windows slide by one step: dst[:, t] = src[:, t+1], dst[:, T-1] = new notes / beat(t).
__global__ void gen_advance_kernel(DjGenState* st, const float* __restrict__ results, const float* __restrict__ nsrc,
                                   float* __restrict__ ndst, const float* __restrict__ bsrc, float* __restrict__ bdst,
                                   int G, int T, int N, int NB, int phase) {
  const int step = st->step;
  const float* nn = results + (int64_t)step * G * N * 3;
  const int tot_n = G * T * N * 3, tot_b = G * T * NB;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot_n + tot_b; i += gridDim.x * blockDim.x) {
    if (i < tot_n) {
      const int e = i % (N * 3), t = (i / (N * 3)) % T, g = i / (N * 3 * T);
      ndst[i] = t + 1 < T ? nsrc[i + N * 3] : nn[g * N * 3 + e];
    } else {
      const int j = i - tot_n, e = j % NB, t = (j / NB) % T;
      bdst[j] = t + 1 < T ? bsrc[j + NB] : ((e == step % NB) ? 1.f : 0.f);   // compute_beat(t, NOTES_PER_BAR)
    }
  }
}
// runs after gen_advance_kernel of the same step (separate launch: every block above reads st->step)
__global__ void gen_state_kernel(DjGenState* st, const float* __restrict__ results, int G, int N, int notes_per_bar) {
  const int g = threadIdx.x;
  if (g < G) {
    const float* nn = results + ((int64_t)st->step * G + g) * N * 3;
    bool any = false;
    for (int i = 0; i < N * 3; ++i) any = any || (nn[i] != 0.f);
    if (!any) {
      st->silent[g] += 1;
      if (st->silent[g] >= notes_per_bar) st->temperature[g] += 0.1;   // NOTES_PER_BAR (generate.py:65-67)
    } else {
      st->silent[g] = 0;
      st->temperature[g] = st->default_temp[g];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) st->step += 1;
}

}  // namespace

int dj_launch_gen_advance(void* state, const float* results, const float* nsrc, float* ndst, const float* bsrc,
                          float* bdst, int G, int T, int N, int NB, hipStream_t st) {
  hipLaunchKernelGGL(gen_advance_kernel, dim3(64), dim3(256), 0, st, (DjGenState*)state, results, nsrc, ndst, bsrc,
                     bdst, G, T, N, NB, 0);
  hipLaunchKernelGGL(gen_state_kernel, dim3(1), dim3(64), 0, st, (DjGenState*)state, results, G, N, NB);
  return (int)hipGetLastError();
}
int dj_gen_state_bytes() { return (int)sizeof(DjGenState); }
// bytes of the bf16 weight fragments of the matrix-core sampler (U0, [W1 ; U1] of a 2 x 128 note axis), behind the
// sampler's float scratch
int dj_gen_wpack_bytes() { return GM_FRAGS * 1024; }
int dj_launch_generate_pack(int dtype, int Hn, int Ln, const float* P, const int64_t* offs, void* wpack, uint32_t kf,
                            hipStream_t st) {
  if (!(dtype == DJ_BF16 && !(kf & DJ_KF_NO_GEN_MFMA) && wpack && Hn == 128 && Ln == 2)) return 0;
  hipLaunchKernelGGL(gen_pack_bf16_kernel, dim3((GM_FRAGS * 512 + 255) / 256), dim3(256), 0, st, P, offs[9], offs[8 + 5],
                     offs[9 + 5], (bf16_t*)wpack);
  return (int)hipGetLastError();
}

// the sampler's style terms alone (gen_prep_kernel into the scratch): dj_generate_prepare
int dj_launch_generate_prep(int G, int T, int N, int Ht, int Hn, int Ln, int S, int SU, const float* P, const int64_t* offs,
                            const float* style_last, int64_t style_stride, float* scratch, hipStream_t st) {
  if (G < 1 || G > GEN_MAXG || Ln < 1 || Ln > 4 || 4 * Hn > 1024 || (Hn % 32) || Ht + 3 > 512 || SU > 64 || S > 64) return 1300;
  GenArgs a;
  memset(&a, 0, sizeof(a));
  a.G = G; a.N = N; a.Hn = Hn; a.Ht = Ht; a.Ln = Ln; a.S = S; a.SU = SU; a.T = T; a.P = P;
  a.p_style_W = offs[0]; a.p_style_b = offs[1];
  for (int l = 0; l < Ln; ++l) {
    a.dW[l] = offs[6 + 5 * l]; a.db[l] = offs[7 + 5 * l];
  }
  a.D0 = Ht + 3;
  a.style_last = style_last;
  a.style_stride = style_stride;
  a.svec = scratch;
  hipLaunchKernelGGL(gen_prep_kernel, dim3(Ln), dim3(512), 0, st, a);
  return (int)hipGetLastError();
}

// Htime: TA-ordered top time-axis h buffer of the window (operand dtype).  scratch: float workspace
// of at least GEN_MAXG*64 + 4*GEN_MAXG*512 + G*N*4Hn floats.
int dj_launch_generate_notes(int dtype, int G, int T, int N, int Ht, int Hn, int Ln, int S, int SU, const float* P,
                             const int64_t* offs /* [6 + 5*Ln] */, const void* Htime, const float* style_last,
                             int64_t style_stride,
                             float* scratch, const double* uniforms, const float* temperature, float* next_notes,
                             int* draws_used, void* state, float* results, int sigm, int static_ready,
                             void* wpack, uint32_t kf, hipStream_t st) {
  if (G < 1 || G > GEN_MAXG || Ln < 1 || Ln > 4 || 4 * Hn > 1024 || (Hn % 32) || Ht + 3 > 512 || SU > 64 || S > 64) return 1300;
  GenArgs a;
  a.G = G; a.N = N; a.Hn = Hn; a.Ht = Ht; a.Ln = Ln; a.S = S; a.SU = SU; a.T = T; a.P = P;
  a.p_style_W = offs[0]; a.p_style_b = offs[1]; a.p_nd_W = offs[2]; a.p_nd_b = offs[3]; a.p_vd_W = offs[4];
  a.p_vd_b = offs[5];
  for (int l = 0; l < Ln; ++l) {
    a.dW[l] = offs[6 + 5 * l]; a.db[l] = offs[7 + 5 * l]; a.W[l] = offs[8 + 5 * l]; a.U[l] = offs[9 + 5 * l];
    a.b[l] = offs[10 + 5 * l];
  }
  a.D0 = Ht + 3;
  a.style_last = style_last;
  a.style_stride = style_stride;
  a.svec = scratch;
  a.zx0 = scratch + GEN_MAXG * 64 + 4 * GEN_MAXG * 512;
  a.uniforms = uniforms; a.temperature = temperature; a.next_notes = next_notes; a.draws_used = draws_used;
  a.state = (DjGenState*)state; a.results = results;
  // static_ready: the style terms in the scratch are current (they depend on the style vector and the weights only)
  if (!static_ready) hipLaunchKernelGGL(gen_prep_kernel, dim3(Ln), dim3(512), 0, st, a);
  dim3 gz((4 * Hn + 255) / 256, G * ((N + ZX_NB - 1) / ZX_NB));
  if (dtype == DJ_F32)
    hipLaunchKernelGGL(gen_zx0_kernel<float>, gz, dim3(256), 0, st, a, (const float*)Htime);
  else
    hipLaunchKernelGGL(gen_zx0_kernel<bf16_t>, gz, dim3(256), 0, st, a, (const bf16_t*)Htime);
  const size_t smem = ((size_t)3 * Ln * G * Hn + (size_t)G * 4 * Hn + (size_t)G * Hn + 3 * Hn + 4 + (size_t)G * N * 3 + 9 * G + 8) *
                          sizeof(float) + (size_t)2 * N * G * sizeof(double) + 16;
  if (smem > 64 * 1024) return 1301;
  // bf16 mode, the reference's note axis (2 x 128 units), up to 4 pieces: the walk on the matrix cores against bf16
  // weight fragments (gen_sample_mfma_kernel); the fragments depend on the parameters only (static_ready: packed already)
  if (dtype == DJ_BF16 && !(kf & DJ_KF_NO_GEN_MFMA) && wpack && Hn == 128 && Ln == 2 && G <= 4) {
    const size_t smem_m = ((size_t)2 * 4 * 512 + 4 * 128 + 4 * 128 + 3 * 128 + 4 + (size_t)4 * N * 3 + 16 + 16 + 8) * sizeof(float) +
                          (size_t)3 * 5 * 128 * sizeof(bf16_t) + (size_t)2 * N * G * sizeof(double) + 16;
    if (smem_m <= 64 * 1024) {
      if (!static_ready)
        hipLaunchKernelGGL(gen_pack_bf16_kernel, dim3((GM_FRAGS * 512 + 255) / 256), dim3(256), 0, st, P, a.U[0], a.W[1],
                           a.U[1], (bf16_t*)wpack);
      if (sigm)
        hipLaunchKernelGGL((gen_sample_mfma_kernel<true>), dim3(1), dim3(512), smem_m, st, a, (const bf16_t*)wpack);
      else
        hipLaunchKernelGGL((gen_sample_mfma_kernel<false>), dim3(1), dim3(512), smem_m, st, a, (const bf16_t*)wpack);
      return (int)hipGetLastError();
    }
  }
  const bool ks_off = (kf & DJ_KF_NO_GEN_KSPLIT) != 0;
  if (!ks_off && G <= 4 && 4 * Hn <= 512 && smem + (size_t)3 * G * 4 * Hn * sizeof(float) <= 64 * 1024) {
    const size_t smem_ks = smem + (size_t)3 * G * 4 * Hn * sizeof(float);      // zp is 4 x the size of zb
    if (sigm)
      hipLaunchKernelGGL((gen_sample_ks_kernel<true>), dim3(1), dim3(4 * Hn), smem_ks, st, a);
    else
      hipLaunchKernelGGL((gen_sample_ks_kernel<false>), dim3(1), dim3(4 * Hn), smem_ks, st, a);
    return (int)hipGetLastError();
  }
  // up to 512 threads (Hn <= 128) the sampler may use 256 VGPRs
#define DJ_GEN_LAUNCH(S, MT) hipLaunchKernelGGL((gen_sample_kernel<S, MT>), dim3(1), dim3(4 * Hn), smem, st, a)
  if (4 * Hn <= 512) {
    if (sigm) DJ_GEN_LAUNCH(true, 512); else DJ_GEN_LAUNCH(false, 512);
  } else {
    if (sigm) DJ_GEN_LAUNCH(true, 1024); else DJ_GEN_LAUNCH(false, 1024);
  }
#undef DJ_GEN_LAUNCH
  return (int)hipGetLastError();
}
