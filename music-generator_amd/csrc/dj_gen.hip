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
#include "dj_kernels.h"

namespace {

constexpr int GEN_MAXG = 8;
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
__global__ void gen_prep_kernel(GenArgs a) {
  __shared__ float st[GEN_MAXG * 64];
  for (int i = threadIdx.x; i < a.G * a.SU; i += blockDim.x) {
    int g = i / a.SU, k = i % a.SU;
    float s = a.P[a.p_style_b + k];
    for (int j = 0; j < a.S; ++j) s += a.style_last[g * a.style_stride + j] * a.P[a.p_style_W + (int64_t)j * a.SU + k];
    st[i] = s;
    a.svec[i] = s;
  }
  __syncthreads();
  for (int l = 0; l < a.Ln; ++l) {
    const int D = l == 0 ? a.D0 : a.Hn;
    float* sp = a.svec + GEN_MAXG * 64 + (int64_t)l * GEN_MAXG * 512;
    for (int i = threadIdx.x; i < a.G * D; i += blockDim.x) {
      int g = i / D, d = i % D;
      float s = a.P[a.db[l] + d];
      for (int k = 0; k < a.SU; ++k) s += st[g * a.SU + k] * a.P[a.dW[l] + (int64_t)k * D + d];
      sp[g * 512 + d] = dj_tanh(s);
    }
  }
}

// zx0[g,n,col] = b0[col] + sum_{k<Ht} (feat[g,n,k] + sp0[g,k]) W0[k,col] + sum_{c<3} sp0[g,Ht+c] W0[Ht+c,col]
// feat = last window step of the time axis, read straight from the TA-ordered h buffer.
template <typename T>
__global__ void gen_zx0_kernel(GenArgs a, const T* __restrict__ Htime) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;     // 0 .. 4Hn-1
  const int gn = blockIdx.y, g = gn / a.N, n = gn % a.N;
  if (col >= 4 * a.Hn) return;
  const float* sp0 = a.svec + GEN_MAXG * 64 + g * 512;
  const T* f = Htime + dj_row_ta(g, a.T - 1, n, a.T, a.N) * a.Ht;
  const float* W0 = a.P + a.W[0];
  const int ldw = 4 * a.Hn;
  float s = a.P[a.b[0] + col];
  for (int k = 0; k < a.Ht; ++k) s += (dj_to_f32(f[k]) + sp0[k]) * W0[(int64_t)k * ldw + col];
  for (int c = 0; c < 3; ++c) s += sp0[a.Ht + c] * W0[(int64_t)(a.Ht + c) * ldw + col];
  a.zx0[((int64_t)g * a.N + n) * ldw + col] = s;
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

// one workgroup, 4*Hn threads (one per gate column)
template <bool SIGM, int MAXT>
__global__ __launch_bounds__(MAXT) void gen_sample_kernel(GenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Hn = a.Hn, G = a.G, C4 = 4 * Hn;
  float* hs = sm;                          // [Ln][G][Hn]
  float* cs = hs + a.Ln * G * Hn;          // [Ln][G][Hn]
  float* zb = cs + a.Ln * G * Hn;          // [G][4Hn]
  float* xs = zb + G * C4;                 // [G][Hn]   input of layers >= 1 (h below + style)
  float* chosen = xs + G * Hn;             // [G][4]    previous note (play, replay, volume)
  float* logit = chosen + G * 4;           // [G][4]
  __shared__ int kdraw, knear;
  const int col = threadIdx.x;
  for (int i = threadIdx.x; i < 2 * a.Ln * G * Hn; i += blockDim.x) hs[i] = 0.f;
  for (int i = threadIdx.x; i < G * 4; i += blockDim.x) chosen[i] = 0.f;
  if (threadIdx.x == 0) {
    kdraw = a.state ? a.state->draw_off : 0;
    knear = 0;
  }
  float* out_notes = a.state ? a.results + (int64_t)a.state->step * G * a.N * 3 : a.next_notes;
  __syncthreads();

  for (int n = 0; n < a.N; ++n) {
    for (int l = 0; l < a.Ln; ++l) {
      // ---- pre-activations of column `col` for every piece
      float z[GEN_MAXG];
      const float* U = a.P + a.U[l];
      const float* hl = hs + l * G * Hn;
      if (l == 0) {
        const float* W0 = a.P + a.W[0];
#pragma unroll
        for (int g = 0; g < GEN_MAXG; ++g)
          if (g < G) {
            float s = a.zx0[((int64_t)g * a.N + n) * C4 + col];
#pragma unroll
            for (int c = 0; c < 3; ++c) s += chosen[g * 4 + c] * W0[(int64_t)(a.Ht + c) * C4 + col];
            z[g] = s;
          }
      } else {
        const float bl = a.P[a.b[l] + col];
#pragma unroll
        for (int g = 0; g < GEN_MAXG; ++g) z[g] = bl;
        gen_dot(z, a.P + a.W[l] + col, C4, xs, Hn, G);
      }
      gen_dot(z, U + col, C4, hl, Hn, G);
#pragma unroll
      for (int g = 0; g < GEN_MAXG; ++g)
        if (g < G) zb[g * C4 + col] = z[g];
      __syncthreads();
      // ---- cell update (Keras gate order i,f,c,o)
      for (int i = threadIdx.x; i < G * Hn; i += blockDim.x) {
        const int g = i / Hn, u = i % Hn;
        const float* zz = zb + g * C4;
        const float ig = dj_ract<SIGM>(zz[u]), fg = dj_ract<SIGM>(zz[Hn + u]), gg = dj_tanh(zz[2 * Hn + u]),
                    og = dj_ract<SIGM>(zz[3 * Hn + u]);
        const float cn = fg * cs[(l * G + g) * Hn + u] + ig * gg;
        cs[(l * G + g) * Hn + u] = cn;
        const float hv = og * dj_tanh(cn);
        hs[(l * G + g) * Hn + u] = hv;
        if (l + 1 < a.Ln) xs[g * Hn + u] = hv + a.svec[GEN_MAXG * 64 + (int64_t)(l + 1) * GEN_MAXG * 512 + g * 512 + u];
      }
      __syncthreads();
    }
    // ---- heads: (play, replay) = sigmoid(h Wn + bn), volume = h Wv + bv   (model.py:94-95)
    {
      const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
      const float* ht = hs + (a.Ln - 1) * G * Hn;
      for (int job = wv; job < G * 3; job += nw) {
        const int g = job / 3, o = job % 3;
        float s = 0.f;
        for (int k = lane; k < Hn; k += 64)
          s += ht[g * Hn + k] * (o < 2 ? a.P[a.p_nd_W + (int64_t)k * 2 + o] : a.P[a.p_vd_W + k]);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) s += __shfl_xor(s, sft);
        if (lane == 0) logit[g * 4 + o] = s + (o < 2 ? a.P[a.p_nd_b + o] : a.P[a.p_vd_b]);
      }
    }
    __syncthreads();
    // ---- sampling, reference draw order (generate.py:47-58,116-118)
    if (threadIdx.x == 0) {
      int k = kdraw;
      for (int g = 0; g < G; ++g) {
        float pp = dj_sigmoid(logit[g * 4]), pr = dj_sigmoid(logit[g * 4 + 1]);
        const float vol = logit[g * 4 + 2];
        const float temp = a.state ? (float)a.state->temperature[g] : a.temperature[g];
        if (temp != 1.0f) {                       // apply_temperature, float32 like the reference (generate.py:81-91)
          float x0 = -logf(1.0f / pp - 1.0f), x1 = -logf(1.0f / pr - 1.0f);
          pp = 1.0f / (1.0f + expf(-x0 / temp));
          pr = 1.0f / (1.0f + expf(-x1 / temp));
        }
        float play = 0.f, rep = 0.f, v = 0.f;
        const double u0 = a.uniforms[k++];
        knear += fabs(u0 - (double)pp) < DJ_GEN_TIE_BAND;
        if (u0 <= (double)pp) {
          play = 1.f;
          v = vol;
          const double u1 = a.uniforms[k++];
          knear += fabs(u1 - (double)pr) < DJ_GEN_TIE_BAND;
          if (u1 <= (double)pr) rep = 1.f;
        }
        chosen[g * 4] = play;
        chosen[g * 4 + 1] = rep;
        chosen[g * 4 + 2] = v;
        float* o = out_notes + ((int64_t)g * a.N + n) * 3;
        o[0] = play;
        o[1] = rep;
        o[2] = v;
      }
      kdraw = k;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (a.state) {
      a.state->draw_off = kdraw;
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
__global__ void gen_state_kernel(DjGenState* st, const float* __restrict__ results, int G, int N) {
  const int g = threadIdx.x;
  if (g < G) {
    const float* nn = results + ((int64_t)st->step * G + g) * N * 3;
    bool any = false;
    for (int i = 0; i < N * 3; ++i) any = any || (nn[i] != 0.f);
    if (!any) {
      st->silent[g] += 1;
      if (st->silent[g] >= 16) st->temperature[g] += 0.1;
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
  hipLaunchKernelGGL(gen_state_kernel, dim3(1), dim3(64), 0, st, (DjGenState*)state, results, G, N);
  return (int)hipGetLastError();
}
int dj_gen_state_bytes() { return (int)sizeof(DjGenState); }

// Htime: TA-ordered top time-axis h buffer of the window (operand dtype).  scratch: float workspace
// of at least GEN_MAXG*64 + 4*GEN_MAXG*512 + G*N*4Hn floats.
int dj_launch_generate_notes(int dtype, int G, int T, int N, int Ht, int Hn, int Ln, int S, int SU, const float* P,
                             const int64_t* offs /* [6 + 5*Ln] */, const void* Htime, const float* style_last,
                             int64_t style_stride,
                             float* scratch, const double* uniforms, const float* temperature, float* next_notes,
                             int* draws_used, void* state, float* results, int sigm, hipStream_t st) {
  if (G < 1 || G > GEN_MAXG || Ln < 1 || Ln > 4 || 4 * Hn > 1024 || (Hn % 32) || Ht + 3 > 512 || SU > 64) return 1300;
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
  hipLaunchKernelGGL(gen_prep_kernel, dim3(1), dim3(256), 0, st, a);
  dim3 gz((4 * Hn + 255) / 256, G * N);
  if (dtype == DJ_F32)
    hipLaunchKernelGGL(gen_zx0_kernel<float>, gz, dim3(256), 0, st, a, (const float*)Htime);
  else
    hipLaunchKernelGGL(gen_zx0_kernel<bf16_t>, gz, dim3(256), 0, st, a, (const bf16_t*)Htime);
  const size_t smem = ((size_t)2 * Ln * G * Hn + (size_t)G * 4 * Hn + (size_t)G * Hn + 8 * G) * sizeof(float);
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
