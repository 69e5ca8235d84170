// Internal C++ interface between the kernel translation units and dj_api.hip.
#pragma once
#include "dj_common.h"
#include "../../include/deepj_hip.h"      // DJ_KF_* kernel-selection flags

// `kf` arguments below: the effective DJ_KF_* bits of the call (dj_config.kernel_flags OR-ed with the DEEPJ_* process
// defaults read once at load / dj_env_reload(): dj_api.hip kflags)

struct FeatArgs {
  const float* notes;   // [B,T,N,3]
  const float* beat;    // [B,T,NB]
  const float* bins;    // [12,Bfull,T]: of the whole batch these B samples are rows bt0 / T ... of (dj_pitch_bins)
  const float* sp0;     // [B*T, F] tanh'd style projection of time layer 0
  const float* Wc;      // [24,3,64] conv kernel (Keras layout)
  const float* bc;      // [64]
  int B, T, N, NB, octave;
  int Bfull, bt0;       // batch the pitch_bins reshape runs over (model.py:47) and this call's first (b, t) row in it
  int F, FP;            // logical / padded feature width (94 / 96)
  DjDrop d_notes, d_beat, d_conv, d_style;
};

struct GlueArgs {
  int B, T, N;
  int Hd;          // width of the producing layer's h
  int D, DP;       // logical / padded width of the consuming layer's input
  int in_na, out_na;   // row order of producer / consumer (0 = TA, 1 = NA)
  const float* sp;     // [B*T, D] tanh'd style projection of the consuming layer (nullable)
  const float* chosen; // [B,T,N,3] or null: append shifted chosen after the Hd columns (model.py:101-106)
  DjDrop d_out, d_style, d_chosen;
};

struct HeadArgs {
  int B, T, N, Hd;
  const float* Wn;   // note_dense kernel [Hd,2]   (model.py:94)
  const float* bn;   // [2]
  const float* Wv;   // volume_dense kernel [Hd,1] (model.py:95)
  const float* bv;   // [1]
  const float* target;   // [B,T,N,3] or null (inference)
  float* out;            // [B,T,N,3] or null
  float* loss;           // scalar accumulator (+=) or null
  float* dWn;
  float* dbn;
  float* dWv;
  float* dbv;
  float inv_count;
  DjDrop d_out;
};

struct NadamArgs {
  float lr, beta1, beta2, eps;
  float mu_t, mu_t1;          // momentum_cache_t, momentum_cache_t_1
  float ms_new, ms_next;      // m_schedule_new, m_schedule_next
  float bc2;                  // 1 - beta2^t
  float gscale;               // gradient pre-scale (1/world_size for data parallel)
};

// up to 8 small dense layers on the same input A [M, K] (the per-layer style projections)
constexpr int DJ_DENSE_BATCH_MAX = 8;
struct DenseBatch {
  int n, M, K;
  const float* A;
  const float* W[DJ_DENSE_BATCH_MAX];    // [K, N_l]
  const float* b[DJ_DENSE_BATCH_MAX];    // [N_l] or null
  float* C[DJ_DENSE_BATCH_MAX];          // forward outputs [M, N_l]
  const float* dC[DJ_DENSE_BATCH_MAX];   // backward inputs [M, N_l]
  float* dW[DJ_DENSE_BATCH_MAX];
  float* db[DJ_DENSE_BATCH_MAX];
  int N[DJ_DENSE_BATCH_MAX];
};

// FORWARD LSTM cell of one recurrence step as the EPILOGUE of that step's GEMM (generic-width path, bf16; dj_step.hip
// drives it, dj_gemm.hip runs it): z_t = [x_t | h_{t-1}] [W ; U] + b in ONE product over K = K1p + H -- A (the kernel's
// A operand) holds the rows x_t (K1 valid columns, taken as zero up to K1p, a multiple of 64), A2 the rows h_{t-1} (null
// at step 0) -- with the rows of the packed B operand [4H][K1p + H] in GATE-INTERLEAVED order (row n' = 32 G + 8 g + e is
// gate column g H + 8 G + e of W and U: dj_launch_pack_wu_gates), so that a lane of the C^T accumulator block holds all
// four gates of 4 units of one row and turns them into c_t, h_t on the spot.  No x W pass, no z round trip, no gate launch.
// Rows are "all sequences at step t" of the sequence-tiled buffers: virtual row v at physical row ((v >> 5) * steps) * 32
// + (v & 31) of the pointers below, which stand at step t.  Writes Hs (row-major) and, when training (Cs != null), the
// two stashes BPTT reads (step_bwd8c_kernel); `carry` (c_{t-1}) is fp32 in FRAGMENT layout [rows/32][H/8][64 lanes][4]
// -- 16 bytes per lane, coalesced; only this epilogue touches it.
struct CellEpi {
  int H, steps, sigm, first;      // first: step 0 (c_{-1} = 0, carry not read)
  int K1, K1p, lda2;
  const void* A2;                 // h_{t-1} rows [.., H] (same row-block stride as A) or null
  float* carry;
  void* Z;                        // gate stash out (training only): 8-bit activated-gate codes, FRAGMENT layout
                                  //   [row block of the step][H/8][64 lanes][16 B = 4 gates x 4 units] (dj_common.h codes)
  void* Hs;                       // h_t out [.., H] row-major
  void* Cs;                       // c_t stash out, bf16 in the same fragment layout (8 B per lane), or null (inference)
};
// dj_gemm.hip
// A [M, K1] (lda, row-block stride a_rbs) and ce.A2 [M, H] x Bt [4H, K1p + H]^T (ldb), bias [4H] in natural gate order
int dj_launch_gemm_nt_cell(int M, const void* A, int lda, int a_rbs, const void* Bt, int ldb, const CellEpi& ce,
                           const float* bias, hipStream_t st);
int dj_launch_gemm_nt(int dtype, int M, int N, int K, const void* A, int lda, const void* Bt, int ldb, void* C, int ldc,
                      int c_mode, const float* bias, hipStream_t st);
// same with row-block strides on A and C (dj_gemm.hip rbs_row): per-step views of sequence-tiled buffers
int dj_launch_gemm_nt_rbs(int dtype, int M, int N, int K, const void* A, int lda, int a_rbs, const void* Bt, int ldb,
                          void* C, int ldc, int c_rbs, int c_mode, const float* bias, hipStream_t st);
// A column-tile-major (a_cts != 0; bf16, lda = 256): element (m, k) at A + (k >> 8) * a_cts + m * 256 + (k & 255)
int dj_launch_gemm_nt_ex(int dtype, int M, int N, int K, const void* A, int lda, int a_rbs, int64_t a_cts, const void* Bt,
                         int ldb, void* C, int ldc, int c_rbs, int c_mode, const float* bias, hipStream_t st);
int dj_launch_gemm_tn(int dtype, int64_t M, int Ka, int ka_valid, int N, const void* A, int lda, const void* B, int ldb, float* C,
                      int ldc, int a_shift, int steps, hipStream_t st);
// dz_cts: 0 = dZ row-major [M, N]; else column-tile-major [N/256][M][256] with dz_cts elements between tiles
int dj_launch_lstm_wgrad(int dtype, int64_t M, int steps, const void* X, int DP, int D, const void* Hs, int H,
                         const void* dZ, int N, int64_t dz_cts, float* dW, float* dU, const void* zeros, hipStream_t st);
// dj_lstm.hip
int dj_launch_lstm_pack(int dtype, int H, const float* U, void* fwd, void* bwd, hipStream_t st);
// Zx: x W + b of all steps (fragment-tiled, operand dtype); Gst: gate stash out (null = inference; fp32 may alias Zx)
int dj_launch_lstm_fwd(int dtype, int H, int ntiles, int steps, const void* Zx, void* Gst, const void* Upack, void* Hout,
                       void* Cout, int sigm, hipStream_t st);
// gate stash bytes per row of a layer with H units: fp32 keeps z (16 H), bf16 the activated gates as 8-bit codes (4 H)
int64_t dj_lstm_stash_row_bytes(int dtype, int H);
// WTpack/D/dX/DP: optional fused input gradient dX = dz W^T (WTpack from dj_launch_lstm_pack_wt; null = off;
// available where dj_lstm_bwd_has_dx says so)
// dz_cts: layout of the dZ output, as for dj_launch_lstm_wgrad (0 = row-major [rows, 4H])
// kf: DJ_KF_* bits (DJ_KF_BWD_PLAIN: bf16 H = 256 on lstm_bwd_kernel instead of the split-gate-math sweep)
int dj_launch_lstm_bwd(int dtype, int H, int ntiles, int steps, const void* Z, const void* UTpack, const void* C,
                       const void* dH, void* dZ, int64_t dz_cts, float* dbias, int sigm, const void* WTpack, int D, void* dX,
                       int DP, uint32_t kf, hipStream_t st);
int dj_lstm_bwd_has_dx(int dtype, int H, int D);
int dj_launch_lstm_pack_wt(int dtype, int H, const float* W, int D, void* out, hipStream_t st);
int dj_lstm_fused_nkx(int dtype, int H, int D);
int dj_launch_lstm_pack_w(int dtype, int H, const float* W, int D, int NKX, void* out, hipStream_t st);
// cluster_scratch: dj_lstm_cluster_scratch_bytes_impl() bytes, 128-byte aligned, owned by the caller's workspace
// (null = per-tile kernel only)
int dj_launch_lstm_fwd_cluster_f32(int ntiles, int steps, const void* Zx, const void* Upack, void* Hout, int sigm,
                                   void* scratch, uint32_t kf, hipStream_t st);
int dj_launch_lstm_fwd_cluster_pair(int ntiles, int steps, const void* X0, int DP0, const void* W0pack, const float* b0,
                                    const void* U0pack, void* X1, const void* W1pack, const float* b1, const void* U1pack,
                                    void* H1, const float* sp1, int sp_D, int n_seq, int n_b, int sigm, void* scratch,
                                    uint32_t kf, hipStream_t st);
int dj_launch_lstm_fwd_fused(int dtype, int H, int ntiles, int steps, const void* X, int DP, int NKX,
                             const void* Wpack, const float* bias, void* Zst, const void* Upack, void* Hout,
                             void* Cout, int sigm, void* cluster_scratch, uint32_t kf, hipStream_t st);
int64_t dj_lstm_cluster_scratch_bytes_impl();
// expired waits + misplaced clusters recorded in that scratch since the last call (0 in a healthy run; the affected
// tiles carry NaN), -1 on a HIP error; drains `st` (asynchronous copy on the caller's stream + synchronise: a blocking
// copy on the null stream does not order against a non-blocking stream)
int dj_lstm_cluster_faults_impl(void* cluster_scratch, hipStream_t st);
int dj_lstm_cluster_faults_take(void* cluster_scratch, int32_t* words_host, hipStream_t st);
// device address of the fault line inside a cluster scratch (32 ints; layout in dj_lstm.hip, "bounded exchange waits")
void* dj_lstm_cluster_fault_words(void* cluster_scratch);
// copy of that line as it stands once `st` has drained; resets nothing
int dj_lstm_cluster_fault_line(void* cluster_scratch, int32_t* words_host, hipStream_t st);
// dj_step.hip -- generic-H path (one GEMM + gate launch per recurrence step)
int64_t dj_lstm_step_scratch_floats(int H, int64_t ntiles);
int dj_launch_lstm_step_fwd(int dtype, int H, int ntiles, int steps, void* Z, const void* Ut, void* Hs, void* Cs,
                            float* scratch, int sigm, hipStream_t st);
// bf16: the whole forward sweep of a generic-width layer as `steps` GEMMs with the cell as their epilogue (CellEpi): X
// [rows, DP] (D valid columns), WU from dj_launch_pack_wu_gates (K1p = dj_step_k1p(DP)), bias [4H]; Z / Cs null = inference
int dj_step_k1p(int DP);
int dj_launch_lstm_step_fwd_fused(int H, int ntiles, int steps, const void* X, int DP, int D, const void* WU,
                                  const float* bias, void* Z, void* Hs, void* Cs, float* scratch, int sigm, hipStream_t st);
// dz_cts: 0 = dZ row-major [rows, 4H]; bf16 with 4H % 256 == 0: column-tile-major [4H/256][rows][256], as dj_launch_lstm_bwd
// codes: Z / Cs are the stashes of dj_launch_lstm_step_fwd_fused (8-bit gate codes, bf16 c, fragment layout) instead of
// row-major z / c in the operand dtype
int dj_launch_lstm_step_bwd(int dtype, int H, int ntiles, int steps, const void* Z, const void* Uc, const void* Cs,
                            const void* dH, void* dZ, int64_t dz_cts, float* dbias, float* scratch, int sigm, int codes,
                            hipStream_t st);
// dj_elem.hip
int dj_launch_dense_small(const float* A, int M, int K, const float* W, const float* b, float* C, int N, int act_tanh,
                          hipStream_t st);
int dj_launch_dense_small_bwd_x(const float* dC, int M, int N, const float* W, int K, float* dA, int accumulate,
                                hipStream_t st);
int dj_launch_dense_small_bwd_w(const float* A, int M, int K, const float* dC, int N, float* dW, float* db,
                                hipStream_t st);
int dj_launch_dense_small_batch(const DenseBatch* d, int act_tanh, hipStream_t st);
int dj_launch_dense_small_batch_bwd(const DenseBatch* d, float* dA, hipStream_t st);
int dj_launch_bins(const float* notes, float* bins, int B, int T, int N, int octave, DjDrop dn, hipStream_t st);
int dj_launch_feature_xcol(int dtype, const void* fa, void* Xcol, hipStream_t st);
int dj_launch_feature_asm(int dtype, const void* fa, void* X, void* Y, int store_y, hipStream_t st);
int dj_launch_feature_bwd(int dtype, const void* fa, const void* dX, void* Ycol, float* dbc, float* dpre0,
                          hipStream_t st);
int dj_launch_glue_fwd(int dtype, const void* ga, const void* Hin, void* X, hipStream_t st);
int dj_launch_glue_bwd(int dtype, const void* ga, const void* dX, void* dH, float* dpre, hipStream_t st);
int dj_launch_head(int dtype, const void* ha, const void* Hn, void* dH, hipStream_t st);
int dj_launch_cvt_transpose(int dtype, const float* W, int K, int N, void* out, int ld, hipStream_t st);
// B operand of the forward cell GEMM (CellEpi): out[n'][k], n' = 32 G + 8 g + e <-> gate column c = g H + 8 G + e,
// k < D: W[k][c]; D <= k < K1p: 0; K1p <= k < K1p + H: U[k - K1p][c]    (W [D, 4H], U [H, 4H] fp32 Keras layouts)
int dj_launch_pack_wu_gates(int dtype, const float* W, const float* U, int D, int K1p, int H, void* out, hipStream_t st);
int dj_launch_cvt_copy(int dtype, const float* W, int64_t n, void* out, hipStream_t st);
int dj_launch_cvt_to_f32(int dtype, const void* in, int64_t n, float* out, hipStream_t st);
int dj_launch_nadam(float* p, const float* g, float* m, float* v, int64_t n, const void* na, hipStream_t st);
int dj_launch_ta_to_canonical(int dtype, const void* Hin, float* out, int B, int T, int N, int Hd, hipStream_t st);
int dj_launch_canonical_to_na(int dtype, const float* in, void* out, int B, int T, int N, int Hd, hipStream_t st);
// dj_gen.hip
int dj_launch_generate_prep(int G, int T, int N, int Ht, int Hn, int Ln, int S, int SU, const float* P, const int64_t* offs,
                            const float* style_last, int64_t style_stride, float* scratch, hipStream_t st);
int dj_launch_generate_notes(int dtype, int G, int T, int N, int Ht, int Hn, int Ln, int S, int SU, const float* P,
                             const int64_t* offs /* [6 + 5*Ln] */, const void* Htime, const float* style_last,
                             int64_t style_stride, float* scratch, const double* uniforms, const float* temperature,
                             float* next_notes, int* draws_used, void* state, float* results, int sigm,
                             int static_ready, void* wpack, uint32_t kf, hipStream_t st);
// wpack: dj_gen_wpack_bytes() bytes for the bf16 weight fragments of the matrix-core sampler (bf16 mode; null = the
// vector-ALU samplers); packed by the call unless static_ready
int dj_gen_wpack_bytes();
// packs them (a no-op where the matrix-core sampler does not apply: fp32, other widths, DJ_KF_NO_GEN_MFMA)
int dj_launch_generate_pack(int dtype, int Hn, int Ln, const float* P, const int64_t* offs, void* wpack, uint32_t kf,
                            hipStream_t st);
int dj_launch_gen_advance(void* state, const float* results, const float* nsrc, float* ndst, const float* bsrc,
                          float* bdst, int G, int T, int N, int NB, hipStream_t st);
int dj_gen_state_bytes();
