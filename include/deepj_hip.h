/* deepj_hip.h -- C ABI of libdeepj_hip.so: the MI355X (gfx950) implementation of the
 * DeepJ biaxial-LSTM hot path (training step + generation sub-models).
 *
 * The reference (calclavia/music-generator) has no FFI on this path: the boundary is
 * the Keras Model object protocol.  Each entry point below names the reference
 * interface it replaces (paths relative to /root/reference); the Python duck-types
 * in music-generator_amd/ (Model.fit / .predict) bind these through ctypes
 * (see INTEGRATION.md).
 *
 * Conventions: every pointer is a DEVICE pointer unless marked "host"; tensors are
 * dense row-major fp32 in the reference's layouts; no allocation and no implicit
 * synchronisation inside any call; all work is enqueued on `stream` (a hipStream_t
 * passed as void*); re-entrant per (stream, workspace).  Return value: 0 = ok,
 * 1..999 = hipError_t, >= 1000 = argument error.  No exceptions cross the ABI.
 */
#ifndef DEEPJ_HIP_H
#define DEEPJ_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DJ_ABI_VERSION 5
#define DJ_DTYPE_F32 0  /* fp32 operands, v_mfma_f32_32x32x2_f32 (parity mode)            */
#define DJ_DTYPE_BF16 1 /* bf16 operands/stash, fp32 accumulate + cell state (throughput) */

/* Hyper-parameters: reference constants.py:42-77 (defaults there), made runtime so the
 * BASELINE shapes (N=128, B=64) can be expressed. */
typedef struct dj_config {
  int32_t batch;            /* B: samples in this call (constants.py:66 BATCH_SIZE)      */
  int32_t time_steps;       /* T: constants.py:67 SEQ_LEN (1 for note_model.predict)     */
  int32_t num_notes;        /* N: constants.py:56 NUM_NOTES                              */
  int32_t num_styles;       /* constants.py:42                                           */
  int32_t notes_per_bar;    /* constants.py:63 (width of the beat input)                 */
  int32_t octave;           /* constants.py:51                                           */
  int32_t octave_units;     /* constants.py:70 (must be 64)                              */
  int32_t style_units;      /* constants.py:71 (<= 64)                                   */
  int32_t note_units;       /* constants.py:72 (must be 3)                               */
  int32_t time_axis_units;  /* constants.py:73; multiple of 32 in 32..2048: 128 / 256 run */
  int32_t note_axis_units;  /* constants.py:74   the persistent kernels, other widths the */
                            /*   per-step GEMM + gate path (scaled model: 1024)          */
  int32_t time_axis_layers; /* constants.py:76 (1..4)                                    */
  int32_t note_axis_layers; /* constants.py:77 (1..4)                                    */
  int32_t dtype;            /* DJ_DTYPE_*                                                */
  int32_t recurrent_sigmoid;/* 0: Keras-2 hard_sigmoid gates (default), 1: sigmoid       */
  float input_dropout;      /* model.py:128 build_models(input_dropout=0.2)              */
  float dropout;            /* model.py:128 build_models(dropout=0.5)                    */
  int32_t kernel_flags;     /* OR of DJ_KF_* below; 0 = the library's own selection      */
  int32_t fuse_xw_min_tiles;/* > 0: fuse x*W into the recurrent sweep from this many     */
                            /*   sequence tiles on (0 = default rule; tests)             */
} dj_config;

/* Kernel-selection flags (dj_config.kernel_flags: per engine; none changes results beyond summation order).  The
 * DEEPJ_* environment variables of the same meaning (DEEPJ_CLUSTER=0, DEEPJ_CLUSTER_PAIR=0, DEEPJ_CLUSTER_F32=0,
 * DEEPJ_CLUSTER_COOP=0, DEEPJ_FUSE_DX=0, DEEPJ_GEN_KSPLIT=0, DEEPJ_STEP_EPILOGUE=0, DEEPJ_DEBUG_CLUSTER_FAULT=1,
 * DEEPJ_DEBUG_CLUSTER_LATE=1, DEEPJ_FUSE_XW_MIN_TILES=n)
 * are read ONCE, at the first call into the library, as process-wide defaults that are OR-ed with these bits; there
 * is no getenv on the launch path.  dj_env_reload() reads them again (tests that switch kernels inside one process).
 * dj_env_reload() is for single-threaded test set-up: it rewrites the process defaults without synchronisation, so no
 * other thread may be inside the library while it runs.  Every bit is honoured PER ENGINE through dj_config.kernel_flags
 * (the effective flags of a call are passed down to the launchers).
 * A hipGraph captured from these calls keeps the selection in force at capture time.  dj_generate_prepare and
 * dj_generate_step_prepared must see the same flags (the packed-weight layout depends on them): same cfg, and no
 * dj_env_reload() in between. */
#define DJ_KF_NO_CLUSTER 1          /* H = 256 sweeps on the per-tile kernels (set by the host after a cluster fault) */
#define DJ_KF_NO_CLUSTER_PAIR 2     /* inference: time-axis layers one launch each instead of the wavefront pair       */
#define DJ_KF_NO_CLUSTER_F32 4      /* fp32 inference with <= 8 tiles on the per-tile kernel                           */
#define DJ_KF_NO_CLUSTER_COOP 8     /* wavefront pair without the cooperative (4 waves per tile) body                  */
#define DJ_KF_NO_FUSE_DX 16         /* dX = dz W^T always as a GEMM                                                    */
#define DJ_KF_NO_GEN_KSPLIT 32      /* note sampler: one thread per gate column                                       */
#define DJ_KF_DEBUG_CLUSTER_FAULT 64 /* cluster launches fail their placement check (fault-handling tests)            */
#define DJ_KF_DEBUG_CLUSTER_LATE 128 /* the last member of every cluster never arrives in round 0: every other wave's  */
                                     /*   bound runs out, once (tests of the expiry path and its cost)                */
#define DJ_KF_NO_STEP_EPILOGUE 256   /* generic-width layers (scaled model) in bf16: one GEMM + one gate launch per step   */
                                     /*   instead of the cell as the GEMM's epilogue                                      */
#define DJ_KF_COUNTED_EXCHANGE 512   /* H = 256 cluster sweeps (training sweep, cooperative inference pair): members close */
                                     /*   a step through the cluster's counter (store acknowledgement, barrier, atomic,     */
                                     /*   poll) instead of tags inside the h slices                                          */
#define DJ_KF_DEBUG_CLUSTER_MUTE 1024 /* tagged sweep: the last member of every cluster stops publishing at step 2 (tests)  */
#define DJ_KF_BWD_PLAIN 2048          /* bf16 H = 256 BPTT sweep on the round-4 kernel (all gate math behind the product)   */
#define DJ_KF_NO_GEN_MFMA 4096        /* bf16 note sampler on the vector ALUs against the fp32 weights (the fp32 mode's kernel)  */
/* (ABI 3 had two opt-in re-decompositions of the H = 256 BPTT sweep, DJ_KF_BWD_PAIR / _DUAL; they were slower and now
 * live in tools/bwd_decompositions/, outside this library) */
int32_t dj_env_reload(void);

int32_t dj_abi_version(void);
int32_t dj_config_size(void);   /* sizeof(dj_config) of the library: a binding checks its own struct against it */

/* Flat fp32 parameter vector in the reference's layer-creation order with Keras tensor
 * layouts (model.py:128-169).  dj_param_info enumerates the tensors:
 * returns 0 and fills name/offset/shape for 0 <= index < count, 1000 past the end. */
int64_t dj_param_count(const dj_config* cfg);
int32_t dj_param_info(const dj_config* cfg, int32_t index, char* name_host, int32_t name_cap, int64_t* offset_host,
                      int32_t* shape4_host, int32_t* ndim_host);

/* Caller-owned scratch.  dj_workspace_init must run once (and again after the config
 * changes) before the first compute call: it zeroes padding rows the kernels rely on. */
int64_t dj_workspace_bytes(const dj_config* cfg);
int32_t dj_workspace_init(const dj_config* cfg, void* workspace, int64_t workspace_bytes, void* stream);

/* One teacher-forced forward + BPTT pass = the body of Model.fit on one batch
 * (train.py:29 -> model.py:151-152; loss = primary_loss, model.py:14-20).
 * inputs  notes/chosen/target [B,T,N,3], beat [B,T,notes_per_bar], style [B,T,num_styles]
 * outputs grads[param_count] (overwritten), loss[1] (mean loss), out [B,T,N,3] or NULL.
 * Dropout masks are a counter hash of (seed, site, element); seed-identical calls are
 * reproducible and the CPU oracle regenerates the same masks.  dropout == 0 disables. */
int32_t dj_train_fwd_bwd(const dj_config* cfg, const float* params, float* grads, const float* notes,
                         const float* chosen, const float* beat, const float* style, const float* target, float* out,
                         float* loss, void* workspace, int64_t workspace_bytes, uint64_t seed, void* stream);
/* Same, micro-batch form: with accumulate != 0 the gradients of this call are ADDED to `grads` (gradient
 * accumulation over micro-batches, e.g. the scaled model's global batch as 2 x 64 sequences); `loss` is this
 * call's mean.  Scale by 1/micro-batches in dj_nadam_step (grad_scale), as for data-parallel ranks. */
int32_t dj_train_fwd_bwd_acc(const dj_config* cfg, const float* params, float* grads, const float* notes,
                             const float* chosen, const float* beat, const float* style, const float* target,
                             float* out, float* loss, void* workspace, int64_t workspace_bytes, uint64_t seed,
                             int32_t accumulate, void* stream);

/* EXACT micro-batching.  The reference's pitch_bins feature (model.py:43-49) is a raw reshape over the whole batch: a
 * sample's feature depends on the other samples it is evaluated with, so splitting a batch naively changes the model.
 * dj_train_fwd_bwd_mb runs samples [batch_offset, batch_offset + cfg->batch) of a batch of full_batch samples as the
 * reference would inside that batch: the feature reads bins_full [octave, full_batch, T] (dj_pitch_bins on the FULL
 * batch's notes with cfg->batch = full_batch, the same seed and train = 1) and every dropout mask is the full batch's
 * mask of these rows.  Summed over the micro-batches (accumulate != 0 from the second on; grad_scale = cfg->batch /
 * full_batch in dj_nadam_step, loss = mean of the calls' losses for equal sizes) the result is the gradient of ONE step
 * on the full batch -- the scaled model's batch of 128 in two halves through one workspace (tests: against the oracle at
 * the full batch).  bins_full == NULL (then full_batch = batch_offset = 0) is dj_train_fwd_bwd_acc.  Code 1211: bad
 * offsets.  dj_pitch_bins' own batch_offset is 0 for a whole batch; a data-parallel rank that holds samples
 * [batch_offset, batch_offset + cfg->batch) of the GLOBAL batch computes its part of the table with it (the input
 * dropout mask is the global batch's), the parts are all-gathered along the sample axis and every rank then runs
 * dj_train_fwd_bwd_mb against the global table: N ranks = one step of the reference on the global batch, exactly
 * (Model.fit with DEEPJ_DDP_EXACT=1). */
int32_t dj_pitch_bins(const dj_config* cfg_full, const float* notes_full, float* bins_full, uint64_t seed, int32_t train,
                      int32_t batch_offset, void* stream);
int32_t dj_train_fwd_bwd_mb(const dj_config* cfg, const float* params, float* grads, const float* notes,
                            const float* chosen, const float* beat, const float* style, const float* target, float* out,
                            float* loss, void* workspace, int64_t workspace_bytes, uint64_t seed, int32_t accumulate,
                            int32_t full_batch, int32_t batch_offset, const float* bins_full, void* stream);

/* Keras-2 Nadam update (model.py:152 optimizer='nadam'): one fused pass over the flat
 * vectors.  step_t is 1-based; m_schedule_host is read and updated on the host.
 * grad_scale pre-multiplies the gradient (1/world_size after an all-reduce(sum)). */
int32_t dj_nadam_step(float* params, const float* grads, float* m, float* v, int64_t count, int64_t step_t,
                      double* m_schedule_host, float lr, float beta1, float beta2, float epsilon, float schedule_decay,
                      float grad_scale, void* stream);

/* The `style` Dense layer alone (model.py:141-142) on style_in [rows, num_styles] -> out [rows, style_units]:
 * model.get_layer('style') applied to the identity in visualize.py:13-23 (style-embedding export). */
int32_t dj_style_embedding(const dj_config* cfg, const float* params, const float* style_in, int32_t rows, float* out,
                           void* stream);

/* Model.predict of the three Keras models built by build_models (model.py:151,155,167),
 * inference mode (no dropout):
 *   dj_predict            model       [notes, chosen, beat, style] -> out [B,T,N,3]
 *   dj_time_model_predict time_model  [notes, beat, style] -> time_out [B,T,N,time_axis_units]
 *                                      (generate.py:108)
 *   dj_note_model_predict note_model  [features [B,T,N,time_axis_units], chosen, style] -> out [B,T,N,3]
 *                                      (generate.py:114; T = 1 there)
 * target may be passed to dj_predict to also obtain the mean loss (loss may be NULL). */
int32_t dj_predict(const dj_config* cfg, const float* params, const float* notes, const float* chosen,
                   const float* beat, const float* style, const float* target, float* out, float* loss,
                   void* workspace, int64_t workspace_bytes, void* stream);
int32_t dj_time_model_predict(const dj_config* cfg, const float* params, const float* notes, const float* beat,
                              const float* style, float* time_out, void* workspace, int64_t workspace_bytes,
                              void* stream);
int32_t dj_note_model_predict(const dj_config* cfg, const float* params, const float* features, const float* chosen,
                              const float* style, float* out, void* workspace, int64_t workspace_bytes, void* stream);

/* One generated time step for cfg->batch (<= 8) pieces = the loop body of generate()
 * (generate.py:104-118): time_model on the sliding window notes_win [G,T,N,3] / beat_win
 * [G,T,16] / style_win [G,T,S] (stateless, from zero state), then the N notes sampled low
 * to high with the note-axis state carried from note to note (equivalent to the reference's
 * N note_model.predict calls).  uniforms [2*N*G] (float64, device) are consumed in the
 * reference's draw order (note-major, piece-minor, replay draw only after a successful play
 * draw, generate.py:52-58); draws_used [2] (int32): [0] = how many were consumed, [1] = how many of them fell
 * within 1e-5 of the probability they were compared with (decisions that depend on the last digits of p: zero
 * certifies the sampled notes against every model whose probabilities agree to 1e-5).  temperature [G]
 * (apply_temperature, generate.py:81-91).  next_notes [G,N,3] = (play, replay, volume). */
int32_t dj_generate_step(const dj_config* cfg, const float* params, const float* notes_win, const float* beat_win,
                         const float* style_win, const double* uniforms, const float* temperature, float* next_notes,
                         int32_t* draws_used, void* workspace, int64_t workspace_bytes, void* stream);

/* Device-resident variant for hipGraph capture: ALL per-run state lives in HBM -- the sliding
 * windows (ping-pong: *_src is read, *_dst receives the window advanced by one step), the
 * MusicGeneration temperature / silent_time schedule (generate.py:60-79), the running offset
 * into a pre-drawn pool of uniforms, and the emitted notes results[step] ([steps_cap,G,N,3]).
 * One call = one time step, no host round trip, identical launch sequence every step, so the
 * caller can capture two calls (src->dst, dst->src) into a graph and replay it. */
typedef struct dj_gen_state {
  int32_t step;                /* time steps generated so far (index into results)            */
  int32_t draw_off;            /* uniforms consumed so far (index into the pool)              */
  int32_t near_ties;           /* draws so far with |u - p| < 1e-5 (see dj_generate_step)      */
  int32_t first_near_step;     /* time step of the first of them; initialise to -1             */
  double temperature[8];       /* per piece, float64 like the reference                        */
  double default_temp[8];
  int32_t silent[8];           /* silent_time, starts at NOTES_PER_BAR (generate.py:24)        */
} dj_gen_state;
int32_t dj_gen_state_size(void);
int32_t dj_generate_step_resident(const dj_config* cfg, const float* params, void* state, float* results,
                                  const double* uniform_pool, const float* notes_src, float* notes_dst,
                                  const float* beat_src, float* beat_dst, const float* style_win, void* workspace,
                                  int64_t workspace_bytes, void* stream);
/* The part of a generated step that depends on the parameters and the style window only -- weight packing, style
 * embedding and projections (model.py:141-142,77,110-113), the transposed conv kernel, the sampler's style terms -- can
 * be done ONCE per run: dj_generate_prepare leaves it in the workspace, dj_generate_step_prepared is
 * dj_generate_step_resident without that part (same arguments, same results bit for bit).  Contract: the same cfg,
 * params, style_win and workspace as the dj_generate_prepare call, and nothing else has used the workspace in
 * between (call dj_generate_prepare again after any other use; it costs ~70 us). */
int32_t dj_generate_prepare(const dj_config* cfg, const float* params, const float* style_win, void* workspace,
                            int64_t workspace_bytes, void* stream);
int32_t dj_generate_step_prepared(const dj_config* cfg, const float* params, void* state, float* results,
                                  const double* uniform_pool, const float* notes_src, float* notes_dst,
                                  const float* beat_src, float* beat_dst, const float* style_win, void* workspace,
                                  int64_t workspace_bytes, void* stream);

/* ---- single-kernel entry points (unit-tested against the oracle one by one) ---- */

/* C[M,N] = A[M,K] * Bt[N,K]^T + bias[N]; operands in `dtype`.  c_mode: 0 = row-major C in the
 * operand dtype, 1 = row-major fp32, 2 = FRAGMENT-TILED C in the operand dtype (the layout
 * dj_lstm_fwd consumes: 32x32 block (rb, cb) stored as [64 lanes][16 accumulator registers];
 * element (s, c) of the block at ((rb*(N/32) + cb)*64 + 32*((s>>2)&1) + c)*16 + (s&3) + 4*(s>>3);
 * needs M % 32 == 0 and N % 32 == 0), 3 = like 0 but ACCUMULATED, C += A Bt^T + bias (bf16 only: the
 * per-step recurrent product h_{t-1} U added to x_t W + b in place, scaled model).  The x*W products of the Keras LSTM layers
 * (model.py:84,122) and the BPTT input gradient dX = dZ * W^T. */
int32_t dj_gemm_nt(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* Bt,
                   int32_t ldb, void* C, int32_t ldc, int32_t c_mode, const float* bias, void* stream);
/* The same with A COLUMN-TILE-MAJOR (bf16): [K/256][M][256], element (m, k) at A + (k >> 8) * a_tile_stride + m * 256 +
 * (k & 255), a_tile_stride >= M * 256 elements -- the dZ layout the bf16 BPTT sweep writes (dj_lstm_bwd below), so that
 * dX = dZ * W^T reads what the sweep wrote. */
int32_t dj_gemm_nt_tiled_a(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int64_t a_tile_stride,
                           const void* Bt, int32_t ldb, void* C, int32_t ldc, int32_t c_mode, const float* bias,
                           void* stream);
/* C[ka_valid,N] += A[M,Ka]^T * B[M,N] (fp32 atomics).  a_shift = 32 with steps > 0 reads
 * A one recurrence step earlier (zeros at step 0): the recurrent-kernel gradient. */
int32_t dj_gemm_tn(int32_t dtype, int64_t M, int32_t Ka, int32_t ka_valid, int32_t N, const void* A, int32_t lda,
                   const void* B, int32_t ldb, float* C, int32_t ldc, int32_t a_shift, int32_t steps, void* stream);
/* Both weight gradients of one LSTM layer in one pass over dZ [M,N] (N = 4H):
 *   dW[D,N] += X[M,:D]^T dZ   (X row-major [M,DP], DP = D padded to a multiple of 8)
 *   dU[H,N] += Hprev^T dZ     (Hs row-major [M,H]; Hprev = Hs one recurrence step (32 rows)
 *                              earlier within a sequence tile, zero at step 0)
 * `zeros` points at >= 16 zero bytes.  dz_tile_stride: 0 = dZ row-major [M, N]; otherwise dZ is column-tile-major
 * [N/256][M][256] with that many elements between column tiles (bf16 only; see dj_lstm_bwd).
 * TF autodiff of the Keras LSTM kernels (model.py:84,122). */
int32_t dj_lstm_wgrad(int32_t dtype, int64_t M, int32_t steps, const void* X, int32_t DP, int32_t D, const void* Hs,
                      int32_t H, const void* dZ, int32_t N, int64_t dz_tile_stride, float* dW, float* dU,
                      const void* zeros, void* stream);
/* Pack a Keras recurrent_kernel U[H,4H] (fp32) into MFMA B-fragment order for the
 * forward (U) and backward (U^T) recurrences; each output holds H*4H operand elements. */
int32_t dj_lstm_pack(int32_t dtype, int32_t H, const float* U, void* upack_fwd, void* upack_bwd, void* stream);
/* Gate stash: what a forward sweep leaves for BPTT, [rows, 4H] logical, FRAGMENT-TILED like dj_gemm_nt c_mode 2
 * (block (rb, cb) at element ((rb*(4H/32) + cb)*64 + lane)*16).  DJ_DTYPE_F32: the pre-activations z as fp32
 * (BPTT recomputes the activations).  DJ_DTYPE_BF16: the ACTIVATED gates as 8-bit codes, one byte per element --
 * i, f, o: code = clamp(ceil(254 y), 0, 255), decoded (code - 0.5)/254 clamped to [0,1], codes 0 / 255 reserved for
 * the saturated hard_sigmoid so that its derivative mask is exact (computed as the saturating round-to-nearest
 * conversion of 254 y + 0.49997: equal to the clamped ceil except for 254 y within 3e-5 above an integer);
 * g: code = round(127 g) + 128 -- half the bytes of a
 * bf16 z stash on kernels that are HBM-bound on them.  dj_lstm_stash_bytes gives the buffer size for `rows` rows
 * (-1 for an unsupported dtype / H). */
int64_t dj_lstm_stash_bytes(int32_t dtype, int32_t H, int64_t rows);
/* Recurrent sweep over `steps` for ntiles*32 sequences.  Zx (fragment-tiled, operand dtype, see dj_gemm_nt
 * c_mode 2; [ntiles*steps*32, 4H] logical) holds x*W+b; `stash` receives the gate stash (may be NULL; in fp32 it may
 * be Zx itself, in bf16 it must not); h -> Hout (row-major [rows, H]); c -> Cout (fragment-tiled [rows, H], may be
 * NULL).  Row order: ((tile*steps + step)*32 + seq_in_tile). */
int32_t dj_lstm_fwd(int32_t dtype, int32_t H, int32_t ntiles, int32_t steps, const void* Zx, void* stash,
                    const void* upack_fwd, void* Hout, void* Cout, int32_t recurrent_sigmoid, void* stream);
/* The same sweep with the input projection fused in (what the training / predict paths use):
 * z_t = x_t W + h_{t-1} U + b with X row-major [rows, DP] (D valid columns), W packed by
 * dj_lstm_pack_w (at most 4H * (roundup(D, 8) + 128) operand elements),
 * bias[4H] fp32.  `stash` (may be NULL) receives the gate stash.  The bf16 build for H = 128 keeps U (and W for
 * D <= 128) in registers.  cluster_scratch: NULL, or dj_lstm_cluster_scratch_bytes() bytes (128-byte aligned,
 * zero-initialised once) that enable the weight-stationary cluster kernel below. */
int32_t dj_lstm_pack_w(int32_t dtype, int32_t H, const float* W, int32_t D, void* wpack, void* stream);
int32_t dj_lstm_fwd_fused(int32_t dtype, int32_t H, int32_t ntiles, int32_t steps, const void* X, int32_t DP,
                          int32_t D, const void* wpack, const float* bias, void* stash, const void* upack_fwd,
                          void* Hout, void* Cout, int32_t recurrent_sigmoid, void* cluster_scratch, void* stream);
/* BPTT sweep: Z / C = the forward's gate stash / fragment-tiled cell states; dH = dL/dh
 * per step (row-major [rows, H]); dZ receives dL/dz: row-major [rows, 4H] when dz_tile_stride == 0, else
 * column-tile-major [4H/256][rows][256] with dz_tile_stride (>= rows * 256) elements between column tiles -- there
 * the 32 rows of a step form one contiguous 16 KiB block per tile, which the weight-gradient GEMM streams per stage
 * (with row-major dZ its 512-byte pieces at 2 KiB stride were fetched from HBM once per row tile: 1.6x the bytes);
 * dbias[4H] += column sums of dz. */
int32_t dj_lstm_bwd(int32_t dtype, int32_t H, int32_t ntiles, int32_t steps, const void* Z, const void* upack_bwd,
                    const void* C, const void* dH, void* dZ, int64_t dz_tile_stride, float* dbias,
                    int32_t recurrent_sigmoid, void* stream);
/* Same sweep that also produces the layer's input gradient dX = dz W^T (row-major [rows, DP], columns
 * D..DP-1 zero) from the dz tile it holds in LDS; W is the Keras kernel [D, 4H], wtpack its fragment stream
 * from dj_lstm_pack_wt (roundup(D, 32) * 4H operand elements).  Replaces one dj_gemm_nt pass over dZ per layer
 * (TF autodiff's dX of model.py:84,122).  Offered by the bf16 H = 128 build: for D <= 128 the
 * whole dX; for D = 256k + 1..4 (note layer 0: 259) only the last 32-column block (4 columns stored), the rest being
 * left to dj_gemm_nt; code 1015 otherwise. */
int32_t dj_lstm_pack_wt(int32_t dtype, int32_t H, const float* W, int32_t D, void* wtpack, void* stream);
int32_t dj_lstm_bwd_dx(int32_t dtype, int32_t H, int32_t ntiles, int32_t steps, const void* Z, const void* upack_bwd,
                       const void* C, const void* dH, void* dZ, int64_t dz_tile_stride, float* dbias,
                       int32_t recurrent_sigmoid, const void* wtpack, int32_t D, void* dX, int32_t DP, void* stream);
/* The bf16 H = 256 forward sweep (any tile count; in inference with <= 64 tiles both time-axis layers in one wavefront
 * launch, with <= 8 tiles four waves per tile) and the fp32 H = 256 inference sweep of <= 8 tiles (dj_predict /
 * dj_time_model_predict / dj_generate_* in the parity mode) run as weight-stationary clusters of 8 workgroups that meet
 * once per step through a counter in L2 (dj_lstm.hip).  Their exchange state (counters, the members' XCC ids, the h
 * slices) lives in a caller-owned scratch -- part of the workspace for the dj_train / dj_predict calls, one per
 * workspace and therefore per engine / stream; concurrent sweeps must not share one.  Two things are checked at run
 * time and can never produce a silent wrong answer or a hung device: a member that never arrives (grid not
 * co-resident, e.g. the device shared with another stream's kernels) lets the bounded wait run out; a cluster whose
 * members report different hardware XCC ids (dispatch not round-robin over the XCDs, which the exchange through one
 * XCD's L2 relies on) is detected in round 0.  In both cases the affected tiles carry NaN from there on (so does the
 * loss) and the event is counted.  The bound counts POLLS (2^19 of them, ~100 ms of actual polling), not elapsed
 * time: time during which the whole queue is off the device (another process's time slice, a driver-side eviction)
 * does not count against it.  The first wave whose bound runs out releases every other waiter of its cluster (a poison
 * bit in the counter) and no poisoned wave waits again, so a faulted launch costs one bound, not one per step.
 * dj_lstm_cluster_faults / dj_workspace_cluster_faults return the number of events recorded in that scratch / workspace
 * since the previous call (they drain `stream`; 0 in a healthy run, -1 on a HIP error).  DJ_KF_NO_CLUSTER
 * (DEEPJ_CLUSTER=0) selects the per-tile kernels instead (DJ_KF_NO_CLUSTER_PAIR / _COOP / _F32 the individual forms);
 * DJ_KF_DEBUG_CLUSTER_FAULT injects a placement fault (tests). */
int64_t dj_lstm_cluster_scratch_bytes(void);
int32_t dj_lstm_cluster_faults(void* cluster_scratch, void* stream);
int32_t dj_workspace_cluster_faults(const dj_config* cfg, void* workspace, int64_t workspace_bytes, void* stream);
/* Diagnostics: the fault line of the workspace as it stands once `stream` has drained, without resetting anything --
 * DJ_FAULT_REPORT_WORDS int32 (host):
 *   [0] expired waits (waves whose bound ran out)   [1] workgroups whose cluster sat on several XCDs   [2] test hook
 *   [4] waits that saw two consecutive polls > 2^20 shader cycles apart, [5] the longest poll-to-poll gap seen in units
 *       of 1024 cycles -- the STALL CENSUS: cumulative over the life of the workspace, filled in healthy runs too; a
 *       gap of milliseconds between two polls of a wave means the wave was off the device
 *   [6] polls at which the shader clock read LOWER than at the poll before (the clock is per XCC: the wave was saved
 *       and restored on another one in the middle of a wait); such differences never enter [4] / [5] (ABI 5)
 *   [8] != 0: [9..17] describe the first expired wait since the last census: [9] kind << 24 | cluster << 12 |
 *       member << 8 | wave (bit 28: the wait was for the producing layer's counter), [10] recurrence step (-1: round
 *       0), [11] counter value seen last, [12] target, [13] polls made, [14..15] shader cycles between first and last
 *       poll (lo, hi), [16] longest poll-to-poll gap in cycles, [17] hardware XCC id + 1.  Kind 4 / 7 = a wait of the
 *       training sweep / the cooperative inference pair for TAGGED h fragments (the members' slices announce themselves, no counter): [11] = fragments
 *       that had arrived, [12] = 16. */
#define DJ_FAULT_REPORT_WORDS 32
int32_t dj_workspace_cluster_fault_report(const dj_config* cfg, void* workspace, int64_t workspace_bytes,
                                          int32_t* words_host, void* stream);
/* Census and report in ONE blocking round trip (ABI 5): words_host receives the fault line as dj_workspace_cluster_fault_report
 * returns it, the counts and the description are reset by a kernel queued on `stream` behind the copy (nothing waits
 * for it), and the number of events is returned (0 in a healthy run, -1 on a HIP or argument error).  What
 * engine.Engine.cluster_faults() calls after every predict / generation chunk (Model.predict of generate.py:108,114). */
int32_t dj_workspace_cluster_faults_take(const dj_config* cfg, void* workspace, int64_t workspace_bytes,
                                         int32_t* words_host, void* stream);
/* The same census without a host round trip: a one-thread kernel on `stream` ADDS (as floats) the event count to
 * out_dev[0], the expired waits to out_dev[1] and the misplaced workgroups to out_dev[2], and resets the counts (not
 * the description of the first expired wait) -- put out_dev next to the loss and one device-to-host copy per training
 * step carries all of it (Model.fit reads [loss, faults, ...] together before the optimizer step; the caller zeroes
 * out_dev). */
int32_t dj_workspace_faults_async(const dj_config* cfg, void* workspace, int64_t workspace_bytes, float* out_dev,
                                  void* stream);
/* mask[rows, cols] (fp32 0 or 1/(1-p)) of dropout site `site` -- exposes the counter
 * hash so tests can pin it against the oracle. */
int32_t dj_dropout_mask(uint64_t seed, int32_t site, float p, int64_t rows, int32_t cols, float* mask, void* stream);

/* ---- live kernel timing (bench.py roofline): when enabled, HIP events are recorded on
 * the caller's stream around every launch, grouped by category.  dj_profile_read waits
 * for the category's events and returns the summed milliseconds and the scope count.
 * Process-global and not thread-safe: a measurement aid, not part of the data path. */
int32_t dj_profile_enable(int32_t on);   /* 0 = off, 1 = every category, 2 + c = category c only (2 events per launch
                                          * cost ~0.2 ms per training step when every launch carries them); clears */
int32_t dj_profile_category_count(void);
const char* dj_profile_category_name(int32_t category);
int32_t dj_profile_read(int32_t category, double* total_ms_host, int64_t* scopes_host);

#ifdef __cplusplus
}
#endif
#endif
