// C ABI of libdeepj_hip.so (include/deepj_hip.h): workspace planning and the kernel
// sequence of one training step / one predict call.  Host code only.
//
// Reference graph being executed: model.py:128-169 (build_models), loss model.py:14-20,
// optimizer model.py:152; driven in the reference by Model.fit (train.py:29) and
// Model.predict (generate.py:108,114).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/deepj_hip.h"
#include "dj_kernels.h"

namespace {

constexpr int MAXL = 4;

// ---- optional per-category HIP-event timing (dj_profile_*): events are recorded on the
// caller's stream around each launch, so bench.py can report live kernel durations.
enum ProfCat {
  PC_PREP, PC_STYLE_FWD, PC_FEATURE_FWD, PC_GLUE_FWD, PC_GEMM_XW, PC_LSTM_FWD_TIME, PC_LSTM_FWD_NOTE, PC_HEAD,
  PC_LSTM_BWD_TIME, PC_LSTM_BWD_NOTE, PC_GEMM_DW, PC_GEMM_DX, PC_GLUE_BWD, PC_FEATURE_BWD, PC_STYLE_BWD, PC_NADAM,
  PC_COUNT
};
const char* const kProfNames[PC_COUNT] = {
    "prep_weights", "style_fwd", "feature_fwd", "glue_fwd", "gemm_xw", "lstm_fwd_time", "lstm_fwd_note", "head_loss",
    "lstm_bwd_time", "lstm_bwd_note", "gemm_dw", "gemm_dx", "glue_bwd", "feature_bwd", "style_bwd", "nadam"};
struct ProfRec { int cat; hipEvent_t a, b; };
struct Prof {
  bool on = false;
  int only = -1;          // >= 0: events around the launches of this category only (dj_profile_enable(2 + category))
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
  }
} g_prof;
struct ProfScope {
  bool act; hipStream_t st; hipEvent_t b;
  ProfScope(int cat, hipStream_t s) : act(g_prof.on && (g_prof.only < 0 || g_prof.only == cat)), st(s) {
    if (!act) return;
    hipEvent_t a = g_prof.get(); b = g_prof.get();
    (void)hipEventRecord(a, st);
    g_prof.recs.push_back({cat, a, b});
  }
  ~ProfScope() { if (act) (void)hipEventRecord(b, st); }
};

struct LstmP {          // parameter offsets (floats) of one LSTM layer + its style Dense
  int64_t dW, db;       // style Dense kernel [SU, D], bias [D]
  int64_t W, U, b;      // LSTM kernel [D,4H], recurrent_kernel [H,4H], bias [4H]
  int D, DP, H;
  int64_t tiles;        // sequence tiles of this layer's axis (32 sequences each)
  int dtype;            // operand dtype of the plan (DJ_F32 / DJ_BF16)
};

struct Plan {
  dj_config c;
  int B, T, N, S, NB, SU, F, FP, Ht, Hn, Lt, Ln, esz;
  int64_t BT, seqT, tilesT, Mt, seqN, tilesN, Mn;
  // parameter offsets
  int64_t p_style_W, p_style_b, p_conv_W, p_conv_b, p_nd_W, p_nd_b, p_vd_W, p_vd_b, nparams;
  LstmP tl[MAXL], nl[MAXL];
  // workspace offsets (bytes)
  int64_t w_style, w_dstyle, w_bins, w_sp_t[MAXL], w_sp_n[MAXL], w_dpre_t[MAXL], w_dpre_n[MAXL];
  int64_t w_Wt_t[MAXL], w_Wc_t[MAXL], w_Uf_t[MAXL], w_Ub_t[MAXL], w_Wp_t[MAXL];
  int64_t w_Wt_n[MAXL], w_Wc_n[MAXL], w_Uf_n[MAXL], w_Ub_n[MAXL], w_Wp_n[MAXL];
  int64_t w_X_t[MAXL], w_Z_t[MAXL], w_H_t[MAXL], w_C_t[MAXL];
  int64_t w_X_n[MAXL], w_Z_n[MAXL], w_H_n[MAXL], w_C_n[MAXL];
  int64_t w_dH_t, w_dX_t, w_dH_n, w_dX_n, w_featin, w_dZ_t, w_dZ_n, w_zero, w_Xcol, w_Ycol, w_WcT, w_step, w_cluster;
  int64_t ws_bytes;
};

inline int64_t up8(int64_t v) { return (v + 7) / 8 * 8; }
// The persistent one-CU-per-sequence-tile recurrent kernels (dj_lstm.hip) exist for the reference's
// layer widths; any other width runs the per-step GEMM + gate path (dj_step.hip).
inline bool rec_persistent(int H) { return H == 128 || H == 256; }
// gate stash of one layer, bytes per row: the persistent kernels keep z (fp32) or 8-bit activated gates (bf16),
// the per-step path (dj_step.hip) keeps z in the operand dtype
inline int64_t stash_row_bytes(const Plan& p, int H) {
  return rec_persistent(H) ? dj_lstm_stash_row_bytes(p.c.dtype, H) : (int64_t)4 * H * p.esz;
}


int make_plan(const dj_config* cfg, Plan& p) {
  if (!cfg) return 1100;
  memset(&p, 0, sizeof(p));
  p.c = *cfg;
  p.B = cfg->batch; p.T = cfg->time_steps; p.N = cfg->num_notes; p.S = cfg->num_styles;
  p.NB = cfg->notes_per_bar; p.SU = cfg->style_units; p.Ht = cfg->time_axis_units; p.Hn = cfg->note_axis_units;
  p.Lt = cfg->time_axis_layers; p.Ln = cfg->note_axis_layers;
  if (p.B < 1 || p.T < 1 || p.N < 1 || p.S < 1 || p.NB < 1) return 1101;
  if (cfg->octave_units != 64 || cfg->note_units != 3 || cfg->octave < 1) return 1102;
  if (p.SU < 1 || p.SU > 64) return 1103;
  if (p.Ht < 32 || p.Ht > 2048 || (p.Ht % 32) || p.Hn < 32 || p.Hn > 2048 || (p.Hn % 32)) return 1104;
  if (p.Lt < 1 || p.Lt > MAXL || p.Ln < 1 || p.Ln > MAXL) return 1105;
  if (cfg->dtype != DJ_F32 && cfg->dtype != DJ_BF16) return 1106;
  if (cfg->input_dropout < 0 || cfg->input_dropout >= 1 || cfg->dropout < 0 || cfg->dropout >= 1) return 1107;
  p.esz = cfg->dtype == DJ_F32 ? 4 : 2;
  p.F = 1 + cfg->octave + 1 + cfg->octave_units + p.NB;    // model.py:61-67
  p.FP = (int)up8(p.F);
  if (p.FP - 64 > 64 || p.F > 128) return 1108;
  p.BT = (int64_t)p.B * p.T;
  p.seqT = (int64_t)p.B * p.N; p.tilesT = (p.seqT + 31) / 32; p.Mt = p.tilesT * 32 * p.T;
  p.seqN = p.BT;               p.tilesN = (p.seqN + 31) / 32; p.Mn = p.tilesN * 32 * p.N;
  if ((int64_t)p.BT * p.N * 264 >= (1LL << 32)) return 1109;   // dropout hash uses 32-bit rows
  if (p.Mt * 4 * p.Ht >= (1LL << 40) || p.Mt > (1LL << 31) - 256 || p.Mn > (1LL << 31) - 256) return 1110;

  // ---- parameters, reference creation order (model.py:128-169)
  int64_t o = 0;
  auto take = [&](int64_t n) { int64_t r = o; o += n; return r; };
  p.p_style_W = take((int64_t)p.S * p.SU); p.p_style_b = take(p.SU);
  p.p_conv_W = take(24 * 3 * 64); p.p_conv_b = take(64);
  for (int l = 0; l < p.Lt; ++l) {
    LstmP& L = p.tl[l];
    L.D = l == 0 ? p.F : p.Ht; L.DP = (int)up8(L.D); L.H = p.Ht; L.tiles = p.tilesT; L.dtype = cfg->dtype;
    L.dW = take((int64_t)p.SU * L.D); L.db = take(L.D);
    L.W = take((int64_t)L.D * 4 * L.H); L.U = take((int64_t)L.H * 4 * L.H); L.b = take(4 * L.H);
  }
  for (int l = 0; l < p.Ln; ++l) {
    LstmP& L = p.nl[l];
    L.D = l == 0 ? p.Ht + 3 : p.Hn; L.DP = (int)up8(L.D); L.H = p.Hn; L.tiles = p.tilesN; L.dtype = cfg->dtype;
    L.dW = take((int64_t)p.SU * L.D); L.db = take(L.D);
    L.W = take((int64_t)L.D * 4 * L.H); L.U = take((int64_t)L.H * 4 * L.H); L.b = take(4 * L.H);
  }
  p.p_nd_W = take(p.Hn * 2); p.p_nd_b = take(2); p.p_vd_W = take(p.Hn); p.p_vd_b = take(1);
  p.nparams = o;

  // ---- workspace
  int64_t w = 0;
  auto wtake = [&](int64_t bytes) { int64_t r = w; w += (bytes + 255) / 256 * 256; return r; };
  p.w_style = wtake(p.BT * p.SU * 4); p.w_dstyle = wtake(p.BT * p.SU * 4);
  p.w_bins = wtake((int64_t)cfg->octave * p.BT * 4 + 256);   // + loss scratch not needed; pad only
  int maxDPt = 0, maxDPn = 0;
  for (int l = 0; l < p.Lt; ++l) {
    const LstmP& L = p.tl[l];
    p.w_sp_t[l] = wtake(p.BT * L.D * 4); p.w_dpre_t[l] = wtake(p.BT * L.D * 4);
    // (generic-width layers: room for the forward cell GEMM's operand [4H][K1p + H], dj_launch_pack_wu_gates)
    p.w_Wt_t[l] = wtake((int64_t)4 * L.H * (L.DP + 128 + (rec_persistent(L.H) ? 0 : L.H)) * p.esz); p.w_Wc_t[l] = wtake((int64_t)L.D * 4 * L.H * p.esz);
    p.w_Uf_t[l] = wtake((int64_t)L.H * 4 * L.H * p.esz); p.w_Ub_t[l] = wtake((int64_t)L.H * 4 * L.H * p.esz);
    p.w_Wp_t[l] = wtake((int64_t)(L.D + 31) / 32 * 32 * 4 * L.H * p.esz);
    p.w_X_t[l] = wtake(p.Mt * L.DP * p.esz); p.w_Z_t[l] = wtake(p.Mt * stash_row_bytes(p, L.H));
    p.w_H_t[l] = wtake(p.Mt * L.H * p.esz); p.w_C_t[l] = wtake(p.Mt * L.H * p.esz);
    if (L.DP > maxDPt) maxDPt = L.DP;
  }
  for (int l = 0; l < p.Ln; ++l) {
    const LstmP& L = p.nl[l];
    p.w_sp_n[l] = wtake(p.BT * L.D * 4); p.w_dpre_n[l] = wtake(p.BT * L.D * 4);
    p.w_Wt_n[l] = wtake((int64_t)4 * L.H * (L.DP + 128 + (rec_persistent(L.H) ? 0 : L.H)) * p.esz); p.w_Wc_n[l] = wtake((int64_t)L.D * 4 * L.H * p.esz);
    p.w_Uf_n[l] = wtake((int64_t)L.H * 4 * L.H * p.esz); p.w_Ub_n[l] = wtake((int64_t)L.H * 4 * L.H * p.esz);
    p.w_Wp_n[l] = wtake((int64_t)(L.D + 31) / 32 * 32 * 4 * L.H * p.esz);
    p.w_X_n[l] = wtake(p.Mn * L.DP * p.esz); p.w_Z_n[l] = wtake(p.Mn * stash_row_bytes(p, L.H));
    p.w_H_n[l] = wtake(p.Mn * L.H * p.esz); p.w_C_n[l] = wtake(p.Mn * L.H * p.esz);
    if (L.DP > maxDPn) maxDPn = L.DP;
  }
  p.w_dH_t = wtake(p.Mt * p.Ht * p.esz); p.w_dX_t = wtake(p.Mt * maxDPt * p.esz);
  p.w_dH_n = wtake(p.Mn * p.Hn * p.esz); p.w_dX_n = wtake(p.Mn * maxDPn * p.esz);
  p.w_featin = wtake(p.Mn * p.Ht * p.esz);   // note_model.predict: features in NA order
  p.w_dZ_t = wtake(p.Mt * 4 * p.Ht * p.esz);  // row-major dz of the layer in flight (BPTT)
  p.w_dZ_n = wtake(p.Mn * 4 * p.Hn * p.esz);
  p.w_Xcol = wtake(p.Mt * 80 * p.esz);       // im2col view of the (dropped-out) notes, conv-kernel tap order
  p.w_Ycol = wtake(p.Mt * 64 * p.esz);       // conv pre-activation -> tanh(conv) stash -> its gradient in BPTT
  p.w_WcT = wtake(64 * 80 * p.esz);         // conv kernel as the k-contiguous Bt operand [64 outputs][80 taps]
  p.w_zero = wtake(256);                      // a zero line (h_{-1} rows of the fused weight-gradient GEMM)
  // exchange state of the weight-stationary cluster forward kernel (bf16, H = 256): per workspace, i.e. per engine
  p.w_cluster = wtake((p.Ht == 256 || p.Hn == 256) ? dj_lstm_cluster_scratch_bytes_impl() : 0);   // bf16 sweeps, fp32 inference
  {                                           // fp32 scratch of the per-step path (layers with H not 128/256)
    int64_t fl = 0;
    if (!rec_persistent(p.Ht)) fl = dj_lstm_step_scratch_floats(p.Ht, p.tilesT);
    if (!rec_persistent(p.Hn) && dj_lstm_step_scratch_floats(p.Hn, p.tilesN) > fl)
      fl = dj_lstm_step_scratch_floats(p.Hn, p.tilesN);
    p.w_step = wtake(fl * 4);
  }
  p.ws_bytes = w;
  return 0;
}

DjDrop mkdrop(uint64_t seed, int site, float prob, bool train, uint32_t row0 = 0) {
  DjDrop d;
  d.key = dj_dropkey(seed, (uint32_t)site);
  d.thr = (train && prob > 0.f) ? (uint32_t)ceil((double)prob * 65536.0) : 0u;
  d.scale = (float)(1.0 / (1.0 - (double)prob));
  d.row0 = row0;
  return d;
}

struct Ctx {
  const Plan& p;
  const float* P;
  char* ws;
  hipStream_t st;
  bool train;
  uint64_t seed;
  // generation with dj_generate_prepare done: packed weights, style embedding / projections and the transposed conv
  // kernel in the workspace are current (they depend on the parameters and the style vector only)
  bool static_ready = false;
  // micro-batch of a larger batch (dj_train_fwd_bwd_mb): this call holds samples [b0, b0 + B) of a batch of Bfull; the
  // dropout masks are those of the full batch's rows and pitch_bins (model.py:43-49, a reshape over the WHOLE batch)
  // reads the full batch's table bins_ext [octave, Bfull, T]
  int b0 = 0, Bfull = 0;
  const float* bins_ext = nullptr;
  template <typename X = void> X* at(int64_t off) const { return (X*)(ws + off); }
};
// dropout site of this call: per-note sites index rows (b T + t) N + n, the beat site rows b T + t (oracle make_masks)
DjDrop mkdrop(const Ctx& c, int site, float prob, bool train) {
  const int64_t bt0 = (int64_t)c.b0 * c.p.T;
  return mkdrop(c.seed, site, prob, train, (uint32_t)(site == DJ_SITE_BEAT ? bt0 : bt0 * c.p.N));
}

#define RUN(x)                 \
  do {                         \
    int rc_ = (x);             \
    if (rc_) return rc_;       \
  } while (0)

// ---- kernel-selection switches.  The DEEPJ_* environment variables are read ONCE (first use of the library, or
// dj_env_reload()) into process defaults; per-engine choices travel in dj_config.kernel_flags / fuse_xw_min_tiles.
// No getenv on the launch path, and nothing the host side does to one engine (the cluster-fault fallback) leaks into
// another engine or into a captured graph's view of the world.
struct EnvDefaults { uint32_t flags; int64_t fuse_xw_min_tiles; };
EnvDefaults read_env() {
  EnvDefaults d{0u, -1};
  auto off = [](const char* name) { const char* e = getenv(name); return e && e[0] == '0'; };
  auto on = [](const char* name) { const char* e = getenv(name); return e && e[0] != '0' && e[0] != 0; };
  if (off("DEEPJ_CLUSTER")) d.flags |= DJ_KF_NO_CLUSTER;
  if (off("DEEPJ_CLUSTER_PAIR")) d.flags |= DJ_KF_NO_CLUSTER_PAIR;
  if (off("DEEPJ_CLUSTER_F32")) d.flags |= DJ_KF_NO_CLUSTER_F32;
  if (off("DEEPJ_CLUSTER_COOP")) d.flags |= DJ_KF_NO_CLUSTER_COOP;
  if (off("DEEPJ_FUSE_DX")) d.flags |= DJ_KF_NO_FUSE_DX;
  if (off("DEEPJ_GEN_KSPLIT")) d.flags |= DJ_KF_NO_GEN_KSPLIT;
  if (off("DEEPJ_GEN_MFMA")) d.flags |= DJ_KF_NO_GEN_MFMA;
  if (off("DEEPJ_STEP_EPILOGUE")) d.flags |= DJ_KF_NO_STEP_EPILOGUE;
  if (off("DEEPJ_TAGGED_EXCHANGE")) d.flags |= DJ_KF_COUNTED_EXCHANGE;
  if (off("DEEPJ_BWD_SPLIT")) d.flags |= DJ_KF_BWD_PLAIN;
  if (on("DEEPJ_DEBUG_CLUSTER_FAULT")) d.flags |= DJ_KF_DEBUG_CLUSTER_FAULT;
  if (on("DEEPJ_DEBUG_CLUSTER_LATE")) d.flags |= DJ_KF_DEBUG_CLUSTER_LATE;
  if (on("DEEPJ_DEBUG_CLUSTER_MUTE")) d.flags |= DJ_KF_DEBUG_CLUSTER_MUTE;
  if (const char* e = getenv("DEEPJ_FUSE_XW_MIN_TILES")) d.fuse_xw_min_tiles = atoll(e);
  return d;
}
EnvDefaults& env_defaults() {
  static EnvDefaults d = read_env();
  return d;
}
inline uint32_t kflags(const dj_config& c) { return (uint32_t)c.kernel_flags | env_defaults().flags; }

// generic-width layers in bf16: the LSTM cell runs as the epilogue of each step's recurrent GEMM (dj_step.hip, CellEpi)
inline bool step_epilogue(const dj_config& c) { return c.dtype == DJ_BF16 && !(kflags(c) & DJ_KF_NO_STEP_EPILOGUE); }
// the weight-stationary cluster kernels may be used by this plan (host side: cleared per engine after a cluster fault)
inline bool cluster_enabled(const dj_config& c) { return !(kflags(c) & DJ_KF_NO_CLUSTER); }

// Fuse x*W into the recurrent kernel when its extra L2 weight stream (D x 4H) is no larger than
// twice the recurrent one (H x 4H); wider inputs (note layer 0: D = 259 vs H = 128) are cheaper as
// a separate GEMM (measured: fused +1.25 ms on the note axis vs 1.24 ms of GEMMs saved).  With few
// sequence tiles (generation: 5) the chip is idle anyway and the recurrence is a pure latency chain:
// there the projection stays a separate (parallel) GEMM and the chain carries h*U only.
// dj_config.fuse_xw_min_tiles / DEEPJ_FUSE_XW_MIN_TILES override the tile threshold (tests run the fused kernel on
// small shapes with it).
inline bool fuse_xw(const dj_config& c, const LstmP& L) {
  const bool forced = c.fuse_xw_min_tiles > 0 || env_defaults().fuse_xw_min_tiles >= 0;
  const int64_t min_tiles = c.fuse_xw_min_tiles > 0 ? c.fuse_xw_min_tiles
                                                    : (env_defaults().fuse_xw_min_tiles >= 0 ? env_defaults().fuse_xw_min_tiles : 128);
  // H = 128 in bf16 keeps U (and W up to H columns) in registers: also the 259-wide note layer 0 is cheaper fused
  const int dmax = (L.H == 128 && L.dtype == DJ_BF16) ? 288 : 2 * L.H;
  if (!rec_persistent(L.H) || L.D > dmax) return false;
  // bf16 H = 256 with the weight-stationary cluster kernel: W and U never leave LDS, so the fused sweep also wins
  // the latency chain of a few tiles (generation: 5 tiles x 128 steps streamed 1 MB of weights per step before)
  if (!forced && L.H == 256 && L.dtype == DJ_BF16 && L.DP <= 256 && cluster_enabled(c)) return true;
  return L.tiles >= min_tiles;
}

// Where the BPTT kernel offers it (bf16, H = 128, D <= H: U^T and W^T both stationary in registers) it also
// produces dX = dz W^T, which replaces one GEMM pass over dZ for that layer.  DJ_KF_NO_FUSE_DX keeps the GEMM.
inline int fuse_dx(const dj_config& c, const LstmP& L) {            // 0: GEMM, 1: whole dX in the kernel, 2: last column block only
  if (!rec_persistent(L.H) || (kflags(c) & DJ_KF_NO_FUSE_DX)) return 0;
  return dj_lstm_bwd_has_dx(L.dtype, L.H, L.D);
}

// weight conversion/packing for one LSTM layer
int prep_layer(const Ctx& c, const LstmP& L, int64_t wWt, int64_t wWc, int64_t wUf, int64_t wUb, int64_t wWp,
               bool need_bwd) {
  const int dt = c.p.c.dtype;
  ProfScope ps(PC_PREP, c.st);
  const bool cell = !rec_persistent(L.H) && step_epilogue(c.p.c);
  if (cell)         // [W ; U]^T with gate-interleaved rows: ONE operand for z_t = [x_t | h_{t-1}] [W ; U] (dj_step.hip)
    RUN(dj_launch_pack_wu_gates(dt, c.P + L.W, c.P + L.U, L.D, dj_step_k1p(L.DP), L.H, c.at(wWt), c.st));
  else if (fuse_xw(c.p.c, L))   // input kernel W as MFMA B fragments for the fused x*W inside the recurrent kernel
    RUN(dj_launch_lstm_pack_w(dt, L.H, c.P + L.W, L.D, dj_lstm_fused_nkx(dt, L.H, L.D), c.at(wWt), c.st));
  else              // k-contiguous Bt operand of the separate x*W GEMM
    RUN(dj_launch_cvt_transpose(dt, c.P + L.W, L.D, 4 * L.H, c.at(wWt), L.DP, c.st));
  if (rec_persistent(L.H)) {
    RUN(dj_launch_lstm_pack(dt, L.H, c.P + L.U, c.at(wUf), need_bwd ? c.at(wUb) : nullptr, c.st));
  } else {          // per-step path: U^T [4H, H] for r = h U, and U [H, 4H] in the operand dtype for dh = dz U^T
    if (!cell) RUN(dj_launch_cvt_transpose(dt, c.P + L.U, L.H, 4 * L.H, c.at(wUf), L.H, c.st));
    if (need_bwd && dt != DJ_F32) RUN(dj_launch_cvt_copy(dt, c.P + L.U, (int64_t)L.H * 4 * L.H, c.at(wUb), c.st));
  }
  if (need_bwd && fuse_dx(c.p.c, L)) RUN(dj_launch_lstm_pack_wt(dt, L.H, c.P + L.W, L.D, c.at(wWp), c.st));
  if (need_bwd && fuse_dx(c.p.c, L) != 1 && dt != DJ_F32)
    RUN(dj_launch_cvt_copy(dt, c.P + L.W, (int64_t)L.D * 4 * L.H, c.at(wWc), c.st));
  return 0;
}

int style_forward(const Ctx& c, const float* style_in) {
  const Plan& p = c.p;
  ProfScope ps(PC_STYLE_FWD, c.st);
  RUN(dj_launch_dense_small(style_in, (int)p.BT, p.S, c.P + p.p_style_W, c.P + p.p_style_b, c.at<float>(p.w_style),
                            p.SU, 0, c.st));                                            // model.py:141-142
  return 0;
}
// sp_l = tanh(style Wd_l + bd_l) of all layers of one axis in one launch      (model.py:77,110-113 + tanh)
int style_proj_all(const Ctx& c, const LstmP* Ls, const int64_t* w_sp, int n) {
  ProfScope ps(PC_STYLE_FWD, c.st);
  DenseBatch d;
  memset(&d, 0, sizeof(d));
  d.n = n; d.M = (int)c.p.BT; d.K = c.p.SU; d.A = c.at<float>(c.p.w_style);
  for (int l = 0; l < n; ++l) {
    d.W[l] = c.P + Ls[l].dW; d.b[l] = c.P + Ls[l].db; d.C[l] = c.at<float>(w_sp[l]); d.N[l] = Ls[l].D;
  }
  return dj_launch_dense_small_batch(&d, 1, c.st);
}

int lstm_layer_fwd(const Ctx& c, const LstmP& L, int64_t tiles, int steps, int64_t M, int64_t wX, int64_t wWt,
                   int64_t wUf, int64_t wZ, int64_t wZx, int64_t wH, int64_t wC, bool is_note) {
  const int dt = c.p.c.dtype;
  if (fuse_xw(c.p.c, L)) {
    // z = x W + h U + b in one persistent kernel (Z receives the gate stash when training)
    ProfScope ps(is_note ? PC_LSTM_FWD_NOTE : PC_LSTM_FWD_TIME, c.st);
    RUN(dj_launch_lstm_fwd_fused(dt, L.H, (int)tiles, steps, c.at(wX), L.DP, dj_lstm_fused_nkx(dt, L.H, L.D), c.at(wWt),
                                 c.P + L.b, c.train ? c.at(wZ) : nullptr, c.at(wUf), c.at(wH),
                                 c.train ? c.at(wC) : nullptr, c.p.c.recurrent_sigmoid,
                                 cluster_enabled(c.p.c) ? c.at(c.p.w_cluster) : nullptr, kflags(c.p.c), c.st));
    return 0;
  }
  if (!rec_persistent(L.H) && step_epilogue(c.p.c)) {
    // generic width, bf16: one launch per recurrence step, z_t = [x_t | h_{t-1}] [W ; U] + b with the cell as its epilogue
    ProfScope ps(is_note ? PC_LSTM_FWD_NOTE : PC_LSTM_FWD_TIME, c.st);
    RUN(dj_launch_lstm_step_fwd_fused(L.H, (int)tiles, steps, c.at(wX), L.DP, L.D, c.at(wWt), c.P + L.b,
                                      c.train ? c.at(wZ) : nullptr, c.at(wH), c.train ? c.at(wC) : nullptr,
                                      c.at<float>(c.p.w_step), c.p.c.recurrent_sigmoid, c.st));
    return 0;
  }
  if (!rec_persistent(L.H)) {        // per-step path: z row-major in the operand dtype, in place in the stash buffer
    {
      ProfScope ps(PC_GEMM_XW, c.st);
      RUN(dj_launch_gemm_nt(dt, (int)M, 4 * L.H, L.DP, c.at(wX), L.DP, c.at(wWt), L.DP, c.at(wZ), 4 * L.H, 0, c.P + L.b,
                            c.st));
    }
    ProfScope ps(is_note ? PC_LSTM_FWD_NOTE : PC_LSTM_FWD_TIME, c.st);
    RUN(dj_launch_lstm_step_fwd(dt, L.H, (int)tiles, steps, c.at(wZ), c.at(wUf), c.at(wH),
                                c.train ? c.at(wC) : nullptr, c.at<float>(c.p.w_step), c.p.c.recurrent_sigmoid, c.st));
    return 0;
  }
  // x W + b of all steps as one GEMM into the axis' dZ buffer (unused until BPTT), fragment-tiled; the sweep reads it
  // from there and leaves the gate stash in Z
  {
    ProfScope ps(PC_GEMM_XW, c.st);
    RUN(dj_launch_gemm_nt(dt, (int)M, 4 * L.H, L.DP, c.at(wX), L.DP, c.at(wWt), L.DP, c.at(wZx), 4 * L.H, 2, c.P + L.b,
                          c.st));
  }
  ProfScope ps(is_note ? PC_LSTM_FWD_NOTE : PC_LSTM_FWD_TIME, c.st);
  // fp32 inference with a handful of tiles (generation in the parity mode): 8 workgroups per tile with their slice of
  // U resident in LDS instead of one workgroup per tile streaming all of it every step (dj_lstm.hip)
  if (!c.train && dt == DJ_F32 && L.H == 256 && tiles <= 8 && cluster_enabled(c.p.c) &&
      !(kflags(c.p.c) & DJ_KF_NO_CLUSTER_F32)) {
    const int rc = dj_launch_lstm_fwd_cluster_f32((int)tiles, steps, c.at(wZx), c.at(wUf), c.at(wH),
                                                  c.p.c.recurrent_sigmoid, c.at(c.p.w_cluster), kflags(c.p.c), c.st);
    if (rc != 1017) return rc;
  }
  RUN(dj_launch_lstm_fwd(dt, L.H, (int)tiles, steps, c.at(wZx), c.train ? c.at(wZ) : nullptr, c.at(wUf), c.at(wH),
                         c.train ? c.at(wC) : nullptr, c.p.c.recurrent_sigmoid, c.st));
  return 0;
}

// time axis (model.py:51-89): fills H_t[Lt-1]
int time_axis_forward(const Ctx& c, const float* notes, const float* beat) {
  const Plan& p = c.p;
  const int dt = p.c.dtype;
  const float pin = p.c.input_dropout, pdr = p.c.dropout;
  DjDrop d_notes = mkdrop(c, DJ_SITE_NOTES, pin, c.train);
  if (!c.bins_ext) RUN(dj_launch_bins(notes, c.at<float>(p.w_bins), p.B, p.T, p.N, p.c.octave, d_notes, c.st));
  if (!c.static_ready) RUN(style_proj_all(c, p.tl, p.w_sp_t, p.Lt));
  FeatArgs fa;
  fa.notes = notes; fa.beat = beat; fa.bins = c.bins_ext ? c.bins_ext : c.at<float>(p.w_bins);
  fa.sp0 = c.at<float>(p.w_sp_t[0]);
  fa.Wc = c.P + p.p_conv_W; fa.bc = c.P + p.p_conv_b;
  fa.B = p.B; fa.T = p.T; fa.N = p.N; fa.NB = p.NB; fa.octave = p.c.octave; fa.F = p.F; fa.FP = p.FP;
  fa.Bfull = c.bins_ext ? c.Bfull : p.B; fa.bt0 = c.bins_ext ? c.b0 * p.T : 0;
  fa.d_notes = d_notes; fa.d_beat = mkdrop(c, DJ_SITE_BEAT, pin, c.train);
  fa.d_conv = mkdrop(c, DJ_SITE_CONV, pdr, c.train);
  fa.d_style = mkdrop(c, DJ_SITE_TSTYLE + 0, pdr, c.train);
  {
    ProfScope ps(PC_FEATURE_FWD, c.st);
    // conv as im2col GEMM on the MFMA units: Xcol -> Y = Xcol Wc + bc -> tanh / dropout / assembly
    if (!c.static_ready) RUN(dj_launch_cvt_transpose(dt, c.P + p.p_conv_W, 72, 64, c.at(p.w_WcT), 80, c.st));
    RUN(dj_launch_feature_xcol(dt, &fa, c.at(p.w_Xcol), c.st));
    RUN(dj_launch_gemm_nt(dt, (int)p.Mt, 64, 80, c.at(p.w_Xcol), 80, c.at(p.w_WcT), 80, c.at(p.w_Ycol), 64, 0,
                          c.P + p.p_conv_b, c.st));
    RUN(dj_launch_feature_asm(dt, &fa, c.at(p.w_X_t[0]), c.at(p.w_Ycol), c.train ? 1 : 0, c.st));
  }
  // inference with the reference's two 256-unit layers and few tiles (generation: 5): both layers in ONE launch, the
  // upper one a few steps behind the lower one, which writes the upper layer's input itself (dj_lstm.hip, ClPair) --
  // the two latency chains of T steps overlap instead of following each other
  if (!c.train && p.Lt == 2 && dt == DJ_BF16 && p.tl[0].H == 256 && p.tl[1].H == 256 && p.tl[0].DP <= 128 &&
      p.tl[1].D == 256 && p.tl[1].DP == 256 && p.tilesT <= 64 && fuse_xw(p.c, p.tl[0]) && fuse_xw(p.c, p.tl[1]) && cluster_enabled(p.c) &&
      dj_lstm_fused_nkx(dt, 256, p.tl[0].D) == 8 && !(kflags(p.c) & DJ_KF_NO_CLUSTER_PAIR)) {
    ProfScope ps(PC_LSTM_FWD_TIME, c.st);
    const int rc = dj_launch_lstm_fwd_cluster_pair(
        (int)p.tilesT, p.T, c.at(p.w_X_t[0]), p.tl[0].DP, c.at(p.w_Wt_t[0]), c.P + p.tl[0].b, c.at(p.w_Uf_t[0]),
        c.at(p.w_X_t[1]), c.at(p.w_Wt_t[1]), c.P + p.tl[1].b, c.at(p.w_Uf_t[1]), c.at(p.w_H_t[1]),
        c.at<float>(p.w_sp_t[1]), p.tl[1].D, p.N, p.B, p.c.recurrent_sigmoid, c.at(p.w_cluster), kflags(p.c), c.st);
    if (rc != 1017) return rc;          // 1017: the device cannot hold both halves -- layer by layer below
  }
  for (int l = 0; l < p.Lt; ++l) {
    const LstmP& L = p.tl[l];
    if (l > 0) {
      GlueArgs g;
      g.B = p.B; g.T = p.T; g.N = p.N; g.Hd = p.Ht; g.D = L.D; g.DP = L.DP; g.in_na = 0; g.out_na = 0;
      g.sp = c.at<float>(p.w_sp_t[l]); g.chosen = nullptr;
      g.d_out = mkdrop(c, DJ_SITE_TOUT + (l - 1), pdr, c.train);
      g.d_style = mkdrop(c, DJ_SITE_TSTYLE + l, pdr, c.train);
      g.d_chosen = mkdrop(c, DJ_SITE_CHOSEN, pin, c.train);
      ProfScope ps(PC_GLUE_FWD, c.st);
      RUN(dj_launch_glue_fwd(dt, &g, c.at(p.w_H_t[l - 1]), c.at(p.w_X_t[l]), c.st));
    }
    RUN(lstm_layer_fwd(c, L, p.tilesT, p.T, p.Mt, p.w_X_t[l], p.w_Wt_t[l], p.w_Uf_t[l], p.w_Z_t[l], p.w_dZ_t, p.w_H_t[l],
                       p.w_C_t[l], false));
  }
  return 0;
}

// note axis (model.py:91-126) from producer buffer `wHin` (TA order if !in_na), then head (+ loss)
int note_axis_forward(const Ctx& c, int64_t wHin, int in_na, int d_out_site, const float* chosen,
                      const float* target, float* out, float* loss, float* grads) {
  const Plan& p = c.p;
  const int dt = p.c.dtype;
  const float pin = p.c.input_dropout, pdr = p.c.dropout;
  RUN(style_proj_all(c, p.nl, p.w_sp_n, p.Ln));
  for (int l = 0; l < p.Ln; ++l) {
    const LstmP& L = p.nl[l];
    GlueArgs g;
    g.B = p.B; g.T = p.T; g.N = p.N; g.D = L.D; g.DP = L.DP; g.out_na = 1;
    g.sp = c.at<float>(p.w_sp_n[l]);
    g.d_style = mkdrop(c, DJ_SITE_NSTYLE + l, pdr, c.train);
    g.d_chosen = mkdrop(c, DJ_SITE_CHOSEN, pin, c.train);
    if (l == 0) {
      g.Hd = p.Ht; g.in_na = in_na; g.chosen = chosen;
      g.d_out = mkdrop(c, d_out_site, pdr, c.train && d_out_site >= 0);
      ProfScope ps(PC_GLUE_FWD, c.st);
      RUN(dj_launch_glue_fwd(dt, &g, c.at(wHin), c.at(p.w_X_n[0]), c.st));
    } else {
      g.Hd = p.Hn; g.in_na = 1; g.chosen = nullptr;
      g.d_out = mkdrop(c, DJ_SITE_NOUT + (l - 1), pdr, c.train);
      ProfScope ps(PC_GLUE_FWD, c.st);
      RUN(dj_launch_glue_fwd(dt, &g, c.at(p.w_H_n[l - 1]), c.at(p.w_X_n[l]), c.st));
    }
    RUN(lstm_layer_fwd(c, L, p.tilesN, p.N, p.Mn, p.w_X_n[l], p.w_Wt_n[l], p.w_Uf_n[l], p.w_Z_n[l], p.w_dZ_n, p.w_H_n[l],
                       p.w_C_n[l], true));
  }
  HeadArgs h;
  h.B = p.B; h.T = p.T; h.N = p.N; h.Hd = p.Hn;
  h.Wn = c.P + p.p_nd_W; h.bn = c.P + p.p_nd_b; h.Wv = c.P + p.p_vd_W; h.bv = c.P + p.p_vd_b;
  h.target = target; h.out = out; h.loss = loss;
  h.dWn = grads ? grads + p.p_nd_W : nullptr; h.dbn = grads ? grads + p.p_nd_b : nullptr;
  h.dWv = grads ? grads + p.p_vd_W : nullptr; h.dbv = grads ? grads + p.p_vd_b : nullptr;
  h.inv_count = (float)(1.0 / ((double)p.BT * p.N));
  h.d_out = mkdrop(c, DJ_SITE_NOUT + (p.Ln - 1), pdr, c.train);
  if (target && !grads) {   // loss only: head gradients land in scratch (dX_n is unused in inference)
    float* dummy = c.at<float>(p.w_dX_n);
    h.dWn = dummy; h.dbn = dummy + 2 * p.Hn; h.dWv = dummy + 2 * p.Hn + 2; h.dbv = dummy + 3 * p.Hn + 2;
  }
  ProfScope ps(PC_HEAD, c.st);
  RUN(dj_launch_head(dt, &h, c.at(p.w_H_n[p.Ln - 1]), (grads && c.train) ? c.at(p.w_dH_n) : nullptr, c.st));
  return 0;
}

// dZ of a bf16 layer on the persistent kernels is column-tile-major ([4H/256][rows][256]: the 32 rows of a recurrence
// step are one contiguous 16 KiB block per 256-column tile, which is what the weight-gradient GEMM streams per stage;
// with row-major dZ its 512-byte pieces at 2 KiB stride were fetched from HBM once per row tile: 1.6x the bytes)
inline int64_t dz_tile_stride(const Ctx& c, const LstmP& L, int64_t M) {
  if (c.p.c.dtype != DJ_BF16) return 0;
  // (round 4: the generic-width path too -- at the scaled shape the weight-gradient GEMM read 3.3x its algorithmic bytes
  // from the row-major dZ; DJ_KF_NO_STEP_EPILOGUE keeps the round-3 layout with the rest of that form)
  if (!rec_persistent(L.H)) return ((4 * L.H) % 256 == 0 && step_epilogue(c.p.c)) ? M * 256 : 0;
  return M * 256;
}

int lstm_layer_bwd(const Ctx& c, const LstmP& L, float* G, int64_t tiles, int steps, int64_t M, int64_t wX,
                   int64_t wWc, int64_t wWp, int64_t wUb, int64_t wZ, int64_t wH, int64_t wC, int64_t wdH, int64_t wdX,
                   int64_t wdZ, bool is_note) {
  const int dt = c.p.c.dtype;
  const int fdx = fuse_dx(c.p.c, L);
  const int64_t cts = dz_tile_stride(c, L, M);
  {
    ProfScope ps(is_note ? PC_LSTM_BWD_NOTE : PC_LSTM_BWD_TIME, c.st);
    if (rec_persistent(L.H)) {
      RUN(dj_launch_lstm_bwd(dt, L.H, (int)tiles, steps, c.at(wZ), c.at(wUb), c.at(wC), c.at(wdH), c.at(wdZ), cts, G + L.b,
                             c.p.c.recurrent_sigmoid, fdx ? c.at(wWp) : nullptr, L.D, fdx ? c.at(wdX) : nullptr, L.DP,
                             kflags(c.p.c), c.st));
    } else {
      const void* Uc = dt == DJ_F32 ? (const void*)(c.P + L.U) : (const void*)c.at(wUb);
      RUN(dj_launch_lstm_step_bwd(dt, L.H, (int)tiles, steps, c.at(wZ), Uc, c.at(wC), c.at(wdH), c.at(wdZ), cts, G + L.b,
                                  c.at<float>(c.p.w_step), c.p.c.recurrent_sigmoid, step_epilogue(c.p.c), c.st));
    }
  }
  {
    ProfScope ps(PC_GEMM_DW, c.st);
    RUN(dj_launch_lstm_wgrad(dt, M, steps, c.at(wX), L.DP, L.D, c.at(wH), L.H, c.at(wdZ), 4 * L.H, cts, G + L.W, G + L.U,
                             c.at(c.p.w_zero), c.st));
  }
  if (fdx == 1) return 0;
  const void* Bt = dt == DJ_F32 ? (const void*)(c.P + L.W) : (const void*)c.at(wWc);
  const int ncols = fdx == 2 ? L.D - L.D % 32 : L.D;       // the kernel above already wrote the last column block
  ProfScope ps(PC_GEMM_DX, c.st);
  RUN(dj_launch_gemm_nt_ex(dt, (int)M, ncols, 4 * L.H, c.at(wdZ), cts ? 256 : 4 * L.H, 1, cts, Bt, 4 * L.H, c.at(wdX), L.DP,
                           1, 0, nullptr, c.st));
  return 0;
}

// gradients of every per-layer style Dense in two launches (weight/bias gradients; input gradient summed into dstyle)
int style_dense_bwd_all(const Ctx& c, float* G) {
  const Plan& p = c.p;
  ProfScope ps(PC_STYLE_BWD, c.st);
  DenseBatch d;
  memset(&d, 0, sizeof(d));
  d.M = (int)p.BT; d.K = p.SU; d.A = c.at<float>(p.w_style);
  auto add = [&](const LstmP& L, int64_t w_dpre) {
    const int l = d.n++;
    d.W[l] = c.P + L.dW; d.dC[l] = c.at<float>(w_dpre); d.dW[l] = G + L.dW; d.db[l] = G + L.db; d.N[l] = L.D;
  };
  for (int l = 0; l < p.Lt; ++l) add(p.tl[l], p.w_dpre_t[l]);
  for (int l = 0; l < p.Ln; ++l) add(p.nl[l], p.w_dpre_n[l]);
  DJ_CHECK(hipMemsetAsync(c.at<float>(p.w_dstyle), 0, p.BT * p.SU * sizeof(float), c.st));
  return dj_launch_dense_small_batch_bwd(&d, c.at<float>(p.w_dstyle), c.st);
}

int check_ws(const Plan& p, void* ws, int64_t bytes) {
  if (!ws || bytes < p.ws_bytes) return 1200;
  if (((uintptr_t)ws) & 255) return 1201;
  return 0;
}

// one kernel: fault counts of the cluster scratch added to out[0..2] as floats (total, expired waits, misplaced), counts
// reset; the description of the first expired wait stays until the host takes it (dj_workspace_faults_async)
__global__ void faults_to_float_kernel(int* fault_words, float* out) {
  if (threadIdx.x == 0) {
    const int e = fault_words[0], m = fault_words[1];
    if (e | m) {
      out[0] += (float)(e + m); out[1] += (float)e; out[2] += (float)m;
      fault_words[0] = 0; fault_words[1] = 0;
    }
  }
}

}  // namespace


// =============================================================================== C ABI
extern "C" {

int32_t dj_env_reload(void) {
  env_defaults() = read_env();
  return 0;
}

int32_t dj_style_embedding(const dj_config* cfg, const float* params, const float* style_in, int32_t rows, float* out,
                           void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  if (!params || !style_in || !out || rows < 1) return 1210;
  return dj_launch_dense_small(style_in, rows, p.S, params + p.p_style_W, params + p.p_style_b, out, p.SU, 0,
                               (hipStream_t)stream);                                     // model.py:141-142
}

int32_t dj_workspace_faults_async(const dj_config* cfg, void* ws, int64_t ws_bytes, float* out, void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  RUN(check_ws(p, ws, ws_bytes));
  if (!out) return 1210;
  if (!(p.Ht == 256 || p.Hn == 256)) return 0;             // no cluster kernels in this plan: nothing to add
  hipLaunchKernelGGL(faults_to_float_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     (int*)dj_lstm_cluster_fault_words((char*)ws + p.w_cluster), out);
  return (int)hipGetLastError();
}

int32_t dj_abi_version(void) { return DJ_ABI_VERSION; }
int32_t dj_config_size(void) { return (int32_t)sizeof(dj_config); }

int64_t dj_param_count(const dj_config* cfg) {
  Plan p;
  if (make_plan(cfg, p)) return -1;
  return p.nparams;
}

int32_t dj_param_info(const dj_config* cfg, int32_t index, char* name, int32_t cap, int64_t* offset, int32_t* shape,
                      int32_t* ndim) {
  Plan p;
  int rc = make_plan(cfg, p);
  if (rc) return rc;
  struct E { char nm[48]; int64_t off; int sh[4]; int nd; };
  E e[8 + 10 * MAXL];
  int n = 0;
  auto add = [&](const char* nm, int64_t off, int nd, int a, int b, int c3) {
    snprintf(e[n].nm, sizeof(e[n].nm), "%s", nm);
    e[n].off = off; e[n].nd = nd; e[n].sh[0] = a; e[n].sh[1] = b; e[n].sh[2] = c3; e[n].sh[3] = 0; ++n;
  };
  add("style/kernel", p.p_style_W, 2, p.S, p.SU, 0); add("style/bias", p.p_style_b, 1, p.SU, 0, 0);
  add("conv/kernel", p.p_conv_W, 3, 24, 3, 64); add("conv/bias", p.p_conv_b, 1, 64, 0, 0);
  char buf[48];
  for (int ax = 0; ax < 2; ++ax) {
    int Lc = ax ? p.Ln : p.Lt;
    for (int l = 0; l < Lc; ++l) {
      const LstmP& L = ax ? p.nl[l] : p.tl[l];
      const char* a = ax ? "note" : "time";
      snprintf(buf, sizeof buf, "%s_dense%d/kernel", a, l); add(buf, L.dW, 2, p.SU, L.D, 0);
      snprintf(buf, sizeof buf, "%s_dense%d/bias", a, l); add(buf, L.db, 1, L.D, 0, 0);
      snprintf(buf, sizeof buf, "%s_lstm%d/kernel", a, l); add(buf, L.W, 2, L.D, 4 * L.H, 0);
      snprintf(buf, sizeof buf, "%s_lstm%d/recurrent_kernel", a, l); add(buf, L.U, 2, L.H, 4 * L.H, 0);
      snprintf(buf, sizeof buf, "%s_lstm%d/bias", a, l); add(buf, L.b, 1, 4 * L.H, 0, 0);
    }
  }
  add("note_dense/kernel", p.p_nd_W, 2, p.Hn, 2, 0); add("note_dense/bias", p.p_nd_b, 1, 2, 0, 0);
  add("volume_dense/kernel", p.p_vd_W, 2, p.Hn, 1, 0); add("volume_dense/bias", p.p_vd_b, 1, 1, 0, 0);
  if (index < 0 || index >= n) return 1000;
  if (name && cap > 0) snprintf(name, cap, "%s", e[index].nm);
  if (offset) *offset = e[index].off;
  if (shape) memcpy(shape, e[index].sh, sizeof(int) * 4);
  if (ndim) *ndim = e[index].nd;
  return 0;
}

int64_t dj_workspace_bytes(const dj_config* cfg) {
  Plan p;
  if (make_plan(cfg, p)) return -1;
  return p.ws_bytes;
}

int32_t dj_workspace_init(const dj_config* cfg, void* ws, int64_t bytes, void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  RUN(check_ws(p, ws, bytes));
  return (int)hipMemsetAsync(ws, 0, p.ws_bytes, (hipStream_t)stream);
}

int32_t dj_train_fwd_bwd(const dj_config* cfg, const float* params, float* grads, const float* notes,
                         const float* chosen, const float* beat, const float* style, const float* target, float* out,
                         float* loss, void* ws, int64_t ws_bytes, uint64_t seed, void* stream) {
  return dj_train_fwd_bwd_acc(cfg, params, grads, notes, chosen, beat, style, target, out, loss, ws, ws_bytes, seed, 0,
                              stream);
}

int32_t dj_train_fwd_bwd_acc(const dj_config* cfg, const float* params, float* grads, const float* notes,
                             const float* chosen, const float* beat, const float* style, const float* target,
                             float* out, float* loss, void* ws, int64_t ws_bytes, uint64_t seed, int32_t accumulate,
                             void* stream) {
  return dj_train_fwd_bwd_mb(cfg, params, grads, notes, chosen, beat, style, target, out, loss, ws, ws_bytes, seed,
                             accumulate, 0, 0, nullptr, stream);
}

// pitch_bins table of a whole batch (model.py:43-45): bins[i, b, t] = sum_k dropped_notes[b, t, i + 12 k, 0]
int32_t dj_pitch_bins(const dj_config* cfg, const float* notes, float* bins, uint64_t seed, int32_t train,
                      int32_t batch_offset, void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  if (!notes || !bins) return 1210;
  if (batch_offset < 0) return 1211;
  // the input-dropout mask of these samples is the one of rows batch_offset * T * N ... of the batch they belong to
  return dj_launch_bins(notes, bins, p.B, p.T, p.N, p.c.octave,
                        mkdrop(seed, DJ_SITE_NOTES, p.c.input_dropout, train != 0,
                               (uint32_t)((int64_t)batch_offset * p.T * p.N)),
                        (hipStream_t)stream);
}

int32_t dj_train_fwd_bwd_mb(const dj_config* cfg, const float* params, float* grads, const float* notes,
                            const float* chosen, const float* beat, const float* style, const float* target,
                            float* out, float* loss, void* ws, int64_t ws_bytes, uint64_t seed, int32_t accumulate,
                            int32_t full_batch, int32_t batch_offset, const float* bins_full, void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  RUN(check_ws(p, ws, ws_bytes));
  if (!params || !grads || !notes || !chosen || !beat || !style || !target || !loss) return 1210;
  if (bins_full && (batch_offset < 0 || full_batch < batch_offset + p.B)) return 1211;
  if (!bins_full && (full_batch || batch_offset)) return 1211;
  Ctx c{p, params, (char*)ws, (hipStream_t)stream, true, seed};
  if (bins_full) {
    c.b0 = batch_offset; c.Bfull = full_batch; c.bins_ext = bins_full;
  }
  const int dt = p.c.dtype;
  float* G = grads;
  if (!accumulate) DJ_CHECK(hipMemsetAsync(G, 0, p.nparams * sizeof(float), c.st));   // every gradient kernel adds into G
  DJ_CHECK(hipMemsetAsync(loss, 0, sizeof(float), c.st));
  for (int l = 0; l < p.Lt; ++l) RUN(prep_layer(c, p.tl[l], p.w_Wt_t[l], p.w_Wc_t[l], p.w_Uf_t[l], p.w_Ub_t[l], p.w_Wp_t[l], true));
  for (int l = 0; l < p.Ln; ++l) RUN(prep_layer(c, p.nl[l], p.w_Wt_n[l], p.w_Wc_n[l], p.w_Uf_n[l], p.w_Ub_n[l], p.w_Wp_n[l], true));

  // ---------------- forward
  RUN(style_forward(c, style));
  RUN(time_axis_forward(c, notes, beat));
  RUN(note_axis_forward(c, p.w_H_t[p.Lt - 1], 0, DJ_SITE_TOUT + (p.Lt - 1), chosen, target, out, loss, G));

  // ---------------- backward (BPTT)
  const float pin = p.c.input_dropout, pdr = p.c.dropout;
  for (int l = p.Ln - 1; l >= 0; --l) {
    const LstmP& L = p.nl[l];
    RUN(lstm_layer_bwd(c, L, G, p.tilesN, p.N, p.Mn, p.w_X_n[l], p.w_Wc_n[l], p.w_Wp_n[l], p.w_Ub_n[l], p.w_Z_n[l], p.w_H_n[l],
                       p.w_C_n[l], p.w_dH_n, p.w_dX_n, p.w_dZ_n, true));
    GlueArgs g;
    g.B = p.B; g.T = p.T; g.N = p.N; g.D = L.D; g.DP = L.DP; g.out_na = 1;
    g.sp = c.at<float>(p.w_sp_n[l]); g.chosen = nullptr;
    g.d_style = mkdrop(c, DJ_SITE_NSTYLE + l, pdr, true);
    g.d_chosen = mkdrop(c, DJ_SITE_CHOSEN, pin, true);
    if (l == 0) {
      g.Hd = p.Ht; g.in_na = 0;
      g.d_out = mkdrop(c, DJ_SITE_TOUT + (p.Lt - 1), pdr, true);
      ProfScope ps(PC_GLUE_BWD, c.st);
      RUN(dj_launch_glue_bwd(dt, &g, c.at(p.w_dX_n), c.at(p.w_dH_t), c.at<float>(p.w_dpre_n[l]), c.st));
    } else {
      g.Hd = p.Hn; g.in_na = 1;
      g.d_out = mkdrop(c, DJ_SITE_NOUT + (l - 1), pdr, true);
      ProfScope ps(PC_GLUE_BWD, c.st);
      RUN(dj_launch_glue_bwd(dt, &g, c.at(p.w_dX_n), c.at(p.w_dH_n), c.at<float>(p.w_dpre_n[l]), c.st));
    }
  }
  for (int l = p.Lt - 1; l >= 0; --l) {
    const LstmP& L = p.tl[l];
    RUN(lstm_layer_bwd(c, L, G, p.tilesT, p.T, p.Mt, p.w_X_t[l], p.w_Wc_t[l], p.w_Wp_t[l], p.w_Ub_t[l], p.w_Z_t[l], p.w_H_t[l],
                       p.w_C_t[l], p.w_dH_t, p.w_dX_t, p.w_dZ_t, false));
    if (l > 0) {
      GlueArgs g;
      g.B = p.B; g.T = p.T; g.N = p.N; g.Hd = p.Ht; g.D = L.D; g.DP = L.DP; g.in_na = 0; g.out_na = 0;
      g.sp = c.at<float>(p.w_sp_t[l]); g.chosen = nullptr;
      g.d_out = mkdrop(c, DJ_SITE_TOUT + (l - 1), pdr, true);
      g.d_style = mkdrop(c, DJ_SITE_TSTYLE + l, pdr, true);
      g.d_chosen = mkdrop(c, DJ_SITE_CHOSEN, pin, true);
      ProfScope ps(PC_GLUE_BWD, c.st);
      RUN(dj_launch_glue_bwd(dt, &g, c.at(p.w_dX_t), c.at(p.w_dH_t), c.at<float>(p.w_dpre_t[l]), c.st));
    } else {
      FeatArgs fa;
      fa.notes = notes; fa.beat = beat; fa.bins = c.at<float>(p.w_bins); fa.sp0 = c.at<float>(p.w_sp_t[0]);
      fa.Wc = c.P + p.p_conv_W; fa.bc = c.P + p.p_conv_b;
      fa.B = p.B; fa.T = p.T; fa.N = p.N; fa.NB = p.NB; fa.octave = p.c.octave; fa.F = p.F; fa.FP = p.FP;
      fa.Bfull = p.B; fa.bt0 = 0;                       // pitch_bins has no gradient (its input is data)
      fa.d_notes = mkdrop(c, DJ_SITE_NOTES, pin, true); fa.d_beat = mkdrop(c, DJ_SITE_BEAT, pin, true);
      fa.d_conv = mkdrop(c, DJ_SITE_CONV, pdr, true); fa.d_style = mkdrop(c, DJ_SITE_TSTYLE + 0, pdr, true);
      ProfScope ps(PC_FEATURE_BWD, c.st);
      RUN(dj_launch_feature_bwd(dt, &fa, c.at(p.w_dX_t), c.at(p.w_Ycol), G + p.p_conv_b, c.at<float>(p.w_dpre_t[0]),
                                c.st));
      // dWc[72,64] = Xcol^T (dropped notes, im2col) * d(conv pre-activation)
      RUN(dj_launch_gemm_tn(dt, p.Mt, 80, 72, 64, c.at(p.w_Xcol), 80, c.at(p.w_Ycol), 64, G + p.p_conv_W, 64, 0, 0, c.st));
    }
  }
  // style Dense layers and the style embedding (model.py:141-142,77,110-113)
  RUN(style_dense_bwd_all(c, G));
  ProfScope ps(PC_STYLE_BWD, c.st);
  RUN(dj_launch_dense_small_bwd_w(style, (int)p.BT, p.S, c.at<float>(p.w_dstyle), p.SU, G + p.p_style_W,
                                  G + p.p_style_b, c.st));
  return 0;
}

int32_t dj_nadam_step(float* params, const float* grads, float* m, float* v, int64_t count, int64_t step_t,
                      double* m_schedule, float lr, float beta1, float beta2, float epsilon, float schedule_decay,
                      float grad_scale, void* stream) {
  if (!params || !grads || !m || !v || !m_schedule || step_t < 1) return 1220;
  // Keras 2.x Nadam.get_updates
  double t = (double)step_t;
  double mu_t = beta1 * (1.0 - 0.5 * pow(0.96, t * schedule_decay));
  double mu_t1 = beta1 * (1.0 - 0.5 * pow(0.96, (t + 1.0) * schedule_decay));
  double ms_new = (*m_schedule) * mu_t;
  double ms_next = ms_new * mu_t1;
  NadamArgs a;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = epsilon;
  a.mu_t = (float)mu_t; a.mu_t1 = (float)mu_t1; a.ms_new = (float)ms_new; a.ms_next = (float)ms_next;
  a.bc2 = (float)(1.0 - pow((double)beta2, t));
  a.gscale = grad_scale;
  *m_schedule = ms_new;
  ProfScope ps(PC_NADAM, (hipStream_t)stream);
  return dj_launch_nadam(params, grads, m, v, count, &a, (hipStream_t)stream);
}

int32_t dj_predict(const dj_config* cfg, const float* params, const float* notes, const float* chosen,
                   const float* beat, const float* style, const float* target, float* out, float* loss, void* ws,
                   int64_t ws_bytes, void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  RUN(check_ws(p, ws, ws_bytes));
  if (!params || !notes || !chosen || !beat || !style || !out) return 1210;
  if (target && !loss) return 1211;
  Ctx c{p, params, (char*)ws, (hipStream_t)stream, false, 0};
  for (int l = 0; l < p.Lt; ++l) RUN(prep_layer(c, p.tl[l], p.w_Wt_t[l], p.w_Wc_t[l], p.w_Uf_t[l], p.w_Ub_t[l], p.w_Wp_t[l], false));
  for (int l = 0; l < p.Ln; ++l) RUN(prep_layer(c, p.nl[l], p.w_Wt_n[l], p.w_Wc_n[l], p.w_Uf_n[l], p.w_Ub_n[l], p.w_Wp_n[l], false));
  if (target) DJ_CHECK(hipMemsetAsync(loss, 0, sizeof(float), c.st));
  RUN(style_forward(c, style));
  RUN(time_axis_forward(c, notes, beat));
  RUN(note_axis_forward(c, p.w_H_t[p.Lt - 1], 0, -1, chosen, target, out, target ? loss : nullptr, nullptr));
  return 0;
}

int32_t dj_time_model_predict(const dj_config* cfg, const float* params, const float* notes, const float* beat,
                              const float* style, float* time_out, void* ws, int64_t ws_bytes, void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  RUN(check_ws(p, ws, ws_bytes));
  if (!params || !notes || !beat || !style || !time_out) return 1210;
  Ctx c{p, params, (char*)ws, (hipStream_t)stream, false, 0};
  for (int l = 0; l < p.Lt; ++l) RUN(prep_layer(c, p.tl[l], p.w_Wt_t[l], p.w_Wc_t[l], p.w_Uf_t[l], p.w_Ub_t[l], p.w_Wp_t[l], false));
  RUN(style_forward(c, style));
  RUN(time_axis_forward(c, notes, beat));
  return dj_launch_ta_to_canonical(p.c.dtype, c.at(p.w_H_t[p.Lt - 1]), time_out, p.B, p.T, p.N, p.Ht, c.st);
}

int32_t dj_note_model_predict(const dj_config* cfg, const float* params, const float* features, const float* chosen,
                              const float* style, float* out, void* ws, int64_t ws_bytes, void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  RUN(check_ws(p, ws, ws_bytes));
  if (!params || !features || !chosen || !style || !out) return 1210;
  Ctx c{p, params, (char*)ws, (hipStream_t)stream, false, 0};
  for (int l = 0; l < p.Ln; ++l) RUN(prep_layer(c, p.nl[l], p.w_Wt_n[l], p.w_Wc_n[l], p.w_Uf_n[l], p.w_Ub_n[l], p.w_Wp_n[l], false));
  RUN(style_forward(c, style));
  RUN(dj_launch_canonical_to_na(p.c.dtype, features, c.at(p.w_featin), p.B, p.T, p.N, p.Ht, c.st));
  RUN(note_axis_forward(c, p.w_featin, 1, -1, chosen, nullptr, out, nullptr, nullptr));
  return 0;
}

int32_t dj_lstm_wgrad(int32_t dtype, int64_t M, int32_t steps, const void* X, int32_t DP, int32_t D, const void* Hs,
                      int32_t H, const void* dZ, int32_t N, int64_t dz_tile_stride, float* dW, float* dU, const void* zeros,
                      void* stream) {
  if (dtype != DJ_F32 && dtype != DJ_BF16) return 1106;
  if ((M % 32) || steps < 1 || D > DP || !zeros) return 1231;
  if (dz_tile_stride && dz_tile_stride < M * 256) return 1233;
  return dj_launch_lstm_wgrad(dtype, M, steps, X, DP, D, Hs, H, dZ, N, dz_tile_stride, dW, dU, zeros, (hipStream_t)stream);
}

int32_t dj_generate_step(const dj_config* cfg, const float* params, const float* notes_win, const float* beat_win,
                         const float* style_win, const double* uniforms, const float* temperature, float* next_notes,
                         int32_t* draws_used, void* ws, int64_t ws_bytes, void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  RUN(check_ws(p, ws, ws_bytes));
  if (!params || !notes_win || !beat_win || !style_win || !uniforms || !temperature || !next_notes || !draws_used)
    return 1210;
  if (p.B > 8) return 1301;
  const int64_t gen_floats = (8 * 64 + 4 * 8 * 512 + (int64_t)p.B * p.N * 4 * p.Hn + 63) / 64 * 64;   // the sampler's float scratch ...
  const int64_t need = gen_floats * 4 + dj_gen_wpack_bytes();            // ... and its bf16 weight fragments (bf16 mode)
  if (need > p.Mn * (int64_t)p.nl[0].DP * p.esz) return 1302;       // scratch lives in the (unused) dX_n area
  Ctx c{p, params, (char*)ws, (hipStream_t)stream, false, 0};
  for (int l = 0; l < p.Lt; ++l) RUN(prep_layer(c, p.tl[l], p.w_Wt_t[l], p.w_Wc_t[l], p.w_Uf_t[l], p.w_Ub_t[l], p.w_Wp_t[l], false));
  RUN(style_forward(c, style_win));
  RUN(time_axis_forward(c, notes_win, beat_win));                   // generate.py:106-109 (whole window, last step used)
  int64_t offs[6 + 5 * MAXL];
  offs[0] = p.p_style_W; offs[1] = p.p_style_b; offs[2] = p.p_nd_W; offs[3] = p.p_nd_b; offs[4] = p.p_vd_W;
  offs[5] = p.p_vd_b;
  for (int l = 0; l < p.Ln; ++l) {
    offs[6 + 5 * l] = p.nl[l].dW; offs[7 + 5 * l] = p.nl[l].db; offs[8 + 5 * l] = p.nl[l].W; offs[9 + 5 * l] = p.nl[l].U;
    offs[10 + 5 * l] = p.nl[l].b;
  }
  return dj_launch_generate_notes(p.c.dtype, p.B, p.T, p.N, p.Ht, p.Hn, p.Ln, p.S, p.SU, params, offs,
                                  c.at(p.w_H_t[p.Lt - 1]), style_win + (int64_t)(p.T - 1) * p.S, (int64_t)p.T * p.S,
                                  c.at<float>(p.w_dX_n), uniforms, temperature, next_notes, draws_used, nullptr,
                                  nullptr, p.c.recurrent_sigmoid, 0, c.at<float>(p.w_dX_n) + gen_floats, kflags(p.c), c.st);
}

int32_t dj_gen_state_size(void) { return dj_gen_state_bytes(); }

namespace {
// the part of a generated step that depends on the parameters and the style vector only: weight packing, style
// embedding and projections, transposed conv kernel, the sampler's style terms
int generate_static(const Ctx& c, const float* style_win, const int64_t* offs) {
  const Plan& p = c.p;
  for (int l = 0; l < p.Lt; ++l) RUN(prep_layer(c, p.tl[l], p.w_Wt_t[l], p.w_Wc_t[l], p.w_Uf_t[l], p.w_Ub_t[l], p.w_Wp_t[l], false));
  RUN(style_forward(c, style_win));
  return 0;
}
void generate_offs(const Plan& p, int64_t* offs) {
  offs[0] = p.p_style_W; offs[1] = p.p_style_b; offs[2] = p.p_nd_W; offs[3] = p.p_nd_b; offs[4] = p.p_vd_W;
  offs[5] = p.p_vd_b;
  for (int l = 0; l < p.Ln; ++l) {
    offs[6 + 5 * l] = p.nl[l].dW; offs[7 + 5 * l] = p.nl[l].db; offs[8 + 5 * l] = p.nl[l].W; offs[9 + 5 * l] = p.nl[l].U;
    offs[10 + 5 * l] = p.nl[l].b;
  }
}
int generate_resident(const dj_config* cfg, const float* params, void* state, float* results, const double* uniform_pool,
                      const float* notes_src, float* notes_dst, const float* beat_src, float* beat_dst,
                      const float* style_win, void* ws, int64_t ws_bytes, void* stream, bool static_ready) {
  Plan p;
  RUN(make_plan(cfg, p));
  RUN(check_ws(p, ws, ws_bytes));
  if (!params || !state || !results || !uniform_pool || !notes_src || !notes_dst || !beat_src || !beat_dst || !style_win)
    return 1210;
  if (p.B > 8) return 1301;
  const int64_t gen_floats = (8 * 64 + 4 * 8 * 512 + (int64_t)p.B * p.N * 4 * p.Hn + 63) / 64 * 64;   // the sampler's float scratch ...
  const int64_t need = gen_floats * 4 + dj_gen_wpack_bytes();            // ... and its bf16 weight fragments (bf16 mode)
  if (need > p.Mn * (int64_t)p.nl[0].DP * p.esz) return 1302;
  Ctx c{p, params, (char*)ws, (hipStream_t)stream, false, 0};
  c.static_ready = static_ready;
  int64_t offs[6 + 5 * MAXL];
  generate_offs(p, offs);
  if (!static_ready) RUN(generate_static(c, style_win, offs));
  RUN(time_axis_forward(c, notes_src, beat_src));
  RUN(dj_launch_generate_notes(p.c.dtype, p.B, p.T, p.N, p.Ht, p.Hn, p.Ln, p.S, p.SU, params, offs,
                               c.at(p.w_H_t[p.Lt - 1]), style_win + (int64_t)(p.T - 1) * p.S, (int64_t)p.T * p.S,
                               c.at<float>(p.w_dX_n), uniform_pool, nullptr, nullptr, nullptr, state, results,
                               p.c.recurrent_sigmoid, static_ready ? 1 : 0, c.at<float>(p.w_dX_n) + gen_floats, kflags(p.c),
                               c.st));
  return dj_launch_gen_advance(state, results, notes_src, notes_dst, beat_src, beat_dst, p.B, p.T, p.N, p.NB, c.st);
}
}  // namespace

int32_t dj_generate_step_resident(const dj_config* cfg, const float* params, void* state, float* results,
                                  const double* uniform_pool, const float* notes_src, float* notes_dst,
                                  const float* beat_src, float* beat_dst, const float* style_win, void* ws,
                                  int64_t ws_bytes, void* stream) {
  return generate_resident(cfg, params, state, results, uniform_pool, notes_src, notes_dst, beat_src, beat_dst, style_win,
                           ws, ws_bytes, stream, false);
}
// the same step for a workspace on which dj_generate_prepare has run with these parameters and this style window
// (and nothing else since): the per-run constants are not recomputed
int32_t dj_generate_step_prepared(const dj_config* cfg, const float* params, void* state, float* results,
                                  const double* uniform_pool, const float* notes_src, float* notes_dst,
                                  const float* beat_src, float* beat_dst, const float* style_win, void* ws,
                                  int64_t ws_bytes, void* stream) {
  return generate_resident(cfg, params, state, results, uniform_pool, notes_src, notes_dst, beat_src, beat_dst, style_win,
                           ws, ws_bytes, stream, true);
}
int32_t dj_generate_prepare(const dj_config* cfg, const float* params, const float* style_win, void* ws, int64_t ws_bytes,
                            void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  RUN(check_ws(p, ws, ws_bytes));
  if (!params || !style_win) return 1210;
  if (p.B > 8) return 1301;
  const int64_t gen_floats = (8 * 64 + 4 * 8 * 512 + (int64_t)p.B * p.N * 4 * p.Hn + 63) / 64 * 64;   // the sampler's float scratch ...
  const int64_t need = gen_floats * 4 + dj_gen_wpack_bytes();            // ... and its bf16 weight fragments (bf16 mode)
  if (need > p.Mn * (int64_t)p.nl[0].DP * p.esz) return 1302;
  Ctx c{p, params, (char*)ws, (hipStream_t)stream, false, 0};
  int64_t offs[6 + 5 * MAXL];
  generate_offs(p, offs);
  RUN(generate_static(c, style_win, offs));
  RUN(style_proj_all(c, p.tl, p.w_sp_t, p.Lt));
  RUN(dj_launch_cvt_transpose(p.c.dtype, c.P + p.p_conv_W, 72, 64, c.at(p.w_WcT), 80, c.st));
  // the matrix-core sampler's bf16 weight fragments depend on the parameters only: packed here, once per run
  RUN(dj_launch_generate_pack(p.c.dtype, p.Hn, p.Ln, params, offs, c.at<float>(p.w_dX_n) + gen_floats, kflags(p.c), c.st));
  return dj_launch_generate_prep(p.B, p.T, p.N, p.Ht, p.Hn, p.Ln, p.S, p.SU, params, offs,
                                 style_win + (int64_t)(p.T - 1) * p.S, (int64_t)p.T * p.S, c.at<float>(p.w_dX_n), c.st);
}

int32_t dj_lstm_pack_w(int32_t dtype, int32_t H, const float* W, int32_t D, void* wpack, void* stream) {
  return dj_launch_lstm_pack_w(dtype, H, W, D, dj_lstm_fused_nkx(dtype, H, D), wpack, (hipStream_t)stream);
}
int32_t dj_lstm_fwd_fused(int32_t dtype, int32_t H, int32_t ntiles, int32_t steps, const void* X, int32_t DP,
                          int32_t D, const void* wpack, const float* bias, void* stash, const void* upack_fwd,
                          void* Hout, void* Cout, int32_t sigm, void* cluster_scratch, void* stream) {
  if (D > DP) return 1232;
  return dj_launch_lstm_fwd_fused(dtype, H, ntiles, steps, X, DP, dj_lstm_fused_nkx(dtype, H, D), wpack, bias, stash,
                                  upack_fwd, Hout, Cout, sigm, (env_defaults().flags & DJ_KF_NO_CLUSTER) ? nullptr : cluster_scratch,
                                  env_defaults().flags, (hipStream_t)stream);
}
int64_t dj_lstm_cluster_scratch_bytes(void) { return dj_lstm_cluster_scratch_bytes_impl(); }
int64_t dj_lstm_stash_bytes(int32_t dtype, int32_t H, int64_t rows) {
  if ((dtype != DJ_F32 && dtype != DJ_BF16) || !rec_persistent(H) || rows < 0) return -1;
  return rows * dj_lstm_stash_row_bytes(dtype, H);
}

// ------------------------------------------------------------------ live kernel timing
int32_t dj_profile_enable(int32_t on) {
  for (auto& r : g_prof.recs) { g_prof.pool.push_back(r.a); g_prof.pool.push_back(r.b); }
  g_prof.recs.clear();
  g_prof.on = on != 0;
  g_prof.only = (on >= 2 && on - 2 < PC_COUNT) ? on - 2 : -1;
  return 0;
}
int32_t dj_profile_category_count(void) { return PC_COUNT; }
const char* dj_profile_category_name(int32_t cat) { return (cat >= 0 && cat < PC_COUNT) ? kProfNames[cat] : ""; }
int32_t dj_profile_read(int32_t cat, double* total_ms, int64_t* scopes) {
  if (cat < 0 || cat >= PC_COUNT || !total_ms || !scopes) return 1240;
  double tot = 0;
  int64_t n = 0;
  for (auto& r : g_prof.recs) {
    if (r.cat != cat) continue;
    DJ_CHECK(hipEventSynchronize(r.b));
    float ms = 0;
    DJ_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
    tot += ms;
    ++n;
  }
  *total_ms = tot;
  *scopes = n;
  return 0;
}

// ------------------------------------------------------------------ single-kernel entry points
int32_t dj_gemm_nt(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* Bt,
                   int32_t ldb, void* C, int32_t ldc, int32_t c_mode, const float* bias, void* stream) {
  if (dtype != DJ_F32 && dtype != DJ_BF16) return 1106;
  if (c_mode < 0 || c_mode > 3) return 1005;
  return dj_launch_gemm_nt(dtype, M, N, K, A, lda, Bt, ldb, C, ldc, c_mode, bias, (hipStream_t)stream);
}
int32_t dj_gemm_nt_tiled_a(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, int64_t a_tile_stride,
                           const void* Bt, int32_t ldb, void* C, int32_t ldc, int32_t c_mode, const float* bias,
                           void* stream) {
  if (dtype != DJ_BF16) return 1106;
  if (c_mode < 0 || c_mode > 2) return 1005;
  if (a_tile_stride < (int64_t)M * 256) return 1233;
  return dj_launch_gemm_nt_ex(dtype, M, N, K, A, 256, 1, a_tile_stride, Bt, ldb, C, ldc, 1, c_mode, bias,
                              (hipStream_t)stream);
}
int32_t dj_gemm_tn(int32_t dtype, int64_t M, int32_t Ka, int32_t ka_valid, int32_t N, const void* A, int32_t lda,
                   const void* B, int32_t ldb, float* C, int32_t ldc, int32_t a_shift, int32_t steps, void* stream) {
  if (dtype != DJ_F32 && dtype != DJ_BF16) return 1106;
  if (M % 32) return 1230;
  return dj_launch_gemm_tn(dtype, M, Ka, ka_valid, N, A, lda, B, ldb, C, ldc, a_shift, steps, (hipStream_t)stream);
}
int32_t dj_lstm_pack(int32_t dtype, int32_t H, const float* U, void* f, void* b, void* stream) {
  return dj_launch_lstm_pack(dtype, H, U, f, b, (hipStream_t)stream);
}
int32_t dj_lstm_fwd(int32_t dtype, int32_t H, int32_t ntiles, int32_t steps, const void* Zx, void* stash,
                    const void* upack, void* Hout, void* Cout, int32_t sigm, void* stream) {
  return dj_launch_lstm_fwd(dtype, H, ntiles, steps, Zx, stash, upack, Hout, Cout, sigm, (hipStream_t)stream);
}
int32_t dj_lstm_bwd(int32_t dtype, int32_t H, int32_t ntiles, int32_t steps, const void* Z, const void* upack,
                    const void* C, const void* dH, void* dZ, int64_t dz_tile_stride, float* dbias, int32_t sigm,
                    void* stream) {
  return dj_launch_lstm_bwd(dtype, H, ntiles, steps, Z, upack, C, dH, dZ, dz_tile_stride, dbias, sigm, nullptr, 0, nullptr,
                            0, env_defaults().flags, (hipStream_t)stream);
}
int32_t dj_lstm_bwd_dx(int32_t dtype, int32_t H, int32_t ntiles, int32_t steps, const void* Z, const void* upack,
                       const void* C, const void* dH, void* dZ, int64_t dz_tile_stride, float* dbias, int32_t sigm,
                       const void* wtpack, int32_t D, void* dX, int32_t DP, void* stream) {
  if (!wtpack) return 1013;
  return dj_launch_lstm_bwd(dtype, H, ntiles, steps, Z, upack, C, dH, dZ, dz_tile_stride, dbias, sigm, wtpack, D, dX, DP,
                            env_defaults().flags, (hipStream_t)stream);
}
int32_t dj_lstm_cluster_faults(void* cluster_scratch, void* stream) {
  return dj_lstm_cluster_faults_impl(cluster_scratch, (hipStream_t)stream);
}
int32_t dj_workspace_cluster_faults(const dj_config* cfg, void* ws, int64_t ws_bytes, void* stream) {
  Plan p;
  if (make_plan(cfg, p) || check_ws(p, ws, ws_bytes)) return -1;
  if (!(p.Ht == 256 || p.Hn == 256)) return 0;
  return dj_lstm_cluster_faults_impl((char*)ws + p.w_cluster, (hipStream_t)stream);
}
int32_t dj_workspace_cluster_fault_report(const dj_config* cfg, void* ws, int64_t ws_bytes, int32_t* words_host,
                                          void* stream) {
  Plan p;
  RUN(make_plan(cfg, p));
  RUN(check_ws(p, ws, ws_bytes));
  if (!words_host) return 1210;
  memset(words_host, 0, DJ_FAULT_REPORT_WORDS * sizeof(int32_t));
  if (!(p.Ht == 256 || p.Hn == 256)) return 0;
  return dj_lstm_cluster_fault_line((char*)ws + p.w_cluster, words_host, (hipStream_t)stream);
}
int32_t dj_workspace_cluster_faults_take(const dj_config* cfg, void* ws, int64_t ws_bytes, int32_t* words_host,
                                         void* stream) {
  Plan p;
  if (!words_host || make_plan(cfg, p) || check_ws(p, ws, ws_bytes)) return -1;
  memset(words_host, 0, DJ_FAULT_REPORT_WORDS * sizeof(int32_t));
  if (!(p.Ht == 256 || p.Hn == 256)) return 0;
  return dj_lstm_cluster_faults_take((char*)ws + p.w_cluster, words_host, (hipStream_t)stream);
}
int32_t dj_lstm_pack_wt(int32_t dtype, int32_t H, const float* W, int32_t D, void* out, void* stream) {
  return dj_launch_lstm_pack_wt(dtype, H, W, D, out, (hipStream_t)stream);
}

__global__ void dropout_mask_kernel(DjDrop d, int64_t rows, int cols, float* mask) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  uint32_t r = (uint32_t)(idx / cols), c = (uint32_t)(idx % cols);
  mask[idx] = dj_keep(d, dj_rowkey(d, r), c);
}
int32_t dj_dropout_mask(uint64_t seed, int32_t site, float prob, int64_t rows, int32_t cols, float* mask,
                        void* stream) {
  DjDrop d = mkdrop(seed, site, prob, true);
  int64_t n = rows * cols;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d, rows,
                     cols, mask);
  return (int)hipGetLastError();
}

}  // extern "C"
