// DeepJ biaxial-LSTM hot path for MI355X (gfx950 / CDNA4) -- shared device helpers.
//
// Conventions used by every kernel in this directory:
//  * "operand type" T is float (parity mode) or __bf16 (throughput mode); all
//    accumulation, cell state and gate math is fp32.
//  * MFMA tiles are 32x32 (v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16);
//    C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//  * Row orders.  A "sequence tile" is 32 independent sequences.  Activations of
//    a recurrent layer are stored [tile][step][32 seqs][cols] so that one step of
//    one tile is a contiguous 32-row block:
//       TA (time axis, reference model.py:72 Permute): seq = b*N + n, step = t
//       NA (note axis, model.py:119-122):              seq = b*T + t, step = n
//    row = ((seq>>5)*steps + step)*32 + (seq&31).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define DJ_F32 0
#define DJ_BF16 1

#define DJ_CHECK(x)                                \
  do {                                             \
    hipError_t e_ = (x);                           \
    if (e_ != hipSuccess) return (int)e_;          \
  } while (0)

// ---------------------------------------------------------------- conversions
__device__ __forceinline__ float dj_to_f32(float v) { return v; }
__device__ __forceinline__ float dj_to_f32(bf16_t v) { return (float)v; }
// per-device slots of host-side caches (kernel attributes, symbol addresses): one process may drive several GPUs
constexpr int DJ_MAX_DEVICES = 64;
inline int dj_current_device() {
  int d = 0;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < DJ_MAX_DEVICES) ? d : 0;
}
template <typename T> __device__ __forceinline__ T dj_from_f32(float v);
template <> __device__ __forceinline__ float dj_from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t dj_from_f32<bf16_t>(float v) { return (bf16_t)v; }

// ---------------------------------------------------------------- row orders
__device__ __forceinline__ int64_t dj_row(int seq, int step, int steps) {
  return ((int64_t)(seq >> 5) * steps + step) * 32 + (seq & 31);
}
__device__ __forceinline__ int64_t dj_row_ta(int b, int t, int n, int T, int N) { return dj_row(b * N + n, t, T); }
__device__ __forceinline__ int64_t dj_row_na(int b, int t, int n, int T, int N) { return dj_row(b * T + t, n, N); }

// C/D fragment row of register r for this lane (32x32 MFMA)
__device__ __forceinline__ int dj_crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------- dropout hash
// Counter-based masks, regenerated in backward instead of stored.  Bit-for-bit
// the same function as oracle/deepj_oracle.py:keep_mask().  Replaces Keras'
// Dropout layers (reference model.py:58,80,85,116,123,136-138).
#define DJ_SITE_NOTES 1
#define DJ_SITE_BEAT 2
#define DJ_SITE_CHOSEN 3
#define DJ_SITE_CONV 4
#define DJ_SITE_TSTYLE 16
#define DJ_SITE_TOUT 32
#define DJ_SITE_NSTYLE 48
#define DJ_SITE_NOUT 64

__host__ __device__ __forceinline__ uint32_t dj_lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
struct DjDrop {
  uint32_t key;     // hashed (seed, site): dj_dropkey()
  uint32_t thr;     // ceil(p * 2^16); 0 => dropout disabled
  float scale;      // 1/(1-p)
  uint32_t row0;    // added to every row index: a micro-batch draws the masks of its rows of the FULL batch
};
// The (seed, site) pair is hashed BEFORE it meets the row, and the row is hashed before the key is added:
// with key = seed ^ site*golden added to the raw row index, the masks of consecutive seeds (step k / k+1,
// rank r / r+1) were the same random field slid by one row.
__host__ __device__ __forceinline__ uint32_t dj_dropkey(uint64_t seed, uint32_t site) {
  return dj_lowbias32(dj_lowbias32((uint32_t)(seed & 0xFFFFFFFFu) ^ (site * 0x9E3779B9u)) + (uint32_t)(seed >> 32));
}
__host__ __device__ __forceinline__ uint32_t dj_rowkey(const DjDrop& d, uint32_t row) {
  return dj_lowbias32(dj_lowbias32(row + d.row0) + d.key);
}
// returns the multiplier (0 or 1/(1-p)) for element (row, c)
// One hash serves a pair of columns (2 x 16 bits): the elementwise kernels walk 8 consecutive columns per
// thread and were partly bound by this integer work at one hash per element.
__host__ __device__ __forceinline__ float dj_keep(const DjDrop& d, uint32_t rowkey, uint32_t c) {
  if (d.thr == 0) return 1.0f;
  const uint32_t h = dj_lowbias32(rowkey + (c >> 1) * 0x9E3779B9u);
  const uint32_t bits = (c & 1u) ? (h >> 16) : (h & 0xFFFFu);
  return (bits >= d.thr) ? d.scale : 0.0f;
}

// ---------------------------------------------------------------- activations
// Branch-free: 1 - 2/(1 + e^{2x}) is exact in the limits (e -> 0 gives -1, e -> inf gives +1);
// absolute error ~1e-7 (v_exp_f32 + v_rcp_f32), far inside the 1e-3 parity budget.
__device__ __forceinline__ float dj_tanh(float x) {
  float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float dj_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// Keras hard_sigmoid = clip(0.2x + 0.5, 0, 1)
__device__ __forceinline__ float dj_hsig(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.0f), 1.0f); }
__device__ __forceinline__ float dj_hsig_grad(float x) { return fabsf(x) < 2.5f ? 0.2f : 0.0f; }   // -2.5 < x < 2.5
template <bool SIGM> __device__ __forceinline__ float dj_ract(float x) {
  if constexpr (SIGM) return dj_sigmoid(x);
  return dj_hsig(x);
}
template <bool SIGM> __device__ __forceinline__ float dj_ract_grad(float x, float y) {
  if constexpr (SIGM) return y * (1.0f - y);
  return dj_hsig_grad(x);
}

// ---------------------------------------------------------------- 8-bit activated-gate codes (bf16 gate stash)
// What a bf16 forward sweep leaves for BPTT per gate value: i, f, o in [0, 1] as code = clamp(ceil(254 y), 0, 255), decoded
// as the interval midpoint (code - 1/2) / 254 clamped to [0, 1] (|error| <= 1/508); codes 0 and 255 are reserved for the
// SATURATED hard_sigmoid, so its derivative mask (0.2 inside, 0 outside) is exact; g = tanh: code = round(127 g) + 128
// (|error| <= 1/254).  (dj_lstm.hip GateEnc / GateDec are the persistent kernels' forms of the same code; these serve the
// generic-width path: the cell epilogue of dj_gemm.hip writes, the gate kernel of dj_step.hip reads.)
// What goes INTO v_cvt_pk_u8_f32, which saturates to [0, 255] and rounds to nearest-even (probed on the device): the code
// ceil(v), v = 254 y, is cvt(v + 0.49997) -- no clamp, no ceil (round 4: 112 of the ~770 vector instructions of a forward
// sweep step, and the sweep is issue-bound).  It differs from ceil(v) only for v within 3e-5 above an integer (the code is
// then one less: the decoded value is 1.5 instead of at most 0.5 code steps off, once in ~30,000 gates) and, at the knees
// of hard_sigmoid, for |z -+ 2.5| < 6e-7 (derivative mask of the neighbouring branch).  g: round(127 g) + 128 with ties to
// even instead of up.
constexpr float DJ_CODE_ROUND = 0.49997f;
template <bool SIGM> __device__ __forceinline__ float dj_gate_code01(float z, float y) {
  if constexpr (SIGM) return fmaf(y, 254.f, DJ_CODE_ROUND);
  return fmaf(z, 50.8f, 127.f + DJ_CODE_ROUND);                                       // 254 (0.2 z + 0.5)
}
__device__ __forceinline__ float dj_gate_code_g(float g) { return fmaf(g, 127.f, 128.f); }
template <bool SIGM> __device__ __forceinline__ void dj_gate_dec01(float code, float& y, float& dy) {
  y = __builtin_amdgcn_fmed3f(fmaf(code, 1.f / 254.f, -0.5f / 254.f), 0.f, 1.f);
  if constexpr (SIGM) dy = y * (1.f - y);
  else dy = (fabsf(code - 127.5f) < 127.25f) ? 0.2f : 0.f;                            // codes 1 .. 254: the linear part
}
__device__ __forceinline__ float dj_gate_dec_g(float code) { return fmaf(code, 1.f / 127.f, -128.f / 127.f); }

// ---------------------------------------------------------------- MFMA wrappers
// One "k-chunk" of operand fragments per lane:
//   float : 4 consecutive k-steps of the 32x32x2 op  -> f32x4  (k = 8 per chunk)
//   bf16  : one 32x32x16 op                          -> bf16x8 (k = 16 per chunk)
template <typename T> struct DjFrag;
template <> struct DjFrag<float> {
  typedef f32x4 type;
  static constexpr int KCHUNK = 8;
};
template <> struct DjFrag<bf16_t> {
  typedef bf16x8 type;
  static constexpr int KCHUNK = 16;
};

__device__ __forceinline__ void dj_mfma(f32x16& acc, const f32x4& a, const f32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
}
__device__ __forceinline__ void dj_mfma(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

// Fragment element -> k index inside a chunk, for lane half h = lane>>5:
//   float : element e (0..3)  -> k = 4h + e   (the hardware sums over (e, h); which
//   bf16  : element e (0..7)  -> k = 8h + e    logical k a slot carries is ours to
//                                              choose as long as A and B agree)
template <typename T> __device__ __forceinline__ int dj_frag_k(int e, int h);
template <> __device__ __forceinline__ int dj_frag_k<float>(int e, int h) { return 4 * h + e; }
template <> __device__ __forceinline__ int dj_frag_k<bf16_t>(int e, int h) { return 8 * h + e; }

// Read an operand fragment from a k-contiguous LDS row: p points at element
// [row][k0] of a tile whose rows hold k contiguously.
__device__ __forceinline__ f32x4 dj_lds_frag(const float* p, int h) { return *(const f32x4*)(p + 4 * h); }
__device__ __forceinline__ bf16x8 dj_lds_frag(const bf16_t* p, int h) { return *(const bf16x8*)(p + 8 * h); }

// ---------------------------------------------------------------- fragment-tiled stores
// 16 accumulator registers of one lane -> 16 contiguous operand-typed elements
// one v_cvt_pk_bf16_f32 (two scalar conversions + shift + or compile to four instructions)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// 4 consecutive elements of one row as one 8-byte (bf16) / 16-byte (f32) store
__device__ __forceinline__ void dj_store4(bf16_t* p, float a, float b, float c, float d) {
  *(uint2*)p = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
__device__ __forceinline__ void dj_store4(float* p, float a, float b, float c, float d) {
  *(float4*)p = make_float4(a, b, c, d);
}
// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: four v_add_f32 with a DPP operand
// (row_mirror, row_half_mirror, quad_perm [1,0,3,2], quad_perm [2,3,0,1]) instead of four ds_bpermute round trips
// through the LDS crossbar, which is what __shfl_xor compiles to.
__device__ __forceinline__ float dj_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  return v;
}
// write 16 fp32 values as a fragment (round to T)
__device__ __forceinline__ void store_frag(float* p, const float (&x)[16]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) ((float4*)p)[i] = make_float4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3]);
}
__device__ __forceinline__ void store_frag(bf16_t* p, const float (&x)[16]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    ((uint4*)p)[i] = make_uint4(pack_bf16x2(x[8 * i], x[8 * i + 1]), pack_bf16x2(x[8 * i + 2], x[8 * i + 3]),
                                pack_bf16x2(x[8 * i + 4], x[8 * i + 5]), pack_bf16x2(x[8 * i + 6], x[8 * i + 7]));
}

