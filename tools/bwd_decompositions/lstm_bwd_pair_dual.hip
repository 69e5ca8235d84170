// Two re-decompositions of the bf16 H = 256 BPTT sweep over PAIRS of workgroups (round 3): correct, parity-tested, and
// 3-25 % SLOWER than the production per-tile kernel (lstm_bwd_kernel<bf16, 256>) at the BASELINE shape -- kept here, outside
// libdeepj_hip.so, as measured experiments (stamps, variants, what would make them pay).
//   dj_lstm_bwd_pair  every tile on a pair of workgroups (lstm_bwd_pair_kernel), two workgroups of different tiles per CU
//   dj_lstm_bwd_dual  two tiles per workgroup pair, the product of one folded into the gate math of the other
// Built as a UNITY translation unit on top of the product's dj_lstm.hip (same fragment layouts, gate decode, cluster
// wait primitives and per-tile launcher; nothing of it is copied) into tools/bwd_decompositions/libdeepj_bwd_exp.so:
//     sh tools/bwd_decompositions/build.sh
// probe / tests: tools/bwd_pair_probe.py, tools/bwd_decompositions/test_bwd_pair_dual.py (GPU box).
// Scratch: dj_bwd_exp_scratch_bytes() bytes, 128-byte aligned, zeroed once: the product's cluster scratch layout followed
// by one 128-byte counter line per pair.
#include "../../music-generator_amd/csrc/dj_lstm.hip"

namespace {

// ---------------------------------------------------------------- backward, two workgroups per tile (bf16, H = 256)
// The per-tile kernel above holds a compute unit with ONE tile whose step is a strict chain: gate math (VALU: ~900
// vector instructions per wave, two waves per SIMD) -> dz tile -> dz U^T (512 KB of U^T fragments through the CU's
// vector-memory path) -> next gate math.  While one phase runs the other pipe idles, and with 256 tiles for 256 compute
// units there is no second tile to fill it.  Here a tile is split over a PAIR of workgroups of four waves (blocks b and
// b + 8: one XCD under round-robin dispatch, verified like the forward cluster's placement): member `part` owns hidden
// units [128 part, 128 part + 128) -- its gate math, its half of dz, its half of the U^T columns (256 KB per step) --
// so every compute unit hosts TWO workgroups of different tiles whose phases interleave.  Per step a member
//   1. computes dz of its units (lane-local, as above) into the LDS tile and stores it to dZ -- the kernel's output IS
//      the exchange: the partner reads it back from the XCD's L2 (sc1 loads), nothing extra is written;
//   2. waits for its own stores (vmcnt), every wave then arrives on the pair's counter line (8 arrivals per round);
//   3. multiplies the OWN half of the k range (its dz columns are already in LDS) -- the partner's stores and
//      arrivals travel meanwhile;
//   4. waits for the partner's round, fetches the partner's half of dz_t (32 KB) into the LDS tile, multiplies it.
// Waits are bounded and counted like the forward cluster's (fault words of the same scratch; the tile's cell gradient is
// poisoned with NaN, the host repeats the step on the per-tile kernel).  The k order differs from the per-tile kernel
// (own half first), so results agree to fp32 summation order, not bit for bit.
constexpr int BP_MAXPAIRS = 256;
constexpr int BP_ARR = 8;                                  // arrivals per round: 2 members x 4 waves
constexpr size_t BP_OFF_CNT = CL_BYTES;                    // behind the forward cluster's region: one 128-byte line per
constexpr size_t BP_BYTES = (size_t)BP_MAXPAIRS * 128;     //   pair, [0] = counter, [8 + part] = XCC ids
constexpr size_t CL_BYTES_ALL = CL_BYTES + BP_BYTES;

template <bool SIGM>
__global__ __launch_bounds__(256, 2) void lstm_bwd_pair_kernel(const uint8_t* __restrict__ Z,
                                                               const bf16_t* __restrict__ UTpack,
                                                               const bf16_t* __restrict__ C,
                                                               const bf16_t* __restrict__ dH, bf16_t* __restrict__ dZ,
                                                               float* __restrict__ dbias, int steps, int64_t dz_cts,
                                                               int ldz, int* __restrict__ cl, int ntiles, int mate) {
  using T = bf16_t;
  constexpr int H = 256, HP = 128;
  using R = RecCfg<T, H>;
  using Frag = typename DjFrag<T>::type;
  constexpr int LDP = HP + R::EPL;                         // row stride of the dH staging tile (own 128 columns)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* dzs = (T*)smem_raw;                                   // [32][LDZ]: dz_t, all 4H columns
  T* dhs = dzs + 32 * R::LDZ;                              // [32][LDP]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  // pair (b, b + mate): `mate` is 8 (neighbours in dispatch order) or half the grid; a multiple of 8 either way, so
  // both members sit on one XCD under round-robin dispatch
  const int bidx = (int)blockIdx.x;
  int part, pid;
  if (mate == 8) {
    part = (bidx >> 3) & 1;
    pid = (bidx & 7) + 8 * (bidx >> 4);
  } else {
    part = bidx >= mate;
    pid = bidx - part * mate;
  }
  if (pid >= ntiles) return;                               // both members of a pair without a tile leave
  const int64_t tile = pid;
  const int wb = 4 * part + w;                             // this wave's 32-unit block
  int* cnt = (int*)((unsigned char*)cl + BP_OFF_CNT) + pid * 32;
  int* xccs = cnt + 8;
  int* fault = cl + CL_CNT_INTS;

  float dcc[16], dbs[4];
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    dcc[r] = 0.f;
    acc[r] = 0.f;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) dbs[g] = 0.f;

  // round 0: publish the XCD this member runs on, meet the partner, compare
  const int my_xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) + 1;   // HW_REG_XCC_ID[3:0]
  const int my_cu = (int)((__builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11))) & 255) + 1;   // HW_REG_HW_ID[15:8]: cu, sh, se
  if (tid == 0) {
    __hip_atomic_store(xccs + part, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(xccs + 2 + part, my_cu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  {
    const bool ok = cl_wait(cnt, BP_ARR, fault, cl_who(5, pid, part, w), -1, lane) >= 0;
    int other = my_xcc;
    if (ok && lane < 2) other = __hip_atomic_load(xccs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // statistic (fault word 3, not a fault): pairs whose members share a compute unit -- they run in lockstep on one
    // tile and gain nothing from each other
    if (ok && tid == 0 && part == 0 &&
        __hip_atomic_load(xccs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_cu)
      atomicAdd(fault + 3, 1);
    const bool same = __all(other == my_xcc) && (__hip_atomic_load(fault + CLF_HOOK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) == 0;
    if (!ok || !same) {
      if (ok && lane == 0 && w == 0) atomicAdd(fault + CLF_MISPLACED, 1);      // (an expired wait counts itself)
#pragma unroll
      for (int r = 0; r < 16; ++r) dcc[r] = __builtin_nanf("");
    }
  }

  // fragment streams and LDS views of the two halves of the k range (k = gate * 256 + unit): half `part` is this
  // member's own, the other the partner's; 8 k-chunks per gate and half
  const Frag* up = (const Frag*)UTpack + (int64_t)wb * R::NKCB * 64 + lane;
  const Frag* up_own = up + 8 * part * 64;
  const Frag* up_oth = up + 8 * (1 - part) * 64;
  const T* ap_own = dzs + l31 * R::LDZ + HP * part;
  const T* ap_oth = dzs + l31 * R::LDZ + HP * (1 - part);
  auto zaddr = [&](int64_t rb, int g) { return Z + ((rb * R::NCB + (g * H + wb * 32) / 32) * 64 + lane) * 16; };
  auto caddr = [&](int64_t rb) { return C + ((rb * R::NCBH + wb) * 64 + lane) * 16; };
  // dH_t, own 128 columns: 32 rows x 16 vectors of 16 bytes = 2 per thread (named scalars: see the kernel above)
  auto dh_ld = [&](int64_t rb, int i) {
    const int v = tid + 256 * i, row = v >> 4, cv = (v & 15) * 8;
    return *(const uint4*)(dH + (rb * 32 + row) * H + HP * part + cv);
  };
  auto dh_st = [&](int i, uint4 val) {
    const int v = tid + 256 * i, row = v >> 4, cv = (v & 15) * 8;
    *(uint4*)(dhs + row * LDP + cv) = val;
  };
  // half a dz tile between LDS and dZ: 32 rows x 4 gates x 256 bytes = 8 vectors of 16 bytes per thread; one column
  // tile (gate) per pass, 4 rows x 256 bytes per wave instruction
  auto dz_off = [&](int i, int half, int& lds_off) {
    const int g = i >> 1, v2 = tid + 256 * (i & 1), row = v2 >> 4, cv = (v2 & 15) * 8 + HP * half;
    lds_off = row * R::LDZ + g * H + cv;
    return (int64_t)g * dz_cts + (int64_t)row * ldz + cv;
  };
  // dz_t U^T in two halves of 32 k-chunks -- the own half of the k range, then the partner's -- each through a ring of
  // RD fragments.  Chunk i of a half: gate i / 8, chunk i % 8 of that half's 8 per gate.  Rolled loops of RD (fully
  // unrolled, hipcc hoists every fragment load and spills).
  constexpr int RD = 8;
  auto choff = [](int i) { return (16 * (i / 8) + i % 8) * 64; };
  auto ring_fill = [&](Frag (&bq)[RD], const Frag* ub) {
#pragma unroll
    for (int p = 0; p < RD; ++p) bq[p] = ub[choff(p)];
  };
  // consume chunks [RD * it, RD * it + RD) of the half at (ab), refill with the RD chunks that follow in the stream
  auto prod_turn = [&](Frag (&bq)[RD], const T* ab, int it, const Frag* refill) {
#pragma unroll
    for (int u = 0; u < RD; ++u) {
      const int i = u;            // chunk within the turn; (RD * it) enters through the pointers
      Frag a = dj_lds_frag(ab + 256 * (i / 8) + 16 * (i % 8), h);
      dj_mfma(acc, a, bq[u]);
      if (refill) bq[u] = refill[choff(u)];
      if ((u & 3) == 3) asm volatile("" ::: "memory");     // keeps the refills where they are written (register budget)
    }
    (void)it;
  };

  Frag16<T> cnext, cprev;
  GateDec<T, SIGM> gd;
  uint4 dh0 = dh_ld(tile * steps + steps - 1, 0), dh1 = dh_ld(tile * steps + steps - 1, 1);
  cnext.load(caddr(tile * steps + steps - 1));

  for (int t = steps - 1; t >= 0; --t) {
    const int64_t rb = tile * steps + t;
    dh_st(0, dh0);
    dh_st(1, dh1);
#pragma unroll
    for (int g = 0; g < 4; ++g) gd.load(g, zaddr(rb, g));
    if (t > 0) {
      cprev.load(caddr(rb - 1));
      dh0 = dh_ld(rb - 1, 0);
      dh1 = dh_ld(rb - 1, 1);
    }
    lds_barrier();     // dH_t staged; every wave has left the previous step's product (its reads of the dz tile)
    float dhv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dhv[r] = dj_to_f32(dhs[dj_crow(r, lane) * LDP + w * 32 + l31]) + acc[r];
    {
      const int u = wb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = dj_crow(r, lane);
        float ig, fg, gg, og, di, df, dO;
        gd.get(r, ig, fg, gg, og, di, df, dO);
        const float ct = cnext.get(r);
        const float cp = (t > 0) ? cprev.get(r) : 0.f;
        const float dh = dhv[r];
        const float tc = dj_tanh(ct);
        const float dzo = dh * tc * dO;
        const float dc = dcc[r] + dh * og * (1.f - tc * tc);
        const float dzi = dc * gg * di;
        const float dzf = dc * cp * df;
        const float dzg = dc * ig * (1.f - gg * gg);
        dcc[r] = dc * fg;
        T* dp = dzs + row * R::LDZ + u;
        dj_lds_put2(dp, dp + H, dzi, dzf);
        dj_lds_put2(dp + 2 * H, dp + 3 * H, dzg, dzo);
        dbs[0] += dzi;
        dbs[1] += dzf;
        dbs[2] += dzg;
        dbs[3] += dzo;
      }
      if (t > 0) cnext.copy_from(cprev);
    }
    lds_barrier();     // this member's half of dz_t is complete in LDS
    bf16_t* dzg_ = dZ + rb * 32 * ldz;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int lo;
      const int64_t go = dz_off(i, part, lo);
      *(uint4*)(dzg_ + go) = *(const uint4*)(dzs + lo);
    }
    if (t > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // the stores above are the exchange: once acknowledged (they are in the XCD's L2), this wave arrives
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      Frag bq[RD];
      // opaque per step (as loop invariants hipcc keeps all 64 fragment addresses in registers and spills them);
      // through an offset, so that the pointers keep their address space (global_load, not flat_load)
      int64_t uoff = 0;
      asm volatile("" : "+v"(uoff));
      const Frag *uo = up_own + uoff, *ut = up_oth + uoff;
      ring_fill(bq, uo);
      constexpr int NTURN = 32 / RD, GPT = RD / 8;         // turns per half, gates per turn
#pragma unroll 1
      for (int it = 0; it + 1 < NTURN; ++it) prod_turn(bq, ap_own + 256 * GPT * it, it, uo + 16 * GPT * (it + 1) * 64);
      prod_turn(bq, ap_own + 256 * GPT * (NTURN - 1), NTURN - 1, nullptr);
      // the partner half's first fragments travel during the exchange.  (Requested from inside the own half's last
      // turn instead -- one continuous stream -- the sweep was 4 % slower; a ring of 16 was 6 % slower: with two
      // workgroups streaming on one compute unit the fragment stream is bound by the CU's vector-memory path, not by
      // bytes in flight.)
      ring_fill(bq, ut);
      if (cl_wait(cnt, BP_ARR * (steps - t + 1), fault, cl_who(5, pid, part, w), t, lane) < 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dcc[r] = __builtin_nanf("");
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      uint4 pz[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int lo;
        const int64_t go = dz_off(i, 1 - part, lo);
        pz[i] = ld_sc1((const uint4*)(dzg_ + go));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int lo;
        dz_off(i, 1 - part, lo);
        *(uint4*)(dzs + lo) = pz[i];
      }
      lds_barrier();   // the partner's half of dz_t is in LDS
#pragma unroll 1
      for (int it = 0; it + 1 < NTURN; ++it) prod_turn(bq, ap_oth + 256 * GPT * it, it, ut + 16 * GPT * (it + 1) * 64);
      prod_turn(bq, ap_oth + 256 * GPT * (NTURN - 1), NTURN - 1, nullptr);
    }
  }
  if (dbias) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = dbs[g];
      v += __shfl_xor(v, 32);
      if (h == 0) atomicAdd(dbias + g * H + wb * 32 + l31, v);
    }
  }
}

// ---------------------------------------------------------------- backward, TWO tiles on a pair of workgroups (bf16, H = 256)
// The pair kernel above showed that two independent workgroups per compute unit do not interleave by themselves.  Here
// the interleave is written into ONE instruction stream: a pair of workgroups (blocks b, b + 8; four waves each, ONE
// per SIMD, so the whole VGPR + AGPR file is theirs) owns TWO sequence tiles, A and B; member `part` owns hidden units
// [128 part, +128) of both, wave w the 32-unit block 4 part + w.  A wave alternates between the tiles:
//     slot A(t):  gate math of A at step t  (VALU)   ||  dz_B(t+1) U^T  (MFMA + the U^T fragment stream)
//     slot B(t):  gate math of B at step t           ||  dz_A(t) U^T
// four k-chunks of the other tile's product behind each of the 16 gate-math elements, so the MFMAs and their
// fragment loads run in the shadow of the gate math's vector instructions instead of after them.  The product's own
// half of the k range comes first (the member's own dz columns, written one slot earlier); the partner's half of that
// dz tile is fetched from L2 in the middle of the slot (element 8) -- the partner stored it half a slot ago, so the
// exchange latency is off the chain -- and multiplied behind elements 8..15.  The U^T fragments are ONE endless stream
// (own half, partner half, own half, ...: the same 64 chunks every slot, both tiles share the wave's slice) through a
// ring that never drains.  Exchange, placement check and fault handling as in lstm_bwd_pair_kernel (counter line of
// the pair: [0] tile A, [1] tile B).
template <bool SIGM>
__global__ __launch_bounds__(256) void lstm_bwd_dual_kernel(const uint8_t* __restrict__ Z, const bf16_t* __restrict__ UTpack,
                                                            const bf16_t* __restrict__ C, const bf16_t* __restrict__ dH,
                                                            bf16_t* __restrict__ dZ, float* __restrict__ dbias, int steps,
                                                            int64_t dz_cts, int ldz, int* __restrict__ cl, int ntiles) {
  using T = bf16_t;
  constexpr int H = 256, HP = 128;
  using R = RecCfg<T, H>;
  using Frag = typename DjFrag<T>::type;
  constexpr int LDP = HP + R::EPL;
  constexpr int TILE_EL = 32 * R::LDZ + 32 * LDP;          // LDS elements per tile: dz tile [32][LDZ] + dH staging [32][LDP]
  constexpr int RD = 16;                                   // U^T ring: 4 waves x 16 KB in flight per compute unit
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* const lds = (T*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int bidx = (int)blockIdx.x, part = (bidx >> 3) & 1, pid = (bidx & 7) + 8 * (bidx >> 4);
  if (2 * pid >= ntiles) return;                           // ntiles is even: both tiles of a pair exist or neither
  const int64_t tile0 = 2 * pid;
  const int wb = 4 * part + w;
  int* line = (int*)((unsigned char*)cl + BP_OFF_CNT) + pid * 32;
  int* xccs = line + 8;
  int* fault = cl + CL_CNT_INTS;

  float dcc[2][16], dbs[4];
  f32x16 acc[2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dcc[x][r] = 0.f;
      acc[x][r] = 0.f;
    }
#pragma unroll
  for (int g = 0; g < 4; ++g) dbs[g] = 0.f;

  // round 0 (on tile A's counter): publish the XCD, meet the partner, compare
  const int my_xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) + 1;
  if (tid == 0) __hip_atomic_store(xccs + part, my_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(line, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  {
    const bool ok = cl_wait(line, BP_ARR, fault, cl_who(6, pid, part, w), -1, lane) >= 0;
    int other = my_xcc;
    if (ok && lane < 2) other = __hip_atomic_load(xccs + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool same = __all(other == my_xcc) && (__hip_atomic_load(fault + CLF_HOOK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) == 0;
    if (!ok || !same) {
      if (ok && lane == 0 && w == 0) atomicAdd(fault + CLF_MISPLACED, 1);      // (an expired wait counts itself)
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) dcc[x][r] = __builtin_nanf("");
    }
  }

  const Frag* up = (const Frag*)UTpack + (int64_t)wb * R::NKCB * 64 + lane;
  const Frag* up_own = up + 8 * part * 64;
  const Frag* up_oth = up + 8 * (1 - part) * 64;
  auto choff = [](int i) { return (16 * (i / 8) + i % 8) * 64; };      // chunk i of a half: gate i / 8, chunk i % 8
  auto zaddr = [&](int64_t rb, int g) { return Z + ((rb * R::NCB + (g * H + wb * 32) / 32) * 64 + lane) * 16; };
  auto caddr = [&](int64_t rb) { return C + ((rb * R::NCBH + wb) * 64 + lane) * 16; };
  auto dh_ld = [&](int64_t rb, int i) {
    const int v = tid + 256 * i, row = v >> 4, cv = (v & 15) * 8;
    return *(const uint4*)(dH + (rb * 32 + row) * H + HP * part + cv);
  };
  auto dz_off = [&](int i, int half, int& lds_off) {
    const int g = i >> 1, v2 = tid + 256 * (i & 1), row = v2 >> 4, cv = (v2 & 15) * 8 + HP * half;
    lds_off = row * R::LDZ + g * H + cv;
    return (int64_t)g * dz_cts + (int64_t)row * ldz + cv;
  };

  Frag16<T> cnext[2], cprev[2];
  GateDec<T, SIGM> gd[2];
  uint4 dhr[2][2];
  Frag bq[RD];
#pragma unroll
  for (int p = 0; p < RD; ++p) bq[p] = up_own[choff(p)];

  // one slot: gate math of tile X at step t, with the product dz_Y(ty) U^T of the other tile folded in (do_p)
  // (do_p is a compile-time flag: with a run-time branch around every product group hipcc's wait-count bookkeeping
  // loses the order of the ring's loads across the joins and waits for nearly all of them before every MFMA)
  auto slot = [&](auto xc, auto pc, int t) {
    constexpr int X = decltype(xc)::value, Y = 1 - X;
    constexpr bool do_p = decltype(pc)::value;
    T* const dzx = lds + X * TILE_EL;
    T* const dzy = lds + Y * TILE_EL;
    const T* const dhx = dzx + 32 * R::LDZ;
    T* const dhy = dzy + 32 * R::LDZ;
    const int64_t rbx = (tile0 + X) * steps + t;
    const int ty = X == 0 ? t + 1 : t;                     // slot A(t) carries P_B(t+1), slot B(t) carries P_A(t)
    const int64_t rby = (tile0 + Y) * steps + ty;
    const int tn = X == 0 ? t : t - 1;                     // the next gate slot is G_Y(tn)
    // its stash is requested in the MIDDLE of this slot, once the exchange's loads have landed: vector memory returns
    // in order, so HBM loads issued at the top of the slot held up every ring fragment behind them (the first eight
    // elements took 6.7 k cycles against 3.9 k for the second eight), and in front of the exchange they held up its
    // counter poll (4.4 k).  Here only the ring refills issued after them can be delayed, and those are not needed
    // for four elements.
    auto prefetch_next = [&]() {
      if (tn >= 0) {
        const int64_t rbn = (tile0 + Y) * steps + tn;
#pragma unroll
        for (int g = 0; g < 4; ++g) gd[Y].load(g, zaddr(rbn, g));
        cprev[Y].load(caddr(tn > 0 ? rbn - 1 : rbn));
        dhr[Y][0] = dh_ld(rbn, 0);
        dhr[Y][1] = dh_ld(rbn, 1);
      }
    };
    float dhv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dhv[r] = dj_to_f32(dhx[dj_crow(r, lane) * LDP + w * 32 + l31]) + acc[X][r];
    if constexpr (do_p) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[Y][r] = 0.f;
    }
    const T* ap_own = dzy + l31 * R::LDZ + HP * part;
    const T* ap_oth = dzy + l31 * R::LDZ + HP * (1 - part);
    // opaque per slot (as loop invariants hipcc keeps all 64 fragment addresses in registers) -- through an OFFSET:
    // a laundered pointer loses its address space and every fragment load becomes a flat_load, which hipcc can only
    // wait for with vmcnt(0) lgkmcnt(0), i.e. each group of products waited for the refills issued just before it
    int64_t uoff = 0;
    asm volatile("" : "+v"(uoff));
    const Frag *uo = up_own + uoff, *ut = up_oth + uoff;
    const int u = wb * 32 + l31;
    Frag an[4];
    auto afrag_read = [&](int rr) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pos = 4 * rr + j, i = pos & 31;
        an[j] = dj_lds_frag((pos < 32 ? ap_own : ap_oth) + 256 * (i / 8) + 16 * (i % 8), h);
      }
    };
    if constexpr (do_p) afrag_read(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r == 8) {
        if constexpr (!do_p) prefetch_next();
        if constexpr (do_p) {
          // the partner's half of dz_Y(ty): stored half a slot ago, fetched into the LDS tile now
          const int target = BP_ARR * ((steps - ty) + (Y == 0 ? 1 : 0));
          if (cl_wait(line + Y, target, fault, cl_who(6, pid, part, w), ty, lane) < 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) dcc[Y][q] = __builtin_nanf("");
          }
          __builtin_amdgcn_wave_barrier();
          asm volatile("" ::: "memory");
          const bf16_t* gsrc = dZ + rby * 32 * ldz;
          uint4 pz[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            int lo;
            const int64_t go = dz_off(i, 1 - part, lo);
            pz[i] = ld_sc1((const uint4*)(gsrc + go));
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            int lo;
            dz_off(i, 1 - part, lo);
            *(uint4*)(dzy + lo) = pz[i];
          }
          asm volatile("" ::: "memory");
          prefetch_next();
        }
        lds_barrier();       // the partner's half of dz_Y is in LDS (every wave passes here, product or not)
        if constexpr (do_p) afrag_read(8);
      }
      {
        const int row = dj_crow(r, lane);
        float ig, fg, gg, og, di, df, dO;
        gd[X].get(r, ig, fg, gg, og, di, df, dO);
        const float ct = cnext[X].get(r);
        const float cp = (t > 0) ? cprev[X].get(r) : 0.f;
        const float dh = dhv[r];
        const float tc = dj_tanh(ct);
        const float dzo = dh * tc * dO;
        const float dc = dcc[X][r] + dh * og * (1.f - tc * tc);
        const float dzi = dc * gg * di;
        const float dzf = dc * cp * df;
        const float dzg = dc * ig * (1.f - gg * gg);
        dcc[X][r] = dc * fg;
        T* dp = dzx + row * R::LDZ + u;
        dj_lds_put2(dp, dp + H, dzi, dzf);
        dj_lds_put2(dp + 2 * H, dp + 3 * H, dzg, dzo);
        dbs[0] += dzi;
        dbs[1] += dzf;
        dbs[2] += dzg;
        dbs[3] += dzo;
      }
      if constexpr (do_p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pos = 4 * r + j;                                   // 0..31 own half, 32..63 the partner's
          dj_mfma(acc[Y], an[j], bq[pos % RD]);
          const int pn = (pos + RD) & 63, in = pn & 31;                // the stream repeats every slot
          bq[pos % RD] = (pn < 32 ? uo : ut)[choff(in)];
        }
        // the A fragments of the next group are read a whole element ahead (read right in front of their MFMAs they
        // cost an LDS round trip per pair); the partner half's first group (element 8) is read behind the mid barrier
        if (r != 7 && r != 15) afrag_read(r + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t > 0) cnext[X].copy_from(cprev[X]);
    if (tn >= 0) {           // stage dH of the next gate slot
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int v = tid + 256 * i, row = v >> 4, cv = (v & 15) * 8;
        *(uint4*)(dhy + row * LDP + cv) = dhr[Y][i];
      }
    }
    lds_barrier();           // dz_X(t), own half, complete in LDS; dH staged; every wave has left this slot's LDS reads
    bf16_t* gdst = dZ + rbx * 32 * ldz;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int lo;
      const int64_t go = dz_off(i, part, lo);
      *(uint4*)(gdst + go) = *(const uint4*)(dzx + lo);
    }
    if (t > 0) {             // the stores are the exchange: acknowledged, this wave arrives on tile X's counter
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(line + X, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  // prologue: the stash of G_A(steps - 1), its dH staged
  {
    const int64_t rb = tile0 * steps + steps - 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) gd[0].load(g, zaddr(rb, g));
    cnext[0].load(caddr(rb));
    cnext[1].load(caddr((tile0 + 1) * steps + steps - 1));
    if (steps > 1) cprev[0].load(caddr(rb - 1));
    T* const dh0 = lds + 32 * R::LDZ;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + 256 * i, row = v >> 4, cv = (v & 15) * 8;
      *(uint4*)(dh0 + row * LDP + cv) = dh_ld(rb, i);
    }
    lds_barrier();
  }
  // slot A(t) carries the product of B at t + 1, slot B(t) that of A at t: the first and the last slot have none
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  slot(I0{}, std::false_type{}, steps - 1);
  for (int t = steps - 1; t > 0; --t) {
    slot(I1{}, std::true_type{}, t);
    slot(I0{}, std::true_type{}, t - 1);
  }
  slot(I1{}, std::false_type{}, 0);
  if (dbias) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = dbs[g];
      v += __shfl_xor(v, 32);
      if (h == 0) atomicAdd(dbias + g * H + wb * 32 + l31, v);
    }
  }
}

// BPTT of a bf16 H = 256 layer on pairs of workgroups (lstm_bwd_pair_kernel): at most 256 tiles per launch, two
// workgroups per compute unit, the whole grid co-resident.  Returns 1017 when the device cannot hold it (the caller then
// uses the per-tile kernel).
int bwd_pair_blocks_per_cu() {
  static int bpc[DJ_MAX_DEVICES] = {};
  const int dev = dj_current_device();
  if (!bpc[dev]) {
    const size_t smem = (size_t)(32 * RecCfg<bf16_t, 256>::LDZ + 32 * (128 + 8)) * sizeof(bf16_t);
    const void* fns[2] = {(const void*)lstm_bwd_pair_kernel<false>, (const void*)lstm_bwd_pair_kernel<true>};
    int least = 1 << 30;
    for (const void* fn : fns) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 0;
      int n = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, smem) != hipSuccess) return 0;
      least = n < least ? n : least;
    }
    bpc[dev] = least > 0 ? least : -1;
  }
  return bpc[dev] > 0 ? bpc[dev] : 0;
}
int launch_bwd_pair(int ntiles, int steps, const void* Z, const void* UTpack, const void* C, const void* dH, void* dZ,
                    int64_t dz_cts_in, float* dbias, int sigm, void* scratch, hipStream_t st) {
  using R = RecCfg<bf16_t, 256>;
  constexpr int H = 256;
  if (!scratch || ((uintptr_t)scratch & 127)) return 1016;
  const int64_t dz_cts = dz_cts_in ? dz_cts_in : 256;
  const int ldz = dz_cts_in ? 256 : 4 * H;
  if (dz_cts_in && dz_cts_in < (int64_t)ntiles * steps * 32 * 256) return 1018;
  const int64_t slots = (int64_t)cluster_cus() * bwd_pair_blocks_per_cu() / 2;       // pairs the device holds at once
  const int cap = (int)(slots < BP_MAXPAIRS ? slots / 8 * 8 : BP_MAXPAIRS);
  if (cap < 8) return 1017;
  const size_t smem = (size_t)(32 * R::LDZ + 32 * (128 + 8)) * sizeof(bf16_t);
  const uint8_t* z = (const uint8_t*)Z;
  const bf16_t *c = (const bf16_t*)C, *dh = (const bf16_t*)dH;
  bf16_t* dz = (bf16_t*)dZ;
  while (ntiles > 0) {
    const int n = ntiles < cap ? ntiles : cap;
    // counter lines of the pairs start at zero in every launch (a kernel, not a memset node: cluster_reset above)
    hipLaunchKernelGGL(cl_reset_kernel, dim3(BP_BYTES / (256 * 16)), dim3(256), 0, st,
                       (uint4*)((unsigned char*)scratch + BP_OFF_CNT), (int*)((char*)scratch + CL_OFF_FAULT) + CLF_HOOK, 0);
    const dim3 grid((n + 7) / 8 * 16);
    const int mate = 8;      // neighbours in dispatch order; (half the grid apart: all 256 pairs on ONE compute unit each, +5 %)
    if (sigm)
      hipLaunchKernelGGL((lstm_bwd_pair_kernel<true>), grid, dim3(256), smem, st, z, (const bf16_t*)UTpack, c, dh, dz, dbias,
                         steps, dz_cts, ldz, (int*)scratch, n, mate);
    else
      hipLaunchKernelGGL((lstm_bwd_pair_kernel<false>), grid, dim3(256), smem, st, z, (const bf16_t*)UTpack, c, dh, dz, dbias,
                         steps, dz_cts, ldz, (int*)scratch, n, mate);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
    const int64_t rows = (int64_t)n * steps * 32;
    z += rows * 4 * H;
    c += rows * H;
    dh += rows * H;
    dz += rows * ldz;
    ntiles -= n;
  }
  return 0;
}

// BPTT of a bf16 H = 256 layer with two tiles per workgroup pair (lstm_bwd_dual_kernel): one workgroup per compute unit
// (150 KB of LDS), at most `compute units` tiles per launch, an even number of them (an odd last tile takes the
// per-tile kernel).  Returns 1017 when the device cannot hold a group of 16 workgroups.
template <typename T, int H, int DX>
int launch_bwd_x(int ntiles, int steps, const void* Z, const void* UTpack, const void* C, const void* dH, void* dZ,
                 int64_t dz_cts_in, float* dbias, int sigm, const void* WTpack, int NQ, void* dX, int DP, hipStream_t st);
int launch_bwd_dual(int ntiles, int steps, const void* Z, const void* UTpack, const void* C, const void* dH, void* dZ,
                    int64_t dz_cts_in, float* dbias, int sigm, void* scratch, hipStream_t st) {
  using R = RecCfg<bf16_t, 256>;
  constexpr int H = 256;
  if (!scratch || ((uintptr_t)scratch & 127)) return 1016;
  const int64_t dz_cts = dz_cts_in ? dz_cts_in : 256;
  const int ldz = dz_cts_in ? 256 : 4 * H;
  if (dz_cts_in && dz_cts_in < (int64_t)ntiles * steps * 32 * 256) return 1018;
  const int cap = cluster_cus() / 16 * 16 < 2 * BP_MAXPAIRS ? cluster_cus() / 16 * 16 : 2 * BP_MAXPAIRS;   // tiles per launch
  if (cap < 16) return 1017;
  const size_t smem = (size_t)2 * (32 * R::LDZ + 32 * (128 + 8)) * sizeof(bf16_t);
  static bool attr_done_dev[DJ_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[dj_current_device()];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)lstm_bwd_dual_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)lstm_bwd_dual_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  const uint8_t* z = (const uint8_t*)Z;
  const bf16_t *c = (const bf16_t*)C, *dh = (const bf16_t*)dH;
  bf16_t* dz = (bf16_t*)dZ;
  int left = ntiles & ~1;
  while (left > 0) {
    const int n = left < cap ? left : cap;
    hipLaunchKernelGGL(cl_reset_kernel, dim3(BP_BYTES / (256 * 16)), dim3(256), 0, st,
                       (uint4*)((unsigned char*)scratch + BP_OFF_CNT), (int*)((char*)scratch + CL_OFF_FAULT) + CLF_HOOK, 0);
    const dim3 grid((n / 2 + 7) / 8 * 16);
    if (sigm)
      hipLaunchKernelGGL((lstm_bwd_dual_kernel<true>), grid, dim3(256), smem, st, z, (const bf16_t*)UTpack, c, dh, dz, dbias,
                         steps, dz_cts, ldz, (int*)scratch, n);
    else
      hipLaunchKernelGGL((lstm_bwd_dual_kernel<false>), grid, dim3(256), smem, st, z, (const bf16_t*)UTpack, c, dh, dz, dbias,
                         steps, dz_cts, ldz, (int*)scratch, n);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
    const int64_t rows = (int64_t)n * steps * 32;
    z += rows * 4 * H;
    c += rows * H;
    dh += rows * H;
    dz += rows * ldz;
    left -= n;
  }
  if (ntiles & 1)      // the odd last tile: per-tile kernel on the same buffers (dz_cts counts from the buffer's start)
    return launch_bwd_x<bf16_t, 256, 0>(1, steps, z, UTpack, c, dh, dz, dz_cts_in, dbias, sigm, nullptr, 0, nullptr, 0, st);
  return 0;
}

}  // namespace

extern "C" {
int64_t dj_bwd_exp_scratch_bytes(void) { return (int64_t)CL_BYTES_ALL; }
// same arguments as dj_lstm_bwd (include/deepj_hip.h) + the scratch; bf16, H = 256 only
int32_t dj_lstm_bwd_pair(int32_t dtype, int32_t H, int32_t ntiles, int32_t steps, const void* Z, const void* upack,
                         const void* C, const void* dH, void* dZ, int64_t dz_tile_stride, float* dbias, int32_t sigm,
                         void* scratch, void* stream) {
  if (!scratch || dtype != DJ_BF16 || H != 256) return 1016;
  if (ntiles <= 0 || steps <= 0) return 0;
  const int rc = launch_bwd_pair(ntiles, steps, Z, upack, C, dH, dZ, dz_tile_stride, dbias, sigm, scratch, (hipStream_t)stream);
  if (rc != 1017) return rc;
  return dj_launch_lstm_bwd(dtype, H, ntiles, steps, Z, upack, C, dH, dZ, dz_tile_stride, dbias, sigm, nullptr, 0, nullptr, 0,
                            (hipStream_t)stream);
}
int32_t dj_lstm_bwd_dual(int32_t dtype, int32_t H, int32_t ntiles, int32_t steps, const void* Z, const void* upack,
                         const void* C, const void* dH, void* dZ, int64_t dz_tile_stride, float* dbias, int32_t sigm,
                         void* scratch, void* stream) {
  if (!scratch || dtype != DJ_BF16 || H != 256) return 1016;
  if (ntiles <= 0 || steps <= 0) return 0;
  const int rc = launch_bwd_dual(ntiles, steps, Z, upack, C, dH, dZ, dz_tile_stride, dbias, sigm, scratch, (hipStream_t)stream);
  if (rc != 1017) return rc;
  return dj_launch_lstm_bwd(dtype, H, ntiles, steps, Z, upack, C, dH, dZ, dz_tile_stride, dbias, sigm, nullptr, 0, nullptr, 0,
                            (hipStream_t)stream);
}
int32_t dj_bwd_exp_faults(void* scratch, void* stream) { return dj_lstm_cluster_faults_impl(scratch, (hipStream_t)stream); }
}
