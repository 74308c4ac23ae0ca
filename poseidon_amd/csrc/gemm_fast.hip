// gemm_fast — the production GEMM of the scOT hot path (operands already in the compute type, 16-byte aligned).
//
// Same three layouts and the same fragment conventions as gemm.hip (the generic fallback for odd shapes / mixed
// dtypes), restructured after the round-1 rocprof trace showed the generic kernel latency-bound (one dependent
// HBM round trip per staged chunk, 2-byte scattered stores):
//   * branch-free staging: every thread issues all of its 16-byte loads for a K-tile back to back (addresses are
//     clamped instead of guarded; only the K tail is zeroed) into registers, two tiles ahead of the MFMAs;
//   * LDS double buffering, one barrier per K-tile (BK = 64 bf16 / 32 f32);
//   * the accumulator tile goes through LDS so that the fused epilogue (bias, column scale, gelu'(aux), residual,
//     column sums for bias gradients) reads and writes 16/32-byte row segments — fully coalesced;
//   * wgrad (TN): split-K with fp32 atomics, and the bias gradient (column sums of dY) comes for free from the
//     dY tile that is already in LDS.
#include <type_traits>
#include "common.h"
#include "wgrad_group.h"   // the grouped weight-gradient argument block, and how a gradient meets the arena (grad_commit8)
#include <stdlib.h>

#define LAYOUT_NT 0
#define LAYOUT_NN 1
#define LAYOUT_TN 2

struct FastArgs {
  const void* A; const void* B; void* C;
  const float* bias; const float* colscale; const void* aux; const void* resid;
  float* colsum_out;   // NT/NN: += column sums of the stored result;  TN: += column sums of A (=dY) over K
  int M, N, K;
  int lda, ldb, ldc, ldaux, ldres;
  int c_dt, aux_dt, res_dt;
  int a_gelu, b_gelu, aux_gelu_grad, atomic;
  int ksplit;
  int use_tr;
  int xcd_swizzle;  // remap workgroup ids so that the tiles sharing a streamed operand run on ONE XCD (its L2 serves the re-reads)
  void* C2;        // optional second output: C = gelu(v), C2 = gelu'(v)   (fc1 epilogue: value and derivative in one pass)
  int aux_mul;     // aux is multiplied in as is (it already holds gelu'(u)) instead of gelu'(aux)
  int pre;         // the epilogue's aux / residual rows are loaded BEFORE the K loop (see gemm_fast_body)
  float* ws;       // TN split-K: partial tiles ws[z][M][N] (fp32), reduced by splitk_reduce_kernel
  size_t ws_plane; // distance (floats) between the partial tiles of consecutive K slices; 0 = M·N (grouped wgrad: Σ_i M_i·N_i)
  int rmw;         // TN, single split: C += acc by the unique owner (no atomics)
  int out_mode;    // rmw only: WgradProblem::mode (0 = C += acc, 1 = C = s·acc, 2 = C += s·acc), s = *out_scale (NULL: 1)
  const float* out_scale;
};

template <typename CT> struct FT;
template <> struct FT<bf16_t> { static constexpr int BK = 64, EPC = 8, KPAD = 8, RPAD = 8; };
template <> struct FT<float> { static constexpr int BK = 32, EPC = 4, KPAD = 4, RPAD = 4; };

// K-contiguous 16-bit tiles with 128-byte rows (BK = 64) are pad-free and XOR-swizzled: the 16-byte chunk c of row r lives at chunk
// c ^ ((r >> 1) & 7).  A fragment read hands ds_read_b128 the chunks (row lane & 15, chunk c0 + (lane >> 4)); the instruction serves
// its lanes in the groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS table), i.e. every group mixes two
// chunk columns over the 16 rows — with ANY row padding two of its lanes then share a 16-byte bank slot (for an odd pitch P the rows'
// slots P·r mod 16 are a permutation, and shifting half of them by one cannot stay inside the other half's complement): 2.1-2.5
// conflict cycles per LDS instruction in the round-3 SQ counters.  With the swizzle the 16 lanes of a group cover the 16 slots once.
template <typename CT, int R, bool KC, int BKT = FT<CT>::BK> struct FTile {
  static constexpr int BK = BKT, EPC = FT<CT>::EPC;
  static constexpr bool SWZ = KC && sizeof(CT) == 2 && BKT == 64;
  static constexpr int pitch = KC ? (SWZ ? BK : BK + FT<CT>::KPAD) : (R + FT<CT>::RPAD);
  static constexpr int elems = KC ? R * pitch : BK * pitch;
  static constexpr int nchunks = R * BK / EPC;      // 16-byte chunks per tile
  static constexpr int per_thread = (nchunks + 255) / 256;
  static constexpr bool exact = nchunks % 256 == 0;   // otherwise the last pass is guarded (e.g. 96 x 32: 384 chunks)
};

// fragment read of a swizzled K-contiguous tile (FTile::SWZ; r0 a multiple of 16): row r0 + (lane & 15), elements kk + 8 (lane >> 4) ..
__device__ __forceinline__ Frag<bf16_t> lds_frag_kc_swz(const bf16_t* t, int r0, int kk, int lane) {
  Frag<bf16_t> f;
  const int r = lane & 15, ch = (kk >> 3) + (lane >> 4);
  f.v = *(const s16x8_t*)(t + (r0 + r) * 64 + ((ch ^ (r >> 1)) << 3));
  return f;
}
__device__ __forceinline__ Frag<float> lds_frag_kc_swz(const float*, int, int, int) { return Frag<float>(); }   // (never instantiated: SWZ is 16-bit only)

// issue this thread's loads for K-tile starting at k0 (raw 16-byte chunks, no waits, no branches)
template <typename CT, int R, bool KC, int BKT, int NCH>
__device__ __forceinline__ void fload(uint4 (&st)[NCH], const CT* __restrict__ src, int ld, int row0,
                                      int rmax, int k0, int kend, int tid) {
  using T = FTile<CT, R, KC, BKT>;
  constexpr int EPC = T::EPC, BK = T::BK;
#pragma unroll
  for (int i = 0; i < T::per_thread; ++i) {
    const int c = tid + i * 256;
    size_t idx;
    if (KC) {
      constexpr int CPR = BK / EPC;
      const int row = min(row0 + c / CPR, rmax - 1);
      const int k = min(k0 + (c % CPR) * EPC, kend - EPC);
      idx = (size_t)row * ld + k;
    } else {
      constexpr int CPR = R / EPC;
      const int k = min(k0 + c / CPR, kend - 1);
      const int r = min(row0 + (c % CPR) * EPC, rmax - EPC);
      idx = (size_t)k * ld + r;
    }
    if (T::exact || c < T::nchunks) st[i] = *(const uint4*)(src + idx);
  }
}

template <typename CT, int R, bool KC, int BKT, int NCH>
__device__ __forceinline__ void fstore(CT* tile, const uint4 (&st)[NCH], int k0, int kend, int tid) {
  using T = FTile<CT, R, KC, BKT>;
  constexpr int EPC = T::EPC, BK = T::BK;
#pragma unroll
  for (int i = 0; i < T::per_thread; ++i) {
    const int c = tid + i * 256;
    uint4 v = st[i];
    int off, k;
    if (KC) {
      constexpr int CPR = BK / EPC;
      k = k0 + (c % CPR) * EPC;
      const int row = c / CPR, ch = c % CPR;
      off = row * T::pitch + (T::SWZ ? (ch ^ ((row >> 1) & 7)) : ch) * EPC;
    } else {
      constexpr int CPR = R / EPC;
      k = k0 + c / CPR;
      off = (c / CPR) * T::pitch + (c % CPR) * EPC;
    }
    if (k >= kend) v = make_uint4(0, 0, 0, 0);   // K tail (K is a multiple of EPC in this kernel)
    if (T::exact || c < T::nchunks) *(uint4*)(tile + off) = v;
  }
}

// bf16x3: the operands are fp32 in memory; every 4-float chunk is split into hi = bf16(x) and lo = bf16(x - hi) while it is
// staged into LDS (two bf16 tiles per operand), and the product is accumulated as hi·hi + hi·lo + lo·hi — three bf16 MFMAs per
// K-step with ~2^-17 operand error instead of bf16's 2^-9 (the dropped lo·lo term is 2^-18).  MFMA time is < 10 % of these
// kernels, so tripling it is affordable; the price is the fp32 operand traffic.
template <int R, bool KC, int BKT, int NCH>
__device__ __forceinline__ void fstore_x3(bf16_t* hi, bf16_t* lo, const uint4 (&st)[NCH], int k0, int kend, int tid) {
  using L = FTile<float, R, KC, BKT>;     // source chunks: 4 floats
  using T = FTile<bf16_t, R, KC, BKT>;    // LDS tile geometry
#pragma unroll
  for (int i = 0; i < L::per_thread; ++i) {
    const int c = tid + i * 256;
    int off, k;
    if (KC) {
      constexpr int CPR = BKT / 4;
      k = k0 + (c % CPR) * 4;
      off = (c / CPR) * T::pitch + (c % CPR) * 4;
    } else {
      constexpr int CPR = R / 4;
      k = k0 + c / CPR;
      off = (c / CPR) * T::pitch + (c % CPR) * 4;
    }
    uint4 v = st[i];
    if (k >= kend) v = make_uint4(0, 0, 0, 0);
    const float x[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    float h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = bf2f(f2bf(x[j])); l[j] = x[j] - h[j]; }
    if (L::exact || c < L::nchunks) {
      *(uint2*)(hi + off) = make_uint2(pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3]));
      *(uint2*)(lo + off) = make_uint2(pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3]));
    }
  }
}

// WM x WN = arrangement of the 4 waves over the BM x BN tile (WM*WN == 4); each wave owns (BM/WM) x (BN/WN).
// X3 (CT = bf16_t, BKT = 32): fp32 operands in memory, split into hi/lo bf16 tiles in LDS, 3 MFMAs per K-step (see fstore_x3).
// LDS footprint of one workgroup of gemm_fast_body (operand double buffer, reused by the epilogue's C tile)
template <typename CT, int BM, int BN, int BKT, int LAYOUT, bool X3, int KG = 1, int GLDS = 0> struct FastLds {
  static constexpr bool A_KC = (LAYOUT != LAYOUT_TN), B_KC = (LAYOUT == LAYOUT_NT);
  static constexpr int STAGE = (X3 ? 2 : 1) * (FTile<CT, BM, A_KC, BKT>::elems + FTile<CT, BN, B_KC, BKT>::elems);
  static constexpr size_t AB = (size_t)KG * (GLDS ? GLDS : 2) * STAGE * sizeof(CT), C = (size_t)KG * BM * (BN + 4) * sizeof(float) + BN * sizeof(float);
  static constexpr size_t bytes = AB > C ? AB : C;
};

// One workgroup's share of a GEMM: output tile (by, bx), K range [bz·ksplit, (bz+1)·ksplit).  `smem`: FastLds<..>::bytes, 16-byte aligned.
// KG = 2: the workgroup has EIGHT waves in two groups of four; group g walks the K-tiles g, g + 2, ... of the same output tile with its
// own LDS double buffer and accumulators, and the two partial tiles meet in LDS before the epilogue.  For small grids with a long
// contraction (the deep stages' weight gradients: 432 workgroups x 64 K-tiles at 4096 tokens) the serial K loop — one exposed
// load -> LDS -> barrier -> MFMA round trip per tile, at one or two workgroups per CU — is what the launch costs; this halves its
// trip count and doubles the loads in flight per CU without a second pass over partial sums in HBM.
// GLDS = 3 | 4 LDS stages (NT, 16-bit swizzled tiles, K a multiple of BK): the operand tiles go global -> LDS directly (global_load_lds_dwordx4: a
// wave instruction lands 64 x 16 bytes at a wave-uniform LDS base + lane x 16, so a 1 KB piece = 8 tile rows and the chunk swizzle is
// applied on the SOURCE side: lane (row, position pc) fetches chunk pc ^ ((row >> 1) & 7) — probed in tools/probes/glds_probe.hip) through
// GLDS LDS stages: no staging registers, no ds_write pass.
template <typename CT, int BM, int BN, int WM, int WN, int BKT, int NSET, int LAYOUT, bool X3 = false, int KG = 1, int GLDS = 0>
__device__ __forceinline__ void gemm_fast_body(const FastArgs& p, const int bx, const int by, const int bz, char* smem) {
  static_assert(NSET == 2 || NSET == 4, "pipeline depth");
  static_assert(KG == 1 || (KG == 2 && !X3), "K groups: plain 16-bit or fp32 operands");
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(!X3 || (sizeof(CT) == 2 && BKT == 32), "bf16x3: bf16 tiles, one MFMA K-step per tile");
  constexpr bool A_KC = (LAYOUT != LAYOUT_TN);
  constexpr bool B_KC = (LAYOUT == LAYOUT_NT);
  using TA = FTile<CT, BM, A_KC, BKT>;
  using TB = FTile<CT, BN, B_KC, BKT>;
  using MT = typename std::conditional<X3, float, CT>::type;     // element type in memory
  using LA = FTile<MT, BM, A_KC, BKT>;                            // load geometry (16-byte chunks of MT)
  using LB = FTile<MT, BN, B_KC, BKT>;
  constexpr int BK = BKT;
  constexpr int MI = BM / (16 * WM), NI = BN / (16 * WN);
  constexpr int WROWS = BM / WM, WCOLS = BN / WN;
  constexpr int STAGE = (X3 ? 2 : 1) * (TA::elems + TB::elems);   // X3: [A hi][A lo][B hi][B lo]
  constexpr int BOFF = (X3 ? 2 : 1) * TA::elems;                  // offset of the B tile(s) in a stage
  constexpr int CP = BN + 4;                                  // C tile pitch (floats)
  static_assert(FastLds<CT, BM, BN, BKT, LAYOUT, X3, KG, GLDS>::STAGE == STAGE, "LDS sizing");
  static_assert(GLDS == 0 || (LAYOUT == LAYOUT_NT && TA::SWZ && TB::SWZ && !X3 && KG == 1 && BM % 32 == 0 && BN % 32 == 0), "direct-to-LDS: NT, swizzled 16-bit tiles");
  const int kg = KG > 1 ? (int)(threadIdx.x >> 8) : 0;            // K group of this wave quartet
  CT* lds = (CT*)smem + (size_t)kg * 2 * STAGE;

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN, g = lane >> 4;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * p.ksplit;
  const int kend = min(p.K, kbeg + p.ksplit);
  const int nk = (kend - kbeg + BK - 1) / BK;
  const MT* A = (const MT*)p.A;
  const MT* B = (const MT*)p.B;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;  // TN bias-grad partial (thread tid < BM owns column m0+tid of dY)

  // Epilogue operands that do not depend on the product — gelu'(u) of a data gradient (aux), the tensor a data gradient is accumulated
  // into (resid) — are requested HERE, before the K loop: loaded inside the epilogue they were a fully exposed round trip per workgroup
  // (the deep stages' `dgrad fc2 · gelu'` ran 39.8 us against 20.8 us for the same product with two OUTPUT tensors instead).
  constexpr int E_CPRW = BN / 8, E_RPP = (256 * KG) / E_CPRW, E_IT = (BM + E_RPP - 1) / E_RPP;
  uint4 eaux[E_IT], eres[E_IT][2];
  const bool e_on = LAYOUT != LAYOUT_TN && p.pre && p.ws == nullptr;
  const bool e_aux = e_on && p.aux_gelu_grad && p.aux_dt == SCOT_BF16, e_res = e_on && p.resid != nullptr;
  if (e_aux || e_res) {
    const int ecol = min(n0 + (int)(threadIdx.x % E_CPRW) * 8, p.N - 8);
#pragma unroll
    for (int it = 0; it < E_IT; ++it) {
      const int grow = min(m0 + min((int)(threadIdx.x / E_CPRW) + it * E_RPP, BM - 1), p.M - 1);
      if (e_aux) eaux[it] = *(const uint4*)((const bf16_t*)p.aux + (size_t)grow * p.ldaux + ecol);
      if (e_res) {
        if (p.res_dt == SCOT_F32) {
          eres[it][0] = *(const uint4*)((const float*)p.resid + (size_t)grow * p.ldres + ecol);
          eres[it][1] = *(const uint4*)((const float*)p.resid + (size_t)grow * p.ldres + ecol + 4);
        } else {
          eres[it][0] = *(const uint4*)((const bf16_t*)p.resid + (size_t)grow * p.ldres + ecol);
        }
      }
    }
  }

  // NSET register sets: the loads of K-tile t+NSET are issued at the TOP of iteration t and written to LDS at the END of
  // iteration t+NSET-1, i.e. they have NSET-1 full MFMA phases to land.  NSET = 2 when many workgroups share a CU (their
  // interleaving hides the latency); NSET = 4 for the small grids of stages 2/3, where ONE workgroup per CU walks 12–48 K-tiles
  // and each iteration used to stall ~1000 cycles on the HBM round trip of a load issued only one iteration earlier.
  uint4 ra[GLDS ? 1 : NSET][LA::per_thread], rb[GLDS ? 1 : NSET][LB::per_thread];
  // stage `st` <- register set: plain copy, or the hi/lo split of bf16x3
  auto stage_store = [&](CT* st, const uint4 (&a)[LA::per_thread], const uint4 (&b)[LB::per_thread], int k0) {
    if constexpr (X3) {
      fstore_x3<BM, A_KC, BKT>((bf16_t*)st, (bf16_t*)st + TA::elems, a, k0, kend, tid);
      fstore_x3<BN, B_KC, BKT>((bf16_t*)st + BOFF, (bf16_t*)st + BOFF + TB::elems, b, k0, kend, tid);
    } else {
      fstore<CT, BM, A_KC, BKT>(st, a, k0, kend, tid);
      fstore<CT, BN, B_KC, BKT>(st + BOFF, b, k0, kend, tid);
    }
  };
  // K-tile j of THIS group starts at kofs(j) (KG = 1: consecutive tiles)
  auto kofs = [&](int j) { return kbeg + (kg + KG * j) * BK; };
  if constexpr (GLDS == 0) {
#pragma unroll
  for (int u = 0; u < NSET; ++u) {   // unconditional: fload clamps its addresses, fstore zero-fills tiles past kend
    fload<MT, BM, A_KC, BKT>(ra[u], A, p.lda, m0, p.M, kofs(u), kend, tid);
    fload<MT, BN, B_KC, BKT>(rb[u], B, p.ldb, n0, p.N, kofs(u), kend, tid);
  }
  stage_store(lds, ra[0], rb[0], kofs(0));
  __syncthreads();
  }

  auto frag_a = [&](const CT* As, int i, int kk) {
    const int r0 = wr * WROWS + i * 16;
    if constexpr (A_KC && TA::SWZ) return lds_frag_kc_swz(As, r0, kk, lane);
    if (A_KC) return lds_frag_kc(As, TA::pitch, r0, kk, lane);
    return lds_frag_ks(As, TA::pitch, r0, kk + g * 8, kk + g * 8 + 4, lane, p.use_tr);
  };
  auto frag_b = [&](const CT* Bs, int j, int kk) {
    const int c0 = wc * WCOLS + j * 16;
    if constexpr (B_KC && TB::SWZ) return lds_frag_kc_swz(Bs, c0, kk, lane);
    if (B_KC) return lds_frag_kc(Bs, TB::pitch, c0, kk, lane);
    return lds_frag_ks(Bs, TB::pitch, c0, kk + g * 8, kk + g * 8 + 4, lane, p.use_tr);
  };
  auto compute = [&](const CT* st) {
    const CT* As = st;
    const CT* Bs = st + BOFF;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 32) {
      Frag<CT> fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = frag_a(As, i, kk);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = frag_b(Bs, j, kk);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) mma16(acc[i][j], fa[i], fb[j]);
      if constexpr (X3) {
        Frag<CT> fl[MI > NI ? MI : NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) fl[j] = frag_b(Bs + TB::elems, j, kk);        // B lo
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) mma16(acc[i][j], fa[i], fl[j]);
#pragma unroll
        for (int i = 0; i < MI; ++i) fl[i] = frag_a(As + TA::elems, i, kk);        // A lo
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) mma16(acc[i][j], fl[i], fb[j]);
      }
    }
    if (LAYOUT == LAYOUT_TN && p.colsum_out && bx == 0 && tid < BM) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) {
        bsum += from_ct(As[k * TA::pitch + tid]);
        if constexpr (X3) bsum += from_ct(As[TA::elems + k * TA::pitch + tid]);
      }
    }
  };

  // Every iteration issues its loads, computes and stores UNCONDITIONALLY (loads past the end re-read clamped addresses,
  // their tiles are zero-filled by fstore and add nothing): with `if (tile exists)` around the loads the compiler cannot count
  // the loads in flight and falls back to s_waitcnt vmcnt(0) before every LDS store — i.e. no prefetch at all.  The trip
  // count is rounded up to a multiple of NSET for the same reason.
  if constexpr (GLDS != 0) {
    typedef __attribute__((address_space(3))) void* lds_p;
    typedef __attribute__((address_space(1))) const void* gbl_p;
    constexpr int LPW = BM / 32 + BN / 32;       // direct loads per wave and K-tile (1 KB = 8 tile rows each)
    auto issue = [&](CT* st, int k0) {
#pragma unroll
      for (int u = 0; u < BM / 32; ++u) {
        const int q = wave + 4 * u, row = 8 * q + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        const CT* src = (const CT*)A + (size_t)min(m0 + row, p.M - 1) * p.lda + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gbl_p)src, (lds_p)(st + q * 512), 16, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < BN / 32; ++u) {
        const int q = wave + 4 * u, row = 8 * q + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        const CT* src = (const CT*)B + (size_t)min(n0 + row, p.N - 1) * p.ldb + k0 + c * 8;
        __builtin_amdgcn_global_load_lds((gbl_p)src, (lds_p)(st + BOFF + q * 512), 16, 0, 0);
      }
    };
    {
      // GLDS >= 3 stages: tiles t+1 .. t+GLDS-1 are in flight while tile t is multiplied.  The wait that retires tile t leaves the
      // loads of the (up to GLDS-2) newer tiles outstanding; the barrier behind it makes every wave's pieces visible and says that
      // stage (t-1) % GLDS — read during iteration t-1 — is free for tile t+GLDS-1
#pragma unroll
      for (int u = 0; u < GLDS - 1; ++u)
        if (u < nk) issue(lds + u * STAGE, kofs(u));
      for (int t = 0; t < nk; ++t) {
        const int newer = min(GLDS - 2, nk - 1 - t);
        if (newer >= 2) SCOT_VMCNT(2 * LPW);
        else if (newer == 1) SCOT_VMCNT(LPW);
        else SCOT_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (t + GLDS - 1 < nk) issue(lds + ((t + GLDS - 1) % GLDS) * STAGE, kofs(t + GLDS - 1));
        compute(lds + (t % GLDS) * STAGE);
      }
      __syncthreads();     // the epilogue's C tile aliases the stages
    }
  } else {
  const int nk_pad = ((nk + KG - 1) / KG + NSET - 1) / NSET * NSET;    // tiles per group (the same trip count in both: barriers)
  for (int t = 0; t < nk_pad; t += NSET) {
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
      const int tt = t + u;
      // set u held tile tt (already in LDS buffer u&1): refill it with tile tt+NSET
      fload<MT, BM, A_KC, BKT>(ra[u], A, p.lda, m0, p.M, kofs(tt + NSET), kend, tid);
      fload<MT, BN, B_KC, BKT>(rb[u], B, p.ldb, n0, p.N, kofs(tt + NSET), kend, tid);
      compute(lds + (u & 1) * STAGE);
      stage_store(lds + ((u + 1) & 1) * STAGE, ra[(u + 1) % NSET], rb[(u + 1) % NSET], kofs(tt + 1));
      __syncthreads();
    }
  }
  }

  if (LAYOUT == LAYOUT_TN && p.colsum_out && bx == 0 && tid < BM && m0 + tid < p.M) atomicAdd(&p.colsum_out[m0 + tid], bsum);

  // ---- epilogue through LDS
  float* Cs = (float*)smem;               // KG partial tiles back to back: group g fills Cs + g·BM·CP
  float* colacc = Cs + KG * BM * CP;
  {
    float* Cg = Cs + kg * BM * CP;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cg[(wr * WROWS + i * 16 + g * 4 + r) * CP + wc * WCOLS + j * 16 + (lane & 15)] = acc[i][j][r];
  }
  const bool want_colsum = (LAYOUT != LAYOUT_TN) && p.colsum_out != nullptr;
  const int etid = threadIdx.x;           // epilogue: all 256·KG threads share the rows of the tile
  if (want_colsum && etid < BN) colacc[etid] = 0.f;
  __syncthreads();

  if (LAYOUT != LAYOUT_TN && p.atomic) {
    // K slices of an NT / NN product (grid z): every slice ADDS its partial tile into the fp32 result — no partial planes, no ticket, no
    // second pass; slice 0 carries the bias.  A wave instruction covers 64 consecutive floats of one row (4 cache lines, one L2 atomic
    // request each).  The result is either an accumulation target (data gradients into the fp32 residual-stream gradient) or was zeroed
    // by the launcher.  fp32 addition order varies run to run: last-bit differences only.
    float* C = (float*)p.C;
    for (int idx = etid; idx < BM * BN; idx += 256 * KG) {
      const int row = idx / BN, c = idx % BN;
      const int grow = m0 + row, gcol = n0 + c;
      if (grow < p.M && gcol < p.N) {
        float v = Cs[row * CP + c];
        if constexpr (KG == 2) v += Cs[BM * CP + row * CP + c];
        if (p.bias && bz == 0) v += p.bias[gcol];
        atomicAdd(C + (size_t)grow * p.ldc + gcol, v);
      }
    }
    return;
  }
  constexpr int CPRW = BN / 8;            // 8-column chunks per tile row
  constexpr int RPP = (256 * KG) / CPRW;  // rows per pass; threads >= RPP*CPRW idle (BN = 96: 252 of 256 active)
  const int cc = etid % CPRW;             // constant per thread over the row loop
  const bool ep_active = etid < RPP * CPRW;
  const int col = n0 + cc * 8;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float bv[8], sv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = col + j < p.N;
    bv[j] = (p.bias && ok && bz == 0) ? p.bias[col + j] : 0.f;
    sv[j] = (p.colscale && ok) ? p.colscale[col + j] : 1.f;
  }
  if (col < p.N && ep_active) {
#pragma unroll
    for (int it = 0; it < E_IT; ++it) {
      const int row = etid / CPRW + it * RPP;
      if (row >= BM) break;
      const int grow = m0 + row;
      if (grow >= p.M) break;
      float v[8];
      const float4 a = *(const float4*)(Cs + row * CP + cc * 8), b = *(const float4*)(Cs + row * CP + cc * 8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      if constexpr (KG == 2) {
        const float4 a2 = *(const float4*)(Cs + BM * CP + row * CP + cc * 8), b2 = *(const float4*)(Cs + BM * CP + row * CP + cc * 8 + 4);
        v[0] += a2.x; v[1] += a2.y; v[2] += a2.z; v[3] += a2.w; v[4] += b2.x; v[5] += b2.y; v[6] += b2.z; v[7] += b2.w;
      }
      const bool partial = (LAYOUT != LAYOUT_TN) && p.ws != nullptr;
      if (!partial) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (v[j] + bv[j]) * sv[j];
      }
      if (p.aux_gelu_grad && !partial) {
        float x[8];
        if (e_aux) {
          const uint4 u = eaux[it];
          unpack_bf16x2(u.x, x[0], x[1]); unpack_bf16x2(u.y, x[2], x[3]); unpack_bf16x2(u.z, x[4], x[5]); unpack_bf16x2(u.w, x[6], x[7]);
        } else {
          ld8(p.aux, p.aux_dt, (size_t)grow * p.ldaux + col, x);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= p.aux_mul ? x[j] : gelu_grad_f(x[j]);
      }
      if (p.resid && !partial) {
        float x[8];
        if (e_res) {
          const uint4 u = eres[it][0], w = eres[it][1];
          if (p.res_dt == SCOT_F32) {
            x[0] = __uint_as_float(u.x); x[1] = __uint_as_float(u.y); x[2] = __uint_as_float(u.z); x[3] = __uint_as_float(u.w);
            x[4] = __uint_as_float(w.x); x[5] = __uint_as_float(w.y); x[6] = __uint_as_float(w.z); x[7] = __uint_as_float(w.w);
          } else {
            unpack_bf16x2(u.x, x[0], x[1]); unpack_bf16x2(u.y, x[2], x[3]); unpack_bf16x2(u.z, x[4], x[5]); unpack_bf16x2(u.w, x[6], x[7]);
          }
        } else {
          ld8(p.resid, p.res_dt, (size_t)grow * p.ldres + col, x);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += x[j];
      }
      const size_t ci = (size_t)grow * p.ldc + col;
      if (LAYOUT == LAYOUT_TN) {
        if (p.ws) {            // split-K partial tile (dense [M][N], 32-byte aligned rows since N % 8 == 0)
          st8(p.ws + (size_t)bz * (p.ws_plane ? p.ws_plane : (size_t)p.M * p.N), SCOT_F32, (size_t)grow * p.N + col, v);
        } else if (p.rmw) {    // single split: this workgroup is the only writer of the tile
          grad_commit8((float*)p.C, ci, v, p.out_mode, p.out_scale);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) atomicAdd((float*)p.C + ci + j, v[j]);
        }
      } else if (p.ws) {   // NT/NN split-K: raw partial sums; bias/aux/resid are applied by splitk_epilogue_kernel
        st8(p.ws + (size_t)bz * p.M * p.N, SCOT_F32, (size_t)grow * p.N + col, v);
      } else if (p.C2) {
        float gv[8], gd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { float cdf, e; gelu_terms(v[j], cdf, e); gv[j] = v[j] * cdf; gd[j] = cdf + v[j] * 0.3989422804014327f * e; }
        st8(p.C, p.c_dt, ci, gv);
        if (p.C2 != p.C) st8(p.C2, p.c_dt, ci, gd);
      } else {
        st8(p.C, p.c_dt, ci, v);
      }
      if (want_colsum) {
#pragma unroll
        for (int j = 0; j < 8; ++j) csum[j] += v[j];
      }
    }
    if (want_colsum) {
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(&colacc[cc * 8 + j], csum[j]);
    }
  }
  if (want_colsum) {
    __syncthreads();
    if (etid < BN && n0 + etid < p.N) atomicAdd(&p.colsum_out[n0 + etid], colacc[etid]);
  }
}

template <typename CT, int BM, int BN, int WM, int WN, int BKT, int NSET, int LAYOUT, bool X3 = false, int KG = 1, int GLDS = 0>
__global__ __launch_bounds__(256 * KG) void gemm_fast_kernel(FastArgs p) {
  __shared__ __attribute__((aligned(1024))) char smem[FastLds<CT, BM, BN, BKT, LAYOUT, X3, KG, GLDS>::bytes];
  // Workgroup b runs on XCD b % 8 (observed dispatch order, used for speed only).  The tiles that re-read the same streamed
  // operand — all output tiles of one token chunk (TN), all column tiles of one row block (NT/NN) — are renumbered so that they
  // are consecutive ON ONE XCD: its 4 MB L2 then serves the re-reads instead of the fabric (PMC: 3.1x algorithmic bytes before).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const int gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
    const int G = (LAYOUT == LAYOUT_TN) ? gx * gy : gx;
    const int NG = (LAYOUT == LAYOUT_TN) ? gz : gy * gz;
    if (p.xcd_swizzle && NG % 8 == 0) {
      const int L = bx + gx * (by + gy * bz);
      const int j = L >> 3;
      const int group = (L & 7) * (NG >> 3) + j / G, member = j % G;
      if (LAYOUT == LAYOUT_TN) { bz = group; bx = member % gx; by = member / gx; }
      else { by = group % gy; bz = group / gy; bx = member; }
    }
  }
  gemm_fast_body<CT, BM, BN, WM, WN, BKT, NSET, LAYOUT, X3, KG, GLDS>(p, bx, by, bz, smem);
}

// ------------------------------------------------------------------------------------------------------------------------
// Grouped weight gradients: the (up to 8) wgrad GEMMs of one ScOTLayer — dW_i[M_i, N_i] += dY_i[K, M_i]^T · X_i[K, N_i], all
// contracting over the same K tokens — in ONE launch.  Round-2 trace: 273 TN launches + 342 split-K reduce launches per step
// were 31 % of the kernel time at 85 TF/s; a stage-0 gradient has only 4 output tiles, so each launch split K 128 ways to fill
// the chip and moved 19 MB of partial tiles.  Together the four problems of a block have 12 tiles: 40 K slices fill the chip,
// the partials shrink 3x, 8 launches become 2 (this kernel + one grouped reduce), and the deep stages' problems (hundreds of
// 64x64 tiles each, no split) share one launch instead of four tail effects.

template <typename CT, int BM, int BN, int BKT, int NSET, int KG = 1>
__global__ __launch_bounds__(256 * KG) void wgrad_group_kernel(WgradGroupArgs g) {
  __shared__ __attribute__((aligned(16))) char smem[FastLds<CT, BM, BN, BKT, LAYOUT_TN, false, KG>::bytes];
  // all tiles of one K slice consecutively on ONE XCD (workgroup b runs on XCD b % 8: speed only): its L2 serves the re-reads of
  // the slice's operand panels by the tiles that share them
  int L = blockIdx.x, slice, tile;
  if (g.nsplit % 8 == 0) {
    const int j = L >> 3;
    slice = (L & 7) * (g.nsplit >> 3) + j / g.tiles;
    tile = j % g.tiles;
  } else {
    slice = L / g.tiles;
    tile = L % g.tiles;
  }
  int q = 0;
#pragma unroll
  for (int i = 1; i < SCOT_WGRAD_GROUP_MAX; ++i) q += (i < g.n && tile >= g.p[i].tile0) ? 1 : 0;
  const WgradProblem& pr = g.p[q];
  const int local = tile - pr.tile0;
  FastArgs a;
  a.A = pr.A; a.B = pr.B; a.C = pr.C; a.bias = nullptr; a.colscale = nullptr; a.aux = nullptr; a.resid = nullptr;
  a.colsum_out = pr.colsum;
  a.M = pr.M; a.N = pr.N; a.K = g.K; a.lda = pr.lda; a.ldb = pr.ldb; a.ldc = pr.ldc; a.ldaux = 0; a.ldres = 0;
  a.c_dt = SCOT_F32; a.aux_dt = 0; a.res_dt = 0; a.a_gelu = 0; a.b_gelu = 0; a.aux_gelu_grad = 0; a.atomic = 0;
  a.ksplit = g.ksplit; a.use_tr = g.use_tr; a.xcd_swizzle = 0; a.C2 = nullptr; a.aux_mul = 0; a.pre = 0;
  a.ws = g.ws ? g.ws + pr.ws_off : nullptr;     // partial tiles of slice z at ws[z·plane + ws_off ..]
  a.ws_plane = g.plane;
  a.rmw = g.ws ? 0 : 1;
  a.out_mode = pr.mode; a.out_scale = g.scale;
  gemm_fast_body<CT, BM, BN, 2, 2, BKT, NSET, LAYOUT_TN, false, KG>(a, local % pr.tiles_n, local / pr.tiles_n, slice, smem);
}

// Σ_z ws[z][e .. e+7]: ZL consecutive lanes share one 8-float group and take every ZL-th partial (independent loads,
// unrolled), then fold with shuffles.  (One thread per group walking all partials in turn was a chain of `nsplit`
// dependent L2 round trips: 36 us for a 96x96 gradient with 64 partials.)
template <int ZL>
__device__ __forceinline__ void splitk_sum(float (&acc)[8], const float* __restrict__ ws, size_t e, size_t plane, int nsplit, int zl) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 4
  for (int z = zl; z < nsplit; z += ZL) {
    float v[8];
    ld8(ws + (size_t)z * plane, SCOT_F32, e, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += v[j];
  }
#pragma unroll
  for (int o = 1; o < ZL; o <<= 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
}

// C[m][n] += Σ_z ws[z][m][n]
template <int ZL>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* C, int M, int N, int ldc, int nsplit) {
  const size_t n8 = (size_t)M * N / 8;
  const int zl = threadIdx.x % ZL;
  // all ZL lanes of a group run the same trip count (the shuffles need them converged)
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / ZL; i < n8; i += (size_t)gridDim.x * blockDim.x / ZL) {
    const size_t e = i * 8;
    const int m = e / N, n = e % N;
    float acc[8];
    splitk_sum<ZL>(acc, ws, e, (size_t)M * N, nsplit, zl);
    if (zl == 0) {
      float c[8];
      ld8(C, SCOT_F32, (size_t)m * ldc + n, c);
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] += acc[j];
      st8(C, SCOT_F32, (size_t)m * ldc + n, c);
    }
  }
}

template <typename CT, int BM, int BN, int WM, int WN, int BKT = FT<CT>::BK, int NSET = 2>
static int flaunch_layout(const FastArgs& a, int layout, int nsplit, hipStream_t s) {
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nsplit), block(256);
  switch (layout) {
    case LAYOUT_NT: hipLaunchKernelGGL((gemm_fast_kernel<CT, BM, BN, WM, WN, BKT, NSET, LAYOUT_NT>), grid, block, 0, s, a); break;
    case LAYOUT_NN: hipLaunchKernelGGL((gemm_fast_kernel<CT, BM, BN, WM, WN, BKT, NSET, LAYOUT_NN>), grid, block, 0, s, a); break;
    case LAYOUT_TN: hipLaunchKernelGGL((gemm_fast_kernel<CT, BM, BN, WM, WN, BKT, NSET, LAYOUT_TN>), grid, block, 0, s, a); break;
    default: return SCOT_ERR_UNSUPPORTED;
  }
  return scot_check_launch();
}
// tile ids: 0 = 64x64 (2x2 waves), 1 = 128x96 (4x1), 2 = 64x96 (2x2), 3 = 128x128 (2x2), 4 = 96x96 (2x2)... bf16 only beyond 0/3
template <typename CT> static int flaunch_tile(int tile, const FastArgs& a, int layout, int nsplit, hipStream_t s);
template <> int flaunch_tile<bf16_t>(int tile, const FastArgs& a, int layout, int nsplit, hipStream_t s) {
  switch (tile) {
    case 1: return flaunch_layout<bf16_t, 128, 96, 4, 1>(a, layout, nsplit, s);
    case 2: return flaunch_layout<bf16_t, 64, 96, 2, 2>(a, layout, nsplit, s);
    case 3: return flaunch_layout<bf16_t, 128, 128, 2, 2>(a, layout, nsplit, s);
    case 4: return flaunch_layout<bf16_t, 96, 96, 2, 2, 64>(a, layout, nsplit, s);
    case 5: return flaunch_layout<bf16_t, 96, 96, 2, 2, 32>(a, layout, nsplit, s);
    case 6: return flaunch_layout<bf16_t, 64, 64, 2, 2, 32>(a, layout, nsplit, s);
    case 7: return flaunch_layout<bf16_t, 128, 96, 4, 1, 32>(a, layout, nsplit, s);
    case 8: return flaunch_layout<bf16_t, 64, 64, 2, 2, 64, 4>(a, layout, nsplit, s);   // deep pipeline (small grids)
    case 9: {                                                                           // three direct-to-LDS stages (NT only)
      if (layout != LAYOUT_NT) return flaunch_layout<bf16_t, 64, 64, 2, 2>(a, layout, nsplit, s);
      dim3 grid((a.N + 63) / 64, (a.M + 63) / 64, nsplit), block(256);
      hipLaunchKernelGGL((gemm_fast_kernel<bf16_t, 64, 64, 2, 2, 64, 2, LAYOUT_NT, false, 1, 3>), grid, block, 0, s, a);
      return scot_check_launch();
    }
    default: return flaunch_layout<bf16_t, 64, 64, 2, 2>(a, layout, nsplit, s);
  }
}
template <> int flaunch_tile<float>(int tile, const FastArgs& a, int layout, int nsplit, hipStream_t s) {
  return tile == 3 ? flaunch_layout<float, 128, 128, 2, 2>(a, layout, nsplit, s) : flaunch_layout<float, 64, 64, 2, 2>(a, layout, nsplit, s);
}
// bf16x3 (fp32 operands in memory): 64x64, 64x96 and 96x96 tiles, BK = 32
template <int BM, int BN>
static int flaunch_x3_layout(const FastArgs& a, int layout, int nsplit, hipStream_t s) {
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nsplit), block(256);
  switch (layout) {
    case LAYOUT_NT: hipLaunchKernelGGL((gemm_fast_kernel<bf16_t, BM, BN, 2, 2, 32, 2, LAYOUT_NT, true>), grid, block, 0, s, a); break;
    case LAYOUT_NN: hipLaunchKernelGGL((gemm_fast_kernel<bf16_t, BM, BN, 2, 2, 32, 2, LAYOUT_NN, true>), grid, block, 0, s, a); break;
    case LAYOUT_TN: hipLaunchKernelGGL((gemm_fast_kernel<bf16_t, BM, BN, 2, 2, 32, 2, LAYOUT_TN, true>), grid, block, 0, s, a); break;
    default: return SCOT_ERR_UNSUPPORTED;
  }
  return scot_check_launch();
}
static int flaunch_x3(int tile, const FastArgs& a, int layout, int nsplit, hipStream_t s) {
  switch (tile) {
    case 2: return flaunch_x3_layout<64, 96>(a, layout, nsplit, s);
    case 4: return flaunch_x3_layout<96, 96>(a, layout, nsplit, s);
    default: return flaunch_x3_layout<64, 64>(a, layout, nsplit, s);
  }
}
static void tile_dims(int tile, int& bm, int& bn, int& bkt) {
  bkt = 64;
  switch (tile) {
    case 1: bm = 128; bn = 96; break;
    case 2: bm = 64; bn = 96; break;
    case 3: bm = 128; bn = 128; break;
    case 4: bm = 96; bn = 96; break;
    case 5: bm = 96; bn = 96; bkt = 32; break;
    case 6: bm = 64; bn = 64; bkt = 32; break;
    case 8: case 9: bm = 64; bn = 64; break;
    case 7: bm = 128; bn = 96; bkt = 32; break;
    default: bm = 64; bn = 64;
  }
}

extern int g_scot_use_tr;

// K slices of the fp32-result NT products (see gemm_fast_impl).  g_nt_splitk: 0 = the policy below, -1 = never, S > 0 = S slices for
// every eligible call (sweeps: tools/bench_deep_gemm.py).
static int g_nt_splitk = 0, g_nt_splitk_zeroed = 1;
extern "C" void scot_gemm_splitk_config(int slices, int zeroed_too) { g_nt_splitk = slices; g_nt_splitk_zeroed = zeroed_too; }
static int nt_splitk_policy(int M, int N, int K, int accumulate, long tiles, long nkt) {
  if (g_nt_splitk < 0) return 1;
  if (!accumulate && !g_nt_splitk_zeroed) return 1;
  if (g_nt_splitk > 0) return nkt >= 2 * g_nt_splitk ? g_nt_splitk : 1;
  // profiles/round6/splitk_atomic_probe.txt (us per launch, operands cold as in the step, unsplit -> 2 / 3 / 4 slices): the atomics cost
  // ~1.8 us per slice per 0.8 M result elements, so only two slices of a long contraction into a small accumulation target pay —
  // [1024, 768] += K 3072: 21.8 -> 16.8 / 18.7 / 18.0;  K 2304: 16.6 -> 13.8 / 15.7 / 16.2;  [4096, 384] += K 1536: 17.7 -> 18.5 (no);
  // the zeroed forward form never does ([1024, 768] = K 3072: 21.8 -> 24.2 with the memset in front).
  // ... and in the step those two launches change nothing (profiles/round6/splitk_in_step_ab.txt: 18.70 / 18.55 ms without, 18.54 / 18.53 with,
  // `gemm NT` family 7.00-7.10 ms either way), so the policy splits NOTHING: results stay bit-reproducible run to run, the mechanism stays
  // for scot_gemm_splitk_config (tests, tools/bench_splitk.py).
  (void)M; (void)N; (void)K; (void)tiles;
  return 1;
}

// Returns SCOT_ERR_UNSUPPORTED when the call does not qualify (the caller then uses the generic kernel).

// `query` != NULL: plan only — the tile / split policy below runs against an unlimited workspace and *query receives the bytes it
// would use; nothing is launched (scot_gemm_workspace_bytes: one policy, two readers).
static int gemm_fast_impl(int layout, int compute, int M, int N, int K, const void* A, int a_dt, int lda, int a_gelu,
                          const void* B, int b_dt, int ldb, int b_gelu, void* C, int c_dt, int ldc, const float* bias,
                          const float* colscale, const void* aux, int aux_dt, int ldaux, const void* resid, int res_dt, int ldres,
                          int accumulate, float* colsum_out, void* workspace, size_t ws_bytes, int aux_mul, void* C2, hipStream_t stream,
                          size_t* query) {
  if (query) { *query = 0; workspace = (void*)(uintptr_t)64; ws_bytes = (size_t)1 << 60; }
  const bool x3 = compute == SCOT_BF16X3;
  const int want = compute == SCOT_BF16 ? SCOT_BF16 : SCOT_F32;   // bf16x3 keeps its operands in fp32
  const int epc = compute == SCOT_BF16 ? 8 : 4;
  if (a_dt != want || b_dt != want) return SCOT_ERR_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)aux | (uintptr_t)resid) & 15) != 0) return SCOT_ERR_UNSUPPORTED;
  if (lda % epc || ldb % epc || ldc % 8 || (aux && ldaux % 8) || (resid && ldres % 8) || N % 8 || K % epc) return SCOT_ERR_UNSUPPORTED;
  if (layout == LAYOUT_TN ? (M % epc || M < epc) : false) return SCOT_ERR_UNSUPPORTED;
  if (layout != LAYOUT_NT && N < epc) return SCOT_ERR_UNSUPPORTED;
  if (K < epc || M < 1) return SCOT_ERR_UNSUPPORTED;
  FastArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.colscale = colscale; a.aux = aux; a.resid = resid; a.colsum_out = colsum_out;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux; a.ldres = ldres;
  a.c_dt = c_dt; a.aux_dt = aux_dt; a.res_dt = res_dt; a.a_gelu = a_gelu; a.b_gelu = b_gelu; a.aux_gelu_grad = aux != nullptr;
  a.use_tr = g_scot_use_tr; a.atomic = 0; a.ws = nullptr; a.ws_plane = 0; a.rmw = 0; a.C2 = C2; a.aux_mul = aux_mul;
  a.out_mode = 0; a.out_scale = nullptr;
  a.xcd_swizzle = 1;
  a.pre = 1;         // epilogue operands requested before the K loop (round 4: dgrad fc2 · gelu' 43.2 -> 36.6 us in step)
  if (C2 && ((((uintptr_t)C2) & 15) != 0 || layout == LAYOUT_TN)) return SCOT_ERR_UNSUPPORTED;
  if (a_gelu || b_gelu) return SCOT_ERR_UNSUPPORTED;   // GELU-on-load is the general kernel's (the engine stores GELU(u) from the fc1 epilogue)
  int bk = compute == SCOT_BF16 ? 64 : 32;
  int nsplit = 1;
  // tile choice: the per-layout policy below (the SCOT_GEMM_TILE* / SCOT_GEMM_GLDS overrides of rounds 1-4 are retired: their sweeps are
  // under profiles/round2..4 — micro_deep_gemm_tiles_r3.txt, gemm_lds_ring_depth_r4.txt — and in profiles/HISTORY.md)
  int tile = -1;
  const int glds = 1;   // direct-to-LDS K loop for the NT products it covers (register-staged: +0.15 ms per step, round 3)
  const int deep = 1;   // four-register-set pipeline for the long-K small-grid products (round 3, in step: 19.76 -> 19.59 ms)
  if (tile < 0) {
    tile = 0;   // policy (see DESIGN.md §3 for the measurements behind it)
    // wgrad with a long token dimension (stages 0/1): 96x96 (cold-cache sweep: 25.6 vs 38.5 us at stage 1); with K <= 4096
    // (stages 2/3) the 64x64 grid is already large enough to run unsplit (no partials, no reduce pass): 22 vs 27 us
    if ((compute == SCOT_BF16 || x3) && layout == LAYOUT_TN && M % 96 == 0 && N % 96 == 0 && K >= 8192) tile = 4;
    else if ((compute == SCOT_BF16 || x3) && N == 96) tile = 2;   // one 64x96 column tile: the A operand streams once (64x64 would read it twice)
    else if (compute == SCOT_BF16 && layout != LAYOUT_TN && K <= 128) tile = 6;   // K = 96: BK = 32 halves LDS -> more workgroups/CU (-12 %)
    // small grids walking a long contraction (stage 3: fc2 forward, fc1 / qkv data gradients — 192 workgroups x 36-48 K-tiles): four
    // register sets of loads in flight instead of two (26.4 -> 22.4 / 24.5 -> 21.7 us alone; shorter contractions lose to the padded trip count)
    else if (compute == SCOT_BF16 && layout != LAYOUT_TN && K >= 36 * 64 && (long)((M + 63) / 64) * ((N + 63) / 64) <= 512 && deep) tile = 8;
  }
  // 64 x 64-tile NT products whose K is whole tiles: three LDS stages filled by global_load_lds (alone -6 % at K = 384, -14..21 % from
  // K = 1536; in step 19.60 -> 19.45 ms; two stages lose to the register pipeline in step, four to three)
  if (glds && compute == SCOT_BF16 && layout == LAYOUT_NT && (tile == 0 || tile == 8) && K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0) tile = 9;
  if (x3 && tile != 2 && tile != 4) tile = 0;                     // bf16x3 instantiates 64x64, 64x96, 96x96
  if (compute == SCOT_F32 && tile != 0 && tile != 3) tile = 0;    // fp32 instantiates 64x64 and 128x128 only
  int bm, bn, bkt;
  tile_dims(tile, bm, bn, bkt);
  if (compute == SCOT_BF16) bk = bkt;
  a.ksplit = ((K + bk - 1) / bk) * bk;
  const long tiles = (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  const long nkt = (K + bk - 1) / bk;
  if (layout == LAYOUT_TN) {
    // wgrad: small output, contraction over all tokens.  Split K so that ~512 workgroups stream the operands; each split
    // writes a partial tile into the workspace and ONE reduce pass adds them into the gradient (12.6 M fp32 atomics per
    // call in the first version of this kernel cost 300 us; the partials cost < 20 MB of traffic).
    if (c_dt != SCOT_F32 || !accumulate) return SCOT_ERR_UNSUPPORTED;
    const int tn_wgs = 512;
    long wantsplit = (tn_wgs + tiles - 1) / tiles;
    const long maxsplit = (K + 8 * bk - 1) / (8 * bk);       // >= 8 K-tiles per workgroup
    long wsmax = workspace ? (long)(ws_bytes / ((size_t)M * N * sizeof(float))) : 1;
    nsplit = (int)(wantsplit < 1 ? 1 : (wantsplit > maxsplit ? maxsplit : wantsplit));
    if (nsplit > wsmax) nsplit = (int)(wsmax < 1 ? 1 : wsmax);
    int per = (K + nsplit - 1) / nsplit;
    per = ((per + bk - 1) / bk) * bk;
    if (nsplit >= 8) {   // make the split count a multiple of 8 so that the XCD remap applies (one token chunk per XCD at a time)
      for (int tries = 0; tries < 64 && ((K + per - 1) / per) % 8 != 0; ++tries) per += bk;
      if (((K + per - 1) / per) % 8 != 0) per = ((((K + nsplit - 1) / nsplit) + bk - 1) / bk) * bk;
    }
    a.ksplit = per;
    nsplit = (K + per - 1) / per;
    if (nsplit == 1) a.rmw = 1;
    else if (workspace && (((uintptr_t)workspace & 31) == 0)) a.ws = (float*)workspace;
    else a.atomic = 1;
  } else {
    // K slices with fp32 atomics into the result (round 6; the hand-off-free form of split-K: partial planes + an epilogue pass lost in
    // rounds 2-3, an in-launch ticket + combine in round 5).  Only where the result is fp32 and the epilogue is bias-only: the deep stages'
    // fc2 forward (result zeroed here first) and the fc1 / qkv data gradients, which accumulate into the fp32 residual-stream gradient.
    int S = 1;
    const bool split_ok = compute == SCOT_BF16 && layout == LAYOUT_NT && c_dt == SCOT_F32 && !aux && !C2 && !colsum_out && !colscale &&
                          resid == nullptr && (accumulate || ldc == N) && (tile == 9 || tile == 8 || tile == 0);
    if (split_ok) S = nt_splitk_policy(M, N, K, accumulate, tiles, nkt);
    if (S > 1) {
      const long per = (nkt + S - 1) / S;
      a.ksplit = (int)(per * bk);
      nsplit = (int)((nkt + per - 1) / per);
    }
    if (nsplit > 1) {
      a.atomic = 1;
      if (!accumulate && !query && hipMemsetAsync(C, 0, (size_t)M * N * sizeof(float), stream) != hipSuccess) return SCOT_ERR_LAUNCH;
    } else if (accumulate) {
      if (resid != nullptr) return SCOT_ERR_UNSUPPORTED;
      a.resid = C; a.res_dt = c_dt; a.ldres = ldc;
    }
  }
  if (query) {
    *query = a.ws ? (size_t)nsplit * M * N * sizeof(float) : 0;
    return SCOT_OK;
  }
  int rc = x3 ? flaunch_x3(tile, a, layout, nsplit, stream)
              : compute == SCOT_BF16 ? flaunch_tile<bf16_t>(tile, a, layout, nsplit, stream) : flaunch_tile<float>(tile, a, layout, nsplit, stream);
  if (rc == SCOT_OK && a.ws) {
    const size_t n8 = (size_t)M * N / 8;
    const int zl = (nsplit >= 64 && layout == LAYOUT_TN) ? 32 : nsplit >= 16 ? 8 : nsplit >= 4 ? 4 : 1;
    size_t blocks = (n8 * zl + 255) / 256; if (blocks > 4096) blocks = 4096;
    const dim3 g((unsigned)blocks), b(256);
    if (zl == 32) hipLaunchKernelGGL(splitk_reduce_kernel<32>, g, b, 0, stream, a.ws, (float*)C, M, N, ldc, nsplit);      // (only TN products split K)
    else if (zl == 8) hipLaunchKernelGGL(splitk_reduce_kernel<8>, g, b, 0, stream, a.ws, (float*)C, M, N, ldc, nsplit);
    else if (zl == 4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, g, b, 0, stream, a.ws, (float*)C, M, N, ldc, nsplit);
    else hipLaunchKernelGGL(splitk_reduce_kernel<1>, g, b, 0, stream, a.ws, (float*)C, M, N, ldc, nsplit);
    rc = scot_check_launch();
  }
  return rc;
}


int scot_gemm_fast(int layout, int compute, int M, int N, int K, const void* A, int a_dt, int lda, int a_gelu,
                   const void* B, int b_dt, int ldb, int b_gelu, void* C, int c_dt, int ldc, const float* bias,
                   const float* colscale, const void* aux, int aux_dt, int ldaux, const void* resid, int res_dt, int ldres,
                   int accumulate, float* colsum_out, void* workspace, size_t ws_bytes, int aux_mul, void* C2, hipStream_t stream) {
  return gemm_fast_impl(layout, compute, M, N, K, A, a_dt, lda, a_gelu, B, b_dt, ldb, b_gelu, C, c_dt, ldc, bias, colscale, aux, aux_dt,
                        ldaux, resid, res_dt, ldres, accumulate, colsum_out, workspace, ws_bytes, aux_mul, C2, stream, nullptr);
}

// include/scot_hip.h: scot_gemm_workspace_bytes — dense operands (leading dimensions = row lengths) in the compute mode's
// operand type, fp32 result for TN (accumulate) / 16-bit otherwise; 0 = the call would not touch the workspace.
extern "C" size_t scot_gemm_workspace_bytes(int layout, int compute, int M, int N, int K) {
  const int dt = compute == SCOT_BF16 ? SCOT_BF16 : SCOT_F32;
  const int lda = layout == LAYOUT_TN ? M : K, ldb = layout == LAYOUT_NT ? K : N;
  void* al = (void*)(uintptr_t)64;
  size_t q = 0;
  const int rc = gemm_fast_impl(layout, compute, M, N, K, al, dt, lda, 0, al, dt, ldb, 0, al, layout == LAYOUT_TN ? SCOT_F32 : dt, N, nullptr,
                                nullptr, nullptr, 0, 0, nullptr, 0, 0, layout == LAYOUT_TN ? 1 : 0, nullptr, nullptr, 0, 0, nullptr, nullptr, &q);
  return rc == SCOT_OK ? q : 0;
}

// C_i[m][n] += Σ_z ws[z·plane + ws_off_i + m·N_i + n] for every problem of the group (one launch)
template <int ZL>
__global__ __launch_bounds__(256) void wgrad_group_reduce_kernel(WgradGroupArgs g) {
  const size_t n8 = g.plane / 8;
  const int zl = threadIdx.x % ZL;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / ZL; i < n8; i += (size_t)gridDim.x * blockDim.x / ZL) {
    const size_t e = i * 8;
    int q = 0;
#pragma unroll
    for (int k = 1; k < SCOT_WGRAD_GROUP_MAX; ++k) q += (k < g.n && e >= g.p[k].ws_off) ? 1 : 0;
    const WgradProblem& pr = g.p[q];
    const size_t le = e - pr.ws_off;
    const int m = le / pr.N, n = le % pr.N;
    float acc[8];
    splitk_sum<ZL>(acc, g.ws, e, g.plane, g.nsplit, zl);
    if (zl == 0) grad_commit8(pr.C, (size_t)m * pr.ldc + n, acc, pr.mode, g.scale);
  }
}

template <int BM, int BN, int BKT, int NSET, int KG = 1>
static int launch_wgrad_group(const WgradGroupArgs& g, hipStream_t s) {
  hipLaunchKernelGGL((wgrad_group_kernel<bf16_t, BM, BN, BKT, NSET, KG>), dim3((unsigned)(g.tiles * g.nsplit)), dim3(256 * KG), 0, s, g);
  return scot_check_launch();
}

int scot_wgrad_group_wide_launch(const WgradGroupArgs& g, int variant, hipStream_t s);      // wgrad_wide.hip
int scot_gemm_wide_mode(int* variant);                                                       // gemm_wide.hip
// include/scot_hip.h: scot_wgrad_group.  dY_i: [K, M_i] (16-bit operands), X_i: [K, N_i], dW_i: [M_i, N_i] fp32 (+=),
// dbias_i: [M_i] fp32 (+= column sums of dY_i) or NULL.  All leading dimensions = the row lengths (dense).
static int wgrad_group_impl(int compute, int n, int K, const void* const* dY, const void* const* X, float* const* dW,
                            float* const* dbias, const int* Ms, const int* Ns, void* workspace, size_t ws_bytes,
                            const int* modes, const float* grad_scale, hipStream_t stream, size_t* query) {
  if (query) { *query = 0; workspace = (void*)(uintptr_t)64; ws_bytes = (size_t)1 << 60; }
  if (n <= 0 || n > SCOT_WGRAD_GROUP_MAX || K <= 0) return SCOT_ERR_SHAPE;
  if (compute != SCOT_BF16) return SCOT_ERR_UNSUPPORTED;     // fp32 / split modes use scot_gemm per problem
  if (K % 8) return SCOT_ERR_UNSUPPORTED;
  bool all96 = true;
  for (int i = 0; i < n; ++i) {
    if (Ms[i] <= 0 || Ns[i] <= 0) return SCOT_ERR_SHAPE;
    if (Ms[i] % 8 || Ns[i] % 8) return SCOT_ERR_UNSUPPORTED;
    if (!query && (((uintptr_t)dY[i] | (uintptr_t)X[i] | (uintptr_t)dW[i]) & 15) != 0) return SCOT_ERR_UNSUPPORTED;
    all96 = all96 && Ms[i] % 96 == 0 && Ns[i] % 96 == 0;
  }
  // 128 x 128 tiles (wgrad_wide.hip) for groups of 128-multiples: eight waves, two LDS stages (two workgroups per CU) — unsplit from 256 tiles,
  // below that, from 8192 tokens, with K cut so that ~860 workgroups exist (>= 8 K-tiles each).  profiles/round5/wgrad_wide_sweep_r5.txt, us per launch,
  // 64 x 64 grouped kernel -> this: Poseidon-B stage 3 57.0 -> 42.8, stage 2 (108 tiles) 58.6 -> 54.4; Poseidon-L 299.8 -> 210.1,
  // 393.5 -> 267.2, 433.7 -> 293.9 (four waves or four stages lose everywhere: 243.9 / 347.5 at L's stage 3)
  int wide = -1, wide_split = 1;
  {
    int forced = 0;
    const int mode = scot_gemm_wide_mode(&forced);
    bool all128 = K % 64 == 0;
    long t128 = 0;
    for (int i = 0; i < n; ++i) { all128 = all128 && Ms[i] % 128 == 0 && Ns[i] % 128 == 0; t128 += (long)(Ms[i] / 128) * (Ns[i] / 128); }
    const long nkt128 = K / 64;
    if (mode != 0 && all128 && t128 > 0) {
      if (t128 >= 256) wide = 1;
      else if (t128 >= 64 && nkt128 >= 128) {      // (shorter K — Poseidon-B's stage 2, 4096 tokens: 58.6 -> 54.4 us alone for 113 MB of partial tiles
                                                   //  written and read back per launch, nothing in step: stays on the unsplit 64 x 64 kernel)
        wide = 1;
        wide_split = (int)((864 + t128 - 1) / t128);
        if (wide_split > nkt128 / 8) wide_split = (int)(nkt128 / 8);
      }
      if (mode == 2) {      // tests / sweeps: bits 0-3 of the forced variant = kernel instantiation, bits 4.. = K slices (0: the policy's)
        wide = forced & 15;
        if (forced >> 4) wide_split = forced >> 4;
      }
    }
  }
  // tile policy of the single-problem path: 96x96 for the long-K gradients of the token-heavy stages, 64x64 otherwise
  const bool t96 = wide < 0 && all96 && K >= 8192;
  const int bm = wide >= 0 ? 128 : (t96 ? 96 : 64), bn = bm, bk = 64;
  WgradGroupArgs g;
  g.n = n; g.K = K; g.use_tr = g_scot_use_tr;
  int tiles = 0;
  size_t plane = 0;
  for (int i = 0; i < n; ++i) {
    WgradProblem& p = g.p[i];
    p.A = query ? nullptr : dY[i]; p.B = query ? nullptr : X[i]; p.C = query ? nullptr : dW[i]; p.colsum = (dbias && !query) ? dbias[i] : nullptr;
    p.M = Ms[i]; p.N = Ns[i]; p.lda = Ms[i]; p.ldb = Ns[i]; p.ldc = Ns[i];
    p.mode = modes ? modes[i] : SCOT_GRAD_ADD;
    if (p.mode < 0 || p.mode > SCOT_GRAD_ADD_SCALED) return SCOT_ERR_SHAPE;
    p.tiles_n = (Ns[i] + bn - 1) / bn;
    p.tile0 = tiles;
    p.ws_off = (unsigned)plane;
    tiles += p.tiles_n * ((Ms[i] + bm - 1) / bm);
    plane += (size_t)Ms[i] * Ns[i];
  }
  for (int i = n; i < SCOT_WGRAD_GROUP_MAX; ++i) { g.p[i] = g.p[0]; g.p[i].tile0 = 0x7fffffff; g.p[i].ws_off = 0xffffffffu; }
  g.tiles = tiles; g.plane = plane; g.scale = grad_scale;
  // K slices: enough workgroups to fill the chip (~2 per CU), at least 8 K-tiles each, a multiple of 8 so that one slice's
  // tiles share an XCD; none when the group already has >= 256 tiles
  const int want_wgs = 256;   // in-step optimum: 192-256 (448 filled the chip better alone and cost the chain 0.1 ms; 128 makes the side stream the wall)
  const long nkt = (K + bk - 1) / bk;
  // (groups with >= 256 tiles — the deep stages — are never split: two K slices for the 432-tile stage-2 group run 66 instead of 85 us
  // alone and cost the step 0.2 ms; the eight-wave workgroups below halve its serial K loop without a second pass)
  long nsplit = tiles >= 256 ? 1 : (want_wgs + tiles - 1) / tiles;
  if (wide >= 0) nsplit = wide_split;
  if (nsplit < 1) nsplit = 1;
  const long maxsplit = nkt / 8 > 0 ? nkt / 8 : 1;
  if (nsplit > maxsplit) nsplit = maxsplit;
  const long wsmax = (workspace && plane) ? (long)(ws_bytes / (plane * sizeof(float))) : 1;
  if (nsplit > wsmax) nsplit = wsmax < 1 ? 1 : wsmax;
  int per = (int)(((K + nsplit - 1) / nsplit + bk - 1) / bk * bk);
  if (nsplit >= 8 && wide < 0) {
    for (int tries = 0; tries < 64 && ((K + per - 1) / per) % 8 != 0; ++tries) per += bk;
    if (((K + per - 1) / per) % 8 != 0) per = (int)(((K + nsplit - 1) / nsplit + bk - 1) / bk * bk);
  }
  g.ksplit = per;
  g.nsplit = (K + per - 1) / per;
  g.ws = nullptr;
  if (g.nsplit > 1) {
    if (!workspace || (((uintptr_t)workspace) & 31) || (size_t)g.nsplit * plane * sizeof(float) > ws_bytes) return SCOT_ERR_UNSUPPORTED;
    g.ws = (float*)workspace;
  }
  if (query) {
    *query = g.nsplit > 1 ? (size_t)g.nsplit * plane * sizeof(float) : 0;
    return SCOT_OK;
  }
  // unsplit 64x64-tile groups (the deep stages: 432 / 1728 tiles walking 64 / 16 K-tiles each): two K groups per workgroup
  const int kg_env = 2;   // (measured: stage 2 84.5 -> 62.5 us alone, step -0.12 ms against four waves)
  int rc;
  if (wide >= 0) rc = scot_wgrad_group_wide_launch(g, wide, stream);
  else if (!t96 && g.nsplit == 1 && kg_env == 2 && nkt >= 4) rc = launch_wgrad_group<64, 64, 64, 2, 2>(g, stream);
  else rc = t96 ? launch_wgrad_group<96, 96, 64, 2>(g, stream) : launch_wgrad_group<64, 64, 64, 2>(g, stream);
  if (rc == SCOT_OK && g.ws) {
    const size_t n8 = plane / 8;
    const int zl = g.nsplit >= 32 ? 8 : g.nsplit >= 4 ? 4 : 1;
    size_t blocks = (n8 * zl + 255) / 256; if (blocks > 4096) blocks = 4096;
    const dim3 gr((unsigned)blocks), b(256);
    if (zl == 8) hipLaunchKernelGGL(wgrad_group_reduce_kernel<8>, gr, b, 0, stream, g);
    else if (zl == 4) hipLaunchKernelGGL(wgrad_group_reduce_kernel<4>, gr, b, 0, stream, g);
    else hipLaunchKernelGGL(wgrad_group_reduce_kernel<1>, gr, b, 0, stream, g);
    rc = scot_check_launch();
  }
  return rc;
}

extern "C" int scot_wgrad_group(int compute, int n, int K, const void* const* dY, const void* const* X, float* const* dW,
                                float* const* dbias, const int* Ms, const int* Ns, void* workspace, size_t ws_bytes,
                                const int* modes, const float* grad_scale, hipStream_t stream) {
  return wgrad_group_impl(compute, n, K, dY, X, dW, dbias, Ms, Ns, workspace, ws_bytes, modes, grad_scale, stream, nullptr);
}

// include/scot_hip.h: scot_wgrad_group_workspace_bytes (0: no split, or the shapes are not covered by the grouped kernel)
extern "C" size_t scot_wgrad_group_workspace_bytes(int n, int K, const int* Ms, const int* Ns) {
  size_t q = 0;
  const int rc = wgrad_group_impl(SCOT_BF16, n, K, nullptr, nullptr, nullptr, nullptr, Ms, Ns, nullptr, 0, nullptr, nullptr, nullptr, &q);
  return rc == SCOT_OK ? q : 0;
}
