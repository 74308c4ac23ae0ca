// gemm_wide — NT products on 128 x 128 output tiles (16-bit operands, K-contiguous: A [M, K], B [N, K]; M % 128 == 0, N % 128 == 0,
// K % 64 == 0).  gemm_fast's 64 x 64 tiles move every operand byte through L2 -> LDS twice as often (32 flop per ingested byte against
// 64 here); a CU ingests ~45-60 GB/s with one resident workgroup and ~90-100 with two or three, whatever the tile (round-5 sweeps,
// profiles/round5/gemm_wide_sweep_*.txt), so the wide tile pays exactly where its grid still gives every CU two workgroups — the
// Linear layers of Poseidon-L (C = 384 .. 1536 at 2048 .. 32768 token rows: -20 .. -48 % per launch, up to 800 TF/s) and of
// Poseidon-B at 256 x 256 — and, with one workgroup per CU, for long contractions and the gelu'-scaled data gradient at 128-256 tiles.
// Three instantiations of one kernel:
//   8 waves (64 x 32 each), 4 LDS stages of 32 KB (131 KB: one workgroup per CU), epilogue operands requested before the K loop;
//   8 waves, 2 stages (68 KB with the epilogue's C tile: two workgroups per CU);   4 waves (64 x 64 each), 2 stages.
// K loop as gemm_fast's direct-to-LDS loop: global_load_lds_dwordx4 (1 KB = 8 tile rows per wave instruction, the 16-byte chunk swizzle
// applied on the SOURCE side), counted vmcnt waits, one raw s_barrier per K-tile.
// MEASURED AND REMOVED (round 5): cutting K into slices inside the launch for outputs with < 128 tiles (fp32 slabs in the workspace,
// per-tile arrival counter, the last arriver adds the slabs in slice order and runs the epilogue).  Correct (emulator + GPU tests,
// bit-identical run to run) and slower than the 64 x 64 tiles at every Poseidon-B / -L shape: with an agent-scope release fence per
// workgroup the launch doubled (every `buffer_wbl2` writes back the XCD's whole L2), with write-through slab stores (sc1) + one acquire
// the hand-off still cost 6-9 us per launch — the last arriver reads (S - 1) x 64 KB alone — against 3-5 us saved in the K loop.
// Reference shapes: the Linear layers of HF modeling_swinv2.py:396-410, 496-506, 536-561 at the widths of scOT/model.py:403-404 and
// scOT/train.py:35-72 (MODEL_MAP).
#include "common.h"
#include <stdlib.h>

#define LAYOUT_NT 0

struct WideArgs {
  const bf16_t* A; const bf16_t* B; void* C; void* C2;
  const float* bias; const void* aux; const void* resid;
  int M, N, K, lda, ldb, ldc, ldaux, ldres;
  int c_dt, aux_mul;
  int tiles, tiles_n, tpx;   // output tiles, tiles along N, tiles per XCD chunk
  int colmajor;              // tile order inside an XCD's chunk: 0 = along N (shares the A row block), 1 = along M (shares the B rows)
};

template <int BM, int BN, int STAGES> struct WideLds {
  static constexpr int STAGE = (BM + BN) * 64;                      // 16-bit elements per stage
  static constexpr size_t AB = (size_t)STAGES * STAGE * 2, C = (size_t)BM * (BN + 4) * 4;
  static constexpr size_t bytes = AB > C ? AB : C;                  // ONE __shared__ object (a second one makes hipcc drain the DMA queue before every ds_read)
};

// fragment of a swizzled K-contiguous tile (128-byte rows, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7)): row r0 + (lane & 15)
__device__ __forceinline__ Frag<bf16_t> wide_frag(const bf16_t* t, int r0, int kk, int lane) {
  Frag<bf16_t> f;
  const int r = lane & 15, ch = (kk >> 3) + (lane >> 4);
  f.v = *(const s16x8_t*)(t + (r0 + r) * 64 + ((ch ^ (r >> 1)) << 3));
  return f;
}

// EPI: which epilogue operand rows the kernel reads — 0 none, 1 `aux` (16-bit: gelu'(u) of a data gradient), 2 `resid` (fp32: the tensor a
// data gradient is accumulated into).  A template parameter so that their loads are straight-line code (a runtime `have_aux` inside the
// unrolled loop made hipcc branch around every load and wait for each).  PRE: request them in front of the K loop (one resident
// workgroup per CU: nobody else hides the epilogue's round trip) instead of inside the epilogue (two per CU: the registers cost the
// second workgroup).
#define WIDE_EPI_NONE 0
#define WIDE_EPI_AUX16 1
#define WIDE_EPI_RES32 2
// WM x WN: arrangement of the NW = WM·WN waves over the tile; a wave owns (BM / WM) x (BN / WN).
template <int BM, int BN, int WM, int WN, int STAGES, int EPI, bool PRE>
__global__ __launch_bounds__(WM * WN * 64) void gemm_wide_kernel(WideArgs p) {
  constexpr int NW = WM * WN, NT = NW * 64, MI = BM / WM / 16, NI = BN / WN / 16, WROWS = BM / WM, WCOLS = BN / WN;
  constexpr int STAGE = WideLds<BM, BN, STAGES>::STAGE, BOFF = BM * 64;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW, LPW = PA + PB;      // 1 KB pieces (8 tile rows) per wave and K-tile
  constexpr int CP = BN + 4;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "pieces per wave");
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  __shared__ __attribute__((aligned(1024))) char smem[WideLds<BM, BN, STAGES>::bytes];
  bf16_t* lds = (bf16_t*)smem;

  // workgroup b runs on XCD b % 8 (dispatch order: speed only).  An XCD owns a contiguous chunk of `tpx` tiles: they share operand
  // rows in its L2
  const int L = blockIdx.x, xcd = L & 7, tl = L >> 3;
  const int tile = xcd * p.tpx + tl;
  if (tile >= p.tiles) return;
  int by, bx;
  if (p.colmajor) { const int tm = p.tiles / p.tiles_n; bx = tile / tm; by = tile % tm; }
  else { by = tile / p.tiles_n; bx = tile % p.tiles_n; }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN, g = lane >> 4;
  const int m0 = by * BM, n0 = bx * BN;
  const int nk = p.K >> 6;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) acc[i][jj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // epilogue geometry: a thread owns one 8-column chunk (cc) of the rows tid / CPRW + it * RPP
  constexpr int CPRW = BN / 8, RPP = NT / CPRW, E_IT = BM / RPP;
  static_assert(NT % CPRW == 0 && BM % RPP == 0, "epilogue passes");
  const int cc = tid % CPRW, erow = tid / CPRW;
  const int col = n0 + cc * 8;
  constexpr bool HOLD = PRE && EPI != WIDE_EPI_NONE;
  constexpr int EW = EPI == WIDE_EPI_RES32 ? 2 : 1;
  uint4 eop[HOLD ? E_IT : 1][EW];
  auto load_epilogue_operand = [&](uint4 (&dst)[EW], int it) {
    const size_t grow = (size_t)(m0 + erow + it * RPP);
    if constexpr (EPI == WIDE_EPI_AUX16) dst[0] = *(const uint4*)((const bf16_t*)p.aux + grow * p.ldaux + col);
    if constexpr (EPI == WIDE_EPI_RES32) {
      dst[0] = *(const uint4*)((const float*)p.resid + grow * p.ldres + col);
      dst[1] = *(const uint4*)((const float*)p.resid + grow * p.ldres + col + 4);
    }
  };
  if constexpr (HOLD) {      // older than every direct-to-LDS load: the loop's counted waits retire them first
#pragma unroll
    for (int it = 0; it < E_IT; ++it) load_epilogue_operand(eop[it], it);
  }

  typedef __attribute__((address_space(3))) void* lds_p;
  typedef __attribute__((address_space(1))) const void* gbl_p;
  auto issue = [&](bf16_t* st, int kt) {
    const int k0 = kt << 6;
#pragma unroll
    for (int u = 0; u < PA; ++u) {
      const int q = wave + NW * u, row = 8 * q + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      const bf16_t* src = p.A + (size_t)(m0 + row) * p.lda + k0 + c * 8;
      __builtin_amdgcn_global_load_lds((gbl_p)src, (lds_p)(st + q * 512), 16, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const int q = wave + NW * u, row = 8 * q + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      const bf16_t* src = p.B + (size_t)(n0 + row) * p.ldb + k0 + c * 8;
      __builtin_amdgcn_global_load_lds((gbl_p)src, (lds_p)(st + BOFF + q * 512), 16, 0, 0);
    }
  };
  auto compute = [&](const bf16_t* st) {
    const bf16_t* As = st;
    const bf16_t* Bs = st + BOFF;
#pragma unroll
    for (int kk = 0; kk < 64; kk += 32) {
      Frag<bf16_t> fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = wide_frag(As, wr * WROWS + i * 16, kk, lane);
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) fb[jj] = wide_frag(Bs, wc * WCOLS + jj * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) mma16(acc[i][jj], fa[i], fb[jj]);
    }
  };

  // STAGES-deep ring: tiles t+1 .. t+STAGES-1 are in flight while tile t is multiplied; the counted wait retires tile t only, the raw
  // barrier behind it makes every wave's pieces visible and frees the stage read during iteration t-1 (gemm_fast's direct-to-LDS loop)
#pragma unroll
  for (int u = 0; u < STAGES - 1; ++u)
    if (u < nk) issue(lds + u * STAGE, u);
  for (int t = 0; t < nk; ++t) {
    const int newer = min(STAGES - 2, nk - 1 - t);
    if (newer >= 2) SCOT_VMCNT(2 * LPW);
    else if (newer == 1) SCOT_VMCNT(LPW);
    else SCOT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    if (t + STAGES - 1 < nk) issue(lds + ((t + STAGES - 1) % STAGES) * STAGE, t + STAGES - 1);
    compute(lds + (t % STAGES) * STAGE);
  }

  // ---- epilogue through LDS (the C tile aliases the stages): 16- / 32-byte row segments per thread
  __syncthreads();
  float* Cs = (float*)smem;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int jj = 0; jj < NI; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wr * WROWS + i * 16 + g * 4 + r) * CP + wc * WCOLS + jj * 16 + (lane & 15)] = acc[i][jj][r];
  __syncthreads();
  float bv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bv[q] = p.bias ? p.bias[col + q] : 0.f;
#pragma unroll
  for (int it = 0; it < E_IT; ++it) {
    const int row = erow + it * RPP;
    const size_t grow = (size_t)(m0 + row);
    uint4 eo[EW];
    if constexpr (EPI != WIDE_EPI_NONE) {
      if constexpr (HOLD) {
#pragma unroll
        for (int q = 0; q < EW; ++q) eo[q] = eop[it][q];
      } else {
        load_epilogue_operand(eo, it);
      }
    }
    float v[8];
    const float4 a = *(const float4*)(Cs + row * CP + cc * 8), b = *(const float4*)(Cs + row * CP + cc * 8 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] += bv[q];
    if constexpr (EPI == WIDE_EPI_AUX16) {
      float x[8];
      const uint4 u = eo[0];
      unpack_bf16x2(u.x, x[0], x[1]); unpack_bf16x2(u.y, x[2], x[3]); unpack_bf16x2(u.z, x[4], x[5]); unpack_bf16x2(u.w, x[6], x[7]);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] *= p.aux_mul ? x[q] : gelu_grad_f(x[q]);
    }
    if constexpr (EPI == WIDE_EPI_RES32) {
      const uint4 u = eo[0], w = eo[EW - 1];
      v[0] += __uint_as_float(u.x); v[1] += __uint_as_float(u.y); v[2] += __uint_as_float(u.z); v[3] += __uint_as_float(u.w);
      v[4] += __uint_as_float(w.x); v[5] += __uint_as_float(w.y); v[6] += __uint_as_float(w.z); v[7] += __uint_as_float(w.w);
    }
    const size_t ci = grow * p.ldc + col;
    if (p.C2) {
      float gv[8], gd[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { float cdf, e; gelu_terms(v[q], cdf, e); gv[q] = v[q] * cdf; gd[q] = cdf + v[q] * 0.3989422804014327f * e; }
      st8(p.C, p.c_dt, ci, gv);
      if (p.C2 != p.C) st8(p.C2, p.c_dt, ci, gd);
    } else {
      st8(p.C, p.c_dt, ci, v);
    }
  }
}

// ---- launch policy -------------------------------------------------------------------------------------------------------------
// mode: 0 = never, 1 = policy (default), 2 = every eligible call with `variant` (tests, tools/bench_deep_gemm.py)
#define WIDE_V_8W4S 0      /* 8 waves, 4 stages, epilogue operands ahead of the K loop: one workgroup per CU */
#define WIDE_V_8W2S 1      /* 8 waves, 2 stages: two workgroups per CU */
#define WIDE_V_4W2S 2      /* 4 waves, 2 stages: two workgroups per CU */
static int g_wide_mode = 1, g_wide_variant = WIDE_V_8W4S;
extern "C" void scot_gemm_wide_config(int mode, int variant) { g_wide_mode = mode; g_wide_variant = variant; }
int scot_gemm_wide_mode(int* variant) { if (variant) *variant = g_wide_variant; return g_wide_mode; }      // (wgrad_wide.hip's policy in gemm_fast.hip)

// Which instantiation, or -1 = leave the call to gemm_fast.  From profiles/round5/gemm_wide_sweep_v3_{B,L,B256}.txt (hipGraph replays of 20
// dependent launches per configuration, binary16 operands; us per launch, 64 x 64 tiles -> chosen instantiation):
//   > 256 tiles: two workgroups per CU — four waves for the plain epilogues (Poseidon-L fc1 93.8 -> 53.6, qkv 70.5 -> 43.6; stage 1 of L, 3072
//     tiles: 106.6 -> 79.1), eight where the epilogue reads a second operand (gelu'-scaled data gradient 92.7 -> 47.5 = 815 TF/s;
//     fp32-accumulating data gradients 67.3 -> 49.4);
//   128-256 tiles (one workgroup per CU, four stages): contractions of >= 48 K-tiles (L fc2 70.4 -> 58.7, fc1 data gradient 71.8 -> 56.3)
//     and the gelu'-scaled data gradient (Poseidon-B stage 3: 14.4 -> 11.6); shorter contractions are even or lose (19.3 -> 20.8);
//   < 128 tiles: the 64 x 64 tiles' three workgroups per CU win everywhere (Poseidon-B stage 3, N = 768: 18.4 vs 30.7).
static int wide_variant(int M, int N, int K, int epi) {
  if (g_wide_mode == 0) return -1;
  if (M % 128 || N % 128 || K % 64 || M < 128 || N < 128 || K < 64) return -1;
  if (g_wide_mode == 2) return g_wide_variant & 15;      // (bits 4..: K slices of the grouped weight gradients, gemm_fast.hip)
  const long tiles = (long)(M / 128) * (N / 128);
  const int nkt = K / 64;
  if (nkt < 6) return -1;
  if (tiles > 256) return epi == WIDE_EPI_NONE ? WIDE_V_4W2S : WIDE_V_8W2S;
  if (tiles >= 128) return (nkt >= 48 || epi == WIDE_EPI_AUX16) ? WIDE_V_8W4S : -1;
  return -1;
}

template <int WM, int WN, int STAGES, bool PRE>
static void wide_launch(int epi, unsigned grid, const WideArgs& a, hipStream_t stream) {
  const dim3 g(grid), b(WM * WN * 64);
  if (epi == WIDE_EPI_AUX16) hipLaunchKernelGGL((gemm_wide_kernel<128, 128, WM, WN, STAGES, WIDE_EPI_AUX16, PRE>), g, b, 0, stream, a);
  else if (epi == WIDE_EPI_RES32) hipLaunchKernelGGL((gemm_wide_kernel<128, 128, WM, WN, STAGES, WIDE_EPI_RES32, PRE>), g, b, 0, stream, a);
  else hipLaunchKernelGGL((gemm_wide_kernel<128, 128, WM, WN, STAGES, WIDE_EPI_NONE, PRE>), g, b, 0, stream, a);
}

// Returns SCOT_ERR_UNSUPPORTED when the call does not qualify (scot_gemm then asks gemm_fast).
int scot_gemm_wide(int layout, int compute, int M, int N, int K, const void* A, int a_dt, int lda, int a_gelu,
                   const void* B, int b_dt, int ldb, int b_gelu, void* C, int c_dt, int ldc, const float* bias,
                   const float* colscale, const void* aux, int aux_dt, int ldaux, const void* resid, int res_dt, int ldres,
                   int accumulate, float* colsum_out, int aux_mul, void* C2, hipStream_t stream) {
  if (layout != LAYOUT_NT || compute != SCOT_BF16 || a_dt != SCOT_BF16 || b_dt != SCOT_BF16) return SCOT_ERR_UNSUPPORTED;
  if (a_gelu || b_gelu || colscale || colsum_out || (aux && aux_dt != SCOT_BF16)) return SCOT_ERR_UNSUPPORTED;
  if (aux && (resid || accumulate)) return SCOT_ERR_UNSUPPORTED;               // (one epilogue operand per instantiation: what the engine's calls use)
  if ((resid && res_dt != SCOT_F32) || (accumulate && c_dt != SCOT_F32)) return SCOT_ERR_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)C2 | (uintptr_t)aux | (uintptr_t)resid) & 15) != 0) return SCOT_ERR_UNSUPPORTED;
  if (lda % 8 || ldb % 8 || ldc % 8 || (aux && ldaux % 8) || (resid && ldres % 8)) return SCOT_ERR_UNSUPPORTED;
  if (accumulate && resid != nullptr) return SCOT_ERR_UNSUPPORTED;
  const int epi = aux ? WIDE_EPI_AUX16 : ((resid || accumulate) ? WIDE_EPI_RES32 : WIDE_EPI_NONE);
  const int variant = wide_variant(M, N, K, epi);
  if (variant < 0) return SCOT_ERR_UNSUPPORTED;
  WideArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.C2 = C2; a.bias = bias; a.aux = aux; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux; a.ldres = ldres;
  a.c_dt = c_dt; a.aux_mul = aux_mul;
  if (accumulate) { a.resid = C; a.ldres = ldc; }
  const int tm = M / 128, tn = N / 128;
  a.tiles = tm * tn; a.tiles_n = tn;
  a.tpx = (a.tiles + 7) / 8;
  // chunk of tpx consecutive tiles per XCD: along N it touches ~ceil(tpx / tn) A row blocks and min(tpx, tn) B row blocks, along M the mirror
  const long rowmaj = (a.tpx + tn - 1) / tn + (a.tpx < tn ? a.tpx : tn), colmaj = (a.tpx + tm - 1) / tm + (a.tpx < tm ? a.tpx : tm);
  a.colmajor = colmaj < rowmaj ? 1 : 0;
  const unsigned grid = 8u * (unsigned)a.tpx;
  switch (variant) {
    case WIDE_V_8W2S: wide_launch<2, 4, 2, false>(epi, grid, a, stream); break;
    case WIDE_V_4W2S: wide_launch<2, 2, 2, false>(epi, grid, a, stream); break;
    default: wide_launch<2, 4, 4, true>(epi, grid, a, stream); break;
  }
  return scot_check_launch();
}
