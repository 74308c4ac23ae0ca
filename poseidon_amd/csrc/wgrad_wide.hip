// wgrad_wide — the grouped weight gradients of a ScOTLayer (dW_i[M_i, N_i] += dY_i[K, M_i]^T · X_i[K, N_i], i < 8, the same K tokens:
// autograd of HF modeling_swinv2.py:396-410, 502-506, 545-561) on 128 x 128 output tiles, for groups whose M_i, N_i are multiples of 128
// (C = 384 / 768 / 1536: the deep stages of Poseidon-B and every GEMM-shaped stage of Poseidon-L).
//
// Why: round 5's what-if (profiles/round5/whatif_side_stream_r5.txt) — the step is 1.05 ms shorter without the deep stages' grouped
// gradients (2.3 ms of kernels on the weight-gradient stream at ~200 TF/s on 64 x 64 tiles), and gemm_wide.hip's sweeps say what a
// 128 x 128 tile does once its grid gives every CU two workgroups (600-800 TF/s on the NT products).
// Both operands are K-STRIDED here (a token per row): the LDS tiles stay in source orientation [64 tokens][128] (256-byte rows, filled
// by global_load_lds_dwordx4: 1 KB = 4 token rows per wave instruction) and the MFMA fragments — 8 consecutive tokens of one output
// row / column per lane — are read with ds_read_b64_tr_b16 (common.h lds_frag_ks).  Without padding the four token rows a 16-lane group
// reads sit 256 bytes apart = on the same banks: the 32-byte block b of token row k is stored at block b ^ f(k), f(k) = (k & 3) |
// ((k >> 1) & 4), which gives the eight rows one half-wave touches eight different 8-bank groups; applied on the SOURCE side of the
// direct-to-LDS load (the destination is lane-linear) and in the fragment address.
// Bias gradients (column sums of dY over the tokens) come from the dY tile in LDS, as in gemm_fast's TN kernel.
// Groups of 64-255 tiles over >= 8192 tokens (Poseidon-L's and B@256²'s middle stages) cut K into slices: partial tiles go to the workspace in
// the layout of wgrad_group_reduce_kernel (gemm_fast.hip), which commits them to the gradients.  Shorter contractions stay on the unsplit
// 64 x 64 grouped kernel — Poseidon-B's stage 2 (108 tiles, 4096 tokens): 58.6 -> 54.4 us alone for 113 MB of partials, nothing in the step
// (the policy is wgrad_group_impl's, gemm_fast.hip).  How a result meets its gradient tensor (add / store scaled / add scaled: the lazy
// zero-grad of round 6) is WgradProblem::mode, applied by grad_commit8 here and in the reduce.
#include "wgrad_group.h"
#include <stdlib.h>

template <int STAGES> struct WgWideLds {
  static constexpr int STAGE = 2 * 64 * 128;                           // 16-bit elements per stage: dY tile + X tile
  static constexpr size_t AB = (size_t)STAGES * STAGE * 2, C = (size_t)128 * 132 * 4;
  static constexpr size_t bytes = AB > C ? AB : C;
};

__device__ __forceinline__ int wg_swz(int k) { return (k & 3) | ((k >> 1) & 4); }

// 8 consecutive tokens kk + 8 g .. + 7 of output row / column c0 + (lane & 15), from a [64][128] tile (see the header): two transposing reads
__device__ __forceinline__ Frag<bf16_t> wg_frag(const bf16_t* t, int blk, int kk, int lane, int use_tr) {
  Frag<bf16_t> f;
  const int i = lane & 15, g = lane >> 4;
  if (use_tr) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const int k0 = kk + 8 * g + (i >> 2), k1 = k0 + 4;
    const bf16_t* p0 = t + k0 * 128 + ((blk ^ wg_swz(k0)) << 4) + (i & 3) * 4;
    const bf16_t* p1 = t + k1 * 128 + ((blk ^ wg_swz(k1)) << 4) + (i & 3) * 4;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p1);
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kk + 8 * g + j;
      f.v[j] = (short)t[k * 128 + ((blk ^ wg_swz(k)) << 4) + i];
    }
  }
  return f;
}

template <int WM, int WN, int STAGES>
__global__ __launch_bounds__(WM * WN * 64) void wgrad_wide_kernel(WgradGroupArgs g) {
  constexpr int NW = WM * WN, NT = NW * 64, MI = 128 / WM / 16, NI = 128 / WN / 16, WROWS = 128 / WM, WCOLS = 128 / WN;
  constexpr int STAGE = WgWideLds<STAGES>::STAGE, BOFF = 64 * 128;
  constexpr int PP = 16 / NW, LPW = 2 * PP;            // 1 KB pieces (4 token rows of 256 bytes) per wave, operand and K-tile
  constexpr int CP = 132;
  static_assert(16 % NW == 0 && STAGES >= 2 && STAGES <= 4, "waves / ring depth");
  __shared__ __attribute__((aligned(1024))) char smem[WgWideLds<STAGES>::bytes];
  bf16_t* lds = (bf16_t*)smem;

  // unsplit: an XCD (workgroup b runs on XCD b % 8: speed only) owns a contiguous chunk of tiles — neighbours share operand panels in its L2;
  // split: all tiles of one K slice on one XCD when the slice count allows, as wgrad_group_kernel
  int L = blockIdx.x, slice, tile;
  if (g.nsplit == 1) {
    const int tpx = (g.tiles + 7) >> 3;
    slice = 0;
    tile = (L & 7) * tpx + (L >> 3);
    if (tile >= g.tiles) return;
  } else if (g.nsplit % 8 == 0) {
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
  const int by = local / pr.tiles_n, bx = local % pr.tiles_n;
  const int m0 = by * 128, n0 = bx * 128;
  const bf16_t* A = (const bf16_t*)pr.A;
  const bf16_t* B = (const bf16_t*)pr.B;
  const int lda = pr.lda, ldb = pr.ldb;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN, gq = lane >> 4;
  const int kbeg = slice * g.ksplit;                      // tokens; ksplit is a multiple of 64
  const int nk = (min(g.K, kbeg + g.ksplit) - kbeg) >> 6;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int jj = 0; jj < NI; ++jj) acc[i][jj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;                                       // thread tid < 128 of a tile in the first column: Σ_tokens dY[., m0 + tid]
  const bool want_bias = pr.colsum != nullptr && bx == 0 && tid < 128;

  typedef __attribute__((address_space(3))) void* lds_p;
  typedef __attribute__((address_space(1))) const void* gbl_p;
  auto issue = [&](bf16_t* st, int kt) {
    const size_t k0 = (size_t)kbeg + ((size_t)kt << 6);
#pragma unroll
    for (int u = 0; u < PP; ++u) {
      const int pc = wave + NW * u, k = 4 * pc + (lane >> 4), c = (lane & 15) ^ (wg_swz(k) << 1);      // 16-byte chunk c of token row k
      __builtin_amdgcn_global_load_lds((gbl_p)(A + (k0 + k) * lda + m0 + c * 8), (lds_p)(st + pc * 512), 16, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < PP; ++u) {
      const int pc = wave + NW * u, k = 4 * pc + (lane >> 4), c = (lane & 15) ^ (wg_swz(k) << 1);
      __builtin_amdgcn_global_load_lds((gbl_p)(B + (k0 + k) * ldb + n0 + c * 8), (lds_p)(st + BOFF + pc * 512), 16, 0, 0);
    }
  };
  auto compute = [&](const bf16_t* st) {
    const bf16_t* As = st;
    const bf16_t* Bs = st + BOFF;
#pragma unroll
    for (int kk = 0; kk < 64; kk += 32) {
      Frag<bf16_t> fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = wg_frag(As, wr * (WROWS / 16) + i, kk, lane, g.use_tr);
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) fb[jj] = wg_frag(Bs, wc * (WCOLS / 16) + jj, kk, lane, g.use_tr);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) mma16(acc[i][jj], fa[i], fb[jj]);
    }
    if (want_bias) {
#pragma unroll 8
      for (int k = 0; k < 64; ++k) bsum += bf2f(As[k * 128 + (tid ^ (wg_swz(k) << 4))]);
    }
  };

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
  if (want_bias) atomicAdd(&pr.colsum[m0 + tid], bsum);

  // ---- epilogue through LDS: fp32 row segments; unsplit: this workgroup is the tile's only writer (read-modify-write, no atomics)
  __syncthreads();
  float* Cs = (float*)smem;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int jj = 0; jj < NI; ++jj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cs[(wr * WROWS + i * 16 + gq * 4 + r) * CP + wc * WCOLS + jj * 16 + (lane & 15)] = acc[i][jj][r];
  __syncthreads();
  constexpr int RPP = NT / 16, E_IT = 128 / RPP;
  const int cc = tid % 16, erow = tid / 16;
  float* part = g.ws ? g.ws + (size_t)slice * g.plane + pr.ws_off : nullptr;
#pragma unroll
  for (int it = 0; it < E_IT; ++it) {
    const int row = erow + it * RPP;
    float v[8];
    const float4 a = *(const float4*)(Cs + row * CP + cc * 8), b = *(const float4*)(Cs + row * CP + cc * 8 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    if (part) {
      st8(part, SCOT_F32, (size_t)(m0 + row) * pr.N + n0 + cc * 8, v);
    } else {
      grad_commit8(pr.C, (size_t)(m0 + row) * pr.ldc + n0 + cc * 8, v, pr.mode, g.scale);
    }
  }
}

// variant: 0 = 8 waves x 4 stages (one workgroup per CU), 1 = 8 waves x 2 stages, 2 = 4 waves x 2 stages (two per CU) — gemm_wide.hip's three
int scot_wgrad_group_wide_launch(const WgradGroupArgs& g, int variant, hipStream_t s) {
  const unsigned grid = g.nsplit == 1 ? 8u * (unsigned)((g.tiles + 7) / 8) : (unsigned)(g.tiles * g.nsplit);
  switch (variant) {
    case 1: hipLaunchKernelGGL((wgrad_wide_kernel<2, 4, 2>), dim3(grid), dim3(512), 0, s, g); break;
    case 2: hipLaunchKernelGGL((wgrad_wide_kernel<2, 2, 2>), dim3(grid), dim3(256), 0, s, g); break;
    default: hipLaunchKernelGGL((wgrad_wide_kernel<2, 4, 4>), dim3(grid), dim3(512), 0, s, g); break;
  }
  return scot_check_launch();
}
