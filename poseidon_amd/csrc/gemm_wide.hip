// gemm_wide — the NT products of the deep stages (C = 384 / 768: 1024-4096 token rows against 384-3072 weight rows) on 128 x 128
// output tiles, split along K INSIDE the launch when the output has too few tiles to fill the chip.
//
// Why (round-4 measurements, profiles/round4/gemm_load_probe_r4.txt): gemm_fast's 64 x 64-tile loop pushes ~13x the operand bytes
// through L2 -> LDS (147 MB for the 11 MB of [1024, 768] x 3072) and a workgroup's ingest is bytes in flight / ~0.55 us: 32 KB in its
// three-stage ring = 58 GB/s with one workgroup per CU, against the ~100 GB/s a CU can take.  A 128 x 128 tile does four times the
// flops per K-tile for twice the bytes (64 flop/B instead of 32), and its three 32 KB stages keep 64 KB in flight.  What made 128-wide
// tiles lose in rounds 1-3 was the grid: 48 tiles for [1024, 3072] -> 768.  Here the K range of such a product is cut into S slices
// (grid = tiles x S); every slice leaves its fp32 accumulators as a slab in the workspace, takes a ticket from the tile's arrival
// counter (agent-scope release before, cdna_hip_programming.md §5 "in-launch split-K reduction"), and the workgroup that draws the last
// ticket adds the slabs IN SLICE ORDER (its own from registers: the sum does not depend on who arrived last) and runs the epilogue —
// bias, gelu / gelu' pair, gelu'(aux) factor, residual / accumulate — once.  No reduce launch, no second kernel boundary.
//
// Operands: 16-bit, K-contiguous (A [M, K], B [N, K]), M % 128 == 0, N % 128 == 0, K % 64 == 0.  Everything else stays with gemm_fast.
// Reference shapes: the Linear layers of HF modeling_swinv2.py:396-410, 496-506, 536-561 at the widths of scOT/model.py:403-404.
#include "common.h"
#include <stdlib.h>

#define LAYOUT_NT 0

// Hand-off of a slab to the tile's last arriver (cdna_hip_programming.md §6 Guideline 16, recipe R1): the slab is stored WRITE-THROUGH
// (16-byte buffer stores with the sc1 bit: the bytes leave this XCD's L2 for memory, no release fence — a `buffer_wbl2` per workgroup
// wrote back the whole L2 each time and doubled the launch, profiles/round5/gemm_wide_sweep_r5.txt), every storing wave drains its
// stores, ONE lane takes the ticket (relaxed agent-scope atomic); the last arriver's ONE acquire drops its CU's L1, then plain loads.
#ifdef SCOT_HIPEMU
#define SCOT_WAIT_VM0() ((void)0)
#define SCOT_ACQUIRE_AGENT() ((void)0)
struct wide_rsrc_t { float* base; };
__device__ __forceinline__ wide_rsrc_t wide_make_rsrc(float* base, int) { return wide_rsrc_t{base}; }
__device__ __forceinline__ void wide_store_wt(f32x4_t v, wide_rsrc_t r, int byte_off) { *(f32x4_t*)((char*)r.base + byte_off) = v; }
#else
#define SCOT_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SCOT_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
typedef __amdgpu_buffer_rsrc_t wide_rsrc_t;
__device__ __forceinline__ wide_rsrc_t wide_make_rsrc(float* base, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000); }
typedef unsigned wide_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wide_store_wt(f32x4_t v, wide_rsrc_t r, int byte_off) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wide_u32x4_t, v), r, byte_off, 0, /*aux: sc1*/ 16);
}
#endif

struct WideArgs {
  const bf16_t* A; const bf16_t* B; void* C; void* C2;
  const float* bias; const void* aux; const void* resid;
  int M, N, K, lda, ldb, ldc, ldaux, ldres;
  int c_dt, aux_mul;
  int nsplit, kt_slice;      // K slices per tile; K-tiles (of 64) per slice
  int tiles, tiles_n, tpx;   // output tiles, tiles along N, tiles per XCD chunk
  int colmajor;              // tile order inside an XCD's chunk: 0 = along N (shares the A row block), 1 = along M (shares the B rows)
  float* slabs; int* counters;
};

template <int BM, int BN, int STAGES> struct WideLds {
  static constexpr int STAGE = (BM + BN) * 64;                      // 16-bit elements per stage
  static constexpr size_t AB = (size_t)STAGES * STAGE * 2, C = (size_t)BM * (BN + 4) * 4;
  static constexpr size_t flag_off = AB > C ? AB : C;               // the last-arriver ticket lives behind both uses (ONE __shared__ object)
  static constexpr size_t bytes = flag_off + 64;
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
// unrolled loop made hipcc branch around every load and wait for each).
#define WIDE_EPI_NONE 0
#define WIDE_EPI_AUX16 1
#define WIDE_EPI_RES32 2
// WM x WN: arrangement of the NW = WM·WN waves over the tile; a wave owns (BM / WM) x (BN / WN).  More waves, not more bytes in flight, is
// what raises a workgroup's direct-to-LDS ingest: a wave's global_load_lds instructions complete one after the other (~1 KB per 300 clk).
template <int BM, int BN, int WM, int WN, int STAGES, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void gemm_wide_kernel(WideArgs p) {
  constexpr int NW = WM * WN, NT = NW * 64, MI = BM / WM / 16, NI = BN / WN / 16, WROWS = BM / WM, WCOLS = BN / WN;
  constexpr int STAGE = WideLds<BM, BN, STAGES>::STAGE, BOFF = BM * 64;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW, LPW = PA + PB;      // 1 KB pieces (8 tile rows) per wave and K-tile
  constexpr int CP = BN + 4;
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "pieces per wave");
  __shared__ __attribute__((aligned(1024))) char smem[WideLds<BM, BN, STAGES>::bytes];
  bf16_t* lds = (bf16_t*)smem;

  // workgroup b runs on XCD b % 8 (dispatch order: speed only).  An XCD owns a contiguous chunk of `tpx` tiles — they share operand rows
  // in its L2 — and all slices of a tile, back to back, so the last arriver reads slabs its own L2 may still hold.
  const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
  const int tl = j / p.nsplit, slice = j % p.nsplit;
  const int tile = xcd * p.tpx + tl;
  if (tl >= p.tpx || tile >= p.tiles) return;
  int by, bx;
  if (p.colmajor) { const int tm = p.tiles / p.tiles_n; bx = tile / tm; by = tile % tm; }
  else { by = tile / p.tiles_n; bx = tile % p.tiles_n; }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN, g = lane >> 4;
  const int m0 = by * BM, n0 = bx * BN;
  const int nkt = p.K >> 6;
  const int kt0 = slice * p.kt_slice;
  const int nk = min(p.kt_slice, nkt - kt0);

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
  uint4 eop[EPI == WIDE_EPI_NONE ? 1 : E_IT][EPI == WIDE_EPI_RES32 ? 2 : 1];
  auto load_epilogue_operands = [&]() {
#pragma unroll
    for (int it = 0; it < E_IT; ++it) {
      const size_t grow = (size_t)(m0 + erow + it * RPP);
      if constexpr (EPI == WIDE_EPI_AUX16) eop[it][0] = *(const uint4*)((const bf16_t*)p.aux + grow * p.ldaux + col);
      if constexpr (EPI == WIDE_EPI_RES32) {
        eop[it][0] = *(const uint4*)((const float*)p.resid + grow * p.ldres + col);
        eop[it][1] = *(const uint4*)((const float*)p.resid + grow * p.ldres + col + 4);
      }
    }
  };
  // unsplit: requested here, in front of the K loop (older than every direct-to-LDS load, so the loop's counted waits retire them first);
  // split: only the last arriver needs them (below)
  if constexpr (EPI != WIDE_EPI_NONE) {
    if (p.nsplit == 1) load_epilogue_operands();
  }

  typedef __attribute__((address_space(3))) void* lds_p;
  typedef __attribute__((address_space(1))) const void* gbl_p;
  auto issue = [&](bf16_t* st, int kt) {
    const int k0 = (kt0 + kt) << 6;
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

  if (p.nsplit > 1) {
    // slab of (tile, slice): the accumulators in fragment order — one 16-byte store per lane and fragment, 1 KB contiguous per wave instruction
    float* slab0 = p.slabs + (size_t)tile * p.nsplit * (BM * BN);
    {
      const wide_rsrc_t mine = wide_make_rsrc(slab0 + (size_t)slice * (BM * BN), BM * BN * 4);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) wide_store_wt(acc[i][jj], mine, ((i * NI + jj) * NT + tid) * 16);
    }
    int* flag = (int*)(smem + WideLds<BM, BN, STAGES>::flag_off);
    SCOT_WAIT_VM0();                 // every storing wave: its write-through stores have reached memory
    __syncthreads();
    if (tid == 0) *flag = atomicAdd(p.counters + tile, 1);
    __syncthreads();
    if (*flag != p.nsplit - 1) return;
    if (tid == 0) {
      p.counters[tile] = 0;          // ready for the next launch that uses this tile index (same stream: ordered by the kernel boundary)
      SCOT_ACQUIRE_AGENT();
    }
    __syncthreads();
    if constexpr (EPI != WIDE_EPI_NONE) load_epilogue_operands();
    // Σ over the slices in slice order, this workgroup's own term from its registers: bit-identical whoever arrives last
    f32x4_t tot[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) tot[i][jj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.nsplit; ++s) {
      if (s == slice) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int jj = 0; jj < NI; ++jj) tot[i][jj] += acc[i][jj];
      } else {
        const float* other = slab0 + (size_t)s * (BM * BN);
        f32x4_t v[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int jj = 0; jj < NI; ++jj) v[i][jj] = *(const f32x4_t*)(other + ((i * NI + jj) * NT + tid) * 4);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int jj = 0; jj < NI; ++jj) tot[i][jj] += v[i][jj];
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int jj = 0; jj < NI; ++jj) acc[i][jj] = tot[i][jj];
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
    float v[8];
    const float4 a = *(const float4*)(Cs + row * CP + cc * 8), b = *(const float4*)(Cs + row * CP + cc * 8 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] += bv[q];
    if constexpr (EPI == WIDE_EPI_AUX16) {
      float x[8];
      const uint4 u = eop[it][0];
      unpack_bf16x2(u.x, x[0], x[1]); unpack_bf16x2(u.y, x[2], x[3]); unpack_bf16x2(u.z, x[4], x[5]); unpack_bf16x2(u.w, x[6], x[7]);
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] *= p.aux_mul ? x[q] : gelu_grad_f(x[q]);
    }
    if constexpr (EPI == WIDE_EPI_RES32) {
      const uint4 u = eop[it][0], w = eop[it][1];
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
// mode: 0 = never, 1 = policy (default), 2 = every eligible call (tests / tools); force_split > 0 overrides the slice count
static int g_wide_mode = 1, g_wide_force_split = 0, g_wide_variant = 1;
extern "C" void scot_gemm_wide_config(int mode, int force_split) {
  g_wide_mode = mode & 0xff; g_wide_force_split = force_split;
  if (mode >> 8) g_wide_variant = (mode >> 8) - 1;       // (tools/bench_deep_gemm.py: bits 8.. = 1 + kernel variant, see wide_launch)
}

#define SCOT_WIDE_COUNTER_BYTES SCOT_WS_RESERVED     /* the LAST bytes of the caller's workspace: tile arrival counters, zero between launches */

struct WidePlan { int S, kt_slice, tiles, tiles_n, tpx, colmajor; size_t slab_bytes; };

// Decided from (M, N, K) alone (scot_gemm_workspace_bytes must give the same answer as the launch).  false = leave the call to gemm_fast.
static bool wide_plan(int M, int N, int K, WidePlan& pl) {
  if (g_wide_mode == 0) return false;
  if (M % 128 || N % 128 || K % 64 || M < 128 || N < 128) return false;
  const int tm = M / 128, tn = N / 128, tiles = tm * tn, nkt = K / 64;
  int S = 1;
  if (g_wide_mode == 1) {
    // The tile pays where a 64 x 64 grid is either small and long (stage 3: 192 workgroups x 36-48 K-tiles) or large and short; it needs
    // >= 6 K-tiles per workgroup to amortise its 0.6 us first-load latency and ~128+ workgroups to beat the narrow tiles' occupancy
    if (M > 8192 || nkt < 6) return false;
    if (tiles < 128) {
      S = (160 + tiles - 1) / tiles;
      while (S > 1 && nkt / S < 6) --S;
      if (S > 8) S = 8;
      if (tiles * S < 96) return false;
    } else if (tiles > 512) {
      return false;
    }
  }
  if (g_wide_force_split > 0) S = g_wide_force_split;
  if (S > nkt) S = nkt;
  if (S > 1 && tiles > (int)(SCOT_WS_RESERVED / sizeof(int))) return false;      // one arrival counter per tile
  pl.kt_slice = (nkt + S - 1) / S;
  pl.S = (nkt + pl.kt_slice - 1) / pl.kt_slice;
  pl.tiles = tiles; pl.tiles_n = tn;
  pl.tpx = (tiles + 7) / 8;
  // chunk of tpx consecutive tiles per XCD: along N it touches ~ceil(tpx / tn) A row blocks and min(tpx, tn) B row blocks, along M the mirror
  const double rowmaj = (double)((pl.tpx + tn - 1) / tn) * 128.0 + (double)(pl.tpx < tn ? pl.tpx : tn) * 128.0;
  const double colmaj = (double)((pl.tpx + tm - 1) / tm) * 128.0 + (double)(pl.tpx < tm ? pl.tpx : tm) * 128.0;
  pl.colmajor = colmaj < rowmaj ? 1 : 0;
  pl.slab_bytes = pl.S > 1 ? (size_t)tiles * pl.S * 128 * 128 * sizeof(float) : 0;
  return true;
}

// bytes of workspace a wide launch of this shape would use (0: not taken / unsplit); scot_gemm_workspace_bytes adds it to its answer
size_t scot_gemm_wide_workspace_bytes(int layout, int compute, int M, int N, int K) {
  WidePlan pl;
  if (layout != LAYOUT_NT || compute != SCOT_BF16 || !wide_plan(M, N, K, pl)) return 0;
  return pl.slab_bytes ? pl.slab_bytes + SCOT_WIDE_COUNTER_BYTES : 0;
}

template <int WM, int WN, int STAGES>
static void wide_launch(int epi, unsigned grid, const WideArgs& a, hipStream_t stream) {
  const dim3 g(grid), b(WM * WN * 64);
  if (epi == WIDE_EPI_AUX16) hipLaunchKernelGGL((gemm_wide_kernel<128, 128, WM, WN, STAGES, WIDE_EPI_AUX16>), g, b, 0, stream, a);
  else if (epi == WIDE_EPI_RES32) hipLaunchKernelGGL((gemm_wide_kernel<128, 128, WM, WN, STAGES, WIDE_EPI_RES32>), g, b, 0, stream, a);
  else hipLaunchKernelGGL((gemm_wide_kernel<128, 128, WM, WN, STAGES, WIDE_EPI_NONE>), g, b, 0, stream, a);
}

// Returns SCOT_ERR_UNSUPPORTED when the call does not qualify (scot_gemm then asks gemm_fast).
int scot_gemm_wide(int layout, int compute, int M, int N, int K, const void* A, int a_dt, int lda, int a_gelu,
                   const void* B, int b_dt, int ldb, int b_gelu, void* C, int c_dt, int ldc, const float* bias,
                   const float* colscale, const void* aux, int aux_dt, int ldaux, const void* resid, int res_dt, int ldres,
                   int accumulate, float* colsum_out, void* workspace, size_t ws_bytes, int aux_mul, void* C2, hipStream_t stream) {
  if (layout != LAYOUT_NT || compute != SCOT_BF16 || a_dt != SCOT_BF16 || b_dt != SCOT_BF16) return SCOT_ERR_UNSUPPORTED;
  if (a_gelu || b_gelu || colscale || colsum_out || (aux && aux_dt != SCOT_BF16)) return SCOT_ERR_UNSUPPORTED;
  if (aux && (resid || accumulate)) return SCOT_ERR_UNSUPPORTED;               // (one epilogue operand per instantiation: what the engine's calls use)
  if ((resid && res_dt != SCOT_F32) || (accumulate && c_dt != SCOT_F32)) return SCOT_ERR_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)C2 | (uintptr_t)aux | (uintptr_t)resid) & 15) != 0) return SCOT_ERR_UNSUPPORTED;
  if (lda % 8 || ldb % 8 || ldc % 8 || (aux && ldaux % 8) || (resid && ldres % 8)) return SCOT_ERR_UNSUPPORTED;
  WidePlan pl;
  if (!wide_plan(M, N, K, pl)) return SCOT_ERR_UNSUPPORTED;
  WideArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.C2 = C2; a.bias = bias; a.aux = aux; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux; a.ldres = ldres;
  a.c_dt = c_dt; a.aux_mul = aux_mul;
  if (accumulate) {
    if (resid != nullptr) return SCOT_ERR_UNSUPPORTED;
    a.resid = C; a.ldres = ldc;
  }
  a.nsplit = pl.S; a.kt_slice = pl.kt_slice; a.tiles = pl.tiles; a.tiles_n = pl.tiles_n; a.tpx = pl.tpx; a.colmajor = pl.colmajor;
  a.slabs = nullptr; a.counters = nullptr;
  if (pl.S > 1) {
    if (!workspace || (((uintptr_t)workspace) & 15) || ws_bytes < pl.slab_bytes + SCOT_WIDE_COUNTER_BYTES || (ws_bytes & 15)) return SCOT_ERR_UNSUPPORTED;
    a.slabs = (float*)workspace;
    a.counters = (int*)((char*)workspace + ws_bytes - SCOT_WIDE_COUNTER_BYTES);
  }
  const unsigned grid = 8u * (unsigned)pl.tpx * (unsigned)pl.S;
  const int epi = a.aux ? WIDE_EPI_AUX16 : (a.resid ? WIDE_EPI_RES32 : WIDE_EPI_NONE);
  switch (g_wide_variant) {
    case 0: wide_launch<2, 2, 3>(epi, grid, a, stream); break;      // 4 waves of 64 x 64, three 32 KB stages
    case 2: wide_launch<2, 4, 4>(epi, grid, a, stream); break;      // 8 waves of 64 x 32, four stages
    case 3: wide_launch<4, 4, 3>(epi, grid, a, stream); break;      // 16 waves of 32 x 32, three stages
    case 4: wide_launch<2, 4, 2>(epi, grid, a, stream); break;      // 8 waves, two stages: 68 KB of LDS = two workgroups per CU (large grids)
    case 5: wide_launch<2, 2, 2>(epi, grid, a, stream); break;      // 4 waves, two stages
    default: wide_launch<2, 4, 3>(epi, grid, a, stream); break;     // 8 waves of 64 x 32, three stages
  }
  return scot_check_launch();
}
