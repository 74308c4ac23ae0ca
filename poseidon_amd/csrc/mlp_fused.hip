// mlp_fused — the MLP half of a ScOTLayer in ONE kernel, for the token-heavy stages (C = 96 / 192), bf16 operands:
//
//     z   = gelu(h16 · W1^T + b1) · W2^T + b2                  (Swinv2Intermediate + Swinv2Output, HF:533-561)
//     out = h + s_b · (gamma_b ⊙ LN(z) + beta_b),  out16 = bf16(out)     (res-post-norm + DropPath, reference model.py:566-579)
//
// The layer-by-layer path runs three kernels here (fc1 with the GELU epilogue, fc2, cond-LN) and moves the [M, 4C] activation
// through HBM twice more than needed (PMC round 1: the 4C tensors are 45 % of the forward bytes).  Here a workgroup owns
// 64·TT rows from the operand load to the normalised output:
//   * GEMM 1 is computed TRANSPOSED (U^T = W1 · h^T: the W1 rows are the MFMA's M dimension, the tokens its N dimension), so an
//     accumulator lane (g, c) ends up holding token c and 8 consecutive hidden units of it — after bias + GELU (fp32) and
//     rounding to bf16 that IS the A operand of GEMM 2 (rows = tokens, K = hidden) in the library's one fragment convention:
//     no LDS round trip, no shuffles between the two GEMMs.  The "8 consecutive" come from storing the W1 rows of every
//     32-hidden block into LDS in the order [t][a][b] of hidden = 8a + 4t + b (two 16-row MFMA tiles t = 0, 1).
//   * W1 / W2 stream through LDS in chunks of HC hidden units (all workgroups read the same weights: L2 traffic); the next
//     chunk's 16-byte loads are in flight in registers while the current one is multiplied.
//   * the [rows, C] result never leaves the CU before the layer norm: accumulators -> per-wave LDS patch (aliasing the dead
//     weight chunk) -> row-contiguous registers -> statistics over the 4 lanes of a row -> 16/32-byte stores.
// Training additionally stores what the backward consumes (gelu(u), gelu'(u), z, mean, rstd) — same tensors, same dtypes and
// the same rounding points as the three-kernel path, so the two paths agree to accumulation order.
//
// STATUS: default path of the 16-bit modes since round 2 (60 GPU parity tests against the launches it replaces, whole-model
// fixtures with it on; stage 0: 58 us forward / 60 us backward chain per block, ~3.4 TB/s of HBM traffic — the tensors saved for
// the backward are what bounds it; the inference variant runs 41 us).  Also runs on the CPU through tests/hipemu.
#include "common.h"
#include <stdlib.h>

#include "block_fused.h"

// prefetch registers: a NATIVE vector type — arrays of HIP's uint4 struct were left in scratch memory by the compiler here
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct MlpArgs {
  const bf16_t* h16; const float* h;
  const bf16_t* W1; const float* b1;
  const bf16_t* W2; const float* b2;
  float* out; bf16_t* out16;
  bf16_t* act; bf16_t* dact;
  void* z; int z_dt; float* mean; float* rstd;
  const float* time; const float* gw_w; const float* gw_b; const float* bw_w; const float* bw_b; const float* sscale;
  int M, rows_per_sample, hid;
  float eps;
};

template <int C, int HC> struct MlpFwdLds {
  static constexpr size_t WBYTES = (size_t)(HC * KPitch<C>::P + C * (HC + 8)) * 2 + (size_t)256 * 4, PBYTES = (size_t)4 * 16 * (C + 4) * 4;
  static constexpr size_t bytes = WBYTES > PBYTES ? WBYTES : PBYTES;
};

// One workgroup's 64·TT rows of the MLP half.  HTILE: the wave's token rows are read from its LDS tile `htile` ([16·TT][C + 8],
// written by the projection half's epilogue in the fused block tail) instead of from p.h16; the tile may alias `smem`.
template <int C, int HC, int TT, bool HTILE>
__device__ __forceinline__ void mlp_fwd_body(const MlpArgs& p, char* smem, const bf16_t* htile, bf16_t* otile = nullptr) {
  constexpr bool RAG = (C % 32) != 0;  // C = 48: the last K-step of GEMM 1 is half empty (zero operands on both sides)
  constexpr int KJ = (C + 31) / 32;    // K-steps of GEMM 1 (K = C)
  constexpr int NT = C / 16;           // channel tiles of GEMM 2 / of the output
  constexpr int NB = HC / 32;          // 32-hidden blocks per chunk
  constexpr int P1 = KPitch<C>::P;     // pitch of the W1 chunk  [HC][P1]  (bf16 elements)
  constexpr int P2 = HC + 8;           // pitch of the W2 chunk  [C][P2]
  constexpr int CP = C + 4;            // pitch of the epilogue patch (floats); CP % 16 == 4: the 4 row groups hit disjoint banks
  constexpr int W1_EL = HC * P1, W2_EL = C * P2;
  constexpr int N1 = HC * C / 8, N2 = C * HC / 8;          // 16-byte pieces per chunk
  constexpr int PW1 = (N1 + 255) / 256, PW2 = (N2 + 255) / 256;
  // LDS (MlpFwdLds<C, HC>::bytes): weight chunks + b1 chunk (one slot per thread, HC used), later the epilogue's fp32 patches
  static_assert(RAG || (N1 % 256 == 0 && N2 % 256 == 0), "weight chunk pieces must divide over the 256 threads");
  static_assert(HC <= 256 && C % 16 == 0 && HC % 32 == 0 && (W1_EL * 2) % 16 == 0 && ((W1_EL + W2_EL) * 2) % 16 == 0, "layout");
  bf16_t* W1c = (bf16_t*)smem;
  bf16_t* W2c = W1c + W1_EL;
  float* b1c = (float*)(W2c + W2_EL);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lc = lane & 15;
  const int HID = p.hid, nch = HID / HC;
  const int row0 = (blockIdx.x * 4 + wave) * (16 * TT);     // first row of this wave

  // ---- this wave's token rows as GEMM-1 B operands (column = token lc, k = 32 j + 8 g ..), straight from HBM
  Frag<bf16_t> hf[TT][KJ];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const s16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (HTILE) {
#pragma unroll
      for (int j = 0; j < KJ; ++j)
        hf[tt][j].v = (!RAG || j * 32 + g * 8 < C) ? *(const s16x8_t*)(htile + (tt * 16 + lc) * (C + 8) + j * 32 + g * 8) : zero8;
    } else {
      const int row = min(row0 + tt * 16 + lc, p.M - 1);
      const bf16_t* src = p.h16 + (size_t)row * C + g * 8;
#pragma unroll
      for (int j = 0; j < KJ; ++j) hf[tt][j].v = (!RAG || j * 32 + g * 8 < C) ? *(const s16x8_t*)(src + j * 32) : zero8;
    }
  }
  if (HTILE) __syncthreads();          // every wave has its rows in registers: the tile may alias the weight chunk filled next

  // ---- weight chunk: global -> registers (unconditional, clamped piece index) -> LDS
  u32x4_t r1[PW1], r2[PW2];
  float rb1;
  auto load_chunk = [&](int c) {
    rb1 = p.b1[c * HC + min(tid, HC - 1)];                  // every thread loads AND stores (no load left pending on a skipped path)
    const bf16_t* s1 = p.W1 + (size_t)c * HC * C;           // HC full rows of W1: one contiguous block
#pragma unroll
    for (int u = 0; u < PW1; ++u) {
      const int i = min(tid + u * 256, N1 - 1);
      r1[u] = *(const u32x4_t*)(s1 + (size_t)i * 8);
    }
#pragma unroll
    for (int u = 0; u < PW2; ++u) {
      const int i = min(tid + u * 256, N2 - 1);
      const int row = i / (HC / 8), c8 = (i % (HC / 8)) * 8;
      r2[u] = *(const u32x4_t*)(p.W2 + (size_t)row * HID + (size_t)c * HC + c8);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int u = 0; u < PW1; ++u) {
      const int i = tid + u * 256;                           // < N1: the pieces divide evenly (static_assert), no guard —
      if (RAG && i >= N1) continue;                          // (C = 48 runs ONE chunk: nothing is left pending across a loop)
      const int x = i / (C / 8), k8 = (i % (C / 8)) * 8;     // a guarded store would leave its load unconsumed on the skipped path
      const int y = x & 31;                                  // x: hidden unit within the chunk
      const int rho = (x & ~31) + (((y >> 2) & 1) << 4) + ((y >> 3) << 2) + (y & 3);   // [blk][t][a][b] of y = 8a + 4t + b
      *(u32x4_t*)(W1c + rho * P1 + k8) = r1[u];
    }
#pragma unroll
    for (int u = 0; u < PW2; ++u) {
      const int i = tid + u * 256;
      if (RAG && i >= N2) continue;
      const int row = i / (HC / 8), c8 = (i % (HC / 8)) * 8;
      *(u32x4_t*)(W2c + row * P2 + c8) = r2[u];
    }
    b1c[tid] = rb1;
    if (RAG) {                                               // columns C .. of every W1 row: zero, once (no chunk store touches them)
      constexpr int PADP = (KPitch<C>::KP - C) / 8;
      const u32x4_t z = {0u, 0u, 0u, 0u};
      for (int i = tid; i < HC * PADP; i += 256) *(u32x4_t*)(W1c + (i / PADP) * P1 + C + (i % PADP) * 8) = z;
    }
  };

  f32x4_t Y[TT][NT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) Y[tt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  load_chunk(0);
  store_chunk();
  __syncthreads();

  auto multiply_chunk = [&](int c) {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      f32x4_t U[TT][2];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) { U[tt][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; U[tt][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const Frag<bf16_t> w = lds_frag_kc(W1c, P1, blk * 32 + t * 16, j * 32, lane);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) mma16(U[tt][t], w, hf[tt][j]);
        }
      }
      // lane (g, lc): token lc, hidden (chunk-local) 32 blk + 8 g + 4 t + r
      const float4 ba = *(const float4*)(b1c + blk * 32 + g * 8), bb = *(const float4*)(b1c + blk * 32 + g * 8 + 4);
      const float bias[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
      Frag<bf16_t> af[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        float av[8], dv[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = U[tt][t][r] + bias[4 * t + r];
            float cdf, e;
            gelu_terms(x, cdf, e);
            av[4 * t + r] = x * cdf;
            dv[4 * t + r] = cdf + x * 0.3989422804014327f * e;
          }
        af[tt] = frag_from_f32<bf16_t>(av);
        if (p.act || p.dact) {        // (either, both or neither: the round-3 backward needs gelu'(u) at most)
          const int row = row0 + tt * 16 + lc;
          if (row < p.M) {
            const size_t o = (size_t)row * HID + (size_t)c * HC + blk * 32 + g * 8;
            if (p.act) *(s16x8_t*)(p.act + o) = af[tt].v;
            if (p.dact) *(s16x8_t*)(p.dact + o) = frag_from_f32<bf16_t>(dv).v;
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const Frag<bf16_t> w = lds_frag_kc(W2c, P2, nt * 16, blk * 32, lane);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) mma16(Y[tt][nt], af[tt], w);
      }
    }
  };
  // The last chunk is peeled: a prefetch that is issued but never consumed inside the loop leaves loads pending on the
  // back edge, and the compiler then waits for EVERYTHING in flight (including the act/dact stores) at the top of each iteration.
  for (int c = 0; c + 1 < nch; ++c) {
    load_chunk(c + 1);                         // in flight during the multiply
    multiply_chunk(c);
    __syncthreads();                           // every wave is done reading chunk c
    store_chunk();
    __syncthreads();
  }
  multiply_chunk(nch - 1);
  __syncthreads();

  // ---- epilogue: + b2, layer norm over the row, conditional affine, DropPath scale, residual.  The weight chunk is dead
  // (barrier above): each wave stages one 16-row tile at a time in its own patch and re-reads it row-contiguously.
  ClnRowsOut e;
  e.bias = p.b2; e.z = p.z; e.z_dt = p.z_dt; e.mean = p.mean; e.rstd = p.rstd; e.time = p.time; e.gw_w = p.gw_w; e.gw_b = p.gw_b; e.bw_w = p.bw_w;
  e.bw_b = p.bw_b; e.sscale = p.sscale; e.resid = p.h; e.out = p.out; e.out16 = p.out16; e.M = p.M; e.rows_per_sample = p.rows_per_sample;
  e.eps = p.eps;
  cln_rows_epilogue<C, TT>(Y, (float*)smem, row0, e, otile);
}

template <int C, int HC, int TT>
__global__ __launch_bounds__(256, 2) void mlp_fused_kernel(MlpArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[MlpFwdLds<C, HC>::bytes];
  mlp_fwd_body<C, HC, TT, false>(p, smem, nullptr);
}

// Hidden units per LDS weight chunk.  The chunk's 16-byte pieces must divide evenly over the 256 threads (HC·C/8 % 256 == 0:
// with a ragged last piece the skipped LDS stores leave loads the compiler cannot prove consumed, and it then waits for all
// memory traffic at the top of every chunk iteration): 64 (27 KB of LDS at C = 96, 54 KB at C = 192).
static int mlp_chunk(int) { return 64; }     // (128-hidden chunks at C = 96: measured no change, round 2)

template <int C, int HC, int TT>
static int launch_mlp(const MlpArgs& a, hipStream_t s) {
  static_assert((HC * C / 8) % 256 == 0, "ragged weight chunk");
  const int rows_per_wg = 64 * TT;
  dim3 grid((a.M + rows_per_wg - 1) / rows_per_wg), block(256);
  hipLaunchKernelGGL((mlp_fused_kernel<C, HC, TT>), grid, block, 0, s, a);
  return scot_check_launch();
}

// token rows per workgroup of the fused block kernels, forced (64 x SCOT_MLP_TT) by the sanitizer / emulator harness (tests/test_hipemu_cpu.py);
// 0 = the launch policies' own choice.  The one environment read left in the library's launch paths.
static int scot_mlp_tt_override() {
  static int tt_env = -1;
  if (tt_env < 0) { const char* e = getenv("SCOT_MLP_TT"); tt_env = e ? atoi(e) : 0; }
  return tt_env;
}

// include/scot_hip.h: scot_mlp_block_fwd.  Returns SCOT_ERR_UNSUPPORTED for shapes this kernel does not cover (the caller
// then runs linear + linear + cln).
extern "C" int scot_mlp_block_fwd(const void* h16, const float* h, const void* W1, const float* b1, const void* W2, const float* b2,
                                  float* out, void* out16, void* act, void* dact, float* z, float* mean, float* rstd,
                                  const float* time, const float* gw_w, const float* gw_b, const float* bw_w, const float* bw_b,
                                  const float* sample_scale, int M, int rows_per_sample, int C, int hid, float eps,
                                  hipStream_t stream) {
  if (M <= 0 || rows_per_sample <= 0) return SCOT_ERR_SHAPE;
  if (C != 96 && C != 192) return SCOT_ERR_UNSUPPORTED;
  const int hc = mlp_chunk(C);
  if (hid < hc || hid % hc != 0) return SCOT_ERR_UNSUPPORTED;
  if (!h16 || !h || !W1 || !b1 || !W2 || !b2 || !out || !gw_b || !bw_b) return SCOT_ERR_SHAPE;
  if ((act == nullptr) != (dact == nullptr) || (mean == nullptr) != (rstd == nullptr) || (gw_w == nullptr) != (bw_w == nullptr))
    return SCOT_ERR_SHAPE;
  MlpArgs a;
  a.h16 = (const bf16_t*)h16; a.h = h; a.W1 = (const bf16_t*)W1; a.b1 = b1; a.W2 = (const bf16_t*)W2; a.b2 = b2;
  a.out = out; a.out16 = (bf16_t*)out16; a.act = (bf16_t*)act; a.dact = (bf16_t*)dact; a.z = z; a.z_dt = SCOT_F32; a.mean = mean; a.rstd = rstd;
  a.time = time; a.gw_w = gw_w; a.gw_b = gw_b; a.bw_w = bw_w; a.bw_b = bw_b; a.sscale = sample_scale;
  a.M = M; a.rows_per_sample = rows_per_sample; a.hid = hid; a.eps = eps;
  const int tt_env = scot_mlp_tt_override();
  // 64·TT rows per workgroup: TT = 2 halves the LDS weight reads per MFMA; TT = 1 when that would leave CUs without work
  // (C = 192 with TT = 2 needs 256 VGPRs + spills: TT = 1 unless forced)
  const int tt = tt_env ? tt_env : ((C == 96 && M >= 64 * 2 * 512) ? 2 : 1);
  if (C == 96) {
    return tt == 2 ? launch_mlp<96, 64, 2>(a, stream) : launch_mlp<96, 64, 1>(a, stream);
  }
  return tt == 2 ? launch_mlp<192, 64, 2>(a, stream) : launch_mlp<192, 64, 1>(a, stream);
}

// ------------------------------------------------------------------------------------------------------------------------
// Backward of the same block along the dependent chain, in one launch (the two weight gradients stay separate GEMMs on
// the side stream; they consume the dz and du written here):
//     dz  = CLN_bwd(s_b · g; z, mean, rstd)            (+= the four cond-LN parameter gradients)
//     du  = (dz · W2) ⊙ gelu'(u)
//     g'  = g + du · W1                                 (gradient wrt the block input h; may be written over g)
// replaces cln_bwd + dgrad fc2 (aux = gelu') + dgrad fc1 (into g) of engine.layer_bwd.  Same structure as the forward:
// GEMM 1 transposed (dA^T = W2^T · dz^T) so that, after the multiply by gelu'(u), a lane holds 8 consecutive hidden units of
// one token = the A operand of GEMM 2 and one 16-byte store of du.  Both weight chunks are K-strided here (W2[c][hidden]
// with k = c; W1[hidden][c] with k = hidden): the transposing LDS read (ds_read_b64_tr_b16) builds the fragments, and the
// permutation "lane 4a+b of tile t <- hidden 8a+4t+b" is folded into its chunk pointer instead of the LDS store.
// t (the conditioning time) must be uniform over a workgroup's rows: rows_per_sample % (64·TT) == 0.
struct MlpBwdArgs {
  const float* g; float* g_out;
  const void* z; int z_dt; const float* mean; const float* rstd;
  const float* time; const float* gw_w; const float* gw_b; const float* sscale;
  const bf16_t* dact; const bf16_t* W1; const bf16_t* W2;
  bf16_t* dz; bf16_t* du;                 // du may be NULL (the recomputing weight-gradient kernel does not read it)
  float* d_gw_w; float* d_gw_b; float* d_bw_w; float* d_bw_b;
  float* partial;                         // see ClnRowsBwd::partial
  const bf16_t* h16; const float* b1;     // RECOMP (dact == NULL): gelu'(u) is recomputed from u = h16·W1^T + b1 instead of loaded
  int M, rows_per_sample, hid, use_tr;
};

// A-operand fragment of GEMM 1 (backward): rows = hidden units 8a+4t+b (a = lane>>2 & 3, b = lane & 3 of the 16-lane
// group), k = channels klo..klo+3, khi..khi+3, from the K-strided chunk T[k][pitch] (column = chunk-local hidden unit).
__device__ __forceinline__ Frag<bf16_t> lds_frag_ks_perm(const bf16_t* t, int pitch, int h0, int tsel, int klo, int khi, int lane,
                                                         int use_tr) {
  Frag<bf16_t> f;
  const int i = lane & 15;
  if (use_tr) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const bf16_t* p0 = t + (klo + (i >> 2)) * pitch + h0 + (i & 3) * 8 + tsel * 4;
    const bf16_t* p1 = t + (khi + (i >> 2)) * pitch + h0 + (i & 3) * 8 + tsel * 4;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p1);
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
    const int col = h0 + (i >> 2) * 8 + tsel * 4 + (i & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f.v[j] = (short)t[(klo + j) * pitch + col];
      f.v[j + 4] = (short)t[(khi + j) * pitch + col];
    }
  }
  return f;
}

template <int C, int HC> struct MlpBwdLds {
  // (the W2 chunk is read with the channels as the fragment's k index: KPitch<C>::KP rows, the rows >= C zero)
  static constexpr size_t WBYTES = (size_t)(HC * (C + 8) + KPitch<C>::KP * (HC + 8)) * 2, PBYTES = (size_t)4 * 16 * (C + 4) * 4;
  static constexpr size_t P1BYTES = ClnBwdLds<C>::bytes;
  static constexpr size_t bytes = WBYTES > PBYTES ? (WBYTES > P1BYTES ? WBYTES : P1BYTES) : (PBYTES > P1BYTES ? PBYTES : P1BYTES);
  // RECOMP: the workgroup's 64·TT rows of h16 stay in LDS BEHIND the region above for the whole hidden loop ([64·TT][C + 8])
  static constexpr size_t htile_bytes(int TT) { return (size_t)64 * TT * (C + 8) * 2; }
};

// One workgroup's 64·TT rows of the MLP half's backward.  KEEP: the rows of g' = g + du·W1 are also returned in registers
// (gkeep[tt][pp][j]: row row0 + 16 tt + (lane >> 2), columns 32 pp + 8 (lane & 3) + j) for the fused block tail.
// GIN: the rows of g arrive in registers (gin, same layout as gkeep: the fused qkv-dgrad prologue produced them and also stored
// them to p.g, which phase 3 re-reads with the same lanes).
// RECOMP: gelu'(u) of the forward is not read from HBM (p.dact) but recomputed: u = h16·W1^T + b1 on the W1 chunk that is in LDS for
// the data gradient anyway (read K-contiguously here, with the row permutation 8a+4t+b of the forward's fragment convention applied
// at the read), bit-identical to the forward's u, and the derivative is rounded to 16 bits exactly as the forward used to store it.
template <int C, int HC, int TT, bool KEEP, bool GIN = false, bool RECOMP = false>
__device__ __forceinline__ void mlp_bwd_body(const MlpBwdArgs& p, char* smem, float (*gkeep)[(C + 31) / 32][8],
                                             const float (*gin)[(C + 31) / 32][8] = nullptr) {
  constexpr bool RAG = (C % 32) != 0;  // C = 48 (one hidden chunk, no recomputation): the channel K-steps' last 16 columns are zero on both sides
  constexpr int KJ = (C + 31) / 32, NT = C / 16, NB = HC / 32;
  constexpr int P1 = C + 8;            // W1 chunk [HC][P1]: k = hidden (rows), columns = channels
  constexpr int P2 = HC + 8;           // W2 chunk [C][P2]:  k = channels (rows), columns = hidden
  constexpr int CP = C + 4;            // epilogue patch pitch (floats)
  constexpr int W1_EL = HC * P1, W2_EL = C * P2;
  constexpr int N1 = HC * C / 8, N2 = C * HC / 8;
  constexpr int PW1 = (N1 + 255) / 256, PW2 = (N2 + 255) / 256;
  // one LDS region (MlpBwdLds<C, HC>::bytes), three lives: [dz patches | column sums] (phase 1) -> weight chunks (phase 2) ->
  // fp32 patches (phase 3)
  static_assert((W1_EL * 2) % 16 == 0, "layout");
  static_assert(RAG || (N1 % 256 == 0 && N2 % 256 == 0), "weight chunk pieces must divide over the 256 threads");
  static_assert(!RAG || !RECOMP, "the recomputing form reads W1 K-contiguously: C % 32 == 0 only");
  bf16_t* W1c = (bf16_t*)smem;
  bf16_t* W2c = W1c + W1_EL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lc = lane & 15;
  const int HID = p.hid, nch = HID / HC;
  const int wg_row0 = blockIdx.x * (64 * TT);
  const int row0 = wg_row0 + wave * (16 * TT);
  const int prow = lane >> 2, q = lane & 3;

  u32x4_t r1[PW1], r2[PW2];
  auto load_chunk = [&](int c) {
    const bf16_t* s1 = p.W1 + (size_t)c * HC * C;
#pragma unroll
    for (int u = 0; u < PW1; ++u) {
      const int i = min(tid + u * 256, N1 - 1);
      r1[u] = *(const u32x4_t*)(s1 + (size_t)i * 8);
    }
#pragma unroll
    for (int u = 0; u < PW2; ++u) {
      const int i = min(tid + u * 256, N2 - 1);
      const int row = i / (HC / 8), c8 = (i % (HC / 8)) * 8;
      r2[u] = *(const u32x4_t*)(p.W2 + (size_t)row * HID + (size_t)c * HC + c8);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int u = 0; u < PW1; ++u) {
      const int i = tid + u * 256;                 // pieces divide evenly (static_assert): unguarded, see the forward kernel
      if (RAG && i >= N1) continue;                // (C = 48 runs one chunk)
      *(u32x4_t*)(W1c + (i / (C / 8)) * P1 + (i % (C / 8)) * 8) = r1[u];
    }
#pragma unroll
    for (int u = 0; u < PW2; ++u) {
      const int i = tid + u * 256;
      if (RAG && i >= N2) continue;
      *(u32x4_t*)(W2c + (i / (HC / 8)) * P2 + (i % (HC / 8)) * 8) = r2[u];
    }
    if (RAG) {                                     // rows C .. KP-1 of the W2 chunk (k index of dz·W2 beyond the channels): zero
      const u32x4_t z = {0u, 0u, 0u, 0u};
      for (int i = tid; i < (KPitch<C>::KP - C) * (HC / 8); i += 256) *(u32x4_t*)(W2c + (C + i / (HC / 8)) * P2 + (i % (HC / 8)) * 8) = z;
    }
  };

  // ---- phase 1: dz = CLN_bwd(s·g) in the row-contiguous layout (4 lanes per row), parameter-gradient column sums
  ClnRowsBwd b;
  b.g = p.g; b.z = p.z; b.z_dt = p.z_dt; b.mean = p.mean; b.rstd = p.rstd; b.time = p.time; b.gw_w = p.gw_w; b.gw_b = p.gw_b; b.sscale = p.sscale;
  b.dz = p.dz; b.d_gw_w = p.d_gw_w; b.d_gw_b = p.d_gw_b; b.d_bw_w = p.d_bw_w; b.d_bw_b = p.d_bw_b; b.partial = p.partial; b.M = p.M;
  b.rows_per_sample = p.rows_per_sample;
  Frag<bf16_t> dzf[TT][KJ];
  cln_bwd_rows<C, TT, GIN>(dzf, smem, wg_row0, b, gin);
  // RECOMP: the wave's token rows of h16 — B operands of u^T = W1·h^T (column = token lc, k = 32 j + 8 g ..) — parked in the wave's own
  // LDS tile behind the weight chunks (24 registers per lane for the whole hidden loop otherwise: the kernel is at the 256-register cap)
  bf16_t* Hw = (bf16_t*)(smem + MlpBwdLds<C, HC>::bytes) + wave * 16 * TT * (C + 8);
  if (RECOMP) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const bf16_t* src = p.h16 + (size_t)min(row0 + tt * 16 + lc, p.M - 1) * C + g * 8;
#pragma unroll
      for (int j = 0; j < KJ; ++j) *(s16x8_t*)(Hw + (tt * 16 + lc) * (C + 8) + j * 32 + g * 8) = *(const s16x8_t*)(src + j * 32);
    }
    __builtin_amdgcn_wave_barrier();       // written and read by the same wave only
  }
  load_chunk(0);
  __syncthreads();                                             // the dz patches and `red` alias the weight chunk
  store_chunk();
  __syncthreads();

  // ---- phase 2: hidden chunks
  f32x4_t Y[TT][NT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) Y[tt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int rowt[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) rowt[tt] = row0 + tt * 16 + lc;

  auto multiply_chunk = [&](int c) {
    // gelu'(u) of this chunk for the lane's token and its 8 hidden units per block: in flight during the first MFMAs
    s16x8_t gpv[RECOMP ? 1 : TT][RECOMP ? 1 : NB];
    if (!RECOMP) {
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
#if SCOT_ABL & 64
          gpv[RECOMP ? 0 : tt][RECOMP ? 0 : blk] = (s16x8_t){15360, 15360, 15360, 15360, 15360, 15360, 15360, 15360};
#else
          gpv[RECOMP ? 0 : tt][RECOMP ? 0 : blk] = *(const s16x8_t*)(p.dact + (size_t)min(rowt[tt], p.M - 1) * HID + (size_t)c * HC + blk * 32 + g * 8);
#endif
    }
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      // RECOMP first, on its own: u^T of this block -> gelu'(u) packed to 16 bits (the value the forward used to store), so that
      // neither the recomputation's accumulators nor the GELU's temporaries are live beside the data-gradient product below
      s16x8_t gpr[RECOMP ? TT : 1];
      if (RECOMP) {
        f32x4_t R[TT][2];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) { R[tt][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; R[tt][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
          for (int ts = 0; ts < 2; ++ts) {
            // A operand: W1 rows (hidden units 8a + 4 ts + b of the block for lane 4a + b), 8 consecutive channels: K-contiguous
            Frag<bf16_t> w;
            w.v = *(const s16x8_t*)(W1c + (blk * 32 + ((lc >> 2) << 3) + ts * 4 + (lc & 3)) * P1 + j * 32 + g * 8);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
              Frag<bf16_t> hfr;
              hfr.v = *(const s16x8_t*)(Hw + (tt * 16 + lc) * (C + 8) + j * 32 + g * 8);
              mma16(R[tt][ts], w, hfr);
            }
          }
        const float4 ba = *(const float4*)(p.b1 + (size_t)c * HC + blk * 32 + g * 8), bb = *(const float4*)(p.b1 + (size_t)c * HC + blk * 32 + g * 8 + 4);
        const float b1v[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          float dg[8];
#pragma unroll
          for (int ts = 0; ts < 2; ++ts)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float x = R[tt][ts][r] + b1v[4 * ts + r];
              float cdf, e;
              gelu_terms(x, cdf, e);
              dg[4 * ts + r] = cdf + x * 0.3989422804014327f * e;
            }
          gpr[RECOMP ? tt : 0] = frag_from_f32<bf16_t>(dg).v;
        }
      }
      f32x4_t U[TT][2];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) { U[tt][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; U[tt][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int j = 0; j < KJ; ++j) {
#pragma unroll
        for (int ts = 0; ts < 2; ++ts) {
          const Frag<bf16_t> w = lds_frag_ks_perm(W2c, P2, blk * 32, ts, j * 32 + g * 8, j * 32 + g * 8 + 4, lane, p.use_tr);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) mma16(U[tt][ts], w, dzf[tt][j]);
        }
      }
      Frag<bf16_t> af[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        float dv[8];
#pragma unroll
        for (int ts = 0; ts < 2; ++ts)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dv[4 * ts + r] = U[tt][ts][r] * bf2f((bf16_t)(RECOMP ? gpr[RECOMP ? tt : 0][4 * ts + r] : gpv[RECOMP ? 0 : tt][RECOMP ? 0 : blk][4 * ts + r]));
        af[tt] = frag_from_f32<bf16_t>(dv);
#if SCOT_ABL & 128
        asm volatile("" ::"v"(af[tt].v));
#else
        if (p.du && rowt[tt] < p.M) *(s16x8_t*)(p.du + (size_t)rowt[tt] * HID + (size_t)c * HC + blk * 32 + g * 8) = af[tt].v;
#endif
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const Frag<bf16_t> w = lds_frag_ks(W1c, P1, nt * 16, blk * 32 + g * 8, blk * 32 + g * 8 + 4, lane, p.use_tr);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) mma16(Y[tt][nt], af[tt], w);
      }
    }
  };
  for (int c = 0; c + 1 < nch; ++c) {          // last chunk peeled, see the forward kernel
    load_chunk(c + 1);
    multiply_chunk(c);
    __syncthreads();
    store_chunk();
    __syncthreads();
  }
  multiply_chunk(nch - 1);
  __syncthreads();

  // ---- phase 3: g' = g + du·W1, through the per-wave patch (aliases the dead weight chunk) for row-contiguous stores
  float* Ct = (float*)smem + wave * 16 * CP;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(g * 4 + r) * CP + nt * 16 + lc] = Y[tt][nt][r];
    __builtin_amdgcn_wave_barrier();
    const int grow = row0 + tt * 16 + prow;
    if (grow < p.M) {
      const size_t base = (size_t)grow * C;
#pragma unroll
      for (int pp = 0; pp < KJ; ++pp) {
        const int col = pp * 32 + q * 8;
        if (RAG && col >= C) {
          if (KEEP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gkeep[tt][pp][j] = 0.f;
          }
          continue;
        }
        const float4 x0 = *(const float4*)(Ct + prow * CP + col), x1 = *(const float4*)(Ct + prow * CP + col + 4);
        float gi[8], o[8];
        ld8(p.g, SCOT_F32, base + col, gi);
        o[0] = gi[0] + x0.x; o[1] = gi[1] + x0.y; o[2] = gi[2] + x0.z; o[3] = gi[3] + x0.w;
        o[4] = gi[4] + x1.x; o[5] = gi[5] + x1.y; o[6] = gi[6] + x1.z; o[7] = gi[7] + x1.w;
        st8(p.g_out, SCOT_F32, base + col, o);
        if (KEEP) {
#pragma unroll
          for (int j = 0; j < 8; ++j) gkeep[tt][pp][j] = o[j];
        }
      }
    } else if (KEEP) {
#pragma unroll
      for (int pp = 0; pp < KJ; ++pp)
#pragma unroll
        for (int j = 0; j < 8; ++j) gkeep[tt][pp][j] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int C, int HC, int TT>
__global__ __launch_bounds__(256, 2) void mlp_bwd_fused_kernel(MlpBwdArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[MlpBwdLds<C, HC>::bytes];
  mlp_bwd_body<C, HC, TT, false>(p, smem, nullptr);
}

extern int g_scot_use_tr;

template <int C, int HC, int TT>
static int launch_mlp_bwd(const MlpBwdArgs& a, hipStream_t s) {
  dim3 grid((a.M + 64 * TT - 1) / (64 * TT)), block(256);
  hipLaunchKernelGGL((mlp_bwd_fused_kernel<C, HC, TT>), grid, block, 0, s, a);
  return scot_check_launch();
}

// include/scot_hip.h: scot_mlp_block_bwd
extern "C" int scot_mlp_block_bwd(const float* g, float* g_out, const float* z, const float* mean, const float* rstd,
                                  const float* time, const float* gw_w, const float* gw_b, const float* sample_scale,
                                  const void* dact, const void* W1, const void* W2, void* dz, void* du, float* d_gw_w,
                                  float* d_gw_b, float* d_bw_w, float* d_bw_b, int M, int rows_per_sample, int C, int hid,
                                  hipStream_t stream) {
  if (M <= 0 || rows_per_sample <= 0) return SCOT_ERR_SHAPE;
  if (C != 96 && C != 192) return SCOT_ERR_UNSUPPORTED;
  const int hc = mlp_chunk(C);
  if (hid < hc || hid % hc != 0) return SCOT_ERR_UNSUPPORTED;
  if (!g || !g_out || !z || !mean || !rstd || !gw_b || !dact || !W1 || !W2 || !dz || !du || !d_gw_b || !d_bw_b) return SCOT_ERR_SHAPE;
  if ((gw_w == nullptr) != (d_gw_w == nullptr) || (d_gw_w == nullptr) != (d_bw_w == nullptr)) return SCOT_ERR_SHAPE;
  const int tt_env = scot_mlp_tt_override();
  int tt = tt_env ? tt_env : ((C == 96 && M >= 64 * 2 * 512) ? 2 : 1);
  if (rows_per_sample % (64 * tt) != 0) tt = 1;
  if (rows_per_sample % 64 != 0) return SCOT_ERR_UNSUPPORTED;      // the conditioning time must be uniform per workgroup
  MlpBwdArgs a;
  a.g = g; a.g_out = g_out; a.z = z; a.z_dt = SCOT_F32; a.mean = mean; a.rstd = rstd; a.time = time; a.gw_w = gw_w; a.gw_b = gw_b; a.sscale = sample_scale;
  a.partial = nullptr; a.h16 = nullptr; a.b1 = nullptr;
  a.dact = (const bf16_t*)dact; a.W1 = (const bf16_t*)W1; a.W2 = (const bf16_t*)W2; a.dz = (bf16_t*)dz; a.du = (bf16_t*)du;
  a.d_gw_w = d_gw_w; a.d_gw_b = d_gw_b; a.d_bw_w = d_bw_w; a.d_bw_b = d_bw_b;
  a.M = M; a.rows_per_sample = rows_per_sample; a.hid = hid; a.use_tr = g_scot_use_tr;
  if (C == 96) {
    return tt == 2 ? launch_mlp_bwd<96, 64, 2>(a, stream) : launch_mlp_bwd<96, 64, 1>(a, stream);
  }
  return tt == 2 ? launch_mlp_bwd<192, 64, 2>(a, stream) : launch_mlp_bwd<192, 64, 1>(a, stream);
}

// ------------------------------------------------------------------------------------------------------------------------
// The attention half's tail, same row ownership:  out = x + s_b · CLN(a · W^T + b)   (Swinv2SelfOutput + res-post-norm,
// HF:478-489, reference model.py:560-565) — the out-projection GEMM with the layer norm in its epilogue (one workgroup owns
// whole rows because N = C <= 192), and its backward along the chain:  dz = CLN_bwd(s_b · g),  da = dz · W.
struct ProjClnArgs {
  const bf16_t* a; const bf16_t* W;      // [M, C] attention output (operand dtype), [C, C] weight (N x K)
  ClnRowsOut e;
};

template <int C> struct ProjFwdLds {
  static constexpr size_t WBYTES = (size_t)C * (96 + 8) * 2, PBYTES = (size_t)4 * 16 * (C + 4) * 4;
  static constexpr size_t bytes = WBYTES > PBYTES ? WBYTES : PBYTES;
};

// `tile16`: see cln_rows_epilogue (per-wave LDS tile of the 16-bit output rows, beyond the fp32 patches), or nullptr.
template <int C, int TT>
__device__ __forceinline__ void proj_cln_fwd_body(const ProjClnArgs& p, char* smem, bf16_t* tile16) {
  constexpr bool RAG = (C % 32) != 0;              // C = 48: one chunk of 64 K columns, the last 16 zero
  constexpr int KJ = (C + 31) / 32, NT = C / 16, KC = RAG ? KJ * 32 : 96, NKC = RAG ? 1 : C / KC;
  constexpr int KCR = RAG ? C : KC;                // columns of a chunk that exist in W
  constexpr int PW = KC + 8;                       // W chunk [C][PW]: K-contiguous columns kc·96 .. +95 of every row
  constexpr int NP = C * KCR / 8, PWN = (NP + 255) / 256;
  static_assert(RAG || C % KC == 0, "K chunking");
  bf16_t* Wc = (bf16_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lc = lane & 15;
  const int row0 = (blockIdx.x * 4 + wave) * (16 * TT);

  Frag<bf16_t> af[TT][KJ];                         // A operand straight from HBM: row = token lc, k = 32 j + 8 g ..
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const bf16_t* src = p.a + (size_t)min(row0 + tt * 16 + lc, p.e.M - 1) * C + g * 8;
    const s16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < KJ; ++j) af[tt][j].v = (!RAG || j * 32 + g * 8 < C) ? *(const s16x8_t*)(src + j * 32) : zero8;
  }
  f32x4_t Y[TT][NT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) Y[tt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  u32x4_t rw[PWN];
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
#pragma unroll
    for (int u = 0; u < PWN; ++u) {
      const int i = min(tid + u * 256, NP - 1);
      rw[u] = *(const u32x4_t*)(p.W + (size_t)(i / (KCR / 8)) * C + kc * KC + (i % (KCR / 8)) * 8);
    }
    if (kc) __syncthreads();                       // every wave is done with the previous chunk
#pragma unroll
    for (int u = 0; u < PWN; ++u) {
      const int i = tid + u * 256;
      if (i < NP) *(u32x4_t*)(Wc + (i / (KCR / 8)) * PW + (i % (KCR / 8)) * 8) = rw[u];
    }
    if (RAG) {
      constexpr int PADP = (KC - KCR) / 8;
      const u32x4_t z = {0u, 0u, 0u, 0u};
      for (int i = tid; i < C * PADP; i += 256) *(u32x4_t*)(Wc + (i / PADP) * PW + KCR + (i % PADP) * 8) = z;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KC / 32; ++j)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const Frag<bf16_t> w = lds_frag_kc(Wc, PW, nt * 16, j * 32, lane);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) mma16(Y[tt][nt], af[tt][kc * (KC / 32) + j], w);
      }
  }
  __syncthreads();                                 // the epilogue patches alias the weight chunk
  cln_rows_epilogue<C, TT>(Y, (float*)smem, row0, p.e, tile16);
}

template <int C, int TT>
__global__ __launch_bounds__(256, 2) void proj_cln_fused_kernel(ProjClnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[ProjFwdLds<C>::bytes];
  proj_cln_fwd_body<C, TT>(p, smem, nullptr);
}

// ------------------------------------------------------------------------------------------------------------------------
// The whole tail of a ScOTLayer's forward in ONE launch: h = x + s·CLN(attn · Wo^T + bo), then out = h + s·CLN(MLP(h)), for the
// same 64·TT rows.  h is written once (fp32 + 16-bit: the backward and the weight gradients read them) but not re-read: the MLP
// half takes its operand rows from the LDS tile the projection half's epilogue filled, and its residual from the values the
// same lanes just stored.
struct TailFwdArgs {
  ProjClnArgs pj; MlpArgs m;
  const bf16_t* Wqkv; const float* bqkv; bf16_t* qkv;    // optional epilogue: qkv[M, 3C] = out16 · Wqkv[3C, C]^T + bqkv (next layer)
};

// The NEXT layer's fused q/k/v projection (HF:396-410) on the rows this workgroup has just produced: qkv = out16 · Wqkv^T + b.
// Same row ownership as the tail, so the stand-alone QKV GEMM launch (and its re-read of out16) leaves the chain.  `tile`: the
// wave's 16-bit output rows in LDS ([16·TT][C + 8], written by the final epilogue); the weight chunk and the patches reuse smem.
template <int C> struct QkvEpi {
  static constexpr int NC = (3 * C) % 96 == 0 ? 96 : 48;
  static constexpr size_t bytes = (size_t)NC * KPitch<C>::P * 2 + (size_t)4 * 16 * (NC + 4) * 4;       // weight chunk + patches
};
template <int C, int TT>
__device__ __forceinline__ void qkv_epilogue(const bf16_t* tile, const bf16_t* Wqkv, const float* bqkv, bf16_t* qkv, int M, char* smem) {
  constexpr bool RAG = (C % 32) != 0;
  constexpr int KJ = (C + 31) / 32, NC = QkvEpi<C>::NC, NCH = 3 * C / NC;          // 96 (C = 48: 48) output columns per weight chunk
  constexpr int PW = KPitch<C>::P, NP = NC * C / 8, PWN = (NP + 255) / 256, CP = NC + 4;
  static_assert((3 * C) % NC == 0 && NC % 16 == 0, "qkv column chunks");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lc = lane & 15;
  const int row0 = (blockIdx.x * 4 + wave) * (16 * TT);
  Frag<bf16_t> af[TT][KJ];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      const s16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
      af[tt][j].v = (!RAG || j * 32 + g * 8 < C) ? *(const s16x8_t*)(tile + (tt * 16 + lc) * (C + 8) + j * 32 + g * 8) : zero8;
    }
  bf16_t* Wc = (bf16_t*)smem;                                    // [NC][PW], K-contiguous rows (= output columns)
  float* Ct = (float*)(smem + (size_t)NC * PW * 2) + wave * 16 * CP;
  const int prow = lane >> 2, q = lane & 3;
  u32x4_t rw[PWN];
#pragma unroll 1
  for (int nc = 0; nc < NCH; ++nc) {
    const bf16_t* src = Wqkv + (size_t)nc * NC * C;              // NC full rows: one contiguous block
#pragma unroll
    for (int u = 0; u < PWN; ++u) rw[u] = *(const u32x4_t*)(src + (size_t)min(tid + u * 256, NP - 1) * 8);
    __syncthreads();                                             // the tile reads (nc = 0) / the previous chunk's reads are done
#pragma unroll
    for (int u = 0; u < PWN; ++u) {
      const int i = tid + u * 256;
      if (i < NP) *(u32x4_t*)(Wc + (i / (C / 8)) * PW + (i % (C / 8)) * 8) = rw[u];
    }
    if (RAG) {
      constexpr int PADP = (KPitch<C>::KP - C) / 8;
      const u32x4_t z = {0u, 0u, 0u, 0u};
      for (int i = tid; i < NC * PADP; i += 256) *(u32x4_t*)(Wc + (i / PADP) * PW + C + (i % PADP) * 8) = z;
    }
    __syncthreads();
    f32x4_t Y[TT][NC / 16];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int nt = 0; nt < NC / 16; ++nt) Y[tt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KJ; ++j)
#pragma unroll
      for (int nt = 0; nt < NC / 16; ++nt) {
        const Frag<bf16_t> w = lds_frag_kc(Wc, PW, nt * 16, j * 32, lane);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) mma16(Y[tt][nt], af[tt][j], w);
      }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
      for (int nt = 0; nt < NC / 16; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ct[(g * 4 + r) * CP + nt * 16 + lc] = Y[tt][nt][r];
      __builtin_amdgcn_wave_barrier();
      const int grow = row0 + tt * 16 + prow;
      if (grow < M) {
#pragma unroll
        for (int pp = 0; pp < (NC + 31) / 32; ++pp) {
          const int col = pp * 32 + q * 8;
          if (NC % 32 != 0 && col >= NC) continue;
          const float4 x0 = *(const float4*)(Ct + prow * CP + col), x1 = *(const float4*)(Ct + prow * CP + col + 4);
          float o[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
          if (bqkv) {
            float bb[8];
            ld8(bqkv, SCOT_F32, nc * NC + col, bb);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += bb[j];
          }
          st8(qkv, SCOT_BF16, (size_t)grow * (3 * C) + nc * NC + col, o);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int C, int HC, int TT, bool QKV>
__global__ __launch_bounds__(256, 2) void tail_fwd_fused_kernel(TailFwdArgs p) {
  constexpr size_t PATCH = (size_t)4 * 16 * (C + 4) * 4, TILE = (size_t)4 * 16 * TT * (C + 8) * 2;
  constexpr size_t A = ProjFwdLds<C>::bytes > PATCH + TILE ? ProjFwdLds<C>::bytes : PATCH + TILE;
  constexpr size_t QB = QkvEpi<C>::bytes;           // qkv epilogue: weight chunk + patches
  constexpr size_t B = A > MlpFwdLds<C, HC>::bytes ? A : MlpFwdLds<C, HC>::bytes;
  constexpr size_t LDS = (QKV && QB > B) ? QB : B;
  __shared__ __attribute__((aligned(16))) char smem[LDS];
  bf16_t* tile = (bf16_t*)(smem + PATCH) + (threadIdx.x >> 6) * 16 * TT * (C + 8);
  proj_cln_fwd_body<C, TT>(p.pj, smem, tile);
  __builtin_amdgcn_wave_barrier();       // the tile is written and read by the same wave
  mlp_fwd_body<C, HC, TT, true>(p.m, smem, tile, QKV ? tile : nullptr);
  if (QKV) {
    __builtin_amdgcn_wave_barrier();
    qkv_epilogue<C, TT>(tile, p.Wqkv, p.bqkv, p.qkv, p.m.M, smem);
  }
}

template <int C, int HC, int TT>
static int launch_tail_fwd(const TailFwdArgs& a, hipStream_t s) {
  dim3 grid((a.m.M + 64 * TT - 1) / (64 * TT)), block(256);
  if (a.qkv) hipLaunchKernelGGL((tail_fwd_fused_kernel<C, HC, TT, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((tail_fwd_fused_kernel<C, HC, TT, false>), grid, block, 0, s, a);
  return scot_check_launch();
}

struct ProjClnBwdArgs {
  const bf16_t* W;                       // [C, C] (N x K): da[:, k] = Σ_n dz[:, n] · W[n, k]
  bf16_t* da;                            // [M, C]
  ClnRowsBwd b;
  int use_tr;
};

template <int C> struct ProjBwdLds {
  static constexpr size_t WBYTES = (size_t)96 * (C + 8) * 2, PBYTES = (size_t)4 * 16 * (C + 4) * 4, P1BYTES = ClnBwdLds<C>::bytes;
  static constexpr size_t bytes = WBYTES > PBYTES ? (WBYTES > P1BYTES ? WBYTES : P1BYTES) : (PBYTES > P1BYTES ? PBYTES : P1BYTES);
};

// GREG: g rows come in registers (fused block tail), see cln_bwd_rows.  `smem`: ProjBwdLds<C>::bytes, dead on entry.
template <int C, int TT, bool GREG>
__device__ __forceinline__ void proj_cln_bwd_body(const ProjClnBwdArgs& p, char* smem, const float (*greg)[(C + 31) / 32][8]) {
  constexpr bool RAG = (C % 32) != 0;              // C = 48: one chunk of 64 rows n, the last 16 zero
  constexpr int NT = C / 16, KC = RAG ? KPitch<C>::KP : 96, NKC = RAG ? 1 : C / KC;
  constexpr int KCR = RAG ? C : KC;                // rows of a chunk that exist in W
  constexpr int PW = C + 8;                        // W chunk [KC rows n][PW]: the contraction index runs over rows (K-strided)
  constexpr int NP = KCR * C / 8, PWN = (NP + 255) / 256;
  constexpr int CP = C + 4;
  bf16_t* Wc = (bf16_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lc = lane & 15;
  const int wg_row0 = blockIdx.x * (64 * TT);
  const int row0 = wg_row0 + wave * (16 * TT);

  Frag<bf16_t> dzf[TT][(C + 31) / 32];
  cln_bwd_rows<C, TT, GREG>(dzf, smem, wg_row0, p.b, greg);
  f32x4_t Y[TT][NT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) Y[tt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  u32x4_t rw[PWN];
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    const bf16_t* src = p.W + (size_t)kc * KC * C;               // KC full rows: one contiguous block
#pragma unroll
    for (int u = 0; u < PWN; ++u) rw[u] = *(const u32x4_t*)(src + (size_t)min(tid + u * 256, NP - 1) * 8);
    __syncthreads();                               // phase-1 LDS (kc = 0) / the previous chunk (kc > 0) is dead
#pragma unroll
    for (int u = 0; u < PWN; ++u) {
      const int i = tid + u * 256;
      if (i < NP) *(u32x4_t*)(Wc + (i / (C / 8)) * PW + (i % (C / 8)) * 8) = rw[u];
    }
    if (RAG) {
      const u32x4_t z = {0u, 0u, 0u, 0u};
      for (int i = tid; i < (KC - KCR) * (C / 8); i += 256) *(u32x4_t*)(Wc + (KCR + i / (C / 8)) * PW + (i % (C / 8)) * 8) = z;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KC / 32; ++j)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const Frag<bf16_t> w = lds_frag_ks(Wc, PW, nt * 16, j * 32 + g * 8, j * 32 + g * 8 + 4, lane, p.use_tr);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) mma16(Y[tt][nt], dzf[tt][kc * (KC / 32) + j], w);
      }
  }
  __syncthreads();
  // da rows through the per-wave fp32 patch -> 16-byte bf16 row segments
  float* Ct = (float*)smem + wave * 16 * CP;
  const int prow = lane >> 2, q = lane & 3;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(g * 4 + r) * CP + nt * 16 + lc] = Y[tt][nt][r];
    __builtin_amdgcn_wave_barrier();
    const int grow = row0 + tt * 16 + prow;
    if (grow < p.b.M) {
#pragma unroll
      for (int pp = 0; pp < (C + 31) / 32; ++pp) {
        const int col = pp * 32 + q * 8;
        if (RAG && col >= C) continue;
        const float4 x0 = *(const float4*)(Ct + prow * CP + col), x1 = *(const float4*)(Ct + prow * CP + col + 4);
        const float o[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        st8(p.da, SCOT_BF16, (size_t)grow * C + col, o);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int C, int TT>
__global__ __launch_bounds__(256, 2) void proj_cln_bwd_fused_kernel(ProjClnBwdArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[ProjBwdLds<C>::bytes];
  proj_cln_bwd_body<C, TT, false>(p, smem, nullptr);
}

// ------------------------------------------------------------------------------------------------------------------------
// The whole tail of a ScOTLayer's backward in ONE launch: MLP half (cond-LN backward -> dgrad fc2 · gelu' -> dgrad fc1 -> + g) and
// attention-output half (cond-LN backward -> projection dgrad) for the same 64·TT rows.  The gradient of the residual stream
// between the two halves stays in registers (it is still written once: the qkv dgrad accumulates into it), which removes one
// launch and one 4·C-byte-per-token read from the dependent chain of the token-heavy stages.
struct TailBwdArgs {
  MlpBwdArgs m; ProjClnBwdArgs pj;
  const bf16_t* dqkv; const bf16_t* Wqkv;     // optional prologue: g += dqkv[M, 3C] · Wqkv[3C, C] (the qkv dgrad of the layer above)
};

// The qkv projection's data gradient of the layer processed just before (HF:396-410: dx = dq Wq + dk Wk + dv Wv), as a prologue of
// this layer's tail: g rows += dqkv rows · Wqkv — same row ownership, so the stand-alone GEMM launch and one read-modify-write
// pass over the residual-stream gradient disappear from the chain.  Writes g in place AND returns the rows in registers.
template <int C, int TT>
__device__ __forceinline__ void qkv_dgrad_prologue(const bf16_t* dqkv, const bf16_t* Wqkv, float* gio, int M, char* smem, int use_tr,
                                                   float (*gk)[C / 32][8]) {
  constexpr int K3 = 3 * C, NT = C / 16, KC = 96, NKC = K3 / KC;
  constexpr int PW = C + 8, NP = KC * C / 8, PWN = (NP + 255) / 256, CP = C + 4;
  bf16_t* Wc = (bf16_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lc = lane & 15;
  const int row0 = (blockIdx.x * 4 + wave) * (16 * TT);
  f32x4_t Y[TT][NT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) Y[tt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  u32x4_t rw[PWN];
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    const bf16_t* src = Wqkv + (size_t)kc * KC * C;              // KC full rows of [3C, C]: one contiguous block
#pragma unroll
    for (int u = 0; u < PWN; ++u) rw[u] = *(const u32x4_t*)(src + (size_t)min(tid + u * 256, NP - 1) * 8);
    // A operand of this K chunk straight from HBM (row = token lc, k = kc·96 + 32 j + 8 g ..): per chunk, not all 3C columns at
    // once — 18 fragments up front spilled at C = 192
    Frag<bf16_t> af[TT][KC / 32];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const bf16_t* arow = dqkv + (size_t)min(row0 + tt * 16 + lc, M - 1) * K3 + kc * KC + g * 8;
#pragma unroll
      for (int j = 0; j < KC / 32; ++j) af[tt][j].v = *(const s16x8_t*)(arow + j * 32);
    }
    if (kc) __syncthreads();                       // every wave is done with the previous chunk
#pragma unroll
    for (int u = 0; u < PWN; ++u) {
      const int i = tid + u * 256;
      if (i < NP) *(u32x4_t*)(Wc + (i / (C / 8)) * PW + (i % (C / 8)) * 8) = rw[u];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KC / 32; ++j)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const Frag<bf16_t> w = lds_frag_ks(Wc, PW, nt * 16, j * 32 + g * 8, j * 32 + g * 8 + 4, lane, use_tr);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) mma16(Y[tt][nt], af[tt][j], w);
      }
  }
  __syncthreads();                                 // the fp32 patches alias the weight chunk
  float* Ct = (float*)smem + wave * 16 * CP;
  const int prow = lane >> 2, q = lane & 3;
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ct[(g * 4 + r) * CP + nt * 16 + lc] = Y[tt][nt][r];
    __builtin_amdgcn_wave_barrier();
    const int grow = row0 + tt * 16 + prow;
#pragma unroll
    for (int pp = 0; pp < C / 32; ++pp) {
      const int col = pp * 32 + q * 8;
      if (grow < M) {
        const float4 x0 = *(const float4*)(Ct + prow * CP + col), x1 = *(const float4*)(Ct + prow * CP + col + 4);
        float gi[8];
        ld8(gio, SCOT_F32, (size_t)grow * C + col, gi);
        gi[0] += x0.x; gi[1] += x0.y; gi[2] += x0.z; gi[3] += x0.w; gi[4] += x1.x; gi[5] += x1.y; gi[6] += x1.z; gi[7] += x1.w;
        st8(gio, SCOT_F32, (size_t)grow * C + col, gi);
#pragma unroll
        for (int j = 0; j < 8; ++j) gk[tt][pp][j] = gi[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) gk[tt][pp][j] = 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int C, int HC, int TT, bool PRO, bool RECOMP>
__global__ __launch_bounds__(256, 2) void tail_bwd_fused_kernel(TailBwdArgs p) {
  constexpr size_t L0 = MlpBwdLds<C, HC>::bytes > ProjBwdLds<C>::bytes ? MlpBwdLds<C, HC>::bytes : ProjBwdLds<C>::bytes;
  constexpr size_t L1 = MlpBwdLds<C, HC>::bytes + (RECOMP ? MlpBwdLds<C, HC>::htile_bytes(TT) : 0);
  constexpr size_t LDS = L0 > L1 ? L0 : L1;        // (the prologue's weight chunk + patches fit inside ProjBwdLds)
  __shared__ __attribute__((aligned(16))) char smem[LDS];
  float gk[TT][(C + 31) / 32][8];
  if constexpr (PRO) {
    static_assert(C % 32 == 0, "the qkv-dgrad prologue exists for C % 32 == 0 only");
    qkv_dgrad_prologue<C, TT>(p.dqkv, p.Wqkv, (float*)p.m.g, p.m.M, smem, p.m.use_tr, gk);
    __syncthreads();                     // the prologue's patches are dead
    // C = 96 without recomputation: the updated rows go on in registers; otherwise (no registers to spare) the norm re-reads the
    // rows its own lanes have just stored
    if (C <= 96 && !RECOMP) mlp_bwd_body<C, HC, TT, true, true, RECOMP>(p.m, smem, gk, gk);
    else mlp_bwd_body<C, HC, TT, true, false, RECOMP>(p.m, smem, gk);
  } else {
    mlp_bwd_body<C, HC, TT, true, false, RECOMP>(p.m, smem, gk);
  }
  __syncthreads();                       // the MLP half's fp32 patches are dead: the norm's dz patches take their place
  proj_cln_bwd_body<C, TT, true>(p.pj, smem, gk);
}

template <int C, int HC, int TT>
static int launch_tail_bwd(const TailBwdArgs& a, hipStream_t s) {
  dim3 grid((a.m.M + 64 * TT - 1) / (64 * TT)), block(256);
  if (a.m.dact == nullptr) {     // gelu'(u) recomputed in the kernel
    if (a.dqkv) hipLaunchKernelGGL((tail_bwd_fused_kernel<C, HC, TT, true, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((tail_bwd_fused_kernel<C, HC, TT, false, true>), grid, block, 0, s, a);
  } else {
    if (a.dqkv) hipLaunchKernelGGL((tail_bwd_fused_kernel<C, HC, TT, true, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((tail_bwd_fused_kernel<C, HC, TT, false, false>), grid, block, 0, s, a);
  }
  return scot_check_launch();
}

static int rows_tile_count(int C, int M, int rows_per_sample) {
  const int tt_env = scot_mlp_tt_override();
  int tt = tt_env ? tt_env : ((C == 96 && M >= 64 * 2 * 512) ? 2 : 1);
  if (rows_per_sample % (64 * tt) != 0) tt = 1;
  return tt;
}

// include/scot_hip.h: scot_proj_cln_fwd
extern "C" int scot_proj_cln_fwd(const void* a, const void* W, const float* bias, const float* resid, float* out, void* out16,
                                 float* z, float* mean, float* rstd, const float* time, const float* gw_w, const float* gw_b,
                                 const float* bw_w, const float* bw_b, const float* sample_scale, int M, int rows_per_sample,
                                 int C, float eps, hipStream_t stream) {
  if (M <= 0 || rows_per_sample <= 0) return SCOT_ERR_SHAPE;
  if (C != 96 && C != 192) return SCOT_ERR_UNSUPPORTED;
  if (!a || !W || !bias || !resid || !out || !gw_b || !bw_b) return SCOT_ERR_SHAPE;
  if ((mean == nullptr) != (rstd == nullptr) || (gw_w == nullptr) != (bw_w == nullptr)) return SCOT_ERR_SHAPE;
  ProjClnArgs p;
  p.a = (const bf16_t*)a; p.W = (const bf16_t*)W;
  p.e.bias = bias; p.e.z = z; p.e.z_dt = SCOT_F32; p.e.mean = mean; p.e.rstd = rstd; p.e.time = time; p.e.gw_w = gw_w; p.e.gw_b = gw_b; p.e.bw_w = bw_w;
  p.e.bw_b = bw_b; p.e.sscale = sample_scale; p.e.resid = resid; p.e.out = out; p.e.out16 = (bf16_t*)out16; p.e.M = M;
  p.e.rows_per_sample = rows_per_sample; p.e.eps = eps;
  const int tt = rows_tile_count(C, M, 64 * 2);    // no per-workgroup uniformity needed in the forward
  dim3 grid((M + 64 * tt - 1) / (64 * tt)), block(256);
  if (C == 96) {
    if (tt == 2) hipLaunchKernelGGL((proj_cln_fused_kernel<96, 2>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((proj_cln_fused_kernel<96, 1>), grid, block, 0, stream, p);
  } else {
    hipLaunchKernelGGL((proj_cln_fused_kernel<192, 1>), dim3((M + 63) / 64), block, 0, stream, p);
  }
  return scot_check_launch();
}

// include/scot_hip.h: scot_proj_cln_bwd
extern "C" int scot_proj_cln_bwd(const float* g, const float* z, const float* mean, const float* rstd, const float* time,
                                 const float* gw_w, const float* gw_b, const float* sample_scale, const void* W, void* dz, void* da,
                                 float* d_gw_w, float* d_gw_b, float* d_bw_w, float* d_bw_b, int M, int rows_per_sample, int C,
                                 hipStream_t stream) {
  if (M <= 0 || rows_per_sample <= 0) return SCOT_ERR_SHAPE;
  if ((C != 96 && C != 192) || rows_per_sample % 64 != 0) return SCOT_ERR_UNSUPPORTED;
  if (!g || !z || !mean || !rstd || !gw_b || !W || !dz || !da || !d_gw_b || !d_bw_b) return SCOT_ERR_SHAPE;
  if ((gw_w == nullptr) != (d_gw_w == nullptr) || (d_gw_w == nullptr) != (d_bw_w == nullptr)) return SCOT_ERR_SHAPE;
  ProjClnBwdArgs p;
  p.W = (const bf16_t*)W; p.da = (bf16_t*)da; p.use_tr = g_scot_use_tr;
  p.b.g = g; p.b.z = z; p.b.z_dt = SCOT_F32; p.b.partial = nullptr; p.b.mean = mean; p.b.rstd = rstd; p.b.time = time; p.b.gw_w = gw_w; p.b.gw_b = gw_b; p.b.sscale = sample_scale;
  p.b.dz = (bf16_t*)dz; p.b.d_gw_w = d_gw_w; p.b.d_gw_b = d_gw_b; p.b.d_bw_w = d_bw_w; p.b.d_bw_b = d_bw_b; p.b.M = M;
  p.b.rows_per_sample = rows_per_sample;
  const int tt = C == 96 ? rows_tile_count(C, M, rows_per_sample) : 1;
  dim3 grid((M + 64 * tt - 1) / (64 * tt)), block(256);
  if (C == 96) {
    if (tt == 2) hipLaunchKernelGGL((proj_cln_bwd_fused_kernel<96, 2>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((proj_cln_bwd_fused_kernel<96, 1>), grid, block, 0, stream, p);
  } else {
    hipLaunchKernelGGL((proj_cln_bwd_fused_kernel<192, 1>), grid, block, 0, stream, p);
  }
  return scot_check_launch();
}


// Rows a workgroup of the block tail owns at (C, M, rows_per_sample) — the geometry both directions use; the backward's partial-sum
// scratch is one row of 4C (2C without conditioning) floats per workgroup and per norm.
static int tail_rows_per_wg(int C, int M, int rows_per_sample) {
  const int tt_env = scot_mlp_tt_override();
  int tt = tt_env ? tt_env : ((C == 96 && M >= 64 * 2 * 512) ? 2 : 1);
  if (C != 96 || rows_per_sample % (64 * tt) != 0) tt = 1;
  return 64 * tt;
}
extern "C" int scot_block_tail_workgroups(int M, int rows_per_sample, int C) {
  if (M <= 0 || rows_per_sample <= 0 || (C != 96 && C != 192 && C != 48)) return 0;
  const int r = tail_rows_per_wg(C, M, rows_per_sample);
  return (M + r - 1) / r;
}

// out[j] += Σ_b partial[b][j]: finishes the per-workgroup column sums of scot_block_tail_bwd / scot_cln_bwd mode 3 (norm_fast.hip)
int scot_cln_bwd_finish_launch(const float* partial, int nblk, int ncol, float* out, hipStream_t s);
extern "C" int scot_partial_colsum(const float* partial, int nblk, int ncol, float* out, hipStream_t stream) {
  if (!partial || !out || nblk <= 0 || ncol <= 0) return SCOT_ERR_SHAPE;
  return scot_cln_bwd_finish_launch(partial, nblk, ncol, out, stream);
}

// include/scot_hip.h: scot_block_tail_bwd = scot_mlp_block_bwd followed by scot_proj_cln_bwd on g_out, in one launch.
extern "C" int scot_block_tail_bwd(const float* g, float* g_out,
                                   /* MLP half */ const void* z2, const float* mean2, const float* rstd2, const float* gw_w2,
                                   const float* gw_b2, const float* sscale2, const void* dact, const void* W1, const void* W2, void* dz2,
                                   void* du, float* d_gw_w2, float* d_gw_b2, float* d_bw_w2, float* d_bw_b2,
                                   /* attention-output half */ const void* z1, const float* mean1, const float* rstd1,
                                   const float* gw_w1, const float* gw_b1, const float* sscale1, const void* Wo, void* dz1, void* da,
                                   float* d_gw_w1, float* d_gw_b1, float* d_bw_w1, float* d_bw_b1,
                                   /* optional prologue g += dqkv · Wqkv (both or neither; needs g_out == g) */ const void* dqkv,
                                   const void* Wqkv,
                                   /* recompute form (dact == NULL): u = h16·W1^T + b1 */ const void* h16, const float* b1,
                                   /* dtype of z1 / z2 */ int z_dt,
                                   /* optional per-workgroup column sums instead of atomics: [workgroups][4C | 2C] each */ float* partial2,
                                   float* partial1,
                                   const float* time, int M, int rows_per_sample, int C, int hid, hipStream_t stream) {
  if (M <= 0 || rows_per_sample <= 0) return SCOT_ERR_SHAPE;
  if (C != 96 && C != 192 && C != 48) return SCOT_ERR_UNSUPPORTED;
  // C = 48 (Poseidon-T / -S stage 0): the stored-gelu' form without the qkv prologue, one 192-wide hidden chunk
  if (C == 48 && (hid != 192 || dact == nullptr || dqkv != nullptr)) return SCOT_ERR_UNSUPPORTED;
  if (mlp_chunk(C) != 64 || hid < 64 || hid % 64 != 0 || rows_per_sample % 64 != 0) return SCOT_ERR_UNSUPPORTED;
  if ((dqkv == nullptr) != (Wqkv == nullptr)) return SCOT_ERR_SHAPE;
  if (dqkv && g_out != g) return SCOT_ERR_UNSUPPORTED;        // the prologue updates g in place
  if (z_dt != SCOT_F32 && z_dt != SCOT_BF16) return SCOT_ERR_DTYPE;
  if ((dact == nullptr) && (!h16 || !b1)) return SCOT_ERR_SHAPE;       // nothing to take gelu'(u) from
  if ((partial2 == nullptr) != (partial1 == nullptr)) return SCOT_ERR_SHAPE;
  const bool atomics = partial2 == nullptr;
  if (!g || !g_out || !z2 || !mean2 || !rstd2 || !gw_b2 || !W1 || !W2 || !dz2 || !z1 || !mean1 || !rstd1 || !gw_b1 || !Wo || !dz1 || !da)
    return SCOT_ERR_SHAPE;
  if (atomics && (!d_gw_b2 || !d_bw_b2 || !d_gw_b1 || !d_bw_b1)) return SCOT_ERR_SHAPE;
  if (atomics && ((gw_w2 == nullptr) != (d_gw_w2 == nullptr) || (d_gw_w2 == nullptr) != (d_bw_w2 == nullptr) ||
                  (gw_w1 == nullptr) != (d_gw_w1 == nullptr) || (d_gw_w1 == nullptr) != (d_bw_w1 == nullptr)))
    return SCOT_ERR_SHAPE;
  if ((gw_w1 == nullptr) != (gw_w2 == nullptr)) return SCOT_ERR_SHAPE;
  const int tt = tail_rows_per_wg(C, M, rows_per_sample) / 64;
  TailBwdArgs a;
  a.m.g = g; a.m.g_out = g_out; a.m.z = z2; a.m.z_dt = z_dt; a.m.mean = mean2; a.m.rstd = rstd2; a.m.time = time; a.m.gw_w = gw_w2; a.m.gw_b = gw_b2;
  a.m.sscale = sscale2; a.m.dact = (const bf16_t*)dact; a.m.W1 = (const bf16_t*)W1; a.m.W2 = (const bf16_t*)W2; a.m.dz = (bf16_t*)dz2;
  a.m.du = (bf16_t*)du; a.m.d_gw_w = d_gw_w2; a.m.d_gw_b = d_gw_b2; a.m.d_bw_w = d_bw_w2; a.m.d_bw_b = d_bw_b2; a.m.partial = partial2;
  a.m.h16 = (const bf16_t*)h16; a.m.b1 = b1;
  a.m.M = M; a.m.rows_per_sample = rows_per_sample; a.m.hid = hid; a.m.use_tr = g_scot_use_tr;
  a.pj.W = (const bf16_t*)Wo; a.pj.da = (bf16_t*)da; a.pj.use_tr = g_scot_use_tr;
  a.pj.b.g = g_out; a.pj.b.z = z1; a.pj.b.z_dt = z_dt; a.pj.b.mean = mean1; a.pj.b.rstd = rstd1; a.pj.b.time = time; a.pj.b.gw_w = gw_w1; a.pj.b.gw_b = gw_b1;
  a.pj.b.sscale = sscale1; a.pj.b.dz = (bf16_t*)dz1; a.pj.b.d_gw_w = d_gw_w1; a.pj.b.d_gw_b = d_gw_b1; a.pj.b.d_bw_w = d_bw_w1;
  a.pj.b.d_bw_b = d_bw_b1; a.pj.b.partial = partial1; a.pj.b.M = M; a.pj.b.rows_per_sample = rows_per_sample;
  a.dqkv = (const bf16_t*)dqkv; a.Wqkv = (const bf16_t*)Wqkv;
  if (C == 48) {
    hipLaunchKernelGGL((tail_bwd_fused_kernel<48, 192, 1, false, false>), dim3((M + 63) / 64), dim3(256), 0, stream, a);
    return scot_check_launch();
  }
  if (C == 96) return tt == 2 ? launch_tail_bwd<96, 64, 2>(a, stream) : launch_tail_bwd<96, 64, 1>(a, stream);
  return launch_tail_bwd<192, 64, 1>(a, stream);
}


// include/scot_hip.h: scot_block_tail_fwd = scot_proj_cln_fwd followed by scot_mlp_block_fwd on its output, in one launch.
extern "C" int scot_block_tail_fwd(/* attention-output half */ const void* a, const void* Wo, const float* bo, const float* x, float* h,
                                   void* h16, void* z1, float* mean1, float* rstd1, const float* gw_w1, const float* gw_b1,
                                   const float* bw_w1, const float* bw_b1, const float* sscale1,
                                   /* MLP half */ const void* W1, const float* b1, const void* W2, const float* b2, float* out,
                                   void* out16, void* act, void* dact, void* z2, float* mean2, float* rstd2, const float* gw_w2,
                                   const float* gw_b2, const float* bw_w2, const float* bw_b2, const float* sscale2,
                                   /* optional: the next layer's qkv = out16 · Wqkv^T + bqkv */ const void* Wqkv, const float* bqkv,
                                   void* qkv,
                                   /* dtype of z1 / z2: fp32, or the 16-bit operand format (only the backward's x-hat reads them) */ int z_dt,
                                   const float* time, int M, int rows_per_sample, int C, int hid, float eps, hipStream_t stream) {
  if (M <= 0 || rows_per_sample <= 0) return SCOT_ERR_SHAPE;
  if (C != 96 && C != 192 && C != 48) return SCOT_ERR_UNSUPPORTED;
  if (C == 48 && hid != 192) return SCOT_ERR_UNSUPPORTED;     // (one 192-wide hidden chunk: Poseidon-T / -S stage 0, mlp_ratio 4)
  if (mlp_chunk(C) != 64 || hid < 64 || hid % 64 != 0) return SCOT_ERR_UNSUPPORTED;
  if (z_dt != SCOT_F32 && z_dt != SCOT_BF16) return SCOT_ERR_DTYPE;
  if (!a || !Wo || !bo || !x || !h || !h16 || !gw_b1 || !bw_b1 || !W1 || !b1 || !W2 || !b2 || !out || !gw_b2 || !bw_b2) return SCOT_ERR_SHAPE;
  if ((act != nullptr && dact == nullptr) || (mean1 == nullptr) != (rstd1 == nullptr) || (mean2 == nullptr) != (rstd2 == nullptr) ||
      (gw_w1 == nullptr) != (bw_w1 == nullptr) || (gw_w2 == nullptr) != (bw_w2 == nullptr) || (gw_w1 == nullptr) != (gw_w2 == nullptr))
    return SCOT_ERR_SHAPE;
  TailFwdArgs t;
  t.pj.a = (const bf16_t*)a; t.pj.W = (const bf16_t*)Wo;
  t.pj.e.bias = bo; t.pj.e.z = z1; t.pj.e.z_dt = z_dt; t.pj.e.mean = mean1; t.pj.e.rstd = rstd1; t.pj.e.time = time; t.pj.e.gw_w = gw_w1; t.pj.e.gw_b = gw_b1;
  t.pj.e.bw_w = bw_w1; t.pj.e.bw_b = bw_b1; t.pj.e.sscale = sscale1; t.pj.e.resid = x; t.pj.e.out = h; t.pj.e.out16 = (bf16_t*)h16;
  t.pj.e.M = M; t.pj.e.rows_per_sample = rows_per_sample; t.pj.e.eps = eps;
  t.m.h16 = (const bf16_t*)h16; t.m.h = h; t.m.W1 = (const bf16_t*)W1; t.m.b1 = b1; t.m.W2 = (const bf16_t*)W2; t.m.b2 = b2;
  t.m.out = out; t.m.out16 = (bf16_t*)out16; t.m.act = (bf16_t*)act; t.m.dact = (bf16_t*)dact; t.m.z = z2; t.m.z_dt = z_dt; t.m.mean = mean2; t.m.rstd = rstd2;
  t.m.time = time; t.m.gw_w = gw_w2; t.m.gw_b = gw_b2; t.m.bw_w = bw_w2; t.m.bw_b = bw_b2; t.m.sscale = sscale2;
  t.m.M = M; t.m.rows_per_sample = rows_per_sample; t.m.hid = hid; t.m.eps = eps;
  if ((Wqkv == nullptr) != (qkv == nullptr)) return SCOT_ERR_SHAPE;
  t.Wqkv = (const bf16_t*)Wqkv; t.bqkv = bqkv; t.qkv = (bf16_t*)qkv;
  const int tt_env = scot_mlp_tt_override();
  const int tt = C == 96 ? (tt_env ? tt_env : (M >= 64 * 2 * 512 ? 2 : 1)) : 1;
  if (C == 48) return launch_tail_fwd<48, 192, 1>(t, stream);       // forward only: the backward of these layers stays layer by layer
  if (C == 96) return tt == 2 ? launch_tail_fwd<96, 64, 2>(t, stream) : launch_tail_fwd<96, 64, 1>(t, stream);
  return launch_tail_fwd<192, 64, 1>(t, stream);
}
