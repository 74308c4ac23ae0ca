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
// STATUS: written at the end of round 1 without GPU time left — compiled for gfx950 only.  OFF unless SCOT_FUSED_MLP=1; the
// parity tests for it (tests/test_kernels_gpu.py::test_mlp_block_fused, SCOT_EXPERIMENTAL=1) have not run yet.
#include "common.h"
#include <stdlib.h>

// prefetch registers: a NATIVE vector type — arrays of HIP's uint4 struct were left in scratch memory by the compiler here
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct MlpArgs {
  const bf16_t* h16; const float* h;
  const bf16_t* W1; const float* b1;
  const bf16_t* W2; const float* b2;
  float* out; bf16_t* out16;
  bf16_t* act; bf16_t* dact;
  float* z; float* mean; float* rstd;
  const float* time; const float* gw_w; const float* gw_b; const float* bw_w; const float* bw_b; const float* sscale;
  int M, rows_per_sample, hid;
  float eps;
};

template <int C, int HC, int TT>
__global__ __launch_bounds__(256, 2) void mlp_fused_kernel(MlpArgs p) {
  constexpr int KJ = C / 32;           // K-steps of GEMM 1 (K = C)
  constexpr int NT = C / 16;           // channel tiles of GEMM 2 / of the output
  constexpr int NB = HC / 32;          // 32-hidden blocks per chunk
  constexpr int P1 = C + 8;            // pitch of the W1 chunk  [HC][P1]  (bf16 elements)
  constexpr int P2 = HC + 8;           // pitch of the W2 chunk  [C][P2]
  constexpr int CP = C + 4;            // pitch of the epilogue patch (floats); CP % 16 == 4: the 4 row groups hit disjoint banks
  constexpr int W1_EL = HC * P1, W2_EL = C * P2;
  constexpr int N1 = HC * C / 8, N2 = C * HC / 8;          // 16-byte pieces per chunk
  constexpr int PW1 = (N1 + 255) / 256, PW2 = (N2 + 255) / 256;
  constexpr size_t WBYTES = (size_t)(W1_EL + W2_EL) * 2 + (size_t)HC * 4;
  constexpr size_t PBYTES = (size_t)4 * 16 * CP * 4;
  constexpr size_t LDS_BYTES = WBYTES > PBYTES ? WBYTES : PBYTES;
  static_assert(C % 32 == 0 && HC % 32 == 0 && (W1_EL * 2) % 16 == 0 && ((W1_EL + W2_EL) * 2) % 16 == 0, "layout");
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
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
    const int row = min(row0 + tt * 16 + lc, p.M - 1);
    const bf16_t* src = p.h16 + (size_t)row * C + g * 8;
#pragma unroll
    for (int j = 0; j < KJ; ++j) hf[tt][j].v = *(const s16x8_t*)(src + j * 32);
  }

  // ---- weight chunk: global -> registers (unconditional, clamped piece index) -> LDS
  u32x4_t r1[PW1], r2[PW2];
  auto load_chunk = [&](int c) {
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
  auto store_chunk = [&](int c) {
#pragma unroll
    for (int u = 0; u < PW1; ++u) {
      const int i = tid + u * 256;
      if (i < N1) {
        const int x = i / (C / 8), k8 = (i % (C / 8)) * 8;   // x: hidden unit within the chunk
        const int y = x & 31;
        const int rho = (x & ~31) + (((y >> 2) & 1) << 4) + ((y >> 3) << 2) + (y & 3);   // [blk][t][a][b] of y = 8a + 4t + b
        *(u32x4_t*)(W1c + rho * P1 + k8) = r1[u];
      }
    }
#pragma unroll
    for (int u = 0; u < PW2; ++u) {
      const int i = tid + u * 256;
      if (i < N2) {
        const int row = i / (HC / 8), c8 = (i % (HC / 8)) * 8;
        *(u32x4_t*)(W2c + row * P2 + c8) = r2[u];
      }
    }
    if (tid < HC) b1c[tid] = p.b1[c * HC + tid];
  };

  f32x4_t Y[TT][NT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) Y[tt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int c = 0; c < nch; ++c) {
    load_chunk(min(c + 1, nch - 1));          // in flight during the multiply (the last iteration re-reads its own chunk: L2 hit)
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
        if (p.act) {
          const int row = row0 + tt * 16 + lc;
          if (row < p.M) {
            const size_t o = (size_t)row * HID + (size_t)c * HC + blk * 32 + g * 8;
            *(s16x8_t*)(p.act + o) = af[tt].v;
            *(s16x8_t*)(p.dact + o) = frag_from_f32<bf16_t>(dv).v;
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
    __syncthreads();                           // every wave is done reading chunk c
    if (c + 1 < nch) {
      store_chunk(c + 1);
      __syncthreads();
    }
  }

  // ---- epilogue: + b2, layer norm over the row, conditional affine, DropPath scale, residual.  The weight chunk is dead
  // (barrier above): each wave stages one 16-row tile at a time in its own patch and re-reads it row-contiguously.
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
    const bool valid = grow < p.M;
    float v[KJ][8];
    float s1 = 0.f;
#pragma unroll
    for (int pp = 0; pp < KJ; ++pp) {
      const int col = pp * 32 + q * 8;
      const float4 x0 = *(const float4*)(Ct + prow * CP + col), x1 = *(const float4*)(Ct + prow * CP + col + 4);
      float bb[8];
      ld8(p.b2, SCOT_F32, col, bb);
      v[pp][0] = x0.x + bb[0]; v[pp][1] = x0.y + bb[1]; v[pp][2] = x0.z + bb[2]; v[pp][3] = x0.w + bb[3];
      v[pp][4] = x1.x + bb[4]; v[pp][5] = x1.y + bb[5]; v[pp][6] = x1.z + bb[6]; v[pp][7] = x1.w + bb[7];
#pragma unroll
      for (int j = 0; j < 8; ++j) s1 += v[pp][j];
    }
    s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
    const float mean = s1 * (1.0f / C);
    float s2 = 0.f;
#pragma unroll
    for (int pp = 0; pp < KJ; ++pp)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[pp][j] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
    const float rstd = 1.0f / sqrtf(s2 * (1.0f / C) + p.eps);
    if (valid) {
      const size_t base = (size_t)grow * C;
      if (p.mean && q == 0) { p.mean[grow] = mean; p.rstd[grow] = rstd; }
      const int samp = grow / p.rows_per_sample;
      const float t = p.time ? p.time[samp] : 0.f;
      const float sc = p.sscale ? p.sscale[samp] : 1.f;
#pragma unroll
      for (int pp = 0; pp < KJ; ++pp) {
        const int col = pp * 32 + q * 8;
        if (p.z) st8(p.z, SCOT_F32, base + col, v[pp]);
        float gw[8], gb[8], bw[8], bbv[8], res[8], o[8];
        ld8(p.gw_b, SCOT_F32, col, gb); ld8(p.bw_b, SCOT_F32, col, bbv);
        if (p.gw_w) { ld8(p.gw_w, SCOT_F32, col, gw); ld8(p.bw_w, SCOT_F32, col, bw); }
        ld8(p.h, SCOT_F32, base + col, res);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float ga = p.gw_w ? gw[j] * t + gb[j] : gb[j];
          const float be = p.gw_w ? bw[j] * t + bbv[j] : bbv[j];
          o[j] = sc * (ga * ((v[pp][j] - mean) * rstd) + be) + res[j];
        }
        st8(p.out, SCOT_F32, base + col, o);
        if (p.out16) st8(p.out16, SCOT_BF16, base + col, o);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int C, int HC, int TT>
static int launch_mlp(const MlpArgs& a, hipStream_t s) {
  const int rows_per_wg = 64 * TT;
  dim3 grid((a.M + rows_per_wg - 1) / rows_per_wg), block(256);
  hipLaunchKernelGGL((mlp_fused_kernel<C, HC, TT>), grid, block, 0, s, a);
  return scot_check_launch();
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
  const int hc = C == 96 ? 96 : 64;          // hidden units per LDS chunk: 40 KB (C = 96) / 54 KB (C = 192) of weights per workgroup
  if (hid < hc || hid % hc != 0) return SCOT_ERR_UNSUPPORTED;
  if (!h16 || !h || !W1 || !b1 || !W2 || !b2 || !out || !gw_b || !bw_b) return SCOT_ERR_SHAPE;
  if ((act == nullptr) != (dact == nullptr) || (mean == nullptr) != (rstd == nullptr) || (gw_w == nullptr) != (bw_w == nullptr))
    return SCOT_ERR_SHAPE;
  MlpArgs a;
  a.h16 = (const bf16_t*)h16; a.h = h; a.W1 = (const bf16_t*)W1; a.b1 = b1; a.W2 = (const bf16_t*)W2; a.b2 = b2;
  a.out = out; a.out16 = (bf16_t*)out16; a.act = (bf16_t*)act; a.dact = (bf16_t*)dact; a.z = z; a.mean = mean; a.rstd = rstd;
  a.time = time; a.gw_w = gw_w; a.gw_b = gw_b; a.bw_w = bw_w; a.bw_b = bw_b; a.sscale = sample_scale;
  a.M = M; a.rows_per_sample = rows_per_sample; a.hid = hid; a.eps = eps;
  static int tt_env = -1;
  if (tt_env < 0) { const char* e = getenv("SCOT_MLP_TT"); tt_env = e ? atoi(e) : 0; }
  // 64·TT rows per workgroup: TT = 2 halves the LDS weight reads per MFMA; TT = 1 when that would leave CUs without work
  // (C = 192 with TT = 2 needs 256 VGPRs + spills: TT = 1 unless forced)
  const int tt = tt_env ? tt_env : ((C == 96 && M >= 64 * 2 * 512) ? 2 : 1);
  if (C == 96) return tt == 2 ? launch_mlp<96, 96, 2>(a, stream) : launch_mlp<96, 96, 1>(a, stream);
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
  const float* z; const float* mean; const float* rstd;
  const float* time; const float* gw_w; const float* gw_b; const float* sscale;
  const bf16_t* dact; const bf16_t* W1; const bf16_t* W2;
  bf16_t* dz; bf16_t* du;
  float* d_gw_w; float* d_gw_b; float* d_bw_w; float* d_bw_b;
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

template <int C, int HC, int TT>
__global__ __launch_bounds__(256, 2) void mlp_bwd_fused_kernel(MlpBwdArgs p) {
  constexpr int KJ = C / 32, NT = C / 16, NB = HC / 32;
  constexpr int P1 = C + 8;            // W1 chunk [HC][P1]: k = hidden (rows), columns = channels
  constexpr int P2 = HC + 8;           // W2 chunk [C][P2]:  k = channels (rows), columns = hidden
  constexpr int PD = C + 8;            // dz patch [16][PD] bf16, K-contiguous
  constexpr int CP = C + 4;            // epilogue patch pitch (floats)
  constexpr int W1_EL = HC * P1, W2_EL = C * P2;
  constexpr int N1 = HC * C / 8, N2 = C * HC / 8;
  constexpr int PW1 = (N1 + 255) / 256, PW2 = (N2 + 255) / 256;
  constexpr size_t WBYTES = (size_t)(W1_EL + W2_EL) * 2;
  constexpr size_t PBYTES = (size_t)4 * 16 * CP * 4;
  constexpr size_t RBYTES = (size_t)4 * 2 * C * 4;
  constexpr size_t DZ_BYTES = (size_t)4 * 16 * PD * 2;
  // one region, three lives: [dz patches | column sums] (phase 1)  ->  weight chunks (phase 2)  ->  fp32 patches (phase 3)
  constexpr size_t P1BYTES = DZ_BYTES + RBYTES;
  constexpr size_t LDS_BYTES = WBYTES > PBYTES ? (WBYTES > P1BYTES ? WBYTES : P1BYTES) : (PBYTES > P1BYTES ? PBYTES : P1BYTES);
  static_assert(DZ_BYTES % 16 == 0 && (W1_EL * 2) % 16 == 0, "layout");
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  bf16_t* W1c = (bf16_t*)smem;
  bf16_t* W2c = W1c + W1_EL;
  float* red = (float*)(smem + DZ_BYTES);                      // [wave][2][C]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lc = lane & 15;
  bf16_t* Dz = (bf16_t*)smem + wave * 16 * PD;
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
      const int i = tid + u * 256;
      if (i < N1) *(u32x4_t*)(W1c + (i / (C / 8)) * P1 + (i % (C / 8)) * 8) = r1[u];
    }
#pragma unroll
    for (int u = 0; u < PW2; ++u) {
      const int i = tid + u * 256;
      if (i < N2) *(u32x4_t*)(W2c + (i / (HC / 8)) * P2 + (i % (HC / 8)) * 8) = r2[u];
    }
  };

  // ---- phase 1: dz = CLN_bwd(s·g) in the row-contiguous layout (4 lanes per row), parameter-gradient column sums
  const int samp = min(wg_row0, p.M - 1) / p.rows_per_sample;  // uniform over the workgroup (host-checked)
  const float t = p.time ? p.time[samp] : 0.f;
  const float sc = p.sscale ? p.sscale[samp] : 1.f;
  float ag[KJ][8], ab[KJ][8];
#pragma unroll
  for (int pp = 0; pp < KJ; ++pp)
#pragma unroll
    for (int j = 0; j < 8; ++j) { ag[pp][j] = 0.f; ab[pp][j] = 0.f; }
  // gamma = gw_w·t + gw_b for 8 columns: re-read (L1/L2) where needed rather than held in 8·KJ registers
  auto gamma8 = [&](int col, float (&ga)[8]) {
    float gb[8], gw[8];
    ld8(p.gw_b, SCOT_F32, col, gb);
    if (p.gw_w) ld8(p.gw_w, SCOT_F32, col, gw);
#pragma unroll
    for (int j = 0; j < 8; ++j) ga[j] = p.gw_w ? gw[j] * t + gb[j] : gb[j];
  };
  Frag<bf16_t> dzf[TT][KJ];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const int grow = row0 + tt * 16 + prow;
    const bool valid = grow < p.M;
    const int rowc = valid ? grow : p.M - 1;
    const size_t base = (size_t)rowc * C;
    const float mean = p.mean[rowc], rstd = p.rstd[rowc];
    float d[KJ][8], xh[KJ][8];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int pp = 0; pp < KJ; ++pp) {
      const int col = pp * 32 + q * 8;
      float zz[8], ga[8];
      ld8(p.g, SCOT_F32, base + col, d[pp]);
      ld8(p.z, SCOT_F32, base + col, zz);
      gamma8(col, ga);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dd = valid ? d[pp][j] * sc : 0.f;
        xh[pp][j] = (zz[j] - mean) * rstd;
        ag[pp][j] += dd * xh[pp][j];
        ab[pp][j] += dd;
        d[pp][j] = dd * ga[j];                 // from here on: dout·gamma
        m1 += d[pp][j]; m2 += d[pp][j] * xh[pp][j];
      }
    }
    m1 += __shfl_xor(m1, 1, 64); m1 += __shfl_xor(m1, 2, 64);
    m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64);
    m1 *= 1.0f / C; m2 *= 1.0f / C;
#pragma unroll
    for (int pp = 0; pp < KJ; ++pp) {
      const int col = pp * 32 + q * 8;
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rstd * (d[pp][j] - m1 - xh[pp][j] * m2);
      if (valid) st8(p.dz, SCOT_BF16, base + col, o);
      store8_ct(Dz + prow * PD + col, o);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < KJ; ++j) dzf[tt][j] = lds_frag_kc(Dz, PD, 0, j * 32, lane);   // column = token lc, k = channel
    __builtin_amdgcn_wave_barrier();
  }
  // column sums over the wave's rows (the 16 rows of a pass live in lanes q, q+4, ...), then over the four waves via LDS
#pragma unroll
  for (int pp = 0; pp < KJ; ++pp)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int o = 4; o < 64; o <<= 1) { ag[pp][j] += __shfl_xor(ag[pp][j], o, 64); ab[pp][j] += __shfl_xor(ab[pp][j], o, 64); }
    }
  if (lane < 4) {
#pragma unroll
    for (int pp = 0; pp < KJ; ++pp)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wave * 2 + 0) * C + pp * 32 + q * 8 + j] = ag[pp][j];
        red[(wave * 2 + 1) * C + pp * 32 + q * 8 + j] = ab[pp][j];
      }
  }
  __syncthreads();
  if (tid < C) {
    float dg = 0.f, db = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { dg += red[(w * 2 + 0) * C + tid]; db += red[(w * 2 + 1) * C + tid]; }
    if (p.d_gw_w) { atomicAdd(&p.d_gw_w[tid], t * dg); atomicAdd(&p.d_bw_w[tid], t * db); }
    atomicAdd(&p.d_gw_b[tid], dg);
    atomicAdd(&p.d_bw_b[tid], db);
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

  for (int c = 0; c < nch; ++c) {
    load_chunk(min(c + 1, nch - 1));
    // gelu'(u) of this chunk for the lane's token and its 8 hidden units per block: in flight during the first MFMAs
    s16x8_t gpv[TT][NB];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
        gpv[tt][blk] = *(const s16x8_t*)(p.dact + (size_t)min(rowt[tt], p.M - 1) * HID + (size_t)c * HC + blk * 32 + g * 8);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
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
          for (int r = 0; r < 4; ++r) dv[4 * ts + r] = U[tt][ts][r] * bf2f((bf16_t)gpv[tt][blk][4 * ts + r]);
        af[tt] = frag_from_f32<bf16_t>(dv);
        if (rowt[tt] < p.M) *(s16x8_t*)(p.du + (size_t)rowt[tt] * HID + (size_t)c * HC + blk * 32 + g * 8) = af[tt].v;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const Frag<bf16_t> w = lds_frag_ks(W1c, P1, nt * 16, blk * 32 + g * 8, blk * 32 + g * 8 + 4, lane, p.use_tr);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) mma16(Y[tt][nt], af[tt], w);
      }
    }
    __syncthreads();
    if (c + 1 < nch) {
      store_chunk();
      __syncthreads();
    }
  }

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
        const float4 x0 = *(const float4*)(Ct + prow * CP + col), x1 = *(const float4*)(Ct + prow * CP + col + 4);
        float gi[8], o[8];
        ld8(p.g, SCOT_F32, base + col, gi);
        o[0] = gi[0] + x0.x; o[1] = gi[1] + x0.y; o[2] = gi[2] + x0.z; o[3] = gi[3] + x0.w;
        o[4] = gi[4] + x1.x; o[5] = gi[5] + x1.y; o[6] = gi[6] + x1.z; o[7] = gi[7] + x1.w;
        st8(p.g_out, SCOT_F32, base + col, o);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
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
  const int hc = C == 96 ? 96 : 64;
  if (hid < hc || hid % hc != 0) return SCOT_ERR_UNSUPPORTED;
  if (!g || !g_out || !z || !mean || !rstd || !gw_b || !dact || !W1 || !W2 || !dz || !du || !d_gw_b || !d_bw_b) return SCOT_ERR_SHAPE;
  if ((gw_w == nullptr) != (d_gw_w == nullptr) || (d_gw_w == nullptr) != (d_bw_w == nullptr)) return SCOT_ERR_SHAPE;
  static int tt_env = -1;
  if (tt_env < 0) { const char* e = getenv("SCOT_MLP_TT"); tt_env = e ? atoi(e) : 0; }
  int tt = tt_env ? tt_env : ((C == 96 && M >= 64 * 2 * 512) ? 2 : 1);
  if (rows_per_sample % (64 * tt) != 0) tt = 1;
  if (rows_per_sample % 64 != 0) return SCOT_ERR_UNSUPPORTED;      // the conditioning time must be uniform per workgroup
  MlpBwdArgs a;
  a.g = g; a.g_out = g_out; a.z = z; a.mean = mean; a.rstd = rstd; a.time = time; a.gw_w = gw_w; a.gw_b = gw_b; a.sscale = sample_scale;
  a.dact = (const bf16_t*)dact; a.W1 = (const bf16_t*)W1; a.W2 = (const bf16_t*)W2; a.dz = (bf16_t*)dz; a.du = (bf16_t*)du;
  a.d_gw_w = d_gw_w; a.d_gw_b = d_gw_b; a.d_bw_w = d_bw_w; a.d_bw_b = d_bw_b;
  a.M = M; a.rows_per_sample = rows_per_sample; a.hid = hid; a.use_tr = g_scot_use_tr;
  if (C == 96) return tt == 2 ? launch_mlp_bwd<96, 96, 2>(a, stream) : launch_mlp_bwd<96, 96, 1>(a, stream);
  return tt == 2 ? launch_mlp_bwd<192, 64, 2>(a, stream) : launch_mlp_bwd<192, 64, 1>(a, stream);
}
