// tail_deep — the tail of a ScOTLayer at the DEEP stages (C = 384 / 768: 4096 / 1024 token rows at batch 64) in one launch per
// direction (reference model.py:560-579, HF modeling_swinv2.py:396-410, 478-489, 533-561):
//
//     h   = x + s1 · CLN1(attn · Wo^T + bo)
//     out = h + s2 · CLN2(gelu(h16 · W1^T + b1) · W2^T + b2)        (+ the NEXT layer's qkv = out16 · Wqkv^T + bqkv)
//
// The layer-by-layer path runs seven launches here (qkv, attention, projection, norm, fc1, fc2, norm); each is a 5-30 us kernel on a
// chip that is half idle: with 4096 / 1024 rows there is one 64x64 GEMM tile per CU or less, and the K loop of such a tile is bound by
// what ONE workgroup can keep in flight.  The block-tail kernels of the token-heavy stages (mlp_fused.hip: 64 rows per workgroup,
// weight chunks staged through LDS) do not carry over either: 4096 rows are 64 such workgroups.
//
// Here a workgroup owns SIXTEEN rows (one MFMA tile: 4096 rows = 256 workgroups = one per CU) and therefore uses every weight element
// exactly once.  So the weights never touch LDS: they stream from L2 straight into MFMA operand fragments, from a FRAGMENT-ORDERED copy
// of each matrix (scot_fragpack: the 64 lanes' 16-byte operands of one 16x32 tile are 1 KiB contiguous).  Measured with
// tools/probes/l2_stream_probe.hip: a CU ingests 91-107 GB/s that way (every CU reading the same 3.5 MB: 35 us) against 36 GB/s
// when the lanes read their 16 bytes from the row-major matrix (adjacent lanes = different rows = one tag lookup per lane).
// The four waves split the OUTPUT columns of each product; the 16 x C activations live in LDS (operand tile) and, for the norms, in
// the "row layout" (16 lanes per row, 8 consecutive columns per lane and 128-column chunk) where a row's statistics are four xor-shuffles
// away and every global access is a 512-byte row segment.  The residual h never leaves the registers of the lanes that normalise it.
//
// C = 768 (1024 rows = 64 row blocks): the hidden dimension is split over `hsplit` workgroups per row block (each recomputes the
// cheap projection + norm, takes hid/hsplit hidden units and writes its partial fc2 sums to ypart); scot_deep_tail_finish adds the
// partial sums and applies the second norm.
#include "common.h"

// ------------------------------------------------------------------------------------------------------------------------
// Fragment-ordered weight copies.  For a matrix W [N][K] (K contiguous) used as the operand whose lane (r, g) owns 8 consecutive k of
// row r:   Wf[((nt·K/32 + ks)·64 + lane)·8 + j] = W[perm(16 nt + (lane & 15))][32 ks + 8 (lane >> 4) + j]
// mode bit 0: the source is stored transposed ([K][N] row-major: the fragments of W^T); bit 1: rows permuted inside every 32-row block,
// fragment row 16 t + 4 a + b <- source row 8 a + 4 t + b (so that the accumulator lanes of the transposed product U^T = W1 · h^T hold 8
// CONSECUTIVE hidden units of a token, see mlp_fused.hip).  desc: int32 [n][6] = {source offset, N, K, first block, mode, dest offset};
// one block = 256 lanes x 16 bytes.
__global__ __launch_bounds__(256) void fragpack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf, const int* __restrict__ desc, int n) {
  int lo = 0, hi = n - 1;
  const int blk = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid * 6 + 3] <= blk) lo = mid; else hi = mid - 1;
  }
  const int* d = desc + lo * 6;
  const int N = d[1], K = d[2], mode = d[4], KS = K / 32;
  const size_t piece = (size_t)(blk - d[3]) * 256 + threadIdx.x;       // 16-byte piece index = (nt·KS + ks)·64 + lane
  if (piece >= (size_t)N * K / 8) return;
  const int lane = (int)(piece & 63);
  const int tile = (int)(piece >> 6), nt = tile / KS, ks = tile - nt * KS;
  int row = nt * 16 + (lane & 15);
  if (mode & 2) {
    const int y = row & 31, t = y >> 4, rho = y & 15;
    row = (row & ~31) + 8 * (rho >> 2) + 4 * t + (rho & 3);
  }
  const int k0 = ks * 32 + (lane >> 4) * 8;
  const float* src = w + d[0];
  float v[8];
  if (mode & 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k0 + j) * N + row];
  } else {
    const float4 a = *(const float4*)(src + (size_t)row * K + k0), b = *(const float4*)(src + (size_t)row * K + k0 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  store8_ct(wf + d[5] + piece * 8, v);
}
extern "C" int scot_fragpack(const float* w, void* wf16, const int* desc, int n, int blocks, hipStream_t s) {
  if (n <= 0 || blocks <= 0) return SCOT_ERR_SHAPE;
  hipLaunchKernelGGL(fragpack_kernel, dim3(blocks), dim3(256), 0, s, w, (bf16_t*)wf16, desc, n);
  return scot_check_launch();
}

// one lane's 16-byte operand of tile (nt, ks) of a fragment-ordered matrix with KS k-steps per row tile
__device__ __forceinline__ Frag<bf16_t> ldw(const bf16_t* __restrict__ Wf, int KS, int nt, int ks, int lane) {
  Frag<bf16_t> f;
  f.v = *(const s16x8_t*)(Wf + ((size_t)(nt * KS + ks) * 64 + lane) * 8);
  return f;
}

struct DeepFwdArgs {
  const bf16_t* a; const bf16_t* Wo; const float* bo; const float* x; float* h; bf16_t* h16; void* z1; float* mean1; float* rstd1;
  const float* gw_w1; const float* gw_b1; const float* bw_w1; const float* bw_b1; const float* ss1;
  const bf16_t* W1; const float* b1; const bf16_t* W2; const float* b2; float* out; bf16_t* out16; bf16_t* act; bf16_t* dact;
  void* z2; float* mean2; float* rstd2;
  const float* gw_w2; const float* gw_b2; const float* bw_w2; const float* bw_b2; const float* ss2;
  const bf16_t* Wqkv; const float* bqkv; bf16_t* qkv;
  int z_dt; const float* time; int M, rows_per_sample, hid; float eps;
  int hsplit; float* ypart;
};

// acc[j] (+)= tile[16][K = C] · Wf[n-tiles nt0 + j][K]^T for j < NTW: the wave's NTW output tiles, weights D k-steps ahead in registers.
// `presync`: a __syncthreads() between the first weight loads and the first read of `tile` (the tile was written just before).
template <int C, int NTW, int D>
__device__ __forceinline__ void rows_gemm(f32x4_t (&acc)[NTW], const bf16_t* __restrict__ Wf, int nt0, const bf16_t* tile, int lane, bool presync) {
  constexpr int KS = C / 32, PA = C + 8;
  Frag<bf16_t> wb[D][NTW];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int j = 0; j < NTW; ++j) wb[d][j] = ldw(Wf, KS, nt0 + j, d, lane);
  if (presync) __syncthreads();
  Frag<bf16_t> a[KS];                      // the 16 rows as A operands, all k-steps at once: no LDS round trip between the MFMAs
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = lds_frag_kc(tile, PA, 0, ks * 32, lane);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) mma16(acc[j], a[ks], wb[ks % D][j]);
    if (ks + D < KS) {
#pragma unroll
      for (int j = 0; j < NTW; ++j) wb[ks % D][j] = ldw(Wf, KS, nt0 + j, ks + D, lane);
    }
  }
}

// the wave's NTW accumulator tiles (C/D layout: row 4 g + r, column 16 (nt0 + j) + lc) -> the fp32 patch [16][C + 4]
template <int C, int NTW>
__device__ __forceinline__ void acc_to_patch(const f32x4_t (&acc)[NTW], float* patch, int nt0, int lane) {
  constexpr int CP = C + 4;
  const int g = lane >> 4, lc = lane & 15;
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) patch[(g * 4 + r) * CP + (nt0 + j) * 16 + lc] = acc[j][r];
}

struct RowNorm {            // one (conditional) layer norm of the row layout
  const float* bias; const float* gw_w; const float* gw_b; const float* bw_w; const float* bw_b;
  float t, sc, eps;
};

// Row layout: thread tid -> row tid >> 4, lane-in-row q = tid & 15; chunk pp holds columns 128 pp + 8 q .. + 7.
// v <- patch + bias; returns mean / rstd of the row (two passes, as the stand-alone norm kernels do)
template <int C>
__device__ __forceinline__ void row_stats(const float* patch, const float* bias, float eps, float (&v)[C / 128][8], float& mean, float& rstd) {
  constexpr int RP = C / 128, CP = C + 4;
  const int rrow = threadIdx.x >> 4, q = threadIdx.x & 15;
  float s1 = 0.f;
#pragma unroll
  for (int pp = 0; pp < RP; ++pp) {
    const int col = pp * 128 + q * 8;
    const float4 x0 = *(const float4*)(patch + rrow * CP + col), x1 = *(const float4*)(patch + rrow * CP + col + 4);
    float bb[8];
    ld8(bias, SCOT_F32, col, bb);
    v[pp][0] = x0.x + bb[0]; v[pp][1] = x0.y + bb[1]; v[pp][2] = x0.z + bb[2]; v[pp][3] = x0.w + bb[3];
    v[pp][4] = x1.x + bb[4]; v[pp][5] = x1.y + bb[5]; v[pp][6] = x1.z + bb[6]; v[pp][7] = x1.w + bb[7];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1 += v[pp][j];
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s1 += __shfl_xor(s1, o, 64);
  mean = s1 * (1.0f / C);
  float s2 = 0.f;
#pragma unroll
  for (int pp = 0; pp < RP; ++pp)
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = v[pp][j] - mean; s2 += d * d; }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s2 += __shfl_xor(s2, o, 64);
  rstd = 1.0f / sqrtf(s2 * (1.0f / C) + eps);
}

// o = res + sc · (gamma ⊙ (v - mean) · rstd + beta) for the 8 columns at `col`
__device__ __forceinline__ void row_affine(const RowNorm& n, int col, const float (&v)[8], float mean, float rstd, const float (&res)[8], float (&o)[8]) {
  float gw[8], gb[8], bw[8], bbv[8];
  ld8(n.gw_b, SCOT_F32, col, gb); ld8(n.bw_b, SCOT_F32, col, bbv);
  if (n.gw_w) { ld8(n.gw_w, SCOT_F32, col, gw); ld8(n.bw_w, SCOT_F32, col, bw); }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float ga = n.gw_w ? gw[j] * n.t + gb[j] : gb[j];
    const float be = n.gw_w ? bw[j] * n.t + bbv[j] : bbv[j];
    o[j] = n.sc * (ga * ((v[j] - mean) * rstd) + be) + res[j];
  }
}

template <int C>
__global__ __launch_bounds__(256, 1) void deep_tail_fwd_kernel(DeepFwdArgs p) {
  constexpr int NT = C / 16, NTW = NT / 4, KS = C / 32, PA = C + 8, CP = C + 4, RP = C / 128, GP = 128 + 8;
  constexpr int DPROJ = NTW >= 12 ? 2 : 4;              // k-steps of weights in flight in the plain row GEMMs (24 loads per wave)
  constexpr int F1 = KS / 12, F2 = NTW / 6;             // sub-stages (24 weight fragments each) of fc1 / fc2 per hidden chunk
  static_assert(C % 128 == 0 && KS % 12 == 0 && NTW % 6 == 0 && (F1 + F2) % 2 == 0, "C = 384 or 768");
  __shared__ __attribute__((aligned(16))) bf16_t tileA[16 * PA];        // the 16 rows as MFMA operand: attn -> h16 -> out16
  constexpr int QPF = (16 * (3 * C + 8) * 2 + 3) / 4;                     // floats covering the 16-bit qkv rows [16][3C + 8]
  __shared__ __attribute__((aligned(16))) float qpatch[QPF > 16 * CP ? QPF : 16 * CP];
  float* patch = qpatch;                                               // accumulators on their way to the row layout [16][C + 4]
  __shared__ __attribute__((aligned(16))) bf16_t gx[2][16 * GP];        // gelu(u) of one 128-hidden chunk, all waves' blocks
  __shared__ __attribute__((aligned(16))) float b1s[4 * C];             // this workgroup's slice of b1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lc = lane & 15;
  const int hs = p.hsplit, rb = blockIdx.x / hs, hq = blockIdx.x - rb * hs;
  const bool lead = hq == 0;
  const int row0 = rb * 16;
  const int rrow = tid >> 4, q = tid & 15, grow = row0 + rrow;
  const int samp = grow / p.rows_per_sample;
  const int HID = p.hid, HL = HID / hs, hb0 = hq * HL, nch = HL / 128;
  const int nt0 = wave * NTW;

  // ---- phase 0: the attention output rows -> operand tile; this workgroup's b1 slice
  for (int i = tid; i < 16 * C / 8; i += 256) {
    const int r = i / (C / 8), c8 = (i % (C / 8)) * 8;
    *(uint4*)(tileA + r * PA + c8) = *(const uint4*)(p.a + (size_t)(row0 + r) * C + c8);
  }
  for (int i = tid; i < HL; i += 256) b1s[i] = p.b1[hb0 + i];

  // ---- phase 1: z1 = attn · Wo^T (+ bo in the row phase)
  {
    f32x4_t acc[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    rows_gemm<C, NTW, DPROJ>(acc, p.Wo, nt0, tileA, lane, true);
    acc_to_patch<C, NTW>(acc, patch, nt0, lane);
  }
  __syncthreads();

  // ---- phase 2: the MLP over this workgroup's hidden units, 128 per chunk (32 per wave).  Weight fragments: two register sets of
  // 24, each refilled as soon as its MFMAs are issued — a chunk's W1 fragments arrive while the previous chunk's fc2 runs, its W2
  // fragments while its fc1 runs.
  f32x4_t Y[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) Y[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  Frag<bf16_t> S[2][24];
  const int KS2 = HID / 32;
  auto issue = [&](Frag<bf16_t> (&s)[24], int u, int c) {        // sub-stage u of chunk c
    if (u < F1) {
      const int tile0 = ((hb0 + c * 128 + wave * 32) >> 5) * 2;
#pragma unroll
      for (int i = 0; i < 24; ++i) s[i] = ldw(p.W1, KS, tile0 + (i & 1), u * 12 + (i >> 1), lane);
    } else {
      const int ks0 = (hb0 + c * 128) >> 5;
#pragma unroll
      for (int i = 0; i < 24; ++i) s[i] = ldw(p.W2, KS2, nt0 + (u - F1) * 6 + (i >> 2), ks0 + (i & 3), lane);
    }
  };
  issue(S[0], 0, 0);                             // (in flight during the row phase, and AHEAD of its stores in the wave's memory queue)
  issue(S[1], 1, 0);

  // ---- row phase A: h = x + s1 · CLN1(z1); h16 -> operand tile; h stays in registers for the second residual
  float hres[RP][8];
  {
    float v[RP][8], mean, rstd;
    row_stats<C>(patch, p.bo, p.eps, v, mean, rstd);
    RowNorm n{p.bo, p.gw_w1, p.gw_b1, p.bw_w1, p.bw_b1, p.time ? p.time[samp] : 0.f, p.ss1 ? p.ss1[samp] : 1.f, p.eps};
    const size_t base = (size_t)grow * C;
    if (lead && p.mean1 && q == 0) { p.mean1[grow] = mean; p.rstd1[grow] = rstd; }
#pragma unroll
    for (int pp = 0; pp < RP; ++pp) {
      const int col = pp * 128 + q * 8;
      if (lead && p.z1) st8(p.z1, p.z_dt, base + col, v[pp]);
      float res[8];
      ld8(p.x, SCOT_F32, base + col, res);
      row_affine(n, col, v[pp], mean, rstd, res, hres[pp]);
      if (lead) {
        if (p.h) st8(p.h, SCOT_F32, base + col, hres[pp]);
        if (p.h16) st8(p.h16, SCOT_BF16, base + col, hres[pp]);
      }
      store8_ct(tileA + rrow * PA + col, hres[pp]);
    }
  }

  __syncthreads();                               // the h16 tile is complete
  Frag<bf16_t> hb[KS];                           // ... and is the B operand of every chunk's fc1: read once
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) hb[ks] = lds_frag_kc(tileA, PA, 0, ks * 32, lane);
  for (int c = 0; c < nch; ++c) {
    // (the prefetches below are UNCONDITIONAL — the last chunk re-loads itself: a load inside a branch makes the compiler wait, at
    // the join, as if it had not been issued, and every chunk would drain the pipeline)
    const int cn = min(c + 1, nch - 1);
    f32x4_t U[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int u = 0; u < F1; ++u) {
#pragma unroll
      for (int ksl = 0; ksl < 12; ++ksl) {
        mma16(U[0], S[u & 1][ksl * 2], hb[u * 12 + ksl]);
        mma16(U[1], S[u & 1][ksl * 2 + 1], hb[u * 12 + ksl]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (u + 2 < F1 + F2) issue(S[u & 1], u + 2, c);
      else issue(S[u & 1], u + 2 - (F1 + F2), cn);
      __builtin_amdgcn_sched_barrier(0);
    }
    // lane (g, lc): token lc, hidden (chunk-local) 32 wave + 8 g + 4 t + r
    {
      const int hl = c * 128 + wave * 32 + g * 8;              // workgroup-local hidden index of the lane's first unit
      const float4 ba = *(const float4*)(b1s + hl), bb = *(const float4*)(b1s + hl + 4);
      const float bias[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
      float av[8], dv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = U[t][r] + bias[4 * t + r];
          float cdf, e;
          gelu_terms(x, cdf, e);
          av[4 * t + r] = x * cdf;
          dv[4 * t + r] = cdf + x * 0.3989422804014327f * e;
        }
      const Frag<bf16_t> af = frag_from_f32<bf16_t>(av);
      *(s16x8_t*)(gx[c & 1] + lc * GP + wave * 32 + g * 8) = af.v;
      if (p.act || p.dact) {
        const size_t o = (size_t)(row0 + lc) * HID + hb0 + hl;
        if (p.act) *(s16x8_t*)(p.act + o) = af.v;
        if (p.dact) *(s16x8_t*)(p.dact + o) = frag_from_f32<bf16_t>(dv).v;
      }
    }
    __syncthreads();                             // gx[c & 1] complete (double-buffered: one barrier per chunk)
    Frag<bf16_t> ga[4];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) ga[k2] = lds_frag_kc(gx[c & 1], GP, 0, k2 * 32, lane);
#pragma unroll
    for (int v = 0; v < F2; ++v) {
      const int u = F1 + v;
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2)
#pragma unroll
        for (int j = 0; j < 6; ++j) mma16(Y[v * 6 + j], ga[k2], S[u & 1][j * 4 + k2]);
      __builtin_amdgcn_sched_barrier(0);
      if (u + 2 < F1 + F2) issue(S[u & 1], u + 2, c);
      else issue(S[u & 1], u + 2 - (F1 + F2), cn);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- phase 3: fc2 sums -> row layout
  acc_to_patch<C, NTW>(Y, patch, nt0, lane);
  __syncthreads();
  const size_t base = (size_t)grow * C;
  if (hs > 1) {                                  // partial sums of this hidden slice; scot_deep_tail_finish continues
    float* dst = p.ypart + ((size_t)hq * p.M + grow) * C;
#pragma unroll
    for (int pp = 0; pp < RP; ++pp) {
      const int col = pp * 128 + q * 8;
      *(float4*)(dst + col) = *(const float4*)(patch + rrow * CP + col);
      *(float4*)(dst + col + 4) = *(const float4*)(patch + rrow * CP + col + 4);
    }
    return;
  }
  {
    float v[RP][8], mean, rstd;
    row_stats<C>(patch, p.b2, p.eps, v, mean, rstd);
    RowNorm n{p.b2, p.gw_w2, p.gw_b2, p.bw_w2, p.bw_b2, p.time ? p.time[samp] : 0.f, p.ss2 ? p.ss2[samp] : 1.f, p.eps};
    if (p.mean2 && q == 0) { p.mean2[grow] = mean; p.rstd2[grow] = rstd; }
#pragma unroll
    for (int pp = 0; pp < RP; ++pp) {
      const int col = pp * 128 + q * 8;
      if (p.z2) st8(p.z2, p.z_dt, base + col, v[pp]);
      float o[8];
      row_affine(n, col, v[pp], mean, rstd, hres[pp], o);
      st8(p.out, SCOT_F32, base + col, o);
      if (p.out16) st8(p.out16, SCOT_BF16, base + col, o);
      if (p.qkv) store8_ct(tileA + rrow * PA + col, o);
    }
  }
  if (!p.qkv) return;

  // ---- phase 4: the next layer's q/k/v projection on the rows just produced, all 3C output columns in one sweep of the weights
  // (C = 384 only: 18 accumulator tiles per wave; at C = 768 the hidden split excludes it anyway)
  if constexpr (C == 384) {
    constexpr int NTQ = 3 * NTW, DQ = NTQ >= 36 ? 1 : 2, QP = 3 * C + 8;
    f32x4_t acc[NTQ];
#pragma unroll
    for (int j = 0; j < NTQ; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float bq[NTQ];
#pragma unroll
    for (int j = 0; j < NTQ; ++j) bq[j] = p.bqkv ? p.bqkv[(wave * NTQ + j) * 16 + lc] : 0.f;
    rows_gemm<C, NTQ, DQ>(acc, p.Wqkv, wave * NTQ, tileA, lane, true);     // (the barrier inside: the out16 tile is complete)
    bf16_t* qp = (bf16_t*)qpatch;                // [16][3C + 8]: the rows in the operand format
#pragma unroll
    for (int j = 0; j < NTQ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) qp[(g * 4 + r) * QP + (wave * NTQ + j) * 16 + lc] = f2bf(acc[j][r] + bq[j]);
    __syncthreads();
    bf16_t* dst = p.qkv + (size_t)grow * (3 * C);
    for (int col = q * 8; col < 3 * C; col += 128) *(uint4*)(dst + col) = *(const uint4*)(qp + rrow * QP + col);
  }
}

// include/scot_hip.h: scot_deep_tail_fwd
extern "C" int scot_deep_tail_fwd(const void* a, const void* Wo_f, const float* bo, const float* x, float* h, void* h16, void* z1, float* mean1,
                                  float* rstd1, const float* gw_w1, const float* gw_b1, const float* bw_w1, const float* bw_b1,
                                  const float* sscale1, const void* W1_f, const float* b1, const void* W2_f, const float* b2, float* out,
                                  void* out16, void* act, void* dact, void* z2, float* mean2, float* rstd2, const float* gw_w2,
                                  const float* gw_b2, const float* bw_w2, const float* bw_b2, const float* sscale2, const void* Wqkv_f,
                                  const float* bqkv, void* qkv, int z_dt, const float* time, int M, int rows_per_sample, int C, int hid,
                                  float eps, int hsplit, float* ypart, hipStream_t stream) {
  if (M <= 0 || rows_per_sample <= 0 || hsplit <= 0) return SCOT_ERR_SHAPE;
  if (C != 384 && C != 768) return SCOT_ERR_UNSUPPORTED;
  if (M % 16 != 0 || rows_per_sample % 16 != 0 || hid != 4 * C || hid % (128 * hsplit) != 0) return SCOT_ERR_UNSUPPORTED;
  if (!a || !Wo_f || !bo || !x || !W1_f || !b1 || !W2_f || !gw_b1 || !bw_b1) return SCOT_ERR_SHAPE;
  if ((gw_w1 == nullptr) != (bw_w1 == nullptr) || (mean1 == nullptr) != (rstd1 == nullptr)) return SCOT_ERR_SHAPE;
  if (hsplit == 1) {
    if (!b2 || !out || !gw_b2 || !bw_b2 || (gw_w2 == nullptr) != (bw_w2 == nullptr) || (mean2 == nullptr) != (rstd2 == nullptr)) return SCOT_ERR_SHAPE;
    if ((Wqkv_f == nullptr) != (qkv == nullptr)) return SCOT_ERR_SHAPE;
    if (qkv && C != 384) return SCOT_ERR_UNSUPPORTED;
  } else if (!ypart || !h) {
    return SCOT_ERR_SHAPE;
  }
  DeepFwdArgs p;
  p.a = (const bf16_t*)a; p.Wo = (const bf16_t*)Wo_f; p.bo = bo; p.x = x; p.h = h; p.h16 = (bf16_t*)h16; p.z1 = z1; p.mean1 = mean1; p.rstd1 = rstd1;
  p.gw_w1 = gw_w1; p.gw_b1 = gw_b1; p.bw_w1 = bw_w1; p.bw_b1 = bw_b1; p.ss1 = sscale1;
  p.W1 = (const bf16_t*)W1_f; p.b1 = b1; p.W2 = (const bf16_t*)W2_f; p.b2 = b2; p.out = out; p.out16 = (bf16_t*)out16; p.act = (bf16_t*)act;
  p.dact = (bf16_t*)dact; p.z2 = z2; p.mean2 = mean2; p.rstd2 = rstd2;
  p.gw_w2 = gw_w2; p.gw_b2 = gw_b2; p.bw_w2 = bw_w2; p.bw_b2 = bw_b2; p.ss2 = sscale2;
  p.Wqkv = (const bf16_t*)Wqkv_f; p.bqkv = bqkv; p.qkv = hsplit == 1 ? (bf16_t*)qkv : nullptr;
  p.z_dt = z_dt; p.time = time; p.M = M; p.rows_per_sample = rows_per_sample; p.hid = hid; p.eps = eps; p.hsplit = hsplit; p.ypart = ypart;
  const dim3 grid((M / 16) * hsplit), block(256);
  if (C == 384) hipLaunchKernelGGL((deep_tail_fwd_kernel<384>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((deep_tail_fwd_kernel<768>), grid, block, 0, stream, p);
  return scot_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------------
// hsplit > 1: out = h + s2 · CLN2(Σ_q ypart[q] + b2).  One wave per row.
struct DeepFinishArgs {
  const float* ypart; int hsplit; const float* b2; const float* h; float* out; bf16_t* out16; void* z2; int z_dt; float* mean2; float* rstd2;
  const float* gw_w2; const float* gw_b2; const float* bw_w2; const float* bw_b2; const float* ss2; const float* time;
  int M, rows_per_sample; float eps;
};

template <int C>
__global__ __launch_bounds__(256) void deep_tail_finish_kernel(DeepFinishArgs p) {
  constexpr int RP = C / 128, CP = C + 4;
  __shared__ __attribute__((aligned(16))) float patch[16 * CP];
  const int tid = threadIdx.x, rrow = tid >> 4, q = tid & 15;
  const int grow = blockIdx.x * 16 + rrow;
  const size_t base = (size_t)grow * C;
#pragma unroll
  for (int pp = 0; pp < RP; ++pp) {
    const int col = pp * 128 + q * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int hq = 0; hq < p.hsplit; ++hq) {
      float v[8];
      ld8(p.ypart, SCOT_F32, ((size_t)hq * p.M + grow) * C + col, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
    store8_ct(patch + rrow * CP + col, s);       // (written and read back by the same thread: row_stats takes its input from the patch)
  }
  float v[RP][8], mean, rstd;
  row_stats<C>(patch, p.b2, p.eps, v, mean, rstd);
  const int samp = grow / p.rows_per_sample;
  RowNorm n{p.b2, p.gw_w2, p.gw_b2, p.bw_w2, p.bw_b2, p.time ? p.time[samp] : 0.f, p.ss2 ? p.ss2[samp] : 1.f, p.eps};
  if (p.mean2 && q == 0) { p.mean2[grow] = mean; p.rstd2[grow] = rstd; }
#pragma unroll
  for (int pp = 0; pp < RP; ++pp) {
    const int col = pp * 128 + q * 8;
    if (p.z2) st8(p.z2, p.z_dt, base + col, v[pp]);
    float res[8], o[8];
    ld8(p.h, SCOT_F32, base + col, res);
    row_affine(n, col, v[pp], mean, rstd, res, o);
    st8(p.out, SCOT_F32, base + col, o);
    if (p.out16) st8(p.out16, SCOT_BF16, base + col, o);
  }
}

// include/scot_hip.h: scot_deep_tail_finish
extern "C" int scot_deep_tail_finish(const float* ypart, int hsplit, const float* b2, const float* h, float* out, void* out16, void* z2, int z_dt,
                                     float* mean2, float* rstd2, const float* gw_w2, const float* gw_b2, const float* bw_w2, const float* bw_b2,
                                     const float* sscale2, const float* time, int M, int rows_per_sample, int C, float eps, hipStream_t stream) {
  if (M <= 0 || rows_per_sample <= 0 || hsplit <= 0) return SCOT_ERR_SHAPE;
  if (C != 384 && C != 768) return SCOT_ERR_UNSUPPORTED;
  if (M % 16 != 0 || rows_per_sample % 16 != 0) return SCOT_ERR_UNSUPPORTED;
  if (!ypart || !b2 || !h || !out || !gw_b2 || !bw_b2 || (gw_w2 == nullptr) != (bw_w2 == nullptr) || (mean2 == nullptr) != (rstd2 == nullptr))
    return SCOT_ERR_SHAPE;
  DeepFinishArgs p{ypart, hsplit, b2, h, out, (bf16_t*)out16, z2, z_dt, mean2, rstd2, gw_w2, gw_b2, bw_w2, bw_b2, sscale2, time, M, rows_per_sample, eps};
  if (C == 384) hipLaunchKernelGGL((deep_tail_finish_kernel<384>), dim3(M / 16), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((deep_tail_finish_kernel<768>), dim3(M / 16), dim3(256), 0, stream, p);
  return scot_check_launch();
}
