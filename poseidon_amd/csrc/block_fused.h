// Row-wise halves shared by the fused block kernels (mlp_fused.hip): a wave owns 16·TT rows; a 16-row tile is handled in the
// "row-contiguous" layout — lane (prow = lane>>2, q = lane&3) holds columns 32·pp + 8·q .. +7 of row prow for pp < C/32 — so
// that a row's statistics are two xor-shuffles away and every global access is a 16/32-byte row segment.
//   cln_rows_epilogue : MFMA accumulators [rows, C] -> + bias -> (conditional) layer norm -> DropPath scale -> + residual
//   cln_bwd_rows      : the backward of that norm for the same rows, ending in MFMA operand fragments of dz
#pragma once
#include "common.h"

// K pitch of an LDS tile whose contraction (or fragment k index) runs over the C channels: C + 8, or — C % 32 != 0 (C = 48: the layer tails of
// Poseidon-T / -S stage 0) — the channels rounded up to whole 32-wide MFMA K-steps + 8, the columns >= C zero-filled
template <int C> struct KPitch { static constexpr int KP = (C + 31) / 32 * 32, P = KP + 8; };

struct ClnRowsOut {
  const float* bias;                    // [C] bias of the GEMM that produced the accumulators
  void* z; int z_dt;                    // training: pre-norm rows [M, C] (fp32, or 16-bit: only the backward's x-hat reads them)
  float* mean; float* rstd;             // ... and their statistics [M]   (all NULL in inference)
  const float* time; const float* gw_w; const float* gw_b; const float* bw_w; const float* bw_b; const float* sscale;
  const float* resid;                   // [M, C] fp32 residual stream
  float* out; bf16_t* out16;            // [M, C]
  int M, rows_per_sample;
  float eps;
};

// Y[tt][nt]: accumulator tiles of the wave's rows row0 + 16 tt .. (C/D layout: row 4g + r, column 16 nt + lc).
// patch_base: LDS, 4 waves x 16 x (C+4) floats, must be dead (the caller has passed a __syncthreads since its last use).
// tile16 (optional): per-wave LDS tile [16·TT][C + 8] that also receives the 16-bit output rows (the fused block tail reads them
// back as MFMA operand fragments instead of re-loading out16 from HBM); must not overlap the fp32 patches.
// C % 32 != 0 (C = 48, Poseidon-T / -S stage 0: forward tail only): the row layout keeps ceil(C / 32) pieces per lane and the pieces
// whose first column is >= C do not exist (no load, no store, nothing in the statistics); every `pv` below is compile-time true otherwise.
template <int C, int TT>
__device__ __forceinline__ void cln_rows_epilogue(f32x4_t (&Y)[TT][C / 16], float* patch_base, int row0, const ClnRowsOut& p,
                                                  bf16_t* tile16 = nullptr) {
  constexpr int KJ = (C + 31) / 32, NT = C / 16, CP = C + 4;   // CP % 16 == 4: the 4 row groups of a tile write disjoint banks
  constexpr bool RAG = (C % 32) != 0;
  static_assert(C % 16 == 0, "channel tiles");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, lc = lane & 15;
  float* Ct = patch_base + wave * 16 * CP;
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
      const bool pv = !RAG || col < C;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[pp][j] = 0.f;
      if (pv) {
        const float4 x0 = *(const float4*)(Ct + prow * CP + col), x1 = *(const float4*)(Ct + prow * CP + col + 4);
        float bb[8];
        ld8(p.bias, SCOT_F32, col, bb);
        v[pp][0] = x0.x + bb[0]; v[pp][1] = x0.y + bb[1]; v[pp][2] = x0.z + bb[2]; v[pp][3] = x0.w + bb[3];
        v[pp][4] = x1.x + bb[4]; v[pp][5] = x1.y + bb[5]; v[pp][6] = x1.z + bb[6]; v[pp][7] = x1.w + bb[7];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s1 += v[pp][j];
    }
    s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
    const float mean = s1 * (1.0f / C);
    float s2 = 0.f;
#pragma unroll
    for (int pp = 0; pp < KJ; ++pp) {
      if (RAG && pp * 32 + q * 8 >= C) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[pp][j] - mean; s2 += d * d; }
    }
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
        if (RAG && col >= C) continue;
        if (p.z) st8(p.z, p.z_dt, base + col, v[pp]);
        float gw[8], gb[8], bw[8], bbv[8], res[8], o[8];
        ld8(p.gw_b, SCOT_F32, col, gb); ld8(p.bw_b, SCOT_F32, col, bbv);
        if (p.gw_w) { ld8(p.gw_w, SCOT_F32, col, gw); ld8(p.bw_w, SCOT_F32, col, bw); }
        ld8(p.resid, SCOT_F32, base + col, res);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float ga = p.gw_w ? gw[j] * t + gb[j] : gb[j];
          const float be = p.gw_w ? bw[j] * t + bbv[j] : bbv[j];
          o[j] = sc * (ga * ((v[pp][j] - mean) * rstd) + be) + res[j];
        }
        st8(p.out, SCOT_F32, base + col, o);
        if (p.out16) st8(p.out16, SCOT_BF16, base + col, o);
        if (tile16) store8_ct(tile16 + (tt * 16 + prow) * (C + 8) + col, o);
      }
    } else if (tile16) {
      const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pp = 0; pp < KJ; ++pp)
        if (!RAG || pp * 32 + q * 8 < C) store8_ct(tile16 + (tt * 16 + prow) * (C + 8) + pp * 32 + q * 8, zero);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

struct ClnRowsBwd {
  const float* g;                        // [M, C] gradient wrt the block output (fp32 residual stream gradient)
  const void* z; int z_dt; const float* mean; const float* rstd;
  const float* time; const float* gw_w; const float* gw_b; const float* sscale;
  bf16_t* dz;                            // [M, C] gradient wrt the pre-norm rows (the weight-gradient GEMM reads it)
  float* d_gw_w; float* d_gw_b; float* d_bw_w; float* d_bw_b;
  float* partial;                        // optional: [workgroups][4·Cp] ([t·dγ | dγ | t·dβ | dβ], each Cp = C rounded up to 64 floats wide, the
                                         // pad zero: the stride of the four tensors in the parameter arena; 2·Cp without conditioning)
                                         // receives the workgroup's column sums instead of 4·C global atomics (scot_partial_colsum)
  int M, rows_per_sample;
};

// LDS needed at `lds` (16-byte aligned, dead): 4 waves x 16 x (C+8) bf16 patches, then 4 x 2 x C floats of column sums.
template <int C> struct ClnBwdLds {
  static constexpr size_t patch_bytes = (size_t)4 * 16 * KPitch<C>::P * 2;
  static constexpr size_t bytes = patch_bytes + (size_t)4 * 2 * C * 4;
};

// dz = CLN_bwd(s·g) for the workgroup's 64·TT rows (wave w: rows wg_row0 + 16 TT w ..): written to HBM (bf16) and returned as
// fragments dzf[tt][j] (lane (r = lane&15, g): row/column r of the tile, k = channels 32 j + 8 g .. +7 — usable as the A or the
// B operand).  The four parameter gradients are reduced over the workgroup's rows and added with one atomic per column; the
// conditioning time must be uniform over the workgroup (rows_per_sample % (64·TT) == 0).  Ends with the atomics ISSUED but no
// barrier after them: the caller must __syncthreads() before it overwrites `lds`.
// GREG: the rows of g are already in registers (greg[tt][pp][j], same lane layout as the loads they replace) — the fused
// block-tail backward hands the MLP half's result straight to the attention half's norm.
template <int C, int TT, bool GREG = false>
__device__ __forceinline__ void cln_bwd_rows(Frag<bf16_t> (&dzf)[TT][(C + 31) / 32], char* lds, int wg_row0, const ClnRowsBwd& p,
                                             const float (*greg)[(C + 31) / 32][8] = nullptr) {
  constexpr int KJ = (C + 31) / 32, PD = KPitch<C>::P;
  constexpr bool RAG = (C % 32) != 0;          // C = 48: the row pieces at columns >= C do not exist; their lanes keep the dz patch's pad zero
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int prow = lane >> 2, q = lane & 3;
  bf16_t* Dz = (bf16_t*)lds + wave * 16 * PD;
  float* red = (float*)(lds + ClnBwdLds<C>::patch_bytes);      // [wave][2][C]
  const int row0 = wg_row0 + wave * (16 * TT);
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
      if (RAG && col >= C) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { d[pp][j] = 0.f; xh[pp][j] = 0.f; }
        continue;
      }
      if (GREG) {
#pragma unroll
        for (int j = 0; j < 8; ++j) d[pp][j] = greg[tt][pp][j];
      } else {
        ld8(p.g, SCOT_F32, base + col, d[pp]);
      }
      ld8(p.z, p.z_dt, base + col, zz);
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
      if (RAG && col >= C) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
      } else if (valid) st8(p.dz, SCOT_BF16, base + col, o);
      store8_ct(Dz + prow * PD + col, o);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < KJ; ++j) dzf[tt][j] = lds_frag_kc(Dz, PD, 0, j * 32, lane);
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
    for (int pp = 0; pp < KJ; ++pp) {
      if (RAG && pp * 32 + q * 8 >= C) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wave * 2 + 0) * C + pp * 32 + q * 8 + j] = ag[pp][j];
        red[(wave * 2 + 1) * C + pp * 32 + q * 8 + j] = ab[pp][j];
      }
    }
  }
  __syncthreads();
  constexpr int CP = (C + 63) / 64 * 64;
  if (p.partial && tid >= C && tid < CP) {       // the pad columns of the partial row
    float* row = p.partial + (size_t)blockIdx.x * (p.gw_w ? 4 : 2) * CP;
    row[tid] = 0.f; row[CP + tid] = 0.f;
    if (p.gw_w) { row[2 * CP + tid] = 0.f; row[3 * CP + tid] = 0.f; }
  }
  if (tid < C) {
    float dg = 0.f, db = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { dg += red[(w * 2 + 0) * C + tid]; db += red[(w * 2 + 1) * C + tid]; }
    if (p.partial) {
      if (p.gw_w) {
        float* row = p.partial + (size_t)blockIdx.x * 4 * CP;
        row[tid] = t * dg; row[CP + tid] = dg; row[2 * CP + tid] = t * db; row[3 * CP + tid] = db;
      } else {
        float* row = p.partial + (size_t)blockIdx.x * 2 * CP;
        row[tid] = dg; row[CP + tid] = db;
      }
    } else {
      if (p.d_gw_w) { atomicAdd(&p.d_gw_w[tid], t * dg); atomicAdd(&p.d_bw_w[tid], t * db); }
      atomicAdd(&p.d_gw_b[tid], dg);
      atomicAdd(&p.d_bw_b[tid], db);
    }
  }
}
