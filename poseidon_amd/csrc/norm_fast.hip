// Fast path of the (conditional) layer norm for C % 8 == 0, 16-byte aligned rows (every real scOT shape).
//
// One row is owned by LPR = 2^k lanes (LPR*8*CPL >= C): the row lives in registers, is read ONCE from HBM with
// 16/32-byte loads, statistics are reduced with xor-shuffles inside the LPR-lane group, and the outputs are written
// with 16/32-byte stores: `out` (fp32 residual stream) and optionally `out2` (a copy in the GEMM operand dtype, so the
// next GEMM streams half the bytes and needs no conversion — traffic-neutral versus converting in the GEMM loader).
// Backward additionally produces, from the same pass, the per-column sums needed for
//   dgamma/dbeta (→ the four cond-LN parameter gradients, reference model.py:147-148) and
//   Σ_rows dx (→ the bias gradient of the Linear that produced x; saves a separate column-sum pass over dx).
#include "common.h"
#include <stdlib.h>

#include "norm.h"

template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPR, int CPL>
__global__ __launch_bounds__(256) void cln_fwd_fast_kernel(ClnFastArgs p) {
  constexpr int RPW = 64 / LPR;  // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int row = (blockIdx.x * 4 + wave) * RPW + sub;
  const bool rvalid = row < p.rows;
  const int C = p.C;
  const size_t base = (size_t)(rvalid ? row : 0) * C;
  float v[CPL][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = (l + i * LPR) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    if (c < C) ld8(p.x, p.x_dt, base + c, v[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1 += v[i][j]; s2 += v[i][j] * v[i][j]; }
  }
  s1 = group_sum<LPR>(s1); s2 = group_sum<LPR>(s2);
  const float mean = s1 / C;
  const float rstd = 1.0f / sqrtf(s2 / C - mean * mean + p.eps);
  if (!rvalid) return;
  if (l == 0 && p.mean) { p.mean[row] = mean; p.rstd[row] = rstd; }
  const float t = p.time ? p.time[row / p.rows_per_sample] : 0.f;
  const float sc = p.sscale ? p.sscale[row / p.rows_per_sample] : 1.f;
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = (l + i * LPR) * 8;
    if (c < C) {
      float gw[8], gb[8], bw[8], bb[8], r[8], o[8];
      ld8(p.gw_b, SCOT_F32, c, gb); ld8(p.bw_b, SCOT_F32, c, bb);
      if (p.gw_w) { ld8(p.gw_w, SCOT_F32, c, gw); ld8(p.bw_w, SCOT_F32, c, bw); }
      if (p.resid) ld8(p.resid, p.res_dt, base + c, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = p.gw_w ? gw[j] * t + gb[j] : gb[j];
        const float b = p.gw_w ? bw[j] * t + bb[j] : bb[j];
        o[j] = sc * (g * ((v[i][j] - mean) * rstd) + b) + (p.resid ? r[j] : 0.f);
      }
      st8(p.out, p.out_dt, base + c, o);
      if (p.out2) st8(p.out2, p.out2_dt, base + c, o);
    }
  }
}

// MODE 0: dx and the parameter gradients;  1: dx only (the dependent chain's half: a pure stream, no reductions over rows);
// 2: parameter gradients only (no row statistics needed: Σ dout·xhat, Σ dout) — the engine runs this half on the side stream;
// 3: dx + the block's partial column sums written to p.partial (no atomics): the small-row-count form of the deep stages
// (1024 / 4096 rows at batch 64), where mode 0 left 64-256 blocks walking 4-8 rows each one HBM round trip after the other and
// ending in 5·C global atomics — 16-20 us of pure latency on the dependent chain, 69 times per step.  Here every wave owns ONE
// pass of rows (all loads of the kernel in flight at once), and the cross-block reduction is a second tiny launch that the engine
// puts on the side stream (scot_cln_bwd_finish): nothing downstream on the chain reads parameter gradients.
// NWV = waves per block (4 or 8).  Every block ends in 4-5 global atomics per column, so with many rows (stages 0/1) twice the
// waves per block = half the blocks = half the atomics at the same number of rows per wave.
template <int LPR, int CPL, int MODE, int NWV>
__global__ __launch_bounds__(NWV * 64) void cln_bwd_fast_kernel(ClnFastArgs p) {
  constexpr int RPW = 64 / LPR;
  constexpr int NCOL = LPR * CPL * 8;      // columns covered (>= C)
  // [copy][dgamma | dbeta | dxsum][j][chunk] for column chunk*8 + j (chunk-major within j: the lanes of one access hit
  // consecutive banks); waves 0-3 combine into copy 0, waves 4-7 into copy 1
  // (MODE 3 parks the four waves' sums side by side instead: [4][dγ | dβ][NCOL] in the same LDS)
  __shared__ float red_raw[MODE == 3 ? 4 * 2 * NCOL : (NWV / 4) * 3 * NCOL];
  float (*red)[3][NCOL] = (float (*)[3][NCOL])red_raw;
  constexpr int NCH = NCOL / 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int b = blockIdx.x / p.chunks_per_sample, chunk = blockIdx.x % p.chunks_per_sample;
  const int r0 = chunk * p.rpb, r1 = min(p.rows_per_sample, r0 + p.rpb);
  const int C = p.C;
  const float t = p.time ? p.time[b] : 0.f;
  const float sc = p.sscale ? p.sscale[b] : 1.f;
  float gam[CPL][8], ag[CPL][8], ab[CPL][8], ax[CPL][8];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = (l + i * LPR) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { gam[i][j] = 0.f; ag[i][j] = 0.f; ab[i][j] = 0.f; ax[i][j] = 0.f; }
    if (c < C) {
      float gb[8], gw[8];
      ld8(p.gw_b, SCOT_F32, c, gb);
      if (p.gw_w) ld8(p.gw_w, SCOT_F32, c, gw);
#pragma unroll
      for (int j = 0; j < 8; ++j) gam[i][j] = p.gw_w ? gw[j] * t + gb[j] : gb[j];
    }
  }
  // software-pipelined over rows: the loads of the wave's next row are in flight while the current one is reduced and
  // stored (the loop was bound by one exposed HBM round trip per row, ~2 us each).  The LPR lanes of a row group share
  // `sub`, hence the trip count: group shuffles below are convergent.
  float d[CPL][8], xr[CPL][8], dn[CPL][8], xn[CPL][8];
  float mean = 0.f, rstd = 0.f, mean_n = 0.f, rstd_n = 0.f;
  auto load_row = [&](int r, float (&dd)[CPL][8], float (&xx)[CPL][8], float& mu, float& rs) {
    const int row = b * p.rows_per_sample + r;
    const size_t base = (size_t)row * C;
    mu = p.mean[row]; rs = p.rstd[row];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = (l + i * LPR) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) { dd[i][j] = 0.f; xx[i][j] = 0.f; }
      if (c < C) {
        ld8(p.dout, p.dout_dt, base + c, dd[i]);
        ld8(p.x, p.x_dt, base + c, xx[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) dd[i][j] *= sc;
      }
    }
  };
  int r = r0 + wave * RPW + sub;   // wave < NWV
  if (r < r1) load_row(r, d, xr, mean, rstd);
  for (; r < r1; r += NWV * RPW) {
    // (an unconditional prefetch — re-reading the last row — was tried so that the compiler can count loads in flight:
    // the extra row costs more than the tighter waits save when a block only makes 2-8 passes)
    const int rn = r + NWV * RPW;
    if (rn < r1) load_row(rn, dn, xn, mean_n, rstd_n);
    const size_t base = (size_t)(b * p.rows_per_sample + r) * C;
    float xh[CPL][8];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[i][j] = (xr[i][j] - mean) * rstd;     // columns >= C hold d = 0, gam = 0: they add nothing below
        const float g = d[i][j] * gam[i][j];
        m1 += g; m2 += g * xh[i][j];
      }
    }
    if (MODE != 2) { m1 = group_sum<LPR>(m1) / C; m2 = group_sum<LPR>(m2) / C; }
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = (l + i * LPR) * 8;
      if (c < C) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o[j] = rstd * (d[i][j] * gam[i][j] - m1 - xh[i][j] * m2);
          if (MODE != 1) { ag[i][j] += d[i][j] * xh[i][j]; ab[i][j] += d[i][j]; }
          if (MODE == 0) ax[i][j] += o[j];
        }
        if (MODE != 2) st8(p.dx, p.dx_dt, base + c, o);
      }
    }
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { d[i][j] = dn[i][j]; xr[i][j] = xn[i][j]; }
    mean = mean_n; rstd = rstd_n;
  }
  if (MODE == 1) return;
  // reduce over the RPW row-groups of the wave (same columns live in lanes l, l+LPR, ...), then over the four waves by
  // taking turns on the LDS copy with plain read-add-write (ds_add_f32 measured ~300 ns per instruction here: 48 of them
  // were 15 us of a 24 us kernel at C = 768)
#pragma unroll
  for (int i = 0; i < CPL; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        ag[i][j] += __shfl_xor(ag[i][j], o, 64); ab[i][j] += __shfl_xor(ab[i][j], o, 64); ax[i][j] += __shfl_xor(ax[i][j], o, 64);
      }
    }
  if (MODE == 3) {
    // one barrier: every wave parks its column sums in its own LDS slice, the first C threads add the four
    float (*part)[2][NCOL] = (float (*)[2][NCOL])red_raw;
    if (sub == 0) {
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = j * NCH + l + i * LPR;
          part[wave][0][c] = ag[i][j]; part[wave][1][c] = ab[i][j];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const int k = (c & 7) * NCH + (c >> 3);
      const float dg = part[0][0][k] + part[1][0][k] + part[2][0][k] + part[3][0][k];
      const float db = part[0][1][k] + part[1][1][k] + part[2][1][k] + part[3][1][k];
      if (p.gw_w) {
        float* row = p.partial + (size_t)blockIdx.x * 4 * C;
        row[c] = t * dg; row[C + c] = dg; row[2 * C + c] = t * db; row[3 * C + c] = db;
      } else {
        float* row = p.partial + (size_t)blockIdx.x * 2 * C;
        row[c] = dg; row[C + c] = db;
      }
    }
    return;
  }
  float (*rc)[NCOL] = red[wave >> 2];
  for (int w = 0; w < 4; ++w) {
    if ((wave & 3) == w && sub == 0) {
#pragma unroll
      for (int i = 0; i < CPL; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = j * NCH + l + i * LPR;
          if (w == 0) { rc[0][c] = ag[i][j]; rc[1][c] = ab[i][j]; rc[2][c] = ax[i][j]; }
          else { rc[0][c] += ag[i][j]; rc[1][c] += ab[i][j]; rc[2][c] += ax[i][j]; }
        }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < C; c += NWV * 64) {
    const int k = (c & 7) * NCH + (c >> 3);
    float dg = red[0][0][k], db = red[0][1][k], dxs = red[0][2][k];
    if (NWV == 8) { dg += red[NWV / 4 - 1][0][k]; db += red[NWV / 4 - 1][1][k]; dxs += red[NWV / 4 - 1][2][k]; }
    if (p.d_gw_w) { atomicAdd(&p.d_gw_w[c], t * dg); atomicAdd(&p.d_bw_w[c], t * db); }
    atomicAdd(&p.d_gw_b[c], dg);
    atomicAdd(&p.d_bw_b[c], db);
    if (p.d_xbias) atomicAdd(&p.d_xbias[c], dxs);
  }
}

template <int LPR, int CPL> static void launch_fwd(const ClnFastArgs& a, hipStream_t s) {
  const int rpb = 4 * (64 / LPR);
  hipLaunchKernelGGL((cln_fwd_fast_kernel<LPR, CPL>), dim3((a.rows + rpb - 1) / rpb), dim3(256), 0, s, a);
}
template <int LPR, int CPL> static void launch_bwd(const ClnFastArgs& a, hipStream_t s) {
  const dim3 grid((a.rows / a.rows_per_sample) * a.chunks_per_sample);
  if (a.mode == 3) hipLaunchKernelGGL((cln_bwd_fast_kernel<LPR, CPL, 3, 4>), grid, dim3(256), 0, s, a);
  else if (a.mode == 1) hipLaunchKernelGGL((cln_bwd_fast_kernel<LPR, CPL, 1, 4>), grid, dim3(256), 0, s, a);
  else if (a.mode == 2) hipLaunchKernelGGL((cln_bwd_fast_kernel<LPR, CPL, 2, 4>), grid, dim3(256), 0, s, a);
  else if (a.nwv == 8) hipLaunchKernelGGL((cln_bwd_fast_kernel<LPR, CPL, 0, 8>), grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL((cln_bwd_fast_kernel<LPR, CPL, 0, 4>), grid, dim3(256), 0, s, a);
}

#define CLN_DISPATCH(FN)                                            \
  const int nch = a.C / 8;                                          \
  if (nch <= 1) FN<1, 1>(a, s);                                     \
  else if (nch <= 2) FN<2, 1>(a, s);                                \
  else if (nch <= 4) FN<4, 1>(a, s);                                \
  else if (nch <= 8) FN<8, 1>(a, s);                                \
  else if (nch <= 16) FN<16, 1>(a, s);                              \
  else if (nch <= 32) FN<32, 1>(a, s);                              \
  else if (nch <= 64) FN<64, 1>(a, s);                              \
  else if (nch <= 128) FN<64, 2>(a, s);                             \
  else if (nch <= 192) FN<64, 3>(a, s);                             \
  else return SCOT_ERR_UNSUPPORTED;

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

int scot_cln_fwd_fast(ClnFastArgs a, hipStream_t s) {
  if (a.C % 8 || !aligned16(a.x) || !aligned16(a.out) || !aligned16(a.resid) || !aligned16(a.out2) || !aligned16(a.gw_w) ||
      !aligned16(a.gw_b) || !aligned16(a.bw_w) || !aligned16(a.bw_b))
    return SCOT_ERR_UNSUPPORTED;
  CLN_DISPATCH(launch_fwd)
  return scot_check_launch();
}
// mode 3: one pass of rows per wave; applies to the small, wide norms of the deep stages (few rows: the partial matrix stays small)
bool scot_cln_bwd_partial_plan(int rows, int rows_per_sample, int C, int* blocks, int* rpb) {
  if (C % 64 || C < 128 || C > 1536 || rows > 8192 || rows <= 0 || rows_per_sample <= 0 || rows % rows_per_sample) return false;
  int lpr = 1;
  while (lpr < 64 && lpr * 8 < C) lpr <<= 1;
  const int rows_per_pass = 4 * (64 / lpr);
  const int r = rows_per_sample < rows_per_pass ? rows_per_sample : rows_per_pass;
  const int cps = (rows_per_sample + r - 1) / r;
  *blocks = (rows / rows_per_sample) * cps;
  *rpb = r;
  return true;
}

// out[j] += Σ_b partial[b][j]  (j < ncol): the cross-block half of mode 3.  grid (ceil(ncol / 64), slices), 64 columns x 4 block lanes
__global__ __launch_bounds__(256) void cln_partial_reduce_kernel(const float* __restrict__ partial, int nblk, int ncol, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float acc = 0.f;
  if (col < ncol)
    for (int b = b0 + ry; b < b1; b += 4) acc += partial[(size_t)b * ncol + col];
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && col < ncol) atomicAdd(&out[col], red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
}

// Up to 32 partial matrices in ONE launch (blockIdx.z picks the matrix): the finishing pass of a whole stage's norm backwards.  Every
// such pass used to be its own 10-14 us launch on the weight-gradient stream — which is as busy as the main chain during the backward.
struct PartialBatch { const float* partial[32]; float* out[32]; int nblk[32]; int ncol[32]; };
__global__ __launch_bounds__(256) void cln_partial_reduce_batch_kernel(PartialBatch b) {
  __shared__ float red[4][64];
  const int k = blockIdx.z;
  const float* __restrict__ partial = b.partial[k];
  const int nblk = b.nblk[k], ncol = b.ncol[k];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  if (blockIdx.x * 64 >= ncol) return;                    // (whole workgroup: the grid is sized for the widest matrix)
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float acc = 0.f;
  if (col < ncol)
    for (int r = b0 + ry; r < b1; r += 4) acc += partial[(size_t)r * ncol + col];
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && col < ncol && b0 < b1) atomicAdd(&b.out[k][col], red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx]);
}
extern "C" int scot_partial_colsum_batch(int n, const float* const* partial, const int* nblk, const int* ncol, float* const* out,
                                         hipStream_t s) {
  if (n <= 0 || n > 32 || !partial || !nblk || !ncol || !out) return SCOT_ERR_SHAPE;
  PartialBatch b;
  int maxcol = 0, maxblk = 0;
  for (int i = 0; i < n; ++i) {
    if (!partial[i] || !out[i] || nblk[i] <= 0 || ncol[i] <= 0) return SCOT_ERR_SHAPE;
    b.partial[i] = partial[i]; b.out[i] = out[i]; b.nblk[i] = nblk[i]; b.ncol[i] = ncol[i];
    if (ncol[i] > maxcol) maxcol = ncol[i];
    if (nblk[i] > maxblk) maxblk = nblk[i];
  }
  for (int i = n; i < 32; ++i) { b.partial[i] = nullptr; b.out[i] = nullptr; b.nblk[i] = 0; b.ncol[i] = 0; }
  int slices = maxblk / 64;
  if (slices < 1) slices = 1;
  if (slices > 8) slices = 8;
  hipLaunchKernelGGL(cln_partial_reduce_batch_kernel, dim3((maxcol + 63) / 64, slices, n), dim3(256), 0, s, b);
  return scot_check_launch();
}

int scot_cln_bwd_finish_launch(const float* partial, int nblk, int ncol, float* out, hipStream_t s) {
  int slices = nblk / 32;
  if (slices < 1) slices = 1;
  if (slices > 16) slices = 16;
  hipLaunchKernelGGL(cln_partial_reduce_kernel, dim3((ncol + 63) / 64, slices), dim3(256), 0, s, partial, nblk, ncol, out);
  return scot_check_launch();
}

int scot_cln_bwd_fast(ClnFastArgs a, void* workspace, size_t ws_bytes, hipStream_t s) {
  if (a.C % 8 || !aligned16(a.x) || !aligned16(a.dout) || !aligned16(a.dx) || !aligned16(a.gw_w) || !aligned16(a.gw_b))
    return SCOT_ERR_UNSUPPORTED;
  if (a.mode == 2 && a.d_xbias) return SCOT_ERR_UNSUPPORTED;   // Σ dx needs dx
  if (a.mode == 3) {
    int blocks, rpb;
    if (a.d_xbias || !scot_cln_bwd_partial_plan(a.rows, a.rows_per_sample, a.C, &blocks, &rpb)) return SCOT_ERR_UNSUPPORTED;
    const size_t need = (size_t)blocks * (a.gw_w ? 4 : 2) * a.C * sizeof(float);
    if (!workspace || (((uintptr_t)workspace) & 15) || ws_bytes < need) return SCOT_ERR_SHAPE;
    a.partial = (float*)workspace;
    a.nwv = 4;
    a.rpb = rpb;
    a.chunks_per_sample = (a.rows_per_sample + rpb - 1) / rpb;
    CLN_DISPATCH(launch_bwd)
    return scot_check_launch();
  }
  // rows per block: enough blocks to cover the chip (256: every block ends in 5·C global atomics) but at least two passes of the four
  // waves, at most 128 rows
  const int rpb_env = 0, blocks_env = 256;
  int lpr = 1;
  while (lpr < 64 && lpr * 8 < a.C) lpr <<= 1;
  a.nwv = (a.mode == 0 && a.rows >= 16384) ? 8 : 4;
  const int rows_per_pass = a.nwv * (64 / lpr);
  const int target_blocks = a.mode == 1 ? 2048 : blocks_env;   // dx only: nothing to flush per block, so many short blocks
  int rpb = rpb_env > 0 ? rpb_env : (a.rows / target_blocks) / rows_per_pass * rows_per_pass;
  if (rpb < 2 * rows_per_pass) rpb = 2 * rows_per_pass;
  if (rpb_env <= 0 && rpb > 32 * a.nwv) rpb = 32 * a.nwv;
  a.rpb = a.rows_per_sample < rpb ? a.rows_per_sample : rpb;
  a.chunks_per_sample = (a.rows_per_sample + a.rpb - 1) / a.rpb;
  CLN_DISPATCH(launch_bwd)
  return scot_check_launch();
}
