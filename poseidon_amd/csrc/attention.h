// Shared device helpers of the window-attention kernels (attention.hip: any window size; attention_w16.hip: the
// 16x16-window fast path).
#pragma once
#include "common.h"

struct AttnArgs {
  const void* qkv;   // [tokens][3C]  (q | k | v), dtype = compute type
  void* out;         // fwd: O [tokens][C];           bwd: dqkv [tokens][3C]
  const void* dout;  // bwd: dO [tokens][C]
  const void* ofwd;  // bwd: O  [tokens][C] (the forward output; delta = rowsum(dO ∘ O))
  float* lse;        // [windows][heads][N]  log-sum-exp of each softmax row (fwd writes, bwd reads)
  const float* bias_table;   // [heads][(2ws-1)^2]  = 16·sigmoid(CPB-MLP)
  const float* logit_scale;  // [heads]
  float* dbias_table;        // bwd, atomically accumulated
  float* dlogit_scale;       // bwd, atomically accumulated
  // bwd: REPLICAS of the two atomically accumulated buffers.  Every (window, head) workgroup of a head adds into the same (2ws-1)^2 + 1
  // addresses — 64 (one window per sample) to 256 same-address atomics in a row, each a serialized L2 read-modify-write: 12 of the 22 us of
  // the 4x4-window backward, 6 of 33 at 8x8, 10 of 88 at 16x16 (ablation, profiles/round4).  Window w adds into replica w % nrep
  // (dbias_table + r·rep_stride_tab, dlogit_scale + r·rep_stride_ls); scot_replica_reduce folds the replicas beside the chain.
  int nrep; size_t rep_stride_tab, rep_stride_ls;
  int C, heads, Hp, Wp, ws, shift, nwx, nw_per_img;
  int use_tr;
};

__device__ __forceinline__ int win_token(const AttnArgs& p, int win, int n) {
  const int b = win / p.nw_per_img, w = win % p.nw_per_img;
  const int wy = w / p.nwx, wx = w % p.nwx;
  int y = wy * p.ws + n / p.ws + p.shift, x = wx * p.ws + n % p.ws + p.shift;
  if (y >= p.Hp) y -= p.Hp;
  if (x >= p.Wp) x -= p.Wp;
  return (b * p.Hp + y) * p.Wp + x;
}
// region id on the shifted grid (reference model.py:450-465), 0 when shift == 0
__device__ __forceinline__ int win_region(const AttnArgs& p, int win, int n) {
  if (p.shift == 0) return 0;
  const int w = win % p.nw_per_img;
  const int ys = (w / p.nwx) * p.ws + n / p.ws, xs = (w % p.nwx) * p.ws + n % p.ws;
  const int ry = (ys >= p.Hp - p.ws) + (ys >= p.Hp - p.shift);
  const int rx = (xs >= p.Wp - p.ws) + (xs >= p.Wp - p.shift);
  return ry * 3 + rx;
}

// per-position info packed for the logit loops: (y*(2ws-1) + x) | region << 20 ; region 15 = padding row
__device__ __forceinline__ int pos_info(const AttnArgs& p, int win, int n, int N) {
  if (n >= N) return 15 << 20;
  return ((n / p.ws) * (2 * p.ws - 1) + n % p.ws) | (win_region(p, win, n) << 20);
}
// LDS tiles are [n][KD + pad] with KD = HD rounded up to the MFMA K-step (32); columns HD..KD are zero so that
// K-contiguous fragment reads of head_dim 16 never touch uninitialised LDS.
template <int HD, typename CT> constexpr int row_pitch() { return ((HD + 31) / 32) * 32 + ct_traits<CT>::kpad; }

// Stage the window's rows of one of q/k/v (column offset `col`) into LDS tile [NP][pitch]; optionally L2-normalise
// each row (F.normalize, eps 1e-12).  256 threads, HD/8 lanes per row.
template <typename CT, int HD, int NP>
__device__ __forceinline__ void stage_rows(CT* tile, const void* src, int ld, int col, const int* tok, int N,
                                           bool normalize, int tid) {
  constexpr int CPR = ((HD + 31) / 32) * 4, pitch = row_pitch<HD, CT>();
  for (int c = tid; c < NP * CPR; c += (int)blockDim.x) {
    const int n = c / CPR, d8 = (c % CPR) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (n < N && d8 < HD) ld8(src, ct_traits<CT>::dtype, (size_t)tok[n] * ld + col + d8, v);
    if (normalize) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
#pragma unroll
      for (int o = 1; o < CPR; o <<= 1) ss += __shfl_xor(ss, o, 64);
      const float r = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= r;
    }
    store8_ct(tile + n * pitch + d8, v);
  }
}

// B-operand fragments of a 16-row block read straight from HBM: lane (c = lane&15, g) takes 8 consecutive
// features d = kk*32 + g*8 .. +7 of row tok(n0+c).  Optionally L2-normalised (returns 1/max(|row|,eps) in *rnorm).
template <typename CT, int HD>
__device__ __forceinline__ void load_rows_frag(Frag<CT> (&f)[(HD + 31) / 32], const void* src, int ld, int col,
                                               const int* tok, int n0, int N, bool normalize, int lane) {
  constexpr int KS = (HD + 31) / 32;
  const int n = n0 + (lane & 15), g = lane >> 4;
  float v[KS][8];
  float ss = 0.f;
  const bool valid = n < N;
  const size_t base = valid ? (size_t)tok[n] * ld + col : 0;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int d = kk * 32 + g * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[kk][j] = 0.f;
    if (valid && d < HD) ld8(src, ct_traits<CT>::dtype, base + d, v[kk]);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += v[kk][j] * v[kk][j];
  }
  float r = 1.f;
  if (normalize) {
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    r = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  }
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[kk][j] *= r;
    f[kk] = frag_from_f32<CT>(v[kk]);
  }
}

// row_shl:k — lane l of each 16-lane row receives lane l+k (0 shifted in);  row_shr:k — lane l receives lane l-k.
template <int CTRL> __device__ __forceinline__ float dpp_row(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Raw fp32 values of a 16-row block in B-operand order (lane (c = lane&15, g): features kk*32 + g*8 .. +7 of row n0+c).
template <typename CT, int HD>
__device__ __forceinline__ void load_rows_f32(float (&v)[(HD + 31) / 32][8], const void* src, int ld, int col, const int* tok,
                                              int n0, int N, int lane) {
  constexpr int KS = (HD + 31) / 32;
  const int n = n0 + (lane & 15), g = lane >> 4;
  const bool valid = n < N;
  const size_t base = valid ? (size_t)tok[n] * ld + col : 0;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    const int d = kk * 32 + g * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[kk][j] = 0.f;
    if (valid && d < HD) ld8(src, ct_traits<CT>::dtype, base + d, v[kk]);
  }
}

// fragments from raw fp32 rows in B-operand order (see load_rows_f32), optionally L2-normalised over the whole row
template <typename CT, int HD>
__device__ __forceinline__ void rows_to_frag(Frag<CT> (&f)[(HD + 31) / 32], float (&v)[(HD + 31) / 32][8], bool normalize) {
  constexpr int KS = (HD + 31) / 32;
  float r = 1.f;
  if (normalize) {
    float ss = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[kk][j] * v[kk][j];
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    r = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  }
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[kk][j] *= r;
    f[kk] = frag_from_f32<CT>(v[kk]);
  }
}
