// Argument block of the grouped weight-gradient launches (gemm_fast.hip: 64 x 64 / 96 x 96 tiles; wgrad_wide.hip: 128 x 128 tiles) and of
// their grouped split-K reduce: the (up to 8) problems dW_i[M_i, N_i] += dY_i[K, M_i]^T · X_i[K, N_i] of one ScOTLayer over the same K tokens.
#pragma once
#include "common.h"

#define SCOT_WGRAD_GROUP_MAX 8
struct WgradProblem {
  const void* A; const void* B; float* C; float* colsum;   // A = dY [K, M] (lda), B = X [K, N] (ldb), C = dW [M, N] (ldc, +=)
  int M, N, lda, ldb, ldc;
  int tiles_n, tile0;          // tiles along N; index of this problem's first tile in the group
  unsigned ws_off;             // offset (floats) of this problem's partial tiles inside one K-slice plane of the workspace
  int mode;                    // how the result meets C: 0 = C += acc, 1 = C = s·acc (first writer of a lazily zeroed gradient), 2 = C += s·acc
};
#define SCOT_GRAD_ADD 0
#define SCOT_GRAD_STORE_SCALED 1
#define SCOT_GRAD_ADD_SCALED 2
struct WgradGroupArgs {
  WgradProblem p[SCOT_WGRAD_GROUP_MAX];
  int n, K, ksplit, nsplit, tiles;
  float* ws; size_t plane;     // ws[z][plane]: partial sums of K slice z (all problems back to back); plane in floats
  int use_tr;
  const float* scale;          // device scalar s of modes 1 / 2 (the fp16 build's 1 / gradient scale), NULL = 1
};

// v (8 consecutive result values) meets the 8 floats at c according to `mode` (see WgradProblem::mode)
__device__ __forceinline__ void grad_commit8(float* C, size_t ci, float (&v)[8], int mode, const float* scale) {
  if (mode != SCOT_GRAD_ADD) {
    const float s = scale ? *scale : 1.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= s;
  }
  if (mode != SCOT_GRAD_STORE_SCALED) {
    float o[8];
    ld8(C, SCOT_F32, ci, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += o[j];
  }
  st8(C, SCOT_F32, ci, v);
}

