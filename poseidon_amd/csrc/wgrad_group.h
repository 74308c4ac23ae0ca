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
};
struct WgradGroupArgs {
  WgradProblem p[SCOT_WGRAD_GROUP_MAX];
  int n, K, ksplit, nsplit, tiles;
  float* ws; size_t plane;     // ws[z][plane]: partial sums of K slice z (all problems back to back); plane in floats
  int use_tr;
};

