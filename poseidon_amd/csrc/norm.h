// Argument block shared by norm.hip (C ABI entry points, general kernels) and norm_fast.hip (vectorised kernels).
#pragma once
#include "common.h"

struct ClnFastArgs {
  const void* x; const void* resid; void* out; void* out2; float* mean; float* rstd;
  const float* time; const float* gw_w; const float* gw_b; const float* bw_w; const float* bw_b;
  int x_dt, res_dt, out_dt, out2_dt;
  int rows, rows_per_sample, C;
  float eps;
  const void* dout; void* dx; int dout_dt, dx_dt;
  float* d_gw_w; float* d_gw_b; float* d_bw_w; float* d_bw_b; float* d_xbias;
  int rpb, chunks_per_sample, nwv;
  const float* sscale;   // optional per-sample scale of the normed branch (DropPath), see norm.hip
  int mode;              // backward: 0 = dx + parameter gradients, 1 = dx only, 2 = parameter gradients only,
                         // 3 = dx + per-block partial sums of the parameter gradients into `partial` (finished by scot_cln_bwd_finish)
  float* partial;        // mode 3: [blocks][ncol] fp32, ncol = 4C ([t·dγ | dγ | t·dβ | dβ]) with conditioning, 2C ([dγ | dβ]) without
};

int scot_cln_fwd_fast(ClnFastArgs a, hipStream_t s);
int scot_cln_bwd_fast(ClnFastArgs a, void* workspace, size_t ws_bytes, hipStream_t s);
// mode 3 geometry for (rows, rows_per_sample, C): blocks and rows per block; false = the variant does not apply
bool scot_cln_bwd_partial_plan(int rows, int rows_per_sample, int C, int* blocks, int* rpb);
