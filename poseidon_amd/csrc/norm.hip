// Time-conditioned layer norm (+ fused residual add) — forward and backward.
//
// Reference: ConditionalLayerNorm.forward (scOT/model.py:143-160):  mean, var = E[x^2]-mean^2 (biased, NOT clamped),
// xhat = (x-mean)/sqrt(var+eps); gamma = W_g·t + b_g, beta = W_b·t + b_b (two nn.Linear(1,C)); y = gamma·xhat + beta.
// With use_conditioning=False the reference uses nn.LayerNorm ignoring `time` (model.py:135-140): pass gw_w = bw_w = NULL
// and the plain weight/bias as gw_b / bw_b.  SwinV2 "res-post-norm" (model.py:570,574) is the fused form
// out = resid + y.  Statistics, gamma/beta and the residual stream stay in fp32 in every compute mode.
// sample_scale (optional, [rows / rows_per_sample]): out = resid + s_b·y — Swinv2DropPath (HF:565-586, applied to the normed
// branch at model.py:570,574): s_b = mask_b / keep_prob drawn by the caller; the backward scales the upstream gradient by s_b.
#include "norm.h"

struct ClnArgs {
  const void* x; const void* resid; void* out; float* mean; float* rstd;
  const float* time; const float* gw_w; const float* gw_b; const float* bw_w; const float* bw_b;
  int x_dt, res_dt, out_dt;
  int rows, rows_per_sample, C;
  float eps;
  // backward
  const void* dout; void* dx; int dout_dt, dx_dt;
  float* d_gw_w; float* d_gw_b; float* d_bw_w; float* d_bw_b;
  int vec;
  const float* sscale;
  int mode;   // backward: 0 = dx + parameter gradients, 1 = dx only, 2 = parameter gradients only
};

// one wave per row
__global__ __launch_bounds__(256) void cln_fwd_kernel(ClnArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const int C = p.C;
  const size_t base = (size_t)row * C;
  float s1 = 0.f, s2 = 0.f;
  if (p.vec) {
    for (int c = lane * 8; c < C; c += 512) {
      float v[8];
      ld8(p.x, p.x_dt, base + c, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1 += v[j]; s2 += v[j] * v[j]; }
    }
  } else {
    for (int c = lane; c < C; c += 64) { const float v = ld1(p.x, p.x_dt, base + c); s1 += v; s2 += v * v; }
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  const float mean = s1 / C;
  const float var = s2 / C - mean * mean;
  const float rstd = 1.0f / sqrtf(var + p.eps);
  if (lane == 0 && p.mean) { p.mean[row] = mean; p.rstd[row] = rstd; }
  const float t = p.time ? p.time[row / p.rows_per_sample] : 0.f;
  const float sc = p.sscale ? p.sscale[row / p.rows_per_sample] : 1.f;
  if (p.vec) {
    for (int c = lane * 8; c < C; c += 512) {
      float v[8], r[8], o[8];
      ld8(p.x, p.x_dt, base + c, v);
      if (p.resid) ld8(p.resid, p.res_dt, base + c, r);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = p.gw_w ? p.gw_w[c + j] * t + p.gw_b[c + j] : p.gw_b[c + j];
        const float b = p.bw_w ? p.bw_w[c + j] * t + p.bw_b[c + j] : p.bw_b[c + j];
        o[j] = sc * (g * ((v[j] - mean) * rstd) + b) + (p.resid ? r[j] : 0.f);
      }
      st8(p.out, p.out_dt, base + c, o);
    }
  } else {
    for (int c = lane; c < C; c += 64) {
      const float g = p.gw_w ? p.gw_w[c] * t + p.gw_b[c] : p.gw_b[c];
      const float b = p.bw_w ? p.bw_w[c] * t + p.bw_b[c] : p.bw_b[c];
      float o = sc * (g * ((ld1(p.x, p.x_dt, base + c) - mean) * rstd) + b);
      if (p.resid) o += ld1(p.resid, p.res_dt, base + c);
      st1(p.out, p.out_dt, base + c, o);
    }
  }
}

// Backward.  Block = (sample b, chunk of RPB rows); wave per row; per-lane partial sums of dgamma/dbeta for its
// columns are reduced over the block in LDS and flushed with 4 atomics per column:
//   dW_g[c] += t_b·Σ dout·xhat   db_g[c] += Σ dout·xhat   dW_b[c] += t_b·Σ dout   db_b[c] += Σ dout
// dx = rstd·(g - mean(g) - xhat·mean(g·xhat)),  g = dout·gamma   (identical to differentiating E[x^2]-mean^2).
constexpr int CLN_MAXC = 2048;  // columns handled per block pass (lane owns columns lane + 64*i, i < 32)
__global__ __launch_bounds__(256) void cln_bwd_kernel(ClnArgs p, int rpb, int chunks_per_sample) {
  __shared__ float sg[4][256], sb[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x / chunks_per_sample, chunk = blockIdx.x % chunks_per_sample;
  const int r0 = chunk * rpb, r1 = min(p.rows_per_sample, r0 + rpb);
  const int C = p.C;
  const float t = p.time ? p.time[b] : 0.f;
  const float sc = p.sscale ? p.sscale[b] : 1.f;
  // columns are processed in passes of 256 so that register arrays stay small and statically indexed
  for (int cb = 0; cb < C; cb += 256) {
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0 + wave; r < r1; r += 4) {
      const int row = b * p.rows_per_sample + r;
      const size_t base = (size_t)row * C;
      const float mean = p.mean[row], rstd = p.rstd[row];
      if (cb == 0 && p.mode != 2) {
        // full-row reductions m1 = mean(g), m2 = mean(g·xhat) and dx for ALL columns (done once, on the first pass)
        float m1 = 0.f, m2 = 0.f;
        for (int c = lane; c < C; c += 64) {
          const float gam = p.gw_w ? p.gw_w[c] * t + p.gw_b[c] : p.gw_b[c];
          const float g = sc * ld1(p.dout, p.dout_dt, base + c) * gam;
          const float xh = (ld1(p.x, p.x_dt, base + c) - mean) * rstd;
          m1 += g; m2 += g * xh;
        }
        m1 = wave_sum(m1) / C; m2 = wave_sum(m2) / C;
        for (int c = lane; c < C; c += 64) {
          const float gam = p.gw_w ? p.gw_w[c] * t + p.gw_b[c] : p.gw_b[c];
          const float g = sc * ld1(p.dout, p.dout_dt, base + c) * gam;
          const float xh = (ld1(p.x, p.x_dt, base + c) - mean) * rstd;
          st1(p.dx, p.dx_dt, base + c, rstd * (g - m1 - xh * m2));
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = cb + lane + 64 * i;
        if (c < C) {
          const float d = sc * ld1(p.dout, p.dout_dt, base + c);
          const float xh = (ld1(p.x, p.x_dt, base + c) - mean) * rstd;
          ag[i] += d * xh; ab[i] += d;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { sg[wave][lane + 64 * i] = ag[i]; sb[wave][lane + 64 * i] = ab[i]; }
    __syncthreads();
    const int c = cb + threadIdx.x;
    if (c < C && p.mode != 1) {
      const float dg = sg[0][threadIdx.x] + sg[1][threadIdx.x] + sg[2][threadIdx.x] + sg[3][threadIdx.x];
      const float db = sb[0][threadIdx.x] + sb[1][threadIdx.x] + sb[2][threadIdx.x] + sb[3][threadIdx.x];
      if (p.d_gw_w) { atomicAdd(&p.d_gw_w[c], t * dg); atomicAdd(&p.d_bw_w[c], t * db); }
      atomicAdd(&p.d_gw_b[c], dg);
      atomicAdd(&p.d_bw_b[c], db);
    }
    __syncthreads();
  }
}

extern "C" int scot_colsum(const void* x, int x_dt, const void* y, int y_dt, float* out, int M, int N, int ld, hipStream_t s);
extern "C" int scot_scale_residual(const void* y, int y_dt, const float* scale, const void* resid, int r_dt, void* out, int o_dt,
                                   size_t rows, int N, hipStream_t s);

// out2 (optional): second copy of the output in dtype out2_dt (the next GEMM's operand type)
extern "C" int scot_cln_fwd(const void* x, int x_dt, const void* resid, int res_dt, void* out, int out_dt, void* out2, int out2_dt,
                            float* mean, float* rstd, const float* time, const float* gw_w, const float* gw_b,
                            const float* bw_w, const float* bw_b, int rows, int rows_per_sample, int C, float eps,
                            const float* sample_scale, hipStream_t stream) {
  if (rows <= 0 || C <= 0 || rows_per_sample <= 0 || rows % rows_per_sample) return SCOT_ERR_SHAPE;
  if (!gw_b || !bw_b || (gw_w && !time)) return SCOT_ERR_SHAPE;
  {
    ClnFastArgs f{};
    f.x = x; f.resid = resid; f.out = out; f.out2 = out2; f.mean = mean; f.rstd = rstd; f.time = time;
    f.gw_w = gw_w; f.gw_b = gw_b; f.bw_w = bw_w; f.bw_b = bw_b; f.x_dt = x_dt; f.res_dt = res_dt; f.out_dt = out_dt; f.out2_dt = out2_dt;
    f.rows = rows; f.rows_per_sample = rows_per_sample; f.C = C; f.eps = eps; f.sscale = sample_scale;
    const int rc = scot_cln_fwd_fast(f, stream);
    if (rc != SCOT_ERR_UNSUPPORTED) return rc;
  }
  ClnArgs a{};
  a.x = x; a.resid = resid; a.out = out; a.mean = mean; a.rstd = rstd; a.time = time;
  a.gw_w = gw_w; a.gw_b = gw_b; a.bw_w = bw_w; a.bw_b = bw_b; a.x_dt = x_dt; a.res_dt = res_dt; a.out_dt = out_dt;
  a.rows = rows; a.rows_per_sample = rows_per_sample; a.C = C; a.eps = eps; a.sscale = sample_scale;
  a.vec = (C % 8 == 0) && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)resid) & 15) == 0;
  hipLaunchKernelGGL(cln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, a);
  int rc = scot_check_launch();
  if (rc == SCOT_OK && out2) rc = scot_scale_residual(out, out_dt, nullptr, nullptr, 0, out2, out2_dt, (size_t)rows, C, stream);
  return rc;
}

// d_xbias (optional): += Σ_rows dx  (bias gradient of the Linear that produced x)
extern "C" int scot_cln_bwd(const void* dout, int dout_dt, const void* x, int x_dt, const float* mean, const float* rstd,
                            const float* time, const float* gw_w, const float* gw_b, void* dx, int dx_dt,
                            float* d_gw_w, float* d_gw_b, float* d_bw_w, float* d_bw_b, float* d_xbias, int rows,
                            int rows_per_sample, int C, void* workspace, size_t ws_bytes, const float* sample_scale, int mode,
                            hipStream_t stream) {
  if (rows <= 0 || C <= 0 || rows_per_sample <= 0 || rows % rows_per_sample) return SCOT_ERR_SHAPE;
  if (mode < 0 || mode > 3 || (mode == 2 && d_xbias)) return SCOT_ERR_SHAPE;
  if (!gw_b || ((mode == 0 || mode == 2) && (!d_gw_b || !d_bw_b || (gw_w && (!d_gw_w || !d_bw_w)))) || (gw_w && !time)) return SCOT_ERR_SHAPE;
  if (mode == 3) {   // dx + per-block partials into `workspace` (scot_cln_bwd_workspace_bytes), finished by scot_cln_bwd_finish; fast path only
    ClnFastArgs f{};
    f.dout = dout; f.dout_dt = dout_dt; f.x = x; f.x_dt = x_dt; f.mean = (float*)mean; f.rstd = (float*)rstd; f.time = time;
    f.gw_w = gw_w; f.gw_b = gw_b; f.dx = dx; f.dx_dt = dx_dt; f.d_xbias = d_xbias;
    f.rows = rows; f.rows_per_sample = rows_per_sample; f.C = C; f.sscale = sample_scale; f.mode = 3;
    return scot_cln_bwd_fast(f, workspace, ws_bytes, stream);
  }
  {
    ClnFastArgs f{};
    f.dout = dout; f.dout_dt = dout_dt; f.x = x; f.x_dt = x_dt; f.mean = (float*)mean; f.rstd = (float*)rstd; f.time = time;
    f.gw_w = gw_w; f.gw_b = gw_b; f.dx = dx; f.dx_dt = dx_dt;
    f.d_gw_w = gw_w ? d_gw_w : nullptr; f.d_gw_b = d_gw_b; f.d_bw_w = gw_w ? d_bw_w : nullptr; f.d_bw_b = d_bw_b; f.d_xbias = d_xbias;
    f.rows = rows; f.rows_per_sample = rows_per_sample; f.C = C; f.sscale = sample_scale; f.mode = mode;
    const int rc = scot_cln_bwd_fast(f, workspace, ws_bytes, stream);
    if (rc != SCOT_ERR_UNSUPPORTED) return rc;
  }
  ClnArgs a{};
  a.dout = dout; a.dout_dt = dout_dt; a.x = x; a.x_dt = x_dt; a.mean = (float*)mean; a.rstd = (float*)rstd; a.time = time;
  a.gw_w = gw_w; a.gw_b = gw_b; a.dx = dx; a.dx_dt = dx_dt;
  a.d_gw_w = gw_w ? d_gw_w : nullptr; a.d_gw_b = d_gw_b; a.d_bw_w = gw_w ? d_bw_w : nullptr; a.d_bw_b = d_bw_b;
  a.rows = rows; a.rows_per_sample = rows_per_sample; a.C = C; a.sscale = sample_scale; a.mode = mode;
  const int rpb = rows_per_sample < 128 ? rows_per_sample : 128;
  const int cps = (rows_per_sample + rpb - 1) / rpb;
  hipLaunchKernelGGL(cln_bwd_kernel, dim3((rows / rows_per_sample) * cps), dim3(256), 0, stream, a, rpb, cps);
  int rc = scot_check_launch();
  if (rc == SCOT_OK && d_xbias && mode != 2) rc = scot_colsum(dx, dx_dt, nullptr, 0, d_xbias, rows, C, C, stream);
  return rc;
}

// include/scot_hip.h: scot_cln_bwd_workspace_bytes / scot_cln_bwd_finish (mode 3 of scot_cln_bwd)
int scot_cln_bwd_finish_launch(const float* partial, int nblk, int ncol, float* out, hipStream_t s);
extern "C" size_t scot_cln_bwd_workspace_bytes(int rows, int rows_per_sample, int C, int conditional) {
  int blocks, rpb;
  if (!scot_cln_bwd_partial_plan(rows, rows_per_sample, C, &blocks, &rpb)) return 0;
  return (size_t)blocks * (conditional ? 4 : 2) * C * sizeof(float);
}
extern "C" int scot_cln_bwd_finish(const void* partial, int rows, int rows_per_sample, int C, float* d_gw_w, float* d_gw_b,
                                   float* d_bw_w, float* d_bw_b, hipStream_t stream) {
  int blocks, rpb;
  if (!partial || !d_gw_b || !d_bw_b || (d_gw_w == nullptr) != (d_bw_w == nullptr)) return SCOT_ERR_SHAPE;
  if (!scot_cln_bwd_partial_plan(rows, rows_per_sample, C, &blocks, &rpb)) return SCOT_ERR_UNSUPPORTED;
  if (d_gw_w) {
    // [t·dγ | dγ | t·dβ | dβ] lands on [weight.weight | weight.bias | bias.weight | bias.bias]: contiguous in the parameter arena
    // (C % 64 == 0: every tensor starts on a 64-float boundary); four launches when a caller keeps them apart
    if (d_gw_b == d_gw_w + C && d_bw_w == d_gw_w + 2 * C && d_bw_b == d_gw_w + 3 * C)
      return scot_cln_bwd_finish_launch((const float*)partial, blocks, 4 * C, d_gw_w, stream);
    return SCOT_ERR_UNSUPPORTED;
  }
  if (d_bw_b == d_gw_b + C) return scot_cln_bwd_finish_launch((const float*)partial, blocks, 2 * C, d_gw_b, stream);
  return SCOT_ERR_UNSUPPORTED;
}
