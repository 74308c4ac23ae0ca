// wgrad_mlp — the fc1 / fc2 weight and bias gradients of a ScOTLayer's MLP at the token-heavy stages (C = 96 / 192), computed
// WITHOUT the 4C-wide tensors ever existing in HBM (autograd of HF modeling_swinv2.py:545-548, 558-561; reference model.py:572-574):
//
//     dW1[hid, C] += du^T · h16        db1[hid] += Σ_rows du          du  = (dz · W2) ⊙ gelu'(u),   u = h16 · W1^T + b1
//     dW2[C, hid] += dz^T · gelu(u)    db2[C]   += Σ_rows dz          dz  = the gradient wrt the MLP's pre-norm output (16-bit)
//
// Round 2 stored gelu(u), gelu'(u) (forward) and du (backward) — 3 x 768 B per token at C = 96, a third of a stage-0 layer's HBM
// traffic — only so that the grouped weight-gradient GEMM could read them back.  Here a workgroup owns a chunk of HW hidden units and a
// slice of the tokens: W1[chunk] and W2^T[chunk] sit in LDS for the whole kernel, the token rows of h16 and dz stream through a 32-row
// LDS tile, and per 32 tokens the workgroup RECOMPUTES u and dz·W2 for its hidden units (2·32·HW·C MACs each: the MFMA pipe idles in
// the kernel this replaces) and feeds gelu(u) / du straight from the accumulator registers into the two gradient products:
//   * U = h·W1c^T and D = dz·W2c (A = token rows, B = hidden rows, both K-contiguous over C) leave lane (g, lc) holding hidden unit lc and
//     tokens 4g..4g+3 of each of the two 16-token tiles — exactly an MFMA operand whose 8 contraction elements are the tokens
//     k(g, j) = {4g+j | 16+4g+(j-4)}: any k-bijection is legal as long as both operands agree (common.h), and the other operand (h rows /
//     dz rows with the contraction over TOKENS) is read from the LDS tile with the transposing read at rows (4g.., 16+4g..);
//   * so gelu(u) (B operand of dW2 = dz^T·act) and du (A operand of dW1 = du^T·h) never leave registers.
// HBM traffic: h16 + dz = 4·C bytes per token (the hidden chunks of one token slice run on ONE XCD: its L2 serves the re-reads),
// against 20·C for the four operands of the GEMMs it replaces; partial sums [slice][W1 | b1 | W2 | b2] in the parameter arena's own
// order, added into the gradient arena by one flat reduce.
#include "wgrad_group.h"      // grad_commit8 / SCOT_GRAD_*: how a weight gradient meets the arena
#include <stdlib.h>

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct WgradMlpArgs {
  const bf16_t* h16; const bf16_t* dz;       // [M, C]
  const bf16_t* W1; const float* b1;         // [hid, C], [hid]
  const bf16_t* W2T;                         // [hid, C]  (= W2^T: the transposed 16-bit copy the data gradients already use)
  float* ws;                                 // [nslice][plane]
  size_t plane;                              // floats per slice = hid·C + hid + C·hid + C
  int M, hid, rows_per_slice, nslice, use_tr;   // nslice: slices that hold rows; the launch may be padded to a multiple of 8 (XCD grouping)
  int nslice_launch;
};

template <int C, int HT> struct WgradMlpLds {
  static constexpr int HW = 64 * HT, PW = C + 8, TK = 32;
  static constexpr size_t bytes = (size_t)(2 * HW * PW + 2 * TK * PW) * 2;
};

template <int C, int HT>
__global__ __launch_bounds__(256, 2) void wgrad_mlp_kernel(WgradMlpArgs p) {
  constexpr int KJ = C / 32, CT = C / 16, HW = 64 * HT, PW = C + 8, TK = 32;
  constexpr int NW = HW * C / 8, PWN = NW / 256;            // 16-byte pieces of one weight chunk per thread
  constexpr int NT = TK * C / 8, PTN = (NT + 255) / 256;    // ... of one token tile
  static_assert(NW % 256 == 0, "weight chunk pieces must divide over the 256 threads");
  __shared__ __attribute__((aligned(16))) char smem[WgradMlpLds<C, HT>::bytes];
  bf16_t* W1c = (bf16_t*)smem;               // [HW][PW]  rows = hidden units of the chunk, K-contiguous over C
  bf16_t* W2c = W1c + HW * PW;               // [HW][PW]  rows of W2^T
  bf16_t* Ht = W2c + HW * PW;                // [TK][PW]  token rows of h16
  bf16_t* Dt = Ht + TK * PW;                 // [TK][PW]  token rows of dz
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lc = lane & 15;
  const int hid = p.hid, nchunk = hid / HW;
  // the chunks of one token slice are consecutive ON ONE XCD (workgroup b runs on XCD b % 8: speed only)
  int L = blockIdx.x, slice, chunk;
  if (p.nslice_launch % 8 == 0) { const int j = L >> 3; slice = (L & 7) * (p.nslice_launch >> 3) + j / nchunk; chunk = j % nchunk; }
  else { slice = L / nchunk; chunk = L % nchunk; }
  if (slice >= p.nslice) return;              // padding of the launch (whole workgroup: no barrier is left waiting)
  const int row_beg = slice * p.rows_per_slice, row_end = min(p.M, row_beg + p.rows_per_slice);

  // ---- the chunk's weights: global -> LDS, once
  {
    const bf16_t* s1 = p.W1 + (size_t)chunk * HW * C;
    const bf16_t* s2 = p.W2T + (size_t)chunk * HW * C;
    u32x4_t r1[PWN], r2[PWN];
#pragma unroll
    for (int u = 0; u < PWN; ++u) { r1[u] = *(const u32x4_t*)(s1 + (size_t)(tid + u * 256) * 8); r2[u] = *(const u32x4_t*)(s2 + (size_t)(tid + u * 256) * 8); }
#pragma unroll
    for (int u = 0; u < PWN; ++u) {
      const int i = tid + u * 256;
      *(u32x4_t*)(W1c + (i / (C / 8)) * PW + (i % (C / 8)) * 8) = r1[u];
      *(u32x4_t*)(W2c + (i / (C / 8)) * PW + (i % (C / 8)) * 8) = r2[u];
    }
  }
  float bias[HT];
#pragma unroll
  for (int hh = 0; hh < HT; ++hh) bias[hh] = p.b1[chunk * HW + (wave * HT + hh) * 16 + lc];

  f32x4_t aW1[HT][CT], aW2[HT][CT];
  float sdu[HT];
#pragma unroll
  for (int hh = 0; hh < HT; ++hh) {
    sdu[hh] = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { aW1[hh][ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; aW2[hh][ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
  }
  float sdz = 0.f;                            // chunk 0: thread tid < C owns column tid of Σ_rows dz

  // token tile: this thread's 16-byte pieces of the next 32 rows, in flight while the current rows are multiplied
  u32x4_t rh[PTN], rd[PTN];
  auto load_rows = [&](int r0) {
#pragma unroll
    for (int u = 0; u < PTN; ++u) {
      const int i = min(tid + u * 256, NT - 1);
      const int row = min(r0 + i / (C / 8), p.M - 1);                 // clamped: rows past the slice are zeroed at the LDS store
      const size_t o = (size_t)row * C + (i % (C / 8)) * 8;
      rh[u] = *(const u32x4_t*)(p.h16 + o);
      rd[u] = *(const u32x4_t*)(p.dz + o);
    }
  };
  auto store_rows = [&](int r0) {
#pragma unroll
    for (int u = 0; u < PTN; ++u) {
      const int i = tid + u * 256;
      if (i < NT) {
        const int tr = i / (C / 8);
        const bool ok = r0 + tr < row_end;
        const u32x4_t z = {0u, 0u, 0u, 0u};
        *(u32x4_t*)(Ht + tr * PW + (i % (C / 8)) * 8) = ok ? rh[u] : z;
        *(u32x4_t*)(Dt + tr * PW + (i % (C / 8)) * 8) = ok ? rd[u] : z;
      }
    }
  };
  // every workgroup runs the same number of steps (a slice that starts past M multiplies zero rows): barriers stay uniform
  const int nsteps = (p.rows_per_slice + TK - 1) / TK;
  load_rows(row_beg);
  for (int st = 0; st < nsteps; ++st) {
    const int r0 = row_beg + st * TK;
    __syncthreads();                          // the previous tile's readers are done (first pass: the weight chunk is stored)
    store_rows(r0);
    __syncthreads();
    load_rows(r0 + TK);                       // (clamped addresses: the last prefetch re-reads valid rows and is never stored)
    if (chunk == 0 && tid < C) {
#pragma unroll 8
      for (int t = 0; t < TK; ++t) sdz += bf2f(Dt[t * PW + tid]);
    }
    // recomputation for the wave's HT hidden tiles at once: every token-row fragment read from LDS feeds HT products
    f32x4_t U[HT][2], D[HT][2];
#pragma unroll
    for (int hh = 0; hh < HT; ++hh)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) { U[hh][tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; D[hh][tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
      Frag<bf16_t> w1[HT], w2[HT];
#pragma unroll
      for (int hh = 0; hh < HT; ++hh) {
        w1[hh] = lds_frag_kc(W1c, PW, (wave * HT + hh) * 16, j * 32, lane);
        w2[hh] = lds_frag_kc(W2c, PW, (wave * HT + hh) * 16, j * 32, lane);
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const Frag<bf16_t> hA = lds_frag_kc(Ht, PW, tt * 16, j * 32, lane), dA = lds_frag_kc(Dt, PW, tt * 16, j * 32, lane);
#pragma unroll
        for (int hh = 0; hh < HT; ++hh) { mma16(U[hh][tt], hA, w1[hh]); mma16(D[hh][tt], dA, w2[hh]); }
      }
    }
    // lane (g, lc): hidden unit h0 + lc, tokens tt·16 + 4g + r
    Frag<bf16_t> actf[HT], duf[HT];
#pragma unroll
    for (int hh = 0; hh < HT; ++hh) {
      float av[8], dv[8];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = U[hh][tt][r] + bias[hh];
          float cdf, e;
          gelu_terms(x, cdf, e);
          av[4 * tt + r] = x * cdf;
          // gelu'(u) rounded to 16 bits first, as scot_block_tail_bwd uses it: the du below is the du the data gradient was taken from
          dv[4 * tt + r] = D[hh][tt][r] * bf2f(f2bf(cdf + x * 0.3989422804014327f * e));
        }
      actf[hh] = frag_from_f32<bf16_t>(av);
      duf[hh] = frag_from_f32<bf16_t>(dv);
      // db1 from the ROUNDED du (what the GEMM path summed: the 16-bit du tile)
#pragma unroll
      for (int j = 0; j < 8; ++j) sdu[hh] += bf2f((bf16_t)duf[hh].v[j]);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      // contraction over the 32 tokens: element j < 4 <-> token 4g + j, j >= 4 <-> token 16 + 4g + (j - 4)
      const Frag<bf16_t> hB = lds_frag_ks(Ht, PW, ct * 16, g * 4, 16 + g * 4, lane, p.use_tr);
      const Frag<bf16_t> dA = lds_frag_ks(Dt, PW, ct * 16, g * 4, 16 + g * 4, lane, p.use_tr);
#pragma unroll
      for (int hh = 0; hh < HT; ++hh) {
        mma16(aW1[hh][ct], duf[hh], hB);      // dW1[hidden h0 + 4g + r][channel 16 ct + lc]
        mma16(aW2[hh][ct], dA, actf[hh]);     // dW2[channel 16 ct + 4g + r][hidden h0 + lc]
      }
    }
  }

  // ---- partial sums of this (slice, chunk) into the slice's plane, laid out like the parameter arena: [W1 | b1 | W2 | b2]
  float* pl = p.ws + (size_t)slice * p.plane;
  const size_t oW1 = 0, ob1 = (size_t)hid * C, oW2 = ob1 + hid, ob2 = oW2 + (size_t)C * hid;
#pragma unroll
  for (int hh = 0; hh < HT; ++hh) {
    const int hg = chunk * HW + (wave * HT + hh) * 16;               // first hidden unit of the tile, global
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pl[oW1 + (size_t)(hg + g * 4 + r) * C + ct * 16 + lc] = aW1[hh][ct][r];
        pl[oW2 + (size_t)(ct * 16 + g * 4 + r) * hid + hg + lc] = aW2[hh][ct][r];
      }
    float s = sdu[hh];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) pl[ob1 + hg + lc] = s;
  }
  if (chunk == 0 && tid < C) pl[ob2 + tid] = sdz;
}

// grad[i] += Σ_z ws[z·plane + i]   (i < plane, plane % 8 == 0; ZL lanes share the slices of one 8-float group).  The two weight matrices
// ([0, wlen) and [w2beg, w2beg + wlen) of the plane) meet the arena according to `mode` / `scale` (WgradProblem::mode); the two bias
// vectors between and behind them are always plain accumulations (they belong to the zero-filled part of the gradient arena).
template <int ZL>
__global__ __launch_bounds__(256) void plane_reduce_kernel(const float* __restrict__ ws, float* __restrict__ grad, size_t plane, int nz,
                                                           size_t wlen, size_t w2beg, int mode, const float* scale) {
  const size_t n8 = plane / 8;
  const int zl = threadIdx.x % ZL;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / ZL; i < n8; i += (size_t)gridDim.x * blockDim.x / ZL) {
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int z = zl; z < nz; z += ZL) {
      float v[8];
      ld8(ws + (size_t)z * plane, SCOT_F32, i * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
#pragma unroll
    for (int o = 1; o < ZL; o <<= 1)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
    if (zl == 0) {
      const size_t e = i * 8;
      const bool weight = e < wlen || (e >= w2beg && e < w2beg + wlen);
      grad_commit8(grad, e, acc, weight ? mode : SCOT_GRAD_ADD, scale);
    }
  }
}

extern int g_scot_use_tr;

// slices of `rows_per_slice` tokens (a multiple of 32) so that chunks x slices ~ the requested number of workgroups
static void wgrad_mlp_plan(int M, int C, int hid, int* nslice, int* rps) {
  // workgroups wanted: two per CU at C = 192 (57 vs 85 us alone: twelve chunks of 64 hidden units leave 21 slices at 256), one at C = 96
  const int want = C == 96 ? 256 : 512;     // (round 3: 512 at C = 96 costs the step +0.1 ms)
  const int hw = C == 96 ? 128 : 64, nchunk = hid / hw;
  int ns = (want + nchunk - 1) / nchunk;
  if (ns < 1) ns = 1;
  if (ns >= 8) ns = ns / 8 * 8;
  int r = ((M + ns - 1) / ns + 31) / 32 * 32;
  if (r < 256) r = 256;                         // at least 8 steps per workgroup: the weight chunk load is amortised
  if (r > ((M + 31) / 32) * 32) r = ((M + 31) / 32) * 32;
  *rps = r;
  *nslice = (M + r - 1) / r;
}
static int round8(int n) { return n >= 8 ? (n + 7) / 8 * 8 : n; }

// include/scot_hip.h: scot_wgrad_mlp_workspace_bytes / scot_wgrad_mlp
extern "C" size_t scot_wgrad_mlp_workspace_bytes(int M, int C, int hid) {
  if (M <= 0 || (C != 96 && C != 192) || hid != 4 * C) return 0;
  int ns, rps;
  wgrad_mlp_plan(M, C, hid, &ns, &rps);
  return (size_t)ns * ((size_t)2 * hid * C + hid + C) * sizeof(float);
}

// dW1 / db1 / dW2 / db2 must be CONTIGUOUS in this order (they are in the gradient arena: intermediate.dense.weight, .bias,
// output.dense.weight, .bias): the partial planes are laid out the same way and one flat pass adds them in.
extern "C" int scot_wgrad_mlp(const void* h16, const void* dz, const void* W1, const float* b1, const void* W2T, float* dW1, float* db1,
                              float* dW2, float* db2, int M, int C, int hid, void* workspace, size_t ws_bytes, int mode,
                              const float* grad_scale, hipStream_t stream) {
  if (M <= 0 || mode < 0 || mode > SCOT_GRAD_ADD_SCALED) return SCOT_ERR_SHAPE;
  if ((C != 96 && C != 192) || hid != 4 * C) return SCOT_ERR_UNSUPPORTED;
  if (!h16 || !dz || !W1 || !b1 || !W2T || !dW1 || !db1 || !dW2 || !db2 || !workspace) return SCOT_ERR_SHAPE;
  if (db1 != dW1 + (size_t)hid * C || dW2 != db1 + hid || db2 != dW2 + (size_t)C * hid) return SCOT_ERR_UNSUPPORTED;
  if ((((uintptr_t)h16 | (uintptr_t)dz | (uintptr_t)W1 | (uintptr_t)W2T | (uintptr_t)workspace | (uintptr_t)dW1) & 31) != 0) return SCOT_ERR_SHAPE;
  WgradMlpArgs a;
  a.h16 = (const bf16_t*)h16; a.dz = (const bf16_t*)dz; a.W1 = (const bf16_t*)W1; a.b1 = b1; a.W2T = (const bf16_t*)W2T;
  a.ws = (float*)workspace; a.M = M; a.hid = hid; a.use_tr = g_scot_use_tr;
  a.plane = (size_t)2 * hid * C + hid + C;
  wgrad_mlp_plan(M, C, hid, &a.nslice, &a.rows_per_slice);
  if ((size_t)a.nslice * a.plane * sizeof(float) > ws_bytes) return SCOT_ERR_SHAPE;
  const int nchunk = hid / (C == 96 ? 128 : 64);
  a.nslice_launch = round8(a.nslice);
  const dim3 grid((unsigned)(nchunk * a.nslice_launch)), block(256);
  if (C == 96) hipLaunchKernelGGL((wgrad_mlp_kernel<96, 2>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((wgrad_mlp_kernel<192, 1>), grid, block, 0, stream, a);
  int rc = scot_check_launch();
  if (rc != SCOT_OK) return rc;
  const size_t n8 = a.plane / 8;
  const int zl = a.nslice >= 32 ? 8 : a.nslice >= 4 ? 4 : 1;
  size_t blocks = (n8 * zl + 255) / 256; if (blocks > 2048) blocks = 2048;
  const dim3 gr((unsigned)blocks), b(256);
  const size_t wlen = (size_t)hid * C, w2beg = wlen + hid;
  if (zl == 8) hipLaunchKernelGGL(plane_reduce_kernel<8>, gr, b, 0, stream, a.ws, dW1, a.plane, a.nslice, wlen, w2beg, mode, grad_scale);
  else if (zl == 4) hipLaunchKernelGGL(plane_reduce_kernel<4>, gr, b, 0, stream, a.ws, dW1, a.plane, a.nslice, wlen, w2beg, mode, grad_scale);
  else hipLaunchKernelGGL(plane_reduce_kernel<1>, gr, b, 0, stream, a.ws, dW1, a.plane, a.nslice, wlen, w2beg, mode, grad_scale);
  return scot_check_launch();
}
