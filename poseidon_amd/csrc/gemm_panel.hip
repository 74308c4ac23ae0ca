// gemm_panel — weights-resident, barrier-free GEMM for the tall-skinny contractions of scOT stages 0/1
// (M = tokens = 16k..65k, K = 96..384, N = 96..768):  C[M,N] = A[M,K] · W^T   (NT: W[N,K])   or   A · W   (NN: W[K,N]).
//
// Why a second GEMM kernel: rocprof/PMC (round 1) showed the tiled kernel latency-bound at these shapes — a 64x64 tile has
// only 2–6 K-tiles, so its waves spend 8.5 k cycles (47 % parked on barriers / first-tile HBM latency) to issue ~550
// instructions and 8–24 MFMAs.  Here
//   * the workgroup's W panel [BN x K] is loaded into LDS ONCE and reused for every row group the workgroup processes;
//   * each WAVE is autonomous: it walks 32-row groups, reads its A fragments straight from HBM into MFMA operand
//     registers (16 B per lane, the 96-wide K chunk after next is always in flight), multiplies against the resident
//     panel (36 MFMAs per chunk, no __syncthreads in the loop), and
//   * transposes its 16x32 accumulator blocks through a private 2.3 KB LDS patch so that the fused epilogue
//     (bias, gelu value+derivative, gelu'/aux multiply, residual) stores 64–128-byte row segments.
// bf16 operands only (the fp32 parity mode keeps the tiled kernel).  K must be a multiple of 96, N of BN.
#include "common.h"
#include <stdlib.h>

#define LAYOUT_NT 0
#define LAYOUT_NN 1

struct PanelArgs {
  const bf16_t* A; const bf16_t* B; void* C; void* C2;
  const float* bias; const void* aux; const void* resid;
  int M, N, K;
  int lda, ldb, ldc, ldaux, ldres;
  int c_dt, aux_dt, res_dt;
  int aux_on, aux_mul;
  int use_tr;
};

template <int NF, int LAYOUT>
__global__ __launch_bounds__(256, 2) void gemm_panel_kernel(PanelArgs p) {
  constexpr int BN = NF * 16, MG = 2, CP = 36;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int K = p.K;
  const int pitch = (LAYOUT == LAYOUT_NT) ? K + 8 : BN + 8;        // elements
  bf16_t* Wp = (bf16_t*)smem;
  const size_t panel_bytes = (LAYOUT == LAYOUT_NT) ? (size_t)BN * pitch * 2 : (size_t)K * pitch * 2;
  float* stage = (float*)(smem + ((panel_bytes + 15) & ~(size_t)15));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, lc = lane & 15;
  const int n0 = blockIdx.x * BN;

  // ---- W panel -> LDS (once per workgroup)
  if (LAYOUT == LAYOUT_NT) {
    const int cpr = K / 8;                       // 16-byte chunks per row
    for (int c = tid; c < BN * cpr; c += 256) {
      const int r = c / cpr, k8 = (c % cpr) * 8;
      *(uint4*)(Wp + r * pitch + k8) = *(const uint4*)(p.B + (size_t)(n0 + r) * p.ldb + k8);
    }
  } else {
    constexpr int cpr = BN / 8;
    for (int c = tid; c < K * cpr; c += 256) {
      const int k = c / cpr, c8 = (c % cpr) * 8;
      *(uint4*)(Wp + k * pitch + c8) = *(const uint4*)(p.B + (size_t)k * p.ldb + n0 + c8);
    }
  }
  __syncthreads();

  const int ngroups = (p.M + 32 * 1 - 1) / 32;
  const int nwaves = gridDim.y * 4;
  const int nkc = K / 96;
  float* Ct = stage + wave * 16 * CP;

  uint4 cur[MG][3], nxt[MG][3];
  auto load_chunk = [&](uint4 (&dst)[MG][3], int gi, int kc) {
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
      const int row = min(gi * 32 + mg * 16 + lc, p.M - 1);
      const bf16_t* src = p.A + (size_t)row * p.lda + kc * 96 + g * 8;
#pragma unroll
      for (int j = 0; j < 3; ++j) dst[mg][j] = *(const uint4*)(src + j * 32);
    }
  };

  int gi = blockIdx.y * 4 + wave;
  if (gi < ngroups) load_chunk(cur, gi, 0);
  while (gi < ngroups) {
    f32x4_t acc[MG][NF];
#pragma unroll
    for (int mg = 0; mg < MG; ++mg)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[mg][nf] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < nkc; ++kc) {
      const bool more = kc + 1 < nkc;
      const int ngi = more ? gi : gi + nwaves, nkcn = more ? kc + 1 : 0;
      if (ngi < ngroups) load_chunk(nxt, ngi, nkcn);     // the chunk after this one is in flight while we multiply
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int k0 = kc * 96 + j * 32;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          Frag<bf16_t> b;
          if (LAYOUT == LAYOUT_NT) b = lds_frag_kc(Wp, pitch, nf * 16, k0, lane);
          else b = lds_frag_ks(Wp, pitch, nf * 16, k0 + g * 8, k0 + g * 8 + 4, lane, p.use_tr);
#pragma unroll
          for (int mg = 0; mg < MG; ++mg) {
            Frag<bf16_t> a;
            a.v = __builtin_bit_cast(s16x8_t, cur[mg][j]);
            mma16(acc[mg][nf], a, b);
          }
        }
      }
#pragma unroll
      for (int mg = 0; mg < MG; ++mg)
#pragma unroll
        for (int j = 0; j < 3; ++j) cur[mg][j] = nxt[mg][j];
    }

    // ---- epilogue: 16x32 blocks through the wave's private LDS patch -> row-segment stores
    const int row = lane >> 2, c8 = (lane & 3) * 8;
#pragma unroll
    for (int mg = 0; mg < MG; ++mg) {
      const int grow = gi * 32 + mg * 16 + row;
#pragma unroll
      for (int pp = 0; pp < (NF + 1) / 2; ++pp) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          if (2 * pp + f < NF) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Ct[(g * 4 + r) * CP + f * 16 + lc] = acc[mg][2 * pp + f][r];
          }
        }
        __builtin_amdgcn_wave_barrier();
        const bool cvalid = (2 * pp) * 16 + c8 < BN;   // NF odd: the last pass carries one fragment (16 columns)
        const int col = n0 + pp * 32 + c8;
        if (grow < p.M && cvalid) {
          float v[8];
          const float4 x0 = *(const float4*)(Ct + row * CP + c8), x1 = *(const float4*)(Ct + row * CP + c8 + 4);
          v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
          if (p.bias) {
            float bb[8];
            ld8(p.bias, SCOT_F32, col, bb);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += bb[j];
          }
          if (p.aux_on) {
            float x[8];
            ld8(p.aux, p.aux_dt, (size_t)grow * p.ldaux + col, x);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= p.aux_mul ? x[j] : gelu_grad_f(x[j]);
          }
          if (p.resid) {
            float x[8];
            ld8(p.resid, p.res_dt, (size_t)grow * p.ldres + col, x);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += x[j];
          }
          const size_t ci = (size_t)grow * p.ldc + col;
          if (p.C2) {
            float gv[8], gd[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { float cdf, e; gelu_terms(v[j], cdf, e); gv[j] = v[j] * cdf; gd[j] = cdf + v[j] * 0.3989422804014327f * e; }
            st8(p.C, p.c_dt, ci, gv);
            if (p.C2 != p.C) st8(p.C2, p.c_dt, ci, gd);
          } else {
            st8(p.C, p.c_dt, ci, v);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    gi += nwaves;
  }
}

extern int g_scot_use_tr;

template <int NF>
static int launch_panel(const PanelArgs& a, int layout, hipStream_t s) {
  constexpr int BN = NF * 16;
  const int pitch = layout == LAYOUT_NT ? a.K + 8 : BN + 8;
  size_t panel = layout == LAYOUT_NT ? (size_t)BN * pitch * 2 : (size_t)a.K * pitch * 2;
  panel = (panel + 15) & ~(size_t)15;
  const size_t sh = panel + 4 * 16 * 36 * sizeof(float);
  const int ntn = a.N / BN;
  const int ngroups = (a.M + 31) / 32;
  int gy = 768 / ntn;
  if (gy < 1) gy = 1;
  if (gy > (ngroups + 3) / 4) gy = (ngroups + 3) / 4;
  dim3 grid(ntn, gy), block(256);
  if (layout == LAYOUT_NT) {
    if (sh > 64 * 1024) (void)hipFuncSetAttribute((const void*)gemm_panel_kernel<NF, LAYOUT_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL((gemm_panel_kernel<NF, LAYOUT_NT>), grid, block, sh, s, a);
  } else {
    if (sh > 64 * 1024) (void)hipFuncSetAttribute((const void*)gemm_panel_kernel<NF, LAYOUT_NN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL((gemm_panel_kernel<NF, LAYOUT_NN>), grid, block, sh, s, a);
  }
  return scot_check_launch();
}

// Returns SCOT_ERR_UNSUPPORTED when the call does not qualify (the caller then uses the tiled kernels).
int scot_gemm_panel(int layout, int compute, int M, int N, int K, const void* A, int a_dt, int lda, int a_gelu, const void* B,
                    int b_dt, int ldb, int b_gelu, void* C, int c_dt, int ldc, const float* bias, const float* colscale,
                    const void* aux, int aux_dt, int ldaux, const void* resid, int res_dt, int ldres, int accumulate,
                    float* colsum_out, int aux_mul, void* C2, hipStream_t stream) {
  if (compute != SCOT_BF16 || (layout != LAYOUT_NT && layout != LAYOUT_NN)) return SCOT_ERR_UNSUPPORTED;
  if (a_dt != SCOT_BF16 || b_dt != SCOT_BF16 || a_gelu || b_gelu || colscale || colsum_out) return SCOT_ERR_UNSUPPORTED;
  const int kmax = 192;   // measured: K = 384 (48-col panels) only ties the tiled kernel
  if (K % 96 || K > kmax || N % 48 || M < 4096) return SCOT_ERR_UNSUPPORTED;
  if ((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)C2 | (uintptr_t)aux | (uintptr_t)resid | (uintptr_t)bias) & 15) != 0)
    return SCOT_ERR_UNSUPPORTED;
  if (lda % 8 || ldb % 8 || ldc % 8 || (aux && ldaux % 8) || (resid && ldres % 8)) return SCOT_ERR_UNSUPPORTED;
  PanelArgs a;
  a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.C2 = C2; a.bias = bias; a.aux = aux; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux; a.ldres = ldres;
  a.c_dt = c_dt; a.aux_dt = aux_dt; a.res_dt = res_dt; a.aux_on = aux != nullptr; a.aux_mul = aux_mul; a.use_tr = g_scot_use_tr;
  if (accumulate) {
    if (resid) return SCOT_ERR_UNSUPPORTED;
    a.resid = C; a.res_dt = c_dt; a.ldres = ldc;
  }
  const bool wide = (N % 96 == 0) && K <= 192;   // 96-column panels while the panel stays <= 40 KB (3 workgroups per CU)
  return wide ? launch_panel<6>(a, layout, stream) : launch_panel<3>(a, layout, stream);
}
