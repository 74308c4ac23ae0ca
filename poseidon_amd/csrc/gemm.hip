// scot_gemm — the dense contractions of the scOT hot path (52 % MLP + 19.5 % QKV + 6.5 % out-proj + ConvNeXt
// pointwise + merge/unmerge + patch embed/recovery) and their dgrad / wgrad forms.
//
//   layout NT : C[M,N] = A[M,K] · B[N,K]^T     forward of every nn.Linear (weights are [out,in]; reference
//                                              HF:545-561, HF:389-410, model.py:709,747,760)
//   layout NN : C[M,N] = A[M,K] · B[K,N]       dgrad (dX = dY · W) and the ConvTranspose2d of model.py:616-621
//   layout TN : C[M,N] += A[K,M]^T · B[K,N]    wgrad (dW = dY^T · X), split over K with fp32 atomics
//
// One workgroup = 256 threads = 4 waves (2x2) computing a BMxBN tile with 16x16 MFMA fragments, BK = 32.
// Operands are staged HBM → registers (coalesced 16-byte loads along the contiguous dimension, optional GELU,
// convert to the compute type) → LDS.  An operand whose contiguous dimension is NOT the contraction (B of NN,
// both of TN) stays in its source orientation in LDS and is read with the transposing fragment read
// (ds_read_b64_tr_b16 for bf16) — no transposed copies of activations or weights are ever written to HBM.
// Epilogue (fused): + bias[n], * colscale[n], * gelu'(aux[m,n]), + resid[m,n], store f32/bf16 or atomicAdd.
#include "common.h"
#include <stdlib.h>

#define LAYOUT_NT 0
#define LAYOUT_NN 1
#define LAYOUT_TN 2

struct GemmArgs {
  const void* A; const void* B; void* C;
  const float* bias; const float* colscale; const void* aux; const void* resid;
  int M, N, K;
  int lda, ldb, ldc, ldaux, ldres;
  int a_dt, b_dt, c_dt, aux_dt, res_dt;
  int a_gelu, b_gelu, aux_gelu_grad, atomic;
  int ksplit;   // K elements per blockIdx.z (multiple of 32)
  int a_vec, b_vec;  // 16-byte vector loads legal
  int use_tr;
  void* C2; int aux_mul;
};

constexpr int BK = 32;

template <typename CT, int R, bool KC> struct TileShape {
  // KC: [R][BK+kpad]  else: [BK][R+8]
  static constexpr int pitch = KC ? (BK + ct_traits<CT>::kpad) : (R + 8);
  static constexpr int elems = KC ? R * pitch : BK * pitch;
  static constexpr int nchunk = (R * (BK / 8) + 255) / 256;  // 8-element chunks per thread
};

// Load this thread's chunks of one operand tile into registers (as f32).
template <int R, bool KC>
__device__ __forceinline__ void stage_load(float (&st)[(R * 4 + 255) / 256][8], const void* src, int dt, int ld, int vec,
                                           int row0, int rmax, int k0, int kend, int tid, bool gelu) {
  constexpr int NCH = (R * 4 + 255) / 256;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * 256;
    float* v = st[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (c < R * 4) {
      if (KC) {
        const int row = row0 + (c >> 2), k = k0 + (c & 3) * 8;
        if (row < rmax && k < kend) {
          const size_t idx = (size_t)row * ld + k;
          if (vec && k + 8 <= kend) ld8(src, dt, idx, v);
          else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (k + j < kend) v[j] = ld1(src, dt, idx + j);
          }
        }
      } else {
        constexpr int CPR = R / 8;  // chunks per k-row
        const int k = k0 + c / CPR, r = row0 + (c % CPR) * 8;
        if (k < kend && r < rmax) {
          const size_t idx = (size_t)k * ld + r;
          if (vec && r + 8 <= rmax) ld8(src, dt, idx, v);
          else {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (r + j < rmax) v[j] = ld1(src, dt, idx + j);
          }
        }
      }
      if (gelu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
      }
    }
  }
}

template <typename CT> __device__ __forceinline__ void lds_store8(CT* p, const float v[8]);
template <> __device__ __forceinline__ void lds_store8<float>(float* p, const float v[8]) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void lds_store8<bf16_t>(bf16_t* p, const float v[8]) {
  *(uint4*)p = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}

template <typename CT, int R, bool KC>
__device__ __forceinline__ void stage_store(CT* tile, const float (&st)[(R * 4 + 255) / 256][8], int tid) {
  constexpr int NCH = (R * 4 + 255) / 256;
  constexpr int pitch = TileShape<CT, R, KC>::pitch;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * 256;
    if (c < R * 4) {
      if (KC) lds_store8<CT>(tile + (c >> 2) * pitch + (c & 3) * 8, st[i]);
      else { constexpr int CPR = R / 8; lds_store8<CT>(tile + (c / CPR) * pitch + (c % CPR) * 8, st[i]); }
    }
  }
}

template <typename CT, int BM, int BN, int LAYOUT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  constexpr bool A_KC = (LAYOUT != LAYOUT_TN);
  constexpr bool B_KC = (LAYOUT == LAYOUT_NT);
  using TA = TileShape<CT, BM, A_KC>;
  using TB = TileShape<CT, BN, B_KC>;
  constexpr int MI = BM / 32, NI = BN / 32;
  __shared__ __attribute__((aligned(16))) CT lds[TA::elems + TB::elems];
  CT* As = lds;
  CT* Bs = lds + TA::elems;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * p.ksplit;
  const int kend = min(p.K, kbeg + p.ksplit);

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  float sa[TA::nchunk][8], sb[TB::nchunk][8];
  stage_load<BM, A_KC>(sa, p.A, p.a_dt, p.lda, p.a_vec, m0, p.M, kbeg, kend, tid, p.a_gelu != 0);
  stage_load<BN, B_KC>(sb, p.B, p.b_dt, p.ldb, p.b_vec, n0, p.N, kbeg, kend, tid, p.b_gelu != 0);

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile's fragment reads are done
    stage_store<CT, BM, A_KC>(As, sa, tid);
    stage_store<CT, BN, B_KC>(Bs, sb, tid);
    __syncthreads();
    if (k0 + BK < kend) {  // prefetch next tile into registers while the MFMAs run
      stage_load<BM, A_KC>(sa, p.A, p.a_dt, p.lda, p.a_vec, m0, p.M, k0 + BK, kend, tid, p.a_gelu != 0);
      stage_load<BN, B_KC>(sb, p.B, p.b_dt, p.ldb, p.b_vec, n0, p.N, k0 + BK, kend, tid, p.b_gelu != 0);
    }
    Frag<CT> fa[MI], fb[NI];
    const int g = lane >> 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r0 = wr * (BM / 2) + i * 16;
      if (A_KC) fa[i] = lds_frag_kc(As, TA::pitch, r0, 0, lane);
      else fa[i] = lds_frag_ks(As, TA::pitch, r0, g * 8, g * 8 + 4, lane, p.use_tr);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int c0 = wc * (BN / 2) + j * 16;
      if (B_KC) fb[j] = lds_frag_kc(Bs, TB::pitch, c0, 0, lane);
      else fb[j] = lds_frag_ks(Bs, TB::pitch, c0, g * 8, g * 8 + 4, lane, p.use_tr);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) mma16(acc[i][j], fa[i], fb[j]);
  }

  // epilogue
  const int colbase = n0 + wc * (BN / 2) + (lane & 15);
  const int rowbase = m0 + wr * (BM / 2) + (lane >> 4) * 4;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = colbase + j * 16;
    if (col >= p.N) continue;
    const float bv = (p.bias && blockIdx.z == 0) ? p.bias[col] : 0.f;
    const float cs = p.colscale ? p.colscale[col] : 1.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rowbase + i * 16 + r;
        if (row >= p.M) continue;
        float v = (acc[i][j][r] + bv) * cs;
        if (p.aux_gelu_grad) {
          const float x = ld1(p.aux, p.aux_dt, (size_t)row * p.ldaux + col);
          v *= p.aux_mul ? x : gelu_grad_f(x);
        }
        if (p.resid) v += ld1(p.resid, p.res_dt, (size_t)row * p.ldres + col);
        const size_t ci = (size_t)row * p.ldc + col;
        if (p.atomic) atomicAdd((float*)p.C + ci, v);
        else if (p.C2) { st1(p.C, p.c_dt, ci, gelu_f(v)); if (p.C2 != p.C) st1(p.C2, p.c_dt, ci, gelu_grad_f(v)); }
        else st1(p.C, p.c_dt, ci, v);
      }
    }
  }
}

template <typename CT, int BM, int BN>
static int launch_layout(const GemmArgs& a, int layout, int nsplit, hipStream_t s) {
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, nsplit), block(256);
  switch (layout) {
    case LAYOUT_NT: hipLaunchKernelGGL((gemm_kernel<CT, BM, BN, LAYOUT_NT>), grid, block, 0, s, a); break;
    case LAYOUT_NN: hipLaunchKernelGGL((gemm_kernel<CT, BM, BN, LAYOUT_NN>), grid, block, 0, s, a); break;
    case LAYOUT_TN: hipLaunchKernelGGL((gemm_kernel<CT, BM, BN, LAYOUT_TN>), grid, block, 0, s, a); break;
    default: return SCOT_ERR_UNSUPPORTED;
  }
  return scot_check_launch();
}

template <typename CT>
static int launch_tile(const GemmArgs& a, int layout, int nsplit, hipStream_t s) {
  return launch_layout<CT, 64, 64>(a, layout, nsplit, s);  // generic fallback: one tile shape (the fast path has the rest)
}

extern int g_scot_use_tr;
int scot_gemm_fast(int layout, int compute, int M, int N, int K, const void* A, int a_dt, int lda, int a_gelu,
                   const void* B, int b_dt, int ldb, int b_gelu, void* C, int c_dt, int ldc, const float* bias,
                   const float* colscale, const void* aux, int aux_dt, int ldaux, const void* resid, int res_dt, int ldres,
                   int accumulate, float* colsum_out, void* workspace, size_t ws_bytes, int aux_mul, void* C2, hipStream_t stream);
// gemm_wide.hip: 128 x 128 tiles for the NT products whose grid keeps every CU busy with them (policy there)
int scot_gemm_wide(int layout, int compute, int M, int N, int K, const void* A, int a_dt, int lda, int a_gelu,
                   const void* B, int b_dt, int ldb, int b_gelu, void* C, int c_dt, int ldc, const float* bias,
                   const float* colscale, const void* aux, int aux_dt, int ldaux, const void* resid, int res_dt, int ldres,
                   int accumulate, float* colsum_out, int aux_mul, void* C2, hipStream_t stream);
extern "C" int scot_colsum(const void* x, int x_dt, const void* y, int y_dt, float* out, int M, int N, int ld, hipStream_t s);
int scot_gemm_panel(int layout, int compute, int M, int N, int K, const void* A, int a_dt, int lda, int a_gelu, const void* B,
                    int b_dt, int ldb, int b_gelu, void* C, int c_dt, int ldc, const float* bias, const float* colscale,
                    const void* aux, int aux_dt, int ldaux, const void* resid, int res_dt, int ldres, int accumulate,
                    float* colsum_out, int aux_mul, void* C2, hipStream_t stream);

extern "C" int scot_gemm(int layout, int compute, int M, int N, int K,
                         const void* A, int a_dt, int lda, int a_gelu,
                         const void* B, int b_dt, int ldb, int b_gelu,
                         void* C, int c_dt, int ldc,
                         const float* bias, const float* colscale,
                         const void* aux, int aux_dt, int ldaux,
                         const void* resid, int res_dt, int ldres,
                         int accumulate, float* colsum_out, void* workspace, size_t ws_bytes, int aux_mul, void* C2,
                         hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return SCOT_ERR_SHAPE;
  if (layout < 0 || layout > 2 || compute < 0 || compute > SCOT_BF16X3) return SCOT_ERR_UNSUPPORTED;
  if ((a_dt | b_dt | c_dt) & ~1) return SCOT_ERR_DTYPE;
  if (compute == SCOT_BF16X3 && (a_dt != SCOT_F32 || b_dt != SCOT_F32)) return SCOT_ERR_DTYPE;   // bf16x3 splits fp32 operands
  {
    const int rc = scot_gemm_panel(layout, compute, M, N, K, A, a_dt, lda, a_gelu, B, b_dt, ldb, b_gelu, C, c_dt, ldc, bias, colscale,
                                   aux, aux_dt, ldaux, resid, res_dt, ldres, accumulate, colsum_out, aux_mul, C2, stream);
    if (rc != SCOT_ERR_UNSUPPORTED) return rc;
  }
  {
    const int rc = scot_gemm_wide(layout, compute, M, N, K, A, a_dt, lda, a_gelu, B, b_dt, ldb, b_gelu, C, c_dt, ldc, bias, colscale,
                                  aux, aux_dt, ldaux, resid, res_dt, ldres, accumulate, colsum_out, aux_mul, C2, stream);
    if (rc != SCOT_ERR_UNSUPPORTED) return rc;
  }
  {
    const int rc = scot_gemm_fast(layout, compute, M, N, K, A, a_dt, lda, a_gelu, B, b_dt, ldb, b_gelu, C, c_dt, ldc, bias, colscale,
                                  aux, aux_dt, ldaux, resid, res_dt, ldres, accumulate, colsum_out, workspace, ws_bytes, aux_mul, C2, stream);
    if (rc != SCOT_ERR_UNSUPPORTED) return rc;
  }
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.colscale = colscale; a.aux = aux; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux; a.ldres = ldres;
  a.a_dt = a_dt; a.b_dt = b_dt; a.c_dt = c_dt; a.aux_dt = aux_dt; a.res_dt = res_dt;
  a.a_gelu = a_gelu; a.b_gelu = b_gelu; a.aux_gelu_grad = aux != nullptr; a.use_tr = g_scot_use_tr;
  a.C2 = C2; a.aux_mul = aux_mul;
  if (C2 && layout == LAYOUT_TN) return SCOT_ERR_UNSUPPORTED;
  a.a_vec = (((uintptr_t)A & 15) == 0) && (lda % 8 == 0);
  a.b_vec = (((uintptr_t)B & 15) == 0) && (ldb % 8 == 0);
  int nsplit = 1;
  a.atomic = 0;
  a.ksplit = ((K + 31) / 32) * 32;
  if (layout == LAYOUT_TN) {
    // wgrad: tiny output, huge K (= tokens) → split K so that >= ~512 workgroups exist; fp32 atomics into C.
    if (c_dt != SCOT_F32) return SCOT_ERR_DTYPE;
    const long tiles = (long)((M + 127) / 128) * ((N + 95) / 96);
    long want = (768 + tiles - 1) / tiles;
    long maxsplit = (K + 255) / 256;
    nsplit = (int)(want < 1 ? 1 : (want > maxsplit ? maxsplit : want));
    int per = (K + nsplit - 1) / nsplit;
    per = ((per + 31) / 32) * 32;
    a.ksplit = per;
    nsplit = (K + per - 1) / per;
    a.atomic = 1;
    if (!accumulate) return SCOT_ERR_UNSUPPORTED;  // caller zeroes C (gradient arena semantics: +=)
  } else if (accumulate) {
    // C += result  ≡ residual = C itself
    if (resid != nullptr) return SCOT_ERR_UNSUPPORTED;
    a.resid = C; a.res_dt = c_dt; a.ldres = ldc;
  }
  int rc = compute == SCOT_BF16 ? launch_tile<bf16_t>(a, layout, nsplit, stream) : launch_tile<float>(a, layout, nsplit, stream);
  if (rc == SCOT_OK && colsum_out) {
    if (layout == LAYOUT_TN) rc = scot_colsum(A, a_dt, nullptr, 0, colsum_out, K, M, lda, stream);  // Σ_k A[k][m]
    else rc = scot_colsum(C, c_dt, nullptr, 0, colsum_out, M, N, ldc, stream);
  }
  return rc;
}
