// Shared device helpers for the scOT HIP kernels (gfx950 / CDNA4 only — no portability shims).
//
// Compute types (template parameter CT):
//   bf16_t : 16-bit operands, v_mfma_f32_16x16x32_{bf16|f16}, fp32 accumulate  (the fast path).  The FORMAT of the 16-bit
//            operand type is a property of the library build: libscot_hip.so = bfloat16 (8-bit significand, fp32 range),
//            libscot_hip_f16.so (-DSCOT_OPERAND_FP16) = IEEE binary16 (11-bit significand).  Same kernels, same data movement,
//            same MFMA rate; fp16 is what meets the north star's 1e-3 output bound (operand rounding 2^-12 instead of 2^-9:
//            7e-4 instead of 6e-3 on trained-like weights, tools/probes/precision_sim.py), at the price of a loss scale in the
//            backward (engine.py).  Everything below that says "bf16" means "the library's 16-bit operand format".
//   float  : exact fp32 v_mfma_f32_16x16x4_f32 (k-ordered fmaf chain)              (the 1e-5 parity path)
// Both use ONE fragment convention so every kernel is written once:
//   a lane (r = lane&15, g = lane>>4) owns 8 contraction elements k(g,j), j=0..7, of row r (A) / column r (B).
//   bf16: the 8 elements are the MFMA's native 16x16x32 operand;  f32: element j feeds the j-th of 8
//   16x16x4 MFMAs, whose native k index is g.  Any k(g,j) bijection is legal as long as A and B agree.
// Accumulator (C/D) layout of both: col = lane&15, row = (lane>>4)*4 + reg.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SCOT_F32 0
#define SCOT_BF16 1
#define SCOT_BF16X3 2   /* compute mode only (never a storage dtype): fp32 operands, hi/lo bf16 split, 3 MFMAs per K-step */

#define SCOT_OK 0
#define SCOT_ERR_SHAPE (-1)
#define SCOT_ERR_DTYPE (-2)
#define SCOT_ERR_UNSUPPORTED (-3)
#define SCOT_ERR_LAUNCH (-4)

typedef uint16_t bf16_t;
#if defined(SCOT_OPERAND_FP16)
typedef _Float16 h16_scalar_t;
#else
typedef __bf16 h16_scalar_t;
#endif
typedef h16_scalar_t bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#if defined(SCOT_OPERAND_FP16)
__device__ __forceinline__ float bf2f(bf16_t x) { return (float)__builtin_bit_cast(_Float16, x); }   // v_cvt_f32_f16
#else
__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }
#endif
// fp32 -> bf16, round-to-nearest-even, via gfx950's v_cvt_pk_bf16_f32 (the compiler selects it for __bf16 casts; the
// first version of these helpers did the rounding with 6 integer VALU ops per element — rocprof PMC showed the attention
// kernels VALU-issue bound with ~1/4 of the instructions being that conversion).
typedef h16_scalar_t bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (h16_scalar_t)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {   // v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (RNE)
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// the two 16-bit operands of a dword -> fp32 (bf16: shift / mask; fp16: v_cvt_f32_f16 and its SDWA high-half form)
__device__ __forceinline__ void unpack_bf16x2(uint32_t u, float& lo, float& hi) {
#if defined(SCOT_OPERAND_FP16)
  const bf16x2_t h = __builtin_bit_cast(bf16x2_t, u);
  lo = (float)h[0]; hi = (float)h[1];
#else
  lo = __uint_as_float(u << 16); hi = __uint_as_float(u & 0xffff0000u);
#endif
}

template <typename CT> __device__ __forceinline__ CT to_ct(float f);
template <> __device__ __forceinline__ float to_ct<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t to_ct<bf16_t>(float f) { return f2bf(f); }
__device__ __forceinline__ float from_ct(float f) { return f; }
__device__ __forceinline__ float from_ct(bf16_t f) { return bf2f(f); }
template <typename CT> struct ct_traits;
template <> struct ct_traits<float> { static constexpr int dtype = SCOT_F32; static constexpr int kpad = 4; };
template <> struct ct_traits<bf16_t> { static constexpr int dtype = SCOT_BF16; static constexpr int kpad = 8; };

// wait until at most n of this wave's vector-memory operations are outstanding (direct-to-LDS loads are counted by nothing else:
// the compiler does not order a ds_read behind a global_load_lds)
#ifdef SCOT_HIPEMU
#define SCOT_VMCNT(n) ((void)0)
#else
#define SCOT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#endif

// ---- ablation hooks (tools/ablate_kernels.py builds extra copies of ONE source with -DSCOT_ABL=<bits>; the product never defines it).
// TIMING ONLY — results are garbage: 1 = st8 stores nothing (values kept alive), 2 = gelu_terms is two multiplies, 4 = ld8 loads
// nothing, 8 = no MFMA (and, dead-code-eliminated with it, no fragment reads), 16 = __syncthreads is a no-op, 32 = no global atomics,
// 64 = mlp_fused.hip: gelu'(u) is not loaded, 128 = mlp_fused.hip: du is not stored, 256 = v_exp_f32 is the identity.
#ifndef SCOT_ABL
#define SCOT_ABL 0
#endif
#if SCOT_ABL & 16
#define __syncthreads() __builtin_amdgcn_wave_barrier()
#endif
#if SCOT_ABL & 32
#define atomicAdd(p, v) asm volatile("" ::"v"(v))
#endif
#if SCOT_ABL & 256
#define __builtin_amdgcn_exp2f(x) (x)
#endif

// ---- runtime-dtype global memory access (dtype is wave-uniform → scalar branch) ----------------------------
__device__ __forceinline__ float ld1(const void* p, int dt, size_t i) {
  return dt == SCOT_F32 ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void st1(void* p, int dt, size_t i, float v) {
  if (dt == SCOT_F32) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = f2bf(v);
}
// 8 consecutive elements; caller guarantees 16-byte alignment of element i (f32: 32-byte span, two 16 B loads)
__device__ __forceinline__ void ld8(const void* p, int dt, size_t i, float v[8]) {
#if SCOT_ABL & 4
  for (int j = 0; j < 8; ++j) v[j] = 1.0f + (float)(i & 7);
  return;
#endif
  if (dt == SCOT_F32) {
    const float4 a = *(const float4*)((const float*)p + i);
    const float4 b = *(const float4*)((const float*)p + i + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const uint4 u = *(const uint4*)((const bf16_t*)p + i);
    unpack_bf16x2(u.x, v[0], v[1]); unpack_bf16x2(u.y, v[2], v[3]); unpack_bf16x2(u.z, v[4], v[5]); unpack_bf16x2(u.w, v[6], v[7]);
  }
}
__device__ __forceinline__ void st8(void* p, int dt, size_t i, const float v[8]) {
#if SCOT_ABL & 1
  asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
  return;
#endif
  if (dt == SCOT_F32) {
    *(float4*)((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)p + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    uint4 u;
    u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]); u.z = pack_bf16x2(v[4], v[5]); u.w = pack_bf16x2(v[6], v[7]);
    *(uint4*)((bf16_t*)p + i) = u;
  }
}

// erf-GELU (hidden_act="gelu", reference train.py:263).  libm erff costs ~50 VALU ops and sat on the critical path of the
// fc2 operand loader (rocprof round 1: +40 us per stage-0 fc2); Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. below
// fp32 GELU resolution for the 1e-5 parity mode) needs one exp, one rcp and a degree-5 Horner chain — and the SAME
// exp(-x^2/2) also gives the Gaussian term of the derivative.
__device__ __forceinline__ void gelu_terms(float x, float& cdf, float& pdf_times_sqrt2pi) {
#if SCOT_ABL & 2
  cdf = 0.5f * x; pdf_times_sqrt2pi = 0.25f * x;
  return;
#endif
  // 15 VALU instructions per element (v_rcp_f32 and v_exp_f32 directly: `__frcp_rn` expands to the 11-instruction IEEE division
  // sequence, which made this function 28 instructions and the fused MLP kernels VALU-bound on it — round-2 microbenchmarks)
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);      // exp(-z^2) = exp(-x^2/2)
  // 0.5 * (A&S 7.1.26 polynomial): h = 0.5 * erfc(|x|/sqrt2) = poly * e
  const float poly = ((((0.5307027145f * t - 0.7265760135f) * t + 0.7107068705f) * t - 0.142248368f) * t + 0.127414796f) * t;
  const float h = fmaf(-poly, e, 0.5f);                  // 0.5 * erf(|x|/sqrt2)
  cdf = 0.5f + copysignf(h, x);                          // Phi(x)
  pdf_times_sqrt2pi = e;                                 // exp(-x^2/2)
}
// 8 consecutive compute-type elements to a 16-byte aligned (LDS or global) address
__device__ __forceinline__ void store8_ct(bf16_t* p, const float v[8]) {
  *(uint4*)p = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ void store8_ct(float* p, const float v[8]) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

__device__ __forceinline__ float gelu_f(float x) {
  float cdf, e;
  gelu_terms(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  float cdf, e;
  gelu_terms(x, cdf, e);
  return cdf + x * 0.3989422804014327f * e;
}

// ---- fragments -------------------------------------------------------------------------------------------
template <typename CT> struct Frag;
template <> struct Frag<float> { float v[8]; };
template <> struct Frag<bf16_t> { s16x8_t v; };

__device__ __forceinline__ void frag_set(Frag<float>& f, int j, float x) { f.v[j] = x; }
__device__ __forceinline__ void frag_set(Frag<bf16_t>& f, int j, float x) { f.v[j] = (short)f2bf(x); }
__device__ __forceinline__ void frag_zero(Frag<float>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = 0.f;
}
__device__ __forceinline__ void frag_zero(Frag<bf16_t>& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = 0;
}
template <typename CT> __device__ __forceinline__ Frag<CT> frag_from_f32(const float x[8]);
template <> __device__ __forceinline__ Frag<float> frag_from_f32<float>(const float x[8]) {
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = x[j];
  return f;
}
template <> __device__ __forceinline__ Frag<bf16_t> frag_from_f32<bf16_t>(const float x[8]) {
  const uint4 u = make_uint4(pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7]));
  Frag<bf16_t> f;
  f.v = __builtin_bit_cast(s16x8_t, u);
  return f;
}

__device__ __forceinline__ void mma16(f32x4_t& c, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
#if SCOT_ABL & 8
  return;
#endif
#if defined(SCOT_OPERAND_FP16)
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v), c, 0, 0, 0);
#else
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ void mma16(f32x4_t& c, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
}

// ---- LDS fragment reads ----------------------------------------------------------------------------------
// K-contiguous tile  T[row][pitch]: lane reads 8 consecutive k of row r0+(lane&15) at k = kk + (lane>>4)*8.
__device__ __forceinline__ Frag<bf16_t> lds_frag_kc(const bf16_t* t, int pitch, int r0, int kk, int lane) {
  Frag<bf16_t> f;
  f.v = *(const s16x8_t*)(t + (r0 + (lane & 15)) * pitch + kk + (lane >> 4) * 8);
  return f;
}
__device__ __forceinline__ Frag<float> lds_frag_kc(const float* t, int pitch, int r0, int kk, int lane) {
  Frag<float> f;
  const float* p = t + (r0 + (lane & 15)) * pitch + kk + (lane >> 4) * 8;
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
  return f;
}
// K-strided tile  T[k][pitch] (the fragment's row/column index runs along the contiguous dimension):
// element j<4 comes from LDS row klo+j, j>=4 from khi+(j-4), column c0+(lane&15).
// bf16 + use_tr: two ds_read_b64_tr_b16.  Each 16-lane group hands the instruction 16 8-byte chunks forming a
// row-major [4][16] b16 block (lane i -> row i>>2, chunk i&3) and lane c receives column c of that block.
// (semantics verified at library init by scot_selftest_tr(); scalar gather otherwise.)
__device__ __forceinline__ Frag<bf16_t> lds_frag_ks(const bf16_t* t, int pitch, int c0, int klo, int khi, int lane, int use_tr) {
  Frag<bf16_t> f;
  const int i = lane & 15;
  if (use_tr) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
    const bf16_t* p0 = t + (klo + (i >> 2)) * pitch + c0 + (i & 3) * 4;
    const bf16_t* p1 = t + (khi + (i >> 2)) * pitch + c0 + (i & 3) * 4;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p1);
    f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f.v[j] = (short)t[(klo + j) * pitch + c0 + i];
      f.v[j + 4] = (short)t[(khi + j) * pitch + c0 + i];
    }
  }
  return f;
}
__device__ __forceinline__ Frag<float> lds_frag_ks(const float* t, int pitch, int c0, int klo, int khi, int lane, int) {
  Frag<float> f;
  const int i = lane & 15;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.v[j] = t[(klo + j) * pitch + c0 + i];
    f.v[j + 4] = t[(khi + j) * pitch + c0 + i];
  }
  return f;
}

// ---- wave reductions ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int scot_check_launch() { return hipGetLastError() == hipSuccess ? SCOT_OK : SCOT_ERR_LAUNCH; }
