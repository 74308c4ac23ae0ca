// Fused AdamW (+ global gradient-norm clip) over the flat parameter / gradient arenas.
//
// Reference: the training step either side of the hot path — HF `Trainer` clips with
// `torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)` and steps `torch.optim.AdamW` over up to four
// parameter groups built by the reference's `create_optimizer` (scOT/trainer.py:295-445; run.yaml: max_grad_norm 5.0).
// With 1580 parameter tensors that is thousands of tiny launches; here the parameters already live in ONE fp32 arena
// (poseidon_amd/arena.py), so a step is three launches over it:
//   scot_grad_sqnorm  per-block partial sums of g² (parameters only: `map8` marks padding / non-parameter regions),
//   scot_clip_coef    total norm and min(1, max_norm / (norm + 1e-6)),
//   scot_adamw_step   p, m, v updated in place in one pass (28 bytes per parameter), group hyper-parameters by value; the same
//                     pass writes the 16-bit operand copy of the new weights the forward's GEMMs read (+2 bytes per parameter:
//                     the engine then has nothing to cast at the start of the next step),
//   scot_optim_finish step counters and the fp16 build's dynamic gradient scale (GradScaler semantics), one thread.
// A step whose gradient norm is not finite (overflow under the fp16 gradient scale) is skipped ON THE DEVICE by every rank alike:
// the norm is taken from the REDUCED gradient, so data-parallel replicas cannot disagree about it.
// torch.optim.AdamW arithmetic, in its order (torch/optim/adamw.py, single-tensor path):
//   p *= 1 - lr·wd;  m += (g - m)(1 - β1);  v = β2·v + (1 - β2)·g²;  p -= (lr / bc1) · m / (sqrt(v)/sqrt(bc2) + eps).
#include "common.h"

constexpr int OPT_MAX_GROUPS = 8;
struct AdamGroups { float lr[OPT_MAX_GROUPS]; float wd[OPT_MAX_GROUPS]; };

// map8[i] = group id of elements 8i .. 8i+7 (255: not a parameter)
__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* __restrict__ g, const uint8_t* __restrict__ map8, size_t n8,
                                                          float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    if (map8[i] == 255) continue;
    float v[8];
    ld8(g, SCOT_F32, i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j] * v[j];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = clip coefficient (1 when max_norm <= 0), out[1] = total norm, out[2] = 1 if the norm is not finite (else 0)
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int nblocks, float max_norm, float* out) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
    const float coef = max_norm / (norm + 1e-6f);   // torch.nn.utils.clip_grad_norm_: clamped to 1
    out[0] = (max_norm > 0.f && coef < 1.f) ? coef : 1.f;
    out[1] = norm;
    out[2] = (fabsf(norm) <= 3.4e38f) ? 0.f : 1.f;
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const uint8_t* __restrict__ map8, size_t n8, AdamGroups grp,
                                                    float beta1, float beta2, float eps, float bc1, float rsqrt_bc2,
                                                    const float* __restrict__ clip, const int* __restrict__ step_state,
                                                    bf16_t* __restrict__ shadow16) {
  // a step whose (reduced) gradient norm is not finite is skipped (what torch.cuda.amp.GradScaler.step does) — decided on the
  // device, no host round trip, identically on every data-parallel rank
  if (clip && clip[2] != 0.f) return;
  if (step_state) {   // bias corrections from the number of steps actually APPLIED (skipped steps do not advance Adam's clock)
    const double t = (double)(step_state[0] + 1);
    bc1 = (float)(1.0 - pow((double)beta1, t));
    rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, t)));
  }
  const float cc = clip ? clip[0] : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const int gi = map8[i];
    if (gi == 255) continue;
    const float lr = grp.lr[gi], wd = grp.wd[gi];
    const float step = lr / bc1;
    float pv[8], gv[8], mv[8], vv[8];
    ld8(p, SCOT_F32, i * 8, pv); ld8(g, SCOT_F32, i * 8, gv); ld8(m, SCOT_F32, i * 8, mv); ld8(v, SCOT_F32, i * 8, vv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gj = gv[j] * cc;
      pv[j] *= 1.f - lr * wd;
      mv[j] += (gj - mv[j]) * (1.f - beta1);
      vv[j] = vv[j] * beta2 + (1.f - beta2) * gj * gj;
      const float denom = sqrtf(vv[j]) * rsqrt_bc2 + eps;
      pv[j] -= step * (mv[j] / denom);
    }
    st8(p, SCOT_F32, i * 8, pv); st8(m, SCOT_F32, i * 8, mv); st8(v, SCOT_F32, i * 8, vv);
    if (shadow16) st8(shadow16, SCOT_BF16, i * 8, pv);
  }
}

// step_state: {steps applied, steps skipped}; scale_state (optional): {S, 1/S, clean steps since S last changed}
__global__ void optim_finish_kernel(int* step_state, const float* clip, float* scale_state, float growth, float backoff, int interval,
                                    float max_scale) {
  if (threadIdx.x != 0) return;
  const bool inf = clip && clip[2] != 0.f;
  if (inf) step_state[1] += 1; else step_state[0] += 1;
  if (scale_state && interval > 0) {
    float S = scale_state[0], clean = scale_state[2];
    if (inf) { S *= backoff; if (S < 1.f) S = 1.f; clean = 0.f; }
    else if (++clean >= (float)interval) { S *= growth; if (S > max_scale) S = max_scale; clean = 0.f; }
    scale_state[0] = S; scale_state[1] = 1.f / S; scale_state[2] = clean;
  }
}

static int opt_blocks(size_t n8) {
  size_t b = (n8 + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// partial: [scot_optim_blocks(n)] floats of scratch owned by the caller
extern "C" int scot_optim_blocks(size_t n) { return opt_blocks(n / 8); }

extern "C" int scot_grad_sqnorm(const float* grad, const uint8_t* map8, size_t n, float* partial, hipStream_t stream) {
  if (!grad || !map8 || !partial || n % 8) return SCOT_ERR_SHAPE;
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(opt_blocks(n / 8)), dim3(256), 0, stream, grad, map8, n / 8, partial);
  return scot_check_launch();
}

extern "C" int scot_clip_coef(const float* partial, int nblocks, float max_norm, float* out2, hipStream_t stream) {
  if (!partial || !out2 || nblocks <= 0) return SCOT_ERR_SHAPE;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, partial, nblocks, max_norm, out2);
  return scot_check_launch();
}

// lr / wd: HOST arrays of ngroups floats (passed to the kernel by value); clip: the 3 floats written by scot_clip_coef (or NULL:
// no clipping, no skip); step_state: device int[2] {applied, skipped} — Adam's step number is applied + 1, read on the device — or
// NULL: `step` (>= 1) is used; shadow16 (optional, 16-byte aligned, n elements): the library's 16-bit operand copy of the
// updated parameters, written in the same pass.
extern "C" int scot_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* map8, size_t n,
                               const float* lr, const float* wd, int ngroups, float beta1, float beta2, float eps, int step,
                               const float* clip, const int* step_state, void* shadow16, hipStream_t stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !map8 || n % 8 || ngroups < 1 || ngroups > OPT_MAX_GROUPS || (!step_state && step < 1))
    return SCOT_ERR_SHAPE;
  if (((uintptr_t)shadow16) & 15) return SCOT_ERR_SHAPE;
  if (step < 1) step = 1;
  AdamGroups grp{};
  for (int i = 0; i < ngroups; ++i) { grp.lr[i] = lr[i]; grp.wd[i] = wd[i]; }
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adamw_kernel, dim3(opt_blocks(n / 8)), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, map8, n / 8, grp,
                     beta1, beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)), clip, step_state, (bf16_t*)shadow16);
  return scot_check_launch();
}

// After scot_adamw_step: step_state[0] += 1 (applied) or step_state[1] += 1 (skipped: clip[2] != 0), and — when scale_state is
// given and interval > 0 — torch.cuda.amp.GradScaler's update of the gradient scale: S *= backoff on a skipped step, S *= growth
// after `interval` applied steps in a row; scale_state = {S, 1/S, clean steps}, S kept within [1, max_scale].
extern "C" int scot_optim_finish(int* step_state, const float* clip, float* scale_state, float growth, float backoff, int interval,
                                 float max_scale, hipStream_t stream) {
  if (!step_state) return SCOT_ERR_SHAPE;
  hipLaunchKernelGGL(optim_finish_kernel, dim3(1), dim3(64), 0, stream, step_state, clip, scale_state, growth, backoff, interval, max_scale);
  return scot_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------------
// Data-parallel wire format (poseidon_amd/dp.py): the fp32 gradient arena goes over xGMI as bfloat16 (half the bytes; the
// reference's DDP sends fp32 — reference scOT/train.py via HF Trainer/accelerate).  One pass each way, the mean's 1/N folded in
// (the first version used two torch elementwise kernels per chunk plus a div).  ALWAYS bfloat16 (gradient range), whatever
// 16-bit operand format this build of the library computes with.
typedef __bf16 wire_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t wire_pack2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, wire_bf16x2_t));
}
// wire[i] = bf16(scale · src[i])
__global__ __launch_bounds__(256) void dp_pack_kernel(const float* __restrict__ src, uint16_t* __restrict__ wire, size_t n, float scale) {
  const size_t n8 = n / 8, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    const float4 a = ((const float4*)src)[2 * i], b = ((const float4*)src)[2 * i + 1];
    ((uint4*)wire)[i] = make_uint4(wire_pack2(a.x * scale, a.y * scale), wire_pack2(a.z * scale, a.w * scale),
                                   wire_pack2(b.x * scale, b.y * scale), wire_pack2(b.z * scale, b.w * scale));
  }
  for (size_t i = n8 * 8 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    wire[i] = (uint16_t)(wire_pack2(src[i] * scale, 0.f) & 0xffffu);
}
// dst[i] = scale · float(wire[i])
__global__ __launch_bounds__(256) void dp_unpack_kernel(const uint16_t* __restrict__ wire, float* __restrict__ dst, size_t n, float scale) {
  const size_t n8 = n / 8, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
    const uint4 u = ((const uint4*)wire)[i];
    ((float4*)dst)[2 * i] = make_float4(__uint_as_float(u.x << 16) * scale, __uint_as_float(u.x & 0xffff0000u) * scale,
                                        __uint_as_float(u.y << 16) * scale, __uint_as_float(u.y & 0xffff0000u) * scale);
    ((float4*)dst)[2 * i + 1] = make_float4(__uint_as_float(u.z << 16) * scale, __uint_as_float(u.z & 0xffff0000u) * scale,
                                            __uint_as_float(u.w << 16) * scale, __uint_as_float(u.w & 0xffff0000u) * scale);
  }
  for (size_t i = n8 * 8 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    dst[i] = __uint_as_float(((uint32_t)wire[i]) << 16) * scale;
}
// include/scot_hip.h: scot_dp_pack / scot_dp_unpack (src / dst 32-byte aligned, wire 16-byte aligned)
extern "C" int scot_dp_pack(const float* src, void* wire, size_t n, float scale, hipStream_t s) {
  if (n == 0) return SCOT_OK;
  if ((((uintptr_t)src) & 31) || (((uintptr_t)wire) & 15)) return SCOT_ERR_SHAPE;
  size_t blocks = (n / 8 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(dp_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, (uint16_t*)wire, n, scale);
  return scot_check_launch();
}
extern "C" int scot_dp_unpack(const void* wire, float* dst, size_t n, float scale, hipStream_t s) {
  if (n == 0) return SCOT_OK;
  if ((((uintptr_t)dst) & 31) || (((uintptr_t)wire) & 15)) return SCOT_ERR_SHAPE;
  size_t blocks = (n / 8 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(dp_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint16_t*)wire, dst, n, scale);
  return scot_check_launch();
}
