/* scot_hip.h — C ABI of libscot_hip.so, the MI355X (gfx950) kernel library behind the scOT forward/backward path.
 *
 * Boundary: the reference exposes this path as a Python nn.Module (`scOT.model.ScOT`), whose
 * arithmetic is stock torch ops.  Each entry point below replaces the torch ops of the cited reference lines
 * (ref = /root/reference/scOT/model.py, HF = transformers/models/swinv2/modeling_swinv2.py) and is what a
 * ctypes / pybind / cgo binding would bind; see INTEGRATION.md for the reference-side stub.
 *
 * Conventions: every pointer is a DEVICE pointer borrowed for the duration of the enqueue; calls only enqueue
 * work on `stream` (never allocate, never synchronise, except scot_selftest_tr); return 0 on success, <0 on
 * bad shape (-1) / dtype (-2) / unsupported configuration (-3) / launch failure (-4).
 * dtype codes: 0 = float32, 1 = the library's 16-bit operand format (raw uint16; bfloat16 or binary16, see
 * scot_operand_format).  compute codes: 0 = exact fp32 MFMA, 1 = 16-bit MFMA, 2 = split (hi + lo) 16-bit MFMA.
 */
#ifndef SCOT_HIP_H
#define SCOT_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* scot_stream_t; /* = hipStream_t */

#define SCOT_DT_F32 0
#define SCOT_DT_BF16 1
#define SCOT_LAYOUT_NT 0 /* C[M,N] = A[M,K] B[N,K]^T : nn.Linear forward (HF:545-561, HF:389-410, ref:709,747,760) */
#define SCOT_LAYOUT_NN 1 /* C[M,N] = A[M,K] B[K,N]   : dgrad of nn.Linear; ConvTranspose2d k=s (ref:616-621)      */
#define SCOT_LAYOUT_TN 2 /* C[M,N] += A[K,M]^T B[K,N] : wgrad of nn.Linear (autograd of the above)                 */

int scot_abi_version(void);   /* 5 (round 6: 4 = scot_wgrad_group / scot_wgrad_mlp modes, scot_segments_scale, scot_gemm_splitk_config; 5 = scot_dp_init .. finalize); the bindings check it at load */
/* Format of dtype code 1 in THIS build of the library: 0 = bfloat16 (libscot_hip.so), 1 = IEEE binary16 (libscot_hip_f16.so, the
 * same sources compiled with -DSCOT_OPERAND_FP16).  The reference computes in fp32 (ref:1318-1509); 16-bit operands are this
 * library's choice and binary16 is the one that keeps ScOT.forward within 1e-3 of it. */
int scot_operand_format(void);
/* x[0..n) *= scale (fp32, in place, x 16-byte aligned); *nonfinite (optional, device int) += number of waves that saw Inf/NaN.
 * Undoes the backward's gradient scale on the gradient arena (the role torch.cuda.amp.GradScaler.unscale_ plays for the
 * reference's fp16 recipe, trainer.py via HF Trainer). */
int scot_scale_inplace(float* x, size_t n, float scale, int* nonfinite, scot_stream_t stream);
/* The same with the factor read from the device (x *= *scale_dev): the fp16 build's DYNAMIC gradient scale — a recorded step holds
 * the address, scot_optim_finish changes the value between steps. */
int scot_scale_inplace_dev(float* x, size_t n, const float* scale_dev, int* nonfinite, scot_stream_t stream);
/* The same on a LIST of pieces of x in one launch: chunks = nchunks pairs (offset, count) of int64 on the device, in floats, offset a
 * multiple of 4, count <= 4096.  scale_dev NULL: the pieces are zeroed instead.  This is how the part of the gradient arena that is
 * NOT written by a storing first writer (biases, norm / bias-MLP / convolution parameters, trunk weights: ~5 % of Poseidon-B's bytes)
 * is cleared by zero_grad and brought back from the fp16 gradient scale after the backward; the Linear weights of the ScOTLayers
 * take the scale in the store of their weight-gradient kernels (scot_wgrad_group / scot_wgrad_mlp `mode`).  Reference semantics:
 * autograd's accumulate-into-.grad, trainer.py:605-635 via HF Trainer. */
int scot_segments_scale(float* x, const long long* chunks, int nchunks, const float* scale_dev, int* nonfinite, scot_stream_t stream);
/* Local power-of-two rescale of a gradient branch behind a tiny per-channel scale (ConvNeXt layer scale, model.py:191-195,212-213)
 * in the binary16 build — all factors stay on the device:
 *   scot_pow2_rescale: out2[0] = c = 2^k (k >= 0) with max|v|·c in (1/2, 1], out2[1] = 1/c;
 *   scot_colscale_dev: out[r,c] = g[r,c] * gamma[c] * mul[0]  (out: fp32 or the 16-bit operand format; C % 8 == 0);
 *   scot_axpy_dev:     dst += alpha[0] * src  (fp32; clear_src: src = 0 afterwards). */
int scot_pow2_rescale(const float* v, int n, float* out2, scot_stream_t stream);
int scot_colscale_dev(const float* g, const float* gamma, const float* mul, void* out, int out_dt, int rows, int C, scot_stream_t stream);
int scot_axpy_dev(float* dst, float* src, size_t n, const float* alpha, int clear_src, scot_stream_t stream);
/* Batch assembly from trajectories resident in HBM (the reference's Dataset.__getitem__ + collate: scOT/problems/base.py:318-334,
 * scOT/problems/fluids/incompressible.py:74-160, compressible.py:84-262):
 *   pv [b,c,y,x] = a[c] * data[i_b, t1_b, src[c], y, x] + b[c],  lab[...] the same at t2_b;  src[c] < 0: constant plane b[c];
 *   transpose: (y, x) read at (x, y).  data fp32 [n, T, nsrc, H, W]; it int32 [3, B] = trajectories, t1, t2. */
int scot_gather_pairs(const float* data, const int* it, const int* src, const float* a, const float* b, float* pv, float* lab,
                      int B, int C, int T, int nsrc, int H, int W, int transpose, scot_stream_t stream);
/* One tensor of a batch with its own recipe (readers whose inputs and labels differ, static or analytic extra channels:
 * scOT/problems/elliptic/*.py, wave/acoustic.py, reaction_diffusion/allen_cahn.py, fluids/incompressible.py:149-243 KolmogorovFlow,
 * fluids/compressible.py:8-53 Airfoil):  out[b,c] = a[c] * P + b[c] with P = data[traj[b], tidx[b], src[c]] (src >= 0), nothing
 * (src = -1: the constant b[c]) or the fixed plane planes[-2 - src[c]] ([*, H, W] fp32; may be NULL when unused). */
int scot_gather_planes(const float* data, const int* traj, const int* tidx, const int* src, const float* a, const float* b,
                       const float* planes, float* out, int B, int C, int T, int nsrc, int H, int W, int transpose, scot_stream_t stream);
/* Transposed operand-format copies of weight matrices, converted from the fp32 master in the same pass:
 * wt16[off + c*rows + r] = w[off + r*cols + c] for every matrix of desc (int32 [n][4], device: element offset, rows, cols, index of
 * its first 64x64 tile; rows and cols multiples of 8; tiles = total tile count).  The data gradient of nn.Linear, dX = dY · W, then
 * runs as the forward's NT product on W^T instead of the slower strided-operand NN product. */
int scot_transpose_cast(const float* w, void* wt16, const int* desc, int n, int tiles, scot_stream_t stream);
/* Mask tokens of ScOTEmbeddings (ref:353-359): x[r,:] = mask[r] ? token : x[r,:] in place (x fp32 [rows, C], mask uint8 [rows],
 * token fp32 [C]); backward: d_token += Σ_r mask[r]·g[r,:], g[r,:] = 0 where mask[r]. */
int scot_mask_tokens(float* x, const void* mask_u8, const float* token, int rows, int C, scot_stream_t stream);
int scot_mask_tokens_bwd(float* g, const void* mask_u8, float* d_token, int rows, int C, scot_stream_t stream);
/* Second half of the spectral resize (ref:1293-1316, `_downsample` / `_upsample`: fft2 -> crop / zero-pad the centred spectrum ->
 * ifft2 -> real part), restated as the linear map Y = Pr X Pr^T - Pi X Pi^T (P built on the host, scOT/model.py).  The first half
 * U = X [Pr; Pi]^T is a scot_gemm (NT, compute 0); this entry does Y[b] = Pr·U[b,:,:t] - Pi·U[b,:,t:] per image.
 * U [nimg, s, 2t], Pr / Pi [t, s], Y [nimg, t, t], all fp32; s <= 512. */
int scot_spectral_apply(const float* U, const float* Pr, const float* Pi, float* Y, int nimg, int s, int t, scot_stream_t stream);
/* Data-parallel wire format of the gradient arena (the exchange itself is torch.distributed's RCCL all-reduce / reduce-scatter on
 * the buffers below; reference: DDP's fp32 bucket all-reduce via HF Trainer/accelerate, scOT/train.py):
 *   scot_dp_pack:   wire[i] = bfloat16(scale * src[i])     (scale = 1/world: the mean is taken before the 16-bit sum)
 *   scot_dp_unpack: dst[i]  = scale * float(wire[i])
 * src/dst fp32 (32-byte aligned), wire bfloat16 (16-byte aligned) — bfloat16 in BOTH builds of the library. */
int scot_dp_pack(const float* src, void* wire, size_t n, float scale, scot_stream_t stream);
int scot_dp_unpack(const void* wire, float* dst, size_t n, float scale, scot_stream_t stream);
/* The exchange itself for a host WITHOUT torch.distributed (reference: the DDP all-reduce behind `accelerate launch`,
 * README.md:50-57, per-device batch scOT/train.py:281).  RCCL is resolved at run time (dlopen: the copy already mapped into the
 * process, else the loader path, else /opt/rocm/lib) — no link-time dependency; without it these return -3 and say so on stderr.
 *   scot_dp_unique_id:        rank 0 draws the 128-byte rendezvous token (ncclGetUniqueId) into host memory; the host distributes it.
 *   scot_dp_init:             collective; joins `world` ranks as `rank` on the CURRENT HIP device (one process per GPU).  -3 if this
 *                             process already holds a communicator.
 *   scot_dp_allreduce_bucket: in-place SUM over ranks of grads[0..n) on comm_stream; dtype 0 = fp32, 1 = the bfloat16 wire format of
 *                             scot_dp_pack (bfloat16 in both builds).  The mean's 1/world is applied by the caller BEFORE the sum
 *                             (scot_dp_pack's scale; on fp32 ranges a scale pass), so a 16-bit wire cannot overflow.  -3 before init.
 *   scot_dp_world / _rank:    0 / -1 outside init..finalize.
 *   scot_dp_finalize:         destroys the communicator (caller has synchronised comm_stream); idempotent.
 * These are the only entry points that block (init) or touch host memory (unique_id). */
int scot_dp_unique_id(void* id128_host);
int scot_dp_init(const void* id128_host, int rank, int world);
int scot_dp_allreduce_bucket(void* grads, size_t n, int dtype, scot_stream_t comm_stream);
int scot_dp_world(void);
int scot_dp_rank(void);
int scot_dp_finalize(void);
int scot_selftest_tr(scot_stream_t stream); /* 1: ds_read_b64_tr_b16 path verified & on, 0: scalar-gather fallback */
void scot_set_use_tr(int v);
int scot_get_use_tr(void);

/* compute: 0 = fp32 (exact v_mfma_f32_16x16x4_f32), 1 = bf16 operands (v_mfma_f32_16x16x32_bf16, fp32 accumulate), 2 = bf16x3:
 * fp32 operands split into hi + lo bf16 on the way into LDS, hi·hi + hi·lo + lo·hi on the bf16 MFMA (~2^-17 operand error).
 * Dense contraction with fused prologue/epilogue.  a_gelu/b_gelu: apply erf-GELU to the operand while loading
 * (Swinv2Intermediate's activation, HF:545-548).  Epilogue: (+bias[n]) (*colscale[n]) (*gelu'(aux[m,n])) (+resid[m,n]);
 * accumulate=1: C += result (required for TN, which splits K and uses fp32 atomics). */
int scot_gemm(int layout, int compute, int M, int N, int K,
              const void* A, int a_dt, int lda, int a_gelu,
              const void* B, int b_dt, int ldb, int b_gelu,
              void* C, int c_dt, int ldc,
              const float* bias, const float* colscale,
              const void* aux, int aux_dt, int ldaux,
              const void* resid, int res_dt, int ldres,
              int accumulate, float* colsum_out, void* workspace, size_t ws_bytes, int aux_mul, void* C2,
              scot_stream_t stream);
/* colsum_out (optional, fp32, +=): NT/NN: column sums of the stored result; TN: Σ_k A[k][m] — i.e. the bias gradient
 * when A = dY, taken from the dY tile already staged in LDS.
 * workspace (optional, 32-byte aligned device scratch owned by the caller): TN splits K over workgroups and writes
 * partial tiles there, reduced by one extra pass; without it TN falls back to fp32 atomics.
 * C2 (optional, NT/NN): the epilogue stores gelu(v) to C and gelu'(v) to C2 (same dtype/ld; C2 == C: gelu(v) only, the
 * inference form) — the fc1 form, so that no
 * later kernel re-evaluates erf; aux_mul=1: `aux` already holds that derivative and is multiplied in as is. */

/* Scratch the call above would use with these dimensions (dense operands in the compute mode's operand type; TN: fp32 result with
 * accumulate = 1): bytes of split-K partial tiles, 0 when it runs unsplit.  scot_gemm adapts to a SMALLER workspace (fewer K slices,
 * fp32 atomics without any) — this is the size at which nothing is clipped.  SURVEY.md §8(b): `scot_<op>_workspace_bytes(dims…)`. */
size_t scot_gemm_workspace_bytes(int layout, int compute, int M, int N, int K);

/* NT products with 16-bit operands, M % 128 == 0, N % 128 == 0, K % 64 == 0 whose grid of 128 x 128 output tiles still gives every CU
 * two workgroups, or one with a long contraction (the Linear layers HF modeling_swinv2.py:396-410, 496-506, 536-561 at the widths and
 * token counts of Poseidon-L, of Poseidon-B at 256 x 256, and the long-K / gelu'-scaled products of Poseidon-B's deep stages), run in
 * csrc/gemm_wide.hip (policy and measurements there); results agree with the 64 x 64-tile kernel to fp32 summation order.
 * scot_gemm_wide_config: mode 0 = never use these tiles, 1 = the library's policy (default), 2 = every eligible call, with kernel
 * `variant` 0 (8 waves, 4 LDS stages), 1 (8 waves, 2 stages) or 2 (4 waves, 2 stages) — tests and tools/bench_deep_gemm.py.
 * Process-wide, not thread-safe against concurrent scot_gemm calls. */
void scot_gemm_wide_config(int mode, int variant);

/* K slices for the NT products whose result is fp32 and whose epilogue is bias-only (HF modeling_swinv2.py:552-561 forward at the deep
 * stages; the data gradients of HF:536-548 and HF:396-410 accumulating into the fp32 residual-stream gradient): every slice adds its
 * partial tile into the result with fp32 atomics (csrc/gemm_fast.hip; a non-accumulating call zeroes the result on `stream` first).
 * Results agree with the unsplit kernel to fp32 summation order and are not bit-reproducible run to run.
 * scot_gemm_splitk_config: slices 0 = the library's policy (default), -1 = never, S > 0 = S slices for every eligible call;
 * zeroed_too 0 = only accumulating calls are split.  Process-wide, like scot_gemm_wide_config. */
void scot_gemm_splitk_config(int slices, int zeroed_too);

/* The weight gradients of one ScOTLayer in ONE launch: for i < n (n <= 8)
 *   dW_i[M_i, N_i] += dY_i[K, M_i]^T · X_i[K, N_i],   dbias_i[M_i] += Σ_k dY_i[k, :]   (dbias / dbias_i may be NULL)
 * i.e. the autograd of query/key/value, attention.output.dense, intermediate.dense and output.dense (HF:396-410, 502-506,
 * 545-561) over the same K tokens.  Operands dense row-major in the 16-bit operand format, dW fp32; the arrays of pointers /
 * sizes are HOST arrays, read before the call returns.  K is split over workgroups, partial tiles go through `workspace`
 * (32-byte aligned device scratch, >= nsplit · Σ M_i N_i floats) and one grouped pass adds them into the gradients.
 * compute must be 1 (16-bit MFMA); returns -3 for anything else (use scot_gemm per problem).
 * modes (host array of n ints, or NULL = all 0) says how each result meets dW_i, with s = *grad_scale (device float, NULL = 1):
 *   0  dW_i += acc            the gradient arena's accumulate semantics (autograd's += into .grad)
 *   1  dW_i  = s · acc        the FIRST writer of a gradient that zero_grad left unfilled: no fill, no re-read of zeros, and the fp16
 *                             build's un-scale 1/S rides in the store (no separate pass over the arena)
 *   2  dW_i += s · acc        later backwards of an accumulation window (the tensor holds unscaled values)
 * dbias_i is always a plain accumulation. */
int scot_wgrad_group(int compute, int n, int K, const void* const* dY, const void* const* X, float* const* dW,
                     float* const* dbias, const int* M, const int* N, void* workspace, size_t ws_bytes, const int* modes,
                     const float* grad_scale, scot_stream_t stream);
/* nsplit · Σ M_i N_i · 4 bytes for the K split scot_wgrad_group chooses for these shapes (0: unsplit, or shapes it does not cover).
 * With less it splits less; with none and a split wanted it returns -3. */
size_t scot_wgrad_group_workspace_bytes(int n, int K, const int* M, const int* N);

/* Shifted-window cosine attention, HF:389-455 + ref:522-559 (roll/partition/mask folded into indexing).
 * qkv: [batch*Hp*Wp][3C] (q|k|v) in the compute dtype; out: [batch*Hp*Wp][C]; lse: [batch*nW][heads][ws*ws] f32;
 * bias_table: [heads][(2ws-1)^2] = 16*sigmoid(CPB MLP) from scot_cpb_fwd; logit_scale: [heads]. */
int scot_window_attn_fwd(int compute, const void* qkv, void* out, float* lse, const float* bias_table,
                         const float* logit_scale, int batch, int Hp, int Wp, int C, int heads, int ws, int shift,
                         scot_stream_t stream);
int scot_window_attn_bwd(int compute, const void* qkv, const void* out_fwd, const void* dout, const float* lse, const float* bias_table,
                         const float* logit_scale, void* dqkv, float* dbias_table, float* dlogit_scale, int batch,
                         int Hp, int Wp, int C, int heads, int ws, int shift, scot_stream_t stream);
/* Attention probabilities of one block, [batch·nW, heads, N, N] fp32, recomputed from qkv and the forward's log-sum-exp — what
 * `output_attentions=True` returns (HF:443-455; the fused kernels never store them).  head_dim <= 64, N·head_dim·4 <= 64 KB. */
int scot_window_attn_probs(const void* qkv, int qkv_dt, const float* lse, const float* bias_table, const float* logit_scale,
                           float* probs, int batch, int Hp, int Wp, int C, int heads, int ws, int shift, scot_stream_t stream);

/* Continuous relative position bias MLP, HF:376-378,418-428 (coords table HF:457-476 is passed in). */
int scot_cpb_fwd(const float* coords, const float* w0, const float* b0, const float* w2, float* table, float* z, int ws,
                 int heads, scot_stream_t stream);
int scot_cpb_bwd(const float* coords, const float* w0, const float* b0, const float* w2, const float* z,
                 const float* dtable, float* dw0, float* db0, float* dw2, int ws, int heads, scot_stream_t stream);

/* Same MLP for MANY layers in one launch.  desc[l] = {w0_off, b0_off, w2_off, coords_off, ws, heads, tab_off, z_off}
 * (int32, offsets in floats from params / coords_base / tables / zbuf; grads uses the params offsets). */
int scot_cpb_fwd_batched(const float* params, const int* desc, int nlayers, int max_ws, const float* coords_base,
                         float* tables, float* zbuf, scot_stream_t stream);
/* backward of layers first .. first + count - 1; max_ws / max_heads: the largest window size and head count among them (sizes the
 * launch's LDS: the table-gradient tile of a layer is staged once per workgroup) */
/* scot_window_attn_bwd with dbias_table / dlogit_scale as `nrep` replicas (strides in floats): window w accumulates into replica w % nrep, so
 * the same-address atomic chains of a head's windows are nrep times shorter; scot_replica_reduce folds the replicas afterwards.
 * (Same reference lines as scot_window_attn_bwd: the autograd of modeling_swinv2.py:395-451 via scOT/model.py:166.) */
int scot_window_attn_bwd_rep(int compute, const void* qkv, const void* out_fwd, const void* dout, const float* lse, const float* bias_table,
                             const float* logit_scale, void* dqkv, float* dbias_table, float* dlogit_scale, int batch, int Hp, int Wp, int C,
                             int heads, int ws, int shift, int nrep, size_t rep_stride_tab, size_t rep_stride_ls, hipStream_t stream);
/* dst[d.dst_off + j] += sum_{r0 <= r < nrep} rep[r*stride + d.src_off + j] for j < d.count, for each of the n entries d = (src_off, dst_off,
 * count) of desc (int32, device); max_count = the largest count. */
int scot_replica_reduce(const float* rep, int r0, int nrep, size_t stride, const int* desc, int n, int max_count, float* dst,
                        hipStream_t stream);
int scot_cpb_bwd_batched(const float* params, const int* desc, int first, int count, int max_ws, int max_heads,
                         const float* coords_base, const float* zbuf, const float* dtables, float* grads, scot_stream_t stream);

/* ConditionalLayerNorm / LayerNorm (+ fused residual), ref:135-160, res-post-norm ref:570,574.
 * sample_scale (optional, one float per sample): out = resid + s_b * norm(x) — Swinv2DropPath (HF:565-586) on the normed
 * branch, s_b = mask_b / keep_prob drawn by the caller; the backward applies the same s_b to dout.
 * scot_cln_bwd mode: 0 = dx and the four parameter gradients (+=), 1 = dx only, 2 = parameter gradients only (dx may be
 * NULL) — lets the caller keep the dependent chain a pure stream and take the column reductions on another stream. */
int scot_cln_fwd(const void* x, int x_dt, const void* resid, int res_dt, void* out, int out_dt, void* out2, int out2_dt,
                 float* mean, float* rstd,
                 const float* time, const float* gw_w, const float* gw_b, const float* bw_w, const float* bw_b, int rows,
                 int rows_per_sample, int C, float eps, const float* sample_scale, scot_stream_t stream);
int scot_cln_bwd(const void* dout, int dout_dt, const void* x, int x_dt, const float* mean, const float* rstd,
                 const float* time, const float* gw_w, const float* gw_b, void* dx, int dx_dt, float* d_gw_w,
                 float* d_gw_b, float* d_bw_w, float* d_bw_b, float* d_xbias, int rows, int rows_per_sample, int C,
                 void* workspace, size_t ws_bytes, const float* sample_scale, int mode, scot_stream_t stream);
/* out2: optional second copy of the output in the next GEMM's operand dtype; d_xbias: optional += Σ_rows dx.
 * mode 3 (the deep stages' small row counts: rows <= 8192, C % 64 == 0, 128 <= C <= 1536): dx as in mode 1, every wave owning ONE
 * pass of rows, plus the block's partial parameter-gradient sums written to `workspace` (>= scot_cln_bwd_workspace_bytes, 16-byte
 * aligned; d_* unused, may be NULL) — no atomics on the dependent chain.  scot_cln_bwd_finish adds the partials into the four (two
 * without conditioning) parameter gradients, which must be contiguous [weight.weight | weight.bias | bias.weight | bias.bias] as they
 * are in the parameter arena; it may run on any stream ordered behind the mode-3 call (the engine: its weight-gradient stream).
 * Returns -3 where the form does not apply (workspace query: 0). */
size_t scot_cln_bwd_workspace_bytes(int rows, int rows_per_sample, int C, int conditional);
int scot_cln_bwd_finish(const void* partial, int rows, int rows_per_sample, int C, float* d_gw_w, float* d_gw_b, float* d_bw_w,
                        float* d_bw_b, scot_stream_t stream);

/* EXPERIMENTAL (off unless SCOT_FUSED_MLP=1; see poseidon_amd/csrc/mlp_fused.hip) — the MLP half of a ScOTLayer in one launch:
 *   z = gelu(h16·W1^T + b1)·W2^T + b2   (Swinv2Intermediate + Swinv2Output, HF modeling_swinv2.py:533-561)
 *   out = h + s_b·CLN(z), out16 = bf16(out)   (res-post-norm, reference scOT/model.py:566-579)
 * h16 [M,C] bf16, h [M,C] fp32, W1 [hid,C] bf16, W2 [C,hid] bf16; training also stores act = gelu(u), dact = gelu'(u)
 * ([M,hid] bf16), z [M,C] fp32, mean/rstd [M] (all-or-nothing per pair; NULL in inference).  bf16 operands only,
 * C in {96, 192}: anything else returns SCOT_ERR_UNSUPPORTED and the caller runs scot_gemm x2 + scot_cln_fwd. */
int scot_mlp_block_fwd(const void* h16, const float* h, const void* W1, const float* b1, const void* W2, const float* b2,
                       float* out, void* out16, void* act, void* dact, float* z, float* mean, float* rstd,
                       const float* time, const float* gw_w, const float* gw_b, const float* bw_w, const float* bw_b,
                       const float* sample_scale, int M, int rows_per_sample, int C, int hid, float eps,
                       scot_stream_t stream);
/* EXPERIMENTAL, same status: the dependent chain of that block's backward in one launch —
 *   dz = CLN_bwd(s_b·g; z, mean, rstd) (+= the four cond-LN parameter gradients), du = (dz·W2) ⊙ dact, g_out = g + du·W1
 * (replaces scot_cln_bwd + two dgrad scot_gemm calls; the weight gradients remain scot_gemm(TN) on dz / du).
 * g_out may alias g.  Requires rows_per_sample % 64 == 0 (one conditioning time per workgroup). */
int scot_mlp_block_bwd(const float* g, float* g_out, const float* z, const float* mean, const float* rstd,
                       const float* time, const float* gw_w, const float* gw_b, const float* sample_scale,
                       const void* dact, const void* W1, const void* W2, void* dz, void* du, float* d_gw_w,
                       float* d_gw_b, float* d_bw_w, float* d_bw_b, int M, int rows_per_sample, int C, int hid,
                       scot_stream_t stream);
/* EXPERIMENTAL, same status: the tail of the attention half with the layer norm in the GEMM epilogue —
 *   out = resid + s_b·CLN(a·W^T + bias), out16 = bf16(out)   (Swinv2SelfOutput, HF modeling_swinv2.py:478-489, + res-post-norm,
 *   reference scOT/model.py:560-565);  a [M,C] bf16, W [C,C] bf16 (N x K);  training also stores z = a·W^T + bias and mean/rstd.
 * scot_proj_cln_bwd: dz = CLN_bwd(s_b·g; z, mean, rstd) (+= the four cond-LN parameter gradients), da = dz·W. */
int scot_proj_cln_fwd(const void* a, const void* W, const float* bias, const float* resid, float* out, void* out16, float* z,
                      float* mean, float* rstd, const float* time, const float* gw_w, const float* gw_b, const float* bw_w,
                      const float* bw_b, const float* sample_scale, int M, int rows_per_sample, int C, float eps,
                      scot_stream_t stream);
int scot_proj_cln_bwd(const float* g, const float* z, const float* mean, const float* rstd, const float* time,
                      const float* gw_w, const float* gw_b, const float* sample_scale, const void* W, void* dz, void* da,
                      float* d_gw_w, float* d_gw_b, float* d_bw_w, float* d_bw_b, int M, int rows_per_sample, int C,
                      scot_stream_t stream);
/* scot_mlp_block_bwd followed by scot_proj_cln_bwd on its g_out, for the same rows, in ONE launch (the backward of HF:533-561 +
 * ref:566-579 and of HF:478-489 + ref:560-565 along the dependent chain): g_out is still written (the qkv dgrad accumulates
 * into it) but not re-read.  Suffix 2 = the MLP half's norm (layernorm_after), 1 = the attention half's (layernorm_before);
 * arguments as in the two entry points above.  C in {96, 192}, rows_per_sample % 64 == 0; returns -3 otherwise.
 * C = 48 with hid = 192 (Poseidon-T / -S stage 0, ref train.py:35-47) in the stored-gelu' form without the qkv prologue (dact != NULL, dqkv == NULL).
 * Round 3 — the form without 4C-wide tensors in HBM: dact == NULL makes the kernel RECOMPUTE gelu'(u) from u = h16·W1^T + b1
 * (h16 [M, C] 16-bit and b1 [hid] then required); du == NULL: the product dz2·W2 ⊙ gelu'(u) is not stored (scot_wgrad_mlp recomputes
 * it); z_dt: dtype of z1 / z2 (0 = fp32, 1 = the 16-bit operand format); partial2 / partial1 (both or neither): instead of 4·C
 * atomics per workgroup and norm, the workgroup's column sums go to row `workgroup` of a [scot_block_tail_workgroups()][4·Cp]
 * matrix ([t·dγ | dγ | t·dβ | dβ], each Cp = C rounded up to a multiple of 64 floats wide with a zero pad — the stride of the four
 * tensors in the parameter arena; [2·Cp] = [dγ | dβ] without conditioning) that scot_partial_colsum adds into the parameter
 * gradients on any stream ordered behind this call; the d_* pointers are then unused. */
int scot_block_tail_bwd(const float* g, float* g_out, const void* z2, const float* mean2, const float* rstd2, const float* gw_w2,
                        const float* gw_b2, const float* sscale2, const void* dact, const void* W1, const void* W2, void* dz2, void* du,
                        float* d_gw_w2, float* d_gw_b2, float* d_bw_w2, float* d_bw_b2, const void* z1, const float* mean1,
                        const float* rstd1, const float* gw_w1, const float* gw_b1, const float* sscale1, const void* Wo, void* dz1,
                        void* da, float* d_gw_w1, float* d_gw_b1, float* d_bw_w1, float* d_bw_b1,
                        const void* dqkv, const void* Wqkv /* optional prologue, both or neither: g += dqkv[M,3C] · Wqkv[3C,C] in place
                        (needs g_out == g) = the qkv projection's data gradient (HF:396-410) of the layer processed before */,
                        const void* h16, const float* b1, int z_dt, float* partial2, float* partial1,
                        const float* time, int M, int rows_per_sample, int C, int hid, scot_stream_t stream);
int scot_block_tail_workgroups(int M, int rows_per_sample, int C);      /* rows of partial2 / partial1 (0: shapes not covered) */
int scot_partial_colsum(const float* partial, int nblk, int ncol, float* out, scot_stream_t stream);   /* out[j] += Σ_b partial[b][j] */
/* the same for n <= 32 matrices in ONE launch (HOST arrays of device pointers / sizes, read before the call returns): the engine
 * finishes a whole stage's norm backwards (scot_cln_bwd mode 3, scot_block_tail_bwd partial rows) with it */
int scot_partial_colsum_batch(int n, const float* const* partial, const int* nblk, const int* ncol, float* const* out,
                              scot_stream_t stream);
/* scot_proj_cln_fwd followed by scot_mlp_block_fwd on its output, for the same rows, in ONE launch (HF:478-489 + ref:560-565, then
 * HF:533-561 + ref:566-579): h / h16 are written (the backward reads them) but not re-read.  Suffix 1 = attention half's norm
 * (layernorm_before), 2 = MLP half's (layernorm_after); arguments as in the two entry points.  C in {96, 192}; -3 otherwise.
 * Also C = 48 with hid = 192 (Poseidon-T / -S stage 0): only this whole-tail form exists at that width.
 * act / dact NULL (and z / statistics given): training without the 4C-wide saves (the backward recomputes, see above); z_dt: dtype
 * z1 / z2 are stored in (0 = fp32, 1 = 16-bit: only the norm backward's x-hat reads them). */
int scot_block_tail_fwd(const void* a, const void* Wo, const float* bo, const float* x, float* h, void* h16, void* z1, float* mean1,
                        float* rstd1, const float* gw_w1, const float* gw_b1, const float* bw_w1, const float* bw_b1,
                        const float* sscale1, const void* W1, const float* b1, const void* W2, const float* b2, float* out, void* out16,
                        void* act, void* dact, void* z2, float* mean2, float* rstd2, const float* gw_w2, const float* gw_b2,
                        const float* bw_w2, const float* bw_b2, const float* sscale2,
                        const void* Wqkv, const float* bqkv, void* qkv /* optional epilogue (Wqkv and qkv both or neither): the NEXT
                        layer's fused q/k/v projection qkv[M,3C] = out16 · Wqkv[3C,C]^T + bqkv (HF:396-410) on the rows just produced */,
                        int z_dt, const float* time, int M, int rows_per_sample, int C, int hid, float eps, scot_stream_t stream);

/* Step tape (csrc/host_tape.hip).  The reference issues a training step op by op from Python (trainer.py via HF Trainer / autograd);
 * this library's engine records the step once as a list of the calls of THIS header with their final arguments and replays the list.
 * scot_tape_replay walks such a list without returning to the host language between launches: prog = records of 64-bit words
 * {entry point address, n_int, n_flt, the integer-class arguments in order (pointers, int, size_t, the stream handle), the float
 * arguments as raw IEEE-754 single bits}; n_int <= 48, n_flt <= 8.  Returns 0, or the first non-zero status with *fail_entry = the
 * index of the record that returned it (nothing after it is issued).  The host-side operations a step needs between launches are
 * entry points too, so that they can be records: memset / device-to-device copy on a stream, event record, stream-wait-event (event = a
 * hipEvent_t). */
int scot_memset_async(void* p, int byte, size_t n, scot_stream_t stream);
int scot_memcpy_async(void* dst, const void* src, size_t n, scot_stream_t stream);
int scot_event_record(void* event, scot_stream_t stream);
int scot_stream_wait_event(scot_stream_t stream, void* event);
int scot_tape_replay(const unsigned long long* prog, size_t n_words, int* fail_entry);

/* The fc1 / fc2 weight and bias gradients of a ScOTLayer's MLP WITHOUT gelu(u), gelu'(u), du in HBM (csrc/wgrad_mlp.hip; autograd of
 * HF:545-548, 558-561): per token slice and hidden chunk the kernel recomputes u = h16·W1^T + b1 and dz·W2 and feeds gelu(u) / du from
 * registers into dW2 += dz^T·gelu(u), dW1 += du^T·h16 (+ both bias gradients).  h16, dz [M, C] 16-bit; W1 [hid, C]; W2T [hid, C] = W2^T;
 * dW1 [hid, C] | db1 [hid] | dW2 [C, hid] | db2 [C] must be CONTIGUOUS in this order (the gradient arena's layout); workspace >=
 * scot_wgrad_mlp_workspace_bytes, 32-byte aligned.  C in {96, 192}, hid = 4C; -3 otherwise.  mode / grad_scale: how dW1 and dW2 meet the
 * arena (scot_wgrad_group's modes); db1 / db2 are always plain accumulations. */
size_t scot_wgrad_mlp_workspace_bytes(int M, int C, int hid);
int scot_wgrad_mlp(const void* h16, const void* dz, const void* W1, const float* b1, const void* W2T, float* dW1, float* db1, float* dW2,
                   float* db2, int M, int C, int hid, void* workspace, size_t ws_bytes, int mode, const float* grad_scale,
                   scot_stream_t stream);

/* Data movement */
int scot_add(const void* a, int a_dt, const void* b, int b_dt, void* out, int out_dt, size_t n, size_t period,
             scot_stream_t stream);                                                   /* ref:847-849,1175-1177,361 */
int scot_batch_sum(const void* x, int x_dt, float* out, int batch, size_t period, scot_stream_t stream);
int scot_copy2d(const void* src, int s_dt, void* dst, int d_dt, int B, int Hs, int Ws, int Hd, int Wd, int C,
                scot_stream_t stream);                                                /* ref:480-498, 563-566 */
int scot_space_to_depth(const void* fine, const void* fine2, int f_dt, void* coarse, int c_dt, int B, int H, int W, int C,
                        int order, scot_stream_t stream);                             /* ref:672-704 (order 0) */
int scot_depth_to_space(const void* coarse, int c_dt, void* fine, int f_dt, int B, int H, int W, int H2, int W2, int C,
                        int order, scot_stream_t stream);                             /* ref:748-756 (order 1) */
int scot_patchify(const float* img, void* cols, int c_dt, int B, int Cc, int H, int W, int p, scot_stream_t stream); /* ref:286-308 */
int scot_unpatchify(const void* cols, int c_dt, const float* bias, float* img, int B, int Cc, int H, int W, int gh, int gw,
                    int p, scot_stream_t stream);                                      /* ref:616-621,632-643 */
int scot_nchw_channel_sum(const float* x, float* out, int B, int Cc, int HW, scot_stream_t stream);
int scot_colsum(const void* x, int x_dt, const void* y, int y_dt, float* out, int M, int N, int ld, scot_stream_t stream);
int scot_scale_residual(const void* y, int y_dt, const float* scale, const void* resid, int r_dt, void* out, int o_dt,
                        size_t rows, int N, scot_stream_t stream);                     /* ref:212-216 */

/* Stencils */
int scot_dwconv7(const void* x, int x_dt, const float* w, const float* bias, void* y, int y_dt, int B, int H, int W, int C,
                 int flip, scot_stream_t stream);                                       /* ref:178-180,206 */
int scot_dwconv7_wgrad(const void* dy, int dy_dt, const void* x, int x_dt, float* dw, float* db, int B, int H, int W, int C,
                       scot_stream_t stream);
int scot_conv5(const float* in, const float* w, float* out, int B, int Cc, int H, int W, int transpose,
               scot_stream_t stream);                                                   /* ref:623-630,647 */
int scot_conv5_wgrad(const float* dout, const float* in, float* dw, int B, int Cc, int H, int W, scot_stream_t stream);

/* Head finalisation + loss, ref:1411-1484 */
int scot_head_finalize(float* pred, const float* pv, int pv_ch, const float* labels, const unsigned char* mask,
                       int mask_full, const int* group_of_channel, float* sums, int B, int Cc, int HW, int p,
                       scot_stream_t stream);
int scot_loss_finish(const float* sums, const float* counts, int G, int normalized, float* loss, scot_stream_t stream);
int scot_loss_bwd(const float* pred, const float* labels, const unsigned char* mask, int mask_full,
                  const int* group_of_channel, const float* sums, const float* counts, int G, int normalized,
                  const float* dloss, float* dpred, int B, int Cc, int HW, int p, scot_stream_t stream);

/* ---- optimizer step over the flat arenas (reference scOT/trainer.py:295-445 builds the groups, HF Trainer
 * clips with clip_grad_norm_ and steps torch.optim.AdamW).  map8[i] = parameter-group id (0..7) of arena elements 8i..8i+7, 255 =
 * not a parameter (alignment padding, the key-bias slot of the fused qkv bias).  n = arena size in floats (multiple of 8). */
int scot_optim_blocks(size_t n);                       /* floats of `partial` scratch scot_grad_sqnorm needs */
int scot_grad_sqnorm(const float* grad, const unsigned char* map8, size_t n, float* partial, scot_stream_t stream);
int scot_clip_coef(const float* partial, int nblocks, float max_norm /* <= 0: no clipping */,
                   float* out3 /* {coef, total_norm, 1 if the norm is not finite else 0} */, scot_stream_t stream);
/* clip: the three floats of scot_clip_coef (NULL: no clipping, no skip) — the update is SKIPPED when clip[2] != 0 (gradients that
 * overflowed under the fp16 build's gradient scale; GradScaler.step semantics, decided on the device from the norm of the REDUCED
 * gradient, so every data-parallel rank decides alike).  step_state: device int[2] {steps applied, steps skipped}: Adam's step number
 * is applied + 1 (NULL: `step`).  shadow16 (optional): n elements in the library's 16-bit operand format, receives the updated
 * parameters in the same pass (the GEMM operand copy the next forward reads). */
int scot_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const unsigned char* map8, size_t n,
                    const float* lr /* host[ngroups] */, const float* wd /* host[ngroups] */, int ngroups, float beta1, float beta2,
                    float eps, int step, const float* clip, const int* step_state, void* shadow16, scot_stream_t stream);
/* After scot_adamw_step: step_state[0 or 1] += 1; with scale_state {S, 1/S, clean steps} and interval > 0 the gradient scale of the
 * fp16 build follows torch.cuda.amp.GradScaler: S *= backoff after a skipped step, S *= growth after `interval` applied steps. */
int scot_optim_finish(int* step_state, const float* clip, float* scale_state, float growth, float backoff, int interval,
                      float max_scale, scot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
