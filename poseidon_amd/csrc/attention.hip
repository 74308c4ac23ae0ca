// Shifted-window cosine attention with continuous relative-position bias — forward and backward.
//
// Reference semantics (transformers swinv2 `Swinv2SelfAttention.forward`, HF:389-455, called from reference
// scOT/model.py:522-559): for every window and head
//     S = normalize(q) · normalize(k)^T * exp(min(logit_scale, ln 100)) + 16·sigmoid(CPB)[rel(i,j)] + 2·mask(i,j)
//     O = softmax(S) · v
// The roll(-s) + window_partition + window_reverse + roll(+s) round trip (model.py:522-559, HF:146-166) is folded
// into the token index  tok(n) = ((wy·ws + n/ws + s) mod H)·W + ((wx·ws + n%ws + s) mod W); the additive shift
// mask (model.py:442-478) is evaluated analytically from region ids; S is never materialised in HBM.
//
// Workgroup = one (window, head); 4 waves; each wave owns 16-query blocks.  K (normalised) and V of the window
// live in LDS; the whole logit row of a query (<=256 keys) lives in the accumulator registers of one wave:
// S^T tiles = mfma(A = Kn[16 keys x d], B = Qn^T) put ONE query per lane column (lane&15) with its keys spread
// over (lane>>4, reg, tile) → row max / sum need only two xor-shuffles (16, 32), and the probabilities are
// already in MFMA A-operand order for P·V (no LDS round trip, no permutes).
#include "attention.h"
// Tile-pair loop of the backward at NT = 4 (8x8 and 7x7 windows; 1536 four-wave workgroups at stage 2): unrolled twice it is fully unrolled and
// holds 178 VGPRs = 2 workgroups per CU = three rounds of the ~12 us per-workgroup chain; rolled it needs 118 = 4 per CU.
#ifndef SCOT_NT4_UNROLL
#define SCOT_NT4_UNROLL 1
#endif

// ================================================================================================= forward
template <typename CT, int HD, int NT>  // NT = number of 16-key tiles (even), NP = 16*NT >= N
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
  constexpr int NP = NT * 16, KS = (HD + 31) / 32, DT = HD / 16, pitch = row_pitch<HD, CT>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CT* Kn = (CT*)smem;
  CT* Vs = Kn + NP * pitch;
  float* tab = (float*)(Vs + NP * pitch);
  const int ws = p.ws, N = ws * ws, TW = 2 * ws - 1, TS = TW * TW;
  int* rid = (int*)(tab + ((TS + 3) & ~3));
  int* tok = rid + NP;   // token index of every window position (the roll/partition index math, done once)

  const int win = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nwv = nthr >> 6;   // 1, 2 or 4 waves per (window, head): small windows leave waves idle otherwise
  const int ld = 3 * p.C;

  for (int i = tid; i < NP; i += nthr) { rid[i] = pos_info(p, win, i, N); tok[i] = i < N ? win_token(p, win, i) : 0; }
  for (int i = tid; i < TS; i += nthr) tab[i] = p.bias_table[h * TS + i];
  __syncthreads();
  stage_rows<CT, HD, NP>(Kn, p.qkv, ld, p.C + h * HD, tok, N, true, tid);
  stage_rows<CT, HD, NP>(Vs, p.qkv, ld, 2 * p.C + h * HD, tok, N, false, tid);
  __syncthreads();

  const float scale = __expf(fminf(p.logit_scale[h], 4.605170185988092f));  // exp(min(ls, ln 100)), HF:416
  const int cen = (ws - 1) * TW + ws - 1;
  const int g = lane >> 4, qc = lane & 15;

  for (int qb = wave; qb * 16 < N; qb += nwv) {
    const int q0 = qb * 16;
    Frag<CT> qf[KS];
    load_rows_frag<CT, HD>(qf, p.qkv, ld, h * HD, tok, q0, N, true, lane);
    const int q = q0 + qc;
    const bool qvalid = q < N;
    const int qinfo = rid[min(q, NP - 1)];
    const int qoff = (qinfo & 0xfffff) + cen, qrid = qinfo >> 20;

    f32x4_t s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) mma16(s[t], lds_frag_kc(Kn, pitch, t * 16, kk * 32, lane), qf[kk]);
      if (NT >= 8 && (t & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from hoisting all tiles' LDS reads
    }
    // logits: lane holds query q (=column), keys t*16 + g*4 + r
    // branch-free: every bias-table gather is in bounds for any encoded position (padding rows encode position 0),
    // so all 64 LDS gathers of a query row can be in flight together; masking is done with selects afterwards.
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int4 ki = *(const int4*)&rid[t * 16 + g * 4];
      const int kia[4] = {ki.x, ki.y, ki.z, ki.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int krid = kia[r] >> 20;
        float v = s[t][r] * scale + tab[qoff - (kia[r] & 0xfffff)];
        v = (krid != qrid) ? v - 200.0f : v;  // the -100 mask is added twice in the installed oracle (HF:433-436)
        v = (krid == 15 || !qvalid) ? -3.0e38f : v;
        s[t][r] = v;
        m = fmaxf(m, v);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[t][r] - m);
        s[t][r] = e;
        l += e;
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (qvalid && g == 0 && p.lse) p.lse[((size_t)win * p.heads + h) * N + q] = m + __logf(l);

    // O = P · V : A = P (already in A-operand order), B = V rows read with the transposing fragment read
    f32x4_t o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < NT / 2; ++tp) {
      float pv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { pv[r] = s[2 * tp][r] * inv; pv[r + 4] = s[2 * tp + 1][r] * inv; }
      const Frag<CT> pf = frag_from_f32<CT>(pv);
#pragma unroll
      for (int d = 0; d < DT; ++d)
        mma16(o[d], pf, lds_frag_ks(Vs, pitch, d * 16, (2 * tp) * 16 + g * 4, (2 * tp + 1) * 16 + g * 4, lane, p.use_tr));
      if (NT >= 8 && (tp & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    // o[d]: col = feature d*16 + (lane&15), rows = query q0 + g*4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = q0 + g * 4 + r;
      if (qq < N) {
        const size_t base = (size_t)tok[qq] * p.C + h * HD + (lane & 15);
#pragma unroll
        for (int d = 0; d < DT; ++d) st1(p.out, ct_traits<CT>::dtype, base + d * 16, o[d][r]);
      }
    }
  }
}

// ================================================================================================= backward
// Phase B (queries on lane columns, S^T tiles): recompute P, dP = dO·V^T, delta = rowsum(P∘dP), dS = P∘(dP-delta),
//   d bias-table (LDS histogram → one atomic pass), d logit_scale, dQn = scale·dS·Kn → dq through the normalisation.
// Phase A (keys on lane columns, S tiles): recompute P, dS; dV = P^T·dO, dKn = scale·dS^T·Qn → dk.
// LDS holds two [NP][HD] tiles that are re-filled between the phases (Kn,V then Qn,dO).
// The un-normalised rows again, in the layout of the accumulators (row n0 + (lane>>4)*4 + r, feature d*16 + (lane&15)).  Loaded
// separately from (and well before) the stores of normalize_bwd_store: behind a store to `dst` the compiler must assume
// aliasing and would issue one dependent global round trip per row — 4 of the ~10 serial round trips that made the backward of
// the 8x8 / 4x4 windows take 25 us for a few kFLOP.
template <typename CT, int HD>
__device__ __forceinline__ void normalize_bwd_load(float (&x)[4][HD / 16], const void* src, int ld, int col, const int* tokt, int n0, int N,
                                                   int lane) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + g * 4 + r;
    const bool valid = n < N;
    const size_t tok = valid ? (size_t)tokt[n] : 0;
#pragma unroll
    for (int d = 0; d < HD / 16; ++d) x[r][d] = valid ? ld1(src, ct_traits<CT>::dtype, tok * ld + col + d * 16 + c) : 0.f;
  }
}
template <typename CT, int HD>
__device__ __forceinline__ void normalize_bwd_store(const f32x4_t (&acc)[HD / 16], float mul, const float (&x)[4][HD / 16], int ld,
                                                    void* dst, int dcol, const int* tokt, int n0, int N, int lane) {
  // acc[d][r]: gradient wrt the NORMALISED row n0 + (lane>>4)*4 + r, feature d*16 + (lane&15) (times `mul`).
  // y = x / max(|x|, eps):  dx = (g - y (y·g)) / |x|   (|x| >= eps),   dx = g / eps otherwise.
  constexpr int DT = HD / 16;
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + g * 4 + r;
    const bool valid = n < N;
    const size_t tok = valid ? (size_t)tokt[n] : 0;
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) ss += x[r][d] * x[r][d];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) ss += __shfl_xor(ss, o, 64);
    const float nrm = sqrtf(ss);
    const bool clamped = nrm < 1e-12f;
    // (16-bit results: g / eps = 1e12 g is out of binary16's range, and the Inf would meet the all-zero input row that produced the clamped
    //  norm — a padded window token under the bias-free key projection — as 0 · Inf = NaN in the weight gradient.  Its true contribution
    //  there is 0 · 1e12 g = 0, and the data gradient of a padded token is cropped: the 16-bit builds store 0 for a clamped row.)
    const float rn = (clamped && sizeof(CT) == 2) ? 0.f : 1.0f / fmaxf(nrm, 1e-12f);
#pragma unroll
    for (int d = 0; d < DT; ++d) dot += (x[r][d] * rn) * (acc[d][r] * mul);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) dot += __shfl_xor(dot, o, 64);
    if (clamped) dot = 0.f;
    if (valid) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
        st1(dst, ct_traits<CT>::dtype, tok * ld + dcol + d * 16 + c, rn * (acc[d][r] * mul - (x[r][d] * rn) * dot));
    }
  }
}

// ---- backward, kernel 1 of 2: dQ, d bias-table, d logit_scale.  Queries on lane columns (S^T tiles); K (normalised) and V in LDS.
// delta = rowsum(dO ∘ O) comes from the forward output, so every 16x32 block of scores is consumed as soon as it is
// produced (no 128-register S/dP pair): S^T, dP^T -> P -> dS -> {table histogram, dQ += dS·Kn}.
template <typename CT, int HD, int NT>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& p) {
  constexpr int NP = NT * 16, KS = (HD + 31) / 32, DT = HD / 16, pitch = row_pitch<HD, CT>(), TPU = NT == 4 ? SCOT_NT4_UNROLL : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CT* X = (CT*)smem;            // Kn
  CT* Y = X + NP * pitch;       // V
  const int ws = p.ws, N = ws * ws, TW = 2 * ws - 1, TS = TW * TW, TSP = (TS + 3) & ~3;
  float* tab = (float*)(Y + NP * pitch);
  // the table-gradient histogram lives in LDS as DOUBLES: ds_add_f64 issues at full rate on gfx950 while ds_add_f32 is
  // serialised at ~3 clk per active lane per CU (tools/probes/lds_atomic_probe.hip) — with fp32 the histogram alone was
  // half of this kernel
  double* dtab = (double*)(tab + TSP);
  int* rid = (int*)(dtab + TSP);
  int* tok = rid + NP;
  float* red = (float*)(tok + NP);  // [4]

  const int win = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nwv = nthr >> 6;   // 1, 2 or 4 waves per (window, head): small windows leave waves idle otherwise
  const int ld = 3 * p.C, g = lane >> 4, lc = lane & 15;

  for (int i = tid; i < TS; i += nthr) { tab[i] = p.bias_table[h * TS + i]; dtab[i] = 0.0; }
  for (int i = tid; i < NP; i += nthr) { rid[i] = pos_info(p, win, i, N); tok[i] = i < N ? win_token(p, win, i) : 0; }
  __syncthreads();
  stage_rows<CT, HD, NP>(X, p.qkv, ld, p.C + h * HD, tok, N, true, tid);
  stage_rows<CT, HD, NP>(Y, p.qkv, ld, 2 * p.C + h * HD, tok, N, false, tid);
  __syncthreads();

  const float scale = __expf(fminf(p.logit_scale[h], 4.605170185988092f));
  const int cen = (ws - 1) * TW + ws - 1;
  // d logit_scale = Σ_qk dS·cos·scale with Σ_k dS = 0 per query: a heavily cancelling sum.  dS uses delta from the
  // stored (rounded) forward output; the row sums D = Σ_k P·dP and B = Σ_k P·cos taken here in fp32 put the exact
  // cancellation back:  Σ_k P (dP - D) cos = Σ_k dS·cos + (delta - D)·B.
  float dls = 0.f;

  for (int qb = wave; qb * 16 < N; qb += nwv) {
    const int q0 = qb * 16;
    // every global load of the query block goes out first (q, dO, O here; lse and the epilogue's rows below): interleaved with
    // their consumers they were dependent round trips
    Frag<CT> qf[KS], gf[KS];
    float qv[KS][8], dov[KS][8], ov[KS][8];
    load_rows_f32<CT, HD>(qv, p.qkv, ld, h * HD, tok, q0, N, lane);
    load_rows_f32<CT, HD>(dov, p.dout, p.C, h * HD, tok, q0, N, lane);
    load_rows_f32<CT, HD>(ov, p.ofwd, p.C, h * HD, tok, q0, N, lane);
    float xq[4][DT];
    normalize_bwd_load<CT, HD>(xq, p.qkv, ld, h * HD, tok, q0, N, lane);
    rows_to_frag<CT, HD>(qf, qv, true);
    float accD = 0.f, accB = 0.f;
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += dov[kk][j] * ov[kk][j];
      gf[kk] = frag_from_f32<CT>(dov[kk]);
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const int q = q0 + lc;
    const bool qvalid = q < N;
    const int qinfo = rid[min(q, NP - 1)];
    const int qoff = (qinfo & 0xfffff) + cen, qrid = qinfo >> 20;
    const float qlse = qvalid ? p.lse[((size_t)win * p.heads + h) * N + q] : 3.0e38f;
    f32x4_t dq[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) dq[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#pragma unroll TPU
    for (int tp = 0; tp < NT / 2; ++tp) {
      float ds8[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * tp + half;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          mma16(s, lds_frag_kc(X, pitch, t * 16, kk * 32, lane), qf[kk]);
          mma16(dp, lds_frag_kc(Y, pitch, t * 16, kk * 32, lane), gf[kk]);
        }
        const int4 ki = *(const int4*)&rid[t * 16 + g * 4];
        const int kia[4] = {ki.x, ki.y, ki.z, ki.w};
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int krid = kia[r] >> 20;
          float v = s[r] * scale + tab[qoff - (kia[r] & 0xfffff)];
          v = (krid != qrid) ? v - 200.0f : v;
          float pr = __expf(v - qlse);
          pr = (krid == 15 || !qvalid) ? 0.f : pr;
          ds[r] = pr * (dp[r] - delta);
          ds8[half * 4 + r] = ds[r];
          accD = fmaf(pr, dp[r], accD);
          accB = fmaf(pr, s[r], accB);
          dls = fmaf(ds[r], s[r], dls);
        }
        if (ws == 16) {
          // one window row of queries against one window row of keys: the 16x16 block of dS feeds the 31 entries
          // (dx = q - k) of ONE table row.  DPP row shifts fold the lane's four keys along the anti-diagonal before the
          // LDS atomics (folding the four 16-lane rows too, with two ds_bpermute, measured slower: 114 vs 100 us).
          const float a = ds[0] + dpp_row<0x101>(ds[1]) + dpp_row<0x102>(ds[2]) + dpp_row<0x103>(ds[3]);
          const float bt = dpp_row<0x11F>(ds[1]) + dpp_row<0x11E>(ds[2]) + dpp_row<0x11D>(ds[3]);
          const int ia = qoff - (kia[0] & 0xfffff);
          atomicAdd(&dtab[ia], (double)a);
          if (lc >= 13) atomicAdd(&dtab[ia - 16], (double)bt);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(&dtab[qoff - (kia[r] & 0xfffff)], (double)ds[r]);
        }
      }
      const Frag<CT> df = frag_from_f32<CT>(ds8);
#pragma unroll
      for (int d = 0; d < DT; ++d)
        mma16(dq[d], df, lds_frag_ks(X, pitch, d * 16, (2 * tp) * 16 + g * 4, (2 * tp + 1) * 16 + g * 4, lane, p.use_tr));
    }
    accD += __shfl_xor(accD, 16, 64); accD += __shfl_xor(accD, 32, 64);
    accB += __shfl_xor(accB, 16, 64); accB += __shfl_xor(accB, 32, 64);
    if (g == 0 && qvalid) dls = fmaf(delta - accD, accB, dls);
    normalize_bwd_store<CT, HD>(dq, scale, xq, ld, p.out, h * HD, tok, q0, N, lane);
  }
  // d/dls [cos * exp(ls)] = cos * scale  (0 when clamped at ln 100, HF:416)
  dls = wave_sum(dls);
  if (lane == 0) red[wave] = dls;
  __syncthreads();
  const int rep = p.nrep > 1 ? win % p.nrep : 0;
  float* dbt = p.dbias_table + (size_t)rep * p.rep_stride_tab;
  for (int i = tid; i < TS; i += nthr) atomicAdd(&dbt[h * TS + i], (float)dtab[i]);
  if (tid == 0 && p.logit_scale[h] <= 4.605170185988092f) {
    float tot = 0.f;
    for (int w = 0; w < nwv; ++w) tot += red[w];
    atomicAdd(&p.dlogit_scale[(size_t)rep * p.rep_stride_ls + h], tot * scale);
  }
}

// ---- backward, kernel 2 of 2: dK, dV.  Keys on lane columns (S tiles); Q (normalised) and dO in LDS.
// Independent of kernel 1 (delta is recomputed from dO ∘ O while dO is staged), so the two can run concurrently.
template <typename CT, int HD, int NT>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs& p) {
  constexpr int NP = NT * 16, KS = (HD + 31) / 32, DT = HD / 16, pitch = row_pitch<HD, CT>(), TPU = NT == 4 ? SCOT_NT4_UNROLL : 2;
  constexpr int CPR = KS * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CT* X = (CT*)smem;            // Qn
  CT* Y = X + NP * pitch;       // dO
  const int ws = p.ws, N = ws * ws, TW = 2 * ws - 1, TS = TW * TW, TSP = (TS + 3) & ~3;
  float* tab = (float*)(Y + NP * pitch);
  float* lse = tab + TSP;       // [NP]
  float* delta = lse + NP;      // [NP]
  int* rid = (int*)(delta + NP);
  int* tok = rid + NP;

  const int win = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = blockDim.x, nwv = nthr >> 6;   // 1, 2 or 4 waves per (window, head): small windows leave waves idle otherwise
  const int ld = 3 * p.C, g = lane >> 4, lc = lane & 15;

  for (int i = tid; i < TS; i += nthr) tab[i] = p.bias_table[h * TS + i];
  for (int i = tid; i < NP; i += nthr) {
    rid[i] = pos_info(p, win, i, N);
    tok[i] = i < N ? win_token(p, win, i) : 0;
    lse[i] = i < N ? p.lse[((size_t)win * p.heads + h) * N + i] : 3.0e38f;
  }
  __syncthreads();
  stage_rows<CT, HD, NP>(X, p.qkv, ld, h * HD, tok, N, true, tid);
  // dO -> LDS and delta[n] = Σ_d dO[n][d]·O[n][d] in the same pass (CPR lanes per row)
  for (int c = tid; c < NP * CPR; c += nthr) {
    const int n = c / CPR, d8 = (c % CPR) * 8;
    float v[8], o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = 0.f; o[j] = 0.f; }
    if (n < N && d8 < HD) {
      ld8(p.dout, ct_traits<CT>::dtype, (size_t)tok[n] * p.C + h * HD + d8, v);
      ld8(p.ofwd, ct_traits<CT>::dtype, (size_t)tok[n] * p.C + h * HD + d8, o);
    }
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dot += v[j] * o[j];
#pragma unroll
    for (int of = 1; of < CPR; of <<= 1) dot += __shfl_xor(dot, of, 64);
    if ((c % CPR) == 0) delta[n] = dot;
    store8_ct(Y + n * pitch + d8, v);
  }
  __syncthreads();

  const float scale = __expf(fminf(p.logit_scale[h], 4.605170185988092f));
  const int cen = (ws - 1) * TW + ws - 1;

  for (int kb = wave; kb * 16 < N; kb += nwv) {
    const int k0 = kb * 16;
    Frag<CT> kf[KS], vf[KS];
    float kv[KS][8], vv[KS][8], xk[4][DT];
    load_rows_f32<CT, HD>(kv, p.qkv, ld, p.C + h * HD, tok, k0, N, lane);
    load_rows_f32<CT, HD>(vv, p.qkv, ld, 2 * p.C + h * HD, tok, k0, N, lane);
    normalize_bwd_load<CT, HD>(xk, p.qkv, ld, p.C + h * HD, tok, k0, N, lane);
    rows_to_frag<CT, HD>(kf, kv, true);
    rows_to_frag<CT, HD>(vf, vv, false);
    const int key = k0 + lc;
    const bool kvalid = key < N;
    const int kinfo = rid[min(key, NP - 1)];
    const int koff = kinfo & 0xfffff, krid = kinfo >> 20;

    f32x4_t dv[DT], dk[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) { dv[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dk[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

#pragma unroll TPU
    for (int tp = 0; tp < NT / 2; ++tp) {
      float pf8[8], df8[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * tp + half;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          mma16(s, lds_frag_kc(X, pitch, t * 16, kk * 32, lane), kf[kk]);   // rows = queries, col = key
          mma16(dp, lds_frag_kc(Y, pitch, t * 16, kk * 32, lane), vf[kk]);
        }
        const int4 qi = *(const int4*)&rid[t * 16 + g * 4];
        const int qia[4] = {qi.x, qi.y, qi.z, qi.w};
        const float4 ql = *(const float4*)&lse[t * 16 + g * 4];
        const float4 qd = *(const float4*)&delta[t * 16 + g * 4];
        const float qla[4] = {ql.x, ql.y, ql.z, ql.w}, qda[4] = {qd.x, qd.y, qd.z, qd.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qrid = qia[r] >> 20;
          float v = s[r] * scale + tab[(qia[r] & 0xfffff) + cen - koff];
          v = (qrid != krid) ? v - 200.0f : v;
          const bool ok = qrid != 15 && kvalid;
          const float pr = ok ? __expf(v - qla[r]) : 0.f;
          pf8[half * 4 + r] = pr;
          df8[half * 4 + r] = ok ? pr * (dp[r] - qda[r]) : 0.f;
        }
      }
      const Frag<CT> pf = frag_from_f32<CT>(pf8), df = frag_from_f32<CT>(df8);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const int klo = (2 * tp) * 16 + g * 4, khi = (2 * tp + 1) * 16 + g * 4;
        mma16(dv[d], pf, lds_frag_ks(Y, pitch, d * 16, klo, khi, lane, p.use_tr));   // dV  += P^T  · dO
        mma16(dk[d], df, lds_frag_ks(X, pitch, d * 16, klo, khi, lane, p.use_tr));   // dKn += dS^T · Qn
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk2 = k0 + g * 4 + r;
      if (kk2 < N) {
        const size_t base = (size_t)tok[kk2] * ld + 2 * p.C + h * HD + lc;
#pragma unroll
        for (int d = 0; d < DT; ++d) st1(p.out, ct_traits<CT>::dtype, base + d * 16, dv[d][r]);
      }
    }
    normalize_bwd_store<CT, HD>(dk, scale, xk, ld, p.out, p.C + h * HD, tok, k0, N, lane);
  }
}

// one launch for both halves (blockIdx.z): they are independent, see attention_w16.hip
template <typename CT, int HD, int NT>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(AttnArgs p) {
  if (blockIdx.z == 0) attn_bwd_dq_body<CT, HD, NT>(p);
  else attn_bwd_dkv_body<CT, HD, NT>(p);
}

// ================================================================================================= host side
extern int g_scot_use_tr;
int scot_attn_w16(const AttnArgs& a, int compute, int hd, int nwin, bool bwd, hipStream_t s);   // attention_w16.hip

template <typename CT, int HD, int NT>
static int launch_attn(const AttnArgs& a, int nwin, bool bwd, hipStream_t s) {
  constexpr int NP = NT * 16, pitch = row_pitch<HD, CT>();
  const int TS = (2 * a.ws - 1) * (2 * a.ws - 1), TSP = (TS + 3) & ~3;
  size_t sh = 2 * NP * pitch * sizeof(CT);
  const size_t sh_dq = sh + TSP * (sizeof(float) + sizeof(double)) + 2 * NP * sizeof(int) + 4 * sizeof(float);
  const size_t sh_dkv = sh + (TSP + 2 * NP) * sizeof(float) + 2 * NP * sizeof(int);
  if (bwd) sh = sh_dq > sh_dkv ? sh_dq : sh_dkv;
  else sh += TSP * sizeof(float) + 2 * NP * sizeof(int);
  if (sh > 160 * 1024) return SCOT_ERR_UNSUPPORTED;
  // waves per workgroup: a 4x4 window has ONE 16-query block — with 4 waves per workgroup three of them only held wave slots and
  // the 3072-workgroup grid needed three rounds of its ~10 us latency chain
  const int nw = NT <= 2 ? 1 : 4;   // measured (bwd, cold): 4x4 windows 22 vs 33 us with 1 wave; 8x8 windows best with 4
  dim3 grid(nwin, a.heads), block(64 * nw);
  if (bwd) {
    if (sh > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<CT, HD, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL((attn_bwd_kernel<CT, HD, NT>), dim3(nwin, a.heads, 2), block, sh, s, a);
  } else {
    if (sh > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<CT, HD, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL((attn_fwd_kernel<CT, HD, NT>), grid, block, sh, s, a);
  }
  return scot_check_launch();
}

template <typename CT, int HD>
static int dispatch_nt(const AttnArgs& a, int nwin, bool bwd, hipStream_t s) {
  const int N = a.ws * a.ws;
  if (N <= 32) return launch_attn<CT, HD, 2>(a, nwin, bwd, s);
  if (N <= 64) return launch_attn<CT, HD, 4>(a, nwin, bwd, s);
  if (N <= 128) return launch_attn<CT, HD, 8>(a, nwin, bwd, s);
  if (N <= 256) return launch_attn<CT, HD, 16>(a, nwin, bwd, s);
  return SCOT_ERR_UNSUPPORTED;
}

template <typename CT>
static int dispatch_hd(const AttnArgs& a, int hd, int nwin, bool bwd, hipStream_t s) {
  switch (hd) {
    case 16: return dispatch_nt<CT, 16>(a, nwin, bwd, s);
    case 32: return dispatch_nt<CT, 32>(a, nwin, bwd, s);
    case 64: return dispatch_nt<CT, 64>(a, nwin, bwd, s);
    default: return SCOT_ERR_UNSUPPORTED;  // head_dim = embed_dim/3 ∈ {16, 32, 64} for T/S, B, L
  }
}

static int fill_args(AttnArgs& a, int batch, int Hp, int Wp, int C, int heads, int ws, int shift) {
  if (batch <= 0 || C % heads || Hp % ws || Wp % ws || shift < 0 || shift >= ws) return SCOT_ERR_SHAPE;
  a.C = C; a.heads = heads; a.Hp = Hp; a.Wp = Wp; a.ws = ws; a.shift = shift;
  a.nwx = Wp / ws; a.nw_per_img = (Hp / ws) * (Wp / ws);
  a.use_tr = g_scot_use_tr;
  return SCOT_OK;
}

// compute: SCOT_BF16 → qkv/out are bf16;  SCOT_F32 → f32.   Grid is the (already padded) Hp x Wp token grid.
extern "C" int scot_window_attn_fwd(int compute, const void* qkv, void* out, float* lse, const float* bias_table,
                                    const float* logit_scale, int batch, int Hp, int Wp, int C, int heads, int ws,
                                    int shift, hipStream_t stream) {
  AttnArgs a{};
  int rc = fill_args(a, batch, Hp, Wp, C, heads, ws, shift);
  if (rc) return rc;
  a.qkv = qkv; a.out = out; a.lse = lse; a.bias_table = bias_table; a.logit_scale = logit_scale;
  const int nwin = batch * a.nw_per_img;
  rc = scot_attn_w16(a, compute, C / heads, nwin, false, stream);
  if (rc != SCOT_ERR_UNSUPPORTED) return rc;
  return compute == SCOT_BF16 ? dispatch_hd<bf16_t>(a, C / heads, nwin, false, stream)
                              : dispatch_hd<float>(a, C / heads, nwin, false, stream);
}

// ------------------------------------------------------------------ attention probabilities (`output_attentions=True`, HF:443-455)
// The fused kernels never materialise the softmax; a caller that asks for it gets it from this separate, simple kernel: workgroup
// = (window, head), thread = query; the window's normalised keys sit in LDS as fp32 (every thread reads the same key row: a
// broadcast), P[q][k] = exp(cos(q, k)·scale + bias(q - k) + mask(q, k) - lse[q]) with the forward's log-sum-exp.  Off the hot path.
__global__ __launch_bounds__(256) void attn_probs_kernel(AttnArgs p, int qkv_dt, float* __restrict__ probs, int HD) {
  extern __shared__ __attribute__((aligned(16))) float kn[];      // [N][HD]
  const int win = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
  const int N = p.ws * p.ws, ld = 3 * p.C, TW = 2 * p.ws - 1;
  for (int i = tid; i < N; i += blockDim.x) {
    const size_t base = (size_t)win_token(p, win, i) * ld + p.C + h * HD;
    float ss = 0.f;
    for (int d = 0; d < HD; ++d) { const float v = ld1(p.qkv, qkv_dt, base + d); kn[i * HD + d] = v; ss += v * v; }
    const float r = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    for (int d = 0; d < HD; ++d) kn[i * HD + d] *= r;
  }
  __syncthreads();
  const float scale = __expf(fminf(p.logit_scale[h], 4.605170185988092f));
  for (int q = tid; q < N; q += blockDim.x) {
    float qv[64];
    const size_t base = (size_t)win_token(p, win, q) * ld + h * HD;
    float ss = 0.f;
    for (int d = 0; d < HD; ++d) { qv[d] = ld1(p.qkv, qkv_dt, base + d); ss += qv[d] * qv[d]; }
    const float r = scale / fmaxf(sqrtf(ss), 1e-12f);
    const float lse = p.lse[((size_t)win * p.heads + h) * N + q];
    const int rq = win_region(p, win, q), qy = q / p.ws, qx = q % p.ws;
    float* out = probs + (((size_t)win * p.heads + h) * N + q) * N;
    for (int k = 0; k < N; ++k) {
      float dot = 0.f;
      for (int d = 0; d < HD; ++d) dot = fmaf(qv[d], kn[k * HD + d], dot);
      float sv = dot * r + p.bias_table[h * TW * TW + (qy - k / p.ws + p.ws - 1) * TW + (qx - k % p.ws + p.ws - 1)];
      if (win_region(p, win, k) != rq) sv -= 200.0f;               // the -100 mask, added twice (HF:433-436)
      out[k] = __expf(sv - lse);
    }
  }
}
extern "C" int scot_window_attn_probs(const void* qkv, int qkv_dt, const float* lse, const float* bias_table, const float* logit_scale,
                                      float* probs, int batch, int Hp, int Wp, int C, int heads, int ws, int shift, hipStream_t stream) {
  AttnArgs a{};
  const int rc = fill_args(a, batch, Hp, Wp, C, heads, ws, shift);
  if (rc) return rc;
  const int HD = C / heads, N = ws * ws;
  if (HD > 64 || (qkv_dt & ~1)) return SCOT_ERR_UNSUPPORTED;
  const size_t sh = (size_t)N * HD * sizeof(float);
  if (sh > 64 * 1024) return SCOT_ERR_UNSUPPORTED;
  a.qkv = qkv; a.lse = (float*)lse; a.bias_table = bias_table; a.logit_scale = logit_scale;
  hipLaunchKernelGGL(attn_probs_kernel, dim3(batch * a.nw_per_img, heads), dim3(256), sh, stream, a, qkv_dt, probs, HD);
  return scot_check_launch();
}

static int attn_bwd_impl(int compute, const void* qkv, const void* out_fwd, const void* dout, const float* lse,
                         const float* bias_table, const float* logit_scale, void* dqkv,
                         float* dbias_table, float* dlogit_scale, int batch, int Hp, int Wp, int C,
                         int heads, int ws, int shift, int nrep, size_t stride_tab, size_t stride_ls, hipStream_t stream);
extern "C" int scot_window_attn_bwd(int compute, const void* qkv, const void* out_fwd, const void* dout, const float* lse,
                                    const float* bias_table, const float* logit_scale, void* dqkv,
                                    float* dbias_table, float* dlogit_scale, int batch, int Hp, int Wp, int C,
                                    int heads, int ws, int shift, hipStream_t stream) {
  return attn_bwd_impl(compute, qkv, out_fwd, dout, lse, bias_table, logit_scale, dqkv, dbias_table, dlogit_scale, batch, Hp, Wp, C, heads, ws,
                       shift, 1, 0, 0, stream);
}
// include/scot_hip.h: scot_window_attn_bwd_rep — the two accumulated buffers as nrep replicas (window w adds into replica w % nrep)
extern "C" int scot_window_attn_bwd_rep(int compute, const void* qkv, const void* out_fwd, const void* dout, const float* lse,
                                        const float* bias_table, const float* logit_scale, void* dqkv,
                                        float* dbias_table, float* dlogit_scale, int batch, int Hp, int Wp, int C,
                                        int heads, int ws, int shift, int nrep, size_t rep_stride_tab, size_t rep_stride_ls,
                                        hipStream_t stream) {
  if (nrep < 1) return SCOT_ERR_SHAPE;
  return attn_bwd_impl(compute, qkv, out_fwd, dout, lse, bias_table, logit_scale, dqkv, dbias_table, dlogit_scale, batch, Hp, Wp, C, heads, ws,
                       shift, nrep, rep_stride_tab, rep_stride_ls, stream);
}
// dst[d.dst_off + j] += Σ_{r0 <= r < nrep} rep[r·stride + d.src_off + j], j < d.count, for every entry d of desc (int32 [n][3], device)
__global__ __launch_bounds__(256) void replica_reduce_kernel(const float* __restrict__ rep, int r0, int nrep, size_t stride,
                                                             const int* __restrict__ desc, float* __restrict__ dst) {
  const int* d = desc + 3 * blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= d[2]) return;
  float acc = 0.f;
  for (int r = r0; r < nrep; ++r) acc += rep[(size_t)r * stride + d[0] + j];
  dst[(size_t)d[1] + j] += acc;
}
extern "C" int scot_replica_reduce(const float* rep, int r0, int nrep, size_t stride, const int* desc, int n, int max_count, float* dst,
                                   hipStream_t stream) {
  if (!rep || !desc || !dst || n <= 0 || max_count <= 0 || r0 < 0 || nrep < r0) return SCOT_ERR_SHAPE;
  hipLaunchKernelGGL(replica_reduce_kernel, dim3((max_count + 255) / 256, n), dim3(256), 0, stream, rep, r0, nrep, stride, desc, dst);
  return scot_check_launch();
}
static int attn_bwd_impl(int compute, const void* qkv, const void* out_fwd, const void* dout, const float* lse,
                         const float* bias_table, const float* logit_scale, void* dqkv,
                         float* dbias_table, float* dlogit_scale, int batch, int Hp, int Wp, int C,
                         int heads, int ws, int shift, int nrep, size_t stride_tab, size_t stride_ls, hipStream_t stream) {
  AttnArgs a{};
  int rc = fill_args(a, batch, Hp, Wp, C, heads, ws, shift);
  if (rc) return rc;
  a.qkv = qkv; a.out = dqkv; a.dout = dout; a.ofwd = out_fwd; a.lse = (float*)lse; a.bias_table = bias_table; a.logit_scale = logit_scale;
  a.dbias_table = dbias_table; a.dlogit_scale = dlogit_scale; a.nrep = nrep; a.rep_stride_tab = stride_tab; a.rep_stride_ls = stride_ls;
  const int nwin = batch * a.nw_per_img;
  rc = scot_attn_w16(a, compute, C / heads, nwin, true, stream);
  if (rc != SCOT_ERR_UNSUPPORTED) return rc;
  return compute == SCOT_BF16 ? dispatch_hd<bf16_t>(a, C / heads, nwin, true, stream)
                              : dispatch_hd<float>(a, C / heads, nwin, true, stream);
}
