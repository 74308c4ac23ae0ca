// Window attention, 16x16-window fast path (ws = 16, N = 256 keys, shift 0 or 8): stages 0 and 1 of every preset at
// 128x128 input, i.e. ~90 % of all attention time.  Same math as attention.hip (reference HF:389-455 through
// scOT/model.py:522-559); what the fixed geometry buys:
//   * a 16-query block is ONE window row and a 16-key tile is ONE window row, so the relative-position bias of a lane's
//     four logits is four CONSECUTIVE table entries at  lane_base + uniform(qy - ky): plain ds_read with immediate
//     offsets, no per-element index arithmetic and no dependent position lookups;
//   * with shift = ws/2 the shift mask of (query, key) is  [row_region(qy) != row_region(ky)]  (uniform per tile)  OR
//     [col_region(qx) != col_region(kx)]  (a per-lane constant): one select per tile instead of two per logit;
//   * logits are kept in the log2 domain (table, scale and lse pre-multiplied by log2 e): v_exp_f32 directly;
//   * outputs are produced TRANSPOSED (O^T = V^T P^T, dQ^T, dK^T, dV^T): a lane then owns 4 consecutive features of one
//     token -> 8/16-byte stores and two-step (xor 16, 32) row reductions in the normalisation backward.
#include <cstdio>
#include <type_traits>
#include "attention.h"

static constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
static constexpr float kMask2 = -200.0f * 1.4426950408889634f;   // the -100 mask, added twice (HF:433-436), log2 domain

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <typename CT> __device__ __forceinline__ void ld4(const void* p, size_t i, float (&v)[4]);
template <> __device__ __forceinline__ void ld4<float>(const void* p, size_t i, float (&v)[4]) {
  const float4 u = *(const float4*)((const float*)p + i);
  v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const void* p, size_t i, float (&v)[4]) {
  const uint2 u = *(const uint2*)((const bf16_t*)p + i);
  unpack_bf16x2(u.x, v[0], v[1]); unpack_bf16x2(u.y, v[2], v[3]);
}
template <typename CT> __device__ __forceinline__ void st4(void* p, size_t i, const float (&v)[4]);
template <> __device__ __forceinline__ void st4<float>(void* p, size_t i, const float (&v)[4]) {
  *(float4*)((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(void* p, size_t i, const float (&v)[4]) {
  *(uint2*)((bf16_t*)p + i) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}

// geometry of the fast path
struct W16 {
  static constexpr int NT = 16, NP = 256, TW = 31, TS = 961, TSP = 964;
};

// LDS tiles of the fast path.  bf16 with the head_dim padded to 32 (Poseidon-T/S/B): PAD-FREE rows of 64 bytes whose four
// 16-byte chunks sit at positions chunk ^ ((row >> 2) & 3) — the 16 rows of a K-contiguous fragment read and the 4x4 blocks
// of a transposing read then spread over all banks without the 8-element row padding: 16 KB per tile instead of 20 KB, which
// is the difference between 3 and 4 workgroups per CU.  Otherwise (fp32 compute, head_dim 64): padded rows as in attention.hip.
// All reads go through per-lane base pointers + compile-time offsets (`ds_read ... offset:imm`).
template <typename CT, int HD> struct Tile16 {
  static constexpr int KD = ((HD + 31) / 32) * 32, DT = HD / 16;
  static constexpr bool swz = KD == 32 && sizeof(CT) == 2;
  static constexpr int pitch = swz ? 32 : row_pitch<HD, CT>();
  static constexpr int elems = W16::NP * pitch;
  const CT* kc;        // K-contiguous reads: row (lane&15), features (lane>>4)*8 ..
  const CT* ks[DT];    // transposing reads of feature block d (bf16) / strided element reads (fp32)
  __device__ __forceinline__ Tile16(const CT* T, int lane) {
    const int lc = lane & 15, g = lane >> 4;
    kc = T + lc * pitch + (swz ? (g ^ ((lc >> 2) & 3)) : g) * 8;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      if (sizeof(CT) == 2) {
        // lane i of a 16-lane group hands the instruction the 8-byte piece (i & 3) of row 4g + (i >> 2), columns d*16 ..
        const int chunk = d * 2 + ((lc & 3) >> 1);
        ks[d] = T + (g * 4 + (lc >> 2)) * pitch + (swz ? (chunk ^ g) : chunk) * 8 + (lc & 1) * 4;
      } else {
        ks[d] = T + g * 4 * pitch + d * 16 + lc;
      }
    }
  }
  // where chunk `ch` (8 elements) of row n is stored
  static __device__ __forceinline__ int store_off(int n, int ch) { return n * pitch + (swz ? (ch ^ ((n >> 2) & 3)) : ch) * 8; }
};
template <int HD> __device__ __forceinline__ Frag<bf16_t> rd_kc(const Tile16<bf16_t, HD>& T, int t, int kk) {
  Frag<bf16_t> f;
  f.v = *(const s16x8_t*)(T.kc + t * 16 * Tile16<bf16_t, HD>::pitch + kk * 32);
  return f;
}
template <int HD> __device__ __forceinline__ Frag<float> rd_kc(const Tile16<float, HD>& T, int t, int kk) {
  Frag<float> f;
  const float* q = T.kc + t * 16 * Tile16<float, HD>::pitch + kk * 32;
  const float4 a = *(const float4*)q, b = *(const float4*)(q + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
  return f;
}
// 8 keys of tile pair tp (4 of tile 2tp at rows 4g.., 4 of tile 2tp+1), feature column d*16 + (lane&15)
template <int HD> __device__ __forceinline__ Frag<bf16_t> rd_ks(const Tile16<bf16_t, HD>& T, int tp, int d) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_v4;
  constexpr int pitch = Tile16<bf16_t, HD>::pitch;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(T.ks[d] + (2 * tp) * 16 * pitch));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(T.ks[d] + (2 * tp + 1) * 16 * pitch));
  Frag<bf16_t> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
template <int HD> __device__ __forceinline__ Frag<float> rd_ks(const Tile16<float, HD>& T, int tp, int d) {
  constexpr int pitch = Tile16<float, HD>::pitch;
  Frag<float> f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.v[j] = T.ks[d][((2 * tp) * 16 + j) * pitch];
    f.v[j + 4] = T.ks[d][((2 * tp + 1) * 16 + j) * pitch];
  }
  return f;
}


// ---- bf16x3 (X3): q/k/v/dO/O and the results are fp32 in memory, every MFMA operand is a (hi, lo) pair of bf16 fragments —
// hi = bf16(x), lo = bf16(x - hi) — and every product is hi·hi + hi·lo + lo·hi on the bf16 MFMA (operand error ~2^-17, the dropped
// lo·lo term 2^-18): the accuracy class of the fp32 kernels at three bf16 MFMAs instead of eight fp32 ones per 32-deep product.
// LDS tiles come in pairs (hi tile, lo tile `elems` further).  With X3 = false everything below degenerates to the single-fragment code.
template <typename CT, bool X3> struct FragX { Frag<CT> hi, lo; };
template <typename CT, bool X3> __device__ __forceinline__ FragX<CT, X3> fragx_from_f32(const float (&x)[8]) {
  FragX<CT, X3> f;
  if constexpr (X3) {
    float h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = bf2f(f2bf(x[j])); l[j] = x[j] - h[j]; }
    f.hi = frag_from_f32<CT>(h);
    f.lo = frag_from_f32<CT>(l);
  } else {
    f.hi = frag_from_f32<CT>(x);
  }
  return f;
}
template <typename CT, bool X3> __device__ __forceinline__ void mmax(f32x4_t& acc, const FragX<CT, X3>& a, const FragX<CT, X3>& b) {
  mma16(acc, a.hi, b.hi);
  if constexpr (X3) { mma16(acc, a.hi, b.lo); mma16(acc, a.lo, b.hi); }
}
template <typename CT, int HD, bool X3> struct TileX {
  Tile16<CT, HD> hi, lo;
  __device__ __forceinline__ TileX(const CT* T, int lane) : hi(T, lane), lo(T + (X3 ? Tile16<CT, HD>::elems : 0), lane) {}
};
template <int HD, typename CT, bool X3> __device__ __forceinline__ FragX<CT, X3> rdx_kc(const TileX<CT, HD, X3>& T, int t, int kk) {
  FragX<CT, X3> f;
  f.hi = rd_kc<HD>(T.hi, t, kk);
  if constexpr (X3) f.lo = rd_kc<HD>(T.lo, t, kk);
  return f;
}
template <int HD, typename CT, bool X3> __device__ __forceinline__ FragX<CT, X3> rdx_ks(const TileX<CT, HD, X3>& T, int tp, int d) {
  FragX<CT, X3> f;
  f.hi = rd_ks<HD>(T.hi, tp, d);
  if constexpr (X3) f.lo = rd_ks<HD>(T.lo, tp, d);
  return f;
}
// 8 floats -> LDS at `off` of the hi tile (and their bf16 remainder into the lo tile)
template <typename CT, int HD, bool X3> __device__ __forceinline__ void store8_x(CT* tile, int off, const float (&v)[8]) {
  if constexpr (X3) {
    float h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = bf2f(f2bf(v[j])); l[j] = v[j] - h[j]; }
    store8_ct(tile + off, h);
    store8_ct(tile + Tile16<CT, HD>::elems + off, l);
  } else {
    store8_ct(tile + off, v);
  }
}

// token of window position n = (y, x) — the roll(-shift) + window_partition index math (reference model.py:522-559), in
// registers: no LDS token table (1 KB that decided between 2 and 3 workgroups per CU for the dQ kernel)
struct W16Tok {
  int base, y0, x0, Hp, Wp;
  __device__ __forceinline__ W16Tok(const AttnArgs& p, int win) {
    const int b = win / p.nw_per_img, w = win % p.nw_per_img;
    base = b * p.Hp * p.Wp; y0 = (w / p.nwx) * 16 + p.shift; x0 = (w % p.nwx) * 16 + p.shift; Hp = p.Hp; Wp = p.Wp;
  }
  __device__ __forceinline__ int operator()(int n) const {
    int y = y0 + (n >> 4), x = x0 + (n & 15);
    if (y >= Hp) y -= Hp;
    if (x >= Wp) x -= Wp;
    return base + y * Wp + x;
  }
};

// stage the window's rows of q/k/v/dO (column offset `col`) into an LDS tile, optionally L2-normalised (F.normalize, eps 1e-12)
template <typename CT, typename MT, int HD, bool NORM, bool X3>
__device__ __forceinline__ void w16_stage(CT* tile, const void* src, int ld, int col, const W16Tok& T, int tid) {
  constexpr int CPR = ((HD + 31) / 32) * 4;
#pragma unroll 2
  for (int c = tid; c < W16::NP * CPR; c += blockDim.x) {
    const int n = c / CPR, ch = c % CPR;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (ch * 8 < HD) ld8(src, ct_traits<MT>::dtype, (size_t)T(n) * ld + col + ch * 8, v);
    if (NORM) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
#pragma unroll
      for (int o = 1; o < CPR; o <<= 1) ss += __shfl_xor(ss, o, 64);
      const float r = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= r;
    }
    store8_x<CT, HD, X3>(tile, Tile16<CT, HD>::store_off(n, ch), v);
  }
}
// accT[d][r]: gradient (times `mul`) wrt the NORMALISED row of token `tokn`, feature d*16 + (lane>>4)*4 + r.
// y = x / max(|x|, eps):  dx = (g - y (y·g)) / |x|   (|x| >= eps),   dx = g / eps otherwise  (F.normalize, eps 1e-12).
template <typename CT, int HD>
__device__ __forceinline__ void normalize_bwd_load(float (&x)[HD / 16][4], const void* src, size_t off, int lane) {
#pragma unroll
  for (int d = 0; d < HD / 16; ++d) ld4<CT>(src, off + d * 16 + (lane >> 4) * 4, x[d]);
}
template <typename CT, int HD>
__device__ __forceinline__ void normalize_bwd_store_t(const f32x4_t (&acc)[HD / 16], float mul, const float (&x)[HD / 16][4], void* dst,
                                                      size_t off, int lane) {
  constexpr int DT = HD / 16;
  const int g = lane >> 4;
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int d = 0; d < DT; ++d) {
#pragma unroll
    for (int r = 0; r < 4; ++r) ss += x[d][r] * x[d][r];
  }
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  const float nrm = sqrtf(ss);
  const float rn = (nrm < 1e-12f && sizeof(CT) == 2) ? 0.f : 1.0f / fmaxf(nrm, 1e-12f);      // (clamped row, 16-bit result: see attention.hip)
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 4; ++r) dot += (x[d][r] * rn) * (acc[d][r] * mul);
  dot += __shfl_xor(dot, 16, 64);
  dot += __shfl_xor(dot, 32, 64);
  if (nrm < 1e-12f) dot = 0.f;
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = rn * (acc[d][r] * mul - (x[d][r] * rn) * dot);
    st4<CT>(dst, off + d * 16 + g * 4, o);
  }
}

// B-operand fragments of ONE row per lane column read straight from HBM: lane (lc, g) takes features kk*32 + g*8 .. +7 of
// the row at element offset `rowoff` (optionally L2-normalised over the whole row: two xor-shuffles)
template <typename CT, int HD>
__device__ __forceinline__ void row_f32(float (&v)[(HD + 31) / 32][8], const void* src, size_t rowoff, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < (HD + 31) / 32; ++kk) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[kk][j] = 0.f;
    if (kk * 32 + g * 8 < HD) ld8(src, ct_traits<CT>::dtype, rowoff + kk * 32 + g * 8, v[kk]);
  }
}
template <typename CT, typename MT, int HD, bool X3>
__device__ __forceinline__ void row_frag(FragX<CT, X3> (&f)[(HD + 31) / 32], const void* src, size_t rowoff, bool normalize, int lane) {
  constexpr int KS = (HD + 31) / 32;
  float v[KS][8];
  row_f32<MT, HD>(v, src, rowoff, lane);
  float r = 1.f;
  if (normalize) {
    float ss = 0.f;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[kk][j] * v[kk][j];
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    r = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  }
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[kk][j] *= r;
    f[kk] = fragx_from_f32<CT, X3>(v[kk]);
  }
}

// ================================================================================================= forward
// Which (window, head, half) a workgroup of the 1-D grid works on.  The heads of a window read interleaved 64-byte segments of the same
// qkv / O / dO rows (two heads per 128-byte line) and the backward's two halves read the same rows again, so the `members` =
// heads x halves workgroups of ONE window are placed back to back ON ONE XCD (workgroup L runs on XCD L % 8: observed dispatch order,
// used for speed only): its L2 then serves the shared lines and the second half's re-read.  With (window, head, half) as grid
// (x, y, z) those workgroups were nwin apart in dispatch order and every line came from HBM two to four times (PMC, round 3: 152 MB
// fetched per stage-0 backward launch for 63 MB of inputs).
__device__ __forceinline__ void w16_block(const AttnArgs& p, int halves, int& win, int& h, int& half) {
  const int L = blockIdx.x, members = p.heads * halves, nwin = (int)gridDim.x / members;
  int m;
  if ((nwin & 7) == 0) { const int j = L >> 3; win = (j / members) * 8 + (L & 7); m = j % members; }
  else { win = L / members; m = L % members; }
  h = m % p.heads; half = m / p.heads;
}

template <typename CT, int HD, bool SHIFTED, bool X3>
__global__ __launch_bounds__(256, 2) void attn16_fwd_kernel(AttnArgs p) {
  using MT = typename std::conditional<X3, float, CT>::type;   // element type of q/k/v/out in memory
  constexpr int NT = W16::NT, NP = W16::NP, KS = (HD + 31) / 32, DT = HD / 16, TE = (X3 ? 2 : 1) * Tile16<CT, HD>::elems;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CT* Kn = (CT*)smem;
  CT* Vs = Kn + TE;
  float* tab2 = (float*)(Vs + TE);

  int win, h, half_;
  w16_block(p, 1, win, h, half_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ld = 3 * p.C, g = lane >> 4, lc = lane & 15;
  const W16Tok tokf(p, win);

  for (int i = tid; i < W16::TS; i += 256) tab2[i] = p.bias_table[h * W16::TS + i] * kLog2e;
  w16_stage<CT, MT, HD, true, X3>(Kn, p.qkv, ld, p.C + h * HD, tokf, tid);
  w16_stage<CT, MT, HD, false, X3>(Vs, p.qkv, ld, 2 * p.C + h * HD, tokf, tid);
  __syncthreads();

  const float scale2 = __expf(fminf(p.logit_scale[h], 4.605170185988092f)) * kLog2e;  // exp(min(ls, ln 100)), HF:416
  const int w = win % p.nw_per_img;
  const bool lastrow = SHIFTED && (w / p.nwx) == p.Hp / 16 - 1, lastcol = SHIFTED && (w % p.nwx) == p.nwx - 1;
  const float mlane = (lastcol && ((lc >= 8) != (g >= 2))) ? kMask2 : 0.f;
  // bias of (query (qy, qx = lc), key (ky = t, kx = 4g + r)) = tab[(qy - t + 15)*31 + lc - 4g - r + 15]
  const float* tabl = tab2 + (lc - 4 * g + 12);
  const TileX<CT, HD, X3> kt(Kn, lane), vt(Vs, lane);

#pragma nounroll
  for (int qb = wave; qb < 16; qb += 4) {
    FragX<CT, X3> qf[KS];
    const int q = qb * 16 + lc;
    const int tokq = tokf(q);
    row_frag<CT, MT, HD, X3>(qf, p.qkv, (size_t)tokq * ld + h * HD, true, lane);
    const float* tq = tabl + qb * 31;

    f32x4_t s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) mmax(s[t], rdx_kc<HD>(kt, t, kk), qf[kk]);
      if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting every tile's LDS reads
    }
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float madd = 0.f;
      if (SHIFTED) madd = (lastrow && ((qb >= 8) != (t >= 8))) ? kMask2 : mlane;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = fmaf(s[t][r], scale2, tq[(15 - t) * 31 + 3 - r]) + madd;
        s[t][r] = v;
        m = fmaxf(m, v);
      }
      if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = fast_exp2(s[t][r] - m);
        s[t][r] = e;
        l += e;
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (g == 0 && p.lse) p.lse[((size_t)win * p.heads + h) * NP + q] = (m + __log2f(l)) * kLn2;

    // O^T = V^T · P^T : A = V^T (transposing fragment read), B = P^T (the lane's 8 keys of query lc: already in order)
    f32x4_t o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < NT / 2; ++tp) {
      float pv[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { pv[r] = s[2 * tp][r]; pv[r + 4] = s[2 * tp + 1][r]; }
      const FragX<CT, X3> pf = fragx_from_f32<CT, X3>(pv);
#pragma unroll
      for (int d = 0; d < DT; ++d)
        mmax(o[d], rdx_ks<HD>(vt, tp, d), pf);
      if ((tp & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    // o[d][r]: feature d*16 + 4g + r of query q
    const size_t base = (size_t)tokq * p.C + h * HD + g * 4;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const float ov[4] = {o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv};
      st4<MT>(p.out, base + d * 16, ov);
    }
  }
}

// ================================================================================================= backward: dQ, d table, d logit_scale
template <typename CT, int HD, bool SHIFTED, bool X3>
__device__ __forceinline__ void attn16_bwd_dq_body(const AttnArgs& p) {
  using MT = typename std::conditional<X3, float, CT>::type;
  constexpr int NT = W16::NT, NP = W16::NP, KS = (HD + 31) / 32, DT = HD / 16, TE = (X3 ? 2 : 1) * Tile16<CT, HD>::elems;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CT* X = (CT*)smem;            // Kn
  CT* Y = X + TE;               // V
  float* tab2 = (float*)(Y + TE);
  double* dtab = (double*)(tab2 + W16::TSP);   // ds_add_f64 is full rate on gfx950, ds_add_f32 is not (see attention.hip)
  float* red = (float*)(dtab + W16::TSP);      // [waves <= 8]

  int win, h, half_;
  w16_block(p, 2, win, h, half_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ld = 3 * p.C, g = lane >> 4, lc = lane & 15;
  const W16Tok tokf(p, win);

  for (int i = tid; i < W16::TS; i += blockDim.x) { tab2[i] = p.bias_table[h * W16::TS + i] * kLog2e; dtab[i] = 0.0; }
  w16_stage<CT, MT, HD, true, X3>(X, p.qkv, ld, p.C + h * HD, tokf, tid);
  w16_stage<CT, MT, HD, false, X3>(Y, p.qkv, ld, 2 * p.C + h * HD, tokf, tid);
  __syncthreads();

  const float scale = __expf(fminf(p.logit_scale[h], 4.605170185988092f));
  const float scale2 = scale * kLog2e;
  const int w = win % p.nw_per_img;
  const bool lastrow = SHIFTED && (w / p.nwx) == p.Hp / 16 - 1, lastcol = SHIFTED && (w % p.nwx) == p.nwx - 1;
  const float mlane = (lastcol && ((lc >= 8) != (g >= 2))) ? kMask2 : 0.f;
  const float* tabl = tab2 + (lc - 4 * g + 12);
  double* dtabl = dtab + (lc - 4 * g + 15);     // entry of (q = lc, key 4g): the lane's anti-diagonal sum lands here
  const TileX<CT, HD, X3> xb(X, lane), yb(Y, lane);
  // d logit_scale = Σ_qk dS·cos·scale with Σ_k dS = 0 per query: a heavily cancelling sum.  dS uses delta from the
  // stored (rounded) forward output; the row sums D = Σ_k P·dP and B = Σ_k P·cos taken here in fp32 put the exact
  // cancellation back:  Σ_k P (dP - D) cos = Σ_k dS·cos + (delta - D)·B.
  float dls = 0.f;

#pragma nounroll
  for (int qb = wave; qb < 16; qb += (int)(blockDim.x >> 6)) {
    const int q = qb * 16 + lc;
    const int tokq = tokf(q);
    float accD = 0.f, accB = 0.f;
    FragX<CT, X3> qf[KS], gf[KS];
    row_frag<CT, MT, HD, X3>(qf, p.qkv, (size_t)tokq * ld + h * HD, true, lane);
    float delta = 0.f;
    {
      float dov[KS][8], ov[KS][8];
      row_f32<MT, HD>(dov, p.dout, (size_t)tokq * p.C + h * HD, lane);
      row_f32<MT, HD>(ov, p.ofwd, (size_t)tokq * p.C + h * HD, lane);
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += dov[kk][j] * ov[kk][j];
        gf[kk] = fragx_from_f32<CT, X3>(dov[kk]);
      }
      delta += __shfl_xor(delta, 16, 64);
      delta += __shfl_xor(delta, 32, 64);
    }
    const float nlse2 = -p.lse[((size_t)win * p.heads + h) * NP + q] * kLog2e;
    const size_t off = (size_t)tokq * ld + h * HD;
    float xq[DT][4];   // the un-normalised q row again, in the layout of the transposed result: loaded here, used by the epilogue
    normalize_bwd_load<MT, HD>(xq, p.qkv, off, lane);
    const float* tq = tabl + qb * 31;
    double* dq_tab = dtabl + qb * 31;

    f32x4_t dq[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) dq[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // (as in the dK/dV half below: the LDS operands of tile pair tp + 1 are requested before pair tp is multiplied)
    constexpr bool PRE = HD <= 32 && !X3;
    struct PairOps { FragX<CT, X3> k[2][KS], v[2][KS], kt[DT]; float tb[2][4]; };
    PairOps po[PRE ? 2 : 1];
    auto fetch = [&](PairOps& o, int tp) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * tp + half;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) { o.k[half][kk] = rdx_kc<HD>(xb, t, kk); o.v[half][kk] = rdx_kc<HD>(yb, t, kk); }
#pragma unroll
        for (int r = 0; r < 4; ++r) o.tb[half][r] = tq[(15 - t) * 31 + 3 - r];
      }
#pragma unroll
      for (int d = 0; d < DT; ++d) o.kt[d] = rdx_ks<HD>(xb, tp, d);
    };
    if constexpr (PRE) fetch(po[0], 0);
#pragma unroll
    for (int tp = 0; tp < NT / 2; ++tp) {
      const PairOps& o = po[PRE ? (tp & 1) : 0];
      if constexpr (PRE) { if (tp + 1 < NT / 2) fetch(po[(tp + 1) & 1], tp + 1); }
      float ds8[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * tp + half;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          if constexpr (PRE) { mmax(s, o.k[half][kk], qf[kk]); mmax(dp, o.v[half][kk], gf[kk]); }
          else { mmax(s, rdx_kc<HD>(xb, t, kk), qf[kk]); mmax(dp, rdx_kc<HD>(yb, t, kk), gf[kk]); }
        }
        float madd = nlse2;
        if (SHIFTED) madd += (lastrow && ((qb >= 8) != (t >= 8))) ? kMask2 : mlane;
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = fast_exp2(fmaf(s[r], scale2, PRE ? o.tb[half][r] : tq[(15 - t) * 31 + 3 - r]) + madd);
          ds[r] = pr * (dp[r] - delta);
          ds8[half * 4 + r] = ds[r];
          accD = fmaf(pr, dp[r], accD);
          accB = fmaf(pr, s[r], accB);
          dls = fmaf(ds[r], s[r], dls);
        }
        // pin the three running sums here: left alone, the scheduler sinks all 192 of a query block's multiply-adds to the
        // block's end and keeps their operands alive (256 VGPRs + scratch instead of ~90)
        asm volatile("" : "+v"(accD), "+v"(accB), "+v"(dls));
        // the 16x16 block of dS feeds the 31 entries (dx = qx - kx) of table row (qy - ky): DPP row shifts fold the
        // lane's four keys along the anti-diagonal, then one LDS atomic per lane (+ 3 lanes for the wrapped tail)
        const float a = ds[0] + dpp_row<0x101>(ds[1]) + dpp_row<0x102>(ds[2]) + dpp_row<0x103>(ds[3]);
        const float bt = dpp_row<0x11F>(ds[1]) + dpp_row<0x11E>(ds[2]) + dpp_row<0x11D>(ds[3]);
        atomicAdd(&dq_tab[(15 - t) * 31], (double)a);
        if (lc >= 13) atomicAdd(&dq_tab[(15 - t) * 31 - 16], (double)bt);
      }
      const FragX<CT, X3> df = fragx_from_f32<CT, X3>(ds8);
#pragma unroll
      for (int d = 0; d < DT; ++d) {  // dQn^T += Kn^T · dS^T
        if constexpr (PRE) mmax(dq[d], o.kt[d], df);
        else mmax(dq[d], rdx_ks<HD>(xb, tp, d), df);
      }
      __builtin_amdgcn_sched_barrier(0);   // one tile pair at a time: unrolled for the immediates, not for hoisting
    }
    accD += __shfl_xor(accD, 16, 64); accD += __shfl_xor(accD, 32, 64);
    accB += __shfl_xor(accB, 16, 64); accB += __shfl_xor(accB, 32, 64);
    if (g == 0) dls = fmaf(delta - accD, accB, dls);
    normalize_bwd_store_t<MT, HD>(dq, scale, xq, p.out, off, lane);
  }
  // d/dls [cos * exp(ls)] = cos * scale  (0 when clamped at ln 100, HF:416)
  dls = wave_sum(dls);
  if (lane == 0) red[wave] = dls;
  __syncthreads();
  // (every (window, head) workgroup of a head flushes the same 961 addresses: start each one somewhere else so that concurrent
  // workgroups do not queue up on the same L2 atomic unit in lockstep)
  const int rep = p.nrep > 1 ? win % p.nrep : 0;
  float* dbt = p.dbias_table + (size_t)rep * p.rep_stride_tab;
  for (int i = tid; i < W16::TS; i += blockDim.x) {
    int k = i + (win % 31) * 31;
    k = k >= W16::TS ? k - W16::TS : k;
    atomicAdd(&dbt[h * W16::TS + k], (float)dtab[k]);
  }
  if (tid == 0 && p.logit_scale[h] <= 4.605170185988092f) {
    float r = 0.f;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) r += red[wv];
    atomicAdd(&p.dlogit_scale[(size_t)rep * p.rep_stride_ls + h], r * scale);
  }
}

// ================================================================================================= backward: dK, dV
template <typename CT, int HD, bool SHIFTED, bool X3>
__device__ __forceinline__ void attn16_bwd_dkv_body(const AttnArgs& p) {
  using MT = typename std::conditional<X3, float, CT>::type;
  constexpr int NT = W16::NT, NP = W16::NP, KS = (HD + 31) / 32, DT = HD / 16, TE = (X3 ? 2 : 1) * Tile16<CT, HD>::elems;
  constexpr int CPR = KS * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CT* X = (CT*)smem;            // Qn
  CT* Y = X + TE;               // dO
  float* tab2 = (float*)(Y + TE);
  float* nlse2 = tab2 + W16::TSP;   // [NP]  -lse * log2 e
  float* delta = nlse2 + NP;        // [NP]

  int win, h, half_;
  w16_block(p, 2, win, h, half_);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ld = 3 * p.C, g = lane >> 4, lc = lane & 15;

  const W16Tok tokf(p, win);
  for (int i = tid; i < NP; i += blockDim.x) nlse2[i] = -p.lse[((size_t)win * p.heads + h) * NP + i] * kLog2e;
  for (int i = tid; i < W16::TS; i += blockDim.x) tab2[i] = p.bias_table[h * W16::TS + i] * kLog2e;
  w16_stage<CT, MT, HD, true, X3>(X, p.qkv, ld, h * HD, tokf, tid);
  // dO -> LDS and delta[n] = Σ_d dO[n][d]·O[n][d] in the same pass (CPR lanes per row)
#pragma unroll
  for (int c = tid; c < NP * CPR; c += blockDim.x) {
    const int n = c / CPR, d8 = (c % CPR) * 8;
    float v[8], o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = 0.f; o[j] = 0.f; }
    if (d8 < HD) {
      const size_t ro = (size_t)tokf(n) * p.C + h * HD + d8;
      ld8(p.dout, ct_traits<MT>::dtype, ro, v);
      ld8(p.ofwd, ct_traits<MT>::dtype, ro, o);
    }
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dot += v[j] * o[j];
#pragma unroll
    for (int of = 1; of < CPR; of <<= 1) dot += __shfl_xor(dot, of, 64);
    if ((c % CPR) == 0) delta[n] = dot;
    store8_x<CT, HD, X3>(Y, Tile16<CT, HD>::store_off(n, c % CPR), v);
  }
  __syncthreads();

  const float scale = __expf(fminf(p.logit_scale[h], 4.605170185988092f));
  const float scale2 = scale * kLog2e;
  const int w = win % p.nw_per_img;
  const bool lastrow = SHIFTED && (w / p.nwx) == p.Hp / 16 - 1, lastcol = SHIFTED && (w % p.nwx) == p.nwx - 1;
  const float mlane = (lastcol && ((lc >= 8) != (g >= 2))) ? kMask2 : 0.f;
  // bias of (query (qy = t, qx = 4g + r), key (ky, kx = lc)) = tab[(t - ky + 15)*31 + 4g + r - lc + 15]
  const float* tabl = tab2 + (4 * g - lc + 15);
  const TileX<CT, HD, X3> xb(X, lane), yb(Y, lane);
  const float* nlg = nlse2 + g * 4;
  const float* dlg = delta + g * 4;

#pragma nounroll
  for (int kb = wave; kb < 16; kb += (int)(blockDim.x >> 6)) {
    const int tokk = tokf(kb * 16 + lc);
    FragX<CT, X3> kf[KS], vf[KS];
    row_frag<CT, MT, HD, X3>(kf, p.qkv, (size_t)tokk * ld + p.C + h * HD, true, lane);
    row_frag<CT, MT, HD, X3>(vf, p.qkv, (size_t)tokk * ld + 2 * p.C + h * HD, false, lane);
    const float* tk = tabl + (15 - kb) * 31;
    const size_t off = (size_t)tokk * ld + h * HD;
    float xk[DT][4];
    normalize_bwd_load<MT, HD>(xk, (const MT*)p.qkv + p.C, off, lane);

    f32x4_t dv[DT], dk[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) { dv[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dk[d] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

    constexpr bool PRE = HD <= 32 && !X3;      // (head_dim 64 / split operands: the second register set would cost a wave per SIMD or spill)
    if constexpr (PRE) {
    // The LDS operands of tile pair tp + 1 are requested before pair tp is multiplied (two register sets): read where they are used,
    // every tile cost five LDS round trips in a row (83 waits per key block in the ISA) and the wave's chain of waits, not its
    // ~650 VALU instructions per key block, set the kernel's time at three workgroups per CU.
    struct PairOps { FragX<CT, X3> q[2][KS], g[2][KS], yt[DT], xt[DT]; float4 nl[2], qd[2]; float tb[2][4]; };
    PairOps po[2];
    auto fetch = [&](PairOps& o, int tp) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * tp + half;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) { o.q[half][kk] = rdx_kc<HD>(xb, t, kk); o.g[half][kk] = rdx_kc<HD>(yb, t, kk); }
        o.nl[half] = *(const float4*)&nlg[t * 16];
        o.qd[half] = *(const float4*)&dlg[t * 16];
#pragma unroll
        for (int r = 0; r < 4; ++r) o.tb[half][r] = tk[t * 31 + r];
      }
#pragma unroll
      for (int d = 0; d < DT; ++d) { o.yt[d] = rdx_ks<HD>(yb, tp, d); o.xt[d] = rdx_ks<HD>(xb, tp, d); }
    };
    fetch(po[0], 0);
#pragma unroll
    for (int tp = 0; tp < NT / 2; ++tp) {
      const PairOps& o = po[tp & 1];
      if (tp + 1 < NT / 2) fetch(po[(tp + 1) & 1], tp + 1);
      float pf8[8], df8[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * tp + half;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          mmax(s, o.q[half][kk], kf[kk]);   // rows = queries 4g + r of row t, col = key
          mmax(dp, o.g[half][kk], vf[kk]);
        }
        const float nla[4] = {o.nl[half].x, o.nl[half].y, o.nl[half].z, o.nl[half].w};
        const float qda[4] = {o.qd[half].x, o.qd[half].y, o.qd[half].z, o.qd[half].w};
        float madd = 0.f;
        if (SHIFTED) madd = (lastrow && ((t >= 8) != (kb >= 8))) ? kMask2 : mlane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = fast_exp2(fmaf(s[r], scale2, o.tb[half][r]) + (madd + nla[r]));
          pf8[half * 4 + r] = pr;
          df8[half * 4 + r] = pr * (dp[r] - qda[r]);
        }
      }
      const FragX<CT, X3> pf = fragx_from_f32<CT, X3>(pf8), df = fragx_from_f32<CT, X3>(df8);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        mmax(dv[d], o.yt[d], pf);   // dV^T  += dO^T · P
        mmax(dk[d], o.xt[d], df);   // dKn^T += Qn^T · dS
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    } else {
#pragma unroll
    for (int tp = 0; tp < NT / 2; ++tp) {
      float pf8[8], df8[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * tp + half;
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          mmax(s, rdx_kc<HD>(xb, t, kk), kf[kk]);   // rows = queries 4g + r of row t, col = key
          mmax(dp, rdx_kc<HD>(yb, t, kk), vf[kk]);
        }
        const float4 nl = *(const float4*)&nlg[t * 16];
        const float4 qd = *(const float4*)&dlg[t * 16];
        const float nla[4] = {nl.x, nl.y, nl.z, nl.w}, qda[4] = {qd.x, qd.y, qd.z, qd.w};
        float madd = 0.f;
        if (SHIFTED) madd = (lastrow && ((t >= 8) != (kb >= 8))) ? kMask2 : mlane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = fast_exp2(fmaf(s[r], scale2, tk[t * 31 + r]) + (madd + nla[r]));
          pf8[half * 4 + r] = pr;
          df8[half * 4 + r] = pr * (dp[r] - qda[r]);
        }
      }
      const FragX<CT, X3> pf = fragx_from_f32<CT, X3>(pf8), df = fragx_from_f32<CT, X3>(df8);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        mmax(dv[d], rdx_ks<HD>(yb, tp, d), pf);   // dV^T  += dO^T · P
        mmax(dk[d], rdx_ks<HD>(xb, tp, d), df);   // dKn^T += Qn^T · dS
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      const float o[4] = {dv[d][0], dv[d][1], dv[d][2], dv[d][3]};
      st4<MT>(p.out, off + 2 * p.C + d * 16 + g * 4, o);
    }
    normalize_bwd_store_t<MT, HD>(dk, scale, xk, (MT*)p.out + p.C, off, lane);
  }
}

// One launch for the whole backward: blockIdx.z selects the half.  The two halves are independent (each recomputes P from
// q, k, lse and takes delta from dO·O), so their workgroups simply share the grid: no launch boundary between them and
// the dK/dV workgroups fill the CUs as the dQ ones drain.
template <typename CT, int HD, bool SHIFTED, bool X3>
__global__ __launch_bounds__(256, 2) void attn16_bwd_kernel(AttnArgs p) {
  int win, h, half;
  w16_block(p, 2, win, h, half);
  if (half == 0) attn16_bwd_dq_body<CT, HD, SHIFTED, X3>(p);
  else attn16_bwd_dkv_body<CT, HD, SHIFTED, X3>(p);
}
// ================================================================================================= host side
template <typename CT, int HD, bool SHIFTED, bool X3>
static int launch_w16(const AttnArgs& a, int nwin, bool bwd, hipStream_t s) {
  constexpr int NP = W16::NP;
  const size_t tiles = (X3 ? 4 : 2) * Tile16<CT, HD>::elems * sizeof(CT);     // bf16x3: a hi and a lo tile per operand
  const size_t sh_fwd = tiles + W16::TSP * sizeof(float);
  const size_t sh_dq = tiles + W16::TSP * (sizeof(float) + sizeof(double)) + 8 * sizeof(float);
  const size_t sh_dkv = tiles + (W16::TSP + 2 * NP) * sizeof(float);
  dim3 grid(nwin * a.heads), block(256);       // 1-D: w16_block() maps a workgroup to its (window, head[, half])
  if (!bwd) {
    if (sh_fwd > 160 * 1024) return SCOT_ERR_UNSUPPORTED;
    if (sh_fwd > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn16_fwd_kernel<CT, HD, SHIFTED, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_fwd);
    hipLaunchKernelGGL((attn16_fwd_kernel<CT, HD, SHIFTED, X3>), grid, block, sh_fwd, s, a);
  } else {
    const size_t sh_b = sh_dq > sh_dkv ? sh_dq : sh_dkv;
    if (sh_b > 160 * 1024) return SCOT_ERR_UNSUPPORTED;
    if (sh_b > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn16_bwd_kernel<CT, HD, SHIFTED, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_b);
    hipLaunchKernelGGL((attn16_bwd_kernel<CT, HD, SHIFTED, X3>), dim3(nwin * a.heads * 2), block, sh_b, s, a);
  }
  return scot_check_launch();
}

template <typename CT, bool X3> static int dispatch_w16(const AttnArgs& a, int hd, int nwin, bool bwd, hipStream_t s) {
  const bool sh = a.shift != 0;
  switch (hd) {
    case 16: return sh ? launch_w16<CT, 16, true, X3>(a, nwin, bwd, s) : launch_w16<CT, 16, false, X3>(a, nwin, bwd, s);
    case 32: return sh ? launch_w16<CT, 32, true, X3>(a, nwin, bwd, s) : launch_w16<CT, 32, false, X3>(a, nwin, bwd, s);
    case 64: return sh ? launch_w16<CT, 64, true, X3>(a, nwin, bwd, s) : launch_w16<CT, 64, false, X3>(a, nwin, bwd, s);
    default: return SCOT_ERR_UNSUPPORTED;
  }
}

// compute: SCOT_F32 / SCOT_BF16 / SCOT_BF16X3 (fp32 tensors, split bf16 MFMAs).
// returns SCOT_ERR_UNSUPPORTED when the geometry is not the fast path's (the caller then uses the general kernels)
int scot_attn_w16(const AttnArgs& a, int compute, int hd, int nwin, bool bwd, hipStream_t s) {
  if (!a.use_tr || a.ws != 16 || (a.shift != 0 && a.shift != 8)) return SCOT_ERR_UNSUPPORTED;
  if (compute == SCOT_BF16X3) return dispatch_w16<bf16_t, true>(a, hd, nwin, bwd, s);
  return compute == SCOT_BF16 ? dispatch_w16<bf16_t, false>(a, hd, nwin, bwd, s) : dispatch_w16<float, false>(a, hd, nwin, bwd, s);
}
