// Runs the fused block kernels of poseidon_amd/csrc/mlp_fused.hip on the CPU through the hipemu shim (hip/hip_runtime.h in
// this directory) and checks them against plain double-precision loops.  Test infrastructure only.
//   usage: emu_fused <case> <C> <B> <L> <cond 0|1> <train 0|1> <use_tr 0|1>      (SCOT_MLP_TT=1|2 selects rows per workgroup)
//   cases: mlp_fwd, mlp_bwd, proj_fwd, proj_bwd          exit code 0 = all checks passed
#include "../../poseidon_amd/csrc/mlp_fused.hip"

#include <cstdio>
#include <random>
#include <string>

int g_scot_use_tr = 1;
// (defined in norm_fast.hip, which this stand-alone harness does not build: scot_partial_colsum is not exercised here)
int scot_cln_bwd_finish_launch(const float*, int, int, float*, void*) { return -3; }

typedef std::vector<float> V;
typedef std::vector<uint16_t> H;
static std::mt19937 rng(1234);
static V randn(size_t n, float scale, float shift = 0.f) {
  std::normal_distribution<float> d(0.f, 1.f);
  V v(n);
  for (auto& x : v) x = d(rng) * scale + shift;
  return v;
}
static uint16_t tobf(float f) { return f2bf(f); }
static H tobf(const V& v) { H h(v.size()); for (size_t i = 0; i < v.size(); ++i) h[i] = tobf(v[i]); return h; }
static V tof(const H& h) { V v(h.size()); for (size_t i = 0; i < h.size(); ++i) v[i] = bf2f(h[i]); return v; }
static double gelu(double x) { return 0.5 * x * (1.0 + std::erf(x / std::sqrt(2.0))); }
static double dgelu(double x) { return 0.5 * (1.0 + std::erf(x / std::sqrt(2.0))) + x * std::exp(-0.5 * x * x) / std::sqrt(2.0 * M_PI); }

static int fails = 0;
template <typename A, typename B> static void check(const char* what, const A& got, const B& ref, double tol) {
  double num = 0, den = 0;
  bool finite = true;
  for (size_t i = 0; i < ref.size(); ++i) {
    const double g = got[i], r = ref[i];
    if (!std::isfinite(g)) finite = false;
    num += (g - r) * (g - r); den += r * r;
  }
  const double e = std::sqrt(num / std::max(den, 1e-300));
  const bool ok = finite && e < tol;
  printf("  %-10s rel-L2 %.3e (tol %.1e) %s\n", what, e, tol, ok ? "ok" : "FAIL");
  if (!ok) ++fails;
}

struct Norm { V gw_w, gw_b, bw_w, bw_b, time, sc; };
static Norm make_norm(int C, int B) {
  Norm n;
  n.gw_w = randn(C, 0.3f); n.gw_b = randn(C, 0.1f, 1.f); n.bw_w = randn(C, 0.1f); n.bw_b = randn(C, 0.1f);
  n.time = V(B); n.sc = V(B);
  for (int b = 0; b < B; ++b) { n.time[b] = 0.1f * (b + 1); n.sc[b] = (b % 3 == 1) ? 0.f : 1.f / 0.7f; }
  return n;
}
// out = resid + sc * (gamma * LN(z) + beta); also mean / rstd
static void ref_cln(const std::vector<double>& z, const V& resid, const Norm& n, bool cond, int M, int L, int C,
                    std::vector<double>& out, std::vector<double>& mean, std::vector<double>& rstd) {
  out.assign((size_t)M * C, 0); mean.assign(M, 0); rstd.assign(M, 0);
  for (int r = 0; r < M; ++r) {
    double mu = 0, var = 0;
    for (int c = 0; c < C; ++c) mu += z[(size_t)r * C + c];
    mu /= C;
    for (int c = 0; c < C; ++c) var += (z[(size_t)r * C + c] - mu) * (z[(size_t)r * C + c] - mu);
    var /= C;
    const double rs = 1.0 / std::sqrt(var + 1e-5);
    mean[r] = mu; rstd[r] = rs;
    const int b = r / L;
    for (int c = 0; c < C; ++c) {
      const double ga = cond ? n.gw_w[c] * n.time[b] + n.gw_b[c] : n.gw_b[c];
      const double be = cond ? n.bw_w[c] * n.time[b] + n.bw_b[c] : n.bw_b[c];
      out[(size_t)r * C + c] = resid[(size_t)r * C + c] + n.sc[b] * (ga * (z[(size_t)r * C + c] - mu) * rs + be);
    }
  }
}
// dz = CLN_bwd(sc * g), parameter gradients
static void ref_cln_bwd(const V& g, const V& z, const V& mean, const V& rstd, const Norm& n, bool cond, int M, int L, int C,
                        std::vector<double>& dz, std::vector<double> (&pg)[4]) {
  dz.assign((size_t)M * C, 0);
  for (auto& p : pg) p.assign(C, 0);
  for (int r = 0; r < M; ++r) {
    const int b = r / L;
    double m1 = 0, m2 = 0;
    std::vector<double> gd(C), xh(C);
    for (int c = 0; c < C; ++c) {
      const double ga = cond ? n.gw_w[c] * n.time[b] + n.gw_b[c] : n.gw_b[c];
      const double d = (double)g[(size_t)r * C + c] * n.sc[b];
      xh[c] = ((double)z[(size_t)r * C + c] - mean[r]) * rstd[r];
      gd[c] = d * ga;
      m1 += gd[c]; m2 += gd[c] * xh[c];
      pg[1][c] += d * xh[c]; pg[0][c] += d * xh[c] * n.time[b];
      pg[3][c] += d; pg[2][c] += d * n.time[b];
    }
    m1 /= C; m2 /= C;
    for (int c = 0; c < C; ++c) dz[(size_t)r * C + c] = rstd[r] * (gd[c] - m1 - xh[c] * m2);
  }
}

static int run_mlp_fwd(int C, int B, int L, bool cond, bool train) {
  const int M = B * L, HID = 4 * C;
  V h = randn((size_t)M * C, 1.f);
  H h16 = tobf(h);
  H W1 = tobf(randn((size_t)HID * C, 1.f / std::sqrt((float)C))), W2 = tobf(randn((size_t)C * HID, 1.f / std::sqrt((float)HID)));
  V b1 = randn(HID, 0.2f), b2 = randn(C, 0.2f);
  Norm n = make_norm(C, B);
  const float nanv = std::nanf("");
  V out((size_t)M * C, nanv), z((size_t)M * C, nanv), mean(M, nanv), rstd(M, nanv);
  H out16((size_t)M * C, 0x7fc0), act((size_t)M * HID, 0x7fc0), dact((size_t)M * HID, 0x7fc0);
  const int rc = scot_mlp_block_fwd(h16.data(), h.data(), W1.data(), b1.data(), W2.data(), b2.data(), out.data(), out16.data(),
                                    train ? act.data() : nullptr, train ? dact.data() : nullptr, train ? z.data() : nullptr,
                                    train ? mean.data() : nullptr, train ? rstd.data() : nullptr, cond ? n.time.data() : nullptr,
                                    cond ? n.gw_w.data() : nullptr, n.gw_b.data(), cond ? n.bw_w.data() : nullptr, n.bw_b.data(),
                                    n.sc.data(), M, L, C, HID, 1e-5f, nullptr);
  if (rc) { printf("  rc = %d\n", rc); return 1; }
  // reference
  V h16f = tof(h16), W1f = tof(W1), W2f = tof(W2);
  std::vector<double> a_ref((size_t)M * HID), d_ref((size_t)M * HID);
  for (int r = 0; r < M; ++r)
    for (int j = 0; j < HID; ++j) {
      double u = b1[j];
      for (int k = 0; k < C; ++k) u += (double)h16f[(size_t)r * C + k] * W1f[(size_t)j * C + k];
      a_ref[(size_t)r * HID + j] = gelu(u); d_ref[(size_t)r * HID + j] = dgelu(u);
    }
  // z from the bf16-rounded activation: the kernel's own when it stored it (sharp check of GEMM 2), else the reference's
  V a_used = train ? tof(act) : V();
  std::vector<double> zr((size_t)M * C);
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < C; ++c) {
      double s = b2[c];
      for (int j = 0; j < HID; ++j)
        s += (train ? (double)a_used[(size_t)r * HID + j] : (double)bf2f(tobf((float)a_ref[(size_t)r * HID + j]))) * W2f[(size_t)c * HID + j];
      zr[(size_t)r * C + c] = s;
    }
  std::vector<double> outr, meanr, rstdr;
  ref_cln(zr, h, n, cond, M, L, C, outr, meanr, rstdr);
  const double tz = train ? 2e-5 : 3e-3;
  check("out", out, outr, tz);
  V o16 = tof(out16);
  check("out16", o16, outr, 4e-3);
  if (train) {
    check("act", tof(act), a_ref, 4e-3);
    check("dact", tof(dact), d_ref, 4e-3);
    check("z", z, zr, 2e-5); check("mean", mean, meanr, 1e-4); check("rstd", rstd, rstdr, 1e-4);
  }
  return 0;
}

static int run_mlp_bwd(int C, int B, int L, bool cond) {
  const int M = B * L, HID = 4 * C;
  V g = randn((size_t)M * C, 1.f), z = randn((size_t)M * C, 1.5f, 0.3f);
  V mean(M), rstd(M);
  for (int r = 0; r < M; ++r) {
    double mu = 0, var = 0;
    for (int c = 0; c < C; ++c) mu += z[(size_t)r * C + c];
    mu /= C;
    for (int c = 0; c < C; ++c) var += (z[(size_t)r * C + c] - mu) * (z[(size_t)r * C + c] - mu);
    mean[r] = (float)mu; rstd[r] = (float)(1.0 / std::sqrt(var / C + 1e-5));
  }
  H gp = tobf(randn((size_t)M * HID, 0.5f));
  H W1 = tobf(randn((size_t)HID * C, 1.f / std::sqrt((float)C))), W2 = tobf(randn((size_t)C * HID, 1.f / std::sqrt((float)HID)));
  Norm n = make_norm(C, B);
  H dz((size_t)M * C, 0x7fc0), du((size_t)M * HID, 0x7fc0);
  V gout((size_t)M * C, std::nanf(""));
  V pg[4] = {V(C, 0.f), V(C, 0.f), V(C, 0.f), V(C, 0.f)};
  const int rc = scot_mlp_block_bwd(g.data(), gout.data(), z.data(), mean.data(), rstd.data(), cond ? n.time.data() : nullptr,
                                    cond ? n.gw_w.data() : nullptr, n.gw_b.data(), n.sc.data(), gp.data(), W1.data(), W2.data(),
                                    dz.data(), du.data(), cond ? pg[0].data() : nullptr, pg[1].data(), cond ? pg[2].data() : nullptr,
                                    pg[3].data(), M, L, C, HID, nullptr);
  if (rc) { printf("  rc = %d\n", rc); return 1; }
  std::vector<double> dzr, pgr[4];
  ref_cln_bwd(g, z, mean, rstd, n, cond, M, L, C, dzr, pgr);
  check("dz", tof(dz), dzr, 4e-3);
  for (int i : {0, 1, 2, 3}) if (cond || (i & 1)) check(i == 0 ? "d_gw_w" : i == 1 ? "d_gw_b" : i == 2 ? "d_bw_w" : "d_bw_b", pg[i], pgr[i], 1e-4);
  // du from the kernel's own dz, g_out from the kernel's own du (sharp checks of the two GEMMs)
  V dzk = tof(dz), W1f = tof(W1), W2f = tof(W2), gpf = tof(gp);
  std::vector<double> dur((size_t)M * HID), gor((size_t)M * C);
  for (int r = 0; r < M; ++r)
    for (int j = 0; j < HID; ++j) {
      double s = 0;
      for (int c = 0; c < C; ++c) s += (double)dzk[(size_t)r * C + c] * W2f[(size_t)c * HID + j];
      dur[(size_t)r * HID + j] = s * gpf[(size_t)r * HID + j];
    }
  check("du", tof(du), dur, 4e-3);
  V duk = tof(du);
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < C; ++c) {
      double s = g[(size_t)r * C + c];
      for (int j = 0; j < HID; ++j) s += (double)duk[(size_t)r * HID + j] * W1f[(size_t)j * C + c];
      gor[(size_t)r * C + c] = s;
    }
  check("g_out", gout, gor, 2e-5);
  // in place
  V gi = g;
  H dz2(dz.size()), du2(du.size());
  V pz[4] = {V(C, 0.f), V(C, 0.f), V(C, 0.f), V(C, 0.f)};
  scot_mlp_block_bwd(gi.data(), gi.data(), z.data(), mean.data(), rstd.data(), cond ? n.time.data() : nullptr,
                     cond ? n.gw_w.data() : nullptr, n.gw_b.data(), n.sc.data(), gp.data(), W1.data(), W2.data(), dz2.data(),
                     du2.data(), cond ? pz[0].data() : nullptr, pz[1].data(), cond ? pz[2].data() : nullptr, pz[3].data(), M, L, C, HID,
                     nullptr);
  check("in-place", gi, gout, 1e-7);
  return 0;
}

static int run_proj_fwd(int C, int B, int L, bool cond, bool train) {
  const int M = B * L;
  H a = tobf(randn((size_t)M * C, 1.f)), W = tobf(randn((size_t)C * C, 1.f / std::sqrt((float)C)));
  V x = randn((size_t)M * C, 1.f), bias = randn(C, 0.2f);
  Norm n = make_norm(C, B);
  const float nanv = std::nanf("");
  V out((size_t)M * C, nanv), z((size_t)M * C, nanv), mean(M, nanv), rstd(M, nanv);
  H out16((size_t)M * C, 0x7fc0);
  const int rc = scot_proj_cln_fwd(a.data(), W.data(), bias.data(), x.data(), out.data(), out16.data(), train ? z.data() : nullptr,
                                   train ? mean.data() : nullptr, train ? rstd.data() : nullptr, cond ? n.time.data() : nullptr,
                                   cond ? n.gw_w.data() : nullptr, n.gw_b.data(), cond ? n.bw_w.data() : nullptr, n.bw_b.data(),
                                   n.sc.data(), M, L, C, 1e-5f, nullptr);
  if (rc) { printf("  rc = %d\n", rc); return 1; }
  V af = tof(a), Wf = tof(W);
  std::vector<double> zr((size_t)M * C);
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < C; ++c) {
      double s = bias[c];
      for (int k = 0; k < C; ++k) s += (double)af[(size_t)r * C + k] * Wf[(size_t)c * C + k];
      zr[(size_t)r * C + c] = s;
    }
  std::vector<double> outr, meanr, rstdr;
  ref_cln(zr, x, n, cond, M, L, C, outr, meanr, rstdr);
  check("out", out, outr, 2e-5);
  check("out16", tof(out16), outr, 4e-3);
  if (train) { check("z", z, zr, 2e-5); check("mean", mean, meanr, 1e-4); check("rstd", rstd, rstdr, 1e-4); }
  return 0;
}

static int run_proj_bwd(int C, int B, int L, bool cond) {
  const int M = B * L;
  V g = randn((size_t)M * C, 1.f), z = randn((size_t)M * C, 1.5f, 0.3f);
  V mean(M), rstd(M);
  for (int r = 0; r < M; ++r) {
    double mu = 0, var = 0;
    for (int c = 0; c < C; ++c) mu += z[(size_t)r * C + c];
    mu /= C;
    for (int c = 0; c < C; ++c) var += (z[(size_t)r * C + c] - mu) * (z[(size_t)r * C + c] - mu);
    mean[r] = (float)mu; rstd[r] = (float)(1.0 / std::sqrt(var / C + 1e-5));
  }
  H W = tobf(randn((size_t)C * C, 1.f / std::sqrt((float)C)));
  Norm n = make_norm(C, B);
  H dz((size_t)M * C, 0x7fc0), da((size_t)M * C, 0x7fc0);
  V pg[4] = {V(C, 0.f), V(C, 0.f), V(C, 0.f), V(C, 0.f)};
  const int rc = scot_proj_cln_bwd(g.data(), z.data(), mean.data(), rstd.data(), cond ? n.time.data() : nullptr,
                                   cond ? n.gw_w.data() : nullptr, n.gw_b.data(), n.sc.data(), W.data(), dz.data(), da.data(),
                                   cond ? pg[0].data() : nullptr, pg[1].data(), cond ? pg[2].data() : nullptr, pg[3].data(), M, L, C,
                                   nullptr);
  if (rc) { printf("  rc = %d\n", rc); return 1; }
  std::vector<double> dzr, pgr[4];
  ref_cln_bwd(g, z, mean, rstd, n, cond, M, L, C, dzr, pgr);
  check("dz", tof(dz), dzr, 4e-3);
  for (int i : {0, 1, 2, 3}) if (cond || (i & 1)) check(i == 0 ? "d_gw_w" : i == 1 ? "d_gw_b" : i == 2 ? "d_bw_w" : "d_bw_b", pg[i], pgr[i], 1e-4);
  V dzk = tof(dz), Wf = tof(W);
  std::vector<double> dar((size_t)M * C);
  for (int r = 0; r < M; ++r)
    for (int k = 0; k < C; ++k) {
      double s = 0;
      for (int c = 0; c < C; ++c) s += (double)dzk[(size_t)r * C + c] * Wf[(size_t)c * C + k];
      dar[(size_t)r * C + k] = s;
    }
  check("da", tof(da), dar, 4e-3);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: emu_fused <case> <C> <B> <L> <cond> <train> <use_tr>\n"); return 2; }
  const std::string what = argv[1];
  const int C = atoi(argv[2]), B = atoi(argv[3]), L = atoi(argv[4]);
  const bool cond = atoi(argv[5]), train = atoi(argv[6]);
  g_scot_use_tr = atoi(argv[7]);
  printf("%s C=%d B=%d L=%d cond=%d train=%d use_tr=%d\n", what.c_str(), C, B, L, (int)cond, (int)train, g_scot_use_tr);
  int rc = 2;
  if (what == "mlp_fwd") rc = run_mlp_fwd(C, B, L, cond, train);
  else if (what == "mlp_bwd") rc = run_mlp_bwd(C, B, L, cond);
  else if (what == "proj_fwd") rc = run_proj_fwd(C, B, L, cond, train);
  else if (what == "proj_bwd") rc = run_proj_bwd(C, B, L, cond);
  return rc ? rc : (fails ? 1 : 0);
}
