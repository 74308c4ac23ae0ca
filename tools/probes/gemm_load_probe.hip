// Probe: what bounds the K loop of the deep-stage NT GEMMs (gemm_fast.hip, 64 x 64 tiles, 4 waves, three direct-to-LDS stages)?
// The stage-3 products ([1024, 768] x K = 3072, 192 workgroups) take ~0.5-0.67 us per 64-wide K tile = 16 KB per workgroup = ~30 GB/s per
// CU, far under the L1's 64 B/clk.  This probe runs that loop's memory side alone and in variants:
//   mode 0  gemm_fast's loads only: A and B tiles global -> LDS (global_load_lds_dwordx4, 8 rows x 128 B per wave instruction), wait, barrier
//   mode 1  mode 0 + the LDS fragment reads and the 8 MFMAs per wave and K tile (the loop as it runs)
//   mode 2  A as in mode 1; B from a FRAGMENT-ORDERED copy straight into MFMA operand registers (1 KiB contiguous per wave instruction, no LDS)
//   mode 3  A and B both fragment-ordered, registers only (no LDS, no barrier): the ceiling of a layout change
//   mode 4  mode 0 with the A loads only (half the bytes)
// Buffers rotate over `nrot` copies (1 = hot in L2 / MALL, 16 = what a step sees).   build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_p;
typedef __attribute__((address_space(1))) const void* gbl_p;

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int MODE>
__global__ __launch_bounds__(256) void kloop(const short* __restrict__ A, const short* __restrict__ B, int M, int N, int K, float* out) {
  constexpr int STAGE = 2 * 64 * 64;   // elements per stage: A tile then B tile
  __shared__ __attribute__((aligned(1024))) short lds[3 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lc = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64, nk = K / 64;
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};

  auto glds_tile = [&](short* st, int k0, bool a_on, bool b_on) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = wave + 4 * u, row = 8 * q + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      if (a_on) __builtin_amdgcn_global_load_lds((gbl_p)(A + (size_t)(m0 + row) * K + k0 + c * 8), (lds_p)(st + q * 512), 16, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = wave + 4 * u, row = 8 * q + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      if (b_on) __builtin_amdgcn_global_load_lds((gbl_p)(B + (size_t)(n0 + row) * K + k0 + c * 8), (lds_p)(st + 4096 + q * 512), 16, 0, 0);
    }
  };
  // swizzled K-contiguous tile read: fragment of rows r0 + lc, k-elements kk*32 + g*8 .. +8
  auto frag = [&](const short* tile, int r0, int kk) -> h16x8 {
    const int row = r0 + lc, pc = (kk * 4 + g) ^ ((row >> 1) & 7);
    return __builtin_bit_cast(h16x8, *(const s16x8*)(tile + row * 64 + pc * 8));
  };
  // fragment-ordered operand: [tile16][k32 step][lane][8]
  auto ffrag = [&](const short* P, int t16, int ks) -> s16x8 { return *(const s16x8*)(P + (((size_t)t16 * (K / 32) + ks) * 64 + lane) * 8); };

  if constexpr (MODE == 0 || MODE == 1 || MODE == 4) {
    constexpr bool BON = MODE != 4;
    constexpr int LPW = BON ? 4 : 2;
    glds_tile(lds, 0, true, BON);
    glds_tile(lds + STAGE, 64, true, BON);
    for (int t = 0; t < nk; ++t) {
      if (t + 1 < nk) VMCNT(LPW); else VMCNT(0);
      __builtin_amdgcn_s_barrier();
      if (t + 2 < nk) glds_tile(lds + ((t + 2) % 3) * STAGE, (t + 2) * 64, true, BON);
      if constexpr (MODE == 1) {
        const short* st = lds + (t % 3) * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          h16x8 fa[2], fb[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) { fa[i] = frag(st, wr * 32 + i * 16, kk); fb[i] = frag(st + 4096, wc * 32 + i * 16, kk); }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  } else if constexpr (MODE == 2) {
    // B fragments of the wave's two 16-column tiles: 4 loads per K tile, three register sets (two tiles in flight), issued BEFORE the
    // tile's A loads so that the counted wait for the A tile covers them
    s16x8 rb[3][4];
    const int bt0 = (n0 + wc * 32) / 16;
    auto bload = [&](s16x8 (&r)[4], int t) {
      t = t < nk ? t : nk - 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) r[kk * 2 + j] = ffrag(B, bt0 + j, t * 2 + kk);
    };
    auto step = [&](s16x8 (&cur)[4], s16x8 (&nxt2)[4], int t) {
      VMCNT(6);
      __builtin_amdgcn_s_barrier();
      bload(nxt2, t + 2);
      glds_tile(lds + ((t + 2) % 3) * STAGE, (t + 2 < nk ? t + 2 : nk - 1) * 64, true, false);
      const short* st = lds + (t % 3) * STAGE;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        h16x8 fa[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = frag(st, wr * 32 + i * 16, kk);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], __builtin_bit_cast(h16x8, cur[kk * 2 + j]), acc[i][j], 0, 0, 0);
      }
    };
    bload(rb[0], 0); glds_tile(lds, 0, true, false);
    bload(rb[1], 1); glds_tile(lds + STAGE, 64, true, false);
    for (int t = 0; t < nk; t += 3) {
      step(rb[0], rb[2], t);
      step(rb[1], rb[0], t + 1);
      step(rb[2], rb[1], t + 2);
    }
  } else {
    s16x8 ra[3][4], rb[3][4];
    const int at0 = (m0 + wr * 32) / 16, bt0 = (n0 + wc * 32) / 16;
    auto load = [&](s16x8 (&a)[4], s16x8 (&b)[4], int t) {
      t = t < nk ? t : nk - 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 2; ++j) { a[kk * 2 + j] = ffrag(A, at0 + j, t * 2 + kk); b[kk * 2 + j] = ffrag(B, bt0 + j, t * 2 + kk); }
    };
    auto step = [&](s16x8 (&ca)[4], s16x8 (&cb)[4], s16x8 (&na)[4], s16x8 (&nb)[4], int t) {
      load(na, nb, t + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, ca[kk * 2 + i]), __builtin_bit_cast(h16x8, cb[kk * 2 + j]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    load(ra[0], rb[0], 0);
    load(ra[1], rb[1], 1);
    for (int t = 0; t < nk; t += 3) {
      step(ra[0], rb[0], ra[2], rb[2], t);
      step(ra[1], rb[1], ra[0], rb[0], t + 1);
      step(ra[2], rb[2], ra[1], rb[1], t + 2);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 1234.5f) out[blockIdx.x] = s;
}

template <int MODE>
static void run(const short* A, const short* B, int M, int N, int K, float* out, int nrot, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t sa = (size_t)M * K, sb = (size_t)N * K;
  const dim3 grid(N / 64, M / 64), block(256);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((kloop<MODE>), grid, block, 0, 0, A + (it % nrot) * sa, B + (it % nrot) * sb, M, N, K, out);
  hipDeviceSynchronize();
  const int reps = 48;
  hipEventRecord(e0, 0);
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((kloop<MODE>), grid, block, 0, 0, A + (it % nrot) * sa, B + (it % nrot) * sb, M, N, K, out);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / reps;
  const double wg_bytes = (MODE == 4 ? 1.0 : 2.0) * 64.0 * K * 2;
  printf("mode %d %-34s M=%5d N=%4d K=%4d wgs=%4d nrot=%2d : %6.1f us/launch  %5.2f us per K tile  %6.1f GB/s per workgroup\n", MODE, what, M, N, K,
         grid.x * grid.y, nrot, us, us / (K / 64), wg_bytes / (us * 1e-6) / 1e9);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  const int NROT = 16, MMAX = 4096;
  short *A, *B; float* out;
  hipMalloc(&A, (size_t)NROT * MMAX * 1536 * 2);          // the largest A: [4096, 1536] or [1024, 3072]
  hipMalloc(&B, (size_t)NROT * MMAX * 1536 * 2);          // (same size: [3072, 768] and [768, 3072] fit)
  hipMalloc(&out, 4096 * 4);
  std::vector<short> h((size_t)MMAX * 1536);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (short)(0x2c00 + (i * 7919u) % 512);
  for (int r = 0; r < NROT; ++r) {
    hipMemcpy(A + (size_t)r * h.size(), h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B + (size_t)r * h.size(), h.data(), h.size() * 2, hipMemcpyHostToDevice);
  }
  struct { int M, N, K; } shapes[] = {{1024, 768, 3072}, {1024, 768, 768}, {4096, 384, 1536}, {4096, 384, 384}, {1024, 3072, 768}};
  for (auto s : shapes)
    for (int nrot : {1, 16}) {
      run<0>(A, B, s.M, s.N, s.K, out, nrot, "A+B direct-to-LDS, loads only");
      run<4>(A, B, s.M, s.N, s.K, out, nrot, "A only direct-to-LDS");
      run<1>(A, B, s.M, s.N, s.K, out, nrot, "A+B direct-to-LDS + LDS reads + MFMA");
      run<2>(A, B, s.M, s.N, s.K, out, nrot, "A to LDS, B fragment-ordered regs");
      run<3>(A, B, s.M, s.N, s.K, out, nrot, "A and B fragment-ordered regs");
    }
  return 0;
}
