// Micro-probe: how fast can EVERY CU stream the SAME weight matrix from L2 straight into MFMA B-operand fragments (no LDS)?
// This is the access pattern of a fused deep-stage layer tail that owns 16 token rows per workgroup (M = 4096 rows -> 256
// workgroups) and therefore uses every weight element exactly once per workgroup: lane (lc, g) of a wave loads 32 contiguous bytes
// of weight row n0 + lc (two 16-byte loads = the k-elements of two 16x16x32 MFMAs), i.e. a wave instruction pair covers 16 rows x
// 128 B.  Reported: us per launch and GB/s per CU for ring depths (U..2U pairs of 16-byte loads in flight per lane), waves per workgroup and grid sizes.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/l2_stream_probe.hip -o gpurun_out/l2_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// W: [NR][K] 16-bit, row-major.  Wave w of WPG takes 16-row tiles t = w, w + WPG, ...; per tile the K/64 double-steps.
template <int U, int WPG, int FRAG>
__global__ __launch_bounds__(WPG * 64) void stream_kernel(const short* __restrict__ W, int NR, int K, float* out, int rot_stride_elems,
                                                          int rot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lc = lane & 15, g = lane >> 4;
  const short* Wb = W + (size_t)rot * rot_stride_elems;
  const int ksteps = K / 64;                 // double k-steps per tile
  const int ntiles = NR / 16;
  const int my_tiles = (ntiles - wave + WPG - 1) / WPG;
  const int total = my_tiles * ksteps;       // pairs of loads this wave issues
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  h16x8 a;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)(0.001f * (lane + j));
  // two register sets of U load pairs each: the loads of one set are issued before the other set is consumed, so U..2U pairs
  // (2U..4U KiB per wave) are always in flight
  s16x8 ra0[U], ra1[U], rb0[U], rb1[U];
  auto addr = [&](int i) -> const short* {
    i = i < total ? i : 0;
    const int t = i / ksteps, ks = i - t * ksteps;
    const int tile = wave + t * WPG;
    if (FRAG) return Wb + ((size_t)tile * ksteps + ks) * 1024 + lane * 8;      // [tile][kstep][2][lane][8]: 2 x 1 KiB contiguous
    return Wb + (size_t)(tile * 16 + lc) * K + ks * 64 + g * 16;
  };
  auto issue = [&](s16x8 (&x0)[U], s16x8 (&x1)[U], int base) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const short* p = addr(base + u);
      x0[u] = *(const s16x8*)p;
      x1[u] = *(const s16x8*)(p + (FRAG ? 512 : 8));
    }
  };
  auto consume = [&](s16x8 (&x0)[U], s16x8 (&x1)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const h16x8 b0 = __builtin_bit_cast(h16x8, x0[u]), b1 = __builtin_bit_cast(h16x8, x1[u]);
      acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, acc[u & 3], 0, 0, 0);
      acc[(u + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc[(u + 2) & 3], 0, 0, 0);
    }
  };
  issue(ra0, ra1, 0);
  for (int i = 0; i < total; i += 2 * U) {
    issue(rb0, rb1, i + U);
    __builtin_amdgcn_sched_barrier(0);
    consume(ra0, ra1);
    __builtin_amdgcn_sched_barrier(0);
    issue(ra0, ra1, i + 2 * U);
    __builtin_amdgcn_sched_barrier(0);
    consume(rb0, rb1);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  if (s == 1234.5f) out[blockIdx.x] = s;
}

template <int U, int WPG, int FRAG = 0>
static void run(const short* W, int NR, int K, float* out, int grid, int nrot, const char* tag) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int rot_stride = NR * K;
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((stream_kernel<U, WPG, FRAG>), dim3(grid), dim3(WPG * 64), 0, 0, W, NR, K, out, rot_stride, it % nrot);
  hipDeviceSynchronize();
  const int reps = 40;
  hipEventRecord(e0, 0);
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((stream_kernel<U, WPG, FRAG>), dim3(grid), dim3(WPG * 64), 0, 0, W, NR, K, out, rot_stride, it % nrot);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1000.0 / reps, mb = (double)NR * K * 2 / 1e6;
  printf("%-10s U=%2d waves=%d grid=%4d NRxK=%5dx%4d (%.2f MB) nrot=%3d : %7.1f us/launch  %6.1f GB/s per WG  %6.2f TB/s chip\n", tag, U, WPG, grid, NR, K, mb,
         nrot, us, mb * 1e6 / (us * 1e-6) / 1e9, mb * 1e6 * grid / (us * 1e-6) / 1e12);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  // up to 128 rotating copies of a 3.54 MB matrix (Wo + W1 + W2 + Wqkv of a C = 384 layer: 4608 rows x 384): 453 MB > Infinity Cache
  const int K = 384, NR = 4608, NROT = 128;
  short* W; float* out;
  hipMalloc(&W, (size_t)NROT * NR * K * 2);
  hipMalloc(&out, 4096 * 4);
  std::vector<short> h((size_t)NR * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (short)(0x2c00 + (i * 7919u) % 512);
  for (int r = 0; r < NROT; ++r) hipMemcpy(W + (size_t)r * NR * K, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int nrot : {1, 128}) {
    run<4, 4>(W, NR, K, out, 256, nrot, "C384");
    run<8, 4>(W, NR, K, out, 256, nrot, "C384");
    run<12, 4>(W, NR, K, out, 256, nrot, "C384");
    run<6, 8>(W, NR, K, out, 256, nrot, "C384");
    run<8, 4>(W, NR, K, out, 512, nrot, "C384");
    run<8, 4>(W, NR, K, out, 64, nrot, "C384");
  }
  for (int nrot : {1, 128}) {
    run<4, 4, 1>(W, NR, K, out, 256, nrot, "C384frag");
    run<8, 4, 1>(W, NR, K, out, 256, nrot, "C384frag");
    run<12, 4, 1>(W, NR, K, out, 256, nrot, "C384frag");
    run<6, 8, 1>(W, NR, K, out, 256, nrot, "C384frag");
    run<8, 4, 1>(W, NR, K, out, 512, nrot, "C384frag");
    run<8, 4, 1>(W, NR, K, out, 64, nrot, "C384frag");
  }
  run<8, 4, 1>(W, 4608, 768, out, 256, 64, "C768qfrag");
  // the C = 768 layer quartered over the hidden dimension (Wo + W1/4 + W2/4 + Wqkv: ~7 MB as [NR][768]): 4608 rows x 768
  run<8, 4>(W, 4608, 768, out, 256, 1, "C768q");
  run<8, 4>(W, 4608, 768, out, 256, 64, "C768q");
  run<6, 8>(W, 4608, 768, out, 256, 64, "C768q");
  return 0;
}
