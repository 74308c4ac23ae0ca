// Probe: direct global -> LDS loads on gfx950 (global_load_lds_dwordx4): where does lane l's 16 bytes land, and does a source-side
// XOR swizzle give the layout gemm_fast.hip's swizzled K-contiguous tiles use?   build: hipcc --offload-arch=gfx950 -O3 glds_probe.hip -o glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

__global__ __launch_bounds__(256) void probe(const unsigned short* G, unsigned short* out) {
  __shared__ __attribute__((aligned(1024))) unsigned short tile[64 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 64 * 64; i += 256) tile[i] = 0xdead;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = wave + 4 * u;                    // 1 KB piece: rows 8q .. 8q+7
    const int row = 8 * q + (lane >> 3), pc = lane & 7;
    const int c = pc ^ ((row >> 1) & 7);           // the chunk of the row that must land at LDS position pc
    const unsigned short* src = G + row * 64 + c * 8;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(tile + q * 512), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 256) out[i] = tile[i];
}

int main() {
  std::vector<unsigned short> h(64 * 64), o(64 * 64);
  for (int i = 0; i < 64 * 64; ++i) h[i] = (unsigned short)i;
  unsigned short *dG, *dO;
  hipMalloc(&dG, h.size() * 2); hipMalloc(&dO, o.size() * 2);
  hipMemcpy(dG, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, dG, dO);
  hipMemcpy(o.data(), dO, o.size() * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int row = 0; row < 64; ++row)
    for (int pc = 0; pc < 8; ++pc)
      for (int j = 0; j < 8; ++j) {
        const int c = pc ^ ((row >> 1) & 7);
        if (o[row * 64 + pc * 8 + j] != h[row * 64 + c * 8 + j]) ++bad;
      }
  printf("glds probe: %d mismatches of 4096 (row 0: %u %u %u ; row 3 pos 0: %u expect %u)\n", bad, o[0], o[1], o[8], o[3 * 64], h[3 * 64 + (1 * 8)]);
  return bad != 0;
}
