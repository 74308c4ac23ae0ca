// Micro-probe: cycles per LDS atomic instruction on gfx950 as a function of active lanes / address pattern / type.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_atomic_probe.hip -o gpurun_out/lds_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void probe(long long* out, int active, int stride, int iters) {
  __shared__ float tabf[4096];
  __shared__ unsigned tabu[4096];
  __shared__ unsigned long long tabl[4096];
  __shared__ double tabd[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) { tabf[i] = 0.f; tabu[i] = 0; tabl[i] = 0; tabd[i] = 0.0; }
  __syncthreads();
  const int idx = (lane * stride) & 4095;
  long long t0 = clock64();
  if (lane < active) {
    for (int it = 0; it < iters; ++it) {
      if (MODE == 0) atomicAdd(&tabf[idx], 1.0f);                      // ds_add_f32
      if (MODE == 1) atomicAdd(&tabu[idx], 1u);                        // ds_add_u32
      if (MODE == 2) { float v = tabf[idx]; tabf[idx] = v + 1.0f; }    // plain read-add-write
      if (MODE == 3) atomicAdd(&tabl[idx], 1ull);                      // ds_add_u64
      if (MODE == 4) atomicAdd(&tabd[idx], 1.0);                       // ds_add_f64
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (tabf[threadIdx.x] == 123.f && tabu[threadIdx.x] == 77 && tabl[threadIdx.x] == 5 && tabd[threadIdx.x] == 3.0) out[0] = 0;
}

int main() {
  long long* d; hipMalloc(&d, 8 * 1024);
  const int iters = 256;
  const char* names[5] = {"ds_add_f32", "ds_add_u32", "read-add-write", "ds_add_u64", "ds_add_f64"};
  for (int mode = 3; mode < 5; ++mode)
    for (int waves = 1; waves <= 4; waves *= 4)
      for (int stride = 0; stride <= 8; stride = stride ? stride * 8 : 1)
        for (int active = 64; active >= 1; active /= 4) {
          long long h = 0;
          for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64 * waves), 0, 0, d, active, stride, iters);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64 * waves), 0, 0, d, active, stride, iters);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64 * waves), 0, 0, d, active, stride, iters);
            if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64 * waves), 0, 0, d, active, stride, iters);
            if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64 * waves), 0, 0, d, active, stride, iters);
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
          }
          printf("%-15s waves=%d stride=%d active=%2d : %7.1f clk/instr\n", names[mode], waves, stride, active, (double)h / iters);
        }
  return 0;
}
