// hipemu — a tiny HOST stand-in for <hip/hip_runtime.h>, test infrastructure only (never linked into libscot_hip.so).
//
// It lets a kernel source file of poseidon_amd/csrc/ be compiled as plain C++ (x86, the ROCm clang) and executed on the CPU
// with one fiber (or, with -DHIPEMU_THREADS, one OS thread) per work-item, so that the index algebra, LDS aliasing and barrier placement of a kernel can be
// checked WITHOUT a GPU (tests/test_hipemu_cpu.py).  What is emulated, and with which semantics:
//   * threadIdx / blockIdx / blockDim / gridDim, __shared__ (a static array: blocks run one after another), __syncthreads
//   * wave64 cross-lane operations by rendezvous on a per-wave barrier: __shfl_xor, v_mfma_f32_16x16x32_bf16 (the fragment
//     convention of csrc/common.h: A lane (r, g) = row r, k = 8g..8g+7; B lane (r, g) = column r; D lane (c, g) = column c,
//     rows 4g..4g+3), v_mfma_f32_16x16x4_f32, ds_read_b64_tr_b16 (as documented at lds_frag_ks in common.h)
//   * __builtin_amdgcn_wave_barrier() is a REAL wave barrier here (lanes are threads; on the GPU a wave runs in lockstep)
//   * atomicAdd(float*), float4/uint4/dim3, hipLaunchKernelGGL (synchronous)
// NOT emulated: timing, bank conflicts, register limits, memory-model subtleties — a pass here says the arithmetic and the
// data movement are right, nothing about speed.
#pragma once
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
inline thread_local uint3_emu threadIdx, blockIdx;
inline thread_local dim3 blockDim, gridDim;

struct float4 { float x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
struct int4 { int x, y, z, w; };
struct float2 { float x, y; };
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
inline float2 make_float2(float x, float y) { return {x, y}; }
inline int2 make_int2(int x, int y) { return {x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
#define hipFuncAttributeMaxDynamicSharedMemorySize 0
inline int g_emu_error = 0;   // set when a launch could not be emulated (see run_block); reported once through hipGetLastError
inline hipError_t hipGetLastError() { const int e = g_emu_error; g_emu_error = 0; return e; }
template <typename T> inline hipError_t hipFuncSetAttribute(T, int, int) { return hipSuccess; }

using std::max;
using std::min;
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __expf(float x) { return std::exp(x); }
inline float __logf(float x) { return std::log(x); }
#define __log2f(x) std::log2((float)(x))        /* glibc declares (but does not export) a function of this name */
inline float emu_exp2f(float x) { return std::exp2(x); }
inline float emu_rcpf(float x) { return 1.0f / x; }
#define __builtin_amdgcn_rcpf emu_rcpf
#define __builtin_amdgcn_exp2f emu_exp2f
#define hipMemcpyDeviceToHost 2
inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, int, hipStream_t) { std::memcpy(dst, src, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* dst, int byte, size_t n, hipStream_t) { std::memset(dst, byte, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)std::malloc(n); return *p ? hipSuccess : 1; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }

namespace hipemu {
#ifdef HIPEMU_THREADS
// one OS thread per work-item: slow (every cross-lane operation is a 64-thread rendezvous in the kernel's scheduler) but it is
// what ThreadSanitizer / AddressSanitizer understand — missing barriers show up as data races
struct Barrier {
  std::barrier<> b;
  explicit Barrier(unsigned n) : b(n) {}
  void arrive_and_wait() { b.arrive_and_wait(); }
};
#else
// default: the work-items of a block are FIBERS of one OS thread, switched by hand (callee-saved registers + stack pointer)
// whenever one of them has to wait at a barrier — deterministic, and ~50x faster than OS threads for these kernels
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  const uint64_t* wait_gen = nullptr;   // parked at a barrier until *wait_gen != wait_val
  uint64_t wait_val = 0;
};
struct Sched {
  std::vector<Fiber> fib;
  void* main_sp = nullptr;
  unsigned cur = 0;
  uint64_t events = 0;                  // barrier arrivals, published exchanges, completions: a scheduler round without any = deadlock
  std::function<void(unsigned)> job;
};
inline Sched* g_sched = nullptr;
__attribute__((naked, noinline)) static void emu_switch(void** /*save_sp*/, void* /*to_sp*/) {
  asm volatile(
      "pushq %rbp\n pushq %rbx\n pushq %r12\n pushq %r13\n pushq %r14\n pushq %r15\n"
      "movq %rsp, (%rdi)\n"
      "movq %rsi, %rsp\n"
      "popq %r15\n popq %r14\n popq %r13\n popq %r12\n popq %rbx\n popq %rbp\n"
      "ret\n");
}
struct Barrier {
  unsigned n, count = 0;
  uint64_t gen = 0;
  explicit Barrier(unsigned n_) : n(n_) {}
  void arrive_and_wait() {
    ++g_sched->events;
    if (++count == n) { count = 0; ++gen; return; }
    Fiber& f = g_sched->fib[g_sched->cur];
    f.wait_gen = &gen;
    f.wait_val = gen;
    emu_switch(&f.sp, g_sched->main_sp);     // resumed by the scheduler once gen has moved on
  }
};
// give the other work-items a turn (the caller re-checks its own condition when it is resumed)
inline void park() {
  Fiber& f = g_sched->fib[g_sched->cur];
  f.wait_gen = nullptr;
  emu_switch(&f.sp, g_sched->main_sp);
}
#endif
struct Wave {
  Barrier bar{64};
  float f[64];
  uint64_t a[64][2], b[64][2];     // 8 x b16 operands
  float fa[64][8], fb[64][8];      // fp32 operands
  const void* ptr[64];
#ifndef HIPEMU_THREADS
  uint32_t xcnt[64][64] = {};      // xcnt[l][p]: exchanges lane l has published for lane p
  float xval[64][64][2];
#endif
};
struct Block {
  std::unique_ptr<Barrier> bar;
  std::vector<std::unique_ptr<Wave>> waves;
};
inline Block* g_block = nullptr;
inline Wave& wave() { return *g_block->waves[threadIdx.x >> 6]; }
inline int lane() { return threadIdx.x & 63; }
inline float bf2f(uint16_t x) { return __uint_as_float(((uint32_t)x) << 16); }
}  // namespace hipemu

inline void __syncthreads() { hipemu::g_block->bar->arrive_and_wait(); }
inline void emu_wave_barrier() { hipemu::wave().bar.arrive_and_wait(); }
#define __builtin_amdgcn_wave_barrier emu_wave_barrier
// global_load_lds_*: lane l's `size` bytes land at the (wave-uniform) LDS base + l * size; synchronous here, so the vmcnt waits of the
// kernels are no-ops (SCOT_HIPEMU) and s_barrier is the block barrier
#define SCOT_HIPEMU 1
template <typename G, typename L> inline void emu_global_load_lds(G g, L l, int size, int off, int) {
  std::memcpy((char*)(const void*)l + (hipemu::lane() & 63) * size + off, (const void*)g, (size_t)size);
}
#define __builtin_amdgcn_global_load_lds emu_global_load_lds
#define __builtin_amdgcn_s_barrier __syncthreads
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

#ifdef HIPEMU_THREADS
inline float __shfl_xor(float v, int mask, int = 64) {
  auto& w = hipemu::wave();
  const int l = hipemu::lane();
  w.f[l] = v;
  w.bar.arrive_and_wait();
  const float r = w.f[l ^ mask];
  w.bar.arrive_and_wait();
  return r;
}
#else
// Pairwise rendezvous instead of a whole-wave one: lane l only needs lane l^mask.  The k-th exchange between a pair matches the
// partner's k-th exchange with it, whatever else either lane has executed in between — so shuffles inside sub-wave groups that
// run different trip counts (cln_bwd_fast_kernel on ragged row counts), and the cross-group shuffles after such a loop, pair up
// exactly as the hardware's lane masking / reconvergence makes them.
inline float __shfl_xor(float v, int mask, int = 64) {
  auto& w = hipemu::wave();
  const int l = hipemu::lane(), p = (l ^ mask) & 63;
  const uint32_t k = w.xcnt[l][p]++;
  w.xval[l][p][k & 1] = v;
  ++hipemu::g_sched->events;
  while (w.xcnt[p][l] <= k) hipemu::park();
  return w.xval[p][l][k & 1];
}
#endif

typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef short emu_s16x4 __attribute__((ext_vector_type(4)));

inline emu_f32x4 emu_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c, int, int, int) {
  auto& w = hipemu::wave();
  const int l = hipemu::lane(), g = l >> 4, lc = l & 15;
  std::memcpy(w.a[l], &a, 16);
  std::memcpy(w.b[l], &b, 16);
  w.bar.arrive_and_wait();
  emu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * g + r;
    float s = c[r];
    for (int gg = 0; gg < 4; ++gg) {
      const uint16_t* pa = (const uint16_t*)w.a[gg * 16 + row];
      const uint16_t* pb = (const uint16_t*)w.b[gg * 16 + lc];
      for (int j = 0; j < 8; ++j) s += hipemu::bf2f(pa[j]) * hipemu::bf2f(pb[j]);
    }
    d[r] = s;
  }
  w.bar.arrive_and_wait();
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 emu_mfma_f32_16x16x32_bf16

// v_mfma_f32_16x16x32_f16: same operand layout, IEEE binary16 elements (the -DSCOT_OPERAND_FP16 build of the kernels)
typedef _Float16 emu_f16x8 __attribute__((ext_vector_type(8)));
inline emu_f32x4 emu_mfma_f32_16x16x32_f16(emu_f16x8 a, emu_f16x8 b, emu_f32x4 c, int, int, int) {
  auto& w = hipemu::wave();
  const int l = hipemu::lane(), g = l >> 4, lc = l & 15;
  std::memcpy(w.a[l], &a, 16);
  std::memcpy(w.b[l], &b, 16);
  w.bar.arrive_and_wait();
  emu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * g + r;
    float s = c[r];
    for (int gg = 0; gg < 4; ++gg) {
      const _Float16* pa = (const _Float16*)w.a[gg * 16 + row];
      const _Float16* pb = (const _Float16*)w.b[gg * 16 + lc];
      for (int j = 0; j < 8; ++j) s += (float)pa[j] * (float)pb[j];
    }
    d[r] = s;
  }
  w.bar.arrive_and_wait();
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 emu_mfma_f32_16x16x32_f16

// v_mfma_f32_16x16x4_f32: A lane (r, g) holds A[r][k = g], B lane (r, g) holds B[k = g][r]
inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  auto& w = hipemu::wave();
  const int l = hipemu::lane(), g = l >> 4, lc = l & 15;
  w.fa[l][0] = a; w.fb[l][0] = b;
  w.bar.arrive_and_wait();
  emu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    float s = c[r];
    for (int k = 0; k < 4; ++k) s = std::fma(w.fa[k * 16 + 4 * g + r][0], w.fb[k * 16 + lc][0], s);
    d[r] = s;
  }
  w.bar.arrive_and_wait();
  return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_f32_16x16x4f32

// ds_read_b64_tr_b16: within a 16-lane group, lane i supplies the address of 4 consecutive b16 = block[i>>2][4(i&3) .. +3];
// lane c receives column c of that [4][16] block.
template <typename P> inline emu_s16x4 emu_ds_read_tr16_b64(P p) {
  auto& w = hipemu::wave();
  const int l = hipemu::lane(), base = l & ~15, c = l & 15;
  w.ptr[l] = (const void*)p;
  w.bar.arrive_and_wait();
  emu_s16x4 r;
  for (int k = 0; k < 4; ++k) r[k] = ((const short*)w.ptr[base + 4 * k + (c >> 2)])[c & 3];
  w.bar.arrive_and_wait();
  return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16 emu_ds_read_tr16_b64

// v_mov_b32 dpp row_shl:n (0x101..0x10F): lane i of a 16-lane row reads lane i+n;  row_shr:n (0x111..0x11F): lane i-n;
// bound_ctrl = true: lanes without a source get 0.  (the only DPP controls the kernels use)
inline int emu_update_dpp(int, int v, int ctrl, int, int, bool) {
  auto& w = hipemu::wave();
  const int l = hipemu::lane(), row = l & ~15, i = l & 15;
  std::memcpy(&w.f[l], &v, 4);
  w.bar.arrive_and_wait();
  int src = -1;
  if (ctrl >= 0x101 && ctrl <= 0x10F) src = i + (ctrl - 0x100);
  else if (ctrl >= 0x111 && ctrl <= 0x11F) src = i - (ctrl - 0x110);
  else { std::fprintf(stderr, "hipemu: unsupported DPP control 0x%x\n", ctrl); std::abort(); }
  int r = 0;
  if (src >= 0 && src < 16) std::memcpy(&r, &w.f[row + src], 4);
  w.bar.arrive_and_wait();
  return r;
}
#define __builtin_amdgcn_update_dpp emu_update_dpp

inline unsigned long long __ballot(int pred) {
  auto& w = hipemu::wave();
  const int l = hipemu::lane();
  w.f[l] = pred ? 1.f : 0.f;
  w.bar.arrive_and_wait();
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) if (w.f[i] != 0.f) m |= 1ull << i;
  w.bar.arrive_and_wait();
  return m;
}

inline float atomicAdd(float* p, float v) {
  std::atomic_ref<float> r(*p);
  float old = r.load();
  while (!r.compare_exchange_weak(old, old + v)) {}
  return old;
}
inline int atomicAdd(int* p, int v) { return std::atomic_ref<int>(*p).fetch_add(v); }
inline double atomicAdd(double* p, double v) {
  std::atomic_ref<double> r(*p);
  double old = r.load();
  while (!r.compare_exchange_weak(old, old + v)) {}
  return old;
}

namespace hipemu {
// dynamic shared memory: the test build rewrites `extern __shared__ T name[];` into `T* name = (T*)hipemu::dyn_smem();`
inline char* dyn_smem() {
  alignas(16) static char buf[160 * 1024];
  return buf;
}
inline std::mutex& launch_mutex() {
  static std::mutex* m = new std::mutex;
  return *m;
}
#ifdef HIPEMU_THREADS
// Worker threads are created once and reused for every block of every launch.  Never destroyed: the pool is leaked on purpose
// so that no joinable thread meets a static destructor.
struct Pool {
  std::mutex m;
  std::condition_variable cv_start, cv_done;
  std::vector<std::thread> th;
  std::function<void(unsigned)> job;
  unsigned nactive = 0, remaining = 0;
  uint64_t gen = 0;
  void ensure(unsigned n) {
    while (th.size() < n) {
      const unsigned id = (unsigned)th.size();
      th.emplace_back([this, id] {
        uint64_t seen = 0;
        for (;;) {
          std::unique_lock<std::mutex> lk(m);
          cv_start.wait(lk, [&] { return gen != seen && id < nactive; });
          seen = gen;
          auto j = job;
          lk.unlock();
          j(id);
          lk.lock();
          if (--remaining == 0) cv_done.notify_one();
        }
      });
      th.back().detach();
    }
  }
  void run(unsigned n, std::function<void(unsigned)> f) {
    std::unique_lock<std::mutex> lk(m);
    ensure(n);
    job = std::move(f);
    nactive = n;
    remaining = n;
    ++gen;
    cv_start.notify_all();
    cv_done.wait(lk, [&] { return remaining == 0; });
    nactive = 0;
  }
};
inline Pool& pool() {
  static Pool* p = new Pool;
  return *p;
}
inline void run_block(unsigned nthr, std::function<void(unsigned)> f) { pool().run(nthr, std::move(f)); }
#else
constexpr size_t kFiberStack = 256 * 1024;
inline void fiber_entry() {
  Sched* s = g_sched;
  const unsigned me = s->cur;
  s->job(me);
  s->fib[me].done = true;
  ++s->events;
  void* dummy;
  emu_switch(&dummy, s->main_sp);
  std::abort();   // a finished fiber is never resumed
}
inline void run_block(unsigned nthr, std::function<void(unsigned)> f) {
  static Sched* s = new Sched;
  g_sched = s;
  s->job = std::move(f);
  while (s->fib.size() < nthr) {
    Fiber fb;
    fb.stack = (char*)std::aligned_alloc(64, kFiberStack);
    s->fib.push_back(fb);
  }
  for (unsigned t = 0; t < nthr; ++t) {
    Fiber& fb = s->fib[t];
    fb.done = false;
    fb.wait_gen = nullptr;
    void** top = (void**)(fb.stack + kFiberStack);     // 64-byte aligned
    top[-1] = nullptr;                                 // fake return address of fiber_entry
    top[-2] = (void*)&fiber_entry;                     // popped by the `ret` of the first switch: rsp = top - 8 at entry
    for (int i = 3; i <= 8; ++i) top[-i] = nullptr;    // rbp rbx r12 r13 r14 r15
    fb.sp = (void*)(top - 8);
  }
  unsigned live = nthr;
  while (live) {
    const uint64_t before = s->events;
    for (unsigned t = 0; t < nthr; ++t) {
      Fiber& fb = s->fib[t];
      if (fb.done) continue;
      if (fb.wait_gen) {
        if (*fb.wait_gen == fb.wait_val) continue;
        fb.wait_gen = nullptr;
      }
      s->cur = t;
      threadIdx = {t, 0, 0};
      emu_switch(&s->main_sp, fb.sp);
      if (fb.done) --live;
    }
    const bool progress = s->events != before;
    if (!progress) {
      // Every live work-item is parked at a rendezvous that cannot complete: a kernel bug (a barrier or a whole-wave operation
      // under divergent control flow) or a construct this emulation cannot express.  The fibers are abandoned and the launch is
      // reported as failed.
      std::fprintf(stderr, "hipemu: block (%u,%u,%u): all live work-items are parked at a barrier — launch abandoned\n", blockIdx.x,
                   blockIdx.y, blockIdx.z);
      g_emu_error = 719;   // hipErrorLaunchFailure
      return;
    }
  }
}
#endif

template <typename K, typename... A> void launch(K kernel, dim3 grid, dim3 block, A... args) {
  const unsigned nthr = block.x * block.y * block.z;
  if (block.y != 1 || block.z != 1 || nthr % 64) { std::fprintf(stderr, "hipemu: unsupported block shape %u x %u x %u\n", block.x, block.y, block.z); std::abort(); }
  std::lock_guard<std::mutex> one_launch_at_a_time(launch_mutex());
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        Block blk;
        blk.bar = std::make_unique<Barrier>(nthr);
        for (unsigned w = 0; w < nthr / 64; ++w) blk.waves.push_back(std::make_unique<Wave>());
        g_block = &blk;
        run_block(nthr, [=](unsigned t) {
          threadIdx = {t, 0, 0};
          blockIdx = {bx, by, bz};
          blockDim = block;
          gridDim = grid;
          kernel(args...);
        });
        g_block = nullptr;
        if (g_emu_error) return;
      }
}
}  // namespace hipemu
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hipemu::launch(kernel, grid, block, __VA_ARGS__)
