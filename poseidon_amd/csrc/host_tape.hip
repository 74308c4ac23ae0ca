// host_tape — replaying a recorded step from C.
//
// engine.py records a training step as a list of C-ABI calls with their final arguments (device addresses, sizes, stream handles) and
// replays the list for every later step.  Replayed from Python that is ~900 ctypes calls of 15-40 arguments each: 7 ms of host time per
// 19 ms step (round 3), a third of a core per rank, and the wall as soon as the GPU side gets 2.7x faster.  scot_tape_replay walks the
// same list inside the library: one ctypes call per run of consecutive launches.  The host-side operations BETWEEN launches that a step
// needs (event record / stream wait for the weight-gradient stream, the few memsets and device-to-device copies) are entry points of
// this file too, so they are ordinary tape entries and a whole forward or backward is normally ONE run.
#include "common.h"
#include <string.h>

extern "C" int scot_memset_async(void* p, int byte, size_t n, hipStream_t s) {
  if (!p && n) return SCOT_ERR_SHAPE;
#ifdef SCOT_HIPEMU
  memset(p, byte, n);
  return SCOT_OK;
#else
  return hipMemsetAsync(p, byte, n, s) == hipSuccess ? SCOT_OK : SCOT_ERR_LAUNCH;
#endif
}
extern "C" int scot_memcpy_async(void* dst, const void* src, size_t n, hipStream_t s) {
  if ((!dst || !src) && n) return SCOT_ERR_SHAPE;
#ifdef SCOT_HIPEMU
  memmove(dst, src, n);
  return SCOT_OK;
#else
  return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s) == hipSuccess ? SCOT_OK : SCOT_ERR_LAUNCH;
#endif
}
// event: a hipEvent_t (torch.cuda.Event.cuda_event)
extern "C" int scot_event_record(void* event, hipStream_t s) {
#ifdef SCOT_HIPEMU
  return SCOT_OK;
#else
  return hipEventRecord((hipEvent_t)event, s) == hipSuccess ? SCOT_OK : SCOT_ERR_LAUNCH;
#endif
}
extern "C" int scot_stream_wait_event(hipStream_t s, void* event) {
#ifdef SCOT_HIPEMU
  return SCOT_OK;
#else
  return hipStreamWaitEvent(s, (hipEvent_t)event, 0) == hipSuccess ? SCOT_OK : SCOT_ERR_LAUNCH;
#endif
}

// One generic call of a C-ABI entry point (x86-64 System V): the integer-class arguments (pointers, int, size_t) go to rdi, rsi, rdx,
// rcx, r8, r9 and then to the stack in order; float arguments to xmm0-7 (an entry point has at most 8 — asserted by the encoder).  A
// callee ignores stack words beyond its own parameters, so ONE prototype with 6 + 8 register arguments and SCOT_TAPE_MAX_STACK stack
// words serves every entry point; a `float` parameter reads the low 32 bits of its xmm register, which is where the raw bits are put.
#define SCOT_TAPE_MAX_INT 48
#define SCOT_TAPE_MAX_STACK (SCOT_TAPE_MAX_INT - 6)
typedef uint64_t W;
typedef int (*tape_fn_t)(W, W, W, W, W, W, double, double, double, double, double, double, double, double,
                         W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W, W);

static inline double bits_to_xmm(W u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}

// prog: n_entries records {fn, n_int, n_flt, int words..., float words (raw IEEE-754 single bits in the low half)...}.  Returns 0, or the
// first non-zero status with *fail_entry = index of the entry that returned it (the rest of the program is not issued).
extern "C" int scot_tape_replay(const uint64_t* prog, size_t n_words, int* fail_entry) {
#if !defined(__x86_64__) || defined(_WIN32)
  // the one-prototype call below IS the System V x86-64 register / stack convention; anywhere else the engine replays its Python list
  (void)prog; (void)n_words; (void)fail_entry;
  return SCOT_ERR_UNSUPPORTED;
#else
  size_t i = 0;
  int entry = 0;
  while (i < n_words) {
    if (i + 3 > n_words) return SCOT_ERR_SHAPE;
    const tape_fn_t fn = (tape_fn_t)(uintptr_t)prog[i];
    const W ni = prog[i + 1], nf = prog[i + 2];
    if (!fn || ni > SCOT_TAPE_MAX_INT || nf > 8 || i + 3 + ni + nf > n_words) return SCOT_ERR_SHAPE;
    W a[SCOT_TAPE_MAX_INT];
    W f[8];
    for (W k = 0; k < SCOT_TAPE_MAX_INT; ++k) a[k] = k < ni ? prog[i + 3 + k] : 0;
    for (W k = 0; k < 8; ++k) f[k] = k < nf ? prog[i + 3 + ni + k] : 0;
    const int rc = fn(a[0], a[1], a[2], a[3], a[4], a[5], bits_to_xmm(f[0]), bits_to_xmm(f[1]), bits_to_xmm(f[2]), bits_to_xmm(f[3]),
                      bits_to_xmm(f[4]), bits_to_xmm(f[5]), bits_to_xmm(f[6]), bits_to_xmm(f[7]),
                      a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18], a[19], a[20], a[21], a[22], a[23],
                      a[24], a[25], a[26], a[27], a[28], a[29], a[30], a[31], a[32], a[33], a[34], a[35], a[36], a[37], a[38], a[39], a[40],
                      a[41], a[42], a[43], a[44], a[45], a[46], a[47]);
    if (rc != 0) {
      if (fail_entry) *fail_entry = entry;
      return rc;
    }
    i += 3 + ni + nf;
    ++entry;
  }
  return SCOT_OK;
#endif
}
