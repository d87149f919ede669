// replay.hip -- the step's launch sequence as a prebuilt call list walked in C (unit_replay), plus the raw event primitives the
// list needs to reproduce the step's stream forks and joins.
//
// Why: one S1 training step is ~650 C-ABI calls (kernel launches + event record / wait pairs) issued from Python through ctypes;
// the interpreter, the tile-policy lookups and the argument marshalling cost the host 9-14 ms per step against ~15 ms of device
// time (BENCH_r05 host_enqueue_ms_per_step), and eight ranks share one node's cores. The step is already free of host
// syncs and data-dependent host branches (it can be captured into a hipGraph), so its call sequence for a given batch key is a
// constant: unit_amd/_lib.py records every enqueueing call of one eagerly executed step -- function address + integer-class
// arguments + float arguments, exactly what ctypes handed over -- and this file re-issues the list: the SAME launches on the
// SAME in-order streams as the eager schedule (a hipGraph replay orders its nodes with barrier packets and measured 0.5-0.9 ms
// slower per step on the device), at the cost of the runtime's launch call alone.
//
// The generic call: under the x86-64 SysV ABI integer-class arguments (int, long, size_t, pointers) take rdi, rsi, rdx, rcx, r8, r9
// and then the stack in declaration order, float arguments take xmm0-7 in declaration order, independently of how the two classes
// interleave (as long as no more than eight floats are passed: the C ABI's maximum is four). A function of ANY prototype in
// include/unit_hip.h can therefore be called through one pointer type taking 32 integer-class and 8 float arguments: surplus
// register and stack arguments are ignored by the callee (caller cleans the stack). The host of this framework is x86-64 Linux (the
// MI355X boxes); on another ABI unit_replay refuses.
#include "common.h"

#define UNIT_CALL_INTS 32
#define UNIT_CALL_FLOATS 8

struct UnitCall {
  void* fn;
  int n_int, n_flt;
  long long i[UNIT_CALL_INTS];
  float f[UNIT_CALL_FLOATS];
};

extern "C" size_t unit_call_bytes(void) { return sizeof(UnitCall); }

typedef int (*unit_fn8_t)(long long, long long, long long, long long, long long, long long, long long, long long, float, float, float, float,
                          float, float, float, float);
typedef int (*unit_fn16_t)(long long, long long, long long, long long, long long, long long, long long, long long, long long, long long, long long,
                           long long, long long, long long, long long, long long, float, float, float, float, float, float, float, float);
typedef int (*unit_fn32_t)(long long, long long, long long, long long, long long, long long, long long, long long, long long, long long, long long,
                           long long, long long, long long, long long, long long, long long, long long, long long, long long, long long, long long,
                           long long, long long, long long, long long, long long, long long, long long, long long, long long, long long, float, float,
                           float, float, float, float, float, float);

// calls[0 .. n): each must return UNIT_OK; on the first failure its index goes to *failed (the library's last-error text is the
// callee's) and the walk stops -- what was enqueued before stays enqueued, as in the eager step.
extern "C" int unit_replay(const void* calls_, int n, int* failed) {
#if !defined(__x86_64__) || defined(_WIN32)
  (void)calls_; (void)n; (void)failed;
  unit_set_error("unit_replay: the generic call assumes the x86-64 SysV calling convention");
  return UNIT_ERR_UNSUPPORTED;
#else
  const UnitCall* calls = (const UnitCall*)calls_;
  for (int k = 0; k < n; ++k) {
    const UnitCall& c = calls[k];
    const long long* a = c.i;
    const float* f = c.f;
    int st;
    if (c.n_int <= 8)
      st = ((unit_fn8_t)c.fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
    else if (c.n_int <= 16)
      st = ((unit_fn16_t)c.fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], f[0], f[1],
                               f[2], f[3], f[4], f[5], f[6], f[7]);
    else
      st = ((unit_fn32_t)c.fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17],
                               a[18], a[19], a[20], a[21], a[22], a[23], a[24], a[25], a[26], a[27], a[28], a[29], a[30], a[31], f[0], f[1], f[2],
                               f[3], f[4], f[5], f[6], f[7]);
    if (st != UNIT_OK) {
      if (failed) *failed = k;
      return st;
    }
  }
  if (failed) *failed = -1;
  return UNIT_OK;
#endif
}

// The step's plan forks and joins its HIP streams through torch.cuda.Event / Stream.wait_stream; while a call list is being
// recorded _lib.py routes those two operations here (same hipEvent_t handle torch owns), so that the list reproduces the edges.
extern "C" int unit_event_record_raw(void* event, void* stream) {
  UNIT_CHECK_ARG(event != nullptr, "unit_event_record_raw: null event");
  hipError_t e = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
  if (e != hipSuccess) { unit_set_error(hipGetErrorString(e)); return UNIT_ERR_LAUNCH; }
  return UNIT_OK;
}

extern "C" int unit_stream_wait_event_raw(void* stream, void* event) {
  UNIT_CHECK_ARG(event != nullptr, "unit_stream_wait_event_raw: null event");
  hipError_t e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
  if (e != hipSuccess) { unit_set_error(hipGetErrorString(e)); return UNIT_ERR_LAUNCH; }
  return UNIT_OK;
}

// self-test target of tests/test_replay_cpu.py (no GPU call): checks that the generic call delivers 29 integer-class and 4 float
// arguments in the right places when the two classes interleave and the integers overflow onto the stack.
extern "C" int unit_replay_selftest(int a0, const void* p1, float f0, long a2, int a3, float f1, int a4, int a5, size_t a6, int a7, int a8, float f2,
                                    int a9, int a10, int a11, int a12, int a13, int a14, int a15, int a16, int a17, int a18, int a19, int a20,
                                    int a21, int a22, int a23, int a24, int a25, int a26, float f3, int a27, long long* out) {
  UNIT_CHECK_ARG(out != nullptr, "unit_replay_selftest: null out");
  long long s = 0;
  const long long v[28] = {a0, (long long)(size_t)p1, a2, a3, a4, a5, (long long)a6, a7, a8, a9, a10, a11, a12, a13, a14,
                           a15, a16, a17, a18, a19, a20, a21, a22, a23, a24, a25, a26, a27};
  for (int i = 0; i < 28; ++i) s += v[i] * (long long)(i + 1);
  out[0] = s;
  out[1] = (long long)(f0 * 1000.0f) + 10 * (long long)(f1 * 1000.0f) + 100 * (long long)(f2 * 1000.0f) + 1000 * (long long)(f3 * 1000.0f);
  return UNIT_OK;
}
