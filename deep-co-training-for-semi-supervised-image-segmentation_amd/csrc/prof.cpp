// Optional per-kernel-class timing (bench.py's roofline leg) + version/status strings.
// The only global state in the library: an opt-in event log guarded by a mutex.
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include "dct_common.h"

int g_dct_prof_on = 0;
int g_dct_skip_families = 0;      // diagnostic: dct_common.h DCT_LAUNCH_FAM

// Diagnostic: which kernel / grid the planner chose for the last dct_conv2d / dct_conv2d_wgrad call of this thread
// (tools/bench_conv.py --plan prints it per layer).  Not part of the ABI contract.
thread_local char g_dct_last_plan[200] = "";
extern "C" const char* dct_debug_last_plan() { return g_dct_last_plan; }

namespace {
struct Rec { int cls; hipEvent_t a, b; };
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
thread_local hipEvent_t t_pending = nullptr;

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void dct_prof_begin(int, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  t_pending = get_event();
  (void)hipEventRecord(t_pending, s);
}
void dct_prof_end(int cls, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r;
  r.cls = cls; r.a = t_pending; r.b = get_event();
  (void)hipEventRecord(r.b, s);
  g_recs.push_back(r);
  t_pending = nullptr;
}

extern "C" int dct_prof_enable(int on) { g_dct_prof_on = on ? 1 : 0; return DCT_OK; }

extern "C" int dct_prof_read(double* ms_per_class, int64_t* launches_per_class, int reset) {
  if (!ms_per_class || !launches_per_class) return DCT_ERR_BAD_ARG;
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < DCT_PROF_NCLASS; ++i) { ms_per_class[i] = 0.0; launches_per_class[i] = 0; }
  for (const Rec& r : g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.cls >= 0 && r.cls < DCT_PROF_NCLASS) {
      ms_per_class[r.cls] += ms;
      launches_per_class[r.cls] += 1;
    }
  }
  if (reset) {
    for (const Rec& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
  }
  return DCT_OK;
}

// Shader-clock probe (bench.py: the clock the chip holds WHILE the captured step runs): one wave reads the shader-clock counter
// (s_memtime) and the constant 100 MHz reference counter (s_memrealtime) when it starts, sleeps in short naps until `ref_ticks`
// reference ticks have passed, and reads both again: out[0] = shader cycles, out[1] = reference ticks -> GHz = 0.1 * out[0] / out[1].
// One wave that is asleep almost all of the time: it takes one wave slot of one SIMD and a few issue cycles per microsecond.
__global__ void dct_clock_probe_kernel(unsigned long long* out, unsigned long long ref_ticks) {
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0) :: "memory");
  do {
    __builtin_amdgcn_s_sleep(32);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1) :: "memory");
  } while (r1 - r0 < ref_ticks);
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
extern "C" int dct_clock_probe(unsigned long long* out2, unsigned long long ref_ticks, dct_stream stream) {
  if (!out2 || ref_ticks == 0 || ref_ticks > 100000000ull) return DCT_ERR_BAD_ARG;      // at most one second
  hipLaunchKernelGGL(dct_clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out2, ref_ticks);
  return hipGetLastError() == hipSuccess ? DCT_OK : DCT_ERR_LAUNCH;
}

// Phase stamp (tools/phase_stamps.py): a one-thread kernel that leaves the 100 MHz reference counter at `slot` -- queued on a model's
// stream between the phases of the step (it is an ordinary kernel node: it replays with the captured graph), it dates the phases of the
// REAL schedule, which a tracing profiler serialises.
// slot[0] counts the stamps of this site, slot[1 + n % ring] holds the n-th: a replayed graph (fixed addresses) leaves a timeline of its last
// `ring` replays, so the steps can be dated while they are PIPELINED (no host synchronisation between replays).
__global__ void dct_stamp_kernel(unsigned long long* slot, unsigned ring) {
  unsigned long long r;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) :: "memory");
  const unsigned long long n = slot[0];
  slot[1 + n % ring] = r;
  slot[0] = n + 1;
}
extern "C" int dct_stamp(unsigned long long* slot, unsigned ring, dct_stream stream) {
  if (!slot || ring < 1) return DCT_ERR_BAD_ARG;
  hipLaunchKernelGGL(dct_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, slot, ring);
  return hipGetLastError() == hipSuccess ? DCT_OK : DCT_ERR_LAUNCH;
}

extern "C" int dct_version(void) { return 100; }

extern "C" const char* dct_status_string(int status) {
  switch (status) {
    case DCT_OK: return "ok";
    case DCT_ERR_BAD_ARG: return "bad argument";
    case DCT_ERR_UNSUPPORTED: return "unsupported shape/dtype/alignment";
    case DCT_ERR_LAUNCH: return "kernel launch failed";
    case DCT_ERR_WORKSPACE: return "workspace missing or too small";
    default: return "unknown status";
  }
}
