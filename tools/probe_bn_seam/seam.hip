// What a BatchNorm statistics seam costs on a chain of short kernels, and whether atomics remove it (round-3 verdict item 4a).
// A "layer" = producer (per-block partial sums of a [pixels][C] fp32 image, 32 pixels per block as enet_mconv's epilogue has them) ->
// statistics (mean, rstd per channel) -> consumer (normalises the image).  Variants of the middle step:
//   rows      the tree's form: partial rows [block][C][2] doubles, folded by a one-block launch of its own (a node on the chain)
//   f64       producer blocks add their sums with global_atomic_add_f64 (order-dependent in the last bits), last block (ticket) finishes
//   limbs     the same with exact, order-independent integer sums: the block sum in 2^-40 fixed point, split into three 32-bit limbs, each
//             accumulated in an int64 of its own (carry-save: no returning atomic)
//   *_rep8    eight replica rows (block b adds into replica b % 8): an eighth of the same-address contention, the last block folds 8 rows
// Build: hipcc --offload-arch=gfx950 -O3 -o seam seam.hip ; run: ./seam   (prints us per layer inside a captured graph of 100 layers)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum Mode { ROWS = 0, F64 = 1, LIMBS = 2, F64_REP8 = 3, LIMBS_REP8 = 4 };
constexpr int PX = 32;           // pixels per producer block
constexpr double FIX = 1099511627776.0;     // 2^40

struct Args {
  const float* x; float* y; double* rows; double* fsum; long long* isum; unsigned* ticket; float2* stats;
  int C, blocks; float eps;
};

__device__ inline void block_sums(const Args& a, int c, double& s1, double& s2) {
  const float* xp = a.x + (size_t)blockIdx.x * PX * a.C + c;
  float v[PX];
#pragma unroll
  for (int i = 0; i < PX; ++i) v[i] = xp[(size_t)i * a.C];
  s1 = 0; s2 = 0;
#pragma unroll
  for (int i = 0; i < PX; ++i) { s1 += v[i]; s2 += (double)v[i] * v[i]; }
}

__device__ inline void finish(const Args& a, int c, double s1, double s2) {
  const double n = (double)a.blocks * PX;
  const double m = s1 / n, var = s2 / n - m * m;
  a.stats[c] = make_float2((float)m, (float)(1.0 / sqrt(var + a.eps)));
}

template <int MODE>
__global__ void producer(Args a) {
  const int c = threadIdx.x;
  double s1 = 0, s2 = 0;
  if (c < a.C) block_sums(a, c, s1, s2);
  if (MODE == ROWS) {
    if (c < a.C) { double* r = a.rows + ((size_t)blockIdx.x * a.C + c) * 2; r[0] = s1; r[1] = s2; }
    return;
  }
  constexpr int REP = (MODE == F64_REP8 || MODE == LIMBS_REP8) ? 8 : 1;
  const int rep = blockIdx.x % REP;
  if (c < a.C) {
    if (MODE == F64 || MODE == F64_REP8) {
      double* f = a.fsum + ((size_t)rep * a.C + c) * 2;
      unsafeAtomicAdd(f, s1); unsafeAtomicAdd(f + 1, s2);
    } else {
      long long* q = a.isum + ((size_t)rep * a.C + c) * 6;
      const double d[2] = {s1, s2};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const __int128 v = (__int128)llrint(d[k] * FIX);     // |sum| * 2^40 < 2^63 for the probe's data; the real kernel would split a double-double
        atomicAdd((unsigned long long*)(q + 3 * k), (unsigned long long)((unsigned long long)v & 0xffffffffull));
        atomicAdd((unsigned long long*)(q + 3 * k + 1), (unsigned long long)(((unsigned long long)v >> 32) & 0xffffffffull));
        atomicAdd((unsigned long long*)(q + 3 * k + 2), (unsigned long long)(long long)(v >> 64));
      }
    }
  }
  // every add of this block is at the L2 before the ticket moves
  __threadfence();
  __syncthreads();
  __shared__ unsigned last;
  if (threadIdx.x == 0) last = atomicAdd(a.ticket, 1u) == (unsigned)a.blocks - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (c < a.C) {
    double t1 = 0, t2 = 0;
    for (int r = 0; r < REP; ++r) {
      if (MODE == F64 || MODE == F64_REP8) {
        double* f = a.fsum + ((size_t)r * a.C + c) * 2;
        t1 += __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t2 += __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(f, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(f + 1, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        long long* q = a.isum + ((size_t)r * a.C + c) * 6;
        double t[2];
        for (int k = 0; k < 2; ++k) {
          __int128 v = 0;
          for (int l = 2; l >= 0; --l) {
            const long long limb = __hip_atomic_load(q + 3 * k + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = (v << 32) + (__int128)limb;
            __hip_atomic_store(q + 3 * k + l, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          t[k] = (double)v / FIX;
        }
        t1 += t[0]; t2 += t[1];
      }
    }
    finish(a, c, t1, t2);
  }
  if (threadIdx.x == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void fold_rows(Args a) {      // one block: the tree's finalize (eight partial rows per round trip)
  const int c = threadIdx.x;
  if (c >= a.C) return;
  double s1 = 0, s2 = 0;
  int b = 0;
  for (; b + 8 <= a.blocks; b += 8) {
    double u[8], w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const double* r = a.rows + ((size_t)(b + i) * a.C + c) * 2; u[i] = r[0]; w[i] = r[1]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1 += u[i]; s2 += w[i]; }
  }
  for (; b < a.blocks; ++b) { const double* r = a.rows + ((size_t)b * a.C + c) * 2; s1 += r[0]; s2 += r[1]; }
  finish(a, c, s1, s2);
}

__global__ void consumer(Args a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)a.blocks * PX * a.C;
  if (i >= n) return;
  const float2 st = a.stats[i % a.C];
  a.y[i] = (a.x[i] - st.x) * st.y;
}

template <int MODE>
static void layer(const Args& a, hipStream_t st) {
  hipLaunchKernelGGL(producer<MODE>, dim3(a.blocks), dim3(128), 0, st, a);
  if (MODE == ROWS) hipLaunchKernelGGL(fold_rows, dim3(1), dim3(128), 0, st, a);
  const size_t n = (size_t)a.blocks * PX * a.C;
  hipLaunchKernelGGL(consumer, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}

template <int MODE>
static double run(Args a, hipStream_t st, std::vector<float2>& out, bool with_stats) {
  constexpr int LAYERS = 100;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int l = 0; l < LAYERS; ++l) layer<MODE>(a, st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  if (with_stats) { out.resize(a.C); CK(hipMemcpy(out.data(), a.stats, sizeof(float2) * a.C, hipMemcpyDeviceToHost)); }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ms * 1e3 / (reps * LAYERS);
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int Cs[] = {32, 128}, Bs[] = {157, 625, 2500};
  printf("us per layer (producer -> statistics -> consumer), 100 layers per graph launch; max |d mean|, |d rstd| rel. to the rows form\n");
  printf("%4s %6s | %8s %8s %8s %8s %8s | %s\n", "C", "blocks", "rows", "f64", "limbs", "f64_rep8", "limb_rep8", "deviation f64 / limbs");
  for (int C : Cs) for (int B : Bs) {
    const size_t n = (size_t)B * PX * C;
    std::vector<float> h(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) % 2001 - 1000) * 1e-3f * (1 + (i % C) * 0.1f); }
    Args a{};
    a.C = C; a.blocks = B; a.eps = 1e-3f;
    float* x; CK(hipMalloc(&x, n * 4)); CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice)); a.x = x;
    CK(hipMalloc(&a.y, n * 4));
    CK(hipMalloc(&a.rows, (size_t)B * C * 16));
    CK(hipMalloc(&a.fsum, 8 * C * 16)); CK(hipMemset(a.fsum, 0, 8 * C * 16));
    CK(hipMalloc(&a.isum, 8 * C * 48)); CK(hipMemset(a.isum, 0, 8 * C * 48));
    CK(hipMalloc(&a.ticket, 4)); CK(hipMemset(a.ticket, 0, 4));
    CK(hipMalloc(&a.stats, sizeof(float2) * C));
    std::vector<float2> r0, r1, r2, r3, r4;
    double t[5];
    for (int round = 0; round < 2; ++round) {
      t[0] = run<ROWS>(a, st, r0, true);
      t[1] = run<F64>(a, st, r1, true);
      t[2] = run<LIMBS>(a, st, r2, true);
      t[3] = run<F64_REP8>(a, st, r3, true);
      t[4] = run<LIMBS_REP8>(a, st, r4, true);
    }
    double d1 = 0, d2 = 0;
    for (int c = 0; c < C; ++c) {
      d1 = fmax(d1, fmax(fabs(r1[c].x - r0[c].x) / (fabs(r0[c].x) + 1e-6), fabs(r1[c].y - r0[c].y) / fabs(r0[c].y)));
      d2 = fmax(d2, fmax(fabs(r2[c].x - r0[c].x) / (fabs(r0[c].x) + 1e-6), fabs(r2[c].y - r0[c].y) / fabs(r0[c].y)));
      d2 = fmax(d2, fmax(fabs(r4[c].x - r0[c].x) / (fabs(r0[c].x) + 1e-6), fabs(r4[c].y - r0[c].y) / fabs(r0[c].y)));
    }
    printf("%4d %6d | %8.2f %8.2f %8.2f %8.2f %8.2f | %.1e / %.1e\n", C, B, t[0], t[1], t[2], t[3], t[4], d1, d2);
    fflush(stdout);
    CK(hipFree(x)); CK(hipFree(a.y)); CK(hipFree(a.rows)); CK(hipFree(a.fsum)); CK(hipFree(a.isum)); CK(hipFree(a.ticket)); CK(hipFree(a.stats));
  }
  return 0;
}
