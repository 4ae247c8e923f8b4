#include <hip/hip_runtime.h>
struct PtrPack { const float* in[4]; float* out[4]; };
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ void block_partial2(float a, float b, float* partial) {
  __shared__ float sm[8];
  a = wave_sum(a); b = wave_sum(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sm[w] = a; sm[4 + w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) { partial[blockIdx.x * 2 + 0] = sm[0] + sm[1] + sm[2] + sm[3]; partial[blockIdx.x * 2 + 1] = sm[4] + sm[5] + sm[6] + sm[7]; }
}
template <int C> __device__ __forceinline__ void load_px(const float* p, long long pix, float v[C]) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p + pix * 4);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <int C> __device__ __forceinline__ float softmax_px(const float x[C], float p[C]) {
  float m = x[0];
#pragma unroll
  for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { p[c] = expf(x[c] - m); s += p[c]; }
  const float inv = 1.f / s;
#pragma unroll
  for (int c = 0; c < C; ++c) p[c] *= inv;
  return m + logf(s);
}
template <int C> __device__ __forceinline__ float entropy_px(const float p[C]) {
  float e = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) e += p[c] * logf(p[c] + 1e-16f);
  return -e;
}
template <int C> __device__ __forceinline__ float entropy_fast(const float p[C]) {     // __logf: v_log_f32 * ln2, no refinement
  float e = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) e += p[c] * __logf(p[c] + 1e-16f);
  return -e;
}
template <int C> __device__ __forceinline__ float entropy_nopk(const float p[C]) {     // logf, but every product behind an asm barrier (no v_pk_mul_f32)
  float e = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { float l = logf(p[c] + 1e-16f); asm volatile("" : "+v"(l)); float t = p[c] * l; asm volatile("" : "+v"(t)); e += t; }
  return -e;
}
template <int C> __device__ __forceinline__ float logsum(const float p[C]) {           // logs only, no products
  float e = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) e += logf(p[c] + 1e-16f);
  return -e;
}
// MODE 0: as shipped.  1: per-thread results written out (no block reduction).  2: no entropy (sum of the mean's first channel).
template <int MODE>
__global__ __launch_bounds__(256) void jsd_var(PtrPack pk, int S, long long P, float* partial, float* perthread) {
  constexpr int C = 4;
  float sum = 0.f;
  const float invS = 1.f / (float)S;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    float mean[C];
#pragma unroll
    for (int c = 0; c < C; ++c) mean[c] = 0.f;
    float hsum = 0.f;
    for (int s = 0; s < S; ++s) {
      float x[C], p[C];
      load_px<C>(pk.in[s], pix, x);
      softmax_px<C>(x, p);
#pragma unroll
      for (int c = 0; c < C; ++c) mean[c] += p[c];
      if (MODE == 3) hsum += entropy_fast<C>(p); else if (MODE == 4) hsum += entropy_nopk<C>(p); else if (MODE == 5) hsum += logsum<C>(p); else if (MODE != 2) hsum += entropy_px<C>(p);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) mean[c] *= invS;
    const float j = MODE == 2 ? mean[0] : MODE == 3 ? entropy_fast<C>(mean) - hsum * invS : MODE == 4 ? entropy_nopk<C>(mean) - hsum * invS : MODE == 5 ? logsum<C>(mean) - hsum * invS : entropy_px<C>(mean) - hsum * invS;
    sum += j;
  }
  if (MODE >= 1) perthread[blockIdx.x * 256 + threadIdx.x] = sum;
  else block_partial2(sum, 0.f, partial);
}
extern "C" int run_var(int mode, const float* a, const float* b, long long P, float* partial, float* perthread, hipStream_t st) {
  PtrPack pk; for (int i = 0; i < 4; ++i) { pk.in[i] = nullptr; pk.out[i] = nullptr; }
  pk.in[0] = a; pk.in[1] = b;
  if (mode == 0) hipLaunchKernelGGL(jsd_var<0>, dim3(1024), dim3(256), 0, st, pk, 2, P, partial, perthread);
  else if (mode == 1) hipLaunchKernelGGL(jsd_var<1>, dim3(1024), dim3(256), 0, st, pk, 2, P, partial, perthread);
  else if (mode == 2) hipLaunchKernelGGL(jsd_var<2>, dim3(1024), dim3(256), 0, st, pk, 2, P, partial, perthread);
  else if (mode == 3) hipLaunchKernelGGL(jsd_var<3>, dim3(1024), dim3(256), 0, st, pk, 2, P, partial, perthread);
  else if (mode == 4) hipLaunchKernelGGL(jsd_var<4>, dim3(1024), dim3(256), 0, st, pk, 2, P, partial, perthread);
  else hipLaunchKernelGGL(jsd_var<5>, dim3(1024), dim3(256), 0, st, pk, 2, P, partial, perthread);
  return (int)hipGetLastError();
}

// Variant 6: what a pre-multiplied sum of two gradient buffers does (RCCL's AVG = PreMulSum): out = a * s + b * s over float2 pairs,
// which the default build turns into v_pk_mul_f32 + v_pk_add_f32 / v_pk_fma_f32 and the flag build into scalar multiplies and adds.
typedef __attribute__((ext_vector_type(2))) float f32x2;
// Variants 7 / 8 narrow it down: 7 = packed products whose operands come straight from the transcendental unit (v_log_f32 ->
// v_pk_mul_f32, the shape of p * log p), 8 = packed adds that read a swapped pair (op_sel forms, as the compiler emits for the
// channel means), 9 = v_pk_fma_f32 with op_sel_hi:[0,1,1] written as inline asm (in both builds); variant 6 has neither.
template <int KIND>
__global__ __launch_bounds__(256) void premul_sum(const float* a, const float* b, float s, long long n4, float* out) {
  const f32x2 s2 = {s, s};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 x = reinterpret_cast<const f32x4*>(a)[i], y = reinterpret_cast<const f32x4*>(b)[i];
    f32x2 x0 = {x[0], x[1]}, x1 = {x[2], x[3]}, y0 = {y[0], y[1]}, y1 = {y[2], y[3]};
    f32x2 r0, r1;
    if (KIND == 0) {
      x0 = x0 * s2; x1 = x1 * s2; y0 = y0 * s2; y1 = y1 * s2;
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(y0), "+v"(y1));      // (the products stay products: no contraction into the sum)
      r0 = x0 + y0; r1 = x1 + y1;
    } else if (KIND == 1) {
      const f32x2 l0 = {__log2f(x[0] * x[0] + 1.f), __log2f(x[1] * x[1] + 1.f)}, l1 = {__log2f(x[2] * x[2] + 1.f), __log2f(x[3] * x[3] + 1.f)};
      r0 = l0 * y0; r1 = l1 * y1;
    } else if (KIND == 2) {
      const f32x2 ys0 = {y0[1], y0[0]}, ys1 = {y1[1], y1[0]};
      r0 = x0 * s2 + ys0; r1 = x1 * s2 + ys1;
    } else {      // the exact instruction form of RCCL's pre-multiplied sum (librccl.so, runRing<float, FuncPreMulSum<float>, ...>): low half of src0 for both lanes
#ifdef VICTIM_NO_PK_ASM                      // (the flag build's assembler refuses the mnemonic)
      r0 = x0 * s2 + y0; r1 = x1 * s2 + y1;
#else
      f32x2 sv = {s, 12345.f};
      asm volatile("v_pk_fma_f32 %0, %2, %3, %4 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %2, %5, %6 op_sel_hi:[0,1,1]"
                   : "=&v"(r0), "=&v"(r1) : "v"(sv), "v"(x0), "v"(y0), "v"(x1), "v"(y1));
#endif
    }
    const f32x4 r = {r0[0], r0[1], r1[0], r1[1]};
    reinterpret_cast<f32x4*>(out)[i] = r;
  }
}
extern "C" int run_premul(int kind, const float* a, const float* b, float s, long long n4, float* out, hipStream_t st) {
  if (kind == 0) hipLaunchKernelGGL(premul_sum<0>, dim3(1024), dim3(256), 0, st, a, b, s, n4, out);
  else if (kind == 1) hipLaunchKernelGGL(premul_sum<1>, dim3(1024), dim3(256), 0, st, a, b, s, n4, out);
  else if (kind == 2) hipLaunchKernelGGL(premul_sum<2>, dim3(1024), dim3(256), 0, st, a, b, s, n4, out);
  else hipLaunchKernelGGL(premul_sum<3>, dim3(1024), dim3(256), 0, st, a, b, s, n4, out);
  return (int)hipGetLastError();
}
