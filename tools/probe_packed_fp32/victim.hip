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

// Synthetic neighbours (which ingredient of the two conv kernels is it?): 0 = LDS reads at full rate (ds_read_b128, no MFMA),
// 1 = v_mfma_f32_16x16x32_bf16 back to back from registers (no LDS), 2 = v_mfma_f32_32x32x16_bf16 likewise, 3 = both: MFMAs whose
// operands are re-read from LDS every step (the shape of a tile loop), 4 = LDS-DMA (global_load_lds_dwordx4) streaming.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int KIND>
__global__ __launch_bounds__(512) void neighbour(const float* src, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];                       // 64 KiB: two blocks per CU
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = src[(blockIdx.x * 16384 + i) & 0xFFFFF];
  __syncthreads();
  f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
  f32x16 acc16;
  for (int k = 0; k < 16; ++k) acc16[k] = 0.f;
  const int lane16 = (threadIdx.x * 4) & 16383;
  if (KIND == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((lane16 + (it * 8 + u) * 2048) & 16380)); acc4 += v; }
    }
  } else if (KIND == 1 || KIND == 2 || KIND == 3) {
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)lds[(lane16 + k) & 16383]; b[k] = (__bf16)lds[(lane16 + 8 + k) & 16383]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (KIND == 3) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((lane16 + (it * 8 + u) * 2048) & 16380));
          a = __builtin_bit_cast(bf16x8, v);
        }
        if (KIND == 2) acc16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc16, 0, 0, 0);
        else acc4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4, 0, 0, 0);
      }
    }
  } else {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (((long long)blockIdx.x * 8192 + (it * 4 + u) * 2048 + threadIdx.x * 4) & 0xFFFFC)),
                                         (lptr_t)(lds + wave * 256 + u * 4096), 16, 0, 0);
      __builtin_amdgcn_s_waitcnt(0);
    }
    __syncthreads();
    acc4[0] = lds[threadIdx.x];
  }
  float r = acc4[0] + acc4[1] + acc4[2] + acc4[3];
  for (int k = 0; k < 16; ++k) r += acc16[k];
  if (r == 12345.678f) sink[threadIdx.x] = r;                                     // (keeps the loop alive, never true in practice)
}
extern "C" int run_neighbour(int kind, const float* src, float* sink, int iters, hipStream_t st) {
  if (kind == 0) hipLaunchKernelGGL(neighbour<0>, dim3(2048), dim3(512), 0, st, src, sink, iters);
  else if (kind == 1) hipLaunchKernelGGL(neighbour<1>, dim3(2048), dim3(512), 0, st, src, sink, iters);
  else if (kind == 2) hipLaunchKernelGGL(neighbour<2>, dim3(2048), dim3(512), 0, st, src, sink, iters);
  else if (kind == 3) hipLaunchKernelGGL(neighbour<3>, dim3(2048), dim3(512), 0, st, src, sink, iters);
  else hipLaunchKernelGGL(neighbour<4>, dim3(2048), dim3(512), 0, st, src, sink, iters);
  return (int)hipGetLastError();
}

// Other operand-routing instruction classes the library's kernels use, as streaming victims (run_class): 0 = DPP (v_add_f32 row_shr:1),
// 1 = packed f16 with swapped halves (v_pk_fma_f16 op_sel), 2 = v_cvt_pk_bf16_f32 round trip, 3 = f64 add / mul, 4 = v_perm_b32 +
// v_alignbit_b32, 5 = plain f32 fma chain (control), 6 = ds_bpermute (__shfl_xor) sums, 7 = ds_swizzle-free LDS transpose through a tile.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
template <int KIND>
__global__ __launch_bounds__(256) void klass(const float* a, const float* b, float s, long long n4, float* out) {
  __shared__ float tile[256 * 5];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 x = reinterpret_cast<const f32x4*>(a)[i], y = reinterpret_cast<const f32x4*>(b)[i];
    f32x4 r;
    if (KIND == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = x[k] + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[k]), 0x111, 0xf, 0xf, true));
    } else if (KIND == 1) {
      const f16x2 x0 = {(_Float16)x[0], (_Float16)x[1]}, x1 = {(_Float16)x[2], (_Float16)x[3]};
      const f16x2 y0 = {(_Float16)y[1], (_Float16)y[0]}, y1 = {(_Float16)y[3], (_Float16)y[2]}, s2 = {(_Float16)s, (_Float16)s};
      f16x2 ys0 = {y0[1], y0[0]}, ys1 = {y1[1], y1[0]};
      asm volatile("" : "+v"(ys0), "+v"(ys1));
      const f16x2 t0 = {ys0[1], ys0[0]}, t1 = {ys1[1], ys1[0]};
      const f16x2 r0 = x0 * s2 + t0, r1 = x1 * s2 + t1;
      r = f32x4{(float)r0[0], (float)r0[1], (float)r1[0], (float)r1[1]};
    } else if (KIND == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = (float)(__bf16)(x[k] * s) + (float)(__bf16)y[k];
    } else if (KIND == 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = (float)((double)x[k] * 1.000000001 + (double)y[k] * (double)s);
    } else if (KIND == 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned p = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x[k]), __builtin_bit_cast(unsigned, y[k]), 0x07060302u);
        const unsigned q = __builtin_amdgcn_alignbit(__builtin_bit_cast(unsigned, x[k]), __builtin_bit_cast(unsigned, y[k]), 16);
        r[k] = __builtin_bit_cast(float, (p & 0x7fffffffu) >> 1) + __builtin_bit_cast(float, (q & 0x7fffffffu) >> 1);
      }
    } else if (KIND == 5) {
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = __builtin_fmaf(x[k], s, y[k]);
    } else if (KIND == 6) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { float v = x[k] * s + y[k]; v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 32, 64); r[k] = v; }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) tile[threadIdx.x * 5 + k] = x[k] * s + y[k];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = tile[((threadIdx.x + 64 * k + 17) & 255) * 5 + k];
      __syncthreads();
    }
    reinterpret_cast<f32x4*>(out)[i] = r;
  }
}
extern "C" int run_class(int kind, const float* a, const float* b, float s, long long n4, float* out, hipStream_t st) {
#define K_(k) if (kind == k) hipLaunchKernelGGL(klass<k>, dim3(1024), dim3(256), 0, st, a, b, s, n4, out);
  K_(0) K_(1) K_(2) K_(3) K_(4) K_(5) K_(6) K_(7)
#undef K_
  return (int)hipGetLastError();
}
