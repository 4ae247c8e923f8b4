// Shared device/host helpers for the gfx950 kernels of libdct_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
#include "../../include/dct.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define DCT_WAVE 64

template <typename T> struct dct_type_of;
template <> struct dct_type_of<float> { static constexpr int id = DCT_F32; };
template <> struct dct_type_of<bf16_t> { static constexpr int id = DCT_BF16; };
template <> struct dct_type_of<f16_t> { static constexpr int id = DCT_F16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }
// 8 consecutive 16-bit elements as one 16-byte load
template <typename T> struct vec8_of;
template <> struct vec8_of<bf16_t> { typedef bf16x8 type; };
template <> struct vec8_of<f16_t> { typedef f16x8 type; };
template <> struct vec8_of<float> { typedef f16x8 type; };   // never used for loads of fp32 data (sizeof(T) == 4 paths branch first)

// Device-side copy of a dct_view (kernel argument).
struct View {
  char* ptr;
  int n, h, w, c;
  long long sn, sh, sw;
};
static inline View to_view(const dct_view* v) {
  View r;
  r.ptr = (char*)v->ptr; r.n = v->n; r.h = v->h; r.w = v->w; r.c = v->c;
  r.sn = v->sn; r.sh = v->sh; r.sw = v->sw;
  return r;
}
static inline bool view_ok(const dct_view* v) {
  return v && v->ptr && v->n > 0 && v->h > 0 && v->w > 0 && v->c > 0 && v->sw >= v->c;
}

// wave-level sum (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// block-level sum for blockDim.x == 256; result valid in thread 0 (and broadcast through smem[0])
__device__ __forceinline__ float block_sum_256(float v, float* smem4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem4[w] = v;
  __syncthreads();
  return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

// ---- streaming stores (round 5) -------------------------------------------------------------------------------------------------
// A 16-byte store of an activation / gradient tensor that nobody reads before hundreds of megabytes of other traffic have gone through
// the memory-side cache.  tools/probe_store_bw: with the default policy a 132 MB fill whose lines are not resident runs at 3.1 TB/s
// behind a stream of reads (the rate of every write-only stream of the step in rounds 1-4, and the reason a written byte was priced at
// two read ones), with the non-temporal policy at 5.4 TB/s.  DCT_NT_STORES = 0 builds the default-policy form (A/B).
#ifndef DCT_NT_STORES
#define DCT_NT_STORES 1
#endif
typedef __attribute__((ext_vector_type(4))) unsigned dct_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned dct_u32x2;
template <typename V> __device__ __forceinline__ void dct_store16_stream(void* p, const V& v) {
  static_assert(sizeof(V) == 16, "16-byte vector");
#if DCT_NT_STORES
  __builtin_nontemporal_store(__builtin_bit_cast(dct_u32x4, v), reinterpret_cast<dct_u32x4*>(p));
#else
  *reinterpret_cast<dct_u32x4*>(p) = __builtin_bit_cast(dct_u32x4, v);
#endif
}
template <typename V> __device__ __forceinline__ void dct_store8_stream(void* p, const V& v) {
  static_assert(sizeof(V) == 8, "8-byte vector");
#if DCT_NT_STORES
  __builtin_nontemporal_store(__builtin_bit_cast(dct_u32x2, v), reinterpret_cast<dct_u32x2*>(p));
#else
  *reinterpret_cast<dct_u32x2*>(p) = __builtin_bit_cast(dct_u32x2, v);
#endif
}

// ---- un-pooling on load (round 5) -------------------------------------------------------------------------------------------------
// The gradient at a max-pooled tensor and the pooling's routing codes (dct_maxpool2x2_fwd_codes: bits 0-1 = window position of
// the first maximum, bit 2 = its ReLU gate, bit 3 = no maximum) ARE the un-pooled gradient, at 3 bytes per window and channel
// instead of 8 -- three of its four positions are zero by construction.  The consumers of an encoder block's un-pooled gradient
// (the block's data gradient and weight gradient) expand {eight pooled bf16 + their eight codes} into the 16-byte chunk of window
// position `pos` while they stage: the same values dct_maxpool2x2_bwd_codes(relu_mask = 1, scale = 1) would have written.
// Per 32-bit pair of elements: the two code bytes move into the 16-bit lanes (v_perm_b32), "== pos | 4" becomes 0 / 0xFFFF
// (packed min / sub on the XORed codes), one AND.
__device__ __forceinline__ uint4 dct_unpool_chunk8(uint4 g, uint2 codes, unsigned pos) {
  const unsigned key = (pos | 4u) * 0x01010101u;
  const unsigned c0 = (codes.x & 0x0F0F0F0Fu) ^ key, c1 = (codes.y & 0x0F0F0F0Fu) ^ key;       // zero byte <=> routed here and gate open
  unsigned m[4];
  const unsigned z[4] = {__builtin_amdgcn_perm(0u, c0, 0x0c010c00u), __builtin_amdgcn_perm(0u, c0, 0x0c030c02u),
                         __builtin_amdgcn_perm(0u, c1, 0x0c010c00u), __builtin_amdgcn_perm(0u, c1, 0x0c030c02u)};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    asm("v_pk_min_u16 %0, %1, %2\n\tv_pk_sub_u16 %0, %0, %2" : "=&v"(m[k]) : "v"(z[k]), "v"(0x00010001u));       // 0 -> 0xFFFF, else 0
  return make_uint4(g.x & m[0], g.y & m[1], g.z & m[2], g.w & m[3]);
}

// ReLU-gate bits of eight 16-bit floats (bf16 or fp16, four packed pairs): bit i = element i > 0.  Sign-magnitude: positive <=> the pattern is > 0
// as a signed 16-bit integer (NaN patterns would count as positive: the callers' values have been through a ReLU).  Packed max / min turn each
// half into 0 / 1, three shift-ors interleave the four words, one more folds the high halves in: 13 vector instructions against the ~24 of eight
// compare + select + or (the stem's forward kernel is bound by vector issue).  Bits 8.. of the result are junk: store it as a byte.
__device__ __forceinline__ unsigned dct_positive_bits8(dct_u32x4 w) {
  unsigned t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    asm("v_pk_max_i16 %0, %1, %2\n\tv_pk_min_u16 %0, %0, %3" : "=&v"(t[k]) : "v"(w[k]), "v"(0u), "v"(0x00010001u));
  const unsigned u = t[0] | (t[1] << 2) | (t[2] << 4) | (t[3] << 6);
  return u | (u >> 15);
}

// ---- profiling hooks (prof.cpp) -------------------------------------------------------------
void dct_prof_begin(int cls, hipStream_t s);
void dct_prof_end(int cls, hipStream_t s);
extern int g_dct_prof_on;

extern thread_local char g_dct_last_plan[200];
#define DCT_PLAN_NOTE(...) snprintf(g_dct_last_plan, sizeof(g_dct_last_plan), __VA_ARGS__)

// Diagnostic (tools/ablate_step.py): launches of a kernel family can be SKIPPED -- results are garbage, timing tells what the family
// costs the captured step (its serialized time minus what the other chain's kernels hide).  dct_tune_set(1100, mask); 0 = nothing skipped.
extern int g_dct_skip_families;
enum { DCT_FAM_IGEMM2 = 1, DCT_FAM_IGEMM3M = 2, DCT_FAM_IGEMM3P = 4, DCT_FAM_WGRAD2 = 8, DCT_FAM_WGRAD3 = 16, DCT_FAM_FOLDS = 32,
       DCT_FAM_ADAM = 64, DCT_FAM_POINTWISE = 128 };
#define DCT_LAUNCH_FAM(fam, cls, kernel, grid, block, shmem, stream, ...)              \
  do {                                                                                 \
    if (g_dct_skip_families & (fam)) break;                                            \
    DCT_LAUNCH(cls, kernel, grid, block, shmem, stream, __VA_ARGS__);                  \
  } while (0)

#define DCT_LAUNCH(cls, kernel, grid, block, shmem, stream, ...)                       \
  do {                                                                                 \
    if (((cls) == DCT_PROF_POINTWISE && (g_dct_skip_families & 128)) || ((cls) == DCT_PROF_ADAM && (g_dct_skip_families & 64))) break; \
    if (g_dct_prof_on) dct_prof_begin((cls), (stream));                                \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);               \
    if (g_dct_prof_on) dct_prof_end((cls), (stream));                                  \
  } while (0)

static inline int dct_check_launch() {
  return hipGetLastError() == hipSuccess ? DCT_OK : DCT_ERR_LAUNCH;
}

static inline unsigned div_up(long long a, long long b) { return (unsigned)((a + b - 1) / b); }
