// Pieces of the implicit-GEMM convolution kernels (igemm.hip: per-tap / shared-halo / packed-rows tiles): the kernel parameter
// block, the LDS-staged row epilogue, and the inline-asm LDS read helpers.
#pragma once
#include "dct_common.h"

namespace {

struct IgemmParams {
  const char* x; const char* w; const float* bias; const char* mask; char* y;
  const unsigned char* mask_bits;   // optional one-bit image of `mask` (dense [n][h][w][c/8]); the staged epilogues read it instead
  unsigned char* bits_out;          // optional: ReLU-gate bits of y (dense y only)
  float* partial;
  int M, N, Cin, R, S;
  int Ho, Wo, Hi, Wi;
  int stride, dil, pad_h, pad_w;
  long long xsN, xsH, xsW;
  long long ysN, ysH, ysW;
  long long msN, msH, msW;
  int relu, scatter, accumulate, mask_channels;
  float mask_scale;
  int kiters, kiters_per_split, cin_iters;
  int cout;  // real Cout (N/4 in scatter mode)
  int staged;  // v2: LDS-staged epilogue with 16-byte row stores (host-checked alignment / 32-bit offsets)
  long long x_bytes, w_bytes;   // lean loops: bytes from x / w to the end of the view / packed weights (buffer descriptor ranges)
  char* pool_y;                 // optional (shared-halo kernel): dense [n][Hp][Wp][cout] 2x2 ceil-mode max pooling of y, from the staged tile
  unsigned char* pool_codes;    // ... and its routing codes (dct_maxpool2x2_fwd_codes), nullable
  int Hp, Wp;
  int pool_only;                // with pool_y: y itself is not wanted (no row stores)
  int xcd_gx, xcd_gy, xcd_total;  // per-tap / packed-rows kernels, weights-heavy layers: 1-D launch of 8 * ceil(total / 8) blocks re-dealt so that
                                // each XCD owns a contiguous range of the (pixel tile fastest, channel tile, split) order -- xcd_total = 0: off
  const unsigned char* up_codes;  // optional (shared-halo kernels, round 5): x is the gradient at a 2x2 max-POOLED tensor [n][up_Hp][up_Wp][Cin] (strides
  int up_Hp, up_Wp;               // xsN / xsH / xsW) and up_codes its dense routing codes; the kernel stages the un-pooled gradient [n][Hi][Wi][Cin] from them
  const float* stem_x;          // optional (64-channel shared-halo data gradient): the stem's input image, dense [n][Ho + 2][Wo + 2]; the block then
  float* stem_slab;             // leaves its [64][10] partial of the stem's weight (taps 0..8) / bias (9) gradient here and does not store y
};

// ---- the per-chunk part of the staged epilogues (eight bf16 of one pixel = one 16-byte chunk) on PACKED 16-bit integer arithmetic.
// Round 4: the epilogues, not the K loops, carried most of the vector instructions of the conv kernels (rocprofv3 SQ_INSTS_VALU:
// 1040 per wave against 144 MFMAs in the 64-channel patch kernel; ~100 of them per chunk here -- every element unpacked to fp32,
// compared, multiplied, re-rounded, re-packed, and the accumulate path computed whether asked for or not).  A bf16 is positive exactly
// when its bits, read as int16, are > 0 (+0 / -0 / negatives are not; a NaN with a clear sign bit would count as positive -- the gate
// producers are ReLU / max-pool / dropout outputs, which never hold one), so gates and masks are two packed min / max per dword,
// and "keep or zero" with a unit scale is one AND.
union Chunk8 { bf16x8 v; unsigned d[4]; };
// per 16-bit half: 1 where the half, as a signed integer, is > 0 (a positive bf16), else 0.  Inline asm: from the vector-extension
// form (__builtin_elementwise_min / max on short2) hipcc builds compares + selects + a byte permute, five instructions per dword.
__device__ __forceinline__ unsigned pk_positive01(unsigned d) {
  unsigned r;
  asm("v_pk_min_i16 %0, %1, %2\n\tv_pk_max_i16 %0, %0, %3" : "=&v"(r) : "v"(d), "v"(0x00010001u), "v"(0u));
  return r;
}
// per 16-bit half: 0xFFFF where pk_positive01 says 1
__device__ __forceinline__ unsigned pk_positive_mask(unsigned d) {
  unsigned r;
  asm("v_pk_min_i16 %0, %1, %2\n\tv_pk_max_i16 %0, %0, %3\n\tv_pk_sub_i16 %0, %3, %0" : "=&v"(r) : "v"(d), "v"(0x00010001u), "v"(0u));
  return r;
}

// ReLU-gate bits of eight bf16 values (bit e: element e > 0) -- the one-bit-per-element image of an activation that the data
// gradient of the layer it feeds needs (1/16 of the bytes of the activation itself).  Bits above 7 of the result are junk: callers
// store the low byte.
__device__ __forceinline__ unsigned relu_bits8(const bf16x8& v) {
  return dct_positive_bits8(__builtin_bit_cast(dct_u32x4, v));       // dct_common.h (shared with the stem's forward kernel)
}
// v = gate bit ? v * scale : 0 (the ReLU / dropout backward of the producer), gates as one byte
__device__ __forceinline__ void chunk_gate_bits(bf16x8& v, unsigned bits, float scale) {
  if (scale == 1.0f) {
    Chunk8 c; c.v = v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // 0 / -1 from each element's bit (v_bfe_i32), merged into one AND mask per dword (v_bfi_b32)
      const int lo = __builtin_amdgcn_sbfe((int)bits, 2 * k, 1), hi = __builtin_amdgcn_sbfe((int)bits, 2 * k + 1, 1);
      unsigned m;
      asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(m) : "v"(0xFFFFu), "v"(lo), "v"(hi));
      c.d[k] &= m;
    }
    v = c.v;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ((bits >> e) & 1u) ? (bf16_t)((float)v[e] * scale) : (bf16_t)0.f;
  }
}
// the same with the gates read off the activation itself (mk > 0)
__device__ __forceinline__ void chunk_gate_act(bf16x8& v, const bf16x8& mk, float scale) {
  if (scale == 1.0f) {
    Chunk8 c, m; c.v = v; m.v = mk;
#pragma unroll
    for (int k = 0; k < 4; ++k) c.d[k] &= pk_positive_mask(m.d[k]);
    v = c.v;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)mk[e] > 0.f ? (bf16_t)((float)v[e] * scale) : (bf16_t)0.f;
  }
}
__device__ __forceinline__ void chunk_add(bf16x8& v, const bf16x8& old) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)old[e]);
}

// Second half of the LDS-staged epilogue of the shared-halo kernels: the block streams the [pixel][channel] image of its tile
// out in whole 16-byte chunks.  The mask / old-value loads of all NCH chunks go out together (one memory round trip), then the
// stores.  The ReLU mask of a data gradient comes from `mask_bits` (one byte per chunk) where the caller has them, else from the
// activation itself; a forward pass with `bits_out` leaves those bits for its consumer's data gradient.
template <int BM, int BN, int NW>
__device__ __forceinline__ void staged_rows_out(const IgemmParams& p, const char* tile, const int* rowY, const int* rowM, int n0, int tid) {
  constexpr int CPR = BN / 8;
  constexpr int NCH = BM * CPR / (NW * 64);
  int yo[NCH];
  bf16x8 mk[NCH], old[NCH];
  unsigned mb[NCH];
#pragma unroll
  for (int t = 0; t < NCH; ++t) {
    const int id = t * (NW * 64) + tid;
    const int row = id / CPR, co = n0 + (id % CPR) * 8;
    yo[t] = rowY[row];
    if (yo[t] >= 0) {
      if (p.mask_bits) mb[t] = p.mask_bits[(unsigned)(rowM[row] + co) >> 3];
      else if (p.mask && co < p.mask_channels) mk[t] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.mask) + rowM[row] + co);
      if (p.accumulate) old[t] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.y) + yo[t] + co);
    }
  }
#pragma unroll
  for (int t = 0; t < NCH; ++t) {
    const int id = t * (NW * 64) + tid;
    const int row = id / CPR, cc = id % CPR;
    if (yo[t] < 0) continue;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + row * (BN * 2) + ((cc ^ (row & (CPR - 1))) * 16));
    const int co = n0 + cc * 8;
    if (p.mask_bits) chunk_gate_bits(v, mb[t], p.mask_scale);
    else if (p.mask && co < p.mask_channels) chunk_gate_act(v, mk[t], p.mask_scale);
    if (p.accumulate) {                                   // a real (scalar) branch: the compiler used to compute the sum for every chunk and select
      asm volatile("" ::: "memory");
      chunk_add(v, old[t]);
    }
    dct_store16_stream(reinterpret_cast<bf16_t*>(p.y) + yo[t] + co, v);
    if (p.bits_out) p.bits_out[(unsigned)(yo[t] + co) >> 3] = (unsigned char)relu_bits8(v);
  }
}

// 2x2 / stride 2 / ceil-mode max pooling of a staged 8 x 16-pixel patch (tile row = py * 16 + px, chunk-swizzled as staged_rows_out
// reads it): the patch starts at even image coordinates, so it holds whole windows -- 4 x 8 pooled pixels x BN / 8 chunks, one
// item per thread.  Same scan order, comparisons and codes as maxpool_fwd_codes_kernel (pointwise.hip), on the same bf16 values
// that go to y: bit-identical to pooling y afterwards, without re-reading it (130 MB per network and step at the first UNet level).
template <int BN, int NW>
__device__ __forceinline__ void staged_pool_out(const IgemmParams& p, const char* tile, int img, int y0, int x0, int n0, int tid) {
  constexpr int CPR = BN / 8;
  for (int id = tid; id < 32 * CPR; id += NW * 64) {
    const int cc = id % CPR, q = id / CPR;
    const int qy = q >> 3, qx = q & 7;
    const int oy = y0 + 2 * qy, ox = x0 + 2 * qx;
    if (oy >= p.Ho || ox >= p.Wo) continue;
    float m[8];
    int arg[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { m[i] = -INFINITY; arg[i] = 8; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (oy + (k >> 1) < p.Ho && ox + (k & 1) < p.Wo) {
        const int row = (2 * qy + (k >> 1)) * 16 + 2 * qx + (k & 1);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + row * (BN * 2) + ((cc ^ (row & (CPR - 1))) * 16));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float f = (float)v[i];
          if (f > m[i]) { m[i] = f; arg[i] = k; }
        }
      }
    }
    const long long o = (((long long)img * p.Hp + (oy >> 1)) * p.Wp + (ox >> 1)) * p.cout + n0 + cc * 8;
    bf16x8 out;
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = (bf16_t)m[i];
    dct_store16_stream(reinterpret_cast<bf16_t*>(p.pool_y) + o, out);
    if (p.pool_codes) {
      union { unsigned char b[8]; uint2 w; } cd;
#pragma unroll
      for (int i = 0; i < 8; ++i) cd.b[i] = (unsigned char)(arg[i] | (m[i] > 0.f ? 4 : 0));
      dct_store8_stream(p.pool_codes + o, cd.w);
    }
  }
}

// Block -> (pixel tile, channel tile, split) for the weights-heavy deep levels.  The hardware deals consecutive workgroups of a
// launch over the 8 XCDs (L2 slices); with the natural 3-D grid every XCD therefore walks ALL channel tiles and fetches every
// weight of the layer into its own L2 (8 x 18.9 MB for the centre's 1024 -> 1024 convolution).  Here XCD j takes the j-th eighth of
// the order (pixel tile fastest), i.e. a few (channel tile, split) slices with all their pixel tiles: each weight is fetched by one
// XCD, the (much smaller) activations by all.  Returns false for the padding blocks of the 1-D launch.
__device__ __forceinline__ bool xcd_remap(const IgemmParams& p, int& bx, int& by, int& bz) {
  if (p.xcd_total == 0) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; return true; }
  const int L = blockIdx.x, per = (p.xcd_total + 7) >> 3;
  const int logical = (L & 7) * per + (L >> 3);
  if (logical >= p.xcd_total) return false;
  bx = logical % p.xcd_gx;
  const int rest = logical / p.xcd_gx;
  by = rest % p.xcd_gy; bz = rest / p.xcd_gy;
  return true;
}

// max(v, 0) as ONE instruction: from fmaxf() hipcc emits a canonicalising v_max_f32 v, v, v in front of the real one (signalling-NaN
// quieting); the values here are sums of MFMA results and biases, which are never signalling
__device__ __forceinline__ float relu1(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
  return r;
}

// 128 B of zeros: the LDS-DMA source of taps / halo pixels that fall outside the image (no branch around the DMA)
__device__ __attribute__((aligned(128))) const uint4 g_zero_page[8] = {};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) const char* lptr_c;
__device__ __forceinline__ void rd128(unsigned addr, bf16x8& dst) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); }
template <int OFF> __device__ __forceinline__ void rd128o(unsigned addr, bf16x8& dst) {     // ds_read_b128 with an immediate byte offset
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int I, int N, int STRIDE> struct RdRows {      // dst[i] <- 16 bytes at addr + i * STRIDE, i = I .. N - 1 (immediate offsets)
  __device__ static __forceinline__ void run(unsigned addr, bf16x8 (&dst)[N]) {
    rd128o<I * STRIDE>(addr, dst[I]);
    if constexpr (I + 1 < N) RdRows<I + 1, N, STRIDE>::run(addr, dst);
  }
};
template <int J, int N, int STRIDE> struct RdCols {      // dst[j] <- 16 bytes at (addr[j] ^ flip) + j * STRIDE, j = J .. N - 1
  __device__ static __forceinline__ void run(const unsigned (&addr)[N], unsigned flip, bf16x8 (&dst)[N]) {
    rd128o<J * STRIDE>(addr[J] ^ flip, dst[J]);
    if constexpr (J + 1 < N) RdCols<J + 1, N, STRIDE>::run(addr, flip, dst);
  }
};
// one LDS-DMA piece through a buffer descriptor: 16 bytes per lane to lds + lane * 16, from base + voff + soff; a lane whose
// voff + soff lies outside the descriptor's range writes ZEROS (tools/probe_buffer_lds).  Kept in a __device__ helper: called
// straight from a kernel template, hipcc's host pass silently fails to instantiate the kernel's stub.
__device__ __forceinline__ void buf_lds16(__amdgpu_buffer_rsrc_t r, unsigned lds, int voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(size_t)lds, 16, voff, (int)soff, 0, 0);
}
#define DCT_BUF_INVALID ((int)0x80000000u)      /* a lane offset every descriptor (< 2 GiB) rejects */
template <int N> __device__ __forceinline__ void lgkm_wait3() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void touch8(bf16x8& r) { asm volatile("" : "+v"(r)); }

}  // namespace
