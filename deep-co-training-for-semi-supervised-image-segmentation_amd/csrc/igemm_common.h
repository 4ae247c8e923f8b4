// Pieces shared by the implicit-GEMM convolution kernels (igemm.hip: per-tap / shared-halo / packed-rows tiles; igemm4.hip: the
// one-block-per-CU ping-pong tile): the kernel parameter block, the LDS-staged row epilogue, and the inline-asm LDS read helpers.
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
  unsigned long long* stamps;   // igemm4 diagnostic builds: per-wave cycle sums
  long long x_bytes, w_bytes;   // lean loops: bytes from x / w to the end of the view / packed weights (buffer descriptor ranges)
};

// ReLU-gate bits of eight bf16 values (bit e: element e > 0) -- the one-bit-per-element image of an activation that the data
// gradient of the layer it feeds needs (1/16 of the bytes of the activation itself).
__device__ __forceinline__ unsigned relu_bits8(const bf16x8& v) {
  unsigned b = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) b |= ((float)v[e] > 0.f ? 1u : 0u) << e;
  return b;
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
    if (p.mask_bits) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((mb[t] >> e) & 1u) ? (bf16_t)((float)v[e] * p.mask_scale) : (bf16_t)0.f;
    } else if (p.mask && co < p.mask_channels) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)mk[t][e] > 0.f ? (bf16_t)((float)v[e] * p.mask_scale) : (bf16_t)0.f;
    }
    if (p.accumulate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)old[t][e]);
    }
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.y) + yo[t] + co) = v;
    if (p.bits_out) p.bits_out[(unsigned)(yo[t] + co) >> 3] = (unsigned char)relu_bits8(v);
  }
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
