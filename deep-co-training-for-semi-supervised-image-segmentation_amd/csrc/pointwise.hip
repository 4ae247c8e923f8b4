// HBM-bound kernels around the conv stack: weight repack, Cin=1 stem conv (fwd/dgrad/wgrad),
// classifier head, 2x2 max-pool, bilinear resize (align_corners), dropout, ReLU backward, cast.
// All operate on NHWC views with 16-byte (bf16x8 / f32x4) accesses along the channel axis.
#include "dct_common.h"

namespace {

// ---- vector helpers: VEC channels as floats -------------------------------------------------
template <typename T, int VEC> struct VecIO;
template <> struct VecIO<bf16_t, 8> {
  __device__ static void load(const bf16_t* p, float* v) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  __device__ static void store(bf16_t* p, const float* v) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    dct_store16_stream(p, t);
  }
};
template <> struct VecIO<float, 4> {
  __device__ static void load(const float* p, float* v) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ static void store(float* p, const float* v) {
    dct_store16_stream(p, f32x4{v[0], v[1], v[2], v[3]});
  }
};
template <typename T> struct VecIO<T, 1> {
  __device__ static void load(const T* p, float* v) { v[0] = to_f32(*p); }
  __device__ static void store(T* p, const float* v) { *p = from_f32<T>(v[0]); }
};
template <typename T> struct VecOf { static constexpr int value = 16 / sizeof(T); };

struct PixIdx { int n, y, x, cv; bool ok; };
// idx -> (pixel of a [n,h,w] grid, channel-vector cv).  32-bit divisions whenever the element count allows
// (64-bit integer division is a ~100-instruction software routine on the VALU and used to dominate these kernels).
__device__ __forceinline__ PixIdx decode(long long idx, int n, int h, int w, int cvecs) {
  PixIdx r;
  const long long total = (long long)n * h * w * cvecs;
  r.ok = idx < total;
  if (!r.ok) { r.n = r.y = r.x = r.cv = 0; return r; }
  if (total <= 0x7fffffffll) {
    const unsigned i = (unsigned)idx, cvu = (unsigned)cvecs, wu = (unsigned)w, hu = (unsigned)h;
    unsigned pix = i / cvu;
    r.cv = (int)(i - pix * cvu);
    const unsigned row = pix / wu;
    r.x = (int)(pix - row * wu);
    const unsigned img = row / hu;
    r.y = (int)(row - img * hu);
    r.n = (int)img;
    return r;
  }
  r.cv = (int)(idx % cvecs);
  long long pix = idx / cvecs;
  r.x = (int)(pix % w); pix /= w;
  r.y = (int)(pix % h);
  r.n = (int)(pix / h);
  return r;
}
__device__ __forceinline__ long long voff(const View& v, int n, int y, int x) {
  return n * v.sn + y * v.sh + x * v.sw;
}

// ---- weight repack -------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const float* src, T* dst, int P, int Tt, int Q, int transpose, int flip) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)P * Tt * Q;
  if (i >= total) return;
  if (!transpose) { dst[i] = from_f32<T>(src[i]); return; }
  const int p = (int)(i % P);
  long long r = i / P;
  int t2, q;
  if (transpose == 1) { t2 = (int)(r % Tt); q = (int)(r / Tt); }   // dst index i = (q, t', p)
  else { q = (int)(r % Q); t2 = (int)(r / Q); }                    // transpose == 2: dst index i = (t', q, p)
  const int t = flip ? Tt - 1 - t2 : t2;
  dst[i] = from_f32<T>(src[((long long)p * Tt + t) * Q + q]);
}

// dst[q][t'][p] = src[p][t][q] (t = flip ? T-1-t' : t'): one 32x32 (p, q) tile per block through LDS, so both
// the fp32 reads (along q) and the packed writes (along p) are coalesced.  P, Q multiples of 32.
template <typename T>
__global__ __launch_bounds__(256) void pack_transpose_kernel(const float* src, T* dst, int P, int Tt, int Q, int flip) {
  __shared__ float tile[32][33];
  const int p0 = blockIdx.x * 32, q0 = blockIdx.y * 32, t2 = blockIdx.z;
  const int t = flip ? Tt - 1 - t2 : t2;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pr = ly + 8 * k;
    tile[pr][lx] = src[((long long)(p0 + pr) * Tt + t) * Q + q0 + lx];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int qr = ly + 8 * k;
    dst[((long long)(q0 + qr) * Tt + t2) * P + p0 + lx] = from_f32<T>(tile[lx][qr]);
  }
}

// All transposed weight packs of a network in ONE launch: job j owns the 32 x 32 (p, q) tiles [tile_begin[j],
// tile_begin[j+1]) of dst[q * dq + t' * dt + p] = src[p][t][q]  (t = flip ? T-1-t' : t').  A UNet has 21 such packs of
// 2-20 us each per step; as one launch they run at HBM speed.
struct PackJob { const void* src; void* dst; int P, T, Q, flip; long long dq, dt; int tile_begin, src_bf16; };
template <typename T>
__global__ __launch_bounds__(256) void pack_transpose_batched_kernel(const PackJob* jobs, int njobs) {
  __shared__ float tile[32][33];
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].tile_begin) ++j;      // <= 32 jobs: a short uniform scan
  const PackJob jb = jobs[j];
  int rel = blockIdx.x - jb.tile_begin;
  const int pt = jb.P / 32, qt = jb.Q / 32;
  const int p0 = (rel % pt) * 32; rel /= pt;
  const int q0 = (rel % qt) * 32;
  const int t2 = rel / qt;
  const int t = jb.flip ? jb.T - 1 - t2 : t2;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pr = ly + 8 * k;
    const long long si = ((long long)(p0 + pr) * jb.T + t) * jb.Q + q0 + lx;
    tile[pr][lx] = jb.src_bf16 ? (float)reinterpret_cast<const bf16_t*>(jb.src)[si] : reinterpret_cast<const float*>(jb.src)[si];
  }
  __syncthreads();
  T* dst = reinterpret_cast<T*>(jb.dst);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int qr = ly + 8 * k;
    dst[(long long)(q0 + qr) * jb.dq + (long long)t2 * jb.dt + p0 + lx] = from_f32<T>(tile[lx][qr]);
  }
}

// The same packs for 16-bit sources and destinations on 64 x 64 (p, q) tiles with 16-byte accesses on both sides (the 32 x 32
// form above moves 2 bytes per lane per access: 1.6 TB/s; a UNet's 21 packs are 2 x 62 MB per step).  tile_begin counts
// 64 x 64 tiles here; P, Q multiples of 64; src and dst 16-byte aligned with dq, dt multiples of 8.
__global__ __launch_bounds__(256) void pack_transpose_batched64_kernel(const PackJob* jobs, int njobs) {
  constexpr int PITCH = 72;                      // 16-bit elements per LDS row (64 + 8: rows stay 16-byte aligned)
  __shared__ __attribute__((aligned(16))) unsigned short tile[64 * PITCH];
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].tile_begin) ++j;
  const PackJob jb = jobs[j];
  int rel = blockIdx.x - jb.tile_begin;
  const int pt = jb.P / 64, qt = jb.Q / 64;
  const int p0 = (rel % pt) * 64; rel /= pt;
  const int q0 = (rel % qt) * 64;
  const int t2 = rel / qt;
  const int t = jb.flip ? jb.T - 1 - t2 : t2;
  const int ch = threadIdx.x & 7, r0 = threadIdx.x >> 3;          // 8 chunks of 8 elements x 32 rows per pass
  const unsigned short* src = reinterpret_cast<const unsigned short*>(jb.src);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int pr = r0 + 32 * k;
    const uint4 v = *reinterpret_cast<const uint4*>(src + ((long long)(p0 + pr) * jb.T + t) * jb.Q + q0 + ch * 8);
    *reinterpret_cast<uint4*>(&tile[pr * PITCH + ch * 8]) = v;
  }
  __syncthreads();
  unsigned short* dst = reinterpret_cast<unsigned short*>(jb.dst);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int qr = r0 + 32 * k;                                    // output row: source column q0 + qr
    unsigned short e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = tile[(ch * 8 + i) * PITCH + qr];
    uint4 v;
    v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
    v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
    *reinterpret_cast<uint4*>(dst + (long long)(q0 + qr) * jb.dq + (long long)t2 * jb.dt + p0 + ch * 8) = v;
  }
}

// ---- Cin = 1 stem --------------------------------------------------------------------------
struct StemGeom { int R, S, stride, dil, pad_h, pad_w, relu, cout; };

// RS > 0: R = S = RS known at compile time (the tap array stays in registers; with run-time R, S the
// dynamically indexed xin[] lives in scratch memory and the kernel runs at a fifth of HBM speed).
template <typename T, int VEC, int RS>
__global__ __launch_bounds__(256) void stem_fwd_kernel(View x, const float* w, const float* bias, View y, StemGeom g) {
  extern __shared__ float sw[];  // [R*S][cout] (a thread's VEC channels of a tap are contiguous: conflict-free), then bias[cout]
  const int R = RS > 0 ? RS : g.R, S = RS > 0 ? RS : g.S;
  const int taps = R * S;
  for (int i = threadIdx.x; i < g.cout * taps; i += 256) { const int c = i / taps, t = i - c * taps; sw[t * g.cout + c] = w[i]; }
  for (int i = threadIdx.x; i < g.cout; i += 256) sw[g.cout * taps + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int cvecs = (g.cout + VEC - 1) / VEC;
  const PixIdx id = decode((long long)blockIdx.x * 256 + threadIdx.x, y.n, y.h, y.w, cvecs);
  if (!id.ok) return;
  constexpr int MAXT = RS > 0 ? RS * RS : 25;
  float xin[MAXT];
  const float* xp = reinterpret_cast<const float*>(x.ptr);
  if constexpr (RS > 0) {
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        const int iy = id.y * g.stride + r * g.dil - g.pad_h, ix = id.x * g.stride + s * g.dil - g.pad_w;
        xin[r * RS + s] = ((unsigned)iy < (unsigned)x.h && (unsigned)ix < (unsigned)x.w) ? xp[voff(x, id.n, iy, ix)] : 0.f;
      }
  } else {
    for (int r = 0; r < R; ++r)
      for (int s = 0; s < S; ++s) {
        const int iy = id.y * g.stride + r * g.dil - g.pad_h, ix = id.x * g.stride + s * g.dil - g.pad_w;
        xin[r * S + s] = ((unsigned)iy < (unsigned)x.h && (unsigned)ix < (unsigned)x.w) ? xp[voff(x, id.n, iy, ix)] : 0.f;
      }
  }
  float out[VEC];
  const bool full = id.cv * VEC + VEC <= g.cout;
  if (RS > 0 && full) {
    const float* swc = sw + id.cv * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int i = 0; i < VEC; ++i) out[i] = fmaf(xin[t], swc[t * g.cout + i], out[i]);   // tap order 0..T-1 per channel, as below
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      out[i] += swc[g.cout * taps + i];
      if (g.relu) out[i] = fmaxf(out[i], 0.f);
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int c = id.cv * VEC + i;
      float a = 0.f;
      if (c < g.cout) {
        for (int t = 0; t < taps; ++t) a = fmaf(xin[t], sw[t * g.cout + c], a);
        a += sw[g.cout * taps + c];
        if (g.relu) a = fmaxf(a, 0.f);
      }
      out[i] = a;
    }
  }
  T* yp = reinterpret_cast<T*>(y.ptr) + voff(y, id.n, id.y, id.x) + id.cv * VEC;
  if (full) VecIO<T, VEC>::store(yp, out);
  else
    for (int i = 0; i < VEC && id.cv * VEC + i < g.cout; ++i) yp[i] = from_f32<T>(out[i]);
}

// 3x3, stride 1, dilation 1, cout a multiple of VEC: one thread per (row, PX adjacent output pixels, VEC-channel slice).  The 3 x (PX + 2) input
// window and the slice's 9xVEC weights sit in registers; per channel the taps accumulate in the same order as stem_fwd_kernel.
// The kernel writes 132 MB per network and step and reads next to nothing, but it was VECTOR-ISSUE bound (round 5, from its ISA: 1041 vector
// instructions per thread for 288 FMAs -- a wave64 instruction occupies its SIMD for four cycles, and this library is built without packed-FP32
// instructions): 64-bit multiply-adds for the address of every one of the 18 window loads, their range predicates, and the weights' LDS reads
// per four pixels.  Now: one 64-bit window base per thread and 32-bit offsets from it, unpredicated loads for threads whose window lies inside
// the image (all but the last of a row with valid padding), eight pixels per thread.
// (Measured and dropped: a thread's pixels EIGHT apart, so that a wave's store instruction writes 1 KB of one row instead of eight 128-byte
// pieces 512 bytes apart: 39.0 -> 52.8 us -- tools/probe_store_bw: the two patterns fill at the same 6.7 TB/s; the per-pixel windows cost it.)
template <typename T, int VEC, int PX>
__global__ __launch_bounds__(256) void stem_fwd3x3_kernel(View x, const float* w, const float* bias, View y, StemGeom g,
                                                          unsigned char* __restrict__ bits) {
  extern __shared__ float sw[];  // [9][cout], then bias[cout]
  for (int i = threadIdx.x; i < g.cout * 9; i += 256) { const int c = i / 9, t = i - c * 9; sw[t * g.cout + c] = w[i]; }
  for (int i = threadIdx.x; i < g.cout; i += 256) sw[g.cout * 9 + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int cvecs = g.cout / VEC, wq = (y.w + PX - 1) / PX;
  const PixIdx id = decode((long long)blockIdx.x * 256 + threadIdx.x, y.n, y.h, wq, cvecs);
  if (!id.ok) return;
  const int x0 = id.x * PX;
  const int iy0 = id.y - g.pad_h, ix0 = x0 - g.pad_w;
  const int xsh = (int)x.sh, xsw = (int)x.sw;                  // (the host checked that a 3-row window's offsets fit 32 bits)
  const float* win = reinterpret_cast<const float*>(x.ptr) + ((long long)id.n * x.sn + (long long)iy0 * x.sh + (long long)ix0 * x.sw);
  float xin[3][PX + 2];
  if (iy0 >= 0 && iy0 + 2 < x.h && ix0 >= 0 && ix0 + PX + 1 < x.w) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < PX + 2; ++j) xin[r][j] = win[r * xsh + j * xsw];
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const bool rok = (unsigned)(iy0 + r) < (unsigned)x.h;
#pragma unroll
      for (int j = 0; j < PX + 2; ++j) xin[r][j] = (rok && (unsigned)(ix0 + j) < (unsigned)x.w) ? win[r * xsh + j * xsw] : 0.f;
    }
  }
  float wv[9][VEC], bv[VEC];
  const float* swc = sw + id.cv * VEC;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) wv[t][i] = swc[t * g.cout + i];
#pragma unroll
  for (int i = 0; i < VEC; ++i) bv[i] = swc[9 * g.cout + i];
  T* yp = reinterpret_cast<T*>(y.ptr) + voff(y, id.n, id.y, x0) + id.cv * VEC;
  const int ysw = (int)y.sw;
  unsigned char* bp = bits ? bits + ((((long long)id.n * y.h + id.y) * y.w + x0) * cvecs + id.cv) : nullptr;
  const int np = min(PX, y.w - x0);
#pragma unroll
  for (int p = 0; p < PX; ++p) {
    if (p >= np) break;
    float out[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < VEC; ++i) out[i] = fmaf(xin[r][p + s], wv[r * 3 + s][i], out[i]);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      out[i] += bv[i];
      if (g.relu) out[i] = fmaxf(out[i], 0.f);
    }
    if constexpr (VEC == 8) {
      typename vec8_of<T>::type pk;
#pragma unroll
      for (int i = 0; i < VEC; ++i) pk[i] = (T)out[i];
      dct_store16_stream(yp + p * ysw, pk);
      // ReLU-gate bits of the ROUNDED outputs (dense y: one byte per pixel and slice), for the consumer's data gradient
      if (bp) bp[p * cvecs] = (unsigned char)dct_positive_bits8(__builtin_bit_cast(dct_u32x4, pk));
    } else {
      VecIO<T, VEC>::store(yp + p * ysw, out);
    }
  }
}

// dx[n,iy,ix] = sum_{co,r,s} dy[n,oy,ox,co] * w[co][r][s]  with oy*stride + r*dil - pad = iy.
// One thread per (input pixel, VEC-channel slice); the cout/VEC slices of a pixel sit in adjacent
// lanes and are summed with xor-shuffles (cout/VEC is a power of two <= 64).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void stem_dgrad_kernel(View dy, const float* w, View dx, StemGeom g) {
  extern __shared__ float sw[];
  const int taps = g.R * g.S;
  for (int i = threadIdx.x; i < g.cout * taps; i += 256) sw[i] = w[i];
  __syncthreads();
  const int cvecs = g.cout / VEC;
  const PixIdx id = decode((long long)blockIdx.x * 256 + threadIdx.x, dx.n, dx.h, dx.w, cvecs);
  float a = 0.f;
  if (id.ok) {
    for (int r = 0; r < g.R; ++r) {
      const int ty = id.y + g.pad_h - r * g.dil;
      if (ty < 0 || ty % g.stride) continue;
      const int oy = ty / g.stride;
      if (oy >= dy.h) continue;
      for (int s = 0; s < g.S; ++s) {
        const int tx = id.x + g.pad_w - s * g.dil;
        if (tx < 0 || tx % g.stride) continue;
        const int ox = tx / g.stride;
        if (ox >= dy.w) continue;
        float v[VEC];
        VecIO<T, VEC>::load(reinterpret_cast<const T*>(dy.ptr) + voff(dy, id.n, oy, ox) + id.cv * VEC, v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) a = fmaf(v[i], sw[(id.cv * VEC + i) * taps + r * g.S + s], a);
      }
    }
  }
  for (int off = 1; off < cvecs; off <<= 1) a += __shfl_xor(a, off, 64);
  if (id.ok && id.cv == 0) reinterpret_cast<float*>(dx.ptr)[voff(dx, id.n, id.y, id.x)] = a;
}

// The same data gradient on the matrix pipe (bf16 dy, 64 channels, 3x3, stride 1, no padding: the UNet stem's FGSM pass).  The kernel above
// is vector-ALU bound (72 FMAs + their conversions and address arithmetic per eight channels of one tap window: 136 us at 16 x 256 x 256).
// Here G[q][t] = sum_c dy[q][c] * w[c][t] for every dy pixel q of a block's 18 x 18 halo is a GEMM (M = pixels, K = 64 channels, N = 9 taps
// padded to 16) on v_mfma_f32_16x16x32_bf16 -- dy is bf16 already, each fp32 weight is split into three bf16 parts (hi + mid + lo = all 24
// mantissa bits, so the products are exact and only the order of the fp32 additions differs from the FMA chain) -- and
// dx[p] = sum_t G[p - t][t] is nine LDS reads per pixel.  A fragment: lane l = pixel l % 16 of the row block, 16-byte channel chunk l / 16,
// loaded straight from global memory (one 16-byte load per lane and MFMA triple).
__global__ __launch_bounds__(256) void stem_dgrad_mfma_kernel(View dy, const float* __restrict__ w, View dx, int tiles_x, int tiles_y) {
  constexpr int HW = 18, HPIX = HW * HW, MB = (HPIX + 15) / 16, GS = 17;      // 324 halo pixels in 21 row blocks; G row stride (floats)
  __shared__ float G[MB * 16 * GS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y; const int n = b / tiles_y;
  const int Y0 = ty * 16, X0 = tx * 16;
  // B fragments: column l15 = tap, channels kb * 32 + kq * 8 + {0..7}; three bf16 parts of every weight
  bf16x8 wb[2][3];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = kb * 32 + kq * 8 + i;
      const float v = l15 < 9 ? w[c * 9 + l15] : 0.f;
      const bf16_t h = (bf16_t)v;
      const float r1 = v - (float)h;
      const bf16_t m = (bf16_t)r1;
      const bf16_t lo = (bf16_t)(r1 - (float)m);
      wb[kb][0][i] = h; wb[kb][1][i] = m; wb[kb][2][i] = lo;
    }
  const bf16_t* dyp = reinterpret_cast<const bf16_t*>(dy.ptr) + (long long)n * dy.sn;
  for (int mb = wave; mb < MB; mb += 4) {
    const int h = mb * 16 + l15;
    const int hy = h / HW, hx = h - hy * HW;
    const int qy = Y0 - 2 + hy, qx = X0 - 2 + hx;
    const bool ok = h < HPIX && (unsigned)qy < (unsigned)dy.h && (unsigned)qx < (unsigned)dy.w;
    bf16x8 a[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (ok) a[kb] = *reinterpret_cast<const bf16x8*>(dyp + (long long)qy * dy.sh + (long long)qx * dy.sw + kb * 32 + kq * 8);
      else
#pragma unroll
        for (int i = 0; i < 8; ++i) a[kb][i] = (bf16_t)0.f;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int part = 2; part >= 0; --part)          // smallest parts first
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb], wb[kb][part], acc, 0, 0, 0);
    // D: column l15 (tap), rows 4 * kq + {0..3} (pixels of the row block)
    if (l15 < 9) {
#pragma unroll
      for (int e = 0; e < 4; ++e) G[(mb * 16 + 4 * kq + e) * GS + l15] = acc[e];
    }
  }
  __syncthreads();
  const int i = tid >> 4, j = tid & 15;
  const int y = Y0 + i, x = X0 + j;
  if (y < dx.h && x < dx.w) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) s += G[((i + 2 - r) * HW + (j + 2 - c)) * GS + r * 3 + c];
    reinterpret_cast<float*>(dx.ptr)[voff(dx, n, y, x)] = s;
  }
}

// ---- classifier head -----------------------------------------------------------------------
// y[pix][co] = sum_ci x[pix][ci]*w[co][ci] + b[co];  thread per pixel, cout <= 8
template <typename T, int VEC>
__global__ __launch_bounds__(256) void head_fwd_kernel(View x, const float* w, const float* bias, View y) {
  extern __shared__ float sw[];  // [cout][cin] + bias
  const int cin = x.c, cout = y.c;
  for (int i = threadIdx.x; i < cout * cin; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < cout; i += 256) sw[cout * cin + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const PixIdx id = decode((long long)blockIdx.x * 256 + threadIdx.x, x.n, x.h, x.w, 1);
  if (!id.ok) return;
  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = 0.f;
  const T* xp = reinterpret_cast<const T*>(x.ptr) + voff(x, id.n, id.y, id.x);
  for (int c0 = 0; c0 < cin; c0 += VEC) {
    float v[VEC];
    VecIO<T, VEC>::load(xp + c0, v);
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < cout)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[o] = fmaf(v[i], sw[o * cin + c0 + i], acc[o]);
  }
  float* yp = reinterpret_cast<float*>(y.ptr) + voff(y, id.n, id.y, id.x);
  for (int o = 0; o < cout; ++o) yp[o] = acc[o] + sw[cout * cin + o];
}
// The same with cin / VEC lanes per pixel (a power of two): a pixel's channels are ONE coalesced read instead of a thread's serial walk over them, and
// the launch has cin / VEC times the threads -- the forward pass's tail runs where nothing else does (tools/phase_stamps.py), 113 k pixels are
// 1.7 waves per SIMD.  Partial dot products folded by a butterfly; lane o of the group stores class o.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void head_fwd_lanes_kernel(View x, const float* w, const float* bias, View y, int lanes_log2) {
  extern __shared__ float sw[];  // [cout][cin] + bias
  const int cin = x.c, cout = y.c;
  for (int i = threadIdx.x; i < cout * cin; i += 256) sw[i] = w[i];
  for (int i = threadIdx.x; i < cout; i += 256) sw[cout * cin + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int L = 1 << lanes_log2, sub = (int)threadIdx.x & (L - 1);
  const PixIdx id = decode(((long long)blockIdx.x * 256 + threadIdx.x) >> lanes_log2, x.n, x.h, x.w, 1);
  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = 0.f;
  if (id.ok) {
    float v[VEC];
    VecIO<T, VEC>::load(reinterpret_cast<const T*>(x.ptr) + voff(x, id.n, id.y, id.x) + sub * VEC, v);
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < cout)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[o] = fmaf(v[i], sw[o * cin + sub * VEC + i], acc[o]);
  }
  for (int m = 1; m < L; m <<= 1)
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < cout) acc[o] += __shfl_xor(acc[o], m);
  if (!id.ok || sub >= cout) return;
  float r = acc[0];
#pragma unroll
  for (int o = 1; o < 8; ++o) r = sub == o ? acc[o] : r;
  reinterpret_cast<float*>(y.ptr)[voff(y, id.n, id.y, id.x) + sub] = r + sw[cout * cin + sub];
}
// dx[pix][ci] = sum_co dy[pix][co]*w[co][ci]  (* x>0 when relu_mask); thread per (pixel, VEC slice)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void head_dx_kernel(View x, View dy, const float* w, View dx, int relu_mask) {
  extern __shared__ float sw[];
  const int cin = x.c, cout = dy.c;
  for (int i = threadIdx.x; i < cout * cin; i += 256) sw[i] = w[i];
  __syncthreads();
  const PixIdx id = decode((long long)blockIdx.x * 256 + threadIdx.x, x.n, x.h, x.w, cin / VEC);
  if (!id.ok) return;
  const float* dp = reinterpret_cast<const float*>(dy.ptr) + voff(dy, id.n, id.y, id.x);
  float out[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) out[i] = 0.f;
  for (int o = 0; o < cout; ++o) {
    const float d = dp[o];
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = fmaf(d, sw[o * cin + id.cv * VEC + i], out[i]);
  }
  if (relu_mask) {
    float xv[VEC];
    VecIO<T, VEC>::load(reinterpret_cast<const T*>(x.ptr) + voff(x, id.n, id.y, id.x) + id.cv * VEC, xv);
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = xv[i] > 0.f ? out[i] : 0.f;
  }
  VecIO<T, VEC>::store(reinterpret_cast<T*>(dx.ptr) + voff(dx, id.n, id.y, id.x) + id.cv * VEC, out);
}
// ---- max pool 2x2 s2 ceil ------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(View x, View y) {
  const PixIdx id = decode((long long)blockIdx.x * 256 + threadIdx.x, y.n, y.h, y.w, y.c / VEC);
  if (!id.ok) return;
  float m[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) m[i] = -INFINITY;
  for (int dy = 0; dy < 2; ++dy)
    for (int dx = 0; dx < 2; ++dx) {
      const int iy = 2 * id.y + dy, ix = 2 * id.x + dx;
      if (iy < x.h && ix < x.w) {
        float v[VEC];
        VecIO<T, VEC>::load(reinterpret_cast<const T*>(x.ptr) + voff(x, id.n, iy, ix) + id.cv * VEC, v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) m[i] = v[i] > m[i] ? v[i] : m[i];
      }
    }
  VecIO<T, VEC>::store(reinterpret_cast<T*>(y.ptr) + voff(y, id.n, id.y, id.x) + id.cv * VEC, m);
}
// thread per (pooled pixel, VEC slice): reads the 2x2 window once, routes dy to the first maximum (strict '>': first
// max wins, as torch) and writes all four dx pixels (ceil-mode edge windows have fewer).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(View x, View dy, View dx, int relu_mask, float scale) {
  const PixIdx id = decode((long long)blockIdx.x * 256 + threadIdx.x, dy.n, dy.h, dy.w, x.c / VEC);
  if (!id.ok) return;
  float v[4][VEC];
  bool in[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int iy = 2 * id.y + (k >> 1), ix = 2 * id.x + (k & 1);
    in[k] = iy < x.h && ix < x.w;
    if (in[k]) VecIO<T, VEC>::load(reinterpret_cast<const T*>(x.ptr) + voff(x, id.n, iy, ix) + id.cv * VEC, v[k]);
  }
  float g[VEC];
  VecIO<T, VEC>::load(reinterpret_cast<const T*>(dy.ptr) + voff(dy, id.n, id.y, id.x) + id.cv * VEC, g);
  int arg[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    float best = -INFINITY;
    arg[i] = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (in[k] && v[k][i] > best) { best = v[k][i]; arg[i] = k; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!in[k]) continue;
    float out[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float o = arg[i] == k ? g[i] : 0.f;
      if (relu_mask) o = v[k][i] > 0.f ? o * scale : 0.f;
      out[i] = o;
    }
    const int iy = 2 * id.y + (k >> 1), ix = 2 * id.x + (k & 1);
    VecIO<T, VEC>::store(reinterpret_cast<T*>(dx.ptr) + voff(dx, id.n, iy, ix) + id.cv * VEC, out);
  }
}

// The same pair with the routing decision kept from the forward pass: one byte per pooled element -- bits 0-1 the window
// position of the first maximum (scan order, as above), bit 2 "the maximum is > 0" (the ReLU / dropout gate of the pool's
// input), bit 3 "no maximum" (a window of NaNs routes nothing, as above).  The backward pass then reads dy and the codes only:
// 1/8 of the bytes of the pool's input, which it no longer re-reads (130 MB per network and step at the first UNet level).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_fwd_codes_kernel(View x, View y, unsigned char* __restrict__ codes) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const PixIdx id = decode(t, y.n, y.h, y.w, y.c / VEC);
  if (!id.ok) return;
  float m[VEC];
  int arg[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { m[i] = -INFINITY; arg[i] = 8; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int iy = 2 * id.y + (k >> 1), ix = 2 * id.x + (k & 1);
    if (iy < x.h && ix < x.w) {
      float v[VEC];
      VecIO<T, VEC>::load(reinterpret_cast<const T*>(x.ptr) + voff(x, id.n, iy, ix) + id.cv * VEC, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        if (v[i] > m[i]) { m[i] = v[i]; arg[i] = k; }
    }
  }
  VecIO<T, VEC>::store(reinterpret_cast<T*>(y.ptr) + voff(y, id.n, id.y, id.x) + id.cv * VEC, m);
  union { unsigned char b[VEC]; unsigned w[VEC / 4]; } cd;             // dense [n][h][w][c], same walk as `decode`
#pragma unroll
  for (int i = 0; i < VEC; ++i) cd.b[i] = (unsigned char)(arg[i] | (m[i] > 0.f ? 4 : 0));
  unsigned* cp = reinterpret_cast<unsigned*>(codes + t * VEC);
  if constexpr (VEC == 8) *reinterpret_cast<uint2*>(cp) = make_uint2(cd.w[0], cd.w[1]);
  else cp[0] = cd.w[0];
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_codes_kernel(const unsigned char* __restrict__ codes, View dy, View dx, int relu_mask,
                                                                float scale) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const PixIdx id = decode(t, dy.n, dy.h, dy.w, dy.c / VEC);
  if (!id.ok) return;
  float g[VEC];
  VecIO<T, VEC>::load(reinterpret_cast<const T*>(dy.ptr) + voff(dy, id.n, id.y, id.x) + id.cv * VEC, g);
  union { unsigned char b[VEC]; unsigned w[VEC / 4]; } cu;
  const unsigned* cp = reinterpret_cast<const unsigned*>(codes + t * VEC);
  if constexpr (VEC == 8) { const uint2 q = *reinterpret_cast<const uint2*>(cp); cu.w[0] = q.x; cu.w[1] = q.y; }
  else cu.w[0] = cp[0];
  const unsigned char* cd = cu.b;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    if (relu_mask) g[i] = (cd[i] & 4) ? g[i] * scale : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int iy = 2 * id.y + (k >> 1), ix = 2 * id.x + (k & 1);
    if (iy >= dx.h || ix >= dx.w) continue;
    float out[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = (cd[i] & 11) == k ? g[i] : 0.f;
    VecIO<T, VEC>::store(reinterpret_cast<T*>(dx.ptr) + voff(dx, id.n, iy, ix) + id.cv * VEC, out);
  }
}

// ---- bilinear, align_corners = True ---------------------------------------------------------
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp lerp_of(int j, float scale, int in_size) {
  // torch: real = scale * j; i0 = min(floor(real), in-1); lambda1 = clamp(real - i0, 0, 1)
  const float real = scale * (float)j;
  Lerp r;
  r.i0 = min((int)floorf(real), in_size - 1);
  r.i1 = min(r.i0 + 1, in_size - 1);
  r.w1 = fminf(fmaxf(real - (float)r.i0, 0.f), 1.f);
  r.w0 = 1.f - r.w1;
  return r;
}
template <typename T, int VEC>
__device__ __forceinline__ void bilinear_fwd_body(const View& x, const View& y, float sh, float sw, long long idx) {
  const PixIdx id = decode(idx, y.n, y.h, y.w, y.c / VEC);
  if (!id.ok) return;
  const Lerp ly = lerp_of(id.y, sh, x.h), lx = lerp_of(id.x, sw, x.w);
  const T* xp = reinterpret_cast<const T*>(x.ptr) + id.cv * VEC;
  float a[VEC], b[VEC], c[VEC], d[VEC], out[VEC];
  VecIO<T, VEC>::load(xp + voff(x, id.n, ly.i0, lx.i0), a);
  VecIO<T, VEC>::load(xp + voff(x, id.n, ly.i0, lx.i1), b);
  VecIO<T, VEC>::load(xp + voff(x, id.n, ly.i1, lx.i0), c);
  VecIO<T, VEC>::load(xp + voff(x, id.n, ly.i1, lx.i1), d);
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    out[i] = ly.w0 * (lx.w0 * a[i] + lx.w1 * b[i]) + ly.w1 * (lx.w0 * c[i] + lx.w1 * d[i]);
  VecIO<T, VEC>::store(reinterpret_cast<T*>(y.ptr) + voff(y, id.n, id.y, id.x) + id.cv * VEC, out);
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(View x, View y, float sh, float sw) {
  bilinear_fwd_body<T, VEC>(x, y, sh, sw, (long long)blockIdx.x * 256 + threadIdx.x);
}
// Several resizes in one launch (dct_bilinear_fwd_batched: a UNet's four skip connections -- every pooled tensor exists once the encoder is through,
// so the four launches of ~7 us on each model's decoder chain are one, in front of the centre): job k owns the blocks [blk0, next blk0).
constexpr int kBilJobs = 8;
struct BilJob { View x, y; float sh, sw; int blk0, pad; };
struct BilJobs { BilJob j[kBilJobs]; int n; };
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bilinear_fwd_batched_kernel(BilJobs js) {
  int k = 0;
  for (int i = 1; i < js.n; ++i) k = (int)blockIdx.x >= js.j[i].blk0 ? i : k;
  const BilJob& jb = js.j[k];
  bilinear_fwd_body<T, VEC>(jb.x, jb.y, jb.sh, jb.sw, (long long)((int)blockIdx.x - jb.blk0) * 256 + threadIdx.x);
}
// gather backward: thread per (source pixel, VEC slice); candidate destination rows/cols are a
// slightly widened analytic range and each is tested with the forward's exact index arithmetic.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(View dy, View dx, float sh, float sw, int accumulate) {
  const PixIdx id = decode((long long)blockIdx.x * 256 + threadIdx.x, dx.n, dx.h, dx.w, dx.c / VEC);
  if (!id.ok) return;
  auto range = [](int i, float scale, int out_size, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }
    lo = max(0, (int)floorf((float)(i - 1) / scale) - 1);
    hi = min(out_size - 1, (int)ceilf((float)(i + 1) / scale) + 1);
  };
  int jlo, jhi, klo, khi;
  range(id.y, sh, dy.h, jlo, jhi);
  range(id.x, sw, dy.w, klo, khi);
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int j = jlo; j <= jhi; ++j) {
    const Lerp ly = lerp_of(j, sh, dx.h);
    const float wy = (ly.i0 == id.y ? ly.w0 : 0.f) + (ly.i1 == id.y ? ly.w1 : 0.f);
    if (ly.i0 != id.y && ly.i1 != id.y) continue;
    for (int k = klo; k <= khi; ++k) {
      const Lerp lx = lerp_of(k, sw, dx.w);
      if (lx.i0 != id.x && lx.i1 != id.x) continue;
      const float wx = (lx.i0 == id.x ? lx.w0 : 0.f) + (lx.i1 == id.x ? lx.w1 : 0.f);
      float g[VEC];
      VecIO<T, VEC>::load(reinterpret_cast<const T*>(dy.ptr) + voff(dy, id.n, j, k) + id.cv * VEC, g);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(g[i], wy * wx, acc[i]);
    }
  }
  T* op = reinterpret_cast<T*>(dx.ptr) + voff(dx, id.n, id.y, id.x) + id.cv * VEC;
  if (accumulate) {
    float old[VEC];
    VecIO<T, VEC>::load(op, old);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += old[i];
  }
  VecIO<T, VEC>::store(op, acc);
}

// The same gather with L lanes per dx pixel (the rows of the candidate window dealt over them, partial sums folded by a butterfly): for a
// steep up-sampling -- the classifier's logits, 84 x 84 -> 256 x 256: ~10 x 10 candidates, ~38 loads per dx pixel -- on a tensor too small to fill
// the chip with one thread per pixel (113 k threads walking their windows one load after the other: 21.8 us between the JSD join and the first
// MFMA kernel of the backward pass, where nothing else runs).
template <typename T, int VEC, int L>
__global__ __launch_bounds__(256) void bilinear_bwd_split_kernel(View dy, View dx, float sh, float sw, int accumulate) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int sub = (int)(threadIdx.x & (L - 1));
  const PixIdx id = decode(t / L, dx.n, dx.h, dx.w, dx.c / VEC);      // a group's lanes are all in or all out (256 % L == 0)
  auto range = [](int i, float scale, int out_size, int& lo, int& hi) {
    lo = max(0, (int)floorf((float)(i - 1) / scale) - 1);
    hi = min(out_size - 1, (int)ceilf((float)(i + 1) / scale) + 1);
  };
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  if (id.ok) {
    int jlo, jhi, klo, khi;
    range(id.y, sh, dy.h, jlo, jhi);
    range(id.x, sw, dy.w, klo, khi);
    for (int j = jlo + sub; j <= jhi; j += L) {
      const Lerp ly = lerp_of(j, sh, dx.h);
      if (ly.i0 != id.y && ly.i1 != id.y) continue;
      const float wy = (ly.i0 == id.y ? ly.w0 : 0.f) + (ly.i1 == id.y ? ly.w1 : 0.f);
      for (int k = klo; k <= khi; ++k) {
        const Lerp lx = lerp_of(k, sw, dx.w);
        if (lx.i0 != id.x && lx.i1 != id.x) continue;
        const float wx = (lx.i0 == id.x ? lx.w0 : 0.f) + (lx.i1 == id.x ? lx.w1 : 0.f);
        float g[VEC];
        VecIO<T, VEC>::load(reinterpret_cast<const T*>(dy.ptr) + voff(dy, id.n, j, k) + id.cv * VEC, g);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(g[i], wy * wx, acc[i]);
      }
    }
  }
#pragma unroll
  for (int m = 1; m < L; m <<= 1)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], m);
  if (!id.ok || sub) return;
  T* op = reinterpret_cast<T*>(dx.ptr) + voff(dx, id.n, id.y, id.x) + id.cv * VEC;
  if (accumulate) {
    float old[VEC];
    VecIO<T, VEC>::load(op, old);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += old[i];
  }
  VecIO<T, VEC>::store(op, acc);
}

// Un-pooling with the skip connection's gradient gathered on the way (dct_maxpool2x2_bwd_codes_skip): the pooled tensor p feeds the next
// encoder block AND, bilinearly resized, the decoder's concatenation, so its gradient is dp = (data gradient of the next block) + (bilinear
// backward of the concatenation's gradient).  The sum used to be formed in memory (the bilinear backward wrote dp, the data gradient
// read it back and added); here the data gradient writes dp alone and this kernel adds the gather -- the loop of bilinear_bwd_kernel, same
// index arithmetic -- to the value it routes: one write and one read of dp less per level, and the sum is not rounded to 16 bits on the way.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_codes_skip_kernel(const unsigned char* __restrict__ codes, View dy, View skip, View dx, int relu_mask,
                                                                     float scale, float sh, float sw) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const PixIdx id = decode(t, dy.n, dy.h, dy.w, dy.c / VEC);
  if (!id.ok) return;
  float g[VEC];
  VecIO<T, VEC>::load(reinterpret_cast<const T*>(dy.ptr) + voff(dy, id.n, id.y, id.x) + id.cv * VEC, g);
  {
    auto range = [](int i, float sc, int out_size, int& lo, int& hi) {
      if (sc <= 0.f) { lo = 0; hi = out_size - 1; return; }
      lo = max(0, (int)floorf((float)(i - 1) / sc) - 1);
      hi = min(out_size - 1, (int)ceilf((float)(i + 1) / sc) + 1);
    };
    int jlo, jhi, klo, khi;
    range(id.y, sh, skip.h, jlo, jhi);
    range(id.x, sw, skip.w, klo, khi);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    for (int j = jlo; j <= jhi; ++j) {
      const Lerp ly = lerp_of(j, sh, dy.h);
      if (ly.i0 != id.y && ly.i1 != id.y) continue;
      const float wy = (ly.i0 == id.y ? ly.w0 : 0.f) + (ly.i1 == id.y ? ly.w1 : 0.f);
      for (int k = klo; k <= khi; ++k) {
        const Lerp lx = lerp_of(k, sw, dy.w);
        if (lx.i0 != id.x && lx.i1 != id.x) continue;
        const float wx = (lx.i0 == id.x ? lx.w0 : 0.f) + (lx.i1 == id.x ? lx.w1 : 0.f);
        float s[VEC];
        VecIO<T, VEC>::load(reinterpret_cast<const T*>(skip.ptr) + voff(skip, id.n, j, k) + id.cv * VEC, s);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(s[i], wy * wx, acc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] += acc[i];
  }
  union { unsigned char b[VEC]; unsigned w[VEC / 4]; } cu;
  const unsigned* cp = reinterpret_cast<const unsigned*>(codes + t * VEC);
  if constexpr (VEC == 8) { const uint2 q = *reinterpret_cast<const uint2*>(cp); cu.w[0] = q.x; cu.w[1] = q.y; }
  else cu.w[0] = cp[0];
  const unsigned char* cd = cu.b;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    if (relu_mask) g[i] = (cd[i] & 4) ? g[i] * scale : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int iy = 2 * id.y + (k >> 1), ix = 2 * id.x + (k & 1);
    if (iy >= dx.h || ix >= dx.w) continue;
    float out[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = (cd[i] & 11) == k ? g[i] : 0.f;
    VecIO<T, VEC>::store(reinterpret_cast<T*>(dx.ptr) + voff(dx, id.n, iy, ix) + id.cv * VEC, out);
  }
}

// ---- dropout (Philox4x32-10) ---------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned long long seed, unsigned long long ctr, unsigned out[4]) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// thread per 4 consecutive channels of a pixel
// calls (nullable): device-resident call counter in TWO 64-bit words that successive launches use in turn.  This launch reads calls[parity] = n, is
// call number n + 1 -- the Philox offset is that << 40 (what the host passes as `offset` otherwise), so a captured HIP graph draws a fresh mask on
// every replay -- and its first thread stores n + 1 to calls[parity ^ 1], the word no block of this launch reads; the next launch comes with the
// other parity.  (Until round 5 a one-thread kernel in front did the increment: two launches per dropout site on every model's chain, where a small
// launch costs the step ~3 us.  One word + a last-block ticket was tried first: 5000 same-address atomics per launch cost 60 us.)
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(View x, View y, unsigned char* mask_out, const unsigned char* mask_in,
                                                       float p, unsigned long long seed, unsigned long long offset,
                                                       unsigned long long* calls, int parity) {
  if (calls) {
    const unsigned long long call = calls[parity] + 1ull;
    offset = call << 40;
    if (blockIdx.x == 0 && threadIdx.x == 0) calls[parity ^ 1] = call;
  }
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const PixIdx id = decode(idx, x.n, x.h, x.w, x.c / 4);
  if (!id.ok) return;
  const float scale = 1.f / (1.f - p);
  bool keep[4];
  const long long dense = (((long long)id.n * x.h + id.y) * x.w + id.x) * x.c + id.cv * 4;
  if (mask_in) {
#pragma unroll
    for (int i = 0; i < 4; ++i) keep[i] = mask_in[dense + i] != 0;
  } else {
    unsigned r[4];
    philox4x32_10(seed, offset + (unsigned long long)idx, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) keep[i] = (float)(r[i] >> 8) * (1.0f / 16777216.0f) >= p;
  }
  const T* xp = reinterpret_cast<const T*>(x.ptr) + voff(x, id.n, id.y, id.x) + id.cv * 4;
  T* yp = reinterpret_cast<T*>(y.ptr) + voff(y, id.n, id.y, id.x) + id.cv * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    yp[i] = from_f32<T>(keep[i] ? to_f32(xp[i]) * scale : 0.f);
    if (mask_out) mask_out[dense + i] = keep[i] ? 1 : 0;
  }
}

// Dropout + the 2 x 2 max-pool behind it in ONE pass (dct_dropout_maxpool2x2_fwd_codes: the fourth encoder level of a training pass -- conv, ReLU,
// dropout, pool): thread per (pooled pixel, VEC channels); every window element gets dropout_kernel's mask (same Philox counter: the element's index
// in the dense [n][h][w][c / 4] walk), value and rounding, then maxpool_fwd_codes_kernel's scan.  The dropped full-resolution tensor is never
// written (the backward pass routes by the codes); one launch and 2 x its bytes less on every model's chain.  Bit-identical to the two launches.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void dropout_maxpool_fwd_codes_kernel(View x, View y, unsigned char* __restrict__ codes, float p, unsigned long long seed,
                                                                         unsigned long long* calls, int parity) {
  const unsigned long long call = calls[parity] + 1ull;
  const unsigned long long offset = call << 40;
  if (blockIdx.x == 0 && threadIdx.x == 0) calls[parity ^ 1] = call;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const PixIdx id = decode(t, y.n, y.h, y.w, y.c / VEC);
  if (!id.ok) return;
  const float scale = 1.f / (1.f - p);
  const int groups = x.c / 4;
  float m[VEC];
  int arg[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { m[i] = -INFINITY; arg[i] = 8; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int iy = 2 * id.y + (k >> 1), ix = 2 * id.x + (k & 1);
    if (iy < x.h && ix < x.w) {
      float v[VEC];
      VecIO<T, VEC>::load(reinterpret_cast<const T*>(x.ptr) + voff(x, id.n, iy, ix) + id.cv * VEC, v);
      const long long pix = ((long long)id.n * x.h + iy) * x.w + ix;
#pragma unroll
      for (int g4 = 0; g4 < VEC / 4; ++g4) {
        unsigned r[4];
        philox4x32_10(seed, offset + (unsigned long long)(pix * groups + id.cv * (VEC / 4) + g4), r);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool keep = (float)(r[i] >> 8) * (1.0f / 16777216.0f) >= p;
          const float d = to_f32(from_f32<T>(keep ? v[g4 * 4 + i] * scale : 0.f));          // the dropped tensor's stored value
          if (d > m[g4 * 4 + i]) { m[g4 * 4 + i] = d; arg[g4 * 4 + i] = k; }
        }
      }
    }
  }
  VecIO<T, VEC>::store(reinterpret_cast<T*>(y.ptr) + voff(y, id.n, id.y, id.x) + id.cv * VEC, m);
  union { unsigned char b[VEC]; unsigned w[VEC / 4]; } cd;             // dense [n][h][w][c], same walk as `decode`
#pragma unroll
  for (int i = 0; i < VEC; ++i) cd.b[i] = (unsigned char)(arg[i] | (m[i] > 0.f ? 4 : 0));
  unsigned* cp = reinterpret_cast<unsigned*>(codes + t * VEC);
  if constexpr (VEC == 8) *reinterpret_cast<uint2*>(cp) = make_uint2(cd.w[0], cd.w[1]);
  else cp[0] = cd.w[0];
}

// ---- relu backward / cast --------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(256) void relu_bwd_kernel(View g, View a, View y, float scale) {
  const PixIdx id = decode((long long)blockIdx.x * 256 + threadIdx.x, y.n, y.h, y.w, y.c / VEC);
  if (!id.ok) return;
  float gv[VEC], av[VEC];
  VecIO<T, VEC>::load(reinterpret_cast<const T*>(g.ptr) + voff(g, id.n, id.y, id.x) + id.cv * VEC, gv);
  VecIO<T, VEC>::load(reinterpret_cast<const T*>(a.ptr) + voff(a, id.n, id.y, id.x) + id.cv * VEC, av);
#pragma unroll
  for (int i = 0; i < VEC; ++i) gv[i] = av[i] > 0.f ? gv[i] * scale : 0.f;
  VecIO<T, VEC>::store(reinterpret_cast<T*>(y.ptr) + voff(y, id.n, id.y, id.x) + id.cv * VEC, gv);
}
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(View x, View y) {
  const PixIdx id = decode((long long)blockIdx.x * 256 + threadIdx.x, y.n, y.h, y.w, y.c);
  if (!id.ok) return;
  reinterpret_cast<TO*>(y.ptr)[voff(y, id.n, id.y, id.x) + id.cv] =
      from_f32<TO>(to_f32(reinterpret_cast<const TI*>(x.ptr)[voff(x, id.n, id.y, id.x) + id.cv]));
}

static inline bool same_nhw(const dct_view* a, const dct_view* b) { return a->n == b->n && a->h == b->h && a->w == b->w; }
static inline bool vec_ok(const dct_view* v, int vec, int esz) {
  return v->c % vec == 0 && v->sw % vec == 0 && v->sh % vec == 0 && v->sn % vec == 0 && ((uintptr_t)v->ptr % (vec * esz)) == 0;
}
static inline StemGeom stem_geom(const dct_conv_desc* d, int cout) {
  StemGeom g; g.R = d->R; g.S = d->S; g.stride = d->stride; g.dil = d->dil; g.pad_h = d->pad_h; g.pad_w = d->pad_w;
  g.relu = d->relu; g.cout = cout; return g;
}
static inline int pix_blocks(long long P, int& ppb, int target_ppb, int max_blocks) {
  long long blocks = (P + target_ppb - 1) / target_ppb;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks < 1) blocks = 1;
  ppb = (int)((P + blocks - 1) / blocks);
  return (int)((P + ppb - 1) / ppb);
}

}  // namespace

#define DISPATCH_T(dtype, ...)                          \
  do {                                                  \
    if ((dtype) == DCT_BF16) { using T = bf16_t; constexpr int VEC = 8; (void)VEC; __VA_ARGS__; } \
    else { using T = float; constexpr int VEC = 4; (void)VEC; __VA_ARGS__; }                    \
  } while (0)

extern "C" int dct_pack_weight(const float* src, void* dst, int P, int T_, int Q, int transpose, int flip_taps,
                               int dtype, dct_stream stream) {
  if (!src || !dst || P < 1 || T_ < 1 || Q < 1) return DCT_ERR_BAD_ARG;
  const long long total = (long long)P * T_ * Q;
  hipStream_t st = (hipStream_t)stream;
  if (transpose == 1 && P % 32 == 0 && Q % 32 == 0) {
    const dim3 grid(P / 32, Q / 32, T_);
    if (dtype == DCT_BF16) DCT_LAUNCH(DCT_PROF_POINTWISE, pack_transpose_kernel<bf16_t>, grid, dim3(256), 0, st, src, (bf16_t*)dst, P, T_, Q, flip_taps);
    else if (dtype == DCT_F32) DCT_LAUNCH(DCT_PROF_POINTWISE, pack_transpose_kernel<float>, grid, dim3(256), 0, st, src, (float*)dst, P, T_, Q, flip_taps);
    else return DCT_ERR_BAD_ARG;
    return dct_check_launch();
  }
  if (dtype == DCT_BF16) DCT_LAUNCH(DCT_PROF_POINTWISE, pack_kernel<bf16_t>, dim3(div_up(total, 256)), dim3(256), 0, st, src, (bf16_t*)dst, P, T_, Q, transpose, flip_taps);
  else if (dtype == DCT_F32) DCT_LAUNCH(DCT_PROF_POINTWISE, pack_kernel<float>, dim3(div_up(total, 256)), dim3(256), 0, st, src, (float*)dst, P, T_, Q, transpose, flip_taps);
  else return DCT_ERR_BAD_ARG;
  return dct_check_launch();
}

extern "C" int dct_pack_weights_batched(const void* jobs_dev, int njobs, int total_tiles, int dtype, dct_stream stream) {
  if (!jobs_dev || njobs < 1 || total_tiles < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCT_BF16) DCT_LAUNCH(DCT_PROF_POINTWISE, pack_transpose_batched_kernel<bf16_t>, dim3(total_tiles), dim3(256), 0, st, (const PackJob*)jobs_dev, njobs);
  else if (dtype == DCT_F32) DCT_LAUNCH(DCT_PROF_POINTWISE, pack_transpose_batched_kernel<float>, dim3(total_tiles), dim3(256), 0, st, (const PackJob*)jobs_dev, njobs);
  else return DCT_ERR_BAD_ARG;
  return dct_check_launch();
}

extern "C" int dct_pack_weights_batched64(const void* jobs_dev, int njobs, int total_tiles, dct_stream stream) {
  if (!jobs_dev || njobs < 1 || total_tiles < 1) return DCT_ERR_BAD_ARG;
  DCT_LAUNCH(DCT_PROF_POINTWISE, pack_transpose_batched64_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs_dev, njobs);
  return dct_check_launch();
}

void dct_relu_bits_launch(const void* y_bf16, unsigned char* bits, long long chunks, hipStream_t st);   // igemm.hip

extern "C" int dct_conv_cin1_fwd(const dct_view* x, const float* w, const float* bias, const dct_view* y,
                                 const dct_conv_desc* d, int dtype, dct_stream stream) {
  if (!view_ok(x) || !view_ok(y) || !w || !d || x->c != 1 || x->n != y->n) return DCT_ERR_BAD_ARG;
  if (d->R * d->S > 25 || y->c > 512) return DCT_ERR_UNSUPPORTED;
  const StemGeom g = stem_geom(d, y->c);
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = (size_t)(y->c * d->R * d->S + y->c) * sizeof(float);
  unsigned char* bits = d->relu_bits_out;
  bool bits_done = false;
  if (bits && (dtype != DCT_BF16 || !d->relu || y->c % 8 || y->sw != y->c || y->sh != (long long)y->w * y->c ||
               y->sn != (long long)y->h * y->w * y->c)) return DCT_ERR_BAD_ARG;
  DISPATCH_T(dtype, {
    if (!vec_ok(y, 1, sizeof(T)) || (y->sw % VEC) || (y->sh % VEC) || (y->sn % VEC) || ((uintptr_t)y->ptr % 16)) return DCT_ERR_UNSUPPORTED;
    const long long total = (long long)y->n * y->h * y->w * ((y->c + VEC - 1) / VEC);
    if (d->R == 3 && d->S == 3 && d->stride == 1 && d->dil == 1 && y->c % VEC == 0) {
      constexpr int PX = 8;
      if (3 * x->sh + (PX + 2) * x->sw > 0x7fffffffll || (long long)PX * y->sw > 0x7fffffffll) return DCT_ERR_UNSUPPORTED;
      const long long groups = (long long)y->n * y->h * ((y->w + PX - 1) / PX) * (y->c / VEC);
      DCT_LAUNCH(DCT_PROF_POINTWISE, (stem_fwd3x3_kernel<T, VEC, PX>), dim3(div_up(groups, 256)), dim3(256), sh, st, to_view(x), w, bias, to_view(y), g,
                 VEC == 8 ? bits : nullptr);
      bits_done = VEC == 8;
    } else if (d->R == 3 && d->S == 3)
      DCT_LAUNCH(DCT_PROF_POINTWISE, (stem_fwd_kernel<T, VEC, 3>), dim3(div_up(total, 256)), dim3(256), sh, st, to_view(x), w, bias, to_view(y), g);
    else
      DCT_LAUNCH(DCT_PROF_POINTWISE, (stem_fwd_kernel<T, VEC, 0>), dim3(div_up(total, 256)), dim3(256), sh, st, to_view(x), w, bias, to_view(y), g);
  });
  if (bits && !bits_done) dct_relu_bits_launch(y->ptr, bits, (long long)y->n * y->h * y->w * (y->c / 8), st);
  return dct_check_launch();
}

int g_stem_dgrad_mfma = 1;      // diagnostic (dct_tune_set 1008): 0 = the vector-ALU kernel for every shape
extern "C" int dct_conv_cin1_dgrad(const dct_view* dy, const float* w, const dct_view* dx,
                                   const dct_conv_desc* d, int dtype, dct_stream stream) {
  if (!view_ok(dy) || !view_ok(dx) || !w || !d || dx->c != 1 || dx->n != dy->n) return DCT_ERR_BAD_ARG;
  const StemGeom g = stem_geom(d, dy->c);
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = (size_t)(dy->c * d->R * d->S) * sizeof(float);
  if (dtype == DCT_BF16 && dy->c == 64 && d->R == 3 && d->S == 3 && d->stride == 1 && d->dil == 1 && d->pad_h == 0 && d->pad_w == 0 &&
      dx->h == dy->h + 2 && dx->w == dy->w + 2 && vec_ok(dy, 8, 2) && g_stem_dgrad_mfma) {
    const int tiles_x = (dx->w + 15) / 16, tiles_y = (dx->h + 15) / 16;
    DCT_LAUNCH(DCT_PROF_POINTWISE, stem_dgrad_mfma_kernel, dim3((unsigned)(dx->n * tiles_y * tiles_x)), dim3(256), 0, st, to_view(dy), w, to_view(dx),
               tiles_x, tiles_y);
    return dct_check_launch();
  }
  DISPATCH_T(dtype, {
    const int cv = dy->c / VEC;
    if (dy->c % VEC || cv > 64 || (cv & (cv - 1)) || !vec_ok(dy, VEC, sizeof(T))) return DCT_ERR_UNSUPPORTED;
    const long long total = (long long)dx->n * dx->h * dx->w * cv;
    DCT_LAUNCH(DCT_PROF_POINTWISE, (stem_dgrad_kernel<T, VEC>), dim3(div_up(total, 256)), dim3(256), sh, st, to_view(dy), w, to_view(dx), g);
  });
  return dct_check_launch();
}

extern "C" int dct_conv1x1_head_fwd(const dct_view* x, const float* w, const float* bias, const dct_view* y,
                                    int dtype, dct_stream stream) {
  if (!view_ok(x) || !view_ok(y) || !w || !same_nhw(x, y)) return DCT_ERR_BAD_ARG;
  if (y->c > 8 || x->c > 1024) return DCT_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = (size_t)(y->c * x->c + y->c) * sizeof(float);
  DISPATCH_T(dtype, {
    if (!vec_ok(x, VEC, sizeof(T))) return DCT_ERR_UNSUPPORTED;
    const long long total = (long long)x->n * x->h * x->w;
    const int lanes = x->c / VEC;
    int lg = 0;
    while ((1 << lg) < lanes) ++lg;
    if (x->c % VEC == 0 && (1 << lg) == lanes && lanes >= 2 && lanes <= 64 && lanes >= y->c)
      DCT_LAUNCH(DCT_PROF_POINTWISE, (head_fwd_lanes_kernel<T, VEC>), dim3(div_up(total * lanes, 256)), dim3(256), sh, st, to_view(x), w, bias, to_view(y), lg);
    else
      DCT_LAUNCH(DCT_PROF_POINTWISE, (head_fwd_kernel<T, VEC>), dim3(div_up(total, 256)), dim3(256), sh, st, to_view(x), w, bias, to_view(y));
  });
  return dct_check_launch();
}

int dct_head_dw_launch(const dct_view* x, const dct_view* dy, float* dw, float* db, int accumulate, int dtype,
                       void* workspace, size_t workspace_bytes, hipStream_t st);  // reduce.hip

extern "C" int dct_conv1x1_head_bwd(const dct_view* x, const dct_view* dy, const float* w, const dct_view* dx,
                                    float* dw, float* db, int relu_mask, int accumulate, int dtype,
                                    void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (!view_ok(x) || !view_ok(dy) || !w || !same_nhw(x, dy)) return DCT_ERR_BAD_ARG;
  if (dy->c > 8) return DCT_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = (size_t)(dy->c * x->c) * sizeof(float);
  if (dx) {
    if (!view_ok(dx) || !same_nhw(x, dx) || dx->c != x->c) return DCT_ERR_BAD_ARG;
    DISPATCH_T(dtype, {
      if (!vec_ok(x, VEC, sizeof(T)) || !vec_ok(dx, VEC, sizeof(T))) return DCT_ERR_UNSUPPORTED;
      const long long total = (long long)x->n * x->h * x->w * (x->c / VEC);
      DCT_LAUNCH(DCT_PROF_POINTWISE, (head_dx_kernel<T, VEC>), dim3(div_up(total, 256)), dim3(256), sh, st, to_view(x), to_view(dy), w, to_view(dx), relu_mask);
    });
  }
  if (dw || db) {
    const int rc = dct_head_dw_launch(x, dy, dw, db, accumulate, dtype, workspace, workspace_bytes, st);
    if (rc != DCT_OK) return rc;
  }
  return dct_check_launch();
}

extern "C" int dct_maxpool2x2_fwd(const dct_view* x, const dct_view* y, int dtype, dct_stream stream) {
  if (!view_ok(x) || !view_ok(y) || x->n != y->n || x->c != y->c) return DCT_ERR_BAD_ARG;
  if (y->h != (x->h + 1) / 2 || y->w != (x->w + 1) / 2) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, {
    if (!vec_ok(x, VEC, sizeof(T)) || !vec_ok(y, VEC, sizeof(T))) return DCT_ERR_UNSUPPORTED;
    const long long total = (long long)y->n * y->h * y->w * (y->c / VEC);
    DCT_LAUNCH(DCT_PROF_POINTWISE, (maxpool_fwd_kernel<T, VEC>), dim3(div_up(total, 256)), dim3(256), 0, st, to_view(x), to_view(y));
  });
  return dct_check_launch();
}
extern "C" int dct_maxpool2x2_bwd(const dct_view* x, const dct_view* dy, const dct_view* dx, int relu_mask,
                                  float scale, int dtype, dct_stream stream) {
  if (!view_ok(x) || !view_ok(dy) || !view_ok(dx) || !same_nhw(x, dx) || x->c != dx->c || x->c != dy->c || x->n != dy->n) return DCT_ERR_BAD_ARG;
  if (dy->h != (x->h + 1) / 2 || dy->w != (x->w + 1) / 2) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, {
    if (!vec_ok(x, VEC, sizeof(T)) || !vec_ok(dy, VEC, sizeof(T)) || !vec_ok(dx, VEC, sizeof(T))) return DCT_ERR_UNSUPPORTED;
    const long long total = (long long)dy->n * dy->h * dy->w * (x->c / VEC);
    DCT_LAUNCH(DCT_PROF_POINTWISE, (maxpool_bwd_kernel<T, VEC>), dim3(div_up(total, 256)), dim3(256), 0, st, to_view(x), to_view(dy), to_view(dx), relu_mask, scale);
  });
  return dct_check_launch();
}

extern "C" int dct_maxpool2x2_fwd_codes(const dct_view* x, const dct_view* y, uint8_t* codes, int dtype, dct_stream stream) {
  if (!view_ok(x) || !view_ok(y) || !codes || x->n != y->n || x->c != y->c) return DCT_ERR_BAD_ARG;
  if (y->h != (x->h + 1) / 2 || y->w != (x->w + 1) / 2) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, {
    if (!vec_ok(x, VEC, sizeof(T)) || !vec_ok(y, VEC, sizeof(T)) || ((uintptr_t)codes % VEC)) return DCT_ERR_UNSUPPORTED;
    const long long total = (long long)y->n * y->h * y->w * (y->c / VEC);
    DCT_LAUNCH(DCT_PROF_POINTWISE, (maxpool_fwd_codes_kernel<T, VEC>), dim3(div_up(total, 256)), dim3(256), 0, st, to_view(x), to_view(y), codes);
  });
  return dct_check_launch();
}
extern "C" int dct_maxpool2x2_bwd_codes(const uint8_t* codes, const dct_view* dy, const dct_view* dx, int relu_mask,
                                        float scale, int dtype, dct_stream stream) {
  if (!codes || !view_ok(dy) || !view_ok(dx) || dy->c != dx->c || dx->n != dy->n) return DCT_ERR_BAD_ARG;
  if (dy->h != (dx->h + 1) / 2 || dy->w != (dx->w + 1) / 2) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, {
    if (!vec_ok(dy, VEC, sizeof(T)) || !vec_ok(dx, VEC, sizeof(T)) || ((uintptr_t)codes % VEC)) return DCT_ERR_UNSUPPORTED;
    const long long total = (long long)dy->n * dy->h * dy->w * (dy->c / VEC);
    DCT_LAUNCH(DCT_PROF_POINTWISE, (maxpool_bwd_codes_kernel<T, VEC>), dim3(div_up(total, 256)), dim3(256), 0, st, codes, to_view(dy), to_view(dx), relu_mask, scale);
  });
  return dct_check_launch();
}

static inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

extern "C" int dct_maxpool2x2_bwd_codes_skip(const uint8_t* codes, const dct_view* dy, const dct_view* skip, const dct_view* dx, int relu_mask,
                                             float scale, int dtype, dct_stream stream) {
  if (!codes || !view_ok(dy) || !view_ok(skip) || !view_ok(dx) || dy->c != dx->c || dx->n != dy->n || skip->n != dy->n || skip->c != dy->c)
    return DCT_ERR_BAD_ARG;
  if (dy->h != (dx->h + 1) / 2 || dy->w != (dx->w + 1) / 2) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const float sh = ac_scale(dy->h, skip->h), sw = ac_scale(dy->w, skip->w);
  DISPATCH_T(dtype, {
    if (!vec_ok(dy, VEC, sizeof(T)) || !vec_ok(dx, VEC, sizeof(T)) || !vec_ok(skip, VEC, sizeof(T)) || ((uintptr_t)codes % VEC)) return DCT_ERR_UNSUPPORTED;
    const long long total = (long long)dy->n * dy->h * dy->w * (dy->c / VEC);
    DCT_LAUNCH(DCT_PROF_POINTWISE, (maxpool_bwd_codes_skip_kernel<T, VEC>), dim3(div_up(total, 256)), dim3(256), 0, st, codes, to_view(dy), to_view(skip),
               to_view(dx), relu_mask, scale, sh, sw);
  });
  return dct_check_launch();
}

extern "C" int dct_bilinear_fwd(const dct_view* x, const dct_view* y, int dtype_in, int dtype_out, dct_stream stream) {
  if (!view_ok(x) || !view_ok(y) || x->n != y->n || x->c != y->c) return DCT_ERR_BAD_ARG;
  if (dtype_in != dtype_out) return DCT_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const float sh = ac_scale(x->h, y->h), sw = ac_scale(x->w, y->w);
  DISPATCH_T(dtype_in, {
    const long long px = (long long)y->n * y->h * y->w;
    if (vec_ok(x, VEC, sizeof(T)) && vec_ok(y, VEC, sizeof(T))) {
      // ONE kernel for the single and the batched call (a job table of one): two instantiations of the same source came out with different
      // FMA contractions of the interpolation weights, i.e. results a last bit apart
      BilJobs js;
      js.n = 1;
      js.j[0].x = to_view(x); js.j[0].y = to_view(y); js.j[0].sh = sh; js.j[0].sw = sw; js.j[0].blk0 = 0; js.j[0].pad = 0;
      DCT_LAUNCH(DCT_PROF_POINTWISE, (bilinear_fwd_batched_kernel<T, VEC>), dim3(div_up(px * (y->c / VEC), 256)), dim3(256), 0, st, js);
    } else
      DCT_LAUNCH(DCT_PROF_POINTWISE, (bilinear_fwd_kernel<T, 1>), dim3(div_up(px * y->c, 256)), dim3(256), 0, st, to_view(x), to_view(y), sh, sw);
  });
  return dct_check_launch();
}
extern "C" int dct_bilinear_fwd_batched(const dct_view* xs, const dct_view* ys, int n, int dtype, dct_stream stream) {
  if (!xs || !ys || n < 1) return DCT_ERR_BAD_ARG;
  if (n > kBilJobs) return DCT_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, {
    BilJobs js;
    js.n = n;
    long long blk = 0;
    for (int k = 0; k < n; ++k) {
      const dct_view* x = xs + k; const dct_view* y = ys + k;
      if (!view_ok(x) || !view_ok(y) || x->n != y->n || x->c != y->c) return DCT_ERR_BAD_ARG;
      if (!vec_ok(x, VEC, sizeof(T)) || !vec_ok(y, VEC, sizeof(T))) return DCT_ERR_UNSUPPORTED;      // (the scalar form: one dct_bilinear_fwd per tensor)
      BilJob& jb = js.j[k];
      jb.x = to_view(x); jb.y = to_view(y);
      jb.sh = ac_scale(x->h, y->h); jb.sw = ac_scale(x->w, y->w);
      jb.blk0 = (int)blk; jb.pad = 0;
      blk += div_up((long long)y->n * y->h * y->w * (y->c / VEC), 256);
      if (blk > 0x7fffffffll) return DCT_ERR_UNSUPPORTED;
    }
    DCT_LAUNCH(DCT_PROF_POINTWISE, (bilinear_fwd_batched_kernel<T, VEC>), dim3((unsigned)blk), dim3(256), 0, st, js);
  });
  return dct_check_launch();
}
extern "C" int dct_bilinear_bwd(const dct_view* dy, const dct_view* dx, int dtype_dy, int dtype_dx, int accumulate,
                                dct_stream stream) {
  if (!view_ok(dy) || !view_ok(dx) || dx->n != dy->n || dx->c != dy->c) return DCT_ERR_BAD_ARG;
  if (dtype_dy != dtype_dx) return DCT_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const float sh = ac_scale(dx->h, dy->h), sw = ac_scale(dx->w, dy->w);
  DISPATCH_T(dtype_dy, {
    const long long px = (long long)dx->n * dx->h * dx->w;
    const bool vec = vec_ok(dx, VEC, sizeof(T)) && vec_ok(dy, VEC, sizeof(T));
    // steep up-sampling on a small tensor (the classifier's logits): eight lanes per dx pixel
    if (vec && sh > 0.f && sw > 0.f && sh < 0.5f && sw < 0.5f && px * (dx->c / VEC) < 262144)
      DCT_LAUNCH(DCT_PROF_POINTWISE, (bilinear_bwd_split_kernel<T, VEC, 8>), dim3(div_up(px * (dx->c / VEC) * 8, 256)), dim3(256), 0, st, to_view(dy), to_view(dx), sh, sw, accumulate);
    else if (vec)
      DCT_LAUNCH(DCT_PROF_POINTWISE, (bilinear_bwd_kernel<T, VEC>), dim3(div_up(px * (dx->c / VEC), 256)), dim3(256), 0, st, to_view(dy), to_view(dx), sh, sw, accumulate);
    else
      DCT_LAUNCH(DCT_PROF_POINTWISE, (bilinear_bwd_kernel<T, 1>), dim3(div_up(px * dx->c, 256)), dim3(256), 0, st, to_view(dy), to_view(dx), sh, sw, accumulate);
  });
  return dct_check_launch();
}

static int dropout_impl(const dct_view* x, const dct_view* y, uint8_t* mask_out, const uint8_t* mask_in, float p,
                        uint64_t seed, uint64_t offset, int dtype, dct_stream stream, uint64_t* calls = nullptr, int parity = 0) {
  if (!view_ok(x) || !view_ok(y) || !same_nhw(x, y) || x->c != y->c || p < 0.f || p >= 1.f) return DCT_ERR_BAD_ARG;
  if (x->c % 4) return DCT_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)x->n * x->h * x->w * (x->c / 4);
  DISPATCH_T(dtype, {
    DCT_LAUNCH(DCT_PROF_POINTWISE, dropout_kernel<T>, dim3(div_up(total, 256)), dim3(256), 0, st, to_view(x), to_view(y),
               (unsigned char*)mask_out, (const unsigned char*)mask_in, p, (unsigned long long)seed, (unsigned long long)offset,
               (unsigned long long*)calls, parity & 1);
  });
  return dct_check_launch();
}
extern "C" int dct_dropout_fwd(const dct_view* x, const dct_view* y, uint8_t* mask_out, float p,
                               uint64_t seed, uint64_t offset, int dtype, dct_stream stream) {
  return dropout_impl(x, y, mask_out, nullptr, p, seed, offset, dtype, stream);
}
extern "C" int dct_dropout_fwd_dev(const dct_view* x, const dct_view* y, uint8_t* mask_out, float p,
                                   uint64_t seed, uint64_t* calls, int parity, int dtype, dct_stream stream) {
  if (!calls || ((uintptr_t)calls & 7) || (parity & ~1)) return DCT_ERR_BAD_ARG;
  return dropout_impl(x, y, mask_out, nullptr, p, seed, 0, dtype, stream, calls, parity);
}
extern "C" int dct_dropout_maxpool2x2_fwd_codes(const dct_view* x, const dct_view* y, uint8_t* codes, float p, uint64_t seed, uint64_t* calls,
                                                int parity, int dtype, dct_stream stream) {
  if (!view_ok(x) || !view_ok(y) || !codes || x->n != y->n || x->c != y->c || p < 0.f || p >= 1.f) return DCT_ERR_BAD_ARG;
  if (y->h != (x->h + 1) / 2 || y->w != (x->w + 1) / 2) return DCT_ERR_BAD_ARG;
  if (!calls || ((uintptr_t)calls & 7) || (parity & ~1)) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, {
    if (!vec_ok(x, VEC, sizeof(T)) || !vec_ok(y, VEC, sizeof(T)) || ((uintptr_t)codes % VEC)) return DCT_ERR_UNSUPPORTED;
    const long long total = (long long)y->n * y->h * y->w * (y->c / VEC);
    DCT_LAUNCH(DCT_PROF_POINTWISE, (dropout_maxpool_fwd_codes_kernel<T, VEC>), dim3(div_up(total, 256)), dim3(256), 0, st, to_view(x), to_view(y), codes, p,
               (unsigned long long)seed, (unsigned long long*)calls, parity & 1);
  });
  return dct_check_launch();
}
extern "C" int dct_dropout_apply(const dct_view* x, const dct_view* y, const uint8_t* mask, float p, int dtype,
                                 dct_stream stream) {
  if (!mask) return DCT_ERR_BAD_ARG;
  return dropout_impl(x, y, nullptr, mask, p, 0, 0, dtype, stream);
}

extern "C" int dct_relu_bwd(const dct_view* g, const dct_view* a, const dct_view* y, float scale, int dtype,
                            dct_stream stream) {
  if (!view_ok(g) || !view_ok(a) || !view_ok(y) || !same_nhw(g, y) || !same_nhw(a, y) || g->c != y->c || a->c != y->c) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, {
    if (!vec_ok(g, VEC, sizeof(T)) || !vec_ok(a, VEC, sizeof(T)) || !vec_ok(y, VEC, sizeof(T))) return DCT_ERR_UNSUPPORTED;
    const long long total = (long long)y->n * y->h * y->w * (y->c / VEC);
    DCT_LAUNCH(DCT_PROF_POINTWISE, (relu_bwd_kernel<T, VEC>), dim3(div_up(total, 256)), dim3(256), 0, st, to_view(g), to_view(a), to_view(y), scale);
  });
  return dct_check_launch();
}

extern "C" int dct_cast(const dct_view* x, const dct_view* y, int dtype_in, int dtype_out, dct_stream stream) {
  if (!view_ok(x) || !view_ok(y) || !same_nhw(x, y) || x->c != y->c) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)y->n * y->h * y->w * y->c;
  const dim3 grid(div_up(total, 256)), blk(256);
  if (dtype_in == DCT_F32 && dtype_out == DCT_BF16) DCT_LAUNCH(DCT_PROF_POINTWISE, (cast_kernel<float, bf16_t>), grid, blk, 0, st, to_view(x), to_view(y));
  else if (dtype_in == DCT_BF16 && dtype_out == DCT_F32) DCT_LAUNCH(DCT_PROF_POINTWISE, (cast_kernel<bf16_t, float>), grid, blk, 0, st, to_view(x), to_view(y));
  else if (dtype_in == DCT_F32 && dtype_out == DCT_F32) DCT_LAUNCH(DCT_PROF_POINTWISE, (cast_kernel<float, float>), grid, blk, 0, st, to_view(x), to_view(y));
  else if (dtype_in == DCT_BF16 && dtype_out == DCT_BF16) DCT_LAUNCH(DCT_PROF_POINTWISE, (cast_kernel<bf16_t, bf16_t>), grid, blk, 0, st, to_view(x), to_view(y));
  else if (dtype_in == DCT_F32 && dtype_out == DCT_F16) DCT_LAUNCH(DCT_PROF_POINTWISE, (cast_kernel<float, f16_t>), grid, blk, 0, st, to_view(x), to_view(y));
  else if (dtype_in == DCT_F16 && dtype_out == DCT_F32) DCT_LAUNCH(DCT_PROF_POINTWISE, (cast_kernel<f16_t, float>), grid, blk, 0, st, to_view(x), to_view(y));
  else if (dtype_in == DCT_F16 && dtype_out == DCT_F16) DCT_LAUNCH(DCT_PROF_POINTWISE, (cast_kernel<f16_t, f16_t>), grid, blk, 0, st, to_view(x), to_view(y));
  else return DCT_ERR_BAD_ARG;
  return dct_check_launch();
}
