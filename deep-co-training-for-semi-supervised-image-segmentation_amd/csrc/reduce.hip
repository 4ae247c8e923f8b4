// Reductions over pixels (HBM-bound): bias gradient, Cin=1 stem weight gradient, classifier-head
// weight gradient.  All three stream a [pixels][C] activation once with 16-byte channel-vector
// loads (a wave reads whole 128-B+ channel runs of consecutive pixels), keep per-thread partial
// sums in registers, fold them across the block through shuffles/LDS and write one partial
// row per block; a second tiny kernel sums the block partials in fixed order (deterministic,
// "+=" into the existing gradient folded in).
//
// Thread layout: tid = row * CV + cv with CV = C / VEC channel vectors; `rows` = 256 / CV pixels
// are in flight per block iteration.
#include <algorithm>
#include "dct_common.h"

// blocks of stem_wgrad_mfma_kernel: one 16-wave block per CU.  Swept on 16 x 254 x 254 x 64 (whole call, fold included): 64 / 128 / 192 / 256 /
// 384 / 512 / 768 blocks -> 107 / 58 / 44 / 36 / 47 / 42 / 47 us (the VALU kernel: 78 us); four-wave blocks: 512-4096 blocks 50-104 us (the
// fold walks one partial row per block)
static const int g_stem_mfma_blocks = 256;

namespace {

template <typename T> struct Vec;
template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float* v) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
};
template <> struct Vec<float> {
  static constexpr int N = 4;
  __device__ static __forceinline__ void load(const float* p, float* v) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
};

// element offset of flat pixel index `pix` in an NHWC view
__device__ __forceinline__ long long pix_off(const View& v, long long pix, int linear) {
  if (linear) return pix * v.sw;
  const unsigned p = (unsigned)pix, row = p / (unsigned)v.w;        // pixel counts are < 2^31 (checked on the host)
  const unsigned x = p - row * (unsigned)v.w, n = row / (unsigned)v.h, y = row - n * (unsigned)v.h;
  return (long long)n * v.sn + (long long)y * v.sh + (long long)x * v.sw;
}
static inline int view_linear(const dct_view* v) {
  return (v->sh == (long long)v->w * v->sw && v->sn == (long long)v->h * v->sh) ? 1 : 0;
}

// ------------------------------------------------------------------------------- bias grad
// partial[blk][c] = sum over the block's pixels of dy[pix][c]
template <typename T>
__device__ __forceinline__ void bias_partial_body(const View& dy, float* partial, int ppb, int linear, int blk) {
  constexpr int VEC = Vec<T>::N;
  __shared__ float red[256 * VEC];
  const int C = dy.c;
  const int CV = C / VEC;               // <= 256, divides 256
  const int rows = 256 / CV;
  const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
  const long long P = (long long)dy.n * dy.h * dy.w;
  const long long pbeg = (long long)blk * ppb, pend = min(P, pbeg + ppb);
  const T* base = reinterpret_cast<const T*>(dy.ptr) + cv * VEC;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  long long pix = pbeg + row;
  for (; pix + 3 * rows < pend; pix += 4 * rows) {
    float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
    Vec<T>::load(base + pix_off(dy, pix, linear), v0);
    Vec<T>::load(base + pix_off(dy, pix + rows, linear), v1);
    Vec<T>::load(base + pix_off(dy, pix + 2 * rows, linear), v2);
    Vec<T>::load(base + pix_off(dy, pix + 3 * rows, linear), v3);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += (v0[i] + v1[i]) + (v2[i] + v3[i]);
  }
  for (; pix < pend; pix += rows) {
    float v0[VEC];
    Vec<T>::load(base + pix_off(dy, pix, linear), v0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += v0[i];
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) red[row * C + cv * VEC + i] = acc[i];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += red[r * C + c];
    partial[(long long)blk * C + c] = s;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bias_partial_kernel(View dy, float* partial, int ppb, int linear) {
  bias_partial_body<T>(dy, partial, ppb, linear, (int)blockIdx.x);
}
// Several bias gradients in one launch pair (dct_bias_grad_batched: a UNet's four up-convolutions -- 2 x 4 launches of ~5 us on each model's chain
// became 2): job k owns the blocks [blk0, blk0 + blocks) of the partial kernel and [red0, red0 + ceil(c / 16)) of the fold; per job the arithmetic
// and its order are those of the single call.
constexpr int kBiasJobs = 8;
struct BiasJob { View dy; float* partial; float* db; int ppb, linear, blk0, blocks, red0, pad; };
struct BiasJobs { BiasJob j[kBiasJobs]; int n; };
template <typename T>
__global__ __launch_bounds__(256) void bias_partial_batched_kernel(BiasJobs js) {
  int k = 0;
  for (int i = 1; i < js.n; ++i) k = (int)blockIdx.x >= js.j[i].blk0 ? i : k;
  const BiasJob& jb = js.j[k];
  bias_partial_body<T>(jb.dy, jb.partial, jb.ppb, jb.linear, (int)blockIdx.x - jb.blk0);
}

// out[i] (=|+=) sum_b partial[b][i]; 16 outputs per block, 16 strided partial sums each, folded in
// fixed order through LDS
__device__ __forceinline__ float fold16(const float* partial, int i, int n, int blocks, float* red) {
  const int part = threadIdx.x >> 4;
  float s = 0.f;
  if (i < n)
    for (int b = part; b < blocks; b += 16) s += partial[(long long)b * n + i];
  red[threadIdx.x] = s;
  __syncthreads();
  float t = 0.f;
  if (part == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k * 16 + threadIdx.x];
  }
  return t;
}
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* partial, float* out, int n, int blocks, int accumulate) {
  __shared__ float red[256];
  const int i = blockIdx.x * 16 + (threadIdx.x & 15);
  const float t = fold16(partial, i, n, blocks, red);
  if (threadIdx.x < 16 && i < n) out[i] = accumulate ? out[i] + t : t;
}

__global__ __launch_bounds__(256) void partial_reduce_batched_kernel(BiasJobs js, int accumulate) {
  __shared__ float red[256];
  int k = 0;
  for (int i = 1; i < js.n; ++i) k = (int)blockIdx.x >= js.j[i].red0 ? i : k;
  const BiasJob& jb = js.j[k];
  const int n = jb.dy.c;
  const int i = ((int)blockIdx.x - jb.red0) * 16 + (threadIdx.x & 15);
  const float t = fold16(jb.partial, i, n, jb.blocks, red);
  if (threadIdx.x < 16 && i < n) jb.db[i] = accumulate ? jb.db[i] + t : t;
}

static int bias_plan(const dct_view* dy, int vec, int& ppb) {
  const long long P = (long long)dy->n * dy->h * dy->w;
  const int rows = 256 / (dy->c / vec);
  long long blocks = (P + 8ll * rows - 1) / (8ll * rows);   // >= 8 iterations per thread
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  ppb = (int)((P + blocks - 1) / blocks);
  return (int)((P + ppb - 1) / ppb);
}

// ------------------------------------------------------------------------------- stem wgrad
struct StemG { int R, S, stride, dil, pad_h, pad_w; };

// partial[blk][co][taps+1] (taps of dw, then db).  x fp32 [N,H,W,1], dy T [N,Ho,Wo,Cout]
// QUAD (3x3, stride 1, dilation 1): the unit of work is 4 adjacent output pixels of a row -- one decode and one
// 3x6 input window per 4 channel-vector loads, which are all in flight together; ppb then counts quads.
template <typename T, bool QUAD>
__global__ __launch_bounds__(256) void stem_wgrad_fast_kernel(View x, View dy, float* partial, StemG g, int ppb) {
  constexpr int VEC = Vec<T>::N;
  constexpr int NT = 10;                 // 9 taps + bias
  __shared__ float red[4 * NT * 64];     // [wave][slot][C], C <= 64
  const int C = dy.c, CV = C / VEC;      // power of two, C <= 64
  const int rows = 256 / CV;
  const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
  const int taps = g.R * g.S;
  const long long P = (long long)dy.n * dy.h * dy.w;
  const long long pbeg = (long long)blockIdx.x * ppb, pend = min(P, pbeg + ppb);
  const float* xp = reinterpret_cast<const float*>(x.ptr);
  const T* dbase = reinterpret_cast<const T*>(dy.ptr) + cv * VEC;
  float acc[NT][VEC];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[t][i] = 0.f;
  if constexpr (QUAD) {
    const int wq = (dy.w + 3) >> 2;
    const long long Q = (long long)dy.n * dy.h * wq;
    const long long qbeg = (long long)blockIdx.x * ppb, qend = min(Q, qbeg + ppb);
    for (long long q = qbeg + row; q < qend; q += rows) {
      const unsigned uq = (unsigned)q, urow = uq / (unsigned)wq;
      const int x0 = (int)(uq - urow * (unsigned)wq) * 4, n = (int)(urow / (unsigned)dy.h), oy = (int)(urow - (unsigned)n * (unsigned)dy.h);
      float d[4][VEC];
      const T* dp = dbase + n * dy.sn + oy * dy.sh + x0 * dy.sw;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (x0 + p < dy.w) Vec<T>::load(dp + p * dy.sw, d[p]);
        else
#pragma unroll
          for (int i = 0; i < VEC; ++i) d[p][i] = 0.f;
      }
      float xw[3][6];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int iy = oy + r - g.pad_h;
        const bool rok = (unsigned)iy < (unsigned)x.h;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int ix = x0 + j - g.pad_w;
          xw[r][j] = (rok && (unsigned)ix < (unsigned)x.w) ? xp[n * x.sn + iy * x.sh + ix * x.sw] : 0.f;
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[t][i] = fmaf(d[p][i], xw[t / 3][p + t % 3], acc[t][i]);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[9][i] += d[p][i];
      }
    }
  } else
  for (long long pix = pbeg + row; pix < pend; pix += rows) {
    const unsigned up = (unsigned)pix, urow = up / (unsigned)dy.w;
    const int ox = (int)(up - urow * (unsigned)dy.w), n = (int)(urow / (unsigned)dy.h), oy = (int)(urow - (unsigned)n * (unsigned)dy.h);
    float d[VEC];
    Vec<T>::load(dbase + n * dy.sn + oy * dy.sh + ox * dy.sw, d);
    float xv[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int iy = oy * g.stride + r * g.dil - g.pad_h, ix = ox * g.stride + s * g.dil - g.pad_w;
        const bool ok = r < g.R && s < g.S && (unsigned)iy < (unsigned)x.h && (unsigned)ix < (unsigned)x.w;
        xv[r * 3 + s] = ok ? xp[n * x.sn + iy * x.sh + ix * x.sw] : 0.f;
      }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[t][i] = fmaf(d[i], xv[t], acc[t][i]);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[9][i] += d[i];
  }
  // fold the rows that share a wave (lanes with equal cv are CV apart)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float v = acc[t][i];
      for (int off = CV; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
      acc[t][i] = v;
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < CV) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < VEC; ++i) red[(wave * NT + t) * C + lane * VEC + i] = acc[t][i];
  }
  __syncthreads();
  // out index (co, t) with t in [0, taps]; tap t of a (R,S) kernel sits at slot (t / S) * 3 + t % S
  for (int o = threadIdx.x; o < C * (taps + 1); o += 256) {
    const int co = o / (taps + 1), t = o - co * (taps + 1);
    const int slot = t < taps ? (t / g.S) * 3 + (t % g.S) : 9;
    const float s = (red[(0 * NT + slot) * C + co] + red[(1 * NT + slot) * C + co]) +
                    (red[(2 * NT + slot) * C + co] + red[(3 * NT + slot) * C + co]);
    partial[(long long)blockIdx.x * C * (taps + 1) + o] = s;
  }
}

// MFMA form of the bf16 stem weight gradient (3x3, stride 1, Cout = 64, x fp32 [N,H,W,1]).  The VALU kernel above is bound by its ten
// FMAs per dy element (66 us for 16 x 252 x 252 x 64, 1.4 TB/s); here the reduction over pixels runs on the matrix pipe and the kernel
// is a stream over dy:  D[64 channels][16 columns] = sum_p dy[p][c] * win[p][j],  columns j = 0..8 the nine window values x[p + (r, s)],
// j = 9 the constant one (the bias gradient), the rest zero.  v_mfma_f32_32x32x16_bf16: K = 16 pixels of a row per step ("unit"),
//   A = dy^T (rows = channels): dy sits [pixel][channel] in memory, the reduction index is the SLOW one, so the unit's 2 KiB tile goes
//       through LDS as it is and comes back through the transposing read ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group);
//   B = win (columns = taps): lane (j, half) loads its eight consecutive x values itself (L2-resident 4-byte loads) and splits each into
//       bf16 high + bf16 low parts -- two MFMAs per channel half keep x to 16 mantissa bits (the products with the bf16 dy are then
//       exact to ~2^-17, as good as the fp32 FMAs of the VALU kernel for a sum that is rounded to fp32 anyway).
// A wave owns a unit at a time (no block barrier in the loop: it reads back only what it wrote), the next unit's loads are in flight
// while the current one is multiplied.  partial[blk][co][10] as the VALU kernel writes it.
typedef __attribute__((address_space(3))) bf16x4* stem_lds_bf16x4_ptr;
constexpr int STEM_NW = 16;          // waves per block: few blocks (= few partial rows for the fold), many waves in flight per CU
__global__ __launch_bounds__(STEM_NW * 64) void stem_wgrad_mfma_kernel(View x, View dy, float* partial, StemG g, int upr, long long nunits) {
  constexpr int NT = 10, C = 64, NWV = STEM_NW;
  __shared__ __attribute__((aligned(16))) char tile[NWV][16 * 128];
  __shared__ float red[NWV][C * NT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, kh = lane >> 5;                 // B column (tap) / K half of this lane
  const int tr = j / 3, ts = j - 3 * tr;                   // tap (r, s) for j < 9
  const float* xp = reinterpret_cast<const float*>(x.ptr);
  const bf16_t* dyp = reinterpret_cast<const bf16_t*>(dy.ptr);
  char* my = tile[wave];
  f32x16 acc[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[h][e] = 0.f;
  // transposing-read addresses: 16-lane group G = lane / 16 reads pixels 8 * (G / 2) + 4 * q .. + 3, channels 32 * hc + 16 * (G & 1) .. + 15;
  // lane 4 q' + p' of the group supplies row q', columns 4 p' .. 4 p' + 3 and receives column (lane % 16) of the four rows
  const int li = lane & 15;
  const unsigned tr_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)my +
                           (unsigned)((8 * kh + (li >> 2)) * 128 + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2);
  // staging: lane -> (pixel row of the unit, 16-byte chunk of its 128 bytes), two loads per lane
  const int srow = lane >> 3, schunk = lane & 7;

  const long long u0 = (long long)blockIdx.x * NWV + wave, ustep = (long long)gridDim.x * NWV;
  uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0;
  float xv[8];
  auto fetch = [&](long long u) {
    const unsigned rowid = (unsigned)(u / upr), ub = (unsigned)(u - (long long)rowid * upr);
    const int n = (int)(rowid / (unsigned)dy.h), oy = (int)(rowid - (unsigned)n * (unsigned)dy.h), ox0 = (int)ub * 16;
    const bf16_t* row = dyp + n * dy.sn + oy * dy.sh + schunk * 8;
    d0 = (ox0 + srow < dy.w) ? *reinterpret_cast<const uint4*>(row + (ox0 + srow) * dy.sw) : make_uint4(0, 0, 0, 0);
    d1 = (ox0 + srow + 8 < dy.w) ? *reinterpret_cast<const uint4*>(row + (ox0 + srow + 8) * dy.sw) : make_uint4(0, 0, 0, 0);
    const int iy = oy + tr - g.pad_h, ix0 = ox0 + 8 * kh + ts - g.pad_w;
    const bool rok = j < 9 && (unsigned)iy < (unsigned)x.h;
    const float* xr = xp + n * x.sn + iy * x.sh;
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = (rok && (unsigned)(ix0 + e) < (unsigned)x.w) ? xr[(ix0 + e) * x.sw] : 0.f;
  };
  if (u0 < nunits) fetch(u0);
  for (long long u = u0; u < nunits; u += ustep) {
    // this unit's operands are in registers: dy tile -> LDS, x window -> bf16 high / low fragments
    *reinterpret_cast<uint4*>(my + srow * 128 + schunk * 16) = d0;
    *reinterpret_cast<uint4*>(my + (srow + 8) * 128 + schunk * 16) = d1;
    bf16x8 whi, wlo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = j == 9 ? 1.f : xv[e];
      const bf16_t hi = (bf16_t)v;
      whi[e] = hi;
      wlo[e] = (bf16_t)(v - (float)hi);
    }
    if (u + ustep < nunits) fetch(u + ustep);              // next unit's loads fly while this one is multiplied
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {
      const unsigned a0 = tr_base + hc * 64;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((stem_lds_bf16x4_ptr)(size_t)(a0));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((stem_lds_bf16x4_ptr)(size_t)(a0 + 4 * 128));
      bf16x8 a;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = lo[e]; a[4 + e] = hi[e]; }
      acc[hc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, whi, acc[hc], 0, 0, 0);
      acc[hc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wlo, acc[hc], 0, 0, 0);
    }
  }
  // accumulator (hc, register 4 q + e) of lane (j, kh): channel 32 hc + 8 q + 4 kh + e, column j
  if (j < NT) {
#pragma unroll
    for (int hc = 0; hc < 2; ++hc)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][(32 * hc + 8 * q + 4 * kh + e) * NT + j] = acc[hc][4 * q + e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * NT; i += NWV * 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) t += red[w][i];          // fixed order: deterministic
    partial[(long long)blockIdx.x * (C * NT) + i] = t;
  }
}

// out: dw[co][inner], db[co] from partial[blk][co][inner+1]
__global__ __launch_bounds__(256) void split_dw_db_kernel(const float* partial, float* dw, float* db, int cout, int inner, int blocks, int accumulate) {
  __shared__ float red[256];
  const int n = cout * (inner + 1);
  const int i = blockIdx.x * 16 + (threadIdx.x & 15);
  const float t = fold16(partial, i, n, blocks, red);
  if (threadIdx.x < 16 && i < n) {
    const int co = i / (inner + 1), k = i - co * (inner + 1);
    if (k < inner) { if (dw) dw[co * inner + k] = accumulate ? dw[co * inner + k] + t : t; }
    else if (db) db[co] = accumulate ? db[co] + t : t;
  }
}

// ------------------------------------------------------------------------------- head wgrad
// partial[blk][co][cin+1]: x T [P][Cin], dy f32 [P][Cout <= 8]
template <typename T, int COUT>
__global__ __launch_bounds__(256) void head_dw_fast_kernel(View x, View dy, float* partial, int ppb, int lin_x, int lin_dy) {
  constexpr int VEC = Vec<T>::N;
  extern __shared__ float red[];         // [4 waves][COUT][C] + [4][COUT]
  const int C = x.c, CV = C / VEC;       // CV <= 64, power of two
  const int rows = 256 / CV;
  const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
  const long long P = (long long)x.n * x.h * x.w;
  const long long pbeg = (long long)blockIdx.x * ppb, pend = min(P, pbeg + ppb);
  const T* xb = reinterpret_cast<const T*>(x.ptr) + cv * VEC;
  const float* db_ = reinterpret_cast<const float*>(dy.ptr);
  float acc[COUT][VEC], accb[COUT];
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
    accb[o] = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[o][i] = 0.f;
  }
  for (long long pix = pbeg + row; pix < pend; pix += rows) {
    float xv[VEC];
    Vec<T>::load(xb + pix_off(x, pix, lin_x), xv);
    const float* dp = db_ + pix_off(dy, pix, lin_dy);
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      const float d = dp[o];
      accb[o] += d;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[o][i] = fmaf(d, xv[i], acc[o][i]);
    }
  }
#pragma unroll
  for (int o = 0; o < COUT; ++o) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float v = acc[o][i];
      for (int off = CV; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
      acc[o][i] = v;
    }
    float v = accb[o];
    for (int off = CV; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    accb[o] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = CV >= 64 ? 1 : 4;       // CV == 64: a wave is one row; rows 0..3 are the 4 waves
  float* redb = red + 4 * COUT * C;
  if (lane < CV) {
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) red[(wave * COUT + o) * C + lane * VEC + i] = acc[o][i];
      if (lane == 0) redb[wave * COUT + o] = accb[o];
    }
  }
  (void)nw;
  __syncthreads();
  for (int i = threadIdx.x; i < COUT * (C + 1); i += 256) {
    const int o = i / (C + 1), k = i - o * (C + 1);
    float s;
    if (k < C) s = (red[(0 * COUT + o) * C + k] + red[(1 * COUT + o) * C + k]) + (red[(2 * COUT + o) * C + k] + red[(3 * COUT + o) * C + k]);
    else s = (redb[0 * COUT + o] + redb[1 * COUT + o]) + (redb[2 * COUT + o] + redb[3 * COUT + o]);
    partial[(long long)blockIdx.x * COUT * (C + 1) + i] = s;
  }
}

static int pix_plan(long long P, int rows, int iters, int max_blocks, int& ppb) {
  long long blocks = (P + (long long)rows * iters - 1) / ((long long)rows * iters);
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks < 1) blocks = 1;
  ppb = (int)((P + blocks - 1) / blocks);
  return (int)((P + ppb - 1) / ppb);
}

static inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// resident blocks of `kernel` on the whole device (one full round of the grid: no half-empty second round)
template <typename K>
static int resident_blocks(K kernel, int fallback) {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fallback;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return fallback;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) return fallback;
  return cus * per_cu;
}

}  // namespace

// igemm.hip (stem weight gradient from the data gradient's epilogue): partial[blk][cout][inner + 1] -> dw, db.  Thousands of slabs
// (one per conv block) are first folded to STEM_FOLD_ROWS by coalesced row sums (block b adds slabs b, b + rows, ...: a fixed order).
constexpr int STEM_FOLD_ROWS = 128;
__global__ __launch_bounds__(640) void slab_rows_fold_kernel(const float* __restrict__ partial, float* __restrict__ out, int n, int blocks) {
  const int i = threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = blockIdx.x;
  for (; b + 3 * STEM_FOLD_ROWS < blocks; b += 4 * STEM_FOLD_ROWS) {
    s0 += partial[(long long)b * n + i]; s1 += partial[(long long)(b + STEM_FOLD_ROWS) * n + i];
    s2 += partial[(long long)(b + 2 * STEM_FOLD_ROWS) * n + i]; s3 += partial[(long long)(b + 3 * STEM_FOLD_ROWS) * n + i];
  }
  for (; b < blocks; b += STEM_FOLD_ROWS) s0 += partial[(long long)b * n + i];
  out[(long long)blockIdx.x * n + i] = (s0 + s1) + (s2 + s3);
}
size_t dct_split_dw_db_scratch(int cout, int inner) { return (size_t)STEM_FOLD_ROWS * cout * (inner + 1) * sizeof(float); }
// scratch: dct_split_dw_db_scratch bytes behind the slabs when blocks > STEM_FOLD_ROWS
void dct_split_dw_db_launch(const float* partial, float* scratch, float* dw, float* db, int cout, int inner, int blocks, int accumulate, hipStream_t st) {
  const int n = cout * (inner + 1);
  if (blocks > STEM_FOLD_ROWS && n <= 640) {
    DCT_LAUNCH(DCT_PROF_POINTWISE, slab_rows_fold_kernel, dim3(STEM_FOLD_ROWS), dim3(640), 0, st, partial, scratch, n, blocks);
    partial = scratch; blocks = STEM_FOLD_ROWS;
  }
  DCT_LAUNCH(DCT_PROF_POINTWISE, split_dw_db_kernel, dim3(div_up(n, 16)), dim3(256), 0, st, partial, dw, db, cout, inner, blocks, accumulate);
}

// =============================================================================== C ABI
extern "C" size_t dct_bias_grad_workspace_bytes(const dct_view* dy) {
  if (!dy || dy->c < 1) return 0;
  return (size_t)1024 * dy->c * sizeof(float);
}

extern "C" int dct_bias_grad(const dct_view* dy, float* db, int accumulate, int dtype,
                             void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (!view_ok(dy) || !db) return DCT_ERR_BAD_ARG;
  if (dtype != DCT_F32 && dtype != DCT_BF16) return DCT_ERR_BAD_ARG;
  const int vec = dtype == DCT_BF16 ? 8 : 4, esz = dtype == DCT_BF16 ? 2 : 4;
  if (dy->c % vec) return DCT_ERR_UNSUPPORTED;
  const int cv = dy->c / vec;
  if (cv > 256 || 256 % cv) return DCT_ERR_UNSUPPORTED;
  if (((uintptr_t)dy->ptr % 16) || (dy->sw % vec) || (dy->sh % vec) || (dy->sn % vec)) return DCT_ERR_UNSUPPORTED;
  (void)esz;
  int ppb;
  const int blocks = bias_plan(dy, vec, ppb);
  if (!workspace || workspace_bytes < (size_t)blocks * dy->c * sizeof(float)) return DCT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const View v = to_view(dy);
  const int lin = view_linear(dy);
  if (dtype == DCT_BF16) DCT_LAUNCH(DCT_PROF_POINTWISE, bias_partial_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, v, (float*)workspace, ppb, lin);
  else DCT_LAUNCH(DCT_PROF_POINTWISE, bias_partial_kernel<float>, dim3(blocks), dim3(256), 0, st, v, (float*)workspace, ppb, lin);
  DCT_LAUNCH(DCT_PROF_POINTWISE, partial_reduce_kernel, dim3(div_up(dy->c, 16)), dim3(256), 0, st,
             (const float*)workspace, db, dy->c, blocks, accumulate);
  return dct_check_launch();
}

extern "C" size_t dct_bias_grad_batched_workspace_bytes(const dct_view* dys, int n) {
  size_t t = 0;
  for (int k = 0; dys && k < n; ++k) t += dct_bias_grad_workspace_bytes(dys + k);
  return t;
}

extern "C" int dct_bias_grad_batched(const dct_view* dys, float* const* dbs, int n, int accumulate, int dtype,
                                     void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (!dys || !dbs || n < 1) return DCT_ERR_BAD_ARG;
  if (n > kBiasJobs) return DCT_ERR_UNSUPPORTED;
  if (dtype != DCT_F32 && dtype != DCT_BF16) return DCT_ERR_BAD_ARG;
  const int vec = dtype == DCT_BF16 ? 8 : 4;
  BiasJobs js;
  js.n = n;
  int blk = 0, red = 0;
  size_t off = 0;
  for (int k = 0; k < n; ++k) {
    const dct_view* dy = dys + k;
    if (!view_ok(dy) || !dbs[k]) return DCT_ERR_BAD_ARG;
    if (dy->c % vec) return DCT_ERR_UNSUPPORTED;
    const int cv = dy->c / vec;
    if (cv > 256 || 256 % cv) return DCT_ERR_UNSUPPORTED;
    if (((uintptr_t)dy->ptr % 16) || (dy->sw % vec) || (dy->sh % vec) || (dy->sn % vec)) return DCT_ERR_UNSUPPORTED;
    BiasJob& jb = js.j[k];
    jb.dy = to_view(dy);
    jb.blocks = bias_plan(dy, vec, jb.ppb);
    jb.linear = view_linear(dy);
    jb.blk0 = blk; jb.red0 = red; jb.pad = 0;
    jb.partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + off);
    jb.db = dbs[k];
    blk += jb.blocks;
    red += div_up(dy->c, 16);
    off += (size_t)jb.blocks * dy->c * sizeof(float);
  }
  if (!workspace || workspace_bytes < off) return DCT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCT_BF16) DCT_LAUNCH(DCT_PROF_POINTWISE, bias_partial_batched_kernel<bf16_t>, dim3(blk), dim3(256), 0, st, js);
  else DCT_LAUNCH(DCT_PROF_POINTWISE, bias_partial_batched_kernel<float>, dim3(blk), dim3(256), 0, st, js);
  DCT_LAUNCH(DCT_PROF_POINTWISE, partial_reduce_batched_kernel, dim3(red), dim3(256), 0, st, js, accumulate);
  return dct_check_launch();
}

extern "C" size_t dct_conv_cin1_wgrad_workspace_bytes(const dct_view* dy, const dct_conv_desc* d) {
  if (!dy || !d) return 0;
  return (size_t)4096 * dy->c * (d->R * d->S + 1) * sizeof(float);
}

extern "C" int dct_conv_cin1_wgrad(const dct_view* x, const dct_view* dy, float* dw, float* db,
                                   const dct_conv_desc* d, int accumulate, int dtype,
                                   void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (!view_ok(x) || !view_ok(dy) || !d || x->c != 1 || x->n != dy->n) return DCT_ERR_BAD_ARG;
  if (dtype != DCT_F32 && dtype != DCT_BF16) return DCT_ERR_BAD_ARG;
  if (d->R < 1 || d->S < 1 || d->R > 3 || d->S > 3) return DCT_ERR_UNSUPPORTED;
  const int vec = dtype == DCT_BF16 ? 8 : 4;
  if (dy->c % vec) return DCT_ERR_UNSUPPORTED;
  const int cv = dy->c / vec;
  if (dy->c > 64 || !pow2(cv)) return DCT_ERR_UNSUPPORTED;
  if (((uintptr_t)dy->ptr % 16) || (dy->sw % vec) || (dy->sh % vec) || (dy->sn % vec)) return DCT_ERR_UNSUPPORTED;
  const int taps = d->R * d->S;
  int ppb;
  const bool quad = d->R == 3 && d->S == 3 && d->stride == 1 && d->dil == 1;
  const long long units = quad ? (long long)dy->n * dy->h * ((dy->w + 3) / 4) : (long long)dy->n * dy->h * dy->w;
  int max_blocks = 1024;
  if (quad) {
    static const int res_bf16 = resident_blocks(stem_wgrad_fast_kernel<bf16_t, true>, 512);
    static const int res_f32 = resident_blocks(stem_wgrad_fast_kernel<float, true>, 512);
    max_blocks = std::min(1024, dtype == DCT_BF16 ? res_bf16 : res_f32);
  }
  const int blocks = pix_plan(units, 256 / cv, quad ? 4 : 8, max_blocks, ppb);
  if (!workspace || workspace_bytes < (size_t)blocks * dy->c * (taps + 1) * sizeof(float)) return DCT_ERR_WORKSPACE;
  StemG g; g.R = d->R; g.S = d->S; g.stride = d->stride; g.dil = d->dil; g.pad_h = d->pad_h; g.pad_w = d->pad_w;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DCT_BF16 && quad && dy->c == 64 && dy->sw % 8 == 0 && (long long)dy->n * dy->sn < (1ll << 31) && (long long)x->n * x->sn < (1ll << 31)) {
    // MFMA form: units of 16 pixels of a row, one per wave at a time; 1024 blocks of four waves at most
    const int upr = (dy->w + 15) / 16;
    const long long nunits = (long long)dy->n * dy->h * upr;
    const int mblocks = (int)std::min<long long>(g_stem_mfma_blocks, (nunits + STEM_NW - 1) / STEM_NW);
    if (workspace_bytes < (size_t)mblocks * dy->c * (taps + 1) * sizeof(float)) return DCT_ERR_WORKSPACE;
    DCT_LAUNCH(DCT_PROF_POINTWISE, stem_wgrad_mfma_kernel, dim3(mblocks), dim3(STEM_NW * 64), 0, st, to_view(x), to_view(dy), (float*)workspace, g, upr, nunits);
    DCT_LAUNCH(DCT_PROF_POINTWISE, split_dw_db_kernel, dim3(div_up(dy->c * (taps + 1), 16)), dim3(256), 0, st,
               (const float*)workspace, dw, db, dy->c, taps, mblocks, accumulate);
    return dct_check_launch();
  }
  if (dtype == DCT_BF16) {
    if (quad) DCT_LAUNCH(DCT_PROF_POINTWISE, (stem_wgrad_fast_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, st, to_view(x), to_view(dy), (float*)workspace, g, ppb);
    else DCT_LAUNCH(DCT_PROF_POINTWISE, (stem_wgrad_fast_kernel<bf16_t, false>), dim3(blocks), dim3(256), 0, st, to_view(x), to_view(dy), (float*)workspace, g, ppb);
  } else {
    if (quad) DCT_LAUNCH(DCT_PROF_POINTWISE, (stem_wgrad_fast_kernel<float, true>), dim3(blocks), dim3(256), 0, st, to_view(x), to_view(dy), (float*)workspace, g, ppb);
    else DCT_LAUNCH(DCT_PROF_POINTWISE, (stem_wgrad_fast_kernel<float, false>), dim3(blocks), dim3(256), 0, st, to_view(x), to_view(dy), (float*)workspace, g, ppb);
  }
  DCT_LAUNCH(DCT_PROF_POINTWISE, split_dw_db_kernel, dim3(div_up(dy->c * (taps + 1), 16)), dim3(256), 0, st,
             (const float*)workspace, dw, db, dy->c, taps, blocks, accumulate);
  return dct_check_launch();
}

extern "C" size_t dct_conv1x1_head_bwd_workspace_bytes(const dct_view* x, int cout) {
  if (!x || cout < 1) return 0;
  return (size_t)1024 * cout * (x->c + 1) * sizeof(float);
}

namespace {
template <typename T>
static void launch_head_dw(int cout, int blocks, size_t sh, hipStream_t st, const View& x, const View& dy, float* ws, int ppb, int lx, int ld) {
  switch (cout) {
#define CASE(N) case N: DCT_LAUNCH(DCT_PROF_POINTWISE, (head_dw_fast_kernel<T, N>), dim3(blocks), dim3(256), sh, st, x, dy, ws, ppb, lx, ld); break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
}
}  // namespace

// dw/db half of the classifier head backward (dct_conv1x1_head_bwd in pointwise.hip calls this)
int dct_head_dw_launch(const dct_view* x, const dct_view* dy, float* dw, float* db, int accumulate, int dtype,
                       void* workspace, size_t workspace_bytes, hipStream_t st) {
  const int vec = dtype == DCT_BF16 ? 8 : 4;
  if (x->c % vec) return DCT_ERR_UNSUPPORTED;
  const int cv = x->c / vec;
  if (cv > 64 || !pow2(cv) || dy->c < 1 || dy->c > 8) return DCT_ERR_UNSUPPORTED;
  if (((uintptr_t)x->ptr % 16) || (x->sw % vec) || (x->sh % vec) || (x->sn % vec)) return DCT_ERR_UNSUPPORTED;
  int ppb;
  const int blocks = pix_plan((long long)x->n * x->h * x->w, 256 / cv, 8, 1024, ppb);
  if (!workspace || workspace_bytes < (size_t)blocks * dy->c * (x->c + 1) * sizeof(float)) return DCT_ERR_WORKSPACE;
  const size_t sh = (size_t)(4 * dy->c * x->c + 4 * dy->c) * sizeof(float);
  if (sh > 64 * 1024) return DCT_ERR_UNSUPPORTED;
  const View vx = to_view(x), vdy = to_view(dy);
  const int lx = view_linear(x), ld = view_linear(dy);
  if (dtype == DCT_BF16) launch_head_dw<bf16_t>(dy->c, blocks, sh, st, vx, vdy, (float*)workspace, ppb, lx, ld);
  else launch_head_dw<float>(dy->c, blocks, sh, st, vx, vdy, (float*)workspace, ppb, lx, ld);
  DCT_LAUNCH(DCT_PROF_POINTWISE, split_dw_db_kernel, dim3(div_up(dy->c * (x->c + 1), 16)), dim3(256), 0, st,
             (const float*)workspace, dw, db, dy->c, x->c, blocks, accumulate);
  return dct_check_launch();
}
