// Implicit-GEMM convolution on MFMA for gfx950 (K1/K2/K3 forward + data gradients).
//
//   D[channel][pixel] = sum_{tap,ci} Wt[channel][tap][ci] * X[pixel shifted by tap][ci]
//
// The MFMA A operand is the weight tile and the B operand the pixel tile, so an accumulator
// register quad holds 4 consecutive output channels of ONE pixel: with NHWC output the
// epilogue stores 8 B (bf16) / 16 B (f32) per lane per quad and 32 lanes x 2 halves cover
// a contiguous 64 B / 128 B channel run of each pixel.
//
// Tile: BN channels x BM pixels x 64 bytes of K per step, 256 threads = 4 waves, each wave
// 64 channels x (BM / WAVES_M) pixels of 32x32 MFMA tiles.  Global->register->LDS staging
// with the next K-step's global loads in flight during the current step's MFMAs (one
// barrier per step, two LDS buffers).  LDS rows are 64 B of data on an 80 B pitch, which
// makes the ds_read_b128 fragment reads bank-conflict free (20*i mod 64 distinct for any
// 16 rows distinct mod 16).
//
// bf16: v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  f32: v_mfma_f32_32x32x2_f32 -- exact
// fp32 FMA chains, used by the parity path.
#include <algorithm>
#include "igemm_common.h"

extern int g_tune_lean;           // wgrad.hip: bit mask of the instruction-lean loop forms (bit 1: this file's packed-rows kernel)

namespace {


template <typename T> struct Mfma;
template <> struct Mfma<bf16_t> {
  static constexpr int KSTEP = 16;
  __device__ static __forceinline__ void run(const char* a_row, const char* b_row, int half, f32x16& acc, int kk) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_row + kk * 32 + half * 16);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(b_row + kk * 32 + half * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};
template <> struct Mfma<float> {
  static constexpr int KSTEP = 2;
  __device__ static __forceinline__ void run(const char* a_row, const char* b_row, int half, f32x16& acc, int kk) {
    const float a = *reinterpret_cast<const float*>(a_row + (kk * 2 + half) * 4);
    const float b = *reinterpret_cast<const float*>(b_row + (kk * 2 + half) * 4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
};

// Epilogue for 4 consecutive channels [c, c+4) of pixel (n, oy, ox).
template <typename T>
__device__ __forceinline__ void epilogue_store4(const IgemmParams& p, int n, int oy, int ox, int c, float v[4]) {
  int co = c;
  if (p.scatter) {
    const int ab = c / p.cout;
    co = c - ab * p.cout;
    oy = 2 * oy + (ab >> 1);
    ox = 2 * ox + (ab & 1);
  }
  if (p.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += p.bias[co + i];
  }
  if (p.relu) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = relu1(v[i]);
  }
  if (p.mask && co < p.mask_channels) {
    const T* mp = reinterpret_cast<const T*>(p.mask) + (n * p.msN + oy * p.msH + ox * p.msW + co);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = to_f32(mp[i]) > 0.f ? v[i] * p.mask_scale : 0.f;
  }
  T* yp = reinterpret_cast<T*>(p.y) + (n * p.ysN + oy * p.ysH + ox * p.ysW + co);
  if (p.accumulate) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += to_f32(yp[i]);
  }
  if constexpr (sizeof(T) == 2) {
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(yp) = o;
  } else {
    f32x4 o = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(yp) = o;
  }
}



template <typename T, int BN, int BM>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
  constexpr int EPV = 16 / sizeof(T);     // elements per 16-byte vector
  constexpr int BK = 64 / sizeof(T);      // K elements per step (64 B rows)
  constexpr int ROWB = 80;                // LDS row pitch
  constexpr int WAVES_N = BN / 64;
  constexpr int WAVES_M = 4 / WAVES_N;
  constexpr int WM = BM / WAVES_M;
  constexpr int TM = WM / 32;
  constexpr int TN = 2;
  constexpr int PA = BN / 64;             // staging passes (64 rows per pass)
  constexpr int PB = BM / 64;
  constexpr int BUF = (BN + BM) * ROWB;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WAVES_M, wm = wave % WAVES_M;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int chunk = tid & 3, srow = tid >> 2;
  const long long Ktot = (long long)p.R * p.S * p.Cin;

  // per-thread staging rows of the pixel tile
  long long bbase[PB];
  int biy0[PB], bix0[PB];
#pragma unroll
  for (int pp = 0; pp < PB; ++pp) {
    const int m = m0 + srow + 64 * pp;
    if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int n = m / hw, rem = m - n * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      bbase[pp] = n * p.xsN;
      biy0[pp] = oy * p.stride - p.pad_h;
      bix0[pp] = ox * p.stride - p.pad_w;
    } else {
      bbase[pp] = 0; biy0[pp] = -(1 << 28); bix0[pp] = -(1 << 28);
    }
  }
  const T* wbase[PA];
#pragma unroll
  for (int pp = 0; pp < PA; ++pp)
    wbase[pp] = reinterpret_cast<const T*>(p.w) + (long long)(n0 + srow + 64 * pp) * Ktot + chunk * EPV;

  const int kbeg = blockIdx.z * p.kiters_per_split;
  const int kend = min(p.kiters, kbeg + p.kiters_per_split);

  // running (r, s, c-iter) for the NEXT tile to load
  int tap = kbeg / p.cin_iters;
  int cit = kbeg - tap * p.cin_iters;
  int r = tap / p.S, s = tap - r * p.S;

  uint4 ra[PA], rb[PB];
  auto load_tile = [&]() {
    const int c0 = cit * BK;
    const long long woff = (long long)(r * p.S + s) * p.Cin + c0;
#pragma unroll
    for (int pp = 0; pp < PA; ++pp) ra[pp] = *reinterpret_cast<const uint4*>(wbase[pp] + woff);
#pragma unroll
    for (int pp = 0; pp < PB; ++pp) {
      const int iy = biy0[pp] + r * p.dil, ix = bix0[pp] + s * p.dil;
      if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) {
        const T* xp = reinterpret_cast<const T*>(p.x) + (bbase[pp] + iy * p.xsH + ix * p.xsW + c0 + chunk * EPV);
        rb[pp] = *reinterpret_cast<const uint4*>(xp);
      } else {
        rb[pp] = make_uint4(0, 0, 0, 0);
      }
    }
    if (++cit == p.cin_iters) { cit = 0; if (++s == p.S) { s = 0; ++r; } }
  };
  auto store_tile = [&](char* buf) {
#pragma unroll
    for (int pp = 0; pp < PA; ++pp)
      *reinterpret_cast<uint4*>(buf + (srow + 64 * pp) * ROWB + chunk * 16) = ra[pp];
#pragma unroll
    for (int pp = 0; pp < PB; ++pp)
      *reinterpret_cast<uint4*>(buf + (BN + srow + 64 * pp) * ROWB + chunk * 16) = rb[pp];
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (kbeg < kend) {
    load_tile();
    store_tile(smem);
  }
  __syncthreads();
  int cur = 0;
  const int half = lane >> 5, l31 = lane & 31;
  for (int it = kbeg; it < kend; ++it) {
    const bool more = it + 1 < kend;
    if (more) load_tile();
    const char* A = smem + cur * BUF;
    const char* B = A + BN * ROWB;
#pragma unroll
    for (int kk = 0; kk < BK / Mfma<T>::KSTEP; ++kk) {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const char* arow = A + (wn * 64 + i * 32 + l31) * ROWB;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const char* brow = B + (wm * WM + j * 32 + l31) * ROWB;
          Mfma<T>::run(arow, brow, half, acc[i][j], kk);
        }
      }
    }
    if (more) store_tile(smem + (cur ^ 1) * BUF);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * WM + j * 32 + l31;
    if (m >= p.M) continue;
    const int hw = p.Ho * p.Wo;
    const int n = m / hw, rem = m - n * hw;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n0 + wn * 64 + i * 32 + 8 * q + 4 * half;
        float v[4] = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        if (p.partial) {
          float* pp = p.partial + ((long long)blockIdx.z * p.M + m) * p.N + c;
          *reinterpret_cast<f32x4*>(pp) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          epilogue_store4<T>(p, n, oy, ox, c, v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 main kernel ("v2"): K-step of 64 elements (128-B LDS rows), tiles staged straight from HBM
// into LDS by global_load_lds_dwordx4 (no VGPR round trip), two LDS stages, one barrier per
// K-step.  One wave-instruction fills 8 rows x 128 B; the LDS image is lane-linear, so the
// bank-conflict swizzle is applied to the per-lane SOURCE chunk and again on the fragment read:
// 16-B chunk c of row r lives at slot c ^ ((r >> 1) & 7), which makes every ds_read_b128 lane group
// hit 16 distinct 16-B slots of the 256-B bank row.  Out-of-image taps (data-gradient halo) read a
// 128-B zero page instead of branching around the DMA; rows past M are clamped to the last pixel
// (their results are never stored).


// LEAN (round 4): pieces as buffer loads to LDS -- a weight piece's lane offset is a constant and its step offset the scalar
// it * 128 (K runs [tap][cin] in the packed weights), a pixel piece's lane offset is its row's constant plus the tap's scalar offset,
// and a tap outside the image (BOUNDS) is one bit of a per-row mask built once (such a lane holds the offset the descriptor rejects
// and stages zeros).  28-58 vector and 41-79 scalar instructions per 8 MFMAs before (tools/isa_loop_mix.py).  Bit-identical.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool BOUNDS, bool LEAN>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64) void igemm2_kernel(IgemmParams p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int ROWB = 128;                 // bytes per LDS row = 64 bf16 of K
  constexpr int BK = 64;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int RPP = NW * 8;               // rows staged per pass (8 per wave-instruction)
  constexpr int PA = BN / RPP, PB = BM / RPP;
  static_assert(BN % RPP == 0 && BM % RPP == 0 && TM >= 1 && TN >= 1, "tile/wave mismatch");
  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = LEAN ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
  const int wn = wave / WAVES_M, wm = wave % WAVES_M;
  int bxi, byi, bzi;
  if (!xcd_remap(p, bxi, byi, bzi)) return;
  const int m0 = bxi * BM, n0 = byi * BN;
  const int srow = wave * 8 + (lane >> 3);                 // staging row within a pass
  const int schunk = (lane & 7) ^ ((srow >> 1) & 7);       // source chunk for this lane's LDS slot
  const long long Ktot = (long long)p.R * p.S * p.Cin;

  long long rowoff[PB];
  int biy0[PB], bix0[PB];
  unsigned tapmask[PB];                       // LEAN + BOUNDS: bit t = tap t of this row lies inside the image (R * S <= 32)
#pragma unroll
  for (int pp = 0; pp < PB; ++pp) {
    int m = m0 + srow + RPP * pp;
    if (m >= p.M) m = p.M - 1;
    const int hw = p.Ho * p.Wo;
    const int n = m / hw, rem = m - n * hw;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad_h, ix0 = ox * p.stride - p.pad_w;
    rowoff[pp] = n * p.xsN + iy0 * p.xsH + ix0 * p.xsW + schunk * 8;
    biy0[pp] = iy0; bix0[pp] = ix0;
    tapmask[pp] = 0;
    if constexpr (LEAN && BOUNDS) {
      for (int rr = 0; rr < p.R; ++rr)
        for (int ss = 0; ss < p.S; ++ss)
          if ((unsigned)(iy0 + rr * p.dil) < (unsigned)p.Hi && (unsigned)(ix0 + ss * p.dil) < (unsigned)p.Wi) tapmask[pp] |= 1u << (rr * p.S + ss);
    }
  }
  const bf16_t* wrow = reinterpret_cast<const bf16_t*>(p.w) + (long long)(n0 + srow) * Ktot + schunk * 8;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
  const char* zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 7) * 16;

  const int kbeg = bzi * p.kiters_per_split;
  const int kend = min(p.kiters, kbeg + p.kiters_per_split);
  int tap = kbeg / p.cin_iters;
  int cit = kbeg - tap * p.cin_iters;
  int r = tap / p.S, s = tap - r * p.S;

  __amdgpu_buffer_rsrc_t rsX, rsW;
  int wvo[PA], xvo[PB];                       // LEAN: constant lane offsets (bytes) of the weight / pixel pieces
  if constexpr (LEAN) {
    rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (long long)n0 * Ktot * 2), 0, (int)(p.w_bytes - (long long)n0 * Ktot * 2), 0x00020000);
#pragma unroll
    for (int pp = 0; pp < PA; ++pp) wvo[pp] = (int)(((long long)(srow + pp * RPP) * Ktot + schunk * 8) * 2);
#pragma unroll
    for (int pp = 0; pp < PB; ++pp) xvo[pp] = (int)(rowoff[pp] * 2);
  }
  const unsigned smem_l2 = (unsigned)(size_t)(lptr_c)(smem);
  int itn = kbeg;                             // LEAN: K-step the next stage() call fetches
  auto stage = [&](char* buf) {
    if constexpr (LEAN) {
      const unsigned la = smem_l2 + (unsigned)(buf - smem) + wave * 8 * ROWB;
#pragma unroll
      for (int pp = 0; pp < PA; ++pp) buf_lds16(rsW, la + pp * RPP * ROWB, wvo[pp], (unsigned)itn * 128u);
      const unsigned lb = la + BN * ROWB;
      const int toffb = ((r * p.dil) * (int)p.xsH + (s * p.dil) * (int)p.xsW + cit * BK) * 2;
      if (BOUNDS) {
        // the tap's offset rides in the lane offset (a row's own offset is negative above / left of the image)
        const int tapi = r * p.S + s;
#pragma unroll
        for (int pp = 0; pp < PB; ++pp)
          buf_lds16(rsX, lb + pp * RPP * ROWB, ((tapmask[pp] >> tapi) & 1u) ? xvo[pp] + toffb : DCT_BUF_INVALID, 0u);
      } else {
#pragma unroll
        for (int pp = 0; pp < PB; ++pp) buf_lds16(rsX, lb + pp * RPP * ROWB, xvo[pp], (unsigned)toffb);
      }
      ++itn;
      if (++cit == p.cin_iters) { cit = 0; if (++s == p.S) { s = 0; ++r; } }
      return;
    }
    const int c0 = cit * BK;
    const long long woff = (long long)(r * p.S + s) * p.Cin + c0;
    char* la = buf + wave * 8 * ROWB;
#pragma unroll
    for (int pp = 0; pp < PA; ++pp)
      __builtin_amdgcn_global_load_lds((gptr_t)(wrow + (long long)pp * RPP * Ktot + woff), (lptr_t)(la + pp * RPP * ROWB), 16, 0, 0);
    char* lb = buf + (BN + wave * 8) * ROWB;
    const long long toff = (long long)(r * p.dil) * p.xsH + (long long)(s * p.dil) * p.xsW + c0;
#pragma unroll
    for (int pp = 0; pp < PB; ++pp) {
      const char* src = reinterpret_cast<const char*>(xb + rowoff[pp] + toff);
      if (BOUNDS) {
        const int iy = biy0[pp] + r * p.dil, ix = bix0[pp] + s * p.dil;
        if (!((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)) src = zero;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lb + pp * RPP * ROWB), 16, 0, 0);
    }
    if (++cit == p.cin_iters) { cit = 0; if (++s == p.S) { s = 0; ++r; } }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (kbeg < kend) stage(smem);
  // bias of the tile's BN channels -> LDS now (behind the two stage buffers), so that the epilogue does not start with a
  // global-memory round trip (stamps: 1.2k of the 2k cycles of "accumulators -> LDS" were the wait for these loads)
  // (only where the extra BN * 4 bytes keep two blocks per CU: the 256 x 64 tile's stages are 80 KiB already)
  constexpr bool BIAS_LDS = 2 * STAGE + BN * 4 <= 80 * 1024;
  float* biasL = reinterpret_cast<float*>(smem + 2 * STAGE);
  if (BIAS_LDS && p.staged && tid < BN) {
    int co = n0 + tid;
    if (p.scatter) co %= p.cout;
    biasL[tid] = p.bias ? p.bias[co] : 0.f;
  }
  __syncthreads();
  int cur = 0;
  const int half = lane >> 5, l31 = lane & 31;
  const int swz = (l31 >> 1) & 7;
  for (int it = kbeg; it < kend; ++it) {
    if (it + 1 < kend) stage(smem + (cur ^ 1) * STAGE);
    const char* A = smem + cur * STAGE + (wn * WTN + l31) * ROWB;
    const char* B = smem + cur * STAGE + (BN + wm * WTM + l31) * ROWB;
    // fragments of sub-step kk+1 are read while the MFMAs of sub-step kk run (two register sets)
    bf16x8 a[2][TN], b[2][TM];
    auto load_frags = [&](int set, int kk) {
      const int coff = ((2 * kk + half) ^ swz) * 16;
#pragma unroll
      for (int i = 0; i < TN; ++i) a[set][i] = *reinterpret_cast<const bf16x8*>(A + i * 32 * ROWB + coff);
#pragma unroll
      for (int j = 0; j < TM; ++j) b[set][j] = *reinterpret_cast<const bf16x8*>(B + j * 32 * ROWB + coff);
    };
    load_frags(0, 0);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      if (kk + 1 < BK / 16) load_frags((kk + 1) & 1, kk + 1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue
  if (p.staged) {
    // Staged through LDS: every wave drops its accumulators (bias + ReLU applied, rounded to bf16) into a
    // [pixel][channel] image of the tile, then the block streams whole pixel rows out with 16-byte stores
    // (a wave-instruction covers 4 x 256-B / 8 x 128-B contiguous runs instead of 32 scattered 16-B pieces --
    // the scattered form kept the last waves in the store queue for 15-18k cycles, a quarter of the block's life).
    // 16-B chunk c of row r sits at chunk c ^ (r % CPR): conflict-light 8-B writes, conflict-free 16-B reads.
    constexpr int CPR = BN / 8;                  // 16-B chunks per tile row
    char* tile = smem;                           // BM * BN * 2 bytes <= one stage
    // row table (second stage buffer, free now): element offset of each tile row in y and in the mask, -1 past M
  int* rowY = reinterpret_cast<int*>(smem + STAGE);
  int* rowM = rowY + BM;
  if (tid < BM) {
    const int m = m0 + tid;
    int oy_ = -1, om_ = -1;
    if (m < p.M) {
      const int hw = p.Ho * p.Wo;
      const int n = m / hw, rem = m - n * hw;
      int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      if (p.scatter) { oy *= 2; ox *= 2; }
      oy_ = (int)(n * p.ysN + oy * p.ysH + ox * p.ysW);
      om_ = (int)(n * p.msN + oy * p.msH + ox * p.msW);
    }
    rowY[tid] = oy_; rowM[tid] = om_;
  }

    // bias vectors from the LDS copy made in the prologue (or, without one, in ONE batch of global loads)
    f32x4 bv[TN][4];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (BIAS_LDS) {
          bv[i][q] = *reinterpret_cast<const f32x4*>(biasL + wn * WTN + i * 32 + 8 * q + 4 * half);
        } else {
          bv[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.bias) {
            int co = n0 + wn * WTN + i * 32 + 8 * q + 4 * half;
            if (p.scatter) co %= p.cout;
            bv[i][q] = *reinterpret_cast<const f32x4*>(p.bias + co);
          }
        }
      }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int row = wm * WTM + j * 32 + l31;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cl = wn * WTN + i * 32 + 8 * q + 4 * half;       // channel within the tile
          float v[4] = {acc[i][j][4 * q + 0] + bv[i][q][0], acc[i][j][4 * q + 1] + bv[i][q][1],
                        acc[i][j][4 * q + 2] + bv[i][q][2], acc[i][j][4 * q + 3] + bv[i][q][3]};
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu1(v[e]);
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          const int chunk = (cl >> 3) ^ (row & (CPR - 1));
          *reinterpret_cast<bf16x4*>(tile + row * (BN * 2) + chunk * 16 + (cl & 4) * 2) = o;
        }
      }
    }
    __syncthreads();
    constexpr int NCH = BM * CPR / (NW * 64);    // chunks per thread
    // the mask loads of all NCH chunks go out together (one memory round trip instead of NCH), then the stores
    long long yoff[NCH];
    bf16x8 mk[NCH];
    unsigned mb[NCH];
    bool use_mask[NCH];
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
      const int id = t * (NW * 64) + tid;
      const int row = id / CPR, cc = id % CPR;
      const int yo = rowY[row];
      yoff[t] = -1;
      use_mask[t] = false;
      if (yo < 0) continue;
      int c = n0 + cc * 8, co = c;
      long long off = yo;
      long long moff = rowM[row];
      if (p.scatter) {
        const int ab = c / p.cout;
        co = c - ab * p.cout;
        off += (ab >> 1) * p.ysH + (ab & 1) * p.ysW;
        moff += (ab >> 1) * p.msH + (ab & 1) * p.msW;
      }
      yoff[t] = off + co;
      if (p.mask_bits) mb[t] = p.mask_bits[(unsigned long long)(moff + co) >> 3];
      else if (p.mask && co < p.mask_channels) {
        use_mask[t] = true;
        mk[t] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.mask) + moff + co);
      }
    }
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
      if (yoff[t] < 0) continue;
      const int id = t * (NW * 64) + tid;
      const int row = id / CPR, cc = id % CPR;
      bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + row * (BN * 2) + ((cc ^ (row & (CPR - 1))) * 16));
      if (p.mask_bits) chunk_gate_bits(v, mb[t], p.mask_scale);
      else if (use_mask[t]) chunk_gate_act(v, mk[t], p.mask_scale);
      dct_store16_stream(reinterpret_cast<bf16_t*>(p.y) + yoff[t], v);
      if (p.bits_out) p.bits_out[(unsigned long long)yoff[t] >> 3] = (unsigned char)relu_bits8(v);
    }
    return;
  }
  if (p.partial) {
    // split-K slab: the fp32 tile goes through LDS the same way (BM * BN * 4 bytes = both stage buffers), so the
    // slab rows (BN * 4 contiguous bytes each) are written with full-row 16-byte stores
    constexpr int CPR4 = BN / 4;                 // 16-B chunks (4 floats) per tile row
    char* tile = smem;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int row = wm * WTM + j * 32 + l31;
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cl = wn * WTN + i * 32 + 8 * q + 4 * half;
          const int chunk = (cl >> 2) ^ (row & (CPR4 - 1));
          *reinterpret_cast<f32x4*>(tile + row * (BN * 4) + chunk * 16) =
              f32x4{acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        }
    }
    __syncthreads();
    constexpr int NCH4 = BM * CPR4 / (NW * 64);
    float* slab = p.partial + ((long long)bzi * p.M + m0) * p.N + n0;
#pragma unroll
    for (int t = 0; t < NCH4; ++t) {
      const int id = t * (NW * 64) + tid;
      const int row = id / CPR4, cc = id % CPR4;
      if (m0 + row >= p.M) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * (BN * 4) + ((cc ^ (row & (CPR4 - 1))) * 16));
      *reinterpret_cast<f32x4*>(slab + (long long)row * p.N + cc * 4) = v;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * WTM + j * 32 + l31;
    if (m >= p.M) continue;
    const int hw = p.Ho * p.Wo;
    const int n = m / hw, rem = m - n * hw;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n0 + wn * WTN + i * 32 + 8 * q + 4 * half;
        float v[4] = {acc[i][j][4 * q + 0], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        epilogue_store4<bf16_t>(p, n, oy, ox, c, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 3x3 stride-1 kernel with a shared input halo ("v3").  At 128 x 128 x 64 per K-step the v2 kernel moves
// 32 KiB from L2 through the CU's vector-memory path into LDS per 2.1 MFLOP = 64 FLOP/B, which is exactly the
// CU's ratio of MFMA rate (4069 FLOP/clk) to L1 fill rate (64 B/clk): it cannot run much beyond half the MFMA
// peak however the loop is scheduled.  Nine taps of a 3x3 filter read the SAME input pixels shifted by one, so
// here a block owns an 8 x 16 patch of output pixels of one image and stages the 10 x 18 input halo of a 64-channel
// slice ONCE for all nine taps (22.5 KiB per nine K-steps instead of 16 KiB per step); only the weight tile of
// the tap (BN rows x 128 B) streams every step: (BN * 128 + 2.5 KiB) per step = 113 FLOP/B at BN = 128.
// A tap's pixel fragment is the same LDS image read at rows (py + r) * 18 + px + s.  On v_mfma_f32_16x16x32_bf16 (MI355X holds a
// higher clock on this shape under load than on 32x32x16: MI355X_MICROARCH.md, DVFS item 7; the 32x32x16 form of this kernel was
// 8 % behind in isolation, level on the step, and is gone) a fragment's 16 lanes are 16 consecutive pixels of one patch row.
// K order: channel slice outermost, taps inside; weights [cout][tap][cin] as for v2; LDS-staged epilogue as v2,
// plus read-modify-write for accumulate.
// ABUFS: halo stages -- 1 when Cin == 64 (a single channel slice: nothing to prefetch), which lets four
// BN = 64 blocks (or two BN = 128 blocks) share a CU's LDS.
// Variants of this tile that were measured and removed: a four-slot ring of 32-channel weight half-stages with
// counted vmcnt and raw barriers (-15 %: twice the barriers for the same MFMAs), four waves of 64 x 64 instead of eight of
// 32 x 64 (-8 %), three unrolled taps per loop trip (-13 %), an XCD-aware tile order (level), amdgpu_waves_per_eu(4) (-1 %).
// UNPOOL (round 5): the input is an encoder block's un-pooled gradient that nobody has written: x = the gradient at the pooled tensor,
// up_codes = the pooling's routing codes.  The 10 x 18 halo of a slice is 5 x 9 whole windows (patches start at multiples of
// (8, 16) and the data gradient's padding is 2: even coordinates); an item = one window x 8 channels -- 16 bytes of gradient and 8
// code bytes through registers, four 16-byte LDS writes -- takes the place of the halo's LDS-DMA pieces.  A window outside the pooled
// tensor stages zeros (the data gradient's padding).  8.6 KB per halo instead of 22.5 KB, and the 130 MB un-pooled tensor of the
// first level (with its producer launch) is gone.
template <int BN, int NWM, int NWN, int ABUFS, bool UNPOOL>
__global__ __launch_bounds__(NWM * NWN * 64) void igemm3m_kernel(IgemmParams p, int tiles_x, int tiles_y) {
  constexpr int NW = NWM * NWN;
  constexpr int TH = 8, TW = 16, BM = TH * TW, HW = TW + 2, HROWS = (TH + 2) * HW;   // 180 halo rows of 128 B
  constexpr int APIECES = (HROWS + 7) / 8, A_BYTES = APIECES * 8 * 128;              // 23 pieces, 23552 B
  constexpr int B_BYTES = BN * 128, BPIECES = BN / 8;
  constexpr int NPA = (APIECES + NW - 1) / NW, NPB = BPIECES / NW;
  constexpr int WTN = BN / NWN, TN = WTN / 32;
  static_assert(NWM == 4 && BPIECES % NW == 0 && TN >= 1, "tile/wave mismatch");
  constexpr int PXB = 8 / NWM;             // 16-pixel column blocks (patch rows) per wave: 2
  extern __shared__ __attribute__((aligned(128))) char smem[];
  char* Abuf = smem;                       // halo stage(s)
  char* Bbuf = smem + ABUFS * A_BYTES;     // two weight stages

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: LDS-DMA destinations and piece bookkeeping stay in SGPRs
  const int wn = wave / NWM, wm = wave % NWM;
  int bx = blockIdx.x;
  const int ntile = blockIdx.y;
  const int tx = bx % tiles_x; bx /= tiles_x;
  const int ty = bx % tiles_y; const int img = bx / tiles_y;
  const int y0 = ty * TH, x0 = tx * TW, n0 = ntile * BN;
  const long long Ktot = 9ll * p.Cin;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
  const char* zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 7) * 16;

  // halo staging: wave w issues pieces w, w + NW, ...; lane -> (row = piece * 8 + lane / 8, its swizzled source chunk)
  int aoff[NPA];                      // element offsets (the host admits this kernel only when x spans < 2^31 elements)
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int piece = wave + i * NW;
    const int row = piece * 8 + (lane >> 3);
    aoff[i] = -1;
    if (piece < APIECES && row < HROWS) {
      const int hy = row / HW, hx = row - hy * HW;
      const int iy = y0 - p.pad_h + hy, ix = x0 - p.pad_w + hx;
      if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
        aoff[i] = (int)(img * p.xsN + iy * p.xsH + ix * p.xsW + (((lane & 7) ^ ((row >> 1) & 7)) * 8));
    }
  }
  auto stageA = [&](char* buf, int c0) {
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int piece = wave + i * NW;
      if (piece < APIECES) {
        const char* src = aoff[i] >= 0 ? reinterpret_cast<const char*>(xb + aoff[i] + c0) : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + piece * 1024), 16, 0, 0);
      }
    }
  };
  // UNPOOL: this thread's windows of the halo.  The pooled gradient is DENSE (host-checked), so the element offset of an item's eight
  // gradients is also the byte offset of its eight codes: one register per item (-1: no item, -2: an item that stages zeros).
  constexpr int UITEMS = UNPOOL ? (45 * 8 + NW * 64 - 1) / (NW * 64) : 1;
  int ugoff[UITEMS];
  uint4 ug[UITEMS];
  uint2 uc[UITEMS];
  if constexpr (UNPOOL) {
#pragma unroll
    for (int i = 0; i < UITEMS; ++i) {
      const int id = tid + i * NW * 64;
      const int w = id >> 3, wy = w / 9, wx = w - wy * 9;
      ugoff[i] = -1;
      if (id < 45 * 8) {
        const int py = (y0 - p.pad_h) / 2 + wy, px = (x0 - p.pad_w) / 2 + wx;       // (exact: even numerators)
        ugoff[i] = ((unsigned)py < (unsigned)p.up_Hp && (unsigned)px < (unsigned)p.up_Wp)
                       ? (int)(((long long)(img * p.up_Hp + py) * p.up_Wp + px) * p.Cin + (id & 7) * 8) : -2;
      }
    }
  }
  auto unpool_load = [&](int c0) {
#pragma unroll
    for (int i = 0; i < UITEMS; ++i) {
      ug[i] = make_uint4(0u, 0u, 0u, 0u); uc[i] = make_uint2(0x08080808u, 0x08080808u);       // code 8: routed nowhere
      if (ugoff[i] >= 0) {
        ug[i] = *reinterpret_cast<const uint4*>(xb + ugoff[i] + c0);
        uc[i] = *reinterpret_cast<const uint2*>(p.up_codes + ugoff[i] + c0);
      }
    }
  };
  auto unpool_store = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < UITEMS; ++i) {
      if (ugoff[i] == -1) continue;
      const int id = tid + i * NW * 64;
      const int w = id >> 3, c8 = id & 7, wy = w / 9, wx = w - wy * 9;
      const int row0 = (2 * wy) * HW + 2 * wx;                   // halo row of the window's first position
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = row0 + (k >> 1) * HW + (k & 1);
        *reinterpret_cast<uint4*>(buf + row * 128 + ((c8 ^ ((row >> 1) & 7)) * 16)) = dct_unpool_chunk8(ug[i], uc[i], (unsigned)k);
      }
    }
  };
  // weight staging: piece = 8 cout rows
  const bf16_t* wsrc[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int row = (wave + i * NW) * 8 + (lane >> 3);
    wsrc[i] = reinterpret_cast<const bf16_t*>(p.w) + (long long)(n0 + row) * Ktot + (((lane & 7) ^ ((row >> 1) & 7)) * 8);
  }
  auto stageB = [&](char* buf, int tap, int c0) {
    const long long woff = (long long)tap * p.Cin + c0;
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + woff), (lptr_t)(buf + (wave + i * NW) * 1024), 16, 0, 0);
  };

  constexpr int TR = WTN / 16;                 // 16-channel row blocks per wave
  f32x4 acc[TR][PXB];
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < PXB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // v_mfma_f32_16x16x32_bf16 fragments: lane l holds row / column l % 16 and the 16-byte K chunk l / 16 of a 32-deep
  // sub-step.  Column block j of the wave's 32 pixels is patch row 2 * wm + j, columns 0..15 in lane order.
  const int l15 = lane & 15, kq = lane >> 4;
  const int rho0 = (PXB * wm) * HW + l15;
  const int aswz = (l15 >> 1) & 7;

  const int nch = p.Cin / 64;
  const unsigned smem_l = (unsigned)(size_t)(lptr_c)(smem);
  // stem weight gradient from the tile (epilogue): its gate bytes and its x-patch value are fetched here, behind the K loop
  unsigned stem_gates[PXB] = {};
  float stem_xv = 0.f;
  if constexpr (BN == 64 && NWN == 1 && ABUFS == 1) {
    if (p.stem_x) {
#pragma unroll
      for (int j = 0; j < PXB; ++j) {
        const int oy = y0 + PXB * wm + j, ox = x0 + (lane & 15);
        if (oy < p.Ho && ox < p.Wo) {
          const long long om = (long long)img * p.msN + (long long)oy * p.msH + (long long)ox * p.msW;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            stem_gates[j] |= (unsigned)p.mask_bits[(unsigned long long)(om + i * 16 + 4 * (lane >> 4)) >> 3] << (8 * i);
        }
      }
      if (tid < 180) {
        const int hy = tid / 18, hx = tid - hy * 18;
        const int iy = y0 + hy, ix = x0 + hx;
        if (iy < p.Ho + 2 && ix < p.Wo + 2) stem_xv = p.stem_x[((long long)img * (p.Ho + 2) + iy) * (p.Wo + 2) + ix];
      }
    }
  }
  if constexpr (UNPOOL) unpool_load(0); else stageA(Abuf, 0);
  stageB(Bbuf, 0, 0);
  float* biasL = reinterpret_cast<float*>(smem + ABUFS * A_BYTES + 2 * B_BYTES);   // bias -> LDS now: no memory round trip in the epilogue
  if (tid < BN) biasL[tid] = p.bias ? p.bias[n0 + tid] : 0.f;
  if constexpr (UNPOOL) unpool_store(Abuf);
  __syncthreads();
  // Fragment addresses with as little vector ALU work as the layout allows.  Weights: a lane's two sub-step addresses are
  // constants of the lane (+ the stage, a scalar); the wave's 16-row blocks are reached through the read's immediate offset.
  // Pixels: LDS row pi = rho0 + (tap offset, a literal once the nine taps are unrolled), 16-byte chunk (4 * k2 + kq) ^ ((pi >> 1) & 7);
  // the second sub-step is the first with address bit 6 flipped, the second column block is HW rows further.
  const unsigned Wb0 = smem_l + ABUFS * A_BYTES + (wn * WTN + l15) * 128 + ((kq ^ aswz) * 16);
                                                             // (second sub-step: chunk (4 + kq) ^ aswz = the first address with bit 6 flipped)
  // Round 5: a wave whose two patch rows both lie below the image (the last patch row of a 57-row image holds ONE image row: three of the
  // four pixel waves have nothing) keeps staging and the barriers but issues no fragment reads and no MFMAs -- its accumulators are never
  // stored.  2-9 % of the matrix and LDS work of the layers whose height is not a multiple of 8, off the resources the co-resident
  // block and the other model's kernels share.  Bit-identical.
  const bool wave_live = y0 + PXB * wm < p.Ho;
  int ab = 0, bb = 0;
  for (int c = 0; c < nch; ++c) {
    const unsigned Xs = smem_l + ab * A_BYTES;                // scalar
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
      const int sx = t % 3, rr = t / 3;
      if (t < 8) stageB(Bbuf + (bb ^ 1) * B_BYTES, t + 1, c * 64);
      else if (c + 1 < nch) stageB(Bbuf + (bb ^ 1) * B_BYTES, 0, (c + 1) * 64);
      if constexpr (UNPOOL) {
        // the next slice's windows: loads in flight over taps 3..8, expanded into the free halo stage in front of the slice's last barrier
        if (ABUFS == 2 && t == 3 && c + 1 < nch) unpool_load((c + 1) * 64);
      } else {
        if (ABUFS == 2 && t == 0 && c + 1 < nch) stageA(Abuf + (ab ^ 1) * A_BYTES, (c + 1) * 64);
      }
      if (wave_live) {
      const unsigned wa0 = Wb0 + bb * B_BYTES, wa1 = wa0 ^ 64u;
      unsigned rho_t = rho0;
      asm volatile("" : "+v"(rho_t));                        // recompute the tap's two addresses here (6 VALU) instead of keeping hoisted ones in registers
      const unsigned pi0 = rho_t + rr * HW + sx;
      const unsigned xrow = Xs + (pi0 << 7);
      unsigned xj[PXB];                                      // column block j: LDS row pi0 + j * HW (+ j * HW * 128 through the read's offset)
#pragma unroll
      for (int j = 0; j < PXB; ++j) xj[j] = xrow | (((kq ^ ((pi0 + j * HW) >> 1)) & 7) << 4);
      bf16x8 a[2][TR], b[2][PXB];
      RdRows<0, TR, 16 * 128>::run(wa0, a[0]);
      RdCols<0, PXB, HW * 128>::run(xj, 0u, b[0]);
      RdRows<0, TR, 16 * 128>::run(wa1, a[1]);
      RdCols<0, PXB, HW * 128>::run(xj, 64u, b[1]);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        if (k2 == 0) lgkm_wait3<TR + PXB>(); else lgkm_wait3<0>();
#pragma unroll
        for (int i = 0; i < TR; ++i) touch8(a[k2][i]);
#pragma unroll
        for (int j = 0; j < PXB; ++j) touch8(b[k2][j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TR; ++i)
#pragma unroll
          for (int j = 0; j < PXB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k2][i], b[k2][j], acc[i][j], 0, 0, 0);
      }
      }
      if constexpr (UNPOOL) {
        if (ABUFS == 2 && t == 8 && c + 1 < nch) unpool_store(Abuf + (ab ^ 1) * A_BYTES);
      }
      __syncthreads();
      bb ^= 1;
    }
    ab ^= 1;
  }

  // ---- epilogue (staged through LDS as in v2): tile row = patch pixel py * 16 + px
  constexpr int CPR = BN / 8;
  static_assert(BM * BN * 2 <= ABUFS * A_BYTES + (BN * 128 * 2 - BM * 8), "epilogue tile does not fit");
  char* tile = smem;                                   // BM * BN * 2 bytes: the halo stage(s) and, if needed, the head of the weight stages
  if constexpr (BN == 64 && NWN == 1 && ABUFS == 1) {
    if (p.stem_x) {
      // The stem's weight gradient from this tile (include/dct.h dct_conv_desc.stem_x): this launch is the data gradient of the UNet's second
      // convolution, its output -- masked by the ReLU gate of the stem's output, rounded to bf16 -- is the stem's dy, and nobody else
      // reads it.  dW[c][t] = sum_px dy[px][c] * x[px + t] over the block's 128 pixels is a [16 taps x 128] x [128 x 64] product:
      // v_mfma_f32_16x16x32_bf16 with A = the x windows (bf16 high + low parts, row 9 = ones for the bias gradient), B = the tile read
      // back through the transposing LDS read (K = pixels), one 32-pixel K block per wave, the four waves' results added through LDS.
      asm volatile("" ::: "memory");
      constexpr int XS = 32 * 1024, RED = 16 * 1024;        // LDS: x patch parts behind the reduction buffer, both in dead stage space
      unsigned short* xs = reinterpret_cast<unsigned short*>(smem + XS);     // [2][180]
      if (tid < 180) {
        const bf16_t vh = (bf16_t)stem_xv;
        const bf16_t vl = (bf16_t)(stem_xv - (float)vh);
        xs[tid] = __builtin_bit_cast(unsigned short, vh);
        xs[180 + tid] = __builtin_bit_cast(unsigned short, vl);
      }
      // masked accumulators -> tile.  Accumulator (i, j): channels 16 i + 4 kq + {0..3} of pixel (patch row 2 wm + j, column l15).
#pragma unroll
      for (int j = 0; j < PXB; ++j) {
        const int py = PXB * wm + j;
        const int row = py * TW + l15;
#pragma unroll
        for (int i = 0; i < TR; ++i) {
          const int cl = i * 16 + 4 * kq;
          const unsigned gate = (stem_gates[j] >> (8 * i + (cl & 4))) & 15u;
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(((gate >> e) & 1u) ? acc[i][j][e] : 0.f);
          const int chunk = (cl >> 3) ^ (row & (CPR - 1));
          *reinterpret_cast<bf16x4*>(tile + row * (BN * 2) + chunk * 16 + (cl & 4) * 2) = o;
        }
      }
      __syncthreads();
      const int kb = wave;                                   // this wave's 32 tile rows (two patch rows)
      // A: row l15 = tap (9: ones, 10..15: zero), K chunk kq = pixels 8 kq .. 8 kq + 7 of the block: patch row 2 kb + (kq >> 1), columns 8 (kq & 1) + j
      union { bf16x8 v; unsigned short u[8]; } ah, al;
      {
        const int t = l15 < 9 ? l15 : 0;
        const int idx0 = (2 * kb + (kq >> 1) + t / 3) * 18 + 8 * (kq & 1) + t % 3;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          ah.u[jj] = l15 < 9 ? xs[idx0 + jj] : (l15 == 9 ? (unsigned short)0x3f80 : (unsigned short)0);
          al.u[jj] = l15 < 9 ? xs[180 + idx0 + jj] : (unsigned short)0;
        }
      }
      // B: column l15 = channel 16 nb + l15, K chunk kq: two transposing reads of 4 rows x 16 columns each; lane 4 q + pp of a 16-lane
      // group supplies the address of row q, columns 4 pp .. 4 pp + 3 (cdna_hip_programming.md T10)
      const int tq = l15 >> 2, tp = l15 & 3;
      f32x4 sacc[4];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        bf16x4 b0, b1;
        const int r0 = 32 * kb + 8 * kq + tq, r1 = r0 + 4;
        const int ch = 2 * nb + (tp >> 1);
        const unsigned a0 = smem_l + r0 * (BN * 2) + ((ch ^ (r0 & (CPR - 1))) * 16) + (tp & 1) * 8;
        const unsigned a1 = smem_l + r1 * (BN * 2) + ((ch ^ (r1 & (CPR - 1))) * 16) + (tp & 1) * 8;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0) : "v"(a0));
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b1) : "v"(a1));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bf16x8 bb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { bb[e] = b0[e]; bb[4 + e] = b1[e]; }
        touch8(bb);
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al.v, bb, z, 0, 0, 0);
        sacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah.v, bb, z, 0, 0, 0);
      }
      // D (nb): rows 4 kq + e = taps, column l15 = channel 16 nb + l15.  Add the four waves through LDS.
      float* red = reinterpret_cast<float*>(smem + RED);          // [wave][nb][lane][4]
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) *reinterpret_cast<f32x4*>(red + ((wave * 4 + nb) * 64 + lane) * 4) = sacc[nb];
      __syncthreads();
      float* slab = p.stem_slab + (long long)blockIdx.x * 640;     // [channel][10]: nine taps, then the bias gradient
      for (int idx = tid; idx < 640; idx += NW * 64) {
        const int ch = idx / 10, t = idx - ch * 10;
        const int o = ((ch >> 4) * 64 + (t >> 2) * 16 + (ch & 15)) * 4 + (t & 3);
        slab[idx] = (red[o] + red[o + 1024]) + (red[o + 2048] + red[o + 3072]);
      }
      return;
    }
  }
  int* rowY = reinterpret_cast<int*>(smem + ABUFS * A_BYTES + 2 * B_BYTES - BM * 8);   // tail of the weight stages
  int* rowM = rowY + BM;
  if (tid < BM) {
    const int oy = y0 + (tid >> 4), ox = x0 + (tid & 15);
    int oy_ = -1, om_ = -1;
    if (oy < p.Ho && ox < p.Wo) {
      oy_ = (int)(img * p.ysN + oy * p.ysH + ox * p.ysW);
      om_ = (int)(img * p.msN + oy * p.msH + ox * p.msW);
    }
    rowY[tid] = oy_; rowM[tid] = om_;
  }
  {
    // accumulator (i, j): channels wn * WTN + 16 * i + 4 * kq + {0..3} of pixel (patch row PXB * wm + j, column l15)
    f32x4 bv[TR];
#pragma unroll
    for (int i = 0; i < TR; ++i) bv[i] = *reinterpret_cast<const f32x4*>(biasL + wn * WTN + i * 16 + 4 * kq);
#pragma unroll
    for (int j = 0; j < PXB; ++j) {
      const int row = (PXB * wm + j) * TW + l15;
#pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int cl = wn * WTN + i * 16 + 4 * kq;
        float v[4] = {acc[i][j][0] + bv[i][0], acc[i][j][1] + bv[i][1], acc[i][j][2] + bv[i][2], acc[i][j][3] + bv[i][3]};
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = relu1(v[e]);
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
        const int chunk = (cl >> 3) ^ (row & (CPR - 1));
        *reinterpret_cast<bf16x4*>(tile + row * (BN * 2) + chunk * 16 + (cl & 4) * 2) = o;
      }
    }
  }
  __syncthreads();
  if (p.pool_y) {
    asm volatile("" ::: "memory");       // a real branch (the forward launches of three encoder levels take it, nobody else)
    staged_pool_out<BN, NW>(p, tile, img, y0, x0, n0, tid);
    if (p.pool_only) return;
  }
  staged_rows_out<BM, BN, NW>(p, tile, rowY, rowM, n0, tid);
}

// ---------------------------------------------------------------------------------------------
// Shared-halo kernel for SMALL images ("v3p"): the 128-pixel tile is PR whole output rows of one image, packed in LDS
// at a pitch of Wo + 2 rows (pixel m = (m / Wo, m % Wo) sits at LDS row (m / Wo) * pitch + m % Wo + tap offset
// r * pitch + s), so the halo of a 64-channel slice is (PR + 2) * (Wo + 2) <= 192 rows, staged once for nine taps.
// These layers have few tiles (16 images x 1-7 tiles x Cout / 128), so the channel slices are split over blockIdx.z
// and the fp32 partial tiles go to slabs folded by splitk_epilogue_kernel (fixed order, as v2's split-K).
// LEAN (round 4): the same loop with a third fewer instructions per K-step (tools/isa_loop_mix.py: 8 MFMAs carried 20-29 vector and
// 30-33 scalar instructions; at one block per CU -- the deep levels -- a wave's own instruction stream, not the matrix pipe, set
// the step).  Halo and weight pieces are buffer loads to LDS with constant lane offsets (out-of-image halo lanes hold an offset the
// descriptor rejects and stage zeros), the step's offsets are 32-bit scalars advanced by adds, tap offsets advance without a
// division.  Same LDS image, same reads, same MFMA order: bit-identical.
template <int BN, int NWM, int NWN, bool LEAN>
__global__ __launch_bounds__(NWM * NWN * 64) void igemm3p_kernel(IgemmParams p, int PR, int tiles_per_img, int chunks_per_split) {
  constexpr int NW = NWM * NWN;
  constexpr int BM = 128, APIECES = 24, A_BYTES = APIECES * 1024;                  // <= 192 halo rows of 128 B
  constexpr int B_BYTES = BN * 128, BPIECES = BN / 8;
  constexpr int NPA = APIECES / NW, NPB = BPIECES / NW;
  constexpr int WTN = BN / NWN, TN = WTN / 32;
  static_assert(NWM == 4 && APIECES % NW == 0 && BPIECES % NW == 0 && TN >= 1, "tile/wave mismatch");
  extern __shared__ __attribute__((aligned(128))) char smem[];
  char* Abuf = smem;
  char* Bbuf = smem + 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / NWM, wm = wave % NWM;
  int bxi, byi, bzi;
  if (!xcd_remap(p, bxi, byi, bzi)) return;
  const int img = bxi / tiles_per_img, ty = bxi - img * tiles_per_img;
  const int y0 = ty * PR, n0 = byi * BN;
  const int pitch = p.Wo + 2, hrows = (PR + 2) * pitch;
  const long long Ktot = 9ll * p.Cin;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
  const char* zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 7) * 16;
  const unsigned smem_l = (unsigned)(size_t)(lptr_c)(smem);

  long long aoff[NPA];
  int aoffb[NPA];                                                // LEAN: the same as a byte offset, or the rejected offset
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int row = (wave + i * NW) * 8 + (lane >> 3);
    aoff[i] = -1;
    if (row < hrows) {
      const int hy = row / pitch, hx = row - hy * pitch;
      const int iy = y0 - p.pad_h + hy, ix = hx - p.pad_w;
      if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
        aoff[i] = img * p.xsN + iy * p.xsH + ix * p.xsW + (((lane & 7) ^ ((row >> 1) & 7)) * 8);
    }
    aoffb[i] = aoff[i] >= 0 ? (int)(aoff[i] * 2) : DCT_BUF_INVALID;
  }
  __amdgpu_buffer_rsrc_t rsX, rsW;
  if constexpr (LEAN) {
    rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (long long)n0 * Ktot * 2), 0, (int)(p.w_bytes - (long long)n0 * Ktot * 2), 0x00020000);
  }
  auto stageA = [&](char* buf, int c0) {
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const char* src = aoff[i] >= 0 ? reinterpret_cast<const char*>(xb + aoff[i] + c0) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(buf + (wave + i * NW) * 1024), 16, 0, 0);
    }
  };
  const bf16_t* wsrc[NPB];
  int woffb[NPB];                                                // LEAN: byte offset of the lane's chunk from the tile's first weight row
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int row = (wave + i * NW) * 8 + (lane >> 3);
    wsrc[i] = reinterpret_cast<const bf16_t*>(p.w) + (long long)(n0 + row) * Ktot + (((lane & 7) ^ ((row >> 1) & 7)) * 8);
    woffb[i] = (int)(((long long)row * Ktot + (((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2);
  }
  auto stageB = [&](char* buf, int tap, int c0) {
    const long long woff = (long long)tap * p.Cin + c0;
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + woff), (lptr_t)(buf + (wave + i * NW) * 1024), 16, 0, 0);
  };

  f32x16 acc[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const int half = lane >> 5, l31 = lane & 31;
  const int mpix = wm * 32 + l31;                              // pixel slot of this lane's fragment column
  int rho0;
  {
    const int pr = mpix / p.Wo, pc = mpix - pr * p.Wo;
    rho0 = pr < PR ? pr * pitch + pc : 0;                      // slots past the tile read row 0 (never stored)
  }
  const int aswz = (l31 >> 1) & 7;

  const int nch = p.Cin / 64;
  const int cbeg = bzi * chunks_per_split, cend = min(nch, cbeg + chunks_per_split);
  if constexpr (LEAN) {
    // LDS byte addresses of this wave's first halo / weight piece in stage 0 (pieces of one kind are NW KiB apart)
    const unsigned ldsA = smem_l + wave * 1024, ldsB = smem_l + 2 * A_BYTES + wave * 1024;
    auto stA = [&](int ab_, unsigned soff) {
#pragma unroll
      for (int i = 0; i < NPA; ++i) buf_lds16(rsX, ldsA + ab_ * A_BYTES + i * NW * 1024, aoffb[i], soff);
    };
    auto stB = [&](int bb_, unsigned soff) {
#pragma unroll
      for (int i = 0; i < NPB; ++i) buf_lds16(rsW, ldsB + bb_ * B_BYTES + i * NW * 1024, woffb[i], soff);
    };
    const unsigned cin2 = (unsigned)p.Cin * 2;
    unsigned xo = (unsigned)cbeg * 128;                         // byte offset of the current channel slice in a pixel
    unsigned wo = xo;                                           // (tap * Cin + c * 64) * 2 of the NEXT weight stage to fetch
    stA(0, xo);
    stB(0, wo);
    __syncthreads();
    // lane constants of the fragment reads: weight rows at a fixed place, pixel rows move with the tap (rho = rho0 + r * pitch + s)
    const unsigned Wl0 = smem_l + 2 * A_BYTES + (wn * WTN + l31) * 128 + ((half ^ aswz) * 16);
    int ab = 0, bb = 0;
    const int nsteps = 9 * (cend - cbeg);
    // Round 5: a wave whose 32 pixel slots all lie past the tile's PR * Wo pixels (two rows of 46 pixels fill 92 of 128 slots: the
    // fourth pixel wave has nothing) stages and keeps the barriers but issues no fragment reads and no MFMAs.  Bit-identical.
    const bool wave_live = wm * 32 < PR * p.Wo;
    int t = 0, tapoff = 0, scol = 0;                            // tap index, r * pitch + s, s
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
      // next weight stage: the following tap of this slice, or tap 0 of the next slice
      if (k + 1 < nsteps) {
        if (t < 8) wo += cin2; else wo += 128u - 8u * cin2;
        stB(bb ^ 1, wo);
      }
      if (t == 0 && k + 9 < nsteps) stA(ab ^ 1, xo + 128u);
      if (wave_live) {
      const int rho = rho0 + tapoff;
      const unsigned Wl = Wl0 + bb * B_BYTES;
      const unsigned Xl = smem_l + ab * A_BYTES + rho * 128 + ((half ^ ((rho >> 1) & 7)) * 16);
      bf16x8 a[2][TN], b[2];
      auto issue = [&](int set, int kk) {
#pragma unroll
        for (int i = 0; i < TN; ++i) rd128((Wl + i * 32 * 128) ^ (kk * 32), a[set][i]);
        rd128(Xl ^ (kk * 32), b[set]);
      };
      issue(0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int set = kk & 1;
        if (kk + 1 < 4) { issue(set ^ 1, kk + 1); lgkm_wait3<TN + 1>(); } else { lgkm_wait3<0>(); }
#pragma unroll
        for (int i = 0; i < TN; ++i) touch8(a[set][i]);
        touch8(b[set]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[set][i], b[set], acc[i], 0, 0, 0);
      }
      }
      __syncthreads();
      bb ^= 1;
      // next tap without branches: s wraps every third step (one halo row down, two columns back), the slice every ninth
      ++t; ++scol;
      const bool w3 = scol == 3, w9 = t == 9;
      tapoff += w3 ? pitch - 2 : 1;
      scol = w3 ? 0 : scol;
      tapoff = w9 ? 0 : tapoff;
      t = w9 ? 0 : t;
      ab ^= w9 ? 1 : 0;
      xo += w9 ? 128u : 0u;
    }
  } else {
  stageA(Abuf, cbeg * 64);
  stageB(Bbuf, 0, cbeg * 64);
  __syncthreads();
  int ab = 0, bb = 0;
  for (int c = cbeg; c < cend; ++c) {
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
      if (t < 8) stageB(Bbuf + (bb ^ 1) * B_BYTES, t + 1, c * 64);
      else if (c + 1 < cend) stageB(Bbuf + (bb ^ 1) * B_BYTES, 0, (c + 1) * 64);
      if (t == 0 && c + 1 < cend) stageA(Abuf + (ab ^ 1) * A_BYTES, (c + 1) * 64);
      const int r = t / 3, s = t - 3 * r;
      const int rho = rho0 + r * pitch + s;
      const int pswz = (rho >> 1) & 7;
      const unsigned Wl = smem_l + 2 * A_BYTES + bb * B_BYTES + (wn * WTN + l31) * 128 + ((half ^ aswz) * 16);
      const unsigned Xl = smem_l + ab * A_BYTES + rho * 128 + ((half ^ pswz) * 16);
      bf16x8 a[2][TN], b[2];
      auto issue = [&](int set, int kk) {
#pragma unroll
        for (int i = 0; i < TN; ++i) rd128((Wl + i * 32 * 128) ^ (kk * 32), a[set][i]);
        rd128(Xl ^ (kk * 32), b[set]);
      };
      issue(0, 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int set = kk & 1;
        if (kk + 1 < 4) { issue(set ^ 1, kk + 1); lgkm_wait3<TN + 1>(); } else { lgkm_wait3<0>(); }
#pragma unroll
        for (int i = 0; i < TN; ++i) touch8(a[set][i]);
        touch8(b[set]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[set][i], b[set], acc[i], 0, 0, 0);
      }
      __syncthreads();
      bb ^= 1;
    }
    ab ^= 1;
  }
  }

  // ---- epilogue: tile row m = pixel slot; row tables behind the tile
  const int npix = PR * p.Wo;
  if (p.partial) {
    // split over channel slices: fp32 tile -> LDS -> whole slab rows (BN * 4 contiguous bytes) with 16-byte stores
    constexpr int CPR4 = BN / 4;
    char* tile = smem;                                   // BM * BN * 4 = 64 KiB of the 80 KiB
    int* rowS = reinterpret_cast<int*>(smem + BM * BN * 4);
    if (tid < BM) {
      const int pr = tid / p.Wo, pc = tid - pr * p.Wo, oy = y0 + pr;
      rowS[tid] = (tid < npix && oy < p.Ho) ? (img * p.Ho + oy) * p.Wo + pc : -1;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cl = wn * WTN + i * 32 + 8 * q + 4 * half;
        const int chunk = (cl >> 2) ^ (mpix & (CPR4 - 1));
        *reinterpret_cast<f32x4*>(tile + mpix * (BN * 4) + chunk * 16) =
            f32x4{acc[i][4 * q + 0], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
      }
    __syncthreads();
    constexpr int NCH4 = BM * CPR4 / (NW * 64);
    float* slab = p.partial + (long long)bzi * p.M * p.N + n0;
#pragma unroll
    for (int t = 0; t < NCH4; ++t) {
      const int id = t * (NW * 64) + tid;
      const int row = id / CPR4, cc = id % CPR4;
      const int mg = rowS[row];
      if (mg < 0) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * (BN * 4) + ((cc ^ (row & (CPR4 - 1))) * 16));
      *reinterpret_cast<f32x4*>(slab + (long long)mg * p.N + cc * 4) = v;
    }
    return;
  }
  constexpr int CPR = BN / 8;
  char* tile = smem;
  int* rowY = reinterpret_cast<int*>(smem + BM * BN * 2);
  int* rowM = rowY + BM;
  if (tid < BM) {
    const int pr = tid / p.Wo, pc = tid - pr * p.Wo, oy = y0 + pr;
    int oy_ = -1, om_ = -1;
    if (tid < npix && oy < p.Ho) {
      oy_ = (int)(img * p.ysN + oy * p.ysH + pc * p.ysW);
      om_ = (int)(img * p.msN + oy * p.msH + pc * p.msW);
    }
    rowY[tid] = oy_; rowM[tid] = om_;
  }
  {
    f32x4 bv[TN][4];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          bv[i][q] = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * WTN + i * 32 + 8 * q + 4 * half);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cl = wn * WTN + i * 32 + 8 * q + 4 * half;
        float v[4] = {acc[i][4 * q + 0] + bv[i][q][0], acc[i][4 * q + 1] + bv[i][q][1], acc[i][4 * q + 2] + bv[i][q][2],
                      acc[i][4 * q + 3] + bv[i][q][3]};
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = relu1(v[e]);
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
        const int chunk = (cl >> 3) ^ (mpix & (CPR - 1));
        *reinterpret_cast<bf16x4*>(tile + mpix * (BN * 2) + chunk * 16 + (cl & 4) * 2) = o;
      }
    }
  }
  __syncthreads();
  staged_rows_out<BM, BN, NW>(p, tile, rowY, rowM, n0, tid);
}

// Sum split-K partial slabs and apply the epilogue.  One thread per (pixel, 4 channels).
template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(IgemmParams p, int splits) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int nq = p.N / 4;
  if (idx >= (long long)p.M * nq) return;
  const int m = (int)(idx / nq), c = (int)(idx - (long long)m * nq) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < splits; ++z) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p.partial + ((long long)z * p.M + m) * p.N + c);
    v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
  }
  const int hw = p.Ho * p.Wo;
  const int n = m / hw, rem = m - n * hw;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  IgemmParams q = p;
  q.partial = nullptr;
  epilogue_store4<T>(q, n, oy, ox, c, v);
  if constexpr (sizeof(T) == 2) {
    if (p.bits_out) {
      // ReLU-gate bits of the rounded output (relu_bits_kernel's test, which used to be a launch of its own behind every split layer):
      // this thread's four channels are half a byte, its neighbour (the other half of the same pixel's 8-channel group) supplies the rest
      unsigned nib = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) nib |= ((float)(bf16_t)v[e] > 0.f ? 1u : 0u) << e;
      const unsigned other = __shfl_xor(nib, 1, 64);
      if ((c & 4) == 0) p.bits_out[(unsigned long long)(n * p.ysN + oy * p.ysH + ox * p.ysW + c) >> 3] = (unsigned char)(nib | (other << 4));
    }
  }
}

// ReLU-gate bits of a dense bf16 tensor (the paths whose epilogue does not leave them behind: split-K, unstaged stores)
__global__ __launch_bounds__(256) void relu_bits_kernel(const bf16_t* __restrict__ y, unsigned char* __restrict__ bits, long long chunks) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= chunks) return;
  bits[i] = (unsigned char)relu_bits8(*reinterpret_cast<const bf16x8*>(y + 8 * i));
}

struct Plan {
  int bn, bm, splits, kiters, kiters_per_split, bk;
  int v2;       // 1: igemm2_kernel (bf16, Cin % 64 == 0)
  int bounds;   // v2: some tap can fall outside the input
  long long tiles;
};

// Planner settings (dct_tune_set).  The per-tap tile always runs on eight waves (32 pixels x 64 channels each: an LDS-DMA piece
// costs ~100 issue cycles, so halving the pieces per wave shortened every K-step by 4-15 %) with the LDS-staged epilogue where
// the alignment allows it (scattered 8-byte stores kept the last waves in the store queue for a quarter of a block's life).
int g_tune_igemm_split_target = 450;   // block target of a split layer (tiles < 200)
int g_tune_igemm_split = -1;    // >= 1 forces the split-K factor
int g_tune_igemm_split_max_tiles = 96;   // layers with at least this many tiles run unsplit
int g_tune_igemm_split_min_kiters = 4;    // ... and so do layers with fewer K-steps than this
int g_tune_igemm_halo_cover = 75;      // percent of the image the 8 x 16 patches must cover
int g_tune_igemm_halo_min_blocks = 400;
int g_tune_igemm_pool = 1;      // diagnostic (1005): 0 = a requested pooling always runs as its own launch behind the conv
int g_tune_igemm_halo = 1;      // 3x3 stride-1 layers with large images: shared-halo kernel (igemm3m_kernel); 0: always the per-tap kernel

static bool make_plan(const dct_view* x, const dct_view* y, const dct_conv_desc* d, int dtype, int M, int N, Plan& pl) {
  const int bk0 = dtype == DCT_BF16 ? 32 : 16;
  if (x->c % bk0 != 0) return false;
  if (N % 64 != 0) return false;
  pl.v2 = (dtype == DCT_BF16 && x->c % 64 == 0) ? 1 : 0;
  if (pl.v2) {
    pl.bk = 64;
    if (N % 128 == 0) { pl.bn = 128; pl.bm = 128; } else { pl.bn = 64; pl.bm = 256; }
    const int Ho = d->scatter2x2 ? y->h / 2 : y->h, Wo = d->scatter2x2 ? y->w / 2 : y->w;
    const bool inside = d->pad_h == 0 && d->pad_w == 0 &&
                        (Ho - 1) * d->stride + (d->R - 1) * d->dil < x->h &&
                        (Wo - 1) * d->stride + (d->S - 1) * d->dil < x->w;
    pl.bounds = inside ? 0 : 1;
  } else {
    pl.bk = bk0;
    pl.bn = (N % 128 == 0) ? 128 : 64;
    pl.bm = 128;
    pl.bounds = 1;
  }
  pl.kiters = d->R * d->S * (x->c / pl.bk);
  pl.tiles = (long long)div_up(M, pl.bm) * (N / pl.bn);
  int splits = 1;
  if (pl.v2) {
    // measured on the UNet layer set (tools/bench_conv.py --ab): a layer with >= 200 tiles runs fastest unsplit
    // (two resident blocks per CU interleave); below that ~450 blocks in total is the sweet spot, and each
    // split must keep >= 4 K-steps to amortise its prologue and its fp32 slab
    if (pl.tiles < g_tune_igemm_split_max_tiles && pl.kiters >= g_tune_igemm_split_min_kiters) {
      splits = (int)((g_tune_igemm_split_target + pl.tiles / 2) / pl.tiles);
      if (splits > 8) splits = 8;
      while (splits > 1 && pl.kiters / splits < 4) --splits;
    }
  } else if (pl.tiles < 384) {
    splits = (int)((768 + pl.tiles - 1) / pl.tiles);
    if (splits > 16) splits = 16;
    while (splits > 1 && pl.kiters / splits < 4) --splits;
  }
  if (g_tune_igemm_split >= 1) splits = g_tune_igemm_split;
  if (splits > pl.kiters) splits = pl.kiters;
  pl.kiters_per_split = (pl.kiters + splits - 1) / splits;
  pl.splits = (pl.kiters + pl.kiters_per_split - 1) / pl.kiters_per_split;
  return true;
}

template <int BM, int BN, int WM, int WN, bool BOUNDS, bool LEAN>
static void launch_v2_k(const IgemmParams& p, dim3 grid, hipStream_t st) {
  constexpr size_t stages = 2 * (size_t)(BM + BN) * 128;
  constexpr size_t lds = stages + (stages + BN * 4 <= 80 * 1024 ? (size_t)BN * 4 : 0);     // two stages [+ the tile's bias vector]
  static bool attr_set = false;   // idempotent one-time opt-in to > 64 KiB dynamic LDS
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<BM, BN, WM, WN, BOUNDS, LEAN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  DCT_LAUNCH_FAM(DCT_FAM_IGEMM2, DCT_PROF_IGEMM, (igemm2_kernel<BM, BN, WM, WN, BOUNDS, LEAN>), grid, dim3(WM * WN * 64), lds, st, p);
}
template <int BM, int BN, int WM, int WN, bool BOUNDS>
static void launch_v2(const IgemmParams& p, dim3 grid, hipStream_t st) {
  // lean loop: activations and packed weights under 2 GiB (buffer descriptors), at most 32 taps (one mask word per row)
  if ((g_tune_lean & 8) && p.x_bytes < (1ll << 31) && p.w_bytes < (1ll << 31) && p.R * p.S <= 32)
    launch_v2_k<BM, BN, WM, WN, BOUNDS, true>(p, grid, st);
  else launch_v2_k<BM, BN, WM, WN, BOUNDS, false>(p, grid, st);
}

template <int BN, int NWN, int ABUFS, bool UNPOOL>
static void launch_v3u(const IgemmParams& p, int tiles_x, int tiles_y, int images, hipStream_t st) {
  constexpr size_t lds = ABUFS * (size_t)(23 * 1024) + 2 * (size_t)BN * 128 + (size_t)BN * 4;   // halo stage(s), two weight stages, bias
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm3m_kernel<BN, 4, NWN, ABUFS, UNPOOL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const dim3 grid((unsigned)(images * tiles_y * tiles_x), p.N / BN, 1);
  DCT_LAUNCH_FAM(DCT_FAM_IGEMM3M, DCT_PROF_IGEMM, (igemm3m_kernel<BN, 4, NWN, ABUFS, UNPOOL>), grid, dim3(4 * NWN * 64), lds, st, p, tiles_x, tiles_y);
}
template <int BN, int NWN, int ABUFS>
static void launch_v3(const IgemmParams& p, int tiles_x, int tiles_y, int images, hipStream_t st) {
  if (p.up_codes) launch_v3u<BN, NWN, ABUFS, true>(p, tiles_x, tiles_y, images, st);
  else launch_v3u<BN, NWN, ABUFS, false>(p, tiles_x, tiles_y, images, st);
}

// Packed-rows shared-halo kernel (igemm3p_kernel) for small images: geometry and split over channel slices.
struct PlanP { int use, PR, tiles_per_img, splits, cps; };
int g_tune_igemm_packed = 1;     // packed-rows shared-halo kernel for the deep levels (0: they stay on the per-tap kernel).  Round 1 had it
                                 // level on the step because every layer under 400 blocks was split over channel slices (fp32 slabs);
                                 // with the split only under 100 blocks -- the two model streams supply the parallelism -- it is
                                 // +0.9 / +2.1 / +1.3 % on the cfg2 step (three in-process A/B rounds, tools/ab_step.py --pre 23=100 --knob 10)
int g_tune_igemm_packed_split = 160;   // packed kernel: layers with fewer blocks than this are split over channel slices (fp32 slabs).  Round 3: 100
                                       // (+0.9 / +2.1 / +1.3 % on the step against 400); round 4, after the instruction diet: 160-200 are 1.7 % ahead of
                                       // 100 on the step (5.52 -> 5.43 ms, three alternating graph-mode runs each; 260: 5.45) -- cen_a, enc4a and enc4b
                                       // (128 blocks each) now run as 256
int g_tune_igemm_packed_fill = 50;     // percent: least fill of the packed 128-pixel tiles.  76 until round 5 (the isolated-layer optimum of round 1); on the
                                       // captured cfg2 step 60 / 50 / 40 are 1.0 / 1.25 / 1.2 % ahead of 76 (two alternating rounds on one box,
                                       // profiles/r05_knob_sweep.txt): the centre's second convolution (81-pixel images: 63 %), enc3a / enc2a / enc2b and
                                       // enc3b's data gradient (69-75 %) leave the per-tap kernel, which re-reads its input once per tap
static PlanP make_plan_p(const dct_view* x, const dct_view* y, const dct_conv_desc* d, int dtype, int N) {
  PlanP pp = {0, 0, 0, 1, 0};
  if (!g_tune_igemm_halo || !g_tune_igemm_packed || dtype != DCT_BF16 || d->R != 3 || d->S != 3 || d->stride != 1 || d->dil != 1 ||
      d->scatter2x2 || x->c % 64 || N % 128 || y->w > 126)
    return pp;
  const int Wo = y->w, Ho = y->h;
  int PR = 128 / Wo;
  const int by_lds = 192 / (Wo + 2) - 2;
  if (PR > by_lds) PR = by_lds;
  if (PR > Ho) PR = Ho;
  if (PR < 1) return pp;
  const int tiles = (Ho + PR - 1) / PR;
  // measured on the UNet deep levels (tools/bench_conv.py --ab-packed): +8-12 % where the tiles are >= 76 % full and every
  // block keeps >= 4 channel slices (36 K-steps); shorter blocks or emptier tiles are level with or behind the per-tap kernel
  if ((double)Ho * Wo / (tiles * 128.0) < g_tune_igemm_packed_fill * 0.01) return pp;
  const int nch = x->c / 64;
  const long long blocks0 = (long long)y->n * tiles * (N / 128);
  int splits = 1;
  if (blocks0 < g_tune_igemm_packed_split) {
    splits = (int)((g_tune_igemm_packed_split + 48 + blocks0 - 1) / blocks0);
    if (splits > nch) splits = nch;
    while (splits > 1 && nch / splits < 4) --splits;
  }
  if (nch / splits < 4 || blocks0 * splits < (g_tune_igemm_packed_split < 256 ? g_tune_igemm_packed_split : 256)) return pp;
  pp.cps = (nch + splits - 1) / splits;
  pp.splits = (nch + pp.cps - 1) / pp.cps;
  pp.use = 1; pp.PR = PR; pp.tiles_per_img = tiles;
  return pp;
}
template <bool LEAN>
static void launch_v3p_k(const IgemmParams& p, const PlanP& pp, dim3 grid, hipStream_t st) {
  constexpr size_t lds = 2 * (size_t)(24 * 1024) + 2 * (size_t)128 * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm3p_kernel<128, 4, 2, LEAN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  DCT_LAUNCH_FAM(DCT_FAM_IGEMM3P, DCT_PROF_IGEMM, (igemm3p_kernel<128, 4, 2, LEAN>), grid, dim3(512), lds, st, p, pp.PR, pp.tiles_per_img, pp.cps);
}
// dct_tune_set(DCT_TUNE_IGEMM_XCD): 0 = natural 3-D grids; 1 = the per-tap and packed-rows kernels deal their blocks XCD by XCD
// (xcd_remap) on layers whose packed weights outweigh their activations; 2 = on every layer they run
int g_tune_igemm_xcd = 1;
static dim3 xcd_grid(IgemmParams& p, dim3 grid, bool per_tap) {
  p.xcd_total = 0; p.xcd_gx = p.xcd_gy = 1;
  const long long total = (long long)grid.x * grid.y * grid.z;
  // measured per layer (profiles/r05_xcd_block_order_ab.txt): the per-tap kernel gains where the weights outweigh the activations
  // (the centre's 1024 -> 1024 convolution 53 -> 42 us, its transposed convolution 17.3 -> 15.7); the packed-rows kernel LOSES 2-10 %
  // on the same layers (its XCDs then stream eight different weight slices at once instead of sharing one through the memory-side
  // cache), so it takes the order only when forced (2: the bit-identity test)
  const bool want = g_tune_igemm_xcd == 2 || (g_tune_igemm_xcd == 1 && per_tap && p.w_bytes > p.x_bytes);
  if (!want || total < 16 || total > (1 << 24)) return grid;
  p.xcd_gx = (int)grid.x; p.xcd_gy = (int)grid.y; p.xcd_total = (int)total;
  return dim3((unsigned)(((total + 7) / 8) * 8), 1, 1);
}
static void launch_v3p(const IgemmParams& p0, const PlanP& pp, int images, hipStream_t st) {
  IgemmParams p = p0;
  const dim3 grid = xcd_grid(p, dim3((unsigned)(images * pp.tiles_per_img), p.N / 128, pp.splits), false);
  // lean loop: buffer descriptors need the activations and the packed weights under 2 GiB each
  if ((g_tune_lean & 2) && p.x_bytes < (1ll << 31) && p.w_bytes < (1ll << 31)) launch_v3p_k<true>(p, pp, grid, st);
  else launch_v3p_k<false>(p, pp, grid, st);
  if (pp.splits > 1) {
    const long long work = (long long)p.M * (p.N / 4);
    DCT_LAUNCH_FAM(DCT_FAM_FOLDS, DCT_PROF_IGEMM, (splitk_epilogue_kernel<bf16_t>), dim3(div_up(work, 256)), dim3(256), 0, st, p, pp.splits);
  }
}

template <typename T>
static int launch(const IgemmParams& p0, const Plan& pl, hipStream_t st) {
  IgemmParams p = p0;
  dim3 grid(div_up(p.M, pl.bm), p.N / pl.bn, pl.splits);
  p.xcd_total = 0; p.xcd_gx = p.xcd_gy = 1;
  if (pl.v2) {
    grid = xcd_grid(p, grid, true);
    if (pl.bn == 128) {
      if (pl.bounds) launch_v2<128, 128, 4, 2, true>(p, grid, st); else launch_v2<128, 128, 4, 2, false>(p, grid, st);
    } else {
      if (pl.bounds) launch_v2<256, 64, 8, 1, true>(p, grid, st); else launch_v2<256, 64, 8, 1, false>(p, grid, st);
    }
  } else if (pl.bn == 128) {
    DCT_LAUNCH(DCT_PROF_IGEMM, (igemm_kernel<T, 128, 128>), grid, dim3(256), 0, st, p);
  } else {
    DCT_LAUNCH(DCT_PROF_IGEMM, (igemm_kernel<T, 64, 128>), grid, dim3(256), 0, st, p);
  }
  if (pl.splits > 1) {
    const long long work = (long long)p.M * (p.N / 4);
    DCT_LAUNCH_FAM(DCT_FAM_FOLDS, DCT_PROF_IGEMM, (splitk_epilogue_kernel<T>), dim3(div_up(work, 256)), dim3(256), 0, st, p, pl.splits);
  }
  return dct_check_launch();
}

}  // namespace

void dct_split_dw_db_launch(const float* partial, float* scratch, float* dw, float* db, int cout, int inner, int blocks, int accumulate, hipStream_t st);   // reduce.hip
size_t dct_split_dw_db_scratch(int cout, int inner);

void dct_relu_bits_launch(const void* y_bf16, unsigned char* bits, long long chunks, hipStream_t st) {
  DCT_LAUNCH(DCT_PROF_POINTWISE, relu_bits_kernel, dim3(div_up(chunks, 256)), dim3(256), 0, st, (const bf16_t*)y_bf16, bits, chunks);
}

// the view the convolution arithmetic runs on: x itself, or (d->unpool_codes) the un-pooled tensor x stands for
static dct_view conv_input_extent(const dct_view* x, const dct_conv_desc* d) {
  dct_view v = *x;
  if (d->unpool_codes) { v.h = d->unpool_h; v.w = d->unpool_w; }
  return v;
}

extern "C" size_t dct_conv2d_workspace_bytes(const dct_view* x0, const dct_view* y, const dct_conv_desc* d, int dtype) {
  if (!x0 || !y || !d) return 0;
  const dct_view xe = conv_input_extent(x0, d);
  const dct_view* x = &xe;
  const int Ho = d->scatter2x2 ? y->h / 2 : y->h, Wo = d->scatter2x2 ? y->w / 2 : y->w;
  const int M = y->n * Ho * Wo;
  const int N = d->scatter2x2 ? 4 * y->c : y->c;
  Plan pl;
  if (!make_plan(x, y, d, dtype, M, N, pl)) return 0;
  size_t need = pl.splits > 1 ? (size_t)pl.splits * M * N * sizeof(float) : 0;
  const PlanP pp = make_plan_p(x, y, d, dtype, N);
  if (pp.use && pp.splits > 1) need = std::max(need, (size_t)pp.splits * M * N * sizeof(float));
  if (d->stem_x) need = std::max(need, (size_t)y->n * ((Ho + 7) / 8) * ((Wo + 15) / 16) * 640 * sizeof(float) + dct_split_dw_db_scratch(64, 9));
  return need;
}

extern "C" int dct_conv2d(const dct_view* x0, const void* w_packed, const float* bias, const dct_view* mask,
                          const dct_view* y, const dct_conv_desc* d, int dtype,
                          void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (!view_ok(x0) || !view_ok(y) || !w_packed || !d) return DCT_ERR_BAD_ARG;
  if (d->unpool_codes) {
    // x0 is the gradient at the pooled tensor; the convolution runs on the un-pooled extent (strides stay x0's: the kernel indexes windows)
    if (d->unpool_h < 1 || d->unpool_w < 1 || x0->h != (d->unpool_h + 1) / 2 || x0->w != (d->unpool_w + 1) / 2) return DCT_ERR_BAD_ARG;
    if (dtype != DCT_BF16 || d->R != 3 || d->S != 3 || d->stride != 1 || d->dil != 1 || d->scatter2x2 || d->pad_h != 2 || d->pad_w != 2 ||
        x0->c % 64 || ((uintptr_t)d->unpool_codes & 7) || x0->sw != x0->c || x0->sh != (long long)x0->w * x0->c ||
        x0->sn != (long long)x0->h * x0->w * x0->c || (long long)x0->n * x0->sn >= (1ll << 31))
      return DCT_ERR_UNSUPPORTED;
  }
  const dct_view xe = conv_input_extent(x0, d);
  const dct_view* x = &xe;
  if (dtype != DCT_F32 && dtype != DCT_BF16) return DCT_ERR_BAD_ARG;
  if (d->R < 1 || d->S < 1 || d->stride < 1 || d->dil < 1) return DCT_ERR_BAD_ARG;
  if (x->n != y->n) return DCT_ERR_BAD_ARG;
  IgemmParams p;
  p.scatter = d->scatter2x2 ? 1 : 0;
  if (p.scatter && (y->h % 2 || y->w % 2 || d->R != 1 || d->S != 1)) return DCT_ERR_BAD_ARG;
  p.Ho = p.scatter ? y->h / 2 : y->h;
  p.Wo = p.scatter ? y->w / 2 : y->w;
  p.Hi = x->h; p.Wi = x->w;
  // the output extent must be reachable: (Ho-1)*stride + (R-1)*dil - 2*pad < Hi is NOT required
  // (out-of-range taps read zero), but shapes must match the conv arithmetic
  {
    const int eh = (x->h + 2 * d->pad_h - d->dil * (d->R - 1) - 1) / d->stride + 1;
    const int ew = (x->w + 2 * d->pad_w - d->dil * (d->S - 1) - 1) / d->stride + 1;
    if (eh != p.Ho || ew != p.Wo) return DCT_ERR_BAD_ARG;
  }
  p.M = y->n * p.Ho * p.Wo;
  p.cout = y->c;
  p.N = p.scatter ? 4 * y->c : y->c;
  p.Cin = x->c; p.R = d->R; p.S = d->S;
  p.stride = d->stride; p.dil = d->dil; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
  const int esz = dtype == DCT_BF16 ? 2 : 4;
  const int epv = 16 / esz;
  // 16-byte vector access requirements
  if (((uintptr_t)x->ptr & 15) || ((uintptr_t)w_packed & 15) || (x->sw % epv) || (x->sh % epv) || (x->sn % epv)) return DCT_ERR_UNSUPPORTED;
  if (((uintptr_t)y->ptr & (4 * esz - 1)) || (y->sw % 4) || (y->sh % 4) || (y->sn % 4) || (y->c % 4)) return DCT_ERR_UNSUPPORTED;
  Plan pl;
  if (!make_plan(x, y, d, dtype, p.M, p.N, pl)) return DCT_ERR_UNSUPPORTED;
  if ((long long)p.M * p.N > (1ll << 40)) return DCT_ERR_UNSUPPORTED;
  p.x = (const char*)x->ptr; p.w = (const char*)w_packed; p.bias = bias; p.y = (char*)y->ptr;
  p.xsN = x->sn; p.xsH = x->sh; p.xsW = x->sw;
  p.ysN = y->sn; p.ysH = y->sh; p.ysW = y->sw;
  p.mask = nullptr; p.msN = p.msH = p.msW = 0;
  p.mask_channels = 0; p.mask_scale = 1.f;
  if (mask) {
    if (!view_ok(mask) || mask->n != y->n || mask->h != y->h || mask->w != y->w) return DCT_ERR_BAD_ARG;
    if (((uintptr_t)mask->ptr & (4 * esz - 1)) || (mask->sw % 4) || (mask->sh % 4) || (mask->sn % 4)) return DCT_ERR_UNSUPPORTED;
    p.mask = (const char*)mask->ptr; p.msN = mask->sn; p.msH = mask->sh; p.msW = mask->sw;
    p.mask_channels = d->mask_channels > 0 ? d->mask_channels : y->c;
    if (p.mask_channels % 4) return DCT_ERR_UNSUPPORTED;
    p.mask_scale = d->mask_scale;
  }
  p.mask_bits = nullptr; p.bits_out = nullptr;
  const auto dense = [](const dct_view* v) { return v->sw == v->c && v->sh == (long long)v->w * v->c && v->sn == (long long)v->h * v->w * v->c; };
  if (d->mask_bits) {
    // a one-bit image of `mask` (which stays the source for the paths without a staged epilogue)
    if (!mask || dtype != DCT_BF16 || !dense(mask) || mask->c != y->c || p.mask_channels != y->c || y->c % 8) return DCT_ERR_BAD_ARG;
    p.mask_bits = d->mask_bits;
  }
  if (d->relu_bits_out) {
    if (dtype != DCT_BF16 || !dense(y) || y->c % 8 || p.scatter || ((uintptr_t)y->ptr & 15)) return DCT_ERR_BAD_ARG;
    p.bits_out = d->relu_bits_out;
  }
  p.stem_x = nullptr; p.stem_slab = nullptr;
  p.xcd_total = 0; p.xcd_gx = p.xcd_gy = 1;
  p.up_codes = d->unpool_codes; p.up_Hp = x0->h; p.up_Wp = x0->w;
  if (d->stem_x) {
    if (!d->stem_dw || !d->stem_db || ((uintptr_t)d->stem_x & 3)) return DCT_ERR_BAD_ARG;
    if (dtype != DCT_BF16 || y->c != 64 || x->c != 64 || d->R != 3 || d->S != 3 || d->stride != 1 || d->dil != 1 || p.scatter || d->accumulate ||
        d->relu || bias || !d->mask_bits || d->mask_scale != 1.f || d->pool_out || d->relu_bits_out || !g_tune_igemm_halo)
      return DCT_ERR_UNSUPPORTED;
  }
  p.pool_y = nullptr; p.pool_codes = nullptr; p.Hp = (y->h + 1) / 2; p.Wp = (y->w + 1) / 2; p.pool_only = 0;
  if (d->pool_only && (!d->pool_out || d->relu_bits_out)) return DCT_ERR_BAD_ARG;
  if (d->pool_codes && !d->pool_out) return DCT_ERR_BAD_ARG;
  if (d->pool_out) {
    if (p.scatter || d->accumulate || mask || d->mask_bits || ((uintptr_t)d->pool_out & 15) || ((uintptr_t)d->pool_codes & 7) || y->c % 8) return DCT_ERR_BAD_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  const auto pool_after = [&]() -> int {   // the launch just issued did not pool its tile: the pooling kernel behind it
    if (!d->pool_out) return DCT_OK;
    dct_view pv;
    pv.ptr = d->pool_out; pv.n = y->n; pv.h = p.Hp; pv.w = p.Wp; pv.c = y->c;
    pv.sw = y->c; pv.sh = (long long)p.Wp * y->c; pv.sn = (long long)p.Hp * p.Wp * y->c;
    return d->pool_codes ? dct_maxpool2x2_fwd_codes(y, &pv, d->pool_codes, dtype, stream) : dct_maxpool2x2_fwd(y, &pv, dtype, stream);
  };
  const auto bits_after = [&]() {      // the launch just issued did not write the bits in its epilogue
    if (!p.bits_out) return;
    dct_relu_bits_launch(y->ptr, p.bits_out, (long long)y->n * y->h * y->w * (y->c / 8), st);
  };
  p.relu = d->relu; p.accumulate = d->accumulate;
  p.kiters = pl.kiters; p.kiters_per_split = pl.kiters_per_split;
  p.cin_iters = x->c / pl.bk;
  p.partial = nullptr;
  p.staged = 0;
  p.x_bytes = ((long long)(x->n - 1) * x->sn + (long long)(x->h - 1) * x->sh + (long long)(x->w - 1) * x->sw + x->c) * esz;
  p.w_bytes = (long long)p.N * d->R * d->S * x->c * esz;
  if (pl.v2 && pl.splits == 1 && !d->accumulate) {
    const bool y16 = !((uintptr_t)y->ptr & 15) && y->sw % 8 == 0 && y->sh % 8 == 0 && y->sn % 8 == 0 && y->c % 8 == 0 &&
                     (long long)y->n * y->sn < (1ll << 31);
    const bool m16 = !mask || (!((uintptr_t)mask->ptr & 15) && mask->sw % 8 == 0 && mask->sh % 8 == 0 && mask->sn % 8 == 0 &&
                               p.mask_channels % 8 == 0 && (long long)mask->n * mask->sn < (1ll << 31));
    const bool b16 = !bias || !((uintptr_t)bias & 15);      // the staged epilogues read the bias as float4
    p.staged = (y16 && m16 && b16) ? 1 : 0;
  }
  if (pl.splits > 1) {
    const size_t need = (size_t)pl.splits * p.M * p.N * sizeof(float);
    if (!workspace || workspace_bytes < need) return DCT_ERR_WORKSPACE;
    p.partial = (float*)workspace;
  }
  if (pl.v2 && pl.splits == 1 && g_tune_igemm_halo && (!bias || !((uintptr_t)bias & 15)) && d->R == 3 && d->S == 3 && d->stride == 1 &&
      d->dil == 1 && !p.scatter) {
    // shared-halo kernel: 8 x 16 output patches; worth it when the patches cover the image well and fill the device
    const bool y16 = !((uintptr_t)y->ptr & 15) && y->sw % 8 == 0 && y->sh % 8 == 0 && y->sn % 8 == 0 && y->c % 8 == 0 &&
                     (long long)y->n * y->sn < (1ll << 31);
    const bool m16 = !mask || (!((uintptr_t)mask->ptr & 15) && mask->sw % 8 == 0 && mask->sh % 8 == 0 && mask->sn % 8 == 0 &&
                               p.mask_channels % 8 == 0 && (long long)mask->n * mask->sn < (1ll << 31));
    const int tiles_y = (p.Ho + 7) / 8, tiles_x = (p.Wo + 15) / 16;
    const int bn = p.N % 128 == 0 ? 128 : 64;
    const long long blocks = (long long)y->n * tiles_y * tiles_x * (p.N / bn);
    const double cover = (double)p.Ho * p.Wo / ((double)tiles_y * 8 * tiles_x * 16);
    // measured (tools/bench_conv.py --ab): the 64-channel tile only pays with a single channel slice (four blocks per CU)
    // 32-bit addressing inside the kernel: x within 2^31 elements, a weight tile's rows within 2^32 bytes of its first
    const bool x32 = (long long)x->n * x->sn < (1ll << 31) && (long long)bn * 9 * x->c * 2 < (1ll << 32);
    if (y16 && m16 && x32 && cover >= g_tune_igemm_halo_cover * 0.01 && blocks >= g_tune_igemm_halo_min_blocks && (bn == 128 || x->c == 64)) {
      if (d->pool_out && g_tune_igemm_pool) { p.pool_y = (char*)d->pool_out; p.pool_codes = d->pool_codes; p.pool_only = d->pool_only ? 1 : 0; }
      if (d->stem_x) {
        const size_t need = (size_t)blocks * 640 * sizeof(float) + dct_split_dw_db_scratch(64, 9);
        if (!workspace || workspace_bytes < need) return DCT_ERR_WORKSPACE;
        p.stem_x = d->stem_x; p.stem_slab = (float*)workspace;
        launch_v3<64, 1, 1>(p, tiles_x, tiles_y, y->n, st);
        dct_split_dw_db_launch((const float*)workspace, (float*)workspace + blocks * 640, d->stem_dw, d->stem_db, 64, 9, (int)blocks, d->stem_accumulate, st);
        DCT_PLAN_NOTE("igemm3m shared-halo 8x16 patches x 64 ch: %lld blocks, stem weight gradient from the tile, y not stored", blocks);
        return dct_check_launch();
      }
      if (bn == 128) {
        if (x->c == 64) launch_v3<128, 2, 1>(p, tiles_x, tiles_y, y->n, st); else launch_v3<128, 2, 2>(p, tiles_x, tiles_y, y->n, st);
      } else launch_v3<64, 1, 1>(p, tiles_x, tiles_y, y->n, st);
      DCT_PLAN_NOTE("igemm3m shared-halo 8x16 patches x %d ch: %lld blocks, cover %.0f %%, %d K-steps%s", bn, blocks, cover * 100, 9 * x->c / 64,
                    p.pool_y ? (p.pool_only ? ", pooled in the epilogue, y not stored" : ", pooled in the epilogue") : "");
      if (!p.pool_y) { const int rp = pool_after(); if (rp != DCT_OK) return rp; }
      return dct_check_launch();
    }
  }
  if (d->stem_x) return DCT_ERR_UNSUPPORTED;      // only the shared-halo 64-channel tile can take the stem along
  if (d->unpool_codes) return DCT_ERR_UNSUPPORTED;      // only the shared-halo kernels expand a pooled gradient while they stage (the caller un-pools first)
  if (pl.v2 && (!bias || !((uintptr_t)bias & 15))) {
    const PlanP pp = make_plan_p(x, y, d, dtype, p.N);
    if (pp.use) {
      const bool y16 = !((uintptr_t)y->ptr & 15) && y->sw % 8 == 0 && y->sh % 8 == 0 && y->sn % 8 == 0 && y->c % 8 == 0 &&
                       (long long)y->n * y->sn < (1ll << 31);
      const bool m16 = !mask || (!((uintptr_t)mask->ptr & 15) && mask->sw % 8 == 0 && mask->sh % 8 == 0 && mask->sn % 8 == 0 &&
                                 p.mask_channels % 8 == 0 && (long long)mask->n * mask->sn < (1ll << 31));
      const size_t need = pp.splits > 1 ? (size_t)pp.splits * p.M * p.N * sizeof(float) : 0;
      if (y16 && m16 && (!need || (workspace && workspace_bytes >= need))) {
        IgemmParams q = p;
        q.partial = pp.splits > 1 ? (float*)workspace : nullptr;
        launch_v3p(q, pp, y->n, st);
        { const int rp = pool_after(); if (rp != DCT_OK) return rp; }
        DCT_PLAN_NOTE("igemm3p packed rows (%d rows of %d px per 128-px tile): %d x %d blocks x %d channel-slice splits", pp.PR, p.Wo,
                      y->n * pp.tiles_per_img, p.N / 128, pp.splits);
        return dct_check_launch();
      }
    }
  }
  const int rc = dtype == DCT_BF16 ? launch<bf16_t>(p, pl, st) : launch<float>(p, pl, st);
  DCT_PLAN_NOTE("%s per-tap %d x %d tile%s: %lld tiles x %d splits, %d K-steps each%s", pl.v2 ? "igemm2" : "igemm", pl.bm, pl.bn,
                pl.bounds ? " (bounds)" : "", pl.tiles, pl.splits, pl.kiters_per_split, p.staged ? ", staged epilogue" : "");
  if (rc == DCT_OK && !p.staged && pl.splits == 1) bits_after();      // (a split layer's fold writes the bits itself)
  if (rc == DCT_OK) { const int rp = pool_after(); if (rp != DCT_OK) return rp; }
  return rc == DCT_OK ? dct_check_launch() : rc;
}


int dct_tune_set_wgrad(int knob, int value);  // wgrad.hip
extern int g_enet_wgrad_max_blocks;           // enet.hip
extern int g_enet_mfma;                       // enet.hip
extern int g_stem_dgrad_mfma;                 // pointwise.hip
extern int g_enet_mwgrad_waves;

extern "C" int dct_tune_set(int knob, int value) {
  switch (knob) {
    case DCT_TUNE_IGEMM_SPLIT: g_tune_igemm_split = value; return DCT_OK;
    case DCT_TUNE_IGEMM_HALO: g_tune_igemm_halo = value; return DCT_OK;
    case DCT_TUNE_IGEMM_PACKED: g_tune_igemm_packed = value; return DCT_OK;
    case DCT_TUNE_IGEMM_HALO_MIN_BLOCKS: g_tune_igemm_halo_min_blocks = value; return DCT_OK;
    case DCT_TUNE_IGEMM_HALO_COVER: g_tune_igemm_halo_cover = value; return DCT_OK;
    case DCT_TUNE_IGEMM_SPLIT_TARGET: if (value < 64) return DCT_ERR_BAD_ARG; g_tune_igemm_split_target = value; return DCT_OK;
    case DCT_TUNE_ENET_MWGRAD_WAVES: if (value < 64) return DCT_ERR_BAD_ARG; g_enet_mwgrad_waves = value; return DCT_OK;
    case DCT_TUNE_ENET_MFMA: if (value < 0 || value > 3) return DCT_ERR_BAD_ARG; g_enet_mfma = value; return DCT_OK;
    case DCT_TUNE_IGEMM_PACKED_SPLIT: if (value < 1) return DCT_ERR_BAD_ARG; g_tune_igemm_packed_split = value; return DCT_OK;
    case DCT_TUNE_IGEMM_PACKED_FILL: if (value < 1 || value > 100) return DCT_ERR_BAD_ARG; g_tune_igemm_packed_fill = value; return DCT_OK;
    case DCT_TUNE_IGEMM_XCD: if (value < 0 || value > 2) return DCT_ERR_BAD_ARG; g_tune_igemm_xcd = value; return DCT_OK;
    case 1100: g_dct_skip_families = value; return DCT_OK;      // diagnostic: skip kernel families (tools/ablate_step.py)
    case 1005: g_tune_igemm_pool = value ? 1 : 0; return DCT_OK;
    case 1008: g_stem_dgrad_mfma = value ? 1 : 0; return DCT_OK;
    case 1002: g_tune_igemm_split_max_tiles = value; return DCT_OK;     // planner studies (tools/bench_conv.py --ab-knob)
    case 1003: g_tune_igemm_split_min_kiters = value; return DCT_OK;
    case DCT_TUNE_ENET_WGRAD_BLOCKS: if (value < 1 || value > 1024) return DCT_ERR_BAD_ARG; g_enet_wgrad_max_blocks = value; return DCT_OK;
    default: return dct_tune_set_wgrad(knob, value);
  }
}
