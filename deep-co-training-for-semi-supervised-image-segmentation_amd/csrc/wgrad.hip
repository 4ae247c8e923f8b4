// Weight-gradient GEMM on MFMA for gfx950:
//
//   dW[p][tap][q] (+)= sum_m P[m][p] * Q[pixel(m)*stride + tap*dil - pad][q]
//
// The reduction runs over pixels, which is the SLOW axis of both NHWC operands, so both MFMA
// operands need a transposed fragment: bf16 tiles are staged [pixel][channel] exactly as they
// sit in HBM (coalesced 16-B loads, ds_write_b128) and read back with ds_read_b64_tr_b16, the
// gfx950 transposing LDS read (4 pixels x 16 channels per 16-lane group, column-major into
// the lanes).  The LDS pitch is row bytes + 64 so the four rows a 32-lane half touches land in
// different 64-B quarters of the 256-B bank row (conflict-free).  f32 (parity path) uses
// v_mfma_f32_32x32x2_f32, whose one-element-per-lane operands need no transpose.
//
// Work split: one block = (p tile, tap, q tile, pixel chunk).  Pixel chunks are the split-K;
// block ids are decoded so that all (tap, tile) blocks of one chunk carry the same id mod 8,
// i.e. share an XCD and re-read the same pixels from that XCD's L2 (speed only).  Partial
// tiles go to a workspace and are summed in fixed order by a second kernel: deterministic,
// and "+=" into an existing gradient is folded into that pass.
#include "dct_common.h"
#include <type_traits>

// bit mask of the instruction-lean loop forms (dct_tune_set(DCT_TUNE_LEAN, ...); all bit-identical to the forms they replace):
// bit 0 = filter-row weight gradient, bit 2 = per-tap weight gradient (this file), bit 1 = packed-rows conv kernel, bit 3 = per-tap conv kernel (igemm.hip),
// bit 4 (with bit 0) = the filter-row loop skips the sub-steps of a K-step that hold no dy pixel
int g_tune_lean = 31;

namespace {

struct WgradParams {
  const char* P; const char* Q; float* out;
  int M, Cp, Cq, R, S;
  int Hp, Wp, Hq, Wq;
  int stride, dil, pad_h, pad_w;
  long long psN, psH, psW, qsN, qsH, qsW;
  int chunks, pix_per_chunk, ptiles, qtiles;
};

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
typedef __attribute__((address_space(3))) char* lds_char_ptr;

// The transposing LDS reads of the LDS-DMA kernels are issued through inline asm: with the builtin,
// hipcc (ROCm 7.2) orders every ds_read_b64_tr_b16 behind ALL outstanding global_load_lds and puts an
// s_waitcnt vmcnt(0) in front of the first read of each K-step, which serialises the prefetch of the
// next stage with the compute of the current one.  The price is hand-placed lgkmcnt waits: a wait is
// followed by empty asm statements that "touch" the destination registers, so no consumer can be
// scheduled above it.
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(lds_char_ptr)(p); }
__device__ __forceinline__ void tr_issue(unsigned addr, bf16x4& dst) {
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr));
}
template <int OFF> __device__ __forceinline__ void tr_issue_o(unsigned addr, bf16x4& dst) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// one LDS-DMA piece through a buffer descriptor: 16 bytes per lane to lds + lane * 16, from base + voff + soff; a lane whose
// voff + soff falls outside the descriptor's range writes zeros.  (A __device__ helper: called straight from the kernel template,
// hipcc's host pass fails to instantiate the kernel's stub -- silently, the .so then lacks the symbol.)
__device__ __forceinline__ void buf_lds16(__amdgpu_buffer_rsrc_t r, char* lds, int voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, (int)soff, 0, 0);
}
template <int N> __device__ __forceinline__ void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void touch(bf16x4& r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ bf16x8 join(const bf16x4& lo, const bf16x4& hi) {
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// Block -> (pixel chunk, tile).  With >= 8 chunks all (tap, tile) blocks of one chunk carry the same
// id mod 8 (they share an XCD and re-read the chunk's pixels from that XCD's L2); with fewer
// chunks the tiles are simply dealt over all XCDs.
__device__ __forceinline__ bool decode_block(const WgradParams& p, int& chunk, int& tile) {
  const int tiles_per_chunk = p.ptiles * p.R * p.S * p.qtiles;
  const int bid = blockIdx.x;
  if (p.chunks >= 8) {
    const int xcd = bid & 7, slot = bid >> 3;
    chunk = (slot / tiles_per_chunk) * 8 + xcd;
    tile = slot % tiles_per_chunk;
  } else {
    chunk = bid / tiles_per_chunk;
    tile = bid - chunk * tiles_per_chunk;
  }
  return chunk < p.chunks;
}

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int pitch, int cbase, int kk, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = g >> 1;
  const char* a0 = tile + (kk * 16 + 8 * h + q) * pitch + (cbase + 16 * (g & 1) + 4 * pp) * 2;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(a0 + 4 * pitch));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

template <typename T, int BP, int BQ>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
  constexpr int ES = sizeof(T);
  constexpr int EPV = 16 / ES;
  constexpr int BKP = (ES == 2) ? 32 : 16;         // pixels per step
  constexpr int PITCH_P = BP * ES + 64, PITCH_Q = BQ * ES + 64;
  constexpr int CPR_P = BP / EPV, CPR_Q = BQ / EPV;  // 16-B chunks per row
  constexpr int RPP_P = 256 / CPR_P, RPP_Q = 256 / CPR_Q;  // rows per pass
  constexpr int NP_P = (BKP + RPP_P - 1) / RPP_P, NP_Q = (BKP + RPP_Q - 1) / RPP_Q;
  constexpr int TP = BP / 64, TQ = BQ / 64;
  constexpr int BUF = BKP * (PITCH_P + PITCH_Q);
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wp = wave >> 1, wq = wave & 1;

  int chunk, tile;
  if (!decode_block(p, chunk, tile)) return;
  const int qt = tile % p.qtiles; tile /= p.qtiles;
  const int tap = tile % (p.R * p.S);
  const int pt = tile / (p.R * p.S);
  const int tr = tap / p.S, ts = tap - tr * p.S;
  const int p0 = pt * BP, q0 = qt * BQ;

  const int mbeg = chunk * p.pix_per_chunk;
  const int mend = min(p.M, mbeg + p.pix_per_chunk);

  const int rowP = tid / CPR_P, chP = tid % CPR_P;
  const int rowQ = tid / CPR_Q, chQ = tid % CPR_Q;
  const int hwp = p.Hp * p.Wp;

  uint4 rp[NP_P], rq[NP_Q];
  auto load_tile = [&](int mit) {
#pragma unroll
    for (int pp = 0; pp < NP_P; ++pp) {
      const int row = rowP + pp * RPP_P;
      const int m = mit + row;
      if (row < BKP && m < mend) {
        const int n = m / hwp, rem = m - n * hwp;
        const int y = rem / p.Wp, x = rem - y * p.Wp;
        const T* src = reinterpret_cast<const T*>(p.P) + (n * p.psN + y * p.psH + x * p.psW + p0 + chP * EPV);
        rp[pp] = *reinterpret_cast<const uint4*>(src);
      } else {
        rp[pp] = make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int pp = 0; pp < NP_Q; ++pp) {
      const int row = rowQ + pp * RPP_Q;
      const int m = mit + row;
      bool ok = row < BKP && m < mend;
      int n = 0, iy = 0, ix = 0;
      if (ok) {
        n = m / hwp;
        const int rem = m - n * hwp;
        const int y = rem / p.Wp, x = rem - y * p.Wp;
        iy = y * p.stride + tr * p.dil - p.pad_h;
        ix = x * p.stride + ts * p.dil - p.pad_w;
        ok = (unsigned)iy < (unsigned)p.Hq && (unsigned)ix < (unsigned)p.Wq;
      }
      if (ok) {
        const T* src = reinterpret_cast<const T*>(p.Q) + (n * p.qsN + iy * p.qsH + ix * p.qsW + q0 + chQ * EPV);
        rq[pp] = *reinterpret_cast<const uint4*>(src);
      } else {
        rq[pp] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_tile = [&](char* buf) {
#pragma unroll
    for (int pp = 0; pp < NP_P; ++pp) {
      const int row = rowP + pp * RPP_P;
      if (row < BKP) *reinterpret_cast<uint4*>(buf + row * PITCH_P + chP * 16) = rp[pp];
    }
#pragma unroll
    for (int pp = 0; pp < NP_Q; ++pp) {
      const int row = rowQ + pp * RPP_Q;
      if (row < BKP) *reinterpret_cast<uint4*>(buf + BKP * PITCH_P + row * PITCH_Q + chQ * 16) = rq[pp];
    }
  };

  f32x16 acc[TP][TQ];
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (mbeg < mend) {
    load_tile(mbeg);
    store_tile(smem);
  }
  __syncthreads();
  int cur = 0;
  const int half = lane >> 5, l31 = lane & 31;
  for (int mit = mbeg; mit < mend; mit += BKP) {
    const bool more = mit + BKP < mend;
    if (more) load_tile(mit + BKP);
    const char* Pt = smem + cur * BUF;
    const char* Qt = Pt + BKP * PITCH_P;
    if constexpr (ES == 2) {
#pragma unroll
      for (int kk = 0; kk < BKP / 16; ++kk) {
        bf16x8 a[TP], b[TQ];
#pragma unroll
        for (int i = 0; i < TP; ++i) a[i] = tr_frag(Pt, PITCH_P, wp * (BP / 2) + i * 32, kk, lane);
#pragma unroll
        for (int j = 0; j < TQ; ++j) b[j] = tr_frag(Qt, PITCH_Q, wq * (BQ / 2) + j * 32, kk, lane);
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BKP / 2; ++kk) {
        float a[TP], b[TQ];
#pragma unroll
        for (int i = 0; i < TP; ++i)
          a[i] = *reinterpret_cast<const float*>(Pt + (kk * 2 + half) * PITCH_P + (wp * (BP / 2) + i * 32 + l31) * 4);
#pragma unroll
        for (int j = 0; j < TQ; ++j)
          b[j] = *reinterpret_cast<const float*>(Qt + (kk * 2 + half) * PITCH_Q + (wq * (BQ / 2) + j * 32 + l31) * 4);
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) store_tile(smem + (cur ^ 1) * BUF);
    __syncthreads();
    cur ^= 1;
  }

  // store: out[chunk][p][tap][q]
  const int taps = p.R * p.S;
  float* base = p.out + (long long)chunk * p.Cp * taps * p.Cq;
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      const int qc = q0 + wq * (BQ / 2) + j * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int pr = p0 + wp * (BP / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        base[((long long)pr * taps + tap) * p.Cq + qc] = acc[i][j][e];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 main kernel ("v2"): 64 pixels per K-step, both tiles staged straight from HBM into LDS with
// global_load_lds_dwordx4 (two stages, one barrier per step).  The LDS image is lane-linear
// ([pixel][channel], one wave-instruction = 1 KiB of consecutive rows), so the conflict-avoiding
// layout for the transposing reads is an XOR on the 64-B column group applied to the per-lane
// SOURCE chunk and again on the read: the four pixel rows a 32-lane half reads then sit in four
// different 64-B quarters of the 256-B bank row.  LDS row rho holds pixel 4*(rho % 16) + rho / 16 of the
// step (the reduction is order-free), which gives every thread four CONSECUTIVE pixels: one
// div-free decode (float reciprocal + fix-up) and three increments per step instead of eight
// integer divisions.
__device__ __attribute__((aligned(128))) const uint4 g_wzero_page[8] = {};
typedef __attribute__((address_space(1))) const void* wg_gptr_t;
typedef __attribute__((address_space(3))) void* wg_lptr_t;

struct FastDiv { int d; float rcp; };
__device__ __forceinline__ int fdiv(int m, const FastDiv& f, int& rem) {   // 0 <= m < 2^24
  int q = (int)((float)m * f.rcp);
  int r = m - q * f.d;
  if (r < 0) { --q; r += f.d; }
  else if (r >= f.d) { ++q; r -= f.d; }
  rem = r;
  return q;
}

struct Wgrad2Params {
  WgradParams w;
  FastDiv dhw, dw_;
  int direct;      // chunks == 1: add straight into out (no partial slab)
  int accumulate;
  float* bias;     // bias gradient (column sums of P): direct -> db itself, else unused (slab tail holds it)
  int with_bias;
  long long slab_stride;   // floats per slab: Cp*taps*Cq (+ Cp with the bias tail)
  long long p_bytes, q_bytes;   // LEAN: bytes from P / Q to the end of the views (buffer descriptor ranges, < 2^31)
};

// Tile rows are RB bytes (128 or 256); the NW waves of a block each issue NI wave-instructions (RPI rows apiece) per
// stage.  A thread's LDS rows are rho_i = RS * i + c (RS = 64 / NI, c = wave * RPI + lane / CPR), and LDS row rho
// holds pixel k = NI * (rho % RS) + rho / RS of the step, so the thread's NI pixels are consecutive.  The 16-B chunk
// index of pixel k's row is XORed with swz_k(k): the four pixels k0 .. k0+3 a 32-lane half reads with one transposing
// read then sit in four different 64-B quarters of the 256-B bank row.
template <int RB, int NW> struct TileGeo {
  static constexpr int CPR = RB / 16;          // chunks per row: 8 | 16
  static constexpr int RPI = 64 / CPR;         // rows per wave-instruction: 8 | 4
  static constexpr int NI = 64 / (NW * RPI);   // instructions per wave per stage
  static constexpr int RS = 64 / NI;           // row stride between a thread's instructions
  static_assert(NI >= 1 && 4 % NI == 0, "wave count / tile width mismatch");
  __device__ static __forceinline__ int row_of(int k) { return RS * (k % NI) + k / NI; }
  // RB = 256: a row spans the bank row, quarter ^= k % 4.  RB = 128: two rows per bank row; the row parity already
  // separates two of the four pixels, the 64-B half is XORed with the bit of k % 4 that the parity does not cover.
  __device__ static __forceinline__ int swz_k(int k) {
    if (RB == 256) return (k & 3) << 2;
    return (NI == 1 ? ((k >> 1) & 1) : (k & 1)) << 2;
  }
};

// LEAN (round 4; views under 2 GiB): the loop carried 107 vector instructions per 8 MFMAs (tools/isa_loop_mix.py) and was bound by
// their issue.  A piece is one buffer_load_dwordx4 ... lds whose lane offset is the decoded pixel's byte offset + the lane's chunk
// (a pixel that contributes zero decodes to an offset the descriptor rejects: that lane stages zeros), so the per-piece 64-bit
// pointer arithmetic and selects go; the fragment reads carry their sub-step / row-group offsets as immediates.  Bit-identical.
template <int BP, int BQ, int NW, bool LEAN>
__global__ __launch_bounds__(NW * 64) void wgrad2_kernel(Wgrad2Params pr) {
  const WgradParams& p = pr.w;
  constexpr int BKP = 64;
  constexpr int RBP = BP * 2, RBQ = BQ * 2;                 // row bytes
  using GP = TileGeo<RBP, NW>;
  using GQ = TileGeo<RBQ, NW>;
  constexpr int WPR = BP / (NW / 2);                        // P rows per wave (waves: NW/2 along P x 2 along Q)
  constexpr int TP = WPR / 32, TQ = BQ / 64;
  constexpr int PPW = 64 / NW;                              // pixels of a step staged by one wave
  static_assert(TP >= 1, "too many waves for this tile");
  constexpr int STAGE = BKP * (RBP + RBQ);
  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = LEAN ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;     // LEAN: piece bookkeeping and the bias branch stay scalar
  const int wp = wave >> 1, wq = wave & 1;

  int chunk, tile;
  if (!decode_block(p, chunk, tile)) return;
  const int qt = tile % p.qtiles; tile /= p.qtiles;
  const int tap = tile % (p.R * p.S);
  const int pt = tile / (p.R * p.S);
  const int tr = tap / p.S, ts = tap - tr * p.S;
  const int p0 = pt * BP, q0 = qt * BQ;
  const int mbeg = chunk * p.pix_per_chunk;
  const int mend = min(p.M, mbeg + p.pix_per_chunk);

  const char* Pb = p.P + (long long)p0 * 2;
  const char* Qb = p.Q + (long long)q0 * 2;
  const char* zero = reinterpret_cast<const char*>(g_wzero_page) + (lane & 7) * 16;
  const int tyo = tr * p.dil - p.pad_h, txo = ts * p.dil - p.pad_w;
  // Staging addresses.  Wave w stages the PPW = 64 / NW pixels [PPW*w, PPW*(w+1)) of a step (rows rho = RS*i + c
  // hold pixel NI*c + i).  Per step, lane l decodes ONE pixel (PPW*w + l % PPW) into 32-bit element offsets of
  // its dy / x rows (-1: contributes zero), and every lane then fetches the offsets of the pixels it
  // stages with a lane permute -- instead of every lane decoding all of its rows.
  const int cP = lane / GP::CPR, cQ = lane / GQ::CPR;          // row within a wave-instruction
  int srcP[GP::NI], srcQ[GQ::NI];                                // permute source lanes (x4 for ds_bpermute)
  int chP[GP::NI], chQ[GQ::NI];                                  // byte offset of this lane's (swizzled) chunk
#pragma unroll
  for (int i = 0; i < GP::NI; ++i) {
    srcP[i] = (GP::NI * cP + i) * 4;
    chP[i] = ((lane % GP::CPR) ^ GP::swz_k(GP::NI * (wave * GP::RPI + cP) + i)) * 16;
  }
#pragma unroll
  for (int i = 0; i < GQ::NI; ++i) {
    srcQ[i] = (GQ::NI * cQ + i) * 4;
    chQ[i] = ((lane % GQ::CPR) ^ GQ::swz_k(GQ::NI * (wave * GQ::RPI + cQ) + i)) * 16;
  }
  const int psN = (int)p.psN, psH = (int)p.psH, psW = (int)p.psW;
  const int qsN = (int)p.qsN, qsH = (int)p.qsH, qsW = (int)p.qsW;

  __amdgpu_buffer_rsrc_t rsP, rsQ;
  if constexpr (LEAN) {
    rsP = __builtin_amdgcn_make_buffer_rsrc((void*)Pb, 0, (int)(pr.p_bytes - (long long)p0 * 2), 0x00020000);
    rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qb, 0, (int)(pr.q_bytes - (long long)q0 * 2), 0x00020000);
  }
  // LEAN: ONE wave per K-step decodes all 64 pixels of the step two ahead (a lane per pixel) into an LDS table of byte offsets
  // (0x80000000 = contributes zero: rejected by the descriptors); the waves take turns, and every wave picks the offsets of the
  // pixels it stages from the table one step later.  The 50 vector instructions of the decode then run once per block and step
  // instead of once per wave.  table[slot][pixel] = {dy offset, x offset}; slot = step & 1.
  int* const tab = reinterpret_cast<int*>(smem + 2 * STAGE);
  auto decode = [&](int slot, int mit) {
    const int m = mit + lane;
    int bP = (int)0x80000000u, bQ = (int)0x80000000u;
    int rem, x;
    const int n = fdiv(min(m, p.M - 1), pr.dhw, rem);
    const int y = fdiv(rem, pr.dw_, x);
    const int iy = y * p.stride + tyo, ix = x * p.stride + txo;
    if (m < mend) {
      bP = (n * psN + y * psH + x * psW) * 2;
      if ((unsigned)iy < (unsigned)p.Hq && (unsigned)ix < (unsigned)p.Wq) bQ = (n * qsN + iy * qsH + ix * qsW) * 2;
    }
    int2 v; v.x = bP; v.y = bQ;
    *reinterpret_cast<int2*>(tab + (slot * 64 + lane) * 2) = v;
  };
  auto stage_l = [&](char* buf, int slot) {
    const int* t = tab + slot * 128 + PPW * wave * 2;
    int vP[GP::NI], vQ[GQ::NI];
#pragma unroll
    for (int i = 0; i < GP::NI; ++i) vP[i] = t[(GP::NI * cP + i) * 2] + chP[i];
#pragma unroll
    for (int i = 0; i < GQ::NI; ++i) vQ[i] = t[(GQ::NI * cQ + i) * 2 + 1] + chQ[i];
#pragma unroll
    for (int i = 0; i < GP::NI; ++i) buf_lds16(rsP, buf + (GP::RS * i + wave * GP::RPI) * RBP, vP[i], 0u);
#pragma unroll
    for (int i = 0; i < GQ::NI; ++i) buf_lds16(rsQ, buf + BKP * RBP + (GQ::RS * i + wave * GQ::RPI) * RBQ, vQ[i], 0u);
  };
  auto stage = [&](char* buf, int mit) {
    const int m = mit + PPW * wave + (lane % PPW);
    int offP = -1, offQ = -1;
    {
      int rem, x;
      const int n = fdiv(min(m, p.M - 1), pr.dhw, rem);
      const int y = fdiv(rem, pr.dw_, x);
      const int iy = y * p.stride + tyo, ix = x * p.stride + txo;
      if (m < mend) {
        offP = n * psN + y * psH + x * psW;
        if ((unsigned)iy < (unsigned)p.Hq && (unsigned)ix < (unsigned)p.Wq) offQ = n * qsN + iy * qsH + ix * qsW;
      }
    }
#pragma unroll
    for (int i = 0; i < GP::NI; ++i) {
      const int off = __builtin_amdgcn_ds_bpermute(srcP[i], offP);
      const char* src = off >= 0 ? Pb + ((long long)off * 2 + chP[i]) : zero;
      __builtin_amdgcn_global_load_lds((wg_gptr_t)src, (wg_lptr_t)(buf + (GP::RS * i + wave * GP::RPI) * RBP), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < GQ::NI; ++i) {
      const int off = __builtin_amdgcn_ds_bpermute(srcQ[i], offQ);
      const char* src = off >= 0 ? Qb + ((long long)off * 2 + chQ[i]) : zero;
      __builtin_amdgcn_global_load_lds((wg_gptr_t)src, (wg_lptr_t)(buf + BKP * RBP + (GQ::RS * i + wave * GQ::RPI) * RBQ), 16, 0, 0);
    }
  };

  f32x16 acc[TP][TQ];
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if constexpr (LEAN) {
    if (wave == 0) { decode(0, mbeg); decode(1, mbeg + BKP); }
    __syncthreads();
    if (mbeg < mend) stage_l(smem, 0);
  } else {
    if (mbeg < mend) stage(smem, mbeg);
  }
  __syncthreads();
  int cur = 0;
  // per-lane fragment bases: everything lane-dependent hoisted; kk and the +4 partner are constants
  int pbase[TP], qbase[TQ];
  {
    const int g = lane >> 4, li = lane & 15, lq = li >> 2, lpp = li & 3, lh = g >> 1;
    const int kq0 = 8 * lh + lq;
    const int rp = GP::row_of(kq0), rq = GQ::row_of(kq0);
#pragma unroll
    for (int i = 0; i < TP; ++i) {
      const int colb = (wp * WPR + i * 32 + 16 * (g & 1) + 4 * lpp) * 2;
      pbase[i] = rp * RBP + (colb ^ (GP::swz_k(kq0) << 4));
    }
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      const int colb = (wq * (BQ / 2) + j * 32 + 16 * (g & 1) + 4 * lpp) * 2;
      qbase[j] = BKP * RBP + rq * RBQ + (colb ^ (GQ::swz_k(kq0) << 4));
    }
  }
  constexpr int P_KK = (16 / GP::NI) * RBP, P_HI = (4 / GP::NI) * RBP;
  constexpr int Q_KK = (16 / GQ::NI) * RBQ, Q_HI = (4 / GQ::NI) * RBQ;
  constexpr int NRD = 2 * (TP + TQ);             // tr reads per 16-pixel sub-step
  const unsigned smem_off = lds_off(smem);
  // bias gradient rides along in the blocks of tap 0 / first cin tile (their wq == 0 waves): see epilogue
  const bool do_bias = pr.with_bias && tap == 0 && qt == 0 && wq == 0;
  f32x16 accb[TP];
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  constexpr std::integral_constant<int, 0> set0{};
  constexpr std::integral_constant<int, 1> set1{};
  static_assert(BKP / 16 == 4, "four sub-steps per K-step");
  int turn = 0;                                  // LEAN: the wave that decodes in this step
  for (int mit = mbeg; mit < mend; mit += BKP) {
    if constexpr (LEAN) {
      // the table slot of step k (= cur) was read at the top of step k - 1: free since that step's barrier
      if (wave == turn && mit + 2 * BKP < mend) decode(cur, mit + 2 * BKP);
      turn = turn + 1 == NW ? 0 : turn + 1;
      if (mit + BKP < mend) stage_l(smem + (cur ^ 1) * STAGE, cur ^ 1);
    } else {
      if (mit + BKP < mend) stage(smem + (cur ^ 1) * STAGE, mit + BKP);
    }
    const unsigned Pl = smem_off + cur * STAGE;
    bf16x4 fa[2][TP][2], fb[2][TQ][2];           // [set][tile][lo/hi], sets alternate per sub-step
    auto compute = [&](auto setc) {
      constexpr int set = decltype(setc)::value;
#pragma unroll
      for (int i = 0; i < TP; ++i) { touch(fa[set][i][0]); touch(fa[set][i][1]); }
#pragma unroll
      for (int j = 0; j < TQ; ++j) { touch(fb[set][j][0]); touch(fb[set][j][1]); }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 a[TP], b[TQ];
#pragma unroll
      for (int i = 0; i < TP; ++i) a[i] = join(fa[set][i][0], fa[set][i][1]);
#pragma unroll
      for (int j = 0; j < TQ; ++j) b[j] = join(fb[set][j][0], fb[set][j][1]);
#pragma unroll
      for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if (do_bias) {        // wave-uniform: column sums of the dy tile = dy^T x ones
#pragma unroll
        for (int i = 0; i < TP; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], ones, accb[i], 0, 0, 0);
      }
    };
    if constexpr (LEAN) {
      unsigned pa[TP], qa[TQ];
#pragma unroll
      for (int i = 0; i < TP; ++i) pa[i] = Pl + pbase[i];
#pragma unroll
      for (int j = 0; j < TQ; ++j) qa[j] = Pl + qbase[j];
#define DCT_W2_ISSUE(set, KK)                                                                  \
      {                                                                                        \
        _Pragma("unroll") for (int i = 0; i < TP; ++i) {                                       \
          tr_issue_o<(KK) * P_KK>(pa[i], fa[set][i][0]);                                       \
          tr_issue_o<(KK) * P_KK + P_HI>(pa[i], fa[set][i][1]);                                \
        }                                                                                      \
        _Pragma("unroll") for (int j = 0; j < TQ; ++j) {                                       \
          tr_issue_o<(KK) * Q_KK>(qa[j], fb[set][j][0]);                                       \
          tr_issue_o<(KK) * Q_KK + Q_HI>(qa[j], fb[set][j][1]);                                \
        }                                                                                      \
      }
      DCT_W2_ISSUE(0, 0)
      DCT_W2_ISSUE(1, 1) lgkm_wait<NRD>(); compute(set0);
      DCT_W2_ISSUE(0, 2) lgkm_wait<NRD>(); compute(set1);
      DCT_W2_ISSUE(1, 3) lgkm_wait<NRD>(); compute(set0);
      lgkm_wait<0>(); compute(set1);
#undef DCT_W2_ISSUE
    } else {
    auto issue = [&](int set, int kk) {
#pragma unroll
      for (int i = 0; i < TP; ++i) {
        tr_issue(Pl + pbase[i] + kk * P_KK, fa[set][i][0]);
        tr_issue(Pl + pbase[i] + kk * P_KK + P_HI, fa[set][i][1]);
      }
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        tr_issue(Pl + qbase[j] + kk * Q_KK, fb[set][j][0]);
        tr_issue(Pl + qbase[j] + kk * Q_KK + Q_HI, fb[set][j][1]);
      }
    };
    issue(0, 0);
    issue(1, 1); lgkm_wait<NRD>(); compute(set0);
    issue(0, 2); lgkm_wait<NRD>(); compute(set1);
    issue(1, 3); lgkm_wait<NRD>(); compute(set0);
    lgkm_wait<0>(); compute(set1);
    }
    __syncthreads();
    cur ^= 1;
  }

  const int half = lane >> 5, l31 = lane & 31;
  const int taps = p.R * p.S;
  float* base = pr.direct ? p.out : p.out + (long long)chunk * pr.slab_stride;
  if (do_bias && l31 == 0) {
    float* bb = pr.direct ? pr.bias : base + (long long)p.Cp * taps * p.Cq;
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int prow = p0 + wp * WPR + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (pr.direct && pr.accumulate) bb[prow] += accb[i][e]; else bb[prow] = accb[i][e];
      }
  }
#pragma unroll
  for (int i = 0; i < TP; ++i)
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
      const int qc = q0 + wq * (BQ / 2) + j * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int prow = p0 + wp * WPR + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        float* dst = base + ((long long)prow * taps + tap) * p.Cq + qc;
        if (pr.direct && pr.accumulate) *dst += acc[i][j][e]; else *dst = acc[i][j][e];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 3x3 stride-1 kernel that shares pixels between taps ("v3").  Per 64-pixel step the v2 tile moves
// 64 x (BP + BQ) x 2 bytes from L2 into LDS for ONE tap: 64 FLOP/B at 128 x 128 and 32 FLOP/B at 64 x 64, at or
// below the CU's ratio of MFMA rate to L1 fill rate (64 FLOP/B) -- the 64-channel layers ran at 300 TFLOP/s.
// Here a block owns one filter ROW r and computes its three taps s = 0..2 together: a K-step is a run of up to
// 64 dy pixels of one image row; the x strip of that run is staged once with two extra pixels (66 rows) and tap s
// reads it shifted by s rows.  (8 + 8.25) KiB per 3 x 0.52 MFLOP = 95 FLOP/B at 64 x 64, 185 FLOP/B at 128 x 128.
// LDS rows are pixels in natural order; the 64-B column group of pixel row k is XORed with a function of k that
// is invariant under k += 4 and takes four different values on any four consecutive rows, so the transposing reads
// stay conflict-free for every shift.  Staging needs no per-pixel decode: (image, row, run) advance incrementally
// and a lane's source address is the run's base plus a constant.
struct Wgrad3Params {
  WgradParams w;
  int segs_per_row, nseg, seg_per_chunk;
  int pitch, nr, units_per_image;     // pitch > 0: narrow images -- a K-step packs nr image rows at a pitch of Wp + 2 rows
  int direct, accumulate, with_bias;
  int skip_empty;                     // LEAN: multiply only the 16-row sub-steps of a K-step that hold dy pixels
  const unsigned char* up_codes;      // UNPOOL: P is the DENSE gradient at the pooled tensor [n][up_Hp][up_Wp][Cp], up_codes its routing codes; w.Hp / w.Wp
  int up_Hp, up_Wp;                   // stay the extent of the un-pooled tensor the reduction runs over
  float* bias;
  long long slab_stride;
  long long p_bytes, q_bytes;         // LEAN: bytes from P / Q to the end of the views (buffer descriptor ranges, < 2^31)
  unsigned long long* stamps;         // diagnostic build (-DDCT_W3_STAMPS, tools/gpu/w3_stamps.py): per-wave cycle sums of the K-step's segments
};
#ifdef DCT_W3_STAMPS
// s_memtime with its own lgkmcnt(0) (cdna_hip_programming.md, In-kernel stamps): placed only where no counted LDS wait is pending
__device__ __forceinline__ unsigned long long w3_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#endif

template <int RB> __device__ __forceinline__ int swz3(int k) {       // in 16-B chunks
  return RB == 256 ? ((k & 3) << 2) : (((k >> 1) & 1) << 2);
}

// G: wave groups per block.  Each group of NW waves is a complete copy of the tile machinery (own stages, own half of the
// block's K-steps); the groups' accumulators are added through LDS before the slab is written.  Two groups halve the
// number of slabs (written once, read once by the fold: 38 MB per layer at 768 four-wave blocks) at the same waves per CU.
// QSHIFT: the x fragments of a row's three taps come from ONE 12-pixel window per lane (three transposing reads) -- tap 2 is the
// window moved by one dword, tap 1 four v_alignbit_b32 -- instead of three separate 8-pixel reads: 5 instead of 8 fragment reads per
// sub-step (0.83 KiB of LDS per MFMA instead of 1.33; the kernel is LDS-bandwidth bound).  Same MFMA operands bit for bit.
// UNPOOL (round 5; LEAN, one image row per K-step): the dy operand is an encoder block's un-pooled gradient, expanded while it is
// staged from the gradient at the pooled tensor and the pooling's routing codes (dct_common.h dct_unpool_chunk8).  A K-step's 64
// pixels of image row y are 32 windows of pooled row y / 2; a thread of the group owns one window x 8 channels: 16 + 8 bytes through
// registers (issued with the step's LDS-DMA pieces, one K-step ahead), two 16-byte LDS writes in front of the step's barrier -- the
// positions (y % 2, 0) and (y % 2, 1) of the window; windows past the row's end write zeros.  The x strip is staged as before.
template <int BP, int BQ, int NW, bool NARROW, int G, bool QSHIFT, bool LEAN, bool UNPOOL = false>
__global__ __launch_bounds__(G * NW * 64) void wgrad3_kernel(Wgrad3Params pr) {
  static_assert(!UNPOOL || (LEAN && BP == 64 && NW == 4), "UNPOOL: lean loop, 64-channel dy tile, 256 threads per group");
  const WgradParams& p = pr.w;
  constexpr int RBP = BP * 2, RBQ = BQ * 2;
  constexpr int CPRP = RBP / 16, CPRQ = RBQ / 16, RPIP = 64 / CPRP, RPIQ = 64 / CPRQ;
  constexpr int QROWS = 72;                                         // 64 + 2 halo pixels, rounded up to whole pieces
  constexpr int NPCP = 64 / RPIP, NPCQ = QROWS / RPIQ, NPC = NPCP + NPCQ;
  constexpr int NPW = (NPC + NW - 1) / NW;                          // pieces per wave per stage
  constexpr int WPR = BP / (NW / 2), TP = WPR / 32, TQ = BQ / 64;
  constexpr int STAGE = 64 * RBP + QROWS * RBQ;
  static_assert(TP >= 1 && TQ >= 1, "tile/wave mismatch");
  static_assert(!LEAN || NPCP % NW == 0, "LEAN: piece i of every wave must be of one kind");
  extern __shared__ __attribute__((aligned(128))) char smem_all[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: piece bookkeeping stays in SGPRs
  const int grp = wave_all / NW, wave = wave_all % NW;
  char* smem = smem_all + grp * (2 * STAGE);
  const int wp = wave >> 1, wq = wave & 1;

  const int tiles_per_chunk = p.ptiles * 3 * p.qtiles;
  int chunk, tile;
  {
    const int bid = blockIdx.x;
    if (p.chunks >= 8) { const int xcd = bid & 7, slot = bid >> 3; chunk = (slot / tiles_per_chunk) * 8 + xcd; tile = slot % tiles_per_chunk; }
    else { chunk = bid / tiles_per_chunk; tile = bid - chunk * tiles_per_chunk; }
    if (chunk >= p.chunks) return;
  }
  const int qt = tile % p.qtiles; tile /= p.qtiles;
  const int tr = tile % 3;
  const int pt = tile / 3;
  const int p0 = pt * BP, q0 = qt * BQ;
  const int cbeg = chunk * pr.seg_per_chunk, cend = min(pr.nseg, cbeg + pr.seg_per_chunk);
  const int per_group = (cend - cbeg + G - 1) / G;                   // K-steps of every group (the last may have fewer)
  const int gbeg = min(cend, cbeg + grp * per_group), gend = min(cend, gbeg + per_group);

  const char* Pb = p.P + (long long)p0 * 2;
  const char* Qb = p.Q + (long long)q0 * 2;
  const char* zero = reinterpret_cast<const char*>(g_wzero_page) + (lane & 7) * 16;
  const int psW = (int)p.psW, qsW = (int)p.qsW;

  // per-lane staging constants: piece (wave + i * NW) is a dy piece (8 or 4 pixel rows) or an x piece.  LDS row k of a
  // tile is pixel (rho, col) of the step: (0, k) for wide images (one run of a row per step), (k / pitch, k % pitch)
  // for narrow ones (nr image rows per step, each followed by two gap rows so that tap s is still "row k + s").
  //
  // LEAN (pad 0, views under 2 GiB; round 4): the loop was bound by instruction issue, not by the matrix pipe -- 144 vector +
  // 111 scalar instructions per 12 MFMAs (tools/isa_loop_mix.py), two thirds of them the staging's per-piece pointer selects
  // and 64-bit step bases.  Here a piece is ONE buffer_load_dwordx4 ... lds: the tensor is a buffer descriptor, the lane's
  // offset inside a step is a constant (a lane that must stage zeros -- gap rows, rows past the tile -- holds an offset the
  // descriptor's range check rejects: such a lane writes zeros to LDS, tools/probe_buffer_lds), the step's base is a 32-bit
  // scalar offset advanced by adds, and only a step that is not full (row tail, last rows of an image) compares the lane's
  // row with the step's limit.  Fragment reads take their sub-step / row-group offsets as immediates.
  int rho[NPW], col[NPW], loff[NPW], ldst[NPW];
  bool isP[NPW], live[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int piece = wave + i * NW;
    live[i] = piece < NPC;
    isP[i] = piece < NPCP;
    const int k = isP[i] ? piece * RPIP + lane / CPRP : (piece - NPCP) * RPIQ + lane / CPRQ;
    rho[i] = NARROW ? k / pr.pitch : 0;
    col[i] = NARROW ? k - rho[i] * pr.pitch : k;
    if (isP[i]) {
      loff[i] = (rho[i] * (int)p.psH + col[i] * psW) * 2 + (((lane % CPRP) ^ swz3<RBP>(k)) * 16);
      ldst[i] = piece * 1024;
    } else {
      loff[i] = (rho[i] * (int)p.qsH + col[i] * qsW) * 2 + (((lane % CPRQ) ^ swz3<RBQ>(k)) * 16);
      ldst[i] = 64 * RBP + (piece - NPCP) * 1024;
    }
    if constexpr (LEAN) {
      // static part of "this lane stages a pixel": dy gap columns / rows past the last packed row, x rows past the strip
      const bool ok = NARROW ? (rho[i] < pr.nr && (!isP[i] || col[i] < p.Wp)) : (isP[i] || k < 66);
      if (!ok) loff[i] = (int)0x80000000u;
    }
  }
  // (image, first dy row, first dy column) of the next step to stage: one decode here, then increments
  int s_n, s_y, s_x;
  {
    s_n = gbeg / pr.units_per_image;
    const int u = gbeg - s_n * pr.units_per_image;
    if (NARROW) { s_y = u * pr.nr; s_x = 0; }
    else { s_y = u / pr.segs_per_row; s_x = (u - s_y * pr.segs_per_row) * 64; }
  }
  // LEAN: byte offsets of the step's first dy / x pixel from the descriptors' bases, as 32-bit scalars
  unsigned offP = 0, offQ = 0;
  __amdgpu_buffer_rsrc_t rsP, rsQ;
  if constexpr (LEAN) {
    offP = (unsigned)((s_n * (int)p.psN + s_y * (int)p.psH + s_x * psW) * 2);
    offQ = (unsigned)((s_n * (int)p.qsN + (s_y + tr) * (int)p.qsH + s_x * qsW) * 2);
    rsP = __builtin_amdgcn_make_buffer_rsrc((void*)Pb, 0, (int)(pr.p_bytes - (long long)p0 * 2), 0x00020000);
    rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qb, 0, (int)(pr.q_bytes - (long long)q0 * 2), 0x00020000);
  }
  const int lrowP = lane / CPRP, lrowQ = lane / CPRQ;
  // LEAN: 16-row sub-steps of the step staged last whose dy rows hold pixels (1..4).  The rows behind them are staged as zeros,
  // so their MFMAs (and bias sums) add exact zeros: the loop skips them -- a row tail of 20 pixels (84-pixel rows: 64 + 20) or
  // a packed step of 48 rows (one 46-pixel row + gap) costs two / three sub-steps instead of four.  Bit-identical.
  int sub_staged = 4;
  // UNPOOL: the thread's window of a step (j = thread of the group / 8) and its 8-channel chunk
  const int utg = wave * 64 + lane, uj = utg >> 3, uc8 = utg & 7;
  const int uoff = uj * p.Cp + p0 + uc8 * 8;                                      // elements (= code bytes) from the step's first window
  const int udst = (2 * uj) * RBP + ((uc8 ^ swz3<RBP>(2 * uj)) * 16);            // LDS bytes of the window's first pixel row in a dy tile (the second: + RBP)
  uint4 ug = make_uint4(0u, 0u, 0u, 0u);
  uint2 ucd = make_uint2(0u, 0u);
  unsigned upos = 0;                 // window position of the staged step's first pixel column: 2 * (y % 2)
  auto stage = [&](char* buf) {
    if constexpr (LEAN) {
      if constexpr (UNPOOL) {
        // the step's windows: pooled row s_y / 2, from column s_x / 2 on; pixels 2 j, 2 j + 1 of the step lie in window j
        const int lim_px = NARROW ? p.Wp : min(64, p.Wp - s_x);
        const int base = ((s_n * pr.up_Hp + (s_y >> 1)) * pr.up_Wp + (s_x >> 1)) * p.Cp;
        upos = (unsigned)(s_y & 1) * 2u;
        ug = make_uint4(0u, 0u, 0u, 0u); ucd = make_uint2(0x08080808u, 0x08080808u);       // code 8: routed nowhere -> zeros
        if (2 * uj < lim_px) {
          ug = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.P) + base + uoff);
          ucd = *reinterpret_cast<const uint2*>(pr.up_codes + base + uoff);
        }
      }
      const unsigned cP = offP, cQ = offQ;
      int lim;                    // LDS rows of this step that hold pixels (dy; the x strip has two more on wide images)
      bool full;
      if (NARROW) {
        const int nrows = min(pr.nr, p.Hp - s_y);
        lim = nrows * pr.pitch; full = nrows == pr.nr;
        sub_staged = pr.skip_empty ? (lim + 13) >> 4 : 4;       // dy rows 0 .. nrows * pitch - 3
        s_y += pr.nr;
        offP += (unsigned)(pr.nr * (int)p.psH * 2); offQ += (unsigned)(pr.nr * (int)p.qsH * 2);
        if (s_y >= p.Hp) {
          s_y = 0; ++s_n;
          offP += (unsigned)(((int)p.psN - pr.units_per_image * pr.nr * (int)p.psH) * 2);
          offQ += (unsigned)(((int)p.qsN - pr.units_per_image * pr.nr * (int)p.qsH) * 2);
        }
      } else {
        lim = min(64, p.Wp - s_x); full = lim == 64;
        sub_staged = pr.skip_empty ? (lim + 15) >> 4 : 4;
        s_x += 64;
        offP += (unsigned)(128 * psW); offQ += (unsigned)(128 * qsW);
        if (s_x >= p.Wp) {
          s_x = 0;
          offP += (unsigned)(((int)p.psH - pr.segs_per_row * 64 * psW) * 2);
          offQ += (unsigned)(((int)p.qsH - pr.segs_per_row * 64 * qsW) * 2);
          if (++s_y == p.Hp) {
            s_y = 0; ++s_n;
            offP += (unsigned)(((int)p.psN - p.Hp * (int)p.psH) * 2);
            offQ += (unsigned)(((int)p.qsN - p.Hp * (int)p.qsH) * 2);
          }
        }
      }
      if (full) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          if (!live[i]) continue;
          if (i < NPCP / NW) { if constexpr (!UNPOOL) buf_lds16(rsP, buf + ldst[i], loff[i], cP); }
          else buf_lds16(rsQ, buf + ldst[i], loff[i], cQ);
        }
      } else {
        const int limQ = NARROW ? lim : lim + 2;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
          if (!live[i]) continue;
          const int piece = wave + i * NW;
          if (i < NPCP / NW) {
            if constexpr (!UNPOOL) {
            const int v = lrowP < lim - piece * RPIP ? loff[i] : (int)0x80000000u;
            buf_lds16(rsP, buf + ldst[i], v, cP);
            }
          } else {
            const int v = lrowQ < limQ - (piece - NPCP) * RPIQ ? loff[i] : (int)0x80000000u;
            buf_lds16(rsQ, buf + ldst[i], v, cQ);
          }
        }
      }
      return;
    }
    const int n = s_n, ybase = s_y, x0 = s_x;
    int len, nrows;
    if (NARROW) {
      len = p.Wp; nrows = min(pr.nr, p.Hp - ybase);
      s_y += pr.nr;
      if (s_y >= p.Hp) { s_y = 0; ++s_n; }
    } else {
      len = min(64, p.Wp - x0); nrows = 1;
      s_x += 64;
      if (s_x >= p.Wp) { s_x = 0; if (++s_y == p.Hp) { s_y = 0; ++s_n; } }
    }
    const int iy0 = ybase + tr - p.pad_h, ix0 = x0 - p.pad_w;
    const long long baseP = ((long long)n * p.psN + (long long)ybase * p.psH + (long long)x0 * p.psW) * 2;
    const long long baseQ = ((long long)n * p.qsN + (long long)iy0 * p.qsH + (long long)ix0 * p.qsW) * 2;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      if (!live[i]) continue;
      const char* src;
      if (isP[i]) src = (rho[i] < nrows && col[i] < len) ? Pb + baseP + loff[i] : zero;
      else
        src = (rho[i] < nrows && col[i] < len + 2 && (unsigned)(iy0 + rho[i]) < (unsigned)p.Hq &&
               (unsigned)(ix0 + col[i]) < (unsigned)p.Wq) ? Qb + baseQ + loff[i] : zero;
      __builtin_amdgcn_global_load_lds((wg_gptr_t)src, (wg_lptr_t)(buf + ldst[i]), 16, 0, 0);
    }
  };

  f32x16 acc[3][TP][TQ];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
      for (int j = 0; j < TQ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][i][j][e] = 0.f;

  // UNPOOL: the staged step's two pixel rows of this thread's window into the dy tile of `buf`
  auto unpool_store = [&](char* buf) {
    *reinterpret_cast<uint4*>(buf + udst) = dct_unpool_chunk8(ug, ucd, upos);
    *reinterpret_cast<uint4*>(buf + udst + RBP) = dct_unpool_chunk8(ug, ucd, upos + 1u);
  };
  if (gbeg < gend) {
    stage(smem);
    if constexpr (UNPOOL) unpool_store(smem);
  }
  __syncthreads();
  int cur = 0;
  int nsub = sub_staged;             // sub-steps of the step about to be multiplied
  int pbase[TP], qbase[3][TQ];
  {
    const int g = lane >> 4, li = lane & 15, lq = li >> 2, lpp = li & 3, lh = g >> 1;
    const int kq0 = 8 * lh + lq;
#pragma unroll
    for (int i = 0; i < TP; ++i) {
      const int colb = (wp * WPR + i * 32 + 16 * (g & 1) + 4 * lpp) * 2;
      pbase[i] = kq0 * RBP + (colb ^ (swz3<RBP>(kq0) << 4));
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        const int colb = (wq * (BQ / 2) + j * 32 + 16 * (g & 1) + 4 * lpp) * 2;
        qbase[s][j] = 64 * RBP + (kq0 + s) * RBQ + (colb ^ (swz3<RBQ>(kq0 + s) << 4));
      }
  }
  constexpr int P_KK = 16 * RBP, P_HI = 4 * RBP, Q_KK = 16 * RBQ, Q_HI = 4 * RBQ;
  constexpr int NRD = 2 * TP + (QSHIFT ? 3 : 6) * TQ;
  const unsigned smem_off = lds_off(smem);
  const bool do_bias = pr.with_bias && tr == 0 && qt == 0 && wq == 0;
  // bias gradient = column sums of dy: a lane's dy fragment is 8 pixels of ONE channel (row l31 of the MFMA A operand),
  // so the sum is 8 VALU adds per sub-step in one register (an MFMA against a ones fragment would pin 16 accumulators
  // in every block of the launch and cost a wave per SIMD of occupancy)
  float accb[TP];
#pragma unroll
  for (int i = 0; i < TP; ++i) accb[i] = 0.f;

  constexpr int QF = QSHIFT ? 1 : 3, QR = QSHIFT ? 3 : 2;     // fragment groups per column block and transposing reads per group
#ifdef DCT_W3_STAMPS
  unsigned long long st_dma = 0, st_comp = 0, st_bar = 0, st_t0 = w3_stamp();
  const unsigned long long st_begin = st_t0;
#endif
  for (int it = 0; it < per_group; ++it) {
    const int gi = gbeg + it;
    if (gi + 1 < gend) stage(smem + (cur ^ 1) * STAGE);
#ifdef DCT_W3_STAMPS
    { const unsigned long long t = w3_stamp(); st_dma += t - st_t0; st_t0 = t; }
#endif
    if (gi < gend) {                    // wave-uniform: a group with one step fewer only keeps the barrier     // all pieces up front: spreading them between the MFMA groups lands the stage later and was 8 % slower
    const unsigned Pl = smem_off + cur * STAGE;
    bf16x4 fa[2][TP][2], fb[2][QF][TQ][QR];
    // the MFMAs (and the bias sums) of one 16-pixel sub-step on fragment set `set`
    auto compute = [&](auto setc) {
      constexpr int set = decltype(setc)::value;
#pragma unroll
      for (int i = 0; i < TP; ++i) { touch(fa[set][i][0]); touch(fa[set][i][1]); }
#pragma unroll
      for (int s = 0; s < QF; ++s)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
          for (int r = 0; r < QR; ++r) touch(fb[set][s][j][r]);
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 a[TP];
#pragma unroll
      for (int i = 0; i < TP; ++i) a[i] = join(fa[set][i][0], fa[set][i][1]);
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
          bf16x8 b;
          if constexpr (QSHIFT) {
            // the lane's 12 consecutive pixels of its x channel as six dwords (two pixels each, the earlier one in the low half)
            union W { bf16x4 v; unsigned d[2]; };
            W w0, w1, w2; w0.v = fb[set][0][j][0]; w1.v = fb[set][0][j][1]; w2.v = fb[set][0][j][2];
            const unsigned D0 = w0.d[0], D1 = w0.d[1], D2 = w1.d[0], D3 = w1.d[1], D4 = w2.d[0];
            union F { bf16x8 v; unsigned d[4]; } f;
            if (s == 0) { f.d[0] = D0; f.d[1] = D1; f.d[2] = D2; f.d[3] = D3; }
            else if (s == 2) { f.d[0] = D1; f.d[1] = D2; f.d[2] = D3; f.d[3] = D4; }
            else {
              f.d[0] = __builtin_amdgcn_alignbit(D1, D0, 16); f.d[1] = __builtin_amdgcn_alignbit(D2, D1, 16);
              f.d[2] = __builtin_amdgcn_alignbit(D3, D2, 16); f.d[3] = __builtin_amdgcn_alignbit(D4, D3, 16);
            }
            b = f.v;
          } else {
            b = join(fb[set][s][j][0], fb[set][s][j][1]);
          }
#pragma unroll
          for (int i = 0; i < TP; ++i) acc[s][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, acc[s][i][j], 0, 0, 0);
        }
      if (do_bias) {
        if constexpr (LEAN) {
          // a real (scalar) branch: without the barrier hipcc computes the four dot products in EVERY wave and selects
          asm volatile("" ::: "memory");
          // v_dot2c_f32_bf16 against (1, 1): two pixels per instruction instead of an unpack + add each (16 -> 4 per fragment)
          typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
          const bf16x2_t ones2 = {(bf16_t)1.0f, (bf16_t)1.0f};
#pragma unroll
          for (int i = 0; i < TP; ++i) {
            union { bf16x8 v; bf16x2_t h[4]; } u; u.v = a[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) accb[i] = __builtin_amdgcn_fdot2_f32_bf16(u.h[e], ones2, accb[i], false);
          }
        } else {
#pragma unroll
          for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) accb[i] += (float)a[i][e];
        }
      }
    };
    constexpr std::integral_constant<int, 0> set0{};
    constexpr std::integral_constant<int, 1> set1{};
    if constexpr (LEAN) {
      // one address register per fragment column block; sub-step (kk) and row-group offsets ride in the reads' immediates
      unsigned pa[TP], qa[QF][TQ];
#pragma unroll
      for (int i = 0; i < TP; ++i) pa[i] = Pl + pbase[i];
#pragma unroll
      for (int s = 0; s < QF; ++s)
#pragma unroll
        for (int j = 0; j < TQ; ++j) qa[s][j] = Pl + qbase[s][j];
#define DCT_W3_ISSUE(set, KK)                                                                                       \
      {                                                                                                             \
        _Pragma("unroll") for (int i = 0; i < TP; ++i) {                                                            \
          tr_issue_o<(KK) * P_KK>(pa[i], fa[set][i][0]);                                                            \
          tr_issue_o<(KK) * P_KK + P_HI>(pa[i], fa[set][i][1]);                                                     \
        }                                                                                                           \
        _Pragma("unroll") for (int s = 0; s < QF; ++s)                                                              \
          _Pragma("unroll") for (int j = 0; j < TQ; ++j) {                                                          \
            tr_issue_o<(KK) * Q_KK>(qa[s][j], fb[set][s][j][0]);                                                    \
            tr_issue_o<(KK) * Q_KK + Q_HI>(qa[s][j], fb[set][s][j][1]);                                             \
            if constexpr (QR == 3) tr_issue_o<(KK) * Q_KK + 2 * Q_HI>(qa[s][j], fb[set][s][j][QR - 1]);             \
          }                                                                                                         \
      }
      // (nsub is wave-uniform: scalar branches)
      DCT_W3_ISSUE(0, 0)
      if (nsub > 1) { DCT_W3_ISSUE(1, 1) lgkm_wait<NRD>(); } else { lgkm_wait<0>(); }
      compute(set0);
      if (nsub > 1) {
        if (nsub > 2) { DCT_W3_ISSUE(0, 2) lgkm_wait<NRD>(); } else { lgkm_wait<0>(); }
        compute(set1);
        if (nsub > 2) {
          if (nsub > 3) { DCT_W3_ISSUE(1, 3) lgkm_wait<NRD>(); } else { lgkm_wait<0>(); }
          compute(set0);
          if (nsub > 3) { lgkm_wait<0>(); compute(set1); }
        }
      }
#undef DCT_W3_ISSUE
    } else {
    auto issue = [&](int set, int kk) {
#pragma unroll
      for (int i = 0; i < TP; ++i) {
        tr_issue(Pl + pbase[i] + kk * P_KK, fa[set][i][0]);
        tr_issue(Pl + pbase[i] + kk * P_KK + P_HI, fa[set][i][1]);
      }
#pragma unroll
      for (int s = 0; s < QF; ++s)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
          for (int r = 0; r < QR; ++r) tr_issue(Pl + qbase[s][j] + kk * Q_KK + r * Q_HI, fb[set][s][j][r]);     // rows +0..3, +4..7 [, +8..11]
    };
    issue(0, 0);
    issue(1, 1); lgkm_wait<NRD>(); compute(set0);
    issue(0, 2); lgkm_wait<NRD>(); compute(set1);
    issue(1, 3); lgkm_wait<NRD>(); compute(set0);
    lgkm_wait<0>(); compute(set1);
    }
    }
#ifdef DCT_W3_STAMPS
    { const unsigned long long t = w3_stamp(); st_comp += t - st_t0; st_t0 = t; }
#endif
    if constexpr (UNPOOL) {
      if (gi + 1 < gend) unpool_store(smem + (cur ^ 1) * STAGE);
    }
    __syncthreads();
#ifdef DCT_W3_STAMPS
    { const unsigned long long t = w3_stamp(); st_bar += t - st_t0; st_t0 = t; }
#endif
    cur ^= 1;
    nsub = sub_staged;
  }
#ifdef DCT_W3_STAMPS
  if (pr.stamps && lane == 0 && blockIdx.x < 4096) {
    unsigned long long* o = pr.stamps + ((size_t)blockIdx.x * (G * NW) + wave_all) * 8;
    o[0] = st_dma; o[1] = st_comp; o[2] = st_bar; o[3] = st_t0 - st_begin; o[4] = (unsigned long long)per_group; o[5] = st_begin;
  }
#endif
  if constexpr (G > 1) {
    // fold the groups: group g > 0 parks its accumulators in LDS (lane-linear), group 0 adds them in order
    float* fold = reinterpret_cast<float*>(smem_all);
    constexpr int PER_WAVE = (3 * TP * TQ * 16 + TP) * 64;
    static_assert((size_t)NW * PER_WAVE * 4 <= (size_t)G * 2 * STAGE, "fold buffer does not fit");
#pragma unroll 1
    for (int g = 1; g < G; ++g) {
      if (grp == g) {
        float* dstp = fold + wave * PER_WAVE + lane;
        int o = 0;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int j = 0; j < TQ; ++j)
#pragma unroll
              for (int e = 0; e < 16; ++e) dstp[64 * (o++)] = acc[s][i][j][e];
#pragma unroll
        for (int i = 0; i < TP; ++i) dstp[64 * (o++)] = accb[i];
      }
      __syncthreads();
      if (grp == 0) {
        const float* srcp = fold + wave * PER_WAVE + lane;
        int o = 0;
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int j = 0; j < TQ; ++j)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[s][i][j][e] += srcp[64 * (o++)];
#pragma unroll
        for (int i = 0; i < TP; ++i) accb[i] += srcp[64 * (o++)];
      }
      __syncthreads();
    }
    if (grp != 0) return;
  }

  const int half = lane >> 5, l31 = lane & 31;
  float* base = pr.direct ? p.out : p.out + (long long)chunk * pr.slab_stride;
  if (do_bias) {
    float* bb = pr.direct ? pr.bias : base + (long long)p.Cp * 9 * p.Cq;
#pragma unroll
    for (int i = 0; i < TP; ++i) {
      const float t = accb[i] + __shfl_xor(accb[i], 32, 64);       // the two 8-pixel halves of each sub-step
      if (half == 0) {
        const int prow = p0 + wp * WPR + i * 32 + l31;
        if (pr.direct && pr.accumulate) bb[prow] += t; else bb[prow] = t;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        const int qc = q0 + wq * (BQ / 2) + j * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int prow = p0 + wp * WPR + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
          float* dst = base + ((long long)prow * 9 + (tr * 3 + s)) * p.Cq + qc;
          if (pr.direct && pr.accumulate) *dst += acc[s][i][j][e]; else *dst = acc[s][i][j][e];
        }
      }
}

// out[i] (=|+=) sum_slab partial[slab][i] for the n4w float4s of the weight gradient and, behind them in every
// slab, the n4b float4s of the bias gradient.  Fixed order (four interleaved partial sums: loads in flight).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, float* dw, float* db, long long n4w, long long n4b,
                                                           long long stride, int slabs, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4w + n4b) return;
  f32x4* outp = i < n4w ? reinterpret_cast<f32x4*>(dw) + i : reinterpret_cast<f32x4*>(db) + (i - n4w);
  f32x4 s0 = accumulate ? *outp : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
  const f32x4* src = reinterpret_cast<const f32x4*>(partial) + i;
  const long long st4 = stride / 4;
  int c = 0;
  for (; c + 3 < slabs; c += 4) {
    s0 += src[(long long)c * st4]; s1 += src[(long long)(c + 1) * st4];
    s2 += src[(long long)(c + 2) * st4]; s3 += src[(long long)(c + 3) * st4];
  }
  for (; c < slabs; ++c) s0 += src[(long long)c * st4];
  *outp = (s0 + s1) + (s2 + s3);
}

struct WPlan { int bp, bq, chunks, ppc, ptiles, qtiles, v2, direct, slabs; int v3, segs_per_row, nseg, seg_per_chunk, pitch, nr, units, groups; };

int g_tune_wgrad_target = 256;     // block target of the per-tap kernel (slab bytes = blocks x 64 KiB; round 4, final sweep: 256 is 0.7 % ahead of 384 / 320 / 192 on the step).  In isolation ~1150 blocks is
                                   // fastest; on the whole step (tools/ab_step.py --knob 14) 256-768 are level and 1.5 % ahead of 1150
int g_tune_wgrad3_target = 768;    // same for the filter-row kernel, in 4-wave units
int g_tune_wgrad_rows_fill = 70;   // percent: minimum fill of the 64-row K-steps for the filter-row kernel
int g_tune_wgrad_rows = 1;     // 3x3 stride-1 layers on wide images: three taps of a filter row per block (wgrad3_kernel)
int g_tune_wgrad_chunks = -1;  // >= 1 forces the number of pixel chunks

static bool make_wplan(const dct_view* p, const dct_view* q, const dct_conv_desc* d, int dtype, WPlan& pl) {
  if (p->c % 64 || q->c % 64) return false;
  const long long M = (long long)p->n * p->h * p->w;
  pl.bp = (p->c % 128 == 0) ? 128 : 64;
  pl.bq = (q->c % 128 == 0) ? 128 : 64;
  pl.ptiles = p->c / pl.bp; pl.qtiles = q->c / pl.bq;
  const bool fits32 = (long long)p->n * p->sn < (1ll << 30) && (long long)q->n * q->sn < (1ll << 30);   // 32-bit element offsets
  pl.v2 = (dtype == DCT_BF16 && M < (1 << 24) && fits32) ? 1 : 0;
  const int bkp = pl.v2 ? 64 : (dtype == DCT_BF16 ? 32 : 16);
  const long long tiles = (long long)pl.ptiles * pl.qtiles * d->R * d->S;
  long long chunks;
  if (pl.v2) {
    // measured (tools/bench_conv.py chunk sweep): ~1150 blocks in total, at most 64 slabs to fold
    chunks = (g_tune_wgrad_target + tiles / 2) / tiles;
    if (chunks > 64) chunks = 64;
    const long long max_by_pix = (M + 4 * bkp - 1) / (4 * bkp);      // >= 4 K-steps per chunk
    if (chunks > max_by_pix) chunks = max_by_pix;
  } else {
    chunks = (1536 + tiles - 1) / tiles;                      // aim for ~1.5k blocks
    const long long max_by_pix = (M + 8 * bkp - 1) / (8 * bkp);      // >= 8 K-steps per chunk
    if (chunks > max_by_pix) chunks = max_by_pix;
  }
  // bound the partial-sum workspace to 192 MiB
  const long long per_chunk = (long long)p->c * q->c * d->R * d->S * 4;
  while (chunks > 1 && chunks * per_chunk > (192ll << 20)) --chunks;
  if (g_tune_wgrad_chunks >= 1) chunks = g_tune_wgrad_chunks;
  if (chunks < 1) chunks = 1;
  long long ppc = (M + chunks - 1) / chunks;
  ppc = (ppc + bkp - 1) / bkp * bkp;
  pl.ppc = (int)ppc;
  pl.chunks = (int)((M + ppc - 1) / ppc);
  pl.slabs = pl.chunks;
  pl.direct = (pl.v2 && pl.chunks == 1) ? 1 : 0;
  pl.v3 = 0;
  if (pl.v2 && g_tune_wgrad_rows && d->R == 3 && d->S == 3 && d->stride == 1 && d->dil == 1) {
    // K-steps of 64 LDS rows: wide images -- runs of <= 64 pixels of one dy row; narrow ones -- nr whole rows at a pitch
    // of Wp + 2.  Worth it when the steps are mostly full (threshold measured with tools/bench_conv.py --ab-wgrad).
    const int Wp = p->w;
    int segs = 1, pitch = 0, nr = 1;
    double fill;
    long long units;
    // the lean loop multiplies only the 16-row sub-steps of a step that hold dy pixels: the fill is counted in those
    const bool skips = (g_tune_lean & 17) == 17 && d->pad_h == 0 && d->pad_w == 0;
    if (Wp > 64) {
      segs = (Wp + 63) / 64; units = (long long)p->h * segs;
      const int tail = Wp - 64 * (segs - 1);
      fill = skips ? (double)Wp / (16.0 * (4 * (segs - 1) + (tail + 15) / 16)) : (double)Wp / (segs * 64.0);
    } else {
      pitch = Wp + 2; nr = 66 / pitch; if (nr < 1) nr = 1;
      const int steps = (p->h + nr - 1) / nr;
      // (narrow images keep the whole-step count: what the skipping saves there is an image's LAST step, and on the layers that
      //  would newly qualify -- cen_a, cen_b, enc4a: one pixel chunk, direct -- the per-tap kernel measured 6-11 % faster:
      //  profiles/r05_wgrad3_substep_skip_ab.txt)
      fill = (double)p->h * Wp / (steps * 64.0);
      units = steps;
    }
    if (fill >= g_tune_wgrad_rows_fill * 0.01 && (long long)p->h * p->sh < (1ll << 29) && (long long)q->h * q->sh < (1ll << 29)) {
      const long long nseg = (long long)p->n * units;
      // measured (tools/bench_conv.py --ab-wgrad): 64 x 64 tiles (4 waves, 113 registers: four waves per SIMD) beat
      // 128 x 64 and 128 x 128 (one 8-wave block per CU at 130 / 205 VGPRs) on every eligible layer, by 20-35 %
      pl.bp = 64; pl.ptiles = p->c / 64; pl.bq = 64; pl.qtiles = q->c / 64;
      const int tiles3 = pl.ptiles * pl.qtiles * 3;
      const int target = g_tune_wgrad3_target;     // 4-wave units; swept 384 / 512 / 640 / 768 / 1024 on the UNet layers: 768 is 10-25 % ahead of the rest
      pl.groups = 2;     // four groups (one 16-wave block per CU) measured 0.7 % behind two     // two wave groups per block: half the slabs at the same waves per CU
      long long ch = (target / pl.groups + tiles3 / 2) / tiles3;
      if (ch > 256) ch = 256;
      if (ch > nseg / (4 * pl.groups)) ch = nseg / (4 * pl.groups);
      while (ch > 1 && ch * per_chunk > (192ll << 20)) --ch;
      if (g_tune_wgrad_chunks >= 1) ch = g_tune_wgrad_chunks;
      if (ch < 1) ch = 1;
      const long long spc = (nseg + ch - 1) / ch;
      pl.v3 = 1;
      pl.segs_per_row = segs; pl.nseg = (int)nseg; pl.seg_per_chunk = (int)spc;
      pl.pitch = pitch; pl.nr = nr; pl.units = (int)units;
      pl.chunks = (int)((nseg + spc - 1) / spc);
      pl.slabs = pl.chunks;
      pl.direct = pl.chunks == 1 ? 1 : 0;
    }
  }
  return true;
}

static unsigned wgrid(const WPlan& pl, int taps) {
  const int tiles = pl.ptiles * pl.qtiles * taps;
  return (unsigned)((pl.chunks >= 8 ? ((pl.chunks + 7) / 8) * 8 : pl.chunks) * tiles);
}

template <typename T>
static void launch_w(const WgradParams& wp, const WPlan& pl, hipStream_t st) {
  const unsigned grid = wgrid(pl, wp.R * wp.S);
  if (pl.bp == 128 && pl.bq == 128) DCT_LAUNCH(DCT_PROF_WGRAD, (wgrad_kernel<T, 128, 128>), dim3(grid), dim3(256), 0, st, wp);
  else if (pl.bp == 128) DCT_LAUNCH(DCT_PROF_WGRAD, (wgrad_kernel<T, 128, 64>), dim3(grid), dim3(256), 0, st, wp);
  else if (pl.bq == 128) DCT_LAUNCH(DCT_PROF_WGRAD, (wgrad_kernel<T, 64, 128>), dim3(grid), dim3(256), 0, st, wp);
  else DCT_LAUNCH(DCT_PROF_WGRAD, (wgrad_kernel<T, 64, 64>), dim3(grid), dim3(256), 0, st, wp);
}

template <int BP, int BQ, int NW, bool LEAN>
static void launch_w2_k(const Wgrad2Params& pr, unsigned grid, hipStream_t st) {
  constexpr size_t lds = 2 * 64 * (size_t)(BP + BQ) * 2 + (LEAN ? 1024 : 0);      // two stages [+ the offset table of the lean form]
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad2_kernel<BP, BQ, NW, LEAN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  DCT_LAUNCH_FAM(DCT_FAM_WGRAD2, DCT_PROF_WGRAD, (wgrad2_kernel<BP, BQ, NW, LEAN>), dim3(grid), dim3(NW * 64), lds, st, pr);
}
template <int BP, int BQ, int NW>
static void launch_w2_t(const Wgrad2Params& pr, unsigned grid, hipStream_t st) {
  if ((g_tune_lean & 4) && pr.p_bytes < (1ll << 31) && pr.q_bytes < (1ll << 31)) launch_w2_k<BP, BQ, NW, true>(pr, grid, st);
  else launch_w2_k<BP, BQ, NW, false>(pr, grid, st);
}
template <int BP, int BQ, int NW, bool NARROW, int G, bool QSHIFT, bool LEAN, bool UNPOOL = false>
static void launch_w3_q(const Wgrad3Params& pr, unsigned grid, hipStream_t st) {
  constexpr size_t lds = G * 2 * (64 * (size_t)BP * 2 + 72 * (size_t)BQ * 2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3_kernel<BP, BQ, NW, NARROW, G, QSHIFT, LEAN, UNPOOL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  DCT_LAUNCH_FAM(DCT_FAM_WGRAD3, DCT_PROF_WGRAD, (wgrad3_kernel<BP, BQ, NW, NARROW, G, QSHIFT, LEAN, UNPOOL>), dim3(grid), dim3(G * NW * 64), lds, st, pr);
}
// lean form of the filter-row kernel: no padding, views under 2 GiB (buffer descriptors)
static bool w3_lean(const Wgrad3Params& pr) {
  return (g_tune_lean & 1) && pr.w.pad_h == 0 && pr.w.pad_w == 0 && pr.p_bytes < (1ll << 31) && pr.q_bytes < (1ll << 31);
}
static void launch_w3(const Wgrad3Params& pr, const WPlan& pl, hipStream_t st) {
  const unsigned grid = wgrid(pl, 3);
  if (pr.up_codes) {        // (the caller has checked: lean form, one image row per K-step)
    if (pr.pitch > 0) launch_w3_q<64, 64, 4, true, 2, true, true, true>(pr, grid, st);
    else launch_w3_q<64, 64, 4, false, 2, true, true, true>(pr, grid, st);
    return;
  }
  // the planner only picks 64 x 64 tiles for this kernel; always two wave groups per block (half the fp32 slabs at the same waves
  // per CU: +4 % on the step; one group and four groups measured behind) and the three taps' x fragments from ONE 12-pixel window
  // per lane (QSHIFT: 5 transposing reads per sub-step instead of 8, +5-6.5 % on the twelve layers that take this kernel)
  // LEAN (the staging as buffer loads with constant lane offsets; bit-identical): layers without padding whose views stay under 2 GiB
  const bool lean = w3_lean(pr);
  if (pr.pitch > 0) {
    if (lean) launch_w3_q<64, 64, 4, true, 2, true, true>(pr, grid, st); else launch_w3_q<64, 64, 4, true, 2, true, false>(pr, grid, st);
  } else {
    if (lean) launch_w3_q<64, 64, 4, false, 2, true, true>(pr, grid, st); else launch_w3_q<64, 64, 4, false, 2, true, false>(pr, grid, st);
  }
}
static void launch_w2(const Wgrad2Params& pr, const WPlan& pl, hipStream_t st) {
  const unsigned grid = wgrid(pl, pr.w.R * pr.w.S);
  if (pl.bp == 128 && pl.bq == 128) {
    launch_w2_t<128, 128, 8>(pr, grid, st);      // eight waves: half the LDS-DMA pieces and decode work per wave
  } else if (pl.bp == 128) launch_w2_t<128, 64, 4>(pr, grid, st);
  else if (pl.bq == 128) launch_w2_t<64, 128, 4>(pr, grid, st);
  else launch_w2_t<64, 64, 4>(pr, grid, st);
}

}  // namespace

unsigned long long* g_w3_stamps = nullptr;    // diagnostic builds only
extern "C" int dct_debug_w3_stamps(void* buf) { g_w3_stamps = (unsigned long long*)buf; return 0; }

// the view the reduction runs over: p itself, or (d->unpool_codes) the dense un-pooled tensor p stands for
static dct_view wgrad_p_extent(const dct_view* p, const dct_conv_desc* d) {
  dct_view v = *p;
  if (d->unpool_codes) {
    v.h = d->unpool_h; v.w = d->unpool_w;
    v.sw = v.c; v.sh = (long long)v.w * v.c; v.sn = (long long)v.h * v.w * v.c;
  }
  return v;
}

extern "C" size_t dct_conv2d_wgrad_workspace_bytes(const dct_view* p0, const dct_view* q, const dct_conv_desc* d, int dtype) {
  if (!p0 || !q || !d) return 0;
  const dct_view pe = wgrad_p_extent(p0, d);
  const dct_view* p = &pe;
  WPlan pl;
  if (!make_wplan(p, q, d, dtype, pl)) return 0;
  if (d->unpool_codes && !(pl.v3 && (pl.pitch == 0 || pl.nr == 1))) return 0;
  return pl.direct ? 16 : (size_t)pl.slabs * ((size_t)p->c * q->c * d->R * d->S + p->c) * sizeof(float);
}

extern "C" int dct_conv2d_wgrad(const dct_view* p, const dct_view* q, float* dw, const dct_conv_desc* d, int dtype,
                                void* workspace, size_t workspace_bytes, dct_stream stream) {
  return dct_conv2d_wgrad_bias(p, q, dw, nullptr, d, dtype, workspace, workspace_bytes, stream);
}

extern "C" int dct_conv2d_wgrad_bias(const dct_view* p0, const dct_view* q, float* dw, float* db, const dct_conv_desc* d, int dtype,
                                     void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (!view_ok(p0) || !view_ok(q) || !dw || !d) return DCT_ERR_BAD_ARG;
  if (dtype != DCT_F32 && dtype != DCT_BF16) return DCT_ERR_BAD_ARG;
  if (d->unpool_codes) {
    // p0 is the DENSE gradient at the pooled tensor; the reduction runs over the un-pooled extent, expanded while the kernel stages
    if (d->unpool_h < 1 || d->unpool_w < 1 || p0->h != (d->unpool_h + 1) / 2 || p0->w != (d->unpool_w + 1) / 2) return DCT_ERR_BAD_ARG;
    if (dtype != DCT_BF16 || d->R != 3 || d->S != 3 || d->stride != 1 || d->dil != 1 || d->pad_h || d->pad_w || ((uintptr_t)d->unpool_codes & 7) ||
        p0->sw != p0->c || p0->sh != (long long)p0->w * p0->c || p0->sn != (long long)p0->h * p0->w * p0->c || !(g_tune_lean & 1))
      return DCT_ERR_UNSUPPORTED;
  }
  const dct_view pe = wgrad_p_extent(p0, d);
  const dct_view* p = &pe;
  if (p->n != q->n) return DCT_ERR_BAD_ARG;
  {
    const int eh = (q->h + 2 * d->pad_h - d->dil * (d->R - 1) - 1) / d->stride + 1;
    const int ew = (q->w + 2 * d->pad_w - d->dil * (d->S - 1) - 1) / d->stride + 1;
    if (eh != p->h || ew != p->w) return DCT_ERR_BAD_ARG;
  }
  const int esz = dtype == DCT_BF16 ? 2 : 4, epv = 16 / esz;
  if (((uintptr_t)p->ptr & 15) || ((uintptr_t)q->ptr & 15) || (p->sw % epv) || (p->sh % epv) || (p->sn % epv) ||
      (q->sw % epv) || (q->sh % epv) || (q->sn % epv) || ((uintptr_t)dw & 15))
    return DCT_ERR_UNSUPPORTED;
  WPlan pl;
  if (!make_wplan(p, q, d, dtype, pl)) return DCT_ERR_UNSUPPORTED;
  if (db && (!pl.v2 || ((uintptr_t)db & 15))) return DCT_ERR_UNSUPPORTED;   // the fused bias gradient lives in the bf16 LDS-DMA kernel
  const long long E = (long long)p->c * q->c * d->R * d->S;
  const long long slab_stride = E + (db ? p->c : 0);
  const size_t need = pl.direct ? 0 : (size_t)pl.slabs * slab_stride * sizeof(float);
  if (need && (!workspace || workspace_bytes < need)) return DCT_ERR_WORKSPACE;
  WgradParams wp;
  wp.P = (const char*)p->ptr; wp.Q = (const char*)q->ptr; wp.out = pl.direct ? dw : (float*)workspace;
  wp.M = p->n * p->h * p->w; wp.Cp = p->c; wp.Cq = q->c; wp.R = d->R; wp.S = d->S;
  wp.Hp = p->h; wp.Wp = p->w; wp.Hq = q->h; wp.Wq = q->w;
  wp.stride = d->stride; wp.dil = d->dil; wp.pad_h = d->pad_h; wp.pad_w = d->pad_w;
  wp.psN = p->sn; wp.psH = p->sh; wp.psW = p->sw; wp.qsN = q->sn; wp.qsH = q->sh; wp.qsW = q->sw;
  wp.chunks = pl.chunks; wp.pix_per_chunk = pl.ppc; wp.ptiles = pl.ptiles; wp.qtiles = pl.qtiles;
  hipStream_t st = (hipStream_t)stream;
  if (pl.v3) {
    Wgrad3Params pr;
    pr.w = wp;
    pr.segs_per_row = pl.segs_per_row; pr.nseg = pl.nseg; pr.seg_per_chunk = pl.seg_per_chunk;
    pr.pitch = pl.pitch; pr.nr = pl.nr; pr.units_per_image = pl.units;
    pr.direct = pl.direct; pr.accumulate = d->accumulate;
    pr.bias = db; pr.with_bias = db ? 1 : 0; pr.slab_stride = slab_stride;
    pr.skip_empty = (g_tune_lean & 16) ? 1 : 0;
    pr.up_codes = d->unpool_codes; pr.up_Hp = p0->h; pr.up_Wp = p0->w;
    pr.p_bytes = ((long long)(p->n - 1) * p->sn + (long long)(p->h - 1) * p->sh + (long long)(p->w - 1) * p->sw + p->c) * 2;
    pr.q_bytes = ((long long)(q->n - 1) * q->sn + (long long)(q->h - 1) * q->sh + (long long)(q->w - 1) * q->sw + q->c) * 2;
    pr.stamps = g_w3_stamps;
    if (d->unpool_codes) {
      pr.w.P = (const char*)p0->ptr;
      if (!w3_lean(pr) || !(pl.pitch == 0 || pl.nr == 1)) return DCT_ERR_UNSUPPORTED;      // (nothing has been launched)
    }
    launch_w3(pr, pl, st);
  } else if (d->unpool_codes) {
    return DCT_ERR_UNSUPPORTED;          // only the filter-row kernel expands a pooled gradient while it stages
  } else if (pl.v2) {
    Wgrad2Params pr;
    pr.w = wp;
    pr.dhw.d = p->h * p->w; pr.dhw.rcp = 1.0f / (float)pr.dhw.d;
    pr.dw_.d = p->w; pr.dw_.rcp = 1.0f / (float)pr.dw_.d;
    pr.direct = pl.direct; pr.accumulate = d->accumulate;
    pr.bias = db; pr.with_bias = db ? 1 : 0; pr.slab_stride = slab_stride;
    pr.p_bytes = ((long long)(p->n - 1) * p->sn + (long long)(p->h - 1) * p->sh + (long long)(p->w - 1) * p->sw + p->c) * 2;
    pr.q_bytes = ((long long)(q->n - 1) * q->sn + (long long)(q->h - 1) * q->sh + (long long)(q->w - 1) * q->sw + q->c) * 2;
    launch_w2(pr, pl, st);
  } else if (dtype == DCT_BF16) launch_w<bf16_t>(wp, pl, st);
  else launch_w<float>(wp, pl, st);
  DCT_PLAN_NOTE("%s %d x %d tile: %d x %d x %d taps, %d pixel chunks%s%s", pl.v3 ? "wgrad3 filter-row" : pl.v2 ? "wgrad2 per-tap" : "wgrad",
                pl.bp, pl.bq, pl.ptiles, pl.qtiles, d->R * d->S, pl.chunks, pl.v3 ? (pl.pitch ? " (narrow rows)" : " (wide rows)") : "",
                pl.direct ? ", direct" : "");
  if (!pl.direct) {
    const long long n4b = db ? p->c / 4 : 0;
    DCT_LAUNCH_FAM(DCT_FAM_FOLDS, DCT_PROF_WGRAD, wgrad_reduce_kernel, dim3(div_up(E / 4 + n4b, 256)), dim3(256), 0, st,
               (const float*)workspace, dw, db, E / 4, n4b, slab_stride, pl.slabs, d->accumulate);
  }
  return dct_check_launch();
}

int dct_tune_set_wgrad(int knob, int value) {
  if (knob == DCT_TUNE_WGRAD_CHUNKS) { g_tune_wgrad_chunks = value; return DCT_OK; }
  if (knob == DCT_TUNE_WGRAD_ROWS) { g_tune_wgrad_rows = value; return DCT_OK; }
  if (knob == DCT_TUNE_WGRAD_ROWS_FILL) { g_tune_wgrad_rows_fill = value; return DCT_OK; }
  if (knob == DCT_TUNE_WGRAD3_TARGET) { if (value < 64) return DCT_ERR_BAD_ARG; g_tune_wgrad3_target = value; return DCT_OK; }
  if (knob == DCT_TUNE_LEAN) { g_tune_lean = value; return DCT_OK; }
  if (knob == DCT_TUNE_WGRAD_TARGET) { if (value < 64) return DCT_ERR_BAD_ARG; g_tune_wgrad_target = value; return DCT_OK; }
  return DCT_ERR_BAD_ARG;
}
