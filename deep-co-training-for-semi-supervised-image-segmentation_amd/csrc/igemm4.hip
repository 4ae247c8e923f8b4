// One-block-per-CU "ping-pong" implicit-GEMM kernel for 3x3 stride-1 convolutions on gfx950 (bf16): the forward pass and the
// data gradient of every UNet 3x3 layer (reference: generalframework/arch/network.py:153-171,196-240 -- the ATen conv2d /
// conv2d-backward-input calls behind UNet's convBatch / upSampleConv blocks).
//
// Why a second tile family.  The shared-halo tiles of igemm.hip (128 pixels x 128 channels, eight waves of 32 x 64, two blocks per
// CU, __syncthreads() per K-step) are bound by their structure: a wave issues 16 MFMAs per barrier, every barrier drains the
// LDS-DMA queue (vmcnt(0)), and a wave reads 768 B of LDS per MFMA (75 % of the LDS read rate at full MFMA rate) -- measured 33 %
// MFMA-pipe busy.  This kernel is the structure cdna_hip_programming.md section 5 ("256^2 8-phase template")
// describes, applied to a convolution:
//   * tile = a TH x TW rectangle of <= 256 output pixels of one image x BN = 128 channels; eight waves as 4 (pixels) x 2 (channels),
//     each wave a 64 x 64 register tile of v_mfma_f32_16x16x32_bf16 (64 accumulator registers, 512 B of LDS reads per MFMA);
//   * the (TH + 2) x (TW + 2) input halo of a 64-channel slice is staged ONCE for the nine taps (two halo buffers: the next
//     slice streams in while the current one is used); only a tap's 128 x 64 weight tile (16 KiB) streams per K-step, through
//     a ring of three stages filled 1.5 K-steps ahead;
//   * everything moves by global_load_lds_dwordx4 with COUNTED s_waitcnt vmcnt(N) and raw s_barrier -- the DMA queue is never
//     drained inside the loop;
//   * the two waves of a SIMD (wave w and w + 4) run half a phase apart ("ping-pong"): while group A issues the 16 MFMAs of a
//     32-deep half-step, group B reads its fragments from LDS and issues its DMA pieces, then they swap.  A phase is
//     {fragment reads, DMA issue, waits} - barrier - {16 MFMAs at raised priority} - barrier; group B enters the loop one barrier late.
//
// Synchronisation (p = phase = 2 * step + half; group A runs phase p between barriers 2p-1 .. 2p+1, group B between 2p .. 2p+2):
//   RAW  weights of step s + 1 are issued in the first half of step s - 1; every wave waits for them (vmcnt(2): only the two
//        pieces of step s + 2 may still be in flight) in the load segment of the SECOND half of step s, i.e. before a barrier that
//        every reader of step s + 1 passes later (the guide's rule "read a staged buffer one phase after the wait that retires it,
//        one barrier more for staggered groups").
//        The halo pieces of slice c + 1 are issued behind that wait in taps 0..5 of slice c, so the same wait retires them one
//        step later, three steps before their first read.
//   WAR  a wave retires its fragment reads (lgkmcnt(0)) BEFORE the phase's first barrier, so a ring stage last read in phase p may
//        be refilled by anyone who has passed a barrier after it: stage (s + 2) % 3 = (s - 1) % 3 is refilled in phase 2s.
//   The accumulation order per output (slice, tap, 32-channel half; one MFMA chain) is that of igemm3m_kernel, so the two kernels
//   agree bit for bit.
#include <algorithm>
#include "igemm_common.h"

namespace {

struct I4Geom {
  int TH, TW, HW;          // tile rows / columns of output pixels, halo pitch TW + 2
  int hrows, npix;         // (TH + 2) * HW halo rows of 128 B; TH * TW pixels (<= 256)
  int tiles_x, tiles_y;
  int cslices;             // 64-channel slices per block (split over blockIdx.z when < Cin / 64)
};

constexpr int I4_APIECES = 45;                       // halo capacity: 360 rows of 128 B
constexpr int I4_A_BYTES = I4_APIECES * 1024;
constexpr int I4_NB = 3;                             // weight ring stages

__device__ __forceinline__ void i4_vmcnt2() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
__device__ __forceinline__ void i4_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void i4_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void i4_barrier() { asm volatile("s_barrier" ::: "memory"); }

// ABL (diagnostic builds only, make EXTRA=-DDCT_I4_ABLATE): 0 = the kernel; bit 0: no weight DMA in the loop, bit 1: no halo DMA in the
// loop, bit 2: no fragment reads, bit 3: no MFMAs, bit 4: no barriers in the loop, bit 5: no vmcnt wait in the loop (timing studies; results are wrong)
template <int BN, int ABL = 0, int RDL = 1>
__global__ __launch_bounds__(512) void igemm4_kernel(IgemmParams p, I4Geom g) {
  constexpr int NW = 8, NWM = 4, NWN = BN / 64;
  static_assert(NWM * NWN == NW, "eight waves of 64 pixels x 64 channels");
  constexpr int BM = 256;
  constexpr int B_BYTES = BN * 128, NPB = BN / 8 / NW;          // weight stage; pieces per wave and stage (2)
  constexpr int NPA = (I4_APIECES + NW - 1) / NW;              // halo pieces per wave and slice (6)
  constexpr int W_OFF = 2 * I4_A_BYTES, BIAS_OFF = W_OFF + I4_NB * B_BYTES;
  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                     // SIMD partners are waves w and w + 4: one of each group per SIMD
  const int wm = wave & 3, wn = (NWN == 2) ? grp : 0;
  int bx = blockIdx.x;
  const int tx = bx % g.tiles_x; bx /= g.tiles_x;
  const int ty = bx % g.tiles_y; const int img = bx / g.tiles_y;
  const int y0 = ty * g.TH, x0 = tx * g.TW, n0 = blockIdx.y * BN;
  const int HW = g.HW;
  const long long Ktot = 9ll * p.Cin;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
  const char* zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 7) * 16;
  const unsigned smem_l = (unsigned)(size_t)(lptr_c)(smem);

  // ---- halo staging: wave w owns pieces w, w + 8, ...; lane -> (halo row = piece * 8 + lane / 8, swizzled source chunk).
  // 16-byte chunk c of halo pixel (hy, hx) sits at chunk c ^ ((hx >> 1) & 7) of LDS row hy * HW + hx: the swizzle depends on the
  // COLUMN only, so a lane's fragment address for tap (r, s) is (a per-lane constant for s) + r * HW * 128 -- no per-tap vector
  // arithmetic (the two waves of a SIMD share its vector issue: ~25 address instructions per load segment beside the partner's
  // MFMAs cost more than the reads themselves, tools/gpu/i4_ablate.py).  A 16-lane read group still covers 16 distinct slots.
  int aoff[NPA];
#pragma unroll
  for (int i = 0; i < NPA; ++i) {
    const int row = (wave + i * NW) * 8 + (lane >> 3);
    aoff[i] = -1;
    if (row < g.hrows) {
      const int hy = row / HW, hx = row - hy * HW;
      const int iy = y0 - p.pad_h + hy, ix = x0 - p.pad_w + hx;
      if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
        aoff[i] = (int)(img * p.xsN + iy * p.xsH + ix * p.xsW + (((lane & 7) ^ ((hx >> 1) & 7)) * 8));
    }
  }
  const int npieces = (g.hrows + 7) >> 3;                         // scalar
  auto stageA1 = [&](int i, int buf, int c0) {                    // piece i of this wave -> halo buffer `buf`, channels c0..c0+63
    if (wave + i * NW < npieces) {
      const char* src = aoff[i] >= 0 ? reinterpret_cast<const char*>(xb + aoff[i] + c0) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * I4_A_BYTES + (wave + i * NW) * 1024), 16, 0, 0);
    }
  };
  // ---- weight staging: piece = 8 cout rows x 128 B; source = (scalar base of the step) + (a lane's constant 32-bit offset)
  unsigned woffL[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int row = (wave + i * NW) * 8 + (lane >> 3);
    woffL[i] = (unsigned)(((long long)row * Ktot + (((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2);
  }
  const char* wtile = p.w + (long long)n0 * Ktot * 2;
  auto stageB = [&](int slot, int tap, int c0) {
    const char* wstep = wtile + ((long long)tap * p.Cin + c0) * 2;          // scalar
#pragma unroll
    for (int i = 0; i < NPB; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wstep + woffL[i]), (lptr_t)(smem + W_OFF + slot * B_BYTES + (wave + i * NW) * 1024), 16, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // v_mfma_f32_16x16x32_bf16 fragments: lane l holds row / column l % 16 and the 16-byte K chunk l / 16 of a 32-deep half-step.
  // Pixel block j of the wave = tile pixels 64 * wm + 16 * j + (0..15); tile pixel m = (m / TW, m % TW) sits at halo row
  // (m / TW) * HW + m % TW (+ r * HW + s for tap (r, s)); slots past the tile read row 0 and are never stored.
  const int l15 = lane & 15, kq = lane >> 4;
  unsigned XA[2][4][3];                                           // [32-channel half][pixel block][tap column s]: byte address in halo buffer 0 at tap row 0
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = wm * 64 + j * 16 + l15;
    int py = m / g.TW, px = m - py * g.TW;
    if (m >= g.npix) { py = 0; px = 0; }
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      XA[0][j][sx] = smem_l + (unsigned)((py * HW + px + sx) * 128 + ((kq ^ (((px + sx) >> 1) & 7)) << 4));
      XA[1][j][sx] = XA[0][j][sx] ^ 64u;                          // second half: 16-byte chunk + 4 = address bit 6 flipped
    }
  }
  unsigned WA[2];
  WA[0] = smem_l + W_OFF + (wn * 64 + l15) * 128 + ((kq ^ ((l15 >> 1) & 7)) * 16);
  WA[1] = WA[0] ^ 64u;

  const int nch = p.Cin / 64;
  const int cbeg = blockIdx.z * g.cslices, cend = min(nch, cbeg + g.cslices);

  // ---- prologue: halo of the first slice, weights of steps 0 and 1, bias
  float* biasL = reinterpret_cast<float*>(smem + BIAS_OFF);
  if (tid < BN) biasL[tid] = p.bias ? p.bias[n0 + tid] : 0.f;     // (before the first DMA: its wait would drain the queue)
#pragma unroll
  for (int i = 0; i < NPA; ++i) stageA1(i, 0, cbeg * 64);
  stageB(0, 0, cbeg * 64);
  stageB(1, 1, cbeg * 64);
  i4_vmcnt0();
  __syncthreads();
  if (grp && !(ABL & 16)) i4_barrier();                           // group B runs one barrier behind group A from here on

  // Fragment reads run one phase ahead of their MFMAs, in two register sets.  Measured (tools/gpu/i4_ablate.py, dec2b): a load segment
  // that issues its own reads and waits for them is a serial chain of ~300 cycles beside a 256-cycle MFMA segment (29 of 94 us).
  // RDL = 0: the reads of phase p + 1 are issued BETWEEN the MFMAs of phase p and retired in the load segment of phase p + 1.
  // RDL = 1: they are issued in the load segment of phase p, which then waits (counted lgkmcnt(8): LDS reads return in order) only
  //          for the reads of phase p, issued a whole phase earlier; the compute segment is 16 bare MFMAs.
  bf16x8 fa[2][4], fb[2][4];
  // the eight fragment reads of phase (tap t, half h) from halo buffer `buf` into register set `set`: weights through immediates
  // (ring stage t % 3, 16-channel row blocks), pixels at XA + (buf, tap row) -- one v_add per read
  auto issue_read = [&](int set, int t, int h, int buf, int k) {
    if (ABL & 4) return;
    const int slot = t % 3;
    if (k < 4) {
      switch (slot * 4 + k) {
#define I4_RDW(S, K) case S * 4 + K: rd128o<S * B_BYTES + K * 2048>(WA[h], fa[set][K]); break;
        I4_RDW(0, 0) I4_RDW(0, 1) I4_RDW(0, 2) I4_RDW(0, 3) I4_RDW(1, 0) I4_RDW(1, 1) I4_RDW(1, 2) I4_RDW(1, 3)
        I4_RDW(2, 0) I4_RDW(2, 1) I4_RDW(2, 2) I4_RDW(2, 3)
#undef I4_RDW
      }
    } else {
      const unsigned soff = (unsigned)(buf * I4_A_BYTES + (t / 3) * HW * 128);       // scalar
      rd128(XA[h][k - 4][t % 3] + soff, fb[set][k - 4]);
    }
  };
#pragma unroll
  for (int k = 0; k < 8; ++k) issue_read(0, 0, 0, 0, k);

  int ab = 0;
  for (int c = cbeg; c < cend; ++c) {
    const bool next_slice = c + 1 < cend;                         // scalar
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool more = h == 0 || t < 8 || next_slice;          // a phase follows this one
        // the phase after this one: (tn, hn) in halo buffer bn
        const int tn = h == 0 ? t : (t + 1) % 9, hn = h ^ 1;
        const int bn = (h == 1 && t == 8) ? ab ^ 1 : ab;
        // ---- load segment
        if (RDL && more) {
#pragma unroll
          for (int k = 0; k < 8; ++k) issue_read(hn, tn, hn, bn, k);
        }
        if (h == 0) {
          // weights of step s + 2 into the stage step s - 1 used (its reads were retired before that phase's first barrier)
          if (!(ABL & 1)) {
            if (t < 7) stageB((t + 2) % 3, t + 2, c * 64);
            else if (next_slice) stageB((t + 2) % 3, t - 7, (c + 1) * 64);
          }
        } else {
          if (!(ABL & 2) && t < NPA && next_slice) stageA1(t, ab ^ 1, (c + 1) * 64);
        }
        if (RDL && h == 0 && !(ABL & 32)) {
          // weights of step s + 1 (and everything older: the halo piece of the step before) have landed; only step s + 2's two
          // pieces may still be in flight.  In the load segment of the step's FIRST half: the first reads of step s + 1 are issued
          // in the load segment of its second half, behind a barrier every wave reaches after this wait.
          if (t < 7 || next_slice) i4_vmcnt2(); else i4_vmcnt0();
        }
        if (RDL && more) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); else i4_lgkm0();
#pragma unroll
        for (int i = 0; i < 4; ++i) touch8(fa[h][i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) touch8(fb[h][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 16)) i4_barrier();
        // ---- compute segment: 16 MFMAs (RDL = 0: the next phase's eight fragment reads between them)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (!(ABL & 8)) {
            const int i = k >> 1, j0 = (k & 1) * 2;
            acc[i][j0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[h][i], fb[h][j0], acc[i][j0], 0, 0, 0);
            acc[i][j0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[h][i], fb[h][j0 + 1], acc[i][j0 + 1], 0, 0, 0);
          }
          if (!RDL) {
            if (more) issue_read(hn, tn, hn, bn, k);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
        if (!RDL && h == 0 && !(ABL & 32)) {
          // (RDL = 0: before this phase's second barrier, which every reader of step s + 1 passes first)
          if (t < 7 || next_slice) i4_vmcnt2(); else i4_vmcnt0();
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 16)) i4_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    ab ^= 1;
  }
  if (!grp && !(ABL & 16)) i4_barrier();                          // group A meets group B's last barrier
  __syncthreads();                                                // every wave is done with the stages: the epilogue reuses them

  // ---- epilogue: tile row = tile pixel m; [pixel][channel] image through LDS, whole rows out with 16-byte stores (igemm.hip)
  if (p.partial) {
    // split over channel slices: the fp32 tile goes out as whole slab rows (BN * 4 contiguous bytes)
    constexpr int CPR4 = BN / 4;
    char* tile = smem;                                            // BM * BN * 4 = 128 KiB
    int* rowS = reinterpret_cast<int*>(smem + BM * BN * 4);
    if (tid < BM) {
      const int py = tid / g.TW, px = tid - py * g.TW;
      const int oy = y0 + py, ox = x0 + px;
      rowS[tid] = (tid < g.npix && oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wm * 64 + j * 16 + l15;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cl = wn * 64 + i * 16 + 4 * kq;
        const int chunk = (cl >> 2) ^ (row & (CPR4 - 1));
        *reinterpret_cast<f32x4*>(tile + row * (BN * 4) + chunk * 16) = acc[i][j];
      }
    }
    __syncthreads();
    constexpr int NCH4 = BM * CPR4 / (NW * 64);
    float* slab = p.partial + (long long)blockIdx.z * p.M * p.N + n0;
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
  char* tile = smem;                                              // BM * BN * 2 = 64 KiB: the halo buffers
  int* rowY = reinterpret_cast<int*>(smem + W_OFF);               // the weight ring is free too
  int* rowM = rowY + BM;
  if (tid < BM) {
    const int py = tid / g.TW, px = tid - py * g.TW;
    const int oy = y0 + py, ox = x0 + px;
    int oy_ = -1, om_ = -1;
    if (tid < g.npix && oy < p.Ho && ox < p.Wo) {
      oy_ = (int)(img * p.ysN + oy * p.ysH + ox * p.ysW);
      om_ = (int)(img * p.msN + oy * p.msH + ox * p.msW);
    }
    rowY[tid] = oy_; rowM[tid] = om_;
  }
  {
    // accumulator (i, j): channels wn * 64 + 16 * i + 4 * kq + {0..3} of tile pixel wm * 64 + 16 * j + l15
    f32x4 bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4*>(biasL + wn * 64 + i * 16 + 4 * kq);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wm * 64 + j * 16 + l15;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cl = wn * 64 + i * 16 + 4 * kq;
        float v[4] = {acc[i][j][0] + bv[i][0], acc[i][j][1] + bv[i][1], acc[i][j][2] + bv[i][2], acc[i][j][3] + bv[i][3]};
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
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
  staged_rows_out<BM, BN, NW>(p, tile, rowY, rowM, n0, tid);
}

// Tile geometry for an Ho x Wo output: the TH x TW rectangle (TH * TW <= 256, halo (TH + 2) * (TW + 2) <= 360 rows) that wastes the
// fewest of the tiles' 256 pixel slots; widths that keep a 16-pixel fragment block inside one tile row are preferred (their LDS
// reads are conflict-free; a block that straddles two rows is 2-way on some lanes).
static bool i4_geometry(int Ho, int Wo, I4Geom& g, double& fill) {
  double best = 0.0;
  int bth = 0, btw = 0;
  for (int tw = 8; tw <= 254 && tw <= ((Wo + 7) & ~7); ++tw) {
    int th = 256 / tw;
    while (th > 1 && (th + 2) * (tw + 2) > I4_APIECES * 8) --th;
    if ((th + 2) * (tw + 2) > I4_APIECES * 8) continue;
    if (th > Ho) th = Ho;
    const int tx = (Wo + tw - 1) / tw, ty = (Ho + th - 1) / th;
    double f = (double)Ho * Wo / ((double)tx * ty * 256.0);
    if (tw % 16) f *= 0.96;
    if (f > best + 1e-9) { best = f; bth = th; btw = tw; }
  }
  if (!btw) return false;
  g.TH = bth; g.TW = btw; g.HW = btw + 2;
  g.hrows = (bth + 2) * (btw + 2); g.npix = bth * btw;
  g.tiles_x = (Wo + btw - 1) / btw; g.tiles_y = (Ho + bth - 1) / bth;
  fill = (double)Ho * Wo / ((double)g.tiles_x * g.tiles_y * 256.0);
  return true;
}

}  // namespace

int g_tune_igemm4 = 1;             // dct_tune_set(DCT_TUNE_IGEMM4, 0): 3x3 stride-1 layers stay on the igemm.hip tiles
int g_tune_igemm4_fill = 70;       // percent: least fill of the 256-pixel tiles
int g_tune_igemm4_min_blocks = 96; // fewest blocks (before a split over channel slices) for which the kernel is taken
int g_tune_igemm4_ablate = 0;      // diagnostic builds (-DDCT_I4_ABLATE): ablation variant, see igemm4_kernel
int g_tune_igemm4_split_below = 200;  // layers with fewer blocks than this are split over channel slices (fp32 slabs)

// Plan of the ping-pong kernel for one layer (shared with dct_conv2d_workspace_bytes): use = 0 when the layer stays on igemm.hip.
struct I4Plan { int use; I4Geom g; int splits; };
static I4Plan i4_plan(int images, int Ho, int Wo, int Cin, int N) {
  I4Plan pl; pl.use = 0; pl.splits = 1;
  if (!g_tune_igemm4 || Cin % 64 || N % 128) return pl;
  double fill;
  if (!i4_geometry(Ho, Wo, pl.g, fill) || fill * 100.0 < g_tune_igemm4_fill) return pl;
  const long long blocks = (long long)images * pl.g.tiles_x * pl.g.tiles_y * (N / 128);
  if (blocks < g_tune_igemm4_min_blocks) return pl;
  const int nch = Cin / 64;
  int splits = 1;
  if (blocks < g_tune_igemm4_split_below) {
    splits = (int)((256 + blocks - 1) / blocks);
    while (splits > 1 && nch / splits < 2) --splits;
  }
  pl.g.cslices = (nch + splits - 1) / splits;
  pl.splits = (nch + pl.g.cslices - 1) / pl.g.cslices;
  pl.use = 1;
  return pl;
}

size_t dct_igemm4_workspace(int images, int Ho, int Wo, int Cin, int N) {
  const I4Plan pl = i4_plan(images, Ho, Wo, Cin, N);
  return (pl.use && pl.splits > 1) ? (size_t)pl.splits * images * Ho * Wo * N * sizeof(float) : 0;
}

// Launch for a layer dct_conv2d has vetted (bf16, 3x3 stride 1, staged-epilogue alignment, 32-bit offsets): returns 0 when the layer is
// not taken, 1 when launched with the epilogue applied, 2 when launched into fp32 slabs (`partial`; the caller folds them).
int dct_igemm4_launch(const void* params, int images, void* workspace, size_t workspace_bytes, hipStream_t st) {
  IgemmParams p = *reinterpret_cast<const IgemmParams*>(params);
  const I4Plan pl = i4_plan(images, p.Ho, p.Wo, p.Cin, p.N);
  if (!pl.use) return 0;
  if (pl.splits > 1) {
    const size_t need = (size_t)pl.splits * p.M * p.N * sizeof(float);
    if (!workspace || workspace_bytes < need) return 0;
    p.partial = (float*)workspace;
  } else p.partial = nullptr;
  constexpr size_t lds = 2 * (size_t)I4_A_BYTES + I4_NB * (size_t)128 * 128 + 128 * 4;
  static_assert(lds <= 160 * 1024 && 256 * 128 * 4 + 256 * 4 <= lds, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm4_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  const dim3 grid((unsigned)(images * pl.g.tiles_y * pl.g.tiles_x), p.N / 128, pl.splits);
#ifdef DCT_I4_ABLATE
  if (g_tune_igemm4_ablate) {
#define I4_ABL_CASE(V) case V: { static bool a_##V = false; if (!a_##V) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm4_kernel<128, V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a_##V = true; } \
      DCT_LAUNCH(DCT_PROF_IGEMM, (igemm4_kernel<128, V>), grid, dim3(512), lds, st, p, pl.g); return pl.splits > 1 ? 2 : 1; }
    switch (g_tune_igemm4_ablate) {
      case 100: { static bool a_r = false; if (!a_r) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm4_kernel<128, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a_r = true; }
        DCT_LAUNCH(DCT_PROF_IGEMM, (igemm4_kernel<128, 0, 0>), grid, dim3(512), lds, st, p, pl.g); return pl.splits > 1 ? 2 : 1; }
      I4_ABL_CASE(1) I4_ABL_CASE(2) I4_ABL_CASE(3) I4_ABL_CASE(4) I4_ABL_CASE(8) I4_ABL_CASE(12) I4_ABL_CASE(7) I4_ABL_CASE(16) I4_ABL_CASE(32) I4_ABL_CASE(35) I4_ABL_CASE(39) I4_ABL_CASE(47) I4_ABL_CASE(63)
      default: break;
    }
  }
#endif
  DCT_LAUNCH(DCT_PROF_IGEMM, (igemm4_kernel<128>), grid, dim3(512), lds, st, p, pl.g);
  return pl.splits > 1 ? 2 : 1;
}
