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
  int cslices;             // 64-channel slices per tile (a layer with few tiles is split over channel slices: fp32 slabs)
  int npatch, ntn, ntiles; // pixel rectangles (images * tiles_y * tiles_x), channel tiles, tiles in all (npatch * ntn * splits)
};

constexpr int I4_APIECES = 45;                       // halo capacity: 360 rows of 128 B
constexpr int I4_A_BYTES = I4_APIECES * 1024;
constexpr int I4_NB = 3;                             // weight ring stages
constexpr int I4_TAIL_BYTES = 76 * 256;              // epilogue rows that do not fit the freed halo buffer (BN = 128: rows 180..255)

// staged_rows_out (igemm_common.h) for a tile image split over two LDS regions: rows < ROWS0 at tile0, the rest at tile1
// (both addressed as base + row * BN * 2).
template <int BM, int BN, int NW, int ROWS0>
__device__ __forceinline__ void i4_rows_out(const IgemmParams& p, const char* tile0, const char* tile1, const int* rowY, const int* rowM, int n0, int tid) {
  constexpr int CPR = BN / 8;
  constexpr int NBATCH = 4;
  constexpr int NCH = BM * CPR / (NW * 64) / NBATCH;              // batches of chunks: few registers held across the loads
#pragma unroll 1
  for (int batch = 0; batch < NBATCH; ++batch) {
  int yo[NCH];
  bf16x8 mk[NCH], old[NCH];
  unsigned mb[NCH];
#pragma unroll
  for (int t = 0; t < NCH; ++t) {
    const int id = (batch * NCH + t) * (NW * 64) + tid;
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
    const int id = (batch * NCH + t) * (NW * 64) + tid;
    const int row = id / CPR, cc = id % CPR;
    if (yo[t] < 0) continue;
    bf16x8 v = *reinterpret_cast<const bf16x8*>((row < ROWS0 ? tile0 : tile1) + row * (BN * 2) + ((cc ^ (row & (CPR - 1))) * 16));
    const int co = n0 + cc * 8;
    if (p.mask_bits) chunk_gate_bits(v, mb[t], p.mask_scale);
    else if (p.mask && co < p.mask_channels) chunk_gate_act(v, mk[t], p.mask_scale);
    if (p.accumulate) {
      asm volatile("" ::: "memory");
      chunk_add(v, old[t]);
    }
    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.y) + yo[t] + co) = v;
    if (p.bits_out) p.bits_out[(unsigned)(yo[t] + co) >> 3] = (unsigned char)relu_bits8(v);
  }
  }
}

// acc += a x b, accumulating IN PLACE (the builtin lets the register allocator give D another register than C; over 18 unrolled
// phases that cost ~60 extra registers and spills).  An MFMA chain on one accumulator needs no wait states; the fragments come from
// LDS reads behind an s_waitcnt; the first reader of acc after the loop sits behind barriers and waits (>= 12 states).
__device__ __forceinline__ void i4_mfma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int N> __device__ __forceinline__ void i4_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void i4_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void i4_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void i4_barrier() { asm volatile("s_barrier" ::: "memory"); }

// ABL (diagnostic builds only, make EXTRA=-DDCT_I4_ABLATE): 0 = the kernel; bit 0: no weight DMA in the loop, bit 1: no halo DMA in the
// loop, bit 2: no fragment reads, bit 3: no MFMAs, bit 4: no barriers in the loop, bit 5: no vmcnt wait in the loop (timing studies; results are wrong)
//
// PERSISTENT blocks.  Measured on the first (one tile per block) form, tools/gpu/i4_ablate.py + profiles/r04_i4_*: at one block per CU
// nothing overlaps a block's prologue (73 KiB of DMA before the first MFMA) and epilogue (64 KiB of stores): 32 of dec2b's 80 us,
// 31 of dec2a's 55.  So a block walks the tile list (tile T, T + gridDim.x, ...):
//   * the next tile is just "the next slice" of the DMA stream: its first halo goes into the free halo buffer during taps 0..5 of the
//     current tile's last slice, its first two weight stages into the ring at taps 7 and 8;
//   * the epilogue stages the bf16 tile in the halo buffer the tile has just finished with (+ a 19 KiB tail), issues the row stores
//     and goes straight on: the stores drain under the next tile's MFMAs (they are older than the weight pieces of its step 2 in the
//     in-order vmcnt queue, so the wait of step 1 -- 1.5 steps later -- is the first that can see them).
// What that bought, and what it did not (profiles/r04_i4_*): per tile of dec2b a block spends 27 k cycles in the K loop (18.4 k of MFMA
// issue), 3.8 k turning accumulators into the staged tile and 6.2 k ISSUING the row stores -- 64 KiB at the ~10 B/clk/CU every CU gets
// while all 256 store at once -- at a 1.65 GHz clock.  The stores' drain overlaps the next tile, their issue cannot: the eight waves
// that own the MFMAs are the ones stuck in the store queue, and one block per CU leaves nobody else to feed the matrix pipes.  Two
// variants measured and removed: only the four "group B" waves issuing all DMA and only "group A" storing (so that no wave waits for
// its own stores behind the in-order vmcnt counter): dec2b 84 -> 95 us; odd blocks started half a tile late (de-synchronised bursts):
// the late blocks run 7 % faster per tile and the launch takes 3 us longer.  The tiles of igemm.hip hide the same work behind the
// second block of the CU, which is why the whole cfg2 step is level between the two families (tools/ab_step.py --knob 35).
template <int BN, int ABL = 0>
__global__ __launch_bounds__(512) void igemm4_kernel(IgemmParams p, I4Geom g) {
  constexpr int NW = 8, NWM = 4, NWN = BN / 64;
  static_assert(NWM * NWN == NW, "eight waves of 64 pixels x 64 channels");
  constexpr int BM = 256;
  constexpr int B_BYTES = BN * 128, NPB = BN / 8 / NW;          // weight stage; pieces per wave and stage (2)
  constexpr int NPA = (I4_APIECES + NW - 1) / NW;              // halo pieces per wave and slice (6)
  constexpr int W_OFF = 2 * I4_A_BYTES, TAIL_OFF = W_OFF + I4_NB * B_BYTES, ROWT_OFF = TAIL_OFF + I4_TAIL_BYTES;
  constexpr int ROWS0 = I4_A_BYTES / (BN * 2);                  // epilogue tile rows that fit the freed halo buffer (180)
  static_assert((BM - ROWS0) * BN * 2 <= I4_TAIL_BYTES, "epilogue tail");
  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                     // SIMD partners are waves w and w + 4: one of each group per SIMD
  const int wm = wave & 3, wn = (NWN == 2) ? grp : 0;
  const int HW = g.HW;
  const long long Ktot = 9ll * p.Cin;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
  const char* zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 7) * 16;
  const unsigned smem_l = (unsigned)(size_t)(lptr_c)(smem);
  const int nch = p.Cin / 64;
  const bool overlap = p.partial == nullptr;                     // fp32 slabs need the whole LDS for their tile: no prefetch across tiles

  // ---- tile list: T -> (channel tile fastest, patch, channel-slice split)
  int img, y0, x0, n0, cbeg, cend, zsplit;
  auto decode = [&](int T, int& img_, int& y0_, int& x0_, int& n0_, int& cb_, int& ce_, int& z_) {
    const int nt = T % g.ntn; T /= g.ntn;
    int pt = T % g.npatch; z_ = T / g.npatch;
    const int tx = pt % g.tiles_x; pt /= g.tiles_x;
    const int ty = pt % g.tiles_y; img_ = pt / g.tiles_y;
    y0_ = ty * g.TH; x0_ = tx * g.TW; n0_ = nt * BN;
    cb_ = z_ * g.cslices; ce_ = min(nch, cb_ + g.cslices);
  };

  // ---- halo staging: wave w owns pieces w, w + 8, ...; lane -> (halo row = piece * 8 + lane / 8, swizzled source chunk).
  // 16-byte chunk c of halo pixel (hy, hx) sits at chunk c ^ ((hx >> 1) & 7) of LDS row hy * HW + hx: the swizzle depends on the
  // COLUMN only, so a lane's fragment address for tap (r, s) is (a per-lane constant for s) + r * HW * 128 -- no per-tap vector
  // arithmetic (the two waves of a SIMD share its vector issue: ~25 address instructions per load segment beside the partner's
  // MFMAs cost more than the reads themselves, tools/gpu/i4_ablate.py).  A 16-lane read group still covers 16 distinct slots.
  // DMA ownership: wave w owns pieces w, w + 8, ... of every stage (6 halo slots, 2 weight pieces)
  constexpr int OWN = NW, NPH = NPA;
  const int dq = wave;
  int hyx[NPH];                                                   // this lane's halo pixel per piece, (hy << 16) | hx (tile-independent); rows past the halo: hy = 0x4000
#pragma unroll
  for (int i = 0; i < NPH; ++i) {
    const int row = (dq + i * OWN) * 8 + (lane >> 3);
    const int hy = row / HW;
    hyx[i] = ((row < g.hrows ? hy : 0x4000) << 16) | (row - hy * HW);
  }
  int aoff[NPH];
  auto halo_offsets = [&](int img_, int y0_, int x0_) {           // element offsets of this lane's halo pixels in tile (img_, y0_, x0_); -1: zero page
#pragma unroll
    for (int i = 0; i < NPH; ++i) {
      const int hx = hyx[i] & 0xffff;
      const int iy = y0_ - p.pad_h + (hyx[i] >> 16), ix = x0_ - p.pad_w + hx;
      aoff[i] = ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                    ? (int)(img_ * p.xsN + iy * p.xsH + ix * p.xsW + (((lane & 7) ^ ((hx >> 1) & 7)) * 8)) : -1;
    }
  };
  const int npieces = (g.hrows + 7) >> 3;                         // scalar
  auto stageA1 = [&](int i, int buf, int c0) {                    // piece slot i of this (group B) wave -> halo buffer `buf`, channels c0..c0+63
    if (dq + i * OWN < npieces) {
      const char* src = aoff[i] >= 0 ? reinterpret_cast<const char*>(xb + aoff[i] + c0) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * I4_A_BYTES + (dq + i * OWN) * 1024), 16, 0, 0);
    }
  };
  // ---- weight staging: piece = 8 cout rows x 128 B; source = (scalar base of the step) + (a lane's constant 32-bit offset)
  // (group B wave 4 + q: pieces q, q + 4, q + 8, q + 12 -- 32 rows apart, the same swizzle: one lane offset, the piece in the scalar base)
  constexpr int NPW = NPB;
  unsigned woffL;
  {
    const int row = dq * 8 + (lane >> 3);
    woffL = (unsigned)(((long long)row * Ktot + (((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2);
  }
  auto stageB = [&](int slot, const char* wt, int tap, int c0) {
    const char* wstep = wt + ((long long)tap * p.Cin + c0) * 2;             // scalar
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wstep + (long long)i * (OWN * 8) * Ktot * 2 + woffL), (lptr_t)(smem + W_OFF + slot * B_BYTES + (dq + i * OWN) * 1024), 16, 0, 0);
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
  unsigned XA[4][3];                                              // [pixel block][tap column s]: byte address in halo buffer 0 at tap row 0, first 32-channel half
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = wm * 64 + j * 16 + l15;
    int py = m / g.TW, px = m - py * g.TW;
    if (m >= g.npix) { py = 0; px = 0; }
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      XA[j][sx] = smem_l + (unsigned)((py * HW + px + sx) * 128 + ((kq ^ (((px + sx) >> 1) & 7)) << 4));
    }
  }
  unsigned WA[2];
  WA[0] = smem_l + W_OFF + (wn * 64 + l15) * 128 + ((kq ^ ((l15 >> 1) & 7)) * 16);
  WA[1] = WA[0] ^ 64u;

  // Fragment reads run one phase ahead of their MFMAs, in two register sets: the reads of phase p + 1 are issued in the load
  // segment of phase p, which then waits (counted lgkmcnt(8): LDS reads return in order) only for the reads of phase p, issued a
  // whole phase earlier; the compute segment is 16 bare MFMAs.  (A load segment that issues its own reads and waits for them is a
  // serial chain of ~300 cycles beside a 256-cycle MFMA segment: 29 of 94 us on dec2b.)
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
      unsigned soff = (unsigned)(buf * I4_A_BYTES + (t / 3) * HW * 128);             // scalar
      asm volatile("" : "+s"(soff));                              // opaque: keeps the 12 x 3 x 2 x 2 sums XA + soff from being hoisted into registers
      // (second half: 16-byte chunk + 4 = address bit 6 flipped; (a ^ 64) + soff is one v_xad_u32)
      rd128(h ? (XA[k - 4][t % 3] ^ 64u) + soff : XA[k - 4][t % 3] + soff, fb[set][k - 4]);
    }
  };

  // ---- first tile: halo of its first slice, weights of steps 0 and 1
  int T = blockIdx.x;
  decode(T, img, y0, x0, n0, cbeg, cend, zsplit);
  const char* wtile = p.w + (long long)n0 * Ktot * 2;
  halo_offsets(img, y0, x0);
#pragma unroll
  for (int i = 0; i < NPH; ++i) stageA1(i, 0, cbeg * 64);
  stageB(0, wtile, 0, cbeg * 64);
  stageB(1, wtile, 1, cbeg * 64);
  i4_vmcnt0();
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) issue_read(0, 0, 0, 0, k);
  if (grp && !(ABL & 16)) i4_barrier();                           // group B runs one barrier behind group A from here on

#ifdef DCT_I4_ABLATE
  // diagnostic: per-wave cycle sums of the tile loop / the epilogue up to the staged tile / the row stores / tile set-up
  unsigned long long st_loop = 0, st_ep1 = 0, st_ep2 = 0, st_setup = 0, st_tiles = 0, st_mark = __builtin_amdgcn_s_memtime();
  const unsigned long long st_begin = st_mark;
#define I4_STAMP(acc_) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - st_mark; st_mark = now_; } while (0)
#else
#define I4_STAMP(acc_) do { } while (0)
#endif
  int ab = 0;
  bool fresh = true;                                              // the tile's step-0 and step-1 weights are known to have landed
  for (;;) {
    const int Tn = T + (int)gridDim.x;
    const bool tile_follows = Tn < g.ntiles;
    int img_n = 0, y0_n = 0, x0_n = 0, n0_n = 0, cbeg_n = 0, cend_n = 0, z_n = 0;
    if (tile_follows) decode(Tn, img_n, y0_n, x0_n, n0_n, cbeg_n, cend_n, z_n);
    const char* wtile_n = p.w + (long long)n0_n * Ktot * 2;

    for (int c = cbeg; c < cend; ++c) {
      const bool last = c + 1 == cend;                            // scalar
      // the slice after this one in the DMA stream: the next slice of the tile, or the first slice of the block's next tile
      const bool nxt = !last || (tile_follows && overlap);
      const char* nxt_w = last ? wtile_n : wtile;
      const int nxt_c0 = last ? cbeg_n * 64 : (c + 1) * 64;
      if (last && nxt) halo_offsets(img_n, y0_n, x0_n);           // (this tile's halo pieces have all been issued)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool more = h == 0 || t < 8 || !last;             // a phase of this tile follows
          // the phase after this one: (tn, hn) in halo buffer bn
          const int tn = h == 0 ? t : (t + 1) % 9, hn = h ^ 1;
          const int bn = (h == 1 && t == 8) ? ab ^ 1 : ab;
          // ---- load segment
          if (more) {
#pragma unroll
            for (int k = 0; k < 8; ++k) issue_read(hn, tn, hn, bn, k);
          }
          {
            if (h == 0) {
              // weights of step s + 2 into the stage step s - 1 used (its reads were retired before that phase's first barrier)
              if (!(ABL & 1)) {
                if (t < 7) stageB((t + 2) % 3, wtile, t + 2, c * 64);
                else if (nxt) stageB((t + 2) % 3, nxt_w, t - 7, nxt_c0);
              }
              // weights of step s + 1 (and everything older: the halo pieces of the step before) have landed; only step s + 2's
              // four pieces may still be in flight.  In the load segment of the step's FIRST half: the first reads of step s + 1
              // are issued in the load segment of its second half, behind a barrier every wave reaches after this wait.
              // (Step 0 of a tile: steps 0 and 1 were drained before the epilogue / in the prologue.)
              if (!(ABL & 32) && !(t == 0 && fresh)) { if (t < 7 || nxt) i4_vmcnt<NPW>(); else i4_vmcnt0(); }
            } else {
              if (!(ABL & 2) && t < NPA && nxt) stageA1(t, ab ^ 1, nxt_c0);
            }
          }
          if (more) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); else i4_lgkm0();
#pragma unroll
          for (int i = 0; i < 4; ++i) touch8(fa[h][i]);
#pragma unroll
          for (int j = 0; j < 4; ++j) touch8(fb[h][j]);
          __builtin_amdgcn_sched_barrier(0);
          if (!(ABL & 16)) i4_barrier();
          // ---- compute segment: 16 MFMAs
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
          if (!(ABL & 8)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                i4_mfma(acc[i][j], fa[h][i], fb[h][j]);
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          if (!(ABL & 16)) i4_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
        fresh = false;
      }
      ab ^= 1;
    }
    if (!grp && !(ABL & 16)) i4_barrier();                        // group A meets group B's last barrier
    I4_STAMP(st_loop);

    // ---- epilogue of tile T: tile row = tile pixel m; [pixel][channel] image through LDS, whole rows out with 16-byte stores
    // (lane constants made opaque here: otherwise every epilogue address is hoisted out of the tile loop into registers and spilled)
    int tid_e = tid, l15_e = l15, kq_e = kq;
    asm volatile("" : "+v"(tid_e), "+v"(l15_e), "+v"(kq_e));
    if (p.partial) {
      __syncthreads();                                            // every wave is done with the stages: the fp32 tile takes them over
      // split over channel slices: the fp32 tile goes out as whole slab rows (BN * 4 contiguous bytes)
      constexpr int CPR4 = BN / 4;
      char* tile = smem;                                          // BM * BN * 4 = 128 KiB
      int* rowS = reinterpret_cast<int*>(smem + BM * BN * 4);
      if (tid_e < BM) {
        const int py = tid_e / g.TW, px = tid_e - py * g.TW;
        const int oy = y0 + py, ox = x0 + px;
        rowS[tid_e] = (tid_e < g.npix && oy < p.Ho && ox < p.Wo) ? (img * p.Ho + oy) * p.Wo + ox : -1;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wm * 64 + j * 16 + l15_e;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cl = wn * 64 + i * 16 + 4 * kq_e;
          const int chunk = (cl >> 2) ^ (row & (CPR4 - 1));
          *reinterpret_cast<f32x4*>(tile + row * (BN * 4) + chunk * 16) = acc[i][j];
        }
      }
      __syncthreads();
      constexpr int NCH4 = BM * CPR4 / (NW * 64);
      float* slab = p.partial + (long long)zsplit * p.M * p.N + n0;
#pragma unroll
      for (int t = 0; t < NCH4; ++t) {
        const int id = t * (NW * 64) + tid_e;
        const int row = id / CPR4, cc = id % CPR4;
        const int mg = rowS[row];
        if (mg < 0) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * (BN * 4) + ((cc ^ (row & (CPR4 - 1))) * 16));
        *reinterpret_cast<f32x4*>(slab + (long long)mg * p.N + cc * 4) = v;
      }
    } else {
      constexpr int CPR = BN / 8;
      // bias of this lane's channels straight from memory (L2-resident; the loads fly during the drain below)
      f32x4 bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        bv[i] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 64 + i * 16 + 4 * kq_e) : f32x4{0.f, 0.f, 0.f, 0.f};
      i4_vmcnt0();                                                // the next tile's first halo and weight stages have landed (issued >= 1 step ago)
      __syncthreads();                                            // ... for every reader; and every wave is done with this tile's last halo buffer
      // staging: rows 0 .. ROWS0 - 1 in the halo buffer this tile used last (ab was flipped: that is ab ^ 1), the rest in the tail
      char* tile0 = smem + (ab ^ 1) * I4_A_BYTES;
      char* tile1 = smem + TAIL_OFF - ROWS0 * (BN * 2);
      int* rowY = reinterpret_cast<int*>(smem + ROWT_OFF);
      int* rowM = rowY + BM;
      if (tid_e < BM) {
        const int py = tid_e / g.TW, px = tid_e - py * g.TW;
        const int oy = y0 + py, ox = x0 + px;
        int oy_ = -1, om_ = -1;
        if (tid_e < g.npix && oy < p.Ho && ox < p.Wo) {
          oy_ = (int)(img * p.ysN + oy * p.ysH + ox * p.ysW);
          om_ = (int)(img * p.msN + oy * p.msH + ox * p.msW);
        }
        rowY[tid_e] = oy_; rowM[tid_e] = om_;
      }
      // accumulator (i, j): channels wn * 64 + 16 * i + 4 * kq_e + {0..3} of tile pixel wm * 64 + 16 * j + l15_e
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wm * 64 + j * 16 + l15_e;
        char* trow = (row < ROWS0 ? tile0 : tile1) + row * (BN * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cl = wn * 64 + i * 16 + 4 * kq_e;
          float v[4] = {acc[i][j][0] + bv[i][0], acc[i][j][1] + bv[i][1], acc[i][j][2] + bv[i][2], acc[i][j][3] + bv[i][3]};
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = relu1(v[e]);
          }
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
          const int chunk = (cl >> 3) ^ (row & (CPR - 1));
          *reinterpret_cast<bf16x4*>(trow + chunk * 16 + (cl & 4) * 2) = o;
        }
      }
      __syncthreads();
      I4_STAMP(st_ep1);
      // the rows go out and the block goes on without waiting for them
      i4_rows_out<BM, BN, NW, ROWS0>(p, tile0, tile1, rowY, rowM, n0, tid_e);
    }
    I4_STAMP(st_ep2);
#ifdef DCT_I4_ABLATE
    ++st_tiles;
#endif
    if (!tile_follows) break;

    // ---- on to the block's next tile
    T = Tn; img = img_n; y0 = y0_n; x0 = x0_n; n0 = n0_n; cbeg = cbeg_n; cend = cend_n; zsplit = z_n; wtile = wtile_n;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!overlap) {                                               // nothing was prefetched: a prologue as for the first tile
      __syncthreads();                                            // (every wave has read its part of the fp32 tile)
      ab = 0;
      halo_offsets(img, y0, x0);
#pragma unroll
      for (int i = 0; i < NPH; ++i) stageA1(i, 0, cbeg * 64);
      stageB(0, wtile, 0, cbeg * 64);
      stageB(1, wtile, 1, cbeg * 64);
      i4_vmcnt0();
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) issue_read(0, 0, 0, ab, k);
    if (grp && !(ABL & 16)) i4_barrier();                         // group B drops one barrier behind again
    fresh = true;
    I4_STAMP(st_setup);
  }
#ifdef DCT_I4_ABLATE
  if (p.stamps && lane == 0) {
    unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
    o[0] = st_loop; o[1] = st_ep1; o[2] = st_ep2; o[3] = st_setup; o[4] = st_tiles; o[5] = __builtin_amdgcn_s_memtime() - st_begin;
  }
#endif
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

int g_tune_igemm4 = 0;             // dct_tune_set(DCT_TUNE_IGEMM4, 0): 3x3 stride-1 layers stay on the igemm.hip tiles
static const int g_tune_igemm4_fill = 70;       // percent: least fill of the 256-pixel tiles
int g_tune_igemm4_min_blocks = 96; // fewest blocks (before a split over channel slices) for which the kernel is taken
int g_tune_igemm4_blocks = 0;      // persistent blocks per launch (0: one per CU); diagnostic knob 1001
unsigned long long* g_igemm4_stamps = nullptr;   // diagnostic builds: per-wave cycle sums (dct_debug_i4_stamps)
extern "C" int dct_debug_i4_stamps(void* buf) { g_igemm4_stamps = (unsigned long long*)buf; return 0; }
int g_tune_igemm4_ablate = 0;      // diagnostic builds (-DDCT_I4_ABLATE): ablation variant, see igemm4_kernel
static const int g_tune_igemm4_split_below = 200;  // layers with fewer blocks than this are split over channel slices (fp32 slabs)

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
  pl.g.npatch = images * pl.g.tiles_x * pl.g.tiles_y;
  pl.g.ntn = N / 128;
  pl.g.ntiles = pl.g.npatch * pl.g.ntn * pl.splits;
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
  p.stamps = g_igemm4_stamps;
  const I4Plan pl = i4_plan(images, p.Ho, p.Wo, p.Cin, p.N);
  if (!pl.use) return 0;
  if (pl.splits > 1) {
    const size_t need = (size_t)pl.splits * p.M * p.N * sizeof(float);
    if (!workspace || workspace_bytes < need) return 0;
    p.partial = (float*)workspace;
  } else p.partial = nullptr;
  constexpr size_t lds = 2 * (size_t)I4_A_BYTES + I4_NB * (size_t)128 * 128 + I4_TAIL_BYTES + 256 * 8;   // halo x 2, ring, epilogue tail, row tables
  static_assert(lds <= 160 * 1024 && 256 * 128 * 4 + 256 * 4 <= lds, "LDS budget");
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  // persistent blocks, one per CU (the LDS footprint admits no second one): block b walks tiles b, b + grid, ...
  const int nblocks = std::min(pl.g.ntiles, g_tune_igemm4_blocks > 0 ? g_tune_igemm4_blocks : cus);
  const dim3 grid((unsigned)nblocks, 1, 1);
#define I4_LAUNCH(ABLV) do { static bool a_ = false; if (!a_) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm4_kernel<128, ABLV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a_ = true; } \
      DCT_LAUNCH(DCT_PROF_IGEMM, (igemm4_kernel<128, ABLV>), grid, dim3(512), lds, st, p, pl.g); } while (0)
#ifdef DCT_I4_ABLATE
  switch (g_tune_igemm4_ablate) {
    case 3: I4_LAUNCH(3); return pl.splits > 1 ? 2 : 1;
    case 4: I4_LAUNCH(4); return pl.splits > 1 ? 2 : 1;
    case 8: I4_LAUNCH(8); return pl.splits > 1 ? 2 : 1;
    case 12: I4_LAUNCH(12); return pl.splits > 1 ? 2 : 1;
    case 63: I4_LAUNCH(63); return pl.splits > 1 ? 2 : 1;
    default: break;
  }
#endif
  I4_LAUNCH(0);
#undef I4_LAUNCH
  return pl.splits > 1 ? 2 : 1;
}
