// Role-split persistent implicit-GEMM kernel for 3x3 stride-1 convolutions on gfx950 (bf16), 64-channel tiles: forward pass and data
// gradient of the UNet 3x3 layers on large images (reference: generalframework/arch/network.py:153-171,196-240).
//
// igemm4.hip (one block per CU, eight MFMA waves in ping-pong pairs) measured what a conv kernel on this chip has to avoid
// (DESIGN.md 9): (1) the row stores of a tile cannot be ISSUED asynchronously by waves that own MFMA work -- 64 KiB per tile at the
// ~10 B/clk/CU every CU gets in a chip-wide burst stalls them 6 k cycles; (2) one in-order vmcnt counter per wave means a wave that
// has issued stores cannot wait for a younger DMA piece without waiting for the stores; (3) a loading wave's instructions crawl beside
// its SIMD partner's MFMAs.  Here every wave has ONE kind of memory traffic:
//   waves 0-3  MFMA waves, one per SIMD: a 64-pixel x 64-channel register tile each (v_mfma_f32_16x16x32_bf16, in place), fragment
//              reads of the next phase issued BETWEEN the MFMAs of this one, no vector-memory instruction at all in the loop;
//   waves 4-5  DMA waves: every global_load_lds of the block (a tap's 64 x 64 weight tile = 8 KiB per K-step through a ring of three
//              stages, the halo of the next (patch, channel slice) into the free halo buffer), counted vmcnt, nothing else in their queue;
//   waves 6-7  store waves: drain the PREVIOUS tile's staged bf16 image (a 32 KiB LDS buffer of its own: a 64-channel tile leaves the
//              room beside two 45 KiB halo buffers and the 24 KiB ring) two 16-byte row stores per lane and K-step -- paced, so the
//              store queue never backs up into a barrier -- with the mask / accumulate / gate-bit epilogue of igemm.hip.
// One s_barrier per K-step (all eight waves), placed between the two 32-channel halves; weight stages in a ring of four:
//   RAW  K-step s's DMA group (halo pieces of the next position, the weights of step s + 3) is waited for at step s + 2 (counted vmcnt: the
//        two youngest groups may be in flight) before barrier s + 2; the MFMA waves issue their first reads of stage s + 3 after it.
//   WAR  MFMA waves retire the reads of stage s (lgkmcnt(0)) before barrier s; DMA waves refill that slot (step s + 4) after barrier s.
// A block walks patches P, P + gridDim.x, ... and, per patch, ALL channel tiles: with one or two channel slices the halo stays
// resident across the channel tiles of a patch (it is loaded once), with more it streams per (tile, slice) as in igemm4.
// Accumulation order per output = igemm3m_kernel's (slice, tap, 32-channel half; one MFMA chain): bit-identical results.
#include <algorithm>
#include "igemm_common.h"

namespace {

struct I5Geom {
  int TH, TW, HW;          // tile rows / columns of output pixels (TH * TW <= 256), halo pitch TW + 2
  int hrows, npix;         // (TH + 2) * HW halo rows of 128 B (<= 360); TH * TW
  int tiles_x, tiles_y, npatch, ntn;     // patches per image, in all (images * tiles_y * tiles_x); channel tiles N / 64
};

constexpr int I5_APIECES = 45, I5_A_BYTES = I5_APIECES * 1024;    // halo capacity: 360 rows of 128 B
constexpr int I5_RING = 4;
constexpr int I5_W_BYTES = 64 * 128;                               // a tap's weight stage: 64 channel rows x 64 input channels
constexpr int I5_W_OFF = 2 * I5_A_BYTES, I5_STG_OFF = I5_W_OFF + I5_RING * I5_W_BYTES, I5_STG_BYTES = 256 * 128;
constexpr int I5_TAB_OFF = I5_STG_OFF + I5_STG_BYTES, I5_BIAS_OFF = I5_TAB_OFF + 256 * 8, I5_MAX_N = 512;
constexpr int I5_LDS = I5_BIAS_OFF + I5_MAX_N * 4;
static_assert(I5_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void i5_mfma(f32x4& acc, const bf16x8& a, const bf16x8& b) {      // in place (see igemm4.hip i4_mfma)
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void i5_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void i5_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int N> __device__ __forceinline__ void i5_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ABL (diagnostic builds, -DDCT_I4_ABLATE): bit 0 no DMA in the loop, bit 2 no fragment reads, bit 3 no MFMAs, bit 6 no row stores
template <int ABL = 0>
__global__ __launch_bounds__(512) void igemm5_kernel(IgemmParams p, I5Geom g) {
  extern __shared__ __attribute__((aligned(128))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave < 4 ? 0 : (wave < 6 ? 1 : 2);           // 0: MFMA, 1: DMA, 2: store
  const int HW = g.HW;
  const long long Ktot = 9ll * p.Cin;
  const int nch = p.Cin / 64;
  const unsigned smem_l = (unsigned)(size_t)(lptr_c)(smem);
  const int G = (int)gridDim.x;
  const int mp = (g.npatch - (int)blockIdx.x + G - 1) / G;        // patches of this block

  auto patch_of = [&](int pi, int& img, int& y0, int& x0) {
    int pt = (int)blockIdx.x + pi * G;
    const int tx = pt % g.tiles_x; pt /= g.tiles_x;
    const int ty = pt % g.tiles_y; img = pt / g.tiles_y;
    y0 = ty * g.TH; x0 = tx * g.TW;
  };
  // halo buffer of stream position (patch pi, channel tile nt, slice c; q = running position count)
  auto hbuf = [&](int pi, int c, int q) { return nch == 1 ? (pi & 1) : (nch == 2 ? c : (q & 1)); };

  // bias of ALL channels -> LDS once (the MFMA waves add it at every tile end and must not wait on global memory there)
  float* biasL = reinterpret_cast<float*>(smem + I5_BIAS_OFF);
  for (int i = tid; i < p.N; i += 512) biasL[i] = p.bias ? p.bias[i] : 0.f;

  if (role == 1) {
    // =============================================================== DMA waves
    const int dq = wave - 4;
    const bf16_t* xb = reinterpret_cast<const bf16_t*>(p.x);
    const char* zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 7) * 16;
    constexpr int NPH = 23;                                        // halo piece slots of a DMA wave: pieces dq, dq + 2, ... (both waves load piece 44)
    int hyx[NPH];                                                  // (hy << 16) | hx of this lane's halo row per slot; rows past the halo: hy = 0x4000
#pragma unroll
    for (int i = 0; i < NPH; ++i) {
      const int row = min(dq + 2 * i, I5_APIECES - 1) * 8 + (lane >> 3);
      const int hy = row / HW;
      hyx[i] = ((row < g.hrows ? hy : 0x4000) << 16) | (row - hy * HW);
    }
    int h_img = 0, h_y0 = 0, h_x0 = 0;                             // patch whose halo is being loaded
    // halo piece slot i -> buffer buf (16-byte chunk swizzled by pixel column).  EVERY slot is issued (rows past the halo copy the
    // zero page): the counted waits below rely on the number of loads per K-step
    auto stageH = [&](int i, int buf, int c0) {
      {
        const int hx = hyx[i] & 0xffff;
        const int iy = h_y0 - p.pad_h + (hyx[i] >> 16), ix = h_x0 - p.pad_w + hx;
        const char* src = zero;
        if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
          src = reinterpret_cast<const char*>(xb + (h_img * p.xsN + iy * p.xsH + ix * p.xsW + (((lane & 7) ^ ((hx >> 1) & 7)) * 8) + c0));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * I5_A_BYTES + min(dq + 2 * i, I5_APIECES - 1) * 1024), 16, 0, 0);
      }
    };
    unsigned woffL;
    {
      const int row = dq * 8 + (lane >> 3);
      woffL = (unsigned)(((long long)row * Ktot + (((lane & 7) ^ ((row >> 1) & 7)) * 8)) * 2);
    }
    auto stageW = [&](int slot, int nt, int tap, int c) {          // pieces dq, dq + 2, dq + 4, dq + 6: 16 rows apart, same swizzle
      const char* wstep = p.w + (((long long)nt * 64) * Ktot + (long long)tap * p.Cin + c * 64) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(wstep + (long long)i * 16 * Ktot * 2 + woffL),
                                         (lptr_t)(smem + I5_W_OFF + slot * I5_W_BYTES + (dq + 2 * i) * 1024), 16, 0, 0);
    };
    // prologue: halo of position 0, weights of steps 0, 1 and 2
    patch_of(0, h_img, h_y0, h_x0);
#pragma unroll
    for (int i = 0; i < NPH; ++i) stageH(i, hbuf(0, 0, 0), 0);
    stageW(0, 0, 0, 0);
    stageW(1, 0, 1, 0);
    stageW(2, 0, 2, 0);
    i5_vmcnt<0>();
    __syncthreads();
    // K-step s (position q, tap t; s = 9 q + t) issues the group G_s = { up to 4 halo pieces of position q + 1, the 4 weight pieces of step
    // s + 3 } and then waits until G_(s-2) has landed, i.e. until at most |G_(s-1)| + |G_s| loads are in flight (vmcnt retires in order):
    // weights get three barrier intervals from issue to first read, halo pieces (taps 0..5 only) at least three.
    int q = 0;
    for (int pi = 0; pi < mp; ++pi)
      for (int nt = 0; nt < g.ntn; ++nt) {
        for (int c = 0; c < nch; ++c, ++q) {
          // the stream position after this one, and whether its halo has to be loaded (resident across channel tiles when nch <= 2)
          int pi2 = pi, nt2 = nt, c2 = c + 1;
          if (c2 == nch) { c2 = 0; if (++nt2 == g.ntn) { nt2 = 0; ++pi2; } }
          const bool nxt = pi2 < mp;
          const bool load_h = nxt && (nch > 2 || nt2 == 0);
          const int hb2 = hbuf(pi2, c2, q + 1);
          if (load_h) patch_of(pi2, h_img, h_y0, h_x0);
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            constexpr int HQ[9] = {4, 4, 4, 4, 4, 3, 0, 0, 0};       // halo pieces per tap (sum NPH)
            constexpr int H0[9] = {0, 4, 8, 12, 16, 20, 23, 23, 23};
            if (!(ABL & 1) && load_h) {
#pragma unroll
              for (int e = 0; e < HQ[t]; ++e) stageH(H0[t] + e, hb2, c2 * 64);
            }
            bool issued = false;
            if (!(ABL & 1)) {
              if (t < 6) { stageW((q + t + 3) & 3, nt, t + 3, c); issued = true; }
              else if (nxt) { stageW((q + t + 3) & 3, nt2, t - 6, c2); issued = true; }
            }
            // in flight afterwards: G_s and G_(s-1).  (The halo pieces of G_(s-1) are counted only inside a position: at t = 0 the
            // previous position's last taps issued none.)
            if (!issued) i5_vmcnt<0>();
            else if (load_h) {
              switch (t) {
                case 0: i5_vmcnt<12>(); break;
                case 1: case 2: case 3: case 4: i5_vmcnt<16>(); break;
                case 5: i5_vmcnt<15>(); break;
                case 6: i5_vmcnt<11>(); break;
                default: i5_vmcnt<8>(); break;
              }
            } else i5_vmcnt<8>();
            i5_barrier();
          }
        }
        i5_barrier();                                              // tile end: the MFMA waves have staged the tile
      }
    return;
  }

  if (role == 2) {
    // =============================================================== store waves
    const int st = tid - 384;                                      // 0 .. 127
    int* rowY = reinterpret_cast<int*>(smem + I5_TAB_OFF);
    int* rowM = rowY + 256;
    const char* stg = smem + I5_STG_OFF;
    int img = 0, y0 = 0, x0 = 0, n0 = 0;
    // element offsets of the staged tile's 256 rows in y and in the mask, -1 outside.  Lane st reads rows 16 u + st / 8 (u = 0..15, see
    // store2); it writes rows 16 (2 (st % 8) + k) + st / 8: every row a wave reads was written by that wave (LDS is in order per wave)
    auto table = [&]() {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int m = 16 * (2 * (st & 7) + k) + (st >> 3);
        const int py = m / g.TW, px = m - py * g.TW;
        const int oy = y0 + py, ox = x0 + px;
        int oy_ = -1, om_ = -1;
        if (m < g.npix && oy < p.Ho && ox < p.Wo) {
          oy_ = (int)(img * p.ysN + oy * p.ysH + ox * p.ysW);
          om_ = (int)(img * p.msN + oy * p.msH + ox * p.msW);
        }
        rowY[m] = oy_; rowM[m] = om_;
      }
    };
    // chunk u (0..15) of this lane: tile row (u * 128 + st) / 8, 16-byte chunk (u * 128 + st) % 8 = st % 8
    auto store2 = [&](int u0) {                                    // two chunks: loads first, then the stores (one round trip)
      if (ABL & 64) return;
      int yo[2]; unsigned mb[2]; bf16x8 mk[2], old[2], v[2];
      const int cc = st & 7, co = n0 + cc * 8;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = ((u0 + k) * 128 + st) >> 3;
        yo[k] = rowY[row];
        v[k] = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((cc ^ (row & 7)) * 16));
        if (yo[k] >= 0) {
          if (p.mask_bits) mb[k] = p.mask_bits[(unsigned)(rowM[row] + co) >> 3];
          else if (p.mask && co < p.mask_channels) mk[k] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.mask) + rowM[row] + co);
          if (p.accumulate) old[k] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(p.y) + yo[k] + co);
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (yo[k] < 0) continue;
        if (p.mask_bits) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[k][e] = ((mb[k] >> e) & 1u) ? (bf16_t)((float)v[k][e] * p.mask_scale) : (bf16_t)0.f;
        } else if (p.mask && co < p.mask_channels) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[k][e] = (float)mk[k][e] > 0.f ? (bf16_t)((float)v[k][e] * p.mask_scale) : (bf16_t)0.f;
        }
        if (p.accumulate) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[k][e] = (bf16_t)((float)v[k][e] + (float)old[k][e]);
        }
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.y) + yo[k] + co) = v[k];
        if (p.bits_out) p.bits_out[(unsigned)(yo[k] + co) >> 3] = (unsigned char)relu_bits8(v[k]);
      }
    };
    __syncthreads();                                               // (prologue barrier of the block)
    bool pending = false;                                          // a staged tile is waiting to be drained
    for (int pi = 0; pi < mp; ++pi)
      for (int nt = 0; nt < g.ntn; ++nt) {
        for (int c = 0; c < nch; ++c) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            // the previous tile drains during this tile's FIRST slice: row tables at tap 0, sixteen chunks per lane at taps 1..8
            if (pending && c == 0) {
              if (t == 0) table(); else store2(2 * (t - 1));
            }
            i5_barrier();
          }
        }
        // (every chunk of the previous tile has been read: the MFMA waves may overwrite the staging buffer now)
        i5_barrier();                                              // tile end: this tile is staged
        patch_of(pi, img, y0, x0);
        n0 = nt * 64;
        pending = true;
      }
    if (pending) {                                                 // the block's last tile: nobody is left to meet at a barrier, and none is needed
      table();                                                     // (a lane reads only table rows written by its own wave, see table())
#pragma unroll 1
      for (int u = 0; u < 16; u += 2) store2(u);
    }
    return;
  }

  // ================================================================= MFMA waves
  const int wm = wave;                                             // pixels 64 * wm .. + 63 of the tile, all 64 channels
  const int l15 = lane & 15, kq = lane >> 4;
  unsigned XA[4][3];                                               // [pixel block][tap column]: byte address in halo buffer 0 at tap row 0, first half
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = wm * 64 + j * 16 + l15;
    int py = m / g.TW, px = m - py * g.TW;
    if (m >= g.npix) { py = 0; px = 0; }
#pragma unroll
    for (int sx = 0; sx < 3; ++sx)
      XA[j][sx] = smem_l + (unsigned)((py * HW + px + sx) * 128 + ((kq ^ (((px + sx) >> 1) & 7)) << 4));
  }
  unsigned WA[2];
  WA[0] = smem_l + I5_W_OFF + l15 * 128 + ((kq ^ ((l15 >> 1) & 7)) * 16);
  WA[1] = WA[0] ^ 64u;
  bf16x8 fa[2][4], fb[2][4];
  // read r (0..7) of phase (tap t, half h) of stream position q: fragments in the order the MFMAs need them -- weights rows 0..15,
  // the four pixel blocks, weights rows 16..63
  auto issue_read = [&](int set, int q, int t, int h, int buf, int r) {
    if (ABL & 4) return;
    if (r == 0 || r >= 5) {
      unsigned wb = WA[h] + (unsigned)(((q + t) & 3) * I5_W_BYTES);
      switch (r) {
        case 0: rd128o<0>(wb, fa[set][0]); break;
        case 5: rd128o<2048>(wb, fa[set][1]); break;
        case 6: rd128o<4096>(wb, fa[set][2]); break;
        default: rd128o<6144>(wb, fa[set][3]); break;
      }
    } else {
      unsigned soff = (unsigned)(buf * I5_A_BYTES + (t / 3) * HW * 128);
      asm volatile("" : "+s"(soff));                               // (keeps the 12 x 3 x 2 x 2 sums XA + soff from being hoisted into registers)
      rd128(h ? (XA[r - 1][t % 3] ^ 64u) + soff : XA[r - 1][t % 3] + soff, fb[set][r - 1]);
    }
  };
  f32x4 acc[4][4];
  __syncthreads();                                                 // prologue: halo of position 0 and the first two weight stages have landed
#pragma unroll
  for (int k = 0; k < 8; ++k) issue_read(0, 0, 0, 0, hbuf(0, 0, 0), k);
  i5_lgkm0();
  int q = 0;
  for (int pi = 0; pi < mp; ++pi)
    for (int nt = 0; nt < g.ntn; ++nt) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < nch; ++c, ++q) {
        int pi2 = pi, nt2 = nt, c2 = c + 1;
        if (c2 == nch) { c2 = 0; if (++nt2 == g.ntn) { nt2 = 0; ++pi2; } }
        const bool nxt = pi2 < mp;
        const int hb = hbuf(pi, c, q), hb2 = hbuf(pi2, c2, q + 1);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const bool more = h == 0 || t < 8 || nxt;              // a phase follows this one (in this tile or the next)
            const int tn = h == 0 ? t : (t + 1) % 9, hn = h ^ 1;
            const int bn = (h == 1 && t == 8) ? hb2 : hb;
            const int qn = (h == 1 && t == 8) ? q + 1 : q;
#pragma unroll
            for (int i = 0; i < 4; ++i) touch8(fa[h][i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) touch8(fb[h][j]);
            __builtin_amdgcn_sched_barrier(0);
            // 16 MFMAs (m = 4 i + j); the next phase's eight fragment reads follow MFMAs 0..7, one each.  After a barrier (h == 1)
            // this phase's fragments have all landed; otherwise they were issued during the previous phase in the order fa0, fb0..3,
            // fa1..3 and MFMA m waits for its own (LDS returns in order): outstanding <= (older reads not needed yet) + (reads issued
            // in this phase so far).
#pragma unroll
            for (int m = 0; m < 16; ++m) {
              if (h == 0 && !(ABL & 4)) {
                if (!more) { if (m == 0) i5_lgkm0(); }
                else if (m <= 4) lgkm_wait3<6>();
                else if (m == 8) lgkm_wait3<9>();
                else if (m == 12) lgkm_wait3<8>();
              }
              if (!(ABL & 8)) i5_mfma(acc[m >> 2][m & 3], fa[h][m >> 2], fb[h][m & 3]);
              if (more && m < 8) issue_read(hn, qn, tn, hn, bn, m);
            }
            if (h == 0) {
              i5_lgkm0();                                          // stage s is read out before barrier s (WAR, see header)
              i5_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      // ---- tile end: ReLU, round, stage [pixel][channel] (16-byte chunk c of row r at chunk c ^ (r & 7)); the store waves drain it
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last MFMAs' results are in the registers (>= 12 wait states)
      {
        char* stg = smem + I5_STG_OFF;
        f32x4 bv[4];                                               // (added last, as igemm.hip's epilogues do: same rounding)
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4*>(biasL + nt * 64 + i * 16 + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = wm * 64 + j * 16 + l15;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int cl = i * 16 + 4 * kq;
            float v[4] = {acc[i][j][0] + bv[i][0], acc[i][j][1] + bv[i][1], acc[i][j][2] + bv[i][2], acc[i][j][3] + bv[i][3]};
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
            *reinterpret_cast<bf16x4*>(stg + row * 128 + (((cl >> 3) ^ (row & 7)) * 16) + (cl & 4) * 2) = o;
          }
        }
      }
      i5_lgkm0();
      i5_barrier();                                                // tile end
    }
}

// TH x TW rectangle of <= 256 pixels (halo <= 360 rows) that wastes the fewest pixel slots; widths that keep a 16-pixel fragment block
// inside one tile row preferred (igemm4.hip i4_geometry)
static bool i5_geometry(int Ho, int Wo, I5Geom& g, double& fill) {
  double best = 0.0;
  int bth = 0, btw = 0;
  for (int tw = 8; tw <= 254 && tw <= ((Wo + 7) & ~7); ++tw) {
    int th = 256 / tw;
    while (th > 1 && (th + 2) * (tw + 2) > I5_APIECES * 8) --th;
    if ((th + 2) * (tw + 2) > I5_APIECES * 8) continue;
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

int g_tune_igemm5 = 0;             // dct_tune_set(DCT_TUNE_IGEMM5, 1): large-image 3x3 stride-1 layers on the role-split kernel
int g_tune_igemm5_min_patches = 200;
extern int g_tune_igemm4_ablate;   // diagnostic builds: shared ablation selector

// Launch for a layer dct_conv2d has vetted (bf16, 3x3 stride 1, 16-byte aligned staged-epilogue views, 32-bit offsets): 1 = launched.
int dct_igemm5_launch(const void* params, int images, hipStream_t st) {
  IgemmParams p = *reinterpret_cast<const IgemmParams*>(params);
  if (!g_tune_igemm5 || p.Cin % 64 || p.N % 64 || p.N > I5_MAX_N) return 0;
  I5Geom g;
  double fill;
  if (!i5_geometry(p.Ho, p.Wo, g, fill) || fill < 0.70) return 0;
  g.npatch = images * g.tiles_x * g.tiles_y;
  g.ntn = p.N / 64;
  if (g.npatch < g_tune_igemm5_min_patches) return 0;
  p.partial = nullptr;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const dim3 grid((unsigned)std::min(g.npatch, cus), 1, 1);
#define I5_LAUNCH(ABLV) do { static bool a_ = false; if (!a_) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm5_kernel<ABLV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)I5_LDS); a_ = true; } \
      DCT_LAUNCH(DCT_PROF_IGEMM, (igemm5_kernel<ABLV>), grid, dim3(512), (size_t)I5_LDS, st, p, g); } while (0)
#ifdef DCT_I4_ABLATE
  switch (g_tune_igemm4_ablate) {
    case 1: I5_LAUNCH(1); return 1;
    case 4: I5_LAUNCH(4); return 1;
    case 8: I5_LAUNCH(8); return 1;
    case 12: I5_LAUNCH(12); return 1;
    case 64: I5_LAUNCH(64); return 1;
    case 77: I5_LAUNCH(77); return 1;
    default: break;
  }
#endif
  I5_LAUNCH(0);
#undef I5_LAUNCH
  return 1;
}
