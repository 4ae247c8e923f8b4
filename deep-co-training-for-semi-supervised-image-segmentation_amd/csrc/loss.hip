// Pixel-wise losses (K10), FGSM tail (K11), flat Adam (K12) and Dice counts.  All HBM-bound:
// one pass over [pixels][C] fp32 logits with C <= 8 channels held in registers; reductions are
// wave shuffle -> LDS -> per-block partial -> single-block fixed-order finalize (deterministic,
// no atomics on floats, no host synchronisation).
#include "dct_common.h"

namespace {

constexpr int kMaxBlocks = 1024;
constexpr float kEntEps = 1e-16f;  // loss.py:80

template <int C> __device__ __forceinline__ void load_px(const float* p, long long pix, float v[C]) {
  if constexpr (C == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + pix * 4);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else if constexpr (C == 2) {
    const f32x2 t = *reinterpret_cast<const f32x2*>(p + pix * 2);
    v[0] = t[0]; v[1] = t[1];
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = p[pix * C + c];
  }
}
template <int C> __device__ __forceinline__ void store_px_plain(float* p, long long pix, const float v[C]) {
  if constexpr (C == 4) *reinterpret_cast<f32x4*>(p + pix * 4) = f32x4{v[0], v[1], v[2], v[3]};
  else if constexpr (C == 2) *reinterpret_cast<f32x2*>(p + pix * 2) = f32x2{v[0], v[1]};
  else {
#pragma unroll
    for (int c = 0; c < C; ++c) p[pix * C + c] = v[c];
  }
}
// (not written as a call of itself with acc = false: a recursive function is not inlined, and every caller then passed its pixel through scratch
// memory to a real call -- round 5, seen in the ISA of every kernel of this file that stores through it)
template <int C> __device__ __forceinline__ void store_px(float* p, long long pix, const float v[C], bool acc) {
  if (acc) {
#pragma clang fp contract(off)    // the value is rounded before it is added, whatever expression the caller formed it from (as when this was a call)
    float o[C];
    load_px<C>(p, pix, o);
    float t[C];
#pragma unroll
    for (int c = 0; c < C; ++c) t[c] = v[c] + o[c];
    store_px_plain<C>(p, pix, t);
    return;
  }
  store_px_plain<C>(p, pix, v);
}
template <int C> __device__ __forceinline__ float softmax_px(const float x[C], float p[C]) {
  float m = x[0];
#pragma unroll
  for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { p[c] = expf(x[c] - m); s += p[c]; }
  const float inv = 1.f / s;
#pragma unroll
  for (int c = 0; c < C; ++c) p[c] *= inv;
  return m + logf(s);  // logsumexp
}

__device__ __forceinline__ void block_partial2(float a, float b, float* partial) {
  __shared__ float sm[8];
  a = wave_sum(a); b = wave_sum(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sm[w] = a; sm[4 + w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2 + 0] = sm[0] + sm[1] + sm[2] + sm[3];
    partial[blockIdx.x * 2 + 1] = sm[4] + sm[5] + sm[6] + sm[7];
  }
}
// out[0] = sum_a / (use_count ? sum_b : denom); out[1] = sum_b
__global__ __launch_bounds__(256) void finalize_kernel(const float* partial, int blocks, float* out, int use_count, float denom, int write_b) {
  __shared__ float sa[256], sb[256];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < blocks; i += 256) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  sa[threadIdx.x] = a; sb[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sa[0] / (use_count ? sb[0] : denom);
    if (write_b) out[1] = sb[0];
  }
}

// ---- cross entropy ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* logits, const long long* tgt, long long P, int ignore, float* partial) {
  float sum = 0.f, cnt = 0.f;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    const long long t = tgt[pix];
    if (t == ignore) continue;
    float x[C], p[C];
    load_px<C>(logits, pix, x);
    const float lse = softmax_px<C>(x, p);
    float xt = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) xt = (t == c) ? x[c] : xt;
    sum += lse - xt; cnt += 1.f;
  }
  block_partial2(sum, cnt, partial);
}
template <int C>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* logits, const long long* tgt, long long P, int ignore,
                                                      const float* count, const float* gscale, float gmul, float* dl, int acc) {
  const float g = (gscale ? gscale[0] : 1.f) * gmul / count[0];
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    const long long t = tgt[pix];
    float x[C], p[C], d[C];
    load_px<C>(logits, pix, x);
    softmax_px<C>(x, p);
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = (t == ignore) ? 0.f : g * (p[c] - (t == c ? 1.f : 0.f));
    store_px<C>(dl, pix, d, acc);
  }
}

// The step's cross-entropy pair in two launches instead of three (round 5: every small launch on a model's chain costs the step ~3 us): the backward
// kernel folds the forward kernel's block partials itself -- every block, in finalize_kernel's order, so the count (and the loss block 0 writes)
// come out bit for bit -- before it writes the logit gradients.
template <int C>
__global__ __launch_bounds__(256) void ce_bwd_fin_kernel(const float* logits, const long long* tgt, long long P, int ignore, const float* partial,
                                                          int blocks, float* out2, const float* gscale, float gmul, float* dl, int acc) {
  __shared__ float sa[256], sb[256];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < blocks; i += 256) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  sa[threadIdx.x] = a; sb[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
    __syncthreads();
  }
  const float count = sb[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) { out2[0] = sa[0] / count; out2[1] = count; }
  const float g = (gscale ? gscale[0] : 1.f) * gmul / count;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    const long long t = tgt[pix];
    float x[C], p[C], d[C];
    load_px<C>(logits, pix, x);
    softmax_px<C>(x, p);
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = (t == ignore) ? 0.f : g * (p[c] - (t == c ? 1.f : 0.f));
    store_px<C>(dl, pix, d, acc);
  }
}

// ---- softmax / entropy (module API) -------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* logits, float* probs, long long P) {
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    float x[C], p[C];
    load_px<C>(logits, pix, x);
    softmax_px<C>(x, p);
    store_px<C>(probs, pix, p, false);
  }
}
template <int C>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* probs, const float* dprobs, float* dl, long long P, int acc) {
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    float p[C], d[C], o[C];
    load_px<C>(probs, pix, p);
    load_px<C>(dprobs, pix, d);
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) dot += d[c] * p[c];
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = p[c] * (d[c] - dot);
    store_px<C>(dl, pix, o, acc);
  }
}
template <int C> __device__ __forceinline__ float entropy_px(const float p[C]) {
  float e = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) e += p[c] * logf(p[c] + kEntEps);
  return -e;
}
template <int C>
__global__ __launch_bounds__(256) void entropy_fwd_kernel(const float* probs, float* map, long long P) {
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    float p[C];
    load_px<C>(probs, pix, p);
    map[pix] = entropy_px<C>(p);
  }
}

// d/dp of  H(p) = -sum p log(p+eps):  -(log(p+eps) + p/(p+eps))
__device__ __forceinline__ float dent(float p) { return -(logf(p + kEntEps) + p / (p + kEntEps)); }
template <int C>
__global__ __launch_bounds__(256) void entropy_bwd_kernel(const float* probs, const float* dmap, float* dprobs, long long P) {
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    float p[C], d[C];
    load_px<C>(probs, pix, p);
    const float g = dmap[pix];
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = g * dent(p[c]);
    store_px<C>(dprobs, pix, d, false);
  }
}

// ---- JSD ------------------------------------------------------------------------------------------
// up to DCT_JSD_MAX_MODELS segmentators per call (the reference sweeps 2 / 4 / 6 views: script/GM/run_multiview.sh:2-6)
constexpr int MAXS = 8;
struct PtrPack { const float* in[MAXS]; float* out[MAXS]; };

template <int C, bool FROM_LOGITS>
__global__ __launch_bounds__(256) void jsd_fwd_kernel(PtrPack pk, int S, long long P, float* map, float* partial) {
  float sum = 0.f;
  const float invS = 1.f / (float)S;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    float mean[C];
#pragma unroll
    for (int c = 0; c < C; ++c) mean[c] = 0.f;
    float hsum = 0.f;
    for (int s = 0; s < S; ++s) {
      float x[C], p[C];
      load_px<C>(pk.in[s], pix, x);
      if constexpr (FROM_LOGITS) softmax_px<C>(x, p);
      else {
#pragma unroll
        for (int c = 0; c < C; ++c) p[c] = x[c];
      }
#pragma unroll
      for (int c = 0; c < C; ++c) mean[c] += p[c];
      hsum += entropy_px<C>(p);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) mean[c] *= invS;   // loss.py:193  reduce(+)/len
    const float j = entropy_px<C>(mean) - hsum * invS;
    if (map) map[pix] = j;
    sum += j;
  }
  if (partial) block_partial2(sum, 0.f, partial);
}
// FROM_LOGITS: dlogits_s (=|+=) g/P * softmax_bwd(dJ/dp_s);  else dprobs_s = dmap[pix] * dJ/dp_s
// SMAX: the models whose probabilities a thread keeps in registers (4: the co-training pairs / triples of the benchmark
// configurations; 8: the multi-view sweeps)
template <int C, bool FROM_LOGITS, int SMAX>
__global__ __launch_bounds__(256) void jsd_bwd_kernel(PtrPack pk, int S, long long P, const float* dmap, const float* gscale,
                                                       float gmul, int acc) {
  const float invS = 1.f / (float)S;
  float g = 0.f;
  if constexpr (FROM_LOGITS) g = (gscale ? gscale[0] : 1.f) * gmul / (float)P;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    float p[SMAX][C], mean[C];
#pragma unroll
    for (int c = 0; c < C; ++c) mean[c] = 0.f;
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      if (s < S) {
        float x[C];
        load_px<C>(pk.in[s], pix, x);
        if constexpr (FROM_LOGITS) softmax_px<C>(x, p[s]);
        else {
#pragma unroll
          for (int c = 0; c < C; ++c) p[s][c] = x[c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) mean[c] += p[s][c];
      }
    }
    float dm[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { mean[c] *= invS; dm[c] = dent(mean[c]); }
    const float gp = FROM_LOGITS ? g : dmap[pix];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      if (s < S) {
        float d[C];
#pragma unroll
        for (int c = 0; c < C; ++c) d[c] = invS * (dm[c] - dent(p[s][c]));
        if constexpr (FROM_LOGITS) {
          float dot = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) dot += d[c] * p[s][c];
          float o[C];
#pragma unroll
          for (int c = 0; c < C; ++c) o[c] = gp * p[s][c] * (d[c] - dot);
          store_px<C>(pk.out[s], pix, o, acc);
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) d[c] *= gp;
          store_px<C>(pk.out[s], pix, d, false);
        }
      }
    }
  }
}

// The co-training step's whole JSD section in ONE pass over the logits (round 5): mean-JSD partial sums (jsd_fwd_kernel<C, true>'s, on its
// grid: same pixels per thread, same order -> the same block partials), the S softmax maps the step returns (softmax_fwd_kernel's) and the S
// logit gradients (jsd_bwd_kernel<C, true>'s arithmetic).  Five launches became two (this + the finalize) in the one stretch of the step where
// nothing else runs: both models' forward passes have just joined (tools/phase_stamps.py).  Bit-identical to the separate launches.
template <int C, int SMAX>
__global__ __launch_bounds__(256) void jsd_step_kernel(PtrPack pk, PtrPack probs, int S, long long P, const float* gscale, float gmul, int acc,
                                                        int want_grad, float* partial) {
  const float invS = 1.f / (float)S;
  const float g = (gscale ? gscale[0] : 1.f) * gmul / (float)P;
  float sum = 0.f;
  // Two of the thread's pixels per trip (two-model case), every load of both (logits and the gradients to add to) issued before the first store: with 1024 blocks
  // a thread walks 2+ pixels and the launch sits where nothing else runs -- the trip used to be six dependent memory round trips per pixel.
  // Per pixel the arithmetic and its order are those of the separate kernels; the thread's pixels are summed in the same order.
  constexpr int NP = SMAX <= 2 ? 2 : 1;       // (two pixels of four or eight models do not fit the register file)
  const long long stride = (long long)gridDim.x * 256;
  for (long long pix0 = (long long)blockIdx.x * 256 + threadIdx.x; pix0 < P; pix0 += NP * stride) {
    const bool two = NP == 2 && pix0 + stride < P;
    float x[NP][SMAX][C], old[NP][SMAX][C];
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
      if (s < S) {
        load_px<C>(pk.in[s], pix0, x[0][s]);
        if constexpr (NP == 2) { if (two) load_px<C>(pk.in[s], pix0 + stride, x[NP - 1][s]); }
        if (want_grad && acc) {
          load_px<C>(pk.out[s], pix0, old[0][s]);
          if constexpr (NP == 2) { if (two) load_px<C>(pk.out[s], pix0 + stride, old[NP - 1][s]); }
        }
      }
    }
#pragma unroll
    for (int h = 0; h < NP; ++h) {
      if (h == 1 && !two) continue;
      const long long pix = pix0 + h * stride;
      float p[SMAX][C], mean[C];
#pragma unroll
      for (int c = 0; c < C; ++c) mean[c] = 0.f;
      float hsum = 0.f;
#pragma unroll
      for (int s = 0; s < SMAX; ++s) {
        if (s < S) {
          softmax_px<C>(x[h][s], p[s]);
#pragma unroll
          for (int c = 0; c < C; ++c) mean[c] += p[s][c];
          hsum += entropy_px<C>(p[s]);
          if (probs.out[s]) store_px<C>(probs.out[s], pix, p[s], false);
        }
      }
      float dm[C];
#pragma unroll
      for (int c = 0; c < C; ++c) mean[c] *= invS;
      sum += entropy_px<C>(mean) - hsum * invS;
      if (want_grad) {
#pragma unroll
        for (int c = 0; c < C; ++c) dm[c] = dent(mean[c]);
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
          if (s < S) {
            float d[C];
#pragma unroll
            for (int c = 0; c < C; ++c) d[c] = invS * (dm[c] - dent(p[s][c]));
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) dot += d[c] * p[s][c];
            float o[C];
#pragma unroll
            for (int c = 0; c < C; ++c) o[c] = g * p[s][c] * (d[c] - dot);
            if (acc) {
#pragma clang fp contract(off)    // store_px's accumulate: round, then add
#pragma unroll
              for (int c = 0; c < C; ++c) o[c] = o[c] + old[h][s][c];
            }
            store_px<C>(pk.out[s], pix, o, false);
          }
        }
      }
    }
  }
  block_partial2(sum, 0.f, partial);
}

// ---- KL(y || p) -----------------------------------------------------------------------------------
template <int C, bool FROM_LOGITS>
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* pin, const float* yin, long long P, float eps, float* map, float* partial) {
  float sum = 0.f;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    float a[C], b[C], p[C], y[C];
    load_px<C>(pin, pix, a);
    load_px<C>(yin, pix, b);
    if constexpr (FROM_LOGITS) { softmax_px<C>(a, p); softmax_px<C>(b, y); }
    else {
#pragma unroll
      for (int c = 0; c < C; ++c) { p[c] = a[c]; y[c] = b[c]; }
    }
    float ylogy = 0.f, ylogp = 0.f;   // loss.py:127-130
#pragma unroll
    for (int c = 0; c < C; ++c) { ylogy += y[c] * logf(y[c] + eps); ylogp += y[c] * logf(p[c] + eps); }
    const float k = ylogy - ylogp;
    if (map) map[pix] = k;
    sum += k;
  }
  if (partial) block_partial2(sum, 0.f, partial);
}
template <int C, bool FROM_LOGITS>
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* pin, const float* yin, long long P, float eps, const float* dmap,
                                                      const float* gscale, float gmul, float* dout, int acc) {
  float g = 0.f;
  if constexpr (FROM_LOGITS) g = (gscale ? gscale[0] : 1.f) * gmul / (float)P;
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    float a[C], b[C], p[C], y[C], d[C];
    load_px<C>(pin, pix, a);
    load_px<C>(yin, pix, b);
    if constexpr (FROM_LOGITS) { softmax_px<C>(a, p); softmax_px<C>(b, y); }
    else {
#pragma unroll
      for (int c = 0; c < C; ++c) { p[c] = a[c]; y[c] = b[c]; }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = -y[c] / (p[c] + eps);
    if constexpr (FROM_LOGITS) {
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) dot += d[c] * p[c];
      float o[C];
#pragma unroll
      for (int c = 0; c < C; ++c) o[c] = g * p[c] * (d[c] - dot);
      store_px<C>(dout, pix, o, acc);
    } else {
      const float gp = dmap[pix];
#pragma unroll
      for (int c = 0; c < C; ++c) d[c] *= gp;
      store_px<C>(dout, pix, d, false);
    }
  }
}

// ---- argmax / dice / fgsm / adam ---------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void argmax_kernel(const float* x, long long* cls, long long P) {
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (long long)gridDim.x * 256) {
    float v[C];
    load_px<C>(x, pix, v);
    int best = 0;
#pragma unroll
    for (int c = 1; c < C; ++c) if (v[c] > v[best]) best = c;
    cls[pix] = best;
  }
}
template <int C>
__global__ __launch_bounds__(256) void dice_kernel(const float* logits, const long long* gt, long long PPI,
                                                    int* inter, int* psum, int* gsum) {
  __shared__ int h[3 * C];
  if (threadIdx.x < 3 * C) h[threadIdx.x] = 0;
  __syncthreads();
  const int b = blockIdx.y;
  const float* lb = logits + (long long)b * PPI * C;
  const long long* gb = gt + (long long)b * PPI;
  int li[C], lp[C], lg[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { li[c] = 0; lp[c] = 0; lg[c] = 0; }
  for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < PPI; pix += (long long)gridDim.x * 256) {
    float v[C];
    load_px<C>(lb, pix, v);
    int best = 0;
#pragma unroll
    for (int c = 1; c < C; ++c) if (v[c] > v[best]) best = c;
    const long long t = gb[pix];
#pragma unroll
    for (int c = 0; c < C; ++c) { lp[c] += best == c; lg[c] += t == c; li[c] += (best == c) && (t == c); }
  }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    if (li[c]) atomicAdd(&h[c], li[c]);
    if (lp[c]) atomicAdd(&h[C + c], lp[c]);
    if (lg[c]) atomicAdd(&h[2 * C + c], lg[c]);
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    if (h[c]) atomicAdd(&inter[b * C + c], h[c]);
    if (h[C + c]) atomicAdd(&psum[b * C + c], h[C + c]);
    if (h[2 * C + c]) atomicAdd(&gsum[b * C + c], h[2 * C + c]);
  }
}

__global__ __launch_bounds__(256) void fgsm_kernel(const float* x, const float* g, float eps, float* xa, float* noise, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gv = g[i];
    const float s = gv > 0.f ? eps : (gv < 0.f ? -eps : 0.f);   // eps*sign(g), sign(0)=0  (AEGenerator.py:42-45)
    if (noise) noise[i] = s;
    xa[i] = x[i] + s;
  }
}

// omb1/omb2 = 1-beta computed in double on the host, as torch does (float(1-0.999) != 1.f-0.999f)
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float step_size, float bc2s, float omb1, float b2, float omb2, float eps, float wd, float gs) {
  g = fmaf(wd, p, g * gs);                      // gs: 1, or the inverse power-of-two loss scale of an fp16 step (exact)
  m = m + (g - m) * omb1;                       // exp_avg.lerp_(grad, 1-b1)
  v = v * b2 + omb2 * g * g;                    // mul_(b2).addcmul_(g, g, 1-b2)
  const float denom = sqrtf(v) / bc2s + eps;
  p = p - step_size * (m / denom);
}
// state (nullable): device-resident {step count t, learning rate, table base, table length} (doubles).  When given,
// the step-dependent scalars come from device memory so that a captured HIP graph of the step needs no per-step
// host value: table[2*(t-base-1)] = {1 - beta1^t, sqrt(1 - beta2^t)} as the HOST computed them (doubles; the step size
// lr / bc1 is then the same correctly rounded division the host does: bit-identical to the scalar-argument path, and a
// learning-rate change touches only state[1]); outside the table they are formed here in double.
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float step_size,
                                                    float bc2s, float omb1, float b2, float omb2, float eps, float wd, float gs, bf16_t* shadow,
                                                    const double* state, const double* table, double beta1, double beta2) {
  if (state) {
    const double t = state[0], lr = state[1];
    const long long idx = (long long)t - (long long)state[2] - 1;
    if (table && idx >= 0 && idx < (long long)state[3]) {
      step_size = (float)(lr / table[2 * idx]);        // the division the host does (lr / bc1 in double, then fp32)
      bc2s = (float)table[2 * idx + 1];
    } else {
      step_size = (float)(lr / (1.0 - pow(beta1, t)));
      bc2s = (float)sqrt(1.0 - pow(beta2, t));
    }
  }
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pp = pv[k], mm = mv[k], v2 = vv[k];
      adam1(pp, gv[k], mm, v2, step_size, bc2s, omb1, b2, omb2, eps, wd, gs);
      pv[k] = pp; mv[k] = mm; vv[k] = v2;
    }
    reinterpret_cast<f32x4*>(p)[i] = pv; reinterpret_cast<f32x4*>(m)[i] = mv; reinterpret_cast<f32x4*>(v)[i] = vv;
    if (shadow) {
      bf16x4 s;
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] = (bf16_t)pv[k];
      reinterpret_cast<bf16x4*>(shadow)[i] = s;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = n4 * 4 + threadIdx.x;
    float pp = p[i], mm = m[i], v2 = v[i];
    adam1(pp, g[i], mm, v2, step_size, bc2s, omb1, b2, omb2, eps, wd, gs);
    p[i] = pp; m[i] = mm; v[i] = v2;
    if (shadow) shadow[i] = (bf16_t)pp;
  }
}

__global__ void adam_advance_kernel(double* state) { state[0] += 1.0; }

static inline unsigned grid_for(long long P) {
  long long b = (P + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}
static inline unsigned wide_grid(long long n) {
  long long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

#define DISPATCH_C(C_, ...)                                                   \
  switch (C_) {                                                               \
    case 2: { constexpr int C = 2; __VA_ARGS__; break; }                      \
    case 3: { constexpr int C = 3; __VA_ARGS__; break; }                      \
    case 4: { constexpr int C = 4; __VA_ARGS__; break; }                      \
    case 5: { constexpr int C = 5; __VA_ARGS__; break; }                      \
    case 6: { constexpr int C = 6; __VA_ARGS__; break; }                      \
    case 7: { constexpr int C = 7; __VA_ARGS__; break; }                      \
    case 8: { constexpr int C = 8; __VA_ARGS__; break; }                      \
    default: return DCT_ERR_UNSUPPORTED;                                      \
  }

extern "C" size_t dct_loss_workspace_bytes(int64_t) { return (size_t)kMaxBlocks * 2 * sizeof(float); }

static inline bool ws_ok(void* ws, size_t bytes) { return ws && bytes >= (size_t)kMaxBlocks * 2 * sizeof(float); }

extern "C" int dct_ce_fwd(const float* logits, const int64_t* targets, int64_t pixels, int C_, int ignore_index,
                          float* out2, void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (!logits || !targets || !out2 || pixels < 1) return DCT_ERR_BAD_ARG;
  if (!ws_ok(workspace, workspace_bytes)) return DCT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(pixels);
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, ce_fwd_kernel<C>, dim3(grid), dim3(256), 0, st, logits, (const long long*)targets, (long long)pixels, ignore_index, (float*)workspace));
  DCT_LAUNCH(DCT_PROF_LOSS, finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, (int)grid, out2, 1, 1.f, 1);
  return dct_check_launch();
}
extern "C" int dct_ce_bwd(const float* logits, const int64_t* targets, int64_t pixels, int C_, int ignore_index,
                          const float* count, const float* gscale, float gmul, float* dlogits, int accumulate,
                          dct_stream stream) {
  if (!logits || !targets || !count || !dlogits || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, ce_bwd_kernel<C>, dim3(wide_grid(pixels)), dim3(256), 0, st, logits, (const long long*)targets, (long long)pixels, ignore_index, count, gscale, gmul, dlogits, accumulate));
  return dct_check_launch();
}
extern "C" int dct_ce_step(const float* logits, const int64_t* targets, int64_t pixels, int C_, int ignore_index, float* out2,
                           const float* gscale, float gmul, float* dlogits, int accumulate, void* workspace, size_t workspace_bytes,
                           dct_stream stream) {
  if (!logits || !targets || !out2 || !dlogits || pixels < 1) return DCT_ERR_BAD_ARG;
  if (!ws_ok(workspace, workspace_bytes)) return DCT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(pixels);       // dct_ce_fwd's grid: the same block partials
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, ce_fwd_kernel<C>, dim3(grid), dim3(256), 0, st, logits, (const long long*)targets, (long long)pixels, ignore_index, (float*)workspace));
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, ce_bwd_fin_kernel<C>, dim3(wide_grid(pixels)), dim3(256), 0, st, logits, (const long long*)targets, (long long)pixels, ignore_index,
                            (const float*)workspace, (int)grid, out2, gscale, gmul, dlogits, accumulate));
  return dct_check_launch();
}
extern "C" int dct_softmax_fwd(const float* logits, float* probs, int64_t pixels, int C_, dct_stream stream) {
  if (!logits || !probs || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, softmax_fwd_kernel<C>, dim3(wide_grid(pixels)), dim3(256), 0, st, logits, probs, (long long)pixels));
  return dct_check_launch();
}
extern "C" int dct_softmax_bwd(const float* probs, const float* dprobs, float* dlogits, int64_t pixels, int C_,
                               int accumulate, dct_stream stream) {
  if (!probs || !dprobs || !dlogits || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, softmax_bwd_kernel<C>, dim3(wide_grid(pixels)), dim3(256), 0, st, probs, dprobs, dlogits, (long long)pixels, accumulate));
  return dct_check_launch();
}
extern "C" int dct_entropy_fwd(const float* probs, float* map, int64_t pixels, int C_, dct_stream stream) {
  if (!probs || !map || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, entropy_fwd_kernel<C>, dim3(wide_grid(pixels)), dim3(256), 0, st, probs, map, (long long)pixels));
  return dct_check_launch();
}

extern "C" int dct_entropy_bwd(const float* probs, const float* dmap, float* dprobs, int64_t pixels, int C_, dct_stream stream) {
  if (!probs || !dmap || !dprobs || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, entropy_bwd_kernel<C>, dim3(wide_grid(pixels)), dim3(256), 0, st, probs, dmap, dprobs, (long long)pixels));
  return dct_check_launch();
}

static bool fill_pack(PtrPack& pk, const float* const* in, float* const* out, int S) {
  if (S < 1 || S > MAXS || !in) return false;
  for (int s = 0; s < MAXS; ++s) { pk.in[s] = nullptr; pk.out[s] = nullptr; }
  for (int s = 0; s < S; ++s) {
    if (!in[s]) return false;
    pk.in[s] = in[s];
    if (out) { if (!out[s]) return false; pk.out[s] = out[s]; }
  }
  return true;
}

extern "C" int dct_jsd_map_fwd(const float* const* probs, int S, float* map, int64_t pixels, int C_, dct_stream stream) {
  PtrPack pk;
  if (!fill_pack(pk, probs, nullptr, S) || !map || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (jsd_fwd_kernel<C, false>), dim3(wide_grid(pixels)), dim3(256), 0, st, pk, S, (long long)pixels, map, (float*)nullptr));
  return dct_check_launch();
}
extern "C" int dct_jsd_map_bwd(const float* const* probs, int S, const float* dmap, float* const* dprobs,
                               int64_t pixels, int C_, dct_stream stream) {
  PtrPack pk;
  if (!fill_pack(pk, probs, dprobs, S) || !dmap || !dprobs || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (S <= 4) { DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (jsd_bwd_kernel<C, false, 4>), dim3(wide_grid(pixels)), dim3(256), 0, st, pk, S, (long long)pixels, dmap, (const float*)nullptr, 1.f, 0)); }
  else { DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (jsd_bwd_kernel<C, false, MAXS>), dim3(wide_grid(pixels)), dim3(256), 0, st, pk, S, (long long)pixels, dmap, (const float*)nullptr, 1.f, 0)); }
  return dct_check_launch();
}
extern "C" int dct_jsd_logits_fwd(const float* const* logits, int S, int64_t pixels, int C_, float* out1,
                                  void* workspace, size_t workspace_bytes, dct_stream stream) {
  PtrPack pk;
  if (!fill_pack(pk, logits, nullptr, S) || !out1 || pixels < 1) return DCT_ERR_BAD_ARG;
  if (!ws_ok(workspace, workspace_bytes)) return DCT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(pixels);
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (jsd_fwd_kernel<C, true>), dim3(grid), dim3(256), 0, st, pk, S, (long long)pixels, (float*)nullptr, (float*)workspace));
  DCT_LAUNCH(DCT_PROF_LOSS, finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, (int)grid, out1, 0, (float)pixels, 0);
  return dct_check_launch();
}
extern "C" int dct_jsd_logits_bwd(const float* const* logits, int S, int64_t pixels, int C_, const float* gscale,
                                  float gmul, float* const* dlogits, int accumulate, dct_stream stream) {
  PtrPack pk;
  if (!fill_pack(pk, logits, dlogits, S) || !dlogits || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (S <= 4) { DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (jsd_bwd_kernel<C, true, 4>), dim3(wide_grid(pixels)), dim3(256), 0, st, pk, S, (long long)pixels, (const float*)nullptr, gscale, gmul, accumulate)); }
  else { DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (jsd_bwd_kernel<C, true, MAXS>), dim3(wide_grid(pixels)), dim3(256), 0, st, pk, S, (long long)pixels, (const float*)nullptr, gscale, gmul, accumulate)); }
  return dct_check_launch();
}

extern "C" int dct_jsd_logits_step(const float* const* logits, int S, int64_t pixels, int C_, float* out1, float* const* probs,
                                   const float* gscale, float gmul, float* const* dlogits, int accumulate,
                                   void* workspace, size_t workspace_bytes, dct_stream stream) {
  PtrPack pk, pp;
  if (!fill_pack(pk, logits, dlogits, S) || !out1 || pixels < 1) return DCT_ERR_BAD_ARG;
  for (int s = 0; s < MAXS; ++s) { pp.in[s] = nullptr; pp.out[s] = (probs && s < S) ? probs[s] : nullptr; }
  if (!dlogits) { for (int s = 0; s < MAXS; ++s) pk.out[s] = nullptr; }
  if (!ws_ok(workspace, workspace_bytes)) return DCT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(pixels);       // dct_jsd_logits_fwd's grid: the block partials, hence the mean, come out bit for bit
  const int want = dlogits ? 1 : 0;
  if (S <= 2) { DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (jsd_step_kernel<C, 2>), dim3(grid), dim3(256), 0, st, pk, pp, S, (long long)pixels, gscale, gmul, accumulate, want, (float*)workspace)); }
  else if (S <= 4) { DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (jsd_step_kernel<C, 4>), dim3(grid), dim3(256), 0, st, pk, pp, S, (long long)pixels, gscale, gmul, accumulate, want, (float*)workspace)); }
  else { DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (jsd_step_kernel<C, MAXS>), dim3(grid), dim3(256), 0, st, pk, pp, S, (long long)pixels, gscale, gmul, accumulate, want, (float*)workspace)); }
  DCT_LAUNCH(DCT_PROF_LOSS, finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, (int)grid, out1, 0, (float)pixels, 0);
  return dct_check_launch();
}

extern "C" int dct_kl_map_fwd(const float* p, const float* y, float* map, int64_t pixels, int C_, float eps, dct_stream stream) {
  if (!p || !y || !map || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (kl_fwd_kernel<C, false>), dim3(wide_grid(pixels)), dim3(256), 0, st, p, y, (long long)pixels, eps, map, (float*)nullptr));
  return dct_check_launch();
}
extern "C" int dct_kl_map_bwd(const float* p, const float* y, const float* dmap, float* dp, int64_t pixels,
                              int C_, float eps, dct_stream stream) {
  if (!p || !y || !dmap || !dp || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (kl_bwd_kernel<C, false>), dim3(wide_grid(pixels)), dim3(256), 0, st, p, y, (long long)pixels, eps, dmap, (const float*)nullptr, 1.f, dp, 0));
  return dct_check_launch();
}
extern "C" int dct_kl_logits_fwd(const float* p_logits, const float* y_logits, int64_t pixels, int C_, float eps,
                                 float* out1, void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (!p_logits || !y_logits || !out1 || pixels < 1) return DCT_ERR_BAD_ARG;
  if (!ws_ok(workspace, workspace_bytes)) return DCT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = grid_for(pixels);
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (kl_fwd_kernel<C, true>), dim3(grid), dim3(256), 0, st, p_logits, y_logits, (long long)pixels, eps, (float*)nullptr, (float*)workspace));
  DCT_LAUNCH(DCT_PROF_LOSS, finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, (int)grid, out1, 0, (float)pixels, 0);
  return dct_check_launch();
}
extern "C" int dct_kl_logits_bwd(const float* p_logits, const float* y_logits, int64_t pixels, int C_, float eps,
                                 const float* gscale, float gmul, float* dp_logits, int accumulate, dct_stream stream) {
  if (!p_logits || !y_logits || !dp_logits || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, (kl_bwd_kernel<C, true>), dim3(wide_grid(pixels)), dim3(256), 0, st, p_logits, y_logits, (long long)pixels, eps, (const float*)nullptr, gscale, gmul, dp_logits, accumulate));
  return dct_check_launch();
}

extern "C" int dct_argmax(const float* x, int64_t* cls, int64_t pixels, int C_, dct_stream stream) {
  if (!x || !cls || pixels < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, argmax_kernel<C>, dim3(wide_grid(pixels)), dim3(256), 0, st, x, (long long*)cls, (long long)pixels));
  return dct_check_launch();
}

extern "C" int dct_dice_counts(const float* logits, const int64_t* gt, int B, int64_t pixels_per_image, int C_,
                               int32_t* inter, int32_t* psum, int32_t* gsum, dct_stream stream) {
  if (!logits || !gt || !inter || !psum || !gsum || B < 1 || pixels_per_image < 1) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  long long bx = (pixels_per_image + 255) / 256;
  if (bx > 64) bx = 64;
  DISPATCH_C(C_, DCT_LAUNCH(DCT_PROF_LOSS, dice_kernel<C>, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, st, logits, (const long long*)gt, (long long)pixels_per_image, inter, psum, gsum));
  return dct_check_launch();
}

// Dice of one DiceMeter.add from the counts, plus the meter's running moments, in ONE launch (one block): rows = B (2-D, per
// slice) or 1 (3-D, counts summed over the batch); dice[row][c] = (2 inter + smooth) / (psum + gsum + smooth) in fp32 as
// the reference's einsum formula gives it; acc[0][j] += value, acc[1][j] += value^2 (double) for the C classes and, at
// j = C, for the row's mean over the report axes (bit c of axes_mask).  Rows are folded in order by one thread per column.
__global__ __launch_bounds__(256) void dice_update_kernel(const int* inter, const int* psum, const int* gsum, int B, int C_, int rows,
                                                          unsigned axes_mask, float smooth, float* dice, double* acc) {
  __shared__ float sd[64 * 8];
  for (int i = threadIdx.x; i < rows * C_; i += 256) {
    const int r = i / C_, c = i - r * C_;
    long long a = 0, b = 0, g = 0;
    if (rows == B) { a = inter[i]; b = psum[i]; g = gsum[i]; }
    else for (int k = 0; k < B; ++k) { a += inter[k * C_ + c]; b += psum[k * C_ + c]; g += gsum[k * C_ + c]; }
    const float d = (2.f * (float)a + smooth) / ((float)(b + g) + smooth);
    dice[i] = d;
    if (r < 64) sd[r * 8 + c] = d;
  }
  __syncthreads();
  const int j = threadIdx.x;
  if (j <= C_) {
    double s0 = 0.0, s1 = 0.0;
    int naxes = 0;
    for (int c = 0; c < C_; ++c) naxes += (axes_mask >> c) & 1;
    for (int r = 0; r < rows; ++r) {
      float v;
      if (j < C_) v = sd[r * 8 + j];
      else {
        float t = 0.f;
        for (int c = 0; c < C_; ++c) if ((axes_mask >> c) & 1) t += sd[r * 8 + c];
        v = t / (float)naxes;
      }
      s0 += (double)v; s1 += (double)v * (double)v;
    }
    acc[j] += s0; acc[(C_ + 1) + j] += s1;
  }
}

extern "C" int dct_dice_update(const int32_t* inter, const int32_t* psum, const int32_t* gsum, int B, int C_, int method3d,
                               uint32_t axes_mask, float smooth, float* dice, double* acc, dct_stream stream) {
  if (!inter || !psum || !gsum || !dice || !acc || B < 1 || B > 64 || C_ < 1 || C_ > 8 || !(axes_mask & ((1u << C_) - 1))) return DCT_ERR_BAD_ARG;
  DCT_LAUNCH(DCT_PROF_LOSS, dice_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, inter, psum, gsum, B, C_, method3d ? 1 : B,
             axes_mask, smooth, dice, acc);
  return dct_check_launch();
}

extern "C" int dct_fgsm_step(const float* x, const float* g, float eps, float* x_adv, float* noise, int64_t n,
                             dct_stream stream) {
  if (!x || !g || !x_adv || n < 1) return DCT_ERR_BAD_ARG;
  DCT_LAUNCH(DCT_PROF_OTHER, fgsm_kernel, dim3(wide_grid(n)), dim3(256), 0, (hipStream_t)stream, x, g, eps, x_adv, noise, (long long)n);
  return dct_check_launch();
}

extern "C" int dct_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float step_size,
                             float bc2_sqrt, double beta1, double beta2, float eps, float weight_decay, float grad_scale,
                             void* bf16_shadow, dct_stream stream) {
  if (!p || !g || !m || !v || n < 1) return DCT_ERR_BAD_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return DCT_ERR_UNSUPPORTED;
  if (bf16_shadow && ((uintptr_t)bf16_shadow & 7)) return DCT_ERR_UNSUPPORTED;
  DCT_LAUNCH(DCT_PROF_ADAM, adam_kernel, dim3(wide_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
             step_size, bc2_sqrt, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, weight_decay, grad_scale, (bf16_t*)bf16_shadow,
             (const double*)nullptr, (const double*)nullptr, beta1, beta2);
  return dct_check_launch();
}

extern "C" int dct_adam_flat_dev(float* p, const float* g, float* m, float* v, int64_t n, double* state,
                                 const double* table, double beta1, double beta2, float eps, float weight_decay, float grad_scale,
                                 void* bf16_shadow, dct_stream stream) {
  if (!p || !g || !m || !v || !state || n < 1) return DCT_ERR_BAD_ARG;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return DCT_ERR_UNSUPPORTED;
  if (bf16_shadow && ((uintptr_t)bf16_shadow & 7)) return DCT_ERR_UNSUPPORTED;
  if ((uintptr_t)state & 7) return DCT_ERR_UNSUPPORTED;
  DCT_LAUNCH(DCT_PROF_ADAM, adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  DCT_LAUNCH(DCT_PROF_ADAM, adam_kernel, dim3(wide_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
             0.f, 1.f, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, weight_decay, grad_scale, (bf16_t*)bf16_shadow,
             (const double*)state, table, beta1, beta2);
  return dct_check_launch();
}
