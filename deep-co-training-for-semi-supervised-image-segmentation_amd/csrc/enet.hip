// Enet kernels (K2/K3/K4/K6/K7/K8 of SURVEY.md 8a): arch/enet.py:8-243 of the reference.
//
// Enet's layers have 1..128 channels (internal widths 3, 16, 32): there is no dense contraction
// worth an MFMA tile and the net is HBM/launch bound, so these are direct VALU
// kernels built around FUSION rather than GEMM shape:
//   * every conv reads its input through the producer's BatchNorm + PReLU/ReLU ("normalise on
//     load": y = act(scale[c] * raw + shift[c])), so a normalised activation is never written;
//   * one generic kernel covers conv / transposed conv / both data gradients (direct or
//     "transposed" gather form, weights addressed through three strides), with optional bias,
//     residual-gradient add and "+=" epilogue;
//   * BatchNorm statistics and the three backward sums come from per-block partials (double
//     accumulators) folded in fixed order by a one-block finalize kernel: deterministic;
//   * the bottleneck tail relu(main + act(bn(raw))) also does the 2x2 max-pool-with-indices /
//     max-unpool / zero-channel-pad of the down- and up-sampling bottlenecks.
// Activations are NHWC in `dtype` (bf16 or f32); all per-channel vectors and the math are fp32.
#include <algorithm>
#include <vector>
#include "dct_common.h"

static const int g_enet_reduce_ppt = 8;           // pixels per thread of a per-channel reduction (sets the number of partial rows)
static const int g_enet_fold_threads = 1024;      // threads of the one-block finalize kernels (256 / 512 swept in round 3: level or slower)
int g_enet_mfma = 3;                 // bf16 / f16 mode: bit 0 = MFMA form of the convolutions with >= 16 input channels, bit 1 = of the
                                     // weight gradients (0: the fp32 VALU kernels)
int g_enet_mwgrad_waves = 2048;      // MFMA weight gradient: waves a launch aims for (pixel slices x tiles) ...
static const int g_enet_mwgrad_min_steps = 4;     // ... with at least this many 16-pixel MFMA steps per slice (multiple of 4)
// (measured and removed, DESIGN.md 4.2: a one-launch channel-owner BatchNorm for small tensors -- 8 x 25 x 25 x 32 backward 60 us against
//  14 us for the three split launches: C / 8 blocks cannot pull the tensor through 4-16 CUs fast enough; a BatchNorm-backward apply
//  kernel on 8 channels per thread -- cfg4 16.6 vs 15.9 ms, cfg5 38.9 vs 36.8)
int g_enet_wgrad_slices = 1;          // 0: one pixel slice per round whatever the tile count
int g_enet_wgrad_max_blocks = 1024;   // dct_tune_set(DCT_TUNE_ENET_WGRAD_BLOCKS, n); <= WG_MAX_BLOCKS

namespace {


// ---- launch plumbing: single and GROUPED launches of the same kernel body -------------------------------------------
// Every Enet kernel body is a functor F { using Args; THREADS; static __device__ run(const Args&) } with ALL of its
// arguments in one struct.  enet_one<F> launches it as before.  enet_grp<F> runs up to GROUP_MAX independent argument
// sets in ONE launch (blockIdx.z picks the set): the 2S forward (or backward) passes of a co-training step have identical
// shapes and no dependencies among themselves, each underfills the device and each of their ~200-500 launches is a seam on
// its queue -- grouped, they are one chain of launches instead of 2S (trainer/cotraining_totalloss.py::_run_step_wide).
// The bodies are the same code on the same operands, so a grouped pass is bit-identical to the pass launched alone.
// Recording (dct_group_begin / _member / _end): while a group is open the entry points below do not launch; they append
// {body, grid, block, LDS bytes, argument blob} to the current member's list.  dct_group_end zips the lists: entry k of
// all members becomes one grouped launch when body and launch geometry agree, else one launch per member.
constexpr int GROUP_MAX = 4;        // (4 x sizeof(ConvP) must fit the 4 KB kernel-argument segment)
template <typename A> struct GroupArgs { A m[GROUP_MAX]; };
template <typename F> __global__ __launch_bounds__(F::THREADS) void enet_one(typename F::Args a) { F::run(a); }
template <typename F> __global__ __launch_bounds__(F::THREADS) void enet_grp(GroupArgs<typename F::Args> g) { F::run(g.m[blockIdx.z]); }

struct GroupRec {
  void (*launch)(const void* const* args, int n, dim3 grid, dim3 block, size_t lds, hipStream_t st, int cls);
  dim3 grid, block;
  size_t lds;
  int cls;
  std::vector<char> blob;
};
struct GroupState {
  bool active = false;
  bool leaves_only = false;   // "side" recording: only the launches of LEAF work (weight gradients, bias sums -- nothing downstream
                              // in the pass reads their results) are held back, everything else launches at once
  int member = 0;
  std::vector<std::vector<GroupRec>> recs;
};
// The recording state belongs to the thread that opened the group: a dct_enet_* call from another thread (a second trainer, a
// background evaluation) launches normally instead of being captured into -- or refused by -- somebody else's open group.
thread_local GroupState g_grp;
thread_local bool g_leaf_scope = false;    // set by the entry points whose launches are leaves (dct_enet_wgrad, dct_enet_channel_sum)
struct LeafScope { bool was; LeafScope() : was(g_leaf_scope) { g_leaf_scope = true; } ~LeafScope() { g_leaf_scope = was; } };

template <typename F>
void enet_launch_n(const void* const* args, int n, dim3 grid, dim3 block, size_t lds, hipStream_t st, int cls) {
  using A = typename F::Args;
  static_assert(sizeof(GroupArgs<A>) <= 4000, "grouped kernel arguments exceed the kernarg segment");
  if (n == 1) {
    DCT_LAUNCH(cls, enet_one<F>, grid, block, lds, st, *reinterpret_cast<const A*>(args[0]));
    return;
  }
  GroupArgs<A> g;
  for (int i = 0; i < n; ++i) g.m[i] = *reinterpret_cast<const A*>(args[i]);
  for (int i = n; i < GROUP_MAX; ++i) g.m[i] = g.m[0];
  grid.z = (unsigned)n;
  DCT_LAUNCH(cls, enet_grp<F>, grid, block, lds, st, g);
}

template <typename F>
void enet_launch(int cls, dim3 grid, dim3 block, size_t lds, hipStream_t st, const typename F::Args& a) {
  if (g_grp.active && (!g_grp.leaves_only || g_leaf_scope)) {
    GroupRec r;
    r.launch = &enet_launch_n<F>; r.grid = grid; r.block = block; r.lds = lds; r.cls = cls;
    r.blob.assign(reinterpret_cast<const char*>(&a), reinterpret_cast<const char*>(&a) + sizeof(a));
    g_grp.recs[g_grp.member].push_back(std::move(r));
    return;
  }
  const void* one[1] = {&a};
  enet_launch_n<F>(one, 1, grid, block, lds, st, cls);
}

// Element access through a view whose storage is `T` or, when its bit of the call's f32 mask is set, fp32
// (raw conv outputs stay fp32 in bf16 mode: BatchNorm subtracts their mean, which would cancel bf16's 8 bits).
template <typename T> __device__ __forceinline__ float ldv(const View& v, long long off, int f32) {
  return f32 ? reinterpret_cast<const float*>(v.ptr)[off] : to_f32(reinterpret_cast<const T*>(v.ptr)[off]);
}
template <typename T> __device__ __forceinline__ void stv(const View& v, long long off, int f32, float val) {
  if (f32) reinterpret_cast<float*>(v.ptr)[off] = val; else reinterpret_cast<T*>(v.ptr)[off] = from_f32<T>(val);
}
__device__ __forceinline__ long long voff(const View& v, int n, int y, int x) { return n * v.sn + y * v.sh + x * v.sw; }
// flat pixel index -> (n, y, x); 32-bit divisions when the index fits (64-bit division is a long VALU routine)
__device__ __forceinline__ void pix3(long long pix, int h, int w, int& n, int& y, int& x) {
  if (pix <= 0x7fffffffll) {
    const unsigned p = (unsigned)pix, row = p / (unsigned)w;
    x = (int)(p - row * (unsigned)w);
    const unsigned img = row / (unsigned)h;
    y = (int)(row - img * (unsigned)h);
    n = (int)img;
  } else {
    x = (int)(pix % w); pix /= w;
    y = (int)(pix % h);
    n = (int)(pix / h);
  }
}
// flat element index -> (pixel, channel)
__device__ __forceinline__ long long split_c(long long idx, int C, int& c) {
  if (idx <= 0x7fffffffll) { const unsigned i = (unsigned)idx, p = i / (unsigned)C; c = (int)(i - p * (unsigned)C); return p; }
  c = (int)(idx % C);
  return idx / C;
}

struct Tf {            // input transform of the producer layer: act(scale*x + shift)
  const float* scale; const float* shift; const float* slope;   // slope: PReLU weights (mode 2)
  int mode;            // 0 none, 1 affine, 2 affine + PReLU, 3 affine + ReLU
};
__device__ __forceinline__ float tf_apply(const Tf& t, int c, float v) {
  if (t.mode == 0) return v;
  float z = fmaf(t.scale[c], v, t.shift[c]);
  if (t.mode == 2) z = z > 0.f ? z : z * t.slope[c];
  else if (t.mode == 3) z = fmaxf(z, 0.f);
  return z;
}
static inline Tf to_tf(const dct_enet_tf* t) {
  Tf r; r.scale = nullptr; r.shift = nullptr; r.slope = nullptr; r.mode = 0;
  if (t) { r.scale = t->scale; r.shift = t->shift; r.slope = t->slope; r.mode = t->mode; }
  return r;
}

// ---- per-channel partial sums -> per-channel results: the fold and the three "finalize" bodies ------------------------------
// Fixed-order fold of the per-block partials: thread (c, part) sums blocks part, part+NP, ... and the NP
// partial sums of a channel are then added in ascending `part` -- deterministic, and blockDim/CP-way parallel
// instead of one thread walking all blocks.  Result valid for threads with part == 0.
constexpr int FT = 1024;      // threads of the one-block finalize kernels: with 256 a 128-channel fold walked 128 partial rows per
                              // thread (11 us of dependent loads, 16 % of a cfg4 step's kernel time); 1024 threads walk 32
__device__ __forceinline__ void fold_partials(const double* partial, int blocks, int C, double* red, double s[3]) {
  int CP = 1;
  while (CP < C) CP <<= 1;
  const int NP = (int)blockDim.x / CP;        // blockDim.x = g_enet_fold_threads (256 ... FT), or the producing kernel's block
  const int c = threadIdx.x % CP, part = threadIdx.x / CP;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  if (c < C && part < NP) {
    // eight rows per round trip (the rolled loop is load -> wait -> add per row: one L2 latency per partial row).  The adds keep
    // their order, so the sums are bit for bit those of the rolled loop.
    int b = part;
    for (; b + 7 * NP < blocks; b += 8 * NP) {
      double q[8][3];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double* p = partial + ((long long)(b + u * NP) * C + c) * 3;
        q[u][0] = *(p); q[u][1] = *(p + 1); q[u][2] = *(p + 2);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a0 += q[u][0]; a1 += q[u][1]; a2 += q[u][2]; }
    }
    for (; b < blocks; b += NP) {
      const double* q = partial + ((long long)b * C + c) * 3;
      a0 += *(q); a1 += *(q + 1); a2 += *(q + 2);
    }
  }
  if (threadIdx.x < NP * CP) { red[threadIdx.x * 3] = a0; red[threadIdx.x * 3 + 1] = a1; red[threadIdx.x * 3 + 2] = a2; }
  __syncthreads();
  s[0] = s[1] = s[2] = 0.0;
  if (part == 0 && c < C)
    for (int k = 0; k < NP; ++k) { s[0] += red[(k * CP + c) * 3]; s[1] += red[(k * CP + c) * 3 + 1]; s[2] += red[(k * CP + c) * 3 + 2]; }
}

// BatchNorm forward finalize: statistics -> scale/shift (+ running statistics)
struct FinP {
  const double* partial; int blocks, C; double count;
  const float* gamma; const float* beta; float eps, momentum;
  float* running_mean; float* running_var; int training;
  float* scale; float* shift; float* save_mean; float* save_invstd; float* save_var;
};
// red: >= blockDim.x * 3 doubles of LDS
__device__ __forceinline__ void bn_fin_body(const FinP& p, double* red) {
  double s[3];
  fold_partials(p.partial, p.training ? p.blocks : 0, p.C, red, s);
  const int c = threadIdx.x;
  int CP = 1;
  while (CP < p.C) CP <<= 1;
  if (c >= p.C || threadIdx.x >= CP) return;
  float mean, var;
  if (p.training) {
    const double m = s[0] / p.count;
    double v = s[1] / p.count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m; var = (float)v;
    const double unbiased = p.count > 1.0 ? v * p.count / (p.count - 1.0) : v;
    if (p.running_mean) {
      p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * mean;
      p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unbiased;
    }
    if (p.save_var) p.save_var[c] = (float)unbiased;       // what a deferred running-statistics update needs
  } else {
    mean = p.running_mean[c]; var = p.running_var[c];
  }
  const float invstd = 1.0f / sqrtf(var + p.eps);
  const float sc = p.gamma[c] * invstd;
  p.scale[c] = sc;
  p.shift[c] = p.beta[c] - mean * sc;
  if (p.save_mean) { p.save_mean[c] = mean; p.save_invstd[c] = invstd; }
}
struct BnFinK {
  using Args = FinP;
  static constexpr int THREADS = FT;
  static __device__ __forceinline__ void run(const Args& p) {
    __shared__ double red[FT * 3];
    bn_fin_body(p, red);
  }
};

// BatchNorm backward finalize: dgamma/dbeta/dslope (+=) and the two per-channel means the apply pass needs
struct BFinP {
  const double* partial; int blocks, C; double count; int training;
  float* dgamma; float* dbeta; float* dslope; float* c1; float* c2;
};
__device__ __forceinline__ void bn_bwd_fin_body(const BFinP& p, double* red) {
  double s[3];
  fold_partials(p.partial, p.blocks, p.C, red, s);
  const int c = threadIdx.x;
  int CP = 1;
  while (CP < p.C) CP <<= 1;
  if (c >= p.C || threadIdx.x >= CP) return;
  if (p.dbeta) p.dbeta[c] += (float)s[0];
  if (p.dgamma) p.dgamma[c] += (float)s[1];
  if (p.dslope) p.dslope[c] += (float)s[2];
  // eval mode: mean / invstd are constants (running statistics), so no correction terms
  p.c1[c] = p.training ? (float)(s[0] / p.count) : 0.f;
  p.c2[c] = p.training ? (float)(s[1] / p.count) : 0.f;
}
struct BnBwdFinK {
  using Args = BFinP;
  static constexpr int THREADS = FT;
  static __device__ __forceinline__ void run(const Args& p) {
    __shared__ double red[FT * 3];
    bn_bwd_fin_body(p, red);
  }
};

// plain per-channel sum finalize (bias gradient): out[c] += sum_b partial[b][c][0]
struct SFinP { const double* partial; int blocks, C; float* out; };
__device__ __forceinline__ void sum_fin_body(const SFinP& p, double* red) {
  double s[3];
  fold_partials(p.partial, p.blocks, p.C, red, s);
  const int c = threadIdx.x;
  int CP = 1;
  while (CP < p.C) CP <<= 1;
  if (c >= p.C || threadIdx.x >= CP) return;
  p.out[c] += (float)s[0];
}
struct SumFinK {
  using Args = SFinP;
  static constexpr int THREADS = FT;
  static __device__ __forceinline__ void run(const Args& p) {
    __shared__ double red[FT * 3];
    sum_fin_body(p, red);
  }
};

// BatchNorm backward, one element: draw = scale (dz - c1 - xhat c2), dz = g act'(z), z = scale raw + shift, xhat = (raw - mean) invstd.
// ONE definition for the apply kernels and for the data-gradient convolution that applies it on load (ConvP::bwd_in): the two must
// agree to the bit, a convolution reading the stored draw and one computing it on load see the same operand.
__device__ __forceinline__ float bn_bwd_draw(float v, float g, float sc, float sh, float sl, float mu, float is, float c1, float c2, int act) {
  const float z = fmaf(sc, v, sh);
  float dz = g;
  if (act == 2) { if (!(z > 0.f)) dz = g * sl; }
  else if (act == 3) { if (!(z > 0.f)) dz = 0.f; }
  const float xh = (v - mu) * is;
  return sc * (dz - c1 - xh * c2);
}

struct ConvP {
  View x, y, rg, rm;              // input, output, residual grad + its ReLU mask (optional)
  const float* w; const float* bias;
  Tf tf;
  int R, S, stride, dil, pad_h, pad_w;
  int transposed, accumulate, has_resid;
  int ws_out, ws_tap, ws_in;      // weight strides (elements): W(o, tap, i)
  int G;                          // output-channel groups of 8
  int fm;                         // f32 mask: bit0 x, bit1 y, bit2 resid grad, bit3 resid mask
  int vec;                        // input rows can be read 8 channels at a time
  int wvec;          // MFMA form: weight rows are K-major and 16-byte aligned (one 32-byte load per B fragment)
  double* stats;     // MFMA form, optional: per-tile channel sums {sum, sum of squares, 0} of the stored outputs, [pixel tile][Cout][3]
  // MFMA form, optional (data gradients): the outputs are g = d/d act(BN(braw)) of the producing layer; per-tile BatchNorm-backward
  // sums {sum dz, sum dz xhat, sum g z [z<0]} (enet_reduce kind 1) go to stats instead
  View braw; const float* bscale; const float* bshift; const float* bslope; const float* bmean; const float* binvstd; int bact; int bn_bwd;
  int ngroups;       // MFMA form: output-channel groups of 32 NT (blockIdx.x = pixel tile * ngroups + group)
  // MFMA form, optional ("de-normalise on load"): the input IS the BatchNorm-backward result of a layer -- x = that layer's raw fp32
  // output, tf = its scale / shift / slope, ig = the gradient wrt its activation (im: ReLU mask of ig, optional) -- and a lane
  // computes draw = bn_bwd_draw(...) for its 8 channels where it would load them: the elementwise apply launch leaves the chain
  int bwd_in, i_act, i_has_mask;
  View ig, im;
  const float* i_mean; const float* i_invstd; const float* i_c1; const float* i_c2;
};

// One thread = one output pixel x 8 output channels.  Weights live in LDS as [tap][i][G*8].
template <typename T> struct ConvK {
  using Args = ConvP;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(const Args& p) {
  extern __shared__ float Ws[];
  const int Cin = p.x.c, Cout = p.y.c, taps = p.R * p.S, CO = p.G * 8;
  for (int e = threadIdx.x; e < taps * Cin * CO; e += 256) {
    const int o = e % CO, i = (e / CO) % Cin, t = e / (CO * Cin);
    Ws[e] = o < Cout ? p.w[(long long)o * p.ws_out + t * p.ws_tap + i * p.ws_in] : 0.f;
  }
  __syncthreads();
  const int g = threadIdx.x % p.G;
  const long long pix = (long long)blockIdx.x * (256 / p.G) + threadIdx.x / p.G;
  const long long P = (long long)p.y.n * p.y.h * p.y.w;
  if (pix >= P) return;
  int n, oy, ox;
  pix3(pix, p.y.h, p.y.w, n, oy, ox);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const int xf = p.fm & 1;
  for (int r = 0; r < p.R; ++r) {
    int iy;
    if (p.transposed) {
      const int ty = oy + p.pad_h - r * p.dil;
      if (ty < 0 || ty % p.stride) continue;
      iy = ty / p.stride;
    } else {
      iy = oy * p.stride - p.pad_h + r * p.dil;
    }
    if ((unsigned)iy >= (unsigned)p.x.h) continue;
    for (int s = 0; s < p.S; ++s) {
      int ix;
      if (p.transposed) {
        const int tx = ox + p.pad_w - s * p.dil;
        if (tx < 0 || tx % p.stride) continue;
        ix = tx / p.stride;
      } else {
        ix = ox * p.stride - p.pad_w + s * p.dil;
      }
      if ((unsigned)ix >= (unsigned)p.x.w) continue;
      const long long xo = voff(p.x, n, iy, ix);
      const float* wt = Ws + (r * p.S + s) * Cin * CO + g * 8;
      auto mac = [&](int i, float raw) {
        const float v = tf_apply(p.tf, i, raw);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + i * CO);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(wt + i * CO + 4);
        acc[0] = fmaf(v, w0[0], acc[0]); acc[1] = fmaf(v, w0[1], acc[1]);
        acc[2] = fmaf(v, w0[2], acc[2]); acc[3] = fmaf(v, w0[3], acc[3]);
        acc[4] = fmaf(v, w1[0], acc[4]); acc[5] = fmaf(v, w1[1], acc[5]);
        acc[6] = fmaf(v, w1[2], acc[6]); acc[7] = fmaf(v, w1[3], acc[7]);
      };
      if (p.vec) {      // Cin % 8 == 0 and 16-byte aligned rows: one (bf16) or two (fp32) 16-B loads per 8 channels
        for (int i = 0; i < Cin; i += 8) {
          float v8[8];
          if (xf || sizeof(T) == 4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.x.ptr) + xo + i);
            const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.x.ptr) + xo + i + 4);
            v8[0] = a[0]; v8[1] = a[1]; v8[2] = a[2]; v8[3] = a[3]; v8[4] = b[0]; v8[5] = b[1]; v8[6] = b[2]; v8[7] = b[3];
          } else {
            typedef typename vec8_of<T>::type V8;
            const V8 a = *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(p.x.ptr) + xo + i);
#pragma unroll
            for (int k = 0; k < 8; ++k) v8[k] = (float)a[k];
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) mac(i + k, v8[k]);
        }
      } else {
        for (int i = 0; i < Cin; ++i) mac(i, ldv<T>(p.x, xo + i, xf));
      }
    }
  }
  const long long yo = n * p.y.sn + oy * p.y.sh + ox * p.y.sw;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = g * 8 + k;
    if (c >= Cout) break;
    float v = acc[k];
    if (p.bias) v += p.bias[c];
    if (p.has_resid) {
      if (ldv<T>(p.rm, voff(p.rm, n, oy, ox) + c, p.fm & 8) > 0.f) v += ldv<T>(p.rg, voff(p.rg, n, oy, ox) + c, p.fm & 4);
    }
    if (p.accumulate) v += ldv<T>(p.y, yo + c, p.fm & 2);
    stv<T>(p.y, yo + c, p.fm & 2, v);
  }
}
};

// 8 consecutive channels of one pixel as one (T) or two (fp32) 16-byte loads
template <typename T> __device__ __forceinline__ void ld8(const View& v, long long off, int f32, float o[8]) {
  if (f32 || sizeof(T) == 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(v.ptr) + off);
    const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(v.ptr) + off + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
  } else {
    typedef typename vec8_of<T>::type V8;
    const V8 a = *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(v.ptr) + off);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)a[i];
  }
}

// ---- MFMA form of the small-channel convolution (bf16 / f16 compute modes) -----------------------
// The VALU kernel above gives one thread a whole reduction (taps x Cin multiply-adds in sequence: 288 for a 3x3 on 32 channels)
// and a stage-3 tensor (8 x 25 x 25 pixels) only ~80 blocks: 50 us per launch at 1.7 TFLOP/s, latency-bound (tools/
// bench_enet_layers.py).  Where the contraction is MFMA-shaped -- Cin a multiple of 8 and taps x Cin a multiple of 16, i.e.
// every convolution of stages 1-3 -- a wave instead owns 32 output pixels x all output channels and walks K = taps x Cin in
// steps of 16 on v_mfma_f32_32x32x16_{bf16,f16}: lane (r, h) loads the 8 consecutive input channels [c0 + 8h, +8) of ITS pixel
// at the tap's offset straight from HBM/L2 (NHWC: one 16- or 32-byte load; the producer's BatchNorm + activation is applied on
// load as before, out-of-image taps are zeros), which IS its A fragment; the B fragment is 8 consecutive K of weight row
// o = 32 j + r (one 32-byte load where the weights are K-major for this role, eight coalesced scalar loads otherwise).  No
// LDS, no barrier.  The operands are rounded to the compute dtype (mixed precision, fp32 accumulate); fp32 mode keeps the
// VALU kernel.  D[pixel][channel]: a lane holds one output channel of 16 pixels, so 32 lanes store 128 contiguous bytes.
template <typename T> struct LowMfma;
template <> struct LowMfma<bf16_t> {
  typedef bf16x8 frag;
  __device__ static __forceinline__ f32x16 run(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct LowMfma<f16_t> {
  typedef f16x8 frag;
  __device__ static __forceinline__ f32x16 run(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct LowMfma<float> {      // never launched (fp32 mode stays on the VALU kernel); keeps ENET_T instantiable
  typedef f16x8 frag;
  __device__ static __forceinline__ f32x16 run(frag, frag, f32x16 c) { return c; }
};

// Straight-line loads.  These kernels are a chain of dependent memory round trips (~1 us each on this chip), so what matters is
// how many loads are in flight per trip: the storage type of every view is a template flag (a run-time `f32 ? a : b` per
// load is a branch per load), out-of-range taps load element 0 and are zeroed afterwards, and the K loop is unrolled by U
// steps whose loads are all issued before the first MFMA.
template <typename T, bool F32> __device__ __forceinline__ float ld1(const void* ptr, long long off) {
  if constexpr (F32 || sizeof(T) == 4) return reinterpret_cast<const float*>(ptr)[off];
  else return to_f32(reinterpret_cast<const T*>(ptr)[off]);
}
template <typename T, bool F32> __device__ __forceinline__ void ld8t(const void* ptr, long long off, float o[8]) {
  if constexpr (F32 || sizeof(T) == 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(ptr) + off);
    const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(ptr) + off + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
  } else {
    typedef typename vec8_of<T>::type V8;
    const V8 a = *reinterpret_cast<const V8*>(reinterpret_cast<const T*>(ptr) + off);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)a[i];
  }
}
__device__ __forceinline__ void pix3u(unsigned pix, int h, int w, int& n, int& y, int& x) {
  const unsigned row = pix / (unsigned)w;
  x = (int)(pix - row * (unsigned)w);
  const unsigned img = row / (unsigned)h;
  y = (int)(row - img * (unsigned)h);
  n = (int)img;
}

// One block = 4 waves on the SAME 32 pixels x 32 NT channels, each wave taking every fourth K-step (a 3x3 on 32 channels is 18
// steps: <= 5 per wave, all in flight after one or two round trips instead of 18 in sequence); the four partial tiles meet in
// LDS, are added in wave order, and each wave stores 8 of the 32 pixel rows.  The producer's BatchNorm + activation sits in LDS
// as (scale, shift, negative-side slope) per input channel -- identity (1, 0, 1) without a transform, slope 0 for ReLU, 1 for the
// affine form -- so applying it is branch-free.
#ifndef DCT_MC_W
#define DCT_MC_W 4
#endif
constexpr int MC_W = DCT_MC_W;     // waves per block = K-split factor (8: the round-5 A/B build, profiles/r05_enet_k_split_8_waves_ab.txt)
constexpr int MC_E = 16 / MC_W;    // accumulators (pixel rows per half-wave) a wave stores after the fold
__device__ __forceinline__ int mc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }      // MFMA 32x32 D layout: row of accumulator e
constexpr int MC_U = 1;            // K-steps of one wave whose loads are issued together.  3 is the fastest ALONE (3x3 32->32: 6.6 us)
                                   // and the slowest in the step: with four hardware queues busy, cfg4 / cfg5 ms per step for
                                   // U = 4 / 3 / 2 / 1 read 18.1 / 17.4 / 16.9 / 16.4 and 44.6 / 41.8 / 39.9 / 38.2 -- registers and loads
                                   // in flight per wave are what the concurrent kernels compete for

template <typename T, int NT, bool XF, bool WV, bool BI = false>
__device__ __forceinline__ void mconv_main(const ConvP& p, const float* tfs, int cbase, bool pvalid, int n, int oy, int ox, int r, int h,
                                           int wave, f32x16 (&acc)[NT]) {
  typedef typename LowMfma<T>::frag frag;
  const int Cin = p.x.c, Cout = p.y.c;
  const int taps = p.R * p.S, nsteps = taps * (Cin >> 4);
  for (int s0 = wave; s0 < nsteps; s0 += MC_W * MC_U) {
    float a8[MC_U][8], w8[MC_U][NT][8];
    float g8[BI ? MC_U : 1][8], m8[BI ? MC_U : 1][8];
    bool v[MC_U];
    int cis[MC_U];
#pragma unroll
    for (int u = 0; u < MC_U; ++u) {
      const int sreal = s0 + MC_W * u;
      const int sidx = min(sreal, nsteps - 1);
      const int cblk = sidx / taps, tap = sidx - cblk * taps;
      const int rr = tap / p.S, ss = tap - rr * p.S;
      const int ci = cblk * 16 + 8 * h;
      cis[u] = ci;
      int iy, ix;
      bool ok = pvalid && sreal < nsteps;
      if (p.transposed) {
        const int ty = oy + p.pad_h - rr * p.dil, tx = ox + p.pad_w - ss * p.dil;
        ok = ok && ty >= 0 && tx >= 0 && (ty % p.stride) == 0 && (tx % p.stride) == 0;
        iy = ty / p.stride; ix = tx / p.stride;
      } else {
        iy = oy * p.stride - p.pad_h + rr * p.dil; ix = ox * p.stride - p.pad_w + ss * p.dil;
      }
      ok = ok && (unsigned)iy < (unsigned)p.x.h && (unsigned)ix < (unsigned)p.x.w;
      v[u] = ok;
      ld8t<T, XF>(p.x.ptr, ok ? voff(p.x, n, iy, ix) + ci : 0, a8[u]);
      if constexpr (BI) {
        ld8t<T, false>(p.ig.ptr, ok ? voff(p.ig, n, iy, ix) + ci : 0, g8[u]);
        if (p.i_has_mask) ld8t<T, false>(p.im.ptr, ok ? voff(p.im, n, iy, ix) + ci : 0, m8[u]);
      }
      const long long wtap = (long long)tap * p.ws_tap + (long long)ci * p.ws_in;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int o = cbase + 32 * j + r;
        const float* wp = p.w + (o < Cout ? (long long)o * p.ws_out + wtap : (long long)ci * p.ws_in);   // (a column past Cout is never stored)
        if constexpr (WV) {
          *reinterpret_cast<f32x4*>(w8[u][j]) = *reinterpret_cast<const f32x4*>(wp);
          *reinterpret_cast<f32x4*>(w8[u][j] + 4) = *reinterpret_cast<const f32x4*>(wp + 4);
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) w8[u][j][k] = wp[(long long)k * p.ws_in];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < MC_U; ++u) {
      float sc[8], sh[8], sl[8];
      *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(tfs + cis[u]);
      *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(tfs + cis[u] + 4);
      *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(tfs + 128 + cis[u]);
      *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(tfs + 128 + cis[u] + 4);
      *reinterpret_cast<f32x4*>(sl) = *reinterpret_cast<const f32x4*>(tfs + 256 + cis[u]);
      *reinterpret_cast<f32x4*>(sl + 4) = *reinterpret_cast<const f32x4*>(tfs + 256 + cis[u] + 4);
      frag A;
      if constexpr (BI) {
        // de-normalise on load: this lane's 8 channels of draw, rounded to T as the stored tensor would have been
        float mu[8], is[8], k1[8], k2[8];
        *reinterpret_cast<f32x4*>(mu) = *reinterpret_cast<const f32x4*>(tfs + 384 + cis[u]);
        *reinterpret_cast<f32x4*>(mu + 4) = *reinterpret_cast<const f32x4*>(tfs + 384 + cis[u] + 4);
        *reinterpret_cast<f32x4*>(is) = *reinterpret_cast<const f32x4*>(tfs + 512 + cis[u]);
        *reinterpret_cast<f32x4*>(is + 4) = *reinterpret_cast<const f32x4*>(tfs + 512 + cis[u] + 4);
        *reinterpret_cast<f32x4*>(k1) = *reinterpret_cast<const f32x4*>(tfs + 640 + cis[u]);
        *reinterpret_cast<f32x4*>(k1 + 4) = *reinterpret_cast<const f32x4*>(tfs + 640 + cis[u] + 4);
        *reinterpret_cast<f32x4*>(k2) = *reinterpret_cast<const f32x4*>(tfs + 768 + cis[u]);
        *reinterpret_cast<f32x4*>(k2 + 4) = *reinterpret_cast<const f32x4*>(tfs + 768 + cis[u] + 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float g = g8[u][k];
          if (p.i_has_mask && !(m8[u][k] > 0.f)) g = 0.f;
          const float t = bn_bwd_draw(a8[u][k], g, sc[k], sh[k], sl[k], mu[k], is[k], k1[k], k2[k], p.i_act);
          A[k] = from_f32<T>(v[u] ? t : 0.f);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float z = fmaf(sc[k], a8[u][k], sh[k]);
          const float t = z > 0.f ? z : z * sl[k];
          A[k] = from_f32<T>(v[u] ? t : 0.f);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        frag B;
#pragma unroll
        for (int k = 0; k < 8; ++k) B[k] = from_f32<T>(w8[u][j][k]);
        acc[j] = LowMfma<T>::run(A, B, acc[j]);
      }
    }
  }
}

template <typename T, int NT, bool YF>
__device__ __forceinline__ void mconv_store(const ConvP& p, const float* red, float* srd, int cbase, long long wbase, long long P, int r, int h,
                                            int wave) {
  // after the K-split fold this wave owns accumulators e = MC_E wave + i, i < MC_E = 16 / MC_W (pixel rows mc_row(e, h)), channel cbase + 32 j + r
  const int Cout = p.y.c;
  float val[MC_E][NT];
#pragma unroll
  for (int i = 0; i < MC_E; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < MC_W; ++w) t += red[((w * NT + j) * 16 + MC_E * wave + i) * 64 + 32 * h + r];   // fixed order
      val[i][j] = t;
    }
  long long yo[MC_E], rgo[MC_E], rmo[MC_E];
  bool pv[MC_E];
#pragma unroll
  for (int i = 0; i < MC_E; ++i) {
    const long long pix = wbase + mc_row(MC_E * wave + i, h);
    pv[i] = pix < P;
    int n, oy, ox;
    pix3u((unsigned)(pv[i] ? pix : 0), p.y.h, p.y.w, n, oy, ox);
    yo[i] = voff(p.y, n, oy, ox);
    rgo[i] = p.has_resid ? voff(p.rg, n, oy, ox) : 0;
    rmo[i] = p.has_resid ? voff(p.rm, n, oy, ox) : 0;
  }
  float add[MC_E][NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int c = cbase + 32 * j + r;
    const bool cv = c < Cout;
    const float b = (p.bias && cv) ? p.bias[c] : 0.f;
#pragma unroll
    for (int i = 0; i < MC_E; ++i) {
      const bool ok = pv[i] && cv;
      float t = b;
      if (p.has_resid) {       // (MFMA form: residual gradient and mask are stored as T -- host check)
        const float m = ld1<T, false>(p.rm.ptr, ok ? rmo[i] + c : 0), g = ld1<T, false>(p.rg.ptr, ok ? rgo[i] + c : 0);
        t += m > 0.f ? g : 0.f;
      }
      if (p.accumulate) t += ld1<T, YF>(p.y.ptr, ok ? yo[i] + c : 0);
      add[i][j] = t;
    }
  }
  float s1[NT], s2[NT], s3[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) s1[j] = s2[j] = s3[j] = 0.f;
  if (p.stats && p.bn_bwd) {
    // the stored value g (rounded to its storage type, as the BatchNorm-backward reduction would read it back) against the
    // producing layer's raw output: dz = g act'(z), xhat -- the reduction kernel's arithmetic, one tile of it
    float braw[MC_E][NT], bsc[NT], bsh[NT], bsl[NT], bmu[NT], bis[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int c = min(cbase + 32 * j + r, Cout - 1);
      bsc[j] = p.bscale[c]; bsh[j] = p.bshift[c]; bmu[j] = p.bmean[c]; bis[j] = p.binvstd[c]; bsl[j] = p.bact == 2 ? p.bslope[c] : 0.f;
#pragma unroll
      for (int i = 0; i < MC_E; ++i) {
        const long long pix = wbase + mc_row(MC_E * wave + i, h);
        int n, oy, ox;
        pix3u((unsigned)(pv[i] ? pix : 0), p.y.h, p.y.w, n, oy, ox);
        braw[i][j] = reinterpret_cast<const float*>(p.braw.ptr)[(pv[i] && cbase + 32 * j + r < Cout) ? voff(p.braw, n, oy, ox) + c : 0];
      }
    }
#pragma unroll
    for (int i = 0; i < MC_E; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c = cbase + 32 * j + r;
        if (pv[i] && c < Cout) {
          const float out = val[i][j] + add[i][j];
          float g;
          if constexpr (YF || sizeof(T) == 4) { reinterpret_cast<float*>(p.y.ptr)[yo[i] + c] = out; g = out; }
          else { const T q = from_f32<T>(out); reinterpret_cast<T*>(p.y.ptr)[yo[i] + c] = q; g = to_f32(q); }
          const float v = braw[i][j];
          const float z = fmaf(bsc[j], v, bsh[j]);
          float dz = g;
          if (p.bact == 2) { if (!(z > 0.f)) { dz = g * bsl[j]; s3[j] = fmaf(g, z, s3[j]); } }
          else if (p.bact == 3) { if (!(z > 0.f)) dz = 0.f; }
          const float xh = (v - bmu[j]) * bis[j];
          s1[j] += dz; s2[j] = fmaf(dz, xh, s2[j]);
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < MC_E; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c = cbase + 32 * j + r;
        if (pv[i] && c < Cout) {
          const float out = val[i][j] + add[i][j];
          if constexpr (YF || sizeof(T) == 4) reinterpret_cast<float*>(p.y.ptr)[yo[i] + c] = out;
          else reinterpret_cast<T*>(p.y.ptr)[yo[i] + c] = from_f32<T>(out);
          s1[j] += out; s2[j] = fmaf(out, out, s2[j]);
        }
      }
  }
  if (p.stats) {
    // BatchNorm statistics of this tile ride along: the consumer's reduction launch (a full read of the tensor and one more
    // seam on the forward chain) is replaced by 32 more partial rows... per-channel sums over the tile's 32 pixels -- 4 rows per
    // lane, the two half-waves by shuffle, the four waves through LDS in wave order -- as doubles in the reduction's layout
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      s1[j] += __shfl_xor(s1[j], 32, 64);
      s2[j] += __shfl_xor(s2[j], 32, 64);
      s3[j] += __shfl_xor(s3[j], 32, 64);
      if (h == 0) {
        srd[((wave * NT + j) * 32 + r) * 3] = s1[j]; srd[((wave * NT + j) * 32 + r) * 3 + 1] = s2[j]; srd[((wave * NT + j) * 32 + r) * 3 + 2] = s3[j];
      }
    }
    __syncthreads();
    if (wave == 0 && h == 0) {
      const long long row = wbase / 32;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c = cbase + 32 * j + r;
        if (c < Cout) {
          double a = 0.0, b = 0.0, d2 = 0.0;
#pragma unroll
          for (int w = 0; w < MC_W; ++w) {
            a += (double)srd[((w * NT + j) * 32 + r) * 3]; b += (double)srd[((w * NT + j) * 32 + r) * 3 + 1];
            d2 += (double)srd[((w * NT + j) * 32 + r) * 3 + 2];
          }
          double* o = p.stats + (row * Cout + c) * 3;
          o[0] = a; o[1] = b; o[2] = d2;
        }
      }
    }
  }
}

template <typename T, int NT> struct MconvK {
  using Args = ConvP;
  static constexpr int THREADS = 64 * MC_W;
  static __device__ __forceinline__ void run(const Args& p) {
    const int ngroups = p.ngroups;
  __shared__ __attribute__((aligned(16))) float tfs[7 * 128];      // scale, shift, slope [, mean, invstd, c1, c2: bwd_in]
  __shared__ __attribute__((aligned(16))) float red[MC_W * NT * 16 * 64];
  __shared__ float srd[MC_W * NT * 32 * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  for (int c = threadIdx.x; c < 128; c += 64 * MC_W) {
    const bool on = p.tf.mode != 0 && c < p.x.c;
    tfs[c] = on ? p.tf.scale[c] : 1.f;
    tfs[128 + c] = on ? p.tf.shift[c] : 0.f;
    tfs[256 + c] = (on && p.tf.mode == 2) ? p.tf.slope[c] : ((on && p.tf.mode == 3) ? 0.f : 1.f);
    if (p.bwd_in) {
      const bool in = c < p.x.c;
      tfs[384 + c] = in ? p.i_mean[c] : 0.f; tfs[512 + c] = in ? p.i_invstd[c] : 0.f;
      tfs[640 + c] = in ? p.i_c1[c] : 0.f; tfs[768 + c] = in ? p.i_c2[c] : 0.f;
    }
  }
  const long long P = (long long)p.y.n * p.y.h * p.y.w;                  // < 2^31 (host check)
  const int cbase = (int)(blockIdx.x % ngroups) * (32 * NT);             // first output channel of this block
  const long long wbase = (long long)(blockIdx.x / ngroups) * 32;        // first output pixel of this block
  const long long pix = wbase + r;                                       // this lane's A row
  const bool pvalid = pix < P;
  int n = 0, oy = 0, ox = 0;
  if (pvalid) pix3u((unsigned)pix, p.y.h, p.y.w, n, oy, ox);
  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  __syncthreads();
  if (p.bwd_in) {          // (x = raw fp32: host check)
    if (p.wvec) mconv_main<T, NT, true, true, true>(p, tfs, cbase, pvalid, n, oy, ox, r, h, wave, acc);
    else mconv_main<T, NT, true, false, true>(p, tfs, cbase, pvalid, n, oy, ox, r, h, wave, acc);
  } else if (p.fm & 1) {
    if (p.wvec) mconv_main<T, NT, true, true>(p, tfs, cbase, pvalid, n, oy, ox, r, h, wave, acc);
    else mconv_main<T, NT, true, false>(p, tfs, cbase, pvalid, n, oy, ox, r, h, wave, acc);
  } else {
    if (p.wvec) mconv_main<T, NT, false, true>(p, tfs, cbase, pvalid, n, oy, ox, r, h, wave, acc);
    else mconv_main<T, NT, false, false>(p, tfs, cbase, pvalid, n, oy, ox, r, h, wave, acc);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[((wave * NT + j) * 16 + e) * 64 + lane] = acc[j][e];
  __syncthreads();
  if (p.fm & 2) mconv_store<T, NT, true>(p, red, srd, cbase, wbase, P, r, h, wave);
  else mconv_store<T, NT, false>(p, red, srd, cbase, wbase, P, r, h, wave);
}
};

// ---- per-channel sums over pixels: partial[blk][c][k], k < NS, double accumulators -------------
// kind 0: {sum x, sum x^2}                                  (BatchNorm statistics; bias grad uses k = 0)
// kind 1: {sum dz, sum dz*xhat, sum g*z*[z<0]}              (BatchNorm / PReLU backward)
struct RedP {
  View x, g, m;                  // x: raw conv output; g: upstream grad; m: ReLU mask of g (optional)
  const float* scale; const float* shift; const float* slope; const float* mean; const float* invstd;
  int act;                       // activation after the BN: 0 none, 2 PReLU, 3 ReLU
  int has_mask, kind, ppb;
  int fm;                        // f32 mask: bit0 x (raw), bit1 g, bit2 g mask, bit3 draw
  double* partial;               // reductions: [block][C][3] partial sums
};

template <typename T>
__device__ __forceinline__ float grad_in(const RedP& p, int n, int y, int x, int c) {
  float g = ldv<T>(p.g, voff(p.g, n, y, x) + c, p.fm & 2);
  if (p.has_mask) {
    if (!(ldv<T>(p.m, voff(p.m, n, y, x) + c, p.fm & 4) > 0.f)) g = 0.f;
  }
  return g;
}

template <typename T> struct ReduceK {
  using Args = RedP;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(const Args& p) {
    double* partial = p.partial;
  __shared__ double red[256 * 3];
  const int C = p.x.c;
  int CP = 1;
  while (CP < C) CP <<= 1;                       // <= 128
  const int rows = 256 / CP;
  const int c = threadIdx.x % CP, row = threadIdx.x / CP;
  const long long P = (long long)p.x.n * p.x.h * p.x.w;
  const long long pbeg = (long long)blockIdx.x * p.ppb, pend = min(P, pbeg + p.ppb);
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  if (c < C) {
    float sc = 0.f, sh = 0.f, sl = 0.f, mu = 0.f, is = 0.f;
    if (p.kind == 1) { sc = p.scale[c]; sh = p.shift[c]; mu = p.mean[c]; is = p.invstd[c]; if (p.act == 2) sl = p.slope[c]; }
    for (long long pix = pbeg + row; pix < pend; pix += rows) {
      int n, y, x;
      pix3(pix, p.x.h, p.x.w, n, y, x);
      const float v = ldv<T>(p.x, voff(p.x, n, y, x) + c, p.fm & 1);
      if (p.kind == 0) {
        a0 += (double)v; a1 += (double)v * (double)v;
      } else {
        const float g = grad_in<T>(p, n, y, x, c);
        const float z = fmaf(sc, v, sh);
        float dz = g;
        if (p.act == 2) { if (!(z > 0.f)) { dz = g * sl; a2 += (double)(g * z); } }
        else if (p.act == 3) { if (!(z > 0.f)) dz = 0.f; }
        const float xh = (v - mu) * is;
        a0 += (double)dz; a1 += (double)dz * (double)xh;
      }
    }
  }
  red[threadIdx.x * 3 + 0] = a0; red[threadIdx.x * 3 + 1] = a1; red[threadIdx.x * 3 + 2] = a2;
  __syncthreads();
  if (row == 0 && c < C) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < rows; ++r) { s0 += red[(r * CP + c) * 3]; s1 += red[(r * CP + c) * 3 + 1]; s2 += red[(r * CP + c) * 3 + 2]; }
    double* o = partial + ((long long)blockIdx.x * C + c) * 3;
    o[0] = s0; o[1] = s1; o[2] = s2;
  }
}
};

// Same partial sums for channel counts that are multiples of 8 (Enet's 16 / 32 / 64 / 128-wide tensors: most of them): a thread
// owns 8 consecutive channels and loads them as one or two 16-byte vectors per pixel (the scalar kernel above issues one 2- or
// 4-byte load per element and a 64-bit-capable index decode per pixel); fp32 running sums over runs of 32 pixels are flushed
// into doubles, rows of threads are folded through LDS in a fixed order.  Same partial layout, same finalize kernels.
template <typename T> struct ReduceVecK {
  using Args = RedP;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(const Args& p) {
    double* partial = p.partial;
  extern __shared__ double redv[];                // [rows][C][3]
  const int C = p.x.c, CV = C / 8;                // CV in {2, 4, 8, 16}
  const int rows = 256 / CV;
  const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
  const long long P = (long long)p.x.n * p.x.h * p.x.w;
  const long long pbeg = (long long)blockIdx.x * p.ppb, pend = min(P, pbeg + p.ppb);
  float sc[8], sh[8], sl[8], mu[8], is[8];
  if (p.kind == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cv * 8 + i;
      sc[i] = p.scale[c]; sh[i] = p.shift[c]; mu[i] = p.mean[c]; is[i] = p.invstd[c]; sl[i] = p.act == 2 ? p.slope[c] : 0.f;
    }
  }
  float a0[8], a1[8], a2[8];
  double d0[8], d1[8], d2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a0[i] = a1[i] = a2[i] = 0.f; d0[i] = d1[i] = d2[i] = 0.0; }
  int run = 0;
  for (long long pix = pbeg + row; pix < pend; pix += rows) {
    int n, y, x;
    pix3(pix, p.x.h, p.x.w, n, y, x);
    float v[8];
    ld8<T>(p.x, voff(p.x, n, y, x) + cv * 8, p.fm & 1, v);
    if (p.kind == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { a0[i] += v[i]; a1[i] = fmaf(v[i], v[i], a1[i]); }
    } else {
      float g[8];
      ld8<T>(p.g, voff(p.g, n, y, x) + cv * 8, p.fm & 2, g);
      if (p.has_mask) {
        float m[8];
        ld8<T>(p.m, voff(p.m, n, y, x) + cv * 8, p.fm & 4, m);
#pragma unroll
        for (int i = 0; i < 8; ++i) if (!(m[i] > 0.f)) g[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float z = fmaf(sc[i], v[i], sh[i]);
        float dz = g[i];
        if (p.act == 2) { if (!(z > 0.f)) { dz = g[i] * sl[i]; a2[i] = fmaf(g[i], z, a2[i]); } }
        else if (p.act == 3) { if (!(z > 0.f)) dz = 0.f; }
        const float xh = (v[i] - mu[i]) * is[i];
        a0[i] += dz; a1[i] = fmaf(dz, xh, a1[i]);
      }
    }
    if (++run == 32) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { d0[i] += (double)a0[i]; d1[i] += (double)a1[i]; d2[i] += (double)a2[i]; a0[i] = a1[i] = a2[i] = 0.f; }
      run = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    double* o = redv + ((long long)row * C + cv * 8 + i) * 3;
    o[0] = d0[i] + (double)a0[i]; o[1] = d1[i] + (double)a1[i]; o[2] = d2[i] + (double)a2[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < rows; ++r) { const double* q = redv + ((long long)r * C + c) * 3; s0 += q[0]; s1 += q[1]; s2 += q[2]; }
    double* o = partial + ((long long)blockIdx.x * C + c) * 3;
    o[0] = s0; o[1] = s1; o[2] = s2;
  }
}
};

struct ApplyP { RedP p; const float* c1; const float* c2; View out; };
// draw = scale * (dz - c1 - xhat * c2)
template <typename T> struct BnApplyK {
  using Args = ApplyP;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(const Args& a) {
    const RedP& p = a.p;
    const float* c1 = a.c1; const float* c2 = a.c2;
    const View& out = a.out;

  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int C = p.x.c;
  const long long total = (long long)p.x.n * p.x.h * p.x.w * C;
  if (idx >= total) return;
  int c, n, y, x;
  pix3(split_c(idx, C, c), p.x.h, p.x.w, n, y, x);
  const float v = ldv<T>(p.x, voff(p.x, n, y, x) + c, p.fm & 1);
  const float g = grad_in<T>(p, n, y, x, c);
  const float r = bn_bwd_draw(v, g, p.scale[c], p.shift[c], p.act == 2 ? p.slope[c] : 0.f, p.mean[c], p.invstd[c], c1[c], c2[c], p.act);
  stv<T>(out, voff(out, n, y, x) + c, p.fm & 8, r);
}
};

// ---- bottleneck tail --------------------------------------------------------------------------
struct TailP {
  View raw, main, out, rawm;      // raw: last conv output; main: x (regular) / pre-pool x (down) / image (initial)
  Tf tf, tfm;                     // BN+act of raw; BN of rawm (up)
  unsigned char* idx;             // [N, h, w, Cm] pooling argmax code (down: written, up: read), dense
  int mode;                       // 0 regular, 1 down, 2 up, 3 initial
  int Cm;                         // channels of the pooled main branch (down) / of idx
  int fm;                         // f32 mask: bit0 raw, bit1 main, bit2 rawm, bit3 out
};

// first maximum of the 2x2 window in scan order (ATen's max_pool2d keeps the first; NaN wins)
template <typename T>
__device__ __forceinline__ float pool4(const View& v, int n, int oy, int ox, int c, int f32, int& code) {
  float best = -INFINITY;
  code = 0;
  for (int k = 0; k < 4; ++k) {
    const float t = ldv<T>(v, voff(v, n, 2 * oy + (k >> 1), 2 * ox + (k & 1)) + c, f32);
    if (t > best || t != t) { best = t; code = k; }
  }
  return best;
}

template <typename T> struct TailFwdK {
  using Args = TailP;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(const Args& p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int C = p.out.c;
  const long long total = (long long)p.out.n * p.out.h * p.out.w * C;
  if (i >= total) return;
  int c, n, oy, ox;
  pix3(split_c(i, C, c), p.out.h, p.out.w, n, oy, ox);
  const long long oo = voff(p.out, n, oy, ox) + c;
  int code;
  if (p.mode == 3) {   // initial block: 13 conv channels through BN+PReLU, channel 13 = maxpool(image)
    float r;
    if (c < p.raw.c) r = tf_apply(p.tf, c, ldv<T>(p.raw, voff(p.raw, n, oy, ox) + c, p.fm & 1));
    else r = pool4<T>(p.main, n, oy, ox, 0, p.fm & 2, code);
    stv<T>(p.out, oo, p.fm & 8, r);
    return;
  }
  const float ext = tf_apply(p.tf, c, ldv<T>(p.raw, voff(p.raw, n, oy, ox) + c, p.fm & 1));
  float mainv = 0.f;
  if (p.mode == 0) {
    mainv = ldv<T>(p.main, voff(p.main, n, oy, ox) + c, p.fm & 2);
  } else if (p.mode == 1) {
    if (c < p.Cm) {
      mainv = pool4<T>(p.main, n, oy, ox, c, p.fm & 2, code);
      p.idx[(((long long)n * p.out.h + oy) * p.out.w + ox) * p.Cm + c] = (unsigned char)code;
    }
  } else {   // up: max_unpool of bn(rawm) at half resolution
    const int sy = oy >> 1, sx = ox >> 1;
    if (p.idx[(((long long)n * p.rawm.h + sy) * p.rawm.w + sx) * p.Cm + c] == (oy & 1) * 2 + (ox & 1))
      mainv = tf_apply(p.tfm, c, ldv<T>(p.rawm, voff(p.rawm, n, sy, sx) + c, p.fm & 4));
  }
  stv<T>(p.out, oo, p.fm & 8, fmaxf(mainv + ext, 0.f));
}
};

// backward of the main branch of down / up / initial blocks
struct TailBP {
  View dout, out, dst;            // upstream grad, ReLU mask (mode 3: the input image), destination
  const unsigned char* idx;
  int mode;                       // 1 down: dst = dx at double resolution (zero where not argmax)
                                  // 2 up:   dst = grad of bn(rawm) at half resolution (gather)
                                  // 3 initial: dst [N,H,W,1] (+)= routed grad of channel Cm of dout (no mask)
  int Cm, accumulate;
  int fm;                         // f32 mask: bit0 dout, bit1 out, bit2 dst
};

template <typename T> struct TailBwdK {
  using Args = TailBP;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(const Args& p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int C = p.dst.c;
  const long long total = (long long)p.dst.n * p.dst.h * p.dst.w * C;
  if (i >= total) return;
  int c, n, y, x;
  pix3(split_c(i, C, c), p.dst.h, p.dst.w, n, y, x);
  const long long dof = voff(p.dst, n, y, x) + c;
  float g = 0.f;
  if (p.mode == 1) {          // dst = x-resolution; (y, x) in the 2x2 window of (y/2, x/2)
    const int oy = y >> 1, ox = x >> 1;
    if (c < p.Cm && p.idx[(((long long)n * p.dout.h + oy) * p.dout.w + ox) * p.Cm + c] == (y & 1) * 2 + (x & 1)) {
      if (ldv<T>(p.out, voff(p.out, n, oy, ox) + c, p.fm & 2) > 0.f) g = ldv<T>(p.dout, voff(p.dout, n, oy, ox) + c, p.fm & 1);
    }
  } else if (p.mode == 2) {   // dst = half resolution; pick the unpooled position
    const int code = p.idx[(((long long)n * p.dst.h + y) * p.dst.w + x) * p.Cm + c];
    const int oy = 2 * y + (code >> 1), ox = 2 * x + (code & 1);
    if (ldv<T>(p.out, voff(p.out, n, oy, ox) + c, p.fm & 2) > 0.f) g = ldv<T>(p.dout, voff(p.dout, n, oy, ox) + c, p.fm & 1);
  } else {                    // initial: p.out = the input image
    const int oy = y >> 1, ox = x >> 1;
    int code;
    (void)pool4<T>(p.out, n, oy, ox, 0, p.fm & 2, code);
    if (code == (y & 1) * 2 + (x & 1)) g = ldv<T>(p.dout, voff(p.dout, n, oy, ox) + p.Cm, p.fm & 1);
  }
  if (p.accumulate) g += ldv<T>(p.dst, dof, p.fm & 4);
  stv<T>(p.dst, dof, p.fm & 4, g);
}
};

// ---- weight gradient: dW[o][tap][i] (physical index e, see ws_* strides of the forward) -------------
// A pixel p = (n, ay, ax) of view A pairs with pixel (ay*stride - pad + r*dil, ...) of view B.
// conv:  A = draw (Cout), B = layer input (Cin, transform tfb);   entry e = (o, tap, i) of [Cout][R][S][Cin]
// convT: A = layer input (Cin, transform tfa), B = dy (Cout);      entry e = (i_A, tap, o_B) of [Cin][R][S][Cout]
struct WgP {
  View a, b;
  Tf tfa, tfb;
  int R, S, stride, dil, pad_h, pad_w;
  int ppb;
  int fm;                         // f32 mask: bit0 a, bit1 b
  float* partial; int E, pps, mtiles, ntiles;      // slabs [slice][E]; MFMA form: pixels per slice, 32 x 32 tiles
};
constexpr int WG_MAX_BLOCKS = 1024;
constexpr int WG_PB = 32;          // pixels staged per round
constexpr int WG_TPT = 2;          // 4x8 register tiles per thread (<= 512 tiles: E <= 16384)

// Register-tiled: the [Ca] x [taps*Cb] gradient is cut into 4 x 8 tiles, a thread owns up to two of them
// and per staged pixel reads 4 + 8 operands (three 16-B LDS reads) for 32 FMAs.  LDS rows are padded to
// multiples of 4 / 8 floats and zero-filled, so ragged channel counts (3, 13, 14) need no branches.
// SL pixel slices: when the gradient has <= 256 / SL register tiles, SL thread groups each take 32 / SL of a round's pixels
// for every tile and their accumulators are added through LDS at the end -- otherwise most of the block idles in the FMA
// phase (a 16 x 16 x 9 gradient has 72 tiles).
template <typename T, int SL> struct WgradK {
  using Args = WgP;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(const Args& p) {
    float* partial = p.partial; const int E = p.E;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Ca = p.a.c, Cb = p.b.c, taps = p.R * p.S;
  const int kb = taps * Cb;
  const int CaP = (Ca + 3) & ~3, kbP = (kb + 7) & ~7;
  const int OT = CaP >> 2, KT = kbP >> 3, ntiles = OT * KT;
  float* As = sm;                             // [WG_PB][CaP]
  float* Bs = sm + WG_PB * CaP;               // [WG_PB][kbP]
  __shared__ int pixn[WG_PB], pixy[WG_PB], pixx[WG_PB];
  const long long P = (long long)p.a.n * p.a.h * p.a.w;
  const long long pbeg = (long long)blockIdx.x * p.ppb, pend = min(P, pbeg + p.ppb);
  constexpr int TPB = 256 / SL;               // tiles a slice group can own
  constexpr int NT = SL == 1 ? WG_TPT : 1;    // tiles per thread
  constexpr int QS = WG_PB / SL;              // pixels of a round per slice
  const int slice = SL == 1 ? 0 : (int)threadIdx.x / TPB, tl = SL == 1 ? (int)threadIdx.x : (int)threadIdx.x % TPB;
  int to[NT], tk[NT];
  float acc[NT][4][8];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = min(j * 256 + tl, ntiles - 1);
    to[j] = (t / KT) * 4; tk[j] = (t % KT) * 8;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[j][a][b] = 0.f;
  }
  for (int e = threadIdx.x; e < WG_PB * (CaP + kbP); e += 256) sm[e] = 0.f;     // the pads stay zero
  for (long long p0 = pbeg; p0 < pend; p0 += WG_PB) {
    __syncthreads();
    if (threadIdx.x < WG_PB) {
      int n = -1, y = 0, x = 0;
      if (p0 + threadIdx.x < pend) pix3(p0 + threadIdx.x, p.a.h, p.a.w, n, y, x);
      pixn[threadIdx.x] = n; pixy[threadIdx.x] = y; pixx[threadIdx.x] = x;
    }
    __syncthreads();
    // staging: 8 pixel slots x 32 channel lanes (no divisions, all of a round's loads in flight together)
    const int tq = threadIdx.x >> 5, tc = threadIdx.x & 31;
    for (int q = tq; q < WG_PB; q += 8) {
      const int n = pixn[q], y = pixy[q], x = pixx[q];
      for (int c = tc; c < Ca; c += 32)
        As[q * CaP + c] = n >= 0 ? tf_apply(p.tfa, c, ldv<T>(p.a, voff(p.a, n, y, x) + c, p.fm & 1)) : 0.f;
      for (int t = 0; t < taps; ++t) {
        const int by = y * p.stride - p.pad_h + (t / p.S) * p.dil, bx = x * p.stride - p.pad_w + (t % p.S) * p.dil;
        const bool ok = n >= 0 && (unsigned)by < (unsigned)p.b.h && (unsigned)bx < (unsigned)p.b.w;
        for (int c = tc; c < Cb; c += 32)
          Bs[q * kbP + t * Cb + c] = ok ? tf_apply(p.tfb, c, ldv<T>(p.b, voff(p.b, n, by, bx) + c, p.fm & 2)) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (SL > 1 ? tl < ntiles : j * 256 < ntiles) {
#pragma unroll 8
        for (int qq = 0; qq < QS; ++qq) {
          const int q = slice * QS + qq;
          const f32x4 av = *reinterpret_cast<const f32x4*>(As + q * CaP + to[j]);
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + q * kbP + tk[j]);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs + q * kbP + tk[j] + 4);
#pragma unroll
          for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              acc[j][a][b] = fmaf(av[a], b0[b], acc[j][a][b]);
              acc[j][a][b + 4] = fmaf(av[a], b1[b], acc[j][a][b + 4]);
            }
          }
        }
      }
    }
  }
  if constexpr (SL > 1) {
    // fold the slices in fixed order through LDS: [slice][tile][32]
    __syncthreads();
    float* fold = sm;
    if (slice > 0 && tl < ntiles) {
      float* d = fold + ((slice - 1) * TPB + tl) * 32;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) d[a * 8 + b] = acc[0][a][b];
    }
    __syncthreads();
    if (slice == 0 && tl < ntiles) {
#pragma unroll 1
      for (int s2 = 1; s2 < SL; ++s2) {
        const float* d = fold + ((s2 - 1) * TPB + tl) * 32;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 8; ++b) acc[0][a][b] += d[a * 8 + b];
      }
    }
    if (slice != 0) return;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = j * 256 + tl;
    if (t < ntiles) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int o = to[j] + a;
        if (o >= Ca) break;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int k = tk[j] + b;
          if (k < kb) partial[(long long)blockIdx.x * E + o * kb + k] = acc[j][a][b];
        }
      }
    }
  }
}
};

// MFMA form of the weight gradient (bf16 / f16 compute modes).  dW[o][k] = sum over pixels of A[pixel][o] * B[pixel + tap(k)][c(k)]
// is a GEMM whose reduction runs over PIXELS, the strided dimension of both NHWC operands, so a v_mfma_f32_32x32x16 fragment
// (8 consecutive K of one row) is 8 pixels of one channel: eight 2- or 4-byte loads per lane, which the 32 lanes of a half-wave
// issue for 32 adjacent channels (64 - 128 contiguous bytes per load instruction).  One wave = one 32 x 32 tile of dW over one
// slice of the pixels; a stage-3 gradient (5000 pixels, 32 x 288) becomes ~180 independent waves of 16 MFMA steps instead of 78
// blocks of two barrier-separated rounds.  Lane r of the B fragment owns column k = 32 nt + r, i.e. its own (tap, channel): ragged
// channel counts and taps need no special case.  The slices are folded in fixed order by enet_wgrad_reduce_kernel as before.
// The layer input is rounded to the compute dtype exactly as the forward's MFMA form rounds it; the gradient operand is stored
// in that dtype already.
template <typename T, bool AF, bool BF>
__device__ __forceinline__ void mwgrad_body(const WgP& p, float* partial, int E, int pps, int mtiles, int ntiles) {
  typedef typename LowMfma<T>::frag frag;
  constexpr int U = 1;                                   // MFMA steps (16 pixels each) whose 16 U loads are in flight together (as MC_U:
                                                         // 4 is fastest alone, 1 in the step -- cfg5 38.2 -> 36.8 ms)
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int Ca = p.a.c, Cb = p.b.c, kb = p.R * p.S * Cb;
  int u0 = blockIdx.x;
  const int nt = u0 % ntiles; u0 /= ntiles;
  const int mt = u0 % mtiles;
  const int split = u0 / mtiles;
  const long long P = (long long)p.a.n * p.a.h * p.a.w;  // < 2^31 (host check)
  const long long pbeg = (long long)split * pps, pend = min(P, pbeg + pps);
  const int o = mt * 32 + r;
  const bool ov = o < Ca;
  const int k = nt * 32 + r;
  const bool kv = k < kb;
  const int tap = kv ? k / Cb : 0, c = kv ? k % Cb : 0;
  const int dy = (tap / p.S) * p.dil - p.pad_h, dx = (tap % p.S) * p.dil - p.pad_w;
  const int amode = ov ? p.tfa.mode : 0, bmode = kv ? p.tfb.mode : 0;
  float asc = 1.f, ash = 0.f, asl = 1.f, bsc = 1.f, bsh = 0.f, bsl = 1.f;      // negative-side slope: PReLU's, 0 for ReLU, 1 = none
  if (amode) { asc = p.tfa.scale[o]; ash = p.tfa.shift[o]; asl = amode == 2 ? p.tfa.slope[o] : (amode == 3 ? 0.f : 1.f); }
  if (bmode) { bsc = p.tfb.scale[c]; bsh = p.tfb.shift[c]; bsl = bmode == 2 ? p.tfb.slope[c] : (bmode == 3 ? 0.f : 1.f); }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // Offsets are 32-bit element counts (host check: every view spans < 2^31 elements) advanced by ADDITION from pixel to
  // pixel: a 64-bit n * sn + y * sh + x * sw per pixel and operand is ~12 quarter-rate multiplies, 64 times per iteration --
  // it was a large part of this kernel's instruction stream.  (With four K-steps of loads in flight this form was 10 % faster alone and
  // 10 % SLOWER in the cfg5 step; with one step in flight it is faster in the step too: 36.8 -> 35.3 ms.)
  const int a_sw = (int)p.a.sw, a_drow = (int)p.a.sh - p.a.w * (int)p.a.sw, a_dimg = (int)p.a.sn - p.a.h * (int)p.a.sh;
  const int b_sw = p.stride * (int)p.b.sw, b_drow = p.stride * (int)p.b.sh - p.a.w * b_sw, b_dimg = (int)p.b.sn - p.a.h * p.stride * (int)p.b.sh;
  for (long long p0 = pbeg; p0 < pend; p0 += 16 * U) {
    float av[U][8], bv[U][8];
    unsigned am = 0, bm = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long q = p0 + 16 * u + 8 * h;
      int n, y, x;
      pix3u((unsigned)min(q, P - 1), p.a.h, p.a.w, n, y, x);
      int ao = n * (int)p.a.sn + y * (int)p.a.sh + x * (int)p.a.sw + o;
      int by = y * p.stride + dy, bx = x * p.stride + dx;
      int bo = n * (int)p.b.sn + by * (int)p.b.sh + bx * (int)p.b.sw + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool pv = q + j < pend;
        const bool va = pv && ov;
        av[u][j] = ld1<T, AF>(p.a.ptr, va ? ao : 0);
        const bool vb = pv && kv && (unsigned)by < (unsigned)p.b.h && (unsigned)bx < (unsigned)p.b.w;
        bv[u][j] = ld1<T, BF>(p.b.ptr, vb ? bo : 0);
        am |= (unsigned)va << (u * 8 + j);
        bm |= (unsigned)vb << (u * 8 + j);
        ++x;
        ao += a_sw; bo += b_sw; bx += p.stride;
        const bool wx = x == p.a.w;
        x = wx ? 0 : x;
        bx = wx ? dx : bx;
        ao += wx ? a_drow : 0; bo += wx ? b_drow : 0;
        y += wx ? 1 : 0; by += wx ? p.stride : 0;
        const bool wy = y == p.a.h;
        y = wy ? 0 : y;
        by = wy ? dy : by;
        ao += wy ? a_dimg : 0; bo += wy ? b_dimg : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      frag A, B;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = av[u][j], b = bv[u][j];
        { const float z = fmaf(asc, a, ash); a = z > 0.f ? z : z * asl; }      // identity (1, 0, 1) without a transform
        { const float z = fmaf(bsc, b, bsh); b = z > 0.f ? z : z * bsl; }
        A[j] = from_f32<T>(((am >> (u * 8 + j)) & 1u) ? a : 0.f);
        B[j] = from_f32<T>(((bm >> (u * 8 + j)) & 1u) ? b : 0.f);
      }
      acc = LowMfma<T>::run(A, B, acc);
    }
  }
  if (kv) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int oo = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (oo < Ca) partial[(long long)split * E + (long long)oo * kb + k] = acc[e];
    }
  }
}

template <typename T> struct MwgradK {
  using Args = WgP;
  static constexpr int THREADS = 64;
  static __device__ __forceinline__ void run(const Args& p) {
    float* partial = p.partial; const int E = p.E, pps = p.pps, mtiles = p.mtiles, ntiles = p.ntiles;
  if (p.fm & 1) {
    if (p.fm & 2) mwgrad_body<T, true, true>(p, partial, E, pps, mtiles, ntiles);
    else mwgrad_body<T, true, false>(p, partial, E, pps, mtiles, ntiles);
  } else {
    if (p.fm & 2) mwgrad_body<T, false, true>(p, partial, E, pps, mtiles, ntiles);
    else mwgrad_body<T, false, false>(p, partial, E, pps, mtiles, ntiles);
  }
}
};

// dw[e] += sum_b partial[b][e]: 16 entries per block, 16 strided partial sums each, folded in fixed order
struct WRedP { const float* partial; float* dw; int E, blocks; };
struct WgradRedK {
  using Args = WRedP;
  static constexpr int THREADS = 256;
  static __device__ __forceinline__ void run(const Args& p) {
    const auto partial = p.partial;
    const auto dw = p.dw;
    const auto E = p.E;
    const auto blocks = p.blocks;

  __shared__ float red[256];
  const int e = blockIdx.x * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
  float s = 0.f;
  if (e < E) {
    int b = part;
    for (; b + 7 * 16 < blocks; b += 8 * 16) {       // eight slabs per round trip, adds in the rolled loop's order (see fold_partials)
      float q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = partial[(long long)(b + u * 16) * E + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += q[u];
    }
    for (; b < blocks; b += 16) s += partial[(long long)b * E + e];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (part == 0 && e < E) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k * 16 + threadIdx.x];
    dw[e] += t;
  }
}
};

static inline bool ok_dtype(int d) { return d == DCT_F32 || d == DCT_BF16 || d == DCT_F16; }
static inline int red_plan(long long P, int C, int& ppb) {
  int CP = 1;
  while (CP < C) CP <<= 1;
  const int rows = 256 / CP;
  // g_enet_reduce_ppt pixels per thread.  Swept on cfg4 (graph replay): 4 / 8 / 16 / 32 / 64 -> 50.0 / 49.8 / 50.5 / 53.9 /
  // 65.4 ms per step: the reductions want parallelism, the longer fold of the single-block finalize costs less
  long long blocks = (P + (long long)g_enet_reduce_ppt * rows - 1) / ((long long)g_enet_reduce_ppt * rows);
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  ppb = (int)((P + blocks - 1) / blocks);
  return (int)((P + ppb - 1) / ppb);
}

}  // namespace

#define ENET_T(dtype, ...) do { if ((dtype) == DCT_BF16) { using T = bf16_t; __VA_ARGS__; } else if ((dtype) == DCT_F16) { using T = f16_t; __VA_ARGS__; } \
                               else { using T = float; __VA_ARGS__; } } while (0)

static int enet_conv_impl(const dct_view* x, const float* w, const float* bias, const dct_enet_tf* tf,
                          const dct_view* y, const dct_conv_desc* d, int transposed,
                          int ws_out, int ws_tap, int ws_in,
                          const dct_view* resid_grad, const dct_view* resid_mask,
                          int f32_mask, int dtype, double* stats_partial, int stats_capacity_rows, int* stats_rows, dct_stream stream,
                          const dct_view* bn_raw = nullptr, const float* bn_scale = nullptr, const float* bn_shift = nullptr,
                          const float* bn_slope = nullptr, int bn_act = 0, const float* bn_mean = nullptr, const float* bn_invstd = nullptr,
                          const dct_enet_bwd_in* bin = nullptr) {
  if (stats_rows) *stats_rows = 0;
  if (!view_ok(x) || !view_ok(y) || !w || !d || !ok_dtype(dtype) || x->n != y->n) return DCT_ERR_BAD_ARG;
  if (d->R < 1 || d->S < 1 || d->stride < 1 || d->dil < 1) return DCT_ERR_BAD_ARG;
  if (y->c > 128 || x->c > 128) return DCT_ERR_UNSUPPORTED;
  ConvP p;
  p.x = to_view(x); p.y = to_view(y);
  p.w = w; p.bias = bias; p.tf = to_tf(tf);
  p.R = d->R; p.S = d->S; p.stride = d->stride; p.dil = d->dil; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
  p.transposed = transposed ? 1 : 0; p.accumulate = d->accumulate ? 1 : 0;
  p.ws_out = ws_out; p.ws_tap = ws_tap; p.ws_in = ws_in;
  p.has_resid = 0; p.wvec = 0; p.stats = nullptr; p.bn_bwd = 0; p.bact = 0; p.braw = p.y;
  p.ngroups = 0;
  p.bwd_in = 0; p.i_act = 0; p.i_has_mask = 0; p.ig = p.x; p.im = p.x;
  p.i_mean = p.i_invstd = p.i_c1 = p.i_c2 = nullptr;
  if (bin) {
    // x is the raw fp32 output of a BatchNorm'd layer, tf its transform; the kernel computes that layer's BatchNorm-backward
    // result on load from (x, bin->g [, bin->g_mask]) -- MFMA form only
    if (!tf || !tf->mode || !(f32_mask & 1) || !view_ok(bin->g) || !bin->mean || !bin->invstd || !bin->c1c2) return DCT_ERR_BAD_ARG;
    if (bin->g->n != x->n || bin->g->h != x->h || bin->g->w != x->w || bin->g->c != x->c) return DCT_ERR_BAD_ARG;
    if (bin->g_mask && (!view_ok(bin->g_mask) || bin->g_mask->n != x->n || bin->g_mask->h != x->h || bin->g_mask->w != x->w ||
                        bin->g_mask->c != x->c)) return DCT_ERR_BAD_ARG;
    auto v8 = [&](const dct_view* v) {     // 16-bit storage, 8 channels per 16-byte load
      return v->c % 8 == 0 && v->sw % 8 == 0 && v->sh % 8 == 0 && v->sn % 8 == 0 && ((uintptr_t)v->ptr % 16) == 0 &&
             (long long)v->n * v->sn < 0x7fffffffLL;
    };
    if (dtype == DCT_F32 || !v8(bin->g) || (bin->g_mask && !v8(bin->g_mask)) || (((uintptr_t)bin->mean | (uintptr_t)bin->invstd |
        (uintptr_t)bin->c1c2) % 16) || (x->c % 4)) return DCT_ERR_UNSUPPORTED;
    p.bwd_in = 1; p.i_act = (tf->mode == 2 || tf->mode == 3) ? tf->mode : 0;
    p.ig = to_view(bin->g);
    if (bin->g_mask) { p.im = to_view(bin->g_mask); p.i_has_mask = 1; }
    p.i_mean = bin->mean; p.i_invstd = bin->invstd; p.i_c1 = bin->c1c2; p.i_c2 = bin->c1c2 + x->c;
  }
  p.bscale = p.bshift = p.bslope = p.bmean = p.binvstd = nullptr;
  p.rg = p.y; p.rm = p.y;
  if (resid_grad) {
    if (!view_ok(resid_grad) || !view_ok(resid_mask)) return DCT_ERR_BAD_ARG;
    p.rg = to_view(resid_grad); p.rm = to_view(resid_mask); p.has_resid = 1;
  }
  int G = (y->c + 7) / 8;
  int Gp = 1;
  while (Gp < G) Gp <<= 1;                  // 1, 2, 4, 8, 16: divides 256
  p.G = Gp;
  p.fm = f32_mask;
  {
    const int xbytes = ((f32_mask & 1) || dtype == DCT_F32) ? 4 : 2;   // bf16 and f16 are both 2 bytes
    p.vec = (x->c % 8 == 0 && x->sw % 8 == 0 && x->sh % 8 == 0 && x->sn % 8 == 0 && ((uintptr_t)x->ptr % (8 * xbytes)) == 0) ? 1 : 0;
  }
  hipStream_t st0 = (hipStream_t)stream;
  if ((g_enet_mfma & 1) && dtype != DCT_F32 && p.vec && x->c >= 16 && x->c % 16 == 0 && y->c <= 128 && (long long)y->n * y->h * y->w < 0x7fffffffLL &&
      (!tf || tf->mode != 2 || tf->slope) && (!resid_grad || (f32_mask & 12) == 0) &&
      (!tf || !tf->mode || (((uintptr_t)tf->scale | (uintptr_t)tf->shift) % 16 == 0))) {
    p.wvec = (ws_in == 1 && (uintptr_t)w % 16 == 0 && ws_out % 4 == 0 && ws_tap % 4 == 0) ? 1 : 0;
    // one wave (= one 64-thread block) per 32 pixels x 32 NT channels; NT > 1 only where the pixel tiles alone fill the chip
    const long long Pm = (long long)y->n * y->h * y->w;
    const long long ptiles = (Pm + 31) / 32;
    const int ntiles = (y->c + 31) / 32;
    if (stats_partial && stats_rows && ptiles <= stats_capacity_rows && !d->accumulate) {
      bool ok = true;
      if (bn_raw) {       // BatchNorm-backward sums: the producing layer's raw fp32 output, same pixels and channels as y
        ok = view_ok(bn_raw) && bn_raw->n == y->n && bn_raw->h == y->h && bn_raw->w == y->w && bn_raw->c == y->c && bn_scale && bn_shift &&
             bn_mean && bn_invstd && (bn_act != 2 || bn_slope) &&
             (long long)bn_raw->n * bn_raw->sn + (long long)bn_raw->h * bn_raw->sh + (long long)bn_raw->w * bn_raw->sw < 0x7fffffffLL;
        if (ok) {
          p.bn_bwd = 1; p.braw = to_view(bn_raw); p.bscale = bn_scale; p.bshift = bn_shift; p.bslope = bn_slope; p.bmean = bn_mean;
          p.binvstd = bn_invstd; p.bact = bn_act;
        }
      }
      if (ok) { p.stats = stats_partial; *stats_rows = (int)ptiles; }
    }
    int nt = 1;
    if (MC_W <= 4 && ntiles >= 4 && ptiles >= 2048) nt = 4;
    else if (ntiles >= 2 && ptiles * ((ntiles + 1) / 2) >= 2048) nt = 2;
    const int ngroups = (ntiles + nt - 1) / nt;
    if (ptiles * ngroups > 0x7fffffffLL) return DCT_ERR_UNSUPPORTED;
    const unsigned gridm = (unsigned)(ptiles * ngroups);
    p.ngroups = ngroups;
    if (nt == 1) ENET_T(dtype, (enet_launch<MconvK<T, 1>>(DCT_PROF_OTHER, dim3(gridm), dim3(64 * MC_W), 0, st0, p)));
    else if (nt == 2) ENET_T(dtype, (enet_launch<MconvK<T, 2>>(DCT_PROF_OTHER, dim3(gridm), dim3(64 * MC_W), 0, st0, p)));
    else if constexpr (MC_W <= 4) ENET_T(dtype, (enet_launch<MconvK<T, 4>>(DCT_PROF_OTHER, dim3(gridm), dim3(64 * MC_W), 0, st0, p)));
    return dct_check_launch();
  }
  if (bin) return DCT_ERR_UNSUPPORTED;        // (the VALU kernel has no de-normalise-on-load form: the caller materialises draw)
  const size_t lds = (size_t)d->R * d->S * x->c * Gp * 8 * sizeof(float);
  if (lds > 64 * 1024) return DCT_ERR_UNSUPPORTED;
  const long long P = (long long)y->n * y->h * y->w;
  const unsigned grid = div_up(P, 256 / Gp);
  hipStream_t st = (hipStream_t)stream;
  p.ngroups = 0;
  ENET_T(dtype, enet_launch<ConvK<T>>(DCT_PROF_OTHER, dim3(grid), dim3(256), lds, st, p));
  return dct_check_launch();
}

extern "C" int dct_enet_conv(const dct_view* x, const float* w, const float* bias, const dct_enet_tf* tf,
                             const dct_view* y, const dct_conv_desc* d, int transposed,
                             int ws_out, int ws_tap, int ws_in,
                             const dct_view* resid_grad, const dct_view* resid_mask,
                             int f32_mask, int dtype, dct_stream stream) {
  return enet_conv_impl(x, w, bias, tf, y, d, transposed, ws_out, ws_tap, ws_in, resid_grad, resid_mask, f32_mask, dtype, nullptr, 0, nullptr,
                        stream);
}

extern "C" int dct_enet_conv_bnbwd_stats(const dct_view* x, const float* w, const dct_view* y, const dct_conv_desc* d, int transposed,
                                         int ws_out, int ws_tap, int ws_in, int f32_mask, int dtype,
                                         const dct_view* bn_raw, const float* bn_scale, const float* bn_shift, const float* bn_slope, int bn_act,
                                         const float* bn_mean, const float* bn_invstd,
                                         double* stats_partial, int stats_capacity_rows, int* stats_rows, dct_stream stream) {
  if (!stats_partial || !stats_rows || stats_capacity_rows < 1 || !bn_raw) return DCT_ERR_BAD_ARG;
  return enet_conv_impl(x, w, nullptr, nullptr, y, d, transposed, ws_out, ws_tap, ws_in, nullptr, nullptr, f32_mask, dtype, stats_partial,
                        stats_capacity_rows, stats_rows, stream, bn_raw, bn_scale, bn_shift, bn_slope, bn_act, bn_mean, bn_invstd);
}

extern "C" int dct_enet_conv_stats(const dct_view* x, const float* w, const float* bias, const dct_enet_tf* tf,
                                   const dct_view* y, const dct_conv_desc* d, int transposed,
                                   int ws_out, int ws_tap, int ws_in, int f32_mask, int dtype,
                                   double* stats_partial, int stats_capacity_rows, int* stats_rows, dct_stream stream) {
  if (!stats_partial || !stats_rows || stats_capacity_rows < 1) return DCT_ERR_BAD_ARG;
  return enet_conv_impl(x, w, bias, tf, y, d, transposed, ws_out, ws_tap, ws_in, nullptr, nullptr, f32_mask, dtype, stats_partial,
                        stats_capacity_rows, stats_rows, stream);
}

extern "C" int dct_enet_conv_bwd_in(const dct_view* raw, const float* w, const dct_enet_tf* tf, const dct_enet_bwd_in* bin,
                                    const dct_view* y, const dct_conv_desc* d, int transposed, int ws_out, int ws_tap, int ws_in,
                                    const dct_view* resid_grad, const dct_view* resid_mask, int f32_mask, int dtype,
                                    const dct_view* bn_raw, const float* bn_scale, const float* bn_shift, const float* bn_slope, int bn_act,
                                    const float* bn_mean, const float* bn_invstd,
                                    double* stats_partial, int stats_capacity_rows, int* stats_rows, dct_stream stream) {
  if (!bin) return DCT_ERR_BAD_ARG;
  if (bn_raw && (!stats_partial || !stats_rows || stats_capacity_rows < 1)) return DCT_ERR_BAD_ARG;
  return enet_conv_impl(raw, w, nullptr, tf, y, d, transposed, ws_out, ws_tap, ws_in, resid_grad, resid_mask, f32_mask, dtype,
                        bn_raw ? stats_partial : nullptr, bn_raw ? stats_capacity_rows : 0, bn_raw ? stats_rows : nullptr, stream,
                        bn_raw, bn_scale, bn_shift, bn_slope, bn_act, bn_mean, bn_invstd, bin);
}

extern "C" size_t dct_enet_reduce_workspace_bytes(int channels) {
  return (size_t)256 * (channels > 0 ? channels : 1) * 3 * sizeof(double);
}


static int enet_reduce_launch(const RedP& p0, int dtype, void* workspace, size_t workspace_bytes, hipStream_t st, int& blocks_out) {
  RedP p = p0;
  const long long P = (long long)p.x.n * p.x.h * p.x.w;
  int ppb;
  const int blocks = red_plan(P, p.x.c, ppb);
  if (!workspace || workspace_bytes < (size_t)blocks * p.x.c * 3 * sizeof(double)) return DCT_ERR_WORKSPACE;
  p.ppb = ppb;
  // 8-channel vector path: every view it reads has 8-aligned strides and a 16-byte (T) / 32-byte (fp32) aligned base
  auto v8 = [&](const View& v, int f32) {
    const int esz = (f32 || dtype == DCT_F32) ? 4 : 2;
    return v.c % 8 == 0 && v.sw % 8 == 0 && v.sh % 8 == 0 && v.sn % 8 == 0 && ((uintptr_t)v.ptr % (8 * esz)) == 0;
  };
  const int C = p.x.c;
  bool vec = C >= 16 && C <= 128 && (C & (C - 1)) == 0 && v8(p.x, p.fm & 1);
  if (vec && p.kind == 1) vec = v8(p.g, p.fm & 2) && (!p.has_mask || v8(p.m, p.fm & 4));
  if (vec) {
    const size_t lds = (size_t)(256 / (C / 8)) * C * 3 * sizeof(double);      // 48 KiB
    p.partial = (double*)workspace;
    ENET_T(dtype, enet_launch<ReduceVecK<T>>(DCT_PROF_OTHER, dim3(blocks), dim3(256), lds, st, p));
  } else {
    p.partial = (double*)workspace;
    ENET_T(dtype, enet_launch<ReduceK<T>>(DCT_PROF_OTHER, dim3(blocks), dim3(256), 0, st, p));
  }
  blocks_out = blocks;
  return DCT_OK;
}

extern "C" int dct_enet_bn_fwd_stats(const dct_view* raw, const float* gamma, const float* beta, float eps, float momentum,
                                     float* running_mean, float* running_var, int training,
                                     float* scale, float* shift, float* save_mean, float* save_invstd, float* save_var,
                                     int f32_mask, int dtype, void* workspace, size_t workspace_bytes, dct_stream stream) {
  return dct_enet_bn_fwd_stats_rows(raw, gamma, beta, eps, momentum, running_mean, running_var, training, scale, shift, save_mean, save_invstd,
                                    save_var, f32_mask, dtype, workspace, workspace_bytes, 0, stream);
}

extern "C" int dct_enet_bn_fwd_stats_rows(const dct_view* raw, const float* gamma, const float* beta, float eps, float momentum,
                                          float* running_mean, float* running_var, int training,
                                          float* scale, float* shift, float* save_mean, float* save_invstd, float* save_var,
                                          int f32_mask, int dtype, void* workspace, size_t workspace_bytes, int partial_rows,
                                          dct_stream stream) {
  if (!view_ok(raw) || !gamma || !beta || !scale || !shift || !ok_dtype(dtype) || raw->c > 128) return DCT_ERR_BAD_ARG;
  if (!training && (!running_mean || !running_var)) return DCT_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  int blocks = 0;
  if (training && partial_rows > 0) {       // the producing convolution wrote the partial rows (dct_enet_conv_stats)
    if (!workspace || workspace_bytes < (size_t)partial_rows * raw->c * 3 * sizeof(double)) return DCT_ERR_WORKSPACE;
    blocks = partial_rows;
  } else if (training) {
    RedP p; p.x = to_view(raw); p.g = p.x; p.m = p.x;
    p.scale = p.shift = p.slope = p.mean = p.invstd = nullptr;
    p.act = 0; p.has_mask = 0; p.kind = 0; p.ppb = 0; p.fm = f32_mask & 1; p.partial = nullptr;
    const int rc = enet_reduce_launch(p, dtype, workspace, workspace_bytes, st, blocks);
    if (rc != DCT_OK) return rc;
  }
  const double count = (double)raw->n * raw->h * raw->w;
  const FinP fp = {(const double*)workspace, blocks, raw->c, count, gamma, beta, eps, momentum, running_mean, running_var, training ? 1 : 0,
                   scale, shift, save_mean, save_invstd, save_var};
  enet_launch<BnFinK>(DCT_PROF_OTHER, dim3(1), dim3(g_enet_fold_threads), 0, st, fp);
  return dct_check_launch();
}

extern "C" int dct_enet_bn_bwd(const dct_view* raw, const dct_view* g, const dct_view* g_mask,
                               const float* scale, const float* shift, const float* slope, int act,
                               const float* mean, const float* invstd,
                               float* dgamma, float* dbeta, float* dslope, float* c1c2, int training,
                               const dct_view* draw, int f32_mask, int dtype, void* workspace, size_t workspace_bytes,
                               dct_stream stream) {
  return dct_enet_bn_bwd_rows(raw, g, g_mask, scale, shift, slope, act, mean, invstd, dgamma, dbeta, dslope, c1c2, training, draw, f32_mask,
                              dtype, workspace, workspace_bytes, 0, stream);
}

// do_sums: reduction (or the rows a convolution wrote) + finalize; do_apply: draw from (raw, g, c1c2)
static int enet_bn_bwd_impl(const dct_view* raw, const dct_view* g, const dct_view* g_mask,
                            const float* scale, const float* shift, const float* slope, int act,
                            const float* mean, const float* invstd,
                            float* dgamma, float* dbeta, float* dslope, float* c1c2, int training,
                            const dct_view* draw, int f32_mask, int dtype, void* workspace, size_t workspace_bytes,
                            int partial_rows, dct_stream stream, bool do_sums, bool do_apply) {
  if (!view_ok(raw) || !view_ok(g) || (do_apply && !view_ok(draw)) || !scale || !shift || !mean || !invstd || !c1c2 || !ok_dtype(dtype))
    return DCT_ERR_BAD_ARG;
  if (do_sums && partial_rows != 0 && g_mask) return DCT_ERR_BAD_ARG;     // (rows from a convolution's epilogue never carry a mask)
  if (!do_sums) partial_rows = -1;
  if (raw->c > 128 || (act == 2 && !slope)) return DCT_ERR_BAD_ARG;
  RedP p; p.x = to_view(raw); p.g = to_view(g); p.m = p.g;
  p.has_mask = 0;
  if (g_mask) { if (!view_ok(g_mask)) return DCT_ERR_BAD_ARG; p.m = to_view(g_mask); p.has_mask = 1; }
  p.scale = scale; p.shift = shift; p.slope = slope; p.mean = mean; p.invstd = invstd;
  p.act = act; p.kind = 1; p.ppb = 0; p.fm = f32_mask; p.partial = nullptr;
  hipStream_t st = (hipStream_t)stream;
  int blocks = 0;
  const bool finalized = partial_rows < 0;    // (apply only: c1c2 and the parameter gradients are set)
  const double count = (double)raw->n * raw->h * raw->w;
  if (partial_rows > 0) {         // the data-gradient convolution that produced g wrote the partial rows (dct_enet_conv_bnbwd_stats)
    if (!workspace || workspace_bytes < (size_t)partial_rows * raw->c * 3 * sizeof(double)) return DCT_ERR_WORKSPACE;
    blocks = partial_rows;
  } else if (!finalized) {
    const int rc = enet_reduce_launch(p, dtype, workspace, workspace_bytes, st, blocks);
    if (rc != DCT_OK) return rc;
  }
  const long long total = (long long)raw->n * raw->h * raw->w * raw->c;
  const View vo = do_apply ? to_view(draw) : p.x;
  if (!finalized) {
    const BFinP bp = {(const double*)workspace, blocks, raw->c, count, training ? 1 : 0, dgamma, dbeta, act == 2 ? dslope : nullptr, c1c2, c1c2 + raw->c};
    enet_launch<BnBwdFinK>(DCT_PROF_OTHER, dim3(1), dim3(g_enet_fold_threads), 0, st, bp);
  }
  if (!do_apply) return dct_check_launch();
  const ApplyP ap = {p, (const float*)c1c2, (const float*)(c1c2 + raw->c), vo};
  ENET_T(dtype, enet_launch<BnApplyK<T>>(DCT_PROF_OTHER, dim3(div_up(total, 256)), dim3(256), 0, st, ap));
  return dct_check_launch();
}

extern "C" int dct_enet_bn_bwd_rows(const dct_view* raw, const dct_view* g, const dct_view* g_mask,
                                    const float* scale, const float* shift, const float* slope, int act,
                                    const float* mean, const float* invstd,
                                    float* dgamma, float* dbeta, float* dslope, float* c1c2, int training,
                                    const dct_view* draw, int f32_mask, int dtype, void* workspace, size_t workspace_bytes,
                                    int partial_rows, dct_stream stream) {
  return enet_bn_bwd_impl(raw, g, g_mask, scale, shift, slope, act, mean, invstd, dgamma, dbeta, dslope, c1c2, training, draw, f32_mask,
                          dtype, workspace, workspace_bytes, partial_rows, stream, true, true);
}

extern "C" int dct_enet_bn_bwd_sums(const dct_view* raw, const dct_view* g, const dct_view* g_mask,
                                    const float* scale, const float* shift, const float* slope, int act,
                                    const float* mean, const float* invstd,
                                    float* dgamma, float* dbeta, float* dslope, float* c1c2, int training,
                                    int f32_mask, int dtype, void* workspace, size_t workspace_bytes, int partial_rows, dct_stream stream) {
  if (partial_rows < 0) return DCT_ERR_BAD_ARG;
  return enet_bn_bwd_impl(raw, g, g_mask, scale, shift, slope, act, mean, invstd, dgamma, dbeta, dslope, c1c2, training, nullptr, f32_mask,
                          dtype, workspace, workspace_bytes, partial_rows, stream, true, false);
}

extern "C" int dct_enet_bn_bwd_apply(const dct_view* raw, const dct_view* g, const dct_view* g_mask,
                                     const float* scale, const float* shift, const float* slope, int act,
                                     const float* mean, const float* invstd, const float* c1c2,
                                     const dct_view* draw, int f32_mask, int dtype, int leaf, dct_stream stream) {
  const bool was = g_leaf_scope;
  if (leaf) g_leaf_scope = true;
  const int rc = enet_bn_bwd_impl(raw, g, g_mask, scale, shift, slope, act, mean, invstd, nullptr, nullptr, nullptr, const_cast<float*>(c1c2),
                                  1, draw, f32_mask, dtype, nullptr, 0, -1, stream, false, true);
  g_leaf_scope = was;
  return rc;
}

extern "C" int dct_enet_channel_sum(const dct_view* x, float* out, int f32_mask, int dtype, void* workspace, size_t workspace_bytes,
                                    dct_stream stream) {
  if (!view_ok(x) || !out || !ok_dtype(dtype) || x->c > 128) return DCT_ERR_BAD_ARG;
  const LeafScope leaf;
  RedP p; p.x = to_view(x); p.g = p.x; p.m = p.x;
  p.scale = p.shift = p.slope = p.mean = p.invstd = nullptr;
  p.act = 0; p.has_mask = 0; p.kind = 0; p.ppb = 0; p.fm = f32_mask & 1; p.partial = nullptr;
  hipStream_t st = (hipStream_t)stream;
  int blocks = 0;
  const int rc = enet_reduce_launch(p, dtype, workspace, workspace_bytes, st, blocks);
  if (rc != DCT_OK) return rc;
  const SFinP sp = {(const double*)workspace, blocks, x->c, out};
  enet_launch<SumFinK>(DCT_PROF_OTHER, dim3(1), dim3(g_enet_fold_threads), 0, st, sp);
  return dct_check_launch();
}

extern "C" int dct_enet_tail_fwd(const dct_view* raw, const dct_enet_tf* tf, const dct_view* main_in,
                                 const dct_view* rawm, const dct_enet_tf* tfm, uint8_t* idx, int idx_channels,
                                 int mode, const dct_view* out, int f32_mask, int dtype, dct_stream stream) {
  if (!view_ok(raw) || !view_ok(out) || !ok_dtype(dtype) || mode < 0 || mode > 3) return DCT_ERR_BAD_ARG;
  TailP p;
  p.raw = to_view(raw); p.out = to_view(out); p.main = p.raw; p.rawm = p.raw;
  p.tf = to_tf(tf); p.tfm = to_tf(tfm); p.idx = idx; p.mode = mode; p.Cm = idx_channels; p.fm = f32_mask;
  if (mode == 0 || mode == 1 || mode == 3) { if (!view_ok(main_in)) return DCT_ERR_BAD_ARG; p.main = to_view(main_in); }
  if (mode == 1 && (!idx || main_in->h != 2 * out->h || main_in->w != 2 * out->w || idx_channels != main_in->c)) return DCT_ERR_BAD_ARG;
  if (mode == 2) {
    if (!view_ok(rawm) || !idx || rawm->h * 2 != out->h || rawm->w * 2 != out->w || idx_channels != out->c) return DCT_ERR_BAD_ARG;
    p.rawm = to_view(rawm);
  }
  if (mode == 3 && (main_in->c != 1 || main_in->h != 2 * out->h || out->c != raw->c + 1)) return DCT_ERR_BAD_ARG;
  const long long total = (long long)out->n * out->h * out->w * out->c;
  hipStream_t st = (hipStream_t)stream;
  ENET_T(dtype, enet_launch<TailFwdK<T>>(DCT_PROF_OTHER, dim3(div_up(total, 256)), dim3(256), 0, st, p));
  return dct_check_launch();
}

extern "C" int dct_enet_tail_bwd(const dct_view* dout, const dct_view* out_mask, const uint8_t* idx, int idx_channels,
                                 int mode, int accumulate, const dct_view* dst, int f32_mask, int dtype, dct_stream stream) {
  if (!view_ok(dout) || !view_ok(out_mask) || !view_ok(dst) || !ok_dtype(dtype) || mode < 1 || mode > 3) return DCT_ERR_BAD_ARG;
  if (mode != 3 && !idx) return DCT_ERR_BAD_ARG;
  TailBP p;
  p.dout = to_view(dout); p.out = to_view(out_mask); p.dst = to_view(dst);
  p.idx = idx; p.mode = mode; p.Cm = idx_channels; p.accumulate = accumulate ? 1 : 0; p.fm = f32_mask;
  const long long total = (long long)dst->n * dst->h * dst->w * dst->c;
  hipStream_t st = (hipStream_t)stream;
  ENET_T(dtype, enet_launch<TailBwdK<T>>(DCT_PROF_OTHER, dim3(div_up(total, 256)), dim3(256), 0, st, p));
  return dct_check_launch();
}

extern "C" size_t dct_enet_wgrad_workspace_bytes(const dct_view* a, const dct_view* b, const dct_conv_desc* d) {
  if (!a || !b || !d) return 0;
  return (size_t)WG_MAX_BLOCKS * a->c * d->R * d->S * b->c * sizeof(float);
}

extern "C" int dct_enet_wgrad(const dct_view* a, const dct_enet_tf* tfa, const dct_view* b, const dct_enet_tf* tfb,
                              float* dw, const dct_conv_desc* d, int f32_mask, int dtype,
                              void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (!view_ok(a) || !view_ok(b) || !dw || !d || !ok_dtype(dtype) || a->n != b->n) return DCT_ERR_BAD_ARG;
  const LeafScope leaf;
  const int E = a->c * d->R * d->S * b->c;
  if ((g_enet_mfma & 2) && dtype != DCT_F32) {
    WgP p;
    p.a = to_view(a); p.b = to_view(b); p.tfa = to_tf(tfa); p.tfb = to_tf(tfb);
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.dil = d->dil; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    p.fm = f32_mask; p.ppb = 0;
    const long long P = (long long)a->n * a->h * a->w;
    const int kb = d->R * d->S * b->c;
    const int mtiles = (a->c + 31) / 32, ntiles = (kb + 31) / 32;
    // pixel slices: enough waves to fill the chip (~2048), >= 4 MFMA steps (64 pixels) each, <= WG_MAX_BLOCKS slices of workspace
    const long long steps = (P + 15) / 16;
    long long ks = g_enet_mwgrad_waves / ((long long)mtiles * ntiles);
    if (ks < 1) ks = 1;
    long long sps = (steps + ks - 1) / ks;
    if (sps < g_enet_mwgrad_min_steps) sps = g_enet_mwgrad_min_steps;
    if ((steps + sps - 1) / sps > WG_MAX_BLOCKS) sps = (steps + WG_MAX_BLOCKS - 1) / WG_MAX_BLOCKS;
    sps = (sps + 3) / 4 * 4;                               // the kernel walks its slice four steps at a time
    auto span = [](const dct_view* v) { return (long long)v->n * v->sn + (long long)v->h * v->sh + (long long)v->w * v->sw + v->c; };
    if (P >= 0x7fffffffLL || span(a) >= 0x7fffffffLL || span(b) >= 0x7fffffffLL) return DCT_ERR_UNSUPPORTED;
    const long long pps = sps * 16;
    const long long nsl = (P + pps - 1) / pps;
    if (pps > 0x7fffffffLL || nsl * mtiles * ntiles > 0x7fffffffLL) return DCT_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < (size_t)nsl * E * sizeof(float)) return DCT_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    p.partial = (float*)workspace; p.E = E; p.pps = (int)pps; p.mtiles = mtiles; p.ntiles = ntiles;
    ENET_T(dtype, enet_launch<MwgradK<T>>(DCT_PROF_OTHER, dim3((unsigned)(nsl * mtiles * ntiles)), dim3(64), 0, st, p));
    const WRedP wr = {(const float*)workspace, dw, E, (int)nsl};
    enet_launch<WgradRedK>(DCT_PROF_OTHER, dim3(div_up(E, 16)), dim3(256), 0, st, wr);
    return dct_check_launch();
  }
  const int CaP = (a->c + 3) & ~3, kbP = (d->R * d->S * b->c + 7) & ~7;
  if ((CaP / 4) * (kbP / 8) > 256 * WG_TPT) return DCT_ERR_UNSUPPORTED;
  const int ntiles = (CaP / 4) * (kbP / 8);
  const int SL = !g_enet_wgrad_slices ? 1 : ntiles <= 64 ? 4 : ntiles <= 128 ? 2 : 1;       // pixel slices per round (see the kernel)
  size_t lds = (size_t)WG_PB * (CaP + kbP) * sizeof(float);
  if (SL > 1) lds = std::max(lds, (size_t)(SL - 1) * (256 / SL) * 32 * sizeof(float));
  if (lds > 64 * 1024) return DCT_ERR_UNSUPPORTED;
  WgP p;
  p.a = to_view(a); p.b = to_view(b); p.tfa = to_tf(tfa); p.tfb = to_tf(tfb);
  p.R = d->R; p.S = d->S; p.stride = d->stride; p.dil = d->dil; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
  p.fm = f32_mask;
  const long long P = (long long)a->n * a->h * a->w;
  // latency-bound (gather, transform, barrier, 32-pixel FMA round): several resident blocks per CU overlap each other's
  // phases; every block keeps >= 2 rounds so the zero fill and the partial-tile write stay amortised
  long long blocks = (P + 2 * WG_PB - 1) / (2 * WG_PB);
  if (blocks > g_enet_wgrad_max_blocks) blocks = g_enet_wgrad_max_blocks;
  if (blocks < 1) blocks = 1;
  long long ppb = (P + blocks - 1) / blocks;
  ppb = (ppb + WG_PB - 1) / WG_PB * WG_PB;
  p.ppb = (int)ppb;
  const int nb = (int)((P + ppb - 1) / ppb);
  if (!workspace || workspace_bytes < (size_t)nb * E * sizeof(float)) return DCT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  p.partial = (float*)workspace; p.E = E; p.pps = 0; p.mtiles = 0; p.ntiles = 0;
  if (SL == 4) ENET_T(dtype, (enet_launch<WgradK<T, 4>>(DCT_PROF_OTHER, dim3(nb), dim3(256), lds, st, p)));
  else if (SL == 2) ENET_T(dtype, (enet_launch<WgradK<T, 2>>(DCT_PROF_OTHER, dim3(nb), dim3(256), lds, st, p)));
  else ENET_T(dtype, (enet_launch<WgradK<T, 1>>(DCT_PROF_OTHER, dim3(nb), dim3(256), lds, st, p)));
  const WRedP wr = {(const float*)workspace, dw, E, nb};
  enet_launch<WgradRedK>(DCT_PROF_OTHER, dim3(div_up(E, 16)), dim3(256), 0, st, wr);
  return dct_check_launch();
}

// ---- grouped passes (see "launch plumbing" at the top) ----------------------------------------------------------------
extern "C" int dct_group_begin(int members) {
  if (members < 1 || members > GROUP_MAX || g_grp.active) return DCT_ERR_BAD_ARG;
  g_grp.recs.assign((size_t)members, std::vector<GroupRec>());
  g_grp.member = 0;
  g_grp.active = true;
  return DCT_OK;
}
// Side recording of LEAF launches: from dct_leaves_begin on, the launches of dct_enet_wgrad / dct_enet_channel_sum (weight and
// bias gradients: leaves of a backward pass's data-gradient chain) are held back while everything else launches as usual;
// dct_leaves_flush(stream) issues what has been held back so far on `stream` (another queue, after an event of the chain) and
// keeps recording, dct_leaves_end(stream) issues the rest and stops.  Operands must stay allocated until their flush has been
// issued AND `stream` has been joined.
extern "C" int dct_leaves_begin(void) {
  if (g_grp.active) return DCT_ERR_BAD_ARG;
  g_grp.recs.assign(1, std::vector<GroupRec>());
  g_grp.member = 0;
  g_grp.leaves_only = true;
  g_grp.active = true;
  return DCT_OK;
}
extern "C" int dct_leaves_flush(dct_stream stream, int* launches_out) {
  if (!g_grp.active || !g_grp.leaves_only) return DCT_ERR_BAD_ARG;
  std::vector<GroupRec> recs;
  recs.swap(g_grp.recs[0]);
  g_grp.active = false;            // (the launches below must go out)
  for (const GroupRec& a : recs) {
    const void* one[1] = {a.blob.data()};
    a.launch(one, 1, a.grid, a.block, a.lds, (hipStream_t)stream, a.cls);
  }
  g_grp.active = true;
  if (launches_out) *launches_out = (int)recs.size();
  return dct_check_launch();
}
extern "C" int dct_leaves_end(dct_stream stream, int* launches_out) {
  const int rc = dct_leaves_flush(stream, launches_out);
  g_grp.active = false;
  g_grp.leaves_only = false;
  g_grp.recs.clear();
  return rc;
}
extern "C" int dct_group_member(int member) {
  if (!g_grp.active || member < 0 || member >= (int)g_grp.recs.size()) return DCT_ERR_BAD_ARG;
  g_grp.member = member;
  return DCT_OK;
}
extern "C" int dct_group_max(void) { return GROUP_MAX; }
extern "C" int dct_group_abort(void) {
  g_grp.active = false;
  g_grp.leaves_only = false;
  g_grp.recs.clear();
  return DCT_OK;
}
// Launches everything recorded since dct_group_begin on `stream`: entry k of every member in ONE launch where the members agree
// on kernel, grid, block and LDS bytes (the same plan on the same shapes always does), otherwise member by member.
// *grouped_out / *single_out (nullable): how many launches of either kind were issued.
extern "C" int dct_group_end(dct_stream stream, int* grouped_out, int* single_out) {
  if (!g_grp.active || g_grp.leaves_only) return DCT_ERR_BAD_ARG;
  g_grp.active = false;
  std::vector<std::vector<GroupRec>> recs;
  recs.swap(g_grp.recs);
  hipStream_t st = (hipStream_t)stream;
  const int n = (int)recs.size();
  size_t longest = 0;
  bool same_len = true;
  for (const auto& r : recs) { longest = std::max(longest, r.size()); same_len = same_len && r.size() == recs[0].size(); }
  int grouped = 0, single = 0;
  for (size_t k = 0; k < longest; ++k) {
    bool same = same_len && n > 1;
    for (int m = 1; same && m < n; ++m) {
      const GroupRec& a = recs[0][k]; const GroupRec& b = recs[m][k];
      same = a.launch == b.launch && a.grid.x == b.grid.x && a.grid.y == b.grid.y && a.grid.z == b.grid.z && a.block.x == b.block.x &&
             a.lds == b.lds && a.blob.size() == b.blob.size();
    }
    if (same && recs[0][k].grid.z == 1) {
      const void* args[GROUP_MAX];
      for (int m = 0; m < n; ++m) args[m] = recs[m][k].blob.data();
      const GroupRec& a = recs[0][k];
      a.launch(args, n, a.grid, a.block, a.lds, st, a.cls);
      ++grouped;
    } else {
      for (int m = 0; m < n; ++m) {
        if (k >= recs[m].size()) continue;
        const GroupRec& a = recs[m][k];
        const void* one[1] = {a.blob.data()};
        a.launch(one, 1, a.grid, a.block, a.lds, st, a.cls);
        ++single;
      }
    }
  }
  if (grouped_out) *grouped_out = grouped;
  if (single_out) *single_out = single;
  return dct_check_launch();
}
