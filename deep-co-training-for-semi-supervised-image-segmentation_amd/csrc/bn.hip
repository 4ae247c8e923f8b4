// Wide-channel BatchNorm2d + ReLU for the BatchNorm'd UNet (`unet_bn`, arch/network.py:243-290 of the reference):
// nn.BatchNorm2d(C) with C = 64 .. 1024 between a valid 3x3 convolution and its ReLU.
//
// HBM-bound, so every pass moves whole 16-byte channel vectors (8 bf16 / 2 x 4 fp32) of NHWC rows:
//   forward : one statistics pass over the raw conv output (per-block partial sums in double, folded in a fixed
//             order: deterministic) -> scale = gamma * invstd, shift = beta - mean * scale (+ running statistics,
//             unbiased variance, momentum) -> one apply pass y = relu(scale * raw + shift);
//   backward: one reduction pass for sum(dz), sum(dz * xhat) with dz = g * [scale * raw + shift > 0] -> dgamma / dbeta and
//             the two batch means -> one apply pass draw = scale * (dz - mean(dz) - xhat * mean(dz * xhat)).
// The Enet kernels (enet.hip) do the same arithmetic for <= 128 channels with "normalise on load" consumers; the UNet
// convolutions stage their input straight from HBM into LDS (no transform on load), hence the explicit apply pass here.
#include "dct_common.h"

namespace {

constexpr int BN_MAX_BLOCKS = 256;

template <typename T> __device__ __forceinline__ void load8(const char* base, long long elem, float v[8]);
template <> __device__ __forceinline__ void load8<bf16_t>(const char* base, long long elem, float v[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(base) + elem);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
template <> __device__ __forceinline__ void load8<float>(const char* base, long long elem, float v[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem);
  const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <typename T> __device__ __forceinline__ void store8(char* base, long long elem, const float v[8]);
template <> __device__ __forceinline__ void store8<bf16_t>(char* base, long long elem, const float v[8]) {
  bf16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(base) + elem) = t;
}
template <> __device__ __forceinline__ void store8<float>(char* base, long long elem, const float v[8]) {
  *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + elem) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + elem + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

__device__ __forceinline__ long long pix_off(const View& v, long long pix) {
  const int x = (int)(pix % v.w);
  const long long r = pix / v.w;
  const int y = (int)(r % v.h);
  const int n = (int)(r / v.h);
  return n * v.sn + y * v.sh + x * v.sw;
}

struct BnP {
  View x, g;                       // raw conv output; upstream gradient wrt relu(bn(raw)) (backward only)
  const float* scale; const float* shift; const float* mean; const float* invstd;
  int ppb;                         // pixels per block
  int relu;
};

// partial[blk][c][2] (double): KIND 0 {sum x, sum x^2}; KIND 1 {sum dz, sum dz * xhat}
template <typename T, int KIND>
__global__ __launch_bounds__(256) void bn_reduce_kernel(BnP p, double* partial) {
  extern __shared__ double red[];                 // [rows][C][2]
  const int C = p.x.c, CV = C / 8;
  const int rows = 256 / CV;                      // CV divides 256 (host-checked: C in {64, 128, 256, 512, 1024, 2048})
  const int cv = threadIdx.x % CV, row = threadIdx.x / CV;
  const long long P = (long long)p.x.n * p.x.h * p.x.w;
  const long long pbeg = (long long)blockIdx.x * p.ppb, pend = min(P, pbeg + p.ppb);
  float a0[8], a1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
  float sc[8], sh[8], mu[8], is[8];
  if (KIND == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = p.scale[cv * 8 + i]; sh[i] = p.shift[cv * 8 + i]; mu[i] = p.mean[cv * 8 + i]; is[i] = p.invstd[cv * 8 + i]; }
  }
  // fp32 running sums over short runs (<= 64 pixels), flushed into doubles: ATen's CPU BatchNorm accumulates in double-ish
  // cascades; the parity tests hold mean / var to 1e-6
  double d0[8], d1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { d0[i] = 0.0; d1[i] = 0.0; }
  int run = 0;
  for (long long pix = pbeg + row; pix < pend; pix += rows) {
    float v[8];
    load8<T>(p.x.ptr, pix_off(p.x, pix) + cv * 8, v);
    if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { a0[i] += v[i]; a1[i] = fmaf(v[i], v[i], a1[i]); }
    } else {
      float g[8];
      load8<T>(p.g.ptr, pix_off(p.g, pix) + cv * 8, g);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float z = fmaf(sc[i], v[i], sh[i]);
        const float dz = (!p.relu || z > 0.f) ? g[i] : 0.f;
        const float xh = (v[i] - mu[i]) * is[i];
        a0[i] += dz; a1[i] = fmaf(dz, xh, a1[i]);
      }
    }
    if (++run == 64) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { d0[i] += (double)a0[i]; d1[i] += (double)a1[i]; a0[i] = 0.f; a1[i] = 0.f; }
      run = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { d0[i] += (double)a0[i]; d1[i] += (double)a1[i]; }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    red[((long long)row * C + cv * 8 + i) * 2 + 0] = d0[i];
    red[((long long)row * C + cv * 8 + i) * 2 + 1] = d1[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    double s0 = 0.0, s1 = 0.0;
    for (int r = 0; r < rows; ++r) { s0 += red[((long long)r * C + c) * 2]; s1 += red[((long long)r * C + c) * 2 + 1]; }
    double* o = partial + ((long long)blockIdx.x * C + c) * 2;
    o[0] = s0; o[1] = s1;
  }
}

__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const double* partial, int blocks, int C, double count,
                                                              const float* gamma, const float* beta, float eps, float momentum,
                                                              float* running_mean, float* running_var, int training,
                                                              float* scale, float* shift, float* save_mean, float* save_invstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < blocks; ++b) { s0 += partial[((long long)b * C + c) * 2]; s1 += partial[((long long)b * C + c) * 2 + 1]; }
    const double m = s0 / count;
    double v = s1 / count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m; var = (float)v;
    if (running_mean) {
      const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  } else {
    mean = running_mean[c]; var = running_var[c];
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  if (save_mean) { save_mean[c] = mean; save_invstd[c] = invstd; }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* partial, int blocks, int C, double count, int training,
                                                              int accumulate, float* dgamma, float* dbeta, float* c1, float* c2) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, s1 = 0.0;
  for (int b = 0; b < blocks; ++b) { s0 += partial[((long long)b * C + c) * 2]; s1 += partial[((long long)b * C + c) * 2 + 1]; }
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s0;
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s1;
  c1[c] = training ? (float)(s0 / count) : 0.f;     // eval mode: the statistics are constants, no correction terms
  c2[c] = training ? (float)(s1 / count) : 0.f;
}

// y = relu?(scale * x + shift)
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(BnP p, View y) {
  const int CV = p.x.c / 8;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long P = (long long)p.x.n * p.x.h * p.x.w;
  if (idx >= P * CV) return;
  const long long pix = idx / CV;
  const int cv = (int)(idx - pix * CV);
  float v[8];
  load8<T>(p.x.ptr, pix_off(p.x, pix) + cv * 8, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float z = fmaf(p.scale[cv * 8 + i], v[i], p.shift[cv * 8 + i]);
    v[i] = p.relu ? fmaxf(z, 0.f) : z;
  }
  store8<T>(y.ptr, pix_off(y, pix) + cv * 8, v);
}

// draw = scale * (dz - c1 - xhat * c2)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnP p, const float* c1, const float* c2, View out) {
  const int CV = p.x.c / 8;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long P = (long long)p.x.n * p.x.h * p.x.w;
  if (idx >= P * CV) return;
  const long long pix = idx / CV;
  const int cv = (int)(idx - pix * CV);
  float v[8], g[8];
  load8<T>(p.x.ptr, pix_off(p.x, pix) + cv * 8, v);
  load8<T>(p.g.ptr, pix_off(p.g, pix) + cv * 8, g);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cv * 8 + i;
    const float sc = p.scale[c];
    const float z = fmaf(sc, v[i], p.shift[c]);
    const float dz = (!p.relu || z > 0.f) ? g[i] : 0.f;
    const float xh = (v[i] - p.mean[c]) * p.invstd[c];
    v[i] = sc * (dz - c1[c] - xh * c2[c]);
  }
  store8<T>(out.ptr, pix_off(out, pix) + cv * 8, v);
}

static bool vec8_ok(const dct_view* v, int esz) {
  return view_ok(v) && v->c % 8 == 0 && v->sw % 8 == 0 && v->sh % 8 == 0 && v->sn % 8 == 0 && ((uintptr_t)v->ptr % (8 * esz)) == 0;
}
static bool chan_ok(int C) { return C >= 8 && C <= 2048 && (C & (C - 1)) == 0; }
static bool same_shape(const dct_view* a, const dct_view* b) { return a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c; }

static int plan_blocks(long long P, int C, int& ppb) {
  const int rows = 256 / (C / 8);
  long long blocks = (P + 16ll * rows - 1) / (16ll * rows);       // >= 16 pixels per thread
  if (blocks > BN_MAX_BLOCKS) blocks = BN_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  ppb = (int)((P + blocks - 1) / blocks);
  return (int)((P + ppb - 1) / ppb);
}

}  // namespace

#define BN_T(dtype, ...) do { if ((dtype) == DCT_BF16) { using T = bf16_t; __VA_ARGS__; } else { using T = float; __VA_ARGS__; } } while (0)

extern "C" size_t dct_bn_workspace_bytes(int channels) {
  return (size_t)BN_MAX_BLOCKS * (channels > 0 ? channels : 1) * 2 * sizeof(double);
}

extern "C" int dct_bn_fwd(const dct_view* raw, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, int training,
                          float* scale, float* shift, float* save_mean, float* save_invstd,
                          const dct_view* y, int relu, int dtype, void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (dtype != DCT_F32 && dtype != DCT_BF16) return DCT_ERR_BAD_ARG;
  const int esz = dtype == DCT_BF16 ? 2 : 4;
  if (!view_ok(raw) || !gamma || !beta || !scale || !shift) return DCT_ERR_BAD_ARG;
  if (!training && (!running_mean || !running_var)) return DCT_ERR_BAD_ARG;
  if (y && (!view_ok(y) || !same_shape(raw, y))) return DCT_ERR_BAD_ARG;
  if (!chan_ok(raw->c) || !vec8_ok(raw, esz) || (y && !vec8_ok(y, esz))) return DCT_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long long P = (long long)raw->n * raw->h * raw->w;
  BnP p; p.x = to_view(raw); p.g = p.x; p.scale = scale; p.shift = shift; p.mean = nullptr; p.invstd = nullptr; p.relu = relu ? 1 : 0;
  int blocks = 0;
  if (training) {
    blocks = plan_blocks(P, raw->c, p.ppb);
    if (!workspace || workspace_bytes < (size_t)blocks * raw->c * 2 * sizeof(double)) return DCT_ERR_WORKSPACE;
    const size_t lds = (size_t)(256 / (raw->c / 8)) * raw->c * 2 * sizeof(double);
    BN_T(dtype, DCT_LAUNCH(DCT_PROF_POINTWISE, (bn_reduce_kernel<T, 0>), dim3(blocks), dim3(256), lds, st, p, (double*)workspace));
  }
  DCT_LAUNCH(DCT_PROF_POINTWISE, bn_fwd_finalize_kernel, dim3(div_up(raw->c, 256)), dim3(256), 0, st, (const double*)workspace, blocks,
             raw->c, (double)P, gamma, beta, eps, momentum, running_mean, running_var, training ? 1 : 0, scale, shift, save_mean, save_invstd);
  if (y) {
    const View vy = to_view(y);
    BN_T(dtype, DCT_LAUNCH(DCT_PROF_POINTWISE, bn_apply_kernel<T>, dim3(div_up(P * (raw->c / 8), 256)), dim3(256), 0, st, p, vy));
  }
  return dct_check_launch();
}

extern "C" int dct_bn_bwd(const dct_view* raw, const dct_view* g, const float* scale, const float* shift,
                          const float* mean, const float* invstd, float* dgamma, float* dbeta, int accumulate,
                          float* c1c2, int training, int relu, const dct_view* draw, int dtype,
                          void* workspace, size_t workspace_bytes, dct_stream stream) {
  if (dtype != DCT_F32 && dtype != DCT_BF16) return DCT_ERR_BAD_ARG;
  const int esz = dtype == DCT_BF16 ? 2 : 4;
  if (!view_ok(raw) || !view_ok(g) || !view_ok(draw) || !scale || !shift || !mean || !invstd || !c1c2) return DCT_ERR_BAD_ARG;
  if (!same_shape(raw, g) || !same_shape(raw, draw)) return DCT_ERR_BAD_ARG;
  if (!chan_ok(raw->c) || !vec8_ok(raw, esz) || !vec8_ok(g, esz) || !vec8_ok(draw, esz)) return DCT_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long long P = (long long)raw->n * raw->h * raw->w;
  BnP p; p.x = to_view(raw); p.g = to_view(g); p.scale = scale; p.shift = shift; p.mean = mean; p.invstd = invstd; p.relu = relu ? 1 : 0;
  const int blocks = plan_blocks(P, raw->c, p.ppb);
  if (!workspace || workspace_bytes < (size_t)blocks * raw->c * 2 * sizeof(double)) return DCT_ERR_WORKSPACE;
  const size_t lds = (size_t)(256 / (raw->c / 8)) * raw->c * 2 * sizeof(double);
  BN_T(dtype, DCT_LAUNCH(DCT_PROF_POINTWISE, (bn_reduce_kernel<T, 1>), dim3(blocks), dim3(256), lds, st, p, (double*)workspace));
  DCT_LAUNCH(DCT_PROF_POINTWISE, bn_bwd_finalize_kernel, dim3(div_up(raw->c, 256)), dim3(256), 0, st, (const double*)workspace, blocks,
             raw->c, (double)P, training ? 1 : 0, accumulate ? 1 : 0, dgamma, dbeta, c1c2, c1c2 + raw->c);
  const View vo = to_view(draw);
  BN_T(dtype, DCT_LAUNCH(DCT_PROF_POINTWISE, bn_bwd_apply_kernel<T>, dim3(div_up(P * (raw->c / 8), 256)), dim3(256), 0, st, p,
                         (const float*)c1c2, (const float*)(c1c2 + raw->c), vo));
  return dct_check_launch();
}

// ---------------------------------------------------------------------------------------------
// nn.BatchNorm2d bookkeeping of a whole network in ONE launch (Enet: 84 layers; the reference does it inside every
// F.batch_norm call, arch/enet.py:22,55-122): r <- (1 - momentum) r + momentum b for running mean / variance from the batch
// statistics a forward pass left in one flat buffer, and num_batches_tracked += 1.  One block per layer.
namespace {
struct BnRunRec { float* running_mean; float* running_var; long long* num_batches_tracked; int c, mean_off, var_off, pad_; };
__global__ __launch_bounds__(128) void bn_running_update_kernel(const BnRunRec* recs, const float* stats, float momentum) {
  const BnRunRec r = recs[blockIdx.x];
  for (int i = threadIdx.x; i < r.c; i += 128) {
    r.running_mean[i] = __fmaf_rn(momentum, stats[r.mean_off + i], r.running_mean[i] * (1.f - momentum));
    r.running_var[i] = __fmaf_rn(momentum, stats[r.var_off + i], r.running_var[i] * (1.f - momentum));
  }
  if (threadIdx.x == 0 && r.num_batches_tracked) *r.num_batches_tracked += 1;
}

// out = a + b (+ c) over n floats, 16 bytes per lane: the flat gradient buffers of a model's concurrent backward passes,
// added in pass order (trainer/cotraining_totalloss.py::_finish_step)
__global__ __launch_bounds__(256) void flat_sum_kernel(float* out, const float* a, const float* b, const float* c, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
  float4 o = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  if (c) { const float4 z = reinterpret_cast<const float4*>(c)[i]; o.x += z.x; o.y += z.y; o.z += z.z; o.w += z.w; }
  reinterpret_cast<float4*>(out)[i] = o;
}
// x *= s over n floats at any 4-byte alignment: 16-byte accesses over the aligned body, block 0 takes the < 4 head and tail floats.
// (ddp.py: the gradient average after a SUM all-reduce, for models whose optimizer does not fold 1/world into its update.)
__global__ __launch_bounds__(256) void flat_scale_kernel(float* x, float s, long long n, int head) {
  const long long n4 = (n - head) / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    float4* p = reinterpret_cast<float4*>(x + head) + i;
    float4 v = *p;
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    *p = v;
  }
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x < head) x[threadIdx.x] *= s;
    const long long t = head + 4 * n4 + threadIdx.x;
    if (threadIdx.x < 4 && t < n) x[t] *= s;
  }
}
}  // namespace

extern "C" int dct_flat_scale(float* x, float scale, long long n, dct_stream stream) {
  if (!x || n < 1 || ((uintptr_t)x & 3)) return DCT_ERR_BAD_ARG;
  int head = (int)(((16 - ((uintptr_t)x & 15)) & 15) / 4);
  if (head > n) head = (int)n;
  const long long n4 = (n - head) / 4;
  DCT_LAUNCH(DCT_PROF_OTHER, flat_scale_kernel, dim3((unsigned)(n4 ? (n4 + 255) / 256 : 1)), dim3(256), 0, (hipStream_t)stream, x, scale, n, head);
  return dct_check_launch();
}

extern "C" int dct_bn_running_update(const void* records_dev, int n_layers, const float* stats, float momentum, dct_stream stream) {
  if (!records_dev || n_layers < 1 || !stats) return DCT_ERR_BAD_ARG;
  DCT_LAUNCH(DCT_PROF_OTHER, bn_running_update_kernel, dim3(n_layers), dim3(128), 0, (hipStream_t)stream,
             (const BnRunRec*)records_dev, stats, momentum);
  return dct_check_launch();
}

extern "C" int dct_flat_sum(float* out, const float* a, const float* b, const float* c, long long n, dct_stream stream) {
  if (!out || !a || !b || n < 1 || (n & 3) || ((uintptr_t)out & 15) || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)c & 15))
    return DCT_ERR_BAD_ARG;
  DCT_LAUNCH(DCT_PROF_OTHER, flat_sum_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, a, b, c, n / 4);
  return dct_check_launch();
}
