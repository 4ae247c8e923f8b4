// How fast can this chip WRITE?  Round 4 measured 3.1 TB/s on every write-only stream of the step (the stem's 132 MB output,
// the pooling-only epilogues) against 5.8-6.5 TB/s for mixed read + write streams, and priced a written byte at twice a read one.
// This probe separates the store path from the kernels around it: a pure fill of N bytes with 16-byte stores, as a function of
// the store policy (default / nt / sc1 sc0), the waves in flight, the bytes per thread, and whether the buffer was just read.
//   hipcc --offload-arch=gfx950 -O3 probe.hip -o probe.bin && ./probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__device__ __forceinline__ void st16(uint4* p, uint4 v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  if (MODE == 3) { const u32x4_ q = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory"); return; }
  if (MODE == 4) { const u32x4_ q = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(q) : "memory"); return; }
  if (MODE == 5) { const u32x4_ q = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(q) : "memory"); return; }
  if (MODE == 6) { const u32x4_ q = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(q) : "memory"); return; }
  if (MODE == 0) *p = v;
  else if (MODE == 1) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p));
  }
  else {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 q = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(q) : "memory");
  }
}

// grid-stride fill: every wave-instruction writes 1 KiB contiguous
template <int MODE>
__global__ __launch_bounds__(256) void fill(uint4* dst, long long n16, unsigned seed) {
  const uint4 v = make_uint4(seed, seed + 1, seed + 2, seed + 3);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) st16<MODE>(dst + i, v);
}
// each block owns a contiguous span (a conv epilogue's pattern: a block drains its own tile rows)
template <int MODE>
__global__ __launch_bounds__(256) void fill_span(uint4* dst, long long n16, unsigned seed, int per_block16) {
  const uint4 v = make_uint4(seed, seed + 1, seed + 2, seed + 3);
  const long long b0 = (long long)blockIdx.x * per_block16;
  for (int i = threadIdx.x; i < per_block16 && b0 + i < n16; i += 256) st16<MODE>(dst + b0 + i, v);
}
// the stem's pattern: a wave owns 4 KiB; each of its four store instructions writes eight 128-byte pieces 512 bytes apart (lane = piece * 8 + sub)
template <int MODE>
__global__ __launch_bounds__(256) void fill_pieces(uint4* dst, long long n16, unsigned seed) {
  const uint4 v = make_uint4(seed, seed + 1, seed + 2, seed + 3);
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63, piece = lane >> 3, sub = lane & 7;
  const long long waves = (long long)gridDim.x * 4;
  for (long long w = wave; w * 256 < n16; w += waves) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const long long i = w * 256 + piece * 32 + p * 8 + sub;
      if (i < n16) st16<MODE>(dst + i, v);
    }
  }
}
// ... and the contiguous form with the same four stores per thread (1 KiB per instruction, a wave's 4 KiB in order)
template <int MODE>
__global__ __launch_bounds__(256) void fill_rows4(uint4* dst, long long n16, unsigned seed) {
  const uint4 v = make_uint4(seed, seed + 1, seed + 2, seed + 3);
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const long long waves = (long long)gridDim.x * 4;
  for (long long w = wave; w * 256 < n16; w += waves) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const long long i = w * 256 + p * 64 + lane;
      if (i < n16) st16<MODE>(dst + i, v);
    }
  }
}
__global__ __launch_bounds__(256) void read_sum(const uint4* src, long long n16, unsigned* out) {
  unsigned acc = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(256) void copy16(const uint4* src, uint4* dst, long long n16) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

template <typename F> static double time_us(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / reps;
}

int main() {
  const long long sizes[] = {16ll << 20, 33ll << 20, 132ll << 20, 528ll << 20};
  uint4 *buf, *src; unsigned* out;
  hipMalloc(&buf, 600ll << 20); hipMalloc(&src, 600ll << 20); hipMalloc(&out, 64);
  hipMemset(buf, 0, 600ll << 20); hipMemset(src, 1, 600ll << 20);
  printf("%-44s %10s %10s\n", "kernel", "us", "TB/s");
  for (long long bytes : sizes) {
    const long long n16 = bytes / 16;
    for (int blocks : {1024, 4096, 16384}) {
      char name[128];
      double t;
      snprintf(name, sizeof name, "fill default   %4lld MB, %5d blocks", bytes >> 20, blocks);
      t = time_us([&] { hipLaunchKernelGGL(fill<0>, dim3(blocks), dim3(256), 0, 0, buf, n16, 7u); }, 20); printf("%-44s %10.1f %10.2f\n", name, t, bytes / t / 1e6);
      snprintf(name, sizeof name, "fill nt        %4lld MB, %5d blocks", bytes >> 20, blocks);
      t = time_us([&] { hipLaunchKernelGGL(fill<1>, dim3(blocks), dim3(256), 0, 0, buf, n16, 7u); }, 20); printf("%-44s %10.1f %10.2f\n", name, t, bytes / t / 1e6);
      snprintf(name, sizeof name, "fill sc0 sc1   %4lld MB, %5d blocks", bytes >> 20, blocks);
      t = time_us([&] { hipLaunchKernelGGL(fill<2>, dim3(blocks), dim3(256), 0, 0, buf, n16, 7u); }, 20); printf("%-44s %10.1f %10.2f\n", name, t, bytes / t / 1e6);
    }
    {
      char name[128]; double t;
      const int per = 2048;      // 32 KiB per block: a 128-pixel x 128-channel bf16 tile
      const int blocks = (int)((n16 + per - 1) / per);
      snprintf(name, sizeof name, "fill_span 32 KiB/block default %4lld MB", bytes >> 20);
      t = time_us([&] { hipLaunchKernelGGL(fill_span<0>, dim3(blocks), dim3(256), 0, 0, buf, n16, 7u, per); }, 20); printf("%-44s %10.1f %10.2f\n", name, t, bytes / t / 1e6);
      snprintf(name, sizeof name, "fill_span 32 KiB/block nt      %4lld MB", bytes >> 20);
      t = time_us([&] { hipLaunchKernelGGL(fill_span<1>, dim3(blocks), dim3(256), 0, 0, buf, n16, 7u, per); }, 20); printf("%-44s %10.1f %10.2f\n", name, t, bytes / t / 1e6);
      snprintf(name, sizeof name, "read           %4lld MB, 4096 blocks", bytes >> 20);
      t = time_us([&] { hipLaunchKernelGGL(read_sum, dim3(4096), dim3(256), 0, 0, src, n16, out); }, 20); printf("%-44s %10.1f %10.2f\n", name, t, bytes / t / 1e6);
      snprintf(name, sizeof name, "copy (R + W)   %4lld MB, 4096 blocks", bytes >> 20);
      t = time_us([&] { hipLaunchKernelGGL(copy16, dim3(4096), dim3(256), 0, 0, src, buf, n16); }, 20); printf("%-44s %10.1f %10.2f (x2 bytes moved)\n", name, t, bytes / t / 1e6);
    }
  }
  // COLD destination: the 132 MB being written are not in the memory-side cache (600 MB of other traffic went by since their last use)
  {
    const long long bytes = 132ll << 20, n16 = bytes / 16;
    uint4* big; hipMalloc(&big, 1200ll << 20); hipMemset(big, 2, 1200ll << 20);
    const double tr = time_us([&] { hipLaunchKernelGGL(read_sum, dim3(4096), dim3(256), 0, 0, big, (600ll << 20) / 16, out); }, 10);
    const double tw = time_us([&] { hipLaunchKernelGGL(fill<0>, dim3(8192), dim3(256), 0, 0, big, (600ll << 20) / 16, 9u); }, 10);
    printf("%-44s %10.1f %10.2f\n", "read 600 MB (the flush)", tr, 600.0 * 1048576 / tr / 1e6);
    printf("%-44s %10.1f %10.2f\n", "fill 600 MB (the other flush)", tw, 600.0 * 1048576 / tw / 1e6);
#define COLD(MODE, label)                                                                                                      \
    {                                                                                                                          \
      double t = time_us([&] {                                                                                                 \
        hipLaunchKernelGGL(read_sum, dim3(4096), dim3(256), 0, 0, big, (600ll << 20) / 16, out);                               \
        hipLaunchKernelGGL(fill<MODE>, dim3(8192), dim3(256), 0, 0, buf, n16, 7u);                                            \
      }, 10);                                                                                                                  \
      printf("%-44s %10.1f %10.2f\n", "fill 132 MB " label " behind a 600 MB read", t - tr, bytes / (t - tr) / 1e6);           \
      t = time_us([&] {                                                                                                        \
        hipLaunchKernelGGL(fill<0>, dim3(8192), dim3(256), 0, 0, big, (600ll << 20) / 16, 9u);                                 \
        hipLaunchKernelGGL(fill<MODE>, dim3(8192), dim3(256), 0, 0, buf, n16, 7u);                                            \
      }, 10);                                                                                                                  \
      printf("%-44s %10.1f %10.2f\n", "fill 132 MB " label " behind a 600 MB fill", t - tw, bytes / (t - tw) / 1e6);           \
    }
    COLD(0, "default") COLD(1, "nt") COLD(2, "sc0 sc1") COLD(3, "sc1") COLD(4, "sc0") COLD(5, "sc0 sc1 nt") COLD(6, "sc1 nt")
#define COLDK(KERNEL, label)                                                                                                   \
    {                                                                                                                          \
      double t = time_us([&] {                                                                                                 \
        hipLaunchKernelGGL(read_sum, dim3(4096), dim3(256), 0, 0, big, (600ll << 20) / 16, out);                               \
        hipLaunchKernelGGL(KERNEL, dim3(8448), dim3(256), 0, 0, buf, n16, 7u);                                                \
      }, 10);                                                                                                                  \
      printf("%-44s %10.1f %10.2f\n", label " behind a 600 MB read", t - tr, bytes / (t - tr) / 1e6);                          \
    }
    // round 5: does it matter how a wave's stores land?  (8448 blocks = one 4 KiB span per wave over 132 MB)
    COLDK(fill_rows4<1>, "rows of 1 KiB, nt") COLDK(fill_pieces<1>, "8 x 128 B pieces, nt") COLDK(fill_rows4<0>, "rows of 1 KiB, default") COLDK(fill_pieces<0>, "8 x 128 B pieces, default")
    // ... and a cold READ of 132 MB for comparison
    {
      double t = time_us([&] {
        hipLaunchKernelGGL(read_sum, dim3(4096), dim3(256), 0, 0, big, (600ll << 20) / 16, out);
        hipLaunchKernelGGL(read_sum, dim3(4096), dim3(256), 0, 0, src, n16, out);
      }, 10);
      printf("%-44s %10.1f %10.2f\n", "read 132 MB behind a 600 MB read", t - tr, bytes / (t - tr) / 1e6);
    }
  }
  // a write-only stream between two reads of OTHER data (what the step does: the stem writes 132 MB, then convs read/write elsewhere)
  {
    const long long bytes = 132ll << 20, n16 = bytes / 16;
    double t = time_us([&] {
      hipLaunchKernelGGL(read_sum, dim3(4096), dim3(256), 0, 0, src, (300ll << 20) / 16, out);
      hipLaunchKernelGGL(fill<0>, dim3(4096), dim3(256), 0, 0, buf, n16, 7u);
    }, 20);
    double tr = time_us([&] { hipLaunchKernelGGL(read_sum, dim3(4096), dim3(256), 0, 0, src, (300ll << 20) / 16, out); }, 20);
    printf("%-44s %10.1f %10.2f\n", "fill 132 MB behind a 300 MB read (fill part)", t - tr, bytes / (t - tr) / 1e6);
  }
  return 0;
}
