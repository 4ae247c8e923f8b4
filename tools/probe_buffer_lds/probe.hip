// What buffer_load_dwordx4 ... lds does with lanes the descriptor's range check rejects, and with EXEC-masked lanes.
//   hipcc --offload-arch=gfx950 -O3 probe.hip -o probe && ./probe
// Findings feed the staging code of wgrad3_kernel (csrc/wgrad.hip): a lane whose voffset lies beyond num_records
// must leave ZEROS in its 16 bytes of LDS, whatever soffset says, and the check must not count soffset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void probe(const char* src, unsigned* out, unsigned nrec, int soff, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* w = reinterpret_cast<unsigned*>(smem);
  for (int i = threadIdx.x; i < 512; i += 64) w[i] = 0xABABABABu;       // stale pattern
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)nrec, 0x00020000);
  const int lane = threadIdx.x;
  int voff = lane * 16;
  if (mode == 0) { if (lane & 1) voff = 0x7fffffff; }                    // far beyond the range
  if (mode == 1) { if (lane & 1) voff = (int)nrec; }                     // first byte beyond
  if (mode == 2) { if (lane & 1) voff = (int)0x80000000u; }              // "negative"
  if (mode == 3) { /* all in range, soffset pushes the odd-lane... (uniform) beyond: see host */ }
  if (mode == 4) {                                                       // EXEC-masked odd lanes
    if ((lane & 1) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)smem, 16, voff, soff, 0, 0);
  } else {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)smem, 16, voff, soff, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = w[i];
}

int main() {
  const unsigned N = 4096;
  std::vector<unsigned> h(N / 4 + 4096);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x10000000u + (unsigned)i;
  char* d; unsigned* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> r(256);
  const char* names[] = {"voffset 0x7fffffff on odd lanes", "voffset == num_records on odd lanes", "voffset 0x80000000 on odd lanes",
                         "all lanes in range, soffset = num_records (beyond for every lane if soffset is checked)", "odd lanes EXEC-masked"};
  for (int mode = 0; mode < 5; ++mode) {
    const int soff = mode == 3 ? (int)N : 256;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, o, N, soff, mode);
    hipDeviceSynchronize();
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int zero = 0, stale = 0, data = 0, other = 0, wrong = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 4; ++j) {
        const unsigned v = r[lane * 4 + j];
        const unsigned expect = 0x10000000u + (unsigned)(soff / 4 + lane * 4 + j);
        if (lane & 1 || mode == 3) {
          if (v == 0) ++zero; else if (v == 0xABABABABu) ++stale; else if (v == expect) ++data; else ++other;
        } else if (v != expect) ++wrong;
      }
    printf("mode %d (%s): probed lanes -> zero %d, stale %d, real data %d, other %d; even lanes wrong %d\n", mode, names[mode], zero, stale, data, other, wrong);
  }
  return 0;
}
