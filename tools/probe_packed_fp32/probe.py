"""Do packed-FP32 VALU instructions survive beside the MFMA conv kernels?  (csrc/Makefile NOPK.)

victim.hip is the JSD forward kernel cut down to per-thread sums (variant 1: as the compiler emits it, with v_pk_mul_f32 /
v_pk_add_f32; 2: without the entropies; 4: the same products behind asm barriers, i.e. without the packed multiplies; 5: logs only).
Each variant is launched 60 times on one stream, alone and beside a stream of dct_conv2d launches, and every per-thread result is
compared with the quiet run.  Built twice: as the compiler likes it, and with -Xclang -target-feature -Xclang -packed-fp32-ops.

    python tools/probe_packed_fp32/probe.py          (needs a GPU; builds the two victim libraries next to this file)
Variants 6-8 are streaming kernels over float2 pairs: 6 = a * s + b * s (v_pk_mul_f32 + v_pk_add_f32: the arithmetic of a
pre-multiplied sum, which is how RCCL forms ReduceOp.AVG), 7 = log2(a^2 + 1) * b (v_log_f32 results straight into
v_pk_mul_f32), 8 = a * s + swap(b) (v_pk_fma_f32 with op_sel), 9 = a * s + b as
v_pk_fma_f32 op_sel_hi:[0,1,1] in inline asm (RCCL's own instruction form; packed in both builds).
Measured (MI355X, ROCm 7.2): default build, beside the shared-halo conv kernel: variants 1, 3 and 8 wrong in 60/60 launches
(<= 5000 of 262144 / 2700 of 2097152 threads), beside the filter-row weight gradient 1 in 22-57 and 8 in 60 of 60; 2 / 4 / 5 / 6 / 7 / 9
exact; flag build: all exact; every build exact without the convs and beside the other loads.  What variants 1, 3 and 8 share and the
exact ones lack is a packed-FP32 instruction that reads a source pair with SWAPPED halves (op_sel:[..1..]); plain packed
multiplies / adds (6), packed products of transcendental results (7) and the broadcast form op_sel_hi:[0,1,1] (9) survive.
Synthetic neighbours (victim.hip::neighbour) instead of the conv kernels: v_mfma_f32_16x16x32_bf16 back to back from registers is
enough (1 / 3: 35 k of 262144 threads, 8: 135 k of 2097152, 60 of 60 launches); LDS reads, LDS-DMA, a dependent 32x32x16 chain
and the 16x16x32 loop throttled by LDS reads are not.  Variants 10-17 (victim.hip::klass) are the other operand-routing
instruction classes the library uses (DPP, v_pk_fma_f16 op_sel, bf16 converts, f64, v_perm / v_alignbit, fma, ds_bpermute, LDS):
exact beside every load, in both builds."""
import ctypes
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import dct_amd  # noqa: E402,F401
from dct_amd import hip_ops as K  # noqa: E402

dev = "cuda:0"


def build(name, extra):
    out = os.path.join(here, name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] + extra +
                          ["-o", out, os.path.join(here, "victim.hip")], stderr=subprocess.DEVNULL)
    lib = ctypes.CDLL(out)
    lib.run_var.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.run_neighbour.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.run_class.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    lib.run_premul.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    return lib


g = torch.Generator(device=dev).manual_seed(0)
P = 8 * 256 * 256
a = torch.randn(P * 4, device=dev, generator=g) * 0.3
b = torch.randn(P * 4, device=dev, generator=g) * 0.3
x = torch.randn(16, 124, 124, 128, device=dev, generator=g).to(torch.bfloat16)
w = (torch.randn(128, 3, 3, 128, device=dev, generator=g) / 34).to(torch.bfloat16)
y = torch.empty(16, 122, 122, 128, device=dev, dtype=torch.bfloat16)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
xt = torch.randn(16, 44, 44, 128, device=dev, generator=g).to(torch.bfloat16)
wt = (torch.randn(4 * 64, 128, device=dev, generator=g) / 11).to(torch.bfloat16)
yt = torch.empty(16, 88, 88, 64, device=dev, dtype=torch.bfloat16)
dw = torch.zeros(128 * 9 * 128, device=dev)
ex = torch.randn(8, 64, 64, 64, device=dev, generator=g)
ew = torch.randn(64, 3, 3, 64, device=dev, generator=g) / 24
ey = torch.empty(8, 64, 64, 64, device=dev)
A = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
Bm = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
big = torch.rand(1 << 26, device=dev)
SYNTH = ("synthetic: LDS reads", "synthetic: MFMA 16x16x32", "synthetic: MFMA 32x32x16", "synthetic: MFMA 16x16x32 on LDS reads", "synthetic: LDS-DMA")
LOADS = ("none", "conv", "convT (per-tap kernel)", "weight gradient", "enet MFMA conv", "rocBLAS bf16 GEMM", "elementwise") + SYNTH
sink = torch.zeros(1024, device=dev)
CLASSES = {10: "DPP row_shr", 11: "v_pk_fma_f16 op_sel", 12: "v_cvt_pk_bf16_f32", 13: "f64 mul / add", 14: "v_perm + v_alignbit", 15: "f32 fma",
           16: "ds_bpermute", 17: "LDS write / read"}
ALL = (1, 2, 3, 4, 5, 6, 7, 8, 9) + tuple(CLASSES)


def run(lib, mode):
    if mode >= 10:                  # other instruction classes (victim.hip::klass): DPP, packed f16 op_sel, bf16 converts, f64, v_perm / v_alignbit, fma, ds_bpermute, LDS
        out = torch.empty_like(a)
        with torch.cuda.stream(sA):
            lib.run_class(mode - 10, a.data_ptr(), b.data_ptr(), 0.125, a.numel() // 4, out.data_ptr(), sA.cuda_stream)
        return out
    if mode >= 6:                   # streaming kernels over the 8 M floats of a and b: 6 = a / 8 + b / 8, 7 = log2(a^2 + 1) * b, 8 = a / 8 + swap(b), 9 = a / 8 + b (asm)
        out = torch.empty_like(a)
        with torch.cuda.stream(sA):
            lib.run_premul(mode - 6, a.data_ptr(), b.data_ptr(), 0.125, a.numel() // 4, out.data_ptr(), sA.cuda_stream)
        return out
    part = torch.zeros(2048, device=dev)
    per = torch.zeros(1024 * 256, device=dev)
    with torch.cuda.stream(sA):
        part.zero_()
        per.zero_()
        lib.run_var(mode, a.data_ptr(), b.data_ptr(), P, part.data_ptr(), per.data_ptr(), sA.cuda_stream)
    return per


for label, extra in (("default build", []), ("-packed-fp32-ops", ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-DVICTIM_NO_PK_ASM"])):
    lib = build("victim_%s.so" % ("nopk" if extra else "pk"), extra)
    refs = {}
    for mode in ALL:
        r = run(lib, mode)
        torch.cuda.synchronize()
        refs[mode] = r.clone()
        torch.cuda.synchronize()
    for load in LOADS:
        for mode in (ALL if load in ("none", "conv", "weight gradient", SYNTH[1]) else (1, 8) if load in SYNTH else (1, 6, 7, 8, 9)):
            bad, nbad = 0, 0
            for it in range(60):
                with torch.cuda.stream(sB):
                    if load == "conv":                      # shared-halo 3x3 kernel: LDS-DMA staging + v_mfma_f32_16x16x32_bf16
                        for _ in range(6):
                            K.conv2d(x, w, None, y, relu=True)
                    elif load == "convT (per-tap kernel)":  # igemm2: LDS-DMA staging + v_mfma_f32_32x32x16_bf16
                        for _ in range(6):
                            K.conv2d(xt, wt, None, yt, R=1, S=1, relu=True, scatter2x2=True)
                    elif load == "weight gradient":
                        for _ in range(3):
                            K.conv2d_wgrad(y, x, dw, accumulate=True)
                    elif load == "enet MFMA conv":          # plain global loads + v_mfma_f32_32x32x16_bf16, no LDS-DMA
                        for _ in range(12):
                            K.enet_conv(ex, ew, None, None, ey, R=3, S=3, pad_h=1, pad_w=1, ws=(9 * 64, 64, 1), compute=torch.bfloat16)
                    elif load == "rocBLAS bf16 GEMM":
                        for _ in range(3):
                            torch.matmul(A, Bm)
                    elif load in SYNTH:                     # victim.hip::neighbour: one ingredient of the conv kernels at a time
                        for _ in range(3):
                            lib.run_neighbour(SYNTH.index(load), big.data_ptr(), sink.data_ptr(), 400, sB.cuda_stream)
                    elif load == "elementwise":
                        for _ in range(6):
                            big.mul_(1.0001)
                t = run(lib, mode)
                torch.cuda.synchronize()
                if not torch.equal(t, refs[mode]):
                    bad += 1
                    nbad = max(nbad, int((t != refs[mode]).sum()))
            print(f"{label:18s} load {load:24s} variant {mode}{' (' + CLASSES[mode] + ')' if mode in CLASSES else ''}: {bad}/60 launches differ (at most {nbad} of {t.numel()} threads)", flush=True)
