import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from dct_amd import hip_ops as K
DEV = "cuda:0"
def run(B, Cin, H, W, Cout, pad, use_mask, dtype=torch.float32, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    ref = F.conv2d(x, w, padding=pad)
    mk = torch.randn(ref.shape, generator=g)
    if use_mask:
        ref = ref * (mk > 0)
    y = torch.full((B, ref.shape[2], ref.shape[3], Cout), float("nan"), dtype=dtype, device=DEV)
    K.conv2d(x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV), w.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV), None, y,
             pad_h=pad, pad_w=pad, mask=mk.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV) if use_mask else None)
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs()
    bad = err > 1e-4 * ref.abs().max()
    msg = f"B{B} Cin{Cin} {H}x{W} Cout{Cout} pad{pad} mask{int(use_mask)}: maxerr {err.max():.3e} nbad {int(bad.sum())}/{bad.numel()} nan {int(torch.isnan(got).sum())}"
    if bad.any():
        idx = bad.nonzero()
        msg += f" first bad {idx[:5].tolist()} ; bad n {sorted(set(idx[:,0].tolist()))} ch-range {idx[:,1].min().item()}-{idx[:,1].max().item()} y {sorted(set(idx[:,2].tolist()))} x {sorted(set(idx[:,3].tolist()))}"
    print(msg)
run(2, 128, 12, 12, 128, 2, True)
run(2, 128, 12, 12, 128, 2, False)
run(1, 128, 12, 12, 128, 2, False)
run(2, 128, 12, 12, 128, 0, False)
run(2, 128, 14, 14, 128, 0, False)
run(2, 64, 12, 12, 128, 2, False)
run(2, 128, 12, 12, 64, 2, False)
run(2, 256, 12, 12, 128, 2, False)
run(3, 128, 10, 10, 128, 2, False)
run(2, 128, 12, 12, 128, 2, False, torch.bfloat16)
