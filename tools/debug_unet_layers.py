"""GPU debug aid: per-layer forward / gradient error of the HIP UNet against the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, oracle
from dct_amd.arch import get_arch

def rel2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

def main(B, H, W, C, dtype):
    torch.manual_seed(5)
    onet = oracle.build_net("unet", C, dropout_p=0.0).eval()
    net = get_arch("unet", {"num_classes": C, "compute_dtype": dtype, "dropout_p": 0.0})
    net.load_state_dict(onet.state_dict()); net = net.to("cuda:0").eval()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, 1, H, W, generator=g); t = torch.randint(0, C, (B, H, W), generator=g)
    taps = {}
    xo = x.clone().requires_grad_(True)
    yo = onet(xo, taps=taps)
    for v in taps.values(): v.retain_grad()
    yo2 = yo.detach().clone().requires_grad_(True)
    gl = torch.autograd.grad(oracle.cross_entropy_2d(yo2, t), yo2)[0]
    yo.backward(gl)
    net._ensure_packs()
    xd = x.to("cuda:0")
    logits, A = net._run_forward(xd, True)
    m = {"dec1": "p1", "dec2": "p2", "dec3": "p3", "dec4": "p4", "enc1": "e1b"}
    print(f"== B{B} {H}x{W} C{C} {dtype}")
    for k, ak in m.items():
        print(f"fwd {k:8s} {rel2(A[ak].float().cpu().permute(0,3,1,2).numpy(), taps[k].detach().numpy()):.2e}")
    for lvl in (4, 3, 2):
        cat = A[f"cat{lvl-1}"]; co = cat.shape[3] // 2
        print(f"fwd enc{lvl}     {rel2(cat[..., :co].float().cpu().permute(0,3,1,2).numpy(), taps[f'enc{lvl}'].detach().numpy()):.2e}")
    print(f"fwd center   {rel2(A['cat4'][..., :512].float().cpu().permute(0,3,1,2).numpy(), taps['center'].detach().numpy()):.2e}")
    print(f"fwd logits   {rel2(logits.cpu().permute(0,3,1,2).numpy(), yo.detach().numpy()):.2e}")
    net.flat_params.ensure_grads()
    net._debug = {}
    dx = net._run_backward(A, gl.permute(0, 2, 3, 1).contiguous().to("cuda:0"), True, True)
    for lvl in (2, 3, 4):
        for ab in "ba":
            k = f"e{lvl}{ab}"
            ref = taps[k].grad * (taps[k].detach() > 0)
            got = net._debug[f"d{k}"].float().cpu().permute(0, 3, 1, 2)
            err = (got - ref).abs()
            bad = (err > 1e-4 * ref.abs().max()).nonzero()
            print(f"bwd d{k} {rel2(got.numpy(), ref.numpy()):.2e} nbad {len(bad)} of {err.numel()} n{sorted(set(bad[:,0].tolist()))} y{sorted(set(bad[:,2].tolist()))} x{sorted(set(bad[:,3].tolist()))} ch[{bad[:,1].min().item() if len(bad) else -1},{bad[:,1].max().item() if len(bad) else -1}]")
    print(f"bwd grad_x   {rel2(dx.cpu().reshape(B,1,H,W).numpy(), xo.grad.numpy()):.2e}")
    for (k, p), (_, po) in zip(net.named_parameters(), onet.named_parameters()):
        print(f"bwd {k:22s} {rel2(p.grad.cpu().numpy(), po.grad.numpy()):.2e}")

if __name__ == "__main__":
    main(2, 200, 200, 2, torch.float32)

