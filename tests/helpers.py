"""Shared test scaffolding: synthetic loaders with the batch format the trainer consumes
(``[[img, gt], meta, names]``, cotraining_totalloss.py:209,221,292)."""
import numpy as np
import torch


class FakeDataset:
    def __init__(self):
        from dct_amd import ModelMode
        self.training = ModelMode.EVAL

    def set_mode(self, mode):
        self.training = mode


class FakeLoader(list):
    def __init__(self, batches, batch_size):
        super().__init__(batches)
        self.batch_size = batch_size
        self.dataset = FakeDataset()


def batches(seed, n, B, H, C, W=None):
    """same generator protocol as tools/capture_golden.py::_batches"""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        img = torch.rand(B, 1, H, W or H, generator=g)
        gt = torch.randint(0, C, (B, 1, H, W or H), generator=g)
        out.append([[img, gt], None, [f"s{seed}_{i}_{j}" for j in range(B)]])
    return out


def digest(t):
    t = t.detach().double().flatten().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), t.norm().item(), t.abs().max().item()])


def blob_batches(seed, n, B, H, C, W=None, noise=0.08):
    """Blob-structured synthetic slices ("blob-structured labels optional for DSC sanity"): background 0 plus
    one to three random ellipses per foreground class; the image is a class-dependent grey level plus smooth noise, in
    [0, 1] -- learnable by a segmentation net, unlike i.i.d. random labels, so losses fall and Dice rises within tens of
    steps.  Same batch format as ``batches``."""
    W = W or H
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    levels = torch.linspace(0.15, 0.85, C)
    out = []
    for i in range(n):
        gt = torch.zeros(B, H, W, dtype=torch.int64)
        for b in range(B):
            for c in range(1, C):
                for _ in range(int(torch.randint(1, 4, (1,), generator=g))):
                    cy, cx = (torch.rand(2, generator=g) * torch.tensor([H, W]) * 0.8 + torch.tensor([H, W]) * 0.1).tolist()
                    ry, rx = (torch.rand(2, generator=g) * (min(H, W) / 5 - min(H, W) / 16) + min(H, W) / 16).tolist()
                    gt[b][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = c
        img = levels[gt] + noise * torch.randn(B, H, W, generator=g)
        # 3x3 box blur: soft edges, spatially correlated noise
        img = torch.nn.functional.avg_pool2d(img.unsqueeze(1), 3, stride=1, padding=1, count_include_pad=False)
        img = img.clamp_(0.0, 1.0)
        out.append([[img.contiguous(), gt.unsqueeze(1).contiguous()], None, [f"blob{seed}_{i}_{j}" for j in range(B)]])
    return out


class MaskReplayNet(torch.nn.Module):
    """Oracle UNet wrapper: each forward pops the next recorded pair of dropout masks (GPU Philox masks cannot equal CPU
    masks, so parity runs replay the masks the HIP network drew)."""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self.queue = []

    def forward(self, x):
        masks = self.queue.pop(0) if self.queue else None
        return self.net(x, dropout_masks=masks) if masks is not None else self.net(x)


def round_conv_operands(onet, dtype):
    """Make the oracle's convolutions round BOTH operands to ``dtype`` wherever the HIP plan's MFMA form does (csrc/enet.hip::
    enet_mconv_kernel: low-precision mode, >= 16 input channels in whole groups of 8, taps x Cin in whole steps of 16, <= 128
    output channels); accumulation stays fp32 on both sides.  The weight gradient still lands on the fp32 parameter.  (The
    MFMA weight gradient rounds the layer input of EVERY convolution; for the ineligible ones -- the 1-channel image, the 3-/13-/14-
    channel ends -- the oracle keeps fp32 there, inside the tests' bands.)"""
    import torch

    def eligible(m):
        if isinstance(m, torch.nn.ConvTranspose2d):
            cin, cout = m.in_channels, m.out_channels
        elif isinstance(m, torch.nn.Conv2d):
            cin, cout = m.in_channels, m.out_channels
        else:
            return False
        taps = m.kernel_size[0] * m.kernel_size[1]
        return cin >= 16 and cin % 8 == 0 and (taps * cin) % 16 == 0 and cout <= 128

    def pre(m, inp):
        m._w_full = m.weight.data
        m.weight.data = m.weight.data.to(dtype).float()
        return (inp[0].to(dtype).float(),) + tuple(inp[1:])

    def post(m, _i, _o):
        m.weight.data = m._w_full
        del m._w_full

    for m in onet.modules():
        if eligible(m):
            m.register_forward_pre_hook(pre)
            m.register_forward_hook(post)
    return onet
