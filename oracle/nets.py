"""fp32 CPU restatement of the two networks on the hot path (TEST INFRASTRUCTURE ONLY).

UNet  restates /root/reference/generalframework/arch/network.py:115-130,153-171,196-240 and, with ``batchnorm=True``
      (``build_net("unet_bn")``), UNet_bn / UNetDec_bn / UNetEnc_bn (:132-150,173-193,243-290)
Enet  restates /root/reference/generalframework/arch/enet.py:8-30,33-152,167-243
init  restates /root/reference/generalframework/arch/__init__.py:60-81

Parameter *names* equal the reference's ``state_dict`` keys so that a
state dict produced here loads into the imported reference and vice versa; the code
itself is table-driven and functional, not a transcription.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Slots(nn.Module):
    """Container that registers children under explicit integer names.

    The reference builds blocks as ``nn.Sequential`` with parameter-free layers in
    between, so its keys look like ``down.0.weight`` / ``down.2.weight``.
    """

    def __init__(self, layers: Dict[int, nn.Module]):
        super().__init__()
        for idx, layer in layers.items():
            self.add_module(str(idx), layer)

    def at(self, idx: int) -> nn.Module:
        return getattr(self, str(idx))


def _resize_bilinear(x: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    # F.upsample_bilinear == interpolate(mode='bilinear', align_corners=True); network.py:232-240
    return F.interpolate(x, size=tuple(int(s) for s in size), mode="bilinear", align_corners=True)


class _Holder(nn.Module):
    pass


class UNet(nn.Module):
    """Valid-conv UNet (network.py:196-240).

    ``dropout_p`` exposes the two ``nn.Dropout(.5)`` sites (network.py:165,210) so parity
    runs can disable them; ``dropout_masks`` (two 0/1 float tensors shaped like the
    tensors they gate) inject explicit masks so a GPU-generated mask can be replayed.
    """

    WIDTHS = (64, 128, 256, 512)

    def __init__(self, in_channels: int = 1, num_classes: int = 2, dropout_p: float = 0.5, batchnorm: bool = False):
        super().__init__()
        self.dropout_p = float(dropout_p)
        self.batchnorm = bool(batchnorm)
        # Slot numbers are the reference's nn.Sequential positions (= state_dict keys).  Per block: (first conv, its
        # BatchNorm or None, second conv, its BatchNorm or None, transposed conv or None).
        if batchnorm:       # UNetDec_bn: BN after the first conv only; centre / UNetEnc_bn: after both; enc1: after the first
            DEC, CEN, ENC, E1 = (0, 1, 3, None, None), (0, 1, 3, 4, 7), (0, 1, 3, 4, 6), (0, 1, 3, None, None)
        else:
            DEC, CEN, ENC, E1 = (0, None, 2, None, None), (0, None, 2, None, 5), (0, None, 2, None, 4), (0, None, 2, None, None)
        self._idx = dict(dec=DEC, center=CEN, enc=ENC, enc1=E1)

        def block(idx, cin, feat, cout=None):
            layers = {idx[0]: nn.Conv2d(cin, feat, 3), idx[2]: nn.Conv2d(feat, feat, 3)}
            if idx[1] is not None:
                layers[idx[1]] = nn.BatchNorm2d(feat)
            if idx[3] is not None:
                layers[idx[3]] = nn.BatchNorm2d(feat)
            if idx[4] is not None:
                layers[idx[4]] = nn.ConvTranspose2d(feat, cout, 2, stride=2)
            return _Slots(dict(sorted(layers.items())))
        cin = in_channels
        for lvl, width in enumerate(self.WIDTHS, start=1):
            blk = _Holder()
            blk.down = block(DEC, cin, width)
            setattr(self, f"dec{lvl}", blk)
            cin = width
        self.center = block(CEN, 512, 1024, 512)
        for lvl, (cin, feat, cout) in {4: (1024, 512, 256), 3: (512, 256, 128), 2: (256, 128, 64)}.items():
            blk = _Holder()
            blk.up = block(ENC, cin, feat, cout)
            setattr(self, f"enc{lvl}", blk)
        self.enc1 = block(E1, 128, 64)
        self.final = nn.Conv2d(64, num_classes, 1)

    @staticmethod
    def _conv_act(slots, idx, which, h):
        """conv `which` (0: first, 1: second) of a block, its BatchNorm when the variant has one, ReLU."""
        h = slots.at(idx[2 * which])(h)
        if idx[2 * which + 1] is not None:
            h = slots.at(idx[2 * which + 1])(h)
        return F.relu(h)

    def _drop(self, x, which: int, masks):
        if masks is not None:
            return x * masks[which] / (1.0 - self.dropout_p)
        if self.training and self.dropout_p > 0:
            return F.dropout(x, self.dropout_p, True)
        return x

    def forward(self, x: torch.Tensor, dropout_masks: Optional[List[torch.Tensor]] = None,
                taps: Optional[dict] = None) -> torch.Tensor:
        skips = []
        h = x
        I = self._idx
        for lvl in (1, 2, 3, 4):
            d = getattr(self, f"dec{lvl}").down
            h = self._conv_act(d, I["dec"], 0, h)
            h = self._conv_act(d, I["dec"], 1, h)
            if lvl == 4:
                h = self._drop(h, 0, dropout_masks)
            h = F.max_pool2d(h, 2, stride=2, ceil_mode=True)
            skips.append(h)
            if taps is not None:
                taps[f"dec{lvl}"] = h
        c = self.center
        h = self._conv_act(c, I["center"], 0, h)
        h = self._conv_act(c, I["center"], 1, h)
        h = self._drop(h, 1, dropout_masks)
        h = F.relu(c.at(I["center"][4])(h))
        if taps is not None:
            taps["center"] = h
        for lvl in (4, 3, 2):
            u = getattr(self, f"enc{lvl}").up
            h = torch.cat([h, _resize_bilinear(skips[lvl - 1], h.shape[2:])], 1)
            h = self._conv_act(u, I["enc"], 0, h)
            if taps is not None:
                taps[f"e{lvl}a"] = h
            h = self._conv_act(u, I["enc"], 1, h)
            if taps is not None:
                taps[f"e{lvl}b"] = h
            h = F.relu(u.at(I["enc"][4])(h))
            if taps is not None:
                taps[f"enc{lvl}"] = h
        h = torch.cat([h, _resize_bilinear(skips[0], h.shape[2:])], 1)
        h = self._conv_act(self.enc1, I["enc1"], 0, h)
        if taps is not None:
            taps["e1a"] = h
        h = self._conv_act(self.enc1, I["enc1"], 1, h)
        if taps is not None:
            taps["enc1"] = h
        return _resize_bilinear(self.final(h), x.shape[2:])


# --------------------------------------------------------------------------- Enet

def _act(channels: int, relu: bool) -> nn.Module:
    return nn.ReLU() if relu else nn.PReLU(channels)


class _Bottleneck(nn.Module):
    """One Enet bottleneck (enet.py:33-152).  kind in {'regular','down','up','dilated','asym'}."""

    def __init__(self, cin: int, cout: int, kind: str = "regular", dilation: int = 1, relu: bool = False):
        super().__init__()
        self.kind, self.cin, self.cout = kind, cin, cout
        mid = cout // 4
        k = 2 if kind == "down" else 1
        self.block1x1_1 = _Slots({0: nn.Conv2d(cin, mid, k, k, bias=False),
                                  1: nn.BatchNorm2d(mid, 1e-3), 2: _act(mid, relu)})
        if kind == "up":
            self.conv_before_unpool = _Slots({0: nn.Conv2d(cin, cout, 1, bias=False),
                                              1: nn.BatchNorm2d(cout, 1e-3)})
            core = nn.ConvTranspose2d(mid, mid, 3, stride=2, padding=1, output_padding=1)
        elif kind == "dilated":
            core = nn.Conv2d(mid, mid, 3, padding=dilation, dilation=dilation)
        elif kind == "asym":
            core = _Slots({0: nn.Conv2d(mid, mid, (5, 1), padding=(2, 0), bias=False),
                           1: nn.Conv2d(mid, mid, (1, 5), padding=(0, 2))})
        else:  # regular / down
            core = nn.Conv2d(mid, mid, 3, padding=1)
        self.middle_block = _Slots({0: core, 1: nn.BatchNorm2d(mid, 1e-3), 2: _act(mid, relu)})
        self.block1x1_2 = _Slots({0: nn.Conv2d(mid, cout, 1, bias=False),
                                  1: nn.BatchNorm2d(cout, 1e-3), 2: _act(cout, relu)})

    @staticmethod
    def _run(slots: _Slots, x):
        core = slots.at(0)
        if isinstance(core, _Slots):
            x = core.at(1)(core.at(0)(x))
        else:
            x = core(x)
        return slots.at(2)(slots.at(1)(x))

    def forward(self, x, unpool_idx=None):
        idx = None
        if self.kind == "down":
            main, idx = F.max_pool2d(x, 2, stride=2, return_indices=True)
            extra = self.cout - self.cin
            if extra:
                main = torch.cat([main, main.new_zeros(main.shape[0], extra, main.shape[2], main.shape[3])], 1)
        elif self.kind == "up":
            cb = self.conv_before_unpool
            main = F.max_unpool2d(cb.at(1)(cb.at(0)(x)), unpool_idx, 2)
        else:
            main = x
        other = self._run(self.block1x1_2, self._run(self.middle_block, self._run(self.block1x1_1, x)))
        out = F.relu(main + other)  # enet.py:146-149 (Dropout2d is constructed but never applied)
        return (out, idx) if self.kind == "down" else out


class _Initial(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(1, 13, 3, stride=2, padding=1)
        self.batch_norm = nn.BatchNorm2d(13, 1e-3)
        self.prelu = nn.PReLU(13)

    def forward(self, x):
        return torch.cat([self.prelu(self.batch_norm(self.conv(x))), F.max_pool2d(x, 2, stride=2)], 1)


_STAGE23 = [("regular", 1), ("dilated", 2), ("asym", 1), ("dilated", 4),
            ("regular", 1), ("dilated", 8), ("asym", 1), ("dilated", 16)]


class _Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.order: List[str] = []

        def add(name, mod):
            self.add_module(name, mod)
            self.order.append(name)

        add("initial", _Initial())
        add("bottleneck_1_0", _Bottleneck(14, 64, "down"))
        for i in range(1, 5):
            add(f"bottleneck_1_{i}", _Bottleneck(64, 64))
        add("bottleneck_2_0", _Bottleneck(64, 128, "down"))
        for stage in (2, 3):
            for i, (kind, dil) in enumerate(_STAGE23, start=1):
                add(f"bottleneck_{stage}_{i}", _Bottleneck(128, 128, kind, dil))

    def forward(self, x):
        stack = []
        for name in self.order:
            mod = getattr(self, name)
            if getattr(mod, "kind", "") == "down":
                x, idx = mod(x)
                stack.append(idx)
            else:
                x = mod(x)
        return x, stack


class _Decoder(nn.Module):
    def __init__(self, num_classes: int):
        super().__init__()
        self.layers = nn.ModuleList([
            _Bottleneck(128, 64, "up", relu=True), _Bottleneck(64, 64, relu=True), _Bottleneck(64, 64, relu=True),
            _Bottleneck(64, 14, "up", relu=True), _Bottleneck(14, 14, relu=True),
            nn.ConvTranspose2d(14, num_classes, 2, stride=2)])

    def forward(self, x, stack):
        for mod in self.layers:
            if getattr(mod, "kind", "") == "up":
                x = mod(x, stack.pop())
            else:
                x = mod(x)
        return x


class Enet(nn.Module):
    def __init__(self, num_classes: int = 2):
        super().__init__()
        self.encoder = _Encoder()
        self.decoder = _Decoder(num_classes)

    def forward(self, x):
        h, stack = self.encoder(x)
        return self.decoder(h, stack)


def init_weights(net: nn.Module) -> nn.Module:
    """arch/__init__.py:60-65: xavier_normal on conv/convT weights, BN gamma~N(1,.02), beta=0."""
    for m in net.modules():
        if type(m) in (nn.Conv2d, nn.ConvTranspose2d):
            nn.init.xavier_normal_(m.weight.data)
        elif type(m) is nn.BatchNorm2d:
            m.weight.data.normal_(1.0, 0.02)
            m.bias.data.fill_(0)
    return net


def build_net(name: str, num_classes: int, **kw) -> nn.Module:
    if name == "unet":
        return init_weights(UNet(num_classes=num_classes, **kw))
    if name == "unet_bn":
        return init_weights(UNet(num_classes=num_classes, batchnorm=True, **kw))
    if name == "enet":
        return init_weights(Enet(num_classes=num_classes))
    raise ValueError(f"oracle has no arch {name!r}")
