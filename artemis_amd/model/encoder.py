"""The Sep-STS encoder of ArTEMIS on stock PyTorch ops (reference: model/sep_sts_encoder.py:5-67,
model/sep_sts_layer.py:10-401).  Harness code around the synthesis hot path: no
custom kernels here; it exists so that configs 1, 3 and 4 run at model level from this repository.

Own structure, upstream parameter names: every sub-module carries the attribute name the
reference gives it (``stem``, ``stage1..4.upper.blocks.N.{norm1,depth_attn,point_attn,norm2,mlp}``,
``down0..3``; inside an attention ``relative_position_bias_table``, ``relative_position_indices``
(a buffer), ``qkv``, ``project``; inside the MLP ``linear1``, ``linear2``), so an upstream
``state_dict`` loads unchanged.

What a block computes (model/sep_sts_layer.py:262-320), restated:

* spatial attention inside (1, 8, 8) windows of each frame, every second block on windows shifted
  by (0, 4, 4) with the usual cyclic roll + region mask (-100 between regions);
* temporal attention over the T = 4 frames at each pixel ((T, 1, 1) windows, no shift, no mask);
* residual, then LayerNorm + GELU MLP (ratio 4) with its own residual.

Attention itself goes through ``torch.nn.functional.scaled_dot_product_attention`` with the
relative-position bias (+ shift mask) as an additive mask: one fused kernel on ROCm instead of the
reference's explicit ``softmax(q k^T + bias) v``.  ``timm`` and ``einops`` are not needed
(``trunc_normal_`` is ``torch.nn.init.trunc_normal_``).
"""
from __future__ import annotations

from functools import lru_cache

import torch
import torch.nn as nn
import torch.nn.functional as F


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.linear1 = nn.Linear(dim, hidden)
        self.activation = nn.GELU()
        self.linear2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.linear2(self.activation(self.linear1(x)))


def _relative_position_indices(window):
    """(N, N) index into the bias table for every (query, key) pair of a (Wd, Wh, Ww) window:
    the three coordinate differences, each shifted to start at 0, in mixed radix
    (2Wd-1, 2Wh-1, 2Ww-1) -- model/sep_sts_layer.py:134-162."""
    axes = [torch.arange(n) for n in window]
    coords = torch.stack(torch.meshgrid(*axes, indexing="ij")).flatten(1)          # (3, N)
    diff = coords[:, :, None] - coords[:, None, :]                                  # (3, N, N)
    index = torch.zeros(diff.shape[1:], dtype=torch.long)
    for axis, n in enumerate(window):
        index = index * (2 * n - 1) + (diff[axis] + n - 1)
    return index


class WindowAttention3D(nn.Module):
    """Multi-head self-attention inside one window with a learned relative-position bias
    (model/sep_sts_layer.py:108-219)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True):
        super().__init__()
        self.window_size = tuple(window_size)
        self.num_heads = num_heads
        table = 1
        for n in self.window_size:
            table *= 2 * n - 1
        self.relative_position_bias_table = nn.Parameter(torch.zeros(table, num_heads))
        self.register_buffer("relative_position_indices",
                             _relative_position_indices(self.window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.project = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def bias(self, n_tokens):
        idx = self.relative_position_indices[:n_tokens, :n_tokens].reshape(-1)
        return self.relative_position_bias_table[idx].view(n_tokens, n_tokens, -1).permute(2, 0, 1)

    def forward(self, x, mask=None):
        """x: (windows * B, N, C); mask: (windows, N, N) additive (0 / -100) or None."""
        Bn, N, C = x.shape
        h = self.num_heads
        q, k, v = self.qkv(x).view(Bn, N, 3, h, C // h).permute(2, 0, 3, 1, 4)     # (Bn, h, N, hd)
        add = self.bias(N).unsqueeze(0)                                             # (1, h, N, N)
        if mask is not None:
            nW = mask.shape[0]
            add = add + mask.unsqueeze(1)                                           # (nW, h, N, N)
            q, k, v = (t.view(Bn // nW, nW, h, N, C // h) for t in (q, k, v))
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=add.to(q.dtype))
        out = out.reshape(Bn, h, N, C // h).transpose(1, 2).reshape(Bn, N, C)
        return self.project(out)


def _windows(x, ws):
    """(B, D, H, W, C) -> (B * windows, Wd*Wh*Ww, C); D, H, W multiples of the window."""
    B, D, H, W, C = x.shape
    x = x.view(B, D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, ws[0] * ws[1] * ws[2], C)


def _unwindows(t, ws, B, D, H, W):
    t = t.view(B, D // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], -1)
    return t.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, -1)


def _fit_window(size, window, shift=None):
    """A window no larger than the tensor; an axis that fits in one window is not shifted
    (model/sep_sts_layer.py:75-105)."""
    ws = tuple(s if s <= w else w for s, w in zip(size, window))
    if shift is None:
        return ws
    return ws, tuple(0 if s <= w else sh for s, w, sh in zip(size, window, shift))


@lru_cache(maxsize=32)
def _shift_mask(size, ws, shift, device):
    """Additive mask (windows, N, N) for attention on the cyclically shifted tensor: tokens that
    come from different sides of the wrap-around must not attend to each other
    (model/sep_sts_layer.py:324-337).  Per axis the padded extent splits into up to three runs --
    [0, P - window), [P - window, P - shift), [P - shift, P) -- and an unshifted axis is one run."""
    label = torch.zeros(size, dtype=torch.long, device=device)
    for axis, (P, w, s) in enumerate(zip(size, ws, shift)):
        run = torch.full((P,), 2, dtype=torch.long, device=device)
        if s > 0:
            pos = torch.arange(P, device=device)
            run = (pos >= P - w).long() + (pos >= P - s).long()
        shape = [1, 1, 1]
        shape[axis] = P
        label = label * 3 + run.view(shape)
    lab = _windows(label.view(1, *size, 1).float(), ws).squeeze(-1)                 # (nW, N)
    differ = lab.unsqueeze(1) != lab.unsqueeze(2)
    return torch.zeros(differ.shape, device=device).masked_fill(differ, -100.0)


class SepSTSBlock(nn.Module):
    def __init__(self, dim, num_heads, depth_window_size, shift_size, point_window_size,
                 mlp_ratio=4.0):
        super().__init__()
        self.depth_window_size = tuple(depth_window_size)
        self.point_window_size = tuple(point_window_size)
        self.shift_size = tuple(shift_size)
        self.norm1 = nn.LayerNorm(dim)
        self.depth_attn = WindowAttention3D(dim, self.depth_window_size, num_heads)
        self.point_attn = WindowAttention3D(dim, self.point_window_size, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def attend(self, x):
        B, D, H, W, C = x.shape
        ws, shift = _fit_window((D, H, W), self.depth_window_size, self.shift_size)
        x = self.norm1(x)
        pad = [(-s) % w for s, w in zip((D, H, W), ws)]                 # up to window multiples
        if any(pad):
            x = F.pad(x, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
        Dp, Hp, Wp = D + pad[0], H + pad[1], W + pad[2]
        shifted = any(s > 0 for s in shift)
        if shifted:
            x = torch.roll(x, shifts=[-s for s in shift], dims=(1, 2, 3))
        mask = _shift_mask((Dp, Hp, Wp), ws, shift, x.device) if shifted else None
        x = _unwindows(self.depth_attn(_windows(x, ws), mask), ws, B, Dp, Hp, Wp)
        if shifted:
            x = torch.roll(x, shifts=list(shift), dims=(1, 2, 3))
        if any(pad):
            x = x[:, :D, :H, :W].contiguous()
        # attention over time at each pixel
        pw = _fit_window((D, H, W), self.point_window_size)
        return _unwindows(self.point_attn(_windows(x, pw)), pw, B, D, H, W)

    def forward(self, x):
        x = x + self.attend(x)
        return x + self.mlp(self.norm2(x))


class SepSTSBasicLayer(nn.Module):
    """`depth` blocks; odd ones work on windows shifted by half a window."""

    def __init__(self, dim, depth, num_heads, depth_window_size, point_window_size):
        super().__init__()
        half = tuple(w // 2 for w in depth_window_size)
        self.blocks = nn.ModuleList([
            SepSTSBlock(dim, num_heads, depth_window_size, half if i % 2 else (0, 0, 0),
                        point_window_size) for i in range(depth)])

    def forward(self, x):                      # (B, C, D, H, W)
        x = x.permute(0, 2, 3, 4, 1)
        for blk in self.blocks:
            x = blk(x)
        return x.permute(0, 4, 1, 2, 3).contiguous()


class SepSTSLayer(nn.Module):
    def __init__(self, plane, depth, num_frames, num_heads, window_size):
        super().__init__()
        self.upper = SepSTSBasicLayer(plane, depth, num_heads, tuple(window_size),
                                      (num_frames, 1, 1))

    def forward(self, x):
        return self.upper(x)


class ResBlock(nn.Module):
    """relu(conv2(relu(conv1(x))) + x) on 3-D convolutions (model/sep_sts_encoder.py:16-30)."""

    def __init__(self, channel, kernel_size):
        super().__init__()
        p = (kernel_size - 1) // 2
        self.conv1 = nn.Conv3d(channel, channel, kernel_size, stride=1, padding=p)
        self.conv2 = nn.Conv3d(channel, channel, kernel_size, stride=1, padding=p)
        self.relu = nn.ReLU()

    def forward(self, x):
        return self.relu(self.conv2(self.relu(self.conv1(x))) + x)


class SepSTSEncoder(nn.Module):
    """Five feature levels x0..x4 at 1, 1/2, 1/4, 1/8, 1/16 of the frame size (ceil-halving:
    stride-(1,2,2) convolutions with kernel 3, padding 1), channels nf[-1]/2, nf[-1], ..., nf[0]."""

    def __init__(self, nf, num_frames, window_size, num_heads):
        super().__init__()
        c = list(reversed(nf))                   # (64, 128, 256, 512) for the large model
        self.stem = nn.Sequential(nn.Conv3d(3, c[0] // 2, 3, stride=1, padding=1),
                                  nn.LeakyReLU(0.2), ResBlock(c[0] // 2, 3))
        depths = (2, 2, 6, 2)
        for i in range(4):
            setattr(self, f"stage{i + 1}", SepSTSLayer(c[i], depths[i], num_frames, num_heads[i],
                                                       window_size[i]))
        chans = [c[0] // 2] + c
        for i in range(4):
            setattr(self, f"down{i}", nn.Conv3d(chans[i], chans[i + 1], 3, stride=(1, 2, 2),
                                                padding=1))

    def forward(self, x):
        feats = [self.stem(x)]
        for i in range(4):
            feats.append(getattr(self, f"stage{i + 1}")(getattr(self, f"down{i}")(feats[-1])))
        return feats
