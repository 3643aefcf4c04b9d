"""ArTEMIS, the model around the synthesis hot path, on stock PyTorch ops + this package's fused
ChronoSynth head (reference: model/artemis.py:8-90, model/helper_modules.py:5-51).  Harness code
(SURVEY.md 8f-2): it exists so that BASELINE configs 1, 3 and 4 run at model level, CuPy-free.

Same constructor, same forward and the same ``state_dict`` keys as upstream (``encoder.*``,
``decoder.{0,1,2}.upconv.*``, ``smooth{1,2,3}.{0.conv.0,1.conv1,1.conv2}.*``,
``predict{1,2,3}.*``): 29,090,528 parameters for the large configuration.

Beyond the reference's forward: everything up to the three ChronoSynth heads depends on the
context frames only, not on the output time, so ``interpolate`` runs encoder, decoder and
SmoothNets ONCE for a 4-frame window and evaluates all requested timesteps as one batch through the
heads (BASELINE configs[3]: 7 timesteps = one N = 7 launch per head and direction), where
upstream's interpolate mode recomputes the whole model per timestep (main.py:118-121, despite the
remark at model/artemis.py:66).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .chrono_synth import ChronoSynth
from .encoder import ResBlock, SepSTSEncoder


class UpSplit(nn.Module):
    """x2 spatial up-sampling by a transposed 3-D convolution told its exact output size
    (model/helper_modules.py:5-13)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.upconv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size=3,
                                         stride=(1, 2, 2), padding=1)

    def forward(self, x, output_size):
        return self.upconv(x, output_size=output_size)


class Conv3dBlock(nn.Module):
    """``.conv = Sequential(Conv3d)``: the upstream parameter path (model/helper_modules.py:39-51,
    batchnorm=False)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride,
                                            padding=padding, bias=bias))

    def forward(self, x):
        return self.conv(x)


def _smooth_net(in_channels, out_channels):
    return nn.Sequential(Conv3dBlock(in_channels, out_channels, 3, stride=1, padding=1),
                         ResBlock(out_channels, 3))


class ArTEMIS(nn.Module):
    #: channels of the encoder levels x4..x1 (the reference's "large model", model/artemis.py:12)
    NUM_FEATURES = (512, 256, 128, 64)
    NUM_HEADS = (2, 4, 8, 16)
    WINDOW = (1, 8, 8)
    HEAD_FEATURES = 64

    def __init__(self, num_inputs=4, joinType="concat", kernel_size=5, dilation=1, fused=True):
        super().__init__()
        nf = list(self.NUM_FEATURES)
        self.joinType = joinType
        self.num_inputs = num_inputs
        growth = 2 if joinType == "concat" else 1
        self.lrelu = nn.LeakyReLU(0.2)
        self.encoder = SepSTSEncoder(nf, num_inputs, [self.WINDOW] * 4, list(self.NUM_HEADS))
        self.decoder = nn.Sequential(UpSplit(nf[0], nf[1]), UpSplit(nf[1] * growth, nf[2]),
                                     UpSplit(nf[2] * growth, nf[3]))
        self.smooth1 = _smooth_net(nf[1] * growth, self.HEAD_FEATURES)
        self.smooth2 = _smooth_net(nf[2] * growth, self.HEAD_FEATURES)
        self.smooth3 = _smooth_net(nf[3] * growth, self.HEAD_FEATURES)
        # the coarsest head predicts normalised kernels, the finer two residual ones (:43-48)
        self.predict1 = ChronoSynth(num_inputs, self.HEAD_FEATURES, kernel_size, dilation,
                                    apply_softmax=True, fused=fused)
        self.predict2 = ChronoSynth(num_inputs, self.HEAD_FEATURES, kernel_size, dilation,
                                    apply_softmax=False, fused=fused)
        self.predict3 = ChronoSynth(num_inputs, self.HEAD_FEATURES, kernel_size, dilation,
                                    apply_softmax=False, fused=fused)

    def _join(self, a, b):
        if self.joinType == "concat":
            return torch.cat([a, b], dim=1)
        return a + b if self.joinType == "add" else a

    def features(self, frames):
        """The time-independent part (model/artemis.py:57-80): three feature pyramids levels
        (B, 64, T, h, w) at 1/8, 1/4 and 1/2 of the frame size, and the head output sizes."""
        images = torch.stack(frames, dim=2)                                  # (B, 3, T, H, W)
        images = images - images.mean(dim=(2, 3, 4), keepdim=True)          # per-sample mean
        x0, x1, x2, x3, x4 = self.encoder(images)
        dx3 = self._join(self.lrelu(self.decoder[0](x4, x3.size())), x3)
        dx2 = self._join(self.lrelu(self.decoder[1](dx3, x2.size())), x2)
        dx1 = self._join(self.lrelu(self.decoder[2](dx2, x1.size())), x1)
        feats = (self.smooth1(dx3), self.smooth2(dx2), self.smooth3(dx1))
        sizes = (tuple(x2.shape[-2:]), tuple(x1.shape[-2:]), tuple(x0.shape[-2:]))
        return feats, sizes

    def heads(self, feats, sizes, frames, output_frame_times):
        """Coarse-to-fine synthesis (model/artemis.py:82-88): each level adds its residual to the
        bilinearly up-sampled result of the level below."""
        ll = self.predict1(feats[0], frames, sizes[0], output_frame_times)
        l = self.predict2(feats[1], frames, sizes[1], output_frame_times)
        l = F.interpolate(ll, size=l.shape[-2:], mode="bilinear") + l
        out = self.predict3(feats[2], frames, sizes[2], output_frame_times)
        out = F.interpolate(l, size=out.shape[-2:], mode="bilinear") + out
        return ll, l, out

    def forward(self, frames, output_frame_times):
        feats, sizes = self.features(frames)
        return self.heads(feats, sizes, frames, output_frame_times)

    @torch.no_grad()
    def interpolate(self, frames, timesteps):
        """All `timesteps` of ONE window of context frames (each (1, 3, H, W)) in one batch:
        encoder/decoder once, heads at batch N = len(timesteps).  Returns (N, 3, H, W)."""
        t = torch.as_tensor(timesteps, dtype=frames[0].dtype, device=frames[0].device).flatten()
        if frames[0].size(0) != 1:
            raise ValueError("interpolate takes one window of context frames (batch 1)")
        feats, sizes = self.features(frames)
        n = t.numel()
        feats = [f.expand(n, *f.shape[1:]) for f in feats]
        frames_n = [f.expand(n, *f.shape[1:]) for f in frames]
        return self.heads(feats, sizes, frames_n, t)[2]
