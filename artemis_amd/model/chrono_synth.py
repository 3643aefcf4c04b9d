"""ChronoSynth, the only caller of the synthesis op (reference: model/chrono_synth.py:6-130),
restated on top of the fused head op.  Same constructor arguments, same sub-module names and
therefore the same ``state_dict`` keys (``ModuleWeight.0.weight`` ... ``feature_fuse.conv.0.bias``),
so a checkpoint trained upstream loads into it unchanged; same ``forward(features, frames,
output_size, output_frame_times)``.

What differs is only how lines :114-129 run: instead of, per context frame, three
``.contiguous()`` copies, a ReplicationPad2d copy, the CuPy op, ``occ *`` and a Python ``sum``,
the four sub-network outputs are handed as they are to ``artemis_amd.chrono_head`` (one forward
kernel, one backward kernel).  ``fused=False`` runs the reference's composition with this
library's per-frame ``FunctionSynth`` instead (bit-identical output; used by the tests).

There is no CuPy import: the late ``import cupy_module.synth`` of :66-67 is gone.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..synth import FunctionSynth, chrono_head

#: context-frame times of the septuplet setup (model/chrono_synth.py:90-93)
CONTEXT_FRAME_TIMES = (-0.25, 0.0, 1.0, 1.25)


class _SizedSequential(nn.Sequential):
    """nn.Sequential whose transposed convolutions are told the output size
    (helper_modules.py:54-61): the stride-2 up-sampling must land exactly on (H, W)."""

    def forward(self, x, output_size):
        for layer in self:
            x = layer(x, output_size) if isinstance(layer, nn.ConvTranspose2d) else layer(x)
        return x


class _Conv2d(nn.Module):
    """Conv wrapped as ``.conv = Sequential(Conv2d)`` to keep the upstream parameter names
    (helper_modules.py:25-36, batchnorm=False case)."""

    def __init__(self, in_ch, out_ch, kernel_size, bias):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, kernel_size=kernel_size, bias=bias))

    def forward(self, x):
        return self.conv(x)


def _subnet(c_in, c_mid, c_up, c_out, last):
    """conv3-lrelu-conv3-lrelu-convT(stride 2)-conv3-<last>: the shape shared by the weight,
    offset and occlusion sub-networks (model/chrono_synth.py:12-56)."""
    return _SizedSequential(
        nn.Conv2d(c_in, c_mid, 3, 1, 1),
        nn.LeakyReLU(0.2),
        nn.Conv2d(c_mid, c_up, 3, 1, 1),
        nn.LeakyReLU(0.2),
        nn.ConvTranspose2d(c_up, c_up, 3, stride=2, padding=1),
        nn.Conv2d(c_up, c_out, 3, 1, 1),
        last,
    )


class ChronoSynth(nn.Module):
    def __init__(self, num_inputs, num_features, kernel_size, dilation, apply_softmax=True,
                 fused=True):
        super().__init__()
        c = num_features + 1          # + the time channel (:10)
        taps = kernel_size ** 2
        self.num_inputs = num_inputs
        self.kernel_size = kernel_size
        self.kernel_pad = int(((kernel_size - 1) * dilation) / 2.0)
        self.dilation = dilation
        self.fused = fused

        self.modulePad = nn.ReplicationPad2d([self.kernel_pad] * 4)
        self.ModuleWeight = _subnet(c, num_features, taps, taps,
                                    nn.Softmax(1) if apply_softmax else nn.Identity())
        self.ModuleAlpha = _subnet(c, num_features, taps, taps, nn.Identity())
        self.ModuleBeta = _subnet(c, num_features, taps, taps, nn.Identity())
        self.ModuleOcclusion = _subnet(c, num_features, num_features, num_inputs, nn.Softmax(dim=1))
        self.feature_fuse = _Conv2d(c * num_inputs, c, kernel_size=1, bias=True)
        self.lrelu = nn.LeakyReLU(0.2)


    def operands(self, features, output_size, output_frame_times):
        """Lines :83-112: the four sub-network outputs for one pyramid level."""
        H, W = output_size
        B, C, T, h, w = features.shape
        # The reference indexes its time list with i = -T//2+1 .. T//2 (:95-98), so for T = 4 the
        # first frame is paired with context_frame_times[-1] = 1.25 (Python's negative index):
        # frame order (1.25, -0.25, 0, 1).  Trained checkpoints embody that pairing; keep it.
        times = torch.as_tensor([CONTEXT_FRAME_TIMES[i] for i in range(-T // 2 + 1, T // 2 + 1)],
                                device=features.device, dtype=features.dtype)
        # |t_context - t_out| as one extra channel per frame (:88-102)
        dt = (times.view(1, T) - output_frame_times.to(features.dtype).view(B, 1)).abs()
        time_tensor = dt.view(B, 1, T, 1, 1).expand(B, 1, T, h, w)
        features = torch.cat([features, time_tensor], 1)
        occ = self.lrelu(self.feature_fuse(torch.cat(torch.unbind(features, 1), 1)))
        features = features.transpose(1, 2).reshape(B * T, C + 1, h, w)
        weights = self.ModuleWeight(features, (H, W)).view(B, T, -1, H, W)
        alphas = self.ModuleAlpha(features, (H, W)).view(B, T, -1, H, W)
        betas = self.ModuleBeta(features, (H, W)).view(B, T, -1, H, W)
        occlusion = self.ModuleOcclusion(occ, (H, W))
        return weights, alphas, betas, occlusion

    def forward(self, features, frames, output_size, output_frame_times):
        H, W = output_size
        weights, alphas, betas, occlusion = self.operands(features, output_size, output_frame_times)
        resized = [F.interpolate(frames[i], size=(H, W), mode="bilinear")
                   for i in range(self.num_inputs)]
        if self.fused and self.kernel_size == 5 and self.dilation == 1:
            return chrono_head([f.contiguous() for f in resized], weights.contiguous(),
                               alphas.contiguous(), betas.contiguous(), occlusion.contiguous(), 1)
        warp = []
        for i in range(self.num_inputs):   # the reference's loop, :114-129
            warp.append(occlusion[:, i:i + 1] * FunctionSynth.apply(
                self.modulePad(resized[i]), weights[:, i].contiguous(), alphas[:, i].contiguous(),
                betas[:, i].contiguous(), self.dilation))
        return sum(warp)
