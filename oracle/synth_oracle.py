"""CPU oracle for the ArTEMIS frame-synthesis op (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``artemis_amd`` never does.

PARITY UNPINNED: the reference ships no golden vectors for this path and cannot run here
(``cupy_module/synth.py:1`` imports CuPy, ``:310-311`` raises on CPU).  See ``synth_oracle.c``.

Two independent restatements of ``/root/reference/cupy_module/synth.py`` live here:

* the scalar C library ``libsynth_oracle.so`` (``synth_oracle.c``), reference operation order,
  reached through ``forward`` / ``grad_weight`` / ``grad_alpha`` / ``grad_beta`` / ``grad_input``;
* ``forward_torch``: a vectorised, differentiable torch restatement (25-tap Python loop over
  whole planes).  Autograd through it yields the true gradInput the reference never computes
  (``cupy_module/synth.py:333,391``) and cross-checks the C library.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsynth_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)


def build(force: bool = False) -> str:
    """Compile ``synth_oracle.c`` with gcc (``make -C oracle``)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "synth_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE, "-B", "libsynth_oracle.so"], check=True,
                       capture_output=True)
    return _LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        six = [ctypes.c_int] * 6
        lib.synth_oracle_forward.argtypes = [_f32p] * 5 + six
        lib.synth_oracle_grad_weight.argtypes = [_f32p] * 5 + six
        lib.synth_oracle_grad_alpha.argtypes = [_f32p] * 6 + six
        lib.synth_oracle_grad_beta.argtypes = [_f32p] * 6 + six
        lib.synth_oracle_grad_input.argtypes = [_f32p] * 5 + six
        ten = [ctypes.c_int] * 10
        lib.synth_oracle_forward_ex.argtypes = [_f32p] * 5 + ten
        lib.synth_oracle_grad_weight_ex.argtypes = [_f32p] * 5 + ten
        lib.synth_oracle_grad_alpha_ex.argtypes = [_f32p] * 6 + ten
        lib.synth_oracle_grad_beta_ex.argtypes = [_f32p] * 6 + ten
        lib.synth_oracle_grad_input_ex.argtypes = [_f32p] * 5 + ten
        for name in ("forward", "grad_weight", "grad_alpha", "grad_beta", "grad_input"):
            getattr(lib, "synth_oracle_" + name).restype = None
            getattr(lib, "synth_oracle_" + name + "_ex").restype = None
        lib.synth_oracle_set_threads.argtypes = [ctypes.c_int]
        lib.synth_oracle_set_threads.restype = None
        lib.synth_oracle_get_threads.restype = ctypes.c_int
        lib.synth_oracle_has_openmp.restype = ctypes.c_int
        _lib = lib
    return _lib


def set_threads(n: int) -> None:
    load().synth_oracle_set_threads(int(n))


def _f32(a) -> np.ndarray:
    if hasattr(a, "detach"):  # torch tensor
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def _dims(input, weight, dilation):
    """Shape contract of FunctionSynth.forward: cupy_module/synth.py:269-278."""
    N, C, Hin, Win = input.shape
    K, H, W = weight.shape[1:]
    k = int(math.sqrt(K))
    assert k * k == K
    assert Hin - ((k - 1) * dilation + 1) == H - 1
    assert Win - ((k - 1) * dilation + 1) == W - 1
    return N, C, H, W, k, int(dilation)


def forward(input, weight, offset_y, offset_x, dilation=1) -> np.ndarray:
    input, weight, offset_y, offset_x = map(_f32, (input, weight, offset_y, offset_x))
    N, C, H, W, k, d = _dims(input, weight, dilation)
    out = np.zeros((N, C, H, W), np.float32)
    load().synth_oracle_forward(_ptr(input), _ptr(weight), _ptr(offset_y), _ptr(offset_x),
                                _ptr(out), N, C, H, W, k, d)
    return out


def grad_weight(grad_out, input, weight, offset_y, offset_x, dilation=1) -> np.ndarray:
    grad_out, input, weight, offset_y, offset_x = map(
        _f32, (grad_out, input, weight, offset_y, offset_x))
    N, C, H, W, k, d = _dims(input, weight, dilation)
    g = np.zeros_like(weight)
    load().synth_oracle_grad_weight(_ptr(grad_out), _ptr(input), _ptr(offset_y), _ptr(offset_x),
                                    _ptr(g), N, C, H, W, k, d)
    return g


def grad_alpha(grad_out, input, weight, offset_y, offset_x, dilation=1) -> np.ndarray:
    grad_out, input, weight, offset_y, offset_x = map(
        _f32, (grad_out, input, weight, offset_y, offset_x))
    N, C, H, W, k, d = _dims(input, weight, dilation)
    g = np.zeros_like(weight)
    load().synth_oracle_grad_alpha(_ptr(grad_out), _ptr(input), _ptr(weight), _ptr(offset_y),
                                   _ptr(offset_x), _ptr(g), N, C, H, W, k, d)
    return g


def grad_beta(grad_out, input, weight, offset_y, offset_x, dilation=1) -> np.ndarray:
    grad_out, input, weight, offset_y, offset_x = map(
        _f32, (grad_out, input, weight, offset_y, offset_x))
    N, C, H, W, k, d = _dims(input, weight, dilation)
    g = np.zeros_like(weight)
    load().synth_oracle_grad_beta(_ptr(grad_out), _ptr(input), _ptr(weight), _ptr(offset_y),
                                  _ptr(offset_x), _ptr(g), N, C, H, W, k, d)
    return g


def grad_input(grad_out, input_shape, weight, offset_y, offset_x, dilation=1) -> np.ndarray:
    """Adjoint of ``forward`` w.r.t. input (fp64 accumulate). Not in the reference."""
    grad_out, weight, offset_y, offset_x = map(_f32, (grad_out, weight, offset_y, offset_x))
    N, C, Hin, Win = input_shape
    K, H, W = weight.shape[1:]
    k = int(math.sqrt(K))
    d = int(dilation)
    assert Hin == H + (k - 1) * d and Win == W + (k - 1) * d
    g = np.zeros((N, C, Hin, Win), np.float32)
    load().synth_oracle_grad_input(_ptr(grad_out), _ptr(weight), _ptr(offset_y), _ptr(offset_x),
                                   _ptr(g), N, C, H, W, k, d)
    return g


def backward(grad_out, input, weight, offset_y, offset_x, dilation=1, with_input=True):
    """(gradInput|None, gradWeight, gradOffset_y, gradOffset_x): cupy_module/synth.py:316-391."""
    gi = grad_input(grad_out, tuple(input.shape), weight, offset_y, offset_x, dilation) \
        if with_input else None
    return (gi,
            grad_weight(grad_out, input, weight, offset_y, offset_x, dilation),
            grad_alpha(grad_out, input, weight, offset_y, offset_x, dilation),
            grad_beta(grad_out, input, weight, offset_y, offset_x, dilation))


def _dims_ex(input, weight, dilation):
    N, C, Hin, Win = input.shape
    K, H, W = weight.shape[1:]
    k = int(math.sqrt(K))
    assert k * k == K
    return N, C, H, W, k, int(dilation), Hin, Win


def forward_ex(input, weight, offset_y, offset_x, dilation=1, origin=(0, 0)) -> np.ndarray:
    """Forward on a frame of any size: tap (r, s) of output pixel (y, x) reads around frame
    coordinate (y + origin[0] + r*d, x + origin[1] + s*d), clamped to the frame."""
    input, weight, offset_y, offset_x = map(_f32, (input, weight, offset_y, offset_x))
    N, C, H, W, k, d, Hin, Win = _dims_ex(input, weight, dilation)
    out = np.zeros((N, C, H, W), np.float32)
    load().synth_oracle_forward_ex(_ptr(input), _ptr(weight), _ptr(offset_y), _ptr(offset_x),
                                   _ptr(out), N, C, H, W, k, d, Hin, Win, int(origin[0]),
                                   int(origin[1]))
    return out


def backward_ex(grad_out, input, weight, offset_y, offset_x, dilation=1, origin=(0, 0),
                with_input=True):
    """(gradInput|None, gradWeight, gradOffset_y, gradOffset_x) of ``forward_ex``."""
    grad_out, input, weight, offset_y, offset_x = map(
        _f32, (grad_out, input, weight, offset_y, offset_x))
    N, C, H, W, k, d, Hin, Win = _dims_ex(input, weight, dilation)
    geo = (N, C, H, W, k, d, Hin, Win, int(origin[0]), int(origin[1]))
    lib = load()
    gw, ga, gb = (np.zeros_like(weight) for _ in range(3))
    lib.synth_oracle_grad_weight_ex(_ptr(grad_out), _ptr(input), _ptr(offset_y), _ptr(offset_x),
                                    _ptr(gw), *geo)
    lib.synth_oracle_grad_alpha_ex(_ptr(grad_out), _ptr(input), _ptr(weight), _ptr(offset_y),
                                   _ptr(offset_x), _ptr(ga), *geo)
    lib.synth_oracle_grad_beta_ex(_ptr(grad_out), _ptr(input), _ptr(weight), _ptr(offset_y),
                                  _ptr(offset_x), _ptr(gb), *geo)
    gi = None
    if with_input:
        gi = np.zeros_like(input)
        lib.synth_oracle_grad_input_ex(_ptr(grad_out), _ptr(weight), _ptr(offset_y),
                                       _ptr(offset_x), _ptr(gi), *geo)
    return gi, gw, ga, gb


def forward_torch(input, weight, offset_y, offset_x, dilation=1):
    """Vectorised differentiable restatement of kernel_Synth_updateOutput
    (cupy_module/synth.py:16-50); works in the dtype of its arguments (fp32 or fp64).

    ``trunc`` is the C ``(int)`` cast (:29-30), the four indices are clamped to the padded frame
    (:32-35), fractions are taken against the truncated value (:37-38) so negative offsets
    extrapolate.  Gradients w.r.t. the offsets flow only through the fractions, which is exactly
    the reference's analytic derivative (:140-145, :190-195: clamp and trunc ignored).
    """
    import torch

    N, C, Hin, Win = input.shape
    K, H, W = weight.shape[1:]
    k = int(math.sqrt(K))
    d = int(dilation)
    assert Hin == H + (k - 1) * d and Win == W + (k - 1) * d
    ys = torch.arange(H, device=input.device).view(1, H, 1)
    xs = torch.arange(W, device=input.device).view(1, 1, W)
    flat = input.reshape(N, C, Hin * Win)
    out = input.new_zeros(N, C, H, W)

    def gather(iy, ix):
        idx = (iy * Win + ix).view(N, 1, H * W).expand(N, C, H * W)
        return torch.gather(flat, 2, idx).view(N, C, H, W)

    for row in range(k):
        for col in range(k):
            t = row * k + col
            w = weight[:, t]
            alpha = offset_y[:, t]
            beta = offset_x[:, t]
            ia = torch.trunc(alpha.detach())
            ib = torch.trunc(beta.detach())
            a = (alpha - ia).unsqueeze(1)
            b = (beta - ib).unsqueeze(1)
            # int32 saturation of the cast, then int64 index math (no wrap for sane offsets)
            ia = ia.clamp(-2**31, 2**31 - 1).nan_to_num(0.0).long()
            ib = ib.clamp(-2**31, 2**31 - 1).nan_to_num(0.0).long()
            yy = ys + row * d + ia
            xx = xs + col * d + ib
            bottom = yy.clamp(0, Hin - 1)
            top = (yy + 1).clamp(0, Hin - 1)
            left = xx.clamp(0, Win - 1)
            right = (xx + 1).clamp(0, Win - 1)
            out = out + w.unsqueeze(1) * (
                gather(bottom, left) * (1 - a) * (1 - b)
                + gather(top, left) * a * (1 - b)
                + gather(bottom, right) * (1 - a) * b
                + gather(top, right) * a * b
            )
    return out


# ---------------------------------------------------------------------------------------------
# The ChronoSynth head around the op: model/chrono_synth.py:114-129 restated with numpy around
# the C oracle (fp32, the reference's operation order: pad, op, `occ *`, Python `sum`).

def head_forward(frames, weights, alphas, betas, occlusion, dilation=1):
    """``framet`` of ChronoSynth.forward for operands already at output size:
    frames: T arrays (B, C, H, W) unpadded; weights/alphas/betas (B, T, k*k, H, W);
    occlusion (B, T, H, W).  Returns (framet, [S_i]) with S_i the per-frame op outputs."""
    weights, alphas, betas, occlusion = map(_f32, (weights, alphas, betas, occlusion))
    T = weights.shape[1]
    k = int(math.sqrt(weights.shape[2]))
    p = int(((k - 1) * dilation) / 2.0)                       # kernel_pad, :60
    warp, synth = [], []
    for i in range(T):                                        # :115
        frame = _f32(frames[i])
        padded = np.pad(frame, ((0, 0), (0, 0), (p, p), (p, p)), mode="edge")   # modulePad, :63,125
        s_i = forward(padded, weights[:, i], alphas[:, i], betas[:, i], dilation)   # :125-126
        synth.append(s_i)
        warp.append(occlusion[:, i:i + 1] * s_i)              # :123-124 (fp32 product)
    framet = warp[0]
    for wpi in warp[1:]:                                      # sum(warp), :129: left to right
        framet = framet + wpi
    return framet.astype(np.float32), synth


def head_backward(grad_out, frames, weights, alphas, betas, occlusion, dilation=1):
    """Gradients of ``head_forward``'s framet w.r.t. (weights, alphas, betas, occlusion) as
    autograd derives them: d(sum) passes grad_out to every warp_i; d(occ * S): gS = occ * g,
    gocc = sum_c g * S; then FunctionSynth.backward (cupy_module/synth.py:316-391) per frame."""
    grad_out, weights, alphas, betas, occlusion = map(
        _f32, (grad_out, weights, alphas, betas, occlusion))
    T = weights.shape[1]
    k = int(math.sqrt(weights.shape[2]))
    p = int(((k - 1) * dilation) / 2.0)
    gw, ga, gb = (np.zeros_like(weights) for _ in range(3))
    gocc = np.zeros_like(occlusion)
    for i in range(T):
        padded = np.pad(_f32(frames[i]), ((0, 0), (0, 0), (p, p), (p, p)), mode="edge")
        w_i, a_i, b_i = (np.ascontiguousarray(t[:, i]) for t in (weights, alphas, betas))
        s_i = forward(padded, w_i, a_i, b_i, dilation)
        gs = (occlusion[:, i:i + 1] * grad_out).astype(np.float32)
        gocc[:, i] = (grad_out.astype(np.float64) * s_i).sum(1)
        gw[:, i] = grad_weight(gs, padded, w_i, a_i, b_i, dilation)
        ga[:, i] = grad_alpha(gs, padded, w_i, a_i, b_i, dilation)
        gb[:, i] = grad_beta(gs, padded, w_i, a_i, b_i, dilation)
    return gw, ga, gb, gocc
