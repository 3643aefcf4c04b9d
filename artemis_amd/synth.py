"""FunctionSynth: the ArTEMIS frame-synthesis op on MI355X, same operator interface as the
reference's ``cupy_module/synth.py`` (paths relative to /root/reference).

Reference interface mirrored here (``cupy_module/synth.py:263-391``)::

    FunctionSynth.apply(input, weight, offset_y, offset_x, dilation) -> output
    forward(context, input, weight, offset_y, offset_x, dilation)
    backward(context, gradOutput) -> (gradInput, gradWeight, gradOffset_y, gradOffset_x, None)

called from ``model/chrono_synth.py:66-67,125-126``.  Same argument meaning, same asserts
(shape relation ``:277-278``, contiguity ``:280-283,331``), same ``NotImplementedError`` for CPU
tensors (``:310-311,388-389``).  The work is done by hand-written gfx950 kernels behind the C ABI
in ``include/artemis_synth.h``; torch only owns the memory and the stream.

Deliberate differences from the reference, all supersets:

* ``gradInput`` is the true adjoint when ``input.requires_grad`` (the reference returns unfilled
  zeros, ``:333,391``).  Accumulation is fp32 atomics, or bit-reproducible when
  ``options.deterministic`` or ``torch.use_deterministic_algorithms(True)`` is set.
* ``gradOffset_x`` is gated on ``needs_input_grad[3]`` (the reference tests index 2, ``:336``) and
  any subset of the three tap gradients may be unneeded (the reference dereferences ``None``,
  ``:343,358,374``).
* One fused backward kernel instead of three launches; tensors larger than 2^31 elements work.
"""
from __future__ import annotations

import contextlib
import ctypes
import math
from dataclasses import dataclass

import torch

from . import _lib


@dataclass
class SynthOptions:
    #: displacement radius (whole pixels) served from the LDS-staged frame tile;
    #: -2 = adaptive (the library moves between 8, 14 and 24 per tensor shape from what its forward
    #: kernels saw on earlier calls: no host synchronisation, include/artemis_synth.h
    #: ARTEMIS_SYNTH_RADIUS_AUTO), -1 = the fixed library default (14), 0 = no LDS staging (direct
    #: gathers).  Any value is correct for any offsets: taps displaced farther are recomputed
    #: through the direct path.  Results of different radii agree to fp32 summation order; with
    #: ``deterministic`` (or torch.use_deterministic_algorithms) adaptive means the fixed default.
    tile_radius: int = _lib.RADIUS_AUTO
    #: bit-reproducible gradInput (fixed-point integer accumulation) instead of fp32 atomics
    deterministic: bool = False


options = SynthOptions()


@contextlib.contextmanager
def synth_options(**kw):
    """Temporarily override fields of :data:`options`."""
    old = {k: getattr(options, k) for k in kw}
    try:
        for k, v in kw.items():
            setattr(options, k, v)
        yield options
    finally:
        for k, v in old.items():
            setattr(options, k, v)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _dims(input, weight, dilation):
    # cupy_module/synth.py:269-278 (forward) and :320-329 (backward)
    intSample = input.size(0)
    intInputDepth = input.size(1)
    intInputHeight = input.size(2)
    intInputWidth = input.size(3)
    intFilterSize = int(math.sqrt(weight.size(1)))
    intOutputHeight = weight.size(2)
    intOutputWidth = weight.size(3)
    assert intFilterSize * intFilterSize == weight.size(1), "weight.size(1) must be a square"
    assert (intInputHeight - ((intFilterSize - 1) * dilation + 1) == intOutputHeight - 1)
    assert (intInputWidth - ((intFilterSize - 1) * dilation + 1) == intOutputWidth - 1)
    return (intSample, intInputDepth, intOutputHeight, intOutputWidth, intFilterSize,
            intInputHeight, intInputWidth)


def _check_operands(input, weight, offset_y, offset_x, bf16_frame_ok=False):
    for name, t in (("input", input), ("weight", weight), ("offset_y", offset_y),
                    ("offset_x", offset_x)):
        if t.dim() != 4:
            raise ValueError(f"{name} must be 4-D NCHW, got {tuple(t.shape)}")
        if name == "input" and bf16_frame_ok and t.dtype == torch.bfloat16:
            continue
        if t.dtype != torch.float32:
            raise TypeError(f"{name} must be float32 (the reference is fp32 only), got {t.dtype}")
        if t.device != input.device:
            raise ValueError(f"{name} is on {t.device}, input on {input.device}")
    if weight.shape != offset_y.shape or weight.shape != offset_x.shape:
        raise ValueError("weight, offset_y and offset_x must have the same shape")
    if weight.size(0) != input.size(0):
        raise ValueError("weight and input batch sizes differ")


def _frame_dtype(t):
    if t.dtype == torch.float32:
        return _lib.FRAME_F32
    if t.dtype == torch.bfloat16:
        return _lib.FRAME_BF16
    raise TypeError(f"frames must be float32 or bfloat16, got {t.dtype}")


def _tile_radius():
    """options.tile_radius as passed to the library: the adaptive choice depends on the calls made
    before, so callers who asked for reproducible results get the fixed default instead -- and so do
    calls under HIP graph capture: a captured launch keeps the shape it was captured with, and the
    first adaptive call on a device allocates its 512 bytes of pinned memory."""
    r = int(options.tile_radius)
    if r == _lib.RADIUS_AUTO and (
            options.deterministic or torch.are_deterministic_algorithms_enabled()
            or (torch.cuda.is_initialized() and torch.cuda.is_current_stream_capturing())):
        r = _lib.RADIUS_DEFAULT
    return r


def _capture_options(context):
    """The options in force when the forward runs also govern its backward: autograd calls
    backward later, on its own thread, usually outside any ``with synth_options(...)`` block."""
    context.synth_tile_radius = _tile_radius()
    context.synth_deterministic = bool(options.deterministic)


def _launch_forward(input, weight, offset_y, offset_x, dilation, geom):
    """geom = (N, C, H, W, k, in_H, in_W, org_y, org_x)"""
    N, C, H, W, k, Hin, Win, oy0, ox0 = geom
    lib = _lib.load()
    with torch.cuda.device(input.device):
        # the kernel writes every element: no need for the reference's new_zeros (:285)
        output = weight.new_empty(N, C, H, W)
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.artemis_synth_forward_ex(
            _ptr(input), _ptr(weight), _ptr(offset_y), _ptr(offset_x), _ptr(output),
            N, C, H, W, k, int(dilation), Hin, Win, oy0, ox0, _frame_dtype(input),
            _tile_radius(), stream), "artemis_synth_forward")
    return output


def _launch_backward(need, gradOutput, input, weight, offset_y, offset_x, dilation, geom,
                     tile_radius=None, deterministic=None):
    """tile_radius / deterministic: the values captured at forward time (None: current options)."""
    N, C, H, W, k, Hin, Win, oy0, ox0 = geom
    tile_radius = int(_tile_radius() if tile_radius is None else tile_radius)
    deterministic = bool(options.deterministic if deterministic is None else deterministic)
    assert (gradOutput.is_contiguous() == True)
    if not input.is_cuda:
        raise NotImplementedError()
    if gradOutput.dtype != torch.float32 or tuple(gradOutput.shape) != (N, C, H, W):
        raise ValueError("gradOutput must be float32 of the forward output's shape")

    lib = _lib.load()
    with torch.cuda.device(input.device):
        # gradients are fp32 whatever the frame's dtype
        gradInput = weight.new_empty(N, C, Hin, Win) if need[0] else None
        gradWeight = weight.new_empty(N, k * k, H, W) if need[1] else None
        gradOffset_y = weight.new_empty(N, k * k, H, W) if need[2] else None
        gradOffset_x = weight.new_empty(N, k * k, H, W) if need[3] else None

        deterministic = deterministic or torch.are_deterministic_algorithms_enabled()
        mode = _lib.GIN_DETERMINISTIC if deterministic else _lib.GIN_ATOMIC
        workspace, ws_bytes = None, 0
        if need[0]:
            ws_bytes = lib.artemis_synth_backward_workspace_bytes_ex(
                N, C, H, W, k, int(dilation), Hin, Win, mode)
            if ws_bytes:
                workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=input.device)
        stream = torch.cuda.current_stream().cuda_stream
        if any(need[:4]):
            _lib.check(lib.artemis_synth_backward_ex(
                _ptr(input), _ptr(weight), _ptr(offset_y), _ptr(offset_x), _ptr(gradOutput),
                _ptr(gradInput), _ptr(gradWeight), _ptr(gradOffset_y), _ptr(gradOffset_x),
                N, C, H, W, k, int(dilation), Hin, Win, oy0, ox0, _frame_dtype(input), mode,
                tile_radius, _ptr(workspace), ws_bytes, stream),
                "artemis_synth_backward")
        if gradInput is not None and gradInput.dtype != input.dtype:
            gradInput = gradInput.to(input.dtype)   # autograd wants the frame's dtype
    return gradInput, gradWeight, gradOffset_y, gradOffset_x


class FunctionSynth(torch.autograd.Function):
    @staticmethod
    def forward(context, input, weight, offset_y, offset_x, dilation):
        context.save_for_backward(input, weight, offset_y, offset_x)
        context.dilation = dilation
        _capture_options(context)

        _check_operands(input, weight, offset_y, offset_x)
        N, C, H, W, k, Hin, Win = _dims(input, weight, dilation)

        assert (input.is_contiguous() == True)
        assert (weight.is_contiguous() == True)
        assert (offset_y.is_contiguous() == True)
        assert (offset_x.is_contiguous() == True)

        if not input.is_cuda:
            raise NotImplementedError()

        return _launch_forward(input, weight, offset_y, offset_x, dilation,
                               (N, C, H, W, k, Hin, Win, 0, 0))

    @staticmethod
    def backward(context, gradOutput):
        input, weight, offset_y, offset_x = context.saved_tensors
        dilation = context.dilation
        N, C, H, W, k, Hin, Win = _dims(input, weight, dilation)
        grads = _launch_backward(context.needs_input_grad, gradOutput, input, weight, offset_y,
                                 offset_x, dilation, (N, C, H, W, k, Hin, Win, 0, 0),
                                 context.synth_tile_radius, context.synth_deterministic)
        return (*grads, None)


class FunctionSynthFrame(torch.autograd.Function):
    """The op on a frame of explicit geometry (``artemis_synth_forward_ex``): tap (r, s) of output
    pixel (y, x) samples ``frame`` around ``(y + origin_y + r*d, x + origin_x + s*d)`` plus its
    offsets, indices clamped to the frame.  ``FunctionSynth`` is ``frame`` = the padded input and
    origin 0.  No counterpart class in the reference; it fuses what the reference composes around
    the op (``model/chrono_synth.py:70,125-126``: ReplicationPad2d, then the op)."""

    @staticmethod
    def forward(context, frame, weight, offset_y, offset_x, dilation, origin_y, origin_x):
        context.save_for_backward(frame, weight, offset_y, offset_x)
        _capture_options(context)
        _check_operands(frame, weight, offset_y, offset_x, bf16_frame_ok=True)
        k = int(math.sqrt(weight.size(1)))
        assert k * k == weight.size(1), "weight.size(1) must be a square"
        for t in (frame, weight, offset_y, offset_x):
            assert (t.is_contiguous() == True)
        if not frame.is_cuda:
            raise NotImplementedError()
        context.dilation = dilation
        context.geom = (frame.size(0), frame.size(1), weight.size(2), weight.size(3), k,
                        frame.size(2), frame.size(3), int(origin_y), int(origin_x))
        return _launch_forward(frame, weight, offset_y, offset_x, dilation, context.geom)

    @staticmethod
    def backward(context, gradOutput):
        frame, weight, offset_y, offset_x = context.saved_tensors
        grads = _launch_backward(context.needs_input_grad, gradOutput, frame, weight, offset_y,
                                 offset_x, context.dilation, context.geom,
                                 context.synth_tile_radius, context.synth_deterministic)
        return (*grads, None, None, None)


def synth_unpadded(frame, weight, offset_y, offset_x, dilation=1):
    """``FunctionSynth.apply(ReplicationPad2d((k-1)d/2)(frame), weight, offset_y, offset_x, d)``
    (model/chrono_synth.py:70,125-126) without materialising the padded frame: ``frame`` is
    (N, C, H, W).  Output and tap gradients are bit-identical to padding first; the gradient
    w.r.t. ``frame`` is what autograd through the padding would give."""
    k = int(math.sqrt(weight.size(1)))
    halo = (k - 1) * int(dilation)
    if halo % 2:
        raise ValueError("(k-1)*dilation must be even for symmetric padding")
    if tuple(frame.shape[2:]) != tuple(weight.shape[2:]):
        raise ValueError("frame and weight must have the same height and width")
    return FunctionSynthFrame.apply(frame, weight, offset_y, offset_x, dilation,
                                    -(halo // 2), -(halo // 2))


def synth_strip(frame, weight, offset_y, offset_x, row0, dilation=1, padded=True):
    """Rows ``[row0, row0 + weight.size(2))`` of the full-frame result: weight/offset tensors hold
    only those rows, ``frame`` is the whole frame (replication-padded if ``padded``)."""
    k = int(math.sqrt(weight.size(1)))
    halo = (k - 1) * int(dilation)
    org = 0 if padded else -(halo // 2)
    return FunctionSynthFrame.apply(frame, weight, offset_y, offset_x, dilation,
                                    int(row0) + org, org)


def synth(input, weight, offset_y, offset_x, dilation=1):
    """Functional form of ``FunctionSynth.apply``."""
    return FunctionSynth.apply(input, weight, offset_y, offset_x, dilation)


class FunctionChronoHead(torch.autograd.Function):
    """``framet`` of ``ChronoSynth.forward`` (model/chrono_synth.py:114-129) as one fused op::

        framet = sum_i occlusion[:, i:i+1] * FunctionSynth(ReplicationPad2d(frames[i]),
                     weights[:, i], alphas[:, i], betas[:, i], dilation)

    ``apply(weights, alphas, betas, occlusion, padded, *frames)``: weights/alphas/betas are the
    (B, T, 25, H, W) tensors the sub-networks produce (``.view(B, T, -1, H, W)``, :109-111), read
    in place; occlusion is (B, T, H, W) (:112); frames are T tensors (B, 3, H, W), or replication-
    padded (B, 3, H+4, W+4) with ``padded=True``.  Gradients flow to weights, alphas, betas and
    occlusion from the fused backward kernel; a frame that requires grad (none does in ArTEMIS)
    gets the op's gradInput for ``gradOutput * occlusion[:, i]`` through the per-frame entry point.
    Only k = 5, dilation = 1."""

    @staticmethod
    def forward(context, weights, alphas, betas, occlusion, padded, *frames):
        T = len(frames)
        if weights.dim() != 5 or weights.shape != alphas.shape or weights.shape != betas.shape:
            raise ValueError("weights, alphas, betas must be (B, T, 25, H, W) of one shape")
        B, Tw, K, H, W = weights.shape
        if Tw != T:
            raise ValueError(f"{T} frames but weights.size(1) = {Tw}")
        if K != 25:
            raise NotImplementedError("the fused head is built for kernel_size 5, dilation 1 "
                                      "(config.py:26-27); compose FunctionSynth for other sizes")
        if tuple(occlusion.shape) != (B, T, H, W):
            raise ValueError("occlusion must be (B, T, H, W)")
        pad = 2 if padded else 0
        for t in (weights, alphas, betas, occlusion):
            if t.dtype != torch.float32:
                raise TypeError("weights, alphas, betas and occlusion must be float32")
        fdt = _frame_dtype(frames[0])
        for t in (weights, alphas, betas, occlusion, *frames):
            if t.device != weights.device:
                raise ValueError("all operands must be on one device")
            assert (t.is_contiguous() == True)
        for f in frames:
            if tuple(f.shape) != (B, 3, H + 2 * pad, W + 2 * pad):
                raise ValueError(f"frame shape {tuple(f.shape)} does not match the head")
            if _frame_dtype(f) != fdt:
                raise TypeError("all frames of a head must have one dtype")
        if not weights.is_cuda:
            raise NotImplementedError()

        context.save_for_backward(weights, alphas, betas, occlusion, *frames)
        _capture_options(context)
        context.geom = (B, T, H, W, H + 2 * pad, W + 2 * pad, pad - 2, fdt)
        lib = _lib.load()
        with torch.cuda.device(weights.device):
            out = weights.new_empty(B, 3, H, W)
            ptrs = (ctypes.c_void_p * T)(*[f.data_ptr() for f in frames])
            _lib.check(lib.artemis_synth_head_forward(
                ptrs, T, _ptr(weights), _ptr(alphas), _ptr(betas), _ptr(occlusion), _ptr(out),
                B, H, W, H + 2 * pad, W + 2 * pad, pad - 2, pad - 2, fdt, _tile_radius(),
                torch.cuda.current_stream().cuda_stream), "artemis_synth_head_forward")
        return out

    @staticmethod
    def backward(context, gradOutput):
        weights, alphas, betas, occlusion, *frames = context.saved_tensors
        B, T, H, W, Hin, Win, org, fdt = context.geom
        assert (gradOutput.is_contiguous() == True)
        if gradOutput.dtype != torch.float32 or tuple(gradOutput.shape) != (B, 3, H, W):
            raise ValueError("gradOutput must be float32 of the forward output's shape")
        need = context.needs_input_grad
        lib = _lib.load()
        with torch.cuda.device(weights.device):
            gw = torch.empty_like(weights) if need[0] else None
            ga = torch.empty_like(alphas) if need[1] else None
            gb = torch.empty_like(betas) if need[2] else None
            gocc = torch.empty_like(occlusion) if need[3] else None
            if any(need[:4]):
                ptrs = (ctypes.c_void_p * T)(*[f.data_ptr() for f in frames])
                _lib.check(lib.artemis_synth_head_backward(
                    ptrs, T, _ptr(weights), _ptr(alphas), _ptr(betas), _ptr(occlusion),
                    _ptr(gradOutput), _ptr(gw), _ptr(ga), _ptr(gb), _ptr(gocc), B, H, W, Hin, Win,
                    org, org, fdt, context.synth_tile_radius,
                    torch.cuda.current_stream().cuda_stream), "artemis_synth_head_backward")
            # Frame gradients (ArTEMIS never asks for them: its frames are inputs) are not part of
            # the fused kernel: per frame, gradInput of the op for gradOutput * occ_f.
            gframes = [None] * T
            for f in range(T):
                if not need[5 + f]:
                    continue
                gs = (occlusion[:, f:f + 1] * gradOutput).contiguous()
                gframes[f] = _launch_backward(
                    (True, False, False, False), gs, frames[f], weights[:, f].contiguous(),
                    alphas[:, f].contiguous(), betas[:, f].contiguous(), 1,
                    (B, 3, H, W, 5, Hin, Win, org, org),
                    context.synth_tile_radius, context.synth_deterministic)[0]
        return (gw, ga, gb, gocc, None, *gframes)


def chrono_head(frames, weights, alphas, betas, occlusion, dilation=1, padded=False):
    """Drop-in for the loop at model/chrono_synth.py:114-129."""
    if int(dilation) != 1:
        raise NotImplementedError("the fused head is built for dilation 1 (config.py:27)")
    return FunctionChronoHead.apply(weights, alphas, betas, occlusion, bool(padded), *frames)


def offset_radius(offset_y, offset_x) -> torch.Tensor:
    """max |trunc(offset)| over both tensors as a 0-d int32 device tensor (no host sync): the
    ``tile_radius`` that would serve every tap from LDS."""
    if not offset_y.is_cuda:
        raise NotImplementedError()
    assert offset_y.is_contiguous() and offset_x.is_contiguous()
    assert offset_y.shape == offset_x.shape and offset_y.dtype == torch.float32
    lib = _lib.load()
    with torch.cuda.device(offset_y.device):
        out = torch.empty((), dtype=torch.int32, device=offset_y.device)
        _lib.check(lib.artemis_synth_offset_radius(
            _ptr(offset_y), _ptr(offset_x), offset_y.numel(), _ptr(out),
            torch.cuda.current_stream().cuda_stream), "artemis_synth_offset_radius")
    return out


def suggest_tile_radius(offset_y, offset_x) -> int:
    """A fixed ``tile_radius`` for offsets like these -- ONE host synchronisation (``.item()``) and a
    pass over both tensors; for callers who want a fixed radius (bit-reproducible forward).  The
    default ``options.tile_radius`` (adaptive) needs neither.  The lean radius-8 shape
    (``RADIUS_SMALL``) when every whole-pixel offset stays within 8, 14 within 14, else 24
    (DESIGN.md 4.4).  (A large but uniform displacement -- a pan -- runs as fast at 14: the scalar
    cannot tell it from a steep flow; the adaptive mode can.)"""
    reach = int(offset_radius(offset_y, offset_x).item())
    return _lib.RADIUS_SMALL if reach <= 8 else (14 if reach <= 14 else 24)


def adaptive_state(height, width, head=False, device=None):
    """What the adaptive tile radius currently holds for output size (height, width) of the
    per-frame op (or the fused head) on ``device``: dict(radius, frac8, frac14, forward_calls).
    Host-only (artemis_synth_auto_query)."""
    lib = _lib.load()
    r, c = ctypes.c_int(0), ctypes.c_ulonglong(0)
    f8, f14 = ctypes.c_double(0), ctypes.c_double(0)
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        lib.artemis_synth_auto_query(int(height), int(width), int(bool(head)), ctypes.byref(r),
                                     ctypes.byref(f8), ctypes.byref(f14), ctypes.byref(c))
    return {"radius": r.value, "frac8": f8.value, "frac14": f14.value, "forward_calls": c.value}

