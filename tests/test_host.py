"""Host-side operator contract of FunctionSynth (no GPU): the reference's asserts and error
behaviour (cupy_module/synth.py:277-283, :310-311) and the option plumbing."""
import pytest
import torch

import artemis_amd
from artemis_amd import FunctionSynth, synth, synth_options, options


def ops(N=1, C=3, H=4, W=5, k=5, d=1, device="cpu"):
    return (torch.rand(N, C, H + (k - 1) * d, W + (k - 1) * d, device=device),
            torch.rand(N, k * k, H, W, device=device), torch.rand(N, k * k, H, W, device=device),
            torch.rand(N, k * k, H, W, device=device), d)


def test_signature_matches_reference():
    import inspect
    assert list(inspect.signature(FunctionSynth.forward).parameters) == [
        "context", "input", "weight", "offset_y", "offset_x", "dilation"]
    assert list(inspect.signature(FunctionSynth.backward).parameters) == ["context", "gradOutput"]
    assert issubclass(FunctionSynth, torch.autograd.Function)


def test_cpu_tensors_raise_like_the_reference():
    with pytest.raises(NotImplementedError):
        FunctionSynth.apply(*ops())


def test_shape_relation_is_asserted():
    i, w, a, b, d = ops()
    with pytest.raises(AssertionError):
        FunctionSynth.apply(i[:, :, :-1].contiguous(), w, a, b, d)
    with pytest.raises(AssertionError):
        FunctionSynth.apply(i, w, a, b, 2)  # dilation inconsistent with the padding


def test_contiguity_is_asserted():
    i, w, a, b, d = ops()
    wt = w.transpose(2, 3).contiguous().transpose(2, 3)
    assert not wt.is_contiguous()
    with pytest.raises(AssertionError):
        FunctionSynth.apply(i, wt, a, b, d)


def test_dtype_and_shape_errors():
    i, w, a, b, d = ops()
    with pytest.raises(TypeError):
        FunctionSynth.apply(i.double(), w, a, b, d)
    with pytest.raises(ValueError):
        FunctionSynth.apply(i, w, a[:, :-1], b, d)
    with pytest.raises(ValueError):
        FunctionSynth.apply(i[0], w, a, b, d)
    with pytest.raises(AssertionError):  # 24 taps is not a square
        FunctionSynth.apply(i, w[:, :24].contiguous(), a[:, :24].contiguous(),
                            b[:, :24].contiguous(), d)


def test_options_context_restores():
    assert options.tile_radius == -1 and options.deterministic is False
    with synth_options(tile_radius=0, deterministic=True):
        assert options.tile_radius == 0 and options.deterministic is True
    assert options.tile_radius == -1 and options.deterministic is False


def test_compat_module_path():
    import os, sys
    compat = os.path.join(os.path.dirname(artemis_amd.__file__), "compat")
    sys.path.insert(0, compat)
    try:
        import cupy_module.synth as s  # what model/chrono_synth.py:66 does
        assert s.FunctionSynth is FunctionSynth
    finally:
        sys.path.remove(compat)
        sys.modules.pop("cupy_module.synth", None)
        sys.modules.pop("cupy_module", None)


def test_functional_alias():
    with pytest.raises(NotImplementedError):
        synth(*ops())
