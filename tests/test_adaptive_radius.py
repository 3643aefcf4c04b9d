"""The package's default tile radius is adaptive (ARTEMIS_SYNTH_RADIUS_AUTO, include/artemis_synth.h):
the library moves between 8, 14 and 24 per tensor shape from reach statistics that its forward
kernels publish to pinned host memory -- no host synchronisation, no reference counterpart (the
reference has no tile at all).  Here: the choice follows the offsets, results stay within the
oracle's tolerances on the way, and the policy / option plumbing on the CPU."""
import numpy as np
import pytest
import torch

from conftest import parity_close

FWD_RTOL, BWD_RTOL = 1e-5, 1e-4


def test_default_option_is_adaptive_and_deterministic_callers_get_the_fixed_default():
    import artemis_amd
    from artemis_amd import _lib, synth as _  # noqa: F401
    import sys
    mod = sys.modules["artemis_amd.synth"]
    assert artemis_amd.SynthOptions().tile_radius == _lib.RADIUS_AUTO == -2 if hasattr(artemis_amd, "SynthOptions") \
        else mod.SynthOptions().tile_radius == _lib.RADIUS_AUTO == -2
    with artemis_amd.synth_options(tile_radius=_lib.RADIUS_AUTO):
        assert mod._tile_radius() == _lib.RADIUS_AUTO
        with artemis_amd.synth_options(deterministic=True):
            assert mod._tile_radius() == _lib.RADIUS_DEFAULT
    with artemis_amd.synth_options(tile_radius=8, deterministic=True):
        assert mod._tile_radius() == 8


def np_(t):
    return np.ascontiguousarray(t.detach().float().cpu().numpy())


def _operands(dev, N, H, W, regime, seed):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    g = torch.Generator(device=dev).manual_seed(seed)
    frame = torch.rand(N, 3, H, W, device=dev, generator=g)
    inp = torch.nn.functional.pad(frame, (2,) * 4, mode="replicate").contiguous()
    w = torch.softmax(torch.randn(N, 25, H, W, device=dev, generator=g), 1)
    oy, ox = bench.make_offsets(regime, (N, 25, H, W), dev, g)
    gout = torch.randn(N, 3, H, W, device=dev, generator=g)
    return inp, w, oy, ox, gout


@pytest.mark.gpu
@pytest.mark.adaptive
def test_adaptive_radius_follows_the_offsets_and_keeps_parity(oracle):
    """One shape, four offset regimes in a row: sigma 1 -> the lean radius-8 kernels, sigma 4 -> 24,
    a pan (large but coherent: the lean shape has no window centres, radius 14's windows follow it)
    -> 14, sigma 1 again -> back to 8; forward and backward against the oracle at every stage,
    whatever radius the stage happens to run."""
    import artemis_amd
    from artemis_amd import _lib
    dev = torch.device("cuda:0")
    lib = _lib.load()
    lib.artemis_synth_auto_reset()
    assert artemis_amd.options.tile_radius == _lib.RADIUS_AUTO      # the package default
    H, W = 272, 448
    oracle.set_threads(8)
    seen = []
    for regime, want in (("sigma:1", 8), ("sigma:4", 24), ("pan", 14), ("sigma:1", 8)):
        inp, w, oy, ox, gout = _operands(dev, 2, H, W, regime, seed=len(seen))
        for it in range(16):                                       # the choice lags the launches in flight
            t = [inp] + [x.clone().requires_grad_(True) for x in (w, oy, ox)]
            out = artemis_amd.FunctionSynth.apply(*t, 1)
            out.backward(gout)
            torch.cuda.synchronize()
            if it in (0, 15):
                tag = f"adaptive {regime} call {it}"
                parity_close(out, oracle.forward(np_(inp), np_(w), np_(oy), np_(ox), 1), FWD_RTOL, tag + " output")
                _, rgw, rga, rgb = oracle.backward(np_(gout), np_(inp), np_(w), np_(oy), np_(ox), 1,
                                                   with_input=False)
                term = float(gout.abs().max() * w.abs().max())
                parity_close(t[1].grad, rgw, BWD_RTOL, tag + " grad_weight")
                parity_close(t[2].grad, rga, BWD_RTOL, tag + " grad_offset_y", term)
                parity_close(t[3].grad, rgb, BWD_RTOL, tag + " grad_offset_x", term)
        st = artemis_amd.adaptive_state(H, W)
        seen.append((regime, st))
        assert st["radius"] == want, seen
    oracle.set_threads(1)
    assert seen[0][1]["frac8"] < 5e-5 and seen[1][1]["frac14"] > 1.5e-4, seen


@pytest.mark.gpu
@pytest.mark.adaptive
def test_adaptive_head_and_op_are_separate_shapes_and_fixed_radius_calls_leave_no_trace():
    import artemis_amd
    from artemis_amd import _lib
    dev = torch.device("cuda:0")
    lib = _lib.load()
    lib.artemis_synth_auto_reset()
    B, T, H, W = 2, 4, 128, 192
    g = torch.Generator(device=dev).manual_seed(5)
    frames = [torch.rand(B, 3, H, W, device=dev, generator=g) for _ in range(T)]
    w = torch.softmax(torch.randn(B, T, 25, H, W, device=dev, generator=g), 2)
    a = torch.randn(B, T, 25, H, W, device=dev, generator=g) * 5
    b = torch.randn(B, T, 25, H, W, device=dev, generator=g) * 5
    occ = torch.softmax(torch.randn(B, T, H, W, device=dev, generator=g), 1)
    with artemis_amd.synth_options(tile_radius=14):
        fixed = artemis_amd.chrono_head(frames, w, a, b, occ)
    torch.cuda.synchronize()
    assert artemis_amd.adaptive_state(H, W, head=True)["forward_calls"] == 0
    for _ in range(10):
        out = artemis_amd.chrono_head(frames, w, a, b, occ)
        torch.cuda.synchronize()
    st = artemis_amd.adaptive_state(H, W, head=True)
    assert st["forward_calls"] == 10 and st["radius"] == 24, st
    assert artemis_amd.adaptive_state(H, W, head=False)["forward_calls"] == 0
    # radius 24 against radius 14: the same sums in another order
    assert float((out - fixed).abs().max()) <= 1e-5 * float(fixed.abs().max())
