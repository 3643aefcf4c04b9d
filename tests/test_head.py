"""The fused ChronoSynth head (artemis_synth_head_forward/backward, FunctionChronoHead): the loop
at model/chrono_synth.py:114-129 as one op.  CPU: the oracle's restatement of that loop against
autograd through the differentiable torch restatement.  GPU: the HIP path against the oracle and,
bit for bit, against the same composition made of this library's per-frame op and torch ops."""
import numpy as np
import pytest
import torch

from conftest import make_operands, parity_close

FWD_RTOL = 1e-5
BWD_RTOL = 1e-4


def head_operands(B, T, H, W, regime="gauss1", seed=0, weights="softmax"):
    rng = np.random.default_rng(seed + 1000)
    frames, ws, als, bes = [], [], [], []
    for i in range(T):
        inp, w, oy, ox, _ = make_operands(B, 3, H, W, 5, 1, regime, seed * 17 + i, weights)
        frames.append(np.ascontiguousarray(inp[:, :, 2:2 + H, 2:2 + W]))
        ws.append(w), als.append(oy), bes.append(ox)
    weights_, alphas, betas = (np.ascontiguousarray(np.stack(t, 1)) for t in (ws, als, bes))
    logits = rng.standard_normal((B, T, H, W), dtype=np.float32)
    e = np.exp(logits - logits.max(1, keepdims=True))
    occ = (e / e.sum(1, keepdims=True)).astype(np.float32)   # Softmax(dim=1), chrono_synth.py:55
    gout = rng.standard_normal((B, 3, H, W), dtype=np.float32)
    return frames, weights_, alphas, betas, occ, gout


def close(actual, desired, rtol, what="", floor=1e-30):
    """conftest.parity_close: norm-relative and elementwise-relative asserts, worst errors recorded."""
    parity_close(actual, desired, rtol, what, floor)


# ---------------------------------------------------------------- oracle (CPU)

def torch_head(oracle, frames, weights, alphas, betas, occ):
    """model/chrono_synth.py:114-129 with the differentiable torch restatement of the op."""
    warp = []
    for i in range(weights.shape[1]):
        padded = torch.nn.functional.pad(frames[i], (2,) * 4, mode="replicate")
        warp.append(occ[:, i:i + 1] * oracle.forward_torch(
            padded, weights[:, i].contiguous(), alphas[:, i].contiguous(), betas[:, i].contiguous(), 1))
    return sum(warp)


@pytest.mark.parametrize("cfg", [(2, 4, 7, 9, "uniform3"), (1, 1, 5, 5, "gauss4"), (1, 3, 6, 11, "clamp40")])
def test_oracle_head_matches_autograd_through_the_torch_restatement(oracle, cfg):
    B, T, H, W, regime = cfg
    frames, w, a, b, occ, gout = head_operands(B, T, H, W, regime, seed=H)
    out, synth = oracle.head_forward(frames, w, a, b, occ, 1)
    tw, ta, tb, tocc = (torch.from_numpy(x).double().requires_grad_(True) for x in (w, a, b, occ))
    tf = [torch.from_numpy(f).double() for f in frames]
    tout = torch_head(oracle, tf, tw, ta, tb, tocc)
    np.testing.assert_allclose(out, tout.detach().numpy(), rtol=2e-6, atol=2e-6)
    tout.backward(torch.from_numpy(gout).double())
    gw, ga, gb, gocc = oracle.head_backward(gout, frames, w, a, b, occ, 1)
    for got, want, name in ((gw, tw.grad, "gw"), (ga, ta.grad, "ga"), (gb, tb.grad, "gb"),
                            (gocc, tocc.grad, "gocc")):
        want = want.numpy()
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(want).max()),
                                   err_msg=name)


def test_head_abi_validation():
    import ctypes
    from artemis_amd import _lib
    lib = _lib.load()
    one = (ctypes.c_void_p * 1)(0)
    args = (None, None, None, None, None, 1, 4, 4, 4, 4, -2, -2, 0, -1, None)
    assert lib.artemis_synth_head_forward(one, 0, *args) == 1
    assert "context frames" in lib.artemis_synth_last_error().decode()
    assert lib.artemis_synth_head_forward(one, 9, *args) == 1
    assert lib.artemis_synth_head_forward(None, 1, *args) == 1
    assert lib.artemis_synth_head_forward(one, 1, *args) == 1      # frames[0] is NULL
    assert "NULL" in lib.artemis_synth_last_error().decode()
    # empty batch: successful no-op before any pointer is looked at
    assert lib.artemis_synth_head_forward(one, 1, None, None, None, None, None, 0, 4, 4, 4, 4, -2,
                                          -2, 0, -1, None) == 0


def test_head_host_contract_on_cpu():
    import artemis_amd
    frames, w, a, b, occ, _ = head_operands(1, 2, 4, 4)
    t = [torch.from_numpy(x) for x in (w, a, b, occ)]
    f = [torch.from_numpy(x) for x in frames]
    with pytest.raises(NotImplementedError):        # CPU tensors, as the reference (:310-311)
        artemis_amd.chrono_head(f, *t)
    with pytest.raises(ValueError):
        artemis_amd.chrono_head(f[:1], *t)
    with pytest.raises(NotImplementedError):
        artemis_amd.chrono_head(f, *t, dilation=2)
    with pytest.raises(ValueError):
        artemis_amd.chrono_head(f, t[0], t[1], t[2], t[3][:, :1])


# ---------------------------------------------------------------- HIP path (GPU)

@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import artemis_amd
    artemis_amd._lib.load()
    return torch.device("cuda:0")


def composed_head(frames, weights, alphas, betas, occ):
    """The reference's loop, with this library's per-frame op."""
    import artemis_amd
    warp = []
    for i in range(weights.shape[1]):
        padded = torch.nn.functional.pad(frames[i], (2,) * 4, mode="replicate")
        warp.append(occ[:, i:i + 1] * artemis_amd.FunctionSynth.apply(
            padded, weights[:, i].contiguous(), alphas[:, i].contiguous(), betas[:, i].contiguous(), 1))
    return sum(warp)


def run_head(dev, fn, frames, w, a, b, occ, gout, radius, need=(True,) * 4):
    import artemis_amd
    t = [torch.from_numpy(x).to(dev).requires_grad_(n) for x, n in zip((w, a, b, occ), need)]
    f = [torch.from_numpy(x).to(dev) for x in frames]
    with artemis_amd.synth_options(tile_radius=radius):
        out = fn(f, *t)
        if any(need):
            out.backward(torch.from_numpy(gout).to(dev))
    torch.cuda.synchronize()
    return out.detach(), [x.grad for x in t]


@pytest.mark.gpu
@pytest.mark.parametrize("radius", [-1, 2, 8])
@pytest.mark.parametrize("cfg", [(2, 4, 33, 70, "gauss1"), (1, 4, 40, 129, "gauss4"), (1, 1, 1, 1, "gauss1"),
                                 (3, 2, 17, 64, "clamp40"), (1, 8, 9, 200, "uniform3"),
                                 (1, 4, 64, 64, "init")])
def test_head_against_oracle_and_composition(dev, oracle, cfg, radius):
    import artemis_amd
    B, T, H, W, regime = cfg
    frames, w, a, b, occ, gout = head_operands(B, T, H, W, regime, seed=H + W)
    out, (gw, ga, gb, gocc) = run_head(dev, artemis_amd.chrono_head, frames, w, a, b, occ, gout, radius)
    cout, (cgw, cga, cgb, cgocc) = run_head(dev, composed_head, frames, w, a, b, occ, gout, radius)
    # forward: the bits of the composition (same per-frame arithmetic, product rounded, then added)
    assert torch.equal(out, cout)
    ref, _ = oracle.head_forward(frames, w, a, b, occ, 1)
    close(out, ref, FWD_RTOL, "framet")
    rgw, rga, rgb, rgocc = oracle.head_backward(gout, frames, w, a, b, occ, 1)
    term = float(np.abs(gout).max() * max(np.abs(f).max() for f in frames) * np.abs(w).max())
    close(gw, rgw, BWD_RTOL, "grad_weights")
    close(ga, rga, BWD_RTOL, "grad_alphas", term)
    close(gb, rgb, BWD_RTOL, "grad_betas", term)
    close(gocc, rgocc, BWD_RTOL, "grad_occlusion")
    # and against autograd through the composition on the device
    close(gw, cgw.cpu().numpy(), BWD_RTOL, "gw vs composed")
    close(ga, cga.cpu().numpy(), BWD_RTOL, "ga vs composed", term)
    close(gb, cgb.cpu().numpy(), BWD_RTOL, "gb vs composed", term)
    close(gocc, cgocc.cpu().numpy(), BWD_RTOL, "gocc vs composed")


@pytest.mark.gpu
def test_head_padded_frames_and_raw_weights(dev, oracle):
    import artemis_amd
    frames, w, a, b, occ, gout = head_operands(2, 4, 30, 77, "gauss4", seed=5, weights="raw")
    padded = [np.pad(f, ((0, 0), (0, 0), (2, 2), (2, 2)), mode="edge") for f in frames]
    out, g1 = run_head(dev, artemis_amd.chrono_head, frames, w, a, b, occ, gout, -1)
    pout, g2 = run_head(dev, lambda f, *t: artemis_amd.chrono_head(f, *t, padded=True), padded, w, a, b,
                        occ, gout, -1)
    assert torch.equal(out, pout)
    for x, y in zip(g1, g2):
        assert torch.equal(x, y)
    ref, _ = oracle.head_forward(frames, w, a, b, occ, 1)
    close(out, ref, FWD_RTOL)


@pytest.mark.gpu
@pytest.mark.parametrize("need", [(False, False, False, True), (True, False, False, False),
                                  (False, True, True, False)])
def test_head_gradient_subsets(dev, oracle, need):
    import artemis_amd
    frames, w, a, b, occ, gout = head_operands(1, 4, 20, 66, "gauss4", seed=9)
    _, grads = run_head(dev, artemis_amd.chrono_head, frames, w, a, b, occ, gout, 2, need)
    ref = oracle.head_backward(gout, frames, w, a, b, occ, 1)
    term = float(np.abs(gout).max() * np.abs(w).max())
    for g, r, n in zip(grads, ref, need):
        assert (g is not None) == n
        if n:
            close(g, r, BWD_RTOL, floor=term)


@pytest.mark.gpu
def test_head_far_offsets_take_both_fixup_paths(dev, oracle):
    # sigma 4 with radius 2: most pixels have far taps (> 12 flagged lanes per wave row: the
    # per-lane fix-up); sigma 1 with radius 2: a few (the cooperative fix-up)
    import artemis_amd
    for regime in ("gauss4", "gauss1"):
        frames, w, a, b, occ, gout = head_operands(1, 3, 33, 130, regime, seed=21)
        out, (gw, ga, gb, gocc) = run_head(dev, artemis_amd.chrono_head, frames, w, a, b, occ, gout, 2)
        ref, _ = oracle.head_forward(frames, w, a, b, occ, 1)
        close(out, ref, FWD_RTOL)
        rgw, rga, rgb, rgocc = oracle.head_backward(gout, frames, w, a, b, occ, 1)
        close(gw, rgw, BWD_RTOL)
        close(ga, rga, BWD_RTOL)
        close(gb, rgb, BWD_RTOL)
        close(gocc, rgocc, BWD_RTOL)


@pytest.mark.gpu
def test_head_full_size_equals_composition(dev):
    # BASELINE.json headline shape: 4 frames at 1080p
    import artemis_amd
    g = torch.Generator(device=dev).manual_seed(103)
    B, T, H, W = 1, 4, 1080, 1920
    frames = [torch.rand(B, 3, H, W, device=dev, generator=g) for _ in range(T)]
    w = torch.softmax(torch.randn(B, T, 25, H, W, device=dev, generator=g), 2).requires_grad_(True)
    a = torch.randn(B, T, 25, H, W, device=dev, generator=g).requires_grad_(True)
    b = torch.randn(B, T, 25, H, W, device=dev, generator=g).requires_grad_(True)
    occ = torch.softmax(torch.randn(B, T, H, W, device=dev, generator=g), 1).requires_grad_(True)
    gout = torch.randn(B, 3, H, W, device=dev, generator=g)
    out = artemis_amd.chrono_head(frames, w, a, b, occ)
    out.backward(gout)
    fused = [x.grad.clone() for x in (w, a, b, occ)]
    for x in (w, a, b, occ):
        x.grad = None
    cout = composed_head(frames, w, a, b, occ)
    cout.backward(gout)
    assert torch.equal(out, cout)
    for f, x, name in zip(fused, (w, a, b, occ), ("gw", "ga", "gb", "gocc")):
        scale = float(x.grad.abs().max())
        err = float((f - x.grad).abs().max())
        assert err <= BWD_RTOL * scale, (name, err, scale)


@pytest.mark.gpu
def test_head_frame_gradients_through_the_per_frame_entry_point(dev, oracle):
    # north_star: "gradients for the input, the kernels, the offsets and the mask".  Frames that
    # require grad get the op's gradInput for gradOutput * occ_f (oracle: its fp64 adjoint).
    import artemis_amd
    frames, w, a, b, occ, gout = head_operands(2, 3, 21, 70, "gauss1", seed=31)
    tf = [torch.from_numpy(f).to(dev).requires_grad_(i != 1) for i, f in enumerate(frames)]
    t = [torch.from_numpy(x).to(dev).requires_grad_(True) for x in (w, a, b, occ)]
    out = artemis_amd.chrono_head(tf, *t)
    out.backward(torch.from_numpy(gout).to(dev))
    torch.cuda.synchronize()
    assert tf[1].grad is None
    for i in (0, 2):
        gs = (occ[:, i:i + 1] * gout).astype(np.float32)
        ref = oracle.backward_ex(gs, frames[i], np.ascontiguousarray(w[:, i]),
                                 np.ascontiguousarray(a[:, i]), np.ascontiguousarray(b[:, i]), 1,
                                 origin=(-2, -2))[0]
        close(tf[i].grad, ref, BWD_RTOL, f"grad frame {i}")
    rgw, rga, rgb, rgocc = oracle.head_backward(gout, frames, w, a, b, occ, 1)
    close(t[0].grad, rgw, BWD_RTOL)
    close(t[3].grad, rgocc, BWD_RTOL)
