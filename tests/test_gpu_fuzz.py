"""Randomised GPU parity: seeded random shapes, regimes, radii, gradient subsets and frame
geometries against the CPU oracle; empty problems through the operator."""
import numpy as np
import pytest
import torch

from conftest import make_operands, parity_close

pytestmark = pytest.mark.gpu

FWD_RTOL = 1e-5
BWD_RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import artemis_amd
    artemis_amd._lib.load()
    return torch.device("cuda:0")


def close(actual, desired, rtol, what="", floor=1e-30):
    """conftest.parity_close: norm-relative and elementwise-relative asserts, worst errors recorded."""
    parity_close(actual, desired, rtol, what, floor)


REGIMES = ["init", "gauss1", "gauss4", "uniform3", "clamp40", "integer", "zero"]


@pytest.mark.parametrize("case", range(40))
def test_random_case_against_oracle(dev, oracle, case):
    import artemis_amd
    rng = np.random.default_rng(9000 + case)
    model_shape = rng.random() < 0.7                      # k=5, d=1, C=3: the LDS-staged kernels
    k, d, C = (5, 1, 3) if model_shape else (int(rng.choice([1, 3, 5, 7])), int(rng.choice([1, 2, 3])),
                                             int(rng.integers(1, 6)))
    N = int(rng.integers(1, 4))
    H = int(rng.choice([1, 2, 7, 31, 32, 33, 63, 64, 65, 90]))
    W = int(rng.choice([1, 3, 63, 64, 65, 127, 128, 129, 200]))
    regime = REGIMES[int(rng.integers(len(REGIMES)))]
    radius = int(rng.choice([-1, 0, 2, 4, 8, 16]))
    det = bool(rng.integers(2))
    weights = "softmax" if rng.random() < 0.5 else "raw"
    inp, w, oy, ox, gout = make_operands(N, C, H, W, k, d, regime, seed=case, weights=weights)
    # a random frame geometry half of the time (else the reference layout)
    if rng.random() < 0.5:
        Hin, Win = int(rng.integers(1, H + 8)), int(rng.integers(1, W + 8))
        org = (int(rng.integers(-6, 7)), int(rng.integers(-6, 7)))
        frame = rng.random((N, C, Hin, Win), dtype=np.float32)
    else:
        frame, org = inp, (0, 0)
    need = tuple(bool(x) for x in rng.integers(0, 2, 4))
    t = [torch.from_numpy(a).to(dev).requires_grad_(n) for a, n in zip((frame, w, oy, ox), need)]
    with artemis_amd.synth_options(tile_radius=radius, deterministic=det):
        out = artemis_amd.FunctionSynthFrame.apply(*t, d, org[0], org[1])
        if any(need):
            out.backward(torch.from_numpy(gout).to(dev))
    torch.cuda.synchronize()
    tag = f"N{N} C{C} {H}x{W} k{k} d{d} {regime} R{radius} det{det} frame{frame.shape[2:]} org{org}"
    close(out, oracle.forward_ex(frame, w, oy, ox, d, origin=org), FWD_RTOL, "output " + tag)
    ref = oracle.backward_ex(gout, frame, w, oy, ox, d, origin=org, with_input=need[0])
    term = float(np.abs(gout).max() * np.abs(w).max() * np.abs(frame).max())
    for g, r, n, name in zip(t, ref, need, ("grad_input", "grad_weight", "grad_offset_y", "grad_offset_x")):
        assert (g.grad is not None) == n
        if n:
            close(g.grad, r, BWD_RTOL, name + " " + tag, floor=term if "offset" in name else 1e-30)


@pytest.mark.parametrize("shape", [(0, 3, 5, 7), (2, 3, 0, 7), (2, 3, 5, 0)])
def test_empty_problems(dev, shape):
    import artemis_amd
    N, C, H, W = shape
    inp = torch.rand(N, C, H + 4, W + 4, device=dev, requires_grad=True)
    w, oy, ox = (torch.rand(N, 25, H, W, device=dev, requires_grad=True) for _ in range(3))
    out = artemis_amd.FunctionSynth.apply(inp, w, oy, ox, 1)
    assert out.shape == (N, C, H, W)
    out.sum().backward()
    torch.cuda.synchronize()
    assert inp.grad.shape == inp.shape and float(inp.grad.abs().sum()) == 0.0   # zero-filled
    assert w.grad.shape == w.shape


def test_forward_is_reproducible_run_to_run(dev):
    # bottom/right edge tiles compute some pixels in several row slots; all store the same bits
    import artemis_amd
    inp, w, oy, ox, _ = make_operands(1, 3, 37, 70, regime="gauss4", seed=3)
    t = [torch.from_numpy(a).to(dev) for a in (inp, w, oy, ox)]
    first = artemis_amd.FunctionSynth.apply(*t, 1)
    for _ in range(5):
        assert torch.equal(artemis_amd.FunctionSynth.apply(*t, 1), first)
