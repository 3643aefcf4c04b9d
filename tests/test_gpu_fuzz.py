"""Randomised GPU parity: seeded random shapes, regimes, radii, gradient subsets and frame
geometries against the CPU oracle; empty problems through the operator."""
import os

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


@pytest.mark.parametrize("case", range(int(os.environ.get("FUZZ_CASES", "40"))))
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
    radius = int(rng.choice([-1, 0, 2, 4, 8, 14, 24, -2, -3]))      # (-2: adaptive, whatever it holds)
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
        if not n:
            continue
        if name == "grad_input":
            # A texel's gradient is a sum of up to thousands of signed contributions (every tap that
            # clamps onto a border texel; the truncating bilinear form also extrapolates with negative
            # coefficients).  Mode ATOMIC adds them in fp32 in an order that differs run to run, so the
            # error of a cancelling sum relative to the element itself is not reproducible (seen:
            # 1.15e-4 once, on an element of 0.5 % of the tensor's scale, in a case that otherwise
            # passes); there the norm-relative assert is the bound.  Mode DETERMINISTIC gives the
            # same bits every run and keeps the elementwise assert.
            parity_close(g.grad, r, BWD_RTOL, name + " " + tag, elementwise=det)
        else:
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


def random_motion_field(rng, N, H, W):
    """Offsets made of 1-4 rectangular regions with their own displacement (up to +-45 px), an
    optional linear gradient, per-tap jitter, and -- sometimes -- a per-tap split (taps of one pixel
    pointing at two places)."""
    shape = (N, 25, H, W)
    oy = rng.standard_normal(shape).astype(np.float32) * np.float32(rng.choice([0.1, 0.5, 1.5]))
    ox = rng.standard_normal(shape).astype(np.float32) * np.float32(rng.choice([0.1, 0.5, 1.5]))
    for _ in range(int(rng.integers(1, 5))):
        y0, y1 = sorted(int(v) for v in rng.integers(0, H + 1, 2))
        x0, x1 = sorted(int(v) for v in rng.integers(0, W + 1, 2))
        oy[:, :, y0:y1, x0:x1] += np.float32(rng.integers(-45, 46))
        ox[:, :, y0:y1, x0:x1] += np.float32(rng.integers(-45, 46))
    if rng.random() < 0.5:
        gx = np.linspace(0, float(rng.uniform(-0.8, 0.8)) * W, W, dtype=np.float32).reshape(1, 1, 1, W)
        gy = np.linspace(0, float(rng.uniform(-0.8, 0.8)) * H, H, dtype=np.float32).reshape(1, 1, H, 1)
        oy += gx
        ox += gy
    if rng.random() < 0.3:
        lo, hi = sorted(int(v) for v in rng.integers(0, 26, 2))
        oy[:, lo:hi] += np.float32(rng.integers(-40, 41))
    return oy, ox


# (FUZZ_MOTION_CASES / FUZZ_CASES: longer one-off soaks, e.g. 600 each after a change to the windows)
@pytest.mark.parametrize("case", range(int(os.environ.get("FUZZ_MOTION_CASES", "24"))))
def test_random_motion_fields_against_oracle(dev, oracle, case):
    """The window-following machinery (first centres, votes, further windows, the three fix-up
    forms) on random piecewise-coherent motion, op and fused head, both tile shapes."""
    import artemis_amd
    from artemis_amd import _lib
    from test_head import head_operands
    lib = _lib.load()
    rng = np.random.default_rng(7000 + case)
    N = int(rng.integers(1, 3))
    H = int(rng.choice([16, 33, 48, 70, 100]))
    W = int(rng.choice([64, 100, 128, 150, 257, 300]))
    radius = int(rng.choice([-1, -2, -3, 2, 4, 8, 14, 24]))
    big_tiles = bool(rng.integers(2))
    old = lib.artemis_synth_set_small_grid_tiles(0 if big_tiles else 1 << 40)
    tag = f"case {case} N{N} {H}x{W} R{radius} big{big_tiles}"
    try:
        if case % 3:
            inp, w, _, _, gout = make_operands(N, 3, H, W, regime="zero", seed=case)
            oy, ox = random_motion_field(rng, N, H, W)
            t = [torch.from_numpy(a).to(dev).requires_grad_(i > 0) for i, a in enumerate((inp, w, oy, ox))]
            with artemis_amd.synth_options(tile_radius=radius):
                out = artemis_amd.FunctionSynth.apply(*t, 1)
                out.backward(torch.from_numpy(gout).to(dev))
            torch.cuda.synchronize()
            close(out, oracle.forward(inp, w, oy, ox, 1), FWD_RTOL, "output " + tag)
            _, rgw, rga, rgb = oracle.backward(gout, inp, w, oy, ox, 1, with_input=False)
            term = float(np.abs(gout).max() * np.abs(w).max() * np.abs(inp).max())
            close(t[1].grad, rgw, BWD_RTOL, "grad_weight " + tag)
            close(t[2].grad, rga, BWD_RTOL, "grad_offset_y " + tag, floor=term)
            close(t[3].grad, rgb, BWD_RTOL, "grad_offset_x " + tag, floor=term)
        else:
            T = int(rng.integers(1, 5))
            frames, w, a, b, occ, gout = head_operands(N, T, H, W, "zero", seed=case)
            for f in range(T):
                a[:, f], b[:, f] = random_motion_field(rng, N, H, W)
            t = [torch.from_numpy(x).to(dev).requires_grad_(True) for x in (w, a, b, occ)]
            with artemis_amd.synth_options(tile_radius=radius):
                out = artemis_amd.chrono_head([torch.from_numpy(f).to(dev) for f in frames], *t)
                out.backward(torch.from_numpy(gout).to(dev))
            torch.cuda.synchronize()
            ref, _ = oracle.head_forward(frames, w, a, b, occ, 1)
            close(out, ref, FWD_RTOL, "framet " + tag)
            term = float(np.abs(gout).max() * np.abs(w).max())
            for g, r, name in zip(t, oracle.head_backward(gout, frames, w, a, b, occ, 1),
                                  ("grad_weights", "grad_alphas", "grad_betas", "grad_occlusion")):
                close(g.grad, r, BWD_RTOL, f"head {name} {tag}", floor=term)
    finally:
        lib.artemis_synth_set_small_grid_tiles(old)
