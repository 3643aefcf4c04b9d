"""Large coherent motion: the LDS tile re-centres on the tile's mean displacement.
Parity against the oracle where every tile is shifted, where the shift varies across the frame and
where it pushes the window across the frame borders."""
import numpy as np
import pytest
import torch

from conftest import make_operands, parity_close
from test_head import head_operands

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


def moving_offsets(shape, kind, rng):
    N, K, H, W = shape
    noise = lambda s: rng.standard_normal(shape, dtype=np.float32) * np.float32(s)
    if kind == "pan":                       # everything displaced by (20, -35) px
        return noise(1) + np.float32(20), noise(1) - np.float32(35)
    if kind == "shear":                     # displacement grows across the frame: -40 .. +40 px
        ramp_y = np.linspace(-40, 40, H, dtype=np.float32).reshape(1, 1, H, 1)
        ramp_x = np.linspace(30, -30, W, dtype=np.float32).reshape(1, 1, 1, W)
        return noise(0.5) + ramp_x + 0 * ramp_y, noise(0.5) + ramp_y + 0 * ramp_x
    if kind == "offframe":                  # window pushed far outside the frame: all clamps
        return noise(2) - np.float32(500), noise(2) + np.float32(700)
    if kind == "split":                     # two objects: taps 0..12 go one way, 13..24 the other
        oy, ox = noise(0.5), noise(0.5)
        oy[:, :13] += 25
        oy[:, 13:] -= 25
        return oy, ox
    raise KeyError(kind)


@pytest.mark.parametrize("radius", [-1, 2, 8])
@pytest.mark.parametrize("kind", ["pan", "shear", "offframe", "split"])
def test_moving_content_against_oracle(dev, oracle, kind, radius):
    import artemis_amd
    rng = np.random.default_rng(77)
    N, H, W = 2, 70, 150
    inp, w, _, _, gout = make_operands(N, 3, H, W, regime="zero", seed=5)
    oy, ox = moving_offsets((N, 25, H, W), kind, rng)
    for det in (False, True):
        t = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (inp, w, oy, ox)]
        with artemis_amd.synth_options(tile_radius=radius, deterministic=det):
            out = artemis_amd.FunctionSynth.apply(*t, 1)
            out.backward(torch.from_numpy(gout).to(dev))
        torch.cuda.synchronize()
        close(out, oracle.forward(inp, w, oy, ox, 1), FWD_RTOL, "output")
        ref = oracle.backward(gout, inp, w, oy, ox, 1)
        term = float(np.abs(gout).max() * np.abs(w).max() * np.abs(inp).max())
        for g, r, name in zip(t, ref, ("grad_input", "grad_weight", "grad_offset_y", "grad_offset_x")):
            close(g.grad, r, BWD_RTOL, f"{name} {kind} R{radius}", floor=term if "offset" in name else 1e-30)


def test_moving_content_head(dev, oracle):
    import artemis_amd
    rng = np.random.default_rng(3)
    B, T, H, W = 1, 4, 40, 130
    frames, w, a, b, occ, gout = head_operands(B, T, H, W, "gauss1", seed=2)
    for f in range(T):                      # a different motion per context frame
        a[:, f] += np.float32((-30, -10, 12, 33)[f])
        b[:, f] += np.float32((18, 6, -7, -21)[f])
    t = [torch.from_numpy(x).to(dev).requires_grad_(True) for x in (w, a, b, occ)]
    out = artemis_amd.chrono_head([torch.from_numpy(f).to(dev) for f in frames], *t)
    out.backward(torch.from_numpy(gout).to(dev))
    torch.cuda.synchronize()
    ref, _ = oracle.head_forward(frames, w, a, b, occ, 1)
    close(out, ref, FWD_RTOL, "framet")
    term = float(np.abs(gout).max() * np.abs(w).max())
    for g, r in zip(t, oracle.head_backward(gout, frames, w, a, b, occ, 1)):
        close(g.grad, r, BWD_RTOL, floor=term)
