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


# ---- several windows per tile (motion boundaries, steep flow): artemis_amd/csrc/synth_device.hpp,
# "Several windows per tile"

def boundary_offsets(shape, kind, rng):
    N, K, H, W = shape
    noise = lambda s: rng.standard_normal(shape, dtype=np.float32) * np.float32(s)
    oy, ox = noise(0.4), noise(0.4)
    if kind == "object":          # an object moving (+28, -22) px in front of a still background
        oy[:, :, 10:45, 40:110] += 28
        ox[:, :, 10:45, 40:110] -= 22
    elif kind == "three":         # three motions meeting inside one tile
        oy[:, :, :, :50] += 30
        ox[:, :, :, 50:90] += 40
        oy[:, :, :, 90:] -= 35
        ox[:, :, :, 90:] -= 30
    elif kind == "four":          # four motions: more than the windows a tile can stage (3)
        oy[:, :, :, :30] += 30
        ox[:, :, :, 30:60] += 40
        oy[:, :, :, 60:90] -= 35
        ox[:, :, :, 90:] -= 45
    elif kind == "steep":         # 0.8 px of displacement per pixel, both axes
        oy += np.linspace(-60, 60, W, dtype=np.float32).reshape(1, 1, 1, W)
        ox += np.linspace(40, -40, H, dtype=np.float32).reshape(1, 1, H, 1)
    elif kind == "taps":          # the taps of every pixel point at two objects and the background
        oy[:, 5:12] += 25
        ox[:, 18:] -= 31
    elif kind == "scattered":     # independent large offsets: no window helps, passes must not hurt
        oy, ox = noise(16), noise(16)
    else:
        raise KeyError(kind)
    return oy, ox


@pytest.mark.parametrize("radius", [-1, 4])
@pytest.mark.parametrize("kind", ["object", "three", "four", "steep", "taps", "scattered"])
def test_several_windows_per_tile_against_oracle(dev, oracle, kind, radius):
    import artemis_amd
    rng = np.random.default_rng(91)
    N, H, W = 1, 70, 150
    inp, w, _, _, gout = make_operands(N, 3, H, W, regime="zero", seed=6)
    oy, ox = boundary_offsets((N, 25, H, W), kind, rng)
    t = [torch.from_numpy(a).to(dev).requires_grad_(i > 0) for i, a in enumerate((inp, w, oy, ox))]
    with artemis_amd.synth_options(tile_radius=radius):
        out = artemis_amd.FunctionSynth.apply(*t, 1)
        out.backward(torch.from_numpy(gout).to(dev))
        again = artemis_amd.FunctionSynth.apply(*[x.detach() for x in t], 1)
    torch.cuda.synchronize()
    assert torch.equal(out, again)                       # the vote is a pure function of the data
    close(out, oracle.forward(inp, w, oy, ox, 1), FWD_RTOL, f"output {kind}")
    _, rgw, rga, rgb = oracle.backward(gout, inp, w, oy, ox, 1, with_input=False)
    term = float(np.abs(gout).max() * np.abs(w).max() * np.abs(inp).max())
    close(t[1].grad, rgw, BWD_RTOL, f"grad_weight {kind}")
    close(t[2].grad, rga, BWD_RTOL, f"grad_offset_y {kind}", floor=term)
    close(t[3].grad, rgb, BWD_RTOL, f"grad_offset_x {kind}", floor=term)


@pytest.mark.parametrize("kind", ["object", "steep", "taps"])
def test_several_windows_head_and_big_tiles(dev, oracle, kind):
    """The fused head (a window sequence per context frame) on a grid large enough for the big
    128 x 16 tiles, with the small-grid switch off."""
    import artemis_amd
    from artemis_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(17)
    B, T, H, W = 1, 3, 70, 300
    frames, w, a, b, occ, gout = head_operands(B, T, H, W, "zero", seed=4)
    for f in range(T):
        oy, ox = boundary_offsets((B, 25, 70, 150), kind, rng)
        a[:, f] = np.tile(oy, (1, 1, 1, 2)) * np.float32((1, -1, 0.5)[f])
        b[:, f] = np.tile(ox, (1, 1, 1, 2)) * np.float32((1, 0.5, -1)[f])
    old = lib.artemis_synth_set_small_grid_tiles(0)
    try:
        t = [torch.from_numpy(x).to(dev).requires_grad_(True) for x in (w, a, b, occ)]
        out = artemis_amd.chrono_head([torch.from_numpy(f).to(dev) for f in frames], *t)
        out.backward(torch.from_numpy(gout).to(dev))
        torch.cuda.synchronize()
    finally:
        lib.artemis_synth_set_small_grid_tiles(old)
    ref, _ = oracle.head_forward(frames, w, a, b, occ, 1)
    close(out, ref, FWD_RTOL, f"framet {kind}")
    term = float(np.abs(gout).max() * np.abs(w).max())
    for g, r, name in zip(t, oracle.head_backward(gout, frames, w, a, b, occ, 1),
                          ("grad_weights", "grad_alphas", "grad_betas", "grad_occlusion")):
        close(g.grad, r, BWD_RTOL, f"{name} {kind}", floor=term)
