"""GPU parity tests (-m gpu): the HIP path, reached through the C ABI, against the CPU oracle on
the same seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full
sizes -- through size-independent properties.

Tolerances (BASELINE.json north_star): forward 1e-5 relative, backward 1e-4 relative, taken
against the scale of the oracle's tensor (|hip - oracle| <= rtol * |oracle| + rtol * max|oracle|):
both sides are fp32 but the device contracts a*b+c into FMAs and sums the four corner products
in a different association, so last-bit differences are expected.  Integer-exact cases (one-hot
taps, integer offsets) are compared bit-exactly.  Deterministic gradInput is bit-exact run to run.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, make_operands, parity_close

pytestmark = pytest.mark.gpu

FWD_RTOL = 1e-5
BWD_RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import artemis_amd
    artemis_amd._lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def close(actual, desired, rtol, what=""):
    """conftest.parity_close: norm-relative and elementwise-relative asserts, worst errors recorded."""
    parity_close(actual, desired, rtol, what)


def run_hip(dev, inp, w, oy, ox, gout, d, radius=-1, deterministic=False, need=(True,) * 4):
    import artemis_amd
    t = [torch.from_numpy(a).to(dev) for a in (inp, w, oy, ox)]
    for x, n in zip(t, need):
        x.requires_grad_(n)
    with artemis_amd.synth_options(tile_radius=radius, deterministic=deterministic):
        out = artemis_amd.FunctionSynth.apply(*t, d)
        if any(need):
            out.backward(torch.from_numpy(gout).to(dev))
    torch.cuda.synchronize()
    return out.detach().cpu().numpy(), [None if x.grad is None else x.grad.cpu().numpy() for x in t]


@pytest.mark.parametrize("radius", [-1, 0, 2, 4, 8, 14, 24])      # default, direct, every built radius
@pytest.mark.parametrize("name", golden_names())
def test_golden(dev, name, radius):
    g = load_golden(name)
    d = int(g["dilation"])
    out, (gi, gw, ga, gb) = run_hip(dev, g["input"], g["weight"], g["offset_y"], g["offset_x"],
                                    g["grad_output"], d, radius)
    close(out, g["output"], FWD_RTOL, "output")
    close(gw, g["grad_weight"], BWD_RTOL, "grad_weight")
    close(ga, g["grad_offset_y"], BWD_RTOL, "grad_offset_y")
    close(gb, g["grad_offset_x"], BWD_RTOL, "grad_offset_x")
    close(gi, g["grad_input"], BWD_RTOL, "grad_input")


SHAPES = [(1, 3, 33, 70), (2, 3, 17, 129), (1, 3, 64, 64), (3, 3, 1, 1), (1, 3, 5, 200),
          (1, 3, 16, 64), (2, 3, 48, 65)]
REGIMES = ["init", "gauss1", "gauss4", "uniform3", "clamp40", "integer"]


@pytest.mark.parametrize("radius", [-1, 0, 2, 8])
@pytest.mark.parametrize("regime", REGIMES)
@pytest.mark.parametrize("shape", SHAPES)
def test_seeded_against_oracle(dev, oracle, shape, regime, radius):
    N, C, H, W = shape
    inp, w, oy, ox, gout = make_operands(N, C, H, W, 5, 1, regime, seed=H * 131 + W)
    out, (gi, gw, ga, gb) = run_hip(dev, inp, w, oy, ox, gout, 1, radius)
    close(out, oracle.forward(inp, w, oy, ox, 1), FWD_RTOL, "output")
    rgi, rgw, rga, rgb = oracle.backward(gout, inp, w, oy, ox, 1)
    close(gw, rgw, BWD_RTOL, "grad_weight")
    close(ga, rga, BWD_RTOL, "grad_offset_y")
    close(gb, rgb, BWD_RTOL, "grad_offset_x")
    close(gi, rgi, BWD_RTOL, "grad_input")


@pytest.mark.parametrize("cfg", [(2, 2, 5, 9, 3, 2), (1, 4, 4, 5, 1, 1), (1, 5, 9, 70, 7, 1),
                                 (2, 1, 6, 66, 5, 3), (1, 9, 3, 4, 3, 1), (1, 3, 20, 37, 5, 2)])
def test_generic_filter_dilation_channels(dev, oracle, cfg):
    N, C, H, W, k, d = cfg
    inp, w, oy, ox, gout = make_operands(N, C, H, W, k, d, "uniform3", seed=k * 7 + d)
    out, (gi, gw, ga, gb) = run_hip(dev, inp, w, oy, ox, gout, d)
    close(out, oracle.forward(inp, w, oy, ox, d), FWD_RTOL)
    rgi, rgw, rga, rgb = oracle.backward(gout, inp, w, oy, ox, d)
    close(gw, rgw, BWD_RTOL)
    close(ga, rga, BWD_RTOL)
    close(gb, rgb, BWD_RTOL)
    close(gi, rgi, BWD_RTOL)


def test_weight_regime_heads_2_and_3(dev, oracle):
    # raw N(0, 0.05^2) weights (heads 2/3 have no softmax: model/chrono_synth.py:39)
    inp, w, oy, ox, gout = make_operands(2, 3, 40, 72, regime="gauss1", seed=3, weights="raw")
    out, (gi, gw, ga, gb) = run_hip(dev, inp, w, oy, ox, gout, 1)
    close(out, oracle.forward(inp, w, oy, ox, 1), FWD_RTOL)
    rgi, rgw, rga, rgb = oracle.backward(gout, inp, w, oy, ox, 1)
    close(gw, rgw, BWD_RTOL)
    close(ga, rga, BWD_RTOL)
    close(gb, rgb, BWD_RTOL)
    close(gi, rgi, BWD_RTOL)


def test_config1_shapes(dev, oracle):
    # BASELINE.json configs[0]: the three heads of a 64x64 forward: (1,25,{16,32,64}^2)
    for hw in (16, 32, 64):
        inp, w, oy, ox, gout = make_operands(1, 3, hw, hw, regime="init", seed=hw)
        out, _ = run_hip(dev, inp, w, oy, ox, gout, 1, need=(False,) * 4)
        close(out, oracle.forward(inp, w, oy, ox, 1), FWD_RTOL)


def test_config2_shape_sample(dev, oracle):
    # BASELINE.json configs[1] shape (256x448), 2 of its 16 samples so the oracle stays quick
    oracle.set_threads(8)
    inp, w, oy, ox, gout = make_operands(2, 3, 256, 448, regime="gauss1", seed=2)
    out, (gi, gw, ga, gb) = run_hip(dev, inp, w, oy, ox, gout, 1)
    close(out, oracle.forward(inp, w, oy, ox, 1), FWD_RTOL)
    rgi, rgw, rga, rgb = oracle.backward(gout, inp, w, oy, ox, 1, with_input=False)
    close(gw, rgw, BWD_RTOL)
    close(ga, rga, BWD_RTOL)
    close(gb, rgb, BWD_RTOL)
    oracle.set_threads(1)


def test_needs_input_grad_subsets(dev, oracle):
    inp, w, oy, ox, gout = make_operands(1, 3, 18, 66, regime="gauss1", seed=9)
    rgi, rgw, rga, rgb = oracle.backward(gout, inp, w, oy, ox, 1)
    for need in [(False, True, True, True), (False, True, False, False),
                 (False, False, True, False), (False, False, False, True),
                 (True, False, False, False), (False, False, False, False)]:
        out, grads = run_hip(dev, inp, w, oy, ox, gout, 1, need=need)
        for g, ref, n in zip(grads, (rgi, rgw, rga, rgb), need):
            if n:
                close(g, ref, BWD_RTOL)
            else:
                assert g is None


def test_deterministic_grad_input_is_bit_exact_and_accurate(dev, oracle):
    inp, w, oy, ox, gout = make_operands(2, 3, 37, 90, regime="gauss4", seed=12)
    ref = oracle.grad_input(gout, inp.shape, w, oy, ox, 1)
    runs = [run_hip(dev, inp, w, oy, ox, gout, 1, deterministic=True,
                    need=(True, False, False, False))[1][0] for _ in range(3)]
    close(runs[0], ref, BWD_RTOL)
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])
    # all contributions clamped onto the border: the worst case for the fixed-point range
    oy2 = np.full_like(oy, 500.0)
    out, grads = run_hip(dev, inp, w, oy2, ox, gout, 1, deterministic=True,
                         need=(True, False, False, False))
    close(grads[0], oracle.grad_input(gout, inp.shape, w, oy2, ox, 1), BWD_RTOL)


def test_deterministic_via_torch_switch(dev, oracle):
    inp, w, oy, ox, gout = make_operands(1, 3, 9, 70, regime="gauss1", seed=13)
    torch.use_deterministic_algorithms(True)
    try:
        a = run_hip(dev, inp, w, oy, ox, gout, 1)[1]
        b = run_hip(dev, inp, w, oy, ox, gout, 1)[1]
    finally:
        torch.use_deterministic_algorithms(False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    close(a[0], oracle.grad_input(gout, inp.shape, w, oy, ox, 1), BWD_RTOL)


def test_nonfinite_and_huge_offsets_do_not_fault(dev, oracle):
    inp, w, oy, ox, gout = make_operands(1, 3, 20, 70, regime="gauss1", seed=14)
    oy[0, 3, 1, 1] = 3e9
    ox[0, 4, 2, 2] = -3e9
    oy[0, 5, 0, 0] = np.nan
    ox[0, 6, 7, 7] = np.inf
    oy[0, 7, 8, 8] = -2147483648.0
    # the five poisoned pixels hold meaningless (huge, cancelling or non-finite) values whose
    # rounding is association-dependent; every other pixel must be unaffected and in tolerance
    poisoned = np.zeros((20, 70), bool)
    for y, x in ((1, 1), (2, 2), (0, 0), (7, 7), (8, 8)):
        poisoned[y, x] = True
    ref = oracle.forward(inp, w, oy, ox, 1)
    rgi, rgw, rga, rgb = oracle.backward(gout, inp, w, oy, ox, 1)
    for radius in (-1, 0):
        out, (gi, gw, ga, gb) = run_hip(dev, inp, w, oy, ox, gout, 1, radius,
                                        need=(False, True, True, True))
        close(out[..., ~poisoned], ref[..., ~poisoned], FWD_RTOL)
        close(gw[..., ~poisoned], rgw[..., ~poisoned], BWD_RTOL)
        close(ga[..., ~poisoned], rga[..., ~poisoned], BWD_RTOL)
        close(gb[..., ~poisoned], rgb[..., ~poisoned], BWD_RTOL)
        # NaN / inf offsets poison exactly their own pixel, as in the oracle
        assert np.array_equal(np.isfinite(out), np.isfinite(ref))


def test_offset_radius_helper(dev):
    import artemis_amd
    oy = torch.zeros(1, 25, 8, 8, device=dev)
    ox = torch.zeros(1, 25, 8, 8, device=dev)
    assert int(artemis_amd.offset_radius(oy, ox)) == 0
    oy[0, 3, 2, 2] = -6.9
    ox[0, 1, 1, 1] = 3.2
    assert int(artemis_amd.offset_radius(oy, ox)) == 6
    ox[0, 0, 0, 0] = float("nan")
    oy[0, 0, 0, 1] = -1e20
    assert int(artemis_amd.offset_radius(oy, ox)) == 2**31 - 1


def test_suggest_tile_radius(dev):
    import artemis_amd
    oy = torch.randn(1, 25, 16, 64, device=dev)
    ox = torch.randn(1, 25, 16, 64, device=dev)
    assert artemis_amd.suggest_tile_radius(oy, ox) == -3     # N(0, 1): every whole-pixel offset within 8: the lean shape
    oy[0, 7, 3, 3] = 8.9
    assert artemis_amd.suggest_tile_radius(oy, ox) == -3
    oy[0, 7, 3, 3] = 14.9
    assert artemis_amd.suggest_tile_radius(oy, ox) == 14
    oy[0, 7, 3, 3] = -15.0
    assert artemis_amd.suggest_tile_radius(oy, ox) == 24
    with artemis_amd.synth_options(tile_radius=artemis_amd.suggest_tile_radius(oy, ox)):
        assert artemis_amd.options.tile_radius == 24


def test_runs_on_the_callers_stream(dev, oracle):
    import artemis_amd
    inp, w, oy, ox, gout = make_operands(1, 3, 32, 64, regime="gauss1", seed=15)
    t = [torch.from_numpy(a).to(dev) for a in (inp, w, oy, ox)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = artemis_amd.synth(*t, 1)
    s.synchronize()
    close(out, oracle.forward(inp, w, oy, ox, 1), FWD_RTOL)


def test_raw_c_abi_call(dev, oracle):
    """Straight through ctypes with raw device pointers: no torch type crosses the boundary."""
    from artemis_amd import _lib
    lib = _lib.load()
    inp, w, oy, ox, gout = make_operands(1, 3, 10, 66, regime="uniform3", seed=16)
    t = [torch.from_numpy(a).to(dev) for a in (inp, w, oy, ox, gout)]
    out = torch.empty(1, 3, 10, 66, device=dev)
    gw, ga, gb = (torch.empty_like(t[1]) for _ in range(3))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.artemis_synth_forward(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                     t[3].data_ptr(), out.data_ptr(), 1, 3, 10, 66, 5, 1, -1, st) == 0
    assert lib.artemis_synth_backward(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                      t[3].data_ptr(), t[4].data_ptr(), None, gw.data_ptr(),
                                      ga.data_ptr(), gb.data_ptr(), 1, 3, 10, 66, 5, 1, 0, -1,
                                      None, 0, st) == 0
    torch.cuda.synchronize()
    close(out, oracle.forward(inp, w, oy, ox, 1), FWD_RTOL)
    _, rgw, rga, rgb = oracle.backward(gout, inp, w, oy, ox, 1, with_input=False)
    close(gw, rgw, BWD_RTOL)
    close(ga, rga, BWD_RTOL)
    close(gb, rgb, BWD_RTOL)
    # deterministic mode without a workspace is refused with a clear message
    gi = torch.empty_like(t[0])
    rc = lib.artemis_synth_backward(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                    t[3].data_ptr(), t[4].data_ptr(), gi.data_ptr(), None, None,
                                    None, 1, 3, 10, 66, 5, 1, 1, -1, None, 0, st)
    assert rc == 1 and "workspace" in lib.artemis_synth_last_error().decode()


# ---- full-size (1080p, BASELINE.json metric shape) size-independent properties ----

@pytest.fixture(scope="module")
def full(dev):
    torch.manual_seed(103)
    N, C, H, W = 1, 3, 1080, 1920
    frame = torch.rand(N, C, H, W, device=dev)
    inp = torch.nn.functional.pad(frame, (2, 2, 2, 2), mode="replicate").contiguous()
    w = torch.softmax(torch.randn(N, 25, H, W, device=dev), 1)
    oy = torch.randn(N, 25, H, W, device=dev)
    ox = torch.randn(N, 25, H, W, device=dev)
    return inp, w, oy, ox


def test_full_size_onehot_zero_offset_is_exact_shift(dev, full):
    import artemis_amd
    inp, w, oy, ox = full
    H, W = w.shape[2:]
    z = torch.zeros_like(oy)
    for r, s in ((0, 0), (2, 2), (4, 1)):
        oh = torch.zeros_like(w)
        oh[:, r * 5 + s] = 1
        out = artemis_amd.synth(inp, oh, z, z, 1)
        assert torch.equal(out, inp[:, :, r:r + H, s:s + W])


def test_full_size_integer_offsets_are_exact_gathers(dev, full):
    import artemis_amd
    inp, w, oy, ox = full
    H, W = w.shape[2:]
    oh = torch.zeros_like(w)
    oh[:, 7] = 1  # tap (1, 2)
    iy = torch.randint(-6, 7, oy.shape, device=dev).float()
    ix = torch.randint(-6, 7, ox.shape, device=dev).float()
    out = artemis_amd.synth(inp, oh, iy, ix, 1)
    ys = (torch.arange(H, device=dev).view(1, H, 1) + 1 + iy[:, 7].long()).clamp(0, H + 3)
    xs = (torch.arange(W, device=dev).view(1, 1, W) + 2 + ix[:, 7].long()).clamp(0, W + 3)
    want = inp[0][:, ys[0], xs[0]].unsqueeze(0)
    assert torch.equal(out, want)


def test_full_size_tiled_equals_direct_and_is_linear(dev, full):
    import artemis_amd
    inp, w, oy, ox = full
    a = artemis_amd.synth(inp, w, oy, ox, 1)
    with artemis_amd.synth_options(tile_radius=0):
        b = artemis_amd.synth(inp, w, oy, ox, 1)
    assert (a - b).abs().max().item() <= FWD_RTOL * a.abs().max().item()
    c = artemis_amd.synth(inp, 2 * w, oy, ox, 1)
    assert torch.equal(c, 2 * a)  # power-of-two scaling commutes with every rounding


def test_full_size_forward_backward_identity(dev, full):
    # sum_t w_t * gradWeight_t == sum_c gradOutput_c * output_c at every pixel, because
    # output_c = sum_t w_t * blend_tc and gradWeight_t = sum_c gradOutput_c * blend_tc.
    import artemis_amd
    inp, w, oy, ox = full
    w = w.clone().requires_grad_(True)
    oy = oy.clone().requires_grad_(True)
    ox = ox.clone().requires_grad_(True)
    out = artemis_amd.synth(inp, w, oy, ox, 1)
    gout = torch.randn_like(out)
    out.backward(gout)
    lhs = (w.detach() * w.grad).sum(1)
    rhs = (gout * out.detach()).sum(1)
    assert (lhs - rhs).abs().max().item() <= BWD_RTOL * rhs.abs().max().item()
    assert torch.isfinite(oy.grad).all() and torch.isfinite(ox.grad).all()
    # offset gradients of a constant frame vanish (all four corners equal)
    const = torch.full_like(inp, 0.37)
    oy.grad = None
    ox.grad = None
    artemis_amd.synth(const, w, oy, ox, 1).backward(gout)
    assert oy.grad.abs().max().item() <= 1e-5 and ox.grad.abs().max().item() <= 1e-5


def test_full_size_grad_input_conserves_mass(dev, full):
    # sum over the frame of gradInput_c == sum over pixels of gradOutput_c * sum_t w_t: the four
    # bilinear coefficients of a tap always sum to 1, clamped or not.
    import artemis_amd
    inp, w, oy, ox = full
    inp = inp.clone().requires_grad_(True)
    out = artemis_amd.synth(inp, w, oy, ox, 1)
    gout = torch.randn_like(out)
    out.backward(gout)
    lhs = inp.grad.double().sum((2, 3))
    rhs = (gout.double() * w.double().sum(1, keepdim=True)).sum((2, 3))
    assert (lhs - rhs).abs().max().item() <= 1e-3 * gout.abs().double().sum().item() / 1e3
    with artemis_amd.synth_options(deterministic=True):
        inp.grad = None
        artemis_amd.synth(inp, w, oy, ox, 1).backward(gout)
        g1 = inp.grad.clone()
        inp.grad = None
        artemis_amd.synth(inp, w, oy, ox, 1).backward(gout)
        assert torch.equal(g1, inp.grad)


def test_nonfinite_frame_texels_poison_only_the_pixels_that_read_them(dev, oracle):
    """Inf / NaN texels in the FRAME (not the offsets) with many out-of-tile taps: a far tap is
    dropped from the LDS pass by a select, so an unrelated non-finite texel under its clamped LDS
    address must not leak into pixels the oracle leaves finite (0 * Inf = NaN otherwise)."""
    inp, w, oy, ox, gout = make_operands(1, 3, 40, 200, regime="gauss4", seed=41)
    rng = np.random.default_rng(41)
    for _ in range(8):
        y, x = int(rng.integers(0, 44)), int(rng.integers(0, 204))
        inp[0, int(rng.integers(0, 3)), y, x] = [np.inf, -np.inf, np.nan][int(rng.integers(0, 3))]
    # the last texels of the frame / of the first tiles: where clamped LDS addresses point
    inp[0, 0, 43, 203] = np.inf
    inp[0, 1, 31, 131] = np.nan
    ref = oracle.forward(inp, w, oy, ox, 1)
    _, rgw, rga, rgb = oracle.backward(gout, inp, w, oy, ox, 1, with_input=False)
    assert np.isfinite(ref).mean() > 0.5
    for radius in (-1, 2, 4, 0):
        out, (gi, gw, ga, gb) = run_hip(dev, inp, w, oy, ox, gout, 1, radius,
                                        need=(False, True, True, True))
        assert np.array_equal(np.isfinite(out), np.isfinite(ref)), f"radius {radius}"
        ok = np.isfinite(ref)
        close(out[ok], ref[ok], FWD_RTOL, "output")
        for g, r, name in ((gw, rgw, "grad_weight"), (ga, rga, "grad_offset_y"),
                           (gb, rgb, "grad_offset_x")):
            assert np.array_equal(np.isfinite(g), np.isfinite(r)), f"{name}, radius {radius}"
            ok = np.isfinite(r)
            close(g[ok], r[ok], BWD_RTOL, name)


def test_tiny_gradients_do_not_overflow_the_fixed_point_scale(dev, oracle):
    """gradOutput ~ 1e-25 and weights ~ 1e-8: the fixed-point scale of gradInput would need 2^S
    beyond fp32's range; it is clamped, and the (tiny) result still follows the fp64 adjoint."""
    inp, w, oy, ox, gout = make_operands(1, 3, 24, 100, regime="gauss1", seed=42)
    w = (w * 1e-8).astype(np.float32)
    gout = (gout * 1e-25).astype(np.float32)
    ref = oracle.grad_input(gout, inp.shape, w, oy, ox, 1)
    assert 0 < np.abs(ref).max() < 1e-30
    for radius, det in ((-1, False), (-1, True), (0, True), (0, False)):
        gi = run_hip(dev, inp, w, oy, ox, gout, 1, radius, deterministic=det,
                     need=(True, False, False, False))[1][0]
        close(gi, ref, BWD_RTOL, "grad_input")


def test_options_in_force_at_forward_govern_the_backward(dev, oracle):
    """deterministic=True around the forward only: autograd runs backward later, outside the
    `with`, and must still use the deterministic gradInput path (bit-equal run to run and equal to
    a run with the option set throughout)."""
    import artemis_amd
    inp, w, oy, ox, gout = make_operands(2, 3, 37, 90, regime="gauss4", seed=43)

    def run(scope):
        t = [torch.from_numpy(a).to(dev).requires_grad_(i == 0) for i, a in enumerate((inp, w, oy, ox))]
        with artemis_amd.synth_options(deterministic=True, tile_radius=4):
            out = artemis_amd.FunctionSynth.apply(*t, 1)
            if scope == "inside":
                out.backward(torch.from_numpy(gout).to(dev))
        if scope == "outside":
            assert artemis_amd.options.deterministic is False
            out.backward(torch.from_numpy(gout).to(dev))
        torch.cuda.synchronize()
        return t[0].grad.cpu().numpy()

    a, b, c = run("outside"), run("outside"), run("inside")
    assert np.array_equal(a, b) and np.array_equal(a, c)
    close(a, oracle.grad_input(gout, inp.shape, w, oy, ox, 1), BWD_RTOL, "grad_input")
