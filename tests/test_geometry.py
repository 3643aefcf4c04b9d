"""Explicit frame geometry (artemis_synth_*_ex, FunctionSynthFrame): the CPU oracle's *_ex
restatement against the compositions the reference makes (ReplicationPad2d + op,
model/chrono_synth.py:70,125-126; row strips of one frame), and -- marked gpu -- the HIP path
against that oracle.  Bit-exact where both sides run the same arithmetic on the same texels."""
import numpy as np
import pytest
import torch

from conftest import make_operands, parity_close

FWD_RTOL = 1e-5
BWD_RTOL = 1e-4


def unpadded_operands(N, C, H, W, k=5, d=1, regime="gauss1", seed=0):
    inp, w, oy, ox, gout = make_operands(N, C, H, W, k, d, regime, seed)
    p = (k - 1) * d // 2
    frame = np.ascontiguousarray(inp[:, :, p:p + H, p:p + W])
    padded = np.pad(frame, ((0, 0), (0, 0), (p, p), (p, p)), mode="edge")
    return frame, padded, w, oy, ox, gout, p


def fold_pad_grad(gpad, p, H, W):
    """Adjoint of replication padding: border gradients add onto the edge rows/columns."""
    g = gpad.astype(np.float64)
    g[:, :, p, :] += g[:, :, :p, :].sum(2)
    g[:, :, p + H - 1, :] += g[:, :, p + H:, :].sum(2)
    g = g[:, :, p:p + H, :]
    g[:, :, :, p] += g[:, :, :, :p].sum(3)
    g[:, :, :, p + W - 1] += g[:, :, :, p + W:].sum(3)
    return g[:, :, :, p:p + W]


# ---------------------------------------------------------------- oracle (CPU)

@pytest.mark.parametrize("cfg", [(2, 3, 9, 14, 5, 1, "uniform3"), (1, 3, 6, 70, 5, 1, "clamp40"),
                                 (1, 2, 7, 8, 3, 2, "gauss4"), (1, 3, 1, 1, 5, 1, "gauss1")])
def test_oracle_unpadded_equals_pad_then_op(oracle, cfg):
    N, C, H, W, k, d, regime = cfg
    frame, padded, w, oy, ox, gout, p = unpadded_operands(N, C, H, W, k, d, regime, seed=H + W)
    ref = oracle.forward(padded, w, oy, ox, d)
    got = oracle.forward_ex(frame, w, oy, ox, d, origin=(-p, -p))
    np.testing.assert_array_equal(got, ref)
    rgi, rgw, rga, rgb = oracle.backward(gout, padded, w, oy, ox, d)
    gi, gw, ga, gb = oracle.backward_ex(gout, frame, w, oy, ox, d, origin=(-p, -p))
    np.testing.assert_array_equal(gw, rgw)
    np.testing.assert_array_equal(ga, rga)
    np.testing.assert_array_equal(gb, rgb)
    np.testing.assert_allclose(gi, fold_pad_grad(rgi, p, H, W), rtol=1e-6, atol=1e-6)


def test_oracle_ex_reference_layout_is_identity(oracle):
    inp, w, oy, ox, gout = make_operands(2, 3, 8, 11, regime="gauss4", seed=4)
    np.testing.assert_array_equal(oracle.forward_ex(inp, w, oy, ox, 1), oracle.forward(inp, w, oy, ox, 1))
    a = oracle.backward_ex(gout, inp, w, oy, ox, 1)
    b = oracle.backward(gout, inp, w, oy, ox, 1)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_oracle_strips_tile_the_full_result(oracle):
    inp, w, oy, ox, gout = make_operands(1, 3, 23, 19, regime="clamp40", seed=8)
    full = oracle.forward(inp, w, oy, ox, 1)
    fgi, fgw, fga, fgb = oracle.backward(gout, inp, w, oy, ox, 1)
    gi_sum = np.zeros_like(fgi, dtype=np.float64)
    for y0, y1 in ((0, 7), (7, 8), (8, 23)):
        sl = np.s_[:, :, y0:y1]
        ws, oys, oxs, gs = (np.ascontiguousarray(a[sl]) for a in (w, oy, ox, gout))
        np.testing.assert_array_equal(oracle.forward_ex(inp, ws, oys, oxs, 1, origin=(y0, 0)), full[sl])
        gi, gw, ga, gb = oracle.backward_ex(gs, inp, ws, oys, oxs, 1, origin=(y0, 0))
        np.testing.assert_array_equal(gw, fgw[sl])
        np.testing.assert_array_equal(ga, fga[sl])
        np.testing.assert_array_equal(gb, fgb[sl])
        gi_sum += gi
    np.testing.assert_allclose(gi_sum, fgi, rtol=1e-5, atol=1e-6)


def test_oracle_ex_known_answer_far_origin(oracle):
    # origin far outside a 2x3 frame: every tap clamps to one corner texel
    frame = np.arange(6, dtype=np.float32).reshape(1, 1, 2, 3)
    w = np.full((1, 9, 2, 2), 1 / 9, np.float32)
    z = np.zeros_like(w)
    np.testing.assert_allclose(oracle.forward_ex(frame, w, z, z, 1, origin=(100, 100)), 5.0, rtol=1e-6)
    np.testing.assert_allclose(oracle.forward_ex(frame, w, z, z, 1, origin=(-100, -100)), 0.0, atol=0)
    np.testing.assert_allclose(oracle.forward_ex(frame, w, z, z, 1, origin=(-100, 100)), 2.0, rtol=1e-6)


# ---------------------------------------------------------------- HIP path (GPU)

@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import artemis_amd
    artemis_amd._lib.load()
    return torch.device("cuda:0")


def close(actual, desired, rtol, what="", floor=1e-30):
    """conftest.parity_close: norm-relative and elementwise-relative asserts, worst errors recorded."""
    parity_close(actual, desired, rtol, what, floor)


def term_size(frame, w, gout):
    return float(np.abs(frame).max() * np.abs(w).max() * np.abs(gout).max())


def run_frame(dev, fn, frame, w, oy, ox, gout, radius, deterministic=False):
    import artemis_amd
    t = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (frame, w, oy, ox)]
    with artemis_amd.synth_options(tile_radius=radius, deterministic=deterministic):
        out = fn(*t)
        out.backward(torch.from_numpy(gout).to(dev))
    torch.cuda.synchronize()
    return out.detach(), [x.grad for x in t]


@pytest.mark.gpu
@pytest.mark.parametrize("radius", [-1, 0, 2, 8])
@pytest.mark.parametrize("cfg", [(2, 3, 33, 70, 5, 1, "gauss1"), (1, 3, 40, 129, 5, 1, "clamp40"),
                                 (1, 3, 1, 1, 5, 1, "gauss4"), (1, 3, 64, 64, 5, 1, "uniform3"),
                                 (1, 2, 9, 20, 3, 2, "gauss4"), (1, 3, 17, 66, 7, 1, "uniform3")])
def test_unpadded_is_bitwise_pad_then_op(dev, oracle, cfg, radius):
    import artemis_amd
    N, C, H, W, k, d, regime = cfg
    frame, padded, w, oy, ox, gout, p = unpadded_operands(N, C, H, W, k, d, regime, seed=H * 3 + W)
    out, (gi, gw, ga, gb) = run_frame(
        dev, lambda f, a, b, c: artemis_amd.synth_unpadded(f, a, b, c, d), frame, w, oy, ox, gout, radius)
    pout, (pgi, pgw, pga, pgb) = run_frame(
        dev, lambda f, a, b, c: artemis_amd.FunctionSynth.apply(f, a, b, c, d), padded, w, oy, ox, gout,
        radius)
    # same kernels, same texels: identical bits to padding first
    assert torch.equal(out, pout)
    assert torch.equal(gw, pgw) and torch.equal(ga, pga) and torch.equal(gb, pgb)
    # and both within tolerance of the oracle
    close(out, oracle.forward_ex(frame, w, oy, ox, d, origin=(-p, -p)), FWD_RTOL, "output")
    rgi, rgw, rga, rgb = oracle.backward_ex(gout, frame, w, oy, ox, d, origin=(-p, -p))
    close(gw, rgw, BWD_RTOL, "grad_weight")
    close(ga, rga, BWD_RTOL, "grad_offset_y", term_size(frame, w, gout))
    close(gb, rgb, BWD_RTOL, "grad_offset_x", term_size(frame, w, gout))
    close(gi, rgi, BWD_RTOL, "grad_input")
    # gradient w.r.t. the unpadded frame = padding's adjoint applied to the padded gradient
    close(gi, fold_pad_grad(pgi.cpu().numpy(), p, H, W).astype(np.float32), BWD_RTOL, "fold")


@pytest.mark.gpu
@pytest.mark.parametrize("radius", [-1, 0, 8])
@pytest.mark.parametrize("padded_frame", [True, False])
def test_row_strips_reproduce_the_full_frame(dev, oracle, padded_frame, radius):
    """Strips reproduce the full frame bit for bit as long as a pixel's taps are computed the same
    way whatever tile it falls in: direct gathers (radius 0), or the first LDS window plus the
    per-pixel fix-up of a few stray taps (sigma 4 px at the default radius 14).  At radius 8 sigma 4
    leaves most pixels with taps outside the first window; whether a tile then stages a further
    window is decided per tile, so results agree to rounding only."""
    import artemis_amd
    from artemis_amd.sharding import row_strips
    H, W = 75, 131
    frame, padded, w, oy, ox, gout, p = unpadded_operands(1, 3, H, W, regime="gauss4", seed=77)
    src = padded if padded_frame else frame
    full_fn = (lambda f, a, b, c: artemis_amd.FunctionSynth.apply(f, a, b, c, 1)) if padded_frame \
        else (lambda f, a, b, c: artemis_amd.synth_unpadded(f, a, b, c, 1))
    out, (gi, gw, ga, gb) = run_frame(dev, full_fn, src, w, oy, ox, gout, radius, deterministic=True)
    gi_sum = torch.zeros_like(gi, dtype=torch.float64)
    # the host-side strip planner (3 ranks) or a ragged hand-made cut incl. a one-row strip
    strips = [(a, b) for a, b, _, _ in row_strips(H, 3, halo=0)] if radius == 0 \
        else [(0, 32), (32, 33), (33, 75)]
    for y0, y1 in strips:
        sl = np.s_[:, :, y0:y1]
        ws, oys, oxs, gs = (np.ascontiguousarray(a[sl]) for a in (w, oy, ox, gout))
        sout, (sgi, sgw, sga, sgb) = run_frame(
            dev, lambda f, a, b, c: artemis_amd.synth_strip(f, a, b, c, y0, 1, padded=padded_frame),
            src, ws, oys, oxs, gs, radius, deterministic=True)
        if radius == 8:
            for a, b, tol in ((sout, out[sl], FWD_RTOL), (sgw, gw[sl], BWD_RTOL), (sga, ga[sl], BWD_RTOL),
                              (sgb, gb[sl], BWD_RTOL)):
                assert float((a - b).abs().max()) <= tol * float(b.abs().max())
        else:
            assert torch.equal(sout, out[sl])
            assert torch.equal(sgw, gw[sl]) and torch.equal(sga, ga[sl]) and torch.equal(sgb, gb[sl])
        gi_sum += sgi.double()
    close(gi_sum.float(), gi.cpu().numpy(), BWD_RTOL, "strip grad_input sum")


@pytest.mark.gpu
def test_ex_arbitrary_frame_and_origin_against_oracle(dev, oracle):
    # frame smaller than the output, origin negative/positive, all radii paths
    import artemis_amd
    rng = np.random.default_rng(5)
    for (Hin, Win, oy0, ox0) in [(7, 9, -3, 5), (40, 100, 11, -17), (1, 1, 0, 0), (90, 64, -40, 70)]:
        _, w, oy, ox, gout = make_operands(1, 3, 37, 83, regime="gauss4", seed=Hin)
        frame = rng.random((1, 3, Hin, Win), dtype=np.float32)
        for radius in (-1, 0, 2):
            out, (gi, gw, ga, gb) = run_frame(
                dev, lambda f, a, b, c: artemis_amd.FunctionSynthFrame.apply(f, a, b, c, 1, oy0, ox0),
                frame, w, oy, ox, gout, radius)
            close(out, oracle.forward_ex(frame, w, oy, ox, 1, origin=(oy0, ox0)), FWD_RTOL, "output")
            rgi, rgw, rga, rgb = oracle.backward_ex(gout, frame, w, oy, ox, 1, origin=(oy0, ox0))
            close(gw, rgw, BWD_RTOL, "grad_weight")
            close(ga, rga, BWD_RTOL, "grad_offset_y", term_size(frame, w, gout))
            close(gb, rgb, BWD_RTOL, "grad_offset_x", term_size(frame, w, gout))
            # frames of 1 to 63 texels: every element is a sum of thousands of clamped contributions,
            # so the rounding of the sum (norm-relative, asserted) is far above 5 ulp of one term and
            # the elementwise criterion for ~100-term sums does not apply
            parity_close(gi, rgi, BWD_RTOL, "grad_input (tiny frames)", elementwise=False)


@pytest.mark.gpu
def test_full_size_unpadded_equals_padded(dev):
    # BASELINE.json headline shape: 1080p, bitwise equality of the fused-pad path
    import artemis_amd
    g = torch.Generator(device=dev).manual_seed(11)
    H, W = 1080, 1920
    frame = torch.rand(1, 3, H, W, device=dev, generator=g)
    w = torch.softmax(torch.randn(1, 25, H, W, device=dev, generator=g), 1)
    oy = torch.randn(1, 25, H, W, device=dev, generator=g) * 2
    ox = torch.randn(1, 25, H, W, device=dev, generator=g) * 2
    padded = torch.nn.functional.pad(frame, (2,) * 4, mode="replicate").contiguous()
    a = artemis_amd.synth_unpadded(frame, w, oy, ox, 1)
    b = artemis_amd.FunctionSynth.apply(padded, w, oy, ox, 1)
    assert torch.equal(a, b)
