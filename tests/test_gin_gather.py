"""gradInput in gather form (artemis_amd/csrc/synth_gin_gather.hpp): when every offset truncates to
zero -- a freshly initialised model -- gradInput is summed per texel in registers instead of scattered
through LDS atomics.  The gather kernel runs optimistically in front of the scatter kernels, checks
every offset it loads and gives up at the first that reaches 1, raising word 2 of the workspace
header; the scatter kernels leave at once unless it did.  These tests check the result against the
oracle's adjoint on every geometry the path accepts, that it is bit-reproducible and the same in both
gin modes (which only the gather form is), and that one large offset anywhere sends the call to the
scatter kernels (header word raised, result still the adjoint)."""
import numpy as np
import pytest
import torch

from conftest import make_operands, parity_close

pytestmark = pytest.mark.gpu

BWD_RTOL = 1e-4


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from artemis_amd import _lib
    return _lib.load()


ONE_BITS = 0x3f800000   # the bits of 1.0f: what the gather kernel raises header word 2 to when it gives up


def backward_ex(lib, inp, w, oy, ox, gout, origin, mode, radius=-1, taps=True, reset_backoff=True):
    """artemis_synth_backward_ex through the C ABI; returns gradInput (and the tap gradients)."""
    from artemis_amd import _lib
    dev = torch.device("cuda:0")
    t_in, t_w, t_oy, t_ox, t_g = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (inp, w, oy, ox, gout))
    N, C, Hin, Win = inp.shape
    H, W = w.shape[2:]
    gin = torch.full_like(t_in, float("nan"))
    gw, ga, gb = (torch.empty_like(t_w) for _ in range(3))
    nbytes = lib.artemis_synth_backward_workspace_bytes_ex(N, C, H, W, 5, 1, Hin, Win, mode)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    p = (lambda t: t.data_ptr()) if taps else (lambda t: None)
    if reset_backoff:   # (after a give-up the library skips the gather kernel for a number of calls)
        lib.artemis_synth_set_gin_gather(lib.artemis_synth_set_gin_gather(-1))
    _lib.check(lib.artemis_synth_backward_ex(
        t_in.data_ptr(), t_w.data_ptr(), t_oy.data_ptr(), t_ox.data_ptr(), t_g.data_ptr(), gin.data_ptr(),
        p(gw), p(ga), p(gb), N, C, H, W, 5, 1, Hin, Win, origin[0], origin[1], _lib.FRAME_F32, mode, radius,
        ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "backward_ex")
    torch.cuda.synchronize()
    gave_up = int(ws[:16].view(torch.int32)[2].item()) >= ONE_BITS if nbytes >= 16 else False
    return gin.cpu().numpy(), [x.cpu().numpy() for x in (gw, ga, gb)], gave_up


def small_offsets(shape, seed, bound=0.999):
    rng = np.random.default_rng(seed)
    return (rng.uniform(-bound, bound, shape).astype(np.float32),
            rng.uniform(-bound, bound, shape).astype(np.float32))


# (N, H, W, frame rows, frame columns, origin): the reference's padded layout, unpadded frames
# (pad folded into the clamp: two to three virtual rows / columns per side), a row strip in the
# middle of a frame and strips that touch its top / bottom, widths that are no multiple of 64
GEOMETRIES = [
    (1, 40, 72, 44, 76, (0, 0)),
    (2, 33, 100, 37, 104, (0, 0)),
    (1, 40, 72, 40, 72, (-2, -2)),
    (2, 70, 131, 70, 131, (-2, -2)),
    (1, 30, 90, 120, 90, (43, -2)),
    (1, 30, 90, 120, 90, (-2, -2)),
    (1, 30, 90, 120, 90, (88, -2)),
    (1, 64, 64, 68, 68, (0, 0)),
]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("geo", GEOMETRIES)
def test_gather_matches_the_adjoint(lib, oracle, geo, mode):
    N, H, W, Hin, Win, origin = geo
    rng = np.random.default_rng(7)
    inp = rng.random((N, 3, Hin, Win), dtype=np.float32)
    w = (rng.standard_normal((N, 25, H, W), dtype=np.float32) * 0.2).astype(np.float32)
    oy, ox = small_offsets(w.shape, 11)
    gout = rng.standard_normal((N, 3, H, W), dtype=np.float32)
    gin, (gw, ga, gb), gave_up = backward_ex(lib, inp, w, oy, ox, gout, origin, mode)
    assert not gave_up
    ref = oracle.backward_ex(gout, inp, w, oy, ox, 1, origin)
    parity_close(gin, ref[0], BWD_RTOL, "grad_input, gather")
    parity_close(gw, ref[1], BWD_RTOL, "grad_weight")
    parity_close(ga, ref[2], BWD_RTOL, "grad_offset_y")
    parity_close(gb, ref[3], BWD_RTOL, "grad_offset_x")
    again, _, _ = backward_ex(lib, inp, w, oy, ox, gout, origin, 1 - mode)
    assert np.array_equal(gin, again), "fixed summation order: same bits in either gin_mode, run to run"
    alone, _, _ = backward_ex(lib, inp, w, oy, ox, gout, origin, mode, taps=False)
    assert np.array_equal(gin, alone), "with or without the tap gradients in the same call"


def test_one_large_offset_sends_the_call_to_the_scatter_kernels(lib, oracle):
    N, H, W = 1, 40, 72
    inp, w, _, _, gout = make_operands(N, 3, H, W, regime="init", seed=5)
    for val in (1.0, -1.0, 1.5, -37.25):
        oy, ox = small_offsets(w.shape, 3)
        ox[0, 13, 17, 29] = val
        ref = oracle.backward_ex(gout, inp, w, oy, ox, 1, (0, 0))[0]
        for mode in (0, 1):
            gin, _, gave_up = backward_ex(lib, inp, w, oy, ox, gout, (0, 0), mode)
            assert gave_up, f"offset {val} does not truncate to zero"
            parity_close(gin, ref, BWD_RTOL, "grad_input, scatter")


def test_radius_zero_and_the_switch_keep_the_scatter_kernels(lib, oracle):
    N, H, W = 1, 40, 72
    inp, w, oy, ox, gout = make_operands(N, 3, H, W, regime="init", seed=9)
    ref = oracle.backward_ex(gout, inp, w, oy, ox, 1, (0, 0))[0]
    gin, _, gave_up = backward_ex(lib, inp, w, oy, ox, gout, (0, 0), 1, radius=0)     # the direct kernels
    assert not gave_up
    parity_close(gin, ref, BWD_RTOL, "grad_input, radius 0")
    assert lib.artemis_synth_set_gin_gather(0) == 1
    try:
        gin, _, gave_up = backward_ex(lib, inp, w, oy, ox, gout, (0, 0), 1)
    finally:
        assert lib.artemis_synth_set_gin_gather(1) == 0
    assert not gave_up
    parity_close(gin, ref, BWD_RTOL, "grad_input, gather off")
    gin, _, _ = backward_ex(lib, inp, w, oy, ox, gout, (0, 0), 1)
    parity_close(gin, ref, BWD_RTOL, "grad_input, gather on again")


def test_tiny_frames_and_far_origins_take_the_scatter_kernels(lib, oracle):
    for (H, W, Hin, Win, origin) in [(16, 16, 20, 20, (0, 0)), (40, 72, 40, 72, (-6, -2)), (40, 40, 44, 44, (0, 0))]:
        rng = np.random.default_rng(2)
        inp = rng.random((1, 3, Hin, Win), dtype=np.float32)
        w = rng.standard_normal((1, 25, H, W), dtype=np.float32)
        oy, ox = small_offsets(w.shape, 4, 0.5)
        gout = rng.standard_normal((1, 3, H, W), dtype=np.float32)
        gin, _, gave_up = backward_ex(lib, inp, w, oy, ox, gout, origin, 0)
        assert not gave_up     # (not tried)
        parity_close(gin, oracle.backward_ex(gout, inp, w, oy, ox, 1, origin)[0], BWD_RTOL, "grad_input")


def test_infinite_weight_outside_the_image_does_not_leak(lib, oracle):
    """Pixels outside the image are read at clamped addresses and dropped by selects: an Inf at the
    image's border pixel must only reach the texels that pixel really touches."""
    N, H, W = 1, 40, 72
    inp, w, oy, ox, gout = make_operands(N, 3, H, W, regime="init", seed=21)
    w[0, :, 0, 0] = np.inf
    w[0, :, H - 1, W - 1] = -np.inf
    gin, _, gave_up = backward_ex(lib, inp, w, oy, ox, gout, (0, 0), 0)
    assert not gave_up
    ref = oracle.backward_ex(gout, inp, w, oy, ox, 1, (0, 0))[0]
    assert np.array_equal(np.isfinite(gin), np.isfinite(ref))
    fin = np.isfinite(ref)
    parity_close(np.where(fin, gin, 0), np.where(fin, ref, 0), BWD_RTOL, "grad_input, finite part")


def test_large_offsets_in_one_corner_only(lib, oracle):
    """Most workgroups finish their tiles before (or without) seeing the flag: the scatter kernels
    must overwrite everything they stored."""
    N, H, W = 2, 96, 200
    inp, w, oy, ox, gout = make_operands(N, 3, H, W, regime="init", seed=41)
    oy[1, :, H - 3:, W - 5:] = 2.75
    ref = oracle.backward_ex(gout, inp, w, oy, ox, 1, (0, 0))[0]
    for mode in (0, 1):
        gin, _, gave_up = backward_ex(lib, inp, w, oy, ox, gout, (0, 0), mode)
        assert gave_up
        parity_close(gin, ref, BWD_RTOL, "grad_input, scatter after a late give-up")


@pytest.mark.adaptive
def test_give_ups_make_the_library_skip_the_kernel_for_a_while(lib, oracle):
    """Once a forward call with the adaptive radius has set up the device's pinned record, the gather
    kernel's give-ups reach the host, and the calls that follow (1, then 2, 4 ...) do not enqueue it:
    header word 2 stays 0 although the offsets are large.  Results are the adjoint either way."""
    import artemis_amd
    from artemis_amd import _lib
    dev = torch.device("cuda:0")
    N, H, W = 1, 40, 72
    inp, w, oy, ox, gout = make_operands(N, 3, H, W, regime="gauss1", seed=13)
    ref = oracle.backward_ex(gout, inp, w, oy, ox, 1, (0, 0))[0]
    t = [torch.from_numpy(a).to(dev) for a in (inp, w, oy, ox)]
    with artemis_amd.synth_options(tile_radius=_lib.RADIUS_AUTO):
        artemis_amd.FunctionSynth.apply(*t, 1)        # sets the record up
    torch.cuda.synchronize()
    # a give-up whose report nobody has looked at yet must not count after the reset below
    # (whatever ran before this test: the reset forgets reports of attempts already issued)
    lib.artemis_synth_set_gin_gather(1)
    assert backward_ex(lib, inp, w, oy, ox, gout, (0, 0), 1, reset_backoff=False)[2]
    lib.artemis_synth_set_gin_gather(1)               # forget earlier back-off and pending reports
    tried = []
    for _ in range(8):
        gin, _, gave_up = backward_ex(lib, inp, w, oy, ox, gout, (0, 0), 1, reset_backoff=False)
        parity_close(gin, ref, BWD_RTOL, "grad_input")
        tried.append(gave_up)
    # call 0 tries and gives up; its report is in by call 1 (backward_ex synchronises), which is
    # skipped; call 2 tries again, calls 3-4 are skipped, call 5 tries, 6-7 (of 4) are skipped
    assert tried == [True, False, True, False, False, True, False, False], tried
    lib.artemis_synth_set_gin_gather(1)


def test_operator_level_random_init_model(lib, oracle):
    import artemis_amd
    dev = torch.device("cuda:0")
    inp, w, oy, ox, gout = make_operands(2, 3, 48, 80, regime="init", seed=33, weights="raw")
    t = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (inp, w, oy, ox)]
    out = artemis_amd.FunctionSynth.apply(*t, 1)
    out.backward(torch.from_numpy(gout).to(dev))
    torch.cuda.synchronize()
    parity_close(t[0].grad, oracle.grad_input(gout, inp.shape, w, oy, ox, 1), BWD_RTOL, "grad_input")
