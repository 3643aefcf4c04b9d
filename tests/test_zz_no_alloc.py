"""artemis_synth_backward_noalloc -- the allocation-free variant of artemis_synth_backward_ex (its
workspace is a device array inside the library) -- against artemis_synth_backward_ex itself,
through the C ABI.  Same kernels, so: tap gradients bit-equal; gradInput bit-equal in
deterministic mode, within the backward tolerance (1e-4 of the tensor's scale; fp32 atomics
arrive in any order) in atomic mode.  Two calls in a row on one stream, the second with a smaller
input: the array keeps the first call's contents and the second must not see them.

Collected last on purpose (file name): the round's end runs `pytest -x`.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import make_operands

pytestmark = pytest.mark.gpu

BWD_RTOL = 1e-4
SCRATCH_BYTES = 64 << 20


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _operands(dev, N, H, W, seed, regime="gauss1"):
    inp, w, oy, ox, gout = make_operands(N, 3, H, W, 5, 1, regime, seed=seed)
    return [torch.from_numpy(a).to(dev) for a in (inp, w, oy, ox, gout)]


def _outputs(t, poison):
    """gradInput, gradWeight, gradOffset_y, gradOffset_x buffers pre-filled with a poison value:
    an entry point that skipped part of its output would show."""
    inp, w = t[0], t[1]
    return [torch.full_like(inp, poison)] + [torch.full_like(w, poison) for _ in range(3)]


def _call_ex(lib, t, outs, mode, radius, st):
    from artemis_amd import _lib
    inp, w, oy, ox, gout = t
    N, C, Hin, Win = inp.shape
    H, W = w.shape[2:]
    nbytes = lib.artemis_synth_backward_workspace_bytes_ex(N, C, H, W, 5, 1, Hin, Win, mode)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=inp.device)
    _lib.check(lib.artemis_synth_backward_ex(
        inp.data_ptr(), w.data_ptr(), oy.data_ptr(), ox.data_ptr(), gout.data_ptr(),
        *[o.data_ptr() for o in outs], N, C, H, W, 5, 1, Hin, Win, 0, 0, _lib.FRAME_F32, mode,
        radius, ws.data_ptr(), nbytes, st), "artemis_synth_backward_ex")
    torch.cuda.synchronize()
    return ws


def _call_noalloc(lib, t, outs, mode, radius, st):
    from artemis_amd import _lib
    inp, w, oy, ox, gout = t
    N, C, Hin, Win = inp.shape
    H, W = w.shape[2:]
    _lib.check(lib.artemis_synth_backward_noalloc(
        inp.data_ptr(), w.data_ptr(), oy.data_ptr(), ox.data_ptr(), gout.data_ptr(),
        *[o.data_ptr() for o in outs], N, C, H, W, 5, 1, Hin, Win, 0, 0, _lib.FRAME_F32, mode,
        radius, st), "artemis_synth_backward_noalloc")


def _close(a, b, rtol, what):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * float(np.abs(b).max()), err_msg=what)


@pytest.mark.parametrize("radius", [-1, 0])
def test_noalloc_matches_backward_ex_two_calls_in_a_row(dev, radius):
    from artemis_amd import _lib
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # first a larger problem (fills much of the scratch), then a smaller one right behind it on
    # the same stream, with no synchronisation in between
    big = _operands(dev, 2, 96, 200, seed=31, regime="gauss4")
    small = _operands(dev, 1, 37, 70, seed=32)
    for mode in (_lib.GIN_DETERMINISTIC, _lib.GIN_ATOMIC):
        want = []
        for t in (big, small):
            outs = _outputs(t, 7.0)
            _call_ex(lib, t, outs, mode, radius, st)
            want.append(outs)
        got = [_outputs(big, -3.0), _outputs(small, -3.0)]
        _call_noalloc(lib, big, got[0], mode, radius, st)
        _call_noalloc(lib, small, got[1], mode, radius, st)
        torch.cuda.synchronize()
        for which, g, w_ in zip(("first", "second (smaller)"), got, want):
            for name, a, b in zip(("grad_input", "grad_weight", "grad_offset_y", "grad_offset_x"),
                                  g, w_):
                if name == "grad_input" and mode == _lib.GIN_ATOMIC:
                    _close(a, b, BWD_RTOL, f"{which} call, {name}, atomic")
                else:
                    assert torch.equal(a, b), f"{which} call, {name}, mode {mode}: bits differ"


def test_noalloc_allocates_nothing_and_skips_scratch_without_grad_input(dev):
    from artemis_amd import _lib
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    t = _operands(dev, 1, 40, 130, seed=33)
    outs = _outputs(t, 1.0)
    _call_noalloc(lib, t, outs, _lib.GIN_DETERMINISTIC, -1, st)   # resolves the scratch address
    _call_noalloc(lib, t, outs, _lib.GIN_ATOMIC, -1, st)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    alloc0 = torch.cuda.memory_allocated()
    for _ in range(5):
        _call_noalloc(lib, t, outs, _lib.GIN_DETERMINISTIC, -1, st)
        _call_noalloc(lib, t, outs, _lib.GIN_ATOMIC, -1, st)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert torch.cuda.memory_allocated() == alloc0
    assert free1 == free0, "device memory in use changed across allocation-free calls"
    # tap gradients only: same result with no grad_input requested (scratch untouched)
    ref = _outputs(t, 2.0)
    _call_ex(lib, t, ref, _lib.GIN_ATOMIC, -1, st)
    only = _outputs(t, 5.0)
    inp, w, oy, ox, gout = t
    N, C, Hin, Win = inp.shape
    H, W = w.shape[2:]
    _lib.check(lib.artemis_synth_backward_noalloc(
        inp.data_ptr(), w.data_ptr(), oy.data_ptr(), ox.data_ptr(), gout.data_ptr(), None,
        only[1].data_ptr(), only[2].data_ptr(), only[3].data_ptr(), N, C, H, W, 5, 1, Hin, Win,
        0, 0, _lib.FRAME_F32, _lib.GIN_ATOMIC, -1, st), "noalloc, taps only")
    torch.cuda.synchronize()
    for a, b in zip(only[1:], ref[1:]):
        assert torch.equal(a, b)
    assert torch.all(only[0] == 5.0)


def test_noalloc_refuses_oversize_with_a_clear_error(dev):
    from artemis_amd import _lib
    lib = _lib.load()
    # deterministic grad_input of a 2-sample 1080p batch needs 100 MB: beyond the 64 MiB scratch;
    # refused on the host before any pointer is touched (the pointers here are never dereferenced)
    N, H, W = 2, 1080, 1920
    need = lib.artemis_synth_backward_workspace_bytes_ex(N, 3, H, W, 5, 1, H + 4, W + 4,
                                                         _lib.GIN_DETERMINISTIC)
    assert need > SCRATCH_BYTES
    rc = lib.artemis_synth_backward_noalloc(16, 16, 16, 16, 16, 16, None, None, None, N, 3, H, W,
                                            5, 1, H + 4, W + 4, 0, 0, _lib.FRAME_F32,
                                            _lib.GIN_DETERMINISTIC, -1, None)
    assert rc == 1
    msg = lib.artemis_synth_last_error().decode()
    assert "scratch" in msg and str(need) in msg
    # the same problem in atomic mode (256-byte header) and one 1080p frame in deterministic mode fit
    assert lib.artemis_synth_backward_workspace_bytes_ex(
        1, 3, H, W, 5, 1, H + 4, W + 4, _lib.GIN_DETERMINISTIC) <= SCRATCH_BYTES


def test_noalloc_full_1080p_frame_deterministic(dev):
    """The largest frame the header names (one padded 1080p frame, 50 MB of scratch) against the
    existing entry point, bit for bit, then a small call behind it."""
    from artemis_amd import _lib
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(5)
    N, H, W = 1, 1080, 1920
    inp = torch.rand(N, 3, H + 4, W + 4, device=dev)
    w = torch.softmax(torch.randn(N, 25, H, W, device=dev), 1)
    oy = torch.randn(N, 25, H, W, device=dev) * 2
    ox = torch.randn(N, 25, H, W, device=dev) * 2
    gout = torch.randn(N, 3, H, W, device=dev)
    t = [inp, w, oy, ox, gout]
    want = _outputs(t, 7.0)
    _call_ex(lib, t, want, _lib.GIN_DETERMINISTIC, -1, st)
    small = _operands(dev, 1, 20, 66, seed=34)
    want_small = _outputs(small, 7.0)
    _call_ex(lib, small, want_small, _lib.GIN_DETERMINISTIC, -1, st)
    got, got_small = _outputs(t, -3.0), _outputs(small, -3.0)
    _call_noalloc(lib, t, got, _lib.GIN_DETERMINISTIC, -1, st)
    _call_noalloc(lib, small, got_small, _lib.GIN_DETERMINISTIC, -1, st)
    torch.cuda.synchronize()
    for a, b in zip(got + got_small, want + want_small):
        assert torch.equal(a, b)
