"""The library only enqueues kernels and memsets on the stream it is given, so its calls can be
captured into a HIP graph (torch.cuda.CUDAGraph) and replayed: the small heads of a training step
are launch-bound, and a replayed graph removes the per-launch host cost."""
import pytest
import torch

from test_head import head_operands

pytestmark = pytest.mark.gpu


def test_head_forward_backward_replays_from_a_hip_graph():
    import ctypes

    from artemis_amd import _lib
    dev = torch.device("cuda:0")
    lib = _lib.load()
    B, T, H, W = 2, 4, 48, 80
    frames, w, a, b, occ, gout = head_operands(B, T, H, W, "gauss1", seed=3)
    tf = [torch.from_numpy(f).to(dev) for f in frames]
    tw, ta, tb, tocc, tg = (torch.from_numpy(x).to(dev) for x in (w, a, b, occ, gout))
    out = torch.empty(B, 3, H, W, device=dev)
    gw, ga, gb = (torch.empty_like(tw) for _ in range(3))
    gocc = torch.empty_like(tocc)
    ptrs = (ctypes.c_void_p * T)(*[f.data_ptr() for f in tf])

    def launch(stream):
        _lib.check(lib.artemis_synth_head_forward(
            ptrs, T, tw.data_ptr(), ta.data_ptr(), tb.data_ptr(), tocc.data_ptr(), out.data_ptr(),
            B, H, W, H, W, -2, -2, 0, -1, stream), "fwd")
        _lib.check(lib.artemis_synth_head_backward(
            ptrs, T, tw.data_ptr(), ta.data_ptr(), tb.data_ptr(), tocc.data_ptr(), tg.data_ptr(),
            gw.data_ptr(), ga.data_ptr(), gb.data_ptr(), gocc.data_ptr(), B, H, W, H, W, -2, -2,
            0, -1, stream), "bwd")

    # eager reference
    launch(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = [t.clone() for t in (out, gw, ga, gb, gocc)]

    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launch(side.cuda_stream)          # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        launch(torch.cuda.current_stream().cuda_stream)
    # new operand values in place, then replay
    tocc.mul_(0.5)
    for t in (out, gw, ga, gb, gocc):
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want[0] * 0.5) or torch.allclose(out, want[0] * 0.5, rtol=1e-6, atol=1e-7)
    assert torch.allclose(gw, want[1] * 0.5, rtol=1e-6, atol=1e-7)
    assert torch.equal(gocc, want[4])      # d out / d occ does not depend on occ
    # back to the original values: bit-identical to eager
    tocc.mul_(2.0)
    graph.replay()
    torch.cuda.synchronize()
    for got, ref in zip((out, gw, ga, gb, gocc), want):
        assert torch.equal(got, ref)


@pytest.mark.adaptive
def test_operator_under_capture_with_the_adaptive_default_uses_the_fixed_radius(oracle):
    """options.tile_radius is adaptive by default; a captured launch keeps the shape it was captured
    with and the first adaptive call on a device allocates, so under capture the wrapper passes the
    fixed default: the replay is bit-identical to an eager call at radius -1."""
    import artemis_amd
    from artemis_amd import _lib
    from conftest import make_operands, parity_close
    assert artemis_amd.options.tile_radius == _lib.RADIUS_AUTO
    dev = torch.device("cuda:0")
    inp, w, oy, ox, _ = make_operands(1, 3, 40, 72, regime="gauss1", seed=11)
    t = [torch.from_numpy(x).to(dev) for x in (inp, w, oy, ox)]
    with artemis_amd.synth_options(tile_radius=_lib.RADIUS_DEFAULT):
        want = artemis_amd.FunctionSynth.apply(*t, 1)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = artemis_amd.FunctionSynth.apply(*t, 1)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    parity_close(out, oracle.forward(inp, w, oy, ox, 1), 1e-5, "output, captured")
