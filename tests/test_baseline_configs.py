"""BASELINE.json configs[1], [2], [3], [4] at their STATED sizes on the GPU against the CPU oracle.

* configs[1] "Synth op forward only, Vimeo-90K septuplet shape 448x256, batch 16, 4 frames": one
  forward launch of N = 16 x 256 x 448 (operand shapes of model/chrono_synth.py:109-127 at the
  dataset's native size, data/preprocessing/vimeo90k_septuplet_process.py:36) at the default
  radius -- 896 big tiles, i.e. the 128 x 16 kernels with N in the tile index -- per context frame;
  samples 0 and 15 in full against the oracle, every sample against its own N = 1 launch with the
  small-grid knob forced to 0 (the same kernels on a grid of 56 tiles), bit for bit; the same for
  the fused head.

* configs[3] "Interpolate mode at 1080p, one 4-frame window, 7 timesteps batched as one launch":
  the op and the fused head at N = 7 x 1080 x 1920 in one launch (main.py:114-121 runs the
  timesteps one by one; batching them is this build's addition).  Samples 0 and 6
  in full against the oracle; all 7 against their own N = 1 launches, bit for bit.
* configs[2] "train step on 256x256 crops, batch 64": the three heads (64 | 8, 25, {64,128,256}^2)
  forward + backward (model/chrono_synth.py:109-127 shapes; 8 = one rank's shard of 64 on 8 GPUs),
  incl. the small-grid 64 x 8 tile path; a sample subset against oracle.backward.
* configs[4] "4K with bf16 frames, spatial tiles sharded over 8 GPUs": 270-row bf16 strips of the
  3840 x 2160 frame against oracle.forward_ex / backward_ex on the rounded frame.

Operands are generated on the device (seeded) and only the compared samples travel to the host.
"""
import numpy as np
import pytest
import torch

from conftest import parity_close

pytestmark = pytest.mark.gpu

FWD_RTOL = 1e-5
BWD_RTOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import artemis_amd
    artemis_amd._lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def fast_oracle(oracle):
    import os
    oracle.set_threads(min(16, os.cpu_count() or 1))
    yield oracle
    oracle.set_threads(1)


def np_(t):
    return np.ascontiguousarray(t.detach().float().cpu().numpy())


# ------------------------------------------------------------------ configs[1]: N = 16 x 256 x 448

def _big_tiles_only(lib):
    """Context: small-grid knob 0 (every launch takes the 128 x 16 tile kernels)."""
    import contextlib

    @contextlib.contextmanager
    def cm():
        old = lib.artemis_synth_set_small_grid_tiles(-1)
        lib.artemis_synth_set_small_grid_tiles(0)
        try:
            yield old
        finally:
            lib.artemis_synth_set_small_grid_tiles(old)
    return cm()


@pytest.mark.parametrize("sigma,radius", [(1.0, -1), (0.045, -1), (1.0, -3)])
def test_config1_op_forward_batch16_vimeo_shape(dev, fast_oracle, sigma, radius):
    """radius -1: the fixed default (two windows per tile); -3: the lean shape the adaptive default settles
    on for offsets like these."""
    import artemis_amd
    from artemis_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(16)
    N, H, W, T = 16, 256, 448, 4
    knob = lib.artemis_synth_set_small_grid_tiles(-1)
    # the launch must be on the big-tile side of the forward's switch (a quarter of the knob)
    assert N * ((H + 31) // 32) * ((W + 63) // 64) >= (knob + 3) // 4      # the library's count: 64 x 32 units
    for f in range(T):                                   # the 4 context frames = the 4 op calls
        inp = torch.nn.functional.pad(torch.rand(N, 3, H, W, device=dev, generator=g), (2,) * 4,
                                      mode="replicate").contiguous()
        w = torch.softmax(torch.randn(N, 25, H, W, device=dev, generator=g), 1)
        oy = torch.randn(N, 25, H, W, device=dev, generator=g) * sigma
        ox = torch.randn(N, 25, H, W, device=dev, generator=g) * sigma
        with artemis_amd.synth_options(tile_radius=radius):
            out = artemis_amd.synth(inp, w, oy, ox, 1)      # ONE launch
            torch.cuda.synchronize()
            assert out.shape == (N, 3, H, W)
            with _big_tiles_only(lib):
                for n in range(N):
                    one = artemis_amd.synth(inp[n:n + 1].contiguous(), w[n:n + 1].contiguous(),
                                            oy[n:n + 1].contiguous(), ox[n:n + 1].contiguous(), 1)
                    assert torch.equal(out[n:n + 1], one), f"frame {f} sample {n}"
        for n in (0, 15):
            ref = fast_oracle.forward(np_(inp[n:n + 1]), np_(w[n:n + 1]), np_(oy[n:n + 1]),
                                      np_(ox[n:n + 1]), 1)
            parity_close(out[n:n + 1], ref, FWD_RTOL, f"config1 op output N16 256x448 sigma {sigma} R{radius}")


def test_config1_head_forward_batch16_vimeo_shape(dev, fast_oracle):
    import artemis_amd
    from artemis_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(17)
    B, T, H, W = 16, 4, 256, 448
    frames = [torch.rand(B, 3, H, W, device=dev, generator=g) for _ in range(T)]
    w = torch.softmax(torch.randn(B, T, 25, H, W, device=dev, generator=g), 2)
    a = torch.randn(B, T, 25, H, W, device=dev, generator=g)
    b = torch.randn(B, T, 25, H, W, device=dev, generator=g)
    occ = torch.softmax(torch.randn(B, T, H, W, device=dev, generator=g), 1)
    out = artemis_amd.chrono_head(frames, w, a, b, occ)  # ONE launch: 16 samples x 4 frames
    torch.cuda.synchronize()
    with _big_tiles_only(lib):
        for n in range(B):
            one = artemis_amd.chrono_head([f[n:n + 1].contiguous() for f in frames],
                                          w[n:n + 1].contiguous(), a[n:n + 1].contiguous(),
                                          b[n:n + 1].contiguous(), occ[n:n + 1].contiguous())
            assert torch.equal(out[n:n + 1], one), f"sample {n}"
    for n in (0, 15):
        ref, _ = fast_oracle.head_forward([np_(f[n:n + 1]) for f in frames], np_(w[n:n + 1]),
                                          np_(a[n:n + 1]), np_(b[n:n + 1]), np_(occ[n:n + 1]), 1)
        parity_close(out[n:n + 1], ref, FWD_RTOL, f"config1 head output B16 256x448")


# ------------------------------------------------------------------ configs[3]: N = 7 x 1080p

def test_config3_op_seven_timesteps_one_launch(dev, fast_oracle):
    import artemis_amd
    g = torch.Generator(device=dev).manual_seed(3)
    N, H, W = 7, 1080, 1920
    # one 4-frame window: the same frame for every timestep, different weights/offsets per timestep
    frame = torch.rand(1, 3, H, W, device=dev, generator=g)
    inp = torch.nn.functional.pad(frame, (2,) * 4, mode="replicate").expand(N, 3, H + 4, W + 4).contiguous()
    w = torch.softmax(torch.randn(N, 25, H, W, device=dev, generator=g), 1)
    oy = torch.randn(N, 25, H, W, device=dev, generator=g) * 1.5
    ox = torch.randn(N, 25, H, W, device=dev, generator=g) * 1.5
    out = artemis_amd.synth(inp, w, oy, ox, 1)          # ONE launch for the 7 timesteps
    torch.cuda.synchronize()
    assert out.shape == (N, 3, H, W)
    for n in range(N):                                   # every sample: the bits of its own launch
        one = artemis_amd.synth(inp[n:n + 1], w[n:n + 1].contiguous(), oy[n:n + 1].contiguous(),
                                ox[n:n + 1].contiguous(), 1)
        assert torch.equal(out[n:n + 1], one), f"timestep {n}"
    for n in (0, 6):                                     # first and last in full against the oracle
        ref = fast_oracle.forward(np_(inp[n:n + 1]), np_(w[n:n + 1]), np_(oy[n:n + 1]),
                                  np_(ox[n:n + 1]), 1)
        parity_close(out[n:n + 1], ref, FWD_RTOL, f"config3 op output, timestep {n}")


def test_config3_head_seven_timesteps_one_launch(dev, fast_oracle):
    import artemis_amd
    g = torch.Generator(device=dev).manual_seed(4)
    B, T, H, W = 7, 4, 1080, 1920
    frames1 = [torch.rand(1, 3, H, W, device=dev, generator=g) for _ in range(T)]
    frames = [f.expand(B, 3, H, W).contiguous() for f in frames1]
    w = torch.softmax(torch.randn(B, T, 25, H, W, device=dev, generator=g), 2)
    a = torch.randn(B, T, 25, H, W, device=dev, generator=g)
    b = torch.randn(B, T, 25, H, W, device=dev, generator=g)
    occ = torch.softmax(torch.randn(B, T, H, W, device=dev, generator=g), 1)
    out = artemis_amd.chrono_head(frames, w, a, b, occ)  # ONE launch: 7 timesteps x 4 frames
    torch.cuda.synchronize()
    for n in range(B):
        one = artemis_amd.chrono_head(frames1, w[n:n + 1].contiguous(), a[n:n + 1].contiguous(),
                                      b[n:n + 1].contiguous(), occ[n:n + 1].contiguous())
        assert torch.equal(out[n:n + 1], one), f"timestep {n}"
    f_np = [np_(f) for f in frames1]
    for n in (0, 6):
        ref, _ = fast_oracle.head_forward(f_np, np_(w[n:n + 1]), np_(a[n:n + 1]), np_(b[n:n + 1]),
                                          np_(occ[n:n + 1]), 1)
        parity_close(out[n:n + 1], ref, FWD_RTOL, f"config3 head output, timestep {n}")


# ------------------------------------------------------------------ configs[2]: batch-64 train heads

def head_case(dev, B, hw, seed, sigma):
    g = torch.Generator(device=dev).manual_seed(seed)
    T = 4
    frames = [torch.rand(B, 3, hw, hw, device=dev, generator=g) for _ in range(T)]
    w = torch.softmax(torch.randn(B, T, 25, hw, hw, device=dev, generator=g), 2)
    a = torch.randn(B, T, 25, hw, hw, device=dev, generator=g) * sigma
    b = torch.randn(B, T, 25, hw, hw, device=dev, generator=g) * sigma
    occ = torch.softmax(torch.randn(B, T, hw, hw, device=dev, generator=g), 1)
    gout = torch.randn(B, 3, hw, hw, device=dev, generator=g)
    return frames, w, a, b, occ, gout


@pytest.mark.parametrize("B,hw,samples", [(8, 64, range(8)), (8, 128, (0, 3, 7)), (8, 256, (0, 7)),
                                          (64, 64, (0, 31, 63)), (64, 128, (63,)), (64, 256, (17,))])
def test_config2_train_heads_forward_backward(dev, fast_oracle, B, hw, samples):
    """The fused head (what artemis_amd.model.ChronoSynth runs in the train step) and, for the
    same operands, the per-frame op of frame 1, forward + backward, against the oracle."""
    import artemis_amd
    frames, w, a, b, occ, gout = head_case(dev, B, hw, seed=B * 1000 + hw, sigma=1.0)
    t = [x.clone().requires_grad_(True) for x in (w, a, b, occ)]
    out = artemis_amd.chrono_head(frames, *t)
    out.backward(gout)
    # per-frame op (FunctionSynth on the padded frame), as the reference composes it
    f = 1
    padded = torch.nn.functional.pad(frames[f], (2,) * 4, mode="replicate").requires_grad_(True)
    opw = [x[:, f].contiguous().requires_grad_(True) for x in (w, a, b)]
    oout = artemis_amd.FunctionSynth.apply(padded, *opw, 1)
    oout.backward(gout)
    torch.cuda.synchronize()
    for n in samples:
        sl = slice(n, n + 1)
        f_np = [np_(x[sl]) for x in frames]
        args = (np_(w[sl]), np_(a[sl]), np_(b[sl]), np_(occ[sl]))
        ref, _ = fast_oracle.head_forward(f_np, *args, 1)
        parity_close(out[sl], ref, FWD_RTOL, f"config2 head output B{B} {hw}^2")
        rgw, rga, rgb, rgocc = fast_oracle.head_backward(np_(gout[sl]), f_np, *args, 1)
        term = float(gout[sl].abs().max() * w[sl].abs().max())
        parity_close(t[0].grad[sl], rgw, BWD_RTOL, "config2 head grad_weights")
        parity_close(t[1].grad[sl], rga, BWD_RTOL, "config2 head grad_alphas", term)
        parity_close(t[2].grad[sl], rgb, BWD_RTOL, "config2 head grad_betas", term)
        parity_close(t[3].grad[sl], rgocc, BWD_RTOL, "config2 head grad_occlusion")
        # the op
        pin = np_(padded[sl])
        wa = [np_(x[sl]) for x in opw]
        parity_close(oout[sl], fast_oracle.forward(pin, *wa, 1), FWD_RTOL, f"config2 op output B{B} {hw}^2")
        rgi, ogw, oga, ogb = fast_oracle.backward(np_(gout[sl]), pin, *wa, 1)
        parity_close(opw[0].grad[sl], ogw, BWD_RTOL, "config2 op grad_weight")
        parity_close(opw[1].grad[sl], oga, BWD_RTOL, "config2 op grad_offset_y", term)
        parity_close(opw[2].grad[sl], ogb, BWD_RTOL, "config2 op grad_offset_x", term)
        parity_close(padded.grad[sl], rgi, BWD_RTOL, "config2 op grad_input")


def test_config2_small_grid_path_is_taken_and_bit_identical(dev):
    """B = 8 at 64^2 is 16 tiles of 64 x 32: far below the small-grid knob, so the 64 x 8 tile
    kernels run; forcing the big tiles (knob 0) must give the same bits."""
    import artemis_amd
    from artemis_amd import _lib
    lib = _lib.load()
    frames, w, a, b, occ, gout = head_case(dev, 8, 64, seed=77, sigma=1.0)
    res = []
    old = lib.artemis_synth_set_small_grid_tiles(-1)
    try:
        for knob in (old, 0):
            lib.artemis_synth_set_small_grid_tiles(knob)
            t = [x.clone().requires_grad_(True) for x in (w, a, b, occ)]
            out = artemis_amd.chrono_head(frames, *t)
            out.backward(gout)
            torch.cuda.synchronize()
            res.append([out.detach()] + [x.grad for x in t])
    finally:
        lib.artemis_synth_set_small_grid_tiles(old)
    assert old > 16 * 4
    for x, y in zip(*res):
        assert torch.equal(x, y)


# ------------------------------------------------------------------ configs[4]: 4K bf16 strips

@pytest.mark.parametrize("strip", [0, 3, 7])
def test_config4_4k_bf16_strip_against_oracle(dev, fast_oracle, strip):
    import artemis_amd
    from artemis_amd.sharding import row_strips
    g = torch.Generator(device=dev).manual_seed(40 + strip)
    H, W = 2160, 3840
    y0, y1 = row_strips(H, 8, halo=0)[strip][:2]
    assert y1 - y0 == 270
    frame = torch.rand(1, 3, H, W, device=dev, generator=g).to(torch.bfloat16)
    h = y1 - y0
    w = torch.softmax(torch.randn(1, 25, h, W, device=dev, generator=g), 1).requires_grad_(True)
    oy = (torch.randn(1, 25, h, W, device=dev, generator=g) * 2).requires_grad_(True)
    ox = (torch.randn(1, 25, h, W, device=dev, generator=g) * 2).requires_grad_(True)
    gout = torch.randn(1, 3, h, W, device=dev, generator=g)
    out = artemis_amd.synth_strip(frame, w, oy, ox, y0, 1, padded=False)
    out.backward(gout)
    torch.cuda.synchronize()
    assert out.dtype == torch.float32
    rounded = np_(frame)                                  # the bf16 frame widened exactly
    org = (y0 - 2, -2)
    ref = fast_oracle.forward_ex(rounded, np_(w), np_(oy), np_(ox), 1, origin=org)
    parity_close(out, ref, FWD_RTOL, f"config4 strip output")
    _, rgw, rga, rgb = fast_oracle.backward_ex(np_(gout), rounded, np_(w), np_(oy), np_(ox), 1,
                                               origin=org, with_input=False)
    term = float(gout.abs().max() * w.abs().max())
    parity_close(w.grad, rgw, BWD_RTOL, "config4 strip grad_weight")
    parity_close(oy.grad, rga, BWD_RTOL, "config4 strip grad_offset_y", term)
    parity_close(ox.grad, rgb, BWD_RTOL, "config4 strip grad_offset_x", term)


# ------------------------------------------------------------------ the bench's other offset regimes
@pytest.mark.parametrize("regime,radius", [("smooth30", -1), ("sigma:4", -1), ("pan", -1), ("smooth30", 24),
                                           ("sigma:8", 24), ("sigma:1", -3), ("pan", -3), ("sigma:4", -3)])
def test_bench_regimes_at_1080p_against_oracle(dev, fast_oracle, regime, radius):
    """The offset regimes bench.py times beside the headline (config.other_offset_regimes), one
    1080p frame each at the default radius (and the steep / scattered ones at radius 24), forward and
    backward against the oracle: smooth30 is where tiles stage two and three windows."""
    import os
    import sys
    import artemis_amd
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    f = bench.make_frames(1, 1.0, dev, 11, regime=regime)[0]
    t = [f["inp"]] + [f[k].clone().requires_grad_(True) for k in ("w", "oy", "ox")]
    with artemis_amd.synth_options(tile_radius=radius):
        out = artemis_amd.FunctionSynth.apply(*t, 1)
        out.backward(f["gout"])
    torch.cuda.synchronize()
    regime = f"{regime} R{radius}"
    inp, w, oy, ox, gout = (np_(f[k]) for k in ("inp", "w", "oy", "ox", "gout"))
    parity_close(out, fast_oracle.forward(inp, w, oy, ox, 1), FWD_RTOL, f"{regime} 1080p output")
    _, rgw, rga, rgb = fast_oracle.backward(gout, inp, w, oy, ox, 1, with_input=False)
    term = float(np.abs(gout).max() * np.abs(w).max() * np.abs(inp).max())
    parity_close(t[1].grad, rgw, BWD_RTOL, f"{regime} 1080p grad_weight")
    parity_close(t[2].grad, rga, BWD_RTOL, f"{regime} 1080p grad_offset_y", floor=term)
    parity_close(t[3].grad, rgb, BWD_RTOL, f"{regime} 1080p grad_offset_x", floor=term)
