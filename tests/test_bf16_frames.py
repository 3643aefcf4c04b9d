"""bf16 frames / fp32 accumulate (BASELINE.json configs[4]).  The reference is fp32 only
(cupy_module/synth.py:9-13); a bf16 texel is widened exactly to fp32 when it is staged, so the
expected result is the oracle run on the frame rounded to bf16 -- same tolerances as fp32."""
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


def to_bf16(a):
    t = torch.from_numpy(a).to(torch.bfloat16)
    return t, t.float().numpy()


@pytest.mark.parametrize("radius", [-1, 0, 2, 8])
@pytest.mark.parametrize("cfg", [(2, 33, 70, "gauss1"), (1, 40, 129, "gauss4"), (1, 1, 1, "gauss1"),
                                 (1, 64, 64, "clamp40")])
def test_bf16_frame_op_against_oracle(dev, oracle, cfg, radius):
    import artemis_amd
    N, H, W, regime = cfg
    inp, w, oy, ox, gout = make_operands(N, 3, H, W, 5, 1, regime, seed=H + W)
    frame = np.ascontiguousarray(inp[:, :, 2:2 + H, 2:2 + W])
    fb, fr = to_bf16(frame)
    t = [fb.to(dev).requires_grad_(True)] + [torch.from_numpy(a).to(dev).requires_grad_(True)
                                             for a in (w, oy, ox)]
    with artemis_amd.synth_options(tile_radius=radius):
        out = artemis_amd.synth_unpadded(*t, 1)
        out.backward(torch.from_numpy(gout).to(dev))
    torch.cuda.synchronize()
    assert out.dtype == torch.float32                      # fp32 accumulate and output
    close(out, oracle.forward_ex(fr, w, oy, ox, 1, origin=(-2, -2)), FWD_RTOL, "output")
    rgi, rgw, rga, rgb = oracle.backward_ex(gout, fr, w, oy, ox, 1, origin=(-2, -2))
    term = float(np.abs(gout).max() * np.abs(w).max() * np.abs(fr).max())
    close(t[1].grad, rgw, BWD_RTOL, "grad_weight")
    close(t[2].grad, rga, BWD_RTOL, "grad_offset_y", term)
    close(t[3].grad, rgb, BWD_RTOL, "grad_offset_x", term)
    assert t[0].grad.dtype == torch.bfloat16               # autograd's contract; computed in fp32
    close(t[0].grad, rgi, 1e-2, "grad_input (rounded to bf16)")


def test_bf16_equals_fp32_run_on_the_rounded_frame(dev):
    # exact widening: identical bits to the fp32 path fed the same (rounded) texels
    import artemis_amd
    inp, w, oy, ox, _ = make_operands(2, 3, 50, 131, regime="gauss4", seed=12)
    fb, fr = to_bf16(inp)
    t = [torch.from_numpy(a).to(dev) for a in (w, oy, ox)]
    for radius in (-1, 8, 24):   # (bf16 frames are built for radii 8, 14, 24; smaller requests run 8)
        with artemis_amd.synth_options(tile_radius=radius):
            a = artemis_amd.FunctionSynthFrame.apply(fb.to(dev), *t, 1, 0, 0)
            b = artemis_amd.FunctionSynthFrame.apply(torch.from_numpy(fr).to(dev), *t, 1, 0, 0)
        assert torch.equal(a, b)


def test_bf16_head(dev, oracle):
    import artemis_amd
    frames, w, a, b, occ, gout = head_operands(2, 4, 30, 77, "gauss4", seed=8)
    pairs = [to_bf16(f) for f in frames]
    t = [torch.from_numpy(x).to(dev).requires_grad_(True) for x in (w, a, b, occ)]
    out = artemis_amd.chrono_head([p[0].to(dev) for p in pairs], *t)
    out.backward(torch.from_numpy(gout).to(dev))
    torch.cuda.synchronize()
    rounded = [p[1] for p in pairs]
    ref, _ = oracle.head_forward(rounded, w, a, b, occ, 1)
    close(out, ref, FWD_RTOL, "framet")
    rgw, rga, rgb, rgocc = oracle.head_backward(gout, rounded, w, a, b, occ, 1)
    term = float(np.abs(gout).max() * np.abs(w).max())
    close(t[0].grad, rgw, BWD_RTOL, "grad_weights")
    close(t[1].grad, rga, BWD_RTOL, "grad_alphas", term)
    close(t[2].grad, rgb, BWD_RTOL, "grad_betas", term)
    close(t[3].grad, rgocc, BWD_RTOL, "grad_occlusion")


def test_bf16_4k_row_strips(dev):
    # configs[4]: 3840 x 2160, bf16 frame, row strips as the 8 ranks of a node would run them;
    # strips reproduce the full-frame result bit for bit
    import artemis_amd
    from artemis_amd.sharding import row_strips
    g = torch.Generator(device=dev).manual_seed(4)
    H, W = 2160, 3840
    frame = torch.rand(1, 3, H, W, device=dev, generator=g).to(torch.bfloat16)
    w = torch.softmax(torch.randn(1, 25, H, W, device=dev, generator=g), 1)
    oy = torch.randn(1, 25, H, W, device=dev, generator=g)
    ox = torch.randn(1, 25, H, W, device=dev, generator=g)
    full = artemis_amd.synth_unpadded(frame, w, oy, ox, 1)
    for y0, y1, _, _ in row_strips(H, 8, halo=0):
        s = artemis_amd.synth_strip(frame, w[:, :, y0:y1].contiguous(), oy[:, :, y0:y1].contiguous(),
                                    ox[:, :, y0:y1].contiguous(), y0, 1, padded=False)
        assert torch.equal(s, full[:, :, y0:y1])
