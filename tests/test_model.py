"""ChronoSynth restated on the fused head (artemis_amd/model/chrono_synth.py) and device
selection.  CPU: state-dict compatibility with the upstream module's parameter names and shapes
(read off model/chrono_synth.py:12-75), the time channel, argument parsing.  GPU: the module's
fused forward/backward against the same module running the reference's per-frame composition."""
import argparse

import pytest
import torch

from artemis_amd import device as dev_mod
from artemis_amd.model import ChronoSynth


def expected_state_dict_shapes(num_inputs=4, nf=64, k=5):
    """Keys and shapes of upstream ChronoSynth(num_inputs, nf, k, 1).state_dict(), derived by hand
    from model/chrono_synth.py:12-75 and model/helper_modules.py:25-36."""
    c, taps = nf + 1, k * k
    exp = {}

    def subnet(name, up, out):
        exp[f"{name}.0.weight"] = (nf, c, 3, 3)
        exp[f"{name}.0.bias"] = (nf,)
        exp[f"{name}.2.weight"] = (up, nf, 3, 3)
        exp[f"{name}.2.bias"] = (up,)
        exp[f"{name}.4.weight"] = (up, up, 3, 3)      # ConvTranspose2d: (in, out, kh, kw)
        exp[f"{name}.4.bias"] = (up,)
        exp[f"{name}.5.weight"] = (out, up, 3, 3)
        exp[f"{name}.5.bias"] = (out,)

    subnet("ModuleWeight", taps, taps)
    subnet("ModuleAlpha", taps, taps)
    subnet("ModuleBeta", taps, taps)
    subnet("ModuleOcclusion", nf, num_inputs)
    exp["feature_fuse.conv.0.weight"] = (c, c * num_inputs, 1, 1)
    exp["feature_fuse.conv.0.bias"] = (c,)
    return exp


def test_state_dict_matches_upstream_names_and_shapes():
    m = ChronoSynth(4, 64, 5, 1)
    sd = m.state_dict()
    exp = expected_state_dict_shapes()
    assert set(sd) == set(exp)
    for k, shape in exp.items():
        assert tuple(sd[k].shape) == shape, k
    # the three heads of ArTEMIS(4, "concat", 5, 1) hold 960,960 parameters
    assert 3 * sum(p.numel() for p in m.parameters()) == 960960


def test_operands_shapes_and_time_channel_pairing():
    torch.manual_seed(0)
    m = ChronoSynth(4, 8, 5, 1)
    feats = torch.randn(2, 8, 4, 6, 7)
    t = torch.tensor([0.5, 0.25])
    w, a, b, occ = m.operands(feats, (11, 13), t)
    assert w.shape == (2, 4, 25, 11, 13) and a.shape == w.shape and b.shape == w.shape
    assert occ.shape == (2, 4, 11, 13)
    torch.testing.assert_close(w.sum(2), torch.ones(2, 4, 11, 13))       # Softmax(1), :39
    torch.testing.assert_close(occ.sum(1), torch.ones(2, 11, 13))         # Softmax(dim=1), :55
    # the reference's loop (:95-98) pairs frame j with context_frame_times[j - 1] (negative index
    # for j = 0): restate it literally and compare the time channel the sub-networks see
    context = [-0.25, 0, 1, 1.25]
    T = 4
    tt = torch.ones(2, 1, T, 6, 7)
    start, end = -T // 2 + 1, T // 2 + 1
    for i in range(start, end):
        tt[:, :, i - start] *= torch.abs(context[i] - t).view(2, 1, 1, 1)
    seen = {}
    def grab(mod, inp, out):
        seen["x"] = inp[0]

    h = m.ModuleAlpha[0].register_forward_hook(grab)
    m.operands(feats, (11, 13), t)
    h.remove()
    got = seen["x"].view(2, T, 9, 6, 7)[:, :, 8]      # last channel = time, per (b, frame)
    torch.testing.assert_close(got, tt[:, 0])


def test_cpu_forward_raises_like_the_reference():
    m = ChronoSynth(4, 8, 5, 1)
    frames = [torch.rand(1, 3, 12, 12) for _ in range(4)]
    with pytest.raises(NotImplementedError):          # cupy_module/synth.py:310-311
        m(torch.randn(1, 8, 4, 6, 6), frames, (12, 12), torch.tensor([0.5]))


def test_device_argument_parsing(monkeypatch):
    assert dev_mod.parse_device("hip:3") == 3
    assert dev_mod.parse_device("cuda:1") == 1
    assert dev_mod.parse_device("2") == 2
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert dev_mod.parse_device("auto") == 5
    assert dev_mod.parse_device(None) == 5
    with pytest.raises(ValueError):
        dev_mod.parse_device("tpu:0")
    p = dev_mod.add_device_argument(argparse.ArgumentParser())
    assert p.parse_args(["--device", "hip:1"]).device == "hip:1"
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            dev_mod.select_device("hip:0")


@pytest.mark.gpu
@pytest.mark.parametrize("softmax", [True, False])
def test_chronosynth_fused_equals_reference_composition(softmax):
    dev = dev_mod.select_device("hip:0")
    torch.manual_seed(103)
    m = ChronoSynth(4, 16, 5, 1, apply_softmax=softmax).to(dev)
    B, H, W = 2, 37, 50
    feats = torch.randn(B, 16, 4, 19, 25, device=dev)
    frames = [torch.rand(B, 3, 74, 100, device=dev) for _ in range(4)]
    t = torch.tensor([0.5, 0.3], device=dev)
    gout = torch.randn(B, 3, H, W, device=dev)

    def run(fused):
        m.fused = fused
        m.zero_grad()
        out = m(feats, frames, (H, W), t)
        out.backward(gout)
        return out.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}

    out_f, g_f = run(True)
    out_c, g_c = run(False)
    assert torch.equal(out_f, out_c)
    for k in g_c:
        scale = float(g_c[k].abs().max())
        err = float((g_f[k] - g_c[k]).abs().max())
        assert err <= 2e-4 * max(scale, 1e-6), (k, err, scale)
