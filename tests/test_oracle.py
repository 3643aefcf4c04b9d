"""CPU tests of the oracle itself (no GPU): the C restatement against hand-derived known
answers, against the independent torch restatement, against fp64 finite differences, and
against the committed golden fixtures.  PARITY UNPINNED by the reference (it has no tests or
vectors for this path); these are the pins this repository can make."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, make_operands


def onehot_weight(N, H, W, tap, k=5):
    w = np.zeros((N, k * k, H, W), np.float32)
    w[:, tap] = 1.0
    return w


# ---- known answers derived by hand from cupy_module/synth.py:24-47 ----

def test_kat_zero_offset_onehot_is_a_shift(oracle):
    # alpha = beta = 0, weight one-hot at (r, s): out[y, x] = in[y + r, x + s]
    N, C, H, W = 1, 3, 6, 8
    inp = np.random.default_rng(0).random((N, C, H + 4, W + 4), dtype=np.float32)
    z = np.zeros((N, 25, H, W), np.float32)
    for r, s in ((0, 0), (2, 3), (4, 4)):
        out = oracle.forward(inp, onehot_weight(N, H, W, r * 5 + s), z, z, 1)
        np.testing.assert_array_equal(out, inp[:, :, r:r + H, s:s + W])


def test_kat_integer_offsets_shift_and_clamp(oracle):
    # integer alpha = m, beta = n: fractions are 0, so out = in[clamp(y+r+m), clamp(x+s+n)]
    N, C, H, W = 1, 2, 5, 6
    inp = np.random.default_rng(1).random((N, C, H + 4, W + 4), dtype=np.float32)
    for m, n in ((1, -2), (-7, 3), (30, 30)):
        oy = np.full((N, 25, H, W), m, np.float32)
        ox = np.full((N, 25, H, W), n, np.float32)
        out = oracle.forward(inp, onehot_weight(N, H, W, 2 * 5 + 1), oy, ox, 1)
        ys = np.clip(np.arange(H) + 2 + m, 0, H + 3)
        xs = np.clip(np.arange(W) + 1 + n, 0, W + 3)
        np.testing.assert_array_equal(out, inp[:, :, ys][:, :, :, xs])


def test_kat_truncation_toward_zero_extrapolates(oracle):
    # alpha = -0.5: (int) gives 0 (not floor = -1), fraction -0.5, so the blend extrapolates:
    # out = 1.5 * in[y+r] - 0.5 * in[y+r+1]   (cupy_module/synth.py:29,37,41-42)
    N, C, H, W = 1, 1, 4, 4
    inp = np.random.default_rng(2).random((N, C, H + 4, W + 4)).astype(np.float32)
    oy = np.full((N, 25, H, W), -0.5, np.float32)
    ox = np.zeros((N, 25, H, W), np.float32)
    out = oracle.forward(inp, onehot_weight(N, H, W, 0), oy, ox, 1)
    want = inp[:, :, 0:H, 0:W] * np.float32(1.5) + inp[:, :, 1:H + 1, 0:W] * np.float32(-0.5)
    np.testing.assert_allclose(out, want, rtol=0, atol=1e-6)
    # +0.5 interpolates: 0.5 * in[y] + 0.5 * in[y+1]
    out = oracle.forward(inp, onehot_weight(N, H, W, 0), -oy, ox, 1)
    want = (inp[:, :, 0:H, 0:W] + inp[:, :, 1:H + 1, 0:W]) * np.float32(0.5)
    np.testing.assert_allclose(out, want, rtol=0, atol=1e-6)


def test_kat_clamp_collapses_top_onto_bottom(oracle):
    # far beyond the frame both rows clamp to Hin-1, so the result is that row whatever the
    # fraction: (1-a) v + a v = v.
    N, C, H, W = 1, 1, 3, 3
    inp = np.random.default_rng(3).random((N, C, H + 4, W + 4)).astype(np.float32)
    oy = np.full((N, 25, H, W), 100.25, np.float32)
    ox = np.zeros((N, 25, H, W), np.float32)
    out = oracle.forward(inp, onehot_weight(N, H, W, 0), oy, ox, 1)
    want = np.broadcast_to(inp[:, :, -1:, 0:W], out.shape)
    np.testing.assert_allclose(out, want, rtol=0, atol=1e-6)


def test_kat_weights_are_linear(oracle):
    inp, w, oy, ox, _ = make_operands(1, 3, 5, 6, regime="uniform3", seed=4)
    a = oracle.forward(inp, w, oy, ox, 1)
    b = oracle.forward(inp, 2 * w, oy, ox, 1)
    np.testing.assert_allclose(b, 2 * a, rtol=1e-6, atol=1e-6)


# ---- the two restatements against each other ----

@pytest.mark.parametrize("regime", ["init", "gauss1", "uniform3", "clamp40", "integer"])
@pytest.mark.parametrize("shape", [(2, 3, 6, 7, 5, 1), (1, 2, 5, 9, 3, 2), (1, 3, 4, 70, 5, 1)])
def test_c_matches_torch_restatement(oracle, regime, shape):
    N, C, H, W, k, d = shape
    inp, w, oy, ox, gout = make_operands(N, C, H, W, k, d, regime, seed=11)
    out_c = oracle.forward(inp, w, oy, ox, d)
    t = [torch.from_numpy(a).clone().requires_grad_(True) for a in (inp, w, oy, ox)]
    out_t = oracle.forward_torch(*t, d)
    # same fp32 operation order, no contraction on either side
    np.testing.assert_allclose(out_c, out_t.detach().numpy(), rtol=1e-6, atol=1e-6)
    out_t.backward(torch.from_numpy(gout))
    gi, gw, ga, gb = oracle.backward(gout, inp, w, oy, ox, d)
    scale = lambda a: 1e-5 * max(1.0, float(np.abs(a).max()))
    np.testing.assert_allclose(gw, t[1].grad.numpy(), rtol=1e-4, atol=scale(gw))
    np.testing.assert_allclose(ga, t[2].grad.numpy(), rtol=1e-4, atol=scale(ga))
    np.testing.assert_allclose(gb, t[3].grad.numpy(), rtol=1e-4, atol=scale(gb))
    np.testing.assert_allclose(gi, t[0].grad.numpy(), rtol=1e-4, atol=scale(gi))


def test_gradients_match_fp64_finite_differences(oracle):
    # away from integer offsets the forward is smooth: central differences in fp64 through the
    # torch restatement pin the C gradients (reference formulas :140-145, :190-195)
    N, C, H, W = 1, 3, 3, 4
    rng = np.random.default_rng(5)
    inp = rng.random((N, C, H + 4, W + 4))
    w = rng.normal(0, 1, (N, 25, H, W))
    frac = rng.uniform(0.2, 0.8, (2, N, 25, H, W))
    whole = rng.integers(-2, 3, (2, N, 25, H, W))
    oy, ox = whole[0] + frac[0] * np.sign(whole[0] + 0.5), whole[1] + frac[1] * np.sign(whole[1] + 0.5)
    gout = rng.normal(0, 1, (N, C, H, W))
    T = lambda a: torch.from_numpy(np.asarray(a, np.float64))

    def loss(i, ww, a, b):
        return float((oracle.forward_torch(T(i), T(ww), T(a), T(b), 1) * T(gout)).sum())

    gi, gw, ga, gb = oracle.backward(gout, inp, w, oy, ox, 1)
    eps = 1e-4
    for arr, grad, which in ((inp, gi, 0), (w, gw, 1), (oy, ga, 2), (ox, gb, 3)):
        flat_idx = rng.choice(arr.size, size=12, replace=False)
        for fi in flat_idx:
            idx = np.unravel_index(fi, arr.shape)
            args = [inp, w, oy, ox]
            plus, minus = arr.copy(), arr.copy()
            plus[idx] += eps
            minus[idx] -= eps
            args[which] = plus
            lp = loss(*args)
            args[which] = minus
            lm = loss(*args)
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - grad[idx]) <= 2e-3 * max(1.0, abs(fd)), (which, idx, fd, grad[idx])


def test_saturating_cast_and_nan(oracle):
    # huge / non-finite offsets: (int) saturates, NaN -> 0, sums wrap; must not crash and the
    # finite lanes must be unaffected
    N, C, H, W = 1, 3, 4, 5
    inp, w, oy, ox, _ = make_operands(N, C, H, W, regime="gauss1", seed=6)
    ref = oracle.forward(inp, w, oy, ox, 1)
    oy2, ox2 = oy.copy(), ox.copy()
    oy2[0, 3, 1, 1] = 3e9
    ox2[0, 4, 2, 2] = -3e9
    oy2[0, 5, 0, 0] = np.nan
    out = oracle.forward(inp, w, oy2, ox2, 1)
    mask = np.ones((H, W), bool)
    mask[1, 1] = mask[2, 2] = mask[0, 0] = False
    np.testing.assert_array_equal(out[..., mask], ref[..., mask])


# ---- golden fixtures freeze the oracle ----

@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_golden(oracle, name):
    g = load_golden(name)
    d = int(g["dilation"])
    out = oracle.forward(g["input"], g["weight"], g["offset_y"], g["offset_x"], d)
    np.testing.assert_array_equal(out, g["output"])
    gi, gw, ga, gb = oracle.backward(g["grad_output"], g["input"], g["weight"], g["offset_y"],
                                     g["offset_x"], d)
    np.testing.assert_array_equal(gw, g["grad_weight"])
    np.testing.assert_array_equal(ga, g["grad_offset_y"])
    np.testing.assert_array_equal(gb, g["grad_offset_x"])
    np.testing.assert_array_equal(gi, g["grad_input"])


def test_golden_set_is_present():
    assert len(golden_names()) >= 9
