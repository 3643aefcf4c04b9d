import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand here, prebuilt on the GPU box."""
    from oracle import synth_oracle
    synth_oracle.load()
    return synth_oracle


def make_operands(N, C, H, W, k=5, d=1, regime="gauss1", seed=0, weights="softmax"):
    """Seeded synthetic operands shaped like ChronoSynth's."""
    rng = np.random.default_rng(seed)
    Hin, Win = H + (k - 1) * d, W + (k - 1) * d
    inp = rng.random((N, C, Hin, Win), dtype=np.float32)
    shape = (N, k * k, H, W)
    if weights == "softmax":
        logits = rng.standard_normal(shape, dtype=np.float32)
        e = np.exp(logits - logits.max(1, keepdims=True))
        weight = (e / e.sum(1, keepdims=True)).astype(np.float32)
    else:
        weight = (rng.standard_normal(shape, dtype=np.float32) * 0.05).astype(np.float32)
    sig = {"init": 0.045, "gauss1": 1.0, "gauss4": 4.0, "gauss16": 16.0}
    if regime in sig:
        oy = rng.standard_normal(shape, dtype=np.float32) * np.float32(sig[regime])
        ox = rng.standard_normal(shape, dtype=np.float32) * np.float32(sig[regime])
    elif regime == "uniform3":
        oy = rng.uniform(-3, 3, shape).astype(np.float32)
        ox = rng.uniform(-3, 3, shape).astype(np.float32)
    elif regime == "clamp40":
        oy = rng.uniform(-40, 40, shape).astype(np.float32)
        ox = rng.uniform(-40, 40, shape).astype(np.float32)
    elif regime == "integer":
        oy = rng.integers(-3, 4, shape).astype(np.float32)
        ox = rng.integers(-3, 4, shape).astype(np.float32)
    elif regime == "zero":
        oy = np.zeros(shape, np.float32)
        ox = np.zeros(shape, np.float32)
    else:
        raise KeyError(regime)
    gout = rng.standard_normal((N, C, H, W), dtype=np.float32)
    return inp, weight, oy, ox, gout
