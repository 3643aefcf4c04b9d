#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/synth_oracle.c).

PARITY UNPINNED: the reference has no golden vectors for this path and cannot run here, so these
fixtures are produced by this repository's restatement of cupy_module/synth.py, not by the
reference.  They freeze the oracle (a later edit that changes its numbers fails
tests/test_oracle.py) and give the GPU tests fixed operands.  Inputs are stored, not re-drawn, so
the files do not depend on any RNG implementation.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import synth_oracle as so  # noqa: E402

# name -> (N, C, H, W, k, d, regime, seed)
CASES = {
    "a_init": (2, 3, 6, 7, 5, 1, "init", 1),
    "a_uniform3": (2, 3, 6, 7, 5, 1, "uniform3", 2),
    "a_integer": (2, 3, 6, 7, 5, 1, "integer", 3),
    "a_clamp40": (2, 3, 6, 7, 5, 1, "clamp40", 4),
    "a_half": (2, 3, 6, 7, 5, 1, "half", 5),
    "b_init": (1, 3, 9, 67, 5, 1, "init", 6),
    "b_gauss2": (1, 3, 9, 67, 5, 1, "gauss2", 7),
    "c_k3d2c2": (2, 2, 5, 9, 3, 2, "uniform3", 8),
    "d_k1": (1, 4, 4, 5, 1, 1, "uniform3", 9),
}


def offsets(rng, shape, regime):
    if regime == "init":  # measured at random init: std ~0.045
        return rng.normal(0, 0.045, shape), rng.normal(0, 0.045, shape)
    if regime == "uniform3":
        return rng.uniform(-3, 3, shape), rng.uniform(-3, 3, shape)
    if regime == "integer":  # exact integers: fractions are exactly 0
        return rng.integers(-3, 4, shape).astype(np.float64), rng.integers(-3, 4, shape).astype(np.float64)
    if regime == "clamp40":  # every border clamp on every side
        return rng.uniform(-40, 40, shape), rng.uniform(-40, 40, shape)
    if regime == "half":  # multiples of 0.5 incl. negatives: exact arithmetic, extrapolation
        return rng.integers(-6, 7, shape) * 0.5, rng.integers(-6, 7, shape) * 0.5
    if regime == "gauss2":
        return rng.normal(0, 2, shape), rng.normal(0, 2, shape)
    raise KeyError(regime)


def make_case(N, C, H, W, k, d, regime, seed):
    rng = np.random.default_rng(seed)
    Hin, Win = H + (k - 1) * d, W + (k - 1) * d
    inp = rng.random((N, C, Hin, Win)).astype(np.float32)
    wshape = (N, k * k, H, W)
    logits = rng.normal(0, 1, wshape)
    weight = (np.exp(logits) / np.exp(logits).sum(1, keepdims=True)).astype(np.float32)
    oy, ox = offsets(rng, wshape, regime)
    oy, ox = oy.astype(np.float32), ox.astype(np.float32)
    gout = rng.normal(0, 1, (N, C, H, W)).astype(np.float32)
    out = so.forward(inp, weight, oy, ox, d)
    gin, gw, ga, gb = so.backward(gout, inp, weight, oy, ox, d)
    return dict(input=inp, weight=weight, offset_y=oy, offset_x=ox, grad_output=gout,
                output=out, grad_input=gin, grad_weight=gw, grad_offset_y=ga, grad_offset_x=gb,
                dilation=np.int32(d))


def main():
    so.build()
    for name, spec in CASES.items():
        np.savez(os.path.join(HERE, name + ".npz"), **make_case(*spec))
        print(name, spec)


if __name__ == "__main__":
    main()
