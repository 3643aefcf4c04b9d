#!/usr/bin/env python3
"""Generate tests/golden/model/config1_cpu_forward.npz: BASELINE.json configs[0] -- ArTEMIS forward on the
CPU, random init under seed 103 (config.py:62), batch 1, 4 x 3 x 64 x 64 context window, t = 0.5,
the synthesis op through the oracle's PyTorch CPU transcription -- as computed by THIS repository's
restatement of the model (artemis_amd/model).  The reference model cannot be built here (CuPy and
timm are not installed), so the fixture freezes this build's numbers: parity unpinned.

Stored: the three model outputs, and for the 12 op calls (3 heads x 4 frames) the operand shapes,
mean |.| of each operand and CRC-32 checksums of their bytes on the generating host.

    python tests/golden/model/make_config1.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import pytest
    import test_artemis_model as tm

    mp = pytest.MonkeyPatch()
    try:
        model, frames, outs = tm.config1_run(mp)
        calls = list(tm._OracleOp.calls)
    finally:
        mp.undo()
    names = ("input", "weight", "offset_y", "offset_x")
    np.savez_compressed(
        os.path.join(HERE, "config1_cpu_forward.npz"),
        out_ll=outs[0].numpy(), out_l=outs[1].numpy(), out=outs[2].numpy(),
        operand_abs_means=np.array([[float(t.abs().mean()) for t in c] for c in calls], np.float64),
        operand_crc32=np.array([[tm.checksum(t) for t in c] for c in calls], np.uint32),
        operand_shapes=np.array([[list(t.shape) for t in c] for c in calls], np.int32),
        operand_names=np.array(names),
        frames_crc32=np.array([tm.checksum(f) for f in frames], np.uint32),
    )
    print("wrote config1_cpu_forward.npz;", [tuple(o.shape) for o in outs],
          "max |offset|", max(float(c[2].abs().max()) for c in calls))


if __name__ == "__main__":
    main()
