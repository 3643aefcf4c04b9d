"""Reentrancy: the reference's op is called from the user's thread (forward) and from autograd's
thread (backward); here several Python threads drive the C ABI at once, each on its own stream
(ctypes releases the GIL during the calls), and every thread must get its own correct result and
its own error text."""
import threading

import numpy as np
import pytest
import torch

from conftest import make_operands

pytestmark = pytest.mark.gpu


def test_concurrent_calls_from_threads(oracle):
    import artemis_amd
    from artemis_amd import _lib
    dev = torch.device("cuda:0")
    lib = _lib.load()
    cases = []
    for i in range(4):
        inp, w, oy, ox, gout = make_operands(1, 3, 30 + 7 * i, 60 + 11 * i, regime="gauss1", seed=40 + i)
        cases.append((inp, w, oy, ox, gout, oracle.forward(inp, w, oy, ox, 1),
                      oracle.backward(gout, inp, w, oy, ox, 1, with_input=False)))
    errors = []

    def worker(i):
        try:
            inp, w, oy, ox, gout, ref_out, ref_g = cases[i]
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for _ in range(20):
                    t = [torch.from_numpy(a).to(dev) for a in (inp, w, oy, ox)]
                    for x in t[1:]:
                        x.requires_grad_(True)
                    out = artemis_amd.FunctionSynth.apply(*t, 1)
                    out.backward(torch.from_numpy(gout).to(dev))
                    # a failing call on this thread must report this thread's message
                    rc = lib.artemis_synth_forward(None, None, None, None, None, 1, 3, 4, 4, i + 6, 0,
                                                   -1, None)
                    assert rc == 1
                    assert f"k={i + 6}" in lib.artemis_synth_last_error().decode()
                stream.synchronize()
            np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out, rtol=1e-5,
                                       atol=1e-5 * np.abs(ref_out).max())
            for g, r in zip(t[1:], ref_g[1:]):
                np.testing.assert_allclose(g.grad.cpu().numpy(), r, rtol=1e-4, atol=1e-4 * np.abs(r).max())
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
