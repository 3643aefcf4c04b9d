"""Data-parallel training step around the head (artemis_amd/train.py): two ranks, each on its half
of the batch, with DDP's gradient all-reduce, must follow the single-process run on the whole batch
(the op shards by sample and needs no collective of its own).  Runs both ranks on the one GPU of
the box over gloo (the 8-GPU RCCL run is the driver's)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_train(tmp_path, world, dump):
    env = dict(os.environ, PYTHONPATH=ROOT, ARTEMIS_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    args = ["--steps", "3", "--batch", "4", "--size", "48", "--features", "16", "--dump", dump, "--cudnn_benchmark", "0"]
    if world == 1:
        cmd = [sys.executable, "-m", "artemis_amd.train", "--device", "hip:0"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "-m", "artemis_amd.train",
               "--backend", "gloo"] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [float(l.split()[3]) for l in r.stdout.splitlines() if l.startswith("step ")]


def test_two_ranks_follow_the_single_process_run(tmp_path):
    assert torch.cuda.is_available()
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    loss1 = run_train(tmp_path, 1, one)
    loss2 = run_train(tmp_path, 2, two)
    assert len(loss1) == 3 and len(loss2) == 3
    # mean-of-shard-means == global mean for equal shards; gradients are averaged by DDP
    for a, b in zip(loss1, loss2):
        assert abs(a - b) <= 2e-5 * max(abs(a), 1.0), (loss1, loss2)
    p1 = torch.load(one, weights_only=True)
    p2 = torch.load(two, weights_only=True)
    assert p1.keys() == p2.keys()
    for k in p1:
        scale = float(p1[k].abs().max())
        err = float((p1[k] - p2[k]).abs().max())
        # Adamax normalises by a running max of |grad|: last-bit gradient differences between one
        # rank and two (fp32 atomics-free, but different summation order) stay tiny
        assert err <= 1e-4 * max(scale, 1e-3), (k, err, scale)
