#!/usr/bin/env python3
"""Model-level runs of the harness on one MI355X (BASELINE.json configs[2] and [3] at model level):

* full ArTEMIS train step (forward, L1, backward, Adamax) on synthetic 256 x 256 crops at batch 8
  (one rank's shard of the batch-64 step on 8 GPUs) and at batch 64 on the one GPU, with the fused
  head op and with the reference's per-frame composition around FunctionSynth;
* interpolate mode at 1080p, one 4-frame window, 7 timesteps as one batch (ArTEMIS.interpolate) and
  as upstream runs it (main.py:118-121: one whole-model forward per timestep).

Random-init weights, synthetic frames.  Prints one JSON object.

    python tools/model_bench.py [--skip-b64] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from artemis_amd.model import ArTEMIS  # noqa: E402
from artemis_amd.train import ModelTrainer, synthetic_frames  # noqa: E402


def timed(fn, steps, warmup=1):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def heartbeat():
    import threading

    def beat():
        while True:
            time.sleep(45)
            print("...", flush=True)
    threading.Thread(target=beat, daemon=True).start()


def main():
    heartbeat()
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--skip-b64", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    res = {"device": torch.cuda.get_device_name(0), "weights": "random init (seed 103)", "data": "synthetic"}

    # ---- configs[2]: train step
    for fused in (True, False):
        tr = ModelTrainer(dev, fused=fused)
        for B in (8,) if args.skip_b64 else (8, 64):
            batch = synthetic_frames(B, (256, 256), 1000, dev)
            try:
                torch.cuda.reset_peak_memory_stats()
                ms = timed(lambda: tr.step(batch), args.steps)
                res[f"train_step_b{B}_256x256_{'fused' if fused else 'composed'}_ms"] = round(ms, 2)
                res[f"train_step_b{B}_256x256_{'fused' if fused else 'composed'}_peak_GB"] = round(
                    torch.cuda.max_memory_allocated() / 1e9, 1)
            except RuntimeError as e:  # e.g. out of memory at batch 64
                res[f"train_step_b{B}_256x256_{'fused' if fused else 'composed'}_error"] = str(e)[:200]
            print(json.dumps(res), flush=True)
        del tr
        torch.cuda.empty_cache()

    # ---- configs[3]: interpolate, 7 timesteps at 1080p
    torch.manual_seed(103)
    model = ArTEMIS().to(dev).eval()
    g = torch.Generator().manual_seed(1)
    frames = [torch.rand(1, 3, 1080, 1920, generator=g).to(dev) for _ in range(4)]
    ts = [i / 8 for i in range(1, 8)]
    torch.cuda.reset_peak_memory_stats()
    res["interpolate_7x1080p_batched_ms"] = round(timed(lambda: model.interpolate(frames, ts), args.steps), 1)
    res["interpolate_7x1080p_batched_peak_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
    print(json.dumps(res), flush=True)

    def per_timestep():
        with torch.no_grad():
            for t in ts:
                model(frames, torch.tensor(t, device=dev))

    res["interpolate_7x1080p_one_forward_per_timestep_ms"] = round(timed(per_timestep, args.steps), 1)
    with torch.no_grad():
        feats_ms = timed(lambda: model.features(frames), args.steps)
    res["encoder_decoder_once_1080p_ms"] = round(feats_ms, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
