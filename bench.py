#!/usr/bin/env python3
"""Headline benchmark: synthesized Mpix/s, forward + backward, 4 context frames at 1080p
(BASELINE.json `metric`), on N GPUs of one node.

A *step* is one pass of the hot path over one batch of synthetic operands: for each of the 4
context frames one `artemis_synth_forward`, then for each frame one `artemis_synth_backward`
(gradWeight + gradOffset_y + gradOffset_x; the frames need no gradient in ArTEMIS,
model/chrono_synth.py:120-126) -- what one ChronoSynth head does per training step for
`batch` samples of 1080 x 1920.  A *synthesized pixel* is one (n, y, x) of that head
(SURVEY.md 8d).  Operands are resident in HBM before the timed region; the kernels are called
through the C ABI on torch's current stream with preallocated outputs, with the LIBRARY DEFAULT
tile radius (what a caller of FunctionSynth gets).

Multi-GPU: the op shards by batch x timestep with no data-path collective (every output element
depends only on its own sample, cupy_module/synth.py:19), so every rank runs the same per-GPU
workload on its own shard (weak scaling) and `value` = pixels of all ranks / max-over-ranks time.

`--workload interp7` is BASELINE.json configs[3] instead (interpolate mode: one 4-frame window,
7 timesteps, forward only): the 7 timesteps are dealt to the ranks with sharding.shard_items and
each rank runs ITS timesteps as one fused-head launch -- all 7 in one launch on 1 GPU, one per GPU
on 8 GPUs with one rank idle (strong scaling; `value` = 7 x 1080p pixels / max-over-ranks time).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from artemis_amd import _lib  # noqa: E402

H, W, C, K_TAPS, FRAMES = 1080, 1920, 3, 25, 4
PAD = 2
# algorithmic (compulsory) bytes per pixel per frame, fp32: SURVEY.md 8d / DESIGN.md
FWD_BYTES = 3 * K_TAPS * 4 + C * 4 + C * 4 * (H + 2 * PAD) * (W + 2 * PAD) / (H * W)
BWD_BYTES = 3 * K_TAPS * 4 + 3 * K_TAPS * 4 + C * 4 + C * 4 * (H + 2 * PAD) * (W + 2 * PAD) / (H * W)
# fused head, forward, per head pixel (T = 4 frames read in place + occlusion, one output)
HEAD_FWD_BYTES = FRAMES * (3 * K_TAPS * 4 + C * 4 + 4) + C * 4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def make_offsets(regime, shape, device, g):
    """Offset regimes: 'sigma:<s>' i.i.d. N(0, s^2) per pixel and tap (the headline
    is sigma:1); 'smooth30' = one low-frequency flow field of about +-30 px shared by the 25 taps
    (control points every ~128 px, N(0, 15^2), bicubic) + 0.3 px noise, vertical and horizontal
    fields independent -- large coherent motion with steep gradients; 'pan' = every tap displaced
    by (20, -35) px + N(0, 1)."""
    N, K, h, w = shape
    rn = lambda *s: torch.randn(*s, device=device, generator=g)
    if regime.startswith("sigma:"):
        s = float(regime.split(":")[1])
        return rn(N, K, h, w) * s, rn(N, K, h, w) * s
    if regime == "smooth30":
        lo_y = rn(N, 1, h // 128 + 2, w // 128 + 2) * 15
        lo_x = rn(N, 1, h // 128 + 2, w // 128 + 2) * 15
        up = lambda lo: torch.nn.functional.interpolate(lo, size=(h, w), mode="bicubic").expand(N, K, h, w)
        return ((up(lo_y) + rn(N, K, h, w) * 0.3).contiguous(),
                (up(lo_x) + rn(N, K, h, w) * 0.3).contiguous())
    if regime == "pan":
        return (rn(N, K, h, w) + 20.0).contiguous(), (rn(N, K, h, w) - 35.0).contiguous()
    raise KeyError(regime)


def make_frames(batch, sigma, device, seed, regime=None, size=None, backward=True):
    """Synthetic operands per SURVEY.md 8d: frames rand in [0,1] replication-padded by 2, weights
    softmax over 25 N(0,1) logits, offsets per `regime` (default N(0, sigma^2)), gradOutput randn.
    size: (h, w) of the output, default 1080 x 1920; backward=False leaves out the gradient tensors."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    regime = regime or f"sigma:{sigma}"
    h, w_ = size or (H, W)
    frames = []
    for _ in range(FRAMES):
        frame = torch.rand(batch, C, h, w_, device=device, generator=g)
        inp = torch.nn.functional.pad(frame, (PAD,) * 4, mode="replicate").contiguous()
        w = torch.softmax(torch.randn(batch, K_TAPS, h, w_, device=device, generator=g), 1)
        oy, ox = make_offsets(regime, (batch, K_TAPS, h, w_), device, g)
        out = torch.empty(batch, C, h, w_, device=device)
        f = dict(inp=inp, w=w, oy=oy, ox=ox, out=out)
        if backward:
            f["gout"] = torch.randn(batch, C, h, w_, device=device, generator=g)
            f["gw"], f["ga"], f["gb"] = (torch.empty(batch, K_TAPS, h, w_, device=device) for _ in range(3))
        frames.append(f)
    return frames


def _cpu_operands(rng, n, h, w, sigma):
    import numpy as np
    shape = (n, K_TAPS, h, w)
    inp = rng.random((n, C, h + 2 * PAD, w + 2 * PAD), dtype=np.float32)
    logits = rng.standard_normal(shape, dtype=np.float32)
    e = np.exp(logits)
    wt = (e / e.sum(1, keepdims=True)).astype(np.float32)
    oy = rng.standard_normal(shape, dtype=np.float32) * np.float32(sigma)
    ox = rng.standard_normal(shape, dtype=np.float32) * np.float32(sigma)
    gout = rng.standard_normal((n, C, h, w), dtype=np.float32)
    return inp, wt, oy, ox, gout


def cpu_baseline(threads, sigma):
    """The oracle's C restatement (kind "port": the reference has no CPU path and cannot be
    built here) timed on the host cores for whole steps of the same workload at batch 1, plus --
    as BASELINE.md's CPU-baseline plan asks -- the PyTorch transcription on the same step and both
    on configs[1] (N = 16, 256 x 448, 4 frames, forward only), each on a bounded sample."""
    import numpy as np
    from oracle import synth_oracle as so
    so.set_threads(threads)
    torch.set_num_threads(threads)
    rng = np.random.default_rng(103)
    sets = [_cpu_operands(rng, 1, H, W, sigma) for _ in range(FRAMES)]
    # whole steps until about 10 s of CPU work (at least 1, at most 12 steps)
    t_total, steps = 0.0, 0
    while steps < 1 or (t_total < 10.0 and steps < 12):
        for inp, w, oy, ox, gout in sets:
            t0 = time.perf_counter()
            so.forward(inp, w, oy, ox, 1)
            so.grad_weight(gout, inp, w, oy, ox, 1)
            so.grad_alpha(gout, inp, w, oy, ox, 1)
            so.grad_beta(gout, inp, w, oy, ox, 1)
            t_total += time.perf_counter() - t0
        steps += 1
    res = {
        "value": round(steps * H * W / 1e6 / t_total, 3),
        "unit": "Mpix/s",
        "cores": threads,
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"{steps} steps of {FRAMES} frames x 1080x1920, batch 1, fwd + gradWeight/Alpha/Beta, "
                  f"oracle/synth_oracle.c (gcc -O2, OpenMP, {threads} threads of "
                  f"{os.cpu_count()} host cpus), {t_total:.1f} s",
    }
    # the same step through the vectorised PyTorch transcription (forward + autograd backward to
    # weight/offsets): ONE frame, scaled to a 4-frame step
    t = [torch.from_numpy(a) for a in sets[0]]
    t0 = time.perf_counter()
    leaves = [x.clone().requires_grad_(True) for x in t[1:4]]
    so.forward_torch(t[0], *leaves, 1).backward(t[4])
    dt = time.perf_counter() - t0
    res["torch_transcription"] = {
        "value": round(H * W / 1e6 / (dt * FRAMES), 3), "unit": "Mpix/s", "cores": threads,
        "sample": f"1 frame of 1080x1920 fwd + autograd bwd (oracle.forward_torch), {dt:.1f} s, x{FRAMES} "
                  "frames per step"}
    res["config2_forward"] = _config2_cpu(so, rng, sigma)
    return res


def cpu_baseline_config2(threads, sigma):
    """cpu_baseline object of `--workload config2`: the oracle's C port (and the PyTorch
    transcription) on configs[1]'s own shape."""
    import numpy as np
    from oracle import synth_oracle as so
    so.set_threads(threads)
    torch.set_num_threads(threads)
    c = _config2_cpu(so, np.random.default_rng(103), sigma, budget_s=12.0)
    return {"value": c["port"], "unit": "Mpix/s", "cores": threads, "host_cpus": os.cpu_count(),
            "kind": "port", "sample": c["sample"],
            "torch_transcription": {"value": c["torch_transcription"], "unit": "Mpix/s", "cores": threads}}


def _config2_cpu(so, rng, sigma, budget_s=3.0):
    # configs[1]: N = 16, 256 x 448, 4 frames, forward only
    n2, h2, w2 = 16, 256, 448
    inp, wt, oy, ox, _ = _cpu_operands(rng, n2, h2, w2, sigma)
    t0 = time.perf_counter()
    reps = 0
    while reps < 1 or (time.perf_counter() - t0 < budget_s and reps < 32):
        so.forward(inp, wt, oy, ox, 1)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    tt = [torch.from_numpy(a) for a in (inp[:2], wt[:2], oy[:2], ox[:2])]
    t0 = time.perf_counter()
    with torch.no_grad():
        so.forward_torch(*tt, 1)
    dtt = (time.perf_counter() - t0) * (n2 / 2)
    return {
        "workload": "N=16, 256x448, 4 frames, forward only (BASELINE.json configs[1])",
        "unit": "Mpix/s (synthesized: 4 frame calls per pixel)",
        "port": round(n2 * h2 * w2 / 1e6 / (dt * FRAMES), 3),
        "torch_transcription": round(n2 * h2 * w2 / 1e6 / (dtt * FRAMES), 3),
        "sample": f"port: {reps} x one frame call of all 16 samples ({dt:.2f} s each); transcription: "
                  f"2 of 16 samples once, scaled; both x{FRAMES} frames",
    }


def head_operands(B, device, regime, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    T = FRAMES
    frames = [torch.rand(B, C, H, W, device=device, generator=g) for _ in range(T)]
    w = torch.softmax(torch.randn(B, T, K_TAPS, H, W, device=device, generator=g), 2)
    ab = [make_offsets(regime, (B, K_TAPS, H, W), device, g) for _ in range(T)]
    a = torch.stack([p[0] for p in ab], 1).contiguous()
    b = torch.stack([p[1] for p in ab], 1).contiguous()
    occ = torch.softmax(torch.randn(B, T, H, W, device=device, generator=g), 1)
    gout = torch.randn(B, C, H, W, device=device, generator=g)
    return frames, w, a, b, occ, gout


def head_timing(device, radius, sigma, iters=20):
    """Beside the headline: the whole ChronoSynth head (model/chrono_synth.py:114-129: pad, 4 x op,
    occlusion product, sum; forward + backward to weights/alphas/betas/occlusion) at 1080p,
    (a) as the fused artemis_synth_head_* pair through the C ABI and (b) as the reference's own
    composition around this library's per-frame op with torch doing the rest (batch 1 and batch 2:
    from batch 2 on the reference's `.contiguous()` slices are real copies)."""
    import artemis_amd
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for B in (1, 2):
        T = FRAMES
        frames, w, a, b, occ, gout = head_operands(B, device, f"sigma:{sigma}", 7 + B)
        out = torch.empty(B, C, H, W, device=device)
        gw, ga, gb = (torch.empty_like(w) for _ in range(3))
        gocc = torch.empty_like(occ)
        ptrs = (ctypes.c_void_p * T)(*[f.data_ptr() for f in frames])

        def fused():
            _lib.check(lib.artemis_synth_head_forward(
                ptrs, T, w.data_ptr(), a.data_ptr(), b.data_ptr(), occ.data_ptr(), out.data_ptr(),
                B, H, W, H, W, -2, -2, 0, radius, stream), "head_forward")
            _lib.check(lib.artemis_synth_head_backward(
                ptrs, T, w.data_ptr(), a.data_ptr(), b.data_ptr(), occ.data_ptr(), gout.data_ptr(),
                gw.data_ptr(), ga.data_ptr(), gb.data_ptr(), gocc.data_ptr(), B, H, W, H, W, -2, -2,
                0, radius, stream), "head_backward")

        leaves = [t.clone().requires_grad_(True) for t in (w, a, b, occ)]

        def composed():
            lw, la, lb, locc = leaves
            warp = []
            with artemis_amd.synth_options(tile_radius=radius):
                for i in range(T):
                    padded = torch.nn.functional.pad(frames[i], (PAD,) * 4, mode="replicate")
                    warp.append(locc[:, i:i + 1] * artemis_amd.FunctionSynth.apply(
                        padded, lw[:, i].contiguous(), la[:, i].contiguous(),
                        lb[:, i].contiguous(), 1))
                sum(warp).backward(gout)
            for t in leaves:
                t.grad = None

        for name, fn in (("fused", fused), ("composed", composed)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            res[f"{name}_b{B}_ms"] = round(ms, 4)
            res[f"{name}_b{B}_Mpix_s"] = round(B * H * W / (ms * 1e-3) / 1e6, 1)
        del leaves, frames, w, a, b, occ, gw, ga, gb
        torch.cuda.empty_cache()
    return res


def source_hash():
    """Hash of the kernel sources: recorded PMC traffic is only quoted for the build it was taken on."""
    h = hashlib.sha256()
    for name in ("synth_device.hpp", "synth_tiles.hip", "synth_kernels.hip"):
        with open(os.path.join(ROOT, "artemis_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def kernel_shape(lib, radius, h, w, head=False):
    """Which kernel shape a launch with `radius` runs: "lean", "14", "24", ... (adaptive: the
    library's current choice for this tensor shape)."""
    if radius == _lib.RADIUS_AUTO:
        radius = adaptive_state(lib, h, w, head)["radius"]
        return "lean" if radius == 8 else str(radius)
    if radius == _lib.RADIUS_SMALL:
        return "lean"
    return "14" if radius == _lib.RADIUS_DEFAULT else str(radius)


def recorded_traffic(key, batch, shape):
    """HBM bytes per launch from the rocprofv3 PMC passes recorded in profiles/pmc_traffic.json
    (counters cannot be read inside this process); None unless recorded for these kernel sources
    and this kernel shape."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(tpath))
    except Exception:
        return None, None
    if rec.get("source_hash") != source_hash():
        return None, "profiles/pmc_traffic.json was recorded for other kernel sources"
    if rec.get("shape", "14") != shape:
        return None, f"profiles/pmc_traffic.json holds the {rec.get('shape', '14')} shape, this run used {shape}"
    v = rec.get(key)
    return (None if v is None else int(v) * batch), "profiles/pmc_traffic.json (rocprofv3 --pmc, recorded)"


def adaptive_state(lib, h, w, head=False):
    """What ARTEMIS_SYNTH_RADIUS_AUTO holds for this shape right now (host-only query)."""
    r, c = ctypes.c_int(0), ctypes.c_ulonglong(0)
    f8, f14 = ctypes.c_double(0), ctypes.c_double(0)
    lib.artemis_synth_auto_query(h, w, int(head), ctypes.byref(r), ctypes.byref(f8), ctypes.byref(f14),
                                 ctypes.byref(c))
    return {"radius": r.value, "taps_beyond_8px": f8.value, "taps_beyond_14px": f14.value,
            "forward_calls": c.value}


class Op:
    """fwd / bwd launchers of one operand set list through the C ABI."""

    def __init__(self, lib, frames, B, radius, stream, size=None):
        self.lib, self.frames, self.B, self.radius, self.stream = lib, frames, B, radius, stream
        self.h, self.w = size or (H, W)

    def fwd(self, f):
        _lib.check(self.lib.artemis_synth_forward(
            f["inp"].data_ptr(), f["w"].data_ptr(), f["oy"].data_ptr(), f["ox"].data_ptr(),
            f["out"].data_ptr(), self.B, C, self.h, self.w, 5, 1, self.radius, self.stream), "forward")

    def bwd(self, f):
        _lib.check(self.lib.artemis_synth_backward(
            f["inp"].data_ptr(), f["w"].data_ptr(), f["oy"].data_ptr(), f["ox"].data_ptr(),
            f["gout"].data_ptr(), None, f["gw"].data_ptr(), f["ga"].data_ptr(),
            f["gb"].data_ptr(), self.B, C, self.h, self.w, 5, 1, _lib.GIN_ATOMIC, self.radius, None, 0,
            self.stream), "backward")

    def step(self, events=None):
        # one event pair around the 4 forward launches and one around the 4 backward launches:
        # per-launch averages without an event between every two kernels
        if events is not None:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
        for f in self.frames:
            self.fwd(f)
        if events is not None:
            e1.record()
        for f in self.frames:
            self.bwd(f)
        if events is not None:
            e2.record()
            events.append((e0, e1, e2))


def regime_timing(lib, device, B, radius, stream, regime, steps, headline_mpix):
    """The same step under another offset regime (few steps, HIP events).  '<regime>@<r>' runs it
    at tile radius r instead of the headline's."""
    if "@" in regime:
        regime, r = regime.split("@")
        radius = int(r)
    frames = make_frames(B, 0.0, device, seed=211, regime=regime)
    op = Op(lib, frames, B, radius, stream)
    # (the adaptive radius follows the new offsets with the lag of the launches in flight, and may
    # take two moves -- lean -> 24 -> 14 on a pan: let some rounds of warm-up complete)
    for _ in range(5 if radius == _lib.RADIUS_AUTO else 2):
        op.step()
        op.step()
        torch.cuda.synchronize()
    ev = []
    for _ in range(steps):
        op.step(ev)
    torch.cuda.synchronize()
    fwd = sum(a.elapsed_time(b) for a, b, _ in ev) / (len(ev) * FRAMES)
    bwd = sum(b.elapsed_time(c) for _, b, c in ev) / (len(ev) * FRAMES)
    mpix = B * H * W / ((fwd + bwd) * FRAMES * 1e-3) / 1e6
    del frames, op
    torch.cuda.empty_cache()
    res = {"Mpix_s": round(mpix, 1), "fwd_kernel_ms": round(fwd, 4), "bwd_kernel_ms": round(bwd, 4),
           "vs_headline": round(mpix / headline_mpix, 3)}
    if radius == _lib.RADIUS_AUTO:
        res["adaptive_radius"] = adaptive_state(lib, H, W)["radius"]
    return res


def warm_up(step, n, radius):
    """n untimed steps.  Under the adaptive radius the first two are each followed by a
    synchronisation: the library picks a shape's radius when a launch is ENQUEUED, from what the
    launches that have FINISHED reported, so n steps enqueued in one go would all run at the start
    radius (14) and the timed region would begin with kernels that never ran."""
    for i in range(n):
        step()
        if radius == _lib.RADIUS_AUTO and i < 2:
            torch.cuda.synchronize()


def run_train(args, rank, world, device, dist, backend):
    lib = _lib.load()  # raises if the HIP library is missing: there is no fallback
    B = args.batch
    frames = make_frames(B, args.sigma, device, seed=103 + rank)
    stream = torch.cuda.current_stream().cuda_stream
    op = Op(lib, frames, B, args.radius, stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    events = []
    warm_up(op.step, args.warmup, args.radius)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        op.step(events)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()

    ms = (t1 - t0) * 1e3 / args.steps
    if dist is not None:
        tt = torch.tensor([ms], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms = float(tt.item())
    if rank != 0:
        return None
    pix_per_step = world * B * H * W
    value = pix_per_step / (ms * 1e-3) / 1e6
    fwd_ms = sum(a.elapsed_time(b) for a, b, _ in events) / (len(events) * FRAMES)
    bwd_ms = sum(b.elapsed_time(c) for _, b, c in events) / (len(events) * FRAMES)
    bwd_gbs = BWD_BYTES * B * H * W / (bwd_ms * 1e-3) / 1e9
    fwd_gbs = FWD_BYTES * B * H * W / (fwd_ms * 1e-3) / 1e9
    traffic, traffic_src = recorded_traffic("synth_bwd_bytes_per_launch", B,
                                            kernel_shape(lib, args.radius, H, W))
    if args.sigma != 1.0:  # (recorded at sigma 1)
        traffic, traffic_src = None, "profiles/pmc_traffic.json holds sigma 1 only"
    line = {
        "metric": "synthesized Mpix/s fwd+bwd (4 frames, 1080p)",
        "value": round(value, 2),
        "unit": "Mpix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "synth op fwd+bwd (gradWeight, gradOffset_y, gradOffset_x), 4 frames x "
                        "1080x1920, k=5 d=1 C=3, fp32",
            "batch_per_gpu": B,
            "offset_regime": f"sigma:{args.sigma:g} (i.i.d. N(0, sigma^2) px per pixel and tap)",
            "tile_radius": args.radius,
            "tile_radius_note": "-2 = adaptive, the package default (what FunctionSynth callers get): the "
                                "library picks 8 / 14 / 24 per shape from its forward kernels' reach "
                                "statistics, no host sync; -1 = fixed 14",
            "adaptive": adaptive_state(lib, H, W) if args.radius == _lib.RADIUS_AUTO else None,
            "sharding": "batch x timestep, no collective",
            "hbm_frac_of_8TBps_whole_step": round(
                value * 1e6 * (FWD_BYTES + BWD_BYTES) * FRAMES / world / 1e9 / HBM_PEAK_GBS, 4),
            "fwd_kernel_ms": round(fwd_ms, 4),
            "fwd_kernel_GBps": round(fwd_gbs, 1),
        },
        "roofline": {
            "kernel": "synth_bwd_tapmajor (fused gradWeight/gradOffset_y/gradOffset_x)"
                      + (f", radius {adaptive_state(lib, H, W)['radius']} (adaptive)"
                         if args.radius == _lib.RADIUS_AUTO else ""),
            "bound": "hbm",
            "achieved": round(bwd_gbs, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(bwd_gbs / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": round(BWD_BYTES * B * H * W),
            "avg_launch_ms": round(bwd_ms, 4),
        },
    }
    if world == 1:
        del frames, op
        torch.cuda.empty_cache()
        if args.regimes:
            line["config"]["other_offset_regimes"] = {
                r: regime_timing(lib, device, B, args.radius, stream, r, 5, value)
                for r in args.regimes.split(",")}
        if not args.no_head:
            line["config"]["chrono_head_fwd_bwd_1080p"] = head_timing(device, args.radius, args.sigma)
            line["config"]["config2_forward_gpu"] = config2_gpu_quick(lib, device, args.radius, stream,
                                                                      args.sigma)
        if not args.no_cpu_baseline:  # reported at N = 1 only
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            line["cpu_baseline"] = cpu_baseline(threads, args.sigma)
    return line


CONFIG2_SHAPE = (16, 256, 448)   # N, H, W of BASELINE.json configs[1]


def config2_fwd_bytes():
    """Algorithmic bytes of ONE forward launch at configs[1]'s shape (324 B/pixel)."""
    n2, h2, w2 = CONFIG2_SHAPE
    return (3 * K_TAPS * 4 + C * 4) * n2 * h2 * w2 + C * 4 * n2 * (h2 + 2 * PAD) * (w2 + 2 * PAD)


def config2_gpu_quick(lib, device, radius, stream, sigma, steps=100, warmup=50):
    """configs[1] on the GPU beside the headline (the full line: --workload config2): 4 forward
    launches of N = 16 x 256 x 448 per step, HIP events.  50 + 100 steps of 0.4 ms: the first ~50 ms
    after a pause run up to 15 % slower (profiles/r3_sustained.log)."""
    n2, h2, w2 = CONFIG2_SHAPE
    frames = make_frames(n2, sigma, device, seed=503, size=(h2, w2), backward=False)
    op = Op(lib, frames, n2, radius, stream, size=(h2, w2))
    for _ in range(warmup):
        for f in frames:
            op.fwd(f)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        for f in frames:
            op.fwd(f)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    gbs = config2_fwd_bytes() * FRAMES / (ms * 1e-3) / 1e9
    del frames, op
    torch.cuda.empty_cache()
    return {"workload": "N=16, 256x448, 4 frames, forward only (BASELINE.json configs[1]), 1 GPU",
            "Mpix_s": round(n2 * h2 * w2 / (ms * 1e-3) / 1e6, 1), "ms_per_step": round(ms, 4),
            "fwd_kernel_ms": round(ms / FRAMES, 4), "GBps": round(gbs, 1),
            "hbm_frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4)}


def run_config2(args, rank, world, device, dist, backend):
    """BASELINE.json configs[1]: "Synth op forward only, Vimeo-90K septuplet shape 448x256, batch 16,
    4 frames, config.py default taps, fp32, 1xMI355X" (operand shapes of model/chrono_synth.py:109-127
    at the dataset's native size, data/preprocessing/vimeo90k_septuplet_process.py:36).  A step = the
    4 per-frame forward launches of one head at N = 16 (2.38 GB of operands per step, so every launch
    streams cold planes: the 4 frames' sets rotate through 9x the Infinity Cache); `value` =
    synthesized pixels (N x 256 x 448 per step) / time.  On N ranks every rank runs its own batch of
    16 (weak scaling, no collective)."""
    lib = _lib.load()
    n2, h2, w2 = CONFIG2_SHAPE
    frames = make_frames(n2, args.sigma, device, seed=503 + rank, size=(h2, w2), backward=False)
    stream = torch.cuda.current_stream().cuda_stream
    op = Op(lib, frames, n2, args.radius, stream, size=(h2, w2))

    def step():
        for f in frames:
            op.fwd(f)

    def barrier():
        if dist is not None:
            dist.barrier()

    warm_up(step, args.warmup, args.radius)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    ms = (t1 - t0) * 1e3 / args.steps
    if dist is not None:
        tt = torch.tensor([ms], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms = float(tt.item())
    if rank != 0:
        return None
    kernel_ms = e0.elapsed_time(e1) / (args.steps * FRAMES)
    fwd_bytes = config2_fwd_bytes()
    gbs = fwd_bytes / (kernel_ms * 1e-3) / 1e9
    value = world * n2 * h2 * w2 / (ms * 1e-3) / 1e6
    line = {
        "metric": "synthesized Mpix/s fwd (4 frames, 448x256, batch 16)",
        "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "config2: synth op forward only, N=16 x 256x448 (Vimeo-90K septuplet), 4 frames "
                        "= 4 launches per step (BASELINE.json configs[1]), k=5 d=1 C=3, fp32",
            "batch_per_gpu": n2, "offset_regime": f"sigma:{args.sigma:g}", "tile_radius": args.radius,
            "tile_radius_note": "-2 = adaptive, the package default (what FunctionSynth callers get); -1 = fixed 14",
            "adaptive": adaptive_state(lib, h2, w2) if args.radius == _lib.RADIUS_AUTO else None,
            "operand_bytes_per_step": FRAMES * fwd_bytes,
            "sharding": "batch, no collective",
        },
        "roofline": {
            "kernel": "synth_fwd_tapmajor (one frame's forward at N = 16)",
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
            "traffic_source": "not recorded for this shape (per-pixel traffic: profiles/pmc_traffic.json, 1080p)",
            "algorithmic_bytes_per_launch": fwd_bytes, "avg_launch_ms": round(kernel_ms, 4),
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        del frames, op
        torch.cuda.empty_cache()
        threads = args.cpu_threads or min(16, os.cpu_count() or 1)
        line["cpu_baseline"] = cpu_baseline_config2(threads, args.sigma)
    return line


def run_interp7(args, rank, world, device, dist, backend):
    """BASELINE.json configs[3]: interpolate mode, one 4-frame 1080p window, 7 timesteps
    (main.py:114-121 runs them one model forward at a time).  sharding.shard_interpolation deals
    whole timesteps when the ranks divide 7 and a row strip of all 7 otherwise (--interp-shard
    timesteps forces whole timesteps: 7 busy ranks + 1 idle on 8 GPUs); a rank runs ITS share as
    one fused-head forward launch."""
    from artemis_amd.sharding import InterpolationShard, shard_interpolation, shard_items
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    TS = 7
    if args.interp_shard == "timesteps":
        sh = InterpolationShard(tuple(i.timestep for i in shard_items(1, TS, rank, world)), 0, H)
    else:
        sh = shard_interpolation(TS, H, rank, world)
    n, Hl = len(sh.timesteps), sh.y1 - sh.y0
    T = FRAMES
    if n and Hl:
        # the window's 4 frames are the same for every timestep and stay whole; weights / offsets /
        # occlusion differ per timestep and exist for this rank's rows only
        frames1, w, a, b, occ, _ = head_operands(n, device, f"sigma:{args.sigma}", 300 + rank)
        frames = [f[:1].expand(n, C, H, W).contiguous() for f in frames1]
        w, a, b = (t[:, :, :, :Hl].contiguous() for t in (w, a, b))
        occ = occ[:, :, :Hl].contiguous()
        out = torch.empty(n, C, Hl, W, device=device)
        ptrs = (ctypes.c_void_p * T)(*[f.data_ptr() for f in frames])

    def step():
        if n and Hl:
            _lib.check(lib.artemis_synth_head_forward(
                ptrs, T, w.data_ptr(), a.data_ptr(), b.data_ptr(), occ.data_ptr(), out.data_ptr(),
                n, Hl, W, H, W, sh.y0 - 2, -2, 0, args.radius, stream), "head_forward")

    def barrier():
        if dist is not None:
            dist.barrier()

    warm_up(step, args.warmup, args.radius)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    ms = (t1 - t0) * 1e3 / args.steps
    if dist is not None:
        tt = torch.tensor([ms], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms = float(tt.item())
    if rank != 0:
        return None
    kernel_ms = e0.elapsed_time(e1) / args.steps
    value = TS * H * W / (ms * 1e-3) / 1e6
    # recorded for one 1080p sample (tools/prof_run.py --head); this launch holds n samples of Hl rows
    traffic, traffic_src = recorded_traffic("head_fwd_bytes_per_launch", 1,
                                            kernel_shape(lib, args.radius, Hl, W, head=True))
    if traffic is not None and args.sigma == 1.0:
        traffic = int(traffic * n * Hl / H)
    else:
        traffic = None
    gbs = HEAD_FWD_BYTES * n * Hl * W / (kernel_ms * 1e-3) / 1e9
    busy = world if sh.y1 - sh.y0 < H else min(world, TS)
    return {
        "metric": "synthesized Mpix/s fwd (interpolate: 7 timesteps, 4 frames, 1080p)",
        "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "interp7: fused ChronoSynth head forward, one 4-frame 1080x1920 window, 7 "
                        "timesteps (BASELINE.json configs[3]), k=5 d=1 C=3, fp32",
            "timesteps_on_rank0": n, "rows_on_rank0": [sh.y0, sh.y1], "busy_ranks": busy,
            "idle_ranks": world - busy,
            "offset_regime": f"sigma:{args.sigma:g}", "tile_radius": args.radius,
            "adaptive": adaptive_state(lib, Hl, W, head=True) if args.radius == _lib.RADIUS_AUTO else None,
            "sharding": ("sharding.shard_items(1, 7, rank, world): whole timesteps in balanced blocks"
                         if args.interp_shard == "timesteps" else
                         "sharding.shard_interpolation(7, 1080, rank, world): whole timesteps when the "
                         "ranks divide 7, else a row strip of all 7 per rank") + ", no collective",
        },
        "roofline": {
            "kernel": "synth_fwd_tapmajor<HEAD> (fused 4-frame head forward), rank 0's launch",
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": round(HEAD_FWD_BYTES * n * Hl * W),
            "avg_launch_ms": round(kernel_ms, 4),
        },
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 20 for train, 200 for config2, 60 for interp7 -- the short "
                         "steps of the latter two need ~100 ms of sustained launches before the rate "
                         "settles, profiles/r3_sustained.log)")
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed steps first (default: 5 / 100 / 20)")
    ap.add_argument("--workload", choices=["train", "config2", "interp7"], default="train",
                    help="train = the headline (op fwd+bwd, 4 frames, 1080p); config2 = configs[1] (op "
                         "forward, N=16 x 256x448, 4 frames); interp7 = configs[3]")
    ap.add_argument("--interp-shard", choices=["auto", "timesteps"], default="auto",
                    help="interp7 on N ranks: auto = sharding.shard_interpolation (row strips of all "
                         "timesteps when N does not divide 7); timesteps = whole timesteps only")
    ap.add_argument("--batch", type=int, default=1, help="samples per GPU (1080p each)")
    ap.add_argument("--sigma", type=float, default=1.0, help="offset std in pixels (headline: 1)")
    ap.add_argument("--radius", type=int, default=_lib.RADIUS_AUTO,
                    help="LDS tile radius passed to the library; -2 = adaptive (the package default), "
                         "-1 = the fixed library default (14)")
    ap.add_argument("--regimes", default="smooth30,sigma:4,pan,sigma:1@14,sigma:1@24,smooth30@14",
                    help="other offset regimes timed beside the headline at N = 1 ('' = none); "
                         "'<regime>@<r>' = at the FIXED tile radius r")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(16, host cpus)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-head", action="store_true", help="skip the fused-head side measurement")
    args = ap.parse_args()
    default_steps, default_warmup = {"train": (20, 5), "config2": (200, 100), "interp7": (60, 20)}[args.workload]
    if args.steps is None:
        args.steps = default_steps
    if args.warmup is None:
        args.warmup = default_warmup

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one rank per GPU; BENCH_DIST_BACKEND=gloo (+ ranks sharing a GPU) only rehearses the
    # multi-rank control flow on a one-GPU box
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    # BENCH_FORCE_DIST=1: initialise the process group for a single rank too (rehearses the RCCL
    # init / barrier / all-reduce calls on a one-GPU box; launch under torch.distributed.run)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    run = {"train": run_train, "config2": run_config2, "interp7": run_interp7}[args.workload]
    line = run(args, rank, world, device, dist, backend)
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
