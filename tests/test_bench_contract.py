"""bench.py's output contract, on the GPU: the one JSON line of a single-GPU run (with `roofline` and
`cpu_baseline`), the multi-rank control flow (two ranks over gloo sharing the card: barrier, MAX over
ranks, whole-job value), the RCCL calls themselves with one forced rank (this box has one GPU; the
N-rank RCCL run is the driver's), and the interp7 workload."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "2", "--warmup", "1", "--no-head", "--regimes", ""]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(args, nproc=0, env=None):
    e = dict(os.environ)
    e.update(env or {})
    if nproc:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "bench.py", *args]
    else:
        cmd = [sys.executable, "bench.py", *args]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def check_line(d, n_gpus):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "Mpix/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    assert "traffic" in r


def test_single_gpu_line_with_cpu_baseline():
    d = run_bench(QUICK)
    check_line(d, 1)
    assert d["scaling"] == "weak" and d["config"]["tile_radius"] == -2      # the package default: adaptive
    # value = pixels of the whole job / time of a step
    assert abs(d["value"] - 1080 * 1920 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mpix/s" and c["value"] > 0
    assert 1 <= c["cores"] <= c["host_cpus"] and "sample" in c


def test_two_ranks_over_gloo_share_the_card():
    d = run_bench(QUICK + ["--gpus", "2", "--no-cpu-baseline"], nproc=2, env={"BENCH_DIST_BACKEND": "gloo"})
    check_line(d, 2)
    # whole-job value: both ranks' pixels over the slowest rank's time
    assert abs(d["value"] - 2 * 1080 * 1920 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3


def test_rccl_calls_with_one_forced_rank():
    d = run_bench(QUICK + ["--gpus", "1", "--no-cpu-baseline"], nproc=1, env={"BENCH_FORCE_DIST": "1"})
    check_line(d, 1)


def test_interp7_two_ranks_take_row_strips_of_all_timesteps():
    d = run_bench(["--workload", "interp7", "--steps", "2", "--warmup", "1", "--gpus", "2", "--no-cpu-baseline"],
                  nproc=2, env={"BENCH_DIST_BACKEND": "gloo"})
    check_line(d, 2)
    c = d["config"]
    assert c["timesteps_on_rank0"] == 7 and c["rows_on_rank0"] == [0, 540] and c["idle_ranks"] == 0
    assert abs(d["value"] - 7 * 1080 * 1920 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3


def test_interp7_workload():
    d = run_bench(["--workload", "interp7", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    check_line(d, 1)
    assert d["scaling"] == "strong" and "interp7" in d["config"]["workload"]
