"""Multi-GPU path on CPU: the batch x timestep partition (host logic) and a world_size-2 gloo run in
which each rank computes its own shard (with the CPU oracle standing in for the device op -- the
product op has no CPU path) and the gathered result equals the single-process one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from artemis_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(0, 1), (1, 8), (7, 8), (8, 8), (9, 8), (64, 8), (16, 3), (5, 2)])
def test_shard_bounds_partition(n, world):
    covered = []
    sizes = []
    for r in range(world):
        b, e = sharding.shard_bounds(n, r, world)
        assert 0 <= b <= e <= n
        covered += list(range(b, e))
        sizes.append(e - b)
    assert covered == list(range(n))           # disjoint, ordered, complete
    assert max(sizes) - min(sizes) <= 1         # balanced
    assert sizes == sorted(sizes, reverse=True)


def test_shard_bounds_errors():
    with pytest.raises(ValueError):
        sharding.shard_bounds(4, 2, 2)
    with pytest.raises(ValueError):
        sharding.shard_bounds(-1, 0, 2)


def test_interpolate_config_7_timesteps_on_8_gpus():
    # BASELINE.json configs[3]: one 4-frame window, 7 timesteps, 8 GPUs -> one timestep per GPU
    items = [sharding.shard_items(1, 7, r, 8) for r in range(8)]
    assert [len(i) for i in items] == [1] * 7 + [0]
    assert [i[0].timestep for i in items[:7]] == list(range(7))
    # train config: batch 64 over 8 GPUs -> 8 samples each
    assert all(len(sharding.shard_items(64, 1, r, 8)) == 8 for r in range(8))


@pytest.mark.parametrize("world", [1, 2, 3, 4, 7, 8, 14])
def test_interpolation_shards_cover_every_pixel_of_every_timestep_once(world):
    TS, Hh = 7, 1080
    seen = np.zeros((TS, Hh), np.int32)
    loads = []
    for r in range(world):
        sh = sharding.shard_interpolation(TS, Hh, r, world)
        for t in sh.timesteps:
            seen[t, sh.y0:sh.y1] += 1
        loads.append(len(sh.timesteps) * (sh.y1 - sh.y0))
    assert (seen == 1).all()
    # whole timesteps when the ranks divide them, else a row strip of all of them: balanced to a row
    if TS % world == 0:
        assert all(sharding.shard_interpolation(TS, Hh, r, world).y1 == Hh for r in range(world))
    assert max(loads) - min(loads) <= TS
    assert max(loads) * world <= TS * Hh * 1.01 + TS * world      # >= 99 % balanced at 1080 rows


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_interpolation_row_strips_reproduce_the_whole_frame_on_gpu(world):
    """The strips of sharding.shard_interpolation through the fused head's frame geometry: the
    ranks' outputs, stacked, are the single-launch result bit for bit (offsets within the window)."""
    import ctypes
    from artemis_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    TS, T, Hh, Ww = 7, 4, 150, 200
    frames = [torch.rand(1, 3, Hh, Ww, device=dev, generator=g).expand(TS, 3, Hh, Ww).contiguous() for _ in range(T)]
    w = torch.softmax(torch.randn(TS, T, 25, Hh, Ww, device=dev, generator=g), 2)
    a, b = (torch.randn(TS, T, 25, Hh, Ww, device=dev, generator=g) * 1.5 for _ in range(2))
    occ = torch.softmax(torch.randn(TS, T, Hh, Ww, device=dev, generator=g), 1)
    ptrs = (ctypes.c_void_p * T)(*[f.data_ptr() for f in frames])
    st = torch.cuda.current_stream().cuda_stream

    def head(wt, at, bt, ot, n, rows, y0):
        out = torch.empty(n, 3, rows, Ww, device=dev)
        _lib.check(lib.artemis_synth_head_forward(ptrs, T, wt.data_ptr(), at.data_ptr(), bt.data_ptr(), ot.data_ptr(),
                                                  out.data_ptr(), n, rows, Ww, Hh, Ww, y0 - 2, -2, 0, -1, st), "head")
        return out

    full = head(w, a, b, occ, TS, Hh, 0)
    parts = []
    for r in range(world):
        sh = sharding.shard_interpolation(TS, Hh, r, world)
        assert sh.timesteps == tuple(range(TS))
        sl = slice(sh.y0, sh.y1)
        parts.append(head(w[:, :, :, sl].contiguous(), a[:, :, :, sl].contiguous(), b[:, :, :, sl].contiguous(),
                          occ[:, :, sl].contiguous(), TS, sh.y1 - sh.y0, sh.y0))
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts, 2), full)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_interpolation_row_strips_on_a_steep_flow_agree_to_rounding(world):
    """The documented other half: on large steep motion (the
    bench's smooth30 field) windows are re-centred and staged again per TILE, tile origins are
    relative to the strip, so the ranks' strips equal the single launch to the forward tolerance,
    not bit for bit."""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from artemis_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(6)
    TS, T, Hh, Ww = 2, 4, 400, 640
    frames = [torch.rand(1, 3, Hh, Ww, device=dev, generator=g).expand(TS, 3, Hh, Ww).contiguous() for _ in range(T)]
    w = torch.softmax(torch.randn(TS, T, 25, Hh, Ww, device=dev, generator=g), 2)
    ab = [bench.make_offsets("smooth30", (TS, 25, Hh, Ww), dev, g) for _ in range(T)]
    a = torch.stack([p[0] for p in ab], 1).contiguous()
    b = torch.stack([p[1] for p in ab], 1).contiguous()
    occ = torch.softmax(torch.randn(TS, T, Hh, Ww, device=dev, generator=g), 1)
    ptrs = (ctypes.c_void_p * T)(*[f.data_ptr() for f in frames])
    st = torch.cuda.current_stream().cuda_stream

    def head(wt, at, bt, ot, n, rows, y0):
        out = torch.empty(n, 3, rows, Ww, device=dev)
        _lib.check(lib.artemis_synth_head_forward(ptrs, T, wt.data_ptr(), at.data_ptr(), bt.data_ptr(), ot.data_ptr(),
                                                  out.data_ptr(), n, rows, Ww, Hh, Ww, y0 - 2, -2, 0, -1, st), "head")
        return out

    full = head(w, a, b, occ, TS, Hh, 0)
    parts = []
    for r in range(world):
        sh = sharding.shard_interpolation(7, Hh, r, world)                # (the row split of 7 timesteps)
        sl = slice(sh.y0, sh.y1)
        parts.append(head(w[:, :, :, sl].contiguous(), a[:, :, :, sl].contiguous(), b[:, :, :, sl].contiguous(),
                          occ[:, :, sl].contiguous(), TS, sh.y1 - sh.y0, sh.y0))
    torch.cuda.synchronize()
    got = torch.cat(parts, 2)
    assert got.shape == full.shape
    assert float((got - full).abs().max()) <= 1e-5 * float(full.abs().max())


def test_row_strips_cover_frame():
    strips = sharding.row_strips(2160, 8, halo=9)
    assert strips[0][0] == 0 and strips[-1][1] == 2160
    for (y0, y1, lo, hi), nxt in zip(strips, strips[1:] + [None]):
        assert y1 - y0 == 270
        assert lo <= y0 and hi >= min(y1 + 5, 2164)
        assert 0 <= lo and hi <= 2164
        if nxt:
            assert nxt[0] == y1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import make_operands
        from oracle import synth_oracle as oracle
        N = 5  # ragged over 2 ranks: 3 + 2
        inp, w, oy, ox, _ = make_operands(N, 3, 9, 70, regime="gauss1", seed=42)
        mine = sharding.shard_batch([torch.from_numpy(a) for a in (inp, w, oy, ox)], rank, world)
        b, e = sharding.shard_bounds(N, rank, world)
        assert mine[0].shape[0] == e - b
        local = torch.from_numpy(oracle.forward(*[t.numpy() for t in mine], 1))
        full = sharding.gather_outputs(local, N)
        ref = oracle.forward(inp, w, oy, ox, 1)
        ok = np.array_equal(full.numpy(), ref)
        # the timing reduction bench.py uses: MAX over ranks
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t) == float(world)
        dist.barrier()
        if rank == 0:
            with open(out_path, "w") as f:
                f.write("ok" if ok else "mismatch")
        assert ok
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shards_reassemble(tmp_path):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    assert out.read_text() == "ok"
