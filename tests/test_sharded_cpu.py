"""The N > 1 path (row sharding + gather of hit records, messyerraytracer_amd/sharded.py)
under gloo with world_size 2 on the CPU.  The per-rank trace is replaced by a
deterministic pattern so that the exchange, chunking and placement are what is tested;
the kernels themselves are covered by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from messyerraytracer_amd import sharded


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pattern(y0, y1, width, view=0):
    """32 bytes per ray that encode (view, y, x)."""
    ys = np.arange(y0, y1, dtype=np.uint32)[:, None]
    xs = np.arange(width, dtype=np.uint32)[None, :]
    rec = np.zeros((y1 - y0, width, 8), dtype=np.uint32)
    rec[..., 0] = ys
    rec[..., 1] = xs
    rec[..., 2] = ys * 65599 + xs * 31 + view
    rec[..., 7] = view
    return rec


def _worker(rank, world, port, width, rows, chunks, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    calls = []

    def tracer(y0, y1, out):
        calls.append((y0, y1))
        out.copy_(torch.from_numpy(_pattern(y0, y1, width, rank if kind == "views" else 0).view(np.uint8).reshape(-1)))

    if kind == "grid":
        job = sharded.ShardedGrid(width, rows, tracer, dev, chunks=chunks)
        want_rows = sharded.row_block(rank, world, rows)
    else:
        job = sharded.ShardedViews(width, rows, tracer, dev, chunks=chunks)
        want_rows = (0, rows)
    for _ in range(2):  # two steps: buffers are reused
        calls.clear()
        img = job.step()
    ok = calls[0][0] == want_rows[0] and calls[-1][1] == want_rows[1] and \
        all(calls[i][1] == calls[i + 1][0] for i in range(len(calls) - 1))
    if rank == 0:
        got = img.numpy()
        if kind == "grid":
            want = _pattern(0, rows, width).view(np.uint8).reshape(-1)
            ok = ok and np.array_equal(got, want)
        else:
            for r in range(world):
                ok = ok and np.array_equal(got[r], _pattern(0, rows, width, r).view(np.uint8).reshape(-1))
    else:
        ok = ok and img is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,width,rows,chunks", [("grid", 64, 48, 4), ("grid", 33, 37, 3), ("grid", 16, 2, 8),
                                                    ("views", 32, 24, 4), ("views", 8, 5, 2)])
def test_world_size_2_gather(kind, width, rows, chunks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, width, rows, chunks, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_row_blocks_partition_the_grid():
    for world in (1, 2, 3, 4, 8):
        for rows in (1, 7, 8, 4096, 8192, 1000):
            blocks = [sharded.row_block(r, world, rows) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == rows
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            for y0, y1 in blocks:
                cs = sharded.chunk_bounds(y0, y1, 4)
                assert cs[0][0] == y0 and cs[-1][1] == y1
                assert all(cs[i][1] == cs[i + 1][0] for i in range(len(cs) - 1))


def test_single_process_is_a_plain_loop():
    calls = []

    def tracer(y0, y1, out):
        calls.append((y0, y1))
        out.fill_(7)

    job = sharded.ShardedGrid(8, 10, tracer, torch.device("cpu"), chunks=3)
    img = job.step()
    assert calls == sharded.chunk_bounds(0, 10, 3) and img.numel() == 10 * 8 * 32 and bool((img == 7).all())
