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


def _pattern(y0, y1, width, view=0, salt=0):
    """32 bytes per ray that encode (view, y, x); `salt` tells frames apart."""
    ys = np.arange(y0, y1, dtype=np.uint32)[:, None]
    xs = np.arange(width, dtype=np.uint32)[None, :]
    rec = np.zeros((y1 - y0, width, 8), dtype=np.uint32)
    rec[..., 0] = ys
    rec[..., 1] = xs
    rec[..., 2] = ys * 65599 + xs * 31 + view + salt
    rec[..., 7] = view
    return rec


def _worker(rank, world, port, width, rows, chunks, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    calls = []
    opts = kind.split("-")
    kind = opts[0]
    tokens = "tokens" in opts  # 4 bytes per ray travel, rank 0 rebuilds the 32-byte records
    depth = 2 if "pipe" in opts else 1  # frames pipelined: the exchange of frame k beside the tracing of frame k+1
    frame = [0]

    def tracer(y0, y1, out):
        calls.append((y0, y1))
        rec = _pattern(y0, y1, width, rank if kind == "views" else 0, 1000 * frame[0])
        src = np.ascontiguousarray(rec[..., 2]) if tokens else rec  # the token is word 2 of the record
        out.copy_(torch.from_numpy(src.view(np.uint8).reshape(-1)))

    def rebuild(view, y0, y1, tok, hits, stream):
        assert stream is None  # no side stream on the CPU
        rec = _pattern(y0, y1, width, view)  # salt unknown here: word 2 is overwritten with what arrived
        rec[..., 2] = tok.numpy().view(np.uint32).reshape(y1 - y0, width)  # what travelled, not what we expect
        hits.copy_(torch.from_numpy(rec.view(np.uint8).reshape(-1)))

    if kind == "grid":
        job = sharded.ShardedGrid(width, rows, tracer, dev, chunks=chunks,
                                  expander=(lambda y0, y1, t, h, s: rebuild(0, y0, y1, t, h, s)) if tokens else None)
        want_rows = sharded.row_block(rank, world, rows)
    else:
        job = sharded.ShardedViews(width, rows, tracer, dev, chunks=chunks, expander=rebuild if tokens else None, depth=depth)
        want_rows = (0, rows)
    assert job.token_mode == tokens
    for k in range(3):  # three steps: buffers (two sets of them when pipelined) are reused
        calls.clear()
        frame[0] = k
        img = job.step()
    if depth > 1:
        job.finish()
    ok = calls[0][0] == want_rows[0] and calls[-1][1] == want_rows[1] and \
        all(calls[i][1] == calls[i + 1][0] for i in range(len(calls) - 1))
    if rank == 0:
        got = img.numpy()
        if kind == "grid":
            want = _pattern(0, rows, width, 0, 2000).view(np.uint8).reshape(-1)
            ok = ok and np.array_equal(got, want)
        else:
            for r in range(world):
                ok = ok and np.array_equal(got[r], _pattern(0, rows, width, r, 2000).view(np.uint8).reshape(-1))
    else:
        ok = ok and img is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,width,rows,chunks", [("grid", 64, 48, 4), ("grid", 33, 37, 3), ("grid", 16, 2, 8),
                                                    ("views", 32, 24, 4), ("views", 8, 5, 2),
                                                    ("grid-tokens", 64, 48, 4), ("grid-tokens", 33, 37, 3),
                                                    ("views-tokens", 32, 24, 4), ("views-tokens", 8, 5, 2),
                                                    ("views-pipe", 32, 24, 1), ("views-tokens-pipe", 32, 24, 1),
                                                    ("views-tokens-pipe", 8, 5, 2)])
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


def _balanced_worker(rank, world, port, width, rows, share, depth, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frame = [0]
    traced = []

    def tracer(view, y0, y1, out):
        traced.append((view, y0, y1))
        rec = _pattern(y0, y1, width, view, 1000 * frame[0])
        out.copy_(torch.from_numpy(np.ascontiguousarray(rec[..., 2]).view(np.uint8).reshape(-1)))

    def rebuild(view, y0, y1, tok, hits, stream):
        rec = _pattern(y0, y1, width, view)
        rec[..., 2] = tok.numpy().view(np.uint32).reshape(y1 - y0, width)
        hits.copy_(torch.from_numpy(rec.view(np.uint8).reshape(-1)))

    job = sharded.BalancedViews(width, rows, tracer, rebuild, torch.device("cpu"), root_share=share, depth=depth)
    for k in range(3):
        traced.clear()
        frame[0] = k
        img = job.step()
    job.finish()
    ok = traced == job.spans[rank]
    if rank == 0:
        got = img.numpy()
        for v in range(world):
            ok = ok and np.array_equal(got[v], _pattern(0, rows, width, v, 2000).view(np.uint8).reshape(-1))
    else:
        ok = ok and img is None
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("width,rows,share,depth", [(16, 20, 0.6, 2), (8, 7, 0.3, 1), (8, 9, 1.0, 2)])
def test_balanced_views_world_size_2(width, rows, share, depth):
    """Rank 0 keeps `share` of its view, rank 1 traces the rest besides its own view; rank 0
    still ends up with every view complete."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_balanced_worker, args=(r, 2, port, width, rows, share, depth, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}


def test_balanced_spans_cover_every_view_once():
    for world in (1, 2, 3, 8):
        for rows in (1, 5, 4096):
            for share in (1.0, 0.55, 0.01):
                spans = sharded.balanced_spans(world, rows, share)
                assert len(spans) == world
                cover = {v: [] for v in range(world)}
                for s in spans:
                    for (v, a, b) in s:
                        assert 0 <= a < b <= rows
                        cover[v].append((a, b))
                for v, parts in cover.items():
                    parts.sort()
                    assert parts[0][0] == 0 and parts[-1][1] == rows
                    assert all(parts[i][1] == parts[i + 1][0] for i in range(len(parts) - 1))
