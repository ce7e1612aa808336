"""Multi-GPU sharding of large ray grids: one process per GPU (torch.distributed,
backend "nccl" == RCCL over xGMI), the read-only BVH replicated on every rank,
grid rows split into contiguous blocks, hit records assembled on rank 0.

There is no exchange during traversal; the only collective is the gather of
results.  It is issued per row chunk with async_op=True so the
copy of chunk c over xGMI overlaps the tracing of chunk c+1; rank 0 receives
each peer over that peer's own link (gather == grouped send/recv in RCCL).

Two exchange formats:
  * records: the 32-byte hit records themselves travel (537 MB for a 4096^2 view);
  * tokens (an `expander` is given): ranks trace with MRT_FLAG_TOKEN_OUT, 4 bytes per
    ray travel, and rank 0 rebuilds the records of every chunk with
    mrt_expand_grid_tokens on a side stream while the next chunk is traced.  The
    records are bit-identical (tests/test_parity_gpu.py); an xGMI link moves about
    a tenth of what a GPU traces, so this is what keeps the gather off the
    critical path.

The reference is single-process / single-device (no collective anywhere in
src/); this module is the MI355X-native addition named by BASELINE.json.
"""
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

HIT_BYTES = 32    # mrt_hit32 / GPUIntersectionPacked (src/api/gpu_types.h:87-92)
TOKEN_BYTES = 4   # MRT_FLAG_TOKEN_OUT on a flat scene; a two-level scene writes 8 ({triangle, instance}): pass token_bytes = ctx.token_bytes()


def row_block(rank: int, world: int, rows: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: y in [r*H/N, (r+1)*H/N)."""
    return rank * rows // world, (rank + 1) * rows // world


def chunk_bounds(y0: int, y1: int, chunks: int) -> List[Tuple[int, int]]:
    n = y1 - y0
    chunks = max(1, min(chunks, n)) if n > 0 else 1
    return [(y0 + c * n // chunks, y0 + (c + 1) * n // chunks) for c in range(chunks)]


class _Done:
    def wait(self):
        return True


def _gather(out: torch.Tensor, dst, group):
    """dist.gather(async_op=True) to rank 0.  With a gloo group and device tensors -- several ranks
    rehearsing on ONE GPU, which RCCL refuses -- the same exchange goes through host memory."""
    if out.is_cuda and dist.get_backend(group) == "gloo":
        host = out.cpu()
        parts = [torch.empty_like(host) for _ in dst] if dst is not None else None
        dist.gather(host, parts, dst=0, group=group)
        if dst is not None:
            for d, p in zip(dst, parts):
                d.copy_(p)
        return _Done()
    return dist.gather(out, dst, dst=0, group=group, async_op=True)


class _SideStream:
    """Where rank 0 expands tokens: a second HIP stream on a GPU (expansion of chunk c runs
    beside the tracing of chunk c+1), nothing on the CPU (gloo tests)."""

    def __init__(self, device: torch.device):
        self.stream = torch.cuda.Stream(device) if device.type == "cuda" else None

    def run_after(self, works, fn):
        """Call fn(raw_stream) ordered after the collectives `works`."""
        if self.stream is None:
            for w in works:
                w.wait()
            fn(None)
            return
        self.stream.wait_stream(torch.cuda.current_stream(self.stream.device))  # and after what the main stream holds
        with torch.cuda.stream(self.stream):
            for w in works:
                w.wait()  # the side stream waits for the collective; the host does not
            fn(self.stream.cuda_stream)

    def join(self):
        if self.stream is not None:
            torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)

    def mark(self):
        """An event after everything queued on the side stream so far (None on the CPU)."""
        if self.stream is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def wait_mark(self, ev):
        if ev is not None:
            torch.cuda.current_stream(self.stream.device).wait_event(ev)


class ShardedGrid:
    """Rows [0, rows) of a `width`-wide hit image, sharded over the ranks of `group`.

    tracer(y0, y1, out) must fill `out` (a uint8 tensor view on this rank's device) with
    the results of rows [y0, y1): 32-byte hit records, or hit tokens (token_bytes each: 4, or 8 for a
    two-level scene) when an `expander` is given.  expander(y0, y1, tokens, hits, stream) (rank 0 only) rebuilds the
    records of rows [y0, y1) on `stream`.  On the GPU these wrap mrt_cast_grid /
    mrt_expand_grid_tokens on the rank's context; the CPU tests inject pattern generators
    to exercise the exchange path under gloo.
    """

    def __init__(self, width: int, rows: int, tracer: Callable[[int, int, torch.Tensor], None],
                 device: torch.device, chunks: int = 4, group=None, gather: bool = True,
                 expander: Optional[Callable] = None, token_bytes: int = TOKEN_BYTES):
        self.width, self.rows, self.tracer, self.device = width, rows, tracer, device
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.y0, self.y1 = row_block(self.rank, self.world, rows)
        self.chunks = chunk_bounds(self.y0, self.y1, chunks)
        self.gather = gather and self.world > 1
        self.expander = expander if self.gather else None
        self.token_mode = self.expander is not None  # what the tracer must write: tokens or records
        # bytes per row of what is traced / exchanged, and of the assembled image
        self.row_bytes = width * HIT_BYTES
        self.xrow_bytes = width * (token_bytes if self.expander else HIT_BYTES)
        self.local = torch.empty((self.y1 - self.y0) * self.xrow_bytes, dtype=torch.uint8, device=device)
        # rank 0 holds the whole image; every rank's block has its own slot
        self.image: Optional[torch.Tensor] = None
        self.staged: Optional[torch.Tensor] = None  # token mode: the gathered tokens of the whole grid
        if self.rank == 0 and self.gather:
            self.image = torch.empty(rows * self.row_bytes, dtype=torch.uint8, device=device)
            self.staged = torch.empty(rows * self.xrow_bytes, dtype=torch.uint8, device=device) if self.expander else self.image
        self.side = _SideStream(device) if self.expander and self.rank == 0 else None
        # gather needs equally sized pieces: chunk c of every rank must have the same row count
        self.uniform = all(row_block(r, self.world, rows)[1] - row_block(r, self.world, rows)[0] == self.y1 - self.y0
                           for r in range(self.world))

    @staticmethod
    def _rows(buf: torch.Tensor, row_bytes: int, y0: int, y1: int, base_row: int) -> torch.Tensor:
        return buf[(y0 - base_row) * row_bytes:(y1 - base_row) * row_bytes]

    def _expand(self, works, spans):
        """Rank 0, token mode: once `works` are done rebuild the records of the row spans."""
        def fn(stream):
            for (a, b) in spans:
                if b > a:
                    self.expander(a, b, self._rows(self.staged, self.xrow_bytes, a, b, 0),
                                  self._rows(self.image, self.row_bytes, a, b, 0), stream)
        self.side.run_after(works, fn)

    def step(self) -> Optional[torch.Tensor]:
        """Trace this rank's rows and assemble everything on rank 0.  Returns the
        full image (uint8, rows*width*32 bytes) on rank 0, None elsewhere; with
        world == 1 returns the local block."""
        pending = []
        for (c0, c1) in self.chunks:
            out = self._rows(self.local, self.xrow_bytes, c0, c1, self.y0)
            self.tracer(c0, c1, out)
            if not self.gather:
                continue
            if self.uniform:
                dst_list, spans = None, []
                if self.rank == 0:
                    for r in range(self.world):
                        ry0, _ = row_block(r, self.world, self.rows)
                        spans.append((ry0 + (c0 - self.y0), ry0 + (c1 - self.y0)))
                    dst_list = [self._rows(self.staged, self.xrow_bytes, a, b, 0) for (a, b) in spans]
                work = _gather(out, dst_list, self.group)
                if self.side is not None:
                    self._expand([work], spans)
                else:
                    pending.append(work)
            else:  # ragged blocks: point-to-point
                if self.rank == 0:
                    self._rows(self.staged, self.xrow_bytes, c0, c1, 0).copy_(out)
                    if self.side is not None:
                        self._expand([], [(c0, c1)])
                else:
                    pending.append(dist.isend(out, dst=0, group=self.group))
        if self.gather and not self.uniform and self.rank == 0:
            for r in range(1, self.world):
                ry0, ry1 = row_block(r, self.world, self.rows)
                for (c0, c1) in chunk_bounds(ry0, ry1, len(self.chunks)):
                    work = dist.irecv(self._rows(self.staged, self.xrow_bytes, c0, c1, 0), src=r, group=self.group)
                    if self.side is not None:
                        self._expand([work], [(c0, c1)])
                    else:
                        pending.append(work)
        for w in pending:
            w.wait()
        if self.side is not None:
            self.side.join()
        if not self.gather:
            return self.local
        return self.image if self.rank == 0 else None


class ShardedViews:
    """N independent views (one full grid per rank), assembled on rank 0: the weak-scaling
    workload of bench.py.  Same exchange as ShardedGrid with one block per rank;
    expander(view, y0, y1, tokens, hits, stream) rebuilds rows [y0, y1) of rank `view`'s image.

    depth = 2 pipelines frames: step() returns once frame k is traced and its gather is
    queued, so the exchange (and rank 0's expansion) of frame k runs beside the tracing of
    frame k+1 on double-buffered exchange buffers; finish() waits for everything queued.
    With depth = 1 step() finishes the frame itself (chunks > 1 then overlaps the copy of
    chunk c with the tracing of chunk c+1 inside the frame)."""

    def __init__(self, width: int, rows: int, tracer: Callable[[int, int, torch.Tensor], None],
                 device: torch.device, chunks: int = 4, group=None, gather: bool = True, force_gather: bool = False,
                 expander: Optional[Callable] = None, depth: int = 1, token_bytes: int = TOKEN_BYTES):
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.width, self.rows, self.tracer, self.group = width, rows, tracer, group
        self.chunks = chunk_bounds(0, rows, chunks)
        # force_gather: run the exchange even with one rank (rehearsal of the N > 1 path on one GPU)
        self.gather = gather and (self.world > 1 or (force_gather and dist.is_initialized()))
        self.expander = expander if self.gather else None
        self.token_mode = self.expander is not None  # what the tracer must write: tokens or records
        self.depth = max(1, min(int(depth), 2)) if self.gather else 1
        self.row_bytes = width * HIT_BYTES
        self.xrow_bytes = width * (token_bytes if self.expander else HIT_BYTES)
        # what this rank traces into / sends from, one buffer per frame in flight
        self.local = torch.empty((self.depth, rows * self.xrow_bytes), dtype=torch.uint8, device=device)
        self.images = None   # rank 0: [frame slot][view] records (token mode: one slot, written in frame order)
        self.staged = None   # rank 0: [frame slot][view] what arrived
        if self.rank == 0 and self.gather:
            self.staged = torch.empty((self.depth, self.world, rows * self.xrow_bytes), dtype=torch.uint8, device=device)
            self.images = torch.empty((1, self.world, rows * self.row_bytes), dtype=torch.uint8, device=device) \
                if self.expander else self.staged
        self.side = _SideStream(device) if self.expander and self.rank == 0 else None
        self.frame = 0
        self.in_flight = [[] for _ in range(self.depth)]  # collectives still using slot b
        self.expanded = [None] * self.depth               # rank 0: event after the expansion out of slot b

    def _drain(self, b: int):
        for w in self.in_flight[b]:
            w.wait()
        self.in_flight[b] = []

    def step(self):
        """Trace one frame and queue its exchange.  Returns this rank's records (no gather), or on
        rank 0 the [view] images of this frame -- complete after finish() when depth > 1."""
        b = self.frame % self.depth
        self.frame += 1
        self._drain(b)              # the exchange that last used this slot (frame - depth) is done with it
        if self.side is not None:   # ... and so is rank 0's expansion out of staged[b] (not the newer
            self.side.wait_mark(self.expanded[b])  # frame's, which may still be waiting for its tokens)
        local = self.local[b]
        for (c0, c1) in self.chunks:
            out = local[c0 * self.xrow_bytes:c1 * self.xrow_bytes]
            self.tracer(c0, c1, out)
            if not self.gather:
                continue
            dst = None
            if self.rank == 0:
                dst = [self.staged[b, r, c0 * self.xrow_bytes:c1 * self.xrow_bytes] for r in range(self.world)]
            work = _gather(out, dst, self.group)
            self.in_flight[b].append(work)
            if self.side is None:
                continue

            def fn(stream, c0=c0, c1=c1, b=b):
                for r in range(self.world):
                    self.expander(r, c0, c1, self.staged[b, r, c0 * self.xrow_bytes:c1 * self.xrow_bytes],
                                  self.images[0, r, c0 * self.row_bytes:c1 * self.row_bytes], stream)
            self.side.run_after([work], fn)
        if self.side is not None:
            self.expanded[b] = self.side.mark()
        if self.depth == 1:
            self.finish()
        if not self.gather:
            return local
        if self.rank != 0:
            return None
        return self.images[0] if self.expander else self.images[b]

    def finish(self):
        """Wait (stream-wise) for every queued exchange and expansion."""
        for b in range(self.depth):
            self._drain(b)
        if self.side is not None:
            self.side.join()


def balanced_spans(world: int, rows: int, root_share: float) -> List[List[Tuple[int, int, int]]]:
    """spans[r] = the (view, y0, y1) row ranges rank r traces when rank 0 keeps only the first
    `root_share` of its own view and the other ranks share the rest of it (each still traces
    its whole own view).  root_share = 1: every rank traces exactly its own view."""
    if world == 1 or root_share >= 1.0:
        return [[(r, 0, rows)] for r in range(world)]
    keep = max(1, min(rows, int(round(rows * root_share))))
    rest = rows - keep
    spans = [[(0, 0, keep)]]
    for r in range(1, world):
        a, b = keep + rest * (r - 1) // (world - 1), keep + rest * r // (world - 1)
        spans.append([(r, 0, rows)] + ([(0, a, b)] if b > a else []))
    return spans


class BalancedViews:
    """The weak-scaling workload of ShardedViews (one view per rank, all views assembled as
    records on rank 0, 4-byte tokens on the wire, frames pipelined `depth` deep) with rank 0's
    extra duty priced in: rank 0 rebuilds everybody's records (0.14 ms per 4096^2 view against
    2.2 ms of tracing), so with 8 ranks it would finish a third later than the others.  It
    therefore keeps only `root_share` of its own view's rows and the peers trace the rest of
    that view besides their own (balanced_spans); every rank sends one equally sized token
    payload per frame (its spans back to back, padded), rank 0 expands each span into the
    image of the view it belongs to.

    tracer(view, y0, y1, out): tokens of rows [y0, y1) of `view` into out (uint8 view);
    expander(view, y0, y1, tokens, hits, stream): rank 0, records of those rows on `stream`."""

    def __init__(self, width: int, rows: int, tracer: Callable, expander: Callable, device: torch.device,
                 root_share: float = 1.0, group=None, depth: int = 2, token_bytes: int = TOKEN_BYTES):
        if not dist.is_initialized():
            raise RuntimeError("BalancedViews needs an initialised process group (use ShardedViews without one)")
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.width, self.rows, self.tracer, self.expander, self.group = width, rows, tracer, expander, group
        self.token_mode, self.gather = True, True
        self.depth = max(1, min(int(depth), 2))
        self.spans = balanced_spans(self.world, rows, root_share)
        self.pay_rows = max(sum(y1 - y0 for (_, y0, y1) in s) for s in self.spans)
        self.row_bytes, self.xrow_bytes = width * HIT_BYTES, width * token_bytes
        self.local = torch.zeros((self.depth, self.pay_rows * self.xrow_bytes), dtype=torch.uint8, device=device)
        self.images = self.staged = self.side = None
        if self.rank == 0:
            self.staged = torch.empty((self.depth, self.world, self.pay_rows * self.xrow_bytes), dtype=torch.uint8, device=device)
            self.images = torch.empty((self.world, rows * self.row_bytes), dtype=torch.uint8, device=device)
            self.side = _SideStream(device)
        self.frame = 0
        self.in_flight = [None] * self.depth
        self.expanded = [None] * self.depth

    def step(self):
        """Trace this rank's spans of one frame and queue the exchange.  Rank 0 gets the [view]
        images (complete after finish() when depth > 1), the others None."""
        b = self.frame % self.depth
        self.frame += 1
        if self.in_flight[b] is not None:
            self.in_flight[b].wait()          # the exchange that last used this slot is done with it
        if self.side is not None:
            self.side.wait_mark(self.expanded[b])
        local, off = self.local[b], 0
        for (view, y0, y1) in self.spans[self.rank]:
            self.tracer(view, y0, y1, local[off * self.xrow_bytes:(off + y1 - y0) * self.xrow_bytes])
            off += y1 - y0
        dst = [self.staged[b, r] for r in range(self.world)] if self.rank == 0 else None
        work = _gather(local, dst, self.group)
        self.in_flight[b] = work
        if self.rank == 0:
            def fn(stream, b=b):
                for r in range(self.world):
                    o = 0
                    for (view, y0, y1) in self.spans[r]:
                        self.expander(view, y0, y1, self.staged[b, r, o * self.xrow_bytes:(o + y1 - y0) * self.xrow_bytes],
                                      self.images[view, y0 * self.row_bytes:y1 * self.row_bytes], stream)
                        o += y1 - y0
            self.side.run_after([work], fn)
            self.expanded[b] = self.side.mark()
        if self.depth == 1:
            self.finish()
        return self.images if self.rank == 0 else None

    def finish(self):
        for b in range(self.depth):
            if self.in_flight[b] is not None:
                self.in_flight[b].wait()
                self.in_flight[b] = None
        if self.side is not None:
            self.side.join()
