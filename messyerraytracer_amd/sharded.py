"""Multi-GPU sharding of large ray grids: one process per GPU (torch.distributed,
backend "nccl" == RCCL over xGMI), the read-only BVH replicated on every rank,
grid rows split into contiguous blocks, 32-byte hit records gathered to rank 0.

There is no exchange during traversal; the only collective is the gather of
results.  It is issued per row chunk with async_op=True so the
copy of chunk c over xGMI overlaps the tracing of chunk c+1; rank 0 receives
each peer over that peer's own link (gather == grouped send/recv in RCCL).

The reference is single-process / single-device (no collective anywhere in
src/); this module is the MI355X-native addition named by BASELINE.json.
"""
from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

HIT_BYTES = 32  # mrt_hit32 / GPUIntersectionPacked (src/api/gpu_types.h:87-92)


def row_block(rank: int, world: int, rows: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: y in [r*H/N, (r+1)*H/N)."""
    return rank * rows // world, (rank + 1) * rows // world


def chunk_bounds(y0: int, y1: int, chunks: int) -> List[Tuple[int, int]]:
    n = y1 - y0
    chunks = max(1, min(chunks, n)) if n > 0 else 1
    return [(y0 + c * n // chunks, y0 + (c + 1) * n // chunks) for c in range(chunks)]


class ShardedGrid:
    """Rows [0, rows) of a `width`-wide hit image, sharded over the ranks of `group`.

    tracer(y0, y1, out) must fill `out` (a uint8 tensor view of (y1-y0)*width*32
    bytes on this rank's device) with the hit records of rows [y0, y1).  On the
    GPU it wraps mrt_cast_grid / mrt_cast on the rank's context; the CPU tests
    inject a pattern generator to exercise the exchange path under gloo.
    """

    def __init__(self, width: int, rows: int, tracer: Callable[[int, int, torch.Tensor], None],
                 device: torch.device, chunks: int = 4, group=None, gather: bool = True):
        self.width, self.rows, self.tracer, self.device = width, rows, tracer, device
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.y0, self.y1 = row_block(self.rank, self.world, rows)
        self.chunks = chunk_bounds(self.y0, self.y1, chunks)
        self.row_bytes = width * HIT_BYTES
        self.local = torch.empty((self.y1 - self.y0) * self.row_bytes, dtype=torch.uint8, device=device)
        self.gather = gather and self.world > 1
        # rank 0 holds the whole image; every rank's block has its own slot
        self.image: Optional[torch.Tensor] = None
        if self.rank == 0 and self.gather:
            self.image = torch.empty(rows * self.row_bytes, dtype=torch.uint8, device=device)
        # gather needs equally sized pieces: chunk c of every rank must have the same row count
        self.uniform = all(row_block(r, self.world, rows)[1] - row_block(r, self.world, rows)[0] == self.y1 - self.y0
                           for r in range(self.world))

    def _view(self, buf: torch.Tensor, y0: int, y1: int, base_row: int) -> torch.Tensor:
        return buf[(y0 - base_row) * self.row_bytes:(y1 - base_row) * self.row_bytes]

    def step(self) -> Optional[torch.Tensor]:
        """Trace this rank's rows and gather everything on rank 0.  Returns the
        full image (uint8, rows*width*32 bytes) on rank 0, None elsewhere; with
        world == 1 returns the local block."""
        pending = []
        for (c0, c1) in self.chunks:
            out = self._view(self.local, c0, c1, self.y0)
            self.tracer(c0, c1, out)
            if not self.gather:
                continue
            if self.uniform:
                dst_list = None
                if self.rank == 0:
                    dst_list = []
                    for r in range(self.world):
                        ry0, _ = row_block(r, self.world, self.rows)
                        dst_list.append(self._view(self.image, ry0 + (c0 - self.y0), ry0 + (c1 - self.y0), 0))
                pending.append(dist.gather(out, dst_list, dst=0, group=self.group, async_op=True))
            else:  # ragged blocks: point-to-point
                if self.rank == 0:
                    self._view(self.image, c0, c1, 0).copy_(out)
                else:
                    pending.append(dist.isend(out, dst=0, group=self.group))
        if self.gather and not self.uniform and self.rank == 0:
            for r in range(1, self.world):
                ry0, ry1 = row_block(r, self.world, self.rows)
                for (c0, c1) in chunk_bounds(ry0, ry1, len(self.chunks)):
                    pending.append(dist.irecv(self._view(self.image, c0, c1, 0), src=r, group=self.group))
        for w in pending:
            w.wait()
        if not self.gather:
            return self.local
        return self.image if self.rank == 0 else None


class ShardedViews:
    """N independent views (one full grid per rank), gathered on rank 0: the weak-scaling
    workload of bench.py.  Same exchange as ShardedGrid with one block per rank."""

    def __init__(self, width: int, rows: int, tracer: Callable[[int, int, torch.Tensor], None],
                 device: torch.device, chunks: int = 4, group=None, gather: bool = True, force_gather: bool = False):
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.width, self.rows, self.tracer, self.group = width, rows, tracer, group
        self.row_bytes = width * HIT_BYTES
        self.chunks = chunk_bounds(0, rows, chunks)
        self.local = torch.empty(rows * self.row_bytes, dtype=torch.uint8, device=device)
        # force_gather: run the exchange even with one rank (rehearsal of the N > 1 path on one GPU)
        self.gather = gather and (self.world > 1 or (force_gather and dist.is_initialized()))
        self.images = None
        if self.rank == 0 and self.gather:
            self.images = torch.empty((self.world, rows * self.row_bytes), dtype=torch.uint8, device=device)

    def step(self):
        pending = []
        for (c0, c1) in self.chunks:
            out = self.local[c0 * self.row_bytes:c1 * self.row_bytes]
            self.tracer(c0, c1, out)
            if self.gather:
                dst = None
                if self.rank == 0:
                    dst = [self.images[r, c0 * self.row_bytes:c1 * self.row_bytes] for r in range(self.world)]
                pending.append(dist.gather(out, dst, dst=0, group=self.group, async_op=True))
        for w in pending:
            w.wait()
        if not self.gather:
            return self.local
        return self.images if self.rank == 0 else None
