"""Digests of hit-record batches: what tests/golden/full_digests.json holds for the
whole C2-C5 batches and what the GPU tests and bench.py recompute from the HIP path's output.

TEST INFRASTRUCTURE ONLY (checker code, like the rest of oracle/).  numpy only.
"""
import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
MASK64 = (1 << 64) - 1


def _pos_hash(x_u32: np.ndarray, first_index: int) -> int:
    """sum_i (x_i + 1) * ((2 i + 1) * GOLDEN) mod 2^64 with i = first_index + position."""
    n = x_u32.shape[0]
    if n == 0:
        return 0
    with np.errstate(over="ignore"):
        i = np.arange(first_index, first_index + n, dtype=np.uint64)
        w = (i * np.uint64(2) + np.uint64(1)) * GOLDEN
        return int(((x_u32.astype(np.uint64) + np.uint64(1)) * w).sum(dtype=np.uint64))


def digest_columns(prim_id: np.ndarray, t: np.ndarray, first_index: int = 0) -> dict:
    """prim_id: int32 (-1 = miss), t: float32, both of the same rays [first_index, first_index + n)."""
    prim_id = np.ascontiguousarray(prim_id, dtype=np.int32)
    t = np.ascontiguousarray(t, dtype=np.float32)
    hit = prim_id >= 0
    pu = prim_id.view(np.uint32)
    return dict(rays=int(prim_id.shape[0]), hit_count=int(hit.sum()),
                prim_xor=int(np.bitwise_xor.reduce(pu[hit])) if hit.any() else 0,
                prim_hash=_pos_hash(pu, first_index), t_hash=_pos_hash(t.view(np.uint32), first_index),
                sum_t=float(t[hit].astype(np.float64).sum()))


def digest_records(hits: np.ndarray, first_index: int = 0) -> dict:
    """hits: structured mrt_hit32 records (fields t, prim_id)."""
    return digest_columns(hits["prim_id"], hits["t"], first_index)


def combine(a: dict, b: dict) -> dict:
    return dict(rays=a["rays"] + b["rays"], hit_count=a["hit_count"] + b["hit_count"], prim_xor=a["prim_xor"] ^ b["prim_xor"],
                prim_hash=(a["prim_hash"] + b["prim_hash"]) & MASK64, t_hash=(a["t_hash"] + b["t_hash"]) & MASK64,
                sum_t=a["sum_t"] + b["sum_t"])


class Accumulator:
    def __init__(self):
        self.d = dict(rays=0, hit_count=0, prim_xor=0, prim_hash=0, t_hash=0, sum_t=0.0)

    def add(self, hits: np.ndarray, first_index: int):
        self.d = combine(self.d, digest_records(hits, first_index))

    def result(self) -> dict:
        return dict(self.d)


def same(a: dict, b: dict) -> bool:
    """Exact fields equal (sum_t is informative: its value depends on the summation order)."""
    return all(a[k] == b[k] for k in ("rays", "hit_count", "prim_xor", "prim_hash", "t_hash"))
