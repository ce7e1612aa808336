#!/usr/bin/env python3
"""Coherent batches of 2^12 .. 2^20 rays (square primary-ray grids, rays and hits resident): the packet walk against the lane
kernel, kernel time and blocking-call wall time.  Where does the packet walk start to pay?
    python tools/bench_small_batches.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth  # noqa: E402


def main():
    cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
    scene = capi.Scene(synth.scene_vertices(cfg))
    ctxs = {"auto": capi.Context(0), "lane": capi.Context(0, kernel=capi.KERNEL_LANE), "packet_asm": capi.Context(0, kernel=capi.KERNEL_PACKET_ASM)}
    for c in ctxs.values():
        scene.upload(c)
    base = ctxs["auto"]
    out = {}
    for side in (64, 128, 256, 512, 1024):
        n = side * side
        cam = capi.camera_look(cfg["origin"], cfg["forward"], side, side, cfg["fov"])
        d_rays, d_hits = base.device_alloc(n * 32), base.device_alloc(n * 32)
        base.generate_grid(cam, side, side, 0, side, d_rays)
        flags = capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE
        row = {}
        for name, c, fl in (("auto_coherent", ctxs["auto"], flags | capi.FLAG_COHERENT), ("packet_asm_coherent", ctxs["packet_asm"], flags | capi.FLAG_COHERENT),
                            ("lane_coherent_flag", ctxs["lane"], flags | capi.FLAG_COHERENT), ("auto_unflagged_sorted", ctxs["auto"], flags)):
            for _ in range(5):
                c.cast(d_rays, d_hits, count=n, flags=fl)
            wall, ker = [], []
            for _ in range(30):
                t0 = time.perf_counter()
                c.cast(d_rays, d_hits, count=n, flags=fl)
                wall.append(time.perf_counter() - t0)
                s = c.stats()
                ker.append(s["last_trace_ms"] + s["last_sort_ms"])
            row[name] = dict(wall_us=float(np.median(wall)) * 1e6, device_us=float(np.median(ker)) * 1e3, kernel=capi.kernel_name(c.stats()["last_kernel"]))
        out[n] = row
        print(n, json.dumps(row), flush=True)
        base.device_free(d_rays); base.device_free(d_hits)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
