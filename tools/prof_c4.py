#!/usr/bin/env python3
"""Runs the lane kernel on config C4 (2^24 incoherent rays, sorted) a few times: the workload
profiled by `rocprofv3 --pmc ... -- python3 tools/prof_c4.py`."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth  # noqa: E402

cfg = synth.CONFIGS["C4"]
scene = capi.Scene(synth.scene_vertices(cfg))
rays = synth.incoherent_rays(cfg["incoherent"], cfg["ray_seed"])
n = rays.shape[0]
ctx = capi.Context(0, kernel=capi.KERNEL_LANE)
scene.upload(ctx)
d_rays, d_hits = ctx.device_alloc(n * 32), ctx.device_alloc(n * 32)
ctx.h2d(d_rays, rays)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ctx.cast(d_rays, d_hits, count=n, flags=capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE)
    print(ctx.stats()["last_trace_ms"], ctx.stats()["last_sort_ms"], flush=True)
