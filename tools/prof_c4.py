#!/usr/bin/env python3
"""Runs one lane-kernel variant on config C4 (2^24 incoherent rays) a few times: the workload
profiled by `rocprofv3 --pmc ... -- python3 tools/prof_c4.py [rounds] [kernel] [sort]`.
kernel: lane | persistent | wide4 | wide8 | auto (default persistent; auto = the library's choice, which also has the row array); sort: 0 | 1 (default 0)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kernel = {"lane": capi.KERNEL_LANE, "persistent": capi.KERNEL_LANE_PERSISTENT,
          "wide4": capi.KERNEL_LANE4_PERSISTENT, "wide8": capi.KERNEL_LANE8_PERSISTENT, "auto": capi.KERNEL_AUTO}[sys.argv[2] if len(sys.argv) > 2 else "persistent"]
sort = len(sys.argv) > 3 and sys.argv[3] == "1"
cfg = synth.CONFIGS["C4"]
scene = capi.Scene(synth.scene_vertices(cfg))
rays = synth.incoherent_rays(cfg["incoherent"], cfg["ray_seed"])
n = rays.shape[0]
ctx = capi.Context(0, kernel=kernel)
scene.upload(ctx)
d_rays, d_hits = ctx.device_alloc(n * 32), ctx.device_alloc(n * 32)
ctx.h2d(d_rays, rays)
flags = capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE | (0 if sort else capi.FLAG_COHERENT)
for _ in range(rounds):
    ctx.cast(d_rays, d_hits, count=n, flags=flags)
    print(ctx.stats()["last_trace_ms"], ctx.stats()["last_sort_ms"], flush=True)
