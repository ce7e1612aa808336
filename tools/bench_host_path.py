#!/usr/bin/env python3
"""End-to-end timing of mrt_cast with HOST arrays (the reference's cast_rays contract):
PCIe in, trace, PCIe out.  Not the headline metric.

Rays come from the library's own grid generator (mrt_generate_grid, copied to the host); the
60-byte host rays carry the fields the path consumes (origin, direction, t_min, t_max;
src/gpu/gpu_ray_caster.cpp:643-650)."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth, types as T  # noqa: E402

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
w, h = cfg["grid"]
scene = capi.Scene(synth.scene_vertices(cfg))
ctx = capi.Context(0)
scene.upload(ctx)
n = w * h
cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
d_rays = ctx.device_alloc(n * 32)
ctx.generate_grid(cam, w, h, 0, h, d_rays)
rays = np.zeros(n, dtype=T.RAY32)
ctx.d2h(rays, d_rays)
ctx.device_free(d_rays)
host = np.zeros(n, dtype=T.HOST_RAY60)
host["origin"], host["direction"], host["t_min"], host["t_max"] = rays["origin"], rays["direction"], rays["t_min"], rays["t_max"]
for name, arr, flags, out in (("packed 32B/32B", rays, capi.FLAG_COHERENT, np.zeros(n, dtype=T.HIT32)),
                              ("host layout 60B/44B", host, capi.FLAG_COHERENT | capi.FLAG_HOST_LAYOUT, np.zeros(n, dtype=T.HOST_HIT44))):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.cast(arr, out, flags=flags)
        ts.append(time.perf_counter() - t0)
    print(f"{name}: wall {min(ts) * 1e3:.1f} ms -> {n / min(ts) / 1e6:.0f} Mrays/s end to end "
          f"({int((out['prim_id'] != (0xFFFFFFFF if out.dtype == T.HOST_HIT44 else -1)).sum())} hits)", flush=True)
