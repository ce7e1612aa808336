#!/usr/bin/env python3
"""End-to-end timing of mrt_cast with HOST arrays (the reference's cast_rays contract):
PCIe in, trace, PCIe out.  Not the headline metric."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth, types as T  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (ray generation only)

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
w, h = cfg["grid"]
scene = capi.Scene(synth.scene_vertices(cfg))
ctx = capi.Context(0)
scene.upload(ctx)
rays = po.grid_rays(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
host = po.make_host_rays(rays)
n = rays.shape[0]
for name, arr, flags, out in (("packed 32B/32B", rays, capi.FLAG_COHERENT, np.zeros(n, dtype=T.HIT32)),
                              ("host layout 60B/44B", host, capi.FLAG_COHERENT | capi.FLAG_HOST_LAYOUT, np.zeros(n, dtype=T.HOST_HIT44))):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.cast(arr, out, flags=flags)
        ts.append(time.perf_counter() - t0)
    s = ctx.stats()
    print(f"{name}: wall {min(ts) * 1e3:.1f} ms -> {n / min(ts) / 1e6:.0f} Mrays/s end to end; h2d {s['last_h2d_ms']:.1f} ms "
          f"trace {s['last_trace_ms']:.2f} ms d2h {s['last_d2h_ms']:.1f} ms", flush=True)
