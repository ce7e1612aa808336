"""mrt_cast(COHERENT) on device-resident rays of renderer-size grids (the reference's cast_rays contract: no width given, the device finds
it; from the second cast on the batch is scheduled and tuned by what the previous cast of as many rays found): kernel time, median of casts 16..29.
    python tools/bench_cast_resolutions.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from messyerraytracer_amd import capi, synth
cfg = synth.CONFIGS["C3"]
scene = capi.Scene(synth.scene_vertices(cfg))
c = capi.Context(0)
scene.upload(c)
for w, h in ((1280, 720), (1280, 960), (1920, 1080), (2560, 1440)):
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    d_rays, d_hits = c.device_alloc(w * h * 32), c.device_alloc(w * h * 32)
    c.generate_grid(cam, w, h, 0, h, d_rays)
    ts = []
    for _ in range(30):
        c.cast(d_rays, d_hits, count=w * h, flags=capi.FLAG_COHERENT | capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE)
        ts.append(c.stats()["last_trace_ms"])
    print(f"mrt_cast(COHERENT) {w}x{h} {float(np.median(ts[16:])):8.3f} ms {c.last_kernel_variant()}", flush=True)
    c.device_free(d_rays); c.device_free(d_hits)
c.close()
