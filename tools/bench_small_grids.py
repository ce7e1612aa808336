"""Fused grid casts (mrt_cast_grid, records resident) below renderer size on one config's scene: kernel time, median of frames 14..29.
    python tools/bench_small_grids.py [C3 [kernel id; 0 = the library's choice, 5 = the 64-ray packet kernel, 1 = one lane per ray]]
MRT_SCHEDULE_MIN_LOG2 / mrt_options.tile_schedule select what is scheduled (api.hip: quarter_small_grid, schedule_plan_kernel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from messyerraytracer_amd import capi, synth
cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
scene = capi.Scene(synth.scene_vertices(cfg))
c = capi.Context(0, kernel=int(sys.argv[2]) if len(sys.argv) > 2 else 5)
scene.upload(c)
for w, h in ((16, 12), (32, 32), (64, 64), (96, 96), (128, 128), (192, 192), (256, 256), (384, 384), (512, 512), (640, 360), (720, 405), (800, 450), (960, 540), (1024, 576)):
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    d_hits = c.device_alloc(w * h * 32)
    ts = []
    for _ in range(30):
        c.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
        ts.append(c.stats()["last_trace_ms"])
    ms = float(np.median(ts[14:]))
    print(f"{w}x{h} {ms:8.3f} ms {c.last_kernel_variant()}", flush=True)
    c.device_free(d_hits)
c.close()
