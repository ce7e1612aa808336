#!/usr/bin/env python3
"""Fused primary-ray casts (mrt_cast_grid, records resident) at renderer resolutions on one config's scene.
    python tools/bench_resolutions.py [C3 [kernel id]]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth  # noqa: E402


def main():
    """python tools/bench_resolutions.py [C3 [kernel ids, comma separated; 0 = the library's choice] [nosched | nopieces]]"""
    cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
    kernels = [int(k) for k in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    nosched = len(sys.argv) > 3 and sys.argv[3] == "nosched"   # mrt_options.tile_schedule = 1: every cast in the plain tile order
    nopieces = len(sys.argv) > 3 and sys.argv[3] == "nopieces"  # 2: longest first, but no unit launched in pieces
    scene = capi.Scene(synth.scene_vertices(cfg))
    out = {}
    for kid in kernels:
        c = capi.Context(0, kernel=kid, tile_schedule=1 if nosched else (2 if nopieces else 0))
        scene.upload(c)
        for w, h in ((640, 360), (1280, 720), (1280, 960), (1920, 1080), (2560, 1440), (3840, 2160), (4096, 4096), (7680, 4320)):
            cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
            d_hits = c.device_alloc(w * h * 32)
            for mode, name in ((capi.MODE_NEAREST, "nearest"), (capi.MODE_ANY_HIT, "any_hit")):
                ts = []
                for _ in range(28):   # (frames 0-11 of a grid are the library's measuring frames: four per candidate)
                    c.cast_grid(cam, w, h, hits=d_hits, mode=mode, flags=capi.FLAG_HITS_ON_DEVICE)
                    ts.append(c.stats()["last_trace_ms"])
                ms = float(np.median(ts[14:]))
                out[f"k{kid}_{w}x{h}_{name}"] = dict(ms=ms, mrays=w * h / ms / 1e3, kernel=c.last_kernel_variant())
                print(f"kernel {kid:2d} {w}x{h} {name:8s} {ms:8.3f} ms  {w * h / ms / 1e3:8.1f} Mrays/s  {c.last_kernel_variant()}", flush=True)
            c.device_free(d_hits)
        c.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
