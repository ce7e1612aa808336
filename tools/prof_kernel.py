#!/usr/bin/env python3
"""Runs ONE kernel variant on one config a few times: the program rocprofv3 wraps for counter passes
(tools/pmc_kernel.sh).  No verification, no baseline; prints the median kernel time.

    python3 tools/prof_kernel.py --config C3 --kernel 11 --iters 8
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--count", type=int, default=0)
    ap.add_argument("--cull", type=int, default=0, help="mrt_options.packet_cull")
    ap.add_argument("--wg", type=int, default=0, help="mrt_options.packet_wg")
    ap.add_argument("--entry", default="fused", choices=["fused", "cast"], help="mrt_cast_grid, or mrt_cast(COHERENT) on device-resident rays")
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    w, h = cfg["grid"]
    scene = capi.Scene(synth.scene_vertices(cfg))
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    c = capi.Context(0, kernel=a.kernel, count_visits=a.count, packet_cull=a.cull, packet_wg=a.wg)
    scene.upload(c)
    d_hits = c.device_alloc(w * h * 32)
    d_rays = None
    if a.entry == "cast":
        d_rays = c.device_alloc(w * h * 32)
        c.generate_grid(cam, w, h, 0, h, d_rays)
    ms = []
    for _ in range(a.iters):
        if d_rays:
            c.cast(d_rays, d_hits, count=w * h, flags=capi.FLAG_COHERENT | capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE)
        else:
            c.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
        ms.append(c.stats()["last_trace_ms"])
    s = c.stats()
    print(json.dumps(dict(config=a.config, kernel=capi.kernel_name(s["last_kernel"]), median_ms=float(np.median(ms)), min_ms=float(min(ms)))))


if __name__ == "__main__":
    main()
