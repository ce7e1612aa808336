#!/usr/bin/env python3
"""Do consecutive frames overlap?  K casts of one config: (a) blocking, one context; (b) ASYNC on one context's stream;
(c) ASYNC, alternating between two contexts (two streams, two hit buffers): the head of frame n+1 fills the wave slots
the tail of frame n leaves empty.   python tools/exp_overlap.py --config C3 --steps 20"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--kernel", type=int, default=0)
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    w, h = cfg["grid"]
    scene = capi.Scene(synth.scene_vertices(cfg))
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    ctxs = [capi.Context(0, kernel=a.kernel) for _ in range(3)]
    for c in ctxs:
        scene.upload(c)
    d_rays = ctxs[0].device_alloc(w * h * 32)
    ctxs[0].generate_grid(cam, w, h, 0, h, d_rays)
    hits = [c.device_alloc(w * h * 32) for c in ctxs]
    base = capi.FLAG_COHERENT | capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE
    out = {}
    for name, n_ctx, flags in (("blocking", 1, base), ("async_1_stream", 1, base | capi.FLAG_ASYNC), ("async_2_streams", 2, base | capi.FLAG_ASYNC),
                               ("async_3_streams", 3, base | capi.FLAG_ASYNC)):
        for rep in range(2):
            for c in ctxs:
                c.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                c = ctxs[i % n_ctx]
                c.cast(d_rays, hits[i % n_ctx], count=w * h, flags=flags)
            for c in ctxs:
                c.synchronize()
            dt = time.perf_counter() - t0
        out[name] = dict(ms_per_step=dt / a.steps * 1e3, mrays=w * h * a.steps / dt / 1e6)
        print(name, json.dumps(out[name]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
