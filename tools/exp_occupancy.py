#!/usr/bin/env python3
"""Occupancy sweep of the packet kernel: the same cast with 8, 7, ... 2 resident waves per SIMD
(mrt_options.extra_lds pads every workgroup's LDS).  If the time grows like 1 / waves the kernel
is bound by the latency of its dependent fetches (more loads in flight would help); if it stays
flat it is bound by issue.

    python tools/exp_occupancy.py --config C3 [--rounds 5] [--two-level]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth  # noqa: E402

# extra LDS bytes -> workgroups per CU = floor(163840 / (static + extra)); 4 waves per workgroup = waves per SIMD
SWEEP = [(0, 8), (18000, 7), (22000, 6), (28000, 5), (36000, 4), (50000, 3), (60000, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--two-level", action="store_true")
    ap.add_argument("--kernel", type=int, default=capi.KERNEL_AUTO)
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    w, h = cfg["grid"]
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    scene = None
    if a.two_level:
        local, inst = synth.multi_mesh_instances(cfg["n_meshes"], cfg["tris_per_mesh"], cfg["s"], cfg["seed"])
    else:
        scene = capi.Scene(synth.scene_vertices(cfg))
    out = {"config": a.config, "two_level": a.two_level, "rows": []}
    d_hits = None
    for extra, waves in SWEEP:
        ctx = capi.Context(0, kernel=a.kernel, extra_lds=extra)
        if a.two_level:
            ctx.upload_two_level_scene(local, inst)
        else:
            scene.upload(ctx)
        d_hits = ctx.device_alloc(w * h * 32)
        ms = []
        for _ in range(a.rounds + 1):
            ctx.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
            ms.append(ctx.stats()["last_trace_ms"])
        t = float(np.median(ms[1:]))
        out["rows"].append({"extra_lds": extra, "waves_per_simd": waves, "ms": t, "grays": w * h / t / 1e6})
        print(f"{a.config} waves/SIMD {waves}: {t:.3f} ms  {w * h / t / 1e6:.2f} Grays/s", flush=True)
        ctx.device_free(d_hits)
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
