#!/usr/bin/env python3
"""Config C5 as a two-level scene (64 meshes of 156 250 triangles, nothing flattened):
upload time, 8192^2 primary grid through the two-level kernel, and the cost of moving every instance
(mrt_update_instances).  Optionally the same scene with every mesh placed `--copies` times.

    python tools/bench_two_level.py [--grid 8192] [--rounds 5] [--copies 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=8192)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--copies", type=int, default=1)
    ap.add_argument("--lane", action="store_true", help="one lane per ray instead of the packet form")
    ap.add_argument("--device-blas", action="store_true", help="build the meshes' BVHs on the device")
    ap.add_argument("--sah", action="store_true", help="with --device-blas: the binned-SAH form of the device builder (MRT_BUILD_SAH)")
    a = ap.parse_args()
    cfg = synth.CONFIGS["C5"]
    local, inst = synth.multi_mesh_instances(cfg["n_meshes"], cfg["tris_per_mesh"], cfg["s"], cfg["seed"])
    if a.copies > 1:  # the same meshes placed again, shifted: instancing proper
        reps = [inst]
        for k in range(1, a.copies):
            more = inst.copy()
            more["origin"] += np.float32([0.37 * k, -0.21 * k, 0.5 * k])
            reps.append(more)
        inst = np.concatenate(reps)
    c = capi.Context(0, kernel=capi.KERNEL_LANE if a.lane else capi.KERNEL_AUTO)
    t0 = time.perf_counter()
    c.upload_two_level_scene(local, inst, blas_on_device=a.device_blas, sah=a.sah and a.device_blas)
    t_up = time.perf_counter() - t0
    info = c.scene_info()
    w = h = a.grid
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    d_hits = c.device_alloc(w * h * 32)

    def run(mode=capi.MODE_NEAREST):
        c.cast_grid(cam, w, h, hits=d_hits, mode=mode, flags=capi.FLAG_HITS_ON_DEVICE)
        return c.stats()["last_trace_ms"]

    run()
    ms = float(np.median([run() for _ in range(a.rounds)]))
    ms_any = float(np.median([run(capi.MODE_ANY_HIT) for _ in range(a.rounds)]))
    t0 = time.perf_counter()
    c.update_instances(inst)
    t_refit = time.perf_counter() - t0
    out = dict(kernel="lane" if a.lane else "packet", blas=("device SAH" if a.sah else "device LBVH") if a.device_blas else "host SAH", build_ms=c.stats()["last_build_ms"], instances=int(inst.shape[0]), mesh_tris=int(local.shape[0]), flat_tris=int(inst["n_tris"].sum()), info=info,
               upload_s=t_up, grid=[w, h], cast_grid_ms=ms, mrays=w * h / ms / 1e3, anyhit_ms=ms_any, anyhit_mrays=w * h / ms_any / 1e3,
               update_instances_ms=t_refit * 1e3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
