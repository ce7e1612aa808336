#!/usr/bin/env python3
"""C4 (2^24 incoherent rays, 1 M triangles): the 8-wide persistent lane kernel's own accounting — cache lines fetched
(one 128-byte line per node step), triangle rows tested, exact leaf boxes read, per ray, from its counting build, and
the kernel time of the plain build.  Prints the roofline row of BASELINE.md section 5.

    python tools/c4_accounting.py [--rounds 5]
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
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    cfg = synth.CONFIGS["C4"]
    scene = capi.Scene(synth.scene_vertices(cfg))
    rays = synth.incoherent_rays(cfg["incoherent"], cfg["ray_seed"])
    n = rays.shape[0]
    out = {}
    for name, kern in (("lane8", capi.KERNEL_LANE8_PERSISTENT), ("lane4", capi.KERNEL_LANE4_PERSISTENT), ("lane2", capi.KERNEL_LANE_PERSISTENT)):
        ctx = capi.Context(0, kernel=kern)
        scene.upload(ctx)
        d_rays, d_hits = ctx.device_alloc(n * 32), ctx.device_alloc(n * 32)
        ctx.h2d(d_rays, rays)
        dev = capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE | capi.FLAG_COHERENT   # COHERENT: no sort (unsorted walk)
        ms = []
        for _ in range(a.rounds + 1):
            ctx.cast(d_rays, d_hits, count=n, flags=dev)
            ms.append(ctx.stats()["last_trace_ms"])
        t = float(np.median(ms[1:]))
        cctx = capi.Context(0, kernel=kern, count_visits=1)
        scene.upload(cctx)
        cctx.cast(d_rays, d_hits, count=n, flags=dev)
        s = cctx.stats()
        line = 128 if kern != capi.KERNEL_LANE_PERSISTENT else 64
        lines, tris, boxes = s["wave_node_fetches"] / n, s["wave_tri_fetches"] / n, s["leaf_box_checks"] / n
        bytes_per_ray = line * lines + 48 * tris + 32 * boxes + 64
        out[name] = dict(kernel=capi.kernel_name(s["last_kernel"]), ms=t, grays=n / t / 1e6, lines_per_ray=lines, tri_rows_per_ray=tris,
                         leaf_boxes_per_ray=boxes, line_bytes=line, bytes_per_ray=bytes_per_ray,
                         requested_tbs=bytes_per_ray * n / (t * 1e-3) / 1e12, frac_of_hbm_peak=bytes_per_ray * n / (t * 1e-3) / 8e12,
                         frac_of_fabric_line_rate=line * lines * n / (t * 1e-3) / 7.5e12)
        print(name, json.dumps(out[name]), flush=True)
        ctx.device_free(d_rays); ctx.device_free(d_hits); ctx.close(); cctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
