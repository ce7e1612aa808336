#!/usr/bin/env python3
"""Measures every BASELINE.json config on one GPU (kernel-only, rays and hits resident in
HBM) and prints the rows of BASELINE.md section 5 / a JSON blob for profiles/.

    python tools/bench_all.py [--configs C2,C3,C4,C5] [--rounds 7]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth  # noqa: E402


def median_ms(fn, rounds):
    fn()
    return float(np.median([fn() for _ in range(rounds)]))


def grid_case(name, rounds, out):
    cfg = synth.CONFIGS[name]
    w, h = cfg["grid"]
    scene = capi.Scene(synth.scene_vertices(cfg))
    ctx = capi.Context(0)
    scene.upload(ctx)
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    n = w * h
    d_rays, d_hits = ctx.device_alloc(n * 32), ctx.device_alloc(n * 32)
    ctx.generate_grid(cam, w, h, 0, h, d_rays)
    dev = capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE

    def cast(mode=capi.MODE_NEAREST):
        ctx.cast(d_rays, d_hits, count=n, mode=mode, flags=dev | capi.FLAG_COHERENT)
        s = ctx.stats()
        return s["last_trace_ms"] + s["last_sort_ms"]

    def fused():
        ctx.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
        return ctx.stats()["last_trace_ms"]

    r = dict(rays=n, tris=int(scene.tris.shape[0]), info=ctx.scene_info())
    r["cast_ms"] = median_ms(cast, rounds)
    r["fused_ms"] = median_ms(fused, rounds)
    r["anyhit_ms"] = median_ms(lambda: cast(capi.MODE_ANY_HIT), rounds)
    for k in ("cast", "fused", "anyhit"):
        r[k + "_mrays"] = n / r[k + "_ms"] / 1e3
    ctx.device_free(d_rays); ctx.device_free(d_hits); ctx.close()
    out[name] = r


def incoherent_case(name, rounds, out):
    cfg = synth.CONFIGS[name]
    scene = capi.Scene(synth.scene_vertices(cfg))
    rays = synth.incoherent_rays(cfg["incoherent"], cfg["ray_seed"])
    n = rays.shape[0]
    r = dict(rays=n, tris=int(scene.tris.shape[0]))
    dev = capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE
    variants = [("lane", dict(kernel=capi.KERNEL_LANE)), ("lane_dirkey", dict(kernel=capi.KERNEL_LANE, sort_key=1)),
                ("persist16", dict(kernel=capi.KERNEL_LANE_PERSISTENT)),
                ("persist16_r8", dict(kernel=capi.KERNEL_LANE_PERSISTENT, refill=8)),
                ("persist16_r32", dict(kernel=capi.KERNEL_LANE_PERSISTENT, refill=32)),
                ("persist24", dict(kernel=capi.KERNEL_LANE_PERSISTENT, stack_override=24)),
                ("persist32", dict(kernel=capi.KERNEL_LANE_PERSISTENT, stack_override=32)),
                ("wide4_16", dict(kernel=capi.KERNEL_LANE4_PERSISTENT)),
                ("wide4_24", dict(kernel=capi.KERNEL_LANE4_PERSISTENT, stack_override=24)),
                ("wide4_32", dict(kernel=capi.KERNEL_LANE4_PERSISTENT, stack_override=32)),
                ("wide4_16_r32", dict(kernel=capi.KERNEL_LANE4_PERSISTENT, refill=32)),
                ("wide8", dict(kernel=capi.KERNEL_LANE8_PERSISTENT)),
                ("wide8_l8", dict(kernel=capi.KERNEL_LANE8_PERSISTENT, leaf_wait=8)),
                ("wide8_l32", dict(kernel=capi.KERNEL_LANE8_PERSISTENT, leaf_wait=32)),
                ("wide8_s24", dict(kernel=capi.KERNEL_LANE8_PERSISTENT, stack_override=24)),
                *[(f"p2_l{lw}_r{rf}", dict(kernel=capi.KERNEL_LANE_PERSISTENT, leaf_wait=lw, refill=rf))
                  for lw in (2, 4, 8, 16, 32, 64) for rf in (8, 16)],
                *[(f"p4_l{lw}_r{rf}", dict(kernel=capi.KERNEL_LANE4_PERSISTENT, leaf_wait=lw, refill=rf))
                  for lw in (2, 4, 8, 16, 32, 64) for rf in (8, 16)],
                ("auto", dict())]
    only = os.environ.get("MRT_BENCH_VARIANTS")  # comma-separated subset
    if only:
        variants = [v for v in variants if v[0] in only.split(",")]
    for kname, kw in variants:
        ctx = capi.Context(0, **kw)
        scene.upload(ctx)
        d_rays, d_hits = ctx.device_alloc(n * 32), ctx.device_alloc(n * 32)
        ctx.h2d(d_rays, rays)

        def run(flags):
            ctx.cast(d_rays, d_hits, count=n, flags=dev | flags)
            s = ctx.stats()
            return s["last_trace_ms"], s["last_sort_ms"]

        for label, flags in (("sort_off", capi.FLAG_COHERENT), ("sort_on", 0)):
            run(flags)
            t = np.array([run(flags) for _ in range(rounds)])
            tr, so = float(np.median(t[:, 0])), float(np.median(t[:, 1]))
            r[f"{kname}_{label}"] = dict(trace_ms=tr, sort_ms=so, mrays_trace=n / tr / 1e3, mrays_total=n / (tr + so) / 1e3)
        ctx.device_free(d_rays); ctx.device_free(d_hits); ctx.close()
    out[name] = r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C2,C3,C4")
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    out = {}
    for name in a.configs.split(","):
        if "incoherent" in synth.CONFIGS[name]:
            incoherent_case(name, a.rounds, out)
        else:
            grid_case(name, a.rounds, out)
        print(name, json.dumps(out[name]), flush=True)
    print("JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
