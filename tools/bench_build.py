#!/usr/bin/env python3
"""Build time and trace time: host-built 8-bin SAH tree (mrt_bvh2_build + mrt_upload_scene) against
the device-built trees (mrt_build_scene_device: the radix tree, PLOC with MRT_BUILD_PLOC, binned SAH with MRT_BUILD_SAH), on one config's scene and primary-ray grid.

    python tools/bench_build.py --config C3 [--rounds 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth, types as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    w, h = cfg["grid"]
    verts = synth.scene_vertices(cfg)
    tris = capi.make_triangles(verts)
    n = tris.shape[0]
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    ctx = capi.Context(0)
    d_hits = ctx.device_alloc(w * h * 32)
    out = dict(config=a.config, tris=int(n), rays=w * h)

    def trace_ms():
        ts = []
        for _ in range(a.rounds):
            ctx.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
            ts.append(ctx.stats()["last_trace_ms"])
        return float(np.median(ts))

    inc = synth.incoherent_rays(1 << 22, 7)
    d_inc, d_inc_hits = ctx.device_alloc(inc.nbytes), ctx.device_alloc(inc.shape[0] * 32)
    ctx.h2d(d_inc, inc)

    def incoherent_ms():
        ts = []
        for _ in range(a.rounds):
            ctx.cast(d_inc, d_inc_hits, count=inc.shape[0],
                     flags=capi.FLAG_COHERENT | capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE)
            ts.append(ctx.stats()["last_trace_ms"])
        return float(np.median(ts))

    def digest():
        hits = np.zeros(w * h, dtype=T.HIT32)
        ctx.d2h(hits, d_hits)
        return hits

    # host: build (threads = all cores of this process) + conversion + upload
    t0 = time.perf_counter()
    verts4 = T.verts4_from_verts9(verts)
    nodes, prim_idx, used = capi.bvh2_build(verts4, 0)
    t1 = time.perf_counter()
    ctx.upload_scene(tris, nodes, prim_idx)
    t2 = time.perf_counter()
    out["host"] = dict(build_s=t1 - t0, upload_s=t2 - t1, threads=len(os.sched_getaffinity(0)),
                       stack_need=ctx.scene_info()["stack_need"], rows=ctx.scene_info()["n_wide_nodes"], trace_ms=trace_ms(), incoherent_4M_ms=incoherent_ms())
    host_hits = digest()

    # device: triangles from host memory (PCIe included in wall time), and already resident; the default form
    # (the radix tree) and locally-ordered clustering (MRT_BUILD_PLOC)
    d_tris = ctx.device_alloc(tris.nbytes)
    ctx.h2d(d_tris, tris)
    for name, kw in (("device", {}), ("device_ploc", {"ploc": True}), ("device_sah", {"sah": True})):
        walls, devs = [], []
        for _ in range(a.rounds):
            t0 = time.perf_counter()
            ctx.build_scene_device(tris, **kw)
            walls.append(time.perf_counter() - t0)
            devs.append(ctx.stats()["last_build_ms"])
        res = []
        for _ in range(a.rounds):
            ctx.build_scene_device(d_tris, n_tris=n, on_device=True, **kw)
            res.append(ctx.stats()["last_build_ms"])
        out[name] = dict(build_ms_from_host_tris=float(np.median(devs)), wall_ms_from_host_tris=float(np.median(walls)) * 1e3,
                         build_ms_resident_tris=float(np.median(res)), stack_need=ctx.scene_info()["stack_need"], rows=ctx.scene_info()["n_wide_nodes"],
                         trace_ms=trace_ms(), incoherent_4M_ms=incoherent_ms())
        out[name]["identical_hits"] = bool(digest().tobytes() == host_hits.tobytes())
        out[name]["trace_vs_host"] = out[name]["trace_ms"] / out["host"]["trace_ms"]
        out[name]["incoherent_vs_host"] = out[name]["incoherent_4M_ms"] / out["host"]["incoherent_4M_ms"]
    # the host tree once more, now that the device has been busy for seconds (the first measurement above runs on a cold part):
    # the ratios are taken against the faster of the two
    ctx.upload_scene(tris, nodes, prim_idx)
    out["host"]["trace_ms_again"], out["host"]["incoherent_4M_ms_again"] = trace_ms(), incoherent_ms()
    ht, hi = min(out["host"]["trace_ms"], out["host"]["trace_ms_again"]), min(out["host"]["incoherent_4M_ms"], out["host"]["incoherent_4M_ms_again"])
    for name in ("device", "device_ploc", "device_sah"):
        out[name]["trace_vs_host"] = out[name]["trace_ms"] / ht
        out[name]["incoherent_vs_host"] = out[name]["incoherent_4M_ms"] / hi
    ctx.build_scene_device(d_tris, n_tris=n, on_device=True)
    ctx.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
    out["identical_hits"] = bool(digest().tobytes() == host_hits.tobytes())
    if cfg.get("scene") == "multi_mesh":
        # the same scene as placed meshes: flatten (Transform3D::xform + Triangle ctor) and build on the device
        local, inst = synth.multi_mesh_instances(cfg["n_meshes"], cfg["tris_per_mesh"], cfg["s"], cfg["seed"])
        d_local = ctx.device_alloc(local.nbytes)
        ctx.h2d(d_local, local)
        ws = []
        for _ in range(a.rounds):
            t0 = time.perf_counter()
            ctx.build_instanced_scene_device(d_local, inst, n_mesh_tris=local.shape[0], on_device=True)
            ws.append(time.perf_counter() - t0)
        ctx.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
        out["instanced"] = dict(instances=int(inst.shape[0]), flatten_plus_build_wall_ms=float(np.median(ws)) * 1e3,
                                build_ms=ctx.stats()["last_build_ms"], identical_hits=bool(digest().tobytes() == host_hits.tobytes()))
        ctx.device_free(d_local)
    out["mtris_per_s_device_build"] = n / out["device"]["build_ms_resident_tris"] / 1e3
    print(json.dumps(out), flush=True)
    ctx.device_free(d_tris)
    ctx.device_free(d_hits)
    ctx.close()


if __name__ == "__main__":
    main()
