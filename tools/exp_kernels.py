#!/usr/bin/env python3
"""A/B timing of kernel variants and tuning knobs on one config, in ONE process with
interleaved rounds (guide rule 24).  Every variant's hit buffer is compared with the
first variant's, so a knob that changes results is caught here.

    python tools/exp_kernels.py --config C3 --rounds 5 [--variants name,name,...]
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth, types as T  # noqa: E402

VARIANTS = {
    # name: (context kwargs, entry)
    "lane_linear": (dict(kernel=capi.KERNEL_LANE), "cast"),
    "lane_tile8x8": (dict(kernel=capi.KERNEL_LANE), "tiled"),
    "lane_tile8x8_fused": (dict(kernel=capi.KERNEL_LANE), "fused"),
    "lane_tile16x4": (dict(kernel=capi.KERNEL_LANE, tile_w_log2=4), "tiled"),
    "lane_tile4x16": (dict(kernel=capi.KERNEL_LANE, tile_w_log2=2), "tiled"),
    "lane_tile8x8_swz": (dict(kernel=capi.KERNEL_LANE, xcd_swizzle=1), "tiled"),
    "lane_tile8x8_stack64": (dict(kernel=capi.KERNEL_LANE, stack_override=64), "tiled"),
    "packet_linear": (dict(kernel=capi.KERNEL_PACKET), "cast"),
    "packet_tile8x8": (dict(kernel=capi.KERNEL_PACKET), "tiled"),
    "packet_tile8x8_fused": (dict(kernel=capi.KERNEL_PACKET), "fused"),
    "packet_tile16x4": (dict(kernel=capi.KERNEL_PACKET, tile_w_log2=4), "tiled"),
    "packet_tile8x8_swz": (dict(kernel=capi.KERNEL_PACKET, xcd_swizzle=1), "tiled"),
    "packet_tile4x16": (dict(kernel=capi.KERNEL_PACKET, tile_w_log2=2), "tiled"),
    "packet_tile8x8_zorder": (dict(kernel=capi.KERNEL_PACKET, tile_order=2), "tiled"),
    "packet_tile8x8_zorder_swz": (dict(kernel=capi.KERNEL_PACKET, tile_order=2, xcd_swizzle=1), "tiled"),
    "lane_tile8x8_zorder": (dict(kernel=capi.KERNEL_LANE, tile_order=2), "tiled"),
    "asm_tile8x8": (dict(kernel=capi.KERNEL_PACKET_ASM), "tiled"),
    "asm_linear": (dict(kernel=capi.KERNEL_PACKET_ASM), "cast"),
    "asm_fused": (dict(kernel=capi.KERNEL_PACKET_ASM), "fused"),
    "asm_fused_zorder": (dict(kernel=capi.KERNEL_PACKET_ASM, tile_order=2), "fused"),
    "asm_linear_zorder": (dict(kernel=capi.KERNEL_PACKET_ASM, tile_order=2), "cast"),
    "asm_fused_zorder32": (dict(kernel=capi.KERNEL_PACKET_ASM, tile_order=3), "fused"),
    "asm_fused_rowmajor": (dict(kernel=capi.KERNEL_PACKET_ASM, tile_order=1), "fused"),
    "asm_fused_zorder_swz": (dict(kernel=capi.KERNEL_PACKET_ASM, tile_order=2, xcd_swizzle=1), "fused"),
    "asm_fused_swz": (dict(kernel=capi.KERNEL_PACKET_ASM, xcd_swizzle=1), "fused"),
    "asm_fused_16x4": (dict(kernel=capi.KERNEL_PACKET_ASM, tile_w_log2=4), "fused"),
    "rows1_fused": (dict(kernel=capi.KERNEL_PACKET_ROWS), "fused"),
    "rows1_cast": (dict(kernel=capi.KERNEL_PACKET_ROWS), "cast"),
    "dual_fused": (dict(kernel=capi.KERNEL_PACKET_DUAL), "fused"),
    "dual_nocull_fused": (dict(kernel=capi.KERNEL_PACKET_DUAL, packet_cull=1), "fused"),
    "dual_nocull_cast": (dict(kernel=capi.KERNEL_PACKET_DUAL, packet_cull=1), "cast"),
    "dual_cast": (dict(kernel=capi.KERNEL_PACKET_DUAL), "cast"),
    "dual_fused_zorder": (dict(kernel=capi.KERNEL_PACKET_DUAL, tile_order=2), "fused"),
    "dual_fused_rowmajor": (dict(kernel=capi.KERNEL_PACKET_DUAL, tile_order=1), "fused"),
    "dual_fused_swz": (dict(kernel=capi.KERNEL_PACKET_DUAL, xcd_swizzle=1), "fused"),
    "dual_fused_zorder32": (dict(kernel=capi.KERNEL_PACKET_DUAL, tile_order=3), "fused"),
    "dual_cull_fused": (dict(kernel=capi.KERNEL_PACKET_DUAL, packet_cull=2), "fused"),
    "dual_cull_cast": (dict(kernel=capi.KERNEL_PACKET_DUAL, packet_cull=2), "cast"),
    "dual_fused_strips": (dict(kernel=capi.KERNEL_PACKET_DUAL, tile_order=4), "fused"),
    "dual_cast_strips": (dict(kernel=capi.KERNEL_PACKET_DUAL, tile_order=4), "cast"),
    "dual_cull_strips": (dict(kernel=capi.KERNEL_PACKET_DUAL, tile_order=4, packet_cull=2), "fused"),
    "dual_fused_wg256": (dict(kernel=capi.KERNEL_PACKET_DUAL, packet_wg=256), "fused"),
    "dual_fused_wg256_strips": (dict(kernel=capi.KERNEL_PACKET_DUAL, packet_wg=256, tile_order=4), "fused"),
    "dual_fused_wg64": (dict(kernel=capi.KERNEL_PACKET_DUAL, packet_wg=64), "fused"),
    "dual_fused_wg64_strips": (dict(kernel=capi.KERNEL_PACKET_DUAL, packet_wg=64, tile_order=4), "fused"),
    "dual_fused_wg64_rowmajor": (dict(kernel=capi.KERNEL_PACKET_DUAL, packet_wg=64, tile_order=1), "fused"),
    "asm_fused_strips": (dict(kernel=capi.KERNEL_PACKET_ASM, tile_order=4), "fused"),
    "quad_fused": (dict(kernel=capi.KERNEL_PACKET_QUAD), "fused"),
    "quad_cast": (dict(kernel=capi.KERNEL_PACKET_QUAD), "cast"),
    "auto_cast": (dict(), "cast"),
    "auto_tiled": (dict(), "tiled"),
    "persist2_linear": (dict(kernel=capi.KERNEL_LANE_PERSISTENT), "cast"),
    "persist4_linear": (dict(kernel=capi.KERNEL_LANE4_PERSISTENT), "cast"),
    "persist8_linear": (dict(kernel=capi.KERNEL_LANE8_PERSISTENT), "cast"),
    "persist4_linear_l32": (dict(kernel=capi.KERNEL_LANE4_PERSISTENT, leaf_wait=32), "cast"),
    "persist4_linear_l8": (dict(kernel=capi.KERNEL_LANE4_PERSISTENT, leaf_wait=8), "cast"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="")
    ap.add_argument("--count", action="store_true", help="also run the counting variants once")
    a = ap.parse_args()
    names = a.variants.split(",") if a.variants else list(VARIANTS)
    cfg = synth.CONFIGS[a.config]
    w, h = cfg["grid"]
    scene = capi.Scene(synth.scene_vertices(cfg))
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    ctxs = {}
    for n in names:
        kw, entry = VARIANTS[n]
        c = capi.Context(0, **kw)
        scene.upload(c)
        ctxs[n] = (c, entry)
    base = ctxs[names[0]][0]
    d_rays = base.device_alloc(w * h * 32)
    d_hits = base.device_alloc(w * h * 32)
    base.generate_grid(cam, w, h, 0, h, d_rays)
    flags = capi.FLAG_COHERENT | capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE

    def run(c, entry):
        if entry == "cast":
            c.cast(d_rays, d_hits, count=w * h, flags=flags)
        elif entry == "tiled":
            c.cast_tiled(d_rays, d_hits, w, h)
        else:
            c.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
        return c.stats()["last_trace_ms"]

    ref_hash = None
    out = np.zeros(w * h, dtype=T.HIT32)
    res = {n: [] for n in names}
    for n in names:  # warm-up + correctness
        c, entry = ctxs[n]
        run(c, entry)
        base.d2h(out, d_hits)
        hsh = hashlib.sha256(out.tobytes()).hexdigest()
        if ref_hash is None:
            ref_hash, ref = hsh, out.copy()
        elif hsh != ref_hash:
            diff = np.nonzero(out["prim_id"] != ref["prim_id"])[0]
            tdiff = np.nonzero(out["t"] != ref["t"])[0]
            ties = int((out["t"][diff] == ref["t"][diff]).sum())
            print(f"!! {n}: result differs from {names[0]}: {diff.size} prim ({ties} exact ties), {tdiff.size} t", flush=True)
    for _ in range(a.rounds):
        for n in names:
            c, entry = ctxs[n]
            res[n].append(run(c, entry))
    print(f"{'variant':28s} {'min ms':>9s} {'median ms':>10s} {'Mrays/s(med)':>13s}")
    summary = {}
    for n in names:
        v = np.array(res[n])
        summary[n] = dict(min_ms=float(v.min()), median_ms=float(np.median(v)), mrays=w * h / np.median(v) / 1e3)
        print(f"{n:28s} {v.min():9.3f} {np.median(v):10.3f} {w * h / np.median(v) / 1e3:13.1f}", flush=True)
    if a.count:
        for kern in (capi.KERNEL_PACKET_DUAL, capi.KERNEL_PACKET_QUAD):
            c = capi.Context(0, kernel=kern, count_visits=True)
            scene.upload(c)
            c.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
            s = c.stats()
            print("counting", capi.kernel_name(s["last_kernel"]), "node rows/packet %.1f tri rows/packet %.1f max_stack %d" % (
                s["wave_node_fetches"] / (w * h / 128), s["wave_tri_fetches"] / (w * h / 128), s["max_stack_depth"]), flush=True)
            c.close()
        for kern in (capi.KERNEL_LANE, capi.KERNEL_PACKET):
            c = capi.Context(0, kernel=kern, count_visits=True)
            scene.upload(c)
            c.cast_tiled(d_rays, d_hits, w, h)
            s = c.stats()
            print("counting", {1: "lane", 2: "packet"}[kern], "nodes/ray %.1f tris/ray %.2f max_stack %d dead_pops/ray %.2f" % (
                s["bvh_nodes_visited"] / s["rays_cast"], s["tri_tests"] / s["rays_cast"], s["max_stack_depth"], s["dead_pops"] / s["rays_cast"]), flush=True)
            c.close()
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
