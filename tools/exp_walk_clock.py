#!/usr/bin/env python3
"""Where a packet's wave spends its cycles: the counting build of trace_packet_rows_kernel<.., 1> clocks every row
fetch with s_memtime (issue -> data there) and the whole kernel body.

    python tools/exp_walk_clock.py --config C3
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    w, h = cfg["grid"]
    scene = capi.Scene(synth.scene_vertices(cfg))
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    out = {}
    for name, kern, mode in (("rows1_clock", capi.KERNEL_PACKET_ROWS, 2), ("rows2_clock", capi.KERNEL_PACKET_DUAL, 2), ("quad_clock", capi.KERNEL_PACKET_QUAD, 2),
                             ("rows1_count", capi.KERNEL_PACKET_ROWS, 1), ("rows2_count", capi.KERNEL_PACKET_DUAL, 1), ("quad_count", capi.KERNEL_PACKET_QUAD, 1)):
        ctx = capi.Context(0, kernel=kern, count_visits=mode)
        scene.upload(ctx)
        d_hits = ctx.device_alloc(w * h * 32)
        ctx.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
        ctx.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
        s = ctx.stats()
        waves = max(1, s["waves"])
        out[name] = dict(kernel=capi.kernel_name(s["last_kernel"]), trace_ms=s["last_trace_ms"], rays=s["rays_cast"],
                         node_rows=s["wave_node_fetches"], tri_rows=s["wave_tri_fetches"], waves=s["waves"],
                         fetch_wait_cycles_per_wave=s["fetch_wait_cycles"] / waves, wave_cycles_per_wave=s["wave_cycles"] / waves,
                         fetches_per_wave=(s["wave_node_fetches"] + s["wave_tri_fetches"]) / max(1, s["waves"]) if s["waves"] else None)
        if s["waves"]:
            f = (s["wave_node_fetches"] + s["wave_tri_fetches"])
            out[name]["cycles_per_fetch_wait"] = s["fetch_wait_cycles"] / max(1, f)
            out[name]["wait_share"] = s["fetch_wait_cycles"] / max(1, s["wave_cycles"])
        print(name, json.dumps(out[name]), flush=True)
        ctx.device_free(d_hits)
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
