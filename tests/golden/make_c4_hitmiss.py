#!/usr/bin/env python3
"""Generates tests/golden/c4_hitmiss.npz: every ray of config C4's WHOLE batch (2^24 incoherent rays) on which the
REFERENCE (oracle/_ref = TinyBVH 1.6.7 compiled from /root/reference) and the oracle disagree about hit / miss.

Why they can: the reference's CPU path ignores Ray::t_min (TinyBVH accepts every t > 0: SURVEY.md section 0, defect 6;
tinybvh_adapter.h:81-83), the GLSL shader and everything here honour t >= t_min; C4's rays start inside the soup, so some
start within t_min of a triangle.  Such a ray is a reference hit and -- unless it hits something else further on -- a miss
here.  The few other disagreements are rays that graze an edge (the reference's approximate reciprocal against the exact
division here).  The fixture lists them all, so tests/test_parity_gpu.py can pin the device's hit count to the
reference's EXACTLY:  device hits = reference hits - (listed reference-only hits) + (listed oracle-only hits).

Run in the build container only (needs /root/reference, ~3 GB of RAM, a few minutes):
    python tests/golden/make_c4_hitmiss.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from messyerraytracer_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    assert po.ref_available(), "oracle/_ref/libmrt_ref.so missing: run `make -C oracle` in the build container"
    cfg = synth.CONFIGS["C4"]
    v = synth.scene_vertices(cfg)
    rays = synth.incoherent_rays(cfg["incoherent"], cfg["ray_seed"])
    rs = po.RefScene(v)
    ref = rs.cast_rays(rays, n_threads=8)
    rs.close()
    osc = po.OracleScene(v)
    ours = osc.trace(rays, n_threads=8)
    ref_hit, our_hit = ref["prim_id"] >= 0, ours["prim_id"] >= 0
    idx = np.nonzero(ref_hit != our_hit)[0].astype(np.int64)
    below = ref_hit[idx] & (ref["t"][idx] < rays["t_min"][idx])      # the reference's hit lies inside [0, t_min)
    # cross-check against the committed sampled fixture's digest of the same batch
    g = np.load(os.path.join(OUT, "c4_sampled.npz"))
    dg = json.loads(str(g["digest"]))
    assert int(ref_hit.sum()) == dg["hit_count"], (int(ref_hit.sum()), dg["hit_count"])
    # how many reference hits below t_min there are in all (most of those rays hit something else further on)
    n_below_all = int((ref_hit & (ref["t"] < rays["t_min"])).sum())
    meta = dict(rays=int(rays.shape[0]), reference_hits=int(ref_hit.sum()), oracle_hits=int(our_hit.sum()),
                disagreements=int(idx.size), reference_only=int((ref_hit[idx]).sum()), oracle_only=int((our_hit[idx]).sum()),
                reference_only_below_t_min=int(below.sum()), reference_hits_below_t_min_in_all=n_below_all)
    path = os.path.join(OUT, "c4_hitmiss.npz")
    np.savez_compressed(path, index=idx, ref_hit=ref_hit[idx], ref_t=ref["t"][idx], ref_prim=ref["prim_id"][idx],
                        below_t_min=below, meta=np.array(json.dumps(meta)))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", json.dumps(meta))


if __name__ == "__main__":
    main()
