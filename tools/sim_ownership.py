"""CPU model of the 128-ray shared walk (packet_rows_kernel.h, MRT_ROWS_LOOPW): who needs each box test.

For sampled 16x8-pixel packets (group A = left 8x8 tile, group B = right one) of a config's primary grid it walks the
product's own BVH2 with per-lane ownership masks as the kernel does, and records per node step
  * popcount(own A), popcount(own B) (histogram, also by tree depth),
  * for each (group, child box) slab test: how many of the group's lanes hit (0 = the whole vector test was wasted),
  * whether the box lies outside the packet's pyramid (what a packet-level cull could skip), for the 16x8 pyramid
    and for each group's own 8x8 pyramid,
next to the one-ray walks of the same rays (what each ray needs on its own).  numpy float32, not the canonical
operation order: a planning tool for counts, not for results.

    python tools/sim_ownership.py --config C3 --packets 200 --out profiles/r03_ownership_sim_c3.json
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(__file__))
from messyerraytracer_amd import synth  # noqa: E402
from sim_packet_walk import LEAF, build_tree, children2, grid_rays, slab, tri_test  # noqa: E402


def pyramid(o, d, idx):
    """inward unit normals of the four side planes through the apex and the corner rays idx = (tl, tr, br, bl)"""
    c = d[list(idx)].astype(np.float64)
    cen = c.sum(axis=0)
    cen /= np.linalg.norm(cen)
    ns = []
    for k in range(4):
        n = np.cross(c[k], c[(k + 1) % 4])
        n /= np.linalg.norm(n)
        if n @ cen < 0:
            n = -n
        ns.append(n + 5e-5 * cen)
    return np.array(ns), o[0].astype(np.float64)


def outside(planes, apex, box):
    """True if the box lies wholly outside one of the planes (its farthest corner along the inward normal is outside)"""
    for n in planes:
        corner = np.where(n >= 0, box[3:], box[:3]).astype(np.float64)
        if n @ (corner - apex) < -1e-6:
            return True
    return False


def one_ray_walks(get_children, leaf_tris, o, d, tmin, tmax):
    """the per-ray ordered walk of every ray: node visits, box tests, triangle tests"""
    n = o.shape[0]
    inv = 1.0 / np.where(np.abs(d) < 1e-12, 1e-12, d)
    out = np.zeros((n, 3), np.int64)
    for r in range(n):
        oo, dd, ii = o[r:r + 1], d[r:r + 1], inv[r:r + 1]
        lim = tmax[r:r + 1].copy()
        tm = tmin[r:r + 1]
        stack = []
        cur = (0, 0)
        nv = nb = nt = 0
        while cur is not None:
            ref, cnt = cur
            if ref & LEAF:
                first = ref & ~LEAF
                for k in range(cnt):
                    nt += 1
                    ok, t = tri_test(leaf_tris[first + k], oo, dd, tm, lim, np.ones(1, bool))
                    if ok[0]:
                        lim = t.astype(np.float32)
                cur = stack.pop() if stack else None
                continue
            nv += 1
            hit = []
            for box, (cref, ccnt) in get_children(ref):
                nb += 1
                te, tx, m = slab(box, oo, ii, tm, lim)
                if m[0]:
                    hit.append((te[0], cref, ccnt))
            if not hit:
                cur = stack.pop() if stack else None
                continue
            hit.sort(key=lambda h: h[0])
            for h in reversed(hit[1:]):
                stack.append((h[1], h[2]))
            cur = (hit[0][1], hit[0][2])
        out[r] = (nv, nb, nt)
    return out


def packet_walk(get_children, leaf_tris, o, d, tmin, tmax, grp_a, pyr_all, pyr_a, pyr_b, rec):
    inv = 1.0 / np.where(np.abs(d) < 1e-12, 1e-12, d)
    lim = tmax.copy()
    n = o.shape[0]
    grp_b = ~grp_a
    stack = []
    cur = (0, 0, np.ones(n, bool), 0)
    while cur is not None:
        ref, cnt, own, depth = cur
        if ref & LEAF:
            first = ref & ~LEAF
            for k in range(cnt):
                rec["tri_rows"] += 1
                for g in (grp_a, grp_b):
                    if (own & g).any():
                        rec["tri_group_tests"] += 1
                        rec["tri_lanes_owning"] += int((own & g).sum())
                ok, t = tri_test(leaf_tris[first + k], o, d, tmin, lim, own)
                lim = np.where(ok, t, lim).astype(np.float32)
            cur = stack.pop() if stack else None
            continue
        rec["node_rows"] += 1
        pa, pb = int((own & grp_a).sum()), int((own & grp_b).sum())
        rec["own_hist"][pa] += 1
        rec["own_hist"][pb] += 1
        dd = min(depth, 39)
        rec["depth_steps"][dd] += 1
        rec["depth_own"][dd] += pa + pb
        hit = []
        for box, (cref, ccnt) in get_children(ref):
            te, tx, m = slab(box, o, inv, tmin, lim)
            m &= own
            out_all = outside(pyr_all[0], pyr_all[1], box)
            for g, pyr, pop in ((grp_a, pyr_a, pa), (grp_b, pyr_b, pb)):
                if pop == 0:
                    rec["group_box_skipped_unowned"] += 1
                    continue
                h = int((m & g).sum())
                rec["group_box_tests"] += 1
                rec["hit_hist"][h] += 1
                rec["lanes_hit"] += h
                rec["lanes_owning"] += pop
                if h == 0:
                    rec["allmiss"] += 1
                    if out_all:
                        rec["allmiss_outside_pyr16x8"] += 1
                    if outside(pyr[0], pyr[1], box):
                        rec["allmiss_outside_pyr8x8"] += 1
                elif out_all:
                    rec["UNSOUND_cull"] += 1
            if m.any():
                key = te[0] if m[0] else te[m][0]
                hit.append((key, cref, ccnt, m))
        if not hit:
            cur = stack.pop() if stack else None
            continue
        hit.sort(key=lambda h: h[0])
        for h in reversed(hit[1:]):
            stack.append((h[1], h[2], h[3], depth + 1))
        cur = (hit[0][1], hit[0][2], hit[0][3], depth + 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--packets", type=int, default=100)
    ap.add_argument("--grid", default=None, help="WxH instead of the config's grid")
    ap.add_argument("--rays", type=int, default=16, help="one-ray walks per packet (sampled lanes)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    wide, leaf_tris = build_tree(cfg)
    W, H = (int(x) for x in a.grid.split("x")) if a.grid else cfg["grid"]
    rng = np.random.default_rng(11)
    cache = {}

    def get2(ref):
        if ref not in cache:
            cache[ref] = children2(wide, ref)
        return cache[ref]

    rec = dict(node_rows=0, tri_rows=0, tri_group_tests=0, tri_lanes_owning=0, group_box_tests=0, group_box_skipped_unowned=0,
               lanes_hit=0, lanes_owning=0, allmiss=0, allmiss_outside_pyr16x8=0, allmiss_outside_pyr8x8=0, UNSOUND_cull=0,
               own_hist=[0] * 65, hit_hist=[0] * 65, depth_steps=[0] * 40, depth_own=[0] * 40)
    one = np.zeros(3, np.float64)
    n_one = 0
    tw, th = 16, 8
    grp_a = (np.arange(tw * th) % tw) < 8
    for p in range(a.packets):
        x0 = int(rng.integers(0, W // tw)) * tw
        y0 = int(rng.integers(0, H // th)) * th
        o, d, tmin, tmax = grid_rays(cfg, W, H, x0, y0, tw, th)
        corners = lambda xa, xb: (0 * tw + xa, 0 * tw + xb, (th - 1) * tw + xb, (th - 1) * tw + xa)
        # pyramids through the corner RAYS pushed outwards (as cull_setup does); pixel-centre rays, so every ray is inside
        pyr_all = pyramid(o, d, corners(0, tw - 1))
        pyr_a = pyramid(o, d, corners(0, 7))
        pyr_b = pyramid(o, d, corners(8, 15))
        packet_walk(get2, leaf_tris, o, d, tmin, tmax, grp_a, pyr_all, pyr_a, pyr_b, rec)
        sel = rng.choice(tw * th, size=min(a.rays, tw * th), replace=False)
        w = one_ray_walks(get2, leaf_tris, o[sel], d[sel], tmin[sel], tmax[sel])
        one += w.sum(axis=0)
        n_one += len(sel)
        if (p + 1) % 20 == 0:
            print(f"{p + 1} packets", flush=True)
    P = a.packets
    gt = rec["group_box_tests"]
    out = {
        "config": a.config, "grid": [W, H], "packets": P,
        "one_ray_walk_per_ray": {"node_visits": one[0] / n_one, "box_tests": one[1] / n_one, "tri_tests": one[2] / n_one},
        "packet_per_wave": {"node_rows": rec["node_rows"] / P, "tri_rows": rec["tri_rows"] / P,
                            "group_box_tests": gt / P, "group_box_tests_skipped_group_owns_nothing": rec["group_box_skipped_unowned"] / P,
                            "tri_group_tests": rec["tri_group_tests"] / P},
        "box_tests": {"all_lanes_miss_share": rec["allmiss"] / gt,
                      "all_miss_and_outside_16x8_pyramid_share": rec["allmiss_outside_pyr16x8"] / gt,
                      "all_miss_and_outside_own_8x8_pyramid_share": rec["allmiss_outside_pyr8x8"] / gt,
                      "mean_lanes_hit_when_some_hit": rec["lanes_hit"] / max(1, gt - rec["allmiss"]),
                      "mean_lanes_owning": rec["lanes_owning"] / gt,
                      "useful_lane_share_vs_one_ray_walk": (one[1] / n_one * 64) / (gt / P / 2 * 64) if gt else None,
                      "unsound_culls": rec["UNSOUND_cull"]},
        "own_popcount_hist_per_group_step": rec["own_hist"],
        "lanes_hit_hist_per_group_box_test": rec["hit_hist"],
        "by_depth": {"steps": rec["depth_steps"], "mean_owning_lanes_of_128": [o_ / s if s else 0 for o_, s in zip(rec["depth_own"], rec["depth_steps"])]},
    }
    s = json.dumps(out, indent=1)
    print(s)
    if a.out:
        open(a.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
