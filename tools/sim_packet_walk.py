"""CPU model of the shared packet walk (packet_rows_kernel.h) over the 2-wide rows and over a 4-wide collapse of the
same tree: how many rows a 128-ray packet fetches, how many child boxes it tests, how many entries it pushes.
A planning tool (numpy, float32 but not the canonical operation order): counts, not results.

    python tools/sim_packet_walk.py --config C3 --packets 64
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from messyerraytracer_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

LEAF = 0x80000000


def build_tree(cfg):
    v = synth.scene_vertices(cfg)
    tris = po.make_triangles(v)
    nodes, prim, used = po.bvh2_build(po.verts4(v))
    wide, leaf_tris = po.to_wide(tris, nodes, prim)
    return wide, leaf_tris


def children2(wide, i):
    w = wide[i]
    out = []
    for side in ("left", "right"):
        cnt = int(w[side + "_count"])
        idx = int(w[side + "_idx"])
        out.append((np.concatenate([w[side + "_min"], w[side + "_max"]]), (LEAF | idx, cnt) if cnt else (idx, 0)))
    return out


def collapse4(wide):
    """scene_prep.cpp's rule: open the inner child with the largest half-area until there are four"""
    def area(b):
        e = b[3:] - b[:3]
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]
    nodes4 = {}
    work = [0]
    while work:
        w2 = work.pop()
        ch = children2(wide, w2)
        while len(ch) < 4:
            best, best_a = -1, -1.0
            for k, (b, (ref, cnt)) in enumerate(ch):
                if not (ref & LEAF) and area(b) > best_a:
                    best, best_a = k, area(b)
            if best < 0:
                break
            l, r = children2(wide, ch[best][1][0])
            ch[best] = l
            ch.append(r)
        nodes4[w2] = ch
        for b, (ref, cnt) in ch:
            if not (ref & LEAF):
                work.append(ref)
    return nodes4


def slab(box, o, inv, tmin, lim):
    t0 = (box[None, :3] - o) * inv
    t1 = (box[None, 3:] - o) * inv
    tn = np.minimum(t0, t1)
    tf = np.maximum(t0, t1)
    te = np.maximum(tn.max(axis=1), tmin)
    tx = np.minimum(tf.min(axis=1), lim)
    return te, tx, te <= tx


def tri_test(tri, o, d, tmin, lim, own):
    e1, e2, v0 = tri["edge1"], tri["edge2"], tri["v0"]
    pv = np.cross(d, e2[None, :])
    det = pv @ e1
    ok = own & (np.abs(det) >= 1e-8)
    with np.errstate(all="ignore"):
        inv = 1.0 / det
        tv = o - v0[None, :]
        u = (tv * pv).sum(axis=1) * inv
        qv = np.cross(tv, e1[None, :])
        vv = (d * qv).sum(axis=1) * inv
        t = (qv @ e2) * inv
    ok &= (u >= 0) & (u <= 1) & (vv >= 0) & (u + vv <= 1) & (t >= tmin) & (t < lim)
    return ok, t


def walk(get_children, leaf_tris, o, d, tmin, tmax, order, grpA=None):
    inv = 1.0 / np.where(np.abs(d) < 1e-12, 1e-12, d)
    lim = tmax.copy()
    n = o.shape[0]
    st = dict(node_rows=0, boxes=0, tri_rows=0, pushes=0, max_stack=0)
    stack = []
    cur = (0, 0, np.ones(n, bool))
    while cur is not None:
        ref, cnt, own = cur
        if ref & LEAF:
            first = ref & ~LEAF
            for k in range(cnt):
                st["tri_rows"] += 1
                ok, t = tri_test(leaf_tris[first + k], o, d, tmin, lim, own)
                lim = np.where(ok, t, lim)
            cur = stack.pop() if stack else None
            continue
        st["node_rows"] += 1
        if grpA is not None:
            a, b = (own & grpA).any(), (own & ~grpA).any()
            st["one_group_steps"] = st.get("one_group_steps", 0) + (0 if (a and b) else 1)
        hit = []
        for box, (cref, ccnt) in get_children(ref):
            st["boxes"] += 1
            te, tx, m = slab(box, o, inv, tmin, lim)
            if m.any():
                key = te[m].min() if order == "min" else (te[0] if m[0] else te[m][0])
                hit.append((key, cref, ccnt, m))
        if not hit:
            cur = stack.pop() if stack else None
            continue
        if order != "fixed":
            hit.sort(key=lambda h: h[0])
        for h in reversed(hit[1:]):
            stack.append((h[1], h[2], h[3]))
            st["pushes"] += 1
        st["max_stack"] = max(st["max_stack"], len(stack))
        st["hits%d" % min(len(hit), 3)] = st.get("hits%d" % min(len(hit), 3), 0) + 1
        cur = (hit[0][1], hit[0][2], hit[0][3])
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--packets", type=int, default=48)
    ap.add_argument("--tile", default="16x8")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.config]
    wide, leaf_tris = build_tree(cfg)
    print("tree built:", wide.shape[0], "wide nodes", flush=True)
    W, H = cfg["grid"]
    tw, th = (int(x) for x in a.tile.split("x"))
    rng = np.random.default_rng(5)
    n4cache = {}

    def get2(ref):
        return children2(wide, ref)

    def get4(ref):
        if ref not in n4cache:
            ch = children2(wide, ref)
            def area(b):
                e = b[3:] - b[:3]
                return e[0] * e[1] + e[1] * e[2] + e[2] * e[0]
            while len(ch) < 4:
                best, best_a = -1, -1.0
                for k, (b, (r, c)) in enumerate(ch):
                    if not (r & LEAF) and area(b) > best_a:
                        best, best_a = k, area(b)
                if best < 0:
                    break
                l, r = children2(wide, ch[best][1][0])
                ch[best] = l
                ch.append(r)
            n4cache[ref] = ch
        return n4cache[ref]

    tot = {k: dict(node_rows=0, boxes=0, tri_rows=0, pushes=0, max_stack=0) for k in ("bvh2", "bvh4_lane0", "bvh4_fixed")}
    for p in range(a.packets):
        tx0 = int(rng.integers(0, W // tw)) * tw
        ty0 = int(rng.integers(0, H // th)) * th
        rows = po.grid_rays(cfg["origin"], cfg["forward"], W, H, cfg["fov"], y0=ty0, y1=ty0 + th)
        rows = rows.reshape(th, W)[:, tx0:tx0 + tw].reshape(-1)
        o = rows["origin"].astype(np.float32)
        d = rows["direction"].astype(np.float32)
        tmin, tmax = rows["t_min"].astype(np.float32), rows["t_max"].astype(np.float32)
        for name, get, order in (("bvh2", get2, "lane0"), ("bvh4_lane0", get4, "lane0"), ("bvh4_fixed", get4, "fixed")):
            s = walk(get, leaf_tris, o, d, tmin, tmax, order, grpA=(np.arange(tw * th) % tw) < tw // 2)
            for k in s:
                tot[name][k] = max(tot[name].get(k, 0), s[k]) if k == "max_stack" else tot[name].get(k, 0) + s[k]
    out = {"config": a.config, "packets": a.packets, "tile": a.tile,
           "per_packet": {n: {k: (v if k == "max_stack" else v / a.packets) for k, v in t.items()} for n, t in tot.items()}}
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
