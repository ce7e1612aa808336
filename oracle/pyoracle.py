"""ctypes bindings for the checkers: oracle/liboracle.so (our C restatement) and
oracle/_ref/libmrt_ref.so (the reference's TinyBVH, when it was built).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmrt_ref.so")

RAY32 = np.dtype([("origin", "<f4", 3), ("t_max", "<f4"), ("direction", "<f4", 3), ("t_min", "<f4")])
HIT32 = np.dtype([("t", "<f4"), ("prim_id", "<i4"), ("bary_u", "<f4"), ("bary_v", "<f4"),
                  ("normal", "<f4", 3), ("hit_layers", "<u4")])
TRI64 = np.dtype([("v0", "<f4", 3), ("id", "<u4"), ("edge1", "<f4", 3), ("layers", "<u4"),
                  ("edge2", "<f4", 3), ("pad2", "<f4"), ("normal", "<f4", 3), ("pad3", "<f4")])
NODE32 = np.dtype([("aabb_min", "<f4", 3), ("left_first", "<u4"), ("aabb_max", "<f4", 3), ("tri_count", "<u4")])
WIDE64 = np.dtype([("left_min", "<f4", 3), ("left_idx", "<u4"), ("left_max", "<f4", 3), ("right_idx", "<u4"),
                   ("right_min", "<f4", 3), ("left_count", "<u4"), ("right_max", "<f4", 3), ("right_count", "<u4")])
HOST_RAY60 = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("inv_direction", "<f4", 3),
                       ("dir_sign", "<i4", 3), ("t_min", "<f4"), ("t_max", "<f4"), ("flags", "<u4")])
HOST_HIT44 = np.dtype([("t", "<f4"), ("position", "<f4", 3), ("normal", "<f4", 3), ("u", "<f4"), ("v", "<f4"),
                       ("prim_id", "<u4"), ("hit_layers", "<u4")])


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("hits", C.c_uint64), ("node_visits", C.c_uint64),
                ("tri_tests", C.c_uint64), ("max_stack", C.c_uint32)]


def build(force: bool = False) -> None:
    """make -C oracle (also builds _ref when /root/reference is present)."""
    if force or not os.path.exists(ORACLE_SO):
        subprocess.run(["make", "-C", HERE], check=True, capture_output=True)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        L.orc_bvh2_build.restype = C.c_int
        L.orc_to_wide.restype = C.c_int
        L.orc_tri_test.restype = C.c_int
        L.orc_morton_key.restype = C.c_uint32
        L.orc_two_level_build.restype = C.c_void_p
        L.orc_two_level_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def make_triangles(verts9, ids=None, layers=None):
    v = np.ascontiguousarray(verts9, dtype=np.float32).reshape(-1, 9)
    n = v.shape[0]
    out = np.zeros(n, dtype=TRI64)
    ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
    lay_a = None if layers is None else np.ascontiguousarray(layers, dtype=np.uint32)
    lib().orc_make_triangles(_p(v), None if ids_a is None else _p(ids_a), None if lay_a is None else _p(lay_a),
                             C.c_uint32(n), _p(out))
    return out


def flatten_instances(verts9, instances):
    """World-space triangles (TRI64) of placed meshes: raytracer_server.cpp:700-711 restated in mrt_oracle.c.
    instances: structured array with the layout of orc_instance (64 bytes)."""
    v = np.ascontiguousarray(verts9, dtype=np.float32).reshape(-1, 9)
    inst = np.ascontiguousarray(instances)
    assert inst.dtype.itemsize == 64
    out = np.zeros(int(inst["n_tris"].sum()), dtype=TRI64)
    lib().orc_flatten_instances(_p(v), _p(inst), C.c_uint32(inst.shape[0]), _p(out))
    return out


def verts4(verts9):
    v = np.ascontiguousarray(verts9, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((v.shape[0], 4), dtype=np.float32)
    out[:, :3] = v
    return out


def bvh2_build(v4):
    n = v4.shape[0] // 3
    nodes = np.zeros(2 * n + 2, dtype=NODE32)
    prim_idx = np.zeros(n, dtype=np.uint32)
    used = C.c_uint32(0)
    rc = lib().orc_bvh2_build(_p(v4), C.c_uint32(n), _p(nodes), _p(prim_idx), C.byref(used))
    assert rc == 0, rc
    return nodes[:used.value].copy(), prim_idx, used.value


def bvh2_info(nodes):
    nc, lc, d, ml = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    sah = C.c_float()
    lib().orc_bvh2_info(_p(nodes), C.byref(nc), C.byref(lc), C.byref(d), C.byref(sah), C.byref(ml))
    return dict(node_count=nc.value, leaf_count=lc.value, depth=d.value, sah_cost=sah.value, max_leaf=ml.value)


def to_wide(tris, nodes, prim_idx):
    n = tris.shape[0]
    wide = np.zeros(max(1, nodes.shape[0]), dtype=WIDE64)
    leaf_tris = np.zeros(n, dtype=TRI64)
    nw = C.c_uint32()
    rc = lib().orc_to_wide(_p(tris), C.c_uint32(n), _p(nodes), C.c_uint32(nodes.shape[0]), _p(prim_idx),
                           _p(wide), C.byref(nw), _p(leaf_tris))
    assert rc == 0, rc
    return wide[:nw.value].copy(), leaf_tris


def trace(wide, leaf_tris, rays, query_mask=0xFFFFFFFF, any_hit=False, counters=False, n_threads=0):
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(rays.shape[0], dtype=HIT32)
    ctr = Counters()
    lib().orc_trace(_p(wide), _p(leaf_tris), _p(rays), _p(hits), C.c_uint64(rays.shape[0]),
                    C.c_uint32(query_mask), C.c_int(1 if any_hit else 0),
                    C.byref(ctr) if counters else None, C.c_int(n_threads))
    if counters:
        return hits, dict(rays=ctr.rays, hits=ctr.hits, node_visits=ctr.node_visits,
                          tri_tests=ctr.tri_tests, max_stack=ctr.max_stack)
    return hits


def trace_brute(tris, rays, query_mask=0xFFFFFFFF, any_hit=False, n_threads=0):
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(rays.shape[0], dtype=HIT32)
    lib().orc_trace_brute(_p(tris), C.c_uint32(tris.shape[0]), _p(rays), _p(hits), C.c_uint64(rays.shape[0]),
                          C.c_uint32(query_mask), C.c_int(1 if any_hit else 0), C.c_int(n_threads))
    return hits


def tri_test(tri, ray):
    t, u, v = C.c_float(), C.c_float(), C.c_float()
    tri = np.ascontiguousarray(tri)
    ray = np.ascontiguousarray(ray)
    ok = lib().orc_tri_test(_p(tri), _p(ray), C.byref(t), C.byref(u), C.byref(v))
    return bool(ok), t.value, u.value, v.value


def camera_basis(forward, w, h, fov_deg):
    f = (C.c_float * 3)(*forward)
    fwd, right, up = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    hw, hh = C.c_float(), C.c_float()
    lib().orc_camera_basis(f, C.c_uint32(w), C.c_uint32(h), C.c_float(fov_deg), fwd, right, up, C.byref(hw), C.byref(hh))
    return (np.array(fwd[:], np.float32), np.array(right[:], np.float32), np.array(up[:], np.float32),
            np.float32(hw.value), np.float32(hh.value))


def grid_rays(origin, forward, w, h, fov_deg, y0=0, y1=None):
    y1 = h if y1 is None else y1
    out = np.zeros((y1 - y0) * w, dtype=RAY32)
    o = (C.c_float * 3)(*origin)
    f = (C.c_float * 3)(*forward)
    lib().orc_grid_rays(o, f, C.c_uint32(w), C.c_uint32(h), C.c_float(fov_deg), C.c_uint32(y0), C.c_uint32(y1), _p(out))
    return out


def ray_camera_rays(origin, basis, w, h, param, ortho=False, jitter=(0.5, 0.5), y0=0, y1=None):
    """RayCamera::generate_rays (ray_camera.h:234-273): basis = 3x3 row-major camera basis; param = vertical
    fov in degrees (perspective) or Camera3D::size (orthographic)."""
    y1 = h if y1 is None else y1
    out = np.zeros((y1 - y0) * w, dtype=RAY32)
    o = (C.c_float * 3)(*origin)
    b = (C.c_float * 9)(*np.asarray(basis, dtype=np.float32).reshape(9))
    lib().orc_ray_camera_rays(o, b, C.c_uint32(w), C.c_uint32(h), C.c_float(param), C.c_int(1 if ortho else 0),
                              C.c_float(jitter[0]), C.c_float(jitter[1]), C.c_uint32(y0), C.c_uint32(y1), _p(out))
    return out


def morton_keys(rays):
    rays = np.ascontiguousarray(rays)
    keys = np.zeros(rays.shape[0], dtype=np.uint32)
    lib().orc_morton_keys(_p(rays), C.c_uint64(rays.shape[0]), _p(keys))
    return keys


def make_host_rays(rays):
    rays = np.ascontiguousarray(rays)
    out = np.zeros(rays.shape[0], dtype=HOST_RAY60)
    lib().orc_make_host_rays(_p(rays), C.c_uint64(rays.shape[0]), _p(out))
    return out


def pack_rays(host_rays):
    host_rays = np.ascontiguousarray(host_rays)
    out = np.zeros(host_rays.shape[0], dtype=RAY32)
    lib().orc_pack_rays(_p(host_rays), C.c_uint64(host_rays.shape[0]), _p(out))
    return out


def unpack_hits(hits, host_rays):
    hits = np.ascontiguousarray(hits)
    host_rays = np.ascontiguousarray(host_rays)
    out = np.zeros(hits.shape[0], dtype=HOST_HIT44)
    lib().orc_unpack_hits(_p(hits), _p(host_rays), C.c_uint64(hits.shape[0]), _p(out))
    return out


class OracleScene:
    """Scene prepared the way the reference prepares it: Triangle ctor ->
    8-bin SAH BVH2 -> Aila-Laine wide nodes."""

    def __init__(self, verts9, ids=None, layers=None):
        self.tris = make_triangles(verts9, ids, layers)
        self.v4 = verts4(verts9)
        self.nodes, self.prim_idx, self.used_nodes = bvh2_build(self.v4)
        self.wide, self.leaf_tris = to_wide(self.tris, self.nodes, self.prim_idx)

    def trace(self, rays, **kw):
        return trace(self.wide, self.leaf_tris, rays, **kw)

    def brute(self, rays, **kw):
        return trace_brute(self.tris, rays, **kw)


class OracleTwoLevelScene:
    """SceneTLAS of the reference (src/accel/scene_tlas.h:140-251): one BVH per distinct mesh in mesh
    space, one over the instances; rays go to mesh space per instance (mrt_oracle.c, two-level section)."""

    def __init__(self, verts9, instances):
        v = np.ascontiguousarray(verts9, dtype=np.float32).reshape(-1, 9)
        inst = np.ascontiguousarray(instances)
        assert inst.dtype.itemsize == 64
        self.h = lib().orc_two_level_build(_p(v), C.c_uint32(v.shape[0]), _p(inst), C.c_uint32(inst.shape[0]))
        if not self.h:
            raise ValueError("orc_two_level_build failed (bad range or singular transform)")

    def trace(self, rays, query_mask=0xFFFFFFFF, any_hit=False, threads=0):
        rays = np.ascontiguousarray(rays)
        assert rays.dtype.itemsize == 32
        hits = np.zeros(rays.shape[0], dtype=HIT32)
        lib().orc_two_level_trace(C.c_void_p(self.h), _p(rays), _p(hits), C.c_uint64(rays.shape[0]), C.c_uint32(query_mask),
                                  C.c_int(1 if any_hit else 0), C.c_int(threads))
        return hits

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_two_level_free(C.c_void_p(self.h))
            self.h = None


# ---------------------------------------------------------------------------
# The reference itself (TinyBVH 1.6.7 compiled from /root/reference).
# ---------------------------------------------------------------------------
def ref_available() -> bool:
    return os.path.exists(REF_SO)


_ref = None


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_scene_create.restype = C.c_void_p
        L.ref_bvh2_used_nodes.restype = C.c_uint32
        L.ref_has_avx2.restype = C.c_int
        _ref = L
    return _ref


class RefScene:
    """RayScene::build + cast_rays of the reference (src/accel/ray_scene.h:62-118,166-185)."""

    def __init__(self, verts9, tris=None, variants=0):
        self.v4 = verts4(verts9)
        self.n = self.v4.shape[0] // 3
        self.tris = make_triangles(verts9) if tris is None else np.ascontiguousarray(tris)
        self.h = C.c_void_p(ref().ref_scene_create(_p(self.v4), _p(self.tris), C.c_uint32(self.n), C.c_uint32(variants)))

    def close(self):
        if self.h:
            ref().ref_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def bvh2(self):
        used = ref().ref_bvh2_used_nodes(self.h)
        nodes = np.zeros(used, dtype=NODE32)
        prim_idx = np.zeros(self.n, dtype=np.uint32)
        ref().ref_bvh2_copy(self.h, _p(nodes), _p(prim_idx))
        return nodes, prim_idx, used

    def bvh2_info(self):
        nc, lc = C.c_int32(), C.c_int32()
        sah = C.c_float()
        ref().ref_bvh2_info(self.h, C.byref(nc), C.byref(lc), C.byref(sah))
        return dict(node_count=nc.value, leaf_count=lc.value, sah_cost=sah.value)

    def cast_rays(self, rays, query_mask=0xFFFFFFFF, variant=0, n_threads=1):
        rays = np.ascontiguousarray(rays)
        hits = np.zeros(rays.shape[0], dtype=HIT32)
        ref().ref_cast_rays(self.h, _p(rays), _p(hits), C.c_int64(rays.shape[0]), C.c_uint32(query_mask),
                            C.c_int(variant), C.c_int(n_threads))
        return hits

    def any_hit(self, rays, variant=0, n_threads=1):
        rays = np.ascontiguousarray(rays)
        out = np.zeros(rays.shape[0], dtype=np.uint8)
        ref().ref_any_hit(self.h, _p(rays), _p(out), C.c_int64(rays.shape[0]), C.c_int(variant), C.c_int(n_threads))
        return out.astype(bool)
