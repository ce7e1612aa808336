"""numpy mirrors of the PODs in include/mrt_hip.h.

Device layouts == GLSL std430 structs of the reference
(src/api/gpu_types.h:44-126, src/gpu/gpu_structs.h:41-47); host layouts == the
reference's Ray / Intersection / Triangle at precision=single
(src/core/ray.h:25-51, src/core/intersection.h:16-40, src/core/triangle.h:22-39).
"""
import numpy as np

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
HOST_TRI80 = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3), ("edge1", "<f4", 3),
                       ("edge2", "<f4", 3), ("normal", "<f4", 3), ("id", "<u4"), ("layers", "<u4")])

INSTANCE = np.dtype([("first_tri", "<u4"), ("n_tris", "<u4"), ("layers", "<u4"), ("reserved", "<u4"),
                     ("basis", "<f4", 9), ("origin", "<f4", 3)])  # mrt_instance

assert INSTANCE.itemsize == 64
assert RAY32.itemsize == 32 and HIT32.itemsize == 32 and TRI64.itemsize == 64
assert NODE32.itemsize == 32 and WIDE64.itemsize == 64
assert HOST_RAY60.itemsize == 60 and HOST_HIT44.itemsize == 44 and HOST_TRI80.itemsize == 80

FLT_MAX = np.float32(3.4028234663852886e38)
NO_HIT = 0xFFFFFFFF


def verts4_from_verts9(verts9: np.ndarray) -> np.ndarray:
    """(n,3,3) float32 vertices -> (3n,4) bvhvec4 array (w = 0), the input of
    tinybvh::BVH::Build (src/accel/tinybvh_adapter.h:42-55)."""
    v = np.ascontiguousarray(verts9, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((v.shape[0], 4), dtype=np.float32)
    out[:, :3] = v
    return out
