"""ctypes binding of the C-ABI in include/mrt_hip.h (libmrt_hip.so).

This is plumbing for tests and bench.py; the product is the shared library.
There is no fallback: if the library is missing or fails to load, importing
callers get an ImportError / MrtError, never a CPU path.
"""
import ctypes as C
import os

import numpy as np

from . import types as T

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRT_LIB_PATH", os.path.join(HERE, "libmrt_hip.so"))  # override: A/B builds in tools/

MRT_OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_NO_SCENE, ERR_PENDING, ERR_NOT_PENDING, ERR_OOM, ERR_UNSUPPORTED, ERR_BAD_BVH = range(1, 10)
MODE_NEAREST, MODE_ANY_HIT = 0, 1
FLAG_COHERENT, FLAG_RAYS_ON_DEVICE, FLAG_HITS_ON_DEVICE, FLAG_HOST_LAYOUT, FLAG_BOOL_OUT, FLAG_FORCE_SORT, FLAG_TOKEN_OUT, FLAG_ASYNC = (1 << i for i in range(8))
TOKEN_MISS = 0xFFFFFFFF
BUILD_TRIS_ON_DEVICE, BUILD_SAFE_HANDOFF, BUILD_BLAS_ON_DEVICE, BUILD_PLOC, BUILD_SAH = 1, 2, 4, 8, 16
KERNEL_AUTO, KERNEL_LANE, KERNEL_PACKET = 0, 1, 2   # (3 and 4: retired experiments, ids not reused)
KERNEL_PACKET_ASM, KERNEL_LANE_PERSISTENT, KERNEL_LANE4_PERSISTENT, KERNEL_LANE8_PERSISTENT, KERNEL_PACKET_DUAL, KERNEL_PACKET_ROWS, KERNEL_PACKET_QUAD = 5, 6, 7, 8, 9, 10, 11
KERNEL_TWO_LEVEL, KERNEL_TWO_LEVEL_PACKET, KERNEL_TWO_LEVEL_PERSISTENT, KERNEL_TWO_LEVEL_PERSISTENT8 = 100, 101, 102, 103  # reported only

# every entry point include/mrt_hip.h declares (tests check they are all exported)
SYMBOLS = [
    "mrt_create", "mrt_destroy", "mrt_last_error", "mrt_status_string", "mrt_version", "mrt_set_stream",
    "mrt_synchronize", "mrt_make_triangles", "mrt_pack_host_triangles", "mrt_bvh2_build", "mrt_bvh2_save", "mrt_bvh2_load", "mrt_upload_scene",
    "mrt_build_scene_device", "mrt_flatten_instances", "mrt_build_instanced_scene_device", "mrt_upload_two_level_scene", "mrt_update_instances", "mrt_two_level_prepare_host", "mrt_two_level_host_arrays", "mrt_two_level_free_host", "mrt_is_available", "mrt_scene_info", "mrt_cast", "mrt_submit", "mrt_collect", "mrt_has_pending",
    "mrt_camera_look", "mrt_camera_perspective", "mrt_camera_orthographic", "mrt_generate_grid", "mrt_cast_grid", "mrt_cast_tiled", "mrt_expand_tokens",
    "mrt_expand_grid_tokens", "mrt_token_bytes", "mrt_morton_keys",
    "mrt_kernel_name", "mrt_struct_size", "mrt_get_stats", "mrt_last_kernel_variant", "mrt_kernel_available", "mrt_device_alloc", "mrt_device_free", "mrt_memcpy_h2d", "mrt_memcpy_d2h",
    "mrt_group_create", "mrt_group_destroy", "mrt_group_size", "mrt_group_context", "mrt_group_last_error", "mrt_group_row_block",
    "mrt_group_upload_scene", "mrt_group_upload_two_level_scene", "mrt_group_cast_grid",
]


class MrtError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"mrt status {status}: {msg}")
        self.status = status


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kernel", C.c_uint32), ("count_visits", C.c_uint32),
                ("sort_threshold", C.c_uint32), ("grid_tile", C.c_uint32), ("tile_w_log2", C.c_uint32),
                ("xcd_swizzle", C.c_uint32), ("stack_override", C.c_uint32), ("tile_order", C.c_uint32),
                ("sort_key", C.c_uint32), ("refill", C.c_uint32), ("leaf_wait", C.c_uint32),
                ("extra_lds", C.c_uint32), ("packet_wg", C.c_uint32), ("packet_cull", C.c_uint32), ("tile_schedule", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("fwd", C.c_float * 3), ("right", C.c_float * 3), ("up", C.c_float * 3),
                ("half_w", C.c_float), ("half_h", C.c_float), ("t_min", C.c_float), ("t_max", C.c_float),
                ("kind", C.c_uint32), ("inv_w", C.c_float), ("inv_h", C.c_float), ("jitter_x", C.c_float), ("jitter_y", C.c_float),
                ("reserved", C.c_uint32 * 3)]


class Stats(C.Structure):
    _fields_ = [("rays_cast", C.c_uint64), ("tri_tests", C.c_uint64), ("bvh_nodes_visited", C.c_uint64),
                ("hits", C.c_uint64), ("last_trace_ms", C.c_float), ("last_sort_ms", C.c_float),
                ("last_h2d_ms", C.c_float), ("last_d2h_ms", C.c_float), ("last_kernel_launches", C.c_uint32),
                ("max_stack_depth", C.c_uint32), ("dead_pops", C.c_uint64), ("detected_grid_w", C.c_uint32), ("reserved", C.c_uint32),
                ("last_build_ms", C.c_float), ("last_kernel", C.c_uint32),
                ("wave_node_fetches", C.c_uint64), ("wave_tri_fetches", C.c_uint64), ("leaf_box_checks", C.c_uint64),
                ("fetch_wait_cycles", C.c_uint64), ("wave_cycles", C.c_uint64), ("waves", C.c_uint64)]


_lib = None


def load():
    """Loads libmrt_hip.so; raises ImportError if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP extension is the product; there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    L.mrt_last_error.restype = C.c_char_p
    L.mrt_status_string.restype = C.c_char_p
    L.mrt_version.restype = C.c_uint32
    L.mrt_struct_size.restype = C.c_uint32
    L.mrt_struct_size.argtypes = [C.c_uint32]
    L.mrt_kernel_name.restype = C.c_char_p
    L.mrt_kernel_name.argtypes = [C.c_uint32]
    L.mrt_token_bytes.restype = C.c_uint32
    L.mrt_token_bytes.argtypes = [C.c_void_p]
    L.mrt_last_kernel_variant.restype = C.c_char_p
    L.mrt_last_kernel_variant.argtypes = [C.c_void_p]
    L.mrt_create.argtypes = [C.c_int, C.POINTER(Options), C.POINTER(C.c_void_p)]
    L.mrt_destroy.argtypes = [C.c_void_p]
    L.mrt_destroy.restype = None
    L.mrt_last_error.argtypes = [C.c_void_p]
    L.mrt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.mrt_synchronize.argtypes = [C.c_void_p]
    L.mrt_make_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.mrt_pack_host_triangles.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.mrt_bvh2_build.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
    L.mrt_bvh2_save.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.mrt_bvh2_load.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    L.mrt_upload_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    L.mrt_build_scene_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    L.mrt_flatten_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.mrt_build_instanced_scene_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    L.mrt_upload_two_level_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    L.mrt_update_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.mrt_is_available.argtypes = [C.c_void_p]
    L.mrt_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mrt_cast.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32]
    L.mrt_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32]
    L.mrt_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.mrt_has_pending.argtypes = [C.c_void_p]
    L.mrt_camera_look.argtypes = [C.POINTER(Camera), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_float]
    L.mrt_camera_perspective.argtypes = [C.POINTER(Camera), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_float]
    L.mrt_camera_orthographic.argtypes = [C.POINTER(Camera), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_float]
    L.mrt_generate_grid.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.mrt_cast_grid.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_void_p, C.c_uint32, C.c_int, C.c_uint32]
    L.mrt_cast_tiled.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.mrt_expand_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    L.mrt_expand_grid_tokens.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    L.mrt_morton_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.mrt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.mrt_device_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.mrt_device_free.argtypes = [C.c_void_p, C.c_void_p]
    L.mrt_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.mrt_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.mrt_group_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(Options), C.POINTER(C.c_void_p)]
    L.mrt_group_destroy.argtypes = [C.c_void_p]
    L.mrt_group_destroy.restype = None
    L.mrt_group_size.argtypes = [C.c_void_p]
    L.mrt_group_context.argtypes = [C.c_void_p, C.c_int]
    L.mrt_group_context.restype = C.c_void_p
    L.mrt_group_last_error.argtypes = [C.c_void_p]
    L.mrt_group_last_error.restype = C.c_char_p
    L.mrt_group_row_block.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mrt_group_row_block.restype = None
    L.mrt_group_upload_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    L.mrt_group_upload_two_level_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    L.mrt_group_cast_grid.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32]
    _lib = L
    return L


def _np(a):
    return a.ctypes.data_as(C.c_void_p)


def _ptr(x):
    """numpy array -> host pointer; int -> raw (device) pointer; torch tensor -> data_ptr()."""
    if isinstance(x, np.ndarray):
        return _np(x)
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError(type(x))


# ---- host-side helpers (no device needed) ---------------------------------------
def make_triangles(verts9, ids=None, layers=None) -> np.ndarray:
    v = np.ascontiguousarray(verts9, dtype=np.float32).reshape(-1, 9)
    out = np.zeros(v.shape[0], dtype=T.TRI64)
    ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
    lay_a = None if layers is None else np.ascontiguousarray(layers, dtype=np.uint32)
    rc = load().mrt_make_triangles(_np(v), None if ids_a is None else _np(ids_a), None if lay_a is None else _np(lay_a),
                                   v.shape[0], _np(out))
    if rc:
        raise MrtError(rc, "mrt_make_triangles")
    return out


def bvh2_build(verts4: np.ndarray, n_threads: int = 0):
    v4 = np.ascontiguousarray(verts4, dtype=np.float32)
    n = v4.shape[0] // 3
    nodes = np.zeros(2 * n + 2, dtype=T.NODE32)
    prim_idx = np.zeros(n, dtype=np.uint32)
    used = C.c_uint32(0)
    rc = load().mrt_bvh2_build(_np(v4), n, _np(nodes), _np(prim_idx), C.byref(used), n_threads)
    if rc:
        raise MrtError(rc, "mrt_bvh2_build")
    return nodes[:used.value].copy(), prim_idx, used.value


def bvh2_save(path: str, nodes: np.ndarray, prim_idx: np.ndarray) -> None:
    """BVH cache file (the counterpart of tinybvh::BVH::Save, tiny_bvh.h:1747-1758)."""
    nodes = np.ascontiguousarray(nodes)
    prim_idx = np.ascontiguousarray(prim_idx, dtype=np.uint32)
    assert nodes.dtype == T.NODE32
    rc = load().mrt_bvh2_save(os.fsencode(path), _np(nodes), nodes.shape[0], _np(prim_idx), prim_idx.shape[0])
    if rc:
        raise MrtError(rc, "mrt_bvh2_save")


def bvh2_load(path: str, n_tris: int):
    """Load a BVH saved for exactly n_tris triangles (BVH::Load, tiny_bvh.h:1770-1799); MrtError(ERR_BAD_BVH)
    if the file is for another triangle count, another version, or damaged."""
    nodes = np.zeros(2 * n_tris, dtype=T.NODE32)
    prim_idx = np.zeros(n_tris, dtype=np.uint32)
    used = C.c_uint32(0)
    rc = load().mrt_bvh2_load(os.fsencode(path), n_tris, _np(nodes), _np(prim_idx), C.byref(used))
    if rc:
        raise MrtError(rc, "mrt_bvh2_load")
    return nodes[:used.value].copy(), prim_idx, used.value


def kernel_available(kernel_id: int) -> bool:
    """Whether this build of libmrt_hip.so contains the kernel (KERNEL_PACKET_QUAD: only with MRT_WITH_QUAD=1)."""
    return bool(load().mrt_kernel_available(C.c_uint32(kernel_id)))


def kernel_name(kernel_id: int) -> str:
    return load().mrt_kernel_name(kernel_id).decode()


def camera_look(origin, forward, grid_w, grid_h, fov_degrees) -> Camera:
    cam = Camera()
    o = (C.c_float * 3)(*origin)
    f = (C.c_float * 3)(*forward)
    rc = load().mrt_camera_look(C.byref(cam), o, f, grid_w, grid_h, fov_degrees)
    if rc:
        raise MrtError(rc, "mrt_camera_look")
    return cam


def ray_camera(origin, basis, width, height, param, ortho=False, jitter=(0.5, 0.5)) -> Camera:
    """RayCamera::setup (ray_camera.h:50-76): basis = 3x3 row-major camera basis; param = vertical fov in degrees
    (perspective) or Camera3D::size (orthographic); jitter = sub-pixel offset of generate_rays_tile_jittered."""
    cam = Camera()
    o = (C.c_float * 3)(*origin)
    b = (C.c_float * 9)(*np.asarray(basis, dtype=np.float32).reshape(9))
    fn = load().mrt_camera_orthographic if ortho else load().mrt_camera_perspective
    rc = fn(C.byref(cam), o, b, width, height, param)
    if rc:
        raise MrtError(rc, "mrt_camera_orthographic" if ortho else "mrt_camera_perspective")
    cam.jitter_x, cam.jitter_y = jitter
    return cam


class Context:
    """mrt_ctx wrapper: one per GPU, externally serialised (SURVEY 8(b) threading)."""

    def __init__(self, device: int = 0, kernel: int = KERNEL_AUTO, count_visits: bool = False,
                 sort_threshold: int = 0, grid_tile: int = 0, tile_w_log2: int = 0, xcd_swizzle: int = 0,
                 stack_override: int = 0, tile_order: int = 0, sort_key: int = 0, refill: int = 0,
                 leaf_wait: int = 0, extra_lds: int = 0, packet_wg: int = 0, packet_cull: int = 0, tile_schedule: int = 0):
        self.L = load()
        opts = Options()
        opts.struct_size = C.sizeof(Options)
        opts.kernel = kernel
        opts.count_visits = int(count_visits)
        opts.sort_threshold = sort_threshold
        opts.grid_tile = grid_tile
        opts.tile_w_log2 = tile_w_log2
        opts.xcd_swizzle = xcd_swizzle
        opts.stack_override = stack_override
        opts.tile_order = tile_order
        opts.sort_key = sort_key
        opts.refill = refill
        opts.leaf_wait = leaf_wait
        opts.extra_lds = extra_lds
        opts.packet_wg = packet_wg
        opts.packet_cull = packet_cull
        opts.tile_schedule = tile_schedule
        self.h = C.c_void_p()
        rc = self.L.mrt_create(device, C.byref(opts), C.byref(self.h))
        if rc:
            raise MrtError(rc, self.L.mrt_status_string(rc).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.mrt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise MrtError(rc, self.L.mrt_last_error(self.h).decode() or self.L.mrt_status_string(rc).decode())

    def set_stream(self, stream_ptr: int):
        self._chk(self.L.mrt_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._chk(self.L.mrt_synchronize(self.h))

    def build_scene_device(self, tris, n_tris=None, on_device=False, safe_handoff=False, ploc=False, sah=False):
        """BVH built on the device from mrt_tri64 triangles (numpy array, or a device pointer with on_device): the radix tree,
        PLOC (ploc) or the host builder's binned-SAH tree (sah)."""
        if isinstance(tris, np.ndarray):
            tris = np.ascontiguousarray(tris)
            assert tris.dtype == T.TRI64
            n_tris = tris.shape[0]
        flags = (BUILD_TRIS_ON_DEVICE if on_device else 0) | (BUILD_SAFE_HANDOFF if safe_handoff else 0) | (BUILD_PLOC if ploc else 0) | (BUILD_SAH if sah else 0)
        self._chk(self.L.mrt_build_scene_device(self.h, _ptr(tris), n_tris, flags))

    def flatten_instances(self, verts9, instances, d_out, n_mesh_tris=None, on_device=False):
        """instances: numpy array of types.INSTANCE; verts9: (n,3,3) float32 mesh-space vertices (or a device pointer)."""
        if isinstance(verts9, np.ndarray):
            verts9 = np.ascontiguousarray(verts9, dtype=np.float32)
            n_mesh_tris = verts9.size // 9
        instances = np.ascontiguousarray(instances)
        assert instances.dtype == T.INSTANCE
        self._chk(self.L.mrt_flatten_instances(self.h, _ptr(verts9), n_mesh_tris, _np(instances), instances.shape[0],
                                               BUILD_TRIS_ON_DEVICE if on_device else 0, _ptr(d_out)))

    def build_instanced_scene_device(self, verts9, instances, n_mesh_tris=None, on_device=False):
        if isinstance(verts9, np.ndarray):
            verts9 = np.ascontiguousarray(verts9, dtype=np.float32)
            n_mesh_tris = verts9.size // 9
        instances = np.ascontiguousarray(instances)
        assert instances.dtype == T.INSTANCE
        self._chk(self.L.mrt_build_instanced_scene_device(self.h, _ptr(verts9), n_mesh_tris, _np(instances), instances.shape[0],
                                                          BUILD_TRIS_ON_DEVICE if on_device else 0))

    def upload_two_level_scene(self, verts9, instances, blas_on_device=False, sah=False):
        """SceneTLAS::build_tlas: one BLAS per distinct mesh, a TLAS over the instances (nothing is flattened)."""
        verts9 = np.ascontiguousarray(verts9, dtype=np.float32)
        instances = np.ascontiguousarray(instances)
        assert instances.dtype == T.INSTANCE
        self._chk(self.L.mrt_upload_two_level_scene(self.h, _np(verts9), verts9.size // 9, _np(instances), instances.shape[0],
                                                    (BUILD_BLAS_ON_DEVICE if blas_on_device else 0) | (BUILD_SAH if sah else 0)))

    def update_instances(self, instances):
        """SceneTLAS::refit_tlas: the same instances with new transforms."""
        instances = np.ascontiguousarray(instances)
        assert instances.dtype == T.INSTANCE
        self._chk(self.L.mrt_update_instances(self.h, _np(instances), instances.shape[0]))

    def upload_scene(self, tris, nodes, prim_idx):
        tris = np.ascontiguousarray(tris)
        nodes = np.ascontiguousarray(nodes)
        prim_idx = np.ascontiguousarray(prim_idx, dtype=np.uint32)
        assert tris.dtype == T.TRI64 and nodes.dtype == T.NODE32
        self._chk(self.L.mrt_upload_scene(self.h, _np(tris), tris.shape[0], _np(nodes), nodes.shape[0], _np(prim_idx)))

    def is_available(self) -> bool:
        return bool(self.L.mrt_is_available(self.h))

    def scene_info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.mrt_scene_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(n_tris=a.value, n_wide_nodes=b.value, stack_need=c.value)

    def cast(self, rays, hits=None, count=None, query_mask=0xFFFFFFFF, mode=MODE_NEAREST, flags=0):
        """rays / hits: numpy arrays (host) or ints / torch tensors (device, with the matching flag)."""
        if count is None:
            count = rays.shape[0]
        if hits is None:
            if flags & FLAG_BOOL_OUT:
                hits = np.zeros(count, dtype=np.uint8)
            elif flags & FLAG_TOKEN_OUT:
                hits = self._token_array(count)
            elif flags & FLAG_HOST_LAYOUT:
                hits = np.zeros(count, dtype=T.HOST_HIT44)
            else:
                hits = np.zeros(count, dtype=T.HIT32)
        if isinstance(rays, np.ndarray):
            rays = np.ascontiguousarray(rays)
        self._chk(self.L.mrt_cast(self.h, _ptr(rays), _ptr(hits), count, query_mask, mode, flags))
        return hits

    def submit(self, rays, count=None, query_mask=0xFFFFFFFF, mode=MODE_NEAREST, flags=0):
        if count is None:
            count = rays.shape[0]
        self._keep = rays  # caller must keep rays alive until collect (gpu_ray_caster.cpp:542)
        self._chk(self.L.mrt_submit(self.h, _ptr(rays), count, query_mask, mode, flags))
        self._pending = (count, mode, flags)

    def collect(self, hits=None, count=None):
        pc, mode, flags = getattr(self, "_pending", (0, 0, 0))
        if count is None:
            count = pc
        if hits is None:
            if flags & FLAG_BOOL_OUT:
                hits = np.zeros(count, dtype=np.uint8)
            elif flags & FLAG_TOKEN_OUT:
                hits = self._token_array(count)
            elif flags & FLAG_HOST_LAYOUT:
                hits = np.zeros(count, dtype=T.HOST_HIT44)
            else:
                hits = np.zeros(count, dtype=T.HIT32)
        self._chk(self.L.mrt_collect(self.h, _ptr(hits), count))
        return hits

    def has_pending(self) -> bool:
        return bool(self.L.mrt_has_pending(self.h))

    def generate_grid(self, cam, grid_w, grid_h, y0, y1, d_rays):
        self._chk(self.L.mrt_generate_grid(self.h, C.byref(cam), grid_w, grid_h, y0, y1, _ptr(d_rays)))

    def cast_grid(self, cam, grid_w, grid_h, y0=0, y1=None, hits=None, query_mask=0xFFFFFFFF, mode=MODE_NEAREST, flags=0):
        y1 = grid_h if y1 is None else y1
        n = grid_w * (y1 - y0)
        if hits is None:
            hits = np.zeros(n, dtype=np.uint8) if (flags & FLAG_BOOL_OUT) else (self._token_array(n) if (flags & FLAG_TOKEN_OUT) else np.zeros(n, dtype=T.HIT32))
        self._chk(self.L.mrt_cast_grid(self.h, C.byref(cam), grid_w, grid_h, y0, y1, _ptr(hits), query_mask, mode, flags))
        return hits

    def cast_tiled(self, d_rays, d_hits, grid_w, rows, query_mask=0xFFFFFFFF, mode=MODE_NEAREST):
        self._chk(self.L.mrt_cast_tiled(self.h, _ptr(d_rays), _ptr(d_hits), grid_w, rows, query_mask, mode))

    def expand_tokens(self, d_rays, d_tokens, d_hits, count, flags=0, stream=None):
        """Device pointers; enqueued on `stream` (raw hipStream_t) or the context's stream, not waited for."""
        self._chk(self.L.mrt_expand_tokens(self.h, _ptr(d_rays), _ptr(d_tokens), _ptr(d_hits), count, flags,
                                           C.c_void_p(stream) if stream else None))

    def expand_grid_tokens(self, cam, grid_w, grid_h, y0, y1, d_tokens, d_hits, stream=None):
        self._chk(self.L.mrt_expand_grid_tokens(self.h, C.byref(cam), grid_w, grid_h, y0, y1, _ptr(d_tokens), _ptr(d_hits),
                                                C.c_void_p(stream) if stream else None))

    def morton_keys(self, d_rays, count, d_keys):
        self._chk(self.L.mrt_morton_keys(self.h, _ptr(d_rays), count, _ptr(d_keys)))

    def stats(self) -> dict:
        s = Stats()
        self._chk(self.L.mrt_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def _token_array(self, count):
        """host array for `count` hit tokens: uint32[count] (flat scene) or uint32[count, 2] (two-level: {triangle, instance})"""
        words = self.token_bytes() // 4
        return np.zeros(count if words == 1 else (count, words), dtype=np.uint32)

    def token_bytes(self) -> int:
        """Bytes per hit token (MRT_FLAG_TOKEN_OUT): 4 for a flat scene, 8 ({triangle, instance}) for a two-level one."""
        return int(self.L.mrt_token_bytes(self.h))

    def last_kernel_variant(self) -> str:
        """The instantiation that ran the last blocking cast, as rocprofv3 spells kernel names."""
        return self.L.mrt_last_kernel_variant(self.h).decode()

    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._chk(self.L.mrt_device_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def device_free(self, ptr: int):
        self._chk(self.L.mrt_device_free(self.h, C.c_void_p(ptr)))

    def h2d(self, d_ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.mrt_memcpy_h2d(self.h, C.c_void_p(d_ptr), _np(arr), arr.nbytes))

    def d2h(self, arr: np.ndarray, d_ptr: int):
        self._chk(self.L.mrt_memcpy_d2h(self.h, _np(arr), C.c_void_p(d_ptr), arr.nbytes))


class Group:
    """mrt_group wrapper: several devices of one node driven from this process (rows of a grid sharded over them)."""

    def __init__(self, devices, **opts_kw):
        self.L = load()
        opts = Options()
        opts.struct_size = C.sizeof(Options)
        for k, v in opts_kw.items():
            setattr(opts, k, v)
        devs = (C.c_int * len(devices))(*devices)
        self.h = C.c_void_p()
        rc = self.L.mrt_group_create(len(devices), devs, C.byref(opts), C.byref(self.h))
        if rc:
            raise MrtError(rc, self.L.mrt_status_string(rc).decode())

    def _chk(self, rc):
        if rc:
            raise MrtError(rc, self.L.mrt_group_last_error(self.h).decode() or self.L.mrt_status_string(rc).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.mrt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self) -> int:
        return self.L.mrt_group_size(self.h)

    def upload(self, scene: "Scene"):
        tris, nodes, prim_idx = np.ascontiguousarray(scene.tris), np.ascontiguousarray(scene.nodes), np.ascontiguousarray(scene.prim_idx, dtype=np.uint32)
        self._chk(self.L.mrt_group_upload_scene(self.h, _np(tris), tris.shape[0], _np(nodes), nodes.shape[0], _np(prim_idx)))

    def upload_two_level_scene(self, verts9, instances, blas_on_device=False, sah=False):
        verts9 = np.ascontiguousarray(verts9, dtype=np.float32)
        instances = np.ascontiguousarray(instances)
        self._chk(self.L.mrt_group_upload_two_level_scene(self.h, _np(verts9), verts9.size // 9, _np(instances), instances.shape[0],
                                                          (BUILD_BLAS_ON_DEVICE if blas_on_device else 0) | (BUILD_SAH if sah else 0)))

    def cast_grid(self, cam, grid_w, grid_h, hits=None, query_mask=0xFFFFFFFF, mode=MODE_NEAREST, flags=0):
        if hits is None:
            hits = np.zeros(grid_w * grid_h, dtype=np.uint8 if (flags & FLAG_BOOL_OUT) else T.HIT32)
        self._chk(self.L.mrt_group_cast_grid(self.h, C.byref(cam), grid_w, grid_h, _ptr(hits), query_mask, mode, flags))
        return hits


def group_row_block(member: int, n_members: int, rows: int):
    y0, y1 = C.c_uint32(), C.c_uint32()
    load().mrt_group_row_block(member, n_members, rows, C.byref(y0), C.byref(y1))
    return y0.value, y1.value


class Scene:
    """Host-side scene: Triangle ctor -> 8-bin SAH BVH2, i.e. what RayScene::build
    hands to upload_scene (src/accel/ray_scene.h:62-86)."""

    def __init__(self, verts9, ids=None, layers=None, n_threads: int = 0):
        self.tris = make_triangles(verts9, ids, layers)
        self.verts4 = T.verts4_from_verts9(verts9)
        self.nodes, self.prim_idx, self.used_nodes = bvh2_build(self.verts4, n_threads)

    def upload(self, ctx: Context):
        ctx.upload_scene(self.tris, self.nodes, self.prim_idx)

    def save_bvh(self, path: str) -> None:
        bvh2_save(path, self.nodes, self.prim_idx)

    @classmethod
    def with_cached_bvh(cls, verts9, path: str, ids=None, layers=None, n_threads: int = 0):
        """The scene with its BVH taken from the cache file if it fits, else built (and saved)."""
        self = cls.__new__(cls)
        self.tris = make_triangles(verts9, ids, layers)
        self.verts4 = T.verts4_from_verts9(verts9)
        try:
            self.nodes, self.prim_idx, self.used_nodes = bvh2_load(path, self.tris.shape[0])
        except MrtError:
            self.nodes, self.prim_idx, self.used_nodes = bvh2_build(self.verts4, n_threads)
            bvh2_save(path, self.nodes, self.prim_idx)
        return self
