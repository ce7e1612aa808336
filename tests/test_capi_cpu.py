"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header
declares, and its host-side functions (triangle packing, BVH builder, camera)
equal the oracle.  No compute entry point is called (there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from messyerraytracer_amd import capi, synth, types as T
from oracle import pyoracle as po


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "mrt_hip.h")).read()
    declared = sorted(set(re.findall(r"^(?:int|void|uint32_t|const char \*|mrt_ctx \*)\s*(mrt_[a-z0-9_]+)\(", hdr, re.M)))
    assert declared, "no declarations parsed"
    assert sorted(capi.SYMBOLS) == declared, "capi.SYMBOLS out of sync with include/mrt_hip.h"
    L = capi.load()
    for s in declared:
        assert hasattr(L, s), f"libmrt_hip.so does not export {s}"
    assert L.mrt_version() == (0 << 16) | 1
    assert L.mrt_status_string(0) == b"ok"
    assert L.mrt_status_string(capi.ERR_PENDING) != L.mrt_status_string(capi.ERR_NOT_PENDING)


def test_struct_sizes_match_reference_layouts(built):
    # the ctypes declarations against the library's own sizeof()
    L = capi.load()
    from messyerraytracer_amd import types as TT
    assert C.sizeof(capi.Options) == L.mrt_struct_size(0) == 64
    assert C.sizeof(capi.Camera) == L.mrt_struct_size(1) == 96
    assert C.sizeof(capi.Stats) == L.mrt_struct_size(2)
    assert TT.INSTANCE.itemsize == L.mrt_struct_size(3) == 64
    assert L.mrt_struct_size(99) == 0
    for dt, size in ((T.RAY32, 32), (T.HIT32, 32), (T.TRI64, 64), (T.NODE32, 32), (T.WIDE64, 64),
                     (T.HOST_RAY60, 60), (T.HOST_HIT44, 44), (T.HOST_TRI80, 80)):
        assert dt.itemsize == size


def test_null_arguments_are_rejected_not_crashed(built):
    L = capi.load()
    assert L.mrt_make_triangles(None, None, None, 1, None) == capi.ERR_INVALID
    assert L.mrt_bvh2_build(None, 0, None, None, None, 1) == capi.ERR_INVALID
    assert L.mrt_camera_look(None, None, None, 0, 0, 0.0) == capi.ERR_INVALID
    assert L.mrt_create(0, None, None) == capi.ERR_INVALID
    assert L.mrt_cast(None, None, None, 0, 0, 0, 0) == capi.ERR_INVALID
    assert L.mrt_has_pending(None) == 0 and L.mrt_is_available(None) == 0
    # the entry points added for the device-side build, instancing and the token exchange
    assert L.mrt_build_scene_device(None, None, 0, 0) == capi.ERR_INVALID
    assert L.mrt_flatten_instances(None, None, 0, None, 0, 0, None) == capi.ERR_INVALID
    assert L.mrt_build_instanced_scene_device(None, None, 0, None, 0, 0) == capi.ERR_INVALID
    assert L.mrt_expand_tokens(None, None, None, None, 0, 0, None) == capi.ERR_INVALID
    assert L.mrt_upload_two_level_scene(None, None, 0, None, 0, 0) == capi.ERR_INVALID
    assert L.mrt_update_instances(None, None, 0) == capi.ERR_INVALID
    assert L.mrt_expand_grid_tokens(None, None, 0, 0, 0, 0, None, None, None) == capi.ERR_INVALID
    assert L.mrt_camera_perspective(None, None, None, 0, 0, 0.0) == capi.ERR_INVALID
    assert L.mrt_kernel_name(capi.KERNEL_PACKET_DUAL) == b"trace_packet_rows_kernel<2>" and L.mrt_kernel_name(77) == b"?"
    L.mrt_destroy(None)  # no-op


def test_create_without_gpu_fails_loudly(built):
    """No silent CPU fallback: without a device mrt_create reports MRT_ERR_NO_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.MrtError) as e:
        capi.Context(0)
    assert e.value.status == capi.ERR_NO_DEVICE


def test_options_are_validated_before_a_device_is_asked_for(built):
    """Unknown kernel ids (the retired 3 and 4 included) and out-of-range tuning knobs are MRT_ERR_INVALID, with or
    without a GPU; a valid selection of a packet walk gets as far as the device check."""
    for kw in (dict(kernel=3), dict(kernel=4), dict(kernel=12), dict(packet_wg=100), dict(packet_cull=3)):
        with pytest.raises(capi.MrtError) as e:
            capi.Context(0, **kw)
        assert e.value.status == capi.ERR_INVALID, kw
    assert capi.kernel_name(capi.KERNEL_PACKET_QUAD) == "trace_packet_quad_kernel"
    import torch
    if not torch.cuda.is_available():
        for kw in (dict(kernel=capi.KERNEL_PACKET_DUAL), dict(packet_wg=64, packet_cull=2, tile_schedule=1)):
            with pytest.raises(capi.MrtError) as e:
                capi.Context(0, **kw)
            assert e.value.status == capi.ERR_NO_DEVICE, kw
    # the four-wide packet walk exists only in builds made with MRT_WITH_QUAD=1: otherwise the selection is refused as such
    assert capi.kernel_available(capi.KERNEL_PACKET_DUAL) and capi.kernel_available(capi.KERNEL_AUTO) and not capi.kernel_available(3)
    if not capi.kernel_available(capi.KERNEL_PACKET_QUAD):
        with pytest.raises(capi.MrtError) as e:
            capi.Context(0, kernel=capi.KERNEL_PACKET_QUAD)
        assert e.value.status == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.MrtError) as e:
        capi.Context(0, tile_schedule=3)     # 0 = longest first with pieces, 1 = plain order, 2 = longest first without pieces
    assert e.value.status == capi.ERR_INVALID


@pytest.mark.parametrize("n,s,seed", [(1, 0.5, 1), (2, 0.5, 2), (3, 0.5, 3), (7, 0.5, 4), (1000, 0.5, 1), (60000, 0.15, 7)])
def test_builder_equals_oracle_builder(built, n, s, seed):
    """mrt_bvh2_build == the restatement of tinybvh::BVH::Build, for any thread count."""
    v = synth.soup(n, s, seed)
    v4 = T.verts4_from_verts9(v)
    on, op, ou = po.bvh2_build(v4)
    for threads in (1, 3, 8):
        pn, pp, pu = capi.bvh2_build(v4, threads)
        assert pu == ou
        assert pn.tobytes() == on.tobytes()
        assert np.array_equal(pp, op)


def test_builder_degenerate_inputs(built):
    # all triangles identical: SAH cannot split; one leaf holding everything
    one = synth.soup(1, 0.5, 3)
    v = np.repeat(one, 40, axis=0)
    v4 = T.verts4_from_verts9(v)
    pn, pp, pu = capi.bvh2_build(v4, 2)
    on, op, ou = po.bvh2_build(v4)
    assert pu == ou == 2 and pn[0]["tri_count"] == 40 and pn.tobytes() == on.tobytes()
    # flat scene (zero extent on z)
    v = synth.soup(500, 0.5, 8)
    v[:, :, 2] = 1.0
    v4 = T.verts4_from_verts9(v)
    pn, pp, pu = capi.bvh2_build(v4, 4)
    on, op, ou = po.bvh2_build(v4)
    assert pn.tobytes() == on.tobytes() and np.array_equal(pp, op)


def test_triangles_and_camera_equal_oracle(built):
    v = synth.soup(5000, 0.3, 11)
    ids = np.arange(5000, dtype=np.uint32)[::-1].copy()
    layers = (np.arange(5000, dtype=np.uint32) % 7) + 1
    assert capi.make_triangles(v, ids, layers).tobytes() == po.make_triangles(v, ids, layers).tobytes()
    for fwd in ((0, 0, 1), (0, 0, -1), (0, 1, 0), (0.3, -0.2, 0.9), (0, -1, 1e-3)):
        for (w, h, fov) in ((16, 12, 60.0), (4096, 4096, 50.0), (1920, 1080, 75.0)):
            cam = capi.camera_look((1, 2, 3), fwd, w, h, fov)
            f, r, u, hw, hh = po.camera_basis(fwd, w, h, fov)
            assert np.array_equal(np.array(cam.fwd[:], np.float32), f)
            assert np.array_equal(np.array(cam.right[:], np.float32), r)
            assert np.array_equal(np.array(cam.up[:], np.float32), u)
            assert np.float32(cam.half_w) == hw and np.float32(cam.half_h) == hh
            assert cam.t_min == np.float32(0.001) and cam.t_max == T.FLT_MAX


def test_product_does_not_reference_the_oracle():
    """The product path may not import, link or call anything under oracle/."""
    for top in ("messyerraytracer_amd", "tools", "include"):  # the product, its headers and the measuring tools
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".sh")):
                    text = open(os.path.join(dirpath, fn), errors="ignore").read()
                    assert "pyoracle" not in text and "liboracle" not in text and "mrt_oracle.h" not in text \
                        and "libmrt_ref" not in text, f"{fn} references the oracle"
    import subprocess
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_instances_flatten_to_the_multi_mesh_scene():
    """synth.multi_mesh_instances + synth.flatten_instances (the host statement of
    raytracer_server.cpp:700-711) reproduce the C5 generator bit for bit."""
    from messyerraytracer_amd import synth
    local, inst = synth.multi_mesh_instances(5, 300, 0.05, 11)
    world = synth.flatten_instances(local, inst)
    assert world.tobytes() == synth.multi_mesh(5, 300, 0.05, 11)[0].tobytes()
    assert inst["n_tris"].sum() == 1500 and inst.dtype.itemsize == 64
    # the oracle's restatement of the flatten + Triangle ctor gives the triangles of the flattened vertices
    from oracle import pyoracle as po
    inst["layers"] = [1, 2, 4, 8, 16]
    flat = po.flatten_instances(local, inst)
    want = po.make_triangles(world, np.arange(1500, dtype=np.uint32), np.repeat(inst["layers"], inst["n_tris"]).astype(np.uint32))
    assert flat.tobytes() == want.tobytes()
    assert capi.make_triangles(world, np.arange(1500, dtype=np.uint32),
                               np.repeat(inst["layers"], inst["n_tris"]).astype(np.uint32)).tobytes() == want.tobytes()


def test_bvh_cache_file_round_trip(built, tmp_path):
    """mrt_bvh2_save / mrt_bvh2_load (BVH::Save / Load, tiny_bvh.h:1747-1799): the loaded tree is the saved one;
    a file for another triangle count, a truncated or a damaged file is refused."""
    v = synth.soup(3000, 0.3, 5)
    v4 = T.verts4_from_verts9(v)
    nodes, prim, used = capi.bvh2_build(v4, 2)
    path = str(tmp_path / "scene.bvh")
    capi.bvh2_save(path, nodes, prim)
    n2, p2, u2 = capi.bvh2_load(path, 3000)
    assert u2 == used and n2.tobytes() == nodes.tobytes() and np.array_equal(p2, prim)
    with pytest.raises(capi.MrtError) as e:
        capi.bvh2_load(path, 2999)                      # saved for another scene
    assert e.value.status == capi.ERR_BAD_BVH
    raw = bytearray(open(path, "rb").read())
    assert len(raw) == 32 + used * 32 + 3000 * 4
    open(path, "wb").write(raw[:-4])                    # truncated
    with pytest.raises(capi.MrtError):
        capi.bvh2_load(path, 3000)
    raw[100] ^= 0x40                                    # one flipped bit in a node
    open(path, "wb").write(raw)
    with pytest.raises(capi.MrtError):
        capi.bvh2_load(path, 3000)
    with pytest.raises(capi.MrtError):
        capi.bvh2_load(str(tmp_path / "missing.bvh"), 3000)
    # Scene.with_cached_bvh builds and saves on a miss, loads on a hit
    a = capi.Scene.with_cached_bvh(v, path)
    b = capi.Scene.with_cached_bvh(v, path)
    assert a.nodes.tobytes() == b.nodes.tobytes() == nodes.tobytes() and np.array_equal(b.prim_idx, prim)
    assert capi.load().mrt_bvh2_save(None, None, 0, None, 0) == capi.ERR_INVALID
    assert capi.load().mrt_bvh2_load(None, 0, None, None, None) == capi.ERR_INVALID
