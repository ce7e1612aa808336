"""mrt_build_scene_device: the acceleration structure built on the device (LBVH) must give the
hits the oracle gives on its host-built SAH tree — results do not depend on which valid BVH is
walked (exact ties go to the lower triangle id), so the comparison is bit-exact."""
import numpy as np
import pytest

from messyerraytracer_amd import capi, synth, types as T
from oracle import pyoracle as po
import parity

pytestmark = pytest.mark.gpu
# the four-wide packet walk is in builds made with MRT_WITH_QUAD=1 only (include/mrt_hip.h, mrt_kernel_available)
QUAD = pytest.param(capi.KERNEL_PACKET_QUAD, marks=pytest.mark.skipif(not capi.kernel_available(capi.KERNEL_PACKET_QUAD), reason="built without MRT_WITH_QUAD"))


def _rays(seed):
    grid = po.grid_rays((0, 0, -12), (0, 0, 1), 100, 70, 50.0)
    inc = synth.incoherent_rays(5000, seed)
    inc["t_min"][:200] = 3.0
    inc["t_max"][:200] = 3.0          # degenerate
    inc["direction"][200:400] = [0, 1, 0]
    inc["t_max"][400:700] = 1.5
    return grid, inc


def _check(c, osc, name, masks=(0xFFFFFFFF,)):
    grid, inc = _rays(11)
    for rays, kind in ((grid, "grid"), (inc, "incoherent")):
        for mask in masks:
            want = osc.trace(rays, query_mask=mask)
            for flags in (capi.FLAG_COHERENT, 0):
                parity.assert_exact(c.cast(rays, query_mask=mask, flags=flags), want, f"{name} {kind} mask={mask:#x} flags={flags}")
            b = c.cast(rays, query_mask=mask, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT)
            assert np.array_equal(b.astype(bool), want["prim_id"] >= 0)
    cam = capi.camera_look((0, 0, -12), (0, 0, 1), 100, 70, 50.0)
    parity.assert_exact(c.cast_grid(cam, 100, 70), osc.trace(grid), f"{name} cast_grid")


FORMS = {"radix_tree": {}, "ploc": {"ploc": True}, "sah": {"sah": True}}


@pytest.mark.parametrize("form", list(FORMS))
@pytest.mark.parametrize("n_tris,scale,seed", [(1, 2.0, 5), (2, 2.0, 6), (3, 1.5, 7), (17, 1.0, 8), (33, 1.0, 18), (1000, 0.5, 1), (20000, 0.25, 33)])
def test_device_built_tree_gives_the_oracles_hits(built, n_tris, scale, seed, form):
    v = synth.soup(n_tris, scale, seed)
    layers = (1 << (np.arange(n_tris) % 3)).astype(np.uint32)
    tris = capi.make_triangles(v, None, layers)
    c = capi.Context(0)
    c.build_scene_device(tris, **FORMS[form])
    assert c.is_available() and c.scene_info()["n_tris"] == n_tris
    _check(c, po.OracleScene(v, None, layers), f"n={n_tris}", masks=(0xFFFFFFFF, 0x5))
    c.close()


@pytest.mark.parametrize("kernel", [capi.KERNEL_LANE, capi.KERNEL_PACKET,
                                    capi.KERNEL_PACKET_ASM, capi.KERNEL_PACKET_DUAL, capi.KERNEL_PACKET_ROWS, QUAD, capi.KERNEL_LANE_PERSISTENT, capi.KERNEL_LANE4_PERSISTENT,
                                    capi.KERNEL_LANE8_PERSISTENT])
def test_every_kernel_walks_a_device_built_tree(built, kernel):
    """Kernels that want the 4-wide layout (absent for device-built trees) must fall back, not fault."""
    v = synth.soup(3000, 0.35, 17)
    c = capi.Context(0, kernel=kernel)
    c.build_scene_device(capi.make_triangles(v))
    _check(c, po.OracleScene(v), f"kernel={kernel}")
    c.close()


@pytest.mark.parametrize("form", list(FORMS))
def test_degenerate_inputs_for_the_device_builders(built, form):
    """Equal Morton keys (coincident triangles, a flat cluster with one outlier): the radix tree splits them by
    index, the clustering merges boxes of equal area by position, the SAH form keeps what no plane separates in one
    leaf; the tree stays a valid BVH of bounded depth."""
    base = synth.soup(1, 0.5, 3)
    same = np.repeat(base, 300, axis=0)                       # 300 copies of one triangle: all keys equal
    v = np.concatenate([same, synth.soup(50, 0.2, 4) * 0.001, synth.soup(1, 0.5, 9) + 4.0]).astype(np.float32)
    ids = np.arange(v.shape[0], dtype=np.uint32)[::-1].copy()  # the winner among exact ties is the LOWEST id
    c = capi.Context(0)
    c.build_scene_device(capi.make_triangles(v, ids), **FORMS[form])
    _check(c, po.OracleScene(v, ids), "degenerate")
    assert c.scene_info()["stack_need"] <= 64
    c.close()


def test_device_resident_triangles_and_rebuild(built):
    """Triangles already in HBM (MRT_BUILD_TRIS_ON_DEVICE); a context is rebuilt in place, and a
    host-built scene can replace a device-built one and vice versa."""
    c = capi.Context(0)
    for seed in (1, 2):
        v = synth.soup(5000, 0.3, seed)
        tris = capi.make_triangles(v)
        d = c.device_alloc(tris.nbytes)
        c.h2d(d, tris)
        c.build_scene_device(d, n_tris=tris.shape[0], on_device=True)
        c.device_free(d)
        _check(c, po.OracleScene(v), f"device tris seed={seed}")
        assert c.stats()["last_build_ms"] > 0.0
    v = synth.soup(2000, 0.4, 9)
    capi.Scene(v).upload(c)                                    # host-built SAH tree over the same context
    _check(c, po.OracleScene(v), "host after device")
    c.build_scene_device(capi.make_triangles(v))
    _check(c, po.OracleScene(v), "device after host")
    # the acquire-release hand-off (what a build falls back to if its verification pass fails)
    c.build_scene_device(capi.make_triangles(v), safe_handoff=True)
    _check(c, po.OracleScene(v), "safe hand-off")
    c.build_scene_device(capi.make_triangles(v), ploc=True)      # clustering after the radix tree and back: one arena serves both
    _check(c, po.OracleScene(v), "clustering after the radix tree")
    c.build_scene_device(capi.make_triangles(v))
    _check(c, po.OracleScene(v), "radix tree after clustering")
    c.build_scene_device(capi.make_triangles(v), sah=True)       # ... and the binned-SAH form (its own, larger share of the arena)
    _check(c, po.OracleScene(v), "SAH after the radix tree")
    c.build_scene_device(capi.make_triangles(v), ploc=True)
    _check(c, po.OracleScene(v), "clustering after SAH")
    # the builder's arena grows when a larger scene arrives (and is kept for the smaller ones after it)
    big = synth.soup(30000, 0.25, 12)
    c.build_scene_device(capi.make_triangles(big), ploc=True)
    _check(c, po.OracleScene(big), "larger scene after smaller ones (clustering)")
    c.build_scene_device(capi.make_triangles(v))
    _check(c, po.OracleScene(v), "smaller scene in the grown arena")
    with pytest.raises(capi.MrtError):
        c.build_scene_device(np.zeros(0, dtype=T.TRI64))
    c.close()


def test_c2_device_build_matches_host_build_on_the_full_grid(built):
    """Config C2 (100 k triangles, 1024^2 rays): the device-built and the host-built tree give the
    same 2^20 hit records, byte for byte."""
    cfg = synth.CONFIGS["C2"]
    w, h = cfg["grid"]
    verts = synth.scene_vertices(cfg)
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    c = capi.Context(0)
    scene = capi.Scene(verts)
    scene.upload(c)
    host = c.cast_grid(cam, w, h)
    c.build_scene_device(scene.tris)
    dev = c.cast_grid(cam, w, h)
    assert dev.tobytes() == host.tobytes()
    inc = synth.incoherent_rays(1 << 18, 77)
    got = c.cast(inc)
    scene.upload(c)
    assert got.tobytes() == c.cast(inc).tobytes()
    c.close()


def test_the_sah_form_builds_the_host_builders_tree(built):
    """MRT_BUILD_SAH makes the host builder's decisions (tinybvh::BVH::Build, restated in host/bvh_builder.cpp) on the
    device's triangle boxes -- the host's widened by one ulp (v0 + e1 is a rounded sum) --, so the tree has the host
    tree's shape: the same number of rows and the same stack need to within the handful of decisions an ulp can flip,
    and the records of a cast are the same bytes."""
    cfg = synth.CONFIGS["C2"]
    w, h = cfg["grid"]
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    c = capi.Context(0)
    scene = capi.Scene(synth.scene_vertices(cfg))
    scene.upload(c)
    host_info, host = c.scene_info(), c.cast_grid(cam, w, h)
    c.build_scene_device(scene.tris, sah=True)
    info = c.scene_info()
    assert abs(info["n_wide_nodes"] - host_info["n_wide_nodes"]) <= 0.002 * host_info["n_wide_nodes"], (info, host_info)
    assert abs(info["stack_need"] - host_info["stack_need"]) <= 2
    assert c.cast_grid(cam, w, h).tobytes() == host.tobytes()
    inc = synth.incoherent_rays(1 << 18, 78)
    got = c.cast(inc)
    scene.upload(c)
    assert got.tobytes() == c.cast(inc).tobytes()
    c.close()


def test_c3_c4_device_build_matches_host_build(built):
    """The headline scene (1 M triangles): 4096^2 primary rays (packet kernel) and 2^22 of C4's
    incoherent rays (persistent lane kernel; 2-wide on the device-built tree, 4-wide on the
    host-built one) give the same records against both trees."""
    cfg = synth.CONFIGS["C3"]
    w, h = cfg["grid"]
    verts = synth.scene_vertices(cfg)
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    c4 = synth.CONFIGS["C4"]
    inc = synth.incoherent_rays(1 << 22, c4["ray_seed"])
    c = capi.Context(0)
    scene = capi.Scene(verts)
    scene.upload(c)
    host_grid = c.cast_grid(cam, w, h)
    host_inc = c.cast(inc, flags=capi.FLAG_COHERENT)
    c.build_scene_device(scene.tris)
    assert c.cast_grid(cam, w, h).tobytes() == host_grid.tobytes()
    assert c.cast(inc, flags=capi.FLAG_COHERENT).tobytes() == host_inc.tobytes()
    assert c.cast(inc).tobytes() == host_inc.tobytes()      # Morton-sorted
    c.build_scene_device(scene.tris, sah=True)               # the binned-SAH form: leaves of several triangles
    assert c.cast_grid(cam, w, h).tobytes() == host_grid.tobytes()
    assert c.cast(inc).tobytes() == host_inc.tobytes()
    c.close()


def test_instances_flattened_and_built_on_the_device(built):
    """mrt_flatten_instances == the host flatten of RayTracerServer::_rebuild_scene + the Triangle
    ctor, bit for bit (meshes shared by several instances, per-mesh layer masks, running ids), and
    mrt_build_instanced_scene_device gives the oracle's hits on the flattened scene."""
    local, inst = synth.multi_mesh_instances(8, 2000, 0.1, 7)
    extra = inst[[0, 3]].copy()                       # true instancing: meshes 0 and 3 placed a second time
    extra["origin"] += np.array([0.5, -0.25, 1.0], dtype=np.float32)
    extra["layers"] = [0x2, 0x4]
    inst = np.concatenate([inst, extra])
    world = synth.flatten_instances(local, inst)
    n = world.shape[0]
    assert n == 10 * 2000
    ids = np.arange(n, dtype=np.uint32)
    layers = np.repeat(inst["layers"], inst["n_tris"]).astype(np.uint32)
    want = po.flatten_instances(local, inst)            # oracle: raytracer_server.cpp:700-711 + Triangle ctor
    assert want.tobytes() == capi.make_triangles(world, ids, layers).tobytes()
    c = capi.Context(0)
    d_out = c.device_alloc(n * 64)
    c.flatten_instances(local, inst, d_out)
    got = np.zeros(n, dtype=T.TRI64)
    c.d2h(got, d_out)
    assert got.tobytes() == want.tobytes()
    # mesh vertices already on the device
    d_local = c.device_alloc(local.nbytes)
    c.h2d(d_local, local)
    c.flatten_instances(d_local, inst, d_out, n_mesh_tris=local.shape[0], on_device=True)
    c.d2h(got, d_out)
    assert got.tobytes() == want.tobytes()
    osc = po.OracleScene(world, ids, layers)
    c.build_instanced_scene_device(local, inst)
    _check(c, osc, "instanced", masks=(0xFFFFFFFF, 0x2))
    c.build_instanced_scene_device(d_local, inst, n_mesh_tris=local.shape[0], on_device=True)
    _check(c, osc, "instanced, resident meshes")
    bad = inst.copy()
    bad["n_tris"][1] = local.shape[0]                 # runs past the mesh array
    with pytest.raises(capi.MrtError):
        c.build_instanced_scene_device(local, bad)
    c.device_free(d_out); c.device_free(d_local)
    c.close()


def test_a_tree_deeper_than_the_stack_is_refused_not_walked(built):
    """An adversarial scene -- one triangle per Morton-key bit (a 63-deep chain in the radix tree) plus 64
    coincident ones at its end -- gives an LBVH deeper than the 64-entry packet stack: the device build
    must say so (MRT_ERR_UNSUPPORTED) and leave the context usable for the host-built tree of the scene."""
    pts = []
    for k in range(63):
        bit = 62 - k                      # key bit 3t + axis holds bit t of that axis' grid coordinate
        p = [0.0, 0.0, 0.0]
        p[bit % 3] = float(1 << (bit // 3))
        pts.append(p)
    pts.append([float((1 << 21) - 1)] * 3)   # fixes the scene extent at 2^21 - 1 grid cells of size 1
    pts += [[0.0, 0.0, 0.0]] * 64
    c0 = np.array(pts, dtype=np.float32)
    v = np.zeros((c0.shape[0], 3, 3), dtype=np.float32)
    v[:, 0] = c0 + np.float32([0.1, 0.1, 0.1]); v[:, 1] = c0 + np.float32([0.4, 0.1, 0.2]); v[:, 2] = c0 + np.float32([0.1, 0.4, 0.3])
    c = capi.Context(0)
    with pytest.raises(capi.MrtError) as e:
        c.build_scene_device(capi.make_triangles(v))
    assert e.value.status == capi.ERR_UNSUPPORTED
    assert not c.is_available()           # no half-installed scene
    scene, osc = capi.Scene(v), po.OracleScene(v)
    scene.upload(c)
    rays = np.zeros(256, dtype=T.RAY32)
    rays["origin"] = np.float32([0.2, 0.2, -5.0]) + np.float32([1.0, 0.0, 0.0]) * (np.arange(256, dtype=np.float32)[:, None] * 0.01)
    rays["direction"] = [0.0, 0.0, 1.0]
    rays["t_min"], rays["t_max"] = 0.001, T.FLT_MAX
    parity.assert_exact(c.cast(rays, flags=capi.FLAG_COHERENT), osc.trace(rays), "host tree after a refused device build")
    c.close()


def test_more_instances_than_one_launch_takes(built):
    """70 000 placements of a two-triangle mesh (a launch's grid.y stops at 65 535): the device flatten still
    equals the oracle's, and the scene built from it gives the oracle's hits."""
    local = synth.soup(2, 0.3, 5)
    n = 70000
    rng = np.random.default_rng(9)
    inst = np.zeros(n, dtype=T.INSTANCE)
    inst["first_tri"], inst["n_tris"], inst["layers"] = 0, 2, 0xFFFFFFFF
    inst["basis"] = np.eye(3, dtype=np.float32).ravel() * np.float32(0.2)
    inst["origin"] = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    want = po.flatten_instances(local, inst)
    c = capi.Context(0)
    d_out = c.device_alloc(2 * n * 64)
    c.flatten_instances(local, inst, d_out)
    got = np.zeros(2 * n, dtype=T.TRI64)
    c.d2h(got, d_out)
    assert got.tobytes() == want.tobytes()
    c.device_free(d_out)
    c.build_instanced_scene_device(local, inst)
    world = synth.flatten_instances(local, inst)
    rays = synth.incoherent_rays(20000, 2)
    parity.assert_exact(c.cast(rays), po.OracleScene(world).trace(rays), "70 000 instances")
    c.close()
