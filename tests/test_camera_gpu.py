"""RayCamera grids on the device (MRT_CAMERA_PERSPECTIVE / _ORTHOGRAPHIC): mrt_generate_grid is bit-identical to
the host loops of src/modules/graphics/ray_camera.h:234-273 (oracle restatement), and the fused
mrt_cast_grid / token paths return what tracing those rays returns."""
import numpy as np
import pytest

from messyerraytracer_amd import capi, synth, types as T
from oracle import pyoracle as po
import parity
from test_camera_cpu import _basis

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def soup(ctx):
    v = synth.soup(20000, 0.25, 11)
    return capi.Scene(v), po.OracleScene(v)


@pytest.mark.parametrize("ortho,param", [(False, 50.0), (True, 9.0)])
@pytest.mark.parametrize("wh", [(256, 192), (200, 77)])  # the second: tiles clipped on the right and at the bottom
def test_ray_camera_grid_on_the_device(ctx, soup, ortho, param, wh):
    scene, osc = soup
    scene.upload(ctx)
    w, h = wh
    basis = _basis(0.35, -0.2)
    origin = (4.0, 2.5, 11.0)
    cam = capi.ray_camera(origin, basis, w, h, param, ortho)
    rays = po.ray_camera_rays(origin, basis, w, h, param, ortho)
    d_rays = ctx.device_alloc(rays.nbytes)
    ctx.generate_grid(cam, w, h, 0, h, d_rays)
    got_rays = np.zeros(w * h, dtype=T.RAY32)
    ctx.d2h(got_rays, d_rays)
    assert got_rays.tobytes() == rays.tobytes(), "device RayCamera rays differ from the host formula"
    want = osc.trace(rays)
    assert (want["prim_id"] >= 0).any() and (want["prim_id"] < 0).any()
    parity.assert_exact(ctx.cast_grid(cam, w, h), want, "RayCamera cast_grid")
    # a band of rows (what a rank of the sharded grid casts)
    y0, y1 = h // 3, h // 3 + 40
    ctx.generate_grid(cam, w, h, y0, y1, d_rays)
    band = np.zeros(w * (y1 - y0), dtype=T.RAY32)
    ctx.d2h(band, d_rays)
    assert band.tobytes() == rays[y0 * w:y1 * w].tobytes()
    parity.assert_exact(ctx.cast_grid(cam, w, h, y0=y0, y1=y1), want[y0 * w:y1 * w], "RayCamera row band")
    # any-hit and hit tokens through the same camera
    any_got = ctx.cast_grid(cam, w, h, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT).astype(bool)
    assert np.array_equal(any_got, want["prim_id"] >= 0)
    d_tok, d_hits = ctx.device_alloc(w * h * 4), ctx.device_alloc(w * h * 32)
    ctx.cast_grid(cam, w, h, hits=d_tok, flags=capi.FLAG_HITS_ON_DEVICE | capi.FLAG_TOKEN_OUT)
    ctx.expand_grid_tokens(cam, w, h, 0, h, d_tok, d_hits)
    ctx.synchronize()
    rebuilt = np.zeros(w * h, dtype=T.HIT32)
    ctx.d2h(rebuilt, d_hits)
    parity.assert_exact(rebuilt, want, "RayCamera tokens")
    # sub-pixel jitter (generate_rays_tile_jittered)
    jcam = capi.ray_camera(origin, basis, w, h, param, ortho, jitter=(0.25, 0.75))
    jrays = po.ray_camera_rays(origin, basis, w, h, param, ortho, jitter=(0.25, 0.75))
    parity.assert_exact(ctx.cast_grid(jcam, w, h), osc.trace(jrays), "RayCamera jittered")
    for p in (d_rays, d_tok, d_hits):
        ctx.device_free(p)


def test_ray_camera_resolution_is_checked(ctx, soup):
    scene, _ = soup
    scene.upload(ctx)
    cam = capi.ray_camera((0, 0, 12), np.eye(3, dtype=np.float32), 64, 64, 50.0)
    with pytest.raises(capi.MrtError) as e:
        ctx.cast_grid(cam, 128, 64)  # RayCamera::generate_rays asserts the set-up resolution (ray_camera.h:150-152)
    assert e.value.status == capi.ERR_INVALID
    cam.kind = 7
    with pytest.raises(capi.MrtError):
        ctx.cast_grid(cam, 64, 64)


def test_ray_camera_on_a_two_level_scene(ctx):
    local, inst = synth.multi_mesh_instances(6, 800, 0.3, 21)
    ctx.upload_two_level_scene(local, inst)
    w, h = 160, 120
    basis = _basis(3.0, 0.1)
    origin = (0.5, 0.5, -12.0)
    cam = capi.ray_camera(origin, basis, w, h, 55.0)
    rays = po.ray_camera_rays(origin, basis, w, h, 55.0)
    want = po.OracleTwoLevelScene(local, inst).trace(rays)
    assert (want["prim_id"] >= 0).any()
    got = ctx.cast_grid(cam, w, h)
    assert np.array_equal(got["prim_id"], want["prim_id"]) and np.array_equal(got["t"], want["t"])
