"""RayCamera grids (src/modules/graphics/ray_camera.h:50-76,208-273): the oracle's restatement against an
independent numpy float32 restatement of the same formulas, and the host half of the C-ABI
(mrt_camera_perspective / mrt_camera_orthographic need no device)."""
import ctypes as C
import math

import numpy as np
import pytest

from messyerraytracer_amd import capi
from oracle import pyoracle as po

F = np.float32


def _basis(yaw, pitch):
    """A rotation as Godot's Basis rows (float32)."""
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return (ry @ rx).astype(np.float32)


def _numpy_camera_rays(origin, basis, w, h, param, ortho, jitter=(0.5, 0.5)):
    """ray_camera.h:234-273 in numpy float32, one operation per numpy call so nothing is fused."""
    origin = np.asarray(origin, dtype=F)
    inv_w, inv_h = F(1.0) / F(w), F(1.0) / F(h)
    aspect = F(w) / F(h)
    if ortho:
        half_h = F(param) * F(0.5)
        half_w = half_h * aspect
    else:
        tan_half = F(math.tan(float(F(param) * F(0.5)) * (3.1415926535897932384626433833 / 180.0)))
        half_w, half_h = tan_half * aspect, tan_half
    x = np.arange(w, dtype=F)
    y = np.arange(h, dtype=F)
    u = (F(2.0) * (x + F(jitter[0])) * inv_w) - F(1.0)
    v = F(1.0) - (F(2.0) * (y + F(jitter[1])) * inv_h)
    uu, vv = np.meshgrid(u, v)  # [h, w]
    rays = np.zeros(h * w, dtype=po.RAY32)
    rays["t_min"], rays["t_max"] = F(0.001), np.finfo(F).max
    if not ortho:
        vx, vy, vz = uu * half_w, vv * half_h, F(-1.0)
        d = np.stack([(basis[k, 0] * vx + basis[k, 1] * vy) + basis[k, 2] * vz for k in range(3)], axis=-1).astype(F)
        l2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        ln = np.sqrt(l2)
        d = d / ln[..., None]
        rays["origin"] = origin
        rays["direction"] = d.reshape(-1, 3)
    else:
        sv, su = vv * half_h, uu * half_w
        o = np.stack([(origin[k] + basis[k, 1] * sv) + basis[k, 0] * su for k in range(3)], axis=-1).astype(F)
        rays["origin"] = o.reshape(-1, 3)
        rays["direction"] = -basis[:, 2]
    return rays


@pytest.mark.parametrize("ortho,param", [(False, 60.0), (False, 33.5), (True, 7.5)])
@pytest.mark.parametrize("wh", [(16, 12), (97, 41)])
def test_oracle_ray_camera_matches_numpy(ortho, param, wh):
    w, h = wh
    basis = _basis(0.7, -0.3)
    origin = (1.5, -2.25, 9.0)
    want = _numpy_camera_rays(origin, basis, w, h, param, ortho)
    got = po.ray_camera_rays(origin, basis, w, h, param, ortho)
    assert got.tobytes() == want.tobytes()
    # row bands are slices of the same grid
    band = po.ray_camera_rays(origin, basis, w, h, param, ortho, y0=3, y1=9)
    assert band.tobytes() == want[3 * w:9 * w].tobytes()
    # a jittered tile is the same formula with another sub-pixel offset (generate_ray_jittered, :106-122)
    jit = po.ray_camera_rays(origin, basis, w, h, param, ortho, jitter=(0.125, 0.875))
    assert jit.tobytes() == _numpy_camera_rays(origin, basis, w, h, param, ortho, (0.125, 0.875)).tobytes()


def test_perspective_rows_run_top_to_bottom_and_look_down_minus_z():
    """v is flipped (row 0 is the top of the image) and the camera looks along -Z of its basis."""
    rays = po.ray_camera_rays((0, 0, 0), np.eye(3, dtype=F), 8, 8, 90.0)
    d = rays["direction"].reshape(8, 8, 3)
    assert (d[..., 2] < 0).all()
    assert (d[0, :, 1] > 0).all() and (d[7, :, 1] < 0).all()
    assert (d[:, 0, 0] < 0).all() and (d[:, 7, 0] > 0).all()
    assert np.allclose(np.linalg.norm(rays["direction"], axis=1), 1.0, atol=1e-6)


def test_camera_setup_through_the_c_abi():
    basis = _basis(-1.1, 0.4)
    cam = capi.ray_camera((1, 2, 3), basis, 640, 360, 75.0)
    assert cam.kind == 1
    assert cam.inv_w == F(1.0) / F(640) and cam.inv_h == F(1.0) / F(360)
    tan_half = F(math.tan(float(F(75.0) * F(0.5)) * (math.pi / 180.0)))
    assert cam.half_h == tan_half and cam.half_w == tan_half * (F(640) / F(360))
    assert list(cam.right) == list(basis[:, 0]) and list(cam.up) == list(basis[:, 1]) and list(cam.fwd) == list(basis[:, 2])
    assert cam.t_min == F(0.001) and cam.t_max == np.finfo(F).max and (cam.jitter_x, cam.jitter_y) == (0.5, 0.5)
    cam = capi.ray_camera((1, 2, 3), basis, 640, 360, 10.0, ortho=True)
    assert cam.kind == 2 and cam.half_h == F(5.0) and cam.half_w == F(5.0) * (F(640) / F(360))
    # the debug-grid camera keeps its kind
    assert capi.camera_look((0, 0, 3), (0, 0, -1), 16, 12, 60.0).kind == 0
    L = capi.load()
    o = (C.c_float * 3)(0, 0, 0)
    b = (C.c_float * 9)(*np.eye(3, dtype=F).reshape(9))
    c = capi.Camera()
    assert L.mrt_camera_perspective(C.byref(c), o, b, 0, 4, 60.0) == capi.ERR_INVALID
    assert L.mrt_camera_perspective(C.byref(c), o, b, 4, 4, 0.0) == capi.ERR_INVALID   # RT_ASSERT(fov > 0), :210
    assert L.mrt_camera_orthographic(C.byref(c), o, b, 4, 4, -1.0) == capi.ERR_INVALID  # RT_ASSERT(size > 0), :222
    assert L.mrt_camera_perspective(None, o, b, 4, 4, 60.0) == capi.ERR_INVALID
