"""The host-side mirror without a device (CPU test tier): RayTracerServer semantics over the
RayDispatcher mirror's CPU backend (a8) and its ThreadPool (a12), held to the oracle bit for bit.  BASELINE config 1
("cube, 16x12 cast_debug_rays on the CPU ThreadPool backend") is the first case."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from messyerraytracer_amd import build as mbuild, synth, types as T
from oracle import pyoracle as po

import server_driver as sd

def test_config_c1_cube_on_the_cpu_backend(built):
    """BASELINE.json config 1: single cube (12 tris), 16x12 debug-ray grid, CPU ThreadPool backend, through the router."""
    c1 = synth.CONFIGS["C1"]
    rays = po.grid_rays(c1["origin"], c1["forward"], *c1["grid"], c1["fov"])
    out = sd.check([(synth.cube(), np.eye(3).reshape(9), (0, 0, 0), 0xFFFFFFFF)], rays, 0xFFFFFFFF)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g1_cube.npz"))
    assert np.array_equal(out["nearest"]["prim_id"] != 0xFFFFFFFF, g["hits_ref"]["prim_id"] >= 0)   # the reference's own 36 hits


def test_placed_meshes_layer_masks_and_the_thread_pool(built):
    """Three placed meshes (rotation + translation, different layer masks), 6 000 rays: above MIN_BATCH_FOR_THREADING, so the
    batch is range-split over the pool and the per-chunk RayStats are merged."""
    local, inst = synth.multi_mesh_instances(3, 1500, 0.3, 77)
    meshes = []
    for i in range(3):
        a, b = int(inst[i]["first_tri"]), int(inst[i]["first_tri"] + inst[i]["n_tris"])
        meshes.append((local[a:b], inst[i]["basis"], inst[i]["origin"], [0x1, 0x6, 0x80000001][i]))
    rays = np.concatenate([po.grid_rays((0, 0, -12), (0, 0, 1), 64, 48, 50.0), synth.incoherent_rays(3000, 5)])
    sd.check(meshes, rays, 0xFFFFFFFF)
    sd.check(meshes, rays, 0x4)           # only the second mesh is visible
    sd.check(meshes[:1], rays[:100], 0x1)  # below the threading threshold: the calling thread alone


def test_root_leaf_scene_on_the_cpu_backend(built):
    v = synth.soup(2, 0.8, 3)
    sd.check([(v, np.eye(3).reshape(9), (0, 0, 0), 0xFFFFFFFF)], po.grid_rays((0, 0, -12), (0, 0, 1), 32, 32, 50.0), 0xFFFFFFFF)


def test_opt_in_fallback_to_the_cpu_pool(built):
    """set_cpu_fallback(true) restores the reference's routing (ray_dispatcher.h:130,152-180; raytracer_server.cpp:346-355):
    BACKEND_AUTO without a usable device casts on the CPU pool, set_backend(BACKEND_GPU) with a failed initialisation
    switches to BACKEND_CPU.  Off by default (the other tests see MRT_ERR_NO_DEVICE).  Without a GPU the records come from
    the pool; on a GPU box the same modes run on the device -- either way they are the oracle's."""
    v = synth.soup(800, 0.4, 31)
    rays = np.concatenate([po.grid_rays((0, 0, -12), (0, 0, 1), 24, 16, 50.0), synth.incoherent_rays(300, 3)])
    mesh = [(v, np.eye(3).reshape(9), (0, 0, 0), 0xFFFFFFFF)]
    sd.check(mesh, rays, 0xFFFFFFFF, mode="auto-fallback")
    sd.check(mesh, rays, 0xFFFFFFFF, mode="gpu-fallback")


def test_cpu_backend_routes_to_the_tlas(built):
    """ray_dispatcher.h:443-452: with a TLAS set the CPU path walks the two-level scene.  Meshes placed twice with different
    masks, 5 000 rays (range-split over the pool), masks that hide whole instances: the records of the oracle's SceneTLAS
    restatement -- flat ids (instance id base + mesh-local index), the instance's layer mask, normalize(basis n)."""
    local, inst = synth.multi_mesh_instances(4, 900, 0.3, 21)
    extra = inst[[1]].copy()
    extra["origin"] += np.float32([0.6, 0.2, -0.8])
    extra["layers"] = 0x4
    inst = np.concatenate([inst, extra])
    inst["layers"][:4] = [0x1, 0x2, 0x1, 0x80000000]
    rays = np.concatenate([po.grid_rays((0, 0, -12), (0, 0, 1), 64, 48, 50.0), synth.incoherent_rays(2000, 13)])
    sd.check_tlas(local, inst, rays, 0xFFFFFFFF)
    sd.check_tlas(local, inst, rays, 0x4)            # only the second placement of mesh 1
    sd.check_tlas(local, inst[:1], rays[:90], 0x1)   # one instance, below the threading threshold
