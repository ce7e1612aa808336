import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
from messyerraytracer_amd import capi, synth
cfg = synth.CONFIGS["C5"]
local, inst = synth.multi_mesh_instances(cfg["n_meshes"], cfg["tris_per_mesh"], cfg["s"], cfg["seed"])
n = 1 << 22
rays = synth.incoherent_rays(n, 7)
out = {}
opts = dict(leaf_wait=int(os.environ.get("MRT_LEAF_WAIT", "0")), refill=int(os.environ.get("MRT_REFILL", "0")),
            stack_override=int(os.environ.get("MRT_LDS_DEPTH", "0")))
for name in ("two_level", "flat"):
    c = capi.Context(0, **opts)
    if name == "two_level":
        c.upload_two_level_scene(local, inst, blas_on_device=True)
    else:
        c.build_instanced_scene_device(local, inst)
    d_rays, d_hits = c.device_alloc(n * 32), c.device_alloc(n * 32)
    c.h2d(d_rays, rays)
    dev = capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE
    for label, flags in (("unsorted", dev | capi.FLAG_COHERENT), ("sorted", dev)):
        ms = []
        for _ in range(4):
            c.cast(d_rays, d_hits, count=n, flags=flags)
            s = c.stats(); ms.append(s["last_trace_ms"] + s["last_sort_ms"])
        out[f"{name}_{label}_ms"] = float(np.median(ms[1:])); out[f"{name}_{label}_mrays"] = n / np.median(ms[1:]) / 1e3
    c.close()
out['opts'] = opts
print(json.dumps(out))
