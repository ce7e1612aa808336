"""One config's scene built on the device a few times in one form (for rocprofv3 --kernel-trace --stats).

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_build -- python3 tools/prof_build.py --config C3 --form sah
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from messyerraytracer_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--form", default="sah", choices=["radix", "ploc", "sah"])
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    tris = capi.make_triangles(synth.scene_vertices(synth.CONFIGS[a.config]))
    ctx = capi.Context(0)
    d = ctx.device_alloc(tris.nbytes)
    ctx.h2d(d, tris)
    kw = {"ploc": a.form == "ploc", "sah": a.form == "sah"}
    ms = []
    for _ in range(a.rounds):
        ctx.build_scene_device(d, n_tris=tris.shape[0], on_device=True, **kw)
        ms.append(ctx.stats()["last_build_ms"])
    print("build_ms", ms, "rows", ctx.scene_info()["n_wide_nodes"])
    ctx.device_free(d)
    ctx.close()


if __name__ == "__main__":
    main()
