#!/usr/bin/env python3
"""bench.py — Mrays/s (primary, closest-hit) on BASELINE.json's headline config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: mrt_cast over the 4096x4096
primary-ray grid (config C3: 1 M-triangle soup, 8-bin SAH BVH2), rays already
resident in HBM in row-major order, 32-byte hit records written to HBM.  At N > 1
every rank traces its own full grid (one view per GPU, BVH replicated: weak
scaling) and the hit records of all views are assembled on rank 0 inside the
timed region: 4-byte hit tokens travel over RCCL and rank 0 rebuilds the 32-byte
records (mrt_expand_grid_tokens) on a side stream; frames are pipelined two deep
(the exchange of frame k runs beside the tracing of frame k+1) and all of them are
complete when the timed region closes.  --gather records sends the records themselves.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     algorithmic bytes / measured kernel time vs the 8 TB/s HBM peak
  cpu_baseline the reference's own CPU path (TinyBVH BVH8 under a range-split
               thread pool, oracle/_ref) on this box's host cores, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", help="C2 | C3 (headline) | C5")
    ap.add_argument("--mode", default="cast", choices=["cast", "tiled", "fused"],
                    help="cast: mrt_cast on row-major device rays (headline); tiled: mrt_cast_tiled; fused: mrt_cast_grid")
    ap.add_argument("--chunks", type=int, default=1, help="row chunks per step at N > 1 (copy of chunk c beside the tracing of chunk c+1)")
    ap.add_argument("--depth", type=int, default=2, choices=[1, 2],
                    help="frames in flight at N > 1: with 2 the exchange of frame k runs beside the tracing of frame k+1 "
                         "(double-buffered); every frame is complete on rank 0 when the timed region closes")
    ap.add_argument("--root-share", default="auto",
                    help="N > 1, token gather: fraction of its own view rank 0 traces itself (auto: 1 - 0.065 (N-1)); 1 = no balancing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--gather", default="tokens", choices=["tokens", "records"],
                    help="what travels to rank 0 at N > 1: 4-byte hit tokens (the 32-byte records are rebuilt on "
                         "rank 0, bit-identical) or the 32-byte records themselves")
    ap.add_argument("--grid-tile", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(cfg, verts):
    """The reference CPU path timed on this box's host cores (bounded: one pass over the
    same 4096^2 grid after a warm-up pass over 1/8 of it).  Checker code, used here only
    as the reported baseline."""
    from oracle import pyoracle as po
    # the box's CPU share for one GPU is 16 cores; never spawn more workers than that
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("MRT_CPU_BASELINE_THREADS", "16")))
    w, h = cfg["grid"]
    rays = po.grid_rays(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    if po.ref_available():
        rs = po.RefScene(verts)
        rs.cast_rays(rays[: rays.shape[0] // 8], n_threads=cores)
        dts = []
        for _ in range(3):  # median of three passes: ~1 s of wall time, ~16 core-seconds
            t0 = time.perf_counter()
            rs.cast_rays(rays, n_threads=cores)
            dts.append(time.perf_counter() - t0)
        dt = sorted(dts)[1]
        kind = "reference"
        what = "tinybvh::BVH8_CPU::Intersect (AVX2) under the ThreadPool range split" if po.ref().ref_has_avx2() else \
            "tinybvh::BVH4_CPU::Intersect (SSE) under the ThreadPool range split"
        rs.close()
    else:
        osc = po.OracleScene(verts)
        osc.trace(rays[: rays.shape[0] // 8], n_threads=cores)
        t0 = time.perf_counter()
        osc.trace(rays, n_threads=cores)
        dt = time.perf_counter() - t0
        kind, what = "port", "oracle/mrt_oracle.c scalar BVH2 walk, OpenMP"
    return dict(value=rays.shape[0] / dt / 1e6, unit="Mrays/s", cores=cores, kind=kind,
                sample=f"one pass (median of 3 for the reference path) over the full {w}x{h} grid of the same scene ({what}; BVH build excluded)")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # MRT_BENCH_ONE_DEVICE=1: every rank on device 0 (rehearsing the N > 1 code path on a one-GPU box; needs a
    # collective backend that accepts it)
    if os.environ.get("MRT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the process group exists even for one rank; with
    # MRT_REHEARSE_GATHER=1 a single rank also runs the chunked gather (self-gather), which is how
    # the N > 1 code path is rehearsed on a one-GPU box
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("MRT_REHEARSE_GATHER") == "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # MRT_BENCH_BACKEND=gloo: rehearsal of the N > 1 path with every rank on one GPU (RCCL refuses that);
        # the exchange then goes through host memory (sharded._gather) and the timing means nothing
        backend = os.environ.get("MRT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    from messyerraytracer_amd import capi, synth, sharded
    cfg = synth.CONFIGS[a.config]
    w, h = cfg["grid"]
    verts = synth.scene_vertices(cfg)
    scene = capi.Scene(verts)
    ctx = capi.Context(local_rank, grid_tile=a.grid_tile)
    stream = torch.cuda.current_stream(device)
    ctx.set_stream(stream.cuda_stream)
    scene.upload(ctx)

    # one view per rank: the camera orbits the scene centre (rank 0 = the config's own camera)
    ang = 2.0 * np.pi * rank / max(world, 1)
    o = np.array(cfg["origin"], dtype=np.float64)
    f = np.array(cfg["forward"], dtype=np.float64)
    rot = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    cam = capi.camera_look(tuple((rot @ o).astype(np.float32)), tuple((rot @ f).astype(np.float32)), w, h, cfg["fov"])

    def view_camera(r):  # the camera of rank r's view (rank 0 rebuilds every view's records from tokens)
        g = 2.0 * np.pi * r / max(world, 1)
        m = np.array([[np.cos(g), 0, np.sin(g)], [0, 1, 0], [-np.sin(g), 0, np.cos(g)]])
        return capi.camera_look(tuple((m @ o).astype(np.float32)), tuple((m @ f).astype(np.float32)), w, h, cfg["fov"])
    cams = [view_camera(r) for r in range(world)]

    n_rays = w * h
    d_rays = torch.empty(n_rays * 32, dtype=torch.uint8, device=device)
    ctx.generate_grid(cam, w, h, 0, h, d_rays)          # untimed: inputs resident in HBM
    trace_ms = []
    dev_flags = capi.FLAG_COHERENT | capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE

    job = None
    events = []  # pipelined frames: the cast is only queued (MRT_FLAG_ASYNC); timed with events on its stream

    # Balanced mode (N > 1, tokens): rank 0 also rebuilds every view's records, so it keeps only part of
    # its own view and the peers trace the rest of view 0 besides their own (sharded.BalancedViews).
    root_share = 1.0
    if world > 1 and a.gather == "tokens" and not a.no_gather:
        # expansion of one view costs ~6.5 % of tracing one (0.14 ms against 2.2 ms); rank 0 expands N of them
        root_share = max(0.3, 1.0 - 0.065 * (world - 1)) if a.root_share == "auto" else float(a.root_share)
    balanced = use_dist and root_share < 1.0
    foreign = {}  # view -> (first row, device rays) for rows of another rank's view this rank traces
    if balanced:
        for (view, y0, y1) in sharded.balanced_spans(world, h, root_share)[rank]:
            if view != rank:
                buf = torch.empty((y1 - y0) * w * 32, dtype=torch.uint8, device=device)
                ctx.generate_grid(cams[view], w, h, y0, y1, buf)
                foreign[view] = (y0, buf)

    def tracer(y0, y1, out, view=None):
        tok = capi.FLAG_TOKEN_OUT if job.token_mode else 0
        own = view is None or view == rank
        rays_ptr = d_rays.data_ptr() + y0 * w * 32 if own else foreign[view][1].data_ptr() + (y0 - foreign[view][0]) * w * 32
        if job.depth > 1:
            if a.mode == "tiled":
                raise SystemExit("--mode tiled is blocking: use --depth 1")
            tok |= capi.FLAG_ASYNC
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        if a.mode == "fused":
            ctx.cast_grid(cam if own else cams[view], w, h, y0=y0, y1=y1, hits=out, flags=capi.FLAG_HITS_ON_DEVICE | tok)
        elif a.mode == "tiled":
            if tok:
                raise SystemExit("--mode tiled writes records only: use --gather records")
            ctx.cast_tiled(rays_ptr, out, w, y1 - y0)
        else:
            ctx.cast(rays_ptr, out, count=(y1 - y0) * w, flags=dev_flags | tok)
        if job.depth > 1:
            e1.record(stream)
            events.append((e0, e1))
        else:
            trace_ms.append(ctx.stats()["last_trace_ms"])

    # rehearsal on one GPU only: make rank 0 rebuild its view K times, the load it carries with K ranks
    rehearse_views = int(os.environ.get("MRT_REHEARSE_VIEWS", "1")) if world == 1 else 1

    def expander(view, y0, y1, tokens, hits, stream):
        for _ in range(rehearse_views):
            ctx.expand_grid_tokens(cams[view], w, h, y0, y1, tokens, hits, stream=stream)

    chunks = a.chunks if use_dist else 1
    if balanced:
        chunks = 1
        job = sharded.BalancedViews(w, h, lambda view, y0, y1, out: tracer(y0, y1, out, view), expander, device,
                                    root_share=root_share, depth=a.depth)
    else:
        job = sharded.ShardedViews(w, h, tracer, device, chunks=chunks, gather=not a.no_gather, force_gather=use_dist,
                                   expander=expander if a.gather == "tokens" else None, depth=a.depth)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(a.warmup):
        job.step()
    job.finish()
    sync()
    trace_ms.clear()
    events.clear()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        job.step()
    job.finish()   # every frame's exchange and expansion is queued behind this; sync() waits for them
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0 and os.environ.get("MRT_BENCH_VERIFY") == "1" and use_dist and not a.no_gather:
        # rehearsal aid: every view assembled on rank 0 must equal a direct cast of that view on this device
        images = job.images if balanced else (job.images[0] if job.token_mode else job.images[(job.frame - 1) % job.depth])
        direct = torch.empty(n_rays * 32, dtype=torch.uint8, device=device)
        for v in range(world):
            ctx.cast_grid(cams[v], w, h, hits=direct, flags=capi.FLAG_HITS_ON_DEVICE)
            if not torch.equal(images[v], direct):
                raise SystemExit(f"view {v} assembled on rank 0 differs from a direct cast")
        print(f"verified {world} views on rank 0", file=sys.stderr, flush=True)

    if rank == 0:
        total_rays = n_rays * world * a.steps
        out = {
            "metric": "Mrays/s (primary, closest-hit)", "value": total_rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.config}: {verts.shape[0]}-triangle soup (seed {cfg.get('seed')}), 8-bin SAH BVH2, "
                                   f"{w}x{h} primary-ray grid per GPU, closest hit, rays+hits HBM-resident",
                       "entry": {"cast": "mrt_cast(COHERENT)", "tiled": "mrt_cast_tiled", "fused": "mrt_cast_grid"}[a.mode],
                       "views": world, "gather": "none" if not (use_dist and not a.no_gather) else
                       ("rccl gather of 4-byte hit tokens to rank 0 in %d chunks, 32-byte records rebuilt there" % chunks
                        if job.token_mode else "rccl gather of 32-byte records to rank 0 in %d chunks" % chunks),
                       "frames_in_flight": job.depth,
                       "balance": ("rank 0 traces %.0f %% of its view (it also rebuilds every view's records); the other ranks "
                                   "share the rest besides their own views" % (100 * root_share)) if balanced else "one view per rank"},
        }
        # roofline of the dominant kernel (trace_lane_kernel): algorithmic bytes / kernel time
        stats_path = os.path.join(ROOT, "tests", "golden", "traversal_stats.json")
        if events:  # queued casts: detection kernel + trace kernel between the two events
            trace_ms.extend(e0.elapsed_time(e1) for (e0, e1) in events)
        kernel_ms = float(np.sum(trace_ms)) / a.steps          # per step (all chunks), rank 0
        # rays rank 0 itself traces per step (balanced mode: part of its view)
        rank0_rays = sum(y1 - y0 for (_, y0, y1) in job.spans[0]) * w if balanced else n_rays
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": "trace_packet_asm_kernel<false>", "kernel_ms": kernel_ms, "rays_per_step_rank0": rank0_rays}
        if os.path.exists(stats_path):
            st = json.load(open(stats_path)).get(a.config)
            if st:
                bytes_per_launch = st["bytes_per_ray"] * rank0_rays
                roof.update(achieved=bytes_per_launch / (kernel_ms * 1e-3) / 1e9, bytes_per_ray=st["bytes_per_ray"],
                            n_int=st["n_int"], n_tri=st["n_tri"])
                roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            roof["traffic"] = json.load(open(pmc)).get(a.config, {}).get("hbm_bytes_per_launch")
        out["roofline"] = roof
        out["kernel_only_mrays"] = rank0_rays / (kernel_ms * 1e-3) / 1e6
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, verts)
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
