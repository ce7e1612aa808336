#!/usr/bin/env python3
"""bench.py — Mrays/s (primary, closest-hit) on BASELINE.json's configs.

    python bench.py --gpus N --steps K --warmup W [--config C3|C2|C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch.

  --config C3 (default; the headline: 1 M-triangle soup, 8-bin SAH BVH2, 4096x4096 grid)
      N = 1: mrt_cast(COHERENT) over the grid's rays, resident in HBM in row-major order; 32-byte hit records
      written to HBM.  N > 1: one such view per GPU (camera orbiting the scene, BVH replicated: WEAK scaling), all
      views assembled as records on rank 0 inside the timed region (4-byte hit tokens over RCCL, records rebuilt on
      rank 0, frames pipelined two deep).
  --config C5 (BASELINE config 5: 10 M-triangle multi-mesh scene, ONE 8192x8192 grid)
      the grid's rows sharded over the N ranks (sharded.ShardedGrid: contiguous row blocks, mrt_cast_grid per chunk,
      token gather + rebuild on rank 0 overlapped chunk by chunk): STRONG scaling; N = 1 is the single-GPU C5 figure.

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline      of the dominant kernel, named from the path actually taken (mrt_stats.last_kernel).  `achieved` =
                the bytes the kernel's algorithm requests per launch — 64 B per row (node or triangle) a packet
                fetches, counted by the counting build of the same kernel in this run, + the rays read and records
                written — over the kernel's mean duration in the timed region (HIP events on its stream).  `traffic`
                = HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json).  The per-ray figure
                of SURVEY 8(d) is kept as `algorithmic_per_ray` (a packet fetches a node once for 64-128 rays, so that
                figure is not a bandwidth).  `valu_issue`: the bound that actually applies — vector instructions per
                launch (committed PMC pass) x 2 cycles / (1024 SIMDs x the clock measured in that pass).
  verified      digest (hit count, prim / t hashes) of the LAST TIMED frame against the oracle's digest of the whole
                batch (tests/golden/full_digests.json): a kernel that skipped work cannot print a line.
  cpu_baseline  the reference's own CPU path (TinyBVH BVH8 under a range-split thread pool, oracle/_ref) on this
                box's host cores, N = 1 only.
  end_to_end_host_mrays  the same batch from pageable host arrays through mrt_cast (PCIe inclusive; never `value`).
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
N_SIMD = 256 * 4       # 256 CUs x 4 SIMD-32 (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)  # (the first ~10 casts after start-up run 1-3 % slower while the clocks settle: 3 warm-up steps 8 450-8 500 Mrays/s, 10: 8 560-8 610)
    ap.add_argument("--config", default="C3", help="C2 | C3 (headline; N > 1: one view per GPU) | C5 (one grid, rows sharded over the GPUs)")
    ap.add_argument("--mode", default="cast", choices=["cast", "tiled", "fused"],
                    help="views: cast = mrt_cast on row-major device rays (headline); tiled: mrt_cast_tiled; fused: mrt_cast_grid")
    ap.add_argument("--chunks", type=int, default=0, help="row chunks per step (copy of chunk c beside the tracing of chunk c+1); "
                                                          "default 1 for views, 4 for the row-sharded grid at N > 1")
    ap.add_argument("--depth", type=int, default=2, choices=[1, 2],
                    help="views at N > 1: frames in flight (2: the exchange of frame k beside the tracing of frame k+1)")
    ap.add_argument("--root-share", default="auto",
                    help="views at N > 1, token gather: fraction of its own view rank 0 traces itself "
                         "(auto: from the expansion / trace times measured at start-up); 1 = no balancing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--gather", default="tokens", choices=["tokens", "records"],
                    help="what travels to rank 0 at N > 1: 4-byte hit tokens (records rebuilt on rank 0, bit-identical) or the records")
    ap.add_argument("--grid-tile", type=int, default=0)
    return ap.parse_args()


def one_socket_cpus():
    """The CPUs of this process's affinity mask that sit on ONE socket (the one holding most of them), their count over
    all sockets, and the CPU model: SURVEY 8(d) asks for the reference's pool on the cores of one socket, stated."""
    allowed = sorted(os.sched_getaffinity(0))
    by_pkg = {}
    for c in allowed:
        try:
            pkg = int(open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id").read())
        except (OSError, ValueError):
            pkg = 0
        by_pkg.setdefault(pkg, []).append(c)
    pkg = max(by_pkg, key=lambda k: len(by_pkg[k]))
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return by_pkg[pkg], len(allowed), len(by_pkg), model


def source_sha16():
    """sha256 over the kernel and host sources of libmrt_hip.so (first 16 hex digits): counter passes committed under
    profiles/ are accepted only for the tree they were taken on."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "messyerraytracer_amd", "csrc")
    for d, _, files in sorted(os.walk(base)):
        for f in sorted(files):
            if f.endswith((".hip", ".h", ".hpp", ".cpp")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(cfg, verts):
    """The reference CPU path timed on this box's host cores (bounded sample): SURVEY 8(d) — the reference's pool shape
    (persistent helpers, caller + helpers = all allowed cores of one socket, src/dispatch/thread_pool.h:41-55), pinned to
    that socket, 1 warm-up, median of 5, BVH build excluded, core count and CPU model stated.  Checker code, used here only
    as the reported baseline."""
    from oracle import pyoracle as po
    socket_cpus, n_allowed, n_sockets, model = one_socket_cpus()
    # `value` is taken on this job's share of the host: 16 CPUs per GPU on the pool's boxes (their affinity mask shows the
    # whole machine, which other jobs are using); MRT_CPU_BASELINE_THREADS sets another count (0 = every CPU of the socket).
    # The all-CPUs-of-one-socket figure SURVEY 8(d) asks for is measured too and reported beside it.
    share = int(os.environ.get("MRT_CPU_BASELINE_THREADS", "16"))
    cores = len(socket_cpus) if share <= 0 else min(len(socket_cpus), share)
    cpus = socket_cpus[:cores]
    w, h = cfg["grid"]
    # a bounded sample: ~2 M rays per allowed core, at most 2^24 (one pass of the reference at ~4 Mrays/s per core takes
    # about half a second; 1 + 5 passes stay well inside the bench's minutes)
    want = min(1 << 24, max(1 << 20, cores << 21))
    rows = min(h, max(1, want // w))
    y0 = (h - rows) // 2
    rays = po.grid_rays(cfg["origin"], cfg["forward"], w, h, cfg["fov"], y0, y0 + rows)
    before = os.sched_getaffinity(0)
    os.sched_setaffinity(0, cpus)          # the pool's helpers are created below and inherit this mask
    try:
        if po.ref_available():
            rs = po.RefScene(verts)
            rs.cast_rays(rays, n_threads=cores)   # warm-up: creates the persistent pool, touches the tree
            dts = []
            for _ in range(5):
                t0 = time.perf_counter()
                rs.cast_rays(rays, n_threads=cores)
                dts.append(time.perf_counter() - t0)
            dt = sorted(dts)[2]
            kind = "reference"
            what = "tinybvh::BVH8_CPU::Intersect (AVX2+FMA)" if po.ref().ref_has_avx2() else "tinybvh::BVH4_CPU::Intersect (SSE)"
            what += " per ray under a persistent range-split pool (chunks = threads, caller runs chunk 0)"
            whole_socket = None
            if len(socket_cpus) > cores:   # every CPU of the socket (SMT siblings included), median of 3
                os.sched_setaffinity(0, socket_cpus)
                rs.cast_rays(rays, n_threads=len(socket_cpus))
                d2 = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    rs.cast_rays(rays, n_threads=len(socket_cpus))
                    d2.append(time.perf_counter() - t0)
                whole_socket = dict(cores=len(socket_cpus), value=rays.shape[0] / sorted(d2)[1] / 1e6,
                                    note="every CPU of one socket in the affinity mask (shared with other jobs on the pool's boxes), median of 3")
            rs.close()
        else:
            osc = po.OracleScene(verts)
            osc.trace(rays[: rays.shape[0] // 8], n_threads=cores)
            dts = []
            for _ in range(5):
                t0 = time.perf_counter()
                osc.trace(rays, n_threads=cores)
                dts.append(time.perf_counter() - t0)
            dt = sorted(dts)[2]
            kind, what, whole_socket = "port", "oracle/mrt_oracle.c scalar BVH2 walk, OpenMP", None
    finally:
        os.sched_setaffinity(0, before)
    return dict(value=rays.shape[0] / dt / 1e6, unit="Mrays/s", cores=cores, kind=kind,
                cpu_model=model, cpus_allowed=n_allowed, sockets_in_mask=n_sockets, pinned_to=f"{cores} CPUs of one socket",
                whole_socket=whole_socket,
                flags="-O2 -mavx2 -mfma -ffp-contract=off (oracle/Makefile)",
                sample=f"rows {y0}..{y0 + rows} of the {w}x{h} grid ({rays.shape[0]} rays), 1 warm-up + median of 5 passes ({what}; BVH build excluded)")


def digest_of(hits_u8: torch.Tensor, first_index: int = 0) -> dict:
    """Digest of 32-byte hit records on the device (checker code: oracle/digests.py)."""
    from oracle import digests
    rec = hits_u8.view(torch.int32).view(-1, 8)
    prim = rec[:, 1].contiguous().cpu().numpy()
    t = rec[:, 0].contiguous().cpu().numpy().view(np.float32)
    return digests.digest_columns(prim, t, first_index)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world == 1 and a.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # MRT_BENCH_ONE_DEVICE=1: every rank on device 0 (rehearsing the N > 1 code path on a one-GPU box; needs a
    # collective backend that accepts it: MRT_BENCH_BACKEND=gloo, the exchange then goes through host memory)
    if os.environ.get("MRT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and os.environ.get("MRT_REHEARSE_GATHER") == "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    # ONE stream for the casts, torch's copies and the collectives' stream-order semantics: a stream of our own made
    # torch's current stream (the default stream's handle is 0, which mrt_set_stream reads as "the context's own
    # stream": the casts would then run unordered beside the exchange)
    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)
    scene.upload(ctx)
    rows_mode = a.config == "C5"   # BASELINE config 5: ONE grid, rows sharded (strong scaling)
    n_rays = w * h
    trace_ms, events = [], []
    kernels_seen = set()

    def note_kernel():
        k = ctx.stats()["last_kernel"]
        if k:
            kernels_seen.add(k)

    # ------------------------------------------------------------------------------------------------------------
    if rows_mode:
        cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
        token_mode = use_dist and a.gather == "tokens" and not a.no_gather

        def tracer(y0, y1, out):
            ctx.cast_grid(cam, w, h, y0=y0, y1=y1, hits=out,
                          flags=capi.FLAG_HITS_ON_DEVICE | (capi.FLAG_TOKEN_OUT if token_mode else 0))
            trace_ms.append(ctx.stats()["last_trace_ms"])
            note_kernel()

        def expander(y0, y1, tokens, hits, s):
            ctx.expand_grid_tokens(cam, w, h, y0, y1, tokens, hits, stream=s)

        chunks = a.chunks or (4 if use_dist else 1)
        job = sharded.ShardedGrid(w, h, tracer, device, chunks=chunks, gather=not a.no_gather,
                                  expander=expander if token_mode else None)
        step = job.step
        finish = lambda: None  # noqa: E731  (ShardedGrid.step completes its frame)
        rank0_rays = (job.y1 - job.y0) * w
        total_rays_per_step = n_rays
        scaling = "strong"
        balance = "contiguous row blocks, one per rank"
        frames_in_flight = 1
        root_share = 1.0
    # ------------------------------------------------------------------------------------------------------------
    else:
        # one view per rank: the camera orbits the scene centre (rank 0 = the config's own camera)
        o = np.array(cfg["origin"], dtype=np.float64)
        f = np.array(cfg["forward"], dtype=np.float64)

        def view_camera(r):
            g = 2.0 * np.pi * r / max(world, 1)
            m = np.array([[np.cos(g), 0, np.sin(g)], [0, 1, 0], [-np.sin(g), 0, np.cos(g)]])
            return capi.camera_look(tuple((m @ o).astype(np.float32)), tuple((m @ f).astype(np.float32)), w, h, cfg["fov"])
        cams = [view_camera(r) for r in range(world)]
        cam = cams[rank]
        d_rays = torch.empty(n_rays * 32, dtype=torch.uint8, device=device)
        ctx.generate_grid(cam, w, h, 0, h, d_rays)          # untimed: inputs resident in HBM
        dev_flags = capi.FLAG_COHERENT | capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE

        # Balanced mode (N > 1, tokens): rank 0 also rebuilds every view's records, so it keeps only part of its own
        # view and the peers trace the rest of view 0 besides their own (sharded.BalancedViews).  The share comes
        # from what one expansion and one trace of a view cost on THIS device, measured now.
        root_share = 1.0
        if world > 1 and a.gather == "tokens" and not a.no_gather:
            if a.root_share == "auto":
                tok = torch.empty(n_rays * 4, dtype=torch.uint8, device=device)
                rec = torch.empty(n_rays * 32, dtype=torch.uint8, device=device)
                t_tr, t_ex = [], []
                for _ in range(3):
                    ctx.cast_grid(cam, w, h, hits=tok, flags=capi.FLAG_HITS_ON_DEVICE | capi.FLAG_TOKEN_OUT)
                    t_tr.append(ctx.stats()["last_trace_ms"])
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    ctx.expand_grid_tokens(cam, w, h, 0, h, tok, rec, stream=stream.cuda_stream)
                    e1.record(stream)
                    torch.cuda.synchronize(device)
                    t_ex.append(e0.elapsed_time(e1))
                ratio = float(np.median(t_ex) / np.median(t_tr))
                root_share = max(0.3, 1.0 - ratio * (world - 1))
                del tok, rec
                share = torch.tensor([root_share], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
                dist.broadcast(share, src=0)          # every rank must cut the views the same way
                root_share = float(share.item())
            else:
                root_share = float(a.root_share)
        balanced = use_dist and root_share < 1.0
        foreign = {}  # view -> (first row, device rays) for rows of another rank's view this rank traces
        if balanced:
            for (view, y0, y1) in sharded.balanced_spans(world, h, root_share)[rank]:
                if view != rank:
                    buf = torch.empty((y1 - y0) * w * 32, dtype=torch.uint8, device=device)
                    ctx.generate_grid(cams[view], w, h, y0, y1, buf)
                    foreign[view] = (y0, buf)
        job = None

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
                note_kernel()

        # rehearsal on one GPU only: make rank 0 rebuild its view K times, the load it carries with K ranks
        rehearse_views = int(os.environ.get("MRT_REHEARSE_VIEWS", "1")) if world == 1 else 1

        def expander(view, y0, y1, tokens, hits, s):
            for _ in range(rehearse_views):
                ctx.expand_grid_tokens(cams[view], w, h, y0, y1, tokens, hits, stream=s)

        chunks = (a.chunks or 1) if use_dist else 1
        if balanced:
            chunks = 1
            job = sharded.BalancedViews(w, h, lambda view, y0, y1, out: tracer(y0, y1, out, view), expander, device,
                                        root_share=root_share, depth=a.depth)
        else:
            job = sharded.ShardedViews(w, h, tracer, device, chunks=chunks, gather=not a.no_gather, force_gather=use_dist,
                                       expander=expander if a.gather == "tokens" else None, depth=a.depth)
        step, finish = job.step, job.finish
        rank0_rays = sum(y1 - y0 for (_, y0, y1) in job.spans[0]) * w if balanced else n_rays
        total_rays_per_step = n_rays * world
        scaling = "weak"
        balance = ("rank 0 traces %.0f %% of its view (it also rebuilds every view's records; share from the expansion / trace "
                   "times measured at start-up); the other ranks share the rest besides their own views" % (100 * root_share)) \
            if balanced else "one view per rank"
        frames_in_flight = job.depth

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    last = None
    for _ in range(a.warmup):
        last = step()
    finish()
    sync()
    trace_ms.clear()
    events.clear()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        last = step()
    finish()   # every frame's exchange and expansion is queued behind this; sync() waits for them
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- what the timed launches wrote: digest of the last frame against the oracle's digest of the whole batch ----
    verified = None
    if rank == 0 and not a.no_verify:
        golden = os.path.join(ROOT, "tests", "golden", "full_digests.json")
        want = json.load(open(golden)).get(a.config) if os.path.exists(golden) else None
        if want is not None:
            from oracle import digests
            if rows_mode:
                image = last if (not job.gather or job.rank == 0) else None
                if not job.gather and world > 1:   # --no-gather: only this rank's block is here
                    got, want_d = digest_of(image, job.y0 * w), None
                    blocks = want.get("row_blocks")
                    if blocks and world == 8:
                        want_d = blocks[0]
                else:
                    got, want_d = digest_of(image), want
            else:
                if not use_dist or a.no_gather:
                    image = last
                elif balanced:
                    image = job.images[0]
                else:
                    image = (job.images[0] if job.token_mode else job.images[(job.frame - 1) % job.depth])[0]
                got, want_d = digest_of(image), want   # view 0 = the config's own camera
            if want_d is not None:
                if not digests.same(got, want_d):
                    raise SystemExit(f"bench: the last timed frame does not match the oracle's digest of {a.config}: got {got}, want {want_d}")
                verified = dict(against="tests/golden/full_digests.json", hit_count=got["hit_count"], prim_hash=got["prim_hash"],
                                t_hash=got["t_hash"], rays=got["rays"])

    if rank == 0 and os.environ.get("MRT_BENCH_VERIFY") == "1" and use_dist and not a.no_gather and not rows_mode:
        # rehearsal aid: every view assembled on rank 0 must equal a direct cast of that view on this device
        images = job.images if balanced else (job.images[0] if job.token_mode else job.images[(job.frame - 1) % job.depth])
        direct = torch.empty(n_rays * 32, dtype=torch.uint8, device=device)
        for v in range(world):
            ctx.cast_grid(cams[v], w, h, hits=direct, flags=capi.FLAG_HITS_ON_DEVICE)
            if not torch.equal(images[v], direct):
                raise SystemExit(f"view {v} assembled on rank 0 differs from a direct cast")
        print(f"verified {world} views on rank 0", file=sys.stderr, flush=True)

    if rank == 0:
        entry = "mrt_cast_grid rows [y0, y1)" if rows_mode else {"cast": "mrt_cast(COHERENT)", "tiled": "mrt_cast_tiled", "fused": "mrt_cast_grid"}[a.mode]
        gather_on = use_dist and not a.no_gather
        token_on = gather_on and job.token_mode
        out = {
            "metric": "Mrays/s (primary, closest-hit)", "value": total_rays_per_step * a.steps / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{a.config}: {verts.shape[0]}-triangle {'multi-mesh scene (64 rotated soups, flattened)' if rows_mode else 'soup'} "
                                    f"(seed {cfg.get('seed')}), 8-bin SAH BVH2, " +
                                    (f"ONE {w}x{h} primary-ray grid, rows sharded over the GPUs" if rows_mode else f"{w}x{h} primary-ray grid per GPU") +
                                    ", closest hit, " + ("rays generated in the kernel" if rows_mode or a.mode == "fused" else "rays+hits HBM-resident")),
                       "entry": entry, "views": 1 if rows_mode else world,
                       "gather": "none" if not gather_on else
                       ("rccl gather of 4-byte hit tokens to rank 0 in %d chunks, 32-byte records rebuilt there" % chunks
                        if token_on else "rccl gather of 32-byte records to rank 0 in %d chunks" % chunks),
                       "frames_in_flight": frames_in_flight, "balance": balance},
        }
        out["source_sha16"] = source_sha16()   # of messyerraytracer_amd/csrc: what committed counter passes are keyed on
        if verified is not None:
            out["verified"] = verified
        # ---- roofline of the dominant kernel --------------------------------------------------------------------
        if events:  # queued casts: detection kernel + trace kernel between the two events
            trace_ms.extend(e0.elapsed_time(e1) for (e0, e1) in events)
            ctx.cast(d_rays.data_ptr(), torch.empty(n_rays * 32, dtype=torch.uint8, device=device), count=n_rays, flags=dev_flags)
            note_kernel()   # (ASYNC casts report no kernel: one blocking cast of the same batch names it)
        # per step (all chunks of a step summed), rank 0: the MEDIAN over the timed steps; the mean beside it
        per_step = np.asarray(trace_ms, dtype=np.float64)
        per_step = per_step[: (per_step.size // a.steps) * a.steps].reshape(a.steps, -1).sum(axis=1) if per_step.size >= a.steps else per_step
        kernel_ms = float(np.median(per_step))
        kernel_ms_mean = float(np.sum(trace_ms)) / a.steps
        kernel_id = sorted(kernels_seen)[0] if kernels_seen else 0
        variant = ctx.last_kernel_variant()     # the instantiation of the last blocking cast, as rocprofv3 spells it
        kname = variant or (capi.kernel_name(kernel_id) + "<any_hit=false>")
        reads_rays = not (rows_mode or a.mode == "fused")
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": kname, "kernels_seen": sorted(capi.kernel_name(k) for k in kernels_seen), "kernel_ms": kernel_ms,
                "kernel_ms_is": "median over the timed steps of the per-step sum of HIP-event kernel times", "kernel_ms_mean": kernel_ms_mean,
                "rays_per_step_rank0": rank0_rays}
        # the rows the kernel fetches: counting build of the same kernel on the same batch, now (outside the timed region)
        cctx = capi.Context(local_rank, grid_tile=a.grid_tile, count_visits=1)
        scene.upload(cctx)
        tmp = torch.empty(rank0_rays * 32, dtype=torch.uint8, device=device)
        if rows_mode:
            cctx.cast_grid(cam, w, h, y0=job.y0, y1=job.y1, hits=tmp, flags=capi.FLAG_HITS_ON_DEVICE)
        elif a.mode == "fused":
            cctx.cast_grid(cam, w, h, y0=0, y1=rank0_rays // w, hits=tmp, flags=capi.FLAG_HITS_ON_DEVICE)
        else:
            cctx.cast(d_rays.data_ptr(), tmp, count=rank0_rays, flags=dev_flags)
        cs = cctx.stats()
        cctx.close()
        del tmp
        if cs["last_kernel"] == kernel_id and cs["wave_node_fetches"]:
            row_bytes = 64 * (cs["wave_node_fetches"] + cs["wave_tri_fetches"])
            io_bytes = (32 if reads_rays else 0) * rank0_rays + 32 * rank0_rays
            roof.update(achieved=(row_bytes + io_bytes) / (kernel_ms * 1e-3) / 1e9,
                        algorithmic={"node_rows": cs["wave_node_fetches"], "tri_rows": cs["wave_tri_fetches"], "row_bytes": row_bytes,
                                     "ray_and_hit_bytes": io_bytes,
                                     "note": "64 B per row a packet fetches (one fetch serves its 64-128 rays) + rays read + records written; "
                                             "counted by the counting build of this kernel in this run"})
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        stats_path = os.path.join(ROOT, "tests", "golden", "traversal_stats.json")
        if os.path.exists(stats_path):
            st = json.load(open(stats_path)).get(a.config)
            if st:
                roof["algorithmic_per_ray"] = {
                    "bytes_per_ray": st["bytes_per_ray"], "n_int": st["n_int"], "n_tri": st["n_tri"],
                    "implied_gbs": st["bytes_per_ray"] * rank0_rays / (kernel_ms * 1e-3) / 1e9,
                    "note": "SURVEY 8(d): every ray charged for every node it needs (the oracle's one-ray walk); a packet kernel "
                            "fetches a node once per packet, so this is a work figure, not a bandwidth"}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            rec = json.load(open(pmc)).get(a.config, {})
            sha = source_sha16()
            # accepted only for the very instantiation this run used, taken on this very source tree
            if rec and not (variant and variant in (rec.get("kernel") or "")):
                roof["traffic_rejected"] = f"profiles/pmc_traffic.json[{a.config}] is of {rec.get('kernel')!r}, this run used {variant!r}"
            elif rec and rec.get("source_sha16") != sha:
                roof["traffic_rejected"] = f"profiles/pmc_traffic.json[{a.config}] was taken on source tree {rec.get('source_sha16')}, this is {sha}"
            elif rec:
                scale = rank0_rays / rec.get("rays_per_launch", rank0_rays)
                roof["traffic"] = rec.get("hbm_bytes_per_launch") * scale if rec.get("hbm_bytes_per_launch") else None
                roof["traffic_source"] = (f"profiles/pmc_traffic.json[{a.config}] ({rec.get('round')}, commit {rec.get('commit')}, source {sha}): separate --pmc "
                                          "passes of this command; FETCH_SIZE x2 (gfx950) + WRITE_SIZE")
                if roof["traffic"]:
                    roof["hbm_frac_measured"] = roof["traffic"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                if rec.get("valu_insts_per_launch"):
                    clk = rec.get("clock_ghz", 2.4)
                    floor_ms = rec["valu_insts_per_launch"] * scale * 2.0 / N_SIMD / (clk * 1e9) * 1e3
                    roof["valu_issue"] = {"insts_per_launch": rec["valu_insts_per_launch"] * scale, "cycles_per_inst": 2, "simds": N_SIMD,
                                          "clock_ghz": clk, "floor_ms": floor_ms, "frac": floor_ms / kernel_ms,
                                          "source": f"SQ_INSTS_VALU and GRBM_GUI_ACTIVE passes in profiles/ ({rec.get('round')})"}
                    roof["limiter"] = "valu_issue"
        out["roofline"] = roof
        out["kernel_only_mrays"] = rank0_rays / (kernel_ms * 1e-3) / 1e6
        if world == 1 and not rows_mode and a.mode != "fused":
            # beside the headline entry (rays resident in HBM, the reference's cast_rays contract): the same grid through
            # mrt_cast_grid, which generates the rays in the kernel -- the device-side form of cast_debug_rays(origin,
            # forward, W, H, fov) (src/godot/raytracer_debug.cpp:539-634).  Outside the timed region; 12 blocking casts.
            rec = torch.empty(n_rays * 32, dtype=torch.uint8, device=device)
            fk, t1 = [], []
            for i in range(12):
                t0 = time.perf_counter()
                ctx.cast_grid(cam, w, h, hits=rec, flags=capi.FLAG_HITS_ON_DEVICE)
                t1.append(time.perf_counter() - t0)
                fk.append(ctx.stats()["last_trace_ms"])
            out["fused_grid"] = {"entry": "mrt_cast_grid", "kernel": ctx.last_kernel_variant(), "kernel_ms": float(np.median(fk[2:])),
                                 "ms_per_step": float(np.median(t1[2:])) * 1e3, "mrays": n_rays / float(np.median(t1[2:])) / 1e6}
            del rec
        if world == 1:
            # the reference's cast_rays(rays, results, count) contract: pageable host arrays in and out (PCIe inclusive)
            if not rows_mode and os.environ.get("MRT_BENCH_NO_HOST_PATH") != "1":
                from messyerraytracer_amd import types as T
                h_rays = np.zeros(n_rays, dtype=T.RAY32)
                ctx.d2h(h_rays.view(np.uint8), d_rays.data_ptr())
                h_hits = np.zeros(n_rays, dtype=T.HIT32)
                ctx.cast(h_rays, h_hits, flags=capi.FLAG_COHERENT)
                t1 = time.perf_counter()
                ctx.cast(h_rays, h_hits, flags=capi.FLAG_COHERENT)
                out["end_to_end_host_mrays"] = n_rays / (time.perf_counter() - t1) / 1e6
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(cfg, verts)
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
