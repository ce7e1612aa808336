"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle
on the same seeded inputs, against the committed golden vectors of the
reference, and through size-independent properties at BASELINE.json's full sizes."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from messyerraytracer_amd import capi, synth, types as T
from oracle import pyoracle as po
from oracle import digests
import parity

pytestmark = pytest.mark.gpu
# the four-wide packet walk is in builds made with MRT_WITH_QUAD=1 only (include/mrt_hip.h, mrt_kernel_available)
QUAD = pytest.param(capi.KERNEL_PACKET_QUAD, marks=pytest.mark.skipif(not capi.kernel_available(capi.KERNEL_PACKET_QUAD), reason="built without MRT_WITH_QUAD"))


def _golden(name):
    return np.load(os.path.join(GOLDEN, name))


def _full_digest(name):
    """The oracle's digest of the WHOLE batch of a config (tests/golden/make_full_digests.py)."""
    return json.load(open(os.path.join(GOLDEN, "full_digests.json")))[name]


def _assert_digest(got, want, what):
    assert digests.same(got, want), f"{what}: the digest of every ray of the batch differs from the oracle's: {got} vs {want}"


def _sample_grid_rays(cfg, idx):
    """Rays of the row-major grid at flat indices idx (generated row by row with the oracle)."""
    w, h = cfg["grid"]
    ys, xs = idx // w, idx % w
    out = np.zeros(idx.shape[0], dtype=T.RAY32)
    for y in np.unique(ys):
        row = po.grid_rays(cfg["origin"], cfg["forward"], w, h, cfg["fov"], int(y), int(y) + 1)
        sel = ys == y
        out[sel] = row[xs[sel]]
    return out


class DeviceArray:
    """Device buffer through the C-ABI helpers (no torch involved)."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        self.ptr = ctx.device_alloc(nbytes)

    def upload(self, arr):
        self.ctx.h2d(self.ptr, arr)
        return self

    def download(self, dtype, count):
        out = np.zeros(count, dtype=dtype)
        self.ctx.d2h(out, self.ptr)
        return out

    def free(self):
        if self.ptr:
            self.ctx.device_free(self.ptr)
            self.ptr = 0


@pytest.fixture(scope="module")
def soup1k(ctx):
    v = synth.soup(1000, 0.5, 1)
    return v, capi.Scene(v), po.OracleScene(v)


# ---------------------------------------------------------------------------
# config C1 and the small fixtures
# ---------------------------------------------------------------------------
def test_c1_cube_all_entry_points(ctx):
    c1 = synth.CONFIGS["C1"]
    w, h = c1["grid"]
    v = synth.cube()
    scene, osc = capi.Scene(v), po.OracleScene(v)
    scene.upload(ctx)
    assert ctx.is_available()
    rays = po.grid_rays(c1["origin"], c1["forward"], w, h, c1["fov"])
    want = osc.trace(rays)
    cam = capi.camera_look(c1["origin"], c1["forward"], w, h, c1["fov"])
    # fused raygen + trace
    parity.assert_exact(ctx.cast_grid(cam, w, h), want, "C1 cast_grid")
    # host rays through mrt_cast (coherent and sorted)
    parity.assert_exact(ctx.cast(rays, flags=capi.FLAG_COHERENT), want, "C1 cast coherent")
    parity.assert_exact(ctx.cast(rays, flags=capi.FLAG_FORCE_SORT), want, "C1 cast sorted")
    # device ray generation is bit-identical to the reference formula
    d_rays = DeviceArray(ctx, rays.nbytes)
    ctx.generate_grid(cam, w, h, 0, h, d_rays.ptr)
    assert d_rays.download(T.RAY32, w * h).tobytes() == rays.tobytes()
    d_hits = DeviceArray(ctx, w * h * 32)
    ctx.cast_tiled(d_rays.ptr, d_hits.ptr, w, h)
    parity.assert_exact(d_hits.download(T.HIT32, w * h), want, "C1 cast_tiled")
    d_rays.free(); d_hits.free()
    # the reference's own result for this config
    g = _golden("g1_cube.npz")
    got = ctx.cast_grid(cam, w, h)
    parity.assert_reference_parity(got["prim_id"], got["t"], g["hits_ref"]["prim_id"], g["hits_ref"]["t"], rays, osc.tris, "C1 vs reference")
    # any-hit
    any_got = ctx.cast(rays, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_COHERENT | capi.FLAG_BOOL_OUT).astype(bool)
    assert np.array_equal(any_got, want["prim_id"] >= 0)
    assert np.array_equal(any_got, g["any_ref"])


def test_g2_g3_small_soup(ctx, soup1k):
    v, scene, osc = soup1k
    scene.upload(ctx)
    for name in ("g2_soup1k_grid.npz", "g3_soup1k_incoherent.npz"):
        g = _golden(name)
        rays = g["rays"]
        want = osc.trace(rays)
        got = ctx.cast(rays, flags=capi.FLAG_COHERENT)
        parity.assert_exact(got, want, name)
        got_plain = ctx.cast(rays)   # not coherent, but at most 8 192 rays: one ray per wave in the lane kernel, no sort
        assert got_plain.tobytes() == got.tobytes()
        assert ctx.stats()["last_kernel_launches"] == 1
        got_sorted = ctx.cast(rays, flags=capi.FLAG_FORCE_SORT)  # the device Morton sort (what batches above 8 192 rays get)
        assert got_sorted.tobytes() == got.tobytes(), "sort on / sort off must give identical results"
        assert ctx.stats()["last_kernel_launches"] == 3
        parity.assert_reference_parity(got["prim_id"], got["t"], g["hits_ref"]["prim_id"], g["hits_ref"]["t"], rays, osc.tris, name)
        any_got = ctx.cast(rays, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT).astype(bool)
        assert np.array_equal(any_got, want["prim_id"] >= 0)
        # any-hit records (not bool): prim >= 0 iff hit, and the hit is a real intersection
        any_rec = ctx.cast(rays, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_COHERENT)
        assert np.array_equal(any_rec["prim_id"] >= 0, want["prim_id"] >= 0)
        hit = any_rec["prim_id"] >= 0
        assert (any_rec["t"][hit] >= want["t"][hit]).all()


def test_host_layout_path(ctx, soup1k):
    """60-byte Ray in, 44-byte Intersection out; conversion loops run on the device."""
    v, scene, osc = soup1k
    scene.upload(ctx)
    rays = np.concatenate([po.grid_rays((0, 0, -12), (0, 0, 1), 40, 30, 50.0), synth.incoherent_rays(777, 5)])
    host = po.make_host_rays(rays)
    want = po.unpack_hits(osc.trace(rays), host)
    for flags in (capi.FLAG_HOST_LAYOUT | capi.FLAG_COHERENT, capi.FLAG_HOST_LAYOUT):
        got = ctx.cast(host, flags=flags)
        assert got.dtype == T.HOST_HIT44
        assert got.tobytes() == want.tobytes()
    b = ctx.cast(host, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_HOST_LAYOUT | capi.FLAG_BOOL_OUT)
    assert np.array_equal(b.astype(bool), want["prim_id"] != 0xFFFFFFFF)


def test_async_submit_collect(ctx, soup1k):
    v, scene, osc = soup1k
    scene.upload(ctx)
    rays = synth.incoherent_rays(5000, 8)
    want = osc.trace(rays)
    assert not ctx.has_pending()
    ctx.submit(rays)
    assert ctx.has_pending()
    with pytest.raises(capi.MrtError) as e:
        ctx.submit(rays)
    assert e.value.status == capi.ERR_PENDING
    with pytest.raises(capi.MrtError) as e:
        ctx.cast(rays)
    assert e.value.status == capi.ERR_PENDING
    got = ctx.collect()
    assert not ctx.has_pending()
    parity.assert_exact(got, want, "async")
    with pytest.raises(capi.MrtError) as e:
        ctx.collect(np.zeros(1, dtype=T.HIT32), 1)
    assert e.value.status == capi.ERR_NOT_PENDING
    # collect fewer than submitted (gpu_ray_caster.cpp:573)
    ctx.submit(rays, flags=capi.FLAG_COHERENT)
    part = ctx.collect(count=100)
    parity.assert_exact(part, want[:100], "async partial")
    # upload drains a pending dispatch (cpp:198-202)
    ctx.submit(rays)
    scene.upload(ctx)
    assert not ctx.has_pending()


# ---------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("count", [1, 63, 64, 65, 255, 256, 257, 1000])
def test_ragged_batch_sizes(ctx, soup1k, count):
    v, scene, osc = soup1k
    scene.upload(ctx)
    rays = synth.incoherent_rays(count, 100 + count)
    want = osc.trace(rays)
    parity.assert_exact(ctx.cast(rays, flags=capi.FLAG_COHERENT), want, f"n={count}")
    parity.assert_exact(ctx.cast(rays), want, f"n={count} sorted")


@pytest.mark.parametrize("cull", [2, 1])
def test_packet_frustum_culling_skips_no_hit(built, cull):
    """mrt_options.packet_cull = 2 (the default since round 3; 1 = the walk without it, kept and held to the same records):
    the 128-ray walk skips a child box that lies wholly outside the pyramid of the packet's
    rays.  Whatever it skips, the records must be the oracle's: pinhole grids (culling active: one apex per packet), clipped
    grids and partial waves (tiles without their corner lanes: culling off for that packet), rays with different origins
    flagged coherent, a camera inside the scene (boxes around the apex: the error bound must keep them), any-hit."""
    v = synth.soup(40000, 0.3, 21)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    c = capi.Context(0, kernel=capi.KERNEL_PACKET_DUAL, packet_cull=cull)
    scene.upload(c)
    for origin, fwd, (w, h), fov in (((0, 0, -12), (0, 0, 1), (256, 128), 50.0), ((0.3, -0.2, 0.1), (0.2, 0.9, -0.3), (128, 96), 75.0),
                                     ((0, 0, -12), (0, 0, 1), (130, 35), 50.0), ((9, 8, -7), (-1, -0.9, 0.8), (64, 64), 20.0)):
        cam = capi.camera_look(origin, fwd, w, h, fov)
        g = po.grid_rays(origin, fwd, w, h, fov)
        want = osc.trace(g)
        parity.assert_exact(c.cast_grid(cam, w, h), want, f"culling walk, grid {w}x{h} from {origin}")
        parity.assert_exact(c.cast(g, flags=capi.FLAG_COHERENT), want, f"culling walk, cast of {w}x{h} from {origin}")
        any_got = c.cast_grid(cam, w, h, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT).astype(bool)
        assert np.array_equal(any_got, want["prim_id"] >= 0)
    assert c.stats()["last_kernel"] == capi.KERNEL_PACKET_DUAL
    inc = synth.incoherent_rays(20000, 5)                         # no common apex: every packet must decline to cull
    parity.assert_exact(c.cast(inc, flags=capi.FLAG_COHERENT), osc.trace(inc), "culling walk, incoherent rays flagged coherent")
    c.close()


@pytest.mark.parametrize("kernel", [capi.KERNEL_AUTO, capi.KERNEL_PACKET_ASM, capi.KERNEL_PACKET_DUAL])
def test_frame_coherent_tile_schedule_changes_no_record(built, kernel, monkeypatch):
    """Grid casts of 2^19 .. 2^24 rays (here from 2^15: MRT_SCHEDULE_MIN_LOG2) launch their tiles longest first by what each cost
    in the previous cast of the same grid, the most expensive units in pieces -- quarter tiles in 16 lanes -- (mrt_options.
    tile_schedule, api.hip schedule_grid / schedule_plan_kernel; here the top 5 % instead of the top 1 %: MRT_SCHED_SPLIT_PCT):
    any launch order and any cut into pieces gives the same records.  The same grid sixteen times, another camera on the same grid
    (the old order is reused: still only a permutation), a row block, a clipped grid, any-hit, tokens, a batch whose width the device
    finds; with pieces, without, and with the schedule off.  (With MRT_KERNEL_AUTO grids of this size are not scheduled: every tile
    goes in quarter tiles, test_small_grids_in_quarter_tiles; the explicit kernels are what is scheduled here, and
    test_renderer_resolutions_on_the_c3_scene runs the tuner on grids of its own size.)"""
    monkeypatch.setenv("MRT_SCHEDULE_MIN_LOG2", "15")
    monkeypatch.setenv("MRT_SCHED_SPLIT_PCT", "5")
    v = synth.soup(20000, 0.25, 33)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    for sched in (0, 2, 1):
        c = capi.Context(0, kernel=kernel, tile_schedule=sched)
        scene.upload(c)
        for (w, h) in ((512, 256), (333, 201)):
            cam = capi.camera_look((0, 0, -12), (0, 0, 1), w, h, 50.0)
            rays = po.grid_rays((0, 0, -12), (0, 0, 1), w, h, 50.0)
            want = osc.trace(rays)
            for frame in range(16 if sched == 0 else 4):
                parity.assert_exact(c.cast_grid(cam, w, h), want, f"kernel {kernel} schedule {sched} {w}x{h} frame {frame}")
            cam2 = capi.camera_look((3, 1, -11), (-0.2, 0, 1), w, h, 40.0)
            parity.assert_exact(c.cast_grid(cam2, w, h), osc.trace(po.grid_rays((3, 1, -11), (-0.2, 0, 1), w, h, 40.0)), "another camera, the old order")
            for frame in range(3):
                parity.assert_exact(c.cast_grid(cam, w, h, y0=8, y1=h - 3), want[8 * w:(h - 3) * w], f"row block frame {frame}")
                b = c.cast_grid(cam, w, h, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT)
                assert np.array_equal(b.astype(bool), want["prim_id"] >= 0)
                tok = c.cast_grid(cam, w, h, flags=capi.FLAG_TOKEN_OUT)
                assert np.array_equal(tok != capi.TOKEN_MISS, want["prim_id"] >= 0)
            if w % 8 == 0:
                for frame in range(4):   # the width found on the device: scheduled from what the previous cast of as many rays found
                    parity.assert_exact(c.cast(rays, flags=capi.FLAG_COHERENT), want, f"mrt_cast(COHERENT) frame {frame}")
        c.close()


def test_interleaved_views_keep_their_schedules(built, monkeypatch):
    """What is learnt about a grid (tile schedule, pieces, how it is cast fastest) is kept per grid and cast mode for the last
    eight of them (api.hip select_grid_state): three views and two modes cast in turn, fourteen rounds -- every record of every
    cast against the oracle, whatever state each cast found."""
    monkeypatch.setenv("MRT_SCHEDULE_MIN_LOG2", "15")
    monkeypatch.setenv("MRT_SCHED_SPLIT_PCT", "5")
    v = synth.soup(20000, 0.25, 37)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    views = []
    for (w, h, origin, fwd) in ((640, 384, (0, 0, -12), (0, 0, 1)), (512, 256, (2, 1, -11), (-0.1, 0, 1)), (400, 300, (0, 0, -12), (0, 0.05, 1))):
        views.append((w, h, capi.camera_look(origin, fwd, w, h, 50.0), osc.trace(po.grid_rays(origin, fwd, w, h, 50.0))))
    for kernel in (capi.KERNEL_PACKET_ASM, capi.KERNEL_PACKET_DUAL):
        c = capi.Context(0, kernel=kernel)
        scene.upload(c)
        for rnd in range(14):
            for (w, h, cam, want) in views:
                parity.assert_exact(c.cast_grid(cam, w, h), want, f"kernel {kernel} round {rnd} {w}x{h}")
                b = c.cast_grid(cam, w, h, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT)
                assert np.array_equal(b.astype(bool), want["prim_id"] >= 0), (kernel, rnd, w, h)
        c.close()


def test_small_grids_in_quarter_tiles(built):
    """Grids of 64 rays up to 3 600 tiles are cast by the packet kernel with every tile launched in pieces -- up to 512 tiles as
    sixteen 2x2-pixel sixteenths (4 rays in lanes 0..3 of a wave), above as four 4x4-pixel quarters (16 rays; api.hip
    quarter_small_grid): a grid of fewer tiles than the device has wave slots lasts as long as its longest walk, and a walk for a
    few rays is short.  Records are the oracle's at every size: widths and heights that are no multiples of 8, 4 or 2 (clipped
    pieces, pieces wholly outside the grid), the fused grid cast, a row block, rays read from memory with a declared width
    (mrt_cast_tiled) and with the width found on the device (mrt_cast(COHERENT)), any-hit, tokens; below 64 rays and above
    3 600 tiles the other kernels."""
    v = synth.soup(20000, 0.25, 36)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    c = capi.Context(0)
    scene.upload(c)
    for (w, h) in ((7, 5), (16, 12), (32, 32), (64, 40), (100, 70), (101, 67), (130, 61), (181, 179), (256, 256), (321, 243), (640, 360), (648, 368)):
        cam = capi.camera_look((0, 0, -12), (0, 0.05, 1), w, h, 50.0)
        rays = po.grid_rays((0, 0, -12), (0, 0.05, 1), w, h, 50.0)
        want = osc.trace(rays)
        parity.assert_exact(c.cast_grid(cam, w, h), want, f"{w}x{h} cast_grid")
        tiles = ((w + 7) // 8) * ((h + 7) // 8)
        in_pieces = w * h >= 64 and tiles <= 3600
        assert c.last_kernel_variant().startswith("trace_packet_asm_kernel" if in_pieces else ("trace_lane_kernel" if w * h < 64 else "trace_packet_")), (w, h, c.last_kernel_variant())
        parity.assert_exact(c.cast_grid(cam, w, h, y0=1, y1=h - 2), want[1 * w:(h - 2) * w], f"{w}x{h} row block")
        b = c.cast_grid(cam, w, h, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT)
        assert np.array_equal(b.astype(bool), want["prim_id"] >= 0)
        tok = c.cast_grid(cam, w, h, flags=capi.FLAG_TOKEN_OUT)
        assert np.array_equal(tok != capi.TOKEN_MISS, want["prim_id"] >= 0)
        parity.assert_exact(c.cast(rays, flags=capi.FLAG_COHERENT), want, f"{w}x{h} mrt_cast(COHERENT)")
        d_rays, d_hits = c.device_alloc(rays.nbytes), c.device_alloc(rays.shape[0] * 32)
        c.h2d(d_rays, rays)
        c.cast_tiled(d_rays, d_hits, w, h)
        got = np.zeros(rays.shape[0], dtype=T.HIT32)
        c.d2h(got, d_hits)
        parity.assert_exact(got, want, f"{w}x{h} mrt_cast_tiled")
        c.device_free(d_rays); c.device_free(d_hits)
    many = synth.incoherent_rays(40000, 24)         # the lane kernel on waves of 1, 2, 4 rays (up to 2^15 rays) and on full ones
    want_many = osc.trace(many)
    for n in (1, 2, 63, 64, 65, 255, 1000, 8192, 8193, 16384, 16385, 32768, 32769, 40000):
        parity.assert_exact(c.cast(many[:n]), want_many[:n], f"{n} incoherent rays")
    inc = synth.incoherent_rays(256 * 64, 23)       # flagged coherent, but not: the device's verdict still routes it to the lane kernel
    parity.assert_exact(c.cast(inc, flags=capi.FLAG_COHERENT), osc.trace(inc), "incoherent rays flagged coherent")
    assert c.stats()["reserved"] == 1
    c.close()


@pytest.mark.parametrize("wh", [(1280, 960), (1920, 1080)])
def test_renderer_resolutions_on_the_c3_scene(built, wh):
    """The reference's own workload size (1280x960: ROADMAP.md:175-181) and 1080p on the 1 M-triangle C3 scene: fifteen frames of
    the same grid -- the twelve measuring frames of the kernel tuner (the 64-ray kernel, the 128-ray walk with its most expensive
    units in pieces, the 128-ray walk whole), the frames launched in a measured tile order, what the library settles on -- every
    record of every frame against the oracle."""
    w, h = wh
    cfg = synth.CONFIGS["C3"]
    verts = synth.scene_vertices(cfg)
    c = capi.Context(0)
    capi.Scene(verts).upload(c)
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    want = po.OracleScene(verts).trace(po.grid_rays(cfg["origin"], cfg["forward"], w, h, cfg["fov"]), n_threads=16)
    d_hits = c.device_alloc(w * h * 32)
    seen = set()
    got = np.zeros(w * h, dtype=T.HIT32)
    for frame in range(15):
        c.cast_grid(cam, w, h, hits=d_hits, flags=capi.FLAG_HITS_ON_DEVICE)
        seen.add(c.last_kernel_variant())
        c.d2h(got, d_hits)
        parity.assert_exact(got, want, f"{w}x{h} frame {frame} ({c.last_kernel_variant()})")
    assert len(seen) == 2, seen       # both packet kernels were measured on this grid
    if wh == (1920, 1080):
        # the same rays read from memory, the width found on the device (the reference's cast_rays contract): from the second cast
        # on the batch is scheduled, and tuned, by what the previous cast of as many rays found
        d_rays = c.device_alloc(w * h * 32)
        c.generate_grid(cam, w, h, 0, h, d_rays)
        seen = set()
        for frame in range(15):
            c.cast(d_rays, d_hits, count=w * h, flags=capi.FLAG_COHERENT | capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE)
            seen.add(c.last_kernel_variant())
            c.d2h(got, d_hits)
            parity.assert_exact(got, want, f"{w}x{h} mrt_cast(COHERENT) frame {frame} ({c.last_kernel_variant()})")
        assert len(seen) == 2, seen
        c.device_free(d_rays)
    c.device_free(d_hits)
    c.close()


@pytest.mark.parametrize("kernel", [capi.KERNEL_PACKET, capi.KERNEL_PACKET_ASM, capi.KERNEL_PACKET_ROWS, capi.KERNEL_PACKET_DUAL, QUAD])
def test_partial_waves_and_clipped_tiles_on_every_packet_kernel(built, kernel):
    """Packets whose wave is not full: COHERENT batches of 1 .. 1000 rays (the last wave partial; at count = 1 lanes
    1..63 have no ray) and grids whose right and bottom tiles are clipped, on the flat and the two-level packet walks.
    Pins the abort of round 1 (gpurun_out/test6.log: a 4-wide packet kernel, since retired, selected child refs with
    v_readlane from lanes 1..3, which had EXITED for want of a ray and held stale registers: wild node index, memory
    fault, SIGABRT inside mrt_cast at count = 1).  No surviving kernel reads another lane's registers except through
    v_readfirstlane (first ACTIVE lane) and ballots; this test keeps it that way."""
    v = synth.soup(3000, 0.4, 8)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    c = capi.Context(0, kernel=kernel)
    scene.upload(c)
    rays = po.grid_rays((0, 0, -12), (0, 0, 1), 40, 25, 50.0)
    want = osc.trace(rays)
    for count in (1, 2, 63, 65, 129, 1000):
        parity.assert_exact(c.cast(rays[:count], flags=capi.FLAG_COHERENT), want[:count], f"kernel {kernel}, {count} coherent rays")
    for (w, h) in ((13, 5), (8, 8), (9, 17), (130, 3)):   # tiles clipped on the right and at the bottom; an odd number of tiles
        cam = capi.camera_look((0, 0, -12), (0, 0, 1), w, h, 50.0)
        g = po.grid_rays((0, 0, -12), (0, 0, 1), w, h, 50.0)
        parity.assert_exact(c.cast_grid(cam, w, h), osc.trace(g), f"kernel {kernel}, {w}x{h} grid")
        parity.assert_exact(c.cast_grid(cam, w, h, y0=1, y1=h - 1), osc.trace(g[w:(h - 1) * w]), f"kernel {kernel}, {w}x{h} rows 1..h-1")
        any_got = c.cast_grid(cam, w, h, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT).astype(bool)
        assert np.array_equal(any_got, osc.trace(g)["prim_id"] >= 0)
    # the two-level packet walk (its BLAS walks are the single-packet asm loop)
    local, inst = synth.multi_mesh_instances(4, 600, 0.3, 13)
    c.upload_two_level_scene(local, inst)
    tl = po.OracleTwoLevelScene(local, inst)
    for (w, h) in ((13, 5), (9, 17)):
        cam = capi.camera_look((0, 0, -12), (0, 0, 1), w, h, 50.0)
        g = po.grid_rays((0, 0, -12), (0, 0, 1), w, h, 50.0)
        got, wt = c.cast_grid(cam, w, h), tl.trace(g)
        assert np.array_equal(got["prim_id"], wt["prim_id"]) and np.array_equal(got["t"], wt["t"])
    wt = tl.trace(rays[:65])
    got = c.cast(rays[:65], flags=capi.FLAG_COHERENT)
    assert np.array_equal(got["prim_id"], wt["prim_id"]) and np.array_equal(got["t"], wt["t"])
    c.close()


def test_empty_and_errors(built):
    c = capi.Context(0)
    rays = synth.incoherent_rays(10, 1)
    with pytest.raises(capi.MrtError) as e:
        c.cast(rays)
    assert e.value.status == capi.ERR_NO_SCENE and not c.is_available()
    scene = capi.Scene(synth.soup(50, 0.5, 2))
    scene.upload(c)
    # count == 0 is a silent no-op (gpu_ray_caster.cpp:419)
    assert c.L.mrt_cast(c.h, None, None, 0, 0xFFFFFFFF, 0, 0) == capi.MRT_OK
    # broken BVHs are rejected on the host, never reach a kernel
    bad = scene.nodes.copy()
    bad[0]["left_first"] = scene.used_nodes + 5
    with pytest.raises(capi.MrtError) as e:
        c.upload_scene(scene.tris, bad, scene.prim_idx)
    assert e.value.status == capi.ERR_BAD_BVH
    bad = scene.nodes.copy()
    bad[2]["left_first"] = 0  # cycle back to the root... or a leaf range out of bounds
    bad[2]["tri_count"] = 0
    with pytest.raises(capi.MrtError):
        c.upload_scene(scene.tris, bad, scene.prim_idx)
    badp = scene.prim_idx.copy()
    badp[3] = 10 ** 6
    with pytest.raises(capi.MrtError) as e:
        c.upload_scene(scene.tris, scene.nodes, badp)
    assert e.value.status == capi.ERR_BAD_BVH
    assert c.is_available()  # a failed upload keeps the previous scene
    with pytest.raises(capi.MrtError) as e:
        c.cast(rays, mode=7)
    assert e.value.status == capi.ERR_INVALID
    c.close()


@pytest.mark.parametrize("n_tris", [1, 2, 3, 5])
def test_tiny_scenes_root_leaf(ctx, n_tris):
    """Root-is-a-leaf scenes (gpu_ray_caster.cpp:255-271 wraps them)."""
    v = synth.soup(n_tris, 3.0, 40 + n_tris)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    scene.upload(ctx)
    rays = np.concatenate([po.grid_rays((0, 0, -12), (0, 0, 1), 64, 64, 50.0), synth.incoherent_rays(3000, n_tris)])
    want = osc.brute(rays)
    parity.assert_exact(ctx.cast(rays, flags=capi.FLAG_COHERENT), want, f"{n_tris} tris")
    assert (want["prim_id"] >= 0).any()


def test_degenerate_and_axis_aligned_rays(ctx, soup1k):
    v, scene, osc = soup1k
    scene.upload(ctx)
    rays = synth.incoherent_rays(4000, 77)
    rays["direction"][:1000] = [0, 0, 1]          # zero components -> safe_inv (+-1e9)
    rays["direction"][1000:1500] = [0, -1, 0]
    rays["direction"][1500:2000] = [-1, 0, 0]
    rays["t_min"][2000:2500] = 4.0                 # t_min >= t_max: miss, t = t_max
    rays["t_max"][2000:2500] = 4.0
    rays["t_max"][2500:3000] = 2.5                 # bounded rays
    rays["origin"][3000:3500] = 0.0                # rays starting inside the soup
    want = osc.trace(rays)
    assert want.tobytes() == osc.brute(rays).tobytes()
    got = ctx.cast(rays, flags=capi.FLAG_COHERENT)
    parity.assert_exact(got, want, "special rays")
    assert (got["prim_id"][2000:2500] == -1).all() and (got["t"][2000:2500] == 4.0).all()
    miss = got["prim_id"] < 0
    assert np.array_equal(got["t"][miss], rays["t_max"][miss])  # miss record: t = t_max, normal 0
    assert not got["normal"][miss].any() and not got["hit_layers"][miss].any()


def test_query_mask_filters_during_traversal(ctx):
    v = synth.soup(2000, 0.6, 21)
    layers = (1 << (np.arange(2000) % 5)).astype(np.uint32)
    ids = (np.arange(2000, dtype=np.uint32) * 3 + 7)   # ids need not be indices
    scene, osc = capi.Scene(v, ids, layers), po.OracleScene(v, ids, layers)
    scene.upload(ctx)
    rays = po.grid_rays((0, 0, -12), (0, 0, 1), 96, 96, 50.0)
    for mask in (0x1, 0x6, 0x1F, 0x10, 0x0, 0xFFFFFFE0):
        want = osc.trace(rays, query_mask=mask)
        got = ctx.cast(rays, query_mask=mask, flags=capi.FLAG_COHERENT)
        parity.assert_exact(got, want, f"mask {mask:#x}")
        hit = got["prim_id"] >= 0
        assert ((got["hit_layers"][hit] & mask) != 0).all()
        b = ctx.cast(rays, query_mask=mask, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_COHERENT | capi.FLAG_BOOL_OUT)
        assert np.array_equal(b.astype(bool), hit)


def test_morton_keys_on_device(ctx):
    rays = synth.incoherent_rays(100000, 31)
    d_rays = DeviceArray(ctx, rays.nbytes).upload(rays)
    d_keys = DeviceArray(ctx, 4 * rays.shape[0])
    ctx.morton_keys(d_rays.ptr, rays.shape[0], d_keys.ptr)
    assert np.array_equal(d_keys.download(np.uint32, rays.shape[0]), po.morton_keys(rays))
    d_rays.free(); d_keys.free()


def test_counting_variant_matches_oracle_counters(built):
    c = capi.Context(0, kernel=capi.KERNEL_LANE, count_visits=True)
    v = synth.soup(20000, 0.2, 3)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    scene.upload(c)
    rays = po.grid_rays((0, 0, -12), (0, 0, 1), 256, 256, 50.0)
    want, ctr = osc.trace(rays, counters=True)
    got = c.cast(rays, flags=capi.FLAG_COHERENT)
    parity.assert_exact(got, want, "counting variant")
    s = c.stats()
    assert s["rays_cast"] == rays.shape[0] and s["hits"] == ctr["hits"]
    # the kernel keeps no entry distance on its stack, so it may fetch a few more nodes
    # than the reference shader's walk; it must never fetch fewer
    assert ctr["node_visits"] <= s["bvh_nodes_visited"] <= 1.35 * ctr["node_visits"]
    assert ctr["tri_tests"] <= s["tri_tests"] <= 1.5 * ctr["tri_tests"]
    assert s["max_stack_depth"] <= c.scene_info()["stack_need"]
    c.close()
    # packet kernel: a wave visits the union of its lanes' nodes, charged to every lane
    c = capi.Context(0, kernel=capi.KERNEL_PACKET, count_visits=True)
    scene.upload(c)
    parity.assert_exact(c.cast(rays, flags=capi.FLAG_COHERENT), want, "counting packet variant")
    s = c.stats()
    assert s["hits"] == ctr["hits"] and s["bvh_nodes_visited"] >= ctr["node_visits"] and s["tri_tests"] >= ctr["tri_tests"]
    assert s["dead_pops"] < s["bvh_nodes_visited"]
    c.close()
    # the hand-written packet walks count too: rows fetched per PACKET (what bench.py's roofline prices)
    n_packets = rays.shape[0] // 64
    for kern in (capi.KERNEL_PACKET_ASM, capi.KERNEL_PACKET_ROWS, capi.KERNEL_PACKET_DUAL):
        c = capi.Context(0, kernel=kern, count_visits=True)
        scene.upload(c)
        parity.assert_exact(c.cast(rays, flags=capi.FLAG_COHERENT), want, f"counting packet kernel {kern}")
        s = c.stats()
        assert s["last_kernel"] == kern and s["rays_cast"] == rays.shape[0] and s["hits"] == ctr["hits"]
        # a packet needs at least what its hungriest ray needs, and at most what its 64 (128) rays need together
        assert ctr["node_visits"] / rays.shape[0] <= s["wave_node_fetches"] / n_packets * (2 if kern == capi.KERNEL_PACKET_DUAL else 1)
        assert 0 < s["wave_node_fetches"] <= ctr["node_visits"] and 0 < s["wave_tri_fetches"] <= ctr["tri_tests"] * 64
        if kern != capi.KERNEL_PACKET_ASM:   # the row walks record their stack's high-water mark: never above what the BVH can need
            assert 0 < s["max_stack_depth"] <= c.scene_info()["stack_need"], (s["max_stack_depth"], c.scene_info())
        c.close()
    # the four-wide packet walk (builds with MRT_WITH_QUAD=1): fewer rows than the two-wide walk fetches, the same hits
    if not capi.kernel_available(capi.KERNEL_PACKET_QUAD):
        return
    c = capi.Context(0, kernel=capi.KERNEL_PACKET_QUAD, count_visits=True)
    scene.upload(c)
    parity.assert_exact(c.cast(rays, flags=capi.FLAG_COHERENT), want, "counting four-wide packet kernel")
    s = c.stats()
    assert s["last_kernel"] == capi.KERNEL_PACKET_QUAD and s["rays_cast"] == rays.shape[0] and s["hits"] == ctr["hits"]
    assert 0 < s["wave_node_fetches"] <= ctr["node_visits"] and 0 < s["wave_tri_fetches"] <= ctr["tri_tests"] * 64
    assert 0 < s["max_stack_depth"] <= 64
    c.close()
    # ... and the persistent lane kernels: node steps = cache lines fetched, per ray
    inc = synth.incoherent_rays(100000, 31)
    want_i, ctr_i = osc.trace(inc, counters=True)
    lines = {}
    for kern in (capi.KERNEL_LANE_PERSISTENT, capi.KERNEL_LANE4_PERSISTENT, capi.KERNEL_LANE8_PERSISTENT):
        c = capi.Context(0, kernel=kern, count_visits=True)
        scene.upload(c)
        parity.assert_exact(c.cast(inc), want_i, f"counting persistent kernel {kern}")
        s = c.stats()
        assert s["last_kernel"] == kern and s["hits"] == ctr_i["hits"] and s["tri_tests"] >= ctr_i["tri_tests"]
        assert s["wave_node_fetches"] == s["bvh_nodes_visited"] > 0
        lines[kern] = s["bvh_nodes_visited"]
        assert (s["leaf_box_checks"] > 0) == (kern == capi.KERNEL_LANE8_PERSISTENT)
        c.close()
    assert ctr_i["node_visits"] <= lines[capi.KERNEL_LANE_PERSISTENT] <= 1.35 * ctr_i["node_visits"]
    assert lines[capi.KERNEL_LANE8_PERSISTENT] < lines[capi.KERNEL_LANE4_PERSISTENT] < lines[capi.KERNEL_LANE_PERSISTENT]  # wider nodes, fewer lines


@pytest.mark.parametrize("kernel", [capi.KERNEL_LANE_PERSISTENT, capi.KERNEL_LANE4_PERSISTENT,
                                    capi.KERNEL_LANE8_PERSISTENT])
@pytest.mark.parametrize("lds_depth,refill,leaf_wait", [(4, 16, 0), (8, 1, 1), (16, 64, 64), (64, 16, 3)])
def test_persistent_lane_kernel_spill_and_refill(built, lds_depth, refill, leaf_wait, kernel):
    """Resident waves pulling rays from a counter; stack entries beyond `lds_depth` spill to HBM.
    A 4-entry LDS stack forces the spill path on almost every ray.  Both node widths; node / leaf
    phase switching from "at the first leaf" to "classic while-while"."""
    c = capi.Context(0, kernel=kernel, stack_override=lds_depth, refill=refill, leaf_wait=leaf_wait)
    v = synth.soup(20000, 0.25, 33)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    scene.upload(c)
    assert c.scene_info()["stack_need"] > 8
    for n in (1, 100, 70001, 300000):
        rays = synth.incoherent_rays(n, 41 + n)
        rays["t_max"][::7] = 2.0
        want = osc.trace(rays)
        parity.assert_exact(c.cast(rays, flags=capi.FLAG_COHERENT), want, f"persistent n={n} unsorted")
        parity.assert_exact(c.cast(rays), want, f"persistent n={n} sorted")
        b = c.cast(rays, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT)
        assert np.array_equal(b.astype(bool), want["prim_id"] >= 0)
    c.close()


def test_row_width_detection_for_coherent_batches(built):
    """mrt_cast(COHERENT) gets no image width (the reference's cast_rays has none): the
    device looks for it.  Whatever it finds, the results are the oracle's."""
    c = capi.Context(0, count_visits=True)
    c_auto = capi.Context(0)
    v = synth.soup(5000, 0.3, 29)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    scene.upload(c)
    scene.upload(c_auto)
    cases = []
    for (w, h, expect) in ((128, 64, 128), (256, 40, 256), (64, 64, 64), (100, 64, 0), (128, 60, 0), (24, 200, 24)):
        cases.append((po.grid_rays((0, 0, -12), (0, 0, 1), w, h, 50.0), expect, f"{w}x{h}"))
    a = po.grid_rays((0, 0, -12), (0, 0, 1), 128, 32, 50.0)
    b = po.grid_rays((0, 0, -12), (0, 0, 1), 64, 64, 50.0)
    cases.append((np.concatenate([a, b]), None, "two grids glued"))   # width changes half way: any answer is fine
    cases.append((synth.incoherent_rays(8192, 3), 0, "incoherent rays declared coherent"))
    cases.append((synth.incoherent_rays(100000, 5), 0, "incoherent rays declared coherent, large batch"))
    same = po.grid_rays((0, 0, -12), (0, 0, 1), 64, 64, 50.0)
    same["direction"][:] = same["direction"][0]
    cases.append((same, 0, "identical directions"))
    for rays, expect, name in cases:
        want = osc.trace(rays)
        got = c.cast(rays, flags=capi.FLAG_COHERENT)
        parity.assert_exact(got, want, name)
        s = c.stats()
        assert s["last_kernel_launches"] == 2
        if expect is not None:
            assert s["detected_grid_w"] == expect, (name, s["detected_grid_w"])
        # the device's own verdict: random rays are sent to the lane kernel, and so is a small batch (< 2^15 rays) in which no
        # row width was found -- packets of 64 consecutive rays are no match for one lane per ray there
        if expect is not None:
            to_lane = name.startswith("incoherent") or (expect == 0 and rays.shape[0] < 32768)
            assert s["reserved"] == (1 if to_lane else 0), (name, s["reserved"])
        parity.assert_exact(c_auto.cast(rays, flags=capi.FLAG_COHERENT), want, name + " (auto: packet or lane launch)")
        host = po.make_host_rays(rays)
        got44 = c.cast(host, flags=capi.FLAG_COHERENT | capi.FLAG_HOST_LAYOUT)
        assert got44.tobytes() == po.unpack_hits(want, host).tobytes()
    c.close()


@pytest.mark.parametrize("kernel", [capi.KERNEL_LANE, capi.KERNEL_PACKET,
                                    capi.KERNEL_PACKET_ASM, capi.KERNEL_PACKET_DUAL, capi.KERNEL_PACKET_ROWS, QUAD, capi.KERNEL_LANE_PERSISTENT, capi.KERNEL_LANE4_PERSISTENT,
                                    capi.KERNEL_LANE8_PERSISTENT])
def test_both_kernels_on_every_kind_of_batch(built, kernel):
    """Either kernel must give the oracle's answer for any batch, coherent or not:
    the kernel choice (MRT_KERNEL_AUTO) is a speed decision only."""
    c = capi.Context(0, kernel=kernel)
    v = synth.soup(3000, 0.35, 17)
    layers = (1 << (np.arange(3000) % 3)).astype(np.uint32)
    scene, osc = capi.Scene(v, None, layers), po.OracleScene(v, None, layers)
    scene.upload(c)
    grid = po.grid_rays((0, 0, -12), (0, 0, 1), 100, 70, 50.0)          # ragged tiles: 100 x 70
    inc = synth.incoherent_rays(5000, 23)
    inc["t_min"][:300] = 3.0
    inc["t_max"][:300] = 3.0                                             # degenerate
    inc["direction"][300:600] = [1, 0, 0]                                # axis aligned
    inc["t_max"][600:900] = 1.5
    mixed = np.concatenate([grid[:777], inc[:1000]])                     # mixed octants inside waves
    for rays, name in ((grid, "grid"), (inc, "incoherent"), (mixed, "mixed")):
        for mask in (0xFFFFFFFF, 0x5):
            want = osc.trace(rays, query_mask=mask)
            for flags in (capi.FLAG_COHERENT, 0):
                parity.assert_exact(c.cast(rays, query_mask=mask, flags=flags), want, f"{name} mask={mask:#x} flags={flags}")
            b = c.cast(rays, query_mask=mask, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_COHERENT | capi.FLAG_BOOL_OUT)
            assert np.array_equal(b.astype(bool), want["prim_id"] >= 0)
    cam = capi.camera_look((0, 0, -12), (0, 0, 1), 100, 70, 50.0)
    want = osc.trace(grid)
    parity.assert_exact(c.cast_grid(cam, 100, 70), want, "cast_grid ragged tiles")
    parity.assert_exact(c.cast_grid(cam, 100, 70, y0=13, y1=41), want[13 * 100:41 * 100], "cast_grid row window")
    for k in (1, 2, 4, 5, 6):
        ck = capi.Context(0, kernel=kernel, tile_w_log2=k)
        scene.upload(ck)
        parity.assert_exact(ck.cast_grid(cam, 100, 70), want, f"tile 2^{k}")
        ck.close()
    c.close()


def test_async_flag_queues_casts_back_to_back(ctx, soup1k):
    """MRT_FLAG_ASYNC: casts only queue work on the context's stream; a frame loop issues several
    (coherent, sorted, grid; records and tokens) and waits once.  They share the context's scratch
    buffers, so stream order alone must keep them apart."""
    v, scene, osc = soup1k
    scene.upload(ctx)
    w, h = 96, 80
    cam = capi.camera_look((0, 0, -12), (0, 0, 1), w, h, 50.0)
    grid = po.grid_rays((0, 0, -12), (0, 0, 1), w, h, 50.0)
    inc = synth.incoherent_rays(70000, 5)      # large enough for the persistent lane kernel
    want_g, want_i = osc.trace(grid), osc.trace(inc)
    dev = capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE | capi.FLAG_ASYNC
    dg, di = DeviceArray(ctx, grid.nbytes).upload(grid), DeviceArray(ctx, inc.nbytes).upload(inc)
    outs = [DeviceArray(ctx, max(grid.shape[0], inc.shape[0]) * 32) for _ in range(5)]
    ctx.cast(dg.ptr, outs[0].ptr, count=grid.shape[0], flags=dev | capi.FLAG_COHERENT)
    ctx.cast(di.ptr, outs[1].ptr, count=inc.shape[0], flags=dev)                       # Morton sort + lane kernel
    ctx.cast(di.ptr, outs[2].ptr, count=inc.shape[0], flags=dev | capi.FLAG_COHERENT)  # judged incoherent on the device
    ctx.cast_grid(cam, w, h, hits=outs[3].ptr, flags=capi.FLAG_HITS_ON_DEVICE | capi.FLAG_ASYNC)
    ctx.cast(dg.ptr, outs[4].ptr, count=grid.shape[0], flags=dev | capi.FLAG_COHERENT | capi.FLAG_TOKEN_OUT)
    ctx.synchronize()
    parity.assert_exact(outs[0].download(T.HIT32, grid.shape[0]), want_g, "async coherent")
    parity.assert_exact(outs[1].download(T.HIT32, inc.shape[0]), want_i, "async sorted")
    parity.assert_exact(outs[2].download(T.HIT32, inc.shape[0]), want_i, "async mislabelled coherent")
    parity.assert_exact(outs[3].download(T.HIT32, grid.shape[0]), want_g, "async cast_grid")
    tok = outs[4].download(np.uint32, grid.shape[0])
    assert np.array_equal(tok != capi.TOKEN_MISS, want_g["prim_id"] >= 0)
    with pytest.raises(capi.MrtError):   # host arrays cannot be left in flight
        ctx.cast(grid, flags=capi.FLAG_ASYNC)
    for d in [dg, di] + outs:
        d.free()


@pytest.mark.parametrize("kernel", [capi.KERNEL_AUTO, capi.KERNEL_LANE, capi.KERNEL_PACKET,
                                    capi.KERNEL_PACKET_ASM, capi.KERNEL_PACKET_DUAL, capi.KERNEL_PACKET_ROWS, QUAD, capi.KERNEL_LANE_PERSISTENT, capi.KERNEL_LANE4_PERSISTENT,
                                    capi.KERNEL_LANE8_PERSISTENT])
def test_hit_tokens_expand_to_identical_records(built, kernel):
    """MRT_FLAG_TOKEN_OUT + mrt_expand_tokens == the records of a plain cast, byte for byte
    (what lets the multi-GPU gather move 4 bytes per ray instead of 32)."""
    c = capi.Context(0, kernel=kernel)
    other = capi.Context(0)          # stands for another rank: same scene uploaded independently
    n_tris = 3000
    v = synth.soup(n_tris, 0.35, 17)
    layers = (1 << (np.arange(n_tris) % 3)).astype(np.uint32)
    scene, osc = capi.Scene(v, None, layers), po.OracleScene(v, None, layers)
    scene.upload(c)
    capi.Scene(v, None, layers).upload(other)
    grid = po.grid_rays((0, 0, -12), (0, 0, 1), 100, 70, 50.0)
    inc = synth.incoherent_rays(5000, 23)
    inc["t_min"][:300] = 3.0
    inc["t_max"][:300] = 3.0                                             # degenerate: misses with t = t_max
    inc["t_max"][600:900] = 1.5
    for rays, name in ((grid, "grid"), (inc, "incoherent")):
        n = rays.shape[0]
        d_rays = DeviceArray(c, rays.nbytes).upload(rays)
        d_tok, d_hits = DeviceArray(c, n * 4), DeviceArray(c, n * 32)
        o_rays = DeviceArray(other, rays.nbytes).upload(rays)
        o_tok, o_hits = DeviceArray(other, n * 4), DeviceArray(other, n * 32)
        for mask in (0xFFFFFFFF, 0x5):
            want = osc.trace(rays, query_mask=mask)
            for flags in (capi.FLAG_COHERENT, 0):
                tok = c.cast(rays, query_mask=mask, flags=flags | capi.FLAG_TOKEN_OUT)
                assert tok.dtype == np.uint32 and np.array_equal(tok != capi.TOKEN_MISS, want["prim_id"] >= 0)
                assert (tok[tok != capi.TOKEN_MISS] < n_tris).all()
                d_tok.upload(tok)
                c.expand_tokens(d_rays.ptr, d_tok.ptr, d_hits.ptr, n)
                c.synchronize()
                parity.assert_exact(d_hits.download(T.HIT32, n), want, f"{name} tokens mask={mask:#x} flags={flags}")
                # the tokens mean the same on a device that uploaded the same scene by itself
                o_tok.upload(tok)
                other.expand_tokens(o_rays.ptr, o_tok.ptr, o_hits.ptr, n)
                other.synchronize()
                assert o_hits.download(T.HIT32, n).tobytes() == want.tobytes()
            # any-hit tokens name SOME intersected triangle: the rebuilt record is a real hit of that triangle
            tok = c.cast(rays, query_mask=mask, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_COHERENT | capi.FLAG_TOKEN_OUT)
            assert np.array_equal(tok != capi.TOKEN_MISS, want["prim_id"] >= 0)
            d_tok.upload(tok)
            c.expand_tokens(d_rays.ptr, d_tok.ptr, d_hits.ptr, n)
            c.synchronize()
            rec = d_hits.download(T.HIT32, n)
            hit = want["prim_id"] >= 0
            assert (rec["t"][hit] >= want["t"][hit]).all() and (rec["prim_id"][~hit] == -1).all()
            same = hit & (rec["prim_id"] == want["prim_id"])
            assert rec[same].tobytes() == want[same].tobytes()
        # tokens that are not from this scene never make the kernel read out of bounds: they read as misses
        bad = np.full(n, n_tris + 12345, dtype=np.uint32)
        d_tok.upload(bad)
        c.expand_tokens(d_rays.ptr, d_tok.ptr, d_hits.ptr, n)
        c.synchronize()
        assert (d_hits.download(T.HIT32, n)["prim_id"] == -1).all()
        # 60-byte rays / 44-byte records
        host = po.make_host_rays(rays)
        want44 = c.cast(host, flags=capi.FLAG_HOST_LAYOUT)
        tok = c.cast(host, flags=capi.FLAG_HOST_LAYOUT | capi.FLAG_TOKEN_OUT)
        h_rays = DeviceArray(c, host.nbytes).upload(host)
        h_hits = DeviceArray(c, n * 44)
        d_tok.upload(tok)
        c.expand_tokens(h_rays.ptr, d_tok.ptr, h_hits.ptr, n, flags=capi.FLAG_HOST_LAYOUT)
        c.synchronize()
        assert h_hits.download(T.HOST_HIT44, n).tobytes() == want44.tobytes()
        for d in (d_rays, d_tok, d_hits, o_rays, o_tok, o_hits, h_rays, h_hits):
            d.free()
    # camera grids: the expanding side regenerates the rays from the camera (rows of a larger grid too)
    cam = capi.camera_look((0, 0, -12), (0, 0, 1), 100, 70, 50.0)
    want = c.cast_grid(cam, 100, 70)
    parity.assert_exact(want, osc.trace(grid), "cast_grid")
    for (y0, y1) in ((0, 70), (13, 41)):
        m = 100 * (y1 - y0)
        tok = c.cast_grid(cam, 100, 70, y0=y0, y1=y1, flags=capi.FLAG_TOKEN_OUT)
        o_tok, o_hits = DeviceArray(other, m * 4).upload(tok), DeviceArray(other, m * 32)
        other.expand_grid_tokens(cam, 100, 70, y0, y1, o_tok.ptr, o_hits.ptr)
        other.synchronize()
        assert o_hits.download(T.HIT32, m).tobytes() == want[y0 * 100:y1 * 100].tobytes()
        o_tok.free(); o_hits.free()
    with pytest.raises(capi.MrtError):
        c.cast(grid, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT | capi.FLAG_TOKEN_OUT)
    c.close(); other.close()


# ---------------------------------------------------------------------------
# BASELINE.json configs at full size
# ---------------------------------------------------------------------------
def _full_grid_case(ctx, name, oracle_rows):
    cfg = synth.CONFIGS[name]
    w, h = cfg["grid"]
    verts = synth.scene_vertices(cfg)
    scene = capi.Scene(verts)
    scene.upload(ctx)
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    got = ctx.cast_grid(cam, w, h)
    # 0. EVERY ray of the grid against the oracle: hit count, prim-id and t hashes over all w*h records (bit-exact)
    _assert_digest(digests.digest_records(got), _full_digest(name), f"{name} whole grid")
    g = _golden(f"{name.lower()}_sampled.npz")
    idx = g["index"]
    osc = po.OracleScene(verts)
    assert osc.nodes.tobytes() == scene.nodes.tobytes()
    # 1. the reference's result on the sampled rays
    rays_s = _sample_grid_rays(cfg, idx)
    st = parity.assert_reference_parity(got["prim_id"][idx], got["t"][idx], g["prim_id"], g["t"], rays_s, osc.tris, name)
    dg = json.loads(str(g["digest"]))
    hits = int((got["prim_id"] >= 0).sum())
    assert abs(hits - dg["hit_count"]) <= max(4, int(parity.MISMATCH_FRACTION * w * h)), (hits, dg["hit_count"])
    # 2. the oracle on a band of full rows: bit-exact
    y0, y1 = oracle_rows
    rays_b = po.grid_rays(cfg["origin"], cfg["forward"], w, h, cfg["fov"], y0, y1)
    parity.assert_exact(got[y0 * w:y1 * w], osc.trace(rays_b), f"{name} rows {y0}..{y1}")
    # 3. properties: row-major lanes == 8x8 tiled lanes; device-resident rays == fused raygen;
    #    row-block shards == whole grid (what each rank of a multi-GPU run computes)
    d_rays = DeviceArray(ctx, w * h * 32)
    d_hits = DeviceArray(ctx, w * h * 32)
    ctx.generate_grid(cam, w, h, 0, h, d_rays.ptr)
    ctx.cast(d_rays.ptr, d_hits.ptr, count=w * h, flags=capi.FLAG_COHERENT | capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE)
    assert d_hits.download(T.HIT32, w * h).tobytes() == got.tobytes(), "linear vs tiled lane mapping"
    ctx.cast_tiled(d_rays.ptr, d_hits.ptr, w, h)
    assert d_hits.download(T.HIT32, w * h).tobytes() == got.tobytes(), "cast_tiled vs cast_grid"
    # 4-byte hit tokens + expansion == the records (the multi-GPU exchange format)
    d_tok = DeviceArray(ctx, w * h * 4)
    ctx.cast(d_rays.ptr, d_tok.ptr, count=w * h, flags=capi.FLAG_COHERENT | capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE | capi.FLAG_TOKEN_OUT)
    ctx.expand_grid_tokens(cam, w, h, 0, h, d_tok.ptr, d_hits.ptr)
    ctx.synchronize()
    assert d_hits.download(T.HIT32, w * h).tobytes() == got.tobytes(), "tokens + expansion vs records"
    d_tok.free()
    d_rays.free(); d_hits.free()
    shards = [ctx.cast_grid(cam, w, h, y0=r * h // 4, y1=(r + 1) * h // 4) for r in range(4)]
    assert np.concatenate(shards).tobytes() == got.tobytes(), "row-sharded result differs"
    return st


def test_c2_full(ctx):
    """Config C2: 100 k-triangle soup, 1024^2 primary rays."""
    _full_grid_case(ctx, "C2", (448, 576))
    # the same batch as HOST arrays (the reference's cast_rays contract): must equal the device-resident cast
    cfg = synth.CONFIGS["C2"]
    w, h = cfg["grid"]
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    want = ctx.cast_grid(cam, w, h)
    rays = po.grid_rays(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    got = ctx.cast(rays, flags=capi.FLAG_COHERENT)
    assert got.tobytes() == want.tobytes()
    host = po.make_host_rays(rays)
    got44 = ctx.cast(host, flags=capi.FLAG_COHERENT | capi.FLAG_HOST_LAYOUT)
    assert got44.tobytes() == po.unpack_hits(want, host).tobytes()


def test_host_arrays_pipelined_in_chunks(ctx):
    """Host arrays of >= 2^21 rays (the reference's cast_rays contract at scale) go through the
    upload / trace / download pipeline in 2^20-ray chunks with a ragged tail: the records must equal
    those of one device-resident cast, for packed and 60/44-byte layouts, coherent, sorted and any-hit."""
    cfg = synth.CONFIGS["C2"]
    verts = synth.scene_vertices(cfg)
    scene = capi.Scene(verts)
    scene.upload(ctx)
    w, h = 2048, 1100                                   # 2 252 800 rays: two full chunks and a part of one
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    want = ctx.cast_grid(cam, w, h)
    rays = po.grid_rays(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    n = rays.shape[0]
    assert n >= 2 * (1 << 20) and n % (1 << 20) != 0
    got = ctx.cast(rays, flags=capi.FLAG_COHERENT)
    assert got.tobytes() == want.tobytes()
    assert ctx.cast(rays).tobytes() == want.tobytes()    # every chunk Morton-sorted on the device
    host = po.make_host_rays(rays)
    got44 = ctx.cast(host, flags=capi.FLAG_COHERENT | capi.FLAG_HOST_LAYOUT)
    assert got44.tobytes() == po.unpack_hits(want, host).tobytes()
    b = ctx.cast(rays, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_COHERENT | capi.FLAG_BOOL_OUT)
    assert np.array_equal(b.astype(bool), want["prim_id"] >= 0)
    tok = ctx.cast(rays, flags=capi.FLAG_COHERENT | capi.FLAG_TOKEN_OUT)
    assert np.array_equal(tok != capi.TOKEN_MISS, want["prim_id"] >= 0)
    inc = synth.incoherent_rays(n, 123)
    osc = po.OracleScene(verts)
    got_inc = ctx.cast(inc)
    parity.assert_exact(got_inc[:100000], osc.trace(inc[:100000]), "pipelined incoherent, first 100k")
    parity.assert_exact(got_inc[-50000:], osc.trace(inc[-50000:]), "pipelined incoherent, ragged tail")
    assert ctx.stats()["rays_cast"] > 0
    with pytest.raises(capi.MrtError):                   # errors of a chunk surface as errors of the cast
        ctx.cast(rays, mode=7)


def test_c3_full_headline(ctx):
    """Config C3 (headline): 1 M-triangle soup, 4096^2 primary rays."""
    _full_grid_case(ctx, "C3", (2040, 2056))


def test_c4_incoherent_sort_on_off(ctx):
    """Config C4: 1 M tris, 2^24 incoherent rays, Morton sort on vs off."""
    cfg = synth.CONFIGS["C4"]
    verts = synth.scene_vertices(cfg)
    scene = capi.Scene(verts)
    scene.upload(ctx)
    rays = synth.incoherent_rays(cfg["incoherent"], cfg["ray_seed"])
    n = rays.shape[0]
    d_rays = DeviceArray(ctx, rays.nbytes).upload(rays)
    d_hits = DeviceArray(ctx, n * 32)
    dev = capi.FLAG_RAYS_ON_DEVICE | capi.FLAG_HITS_ON_DEVICE
    ctx.cast(d_rays.ptr, d_hits.ptr, count=n, flags=dev | capi.FLAG_COHERENT)
    unsorted = d_hits.download(T.HIT32, n)
    ctx.cast(d_rays.ptr, d_hits.ptr, count=n, flags=dev)
    assert ctx.stats()["last_kernel_launches"] == 3
    assert d_hits.download(T.HIT32, n).tobytes() == unsorted.tobytes(), "sort on / sort off differ"
    d_rays.free(); d_hits.free()
    _assert_digest(digests.digest_records(unsorted), _full_digest("C4"), "C4 whole batch")  # every one of the 2^24 rays vs the oracle
    g = _golden("c4_sampled.npz")
    idx = g["index"]
    osc = po.OracleScene(verts)
    parity.assert_reference_parity(unsorted["prim_id"][idx], unsorted["t"][idx], g["prim_id"], g["t"], rays[idx], osc.tris, "C4")
    parity.assert_exact(unsorted[:200000], osc.trace(rays[:200000]), "C4 first 200k rays")
    # The reference's own hit count over the WHOLE batch, pinned exactly.  tests/golden/c4_hitmiss.npz (made by
    # tests/golden/make_c4_hitmiss.py from the reference compiled here) lists every ray of the 2^24 on which the reference and
    # the oracle disagree about hit / miss: 1 317 rays, 1 310 of them reference hits at 0 < t < t_min (the reference's CPU path
    # ignores Ray::t_min: SURVEY.md section 0, defect 6) on rays that hit nothing further on, 3 + 4 edge grazes either way.
    hm = _golden("c4_hitmiss.npz")
    meta = json.loads(str(hm["meta"]))
    dg = json.loads(str(g["digest"]))
    ours_hit = unsorted["prim_id"] >= 0
    assert meta["reference_hits"] == dg["hit_count"] and meta["rays"] == n
    assert int(ours_hit.sum()) == meta["reference_hits"] - meta["reference_only"] + meta["oracle_only"]
    assert np.array_equal(ours_hit[hm["index"]], ~hm["ref_hit"]), "the listed rays are exactly where the device and the reference differ"
    assert int(hm["below_t_min"].sum()) == meta["reference_only_below_t_min"] == 1310 and meta["disagreements"] == 1317
    assert (hm["ref_t"][hm["below_t_min"]] < rays["t_min"][hm["index"]][hm["below_t_min"]]).all()


def test_c5_multi_mesh_sharded_rows(ctx):
    """Config C5: 64 meshes x 156 250 tris flattened (raytracer_server.cpp:700-711), 8192^2 grid
    traced as 8 row blocks — the work of the 8 ranks — and checked on the reference's sampled rows."""
    cfg = synth.CONFIGS["C5"]
    w, h = cfg["grid"]
    verts = synth.scene_vertices(cfg)
    scene = capi.Scene(verts)
    scene.upload(ctx)
    cam = capi.camera_look(cfg["origin"], cfg["forward"], w, h, cfg["fov"])
    g = _golden("c5_sampled.npz")
    idx = g["index"]
    prim = np.empty(idx.shape[0], dtype=np.int32)
    t = np.empty(idx.shape[0], dtype=np.float32)
    hits_total = 0
    d_hits = DeviceArray(ctx, w * (h // 8) * 32)
    full = _full_digest("C5")
    acc = digests.Accumulator()
    for r in range(8):
        y0, y1 = r * h // 8, (r + 1) * h // 8
        ctx.cast_grid(cam, w, h, y0=y0, y1=y1, hits=d_hits.ptr, flags=capi.FLAG_HITS_ON_DEVICE)
        block = d_hits.download(T.HIT32, w * (y1 - y0))
        # every ray of the rank's block against the oracle's digest of that block
        _assert_digest(digests.digest_records(block, y0 * w), full["row_blocks"][r], f"C5 row block {r}")
        acc.add(block, y0 * w)
        sel = (idx >= y0 * w) & (idx < y1 * w)
        prim[sel] = block["prim_id"][idx[sel] - y0 * w]
        t[sel] = block["t"][idx[sel] - y0 * w]
        hits_total += int((block["prim_id"] >= 0).sum())
        if r == 3:  # one row of this block against the oracle, bit-exact
            osc_rows = po.grid_rays(cfg["origin"], cfg["forward"], w, h, cfg["fov"], y0 + 5, y0 + 6)
            band = block[5 * w:6 * w]
    d_hits.free()
    _assert_digest(acc.result(), full, "C5 whole grid (8 row blocks)")
    tris = capi.make_triangles(verts)
    rays_s = _sample_grid_rays(cfg, idx)
    parity.assert_reference_parity(prim, t, g["prim_id"], g["t"], rays_s, tris, "C5")
    assert hits_total > 0.5 * w * h
    osc = po.OracleScene(verts)
    parity.assert_exact(band, osc.trace(osc_rows), "C5 one row vs oracle")


def _tiled_wall(n=12, pitch=1.0):
    """An axis-aligned wall of n x n square tiles (two triangles each) at z = 0 plus a second, offset layer at
    z = 1.5 * pitch: every triangle edge lies on a multiple of the pitch, i.e. on faces of leaf and inner boxes."""
    tris = []
    for z, off in ((0.0, 0.0), (1.5 * pitch, 0.5)):
        for i in range(n):
            for j in range(n):
                x0, y0, x1, y1 = (i + off) * pitch, (j + off) * pitch, (i + off + 1) * pitch, (j + off + 1) * pitch
                a, b, c, d = (x0, y0, z), (x1, y0, z), (x1, y1, z), (x0, y1, z)
                tris += [(a, b, c), (a, c, d)]
    return np.array(tris, dtype=np.float32)


@pytest.mark.parametrize("pitch", [1.0, 0.3, 0.7])
@pytest.mark.parametrize("kernel", [capi.KERNEL_AUTO, capi.KERNEL_LANE, capi.KERNEL_PACKET, capi.KERNEL_PACKET_ASM, capi.KERNEL_PACKET_DUAL, capi.KERNEL_PACKET_ROWS, QUAD,
                                    capi.KERNEL_LANE_PERSISTENT, capi.KERNEL_LANE4_PERSISTENT, capi.KERNEL_LANE8_PERSISTENT])
def test_rays_that_graze_box_faces(built, kernel, pitch):
    """Axis-parallel rays whose origins lie exactly on tile edges: the ray runs IN a face plane of leaf and
    inner boxes, where the rounded slab test (direction component 0 -> inverse 1e9) and the triangle test
    can disagree.  The oracle walks the tree ray by ray; every kernel must give the same answers:
    a lane of a packet may only accept hits in leaves its OWN ray entered, and the 8-wide kernel, whose
    quantised boxes are looser than the tree's (pitch 0.3 is not on its power-of-two grids), only accepts a
    hit if the ray passes the leaf's exact box (Arithmetic)."""
    v = _tiled_wall(12, pitch)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    c = capi.Context(0, kernel=kernel)
    scene.upload(c)
    xs = (np.arange(0.0, 12.5, 0.25) * pitch).astype(np.float32)   # every 4th origin coordinate is a tile edge
    gx, gy = np.meshgrid(xs, xs, indexing="xy")
    n = gx.size
    rays = np.zeros(n, dtype=T.RAY32)
    rays["origin"][:, 0], rays["origin"][:, 1], rays["origin"][:, 2] = gx.ravel(), gy.ravel(), -3.0
    rays["direction"] = [0.0, 0.0, 1.0]
    rays["t_min"], rays["t_max"] = 0.001, T.FLT_MAX
    side = rays.copy()                                          # along the wall, inside its plane and just off it
    side["origin"][:, 0], side["origin"][:, 1] = -2.0, gy.ravel()
    side["origin"][:, 2] = np.where(np.arange(n) % 2 == 0, 0.0, np.float32(1.5 * pitch))
    side["direction"] = [1.0, 0.0, 0.0]
    diag = rays.copy()
    d = np.float32(1.0) / np.sqrt(np.float32(2.0))
    diag["direction"] = [d, 0.0, d]
    for batch, name in ((rays, "normal"), (side, "in-plane"), (diag, "diagonal"), (np.concatenate([rays, side, diag]), "mixed")):
        want = osc.trace(batch)
        for flags in (capi.FLAG_COHERENT, 0):
            parity.assert_exact(c.cast(batch, flags=flags), want, f"graze {name} kernel={kernel} flags={flags}")
        b = c.cast(batch, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_COHERENT | capi.FLAG_BOOL_OUT)
        assert np.array_equal(b.astype(bool), want["prim_id"] >= 0)
    # a batch large enough for the persistent kernels (AUTO: the 8-wide one)
    big = np.tile(np.concatenate([rays, side, diag]), 12)
    want = osc.trace(big)
    for flags in (capi.FLAG_COHERENT, 0):
        parity.assert_exact(c.cast(big, flags=flags), want, f"graze big kernel={kernel} flags={flags}")
    assert int((want["prim_id"] >= 0).sum()) > n and int((want["prim_id"] < 0).sum()) > 0
    c.close()


@pytest.mark.parametrize("kernel", [capi.KERNEL_AUTO, capi.KERNEL_LANE, capi.KERNEL_PACKET, capi.KERNEL_PACKET_ASM, capi.KERNEL_PACKET_DUAL, capi.KERNEL_PACKET_ROWS, QUAD,
                                    capi.KERNEL_LANE_PERSISTENT, capi.KERNEL_LANE4_PERSISTENT, capi.KERNEL_LANE8_PERSISTENT])
def test_non_finite_rays_do_not_disturb_their_neighbours(built, kernel):
    """The reference only asserts ray validity in debug builds (RT_ASSERT_VALID_RAY); a release caller can hand
    over NaN / infinite / zero-length rays.  They must terminate, must not change the answers of the valid rays
    that share their wave or packet, and -- the comparisons being written the same way -- give what the oracle
    gives for them."""
    v = synth.soup(4000, 0.4, 13)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    c = capi.Context(0, kernel=kernel)
    scene.upload(c)
    rays = np.tile(po.grid_rays((0, 0, -12), (0, 0, 1), 96, 96, 50.0), 8)[:70000].copy()   # large enough for the persistent kernels
    rng = np.random.default_rng(3)
    bad = rng.choice(rays.shape[0], 3000, replace=False)
    vals = np.float32([np.nan, np.inf, -np.inf, 0.0, -0.0, 3.0e38, -3.0e38, 1e-30])
    for k, i in enumerate(bad):
        what = k % 6
        if what == 0:
            rays["direction"][i] = vals[rng.integers(0, 8, 3)]
        elif what == 1:
            rays["origin"][i] = vals[rng.integers(0, 8, 3)]
        elif what == 2:
            rays["direction"][i] = 0.0
        elif what == 3:
            rays["t_min"][i] = vals[rng.integers(0, 8)]
        elif what == 4:
            rays["t_max"][i] = vals[rng.integers(0, 8)]
        else:
            rays["origin"][i, rng.integers(0, 3)] = vals[rng.integers(0, 3)]
            rays["direction"][i, rng.integers(0, 3)] = vals[rng.integers(0, 3)]
    good = np.ones(rays.shape[0], dtype=bool)
    good[bad] = False
    want = osc.trace(rays)
    for flags in (capi.FLAG_COHERENT, 0):
        got = c.cast(rays, flags=flags)
        parity.assert_exact(got[good], want[good], f"valid rays next to non-finite ones, kernel={kernel} flags={flags}")
        assert np.array_equal(got["prim_id"][bad], want["prim_id"][bad]), f"non-finite rays, kernel={kernel} flags={flags}"
        b = c.cast(rays, mode=capi.MODE_ANY_HIT, flags=flags | capi.FLAG_BOOL_OUT)
        assert np.array_equal(b.astype(bool)[good], want["prim_id"][good] >= 0)
    c.close()


@pytest.mark.parametrize("tile_order", [1, 2, 3])
@pytest.mark.parametrize("w,h", [(1024, 512), (1000, 520), (128, 128)])
def test_tile_order_never_changes_a_result(built, tile_order, w, h):
    """Row-major tiles or Z-order inside 16x16-tile super-tiles (the default for scenes beyond the Infinity
    Cache): the same records from every grid entry point, also when the grid does not divide into super-tiles
    (the mapping then falls back to row-major)."""
    v = synth.soup(20000, 0.25, 21)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    cam = capi.camera_look((0, 0, -12), (0, 0, 1), w, h, 50.0)
    rays = po.grid_rays((0, 0, -12), (0, 0, 1), w, h, 50.0)
    want = osc.trace(rays)
    for kernel in (capi.KERNEL_AUTO, capi.KERNEL_LANE):
        c = capi.Context(0, kernel=kernel, tile_order=tile_order)
        scene.upload(c)
        parity.assert_exact(c.cast_grid(cam, w, h), want, f"cast_grid order={tile_order} kernel={kernel}")
        parity.assert_exact(c.cast(rays, flags=capi.FLAG_COHERENT), want, f"cast order={tile_order} kernel={kernel}")
        d_rays, d_hits = c.device_alloc(w * h * 32), c.device_alloc(w * h * 32)
        c.h2d(d_rays, rays)
        c.cast_tiled(d_rays, d_hits, w, h)
        got = np.zeros(w * h, dtype=T.HIT32)
        c.d2h(got, d_hits)
        parity.assert_exact(got, want, f"cast_tiled order={tile_order} kernel={kernel}")
        part = c.cast_grid(cam, w, h, y0=h // 4, y1=h // 2)              # a row block of the grid (multi-GPU sharding)
        parity.assert_exact(part, want[(h // 4) * w:(h // 2) * w], f"row block order={tile_order} kernel={kernel}")
        c.device_free(d_rays); c.device_free(d_hits)
        c.close()


@pytest.mark.parametrize("opts", [dict(xcd_swizzle=1), dict(grid_tile=1), dict(sort_key=1), dict(sort_threshold=100000),
                                  dict(sort_threshold=1), dict(tile_w_log2=2), dict(tile_w_log2=4), dict(xcd_swizzle=1, kernel=capi.KERNEL_LANE),
                                  dict(xcd_swizzle=1, tile_order=2, tile_w_log2=4)])
def test_tuning_options_never_change_a_result(built, opts):
    """Every knob of mrt_options moves work around (lane mapping, workgroup order, sort key, sort threshold);
    none may move a result: grids and incoherent batches against the oracle under each of them."""
    v = synth.soup(15000, 0.3, 8)
    scene, osc = capi.Scene(v), po.OracleScene(v)
    c = capi.Context(0, **opts)
    scene.upload(c)
    w, h = 512, 256
    cam = capi.camera_look((0, 0, -12), (0, 0, 1), w, h, 50.0)
    grid = po.grid_rays((0, 0, -12), (0, 0, 1), w, h, 50.0)
    want = osc.trace(grid)
    parity.assert_exact(c.cast_grid(cam, w, h), want, f"cast_grid {opts}")
    parity.assert_exact(c.cast(grid, flags=capi.FLAG_COHERENT), want, f"cast coherent {opts}")
    inc = synth.incoherent_rays(70000, 12)
    want = osc.trace(inc)
    parity.assert_exact(c.cast(inc), want, f"cast incoherent {opts}")
    parity.assert_exact(c.cast(inc[:300]), want[:300], f"cast small {opts}")
    b = c.cast(inc, mode=capi.MODE_ANY_HIT, flags=capi.FLAG_BOOL_OUT)
    assert np.array_equal(b.astype(bool), want["prim_id"] >= 0)
    c.close()
