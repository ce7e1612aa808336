"""Deterministic synthetic scenes and ray batches.

Only IEEE-exact operations (+ - * / sqrt on float32, integer mixing) are used,
so the same arrays come out on every machine: the golden fixtures made in the
build container and the inputs regenerated on the GPU box are bit-identical.

PRNG: splitmix64 used as a counter-based generator; draw k of stream `seed` is
mix(seed + (k+1)*GAMMA); uniform float32 in [0,1) from the top 24 bits.
"""
import numpy as np

from .types import RAY32, FLT_MAX

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, start: int, count: int) -> np.ndarray:
    """Draws start .. start+count-1 of stream `seed` as uint64."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + k * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, start: int, count: int) -> np.ndarray:
    """float32 uniforms in [0,1): top 24 bits * 2^-24 (exact)."""
    return (splitmix64(seed, start, count) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def _soup_block(n: int, s: float, seed: int, start_tri: int) -> np.ndarray:
    u = uniform01(seed, start_tri * 12, n * 12).reshape(n, 12)
    c = u[:, 0:3] * np.float32(10.0) - np.float32(5.0)
    off = (u[:, 3:12] * np.float32(2.0) - np.float32(1.0)) * np.float32(s)
    return (c[:, None, :] + off.reshape(n, 3, 3)).astype(np.float32)


def soup(n_tris: int, s: float, seed: int) -> np.ndarray:
    """Soup(N, s, seed): centre c ~ U[-5,5]^3, vertices c + U[-s,s]^3.
    Returns (n,3,3) float32.  Degenerate triangles (|e1 x e2|^2 < 1e-12) are
    redrawn from stream seed + 0x51ED270B."""
    out = np.empty((n_tris, 3, 3), dtype=np.float32)
    step = 1 << 20
    for a in range(0, n_tris, step):
        b = min(n_tris, a + step)
        out[a:b] = _soup_block(b - a, s, seed, a)
    redraw_seed, rnd = seed + 0x51ED270B, 0
    while True:
        e1 = out[:, 1] - out[:, 0]
        e2 = out[:, 2] - out[:, 0]
        cr = np.cross(e1.astype(np.float64), e2.astype(np.float64))
        bad = np.nonzero((cr * cr).sum(axis=1) < 1e-12)[0]
        if bad.size == 0:
            return out
        out[bad] = _soup_block(bad.size, s, redraw_seed, rnd)
        rnd += bad.size


def cube() -> np.ndarray:
    """Unit cube at the origin, 12 triangles, outward winding (config C1)."""
    p = np.array([[-.5, -.5, -.5], [.5, -.5, -.5], [.5, .5, -.5], [-.5, .5, -.5],
                  [-.5, -.5, .5], [.5, -.5, .5], [.5, .5, .5], [-.5, .5, .5]], dtype=np.float32)
    quads = [(4, 5, 6, 7), (1, 0, 3, 2), (5, 1, 2, 6), (0, 4, 7, 3), (7, 6, 2, 3), (0, 1, 5, 4)]
    tris = []
    for a, b, c, d in quads:
        tris.append([p[a], p[b], p[c]])
        tris.append([p[a], p[c], p[d]])
    return np.asarray(tris, dtype=np.float32)


def _unit_vectors(seed: int, n: int, dims: int) -> np.ndarray:
    """n uniformly distributed unit vectors in `dims` dimensions by rejection
    from the cube (8 candidates per round), float32, IEEE-exact ops only."""
    out = np.zeros((n, dims), dtype=np.float32)
    k = 8
    step = 1 << 19
    for a in range(0, n, step):
        b = min(n, a + step)
        todo = np.arange(a, b)
        rnd = 0
        while todo.size:
            # vector i of round rnd uses draws [i*k*dims, (i+1)*k*dims) of stream seed + rnd*7919
            u = uniform01(seed + rnd * 7919, a * k * dims, (b - a) * k * dims).reshape(b - a, k, dims)[todo - a]
            p = u * np.float32(2.0) - np.float32(1.0)
            l2 = (p * p).sum(axis=2, dtype=np.float32)
            ok = (l2 <= np.float32(1.0)) & (l2 >= np.float32(1e-4))
            first = ok.argmax(axis=1)
            has = ok.any(axis=1)
            rows = np.arange(todo.size)
            sel = p[rows, first]
            ln = np.sqrt(l2[rows, first])
            vec = (sel / ln[:, None]).astype(np.float32)
            out[todo[has]] = vec[has]
            todo = todo[~has]
            rnd += 1
    return out


def incoherent_rays(m: int, seed: int = 7, extent: float = 6.0) -> np.ndarray:
    """Incoherent(M, seed): origin ~ U[-extent,extent]^3, direction uniform on
    the sphere; t_min = 0.001, t_max = FLT_MAX (Ray() defaults, src/core/ray.h:59)."""
    rays = np.zeros(m, dtype=RAY32)
    u = uniform01(seed, 0, m * 3).reshape(m, 3)
    rays["origin"] = u * np.float32(2.0 * extent) - np.float32(extent)
    d = _unit_vectors(seed + 0x0DD5EED, m, 3)
    # renormalise once in float32 so |d| is as close to 1 as float32 allows
    l = np.sqrt((d * d).sum(axis=1, dtype=np.float32))
    rays["direction"] = (d / l[:, None]).astype(np.float32)
    rays["t_min"] = np.float32(0.001)
    rays["t_max"] = FLT_MAX
    return rays


def incoherent_rays_at(index: np.ndarray, seed: int = 7, extent: float = 6.0) -> np.ndarray:
    """Rays index[0], index[1], ... of incoherent_rays(m, seed, extent) for any m > max(index), without generating the
    whole batch (the generator is counter based: ray i takes draws [3 i, 3 i + 3) of the origin stream and draws
    [24 i, 24 i + 24) of each rejection round of the direction stream).  Same arithmetic, same bits."""
    index = np.asarray(index, dtype=np.int64)
    rays = np.zeros(index.shape[0], dtype=RAY32)
    d = np.zeros((index.shape[0], 3), dtype=np.float32)
    for j, i in enumerate(index):
        i = int(i)
        u = uniform01(seed, i * 3, 3)
        rays["origin"][j] = u * np.float32(2.0 * extent) - np.float32(extent)
        rnd = 0
        while True:   # _unit_vectors(seed + 0x0DD5EED, m, 3), element i: first of 8 candidates per round inside the unit ball
            c = uniform01(seed + 0x0DD5EED + rnd * 7919, i * 24, 24).reshape(8, 3) * np.float32(2.0) - np.float32(1.0)
            l2 = (c * c).sum(axis=1, dtype=np.float32)
            ok = (l2 <= np.float32(1.0)) & (l2 >= np.float32(1e-4))
            if ok.any():
                k = int(ok.argmax())
                d[j] = (c[k] / np.sqrt(l2[k])).astype(np.float32)
                break
            rnd += 1
    l = np.sqrt((d * d).sum(axis=1, dtype=np.float32))
    rays["direction"] = (d / l[:, None]).astype(np.float32)
    rays["t_min"] = np.float32(0.001)
    rays["t_max"] = FLT_MAX
    return rays


def multi_mesh(n_meshes: int = 64, tris_per_mesh: int = 156250, s: float = 0.025, seed: int = 100):
    """Config C5: n_meshes soups, each scaled into a cell of a 4x4x4 lattice
    spanning [-5,5]^3 with a seeded rigid rotation, flattened to world space
    with running triangle ids (raytracer_server.cpp:700-711 semantics).
    Returns (verts (N,3,3) float32, mesh_of_tri (N,) uint32)."""
    side = 4
    cell = np.float32(10.0 / side)
    scale = np.float32(0.24)          # [-5,5] -> [-1.2,1.2]: a 2.4-unit cell
    verts = np.empty((n_meshes * tris_per_mesh, 3, 3), dtype=np.float32)
    mesh_id = np.empty(n_meshes * tris_per_mesh, dtype=np.uint32)
    quats = _unit_vectors(seed + 0xA11CE, n_meshes, 4)
    for m in range(n_meshes):
        local = soup(tris_per_mesh, s, seed + m) * scale
        w, x, y, z = [np.float32(v) for v in quats[m]]
        two = np.float32(2.0)
        one = np.float32(1.0)
        rot = np.array([[one - two * (y * y + z * z), two * (x * y - w * z), two * (x * z + w * y)],
                        [two * (x * y + w * z), one - two * (x * x + z * z), two * (y * z - w * x)],
                        [two * (x * z - w * y), two * (y * z + w * x), one - two * (x * x + y * y)]], dtype=np.float32)
        lattice = m % (side ** 3)
        cx = np.float32(-5.0) + cell * np.float32((lattice % side) + 0.5)
        cy = np.float32(-5.0) + cell * np.float32(((lattice // side) % side) + 0.5)
        cz = np.float32(-5.0) + cell * np.float32((lattice // (side * side)) + 0.5)
        p = local.reshape(-1, 3)
        wx = p[:, 0] * rot[0, 0] + p[:, 1] * rot[0, 1] + p[:, 2] * rot[0, 2] + cx
        wy = p[:, 0] * rot[1, 0] + p[:, 1] * rot[1, 1] + p[:, 2] * rot[1, 2] + cy
        wz = p[:, 0] * rot[2, 0] + p[:, 1] * rot[2, 1] + p[:, 2] * rot[2, 2] + cz
        a = m * tris_per_mesh
        verts[a:a + tris_per_mesh] = np.stack([wx, wy, wz], axis=1).reshape(-1, 3, 3)
        mesh_id[a:a + tris_per_mesh] = m
    return verts, mesh_id


def multi_mesh_instances(n_meshes: int = 64, tris_per_mesh: int = 156250, s: float = 0.025, seed: int = 100):
    """The same scene as multi_mesh() before flattening: mesh-space vertices (n_meshes*tris_per_mesh,3,3)
    and one types.INSTANCE per mesh (rotation + lattice-cell origin), the input of
    mrt_build_instanced_scene_device.  flatten_instances(*multi_mesh_instances(...)) == multi_mesh(...)[0]."""
    from . import types as T
    side = 4
    cell = np.float32(10.0 / side)
    scale = np.float32(0.24)
    local = np.empty((n_meshes * tris_per_mesh, 3, 3), dtype=np.float32)
    inst = np.zeros(n_meshes, dtype=T.INSTANCE)
    quats = _unit_vectors(seed + 0xA11CE, n_meshes, 4)
    for m in range(n_meshes):
        local[m * tris_per_mesh:(m + 1) * tris_per_mesh] = soup(tris_per_mesh, s, seed + m) * scale
        w, x, y, z = [np.float32(v) for v in quats[m]]
        two, one = np.float32(2.0), np.float32(1.0)
        rot = np.array([[one - two * (y * y + z * z), two * (x * y - w * z), two * (x * z + w * y)],
                        [two * (x * y + w * z), one - two * (x * x + z * z), two * (y * z - w * x)],
                        [two * (x * z - w * y), two * (y * z + w * x), one - two * (x * x + y * y)]], dtype=np.float32)
        lattice = m % (side ** 3)
        inst[m]["first_tri"], inst[m]["n_tris"], inst[m]["layers"] = m * tris_per_mesh, tris_per_mesh, 0xFFFFFFFF
        inst[m]["basis"] = rot.reshape(9)
        inst[m]["origin"] = [np.float32(-5.0) + cell * np.float32((lattice % side) + 0.5),
                             np.float32(-5.0) + cell * np.float32(((lattice // side) % side) + 0.5),
                             np.float32(-5.0) + cell * np.float32((lattice // (side * side)) + 0.5)]
    return local, inst


def flatten_instances(local: np.ndarray, inst: np.ndarray) -> np.ndarray:
    """World-space vertices of every instance, in instance order: Transform3D::xform per vertex as
    RayTracerServer::_rebuild_scene applies it (raytracer_server.cpp:700-711): basis row . v summed
    left to right, plus origin, all in float32."""
    out = []
    for i in inst:
        p = local[int(i["first_tri"]):int(i["first_tri"]) + int(i["n_tris"])].reshape(-1, 3)
        b, o = i["basis"].astype(np.float32), i["origin"].astype(np.float32)
        w = [p[:, 0] * b[3 * r] + p[:, 1] * b[3 * r + 1] + p[:, 2] * b[3 * r + 2] + o[r] for r in range(3)]
        out.append(np.stack(w, axis=1).reshape(-1, 3, 3))
    return np.concatenate(out).astype(np.float32)


# Named workloads of BASELINE.json `configs`.
CONFIGS = {
    "C1": dict(scene="cube", grid=(16, 12), origin=(0.0, 0.0, 3.0), forward=(0.0, 0.0, -1.0), fov=60.0),
    "C2": dict(scene="soup", n_tris=100_000, s=0.10, seed=1, grid=(1024, 1024),
               origin=(0.0, 0.0, -12.0), forward=(0.0, 0.0, 1.0), fov=50.0),
    "C3": dict(scene="soup", n_tris=1_000_000, s=0.05, seed=1, grid=(4096, 4096),
               origin=(0.0, 0.0, -12.0), forward=(0.0, 0.0, 1.0), fov=50.0),
    "C4": dict(scene="soup", n_tris=1_000_000, s=0.05, seed=1, incoherent=1 << 24, ray_seed=7),
    "C5": dict(scene="multi_mesh", n_meshes=64, tris_per_mesh=156_250, s=0.025, seed=100, grid=(8192, 8192),
               origin=(0.0, 0.0, -12.0), forward=(0.0, 0.0, 1.0), fov=50.0),
}


def scene_vertices(cfg: dict) -> np.ndarray:
    if cfg["scene"] == "cube":
        return cube()
    if cfg["scene"] == "soup":
        return soup(cfg["n_tris"], cfg["s"], cfg["seed"])
    if cfg["scene"] == "multi_mesh":
        return multi_mesh(cfg["n_meshes"], cfg["tris_per_mesh"], cfg["s"], cfg["seed"])[0]
    raise ValueError(cfg["scene"])
