// kernels.hip — gfx950 (CDNA4, wave64) kernels of the batch ray-cast path.
//
// Replaces the reference's GLSL compute shader
// src/gpu/shaders/bvh_traverse.comp.glsl (one thread = one ray, stack-based
// ordered BVH2 traversal over 64-byte dual-AABB nodes, slab test, Moller-
// Trumbore) and the two host conversion loops around it
// (src/gpu/gpu_ray_caster.cpp:639-650, 442-456), which run on the device here.
//
// Arithmetic is the canonical form documented in DESIGN.md ("Arithmetic"):
// compiled with -ffp-contract=off, every fused operation is an explicit
// __builtin_fmaf, so results are bit-identical to oracle/mrt_oracle.c.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "mrt_internal.h"

namespace mrt {

#define MRT_WG 256
#define MRT_WAVE 64

// ---- canonical arithmetic ------------------------------------------------------
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
	return fma_(ax, bx, fma_(ay, by, az * bz));
}
// safe_inv_direction, bvh_traverse.comp.glsl:137-145 == Ray::_precompute, src/core/ray.h:78-89
__device__ __forceinline__ float safe_inv(float d)
{
	const float eps = 1e-9f;
	const float big = 1.0f / eps;
	return __builtin_fabsf(d) > eps ? 1.0f / d : (d >= 0.0f ? big : -big);
}

struct RayRegs {
	float ox, oy, oz, dx, dy, dz, t_min, t_max;
};

// Two launches are queued for a batch declared coherent: the packet kernel and, behind it, the
// lane kernel.  detect_grid_kernel decides on the device which one does the work; the other
// returns here (a few microseconds for an empty grid, no host round trip).
__device__ __forceinline__ bool skip_launch(const TraceParams &p)
{
	return p.skip_flag != nullptr && *p.skip_flag == p.skip_when;
}

// ---- lane -> ray mapping ---------------------------------------------------------
// MAP_LINEAR: thread g traces ray g (or perm[g]).  MAP_TILE8X8: a wave owns an
// 8x8 pixel tile of the row-major grid so its 64 rays share most of their path.
__device__ __forceinline__ bool lane_ray_index_g(const TraceParams &p, uint64_t g, uint64_t &ray_idx, uint32_t &px, uint32_t &py);
__device__ __forceinline__ bool lane_ray_index(const TraceParams &p, uint32_t block, uint64_t &ray_idx, uint32_t &px, uint32_t &py)
{
	return lane_ray_index_g(p, (uint64_t)block * MRT_WG + threadIdx.x, ray_idx, px, py);
}
// tile_order 3: every XCD works on its own column strips of the image.  Workgroups are dealt to the 8 XCDs round-robin
// (workgroup i runs on XCD i & 7), and each XCD has its own 4 MB L2: in row-major launch order every XCD sees every
// tile column, so the rows a band of tiles needs are fetched into all eight L2s (C5: 13 GB of L2 fills per launch for a
// 1.8 GB scene).  Here the image is cut into 8 m strips (about 256 pixels wide), XCD k takes strips k, k + 8, ..., one
// after the other, each from top to bottom: the ~1000 waves an XCD has in flight cover one compact region, and the
// strips of every XCD are spread evenly over the image, so cheap and expensive regions balance (a contiguous band per
// XCD, xcd_swizzle = 1, measured 6 % slower for that reason).  tile_group = consecutive tiles per workgroup.  Returns
// false (row-major order) when the width does not split into 8 m strips of whole workgroups.
__device__ __forceinline__ bool xcd_strips(const TraceParams &p, uint64_t tile, uint32_t tiles_x, uint32_t tiles_y, uint32_t &tx, uint32_t &ty)
{
	const uint32_t tg = p.tile_group ? p.tile_group : 1u;
	uint32_t m = (tiles_x + 128u) >> 8;
	if (m == 0u) m = 1u;
	const uint32_t S = tiles_x / (8u * m);
	if (S == 0u || S * 8u * m != tiles_x || S % tg != 0u) return false;
	const uint64_t wg = tile / tg;
	const uint32_t k = (uint32_t)wg & 7u;
	const uint64_t j = (wg >> 3) * tg + tile % tg; // the tile's place in its XCD's own sequence
	const uint64_t per_strip = (uint64_t)S * tiles_y;
	const uint32_t sl = (uint32_t)(j / per_strip), r = (uint32_t)(j % per_strip);
	ty = r / S;
	tx = (sl * 8u + k) * S + r % S;
	return true;
}

// g = virtual thread index: 64 consecutive g form one wave-sized group of rays
__device__ __forceinline__ bool lane_ray_index_g(const TraceParams &p, uint64_t g, uint64_t &ray_idx, uint32_t &px, uint32_t &py)
{
	uint32_t lane_map = p.lane_map, grid_w = p.grid_w, rows = p.rows, tiles_x = p.tiles_x;
	if (lane_map == MAP_AUTO) { // row width found on the device by detect_grid_kernel (0 = not a grid)
		const uint32_t w = p.auto_grid[0];
		lane_map = w ? MAP_TILE8X8 : MAP_LINEAR;
		grid_w = w; rows = p.auto_grid[1]; tiles_x = p.auto_grid[2];
	}
	if (lane_map == MAP_TILE8X8) {
		uint64_t tile = g >> 6;
		const uint32_t l = (uint32_t)g & 63u;
		uint32_t tx, ty;
		const uint32_t k = p.tile_w_log2; // tile is 2^k wide, 64 / 2^k high
		const uint32_t tiles_y = (rows + (64u >> k) - 1u) >> (6u - k);
		// the schedule of the previous frame: launch slot -> unit of tile_unit consecutive tiles.  Only if it is a schedule
		// of THIS grid (a batch whose row width is found on the device, MAP_AUTO, was scheduled from the last cast's width)
		uint32_t quarter = 4u; // 0..3: the slot's wave works on that 4x4 quarter of its tile, in lanes 0..15
		uint32_t sixteenth = 16u; // 0..15 (quarter_all == 2): on that 2x2 sixteenth of its tile, in lanes 0..3
		if (p.quarter_all == 2u) { sixteenth = (uint32_t)tile & 15u; tile >>= 4; if (l >= 4u || tile >= (uint64_t)tiles_x * tiles_y) return false; }
		else if (p.quarter_all) { quarter = (uint32_t)tile & 3u; tile >>= 2; if (l >= 16u || tile >= (uint64_t)tiles_x * tiles_y) return false; }
		else if (p.tile_sched != nullptr && ((uint64_t)tiles_x * tiles_y + p.tile_unit - 1u) / p.tile_unit == p.n_units) {
			const uint64_t slot = tile / p.tile_unit;
			if (slot >= (p.sched_hdr ? p.sched_hdr[2] : p.n_units)) return false;
			const uint32_t e = p.tile_sched[slot], what = e >> 28, id = e & 0x0FFFFFFFu;
			if (what == 0u) tile = (uint64_t)id * p.tile_unit + tile % p.tile_unit;
			else {
				if (tile % p.tile_unit != 0u) return false; // a piece is (part of) one tile: a second group of the wave has nothing to do
				tile = id;
				if (what >= 2u) { quarter = what - 2u; if (l >= 16u) return false; }
			}
		}
		if (p.tile_order == 1u && (tiles_x & 15u) == 0u && (tiles_y & 15u) == 0u) {
			// 16x16-tile super-tiles in row-major order, Z-order inside: the tiles in flight at
			// any moment cover a compact image region, so they share deep BVH nodes in L2
			const uint32_t st = (uint32_t)(tile >> 8), in = (uint32_t)tile & 255u;
			uint32_t mx = in & 0x55u, my = (in >> 1) & 0x55u; // de-interleave 4+4 bits
			mx = (mx | (mx >> 1)) & 0x33u; mx = (mx | (mx >> 2)) & 0x0Fu;
			my = (my | (my >> 1)) & 0x33u; my = (my | (my >> 2)) & 0x0Fu;
			const uint32_t stx = st % (tiles_x >> 4), sty = st / (tiles_x >> 4);
			tx = (stx << 4) + mx; ty = (sty << 4) + my;
		} else if (p.tile_order == 2u && (tiles_x & 31u) == 0u && (tiles_y & 31u) == 0u) {
			// the same with 32x32-tile super-tiles
			const uint32_t st = (uint32_t)(tile >> 10), in = (uint32_t)tile & 1023u;
			uint32_t mx = in & 0x155u, my = (in >> 1) & 0x155u; // de-interleave 5+5 bits
			mx = (mx | (mx >> 1)) & 0x133u; mx = (mx | (mx >> 2)) & 0x10Fu; mx = (mx | (mx >> 4)) & 0x1Fu;
			my = (my | (my >> 1)) & 0x133u; my = (my | (my >> 2)) & 0x10Fu; my = (my | (my >> 4)) & 0x1Fu;
			const uint32_t stx = st % (tiles_x >> 5), sty = st / (tiles_x >> 5);
			tx = (stx << 5) + mx; ty = (sty << 5) + my;
		} else if (p.tile_order == 3u && xcd_strips(p, tile, tiles_x, tiles_y, tx, ty)) {
		} else { tx = (uint32_t)(tile % tiles_x); ty = (uint32_t)(tile / tiles_x); }
		if (sixteenth < 16u) {
			px = (tx << 3) + ((sixteenth & 3u) << 1) + (l & 1u);
			py = (ty << 3) + ((sixteenth >> 2) << 1) + (l >> 1);
		} else if (quarter < 4u) { // (schedule pieces exist for 8x8 tiles only: k == 3)
			px = (tx << 3) + ((quarter & 1u) << 2) + (l & 3u);
			py = (ty << 3) + ((quarter >> 1) << 2) + (l >> 2);
		} else {
			px = (tx << k) + (l & ((1u << k) - 1u));
			py = (ty << (6u - k)) + (l >> k);
		}
		if (px >= grid_w || py >= rows) return false;
		ray_idx = (uint64_t)py * grid_w + px;
		return true;
	}
	if (p.sparse_lanes) { const uint32_t l = (uint32_t)g & 63u; if (l >= p.sparse_lanes) return false; g = (g >> 6) * p.sparse_lanes + l; }
	if (g >= p.count) return false;
	ray_idx = p.perm ? (uint64_t)p.perm[g] : g;
	if (p.in_fmt == IN_GRID) { px = (uint32_t)(ray_idx % p.grid_w); py = (uint32_t)(ray_idx / p.grid_w); }
	return true;
}

// What a wave's schedule unit cost (shader cycles, modulo 2^32), for the next frame's longest-first launch order.  The
// start time is parked in the cost word itself (note_tile_start) and replaced by the difference at the end
// (note_tile_cost): nothing stays in registers across the walk.  One lane per wave calls; a unit belongs to one wave.
__device__ __forceinline__ bool tile_cost_word(const TraceParams &p, uint64_t g_first, uint32_t *&park, uint32_t *&sum, uint32_t &what)
{
	if (p.tile_cost == nullptr) return false;
	uint32_t rows = p.rows, tiles_x = p.tiles_x;
	if (p.lane_map == MAP_AUTO) { if (p.auto_grid[0] == 0u) return false; rows = p.auto_grid[1]; tiles_x = p.auto_grid[2]; }
	else if (p.lane_map != MAP_TILE8X8) return false;
	const uint32_t k = p.tile_w_log2, tiles_y = (rows + (64u >> k) - 1u) >> (6u - k);
	if (((uint64_t)tiles_x * tiles_y + p.tile_unit - 1u) / p.tile_unit != p.n_units) return false; // not the grid the arrays were sized for
	const uint64_t slot = (g_first >> 6) / p.tile_unit;
	what = 0u;
	if (p.tile_sched == nullptr) { if (slot >= p.n_units) return false; park = sum = p.tile_cost + slot; return true; }
	if (slot >= (p.sched_hdr ? p.sched_hdr[2] : p.n_units)) return false;
	const uint32_t e = p.tile_sched[slot], id = e & 0x0FFFFFFFu;
	what = e >> 28;
	if (what == 0u) { park = sum = p.tile_cost + id; return true; }
	park = p.tile_cost + p.n_units + slot; // a piece of a unit: its own word for the start time, its share added to the unit's
	sum = p.tile_cost + id / p.tile_unit;
	return true;
}
__device__ __forceinline__ void note_tile_start(const TraceParams &p, uint64_t g_first)
{
	uint32_t *park, *sum, what;
	if (tile_cost_word(p, g_first, park, sum, what)) *park = (uint32_t)__builtin_amdgcn_s_memtime();
}
// A unit launched in pieces notes what it would have cost in one piece, as well as that can be said: two single tiles take
// about 1.3 x their pair, the eight quarter tiles of a pair 1.5 x the pair, the four of a tile 1.15 x the tile (MRT_SCHED_DUMP
// of consecutive renewals of one grid, 1920x1080 on the C3 scene: the same pair 1.43 M cycles whole, 1.82 M as two tiles,
// 2.1 M as eight quarters) -- so that a unit is ranked as what it is, not as the sum of its pieces.
__device__ __forceinline__ void note_tile_cost(const TraceParams &p, uint64_t g_first)
{
	uint32_t *park, *sum, what;
	if (!tile_cost_word(p, g_first, park, sum, what)) return;
	uint32_t d = (uint32_t)__builtin_amdgcn_s_memtime() - *park;
	if (park == sum) { *sum = d ? d : 1u; return; }
	if (what == 1u) d = d - (d >> 2);                                  // x 3/4
	else d = p.tile_unit == 2u ? (d >> 1) + (d >> 3) + (d >> 4) : d - (d >> 3); // quarters: x 11/16 of a pair's eight, x 7/8 of a tile's four
	atomicAdd(sum, d ? d : 1u);
}

// Primary-ray grids.  MRT_CAMERA_DEBUG_GRID: RayTracerDebug::cast_debug_rays, src/godot/raytracer_debug.cpp:585-596
// (basis / half extents precomputed on the host, mrt_camera_look, :573-583).  MRT_CAMERA_PERSPECTIVE /
// _ORTHOGRAPHIC: RayCamera::_generate_perspective / _generate_orthographic, src/modules/graphics/
// ray_camera.h:234-273 (v flipped; Basis::xform = one dot product per row, summed left to right; the
// jittered form of :106-122 with the pixel centre 0.5 as the default offset).  Plain float operations in the
// reference's order (nothing is contracted): bit-identical to the host loops.
__device__ __forceinline__ void grid_ray(const TraceParams &p, uint32_t px, uint32_t py, RayRegs &r)
{
	const mrt_camera &c = p.cam;
	float dx, dy, dz;
	r.ox = c.origin[0]; r.oy = c.origin[1]; r.oz = c.origin[2];
	if (c.kind == MRT_CAMERA_DEBUG_GRID) {
		const float u = (2.0f * ((float)px + 0.5f) / (float)p.grid_w - 1.0f) * c.half_w;
		const float v = (2.0f * ((float)(py + p.y0) + 0.5f) / (float)p.grid_h - 1.0f) * c.half_h;
		dx = c.fwd[0] + c.right[0] * u + c.up[0] * v;
		dy = c.fwd[1] + c.right[1] * u + c.up[1] * v;
		dz = c.fwd[2] + c.right[2] * u + c.up[2] * v;
	} else {
		const float u = (2.0f * ((float)px + c.jitter_x) * c.inv_w) - 1.0f;
		const float v = 1.0f - (2.0f * ((float)(py + p.y0) + c.jitter_y) * c.inv_h);
		if (c.kind == MRT_CAMERA_PERSPECTIVE) {
			const float vx = u * c.half_w, vy = v * c.half_h; // view_dir = (vx, vy, -1)
			dx = c.right[0] * vx + c.up[0] * vy + c.fwd[0] * -1.0f;
			dy = c.right[1] * vx + c.up[1] * vy + c.fwd[1] * -1.0f;
			dz = c.right[2] * vx + c.up[2] * vy + c.fwd[2] * -1.0f;
		} else { // parallel rays: the direction is -column 2 as it stands (Ray(ray_origin, forward_): not normalised)
			const float sv = v * c.half_h, su = u * c.half_w;
			r.ox = (c.origin[0] + c.up[0] * sv) + c.right[0] * su;
			r.oy = (c.origin[1] + c.up[1] * sv) + c.right[1] * su;
			r.oz = (c.origin[2] + c.up[2] * sv) + c.right[2] * su;
			r.dx = -c.fwd[0]; r.dy = -c.fwd[1]; r.dz = -c.fwd[2];
			r.t_min = c.t_min; r.t_max = c.t_max;
			return;
		}
	}
	const float l2 = dx * dx + dy * dy + dz * dz;
	if (l2 == 0.0f) { dx = dy = dz = 0.0f; }
	else { const float l = __builtin_sqrtf(l2); dx /= l; dy /= l; dz /= l; }
	r.dx = dx; r.dy = dy; r.dz = dz;
	r.t_min = c.t_min; r.t_max = c.t_max;
}

__device__ __forceinline__ void load_ray(const TraceParams &p, uint64_t idx, uint32_t px, uint32_t py, RayRegs &r)
{
	if (p.in_fmt == IN_GRID) { grid_ray(p, px, py, r); return; }
	float ox, oy, oz, dx, dy, dz, t0, t1;
	if (p.in_fmt == IN_HOST60) { // Ray -> GPURayPacked, gpu_ray_caster.cpp:643-650
		const float *h = reinterpret_cast<const float *>(p.rays) + idx * 15u;
		ox = h[0]; oy = h[1]; oz = h[2]; dx = h[3]; dy = h[4]; dz = h[5];
		t0 = h[12]; t1 = h[13];
	} else {
		const float4 *q = reinterpret_cast<const float4 *>(p.rays) + idx * 2u;
		const float4 a = q[0], b = q[1];
		ox = a.x; oy = a.y; oz = a.z; t1 = a.w;
		dx = b.x; dy = b.y; dz = b.z; t0 = b.w;
	}
	r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz; r.t_min = t0; r.t_max = t1;
}

// Result store: bvh_traverse.comp.glsl:322-327, plus the readback conversion of
// gpu_ray_caster.cpp:442-456 (OUT_HOST44) / :482-487 (OUT_BOOL8) fused in.
__device__ __forceinline__ void store_hit(const TraceParams &p, uint64_t idx, const RayRegs &r,
		float t, int32_t prim, float u, float v, float nx, float ny, float nz, uint32_t layers, uint32_t slot)
{
	if (p.out_fmt == OUT_BOOL8) { reinterpret_cast<uint8_t *>(p.hits)[idx] = prim >= 0 ? 1 : 0; return; }
	if (p.out_fmt == OUT_TOKEN4) { reinterpret_cast<uint32_t *>(p.hits)[idx] = prim >= 0 ? slot : 0xFFFFFFFFu; return; }
	if (p.out_fmt == OUT_HOST44) {
		float *h = reinterpret_cast<float *>(p.hits) + idx * 11u;
		uint32_t *hu = reinterpret_cast<uint32_t *>(h);
		if (prim >= 0) {
			h[0] = t;
			h[1] = r.ox + r.dx * t; h[2] = r.oy + r.dy * t; h[3] = r.oz + r.dz * t;
			h[4] = nx; h[5] = ny; h[6] = nz; h[7] = u; h[8] = v;
			hu[9] = (uint32_t)prim; hu[10] = layers;
		} else { // Intersection::set_miss on a default-constructed record
			h[0] = FLT_MAX; h[1] = h[2] = h[3] = 0.0f; h[4] = h[5] = h[6] = 0.0f; h[7] = h[8] = 0.0f;
			hu[9] = 0xFFFFFFFFu; hu[10] = 0u;
		}
		return;
	}
	float4 *q = reinterpret_cast<float4 *>(p.hits) + idx * 2u;
	float4 a, b;
	a.x = t; a.y = __int_as_float(prim); a.z = u; a.w = v;
	b.x = nx; b.y = ny; b.z = nz; b.w = __uint_as_float(layers);
	q[0] = a; q[1] = b;
}

// End of a ray in every kernel: look up what the record needs about the winning triangle (id,
// layers, the cold normal row) and store it.  Bool and token outputs need none of that.
__device__ __forceinline__ void finish_ray(const TraceParams &p, uint64_t ray_idx, const RayRegs &r,
		float best_t, float best_u, float best_v, uint32_t best_slot)
{
	int32_t prim = -1; float nx = 0.0f, ny = 0.0f, nz = 0.0f; uint32_t layers = 0u;
	if (best_slot != 0xFFFFFFFFu) {
		if (p.out_fmt == OUT_BOOL8 || p.out_fmt == OUT_TOKEN4) prim = 0; // only "hit or not" (and the slot) is stored
		else {
			prim = (int32_t)p.tri_hot[best_slot].id;
			layers = p.tri_hot[best_slot].layers;
			const float4 nn = reinterpret_cast<const float4 *>(p.tri_cold)[best_slot];
			nx = nn.x; ny = nn.y; nz = nn.z;
		}
	}
	store_hit(p, ray_idx, r, best_t, prim, best_u, best_v, nx, ny, nz, layers, best_slot);
}

// The same for a two-level scene (SceneTLAS::cast_ray, src/accel/scene_tlas.h:217-244): prim_id = the flat
// id (instance id base + mesh-local index, already in best_id), hit_layers = the instance's mask, normal =
// normalize(basis * mesh-space normal); DevInstance row = 8 float4: basis at words 12..20, mask at word 23.
__device__ __forceinline__ void finish_two_level_ray(const TraceParams &p, uint64_t ray_idx, const RayRegs &r,
		float best_t, float best_u, float best_v, uint32_t best_slot, uint32_t best_id, uint32_t best_inst)
{
	int32_t prim = -1; float nx = 0.0f, ny = 0.0f, nz = 0.0f; uint32_t layers = 0u;
	if (p.out_fmt == OUT_TOKEN8) { // {triangle slot, instance row}: expand_two_level_tokens_kernel rebuilds the record
		reinterpret_cast<uint2 *>(p.hits)[ray_idx] = make_uint2(best_slot, best_slot != 0xFFFFFFFFu ? best_inst : 0u);
		return;
	}
	if (best_slot != 0xFFFFFFFFu) {
		prim = (int32_t)best_id;
		if (p.out_fmt != OUT_BOOL8) {
			const float4 *row = reinterpret_cast<const float4 *>(p.instances) + (size_t)best_inst * 8u;
			const float4 b0 = row[3], b1 = row[4], b2 = row[5];
			const float4 no = reinterpret_cast<const float4 *>(p.tri_cold)[best_slot];
			nx = fma_(b0.x, no.x, fma_(b0.y, no.y, b0.z * no.z));
			ny = fma_(b0.w, no.x, fma_(b1.x, no.y, b1.y * no.z));
			nz = fma_(b1.z, no.x, fma_(b1.w, no.y, b2.x * no.z));
			const float l2 = fma_(nx, nx, fma_(ny, ny, nz * nz));
			if (l2 == 0.0f) { nx = ny = nz = 0.0f; }
			else { const float l = __builtin_sqrtf(l2); nx /= l; ny /= l; nz /= l; }
			layers = __float_as_uint(b2.w);
		}
	}
	store_hit(p, ray_idx, r, best_t, prim, best_u, best_v, nx, ny, nz, layers, best_slot);
}

// ---- the traversal kernel: one lane = one ray -------------------------------------
// LDS: per-lane stack, entry d of lane l at dword d*64 + l of the wave's region
// (conflict-free: the 64 lanes of a push/pop hit 64 consecutive dwords).
template <bool ANY_HIT, bool COUNT>
__global__ __launch_bounds__(MRT_WG) void trace_lane_kernel(const TraceParams p)
{
	extern __shared__ uint32_t lds_stack[];
	if (skip_launch(p)) return;
	uint32_t block = blockIdx.x;
	if (p.xcd_swizzle) { // contiguous band of the batch per XCD (blocks are dealt round-robin over 8 XCDs)
		const uint32_t per = gridDim.x >> 3;
		if (block < (per << 3)) block = (block & 7u) * per + (block >> 3);
	}
	uint64_t ray_idx = 0; uint32_t px = 0, py = 0;
	if (!lane_ray_index(p, block, ray_idx, px, py)) return;
	RayRegs r;
	load_ray(p, ray_idx, px, py, r);

	float best_t = r.t_max, best_u = 0.0f, best_v = 0.0f;
	uint32_t best_slot = 0xFFFFFFFFu, best_id = 0xFFFFFFFFu;
	uint32_t n_nodes = 0, n_tris = 0, max_sp = 0;

	if (!(r.t_min >= r.t_max)) { // degenerate rays are misses, glsl:214-222
		const float ix = safe_inv(r.dx), iy = safe_inv(r.dy), iz = safe_inv(r.dz);
		const float nrx = -(r.ox * ix), nry = -(r.oy * iy), nrz = -(r.oz * iz);
		const uint32_t lane = threadIdx.x & (MRT_WAVE - 1);
		const uint32_t wave = threadIdx.x / MRT_WAVE;
		uint32_t sp = wave * (p.stack_depth * MRT_WAVE) + lane; // dword index of this lane's stack bottom
		lds_stack[sp] = kSentinel; sp += MRT_WAVE;
		uint32_t cur = 0; // the root is always a wide node (root leaves are wrapped on the host)
		const float4 *nodes = reinterpret_cast<const float4 *>(p.nodes);
		const float4 *hot = reinterpret_cast<const float4 *>(p.tri_hot);

		while (cur != kSentinel) {
			// ---- inner nodes: glsl:243-318 ----
			while (cur < kSentinel) {
				const float4 *n = nodes + (size_t)cur * 4u;
				const float4 a = n[0], b = n[1], c = n[2], d = n[3];
				if (COUNT) n_nodes++;
				// ray_aabb (glsl:84-99) for both children, clamped to [t_min, best_t]
				const float l0x = fma_(a.x, ix, nrx), l1x = fma_(b.x, ix, nrx);
				const float l0y = fma_(a.y, iy, nry), l1y = fma_(b.y, iy, nry);
				const float l0z = fma_(a.z, iz, nrz), l1z = fma_(b.z, iz, nrz);
				const float r0x = fma_(c.x, ix, nrx), r1x = fma_(d.x, ix, nrx);
				const float r0y = fma_(c.y, iy, nry), r1y = fma_(d.y, iy, nry);
				const float r0z = fma_(c.z, iz, nrz), r1z = fma_(d.z, iz, nrz);
				const float tl = fmaxf(fmaxf(fminf(l0x, l1x), fminf(l0y, l1y)), fmaxf(fminf(l0z, l1z), r.t_min));
				const float tlx = fminf(fminf(fmaxf(l0x, l1x), fmaxf(l0y, l1y)), fminf(fmaxf(l0z, l1z), best_t));
				const float tr = fmaxf(fmaxf(fminf(r0x, r1x), fminf(r0y, r1y)), fmaxf(fminf(r0z, r1z), r.t_min));
				const float trx = fminf(fminf(fmaxf(r0x, r1x), fmaxf(r0y, r1y)), fminf(fmaxf(r0z, r1z), best_t));
				const bool hl = tl <= tlx, hr = tr <= trx;
				const uint32_t lref = __float_as_uint(a.w), rref = __float_as_uint(b.w);
				if (hl && hr) { // near child first, far child pushed (glsl:290-305)
					const bool left_near = tl < tr;
					cur = left_near ? lref : rref;
					lds_stack[sp] = left_near ? rref : lref; sp += MRT_WAVE;
					if (COUNT) { const uint32_t dpt = (sp - lane) / MRT_WAVE - wave * p.stack_depth; max_sp = dpt > max_sp ? dpt : max_sp; }
				} else if (hl) cur = lref;
				else if (hr) cur = rref;
				else { sp -= MRT_WAVE; cur = lds_stack[sp]; }
			}
			// ---- leaves: INTERSECT_LEAF, glsl:166-192 ----
			while (cur >= kLeafBit) {
				uint32_t slot = cur & 0x7FFFFFFFu;
				bool last;
				do {
					const float4 *t3 = hot + (size_t)slot * 3u;
					const float4 q0 = t3[0], q1 = t3[1], q2 = t3[2];
					last = (__float_as_uint(q2.w) & kLastInLeaf) != 0u;
					if ((__float_as_uint(q1.w) & p.query_mask) != 0u) {
						if (COUNT) n_tris++;
						// ray_triangle, glsl:105-131 == Triangle::intersect, src/core/triangle.h:56-105
						const float pvx = fma_(r.dy, q2.z, -(r.dz * q2.y));
						const float pvy = fma_(r.dz, q2.x, -(r.dx * q2.z));
						const float pvz = fma_(r.dx, q2.y, -(r.dy * q2.x));
						const float det = dot3(q1.x, q1.y, q1.z, pvx, pvy, pvz);
						if (!(__builtin_fabsf(det) < 1e-8f)) {
							const float inv_det = 1.0f / det;
							const float tvx = r.ox - q0.x, tvy = r.oy - q0.y, tvz = r.oz - q0.z;
							const float u = dot3(tvx, tvy, tvz, pvx, pvy, pvz) * inv_det;
							if (!(u < 0.0f || u > 1.0f)) {
								const float qvx = fma_(tvy, q1.z, -(tvz * q1.y));
								const float qvy = fma_(tvz, q1.x, -(tvx * q1.z));
								const float qvz = fma_(tvx, q1.y, -(tvy * q1.x));
								const float v = dot3(r.dx, r.dy, r.dz, qvx, qvy, qvz) * inv_det;
								if (!(v < 0.0f || u + v > 1.0f)) {
									const float t = dot3(q2.x, q2.y, q2.z, qvx, qvy, qvz) * inv_det;
									// glsl:124 accepts t_min <= t < best_t; an exact tie goes to the lower
									// triangle id so the answer does not depend on the visiting order
									const uint32_t id = __float_as_uint(q0.w);
									if (!(t < r.t_min) && (t < best_t || (t == best_t && best_slot != 0xFFFFFFFFu && id < best_id))) {
										best_t = t; best_u = u; best_v = v; best_slot = slot; best_id = id;
										if (ANY_HIT) last = true;
									}
								}
							}
						}
					}
					slot++;
				} while (!last);
				if (ANY_HIT && best_slot != 0xFFFFFFFFu) { cur = kSentinel; break; }
				sp -= MRT_WAVE; cur = lds_stack[sp];
			}
		}
	}

	// ---- result: glsl:322-327 ----
	finish_ray(p, ray_idx, r, best_t, best_u, best_v, best_slot);

	if (COUNT) {
		atomicAdd(&p.counters[kCntRays], 1ull);
		atomicAdd(&p.counters[kCntTris], (unsigned long long)n_tris);
		atomicAdd(&p.counters[kCntNodes], (unsigned long long)n_nodes);
		if (best_slot != 0xFFFFFFFFu) atomicAdd(&p.counters[kCntHits], 1ull);
		atomicMax(&p.counters[kCntMaxStack], (unsigned long long)max_sp);
		// one lane = one ray: every node step is a (divergent) node fetch, every test a triangle row
		atomicAdd(&p.counters[kCntWaveNodeFetch], (unsigned long long)n_nodes);
		atomicAdd(&p.counters[kCntWaveTriFetch], (unsigned long long)n_tris);
	}
}

#include "lane_persistent_kernel.h"
#include "packet_kernel.h"
#include "packet_asm_kernel.h"
#include "packet_rows_kernel.h"
#ifdef MRT_WITH_QUAD   // the four-wide packet walk: measured, not faster; build.py MRT_WITH_QUAD=1 compiles it in
#include "packet_quad_kernel.h"
#endif
#include "two_level_kernel.h"

// ---- the unified row array of packet_rows_kernel.h ------------------------------------------------------------
// rows[0, n_nodes) = the wide nodes with leaf refs rebased to row indices (0x80000000 | (n_nodes + first slot));
// rows[n_nodes + s] = triangle slot s as {v0,id | e1,layers | e2,flags | normal}: the hot and the cold row of the
// triangle in one 64-byte line, which is the reference's GPUTrianglePacked row (src/api/gpu_types.h:44-51).
__global__ __launch_bounds__(MRT_WG) void build_rows_kernel(const DevNode *nodes, const TriHot *hot, const TriCold *cold,
		uint32_t n_nodes, uint32_t n_tris, float4 *rows)
{
	const uint64_t g = (uint64_t)blockIdx.x * MRT_WG + threadIdx.x;
	if (g >= (uint64_t)n_nodes + n_tris) return;
	float4 *out = rows + g * 4u;
	if (g < n_nodes) {
		const float4 *n = reinterpret_cast<const float4 *>(nodes) + g * 4u;
		float4 a = n[0], b = n[1];
		uint32_t l = __float_as_uint(a.w), r = __float_as_uint(b.w);
		if (l >= kLeafBit) l = kLeafBit | (n_nodes + (l & 0x7FFFFFFFu));
		if (r >= kLeafBit) r = kLeafBit | (n_nodes + (r & 0x7FFFFFFFu));
		a.w = __uint_as_float(l); b.w = __uint_as_float(r);
		out[0] = a; out[1] = b; out[2] = n[2]; out[3] = n[3];
	} else {
		const uint64_t s = g - n_nodes;
		const float4 *t = reinterpret_cast<const float4 *>(hot) + s * 3u;
		out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
		out[3] = reinterpret_cast<const float4 *>(cold)[s];
	}
}

hipError_t launch_build_rows(const DevNode *nodes, const TriHot *hot, const TriCold *cold, uint32_t n_nodes, uint32_t n_tris,
		void *rows, hipStream_t stream)
{
	const uint64_t total = (uint64_t)n_nodes + n_tris;
	hipLaunchKernelGGL(build_rows_kernel, dim3((uint32_t)((total + MRT_WG - 1) / MRT_WG)), dim3(MRT_WG), 0, stream,
			nodes, hot, cold, n_nodes, n_tris, reinterpret_cast<float4 *>(rows));
	return hipGetLastError();
}

// ---- the row array of packet_quad_kernel.h: units of 64 bytes; 4-wide node i = the 128-byte row at unit 2i with
// its boxes as {min, max} pairs per axis and its refs rebased (inner -> 2 * index, leaf -> 0x80000000 |
// (2 * n_nodes4 + first slot)); triangle slot s = the
// 64-byte row at unit 2 * n_nodes4 + s ----
__global__ __launch_bounds__(MRT_WG) void build_rows4_kernel(const Dev4Node *nodes4, const TriHot *hot, const TriCold *cold,
		uint32_t n_nodes4, uint32_t n_tris, float4 *rows)
{
	const uint64_t g = (uint64_t)blockIdx.x * MRT_WG + threadIdx.x;
	if (g >= (uint64_t)n_nodes4 + n_tris) return;
	if (g < n_nodes4) {
		const Dev4Node &n = nodes4[g];
		float *out = reinterpret_cast<float *>(rows + g * 8u);
		for (int k = 0; k < 4; k++)
			for (int c = 0; c < 3; c++) { out[6 * k + 2 * c] = n.box[k][c]; out[6 * k + 2 * c + 1] = n.box[k][3 + c]; } // {min, max} per axis
		uint32_t *oref = reinterpret_cast<uint32_t *>(out) + 24;
		for (int i = 0; i < 4; i++) {
			const uint32_t ref = n.ref[i];
			oref[i] = ref == kSentinel ? ref : (ref >= kLeafBit ? (kLeafBit | (2u * n_nodes4 + (ref & 0x7FFFFFFFu))) : 2u * ref);
		}
		oref[4] = n.n_children; oref[5] = 0u; oref[6] = 0u; oref[7] = 0u;
	} else {
		const uint64_t s = g - n_nodes4;
		float4 *out = rows + ((uint64_t)2u * n_nodes4 + s) * 4u;
		const float4 *t = reinterpret_cast<const float4 *>(hot) + s * 3u;
		out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
		out[3] = reinterpret_cast<const float4 *>(cold)[s];
	}
}

hipError_t launch_build_rows4(const Dev4Node *nodes4, const TriHot *hot, const TriCold *cold, uint32_t n_nodes4, uint32_t n_tris,
		void *rows, hipStream_t stream)
{
	const uint64_t total = (uint64_t)n_nodes4 + n_tris;
	hipLaunchKernelGGL(build_rows4_kernel, dim3((uint32_t)((total + MRT_WG - 1) / MRT_WG)), dim3(MRT_WG), 0, stream,
			nodes4, hot, cold, n_nodes4, n_tris, reinterpret_cast<float4 *>(rows));
	return hipGetLastError();
}

// ---- standalone ray generation (mrt_generate_grid) ---------------------------------
__global__ __launch_bounds__(MRT_WG) void grid_rays_kernel(const TraceParams p, mrt_ray32 *out)
{
	const uint64_t g = (uint64_t)blockIdx.x * MRT_WG + threadIdx.x;
	if (g >= p.count) return;
	RayRegs r;
	grid_ray(p, (uint32_t)(g % p.grid_w), (uint32_t)(g / p.grid_w), r);
	float4 *q = reinterpret_cast<float4 *>(out) + g * 2u;
	float4 a, b;
	a.x = r.ox; a.y = r.oy; a.z = r.oz; a.w = r.t_max;
	b.x = r.dx; b.y = r.dy; b.z = r.dz; b.w = r.t_min;
	q[0] = a; q[1] = b;
}

// ---- hit tokens -> full hit records (mrt_expand_tokens) -----------------------------------
// A token names the winning triangle of a ray (leaf-order slot, 0xFFFFFFFF = miss).  Everything
// else in the record is a function of (ray, triangle): t, u, v come out of one Moller-Trumbore
// evaluation written exactly as in the traversal kernels, so the rebuilt record is the record
// the trace would have stored, bit for bit.  This is what lets a multi-GPU gather move 4 bytes
// per ray over xGMI instead of 32 (sharded.py): the root rebuilds the records from its own copy
// of the scene and the sender's camera.
__global__ __launch_bounds__(MRT_WG) void expand_tokens_kernel(const TraceParams p, const uint32_t *tokens)
{
	const uint64_t g = (uint64_t)blockIdx.x * MRT_WG + threadIdx.x;
	if (g >= p.count) return;
	RayRegs r;
	uint32_t px = 0, py = 0;
	if (p.in_fmt == IN_GRID) { px = (uint32_t)(g % p.grid_w); py = (uint32_t)(g / p.grid_w); }
	load_ray(p, g, px, py, r);
	const uint32_t slot = tokens[g];
	if (slot >= p.n_tris) { // miss (0xFFFFFFFF), or a token that is not from this scene: never read out of bounds
		store_hit(p, g, r, r.t_max, -1, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0u, slot);
		return;
	}
	const float4 *t3 = reinterpret_cast<const float4 *>(p.tri_hot) + (size_t)slot * 3u;
	const float4 q0 = t3[0], q1 = t3[1], q2 = t3[2];
	const float pvx = fma_(r.dy, q2.z, -(r.dz * q2.y));
	const float pvy = fma_(r.dz, q2.x, -(r.dx * q2.z));
	const float pvz = fma_(r.dx, q2.y, -(r.dy * q2.x));
	const float det = dot3(q1.x, q1.y, q1.z, pvx, pvy, pvz);
	const float inv_det = 1.0f / det;
	const float tvx = r.ox - q0.x, tvy = r.oy - q0.y, tvz = r.oz - q0.z;
	const float u = dot3(tvx, tvy, tvz, pvx, pvy, pvz) * inv_det;
	const float qvx = fma_(tvy, q1.z, -(tvz * q1.y));
	const float qvy = fma_(tvz, q1.x, -(tvx * q1.z));
	const float qvz = fma_(tvx, q1.y, -(tvy * q1.x));
	const float v = dot3(r.dx, r.dy, r.dz, qvx, qvy, qvz) * inv_det;
	const float t = dot3(q2.x, q2.y, q2.z, qvx, qvy, qvz) * inv_det;
	const float4 nn = reinterpret_cast<const float4 *>(p.tri_cold)[slot];
	store_hit(p, g, r, t, (int32_t)__float_as_uint(q0.w), u, v, nn.x, nn.y, nn.z, __float_as_uint(q1.w), slot);
}

// The same for a two-level scene: a token is {triangle slot in the mesh arrays, DevInstance row}.  The ray goes to the
// instance's mesh space with the kernels' own sequence (trace_two_level_kernel: o' = M o + t, d' = M d, every fused
// operation an explicit fma), Moller-Trumbore runs there, and finish_two_level_ray writes the record: flat id = the
// instance's id base + the mesh-local id, the instance's layer mask, normalize(basis * n), position on the world ray.
__global__ __launch_bounds__(MRT_WG) void expand_two_level_tokens_kernel(const TraceParams p, const uint2 *tokens)
{
	const uint64_t g = (uint64_t)blockIdx.x * MRT_WG + threadIdx.x;
	if (g >= p.count) return;
	RayRegs r;
	uint32_t px = 0, py = 0;
	if (p.in_fmt == IN_GRID) { px = (uint32_t)(g % p.grid_w); py = (uint32_t)(g / p.grid_w); }
	load_ray(p, g, px, py, r);
	const uint2 tok = tokens[g];
	const uint32_t slot = tok.x, inst = tok.y;
	if (slot >= p.n_tris || inst >= p.n_instances) { // miss, or a token that is not from this scene: never read out of bounds
		finish_two_level_ray(p, g, r, r.t_max, 0.0f, 0.0f, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u);
		return;
	}
	const float4 *row = reinterpret_cast<const float4 *>(p.instances) + (size_t)inst * 8u;
	const float4 m0 = row[0], m1 = row[1], m2 = row[2], meta = row[5];
	const float ox = fma_(m0.x, r.ox, fma_(m0.y, r.oy, fma_(m0.z, r.oz, m0.w)));
	const float oy = fma_(m1.x, r.ox, fma_(m1.y, r.oy, fma_(m1.z, r.oz, m1.w)));
	const float oz = fma_(m2.x, r.ox, fma_(m2.y, r.oy, fma_(m2.z, r.oz, m2.w)));
	const float dx = fma_(m0.x, r.dx, fma_(m0.y, r.dy, m0.z * r.dz));
	const float dy = fma_(m1.x, r.dx, fma_(m1.y, r.dy, m1.z * r.dz));
	const float dz = fma_(m2.x, r.dx, fma_(m2.y, r.dy, m2.z * r.dz));
	const float4 *t3 = reinterpret_cast<const float4 *>(p.tri_hot) + (size_t)slot * 3u;
	const float4 q0 = t3[0], q1 = t3[1], q2 = t3[2];
	const float pvx = fma_(dy, q2.z, -(dz * q2.y));
	const float pvy = fma_(dz, q2.x, -(dx * q2.z));
	const float pvz = fma_(dx, q2.y, -(dy * q2.x));
	const float det = dot3(q1.x, q1.y, q1.z, pvx, pvy, pvz);
	const float inv_det = 1.0f / det;
	const float tvx = ox - q0.x, tvy = oy - q0.y, tvz = oz - q0.z;
	const float u = dot3(tvx, tvy, tvz, pvx, pvy, pvz) * inv_det;
	const float qvx = fma_(tvy, q1.z, -(tvz * q1.y));
	const float qvy = fma_(tvz, q1.x, -(tvx * q1.z));
	const float qvz = fma_(tvx, q1.y, -(tvy * q1.x));
	const float v = dot3(dx, dy, dz, qvx, qvy, qvz) * inv_det;
	const float t = dot3(q2.x, q2.y, q2.z, qvx, qvy, qvz) * inv_det;
	finish_two_level_ray(p, g, r, t, u, v, slot, __float_as_uint(meta.z) + __float_as_uint(q0.w), inst);
}

// ---- row-width detection for coherent batches ------------------------------------------
// RayQuery::coherent (src/api/ray_query.h:69-76) says "these are primary camera rays" but
// the reference's cast_rays(rays, results, count) carries no image width, and a wave of 64
// consecutive rays is a 64x1 pixel strip.  One small block looks at the first rows: inside
// a row consecutive directions differ by one pixel step, at a row end they jump back by a
// whole row.  If the first two jumps sit at w and 2w and w x rows tiles the batch exactly,
// the trace kernel maps its lanes to 2^k x 64/2^k pixel tiles instead.  Purely a speed
// decision: any lane -> ray mapping gives the same results.
#define MRT_DETECT_THREADS 1024
#define MRT_DETECT_MAX_RAYS 65536u
// scratch layout (uint64 words): [0 .. 1023] jump bit masks, [1024] finished-block ticket
__device__ __forceinline__ void ray_dir(const void *rays, uint32_t in_fmt, uint64_t i, float &x, float &y, float &z)
{
	if (in_fmt == IN_HOST60) {
		const float *h = reinterpret_cast<const float *>(rays) + i * 15u;
		x = h[3]; y = h[4]; z = h[5];
	} else {
		const float4 b = reinterpret_cast<const float4 *>(rays)[i * 2u + 1u];
		x = b.x; y = b.y; z = b.z;
	}
}
__global__ __launch_bounds__(MRT_DETECT_THREADS) void detect_grid_kernel(const void *rays, uint32_t in_fmt, uint64_t count,
		uint32_t tile_w_log2, unsigned long long *scratch, uint32_t *out, uint32_t *host_out)
{
	__shared__ uint32_t first, second, is_last;
	const uint32_t m = (uint32_t)(count < (uint64_t)MRT_DETECT_MAX_RAYS ? count : (uint64_t)MRT_DETECT_MAX_RAYS);
	float ax, ay, az, bx, by, bz;
	ray_dir(rays, in_fmt, 0, ax, ay, az);
	ray_dir(rays, in_fmt, 1, bx, by, bz);
	const float step2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay) + (bz - az) * (bz - az);
	const float thr = 36.0f * step2; // a jump of more than 6 pixel steps
	// phase 1: every thread looks at one pair (i-1, i); one 64-bit jump mask per wave
	const uint32_t i = blockIdx.x * MRT_DETECT_THREADS + threadIdx.x;
	bool jump = false, wide = false;
	if (i >= 1 && i < m) {
		ray_dir(rays, in_fmt, i - 1, ax, ay, az);
		ray_dir(rays, in_fmt, i, bx, by, bz);
		const float d2 = (bx - ax) * (bx - ax) + (by - ay) * (by - ay) + (bz - az) * (bz - az);
		jump = d2 > thr;
		wide = !(d2 <= 0.01f); // neighbours more than ~6 degrees apart (or NaN): not what a packet wants
	}
	const unsigned long long mask = __ballot(jump);
	const unsigned long long wmask_dir = __ballot(wide);
	if ((threadIdx.x & 63u) == 0u) {
		__hip_atomic_store(&scratch[i >> 6], mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (wmask_dir) atomicAdd(&scratch[1025], (unsigned long long)__builtin_popcountll(wmask_dir));
	}
	// hand-off to the block that finishes last (agent-scope release / acquire, guide G16)
	__threadfence();
	__syncthreads();
	if (threadIdx.x == 0) {
		const unsigned long long t = atomicAdd(&scratch[1024], 1ull);
		is_last = (t == (unsigned long long)gridDim.x - 1ull) ? 1u : 0u;
		first = 0xFFFFFFFFu; second = 0xFFFFFFFFu;
	}
	__syncthreads();
	if (!is_last) return;
	__threadfence();
	// phase 2 (one block): first and second jump over the <= 1024 mask words
	const uint32_t words = (m + 63u) >> 6;
	unsigned long long wmask = 0ull;
	if (threadIdx.x < words) wmask = __hip_atomic_load(&scratch[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (wmask) atomicMin(&first, threadIdx.x * 64u + (uint32_t)__builtin_ctzll(wmask));
	__syncthreads();
	const uint32_t f = first;
	if (f != 0xFFFFFFFFu && threadIdx.x >= (f >> 6)) {
		unsigned long long rest = wmask;
		if (threadIdx.x == (f >> 6)) rest &= ~((2ull << (f & 63u)) - 1ull); // clear bits <= f
		if (rest) atomicMin(&second, threadIdx.x * 64u + (uint32_t)__builtin_ctzll(rest));
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t w = first, rows = 0, tiles_x = 0;
		const uint32_t tw = 1u << tile_w_log2, th = 64u >> tile_w_log2;
		bool ok = step2 > 0.0f && w != 0xFFFFFFFFu && w >= 16u && (w % tw) == 0u && (count % w) == 0ull;
		if (ok) {
			const uint64_t r = count / w;
			ok = r <= 0xFFFFFFFFull && (r % th) == 0ull && (2ull * w >= m || second == 2u * w);
			rows = (uint32_t)r; tiles_x = w >> tile_w_log2;
		}
		out[0] = ok ? w : 0u; out[1] = ok ? rows : 0u; out[2] = ok ? tiles_x : 0u;
		// "coherent" was only the caller's word: if more than 1 in 8 neighbouring rays point
		// somewhere else, the batch goes to the lane kernel (out[3] = 1) instead of packets
		const unsigned long long n_wide = __hip_atomic_load(&scratch[1025], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		// ... and so does a small batch in which no row width was found: packets of 64 consecutive rays are no match for one lane
		// per ray there (with a width, small batches go in pieces of 4 or 16 rays: api.hip quarter_small_grid)
		out[3] = (n_wide * 8ull > (unsigned long long)m || (!ok && count < 32768ull)) ? 1u : 0u;
		// the same four words to host-mapped memory: read by the host after it has waited for the stream
		if (host_out) { host_out[0] = out[0]; host_out[1] = out[1]; host_out[2] = out[2]; host_out[3] = out[3]; }
		scratch[1024] = 0ull; scratch[1025] = 0ull; // ticket / counter for the next launch (stream ordered)
	}
}

hipError_t launch_detect_grid(const void *rays, uint32_t in_fmt, uint64_t count, uint32_t tile_w_log2,
		unsigned long long *scratch, uint32_t *out, uint32_t *host_out, hipStream_t stream)
{
	const uint32_t m = (uint32_t)(count < (uint64_t)MRT_DETECT_MAX_RAYS ? count : (uint64_t)MRT_DETECT_MAX_RAYS);
	const uint32_t blocks = (m + MRT_DETECT_THREADS - 1) / MRT_DETECT_THREADS;
	hipLaunchKernelGGL(detect_grid_kernel, dim3(blocks), dim3(MRT_DETECT_THREADS), 0, stream, rays, in_fmt, count, tile_w_log2, scratch, out, host_out);
	return hipGetLastError();
}

// ---- Morton keys: src/dispatch/ray_sort.h:41-76 -------------------------------------
__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
	v &= 0x000003FFu;
	v = (v | (v << 16)) & 0x030000FFu;
	v = (v | (v << 8)) & 0x0300F00Fu;
	v = (v | (v << 4)) & 0x030C30C3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}
__device__ __forceinline__ uint32_t quant10(float v)
{
	float n = (v + 1.0f) * 0.5f;
	n = fmaxf(0.0f, fminf(1.0f, n));
	return (uint32_t)(n * 1023.0f);
}
__global__ __launch_bounds__(MRT_WG) void morton_keys_kernel(const void *rays, uint32_t in_fmt, uint64_t count,
		uint32_t *keys, uint32_t *index)
{
	const uint64_t g = (uint64_t)blockIdx.x * MRT_WG + threadIdx.x;
	if (g >= count) return;
	float dx, dy, dz;
	if (in_fmt == IN_HOST60) {
		const float *h = reinterpret_cast<const float *>(rays) + g * 15u;
		dx = h[3]; dy = h[4]; dz = h[5];
	} else {
		const float4 b = reinterpret_cast<const float4 *>(rays)[g * 2u + 1u];
		dx = b.x; dy = b.y; dz = b.z;
	}
	keys[g] = (spread10(quant10(dx)) << 2) | (spread10(quant10(dy)) << 1) | spread10(quant10(dz));
	if (index) index[g] = (uint32_t)g;
}

// ---- sort key for incoherent batches --------------------------------------------------
// The reference sorts by direction only (ray_sort.h:64-76), which groups nothing when the
// origins are scattered (config C4).  Results do not depend on the order, so the sort that
// feeds the lane kernel uses origin first: 6 bits per axis of the origin inside the scene
// bounds (Morton, 18 bits) above 4 bits per axis of the direction (Morton, 12 bits).
// Rays that start in the same ~1/64-of-the-scene cell and point the same way share a wave.
__global__ __launch_bounds__(MRT_WG) void origin_dir_keys_kernel(const void *rays, uint32_t in_fmt, uint64_t count,
		float bx, float by, float bz, float sx, float sy, float sz, uint32_t *keys, uint32_t *index)
{
	const uint64_t g = (uint64_t)blockIdx.x * MRT_WG + threadIdx.x;
	if (g >= count) return;
	float ox, oy, oz, dx, dy, dz;
	if (in_fmt == IN_HOST60) {
		const float *h = reinterpret_cast<const float *>(rays) + g * 15u;
		ox = h[0]; oy = h[1]; oz = h[2]; dx = h[3]; dy = h[4]; dz = h[5];
	} else {
		const float4 a = reinterpret_cast<const float4 *>(rays)[g * 2u], b = reinterpret_cast<const float4 *>(rays)[g * 2u + 1u];
		ox = a.x; oy = a.y; oz = a.z; dx = b.x; dy = b.y; dz = b.z;
	}
	auto q = [](float v, float lo, float scale, float top) { // clamp handles NaN / out-of-scene origins
		const float n = fmaxf(0.0f, fminf(top, (v - lo) * scale));
		return (uint32_t)n;
	};
	const uint32_t qx = q(ox, bx, sx, 63.0f), qy = q(oy, by, sy, 63.0f), qz = q(oz, bz, sz, 63.0f);
	const uint32_t ex = q(dx, -1.0f, 8.0f, 15.0f), ey = q(dy, -1.0f, 8.0f, 15.0f), ez = q(dz, -1.0f, 8.0f, 15.0f);
	const uint32_t ko = (spread10(qx) << 2) | (spread10(qy) << 1) | spread10(qz); // 18 bits
	const uint32_t kd = (spread10(ex) << 2) | (spread10(ey) << 1) | spread10(ez); // 12 bits
	keys[g] = (ko << 12) | kd;
	if (index) index[g] = (uint32_t)g;
}

hipError_t launch_origin_dir_keys(const void *rays, uint32_t in_fmt, uint64_t count, const float lo[3], const float hi[3],
		uint32_t *keys, uint32_t *index, hipStream_t stream)
{
	if (count == 0) return hipSuccess;
	const uint64_t blocks = (count + MRT_WG - 1) / MRT_WG;
	float s[3];
	for (int k = 0; k < 3; k++) { const float e = hi[k] - lo[k]; s[k] = e > 0.0f ? 64.0f / e : 0.0f; }
	hipLaunchKernelGGL(origin_dir_keys_kernel, dim3((uint32_t)blocks), dim3(MRT_WG), 0, stream, rays, in_fmt, count,
			lo[0], lo[1], lo[2], s[0], s[1], s[2], keys, index);
	return hipGetLastError();
}

// ---- launch wrappers (called from api.hip) -------------------------------------------
bool quad_kernel_built()
{
#ifdef MRT_WITH_QUAD
	return true;
#else
	return false;
#endif
}
// The instantiation the last launch_trace / launch_trace_persistent of this thread put on a stream, spelled as rocprofv3
// prints it ("trace_packet_rows_kernel<false, false, 2, 64, true>"): mrt_last_kernel_variant, which bench.py uses to
// accept committed counter passes only for the very kernel the run used.
static thread_local char g_variant[96] = "";
const char *last_trace_variant() { return g_variant; }
static void note_variant(const char *fmt, ...)
{
	va_list ap; va_start(ap, fmt); vsnprintf(g_variant, sizeof(g_variant), fmt, ap); va_end(ap);
}
#define MRT_B(x) ((x) ? "true" : "false")
#ifndef MRT_ROWS_WG_LARGE
#define MRT_ROWS_WG_LARGE MRT_WG // threads per workgroup of the rows kernel on large scenes
#endif
constexpr uint32_t kPrefetchMaxWaves = 10240u; // 1.25 rounds of the device's 8 192 wave slots
hipError_t launch_trace(const TraceParams &p_in, bool any_hit, bool count, hipStream_t stream)
{
	TraceParams p = p_in;
	// consecutive tiles per workgroup (tile_order 3): the rows kernel's workgroup holds rows_wg / 64 waves of two tiles each
	// (MRT_KERNEL_PACKET_DUAL) or 4 waves of one; every other kernel 4 waves of one tile
	p.tile_group = (p.kernel == MRT_KERNEL_PACKET_DUAL && p.row_array != nullptr) ? 2u * ((p.rows_wg == 64u ? 64u : (uint32_t)MRT_ROWS_WG_LARGE) / MRT_WAVE) : MRT_WG / MRT_WAVE;
	uint64_t threads;
	if (p.tile_sched != nullptr && p.sched_hdr != nullptr && p.n_slots_max != 0u) threads = (uint64_t)p.n_slots_max * p.tile_unit * 64u; // (slots past sched_hdr[2] have nothing to do)
	else if (p.lane_map == MAP_TILE8X8) {
		const uint32_t th = 64u >> p.tile_w_log2;
		threads = (uint64_t)p.tiles_x * ((p.rows + th - 1u) / th) * 64u * (p.quarter_all == 2u ? 16u : (p.quarter_all ? 4u : 1u));
	} else if (p.lane_map == MAP_LINEAR && p.sparse_lanes) threads = (p.count + p.sparse_lanes - 1u) / p.sparse_lanes * 64u;
	else threads = p.count * ((p.lane_map == MAP_AUTO && p.quarter_all) ? (p.quarter_all == 2u ? 16u : 4u) : 1u); // (a width found on the device: whole tiles, count / 64 of them)
	if (threads == 0) return hipSuccess;
	const uint64_t blocks = (threads + MRT_WG - 1) / MRT_WG;
	if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
	dim3 grid((uint32_t)blocks), wg(MRT_WG);
	if (p.kernel == MRT_KERNEL_TWO_LEVEL_PACKET) { // two-level scene, coherent batch: one wave per packet
		if (any_hit) hipLaunchKernelGGL((trace_two_level_packet_kernel<true>), grid, wg, p.extra_lds, stream, p);
		else hipLaunchKernelGGL((trace_two_level_packet_kernel<false>), grid, wg, p.extra_lds, stream, p);
		note_variant("trace_two_level_packet_kernel<%s>", MRT_B(any_hit));
		return hipGetLastError();
	}
	if (p.kernel == MRT_KERNEL_TWO_LEVEL) { // two-level scene: one lane per ray, per-lane LDS stack
		const size_t lds2 = (size_t)(MRT_WG / MRT_WAVE) * p.stack_depth * MRT_WAVE * sizeof(uint32_t);
		if (any_hit) hipLaunchKernelGGL((trace_two_level_kernel<true>), grid, wg, lds2, stream, p);
		else hipLaunchKernelGGL((trace_two_level_kernel<false>), grid, wg, lds2, stream, p);
		note_variant("trace_two_level_kernel<%s>", MRT_B(any_hit));
		return hipGetLastError();
	}
#ifdef MRT_WITH_QUAD
	if (p.kernel == MRT_KERNEL_PACKET_QUAD && p.row_array4 != nullptr) {
		// the 128-ray walk over 4-wide node rows: two packets per wave (half the waves)
		const uint64_t rblocks = (threads + 2u * MRT_WG - 1) / (2u * MRT_WG);
		dim3 rgrid((uint32_t)rblocks);
		if (count) {
			if (any_hit) hipLaunchKernelGGL((trace_packet_quad_kernel<true, true>), rgrid, wg, p.extra_lds, stream, p);
			else hipLaunchKernelGGL((trace_packet_quad_kernel<false, true>), rgrid, wg, p.extra_lds, stream, p);
		} else if (any_hit) hipLaunchKernelGGL((trace_packet_quad_kernel<true, false>), rgrid, wg, p.extra_lds, stream, p);
		else hipLaunchKernelGGL((trace_packet_quad_kernel<false, false>), rgrid, wg, p.extra_lds, stream, p);
		note_variant("trace_packet_quad_kernel<%s, %s>", MRT_B(any_hit), MRT_B(count));
		return hipGetLastError();
	}
#endif
	if ((p.kernel == MRT_KERNEL_PACKET_DUAL || p.kernel == MRT_KERNEL_PACKET_ROWS) && p.row_array != nullptr) {
		// the walk over the unified row array: one or two packets per wave (two: half the waves)
		const uint32_t packets = p.kernel == MRT_KERNEL_PACKET_DUAL ? 2u : 1u;
		const uint32_t rows_wg = packets == 2u && p.rows_wg == 64u ? 64u : (packets == 2u ? (uint32_t)MRT_ROWS_WG_LARGE : (uint32_t)MRT_WG);
		const uint64_t rblocks = (threads + packets * rows_wg - 1) / (packets * rows_wg);
		dim3 rgrid((uint32_t)rblocks), rwg(rows_wg);
#define MRT_LAUNCH_ROWS(A, C, N, W, F) hipLaunchKernelGGL((trace_packet_rows_kernel<A, C, N, W, F>), rgrid, rwg, p.extra_lds, stream, p)
#define MRT_LAUNCH_ROWS_AC(N, W, F)                                                                                    \
		do {                                                                                                        \
			if (count) { if (any_hit) MRT_LAUNCH_ROWS(true, true, N, W, F); else MRT_LAUNCH_ROWS(false, true, N, W, F); } \
			else { if (any_hit) MRT_LAUNCH_ROWS(true, false, N, W, F); else MRT_LAUNCH_ROWS(false, false, N, W, F); }   \
		} while (0)
		const bool cull = packets == 2u && (p.rows_cull == 1u || (p.rows_cull == 2u && p.in_fmt == IN_GRID));
		if (packets == 2u && rows_wg == 64u) { if (cull) MRT_LAUNCH_ROWS_AC(2, 64, true); else MRT_LAUNCH_ROWS_AC(2, 64, false); }
		else if (packets == 2u) { if (cull) MRT_LAUNCH_ROWS_AC(2, MRT_ROWS_WG_LARGE, true); else MRT_LAUNCH_ROWS_AC(2, MRT_ROWS_WG_LARGE, false); }
		else MRT_LAUNCH_ROWS_AC(1, MRT_WG, false);
#undef MRT_LAUNCH_ROWS_AC
#undef MRT_LAUNCH_ROWS
		note_variant("trace_packet_rows_kernel<%s, %s, %u, %u, %s>", MRT_B(any_hit), MRT_B(count), packets, rows_wg, MRT_B(cull));
		return hipGetLastError();
	}
	// scenes whose node offsets pass the asm loop's 32 bits use the C++ packet kernel
	if ((p.kernel == MRT_KERNEL_PACKET_ASM || p.kernel == MRT_KERNEL_PACKET_DUAL || p.kernel == MRT_KERNEL_PACKET_ROWS || p.kernel == MRT_KERNEL_PACKET_QUAD) && p.n_nodes < kAsmNodeLimit) {
		if (count) {
			if (any_hit) hipLaunchKernelGGL((trace_packet_asm_kernel<true, true>), grid, wg, p.extra_lds, stream, p);
			else hipLaunchKernelGGL((trace_packet_asm_kernel<false, true>), grid, wg, p.extra_lds, stream, p);
		} else {
			// the scalar-cache prefetch of both children: where the launch is about one round of waves (packet_asm_kernel.h)
			// (a scheduled launch covers the slots the list MAY use: what counts is the units, or the one round the fill rule makes of fewer)
			const uint64_t waves = p.tile_sched != nullptr && p.n_slots_max != 0u ? (p.n_units > 8192u ? p.n_units : (p.n_slots_max < 8192u ? p.n_slots_max : 8192u)) : threads / MRT_WAVE;
			const bool kpf = waves <= kPrefetchMaxWaves;
			if (kpf) { if (any_hit) hipLaunchKernelGGL((trace_packet_asm_kernel<true, false, true>), grid, wg, p.extra_lds, stream, p);
				else hipLaunchKernelGGL((trace_packet_asm_kernel<false, false, true>), grid, wg, p.extra_lds, stream, p); }
			else if (any_hit) hipLaunchKernelGGL((trace_packet_asm_kernel<true>), grid, wg, p.extra_lds, stream, p);
			else hipLaunchKernelGGL((trace_packet_asm_kernel<false>), grid, wg, p.extra_lds, stream, p);
			if (kpf) { note_variant("trace_packet_asm_kernel<%s, false, true>", MRT_B(any_hit)); return hipGetLastError(); }
		}
		note_variant("trace_packet_asm_kernel<%s, %s>", MRT_B(any_hit), MRT_B(count));
		return hipGetLastError();
	}
	if (p.kernel == MRT_KERNEL_PACKET || p.kernel == MRT_KERNEL_PACKET_ASM || p.kernel == MRT_KERNEL_PACKET_DUAL || p.kernel == MRT_KERNEL_PACKET_ROWS || p.kernel == MRT_KERNEL_PACKET_QUAD) {
		if (any_hit) {
			if (count) hipLaunchKernelGGL((trace_packet_kernel<true, true>), grid, wg, 0, stream, p);
			else hipLaunchKernelGGL((trace_packet_kernel<true, false>), grid, wg, 0, stream, p);
		} else {
			if (count) hipLaunchKernelGGL((trace_packet_kernel<false, true>), grid, wg, 0, stream, p);
			else hipLaunchKernelGGL((trace_packet_kernel<false, false>), grid, wg, 0, stream, p);
		}
		note_variant("trace_packet_kernel<%s, %s>", MRT_B(any_hit), MRT_B(count));
		return hipGetLastError();
	}
	const size_t lds = (size_t)(MRT_WG / MRT_WAVE) * p.stack_depth * MRT_WAVE * sizeof(uint32_t);
	if (any_hit) {
		if (count) hipLaunchKernelGGL((trace_lane_kernel<true, true>), grid, wg, lds, stream, p);
		else hipLaunchKernelGGL((trace_lane_kernel<true, false>), grid, wg, lds, stream, p);
	} else {
		if (count) hipLaunchKernelGGL((trace_lane_kernel<false, true>), grid, wg, lds, stream, p);
		else hipLaunchKernelGGL((trace_lane_kernel<false, false>), grid, wg, lds, stream, p);
	}
	note_variant("trace_lane_kernel<%s, %s>", MRT_B(any_hit), MRT_B(count));
	return hipGetLastError();
}

// Persistent lane kernel: `blocks` workgroups stay resident and pull rays from *next_ray.
hipError_t launch_trace_persistent(const TraceParams &p, unsigned long long *next_ray, uint32_t *overflow,
		uint32_t lds_depth, uint32_t refill, uint32_t leaf_wait, uint32_t blocks, bool any_hit, bool count, hipStream_t stream)
{
	if (p.count == 0 || blocks == 0) return hipSuccess;
	PersistParams q;
	q.next_ray = next_ray; q.overflow = overflow; q.overflow_stride = blocks * MRT_WG;
	q.lds_depth = lds_depth; q.refill = refill; q.leaf_wait = leaf_wait ? leaf_wait : 1u;
	// about 32 chunks per wave, between one wave's worth of rays and MRT_RAY_CHUNK
	const uint64_t per_wave = p.count / ((uint64_t)blocks * (MRT_WG / MRT_WAVE)) / 32u;
	q.chunk = per_wave >= MRT_RAY_CHUNK ? MRT_RAY_CHUNK : (per_wave <= MRT_WAVE ? MRT_WAVE : (uint32_t)(per_wave & ~63ull));
	const size_t lds = (size_t)(MRT_WG / MRT_WAVE) * lds_depth * MRT_WAVE * sizeof(uint32_t);
	const bool wide8 = p.kernel == MRT_KERNEL_LANE8_PERSISTENT && p.nodes8 != nullptr;
	const bool wide4 = p.kernel == MRT_KERNEL_LANE4_PERSISTENT && p.nodes4 != nullptr;
	if (p.kernel == MRT_KERNEL_TWO_LEVEL_PERSISTENT8 && p.nodes8 != nullptr && p.leaf_box != nullptr) {
		if (any_hit) hipLaunchKernelGGL((trace_lane_persistent_kernel<true, 8, true>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q);
		else hipLaunchKernelGGL((trace_lane_persistent_kernel<false, 8, true>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q);
	} else if (p.kernel == MRT_KERNEL_TWO_LEVEL_PERSISTENT || p.kernel == MRT_KERNEL_TWO_LEVEL_PERSISTENT8) {
		if (any_hit) hipLaunchKernelGGL((trace_lane_persistent_kernel<true, 2, true>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q);
		else hipLaunchKernelGGL((trace_lane_persistent_kernel<false, 2, true>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q);
	} else if (count) { // counting builds (flat scenes)
#define MRT_LP(A, W) hipLaunchKernelGGL((trace_lane_persistent_kernel<A, W, false, true>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q)
		if (wide8) { if (any_hit) MRT_LP(true, 8); else MRT_LP(false, 8); }
		else if (wide4) { if (any_hit) MRT_LP(true, 4); else MRT_LP(false, 4); }
		else { if (any_hit) MRT_LP(true, 2); else MRT_LP(false, 2); }
#undef MRT_LP
	} else if (wide8) {
		if (any_hit) hipLaunchKernelGGL((trace_lane_persistent_kernel<true, 8>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q);
		else hipLaunchKernelGGL((trace_lane_persistent_kernel<false, 8>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q);
	} else if (wide4) {
		if (any_hit) hipLaunchKernelGGL((trace_lane_persistent_kernel<true, 4>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q);
		else hipLaunchKernelGGL((trace_lane_persistent_kernel<false, 4>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q);
	} else {
		if (any_hit) hipLaunchKernelGGL((trace_lane_persistent_kernel<true, 2>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q);
		else hipLaunchKernelGGL((trace_lane_persistent_kernel<false, 2>), dim3(blocks), dim3(MRT_WG), lds, stream, p, q);
	}
	{
		const bool tl = p.kernel == MRT_KERNEL_TWO_LEVEL_PERSISTENT || p.kernel == MRT_KERNEL_TWO_LEVEL_PERSISTENT8;
		const int width = tl ? ((p.kernel == MRT_KERNEL_TWO_LEVEL_PERSISTENT8 && p.nodes8 != nullptr && p.leaf_box != nullptr) ? 8 : 2) : (wide8 ? 8 : (wide4 ? 4 : 2));
		note_variant("trace_lane_persistent_kernel<%s, %d, %s, %s>", MRT_B(any_hit), width, MRT_B(tl), MRT_B(count && !tl));
	}
	return hipGetLastError();
}

hipError_t launch_grid_rays(const TraceParams &p, mrt_ray32 *out, hipStream_t stream)
{
	if (p.count == 0) return hipSuccess;
	const uint64_t blocks = (p.count + MRT_WG - 1) / MRT_WG;
	hipLaunchKernelGGL(grid_rays_kernel, dim3((uint32_t)blocks), dim3(MRT_WG), 0, stream, p, out);
	return hipGetLastError();
}

hipError_t launch_expand_tokens(const TraceParams &p, const uint32_t *tokens, hipStream_t stream)
{
	if (p.count == 0) return hipSuccess;
	const uint64_t blocks = (p.count + MRT_WG - 1) / MRT_WG;
	if (p.instances != nullptr) // a two-level scene: 8-byte tokens {triangle slot, instance row}
		hipLaunchKernelGGL(expand_two_level_tokens_kernel, dim3((uint32_t)blocks), dim3(MRT_WG), 0, stream, p, reinterpret_cast<const uint2 *>(tokens));
	else hipLaunchKernelGGL(expand_tokens_kernel, dim3((uint32_t)blocks), dim3(MRT_WG), 0, stream, p, tokens);
	return hipGetLastError();
}

hipError_t launch_morton_keys(const void *rays, uint32_t in_fmt, uint64_t count, uint32_t *keys, uint32_t *index, hipStream_t stream)
{
	if (count == 0) return hipSuccess;
	const uint64_t blocks = (count + MRT_WG - 1) / MRT_WG;
	hipLaunchKernelGGL(morton_keys_kernel, dim3((uint32_t)blocks), dim3(MRT_WG), 0, stream, rays, in_fmt, count, keys, index);
	return hipGetLastError();
}

} // namespace mrt
