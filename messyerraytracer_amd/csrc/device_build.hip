// device_build.hip — builds the acceleration structure on the device (mrt_build_scene_device).
//
// The reference builds its BVH on the host (tinybvh::BVH::Build under RayScene::build,
// src/accel/ray_scene.h:62-86: 1.3 s for 1 M triangles on 8 cores, 10-15 s for 10 M) and
// uploads it (GPURayCaster::upload_scene, src/gpu/gpu_ray_caster.cpp:193-341); a scene that
// changes every frame cannot pay that.  This file is the device-side alternative (SURVEY.md
// 8(f) rank 4): an LBVH —
//   1. triangle boxes + scene bounds            (lbvh_bounds_kernel)
//   2. 63-bit Morton key of every box centre    (lbvh_keys_kernel)
//   3. radix sort of (key, triangle) pairs      (rocPRIM)
//   4. the binary radix tree over the keys      (lbvh_hierarchy_kernel; Karras 2012, "Maximizing
//      Parallelism in the Construction of BVHs, Octrees, and k-d Trees", sections 3-4)
//   5. boxes bottom-up, emitting DevNode rows   (lbvh_fit_kernel; one thread per leaf climbs, the
//      second thread to reach a node owns it)
//   6. leaf-ordered TriHot / TriCold rows       (lbvh_leaves_kernel)
// written straight into the layout the trace kernels read (mrt_internal.h).  One triangle per leaf.
// The tree is a valid BVH over the same triangles, so by the tie rule ("Arithmetic")
// casts against it return what casts against the host-built SAH tree return; it is a worse tree
// (more node visits per ray), which is the price of building it in a few milliseconds.
//
// Steps 4-5 have two forms: the radix tree (the default) and, with MRT_BUILD_PLOC, PLOC — Meister and Bittner 2018, "Parallel Locally-
// Ordered Clustering for Bounding Volume Hierarchy Construction": the Morton-sorted triangles are clusters on a
// line; every round each cluster finds, among its 8 neighbours to either side, the one whose union with it
// has the smallest surface area; clusters that choose each other merge into a node; a prefix sum compacts the
// line; until one cluster is left.  It is agglomerative clustering restricted to the Morton neighbourhood: the
// merges follow surface area instead of key bits, which is what the radix tree of step 4 cannot do.  About 30
// rounds of four small launches per million triangles (2.9 against 1.2 ms).  Measured on the 1 M-triangle soup of
// BASELINE config 3 (tools/bench_build.py, profiles/r02d_build_*): primary rays trace 1.05 x the host SAH tree's
// time on the PLOC tree (radius 8; 1.07 at 16, 1.14 at 32, 1.16 at 64), 1.06-1.07 x on the radix tree; incoherent
// rays 1.00 x against 0.93-0.94 x.  A uniform soup is the radix tree's best case (its cells are the cubes a SAH
// builder would cut), so the radix tree stays the default and PLOC is there for scenes with structure.
// Temporaries come from an arena the context owns (grown when a larger scene arrives, never shrunk): a rebuild
// allocates nothing but the scene's own arrays.
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h> // before rocprim: its texture_cache_iterator.hpp calls ::memset
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "mrt_internal.h"

namespace mrt {

namespace {

#define LBVH_WG 256

struct Box { float mn[3], mx[3]; };

// order-preserving float <-> uint (for atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float ord2f(uint32_t o) {
	const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
	float f; memcpy(&f, &u, 4); return f;
}

// One ulp outwards: v0 + e1 is a rounded sum, the true vertex may lie half an ulp outside it.
__device__ __forceinline__ float ulp_down(float f) { return f == 0.0f ? -FLT_MIN : __uint_as_float(__float_as_uint(f) + (f > 0.0f ? -1 : 1)); }
__device__ __forceinline__ float ulp_up(float f) { return f == 0.0f ? FLT_MIN : __uint_as_float(__float_as_uint(f) + (f > 0.0f ? 1 : -1)); }

// 1. box of every triangle (vertices v0, v0+e1, v0+e2 as the intersection test sees them) and the
//    bounds of all boxes: bounds[0..2] = min (ordered uint), bounds[3..5] = max.
//    Grid-stride over at most LBVH_BOUNDS_BLOCKS blocks, one atomic per block and component: atomics on
//    one address cost ~10 ns each chip-wide (one per wave made this kernel 1 ms per million triangles).
#define LBVH_BOUNDS_BLOCKS 1024u
__global__ __launch_bounds__(LBVH_WG) void lbvh_bounds_kernel(const mrt_tri64 *tris, uint32_t n, Box *boxes, uint32_t *bounds)
{
	__shared__ float part[LBVH_WG / 64][6];
	float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
	for (uint32_t i = blockIdx.x * LBVH_WG + threadIdx.x; i < n; i += gridDim.x * LBVH_WG) {
		const float4 *t = reinterpret_cast<const float4 *>(tris + i);
		const float4 a = t[0], b = t[1], c = t[2];
		const float v0[3] = { a.x, a.y, a.z }, e1[3] = { b.x, b.y, b.z }, e2[3] = { c.x, c.y, c.z };
		Box bx;
		for (int k = 0; k < 3; k++) {
			const float p1 = v0[k] + e1[k], p2 = v0[k] + e2[k];
			bx.mn[k] = ulp_down(fminf(v0[k], fminf(p1, p2)));
			bx.mx[k] = ulp_up(fmaxf(v0[k], fmaxf(p1, p2)));
			mn[k] = fminf(mn[k], bx.mn[k]); mx[k] = fmaxf(mx[k], bx.mx[k]);
		}
		boxes[i] = bx;
	}
	for (int k = 0; k < 3; k++) {
		float lo = mn[k], hi = mx[k];
		for (int off = 32; off > 0; off >>= 1) { lo = fminf(lo, __shfl_xor(lo, off)); hi = fmaxf(hi, __shfl_xor(hi, off)); }
		if ((threadIdx.x & 63u) == 0u) { part[threadIdx.x >> 6][k] = lo; part[threadIdx.x >> 6][3 + k] = hi; }
	}
	__syncthreads();
	if (threadIdx.x < 6u) {
		const bool is_min = threadIdx.x < 3u;
		float v = part[0][threadIdx.x];
		for (uint32_t w = 1; w < LBVH_WG / 64; w++) v = is_min ? fminf(v, part[w][threadIdx.x]) : fmaxf(v, part[w][threadIdx.x]);
		if (is_min) atomicMin(&bounds[threadIdx.x], f2ord(v)); else atomicMax(&bounds[threadIdx.x], f2ord(v));
	}
}

__device__ __forceinline__ uint64_t spread21(uint32_t v)
{
	uint64_t x = v & 0x1FFFFFu;
	x = (x | x << 32) & 0x1F00000000FFFFull;
	x = (x | x << 16) & 0x1F0000FF0000FFull;
	x = (x | x << 8) & 0x100F00F00F00F00Full;
	x = (x | x << 4) & 0x10C30C30C30C30C3ull;
	x = (x | x << 2) & 0x1249249249249249ull;
	return x;
}

// 2. key = 63-bit Morton code of the box centre on a 2^21 grid over the scene bounds
__global__ __launch_bounds__(LBVH_WG) void lbvh_keys_kernel(const Box *boxes, uint32_t n, const uint32_t *bounds, uint64_t *keys, uint32_t *index)
{
	const uint32_t i = blockIdx.x * LBVH_WG + threadIdx.x;
	if (i >= n) return;
	const Box b = boxes[i];
	uint32_t q[3];
	for (int k = 0; k < 3; k++) {
		const float lo = ord2f(bounds[k]), hi = ord2f(bounds[3 + k]);
		const float ext = hi - lo;
		const float c = 0.5f * b.mn[k] + 0.5f * b.mx[k];
		float u = ext > 0.0f ? (c - lo) / ext : 0.0f;
		u = fminf(fmaxf(u, 0.0f), 1.0f);
		const uint32_t g = (uint32_t)(u * 2097151.0f);
		q[k] = g > 2097151u ? 2097151u : g;
	}
	keys[i] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
	index[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j (equal keys: the positions
// themselves break the tie, so every leaf has a distinct code); -1 outside the array
__device__ __forceinline__ int lbvh_delta(const uint64_t *keys, int64_t n, int64_t i, int64_t j)
{
	if (j < 0 || j >= n) return -1;
	const uint64_t a = keys[i], b = keys[j];
	if (a != b) return __builtin_clzll(a ^ b);
	return 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);
}

// 4. Karras 2012, figure 4: internal node i covers the sorted leaves [min(i,j), max(i,j)] and splits
//    after position gamma.  child refs: < n-1 internal node, kLeafBit | position for a leaf.
__global__ __launch_bounds__(LBVH_WG) void lbvh_hierarchy_kernel(const uint64_t *keys, uint32_t n,
		uint32_t *left, uint32_t *right, uint32_t *parent_of_node, uint32_t *parent_of_leaf)
{
	const int64_t i = (int64_t)blockIdx.x * LBVH_WG + threadIdx.x;
	const int64_t nn = n;
	if (i >= nn - 1) return;
	const int d = lbvh_delta(keys, nn, i, i + 1) - lbvh_delta(keys, nn, i, i - 1) > 0 ? 1 : -1;
	const int dmin = lbvh_delta(keys, nn, i, i - d);
	int64_t lmax = 2;
	while (lbvh_delta(keys, nn, i, i + lmax * d) > dmin) lmax *= 2;
	int64_t l = 0;
	for (int64_t t = lmax / 2; t >= 1; t /= 2)
		if (lbvh_delta(keys, nn, i, i + (l + t) * d) > dmin) l += t;
	const int64_t j = i + l * d;
	const int dnode = lbvh_delta(keys, nn, i, j);
	int64_t s = 0, t = l;
	do {
		t = (t + 1) / 2;
		if (lbvh_delta(keys, nn, i, i + (s + t) * d) > dnode) s += t;
	} while (t > 1);
	const int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
	const int64_t lo = i < j ? i : j, hi = i < j ? j : i;
	if (lo == gamma) { left[i] = kLeafBit | (uint32_t)gamma; parent_of_leaf[gamma] = (uint32_t)i; }
	else { left[i] = (uint32_t)gamma; parent_of_node[gamma] = (uint32_t)i; }
	if (hi == gamma + 1) { right[i] = kLeafBit | (uint32_t)(gamma + 1); parent_of_leaf[gamma + 1] = (uint32_t)i; }
	else { right[i] = (uint32_t)(gamma + 1); parent_of_node[gamma + 1] = (uint32_t)i; }
	if (i == 0) parent_of_node[0] = 0xFFFFFFFFu;
}

// 5. One thread per leaf climbs towards the root.  The first thread to reach a node leaves (its
//    sibling subtree is not finished); the second owns the node: both child boxes are final, it
//    writes the DevNode row and the node's own box and climbs on.  The hand-off between the two
//    threads (any two CUs, any two XCDs; per-XCD L2s and per-CU L1s are not coherent) follows the
//    write-through form of the guide's inter-workgroup recipe: the handed-off words (a node's box
//    and depth) are stored and loaded ONLY by agent-scope relaxed atomics (sc1: they bypass the
//    non-coherent caches), the storing thread drains them (s_waitcnt vmcnt(0)) before it adds to
//    the parent's arrival counter, itself an agent-scope atomic.  (An acquire-release counter
//    with plain loads and stores is also correct and measured 3x slower here: every one of the
//    two million atomics then writes back and invalidates caches.)
typedef __attribute__((address_space(1))) unsigned long long lbvh_gu64;
typedef __attribute__((address_space(1))) unsigned int lbvh_gu32;
#define LBVH_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ Box load_box_coherent(const Box *src)
{
	lbvh_gu64 *p = (lbvh_gu64 *)src;
	const unsigned long long a = __hip_atomic_load(p, LBVH_RLX), b = __hip_atomic_load(p + 1, LBVH_RLX), c = __hip_atomic_load(p + 2, LBVH_RLX);
	Box r;
	r.mn[0] = __uint_as_float((uint32_t)a); r.mn[1] = __uint_as_float((uint32_t)(a >> 32));
	r.mn[2] = __uint_as_float((uint32_t)b); r.mx[0] = __uint_as_float((uint32_t)(b >> 32));
	r.mx[1] = __uint_as_float((uint32_t)c); r.mx[2] = __uint_as_float((uint32_t)(c >> 32));
	return r;
}
__device__ __forceinline__ void store_box_coherent(Box *dst, const Box &b)
{
	lbvh_gu64 *p = (lbvh_gu64 *)dst;
	__hip_atomic_store(p, (unsigned long long)__float_as_uint(b.mn[0]) | ((unsigned long long)__float_as_uint(b.mn[1]) << 32), LBVH_RLX);
	__hip_atomic_store(p + 1, (unsigned long long)__float_as_uint(b.mn[2]) | ((unsigned long long)__float_as_uint(b.mx[0]) << 32), LBVH_RLX);
	__hip_atomic_store(p + 2, (unsigned long long)__float_as_uint(b.mx[1]) | ((unsigned long long)__float_as_uint(b.mx[2]) << 32), LBVH_RLX);
}

// SAFE: the counter is acquire-release on top of the write-through payload (3x slower; the retry of a
// build whose verification pass found a stale hand-off -- never seen, but the fast form rests on
// measured behaviour of the part, not on an architectural guarantee).
template <bool SAFE>
__global__ __launch_bounds__(LBVH_WG) void lbvh_fit_kernel(uint32_t n, const Box *tri_boxes, const uint32_t *sorted_tri,
		const uint32_t *left, const uint32_t *right, const uint32_t *parent_of_node, const uint32_t *parent_of_leaf,
		uint32_t *arrivals, Box *node_box, uint32_t *node_depth, DevNode *nodes, uint32_t *max_depth)
{
	const uint32_t leaf = blockIdx.x * LBVH_WG + threadIdx.x;
	if (leaf >= n) return;
	uint32_t node = parent_of_leaf[leaf];
	for (;;) {
		const uint32_t before = SAFE ? __hip_atomic_fetch_add((lbvh_gu32 *)&arrivals[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
		                             : __hip_atomic_fetch_add((lbvh_gu32 *)&arrivals[node], 1u, LBVH_RLX);
		if (before == 0u) return;
		const uint32_t l = left[node], r = right[node];
		const Box lb = (l & kLeafBit) ? tri_boxes[sorted_tri[l & 0x7FFFFFFFu]] : load_box_coherent(&node_box[l]);
		const Box rb = (r & kLeafBit) ? tri_boxes[sorted_tri[r & 0x7FFFFFFFu]] : load_box_coherent(&node_box[r]);
		const uint32_t dl = (l & kLeafBit) ? 0u : __hip_atomic_load((lbvh_gu32 *)&node_depth[l], LBVH_RLX);
		const uint32_t dr = (r & kLeafBit) ? 0u : __hip_atomic_load((lbvh_gu32 *)&node_depth[r], LBVH_RLX);
		DevNode g;
		Box u;
		for (int k = 0; k < 3; k++) {
			g.lmin[k] = lb.mn[k]; g.lmax[k] = lb.mx[k]; g.rmin[k] = rb.mn[k]; g.rmax[k] = rb.mx[k];
			u.mn[k] = fminf(lb.mn[k], rb.mn[k]); u.mx[k] = fmaxf(lb.mx[k], rb.mx[k]);
		}
		g.left_ref = l; g.right_ref = r;
		g.left_count = (l & kLeafBit) ? 1u : 0u; g.right_count = (r & kLeafBit) ? 1u : 0u;
		nodes[node] = g;
		const uint32_t depth = (dl > dr ? dl : dr) + 1u;
		if (node == 0u) { *max_depth = depth; return; }
		store_box_coherent(&node_box[node], u);
		__hip_atomic_store((lbvh_gu32 *)&node_depth[node], depth, LBVH_RLX);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the box and depth are out before the parent's counter moves
		node = parent_of_node[node];
	}
}

// 5a. Verification, after the kernel boundary (plain loads are coherent now): every box a node row holds
//     for a child must be what that child's own row (or the leaf's triangle box) says, and depths must
//     add up.  A hand-off that delivered a stale box or depth in the fit shows here as a mismatch.
__global__ __launch_bounds__(LBVH_WG) void lbvh_verify_kernel(const DevNode *nodes, uint32_t n_nodes, const Box *tri_boxes,
		const uint32_t *sorted_tri, const uint32_t *node_depth, const uint32_t *max_depth, uint32_t *bad)
{
	const uint32_t b = blockIdx.x * LBVH_WG + threadIdx.x;
	if (b >= n_nodes) return;
	const DevNode g = nodes[b];
	bool ok = true;
	uint32_t deepest = 0u;
	for (int side = 0; side < 2; side++) {
		const uint32_t ref = side ? g.right_ref : g.left_ref;
		const float *mn = side ? g.rmin : g.lmin, *mx = side ? g.rmax : g.lmax;
		Box want;
		if (ref & kLeafBit) want = tri_boxes[sorted_tri[ref & 0x7FFFFFFFu]];
		else {
			const DevNode c = nodes[ref];
			for (int k = 0; k < 3; k++) { want.mn[k] = fminf(c.lmin[k], c.rmin[k]); want.mx[k] = fmaxf(c.lmax[k], c.rmax[k]); }
			const uint32_t d = node_depth[ref];
			deepest = d > deepest ? d : deepest;
		}
		for (int k = 0; k < 3; k++)
			ok = ok && __float_as_uint(mn[k]) == __float_as_uint(want.mn[k]) && __float_as_uint(mx[k]) == __float_as_uint(want.mx[k]);
	}
	const uint32_t mine = b == 0u ? *max_depth : node_depth[b];
	if (!ok || mine != deepest + 1u) atomicAdd(bad, 1u);
}

// ---- PLOC (steps 4-5, quality form) ----------------------------------------------------------------------
#define PLOC_R_MAX 64 // search radius on the line: positions to either side (run-time, <= this)

__device__ __forceinline__ float ploc_union_area(const Box &a, const Box &b)
{
	const float ex = fmaxf(a.mx[0], b.mx[0]) - fminf(a.mn[0], b.mn[0]);
	const float ey = fmaxf(a.mx[1], b.mx[1]) - fminf(a.mn[1], b.mn[1]);
	const float ez = fmaxf(a.mx[2], b.mx[2]) - fminf(a.mn[2], b.mn[2]);
	return ex * ey + ey * ez + ez * ex;
}

// the clusters of round 0: the sorted triangles
__global__ __launch_bounds__(LBVH_WG) void ploc_init_kernel(const Box *tri_boxes, const uint32_t *sorted_tri, uint32_t n, Box *cbox, uint32_t *cref, uint32_t *cdepth)
{
	const uint32_t i = blockIdx.x * LBVH_WG + threadIdx.x;
	if (i >= n) return;
	cbox[i] = tri_boxes[sorted_tri[i]]; cref[i] = kLeafBit | i; cdepth[i] = 0u;
}

// nearest neighbour on the line: the cluster within `radius` positions whose union with this one has the smallest
// half-area.  Ties go to the smaller i ^ j — a key both ends of a pair compute alike, and smallest for the
// neighbour that completes an even / odd pair, so a run of equal boxes pairs up completely in one round
// instead of forming a chain with one mutual pair (300 coincident triangles: 9 rounds, not 299).
__global__ __launch_bounds__(LBVH_WG) void ploc_nn_kernel(const Box *cbox, uint32_t m, int radius, uint32_t *nn)
{
	__shared__ Box sh[LBVH_WG + 2 * PLOC_R_MAX];
	const int PLOC_R = radius;
	const int64_t base = (int64_t)blockIdx.x * LBVH_WG - PLOC_R;
	for (uint32_t t = threadIdx.x; t < (uint32_t)(LBVH_WG + 2 * PLOC_R); t += LBVH_WG) {
		const int64_t g = base + t;
		if (g >= 0 && g < (int64_t)m) sh[t] = cbox[g];
	}
	__syncthreads();
	const uint32_t i = blockIdx.x * LBVH_WG + threadIdx.x;
	if (i >= m) return;
	const Box b = sh[threadIdx.x + PLOC_R];
	float best = FLT_MAX; uint32_t bj = i;
	for (int d = -PLOC_R; d <= PLOC_R; d++) {
		const int64_t j = (int64_t)i + d;
		if (d == 0 || j < 0 || j >= (int64_t)m) continue;
		const float a = ploc_union_area(b, sh[threadIdx.x + PLOC_R + d]);
		if (a < best || (a == best && (i ^ (uint32_t)j) < (i ^ bj))) { best = a; bj = (uint32_t)j; }
	}
	nn[i] = bj;
}

// mutual nearest neighbours merge: the lower of the two stays (and becomes the node), the upper one leaves the line.
// flags = keep | merged << 32, for one prefix sum of both
__global__ __launch_bounds__(LBVH_WG) void ploc_flags_kernel(const uint32_t *nn, uint32_t m, unsigned long long *flags)
{
	const uint32_t i = blockIdx.x * LBVH_WG + threadIdx.x;
	if (i >= m) return;
	const uint32_t j = nn[i];
	const bool mutual = j != i && nn[j] == i;
	const unsigned long long keep = (mutual && j < i) ? 0ull : 1ull, merged = (mutual && i < j) ? 1ull : 0ull;
	flags[i] = keep | (merged << 32);
}

// the next line: survivors at their prefix position; a merged pair becomes node node_base + its merge rank
// (creation order: children before parents, the root last) with both children's boxes in its row
__global__ __launch_bounds__(LBVH_WG) void ploc_merge_kernel(const Box *cbox, const uint32_t *cref, const uint32_t *cdepth, const uint32_t *nn,
		const unsigned long long *flags, const unsigned long long *pos, uint32_t m, uint32_t node_base,
		Box *obox, uint32_t *oref, uint32_t *odepth, DevNode *staged, uint32_t *staged_depth, uint32_t *totals)
{
	const uint32_t i = blockIdx.x * LBVH_WG + threadIdx.x;
	if (i >= m) return;
	const unsigned long long f = flags[i], p = pos[i];
	if (i == m - 1u) { totals[0] = (uint32_t)(p + f); totals[1] = (uint32_t)((p + f) >> 32); } // survivors, merges of this round
	if ((f & 1ull) == 0ull) return;
	const uint32_t at = (uint32_t)p;
	if ((f >> 32) == 0ull) { obox[at] = cbox[i]; oref[at] = cref[i]; odepth[at] = cdepth[i]; return; }
	const uint32_t j = nn[i], id = node_base + (uint32_t)(p >> 32);
	const Box lb = cbox[i], rb = cbox[j];
	const uint32_t l = cref[i], r = cref[j], dl = cdepth[i], dr = cdepth[j];
	DevNode g; Box u;
	for (int k = 0; k < 3; k++) {
		g.lmin[k] = lb.mn[k]; g.lmax[k] = lb.mx[k]; g.rmin[k] = rb.mn[k]; g.rmax[k] = rb.mx[k];
		u.mn[k] = fminf(lb.mn[k], rb.mn[k]); u.mx[k] = fmaxf(lb.mx[k], rb.mx[k]);
	}
	g.left_ref = l; g.right_ref = r;
	g.left_count = (l & kLeafBit) ? 1u : 0u; g.right_count = (r & kLeafBit) ? 1u : 0u;
	staged[id] = g;
	const uint32_t depth = (dl > dr ? dl : dr) + 1u;
	staged_depth[id] = depth;
	obox[at] = u; oref[at] = id; odepth[at] = depth;
}

// creation order -> root first: node id becomes n_nodes - 1 - id (the root, created last, is node 0)
__global__ __launch_bounds__(LBVH_WG) void ploc_finish_kernel(const DevNode *staged, const uint32_t *staged_depth, uint32_t n_nodes,
		DevNode *nodes, uint32_t *node_depth, uint32_t *max_depth)
{
	const uint32_t id = blockIdx.x * LBVH_WG + threadIdx.x;
	if (id >= n_nodes) return;
	DevNode g = staged[id];
	if (!(g.left_ref & kLeafBit)) g.left_ref = n_nodes - 1u - g.left_ref;
	if (!(g.right_ref & kLeafBit)) g.right_ref = n_nodes - 1u - g.right_ref;
	const uint32_t at = n_nodes - 1u - id;
	nodes[at] = g;
	node_depth[at] = staged_depth[id];
	if (at == 0u) *max_depth = staged_depth[id];
}

// 5b. 4-wide collapse for the incoherent-ray kernel (one 128-byte line per step): the rule of
//     scene_prep.cpp (start from a node's two children, keep opening the internal child with the
//     largest half-area until there are four).  Every binary node gets the 4-wide node it WOULD be
//     the root of, at its own index, so no allocation or top-down pass is needed; only the ones
//     reachable from node 0 are ever read (the others cost HBM capacity, not bandwidth).
__global__ __launch_bounds__(LBVH_WG) void lbvh_collapse4_kernel(const DevNode *nodes, uint32_t n_nodes, Dev4Node *nodes4)
{
	const uint32_t b = blockIdx.x * LBVH_WG + threadIdx.x;
	if (b >= n_nodes) return;
	float box[4][6]; uint32_t ref[4]; uint32_t n = 2;
	auto take = [&](const DevNode &g, uint32_t at_l, uint32_t at_r) {
		for (int k = 0; k < 3; k++) {
			box[at_l][k] = g.lmin[k]; box[at_l][3 + k] = g.lmax[k];
			box[at_r][k] = g.rmin[k]; box[at_r][3 + k] = g.rmax[k];
		}
		ref[at_l] = g.left_ref; ref[at_r] = g.right_ref;
	};
	take(nodes[b], 0, 1);
	while (n < 4) {
		int best = -1; float best_a = -1.0f;
		for (uint32_t i = 0; i < n; i++) {
			if (ref[i] >= kSentinel) continue; // a leaf
			const float e0 = box[i][3] - box[i][0], e1 = box[i][4] - box[i][1], e2 = box[i][5] - box[i][2];
			const float a = e0 * e1 + e1 * e2 + e2 * e0;
			if (a > best_a) { best_a = a; best = (int)i; }
		}
		if (best < 0) break;
		take(nodes[ref[best]], (uint32_t)best, n);
		n++;
	}
	Dev4Node out;
	for (uint32_t i = 0; i < 4; i++) {
		for (int k = 0; k < 6; k++) out.box[i][k] = i < n ? box[i][k] : __builtin_inff(); // unused slot: never hit
		out.ref[i] = i < n ? ref[i] : kSentinel;
	}
	out.n_children = n; out.pad[0] = out.pad[1] = out.pad[2] = 0u;
	nodes4[b] = out;
}

// 5c. 8-wide compressed collapse (Dev8Node), same scheme as 5b: every binary node gets the 8-wide
//     node it would be the root of, at its own index.  Quantisation exactly as on the host
//     (scene_prep.cpp): power-of-two grid step with one step of headroom, outward rounding, and every
//     quantised coordinate checked on the value the trace kernel will decode, fmaf(q, step, origin).
//     A box that cannot be put on a grid (non-finite) raises *bad: the scene then goes without this layout.
//     Also writes the exact box of every leaf (leaf_box, 8 floats per slot; each leaf once, by its binary
//     parent): the 8-wide walk checks a candidate hit against it (mrt_internal.h, Dev8Node).
__global__ __launch_bounds__(LBVH_WG) void lbvh_collapse8_kernel(const DevNode *nodes, uint32_t n_nodes, Dev8Node *nodes8, float *leaf_box, uint32_t *bad)
{
	const uint32_t b = blockIdx.x * LBVH_WG + threadIdx.x;
	if (b >= n_nodes) return;
	float box[8][6]; uint32_t ref[8]; uint32_t n = 2;
	auto take = [&](const DevNode &g, uint32_t at_l, uint32_t at_r) {
		for (int k = 0; k < 3; k++) {
			box[at_l][k] = g.lmin[k]; box[at_l][3 + k] = g.lmax[k];
			box[at_r][k] = g.rmin[k]; box[at_r][3 + k] = g.rmax[k];
		}
		ref[at_l] = g.left_ref; ref[at_r] = g.right_ref;
	};
	take(nodes[b], 0, 1);
	for (uint32_t i = 0; i < 2; i++)
		if (ref[i] >= kLeafBit) {
			float4 *lb = (float4 *)leaf_box + (size_t)(ref[i] & 0x7FFFFFFFu) * 2u;
			lb[0] = make_float4(box[i][0], box[i][1], box[i][2], 0.0f);
			lb[1] = make_float4(box[i][3], box[i][4], box[i][5], 0.0f);
		}
	while (n < 8) {
		int best = -1; float best_a = -1.0f;
		for (uint32_t i = 0; i < n; i++) {
			if (ref[i] >= kSentinel) continue; // a leaf
			const float e0 = box[i][3] - box[i][0], e1 = box[i][4] - box[i][1], e2 = box[i][5] - box[i][2];
			const float a = e0 * e1 + e1 * e2 + e2 * e0;
			if (a > best_a) { best_a = a; best = (int)i; }
		}
		if (best < 0) break;
		take(nodes[ref[best]], (uint32_t)best, n);
		n++;
	}
	Dev8Node out;
	memset(&out, 0, sizeof(out));
	out.n_children = (uint8_t)n;
	float step[3] = { 1.0f, 1.0f, 1.0f };
	bool ok = true;
	for (int a = 0; a < 3; a++) {
		float lo = box[0][a], hi = box[0][3 + a];
		for (uint32_t i = 1; i < n; i++) { lo = fminf(lo, box[i][a]); hi = fmaxf(hi, box[i][3 + a]); }
		out.org[a] = lo;
		const double ext = (double)hi - (double)lo;
		if (!(ext >= 0.0) || !(ext < 1.0e300)) { ok = false; continue; }
		int e = 1;
		if (ext > 0.0) { e = (int)ceil(log2(ext / 254.0)) + 127; if (e < 1) e = 1; if (e > 254) e = 254; }
		for (;;) {
			const float s = __uint_as_float((uint32_t)e << 23);
			if ((double)s * 254.0 >= ext || e >= 254) { step[a] = s; break; }
			e++;
		}
		out.exp[a] = (uint8_t)e;
	}
	for (uint32_t i = 0; i < 8; i++) {
		if (i >= n) { out.ref[i] = kSentinel; continue; }
		out.ref[i] = ref[i];
		for (int a = 0; a < 3 && ok; a++) {
			int ql = (int)floor(((double)box[i][a] - (double)out.org[a]) / (double)step[a]);
			ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
			while (ql > 0 && __builtin_fmaf((float)ql, step[a], out.org[a]) > box[i][a]) ql--;
			int qh = (int)ceil(((double)box[i][3 + a] - (double)out.org[a]) / (double)step[a]);
			qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
			while (qh < 255 && __builtin_fmaf((float)qh, step[a], out.org[a]) < box[i][3 + a]) qh++;
			if (__builtin_fmaf((float)ql, step[a], out.org[a]) > box[i][a] || __builtin_fmaf((float)qh, step[a], out.org[a]) < box[i][3 + a]) ok = false;
			out.qlo[a][i] = (uint8_t)ql; out.qhi[a][i] = (uint8_t)qh;
		}
	}
	if (!ok) atomicOr(bad, 1u);
	nodes8[b] = out;
}

// 6. triangle rows in leaf order (sorted position = slot); every leaf holds one triangle
__global__ __launch_bounds__(LBVH_WG) void lbvh_leaves_kernel(const mrt_tri64 *tris, uint32_t n, const uint32_t *sorted_tri, TriHot *hot, TriCold *cold)
{
	const uint32_t slot = blockIdx.x * LBVH_WG + threadIdx.x;
	if (slot >= n) return;
	const float4 *t = reinterpret_cast<const float4 *>(tris + sorted_tri[slot]);
	const float4 a = t[0], b = t[1], c = t[2], d = t[3];
	float4 *h = reinterpret_cast<float4 *>(hot + slot);
	h[0] = a; h[1] = b;
	float4 c2 = c; c2.w = __uint_as_float(kLastInLeaf);
	h[2] = c2;
	float4 nn = d; nn.w = 0.0f;
	reinterpret_cast<float4 *>(cold)[slot] = nn;
}

// Instances -> world-space triangles: the loop of RayTracerServer::_rebuild_scene
// (src/godot/raytracer_server.cpp:700-711).  Per vertex Transform3D::xform = basis row . v + origin
// (dot product summed left to right), then the Triangle ctor (src/core/triangle.h:41-51, the
// arithmetic of mrt_make_triangles), id = running triangle offset in instance order, layers = the
// mesh's layer mask.  blockIdx.y = instance.
__global__ __launch_bounds__(LBVH_WG) void flatten_instances_kernel(const float *verts9, const mrt_instance *instances,
		const uint32_t *first_out, mrt_tri64 *out)
{
	const mrt_instance in = instances[blockIdx.y];
	const uint32_t k = blockIdx.x * LBVH_WG + threadIdx.x;
	if (k >= in.n_tris) return;
	const float *src = verts9 + 9u * (size_t)(in.first_tri + k);
	float w[3][3];
	for (int v = 0; v < 3; v++) {
		const float x = src[3 * v], y = src[3 * v + 1], z = src[3 * v + 2];
		for (int r = 0; r < 3; r++)
			w[v][r] = ((in.basis[3 * r] * x + in.basis[3 * r + 1] * y) + in.basis[3 * r + 2] * z) + in.origin[r];
	}
	float e1[3], e2[3], n[3];
	for (int c = 0; c < 3; c++) { e1[c] = w[1][c] - w[0][c]; e2[c] = w[2][c] - w[0][c]; }
	n[0] = e1[1] * e2[2] - e1[2] * e2[1];
	n[1] = e1[2] * e2[0] - e1[0] * e2[2];
	n[2] = e1[0] * e2[1] - e1[1] * e2[0];
	const float l2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
	if (l2 == 0.0f) { n[0] = n[1] = n[2] = 0.0f; }
	else { const float l = __builtin_sqrtf(l2); n[0] /= l; n[1] /= l; n[2] /= l; }
	const uint32_t id = first_out[blockIdx.y] + k;
	float4 *dst = reinterpret_cast<float4 *>(out + id);
	float4 a, b, c, d;
	a.x = w[0][0]; a.y = w[0][1]; a.z = w[0][2]; a.w = __uint_as_float(id);
	b.x = e1[0]; b.y = e1[1]; b.z = e1[2]; b.w = __uint_as_float(in.layers);
	c.x = e2[0]; c.y = e2[1]; c.z = e2[2]; c.w = 0.0f;
	d.x = n[0]; d.y = n[1]; d.z = n[2]; d.w = 0.0f;
	dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
}

// A BLAS built on its own (refs from 0) into its place in a two-level scene's node array
__global__ __launch_bounds__(LBVH_WG) void offset_refs_kernel(DevNode *dst, const DevNode *src, uint32_t n, uint32_t node_base, uint32_t tri_base)
{
	const uint32_t i = blockIdx.x * LBVH_WG + threadIdx.x;
	if (i >= n) return;
	DevNode g = src[i];
	auto fix = [&](uint32_t ref) { return ref < kSentinel ? ref + node_base : (ref >= kLeafBit ? (kLeafBit | ((ref & 0x7FFFFFFFu) + tri_base)) : ref); };
	g.left_ref = fix(g.left_ref); g.right_ref = fix(g.right_ref);
	dst[i] = g;
}

// ... and its 8-wide layout (one Dev8Node per binary node, at the binary node's own offset)
__global__ __launch_bounds__(LBVH_WG) void offset_refs8_kernel(Dev8Node *dst, const Dev8Node *src, uint32_t n, uint32_t node_base, uint32_t tri_base)
{
	const uint32_t i = blockIdx.x * LBVH_WG + threadIdx.x;
	if (i >= n) return;
	Dev8Node g = src[i];
	for (int c = 0; c < 8; c++) {
		const uint32_t ref = g.ref[c];
		g.ref[c] = ref < kSentinel ? ref + node_base : (ref >= kLeafBit ? (kLeafBit | ((ref & 0x7FFFFFFFu) + tri_base)) : ref);
	}
	dst[i] = g;
}

} // namespace

hipError_t launch_offset_refs8(Dev8Node *dst, const Dev8Node *src, uint32_t n, uint32_t node_base, uint32_t tri_base, void *stream)
{
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(offset_refs8_kernel, dim3((n + LBVH_WG - 1) / LBVH_WG), dim3(LBVH_WG), 0, (hipStream_t)stream, dst, src, n, node_base, tri_base);
	return hipGetLastError();
}

hipError_t launch_offset_refs(DevNode *dst, const DevNode *src, uint32_t n, uint32_t node_base, uint32_t tri_base, void *stream)
{
	if (n == 0) return hipSuccess;
	hipLaunchKernelGGL(offset_refs_kernel, dim3((n + LBVH_WG - 1) / LBVH_WG), dim3(LBVH_WG), 0, (hipStream_t)stream, dst, src, n, node_base, tri_base);
	return hipGetLastError();
}
hipError_t launch_flatten_instances(const float *d_verts9, const mrt_instance *d_instances, const uint32_t *d_first_out,
		uint32_t n_instances, uint32_t max_tris_per_instance, mrt_tri64 *d_out, void *stream)
{
	if (n_instances == 0 || max_tris_per_instance == 0) return hipSuccess;
	for (uint32_t first = 0; first < n_instances; first += 65535u) { // grid.y is limited to 65535
		const uint32_t n = n_instances - first < 65535u ? n_instances - first : 65535u;
		hipLaunchKernelGGL(flatten_instances_kernel, dim3((max_tris_per_instance + LBVH_WG - 1) / LBVH_WG, n), dim3(LBVH_WG), 0,
				(hipStream_t)stream, d_verts9, d_instances + first, d_first_out + first, d_out);
	}
	return hipGetLastError();
}

#define DB_TRY(call)                                                                                    \
	do {                                                                                                 \
		hipError_t e_ = (call);                                                                          \
		if (e_ != hipSuccess) {                                                                          \
			std::snprintf(err, err_len, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
			cleanup();                                                                                   \
			return MRT_ERR_HIP;                                                                          \
		}                                                                                                \
	} while (0)

// Builds nodes / hot / cold (and nodes4 / nodes8 if wanted; hipMalloc'ed, owned by the caller on success) for
// the n >= 2 triangles at d_tris (device).  depth = stack entries a traversal can need (incl. the sentinel).
// Temporaries are carved from *arena (grown here if it is too small; owned by the caller, kept between builds).
int device_build_lbvh(const mrt_tri64 *d_tris, uint32_t n, bool want4, bool want8, bool safe_handoff, bool fast_lbvh, BuildArena *arena, void *stream_,
		DeviceBuildResult *out, char *err, size_t err_len)
{
	hipStream_t stream = (hipStream_t)stream_;
	DevNode *nodes = nullptr; TriHot *hot = nullptr; TriCold *cold = nullptr; Dev4Node *nodes4 = nullptr; Dev8Node *nodes8 = nullptr;
	float *leaf_box = nullptr;
	auto cleanup = [&] {
		if (nodes) (void)hipFree(nodes);
		if (nodes4) (void)hipFree(nodes4);
		if (nodes8) (void)hipFree(nodes8);
		if (leaf_box) (void)hipFree(leaf_box);
		if (hot) (void)hipFree(hot);
		if (cold) (void)hipFree(cold);
	};
	const size_t nn = n;
	// ---- the arena: one allocation, carved in 256-byte steps ----
	size_t sort_bytes = 0, scan_bytes = 0;
	DB_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, nn, 0, 63, stream));
	DB_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr, 0ull, nn, rocprim::plus<unsigned long long>(), stream));
	size_t need = 0;
	auto reserve = [&](size_t bytes) { const size_t at = need; need += (bytes + 255u) & ~(size_t)255u; return at; };
	const size_t o_boxes = reserve(nn * sizeof(Box)), o_scal = reserve(64), o_keys_a = reserve(nn * 8), o_keys_b = reserve(nn * 8),
			o_idx_a = reserve(nn * 4), o_idx_b = reserve(nn * 4), o_depth = reserve(nn * 4), o_sort = reserve(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
	// radix-tree form: left / right / parents / arrivals / node boxes; PLOC form: two cluster lines, nn, flags, positions, staged rows
	const size_t o_left = reserve(nn * 4), o_right = reserve(nn * 4), o_par_node = reserve(nn * 4), o_par_leaf = reserve(nn * 4),
			o_arrivals = reserve(nn * 4), o_node_box = reserve(nn * sizeof(Box));
	const size_t lbvh_end = need;
	need = o_left; // the two forms never run together: their temporaries share the space
	const size_t o_cbox0 = reserve(nn * sizeof(Box)), o_cbox1 = reserve(nn * sizeof(Box)), o_cref0 = reserve(nn * 4), o_cref1 = reserve(nn * 4),
			o_cdep0 = reserve(nn * 4), o_cdep1 = reserve(nn * 4), o_nn = reserve(nn * 4), o_flags = reserve(nn * 8), o_pos = reserve(nn * 8),
			o_staged = reserve(nn * sizeof(DevNode)), o_sdepth = reserve(nn * 4);
	if (need < lbvh_end) need = lbvh_end;
	if (arena->cap < need) {
		if (arena->ptr) { DB_TRY(hipStreamSynchronize(stream)); (void)hipFree(arena->ptr); arena->ptr = nullptr; arena->cap = 0; }
		if (hipMalloc(&arena->ptr, need) != hipSuccess) { arena->ptr = nullptr; std::snprintf(err, err_len, "device build: out of device memory"); return MRT_ERR_OOM; }
		arena->cap = need;
	}
	char *A = (char *)arena->ptr;
	Box *boxes = (Box *)(A + o_boxes);
	uint32_t *scal = (uint32_t *)(A + o_scal); // bounds[6], max_depth, 8-wide "bad" flag, verification failures, [10..11] PLOC round totals
	uint64_t *keys_a = (uint64_t *)(A + o_keys_a), *keys_b = (uint64_t *)(A + o_keys_b);
	uint32_t *idx_a = (uint32_t *)(A + o_idx_a), *idx_b = (uint32_t *)(A + o_idx_b);
	uint32_t *node_depth = (uint32_t *)(A + o_depth);
	void *sort_tmp = A + o_sort;
	bool ok = hipMalloc(&nodes, (nn - 1) * sizeof(DevNode)) == hipSuccess && hipMalloc(&hot, nn * sizeof(TriHot) + 16) == hipSuccess &&
			hipMalloc(&cold, nn * sizeof(TriCold)) == hipSuccess &&
			(!want4 || hipMalloc(&nodes4, (nn - 1) * sizeof(Dev4Node)) == hipSuccess) &&
			(!want8 || (hipMalloc(&nodes8, (nn - 1) * sizeof(Dev8Node)) == hipSuccess && hipMalloc(&leaf_box, nn * 32) == hipSuccess));
	if (!ok) { std::snprintf(err, err_len, "device build: out of device memory"); cleanup(); return MRT_ERR_OOM; }

	const uint32_t blocks = (uint32_t)((nn + LBVH_WG - 1) / LBVH_WG);
	const uint32_t init[16] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
	DB_TRY(hipMemcpyAsync(scal, init, sizeof(init), hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(lbvh_bounds_kernel, dim3(blocks < LBVH_BOUNDS_BLOCKS ? blocks : LBVH_BOUNDS_BLOCKS), dim3(LBVH_WG), 0, stream,
			d_tris, n, boxes, scal);
	hipLaunchKernelGGL(lbvh_keys_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, boxes, n, scal, keys_a, idx_a);
	DB_TRY(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_a, keys_b, idx_a, idx_b, nn, 0, 63, stream));
	hipLaunchKernelGGL(lbvh_leaves_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, d_tris, n, idx_b, hot, cold);
	uint32_t h[16];
	if (!fast_lbvh) {
		// ---- PLOC: rounds of nearest neighbour / flags / prefix sum / merge on the line of clusters ----
		Box *cbox[2] = { (Box *)(A + o_cbox0), (Box *)(A + o_cbox1) };
		uint32_t *cref[2] = { (uint32_t *)(A + o_cref0), (uint32_t *)(A + o_cref1) }, *cdep[2] = { (uint32_t *)(A + o_cdep0), (uint32_t *)(A + o_cdep1) };
		uint32_t *nnb = (uint32_t *)(A + o_nn);
		unsigned long long *flags = (unsigned long long *)(A + o_flags), *pos = (unsigned long long *)(A + o_pos);
		DevNode *staged = (DevNode *)(A + o_staged);
		uint32_t *sdepth = (uint32_t *)(A + o_sdepth);
		hipLaunchKernelGGL(ploc_init_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, boxes, idx_b, n, cbox[0], cref[0], cdep[0]);
		uint32_t m = n, node_base = 0u;
		int cur = 0;
		int ploc_radius = 8;
		if (const char *e = std::getenv("MRT_PLOC_RADIUS")) { const int r = std::atoi(e); if (r >= 1 && r <= PLOC_R_MAX) ploc_radius = r; } // tuning knob (tools/bench_build.py)
		for (uint32_t round = 0; m > 1u; round++) {
			if (round > 4096u) { std::snprintf(err, err_len, "device build: clustering did not converge"); cleanup(); return MRT_ERR_HIP; }
			const uint32_t mb = (m + LBVH_WG - 1) / LBVH_WG;
			hipLaunchKernelGGL(ploc_nn_kernel, dim3(mb), dim3(LBVH_WG), 0, stream, cbox[cur], m, ploc_radius, nnb);
			hipLaunchKernelGGL(ploc_flags_kernel, dim3(mb), dim3(LBVH_WG), 0, stream, nnb, m, flags);
			DB_TRY(rocprim::exclusive_scan(sort_tmp, scan_bytes, flags, pos, 0ull, (size_t)m, rocprim::plus<unsigned long long>(), stream));
			hipLaunchKernelGGL(ploc_merge_kernel, dim3(mb), dim3(LBVH_WG), 0, stream, cbox[cur], cref[cur], cdep[cur], nnb, flags, pos, m, node_base,
					cbox[cur ^ 1], cref[cur ^ 1], cdep[cur ^ 1], staged, sdepth, scal + 10);
			uint32_t totals[2];
			DB_TRY(hipMemcpyAsync(totals, scal + 10, sizeof(totals), hipMemcpyDeviceToHost, stream));
			DB_TRY(hipStreamSynchronize(stream));
			if (totals[1] == 0u || totals[0] + totals[1] != m) { std::snprintf(err, err_len, "device build: a clustering round made no progress"); cleanup(); return MRT_ERR_HIP; }
			m = totals[0]; node_base += totals[1];
			cur ^= 1;
		}
		if (node_base != n - 1u) { std::snprintf(err, err_len, "device build: clustering produced %u nodes for %u triangles", node_base, n); cleanup(); return MRT_ERR_HIP; }
		hipLaunchKernelGGL(ploc_finish_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, staged, sdepth, n - 1, nodes, node_depth, scal + 6);
		hipLaunchKernelGGL(lbvh_verify_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, nodes, n - 1, boxes, idx_b, node_depth, scal + 6, scal + 8);
		if (want4) hipLaunchKernelGGL(lbvh_collapse4_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, nodes, n - 1, nodes4);
		if (want8) hipLaunchKernelGGL(lbvh_collapse8_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, nodes, n - 1, nodes8, leaf_box, scal + 7);
		DB_TRY(hipGetLastError());
		DB_TRY(hipMemcpyAsync(h, scal, sizeof(h), hipMemcpyDeviceToHost, stream));
		DB_TRY(hipStreamSynchronize(stream));
		if (h[8] != 0u) { std::snprintf(err, err_len, "device build: the tree failed its verification pass (%u nodes)", h[8]); cleanup(); return MRT_ERR_HIP; }
	} else {
		uint32_t *left = (uint32_t *)(A + o_left), *right = (uint32_t *)(A + o_right), *par_node = (uint32_t *)(A + o_par_node), *par_leaf = (uint32_t *)(A + o_par_leaf);
		uint32_t *arrivals = (uint32_t *)(A + o_arrivals);
		Box *node_box = (Box *)(A + o_node_box);
		DB_TRY(hipMemsetAsync(arrivals, 0, nn * 4, stream));
		hipLaunchKernelGGL(lbvh_hierarchy_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, keys_b, n, left, right, par_node, par_leaf);
		// attempt 0: write-through hand-off; attempt 1: acquire-release hand-off, if the verification failed (or asked for)
		for (int attempt = safe_handoff ? 1 : 0;; attempt++) {
			if (attempt == 0)
				hipLaunchKernelGGL(lbvh_fit_kernel<false>, dim3(blocks), dim3(LBVH_WG), 0, stream, n, boxes, idx_b, left, right, par_node, par_leaf,
						arrivals, node_box, node_depth, nodes, scal + 6);
			else {
				DB_TRY(hipMemsetAsync(arrivals, 0, nn * 4, stream));
				DB_TRY(hipMemsetAsync(scal + 6, 0, 3 * sizeof(uint32_t), stream));
				hipLaunchKernelGGL(lbvh_fit_kernel<true>, dim3(blocks), dim3(LBVH_WG), 0, stream, n, boxes, idx_b, left, right, par_node, par_leaf,
						arrivals, node_box, node_depth, nodes, scal + 6);
			}
			hipLaunchKernelGGL(lbvh_verify_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, nodes, n - 1, boxes, idx_b, node_depth, scal + 6, scal + 8);
			if (want4) hipLaunchKernelGGL(lbvh_collapse4_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, nodes, n - 1, nodes4);
			if (want8) hipLaunchKernelGGL(lbvh_collapse8_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, nodes, n - 1, nodes8, leaf_box, scal + 7);
			DB_TRY(hipGetLastError());
			DB_TRY(hipMemcpyAsync(h, scal, sizeof(h), hipMemcpyDeviceToHost, stream));
			DB_TRY(hipStreamSynchronize(stream));
			if (h[8] == 0u) break;
			if (attempt >= 1) { std::snprintf(err, err_len, "device build: the tree failed its verification pass (%u nodes)", h[8]); cleanup(); return MRT_ERR_HIP; }
		}
	}
	out->nodes = nodes; out->hot = hot; out->cold = cold;
	out->n_nodes = n - 1; out->n_tris = n;
	out->depth = h[6] + 1u; // pending entries on the deepest path + the sentinel
	// 4-wide walk: every 4-wide node on a path leaves at most 3 entries pending and descends at least one binary level
	out->nodes4 = nodes4; out->stack4 = nodes4 ? 3u * h[6] + 1u : 0u;
	if (nodes8 && h[7] != 0u) { (void)hipFree(nodes8); (void)hipFree(leaf_box); nodes8 = nullptr; leaf_box = nullptr; } // a box that fits no grid: go without this layout
	out->nodes8 = nodes8; out->leaf_box = leaf_box; out->stack8 = nodes8 ? 7u * h[6] + 1u : 0u;
	for (int k = 0; k < 3; k++) { out->bounds_lo[k] = ord2f(h[k]); out->bounds_hi[k] = ord2f(h[3 + k]); }
	return MRT_OK;
}

} // namespace mrt
