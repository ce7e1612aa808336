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
#include <vector>
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
		if (ref & kLeafBit) { // the union of the leaf's triangle boxes (one triangle per leaf in the radix and PLOC forms)
			const uint32_t first = ref & 0x7FFFFFFFu, cnt = side ? g.right_count : g.left_count;
			want = tri_boxes[sorted_tri[first]];
			for (uint32_t j = 1; j < cnt; j++) {
				const Box t = tri_boxes[sorted_tri[first + j]];
				for (int k = 0; k < 3; k++) { want.mn[k] = fminf(want.mn[k], t.mn[k]); want.mx[k] = fmaxf(want.mx[k], t.mx[k]); }
			}
		} else {
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

// ---- binned SAH (steps 2-5, the host builder's tree; MRT_BUILD_SAH) -----------------------------------------
// tinybvh::BVH::Build (tiny_bvh.h:2332-2466; restated for the host in host/bvh_builder.cpp) as a level-synchronous
// device build: the decisions — 8 bins per axis over the node's box by primitive-box centre, the sweep for the smallest
// A_L N_L + A_R N_R, "split iff 1 + cost / A(node) < N", a side left empty = leaf — are the host's, operation for
// operation (explicitly rounded fp32, no contraction), so the topology is the tree the host builder makes of the same
// boxes.  What differs is how the work is laid out:
//   * node_of[i] is the open node triangle i is in; per level every triangle still in an open node goes down the split
//     its node got at the previous level and adds its box to the bins of the node it arrives in (3 axes x {min, max}
//     x 3 + a count; max on an order-preserving encoding whose empty value is 0, so the bins of a level start as a
//     memset).  21 global atomics per triangle and level made a million-triangle build 19 ms (25 G atomics/s is what
//     the L2 gives); so the bins are accumulated in LDS: while at most 8 nodes are open every block keeps all of them
//     (sah_bin_kernel), afterwards the active triangles are radix-sorted by the slot of their node each level
//     (closed leaves drop off the end of the list), a block's 256 triangles then cover a run of consecutive slots,
//     and it stores the bins of the slots it holds alone with plain coalesced stores, atomics only for the two it may
//     share with its neighbours (sah_bin_sorted_kernel);
//   * one thread per open node then sweeps its 24 bins, decides, and allocates the pair of children (an atomic counter:
//     creation order = breadth first; a child with one triangle is a leaf at once);
//   * when no node is open: subtree sizes bottom-up and preorder ranks top-down over the levels' id ranges give every
//     split the row index the host's depth-first numbering gives it (scene_prep.cpp) and every leaf its first slot;
//     a stable radix sort of (first slot, triangle) pairs is the leaf order.
// Node ids are creation order and nondeterministic; ranks, rows and leaf order are not.
struct SahNode { float mn[3]; uint32_t count; float mx[3]; uint32_t left; }; // left: id of the left child (right = left + 1); 0 = leaf / undecided
#define SAH_BINS 8
#define SAH_BIN_WORDS 8                         // ~ord(min x y z), ord(max x y z), count, unused
#define SAH_SLOT_WORDS (3 * SAH_BINS * SAH_BIN_WORDS)
#define SAH_LDS_SLOTS 8u
#define SAH_NO_SLOT 0xFFFFFFFFu
#define SAH_FAR 1e30f                           // BVH_FAR, tiny_bvh.h:140

__device__ __forceinline__ int sah_bin(float bmn, float bmx, float nmn, float nmx)
{
	const float rpd = __fdiv_rn((float)SAH_BINS, __fsub_rn(nmx, nmn));
	const float x = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(bmn, bmx), 0.5f), nmn), rpd);
	const int bi = (x > -2147483648.0f && x < 2147483648.0f) ? (int)x : (int)0x80000000u; // cvttss2si
	return bi > 0 ? (bi < SAH_BINS - 1 ? bi : SAH_BINS - 1) : 0;
}

// the top levels (at most SAH_LDS_SLOTS open nodes), triangle side: descend the split of the previous level, then bin
// into the node arrived at (if it is open); every block keeps all open slots in LDS
__global__ __launch_bounds__(LBVH_WG) void sah_bin_kernel(const Box *boxes, uint32_t n, uint32_t *node_of, const SahNode *nodes, const uint32_t *slot_of,
		const uint32_t *split_of, uint32_t *bins)
{
	__shared__ uint32_t sh[SAH_LDS_SLOTS * SAH_SLOT_WORDS];
	for (uint32_t t = threadIdx.x; t < SAH_LDS_SLOTS * SAH_SLOT_WORDS; t += LBVH_WG) sh[t] = 0u;
	__syncthreads();
	for (uint32_t i = blockIdx.x * LBVH_WG + threadIdx.x; i < n; i += gridDim.x * LBVH_WG) {
		uint32_t nd = node_of[i];
		SahNode N = nodes[nd];
		const Box b = boxes[i];
		if (N.left != 0u) {
			const uint32_t sp = split_of[nd], axis = sp >> 8, pos = sp & 0xFFu;
			nd = N.left + ((uint32_t)sah_bin(b.mn[axis], b.mx[axis], N.mn[axis], N.mx[axis]) > pos ? 1u : 0u);
			node_of[i] = nd;
			N = nodes[nd];
		}
		const uint32_t slot = slot_of[nd];
		if (slot == SAH_NO_SLOT) continue;
		uint32_t *base = sh + slot * SAH_SLOT_WORDS;
		for (int a = 0; a < 3; a++) {
			uint32_t *w = base + (a * SAH_BINS + sah_bin(b.mn[a], b.mx[a], N.mn[a], N.mx[a])) * SAH_BIN_WORDS;
			for (int k = 0; k < 3; k++) { atomicMax(w + k, ~f2ord(b.mn[k])); atomicMax(w + 3 + k, f2ord(b.mx[k])); }
			atomicAdd(w + 6, 1u);
		}
	}
	__syncthreads();
	for (uint32_t t = threadIdx.x; t < SAH_LDS_SLOTS * SAH_SLOT_WORDS; t += LBVH_WG) {
		const uint32_t v = sh[t];
		if (v == 0u) continue;
		if ((t & (SAH_BIN_WORDS - 1u)) == 6u) atomicAdd(bins + t, v); else atomicMax(bins + t, v);
	}
}

// the levels below, step 1: the active triangles (act[0 .. n_act), or all of them when act == nullptr) descend the split of the
// previous level; key = the slot of the node arrived at (SAH_NO_SLOT: a leaf, the triangle leaves the list at the sort)
__global__ __launch_bounds__(LBVH_WG) void sah_descend_kernel(const Box *boxes, const uint32_t *act, uint32_t n_act, uint32_t *node_of, const SahNode *nodes,
		const uint32_t *slot_of, const uint32_t *split_of, uint32_t *keys, uint32_t *vals)
{
	const uint32_t j = blockIdx.x * LBVH_WG + threadIdx.x;
	if (j >= n_act) return;
	const uint32_t i = act ? act[j] : j;
	uint32_t nd = node_of[i];
	const SahNode N = nodes[nd];
	if (N.left != 0u) {
		const uint32_t sp = split_of[nd], axis = sp >> 8, pos = sp & 0xFFu;
		nd = N.left + ((uint32_t)sah_bin(boxes[i].mn[axis], boxes[i].mx[axis], N.mn[axis], N.mx[axis]) > pos ? 1u : 0u);
		node_of[i] = nd;
	}
	keys[j] = slot_of[nd]; vals[j] = i;
}

// step 2 (after the sort by slot): a block's 256 list positions cover the consecutive slots first .. last; SAH_WIN of them at a time
// are accumulated in LDS and written out -- plain stores for a slot this block holds alone, atomics for `first` / `last` when the
// neighbouring block has triangles of the same node
#define SAH_WIN 64u
__global__ __launch_bounds__(LBVH_WG) void sah_bin_sorted_kernel(const Box *boxes, const uint32_t *act, const uint32_t *keys, uint32_t n_act, const uint32_t *node_of,
		const SahNode *nodes, uint32_t *bins)
{
	__shared__ uint32_t sh[SAH_WIN * SAH_SLOT_WORDS];
	const uint32_t p0 = blockIdx.x * LBVH_WG, p1 = p0 + LBVH_WG < n_act ? p0 + LBVH_WG : n_act, j = p0 + threadIdx.x;
	const uint32_t first = keys[p0], last = keys[p1 - 1u];
	const bool shared_first = p0 > 0u && keys[p0 - 1u] == first, shared_last = p1 < n_act && keys[p1] == last;
	uint32_t slot = SAH_NO_SLOT;
	Box b; SahNode N;
	if (j < p1) { const uint32_t i = act[j]; slot = keys[j]; b = boxes[i]; N = nodes[node_of[i]]; }
	for (uint32_t w0 = first; w0 <= last; w0 += SAH_WIN) {
		const uint32_t nw = last - w0 + 1u < SAH_WIN ? last - w0 + 1u : SAH_WIN;
		for (uint32_t t = threadIdx.x; t < nw * SAH_SLOT_WORDS; t += LBVH_WG) sh[t] = 0u;
		__syncthreads();
		if (slot != SAH_NO_SLOT && slot >= w0 && slot - w0 < nw) {
			uint32_t *base = sh + (slot - w0) * SAH_SLOT_WORDS;
			for (int a = 0; a < 3; a++) {
				uint32_t *w = base + (a * SAH_BINS + sah_bin(b.mn[a], b.mx[a], N.mn[a], N.mx[a])) * SAH_BIN_WORDS;
				for (int k = 0; k < 3; k++) { atomicMax(w + k, ~f2ord(b.mn[k])); atomicMax(w + 3 + k, f2ord(b.mx[k])); }
				atomicAdd(w + 6, 1u);
			}
		}
		__syncthreads();
		for (uint32_t t = threadIdx.x; t < nw * SAH_SLOT_WORDS; t += LBVH_WG) {
			const uint32_t s_ = w0 + t / SAH_SLOT_WORDS, v = sh[t];
			uint32_t *dst = bins + (size_t)w0 * SAH_SLOT_WORDS + t;
			if ((s_ == first && shared_first) || (s_ == last && shared_last)) {
				if (v != 0u) { if ((t & (SAH_BIN_WORDS - 1u)) == 6u) atomicAdd(dst, v); else atomicMax(dst, v); }
			} else *dst = v;
		}
		__syncthreads();
	}
}

__device__ __forceinline__ float sah_half_area(const float mn[3], const float mx[3])
{
	const float e0 = __fsub_rn(mx[0], mn[0]), e1 = __fsub_rn(mx[1], mn[1]), e2 = __fsub_rn(mx[2], mn[2]);
	return e0 < -SAH_FAR ? 0.0f : __fadd_rn(__fadd_rn(__fmul_rn(e0, e1), __fmul_rn(e1, e2)), __fmul_rn(e2, e0));
}
__device__ __forceinline__ void sah_bin_box(const uint32_t *w, float mn[3], float mx[3])
{
	for (int k = 0; k < 3; k++) {
		mn[k] = w[k] == 0u ? SAH_FAR : ord2f(~w[k]);
		mx[k] = w[3 + k] == 0u ? -SAH_FAR : ord2f(w[3 + k]);
	}
}

// one level, node side.  ctr[0] = next free node id, ctr[1] = open nodes of the next level, ctr[2] = the triangles in them
__global__ __launch_bounds__(64) void sah_split_kernel(SahNode *nodes, const uint32_t *open, uint32_t n_open, const uint32_t *bins, uint32_t *slot_of,
		uint32_t *split_of, uint32_t *ctr, uint32_t *open_next, const uint32_t *bounds)
{
	// the block's 64 slots are staged through LDS (one coalesced read of 48 KB instead of 64 lanes striding 768 bytes apart: the
	// widest level of a million-triangle build took 0.43 ms); odd stride: lane l's word w sits in bank (l + w) & 31
	__shared__ uint32_t sh[64 * (SAH_SLOT_WORDS + 1)];
	const uint32_t t0 = blockIdx.x * 64u, nt = n_open - t0 < 64u ? n_open - t0 : 64u;
	for (uint32_t k = threadIdx.x; k < nt * SAH_SLOT_WORDS; k += 64u)
		sh[(k / SAH_SLOT_WORDS) * (SAH_SLOT_WORDS + 1) + k % SAH_SLOT_WORDS] = bins[(size_t)t0 * SAH_SLOT_WORDS + k];
	__syncthreads();
	const uint32_t t = t0 + threadIdx.x;
	if (t >= n_open) return;
	const uint32_t id = open[t];
	const SahNode N = nodes[id];
	const uint32_t *slot = sh + threadIdx.x * (SAH_SLOT_WORDS + 1);
	float ext[3];
	for (int k = 0; k < 3; k++) ext[k] = __fsub_rn(N.mx[k], N.mn[k]);
	const float inv_area = __fdiv_rn(1.0f, __fadd_rn(__fadd_rn(__fmul_rn(ext[0], ext[1]), __fmul_rn(ext[1], ext[2])), __fmul_rn(ext[2], ext[0])));
	float split_cost = SAH_FAR;
	int best_axis = 0, best_pos = 0;
	for (int a = 0; a < 3; a++) {
		const float min_dim = __fmul_rn(__fsub_rn(ord2f(bounds[3 + a]), ord2f(bounds[a])), 1e-20f);
		if (!(ext[a] > min_dim)) continue;
		const uint32_t *ax = slot + a * SAH_BINS * SAH_BIN_WORDS;
		float lmn[3] = { SAH_FAR, SAH_FAR, SAH_FAR }, lmx[3] = { -SAH_FAR, -SAH_FAR, -SAH_FAR };
		float rmn[3] = { SAH_FAR, SAH_FAR, SAH_FAR }, rmx[3] = { -SAH_FAR, -SAH_FAR, -SAH_FAR };
		float cost_below[SAH_BINS - 1], cost_above[SAH_BINS - 1];
		uint32_t ln = 0, rn = 0;
		for (int i = 0; i < SAH_BINS - 1; i++) {
			float bmn[3], bmx[3];
			sah_bin_box(ax + i * SAH_BIN_WORDS, bmn, bmx);
			for (int k = 0; k < 3; k++) { lmn[k] = fminf(lmn[k], bmn[k]); lmx[k] = fmaxf(lmx[k], bmx[k]); }
			sah_bin_box(ax + (SAH_BINS - 1 - i) * SAH_BIN_WORDS, bmn, bmx);
			for (int k = 0; k < 3; k++) { rmn[k] = fminf(rmn[k], bmn[k]); rmx[k] = fmaxf(rmx[k], bmx[k]); }
			ln += ax[i * SAH_BIN_WORDS + 6]; rn += ax[(SAH_BINS - 1 - i) * SAH_BIN_WORDS + 6];
			cost_below[i] = ln == 0u ? SAH_FAR : __fmul_rn(sah_half_area(lmn, lmx), (float)ln);
			cost_above[SAH_BINS - 2 - i] = rn == 0u ? SAH_FAR : __fmul_rn(sah_half_area(rmn, rmx), (float)rn);
		}
		for (int i = 0; i < SAH_BINS - 1; i++) {
			const float c = __fadd_rn(cost_below[i], cost_above[i]);
			if (c < split_cost) { split_cost = c; best_axis = a; best_pos = i; }
		}
	}
	split_cost = __fadd_rn(1.0f, __fmul_rn(__fmul_rn(1.0f, inv_area), split_cost));
	uint32_t lc = 0;
	const uint32_t *ax = slot + best_axis * SAH_BINS * SAH_BIN_WORDS;
	for (int i = 0; i <= best_pos; i++) lc += ax[i * SAH_BIN_WORDS + 6];
	const uint32_t rc = N.count - lc;
	if (split_cost >= (float)N.count || lc == 0u || rc == 0u) { slot_of[id] = SAH_NO_SLOT; return; } // stays a leaf
	SahNode L, R;
	for (int k = 0; k < 3; k++) { L.mn[k] = R.mn[k] = SAH_FAR; L.mx[k] = R.mx[k] = -SAH_FAR; }
	for (int i = 0; i < SAH_BINS; i++) {
		float bmn[3], bmx[3];
		sah_bin_box(ax + i * SAH_BIN_WORDS, bmn, bmx);
		SahNode &D = i <= best_pos ? L : R;
		for (int k = 0; k < 3; k++) { D.mn[k] = fminf(D.mn[k], bmn[k]); D.mx[k] = fmaxf(D.mx[k], bmx[k]); }
	}
	L.count = lc; R.count = rc; L.left = R.left = 0u;
	const uint32_t pair = atomicAdd(&ctr[0], 2u);
	nodes[pair] = L; nodes[pair + 1u] = R;
	nodes[id].left = pair;
	split_of[id] = ((uint32_t)best_axis << 8) | (uint32_t)best_pos;
	for (uint32_t side = 0; side < 2u; side++) {
		const uint32_t c = pair + side, cnt = side ? rc : lc;
		if (cnt >= 2u) { const uint32_t s = atomicAdd(&ctr[1], 1u); open_next[s] = c; slot_of[c] = s; atomicAdd(&ctr[2], cnt); }
		else slot_of[c] = SAH_NO_SLOT;
	}
}

__global__ void sah_root_kernel(SahNode *nodes, uint32_t n, const uint32_t *bounds, uint32_t *slot_of, uint32_t *open, uint32_t *rank, uint32_t *first, uint32_t *ctr)
{
	ctr[0] = 1u;
	SahNode r;
	for (int k = 0; k < 3; k++) { r.mn[k] = ord2f(bounds[k]); r.mx[k] = ord2f(bounds[3 + k]); }
	r.count = n; r.left = 0u;
	nodes[0] = r; slot_of[0] = 0u; open[0] = 0u; rank[0] = 0u; first[0] = 0u;
}

// nodes [lo, hi) of one depth, deepest first: splits in the subtree, height
__global__ __launch_bounds__(LBVH_WG) void sah_size_kernel(const SahNode *nodes, uint32_t lo, uint32_t hi, uint32_t *size, uint32_t *height)
{
	const uint32_t id = lo + blockIdx.x * LBVH_WG + threadIdx.x;
	if (id >= hi) return;
	const uint32_t l = nodes[id].left;
	if (l == 0u) { size[id] = 0u; height[id] = 0u; return; }
	size[id] = 1u + size[l] + size[l + 1u];
	const uint32_t hl = height[l], hr = height[l + 1u];
	height[id] = 1u + (hl > hr ? hl : hr);
}

// nodes [lo, hi) of one depth, root first: a split's row (its preorder rank), its children's ranks and first slots
__global__ __launch_bounds__(LBVH_WG) void sah_emit_kernel(const SahNode *nodes, uint32_t lo, uint32_t hi, const uint32_t *size, const uint32_t *height,
		uint32_t *rank, uint32_t *first, DevNode *rows, uint32_t *node_depth, uint32_t *max_depth)
{
	const uint32_t id = lo + blockIdx.x * LBVH_WG + threadIdx.x;
	if (id >= hi) return;
	const SahNode N = nodes[id];
	if (N.left == 0u) return;
	const uint32_t l = N.left, r = l + 1u, rk = rank[id], f = first[id];
	const SahNode L = nodes[l], R = nodes[r];
	rank[l] = rk + 1u; rank[r] = rk + 1u + size[l];
	first[l] = f; first[r] = f + L.count;
	DevNode g;
	for (int k = 0; k < 3; k++) { g.lmin[k] = L.mn[k]; g.lmax[k] = L.mx[k]; g.rmin[k] = R.mn[k]; g.rmax[k] = R.mx[k]; }
	g.left_ref = L.left ? rk + 1u : (kLeafBit | f); g.left_count = L.left ? 0u : L.count;
	g.right_ref = R.left ? rk + 1u + size[l] : (kLeafBit | (f + L.count)); g.right_count = R.left ? 0u : R.count;
	rows[rk] = g;
	node_depth[rk] = height[id];
	if (id == 0u) *max_depth = height[id];
}

// leaf order: key = first slot of the triangle's leaf.  wrap_lc != 0: the root stayed a leaf and is wrapped in a row whose
// two children are its halves (scene_prep.cpp does the same to a host tree): the first wrap_lc triangles, and the rest
__global__ __launch_bounds__(LBVH_WG) void sah_keys_kernel(const uint32_t *node_of, const uint32_t *first, uint32_t n, uint32_t wrap_lc, uint32_t *keys, uint32_t *index)
{
	const uint32_t i = blockIdx.x * LBVH_WG + threadIdx.x;
	if (i >= n) return;
	keys[i] = wrap_lc ? (i < wrap_lc ? 0u : wrap_lc) : first[node_of[i]];
	index[i] = i;
}
__global__ void sah_wrap_kernel(const uint32_t *bounds, uint32_t n, uint32_t lc, DevNode *rows, uint32_t *node_depth, uint32_t *max_depth)
{
	DevNode g;
	for (int k = 0; k < 3; k++) { g.lmin[k] = g.rmin[k] = ord2f(bounds[k]); g.lmax[k] = g.rmax[k] = ord2f(bounds[3 + k]); }
	g.left_ref = kLeafBit; g.left_count = lc; g.right_ref = kLeafBit | lc; g.right_count = n - lc;
	rows[0] = g; node_depth[0] = 1u; *max_depth = 1u;
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
//    (leaf_key == nullptr), or the slots with one key form a leaf (the binned-SAH form: key = the leaf's first slot)
__global__ __launch_bounds__(LBVH_WG) void lbvh_leaves_kernel(const mrt_tri64 *tris, uint32_t n, const uint32_t *sorted_tri, const uint32_t *leaf_key,
		TriHot *hot, TriCold *cold)
{
	const uint32_t slot = blockIdx.x * LBVH_WG + threadIdx.x;
	if (slot >= n) return;
	const bool last = leaf_key == nullptr || slot + 1u == n || leaf_key[slot + 1u] != leaf_key[slot];
	const float4 *t = reinterpret_cast<const float4 *>(tris + sorted_tri[slot]);
	const float4 a = t[0], b = t[1], c = t[2], d = t[3];
	float4 *h = reinterpret_cast<float4 *>(hot + slot);
	h[0] = a; h[1] = b;
	float4 c2 = c; c2.w = __uint_as_float(last ? kLastInLeaf : 0u);
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
int device_build_lbvh(const mrt_tri64 *d_tris, uint32_t n, bool want4, bool want8, bool safe_handoff, int form, BuildArena *arena, void *stream_,
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
	const size_t ploc_end = need;
	need = o_left;
	// binned-SAH form: 2 n nodes with their side arrays, the open lists, node_of, the bins of the widest level (<= n / 2 open nodes)
	const size_t o_snodes = reserve(2 * nn * sizeof(SahNode)), o_slot = reserve(2 * nn * 4), o_split = reserve(2 * nn * 4), o_size = reserve(2 * nn * 4),
			o_height = reserve(2 * nn * 4), o_rank = reserve(2 * nn * 4), o_first = reserve(2 * nn * 4), o_open0 = reserve(nn * 4), o_open1 = reserve(nn * 4),
			o_node_of = reserve(nn * 4), o_bins = reserve((nn / 2 + SAH_LDS_SLOTS) * SAH_SLOT_WORDS * 4);
	if (form != 2) need = o_left; // (the bins are 384 bytes per triangle: only a build that asks for this form pays for them)
	if (need < lbvh_end) need = lbvh_end;
	if (need < ploc_end) need = ploc_end;
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
	if (form != 2) {
		hipLaunchKernelGGL(lbvh_keys_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, boxes, n, scal, keys_a, idx_a);
		DB_TRY(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_a, keys_b, idx_a, idx_b, nn, 0, 63, stream));
	}
	if (form != 2) hipLaunchKernelGGL(lbvh_leaves_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, d_tris, n, idx_b, (const uint32_t *)nullptr, hot, cold);
	uint32_t h[16];
	uint32_t n_rows = n - 1u;
	if (form == 2) {
		// ---- binned SAH, level by level ----
		SahNode *snodes = (SahNode *)(A + o_snodes);
		uint32_t *slot_of = (uint32_t *)(A + o_slot), *split_of = (uint32_t *)(A + o_split), *size = (uint32_t *)(A + o_size), *height = (uint32_t *)(A + o_height),
				*rank = (uint32_t *)(A + o_rank), *first = (uint32_t *)(A + o_first), *node_of = (uint32_t *)(A + o_node_of), *bins = (uint32_t *)(A + o_bins);
		uint32_t *open[2] = { (uint32_t *)(A + o_open0), (uint32_t *)(A + o_open1) };
		if (!arena->pinned) DB_TRY(hipHostMalloc((void **)&arena->pinned, 64));
		uint32_t *ctr = scal + 12; // [12] next free node id, [13] open nodes of the next level, [14] the triangles in them
		// the list of active triangles and its sort keys, in / out of the sort: the halves of the two Morton-key arrays (unused by this form)
		uint32_t *key_in = (uint32_t *)keys_a, *val_in = key_in + nn, *key_out = (uint32_t *)keys_b, *val_out = key_out + nn;
		DB_TRY(hipMemsetAsync(node_of, 0, nn * 4, stream));
		hipLaunchKernelGGL(sah_root_kernel, dim3(1), dim3(1), 0, stream, snodes, n, scal, slot_of, open[0], rank, first, ctr);
		std::vector<uint32_t> level_lo; // ids of the nodes of depth d: [level_lo[d], level_lo[d + 1])
		level_lo.push_back(0u); level_lo.push_back(1u);
		uint32_t n_open = 1u, n_ids = 1u, n_list = n, n_act = n; // n_list: triangles on the list; n_act: those in this level's open nodes
		const uint32_t *act = nullptr;                          // nullptr: the list is every triangle, in index order
		int cur = 0;
		while (n_open > 0u) {
			if (level_lo.size() > 4096u) { std::snprintf(err, err_len, "device build: the SAH subdivision did not end"); cleanup(); return MRT_ERR_HIP; }
			if ((size_t)n_open > nn / 2 + SAH_LDS_SLOTS || n_act > n_list) { std::snprintf(err, err_len, "device build: inconsistent SAH level (%u open nodes, %u of %u triangles)", n_open, n_act, n_list); cleanup(); return MRT_ERR_HIP; }
			DB_TRY(hipMemsetAsync(ctr + 1, 0, 8, stream)); // (ctr[0], the next free id, runs on from level to level)
			DB_TRY(hipMemsetAsync(bins, 0, (size_t)(n_open < SAH_LDS_SLOTS ? SAH_LDS_SLOTS : n_open) * SAH_SLOT_WORDS * 4, stream));
			if (act == nullptr && n_open <= SAH_LDS_SLOTS)
				hipLaunchKernelGGL(sah_bin_kernel, dim3(blocks < 1024u ? blocks : 1024u), dim3(LBVH_WG), 0, stream, boxes, n, node_of, snodes, slot_of, split_of, bins);
			else {
				hipLaunchKernelGGL(sah_descend_kernel, dim3((n_list + LBVH_WG - 1) / LBVH_WG), dim3(LBVH_WG), 0, stream, boxes, act, n_list, node_of, snodes, slot_of, split_of, key_in, val_in);
				// the closed triangles' key is all ones: above every slot in the low `bits` bits too
				const unsigned bits = 32u - (unsigned)__builtin_clz(n_open);
				size_t sort32 = 0;
				DB_TRY(rocprim::radix_sort_pairs(nullptr, sort32, key_in, key_out, val_in, val_out, (size_t)n_list, 0, bits, stream));
				if (sort32 > (sort_bytes > scan_bytes ? sort_bytes : scan_bytes)) { std::snprintf(err, err_len, "device build: sort workspace"); cleanup(); return MRT_ERR_HIP; }
				DB_TRY(rocprim::radix_sort_pairs(sort_tmp, sort32, key_in, key_out, val_in, val_out, (size_t)n_list, 0, bits, stream));
				act = val_out; n_list = n_act;
				hipLaunchKernelGGL(sah_bin_sorted_kernel, dim3((n_list + LBVH_WG - 1) / LBVH_WG), dim3(LBVH_WG), 0, stream, boxes, act, key_out, n_list, node_of, snodes, bins);
			}
			hipLaunchKernelGGL(sah_split_kernel, dim3((n_open + 63u) / 64u), dim3(64), 0, stream, snodes, open[cur], n_open, bins, slot_of, split_of, ctr, open[cur ^ 1], scal);
			uint32_t *got = arena->pinned;
			DB_TRY(hipMemcpyAsync(got, ctr, 12, hipMemcpyDeviceToHost, stream));
			DB_TRY(hipStreamSynchronize(stream));
			if (got[0] < n_ids || got[0] > 2u * n - 1u) { std::snprintf(err, err_len, "device build: the SAH subdivision made %u nodes for %u triangles", got[0], n); cleanup(); return MRT_ERR_HIP; }
			if (got[0] > n_ids) level_lo.push_back(got[0]);
			n_ids = got[0]; n_open = got[1]; n_act = got[2];
			cur ^= 1;
		}
		// the triangles of the nodes split last still sit in their parent: one more descent
		hipLaunchKernelGGL(sah_descend_kernel, dim3((n_list + LBVH_WG - 1) / LBVH_WG), dim3(LBVH_WG), 0, stream, boxes, act, n_list, node_of, snodes, slot_of, split_of, key_in, val_in);
		const uint32_t depths = (uint32_t)level_lo.size() - 1u;
		uint32_t wrap_lc = 0u;
		if (n_ids == 1u) { // the root stayed a leaf (coincident triangles): one row, its halves
			wrap_lc = (n + 1u) / 2u; n_rows = 1u;
			hipLaunchKernelGGL(sah_wrap_kernel, dim3(1), dim3(1), 0, stream, scal, n, wrap_lc, nodes, node_depth, scal + 6);
		} else {
			n_rows = (n_ids - 1u) / 2u;
			for (uint32_t d = depths; d-- > 0u;) {
				const uint32_t lo = level_lo[d], hi = level_lo[d + 1];
				hipLaunchKernelGGL(sah_size_kernel, dim3((hi - lo + LBVH_WG - 1) / LBVH_WG), dim3(LBVH_WG), 0, stream, snodes, lo, hi, size, height);
			}
			for (uint32_t d = 0; d < depths; d++) {
				const uint32_t lo = level_lo[d], hi = level_lo[d + 1];
				hipLaunchKernelGGL(sah_emit_kernel, dim3((hi - lo + LBVH_WG - 1) / LBVH_WG), dim3(LBVH_WG), 0, stream, snodes, lo, hi, size, height, rank, first,
						nodes, node_depth, scal + 6);
			}
		}
		uint32_t *key32_a = (uint32_t *)keys_a, *key32_b = (uint32_t *)keys_b;
		hipLaunchKernelGGL(sah_keys_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, node_of, first, n, wrap_lc, key32_a, idx_a);
		size_t sort32 = 0;
		DB_TRY(rocprim::radix_sort_pairs(nullptr, sort32, key32_a, key32_b, idx_a, idx_b, nn, 0, 32, stream));
		if (sort32 > (sort_bytes > scan_bytes ? sort_bytes : scan_bytes)) { std::snprintf(err, err_len, "device build: sort workspace"); cleanup(); return MRT_ERR_HIP; }
		DB_TRY(rocprim::radix_sort_pairs(sort_tmp, sort32, key32_a, key32_b, idx_a, idx_b, nn, 0, 32, stream));
		hipLaunchKernelGGL(lbvh_leaves_kernel, dim3(blocks), dim3(LBVH_WG), 0, stream, d_tris, n, idx_b, (const uint32_t *)key32_b, hot, cold);
		const uint32_t rb = (n_rows + LBVH_WG - 1) / LBVH_WG;
		if (!wrap_lc) hipLaunchKernelGGL(lbvh_verify_kernel, dim3(rb), dim3(LBVH_WG), 0, stream, nodes, n_rows, boxes, idx_b, node_depth, scal + 6, scal + 8);
		if (want4) hipLaunchKernelGGL(lbvh_collapse4_kernel, dim3(rb), dim3(LBVH_WG), 0, stream, nodes, n_rows, nodes4);
		if (want8) hipLaunchKernelGGL(lbvh_collapse8_kernel, dim3(rb), dim3(LBVH_WG), 0, stream, nodes, n_rows, nodes8, leaf_box, scal + 7);
		DB_TRY(hipGetLastError());
		DB_TRY(hipMemcpyAsync(h, scal, sizeof(h), hipMemcpyDeviceToHost, stream));
		DB_TRY(hipStreamSynchronize(stream));
		if (h[8] != 0u) { std::snprintf(err, err_len, "device build: the tree failed its verification pass (%u nodes)", h[8]); cleanup(); return MRT_ERR_HIP; }
	} else if (form == 1) {
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
	out->n_nodes = n_rows; out->n_tris = n;
	out->depth = h[6] + 1u; // pending entries on the deepest path + the sentinel
	// 4-wide walk: every 4-wide node on a path leaves at most 3 entries pending and descends at least one binary level
	out->nodes4 = nodes4; out->stack4 = nodes4 ? 3u * h[6] + 1u : 0u;
	if (nodes8 && h[7] != 0u) { (void)hipFree(nodes8); (void)hipFree(leaf_box); nodes8 = nullptr; leaf_box = nullptr; } // a box that fits no grid: go without this layout
	out->nodes8 = nodes8; out->leaf_box = leaf_box; out->stack8 = nodes8 ? 7u * h[6] + 1u : 0u;
	for (int k = 0; k < 3; k++) { out->bounds_lo[k] = ord2f(h[k]); out->bounds_hi[k] = ord2f(h[3 + k]); }
	return MRT_OK;
}

} // namespace mrt
