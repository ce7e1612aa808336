// bvh_builder.cpp — host-side 8-bin SAH BVH2 builder (mrt_bvh2_build).
//
// Produces exactly what tinybvh::BVH::Build produces for the reference
// (thirdparty/tinybvh/tiny_bvh.h:2124-2136, 2261-2466, called from
// src/accel/ray_scene.h:62-86): 32-byte nodes, root = 0, node 1 a hole,
// children as adjacent pairs allocated in split order, leaves indexing prim_idx.
// The result is deterministic for any thread count: the tree is cut into
// independent sub-ranges that are built concurrently with local numbering, and
// the final node numbers are assigned afterwards in the single-threaded
// builder's order (pairs numbered in preorder of the split nodes).  So the
// output equals the reference's single-threaded build bit for bit, which the
// reference's own threaded build (>= 50 000 triangles) only matches in topology.
//
// Off the measured path (the metric is Mrays/s); compiled -ffp-contract=off.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../mrt_internal.h"

namespace {

constexpr int kBins = 8;        // BVHBINS, tiny_bvh.h:104-106
constexpr float kFar = 1e30f;   // BVH_FAR, tiny_bvh.h:140

struct Box { float mn[3], mx[3]; };

inline float fmin_(float a, float b) { return a < b ? a : b; }
inline float fmax_(float a, float b) { return a > b ? a : b; }
inline int clampi(int x, int a, int b) { return x > a ? (x < b ? x : b) : a; }
inline int f2i(float f) { return (f > -2147483648.0f && f < 2147483648.0f) ? (int)f : (int)0x80000000u; } // cvttss2si
inline float half_area(const float mn[3], const float mx[3])
{
	const float e0 = mx[0] - mn[0], e1 = mx[1] - mn[1], e2 = mx[2] - mn[2];
	return e0 < -kFar ? 0.0f : (e0 * e1 + e1 * e2 + e2 * e0);
}

struct Bins {
	Box box[3][kBins];
	uint32_t count[3][kBins];
	void clear()
	{
		for (int a = 0; a < 3; a++) for (int i = 0; i < kBins; i++) {
			for (int k = 0; k < 3; k++) { box[a][i].mn[k] = kFar; box[a][i].mx[k] = -kFar; }
			count[a][i] = 0;
		}
	}
	void merge(const Bins &o)
	{
		for (int a = 0; a < 3; a++) for (int i = 0; i < kBins; i++) {
			for (int k = 0; k < 3; k++) {
				box[a][i].mn[k] = fmin_(box[a][i].mn[k], o.box[a][i].mn[k]);
				box[a][i].mx[k] = fmax_(box[a][i].mx[k], o.box[a][i].mx[k]);
			}
			count[a][i] += o.count[a][i];
		}
	}
};

struct Builder {
	const Box *frag;
	uint32_t *prim_idx;
	float min_dim[3];
	int n_threads;

	void bin_range(const mrt_bvh_node32 &node, const float rpd3[3], uint32_t a0, uint32_t a1, Bins &b) const
	{
		for (uint32_t i = a0; i < a1; i++) {
			const Box &f = frag[prim_idx[i]];
			for (int a = 0; a < 3; a++) {
				int bi = clampi(f2i(((f.mn[a] + f.mx[a]) * 0.5f - node.aabb_min[a]) * rpd3[a]), 0, kBins - 1);
				Box &bb = b.box[a][bi];
				for (int k = 0; k < 3; k++) { bb.mn[k] = fmin_(bb.mn[k], f.mn[k]); bb.mx[k] = fmax_(bb.mx[k], f.mx[k]); }
				b.count[a][bi]++;
			}
		}
	}

	// One split decision + partition.  Returns false when the node stays a leaf.
	bool split(mrt_bvh_node32 &node, mrt_bvh_node32 &left, mrt_bvh_node32 &right, bool parallel) const
	{
		float rpd3[3];
		for (int k = 0; k < 3; k++) rpd3[k] = (float)kBins / (node.aabb_max[k] - node.aabb_min[k]);
		Bins bins; bins.clear();
		const uint32_t first = node.left_first, cnt = node.tri_count;
		if (parallel && n_threads > 1 && cnt > 65536) {
			// min / max / integer adds are order independent, so the merged bins equal the serial ones
			std::vector<Bins> part(n_threads);
			std::vector<std::thread> th;
			const uint32_t chunk = (cnt + n_threads - 1) / n_threads;
			for (int t = 0; t < n_threads; t++) th.emplace_back([&, t] {
				part[t].clear();
				const uint32_t a0 = first + std::min<uint32_t>(cnt, t * chunk), a1 = first + std::min<uint32_t>(cnt, (t + 1) * chunk);
				bin_range(node, rpd3, a0, a1, part[t]);
			});
			for (auto &x : th) x.join();
			for (int t = 0; t < n_threads; t++) bins.merge(part[t]);
		} else bin_range(node, rpd3, first, first + cnt, bins);

		float split_cost = kFar;
		const float ext[3] = { node.aabb_max[0] - node.aabb_min[0], node.aabb_max[1] - node.aabb_min[1], node.aabb_max[2] - node.aabb_min[2] };
		const float rsav = 1.0f / (ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0]);
		int best_axis = 0, best_pos = 0;
		Box best_l{}, best_r{};
		for (int a = 0; a < 3; a++) if (ext[a] > min_dim[a]) {
			Box lb[kBins - 1], rb[kBins - 1], l, r;
			for (int k = 0; k < 3; k++) { l.mn[k] = r.mn[k] = kFar; l.mx[k] = r.mx[k] = -kFar; }
			float anl[kBins - 1], anr[kBins - 1];
			uint32_t ln = 0, rn = 0;
			for (int i = 0; i < kBins - 1; i++) {
				for (int k = 0; k < 3; k++) {
					l.mn[k] = fmin_(l.mn[k], bins.box[a][i].mn[k]); l.mx[k] = fmax_(l.mx[k], bins.box[a][i].mx[k]);
					r.mn[k] = fmin_(r.mn[k], bins.box[a][kBins - 1 - i].mn[k]); r.mx[k] = fmax_(r.mx[k], bins.box[a][kBins - 1 - i].mx[k]);
				}
				lb[i] = l; rb[kBins - 2 - i] = r;
				ln += bins.count[a][i]; rn += bins.count[a][kBins - 1 - i];
				anl[i] = ln == 0 ? kFar : half_area(l.mn, l.mx) * (float)ln;
				anr[kBins - 2 - i] = rn == 0 ? kFar : half_area(r.mn, r.mx) * (float)rn;
			}
			for (int i = 0; i < kBins - 1; i++) {
				const float c = anl[i] + anr[i];
				if (c < split_cost) { split_cost = c; best_axis = a; best_pos = i; best_l = lb[i]; best_r = rb[i]; }
			}
		}
		split_cost = 1.0f + 1.0f * rsav * split_cost;           // c_trav + c_int * rSAV * cost
		if (split_cost >= (float)cnt * 1.0f) return false;        // not splitting is better
		uint32_t j = first + cnt, src = first;
		const float rpd = rpd3[best_axis], nmin = node.aabb_min[best_axis];
		for (uint32_t i = 0; i < cnt; i++) {
			const Box &f = frag[prim_idx[src]];
			int bi = (int)(uint32_t)(long long)(((f.mn[best_axis] + f.mx[best_axis]) * 0.5f - nmin) * rpd);
			bi = clampi(bi, 0, kBins - 1);
			if (bi <= best_pos) src++; else std::swap(prim_idx[src], prim_idx[--j]);
		}
		const uint32_t lc = src - first, rc = cnt - lc;
		if (lc == 0 || rc == 0) return false;
		for (int k = 0; k < 3; k++) {
			left.aabb_min[k] = best_l.mn[k]; left.aabb_max[k] = best_l.mx[k];
			right.aabb_min[k] = best_r.mn[k]; right.aabb_max[k] = best_r.mx[k];
		}
		left.left_first = first; left.tri_count = lc;
		right.left_first = j; right.tri_count = rc;
		return true;
	}
};

// A tree under construction with local pair numbering: the children of the
// k-th split node (preorder) live at pairs[2k], pairs[2k+1]; an internal child
// stores the LOCAL rank of its own split in left_first (tri_count = 0), or
// kCut | task id when it was cut off for a concurrent sub-build.
constexpr uint32_t kCut = 0x80000000u;
struct LocalTree { std::vector<mrt_bvh_node32> pairs; };

// Sequential subdivision in the reference's order (explicit stack, left first).
// root_is_split tells whether `root` itself was split (then rank 0 is its split).
void build_local(const Builder &b, mrt_bvh_node32 root, LocalTree &out, bool &root_is_split,
		uint32_t cut_threshold, std::vector<mrt_bvh_node32> *cut_tasks, bool parallel_bins)
{
	out.pairs.clear();
	root_is_split = false;
	std::vector<uint32_t> stack;
	// The node being subdivided is either the root (slot = UINT32_MAX) or out.pairs[slot].
	uint32_t cur = 0xFFFFFFFFu;
	for (;;) {
		for (;;) {
			mrt_bvh_node32 node = (cur == 0xFFFFFFFFu) ? root : out.pairs[cur];
			if (cut_tasks && cur != 0xFFFFFFFFu && node.tri_count <= cut_threshold) {
				// hand this range to a concurrent sub-build; remember which task
				const uint32_t id = (uint32_t)cut_tasks->size();
				cut_tasks->push_back(node);
				out.pairs[cur].tri_count = 0; out.pairs[cur].left_first = kCut | id;
				break;
			}
			mrt_bvh_node32 l{}, r{};
			if (!b.split(node, l, r, parallel_bins)) break;
			const uint32_t rank = (uint32_t)(out.pairs.size() / 2);
			out.pairs.push_back(l); out.pairs.push_back(r);
			if (cur == 0xFFFFFFFFu) root_is_split = true;
			else { out.pairs[cur].left_first = rank; out.pairs[cur].tri_count = 0; }
			stack.push_back(2 * rank + 1);
			cur = 2 * rank;
		}
		if (stack.empty()) break;
		cur = stack.back(); stack.pop_back();
	}
}

} // namespace

extern "C" int mrt_bvh2_build(const float *verts4, uint32_t n_tris, mrt_bvh_node32 *nodes,
		uint32_t *prim_idx, uint32_t *used_nodes, uint32_t n_threads)
{
	if (!verts4 || !nodes || !prim_idx || !used_nodes || n_tris == 0) return MRT_ERR_INVALID;
	if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
	n_threads = std::min(n_threads, 64u);
	std::vector<Box> frag(n_tris);
	// PrepareBuild, tiny_bvh.h:2290-2311
	mrt_bvh_node32 root{};
	root.left_first = 0; root.tri_count = n_tris;
	for (int k = 0; k < 3; k++) { root.aabb_min[k] = kFar; root.aabb_max[k] = -kFar; }
	for (uint32_t i = 0; i < n_tris; i++) {
		const float *v0 = verts4 + 12 * (size_t)i, *v1 = v0 + 4, *v2 = v0 + 8;
		for (int k = 0; k < 3; k++) {
			frag[i].mn[k] = fmin_(v0[k], fmin_(v1[k], v2[k]));
			frag[i].mx[k] = fmax_(v0[k], fmax_(v1[k], v2[k]));
			root.aabb_min[k] = fmin_(root.aabb_min[k], frag[i].mn[k]);
			root.aabb_max[k] = fmax_(root.aabb_max[k], frag[i].mx[k]);
		}
		prim_idx[i] = i;
	}
	Builder b;
	b.frag = frag.data(); b.prim_idx = prim_idx; b.n_threads = (int)n_threads;
	for (int k = 0; k < 3; k++) b.min_dim[k] = (root.aabb_max[k] - root.aabb_min[k]) * 1e-20f;

	// ---- top of the tree (sequential control, parallel binning), cut into tasks ----
	const bool threaded = n_threads > 1 && n_tris >= 32768;
	const uint32_t cut = threaded ? std::max<uint32_t>(4096u, n_tris / (n_threads * 8u)) : 0u;
	LocalTree top; bool root_split = false;
	std::vector<mrt_bvh_node32> tasks;
	build_local(b, root, top, root_split, cut, threaded ? &tasks : nullptr, threaded);

	// ---- sub-builds, concurrently; each touches only its own prim_idx range ----
	std::vector<LocalTree> sub(tasks.size());
	std::vector<uint8_t> sub_split(tasks.size(), 0);
	if (!tasks.empty()) {
		std::atomic<uint32_t> next{0};
		std::vector<std::thread> th;
		for (uint32_t t = 0; t < n_threads; t++) th.emplace_back([&] {
			for (;;) {
				const uint32_t i = next.fetch_add(1);
				if (i >= tasks.size()) break;
				bool s = false;
				build_local(b, tasks[i], sub[i], s, 0, nullptr, false);
				sub_split[i] = s ? 1 : 0;
			}
		});
		for (auto &x : th) x.join();
	}

	// ---- final numbering: pairs in preorder of the split nodes (tiny_bvh.h:2424) ----
	// Walk the top tree in preorder; a cut node contributes its whole sub-tree's splits.
	std::memset(&nodes[1], 0, sizeof(mrt_bvh_node32)); // node 1 stays unused (:2285)
	nodes[0] = root;
	uint32_t next_rank = 0; // global preorder rank of the next split
	const uint32_t n_top_splits = (uint32_t)(top.pairs.size() / 2);
	std::vector<uint32_t> top_rank(n_top_splits, 0), task_base(tasks.size(), 0);
	if (root_split) {
		// iterative preorder over top split nodes; local rank order IS preorder within `top`,
		// but cut sub-trees interleave, so walk explicitly.
		// children of a split are visited left then right: an explicit stack of "nodes"
		// where a node is either a top split (local rank) or a cut task (kCut | id).
		std::vector<uint32_t> work; work.push_back(0); // start at top split rank 0 (the root's split)
		while (!work.empty()) {
			const uint32_t w = work.back(); work.pop_back();
			if (w & kCut) {
				const uint32_t id = w & ~kCut;
				task_base[id] = next_rank;
				if (sub_split[id]) next_rank += (uint32_t)(sub[id].pairs.size() / 2);
				continue;
			}
			top_rank[w] = next_rank++;
			const mrt_bvh_node32 &l = top.pairs[2 * w], &r = top.pairs[2 * w + 1];
			// push right first so left is processed first
			if (r.tri_count == 0) work.push_back(r.left_first);
			if (l.tri_count == 0) work.push_back(l.left_first);
		}
		// emit top pairs
		for (uint32_t w = 0; w < n_top_splits; w++) {
			const uint32_t p = 2 + 2 * top_rank[w];
			for (int side = 0; side < 2; side++) {
				mrt_bvh_node32 c = top.pairs[2 * w + side];
				if (c.tri_count == 0) {
					if (c.left_first & kCut) {
						const uint32_t id = c.left_first & ~kCut;
						if (sub_split[id]) { c.left_first = 2 + 2 * task_base[id]; c.tri_count = 0; }
						else { c.left_first = tasks[id].left_first; c.tri_count = tasks[id].tri_count; }
					} else c.left_first = 2 + 2 * top_rank[c.left_first];
				}
				nodes[p + side] = c;
			}
		}
		nodes[0].left_first = 2 + 2 * top_rank[0]; nodes[0].tri_count = 0;
		// emit sub-tree pairs
		for (size_t id = 0; id < tasks.size(); id++) if (sub_split[id]) {
			const uint32_t base = task_base[id];
			const uint32_t ns = (uint32_t)(sub[id].pairs.size() / 2);
			for (uint32_t w = 0; w < ns; w++) for (int side = 0; side < 2; side++) {
				mrt_bvh_node32 c = sub[id].pairs[2 * w + side];
				if (c.tri_count == 0) c.left_first = 2 + 2 * (base + c.left_first);
				nodes[2 + 2 * (base + w) + side] = c;
			}
		}
	}
	*used_nodes = 2 + 2 * next_rank;
	return MRT_OK;
}

// ---- BVH cache file: the counterpart of tinybvh::BVH::Save / Load (tiny_bvh.h:1747-1799) -----------
// Layout: 32-byte header {magic "MRTBVH2\0", version, n_tris, used_nodes, 64-bit FNV-1a of the payload},
// then used_nodes nodes (32 B each) and n_tris prim indices.  Like the reference's Load, a file is
// only accepted for the triangle count it was saved for; unlike it, the payload is checksummed and the
// node and index ranges are validated by mrt_upload_scene, so a damaged file cannot reach the device.
namespace {
struct CacheHeader { char magic[8]; uint32_t version, n_tris, used_nodes, reserved; uint64_t checksum; };
static_assert(sizeof(CacheHeader) == 32, "cache header must be 32 bytes");
const char kCacheMagic[8] = { 'M', 'R', 'T', 'B', 'V', 'H', '2', 0 };
constexpr uint32_t kCacheVersion = 1;
uint64_t fnv1a(uint64_t h, const void *p, size_t n)
{
	const unsigned char *b = (const unsigned char *)p;
	for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
	return h;
}
} // namespace

extern "C" int mrt_bvh2_save(const char *path, const mrt_bvh_node32 *nodes, uint32_t used_nodes, const uint32_t *prim_idx, uint32_t n_tris)
{
	if (!path || !nodes || !prim_idx || used_nodes == 0 || n_tris == 0) return MRT_ERR_INVALID;
	CacheHeader h;
	std::memset(&h, 0, sizeof(h));
	std::memcpy(h.magic, kCacheMagic, 8);
	h.version = kCacheVersion; h.n_tris = n_tris; h.used_nodes = used_nodes;
	h.checksum = fnv1a(fnv1a(14695981039346656037ull, nodes, (size_t)used_nodes * sizeof(mrt_bvh_node32)), prim_idx, (size_t)n_tris * 4);
	std::FILE *f = std::fopen(path, "wb");
	if (!f) return MRT_ERR_INVALID;
	bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 &&
			std::fwrite(nodes, sizeof(mrt_bvh_node32), used_nodes, f) == used_nodes &&
			std::fwrite(prim_idx, 4, n_tris, f) == n_tris;
	ok = (std::fclose(f) == 0) && ok;
	return ok ? MRT_OK : MRT_ERR_INVALID;
}

extern "C" int mrt_bvh2_load(const char *path, uint32_t n_tris, mrt_bvh_node32 *nodes, uint32_t *prim_idx, uint32_t *used_nodes)
{
	if (!path || !nodes || !prim_idx || !used_nodes || n_tris == 0) return MRT_ERR_INVALID;
	std::FILE *f = std::fopen(path, "rb");
	if (!f) return MRT_ERR_INVALID;
	CacheHeader h;
	int rc = MRT_ERR_BAD_BVH;
	if (std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, kCacheMagic, 8) == 0 && h.version == kCacheVersion && h.reserved == 0u &&
			h.n_tris == n_tris && h.used_nodes >= 1 && (uint64_t)h.used_nodes <= 2ull * n_tris) { // the caller's array holds 2 n_tris nodes
		if (std::fread(nodes, sizeof(mrt_bvh_node32), h.used_nodes, f) == h.used_nodes &&
				std::fread(prim_idx, 4, n_tris, f) == n_tris && std::fgetc(f) == EOF &&
				h.checksum == fnv1a(fnv1a(14695981039346656037ull, nodes, (size_t)h.used_nodes * sizeof(mrt_bvh_node32)), prim_idx, (size_t)n_tris * 4)) {
			*used_nodes = h.used_nodes;
			rc = MRT_OK;
		}
	}
	std::fclose(f);
	return rc;
}
