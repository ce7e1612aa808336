// scene_prep.cpp — host-side scene preparation for the device layout.
//
//  * mrt_make_triangles / mrt_pack_host_triangles: Triangle ctor and the
//    Triangle -> GPUTrianglePacked loop (src/core/triangle.h:41-51,
//    src/gpu/gpu_ray_caster.cpp:205-217).
//  * prepare_scene: TinyBVH BVH2 -> dual-AABB nodes, the conversion half of
//    GPURayCaster::upload_scene (src/gpu/gpu_ray_caster.cpp:219-311), with the
//    prim_idx indirection resolved and arrays sized by used_nodes
//    (defects 1 and 2), validated before anything is
//    handed to a kernel.
//  * mrt_camera_look: camera basis of RayTracerDebug::cast_debug_rays
//    (src/godot/raytracer_debug.cpp:573-583).
//
// Compiled with -ffp-contract=off: every rounding is the one written.
#include <cmath>
#include <limits>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>
#include "../mrt_internal.h"

namespace {

// godot::Vector3::normalized(): zero stays zero, otherwise divide by the length.
inline void normalize3(float v[3])
{
	float l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
	if (l2 == 0.0f) { v[0] = v[1] = v[2] = 0.0f; return; }
	float l = std::sqrt(l2);
	v[0] /= l; v[1] /= l; v[2] /= l;
}
inline void cross3(const float a[3], const float b[3], float r[3])
{
	r[0] = a[1] * b[2] - a[2] * b[1];
	r[1] = a[2] * b[0] - a[0] * b[2];
	r[2] = a[0] * b[1] - a[1] * b[0];
}

} // namespace

extern "C" int mrt_make_triangles(const float *verts9, const uint32_t *ids, const uint32_t *layers,
		uint32_t n_tris, mrt_tri64 *out)
{
	if (!verts9 || !out) return MRT_ERR_INVALID;
	for (uint32_t i = 0; i < n_tris; i++) {
		const float *a = verts9 + 9 * (size_t)i, *b = a + 3, *c = a + 6;
		mrt_tri64 &t = out[i];
		for (int k = 0; k < 3; k++) { t.v0[k] = a[k]; t.edge1[k] = b[k] - a[k]; t.edge2[k] = c[k] - a[k]; }
		cross3(t.edge1, t.edge2, t.normal);
		normalize3(t.normal);
		t.id = ids ? ids[i] : i;
		t.layers = layers ? layers[i] : 0xFFFFFFFFu;
		t.pad2 = 0.0f; t.pad3 = 0.0f;
	}
	return MRT_OK;
}

extern "C" int mrt_pack_host_triangles(const mrt_host_tri80 *tris, uint32_t n_tris, mrt_tri64 *out)
{
	if (!tris || !out) return MRT_ERR_INVALID;
	for (uint32_t i = 0; i < n_tris; i++) {
		const mrt_host_tri80 &t = tris[i];
		mrt_tri64 &g = out[i];
		for (int k = 0; k < 3; k++) { g.v0[k] = t.v0[k]; g.edge1[k] = t.edge1[k]; g.edge2[k] = t.edge2[k]; g.normal[k] = t.normal[k]; }
		g.id = t.id; g.layers = t.layers; g.pad2 = 0.0f; g.pad3 = 0.0f;
	}
	return MRT_OK;
}

extern "C" int mrt_camera_look(mrt_camera *cam, const float origin[3], const float forward[3],
		uint32_t grid_w, uint32_t grid_h, float fov_degrees)
{
	if (!cam || !origin || !forward || grid_w == 0 || grid_h == 0) return MRT_ERR_INVALID;
	float fwd[3] = { forward[0], forward[1], forward[2] };
	normalize3(fwd);
	float hint[3] = { 0.0f, 1.0f, 0.0f };
	if (std::fabs(fwd[0] * hint[0] + fwd[1] * hint[1] + fwd[2] * hint[2]) > 0.99f) { hint[0] = 1.0f; hint[1] = 0.0f; }
	float right[3], up[3];
	cross3(fwd, hint, right); normalize3(right);
	cross3(right, fwd, up); normalize3(up);
	const float half_fov_rad = (fov_degrees * 0.5f) * (float)(3.14159265358979323846 / 180.0); // Math::deg_to_rad
	const float half_w = std::tan(half_fov_rad);
	const float half_h = half_w * ((float)grid_h / (float)grid_w);
	for (int k = 0; k < 3; k++) { cam->origin[k] = origin[k]; cam->fwd[k] = fwd[k]; cam->right[k] = right[k]; cam->up[k] = up[k]; }
	cam->half_w = half_w; cam->half_h = half_h;
	cam->t_min = 0.001f; cam->t_max = FLT_MAX; // Ray(origin, dir) defaults, src/core/ray.h:59
	cam->kind = MRT_CAMERA_DEBUG_GRID;
	cam->inv_w = 1.0f / (float)grid_w; cam->inv_h = 1.0f / (float)grid_h; // unused by this kind
	cam->jitter_x = cam->jitter_y = 0.5f;
	cam->reserved[0] = cam->reserved[1] = cam->reserved[2] = 0u;
	return MRT_OK;
}

// RayCamera::setup (ray_camera.h:50-76): origin, basis, 1 / resolution; columns 0 / 1 / 2 of the basis
static void ray_camera_common(mrt_camera *cam, const float origin[3], const float basis[9], uint32_t width, uint32_t height)
{
	for (int k = 0; k < 3; k++) {
		cam->origin[k] = origin[k];
		cam->right[k] = basis[3 * k + 0]; cam->up[k] = basis[3 * k + 1]; cam->fwd[k] = basis[3 * k + 2];
	}
	cam->inv_w = 1.0f / (float)width; cam->inv_h = 1.0f / (float)height;
	cam->jitter_x = cam->jitter_y = 0.5f;
	cam->t_min = 0.001f; cam->t_max = FLT_MAX;
	cam->reserved[0] = cam->reserved[1] = cam->reserved[2] = 0u;
}

// RayCamera::_setup_perspective, ray_camera.h:208-218: Math_PI is a double constant, so the tangent is
// taken in double and rounded once; half_w = tan_half * aspect in float.
extern "C" int mrt_camera_perspective(mrt_camera *cam, const float origin[3], const float basis[9],
		uint32_t width, uint32_t height, float fov_degrees)
{
	if (!cam || !origin || !basis || width == 0 || height == 0 || !(fov_degrees > 0.0f)) return MRT_ERR_INVALID;
	ray_camera_common(cam, origin, basis, width, height);
	const float aspect = (float)width / (float)height;
	const float tan_half = (float)std::tan((double)(fov_degrees * 0.5f) * (3.1415926535897932384626433833 / 180.0));
	cam->half_w = tan_half * aspect; cam->half_h = tan_half;
	cam->kind = MRT_CAMERA_PERSPECTIVE;
	return MRT_OK;
}

// RayCamera::_setup_orthographic, ray_camera.h:220-230: size = full vertical extent in world units
extern "C" int mrt_camera_orthographic(mrt_camera *cam, const float origin[3], const float basis[9],
		uint32_t width, uint32_t height, float size)
{
	if (!cam || !origin || !basis || width == 0 || height == 0 || !(size > 0.0f)) return MRT_ERR_INVALID;
	ray_camera_common(cam, origin, basis, width, height);
	const float aspect = (float)width / (float)height;
	cam->half_h = size * 0.5f; cam->half_w = cam->half_h * aspect;
	cam->kind = MRT_CAMERA_ORTHOGRAPHIC;
	return MRT_OK;
}

namespace mrt {

int prepare_scene(const mrt_tri64 *tris, uint32_t n_tris, const mrt_bvh_node32 *nodes, uint32_t used_nodes,
		const uint32_t *prim_idx, DeviceSceneHost *out, char *err, size_t err_len)
{
	auto fail = [&](int code, const char *msg) { if (err && err_len) std::snprintf(err, err_len, "%s", msg); return code; };
	if (!tris || !nodes || !prim_idx || !out) return fail(MRT_ERR_INVALID, "upload_scene: null argument");
	if (n_tris == 0 || used_nodes == 0) return fail(MRT_ERR_INVALID, "upload_scene: empty scene");
	if (n_tris >= 0x7FFFFFFFu) return fail(MRT_ERR_UNSUPPORTED, "upload_scene: more than 2^31-1 triangles");

	// ---- leaf-ordered triangle arrays: slot k holds triangle prim_idx[k] ----
	TriHot *hot = (TriHot *)std::malloc((size_t)n_tris * sizeof(TriHot));
	TriCold *cold = (TriCold *)std::malloc((size_t)n_tris * sizeof(TriCold));
	if (!hot || !cold) { std::free(hot); std::free(cold); return fail(MRT_ERR_OOM, "upload_scene: out of host memory"); }
	for (uint32_t k = 0; k < n_tris; k++) {
		const uint32_t pi = prim_idx[k];
		if (pi >= n_tris) { std::free(hot); std::free(cold); return fail(MRT_ERR_BAD_BVH, "upload_scene: prim_idx entry out of range"); }
		const mrt_tri64 &t = tris[pi];
		TriHot &h = hot[k];
		for (int c = 0; c < 3; c++) { h.v0[c] = t.v0[c]; h.e1[c] = t.edge1[c]; h.e2[c] = t.edge2[c]; cold[k].normal[c] = t.normal[c]; }
		h.id = t.id; h.layers = t.layers; h.flags = 0u; cold[k].pad = 0u;
	}

	auto leaf_ok = [&](const mrt_bvh_node32 &n) { return n.tri_count > 0 && n.left_first < n_tris && n.tri_count <= n_tris - n.left_first; };
	auto mark_leaf = [&](uint32_t first, uint32_t count) { hot[first + count - 1].flags |= kLastInLeaf; return kLeafBit | first; };

	std::vector<DevNode> dev;
	uint32_t depth = 0;
	const mrt_bvh_node32 &root = nodes[0];
	if (root.tri_count > 0) {
		// Root is a leaf (tiny scene).  The reference wraps it with a NaN right box
		// (gpu_ray_caster.cpp:255-271); v_min/v_max drop NaNs, so the range is split
		// into two leaf children that both carry the root box instead.
		if (!leaf_ok(root)) { std::free(hot); std::free(cold); return fail(MRT_ERR_BAD_BVH, "upload_scene: root leaf range out of bounds"); }
		DevNode g{};
		const uint32_t lc = (root.tri_count + 1) / 2, rc = root.tri_count - lc;
		for (int c = 0; c < 3; c++) { g.lmin[c] = g.rmin[c] = root.aabb_min[c]; g.lmax[c] = g.rmax[c] = root.aabb_max[c]; }
		g.left_ref = mark_leaf(root.left_first, lc); g.left_count = lc;
		if (rc == 0) { g.right_ref = g.left_ref; g.right_count = lc; }
		else { g.right_ref = mark_leaf(root.left_first + lc, rc); g.right_count = rc; }
		dev.push_back(g);
		depth = 1;
	} else {
		// DFS preorder numbering of internal nodes, with cycle / range validation.
		std::vector<uint32_t> map(used_nodes, 0xFFFFFFFFu);
		struct Item { uint32_t node, d; };
		std::vector<Item> stack;
		stack.push_back({ 0u, 1u });
		uint32_t n_wide = 0;
		std::vector<uint32_t> order;
		while (!stack.empty()) {
			const Item it = stack.back(); stack.pop_back();
			if (map[it.node] != 0xFFFFFFFFu) { std::free(hot); std::free(cold); return fail(MRT_ERR_BAD_BVH, "upload_scene: BVH node referenced twice"); }
			map[it.node] = n_wide++;
			order.push_back(it.node);
			if (it.d > depth) depth = it.d;
			const uint32_t l = nodes[it.node].left_first, r = l + 1;
			if (l == 0 || r >= used_nodes || r < l) { std::free(hot); std::free(cold); return fail(MRT_ERR_BAD_BVH, "upload_scene: child index out of range"); }
			if (nodes[r].tri_count == 0) stack.push_back({ r, it.d + 1 });
			else if (!leaf_ok(nodes[r])) { std::free(hot); std::free(cold); return fail(MRT_ERR_BAD_BVH, "upload_scene: leaf range out of bounds"); }
			if (nodes[l].tri_count == 0) stack.push_back({ l, it.d + 1 });
			else if (!leaf_ok(nodes[l])) { std::free(hot); std::free(cold); return fail(MRT_ERR_BAD_BVH, "upload_scene: leaf range out of bounds"); }
		}
		dev.resize(n_wide);
		for (uint32_t w = 0; w < n_wide; w++) {
			const mrt_bvh_node32 &n = nodes[order[w]];
			const mrt_bvh_node32 &lc = nodes[n.left_first], &rc = nodes[n.left_first + 1];
			DevNode &g = dev[w];
			for (int c = 0; c < 3; c++) {
				g.lmin[c] = lc.aabb_min[c]; g.lmax[c] = lc.aabb_max[c];
				g.rmin[c] = rc.aabb_min[c]; g.rmax[c] = rc.aabb_max[c];
			}
			if (lc.tri_count > 0) { g.left_ref = mark_leaf(lc.left_first, lc.tri_count); g.left_count = lc.tri_count; }
			else { g.left_ref = map[n.left_first]; g.left_count = 0; }
			if (rc.tri_count > 0) { g.right_ref = mark_leaf(rc.left_first, rc.tri_count); g.right_count = rc.tri_count; }
			else { g.right_ref = map[n.left_first + 1]; g.right_count = 0; }
		}
	}
	out->n_nodes = (uint32_t)dev.size();
	out->nodes = (DevNode *)std::malloc(dev.size() * sizeof(DevNode));
	if (!out->nodes) { std::free(hot); std::free(cold); return fail(MRT_ERR_OOM, "upload_scene: out of host memory"); }
	std::memcpy(out->nodes, dev.data(), dev.size() * sizeof(DevNode));
	out->hot = hot; out->cold = cold; out->n_tris = n_tris;
	out->depth = depth + 1; // one pending entry per wide node on the current path + the sentinel
	for (int c = 0; c < 3; c++) { // scene bounds = union of the root's two child boxes
		out->bounds_lo[c] = dev[0].lmin[c] < dev[0].rmin[c] ? dev[0].lmin[c] : dev[0].rmin[c];
		out->bounds_hi[c] = dev[0].lmax[c] > dev[0].rmax[c] ? dev[0].lmax[c] : dev[0].rmax[c];
	}

	// ---- 4-wide collapse of the same tree (MRT_KERNEL_LANE4_PERSISTENT) ----
	// Children of a 4-node: start from the two children of a BVH2 node and keep opening the
	// internal child with the largest half-area until there are four (or only leaves remain).
	{
		struct Child { float mn[3], mx[3]; uint32_t ref; };
		auto area = [](const Child &c) {
			const float e0 = c.mx[0] - c.mn[0], e1 = c.mx[1] - c.mn[1], e2 = c.mx[2] - c.mn[2];
			return e0 * e1 + e1 * e2 + e2 * e0;
		};
		auto children_of = [&](uint32_t w, Child &l, Child &r) {
			const DevNode &g = dev[w];
			for (int c = 0; c < 3; c++) { l.mn[c] = g.lmin[c]; l.mx[c] = g.lmax[c]; r.mn[c] = g.rmin[c]; r.mx[c] = g.rmax[c]; }
			l.ref = g.left_ref; r.ref = g.right_ref;
		};
		struct Work { uint32_t w2, idx4, need; };
		std::vector<Dev4Node> dev4;
		std::vector<Work> work;
		dev4.emplace_back();
		work.push_back({ 0u, 0u, 0u });
		uint32_t stack4 = 1;
		while (!work.empty()) {
			const Work wk = work.back(); work.pop_back();
			Child ch[4]; uint32_t n = 2;
			children_of(wk.w2, ch[0], ch[1]);
			while (n < 4) {
				int best = -1; float best_a = -1.0f;
				for (uint32_t i = 0; i < n; i++) if (ch[i].ref < kSentinel) { const float a = area(ch[i]); if (a > best_a) { best_a = a; best = (int)i; } }
				if (best < 0) break;
				Child l, r; children_of(ch[best].ref, l, r);
				ch[best] = l; ch[n++] = r;
			}
			const uint32_t need = wk.need + (n - 1);
			if (need + 1 > stack4) stack4 = need + 1;
			Dev4Node node{};
			node.n_children = n;
			for (uint32_t i = 0; i < 4; i++) {
				if (i < n) {
					for (int c = 0; c < 3; c++) { node.box[i][c] = ch[i].mn[c]; node.box[i][3 + c] = ch[i].mx[c]; }
					if (ch[i].ref < kSentinel) {
						const uint32_t idx = (uint32_t)dev4.size();
						dev4.emplace_back();
						work.push_back({ ch[i].ref, idx, need });
						node.ref[i] = idx;
					} else node.ref[i] = ch[i].ref;
				} else {
					// unused slot: the point box at +inf; its slab test fails for every finite ray interval
					for (int c = 0; c < 6; c++) node.box[i][c] = std::numeric_limits<float>::infinity();
					node.ref[i] = kSentinel;
				}
			}
			dev4[wk.idx4] = node;
		}
		out->n_nodes4 = (uint32_t)dev4.size();
		out->nodes4 = (Dev4Node *)std::malloc(dev4.size() * sizeof(Dev4Node));
		if (!out->nodes4) { std::free(hot); std::free(cold); std::free(out->nodes); out->nodes = nullptr; return fail(MRT_ERR_OOM, "upload_scene: out of host memory"); }
		std::memcpy(out->nodes4, dev4.data(), dev4.size() * sizeof(Dev4Node));
		out->stack4 = stack4;
	}

	// ---- 8-wide compressed collapse (built when out->want8): same greedy rule, up to 8 children,
	//      child boxes quantised outwards to 8 bits on the node's own grid.  An optional layout: a box
	//      that cannot be put on a grid (non-finite) only means the scene goes without it ----
	bool ok8 = out->want8;
	if (ok8) {
		struct Child { float mn[3], mx[3]; uint32_t ref; };
		auto area = [](const Child &c) {
			const float e0 = c.mx[0] - c.mn[0], e1 = c.mx[1] - c.mn[1], e2 = c.mx[2] - c.mn[2];
			return e0 * e1 + e1 * e2 + e2 * e0;
		};
		auto children_of = [&](uint32_t w, Child &l, Child &r) {
			const DevNode &g = dev[w];
			for (int c = 0; c < 3; c++) { l.mn[c] = g.lmin[c]; l.mx[c] = g.lmax[c]; r.mn[c] = g.rmin[c]; r.mx[c] = g.rmax[c]; }
			l.ref = g.left_ref; r.ref = g.right_ref;
		};
		// phase 1 (sequential, cheap): topology -- which boxes and refs every 8-wide node holds
		struct Kids { Child ch[8]; uint32_t n; };
		struct Work { uint32_t w2, idx8, need; };
		std::vector<Kids> kids;
		std::vector<Work> work;
		kids.emplace_back();
		work.push_back({ 0u, 0u, 0u });
		uint32_t stack8 = 1;
		while (!work.empty()) {
			const Work wk = work.back(); work.pop_back();
			Kids k; k.n = 2;
			children_of(wk.w2, k.ch[0], k.ch[1]);
			while (k.n < 8) {
				int best = -1; float best_a = -1.0f;
				for (uint32_t i = 0; i < k.n; i++) if (k.ch[i].ref < kSentinel) { const float a = area(k.ch[i]); if (a > best_a) { best_a = a; best = (int)i; } }
				if (best < 0) break;
				Child l, r; children_of(k.ch[best].ref, l, r);
				k.ch[best] = l; k.ch[k.n++] = r;
			}
			const uint32_t need = wk.need + (k.n - 1);
			if (need + 1 > stack8) stack8 = need + 1;
			for (uint32_t i = 0; i < k.n; i++) {
				if (k.ch[i].ref < kSentinel) { // an inner child becomes an 8-wide node of its own
					const uint32_t idx = (uint32_t)kids.size();
					kids.emplace_back();
					work.push_back({ k.ch[i].ref, idx, need });
					k.ch[i].ref = idx;
				}
			}
			kids[wk.idx8] = k;
		}
		// phase 2 (parallel): the grid of every node and the outward-rounded child boxes on it
		std::vector<Dev8Node> dev8(kids.size());
		std::atomic<bool> bad{false};
		// exact boxes of the leaves (every leaf is the child of exactly one 8-wide node): the walk checks a
		// candidate hit against them, so the looser quantised boxes never change a result
		float *leaf_box = (float *)std::calloc((size_t)n_tris * 8u, sizeof(float));
		if (!leaf_box) bad = true;
		auto quantise = [&](size_t first, size_t last) {
			for (size_t w = first; w < last; w++) {
				const Kids &k = kids[w];
				Dev8Node node;
				std::memset(&node, 0, sizeof(node));
				node.n_children = (uint8_t)k.n;
				float scale[3] = { 1.0f, 1.0f, 1.0f };
				for (int a = 0; a < 3; a++) {
					float lo = k.ch[0].mn[a], hi = k.ch[0].mx[a];
					for (uint32_t i = 1; i < k.n; i++) { lo = std::min(lo, k.ch[i].mn[a]); hi = std::max(hi, k.ch[i].mx[a]); }
					node.org[a] = lo;
					// smallest power-of-two step with (hi - lo) / step <= 254 (one step of headroom for the outward rounding)
					int e = 1;
					const double ext = (double)hi - (double)lo;
					if (!(ext >= 0.0) || !std::isfinite(ext)) { bad = true; continue; } // NaN / infinite box
					if (ext > 0.0) { e = (int)std::ceil(std::log2(ext / 254.0)) + 127; if (e < 1) e = 1; if (e > 254) e = 254; }
					for (;;) { // guard against log2 rounding: make sure the extent fits
						uint32_t bits = (uint32_t)e << 23; float s; std::memcpy(&s, &bits, 4);
						if ((double)s * 254.0 >= ext || e >= 254) { scale[a] = s; break; }
						e++;
					}
					node.exp[a] = (uint8_t)e;
				}
				for (uint32_t i = 0; i < 8; i++) {
					if (i >= k.n) { node.ref[i] = kSentinel; continue; }
					node.ref[i] = k.ch[i].ref;
					if (k.ch[i].ref >= kLeafBit && leaf_box) {
						float *lb = leaf_box + (size_t)(k.ch[i].ref & 0x7FFFFFFFu) * 8u;
						for (int a = 0; a < 3; a++) { lb[a] = k.ch[i].mn[a]; lb[4 + a] = k.ch[i].mx[a]; }
					}
					for (int a = 0; a < 3 && !bad; a++) {
						// outward rounding, verified on the value the kernel will decode: fmaf(q, scale, org)
						int ql = (int)std::floor(((double)k.ch[i].mn[a] - (double)node.org[a]) / (double)scale[a]);
						if (ql < 0) ql = 0; if (ql > 255) ql = 255;
						while (ql > 0 && std::fmaf((float)ql, scale[a], node.org[a]) > k.ch[i].mn[a]) ql--;
						int qh = (int)std::ceil(((double)k.ch[i].mx[a] - (double)node.org[a]) / (double)scale[a]);
						if (qh < 0) qh = 0; if (qh > 255) qh = 255;
						while (qh < 255 && std::fmaf((float)qh, scale[a], node.org[a]) < k.ch[i].mx[a]) qh++;
						if (std::fmaf((float)ql, scale[a], node.org[a]) > k.ch[i].mn[a] || std::fmaf((float)qh, scale[a], node.org[a]) < k.ch[i].mx[a])
							bad = true; // cannot happen for finite boxes (one step of headroom); never ship a box that does not contain its child
						node.qlo[a][i] = (uint8_t)ql; node.qhi[a][i] = (uint8_t)qh;
					}
				}
				dev8[w] = node;
			}
		};
		unsigned n_thr = std::thread::hardware_concurrency();
		if (n_thr > 16u) n_thr = 16u;
		if (n_thr < 2u || kids.size() < 4096u) quantise(0, kids.size());
		else {
			std::vector<std::thread> pool;
			for (unsigned t = 0; t < n_thr; t++)
				pool.emplace_back(quantise, kids.size() * t / n_thr, kids.size() * (t + 1) / n_thr);
			for (auto &th : pool) th.join();
		}
		ok8 = !bad;
		if (ok8) {
			out->nodes8 = (Dev8Node *)std::malloc(dev8.size() * sizeof(Dev8Node));
			if (out->nodes8) {
				out->n_nodes8 = (uint32_t)dev8.size();
				std::memcpy(out->nodes8, dev8.data(), dev8.size() * sizeof(Dev8Node));
				out->stack8 = stack8;
				out->leaf_box = leaf_box; leaf_box = nullptr;
			}
		}
		std::free(leaf_box);
	}
	return MRT_OK;
}

} // namespace mrt
