// two_level_prep.cpp -- host side of the two-level scene:
//   MeshBLAS::build      (src/accel/mesh_blas.h:86-138)   one binned-SAH BVH2 per distinct mesh, mesh space
//   BLASInstance         (src/accel/blas_instance.h:47-107) inverse transform, world box of the mesh box
//   SceneTLAS::build_tlas / refit_tlas (src/accel/scene_tlas.h:140-196) BVH2 over the instances' world boxes
// The flat ids follow RayTracerServer::_rebuild_scene (raytracer_server.cpp:700-711): an instance's
// first triangle has the running triangle count of the instances registered before it (the reference's
// own TLAS path reports mesh-local ids; the flat id is what its callers index by).
#include "../mrt_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <utility>
#include <vector>

namespace mrt {

void free_two_level(TwoLevelHost *h)
{
	if (!h) return;
	std::free(h->nodes); std::free(h->hot); std::free(h->cold); std::free(h->inst); std::free(h->blas); std::free(h->inst_blas);
	std::free(h->nodes8); std::free(h->leaf_box);
	*h = TwoLevelHost();
}

namespace {

int fail_(char *err, size_t err_len, int code, const char *msg)
{
	if (err && err_len) std::snprintf(err, err_len, "%s", msg);
	return code;
}

// Inverse of the affine map x -> B x + o in double (cofactors), rounded once to float.
// false if B is singular (or not finite).
bool invert_affine(const float basis[9], const float origin[3], float inv[12])
{
	const double a = basis[0], b = basis[1], c = basis[2], d = basis[3], e = basis[4], f = basis[5], g = basis[6], h = basis[7], i = basis[8];
	const double c00 = e * i - f * h, c01 = c * h - b * i, c02 = b * f - c * e;
	const double c10 = f * g - d * i, c11 = a * i - c * g, c12 = c * d - a * f;
	const double c20 = d * h - e * g, c21 = b * g - a * h, c22 = a * e - b * d;
	const double det = a * c00 + b * c10 + c * c20;
	if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
	const double m[9] = { c00 / det, c01 / det, c02 / det, c10 / det, c11 / det, c12 / det, c20 / det, c21 / det, c22 / det };
	for (int r = 0; r < 3; r++) {
		const double t = -(m[3 * r] * (double)origin[0] + m[3 * r + 1] * (double)origin[1] + m[3 * r + 2] * (double)origin[2]);
		inv[4 * r] = (float)m[3 * r]; inv[4 * r + 1] = (float)m[3 * r + 1]; inv[4 * r + 2] = (float)m[3 * r + 2]; inv[4 * r + 3] = (float)t;
		if (!std::isfinite(inv[4 * r]) || !std::isfinite(inv[4 * r + 1]) || !std::isfinite(inv[4 * r + 2]) || !std::isfinite(inv[4 * r + 3])) return false;
	}
	return true;
}

// World box of a mesh box under x -> B x + o: the eight corners (BLASInstance::compute_world_bounds,
// blas_instance.h:76-107), evaluated in double and rounded outwards to float so that the box still
// contains the exact image.
void world_box(const float lo[3], const float hi[3], const float basis[9], const float origin[3], float wlo[3], float whi[3])
{
	double mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (int k = 0; k < 8; k++) {
		const double x = (k & 1) ? hi[0] : lo[0], y = (k & 2) ? hi[1] : lo[1], z = (k & 4) ? hi[2] : lo[2];
		for (int r = 0; r < 3; r++) {
			const double w = ((double)basis[3 * r] * x + (double)basis[3 * r + 1] * y) + (double)basis[3 * r + 2] * z + (double)origin[r];
			if (w < mn[r]) mn[r] = w;
			if (w > mx[r]) mx[r] = w;
		}
	}
	for (int r = 0; r < 3; r++) {
		float l = (float)mn[r], u = (float)mx[r];
		if ((double)l > mn[r]) l = std::nextafterf(l, -INFINITY);
		if ((double)u < mx[r]) u = std::nextafterf(u, INFINITY);
		wlo[r] = l; whi[r] = u;
	}
}

// BVH2 over n boxes with the scene builder: box k goes in as the triangle {lo, hi, centre}, whose
// AABB is the box.  Result in the device layout via prepare_scene (ids = box index, last-in-leaf flags).
int build_over_boxes(const float *lo, const float *hi, uint32_t n, DeviceSceneHost *out, char *err, size_t err_len)
{
	std::vector<float> v9((size_t)n * 9), v4((size_t)n * 12);
	for (uint32_t k = 0; k < n; k++) {
		for (int c = 0; c < 3; c++) {
			const float a = lo[3 * k + c], b = hi[3 * k + c], m = 0.5f * a + 0.5f * b;
			v9[(size_t)k * 9 + c] = a; v9[(size_t)k * 9 + 3 + c] = b; v9[(size_t)k * 9 + 6 + c] = m;
			v4[(size_t)k * 12 + c] = a; v4[(size_t)k * 12 + 4 + c] = b; v4[(size_t)k * 12 + 8 + c] = m;
		}
		v4[(size_t)k * 12 + 3] = v4[(size_t)k * 12 + 7] = v4[(size_t)k * 12 + 11] = 0.0f;
	}
	std::vector<mrt_tri64> tris(n);
	int rc = mrt_make_triangles(v9.data(), nullptr, nullptr, n, tris.data());
	if (rc) return fail_(err, err_len, rc, "two-level scene: bad instance box");
	std::vector<mrt_bvh_node32> nodes((size_t)2 * n);
	std::vector<uint32_t> prim(n);
	uint32_t used = 0;
	if ((rc = mrt_bvh2_build(v4.data(), n, nodes.data(), prim.data(), &used, 1))) return fail_(err, err_len, rc, "two-level scene: TLAS build failed");
	return prepare_scene(tris.data(), n, nodes.data(), used, prim.data(), out, err, err_len);
}

void free_scene_host(DeviceSceneHost &s)
{
	std::free(s.nodes); std::free(s.nodes4); std::free(s.nodes8); std::free(s.leaf_box); std::free(s.hot); std::free(s.cold);
	s = DeviceSceneHost();
}

} // namespace

int refit_two_level(TwoLevelHost *h, const mrt_instance *instances, uint32_t n, char *err, size_t err_len)
{
	if (!h || !instances || n != h->n_inst) return fail_(err, err_len, MRT_ERR_INVALID, "two-level scene: the instance count of a refit must match the upload");
	std::vector<float> lo((size_t)n * 3), hi((size_t)n * 3);
	std::vector<DevInstance> reg(n);
	uint64_t id_base = 0;
	for (uint32_t i = 0; i < n; i++) {
		const mrt_instance &in = instances[i];
		const TwoLevelBlas &b = h->blas[h->inst_blas[i]];
		if (in.first_tri != b.first_tri || in.n_tris != b.n_tris) return fail_(err, err_len, MRT_ERR_INVALID, "two-level scene: a refit may move instances, not change their meshes");
		DevInstance &d = reg[i];
		std::memset(&d, 0, sizeof(d));
		if (!invert_affine(in.basis, in.origin, d.inv)) return fail_(err, err_len, MRT_ERR_INVALID, "two-level scene: singular instance transform");
		for (int k = 0; k < 9; k++) d.basis[k] = in.basis[k];
		d.root = b.root; d.root8 = b.root8; d.id_base = (uint32_t)id_base; d.layers = in.layers; d.index = i;
		id_base += in.n_tris;
		world_box(b.lo, b.hi, in.basis, in.origin, &lo[(size_t)3 * i], &hi[(size_t)3 * i]);
	}
	DeviceSceneHost t;
	int rc = build_over_boxes(lo.data(), hi.data(), n, &t, err, err_len);
	if (rc) return rc;
	if (t.n_nodes > h->tlas_cap) { free_scene_host(t); return fail_(err, err_len, MRT_ERR_BAD_BVH, "two-level scene: TLAS larger than its reserved range"); }
	uint32_t max_blas = 0, max_blas8 = 0;
	for (uint32_t k = 0; k < h->n_blas; k++) {
		if (h->blas[k].depth > max_blas) max_blas = h->blas[k].depth;
		if (h->blas[k].stack8 > max_blas8) max_blas8 = h->blas[k].stack8;
	}
	// TLAS nodes keep their indices (the TLAS occupies the front of the array); leaf refs already are instance slots
	std::memcpy(h->nodes, t.nodes, (size_t)t.n_nodes * sizeof(DevNode));
	h->n_tlas_nodes = t.n_nodes;
	for (uint32_t k = 0; k < n; k++) { // leaf order: slot k holds instance hot[k].id
		h->inst[k] = reg[t.hot[k].id];
		h->inst[k].flags = t.hot[k].flags & kLastInLeaf;
	}
	// pending entries: TLAS path (t.depth incl. sentinel) + the rest of a TLAS leaf + the return marker + a BLAS path
	h->depth = t.depth + 2u + max_blas;
	h->depth8 = t.depth + 2u + max_blas8;
	free_scene_host(t);
	return MRT_OK;
}

int prepare_two_level(const float *verts9, uint32_t n_mesh_tris, const mrt_instance *instances, uint32_t n,
		uint32_t n_threads, bool build_blas, TwoLevelHost *out, char *err, size_t err_len)
{
	if (!verts9 || !instances || !out || n == 0 || n_mesh_tris == 0) return fail_(err, err_len, MRT_ERR_INVALID, "two-level scene: null or empty argument");
	*out = TwoLevelHost();
	// distinct meshes, in order of first use
	std::map<std::pair<uint32_t, uint32_t>, uint32_t> seen;
	std::vector<TwoLevelBlas> blas;
	std::vector<uint32_t> inst_blas(n);
	uint64_t flat = 0, unique_tris = 0;
	for (uint32_t i = 0; i < n; i++) {
		const mrt_instance &in = instances[i];
		if (in.n_tris == 0 || in.first_tri >= n_mesh_tris || in.n_tris > n_mesh_tris - in.first_tri)
			return fail_(err, err_len, MRT_ERR_INVALID, "two-level scene: an instance's triangle range lies outside the mesh array");
		const auto key = std::make_pair(in.first_tri, in.n_tris);
		auto it = seen.find(key);
		if (it == seen.end()) {
			it = seen.emplace(key, (uint32_t)blas.size()).first;
			TwoLevelBlas b{}; b.first_tri = in.first_tri; b.n_tris = in.n_tris;
			blas.push_back(b);
			unique_tris += in.n_tris;
		}
		inst_blas[i] = it->second;
		flat += in.n_tris;
	}
	if (flat >= 0x7FFFFFFFull || unique_tris >= 0x7FFFFFFFull) return fail_(err, err_len, MRT_ERR_UNSUPPORTED, "two-level scene: more than 2^31-1 triangles");
	const uint32_t tlas_cap = 2u * n; // a BVH2 over n leaves has fewer than n inner nodes; the wrapped root leaf needs 1
	if (!build_blas) {
		// the caller builds every BLAS on the device (an LBVH over k >= 2 triangles has k - 1 nodes, one
		// triangle per leaf) at the roots laid out here, fills in blas[].lo/hi/depth and calls refit_two_level
		TwoLevelHost h;
		uint64_t n_nodes = tlas_cap;
		for (auto &b : blas) {
			if (b.n_tris < 2) return fail_(err, err_len, MRT_ERR_UNSUPPORTED, "two-level scene: a mesh of one triangle needs the host builder");
			b.root = (uint32_t)n_nodes; b.root8 = b.root - tlas_cap; n_nodes += b.n_tris - 1u; // the 8-wide node of a binary node sits at the same offset
		}
		if (n_nodes >= kInstanceReturn) return fail_(err, err_len, MRT_ERR_UNSUPPORTED, "two-level scene: too many nodes");
		h.nodes = (DevNode *)std::calloc((size_t)tlas_cap, sizeof(DevNode));
		h.inst = (DevInstance *)std::calloc(n, sizeof(DevInstance));
		h.blas = (TwoLevelBlas *)std::malloc(blas.size() * sizeof(TwoLevelBlas));
		h.inst_blas = (uint32_t *)std::malloc((size_t)n * sizeof(uint32_t));
		if (!h.nodes || !h.inst || !h.blas || !h.inst_blas) { free_two_level(&h); return fail_(err, err_len, MRT_ERR_OOM, "two-level scene: out of host memory"); }
		h.n_nodes = (uint32_t)n_nodes; h.tlas_cap = tlas_cap; h.n_tris = (uint32_t)unique_tris; h.n_inst = n; h.n_blas = (uint32_t)blas.size();
		h.flat_tris = flat;
		std::memcpy(h.blas, blas.data(), blas.size() * sizeof(TwoLevelBlas));
		std::memcpy(h.inst_blas, inst_blas.data(), (size_t)n * sizeof(uint32_t));
		*out = h;
		return MRT_OK;
	}
	// every BLAS: triangles with mesh-local ids and all layers (the mask lives in the instance), SAH BVH2, device layout
	std::vector<DeviceSceneHost> built(blas.size());
	auto cleanup = [&] { for (auto &s : built) free_scene_host(s); };
	uint64_t n_nodes = tlas_cap;
	for (size_t k = 0; k < blas.size(); k++) {
		const uint32_t nt = blas[k].n_tris;
		const float *src = verts9 + (size_t)9 * blas[k].first_tri;
		std::vector<mrt_tri64> tris(nt);
		int rc = mrt_make_triangles(src, nullptr, nullptr, nt, tris.data());
		if (rc) { cleanup(); return fail_(err, err_len, rc, "two-level scene: bad mesh triangles"); }
		std::vector<float> v4((size_t)nt * 12);
		for (size_t t = 0; t < (size_t)nt * 3; t++) { v4[4 * t] = src[3 * t]; v4[4 * t + 1] = src[3 * t + 1]; v4[4 * t + 2] = src[3 * t + 2]; v4[4 * t + 3] = 0.0f; }
		std::vector<mrt_bvh_node32> nodes((size_t)2 * nt);
		std::vector<uint32_t> prim(nt);
		uint32_t used = 0;
		if ((rc = mrt_bvh2_build(v4.data(), nt, nodes.data(), prim.data(), &used, n_threads))) { cleanup(); return fail_(err, err_len, rc, "two-level scene: BLAS build failed"); }
		built[k].want8 = true;
		if ((rc = prepare_scene(tris.data(), nt, nodes.data(), used, prim.data(), &built[k], err, err_len))) { cleanup(); return rc; }
		blas[k].depth = built[k].depth;
		for (int c = 0; c < 3; c++) { blas[k].lo[c] = built[k].bounds_lo[c]; blas[k].hi[c] = built[k].bounds_hi[c]; }
		n_nodes += built[k].n_nodes;
	}
	if (n_nodes >= kInstanceReturn) { cleanup(); return fail_(err, err_len, MRT_ERR_UNSUPPORTED, "two-level scene: too many nodes"); }
	TwoLevelHost h;
	h.nodes = (DevNode *)std::calloc((size_t)n_nodes, sizeof(DevNode));
	h.hot = (TriHot *)std::malloc((size_t)unique_tris * sizeof(TriHot));
	h.cold = (TriCold *)std::malloc((size_t)unique_tris * sizeof(TriCold));
	h.inst = (DevInstance *)std::calloc(n, sizeof(DevInstance));
	h.blas = (TwoLevelBlas *)std::malloc(blas.size() * sizeof(TwoLevelBlas));
	h.inst_blas = (uint32_t *)std::malloc((size_t)n * sizeof(uint32_t));
	if (!h.nodes || !h.hot || !h.cold || !h.inst || !h.blas || !h.inst_blas) { cleanup(); free_two_level(&h); return fail_(err, err_len, MRT_ERR_OOM, "two-level scene: out of host memory"); }
	h.n_nodes = (uint32_t)n_nodes; h.tlas_cap = tlas_cap; h.n_tris = (uint32_t)unique_tris; h.n_inst = n; h.n_blas = (uint32_t)blas.size();
	h.flat_tris = flat;
	// the 8-wide layout goes along if every BLAS has it (a non-finite box leaves a mesh without one)
	uint64_t n8 = 0;
	bool all8 = true;
	for (size_t k = 0; k < blas.size(); k++) { all8 = all8 && built[k].nodes8 && built[k].leaf_box; n8 += built[k].n_nodes8; }
	if (all8 && n8 < kInstanceReturn) {
		h.nodes8 = (Dev8Node *)std::malloc((size_t)n8 * sizeof(Dev8Node));
		h.leaf_box = (float *)std::malloc((size_t)unique_tris * 8 * sizeof(float));
		if (h.nodes8 && h.leaf_box) { h.wide8 = true; h.n_nodes8 = (uint32_t)n8; }
		else { std::free(h.nodes8); std::free(h.leaf_box); h.nodes8 = nullptr; h.leaf_box = nullptr; }
	}
	uint32_t node_base = tlas_cap, tri_base = 0, base8 = 0;
	for (size_t k = 0; k < blas.size(); k++) { // concatenate, making node and leaf refs global
		const DeviceSceneHost &s = built[k];
		auto fix = [&](uint32_t ref) { return ref < kSentinel ? ref + node_base : (ref >= kLeafBit ? (kLeafBit | ((ref & 0x7FFFFFFFu) + tri_base)) : ref); };
		for (uint32_t w = 0; w < s.n_nodes; w++) {
			DevNode g = s.nodes[w];
			g.left_ref = fix(g.left_ref); g.right_ref = fix(g.right_ref);
			h.nodes[node_base + w] = g;
		}
		std::memcpy(h.hot + tri_base, s.hot, (size_t)s.n_tris * sizeof(TriHot));
		std::memcpy(h.cold + tri_base, s.cold, (size_t)s.n_tris * sizeof(TriCold));
		if (h.wide8) {
			auto fix8 = [&](uint32_t ref) { return ref < kSentinel ? ref + base8 : (ref >= kLeafBit ? (kLeafBit | ((ref & 0x7FFFFFFFu) + tri_base)) : ref); };
			for (uint32_t w = 0; w < s.n_nodes8; w++) {
				Dev8Node g = s.nodes8[w];
				for (int c = 0; c < 8; c++) g.ref[c] = fix8(g.ref[c]);
				h.nodes8[base8 + w] = g;
			}
			std::memcpy(h.leaf_box + (size_t)tri_base * 8, s.leaf_box, (size_t)s.n_tris * 8 * sizeof(float));
			blas[k].root8 = base8; blas[k].stack8 = s.stack8;
			base8 += s.n_nodes8;
		}
		blas[k].root = node_base;
		node_base += s.n_nodes; tri_base += s.n_tris;
	}
	cleanup();
	std::memcpy(h.blas, blas.data(), blas.size() * sizeof(TwoLevelBlas));
	std::memcpy(h.inst_blas, inst_blas.data(), (size_t)n * sizeof(uint32_t));
	const int rc = refit_two_level(&h, instances, n, err, err_len);
	if (rc) { free_two_level(&h); return rc; }
	*out = h;
	return MRT_OK;
}

} // namespace mrt

// ---- the prepared scene on the host, through the C-ABI (include/mrt_hip.h: mrt_two_level_prepare_host) ----------------
struct mrt_two_level_host { mrt::TwoLevelHost h; };

extern "C" {

int mrt_two_level_prepare_host(const float *verts9, uint32_t n_mesh_tris, const mrt_instance *instances, uint32_t n_instances,
		uint32_t n_threads, mrt_two_level_host **out)
{
	if (!out) return MRT_ERR_INVALID;
	*out = nullptr;
	if (!verts9 || !instances || n_instances == 0 || n_mesh_tris == 0) return MRT_ERR_INVALID;
	mrt_two_level_host *w = new (std::nothrow) mrt_two_level_host();
	if (!w) return MRT_ERR_OOM;
	char err[256];
	const int rc = mrt::prepare_two_level(verts9, n_mesh_tris, instances, n_instances, n_threads, true, &w->h, err, sizeof(err));
	if (rc != MRT_OK) { mrt::free_two_level(&w->h); delete w; return rc; }
	*out = w;
	return MRT_OK;
}

int mrt_two_level_host_arrays(const mrt_two_level_host *w, mrt_two_level_arrays *out)
{
	if (!w || !out) return MRT_ERR_INVALID;
	static_assert(sizeof(mrt::DevNode) == sizeof(mrt_bvh_node_wide64), "the device node is the wide node");
	out->nodes = reinterpret_cast<const mrt_bvh_node_wide64 *>(w->h.nodes); out->n_nodes = w->h.n_nodes; out->n_tlas_nodes = w->h.n_tlas_nodes;
	out->tri_hot = reinterpret_cast<const float *>(w->h.hot); out->tri_cold = reinterpret_cast<const float *>(w->h.cold); out->n_tris = w->h.n_tris;
	out->instances = reinterpret_cast<const float *>(w->h.inst); out->n_instances = w->h.n_inst;
	out->depth = w->h.depth;
	return MRT_OK;
}

void mrt_two_level_free_host(mrt_two_level_host *w)
{
	if (!w) return;
	mrt::free_two_level(&w->h);
	delete w;
}

} // extern "C"

