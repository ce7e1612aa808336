// ray_tracer_server.hpp — the semantics of RayTracerServer's ray API (src/godot/raytracer_server.cpp) over the
// RayDispatcher mirror, without Godot: SURVEY.md 8(a) row a16 and Appendix B.
//
// Same names and argument meaning as the GDExtension singleton, Variant types replaced by plain ones
// (Dictionary -> RayHit, Node -> vertices + Transform3D):
//   register_mesh / unregister_mesh / clear / build            (:83-181; _rebuild_scene :669-711: every mesh's
//        triangles to world space by Transform3D::xform, ids = running triangle offset in registration order,
//        layers = the mesh's layer mask)
//   cast_ray(origin, direction, layer_mask = 0x7FFFFFFF)        (:253-272: direction normalised, RayHit fields =
//        the Dictionary's keys hit / position / normal / distance / prim_id / hit_layers)
//   any_hit(origin, direction, max_distance, layer_mask)        (:274-283: t_max = max_distance)
//   cast_rays_batch(rays, results, count, stats, query_mask)    (:285-289)
//   submit(RayQuery, RayQueryResult)                            (:295-328, src/api/ray_query.h:52-118: mode, layer
//        mask, coherent hint, collect_stats, elapsed_ms of the dispatch)
//   set_backend / get_backend with BACKEND_CPU = 0, BACKEND_GPU = 1, BACKEND_AUTO = 2 (:334-366: lazy GPU
//        initialisation + upload), is_gpu_available, get_triangle_count, get_mesh_count, get_bvh_node_count,
//        get_bvh_depth, get_thread_count, get_last_cast_ms.
// One deliberate difference: the reference's set_backend(BACKEND_GPU) prints "GPU init failed -- falling back to
// CPU" and switches the mode; here a failed initialisation leaves the mode on GPU and every cast reports
// MRT_ERR_NO_DEVICE (last_status()): the device path fails loudly, the CPU backend runs only when it is selected.
// set_cpu_fallback(true) (off by default) restores the reference's behaviour: a failed initialisation prints the
// reference's message and switches the mode to BACKEND_CPU (:346-355), and BACKEND_AUTO without a device casts on the pool.
// Locking (shared_mutex around casts, unique around builds, :162,257,287,302) is kept.
#pragma once
#include <chrono>
#include <shared_mutex>
#include <vector>
#include "ray_dispatcher.hpp"

namespace mrt {

// Transform3D: world = basis * v + origin, basis row-major (godot::Basis rows)
struct Transform3D {
	float basis[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	Vector3 origin;
	Vector3 xform(const Vector3 &v) const // Basis::xform + origin: one dot product per row, summed left to right
	{
		return Vector3(basis[0] * v.x + basis[1] * v.y + basis[2] * v.z + origin.x,
				basis[3] * v.x + basis[4] * v.y + basis[5] * v.z + origin.y,
				basis[6] * v.x + basis[7] * v.y + basis[8] * v.z + origin.z);
	}
};

// src/api/ray_query.h:52-118
struct RayQuery {
	enum Mode { NEAREST = 0, ANY_HIT = 1 };
	const Ray *rays = nullptr;
	int count = 0;
	uint32_t layer_mask = 0xFFFFFFFF;
	Mode mode = NEAREST;
	bool collect_stats = false;
	bool coherent = false;
	static RayQuery nearest(const Ray *rays, int count, uint32_t layer_mask = 0xFFFFFFFF)
	{
		RayQuery q; q.rays = rays; q.count = count; q.layer_mask = layer_mask; q.mode = NEAREST; return q;
	}
	static RayQuery any_hit(const Ray *rays, int count, uint32_t layer_mask = 0xFFFFFFFF)
	{
		RayQuery q; q.rays = rays; q.count = count; q.layer_mask = layer_mask; q.mode = ANY_HIT; return q;
	}
};
struct RayQueryResult {
	Intersection *hits = nullptr;
	bool *hit_flags = nullptr;
	int count = 0;
	RayStats stats;
	float elapsed_ms = 0.0f;
};

// the Dictionary of RayTracerServer::cast_ray (:262-269)
struct RayHit {
	bool hit = false;
	Vector3 position, normal;
	float distance = FLT_MAX;
	int prim_id = -1;
	int hit_layers = 0;
};

class RayTracerServer {
public:
	enum BackendMode { BACKEND_CPU = 0, BACKEND_GPU = 1, BACKEND_AUTO = 2 };

	// ---- scene (raytracer_server.cpp:83-181) ----
	// mesh-space vertices, 3 per triangle; returns the mesh id
	int register_mesh(const std::vector<Vector3> &verts, const Transform3D &xf, uint32_t layer_mask = 0xFFFFFFFF)
	{
		std::unique_lock<std::shared_mutex> lock(scene_mutex_);
		meshes_.push_back(Mesh{verts, xf, layer_mask, true});
		scene_dirty_ = true;
		return (int)meshes_.size() - 1;
	}
	void unregister_mesh(int mesh_id)
	{
		std::unique_lock<std::shared_mutex> lock(scene_mutex_);
		if (mesh_id < 0 || mesh_id >= (int)meshes_.size() || !meshes_[mesh_id].valid) return;
		meshes_[mesh_id].valid = false; meshes_[mesh_id].verts.clear();
		scene_dirty_ = true;
	}
	void build()
	{
		std::unique_lock<std::shared_mutex> lock(scene_mutex_);
		_rebuild_scene();
	}
	void clear()
	{
		std::unique_lock<std::shared_mutex> lock(scene_mutex_);
		meshes_.clear(); dispatcher_.scene().clear(); scene_dirty_ = false;
	}

	// ---- ray casting (:253-289) ----
	RayHit cast_ray(const Vector3 &origin, const Vector3 &direction, int layer_mask = 0x7FFFFFFF)
	{
		std::shared_lock<std::shared_mutex> lock(scene_mutex_);
		RayHit result;
		if (!(direction.length_squared() > 0.0f) || !origin.is_finite()) return result; // RT_ASSERTs of :255-256
		Ray r(origin, direction.normalized());
		const Intersection hit = dispatcher_.cast_ray(r, nullptr, (uint32_t)layer_mask);
		result.hit = hit.hit(); result.position = hit.position; result.normal = hit.normal; result.distance = hit.t;
		result.prim_id = (int)hit.prim_id; result.hit_layers = (int)hit.hit_layers;
		return result;
	}
	bool any_hit(const Vector3 &origin, const Vector3 &direction, float max_distance, int layer_mask = 0x7FFFFFFF)
	{
		std::shared_lock<std::shared_mutex> lock(scene_mutex_);
		if (!(direction.length_squared() > 0.0f) || !(max_distance > 0.0f)) return false;
		Ray r(origin, direction.normalized());
		r.t_max = max_distance;
		return dispatcher_.any_hit(r, nullptr, (uint32_t)layer_mask);
	}
	int cast_rays_batch(const Ray *rays, Intersection *results, int count, RayStats *stats = nullptr, uint32_t query_mask = 0xFFFFFFFF)
	{
		std::shared_lock<std::shared_mutex> lock(scene_mutex_);
		return last_status_ = dispatcher_.cast_rays(rays, results, count, stats, query_mask);
	}
	// :295-328
	int submit(const RayQuery &query, RayQueryResult &result)
	{
		if (query.rays == nullptr || query.count <= 0) return last_status_ = MRT_ERR_INVALID; // ERR_FAIL_COND_MSG
		std::shared_lock<std::shared_mutex> lock(scene_mutex_);
		const auto t0 = std::chrono::steady_clock::now();
		RayStats *stats_ptr = query.collect_stats ? &result.stats : nullptr;
		int rc = MRT_ERR_INVALID;
		switch (query.mode) {
			case RayQuery::NEAREST:
				if (result.hits == nullptr) return last_status_ = MRT_ERR_INVALID;
				rc = dispatcher_.cast_rays(query.rays, result.hits, query.count, stats_ptr, query.layer_mask, query.coherent);
				break;
			case RayQuery::ANY_HIT:
				if (result.hit_flags == nullptr) return last_status_ = MRT_ERR_INVALID;
				rc = dispatcher_.any_hit_rays(query.rays, result.hit_flags, query.count, stats_ptr, query.layer_mask, query.coherent);
				break;
		}
		const auto t1 = std::chrono::steady_clock::now();
		result.elapsed_ms = last_cast_ms_ = std::chrono::duration<float, std::milli>(t1 - t0).count();
		result.count = rc == MRT_OK ? query.count : 0;
		return last_status_ = rc;
	}

	// ---- backend control (:334-366) ----
	void set_backend(int mode, int device_ordinal = 0)
	{
		if (mode < 0 || mode > (int)BACKEND_AUTO) return;
		backend_mode_ = (BackendMode)mode;
		switch (backend_mode_) {
			case BACKEND_CPU: dispatcher_.set_backend(RayDispatcher::Backend::CPU); break;
			case BACKEND_GPU:
				dispatcher_.set_backend(RayDispatcher::Backend::GPU);
				if (!dispatcher_.gpu_available()) {
					if (!dispatcher_.initialize_gpu(device_ordinal)) {
						if (dispatcher_.cpu_fallback()) { // opt-in: raytracer_server.cpp:346-355
							std::fprintf(stderr, "[RayTracerServer] GPU init failed -- falling back to CPU\n");
							backend_mode_ = BACKEND_CPU; dispatcher_.set_backend(RayDispatcher::Backend::CPU);
							return;
						}
						last_status_ = MRT_ERR_NO_DEVICE; return; // stays GPU: casts fail loudly
					}
					dispatcher_.upload_to_gpu();
				}
				break;
			case BACKEND_AUTO:
				dispatcher_.set_backend(RayDispatcher::Backend::AUTO);
				if (!dispatcher_.gpu_available() && dispatcher_.initialize_gpu(device_ordinal)) dispatcher_.upload_to_gpu(); // best effort
				break;
		}
	}
	int get_backend() const { return (int)backend_mode_; }
	void set_cpu_fallback(bool on) { dispatcher_.set_cpu_fallback(on); } // opt-in, see the header comment
	bool is_gpu_available() const { return dispatcher_.gpu_available(); }
	int get_triangle_count() const { return dispatcher_.triangle_count(); }
	int get_mesh_count() const { int n = 0; for (const auto &m : meshes_) n += m.valid ? 1 : 0; return n; }
	int get_bvh_node_count() const { return dispatcher_.bvh_node_count(); }
	int get_bvh_depth() const { return dispatcher_.bvh_depth(); }
	int get_thread_count() const { return (int)dispatcher_.thread_count(); }
	float get_last_cast_ms() const { return last_cast_ms_; }
	int last_status() const { return last_status_; }
	RayDispatcher &dispatcher() { return dispatcher_; }

private:
	struct Mesh { std::vector<Vector3> verts; Transform3D xf; uint32_t layer_mask; bool valid; };
	std::vector<Mesh> meshes_;
	RayDispatcher dispatcher_;
	BackendMode backend_mode_ = BACKEND_CPU;
	bool scene_dirty_ = false;
	float last_cast_ms_ = 0.0f;
	int last_status_ = MRT_OK;
	mutable std::shared_mutex scene_mutex_;

	// :669-711: world triangle = xform(v0, v1, v2), id = running tri_offset over the valid meshes in registration order,
	// layers = the mesh's layer mask; then RayDispatcher::build (BVH + upload when a GPU backend is active)
	void _rebuild_scene()
	{
		RayScene &sc = dispatcher_.scene();
		sc.clear();
		uint32_t tri_offset = 0;
		for (const Mesh &m : meshes_) {
			if (!m.valid) continue;
			for (size_t i = 0; i + 2 < m.verts.size(); i += 3)
				sc.triangles.push_back(Triangle(m.xf.xform(m.verts[i]), m.xf.xform(m.verts[i + 1]), m.xf.xform(m.verts[i + 2]), tri_offset++, m.layer_mask));
		}
		dispatcher_.build();
		scene_dirty_ = false;
	}
};

} // namespace mrt
