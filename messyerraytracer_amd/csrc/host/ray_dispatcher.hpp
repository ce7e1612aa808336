// ray_dispatcher.hpp — RayDispatcher (src/dispatch/ray_dispatcher.h:38-464) for
// the MI355X backend.
//
// Same names and argument meaning as the reference's router: Backend
// {CPU, GPU, AUTO}, scene(), build(), set_backend / get_backend,
// gpu_available / gpu_initialized / initialize_gpu / upload_to_gpu / using_gpu,
// cast_rays / any_hit_rays / cast_ray / any_hit, submit_gpu_async /
// collect_gpu_nearest / submit_gpu_async_any_hit / collect_gpu_any_hit /
// has_gpu_pending, triangle_count / bvh_node_count / bvh_depth.
//
// Backends.  GPU: the MI355X path through the C-ABI (gpu_ray_caster.hpp).  CPU: what the reference routes to its
// ThreadPool + TinyBVH (ray_dispatcher.h:152-180,214-240,443-463) — here cpu_backend.hpp: the same range split
// over a persistent pool, per-chunk RayStats merged afterwards, a BVH2 walk with the device kernels' arithmetic over the
// arrays the device is fed, so both backends return the same records.  It is an EXPLICIT backend: the caller
// selects Backend::CPU (the reference's default, kept).  Nothing degrades to it silently: Backend::GPU without a
// device, and Backend::AUTO without one, return MRT_ERR_NO_DEVICE and say so (the reference's AUTO falls back to
// its CPU pool; this library's rule is that the device path fails loudly).  The casts therefore return an int status
// (the reference's return void).  A host that wants the reference's behaviour asks for it: set_cpu_fallback(true)
// makes every cast that finds no usable device take the CPU pool, exactly as ray_dispatcher.h:130,152-180,214-240,
// 247-273 do when using_gpu() is false (one line on stderr the first time it happens; used_cpu_fallback() tells).
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>
#include "cpu_backend.hpp"
#include "gpu_ray_caster.hpp"

namespace mrt {

// RayScene (src/accel/ray_scene.h:34-60): triangles + the BVH2 that build() makes.
struct RayScene {
	std::vector<Triangle> triangles;
	std::vector<float> vertices;           // 3 x bvhvec4 per triangle (tinybvh_adapter.h:42-55)
	std::vector<mrt_bvh_node32> bvh2;      // tinybvh::BVH::bvhNode
	std::vector<uint32_t> prim_idx;        // tinybvh::BVH::primIdx
	uint32_t used_nodes = 0;
	bool built = false;

	void build() // RayScene::build, ray_scene.h:62-86 (BVH2 only: the device layout is made at upload)
	{
		if (triangles.empty()) return;
		const uint32_t n = (uint32_t)triangles.size();
		vertices.assign((size_t)n * 12, 0.0f);
		for (uint32_t i = 0; i < n; i++) {
			const Triangle &t = triangles[i];
			float *v = &vertices[(size_t)i * 12];
			v[0] = t.v0.x; v[1] = t.v0.y; v[2] = t.v0.z;
			v[4] = t.v1.x; v[5] = t.v1.y; v[6] = t.v1.z;
			v[8] = t.v2.x; v[9] = t.v2.y; v[10] = t.v2.z;
		}
		bvh2.assign((size_t)2 * n + 2, mrt_bvh_node32{});
		prim_idx.assign(n, 0u);
		built = mrt_bvh2_build(vertices.data(), n, bvh2.data(), prim_idx.data(), &used_nodes, 0) == MRT_OK;
	}
	void clear() { triangles.clear(); vertices.clear(); built = false; } // ray_scene.h:198-205
	int triangle_count() const { return (int)triangles.size(); }
};

// SceneTLAS (src/accel/scene_tlas.h:140-251) with its MeshBLAS / BLASInstance parts (mesh_blas.h:86-138,
// blas_instance.h:47-107): mesh-space triangles stored once, placed instances on top.  build_tlas() prepares the
// two-level arrays on the host (one binned-SAH BVH per distinct mesh, one over the instances' world boxes: exactly what
// the device is fed); set_instance_transform + refit_tlas = new transforms, meshes untouched.
struct SceneTLAS {
	std::vector<float> mesh_vertices;        // 9 floats per triangle, mesh space, all meshes back to back
	std::vector<mrt_instance> instances;     // the mesh (first_tri, n_tris), its Transform3D and layer mask, in registration order
	SceneTLAS() = default;
	SceneTLAS(const SceneTLAS &) = delete;
	SceneTLAS &operator=(const SceneTLAS &) = delete;
	~SceneTLAS() { clear_built(); }
	bool build_tlas(uint32_t n_threads = 0)
	{
		clear_built();
		if (instances.empty() || mesh_vertices.empty()) return false;
		if (mrt_two_level_prepare_host(mesh_vertices.data(), (uint32_t)(mesh_vertices.size() / 9), instances.data(), (uint32_t)instances.size(),
				n_threads, &host_) != MRT_OK) return false;
		return mrt_two_level_host_arrays(host_, &arrays_) == MRT_OK;
	}
	bool is_built() const { return host_ != nullptr; }
	const mrt_two_level_arrays &arrays() const { return arrays_; }
	int instance_count() const { return (int)instances.size(); }

private:
	mrt_two_level_host *host_ = nullptr;
	mrt_two_level_arrays arrays_{};
	void clear_built() { if (host_) { mrt_two_level_free_host(host_); host_ = nullptr; } arrays_ = mrt_two_level_arrays{}; }
};

class RayDispatcher {
public:
	enum class Backend { CPU, GPU, AUTO }; // ray_dispatcher.h:40-44

	RayScene &scene() { return scene_; }
	const RayScene &scene() const { return scene_; }

	// ray_dispatcher.h:82-88: with a TLAS set, the CPU path walks the two-level scene (_cpu_cast_rays, :443-452); here the
	// device backend does too (the scene is uploaded as a two-level scene: nothing flattened).  nullptr = flat scene again.
	void set_tlas(SceneTLAS *tlas) { tlas_ = tlas; }
	bool has_tlas() const { return tlas_ != nullptr && tlas_->is_built(); }

	void build() // ray_dispatcher.h:67-80
	{
		scene_.build();
		// A device backend with an initialised context uploads.  (The reference asks _should_use_gpu() here, which for
		// AUTO is is_available() = "a scene is already uploaded": its AUTO never reaches the device on a first build,
		// ray_dispatcher.h:74-79 with :433.  Here AUTO with a context uploads like GPU does.)
		if (backend_ != Backend::CPU && gpu_caster_.is_initialized()) upload_to_gpu();
	}

	void set_backend(Backend b) { backend_ = b; }
	Backend get_backend() const { return backend_; }
	// Opt-in: Backend::GPU / AUTO without a usable device hand the batch to the CPU pool (the reference's routing,
	// ray_dispatcher.h:130,152-180) instead of returning MRT_ERR_NO_DEVICE.  Off by default.
	void set_cpu_fallback(bool on) { cpu_fallback_ = on; }
	bool cpu_fallback() const { return cpu_fallback_; }
	bool used_cpu_fallback() const { return fell_back_; }
	bool gpu_available() const { return gpu_caster_.is_available(); }
	bool gpu_initialized() const { return gpu_caster_.is_initialized(); }
	bool initialize_gpu(int device_ordinal = 0) { return gpu_caster_.initialize(device_ordinal); }
	void upload_to_gpu() // ray_dispatcher.h:99-107
	{
		if (gpu_caster_.is_initialized() && has_tlas())
			gpu_caster_.upload_two_level_scene(tlas_->mesh_vertices.data(), (uint32_t)(tlas_->mesh_vertices.size() / 9), tlas_->instances);
		else if (gpu_caster_.is_initialized() && scene_.built)
			gpu_caster_.upload_scene(scene_.triangles, scene_.bvh2.data(), scene_.used_nodes, scene_.prim_idx.data());
	}
	bool using_gpu() const { return _should_use_gpu() && gpu_caster_.is_available(); }

	// ray_dispatcher.h:124-181.  stats: only rays_cast is maintained on the GPU path.
	int cast_rays(const Ray *rays, Intersection *results, int count, RayStats *stats = nullptr,
			uint32_t query_mask = 0xFFFFFFFF, bool coherent = false)
	{
		if (count < 0 || (count > 0 && (!rays || !results))) return MRT_ERR_INVALID;
		if (count == 0) return MRT_OK; // a silent no-op on every backend (gpu_ray_caster.cpp:419)
		if (_route_to_cpu("cast_rays")) {
			return _cpu_dispatch(count, stats, [&](const auto &w, int i, RayStats *s) { results[i] = w.cast(rays[i], query_mask, false, s); });
		}
		if (!using_gpu()) return MRT_ERR_NO_DEVICE;
		if (!coherent && count >= MIN_BATCH_FOR_SORTING) gpu_caster_.cast_rays_sorted(rays, results, count, query_mask);
		else gpu_caster_.cast_rays(rays, results, count, query_mask);
		if (stats) stats->rays_cast += (uint64_t)count;
		return gpu_caster_.last_status();
	}
	// ray_dispatcher.h:191-241
	int any_hit_rays(const Ray *rays, bool *hit_results, int count, RayStats *stats = nullptr,
			uint32_t query_mask = 0xFFFFFFFF, bool coherent = false)
	{
		if (count < 0 || (count > 0 && (!rays || !hit_results))) return MRT_ERR_INVALID;
		if (count == 0) return MRT_OK;
		if (_route_to_cpu("any_hit_rays")) {
			return _cpu_dispatch(count, stats, [&](const auto &w, int i, RayStats *s) { hit_results[i] = w.cast(rays[i], query_mask, true, s).hit(); });
		}
		if (!using_gpu()) return MRT_ERR_NO_DEVICE;
		if (!coherent && count >= MIN_BATCH_FOR_SORTING) gpu_caster_.cast_rays_any_hit_sorted(rays, hit_results, count, query_mask);
		else gpu_caster_.cast_rays_any_hit(rays, hit_results, count, query_mask);
		if (stats) stats->rays_cast += (uint64_t)count;
		return gpu_caster_.last_status();
	}
	// ray_dispatcher.h:247-273
	Intersection cast_ray(const Ray &ray, RayStats *stats = nullptr, uint32_t query_mask = 0xFFFFFFFF)
	{
		Intersection result;
		if (_route_to_cpu("cast_ray")) {
			if (has_tlas()) result = CpuTwoLevelWalker(tlas_->arrays()).cast(ray, query_mask, false, stats);
			else if (scene_.built) result = _walker().cast(ray, query_mask, false, stats);
		}
		else if (using_gpu()) { gpu_caster_.cast_rays(&ray, &result, 1, query_mask); if (stats) stats->rays_cast++; }
		return result;
	}
	bool any_hit(const Ray &ray, RayStats *stats = nullptr, uint32_t query_mask = 0xFFFFFFFF)
	{
		bool result = false;
		if (_route_to_cpu("any_hit")) {
			if (has_tlas()) result = CpuTwoLevelWalker(tlas_->arrays()).cast(ray, query_mask, true, stats).hit();
			else if (scene_.built) result = _walker().cast(ray, query_mask, true, stats).hit();
		}
		else if (using_gpu()) { gpu_caster_.cast_rays_any_hit(&ray, &result, 1, query_mask); if (stats) stats->rays_cast++; }
		return result;
	}

	// ray_dispatcher.h:290-356: sorting for >= 256 rays happens on the device inside submit.
	void submit_gpu_async(const Ray *rays, int count)
	{
		if (using_gpu()) gpu_caster_.submit_async(rays, count, 0xFFFFFFFF, count < MIN_BATCH_FOR_SORTING);
	}
	void collect_gpu_nearest(Intersection *results, int count) { gpu_caster_.collect_nearest(results, count); }
	void submit_gpu_async_any_hit(const Ray *rays, int count)
	{
		if (using_gpu()) gpu_caster_.submit_async_any_hit(rays, count, 0xFFFFFFFF, count < MIN_BATCH_FOR_SORTING);
	}
	void collect_gpu_any_hit(bool *hit_results, int count) { gpu_caster_.collect_any_hit(hit_results, count); }
	bool has_gpu_pending() const { return gpu_caster_.has_pending(); }

	int triangle_count() const { return scene_.triangle_count(); }
	int bvh_node_count() const // BVH::NodeCount(): reachable nodes = usedNodes - 1 (node 1 is a hole)
	{
		return scene_.built ? (int)scene_.used_nodes - 1 : 0;
	}
	int bvh_depth() const // ray_dispatcher.h:375-384: ceil(log2(nodes))
	{
		int nodes = bvh_node_count();
		if (nodes <= 0) return 0;
		int depth = 0;
		while ((1 << depth) < nodes) depth++;
		return depth;
	}
	uint32_t thread_count() const { return pool_.thread_count(); } // workers of the CPU backend's pool (ray_dispatcher.h:386-388)

	GPURayCaster &gpu_caster() { return gpu_caster_; }
	const GPURayCaster &gpu_caster() const { return gpu_caster_; }

private:
	RayScene scene_;
	SceneTLAS *tlas_ = nullptr;  // not owned (ray_dispatcher.h:405)
	GPURayCaster gpu_caster_;
	ThreadPool pool_;            // persistent worker threads of the CPU backend (ray_dispatcher.h:403)
	Backend backend_ = Backend::CPU;
	bool cpu_fallback_ = false;  // opt-in: no usable device -> the CPU pool (the reference's routing)
	bool fell_back_ = false;
	static constexpr int MIN_BATCH_FOR_THREADING = 128; // ray_dispatcher.h:422
	static constexpr int MIN_BATCH_FOR_SORTING = 256;   // ray_dispatcher.h:427

	CpuWalker _walker() const
	{
		return CpuWalker(scene_.triangles.data(), scene_.bvh2.data(), scene_.prim_idx.data(), (uint32_t)scene_.triangles.size());
	}
	// ray_dispatcher.h:152-180: without stats one parallel dispatch; with stats every chunk accumulates into its own
	// slot (taken from an atomic counter) and the slots are merged afterwards
	template <typename PerRay>
	int _cpu_dispatch(int count, RayStats *stats, PerRay per_ray)
	{
		if (count == 0) return MRT_OK;
		if (has_tlas()) return _cpu_run(CpuTwoLevelWalker(tlas_->arrays()), count, stats, per_ray); // ray_dispatcher.h:447-449
		if (!scene_.built) { std::fprintf(stderr, "[RayDispatcher] CPU backend: no scene built\n"); return MRT_ERR_NO_SCENE; }
		return _cpu_run(_walker(), count, stats, per_ray);
	}
	template <typename Walker, typename PerRay>
	int _cpu_run(const Walker &w, int count, RayStats *stats, PerRay per_ray)
	{
		// one tally per span of the pool's split (at most helpers + 1), handed out in arrival order, summed at the end
		std::vector<RayStats> tallies(stats ? pool_.thread_count() + 1 : 0);
		std::atomic<uint32_t> tally_cursor{0};
		pool_.dispatch_and_wait(count, MIN_BATCH_FOR_THREADING, [&](int first, int last) {
			RayStats *mine = stats ? &tallies[tally_cursor.fetch_add(1, std::memory_order_relaxed)] : nullptr;
			for (int i = first; i < last; i++) per_ray(w, i, mine);
		});
		for (const RayStats &t : tallies) *stats += t;
		return MRT_OK;
	}

	bool _should_use_gpu() const // ray_dispatcher.h:429-439
	{
		switch (backend_) {
			case Backend::GPU: return true;
			case Backend::AUTO: return gpu_caster_.is_available();
			case Backend::CPU: return false;
		}
		return false;
	}
	// true: this cast runs on the CPU pool.  Backend::CPU always; Backend::GPU / AUTO without a usable device only with
	// set_cpu_fallback(true) (and says so once) -- otherwise the reason is printed and the cast returns MRT_ERR_NO_DEVICE.
	bool _route_to_cpu(const char *what)
	{
		if (backend_ == Backend::CPU) return true;
		if (using_gpu()) return false;
		if (cpu_fallback_) {
			if (!fell_back_) std::fprintf(stderr, "[RayDispatcher] %s: Backend::%s selected but no usable MI355X context; set_cpu_fallback(true): "
					"routing to the CPU pool as the reference does (ray_dispatcher.h:152-180)\n", what, backend_ == Backend::AUTO ? "AUTO" : "GPU");
			fell_back_ = true;
			return true;
		}
		std::fprintf(stderr, "[RayDispatcher] %s: Backend::%s selected but no initialized MI355X context with an uploaded scene; "
				"nothing falls back to the CPU silently (select Backend::CPU, or opt in with set_cpu_fallback(true))\n", what,
				backend_ == Backend::AUTO ? "AUTO" : "GPU");
		return false;
	}
};

} // namespace mrt
