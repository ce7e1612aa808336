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
// What is NOT here, on purpose: the CPU backend.  In the reference the CPU
// path is TinyBVH under a ThreadPool (ray_dispatcher.h:152-180,443-463); a
// maintainer keeps that code as is.  This library ships only the device path
// and never falls back to a CPU: with Backend::CPU, or with the GPU
// unavailable, every cast returns MRT_ERR_UNSUPPORTED and prints why.  The
// casts therefore return an int status (the reference's return void).
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>
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

class RayDispatcher {
public:
	enum class Backend { CPU, GPU, AUTO }; // ray_dispatcher.h:40-44

	RayScene &scene() { return scene_; }
	const RayScene &scene() const { return scene_; }

	void build() // ray_dispatcher.h:67-80
	{
		scene_.build();
		if (_should_use_gpu() && gpu_caster_.is_initialized()) upload_to_gpu();
	}

	void set_backend(Backend b) { backend_ = b; }
	Backend get_backend() const { return backend_; }
	bool gpu_available() const { return gpu_caster_.is_available(); }
	bool gpu_initialized() const { return gpu_caster_.is_initialized(); }
	bool initialize_gpu(int device_ordinal = 0) { return gpu_caster_.initialize(device_ordinal); }
	void upload_to_gpu() // ray_dispatcher.h:99-107
	{
		if (gpu_caster_.is_initialized() && scene_.built)
			gpu_caster_.upload_scene(scene_.triangles, scene_.bvh2.data(), scene_.used_nodes, scene_.prim_idx.data());
	}
	bool using_gpu() const { return _should_use_gpu() && gpu_caster_.is_available(); }

	// ray_dispatcher.h:124-181.  stats: only rays_cast is maintained on the GPU path.
	int cast_rays(const Ray *rays, Intersection *results, int count, RayStats *stats = nullptr,
			uint32_t query_mask = 0xFFFFFFFF, bool coherent = false)
	{
		if (count < 0 || (count > 0 && (!rays || !results))) return MRT_ERR_INVALID;
		if (!using_gpu()) return _no_cpu("cast_rays");
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
		if (!using_gpu()) return _no_cpu("any_hit_rays");
		if (!coherent && count >= MIN_BATCH_FOR_SORTING) gpu_caster_.cast_rays_any_hit_sorted(rays, hit_results, count, query_mask);
		else gpu_caster_.cast_rays_any_hit(rays, hit_results, count, query_mask);
		if (stats) stats->rays_cast += (uint64_t)count;
		return gpu_caster_.last_status();
	}
	// ray_dispatcher.h:247-273
	Intersection cast_ray(const Ray &ray, RayStats *stats = nullptr, uint32_t query_mask = 0xFFFFFFFF)
	{
		Intersection result;
		if (using_gpu()) { gpu_caster_.cast_rays(&ray, &result, 1, query_mask); if (stats) stats->rays_cast++; }
		else _no_cpu("cast_ray");
		return result;
	}
	bool any_hit(const Ray &ray, RayStats *stats = nullptr, uint32_t query_mask = 0xFFFFFFFF)
	{
		bool result = false;
		if (using_gpu()) { gpu_caster_.cast_rays_any_hit(&ray, &result, 1, query_mask); if (stats) stats->rays_cast++; }
		else _no_cpu("any_hit");
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
	uint32_t thread_count() const { return 0; } // no CPU pool in this backend

	GPURayCaster &gpu_caster() { return gpu_caster_; }
	const GPURayCaster &gpu_caster() const { return gpu_caster_; }

private:
	RayScene scene_;
	GPURayCaster gpu_caster_;
	Backend backend_ = Backend::CPU;
	static constexpr int MIN_BATCH_FOR_SORTING = 256; // ray_dispatcher.h:427

	bool _should_use_gpu() const // ray_dispatcher.h:429-439
	{
		switch (backend_) {
			case Backend::GPU: return true;
			case Backend::AUTO: return gpu_caster_.is_available();
			case Backend::CPU: return false;
		}
		return false;
	}
	int _no_cpu(const char *what) const
	{
		std::fprintf(stderr, "[RayDispatcher] %s: the CPU backend is the reference's own TinyBVH path and is not part of "
				"this library; select Backend::GPU with an initialized, uploaded scene\n", what);
		return MRT_ERR_UNSUPPORTED;
	}
};

} // namespace mrt
