// gpu_ray_caster.hpp — GPURayCaster over the MI355X C-ABI (include/mrt_hip.h).
//
// Same public surface as the reference's Vulkan caster
// (src/gpu/gpu_ray_caster.h:50-136): initialize / is_available /
// is_initialized / upload_scene / upload_cwbvh / cast_rays / cast_rays_any_hit
// / submit_async[_any_hit] / collect_nearest / collect_any_hit / has_pending /
// cleanup.  Differences a maintainer sees:
//   * upload_scene takes the three arrays of a tinybvh::BVH (bvhNode,
//     usedNodes, primIdx) instead of the object, so this header needs no
//     TinyBVH include;
//   * Ray -> packed ray, packed hit -> Intersection (position = origin +
//     direction * t) and the bool extraction run on the device
//     (gpu_ray_caster.cpp:639-650, 442-456, 482-487 are host loops);
//   * there is no 512 K batching (cpp:56,427): no TDR on a compute device;
//   * cast_rays_sorted / submit_async_sorted replace RayDispatcher's host
//     std::sort + gather + unshuffle (ray_dispatcher.h:135-146) with the device
//     radix sort;
//   * the RenderingDevice / RID getters (h:121-136) do not exist.
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../../include/mrt_hip.h"
#include "host_types.hpp"

namespace mrt {

class GPURayCaster {
public:
	GPURayCaster() = default;
	~GPURayCaster() { cleanup(); }
	GPURayCaster(const GPURayCaster &) = delete;
	GPURayCaster &operator=(const GPURayCaster &) = delete;

	// Create the device context.  Idempotent (gpu_ray_caster.cpp:72-76).
	bool initialize(int device_ordinal = 0)
	{
		if (ctx_) return true;
		int rc = mrt_create(device_ordinal, nullptr, &ctx_);
		if (rc != MRT_OK) {
			std::fprintf(stderr, "[GPU RayCaster] initialize failed: %s\n", mrt_status_string(rc));
			ctx_ = nullptr;
			return false;
		}
		return true;
	}
	bool is_available() const { return ctx_ && mrt_is_available(ctx_); }
	bool is_initialized() const { return ctx_ != nullptr; }

	// upload_scene(triangles, bvh2): pass bvh2.bvhNode, bvh2.usedNodes, bvh2.primIdx.
	void upload_scene(const std::vector<Triangle> &triangles, const mrt_bvh_node32 *bvh_nodes,
			uint32_t used_nodes, const uint32_t *prim_idx)
	{
		if (!ctx_ || triangles.empty()) return; // cpp:194
		std::vector<mrt_tri64> packed(triangles.size());
		mrt_pack_host_triangles(reinterpret_cast<const mrt_host_tri80 *>(triangles.data()),
				(uint32_t)triangles.size(), packed.data());
		report(mrt_upload_scene(ctx_, packed.data(), (uint32_t)packed.size(), bvh_nodes, used_nodes, prim_idx), "upload_scene");
	}
	// No reference counterpart: build the acceleration structure on the device from the triangles
	// alone (LBVH, milliseconds) instead of scene.build() + upload_scene(); same hit records.
	void build_scene_on_device(const std::vector<Triangle> &triangles)
	{
		if (!ctx_ || triangles.empty()) return;
		std::vector<mrt_tri64> packed(triangles.size());
		mrt_pack_host_triangles(reinterpret_cast<const mrt_host_tri80 *>(triangles.data()),
				(uint32_t)triangles.size(), packed.data());
		report(mrt_build_scene_device(ctx_, packed.data(), (uint32_t)packed.size(), 0), "build_scene_on_device");
	}
	// SceneTLAS on the device (scene_tlas.h:140-196): one BLAS per distinct mesh, a TLAS over the placed
	// instances, nothing flattened; move_instances = set_instance_transform + refit_tlas.  Hit records carry
	// the flat triangle ids of RayTracerServer::_rebuild_scene.
	void upload_two_level_scene(const float *mesh_vertices9, uint32_t n_mesh_tris, const std::vector<mrt_instance> &instances)
	{
		if (!ctx_ || instances.empty()) return;
		report(mrt_upload_two_level_scene(ctx_, mesh_vertices9, n_mesh_tris, instances.data(), (uint32_t)instances.size(), 0), "upload_two_level_scene");
	}
	void move_instances(const std::vector<mrt_instance> &instances)
	{
		if (!ctx_ || instances.empty()) return;
		report(mrt_update_instances(ctx_, instances.data(), (uint32_t)instances.size()), "move_instances");
	}
	// CWBVH is a Vulkan-path layout (cpp:351-411); accepted and ignored.
	void upload_cwbvh(const void * /*cwbvh*/) {}

	void cast_rays(const Ray *rays, Intersection *results, int count, uint32_t query_mask = 0xFFFFFFFF)
	{
		if (!is_available() || count <= 0) return; // cpp:419
		report(mrt_cast(ctx_, rays, results, (uint64_t)count, query_mask, MRT_MODE_NEAREST,
				MRT_FLAG_HOST_LAYOUT | MRT_FLAG_COHERENT), "cast_rays");
	}
	void cast_rays_any_hit(const Ray *rays, bool *hit_results, int count, uint32_t query_mask = 0xFFFFFFFF)
	{
		if (!is_available() || count <= 0) return; // cpp:466
		static_assert(sizeof(bool) == 1, "bool results are written as bytes");
		report(mrt_cast(ctx_, rays, hit_results, (uint64_t)count, query_mask, MRT_MODE_ANY_HIT,
				MRT_FLAG_HOST_LAYOUT | MRT_FLAG_COHERENT | MRT_FLAG_BOOL_OUT), "cast_rays_any_hit");
	}
	// Incoherent batches: Morton sort + permuted trace + unshuffle, all on the device.
	void cast_rays_sorted(const Ray *rays, Intersection *results, int count, uint32_t query_mask = 0xFFFFFFFF)
	{
		if (!is_available() || count <= 0) return;
		report(mrt_cast(ctx_, rays, results, (uint64_t)count, query_mask, MRT_MODE_NEAREST, MRT_FLAG_HOST_LAYOUT), "cast_rays_sorted");
	}
	void cast_rays_any_hit_sorted(const Ray *rays, bool *hit_results, int count, uint32_t query_mask = 0xFFFFFFFF)
	{
		if (!is_available() || count <= 0) return;
		report(mrt_cast(ctx_, rays, hit_results, (uint64_t)count, query_mask, MRT_MODE_ANY_HIT,
				MRT_FLAG_HOST_LAYOUT | MRT_FLAG_BOOL_OUT), "cast_rays_any_hit_sorted");
	}

	// ---- async (cpp:536-623): one pending dispatch; rays must outlive collect ----
	void submit_async(const Ray *rays, int count, uint32_t query_mask = 0xFFFFFFFF, bool coherent = true)
	{
		if (!is_available() || count <= 0) return;
		report(mrt_submit(ctx_, rays, (uint64_t)count, query_mask, MRT_MODE_NEAREST,
				MRT_FLAG_HOST_LAYOUT | (coherent ? MRT_FLAG_COHERENT : 0u)), "submit_async");
	}
	void submit_async_any_hit(const Ray *rays, int count, uint32_t query_mask = 0xFFFFFFFF, bool coherent = true)
	{
		if (!is_available() || count <= 0) return;
		report(mrt_submit(ctx_, rays, (uint64_t)count, query_mask, MRT_MODE_ANY_HIT,
				MRT_FLAG_HOST_LAYOUT | MRT_FLAG_BOOL_OUT | (coherent ? MRT_FLAG_COHERENT : 0u)), "submit_async_any_hit");
	}
	void collect_nearest(Intersection *results, int count)
	{
		if (!ctx_ || !mrt_has_pending(ctx_)) return; // cpp:557
		report(mrt_collect(ctx_, results, (uint64_t)count), "collect_nearest");
	}
	void collect_any_hit(bool *hit_results, int count)
	{
		if (!ctx_ || !mrt_has_pending(ctx_)) return; // cpp:599
		report(mrt_collect(ctx_, hit_results, (uint64_t)count), "collect_any_hit");
	}
	bool has_pending() const { return ctx_ && mrt_has_pending(ctx_); }

	void cleanup()
	{
		if (ctx_) { mrt_destroy(ctx_); ctx_ = nullptr; }
	}

	mrt_ctx *context() const { return ctx_; }
	int last_status() const { return last_status_; }

private:
	mrt_ctx *ctx_ = nullptr;
	int last_status_ = MRT_OK;
	void report(int rc, const char *what)
	{
		last_status_ = rc;
		if (rc != MRT_OK) std::fprintf(stderr, "[GPU RayCaster] %s: %s (%s)\n", what, mrt_status_string(rc), mrt_last_error(ctx_));
	}
};

} // namespace mrt
