// cpu_backend.hpp — the CPU half of RayDispatcher (src/dispatch/ray_dispatcher.h:152-180,214-240,443-463) and the
// ThreadPool it splits batches over (src/dispatch/thread_pool.h:77-133,173-222), for the host-side mirror.
//
// In the reference the CPU path is TinyBVH (BVH4 / BVH8 SIMD walks) behind RayScene::cast_rays.  TinyBVH is the
// reference's vendored dependency and stays with the reference; what a host needs from this repository is a router
// that still answers when the caller selects Backend::CPU (BASELINE.json config 1 is "cast_debug_rays on the CPU
// ThreadPool backend").  This walk goes over the SAME arrays the device is fed — the 8-bin SAH BVH2 that
// mrt_bvh2_build makes, the reference's Triangle PODs — with the acceptance rules and the operation order of the
// device kernels ("Arithmetic": explicit fma, slab test of bvh_traverse.comp.glsl:84-99, Moller-Trumbore
// of :105-131, an exact tie to the lower triangle id), so the two backends of one router return the same records bit
// for bit.  It is an explicit backend, never a fallback: Backend::GPU and Backend::AUTO do not degrade to it
// (ray_dispatcher.hpp), and libmrt_hip.so itself has no CPU path at all.
//
// Compile with -ffp-contract=off (as everything here): the fused operations are the explicit std::fmaf calls.
#pragma once
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "host_types.hpp"
#include "../../../include/mrt_hip.h"

namespace mrt {

// The pool behind RayDispatcher's CPU path.  Contract taken from src/dispatch/thread_pool.h:41-55,77-133 (its text is not
// followed): helpers = hardware_concurrency - 1 persistent threads; dispatch_and_wait(count, min_batch, body) runs
// body(0, count) inline for small batches, otherwise cuts [0, count) into helpers + 1 equal spans, runs span 0 on the
// calling thread while the helpers claim the others, and returns when every span has run.
//
// How this one works: a dispatch is a "ticket" = (epoch << 32 | next unclaimed span) in ONE atomic word.  A helper that
// wakes for epoch e claims spans by compare-and-swap on that word and stops the moment the word carries another epoch,
// so a helper that is late for e can never touch the spans of e + 1 with e's parameters; completion is a countdown of
// spans (not of helpers), so a dispatch does not wait for helpers that found nothing left to do.
class ThreadPool {
public:
	explicit ThreadPool(uint32_t helpers = 0)
	{
		const uint32_t cores = std::thread::hardware_concurrency();
		n_helpers_ = helpers ? helpers : (cores > 1 ? cores - 1 : 0);
		helpers_.reserve(n_helpers_);
		for (uint32_t i = 0; i < n_helpers_; i++) helpers_.emplace_back(&ThreadPool::helper_main, this);
	}
	~ThreadPool()
	{
		{ std::lock_guard<std::mutex> g(gate_); quitting_ = true; }
		wake_.notify_all();
		for (std::thread &t : helpers_) t.join();
	}
	ThreadPool(const ThreadPool &) = delete;
	ThreadPool &operator=(const ThreadPool &) = delete;

	void dispatch_and_wait(int count, int min_batch_size, const std::function<void(int, int)> &body)
	{
		if (count <= 0) return;
		if (n_helpers_ == 0 || count <= min_batch_size) { body(0, count); return; }
		const int spans = (int)n_helpers_ + 1, len = (count + spans - 1) / spans;
		const int used = (count + len - 1) / len;           // spans that hold at least one item
		uint32_t epoch;
		{
			std::lock_guard<std::mutex> g(gate_);
			epoch = ++epoch_;
			body_ = &body; total_ = count; span_len_ = len; span_count_ = used;
			unfinished_.store(used, std::memory_order_relaxed);
			ticket_.store((uint64_t)epoch << 32 | 1u, std::memory_order_release); // span 0 belongs to the caller
		}
		wake_.notify_all();
		run_span(body, 0, len, count);
		std::unique_lock<std::mutex> g(gate_);
		all_done_.wait(g, [this] { return unfinished_.load(std::memory_order_acquire) == 0; });
	}
	uint32_t thread_count() const { return n_helpers_; }

private:
	std::vector<std::thread> helpers_;
	uint32_t n_helpers_ = 0;
	std::mutex gate_;
	std::condition_variable wake_, all_done_;
	bool quitting_ = false;
	uint32_t epoch_ = 0;                          // guarded by gate_
	const std::function<void(int, int)> *body_ = nullptr;
	int total_ = 0, span_len_ = 0, span_count_ = 0;
	std::atomic<uint64_t> ticket_{0};
	std::atomic<int> unfinished_{0};

	void run_span(const std::function<void(int, int)> &body, int span, int len, int total)
	{
		const long long lo = (long long)span * len;
		body((int)lo, (int)std::min<long long>(lo + len, total));
		if (unfinished_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
			std::lock_guard<std::mutex> g(gate_); // the waiter is either before its predicate check or inside wait()
			all_done_.notify_all();
		}
	}
	void helper_main()
	{
		uint32_t seen = 0;
		for (;;) {
			const std::function<void(int, int)> *body; int total, len, n_spans; uint32_t epoch;
			{
				std::unique_lock<std::mutex> g(gate_);
				wake_.wait(g, [&] { return quitting_ || epoch_ != seen; });
				if (quitting_) return;
				seen = epoch = epoch_;
				body = body_; total = total_; len = span_len_; n_spans = span_count_;
			}
			uint64_t t = ticket_.load(std::memory_order_acquire);
			while ((uint32_t)(t >> 32) == epoch && (int)(uint32_t)t < n_spans) {
				if (ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel, std::memory_order_acquire)) {
					run_span(*body, (int)(uint32_t)t, len, total);
					t = ticket_.load(std::memory_order_acquire);
				}
			}
		}
	}
};

// One ray against the scene's BVH2 (TinyBVH node layout: leftFirst = left child / first prim, triCount > 0 = leaf;
// children are adjacent; node 1 is unused).  RayScene::cast_ray / any_hit (src/accel/ray_scene.h:90-118,136-163) with
// the device kernels' arithmetic.
class CpuWalker {
public:
	CpuWalker(const Triangle *tris, const mrt_bvh_node32 *nodes, const uint32_t *prim_idx, uint32_t n_tris)
		: tris_(tris), nodes_(nodes), prim_idx_(prim_idx), n_tris_(n_tris) {}

	// closest hit (any_hit = false) or first hit found (any_hit = true); stats may be null
	Intersection cast(const Ray &ray, uint32_t query_mask, bool any_hit, RayStats *stats) const
	{
		Intersection out; // a miss record as the default constructor leaves it (Intersection::set_miss, intersection.h:45-52)
		if (stats) stats->rays_cast++;
		if (n_tris_ == 0 || ray.t_min >= ray.t_max) return out; // degenerate rays are misses (glsl:214-222)
		const float ox = ray.origin.x, oy = ray.origin.y, oz = ray.origin.z;
		const float dx = ray.direction.x, dy = ray.direction.y, dz = ray.direction.z;
		const float ix = safe_inv(dx), iy = safe_inv(dy), iz = safe_inv(dz);
		const float nrx = -(ox * ix), nry = -(oy * iy), nrz = -(oz * iz);
		float best_t = ray.t_max, best_u = 0.0f, best_v = 0.0f;
		uint32_t best_tri = UINT32_MAX, best_id = UINT32_MAX;
		uint32_t stack[128];
		int sp = 0;
		uint32_t cur = 0;
		// The root's own box is not tested (the device's wide root holds its children's boxes) -- except when the root is
		// a leaf: upload_scene wraps it as the left child of a wide root (gpu_ray_caster.cpp:255-271), box test included.
		if (nodes_[0].tri_count > 0) {
			float tn;
			if (stats) stats->bvh_nodes_visited++;
			if (!slab(nodes_[0], ix, iy, iz, nrx, nry, nrz, ray.t_min, best_t, tn)) return out;
		}
		for (;;) {
			const mrt_bvh_node32 &n = nodes_[cur];
			if (n.tri_count > 0) {
				for (uint32_t k = 0; k < n.tri_count; k++) {
					const uint32_t ti = prim_idx_[n.left_first + k];
					const Triangle &t = tris_[ti];
					if ((t.layers & query_mask) == 0u) continue;
					if (stats) stats->tri_tests++;
					// ray_triangle, glsl:105-131 == Triangle::intersect, src/core/triangle.h:56-105
					const float pvx = std::fmaf(dy, t.edge2.z, -(dz * t.edge2.y));
					const float pvy = std::fmaf(dz, t.edge2.x, -(dx * t.edge2.z));
					const float pvz = std::fmaf(dx, t.edge2.y, -(dy * t.edge2.x));
					const float det = dot3(t.edge1.x, t.edge1.y, t.edge1.z, pvx, pvy, pvz);
					if (std::fabs(det) < 1e-8f) continue;
					const float inv_det = 1.0f / det;
					const float tvx = ox - t.v0.x, tvy = oy - t.v0.y, tvz = oz - t.v0.z;
					const float u = dot3(tvx, tvy, tvz, pvx, pvy, pvz) * inv_det;
					if (u < 0.0f || u > 1.0f) continue;
					const float qvx = std::fmaf(tvy, t.edge1.z, -(tvz * t.edge1.y));
					const float qvy = std::fmaf(tvz, t.edge1.x, -(tvx * t.edge1.z));
					const float qvz = std::fmaf(tvx, t.edge1.y, -(tvy * t.edge1.x));
					const float v = dot3(dx, dy, dz, qvx, qvy, qvz) * inv_det;
					if (v < 0.0f || u + v > 1.0f) continue;
					const float tt = dot3(t.edge2.x, t.edge2.y, t.edge2.z, qvx, qvy, qvz) * inv_det;
					// glsl:124 accepts t_min <= t < best_t; an exact tie goes to the lower triangle id
					if (!(tt < ray.t_min) && (tt < best_t || (tt == best_t && best_tri != UINT32_MAX && t.id < best_id))) {
						best_t = tt; best_u = u; best_v = v; best_tri = ti; best_id = t.id;
						if (any_hit) goto done;
					}
				}
				if (sp == 0) break;
				cur = stack[--sp];
				continue;
			}
			if (stats) stats->bvh_nodes_visited++;
			const uint32_t l = n.left_first, r = l + 1u;
			float tl, tr;
			const bool hl = slab(nodes_[l], ix, iy, iz, nrx, nry, nrz, ray.t_min, best_t, tl);
			const bool hr = slab(nodes_[r], ix, iy, iz, nrx, nry, nrz, ray.t_min, best_t, tr);
			if (hl && hr) { // near child first, far child pushed (glsl:290-305)
				const bool left_near = tl < tr;
				if (sp < 128) stack[sp++] = left_near ? r : l;
				cur = left_near ? l : r;
			} else if (hl) cur = l;
			else if (hr) cur = r;
			else { if (sp == 0) break; cur = stack[--sp]; }
		}
	done:
		if (best_tri != UINT32_MAX) { // packed -> Intersection, gpu_ray_caster.cpp:442-456
			const Triangle &t = tris_[best_tri];
			out.t = best_t;
			out.position = Vector3(ox + dx * best_t, oy + dy * best_t, oz + dz * best_t);
			out.normal = t.normal; out.u = best_u; out.v = best_v; out.prim_id = t.id; out.hit_layers = t.layers;
			if (stats) stats->hits++;
		}
		return out;
	}

private:
	const Triangle *tris_; const mrt_bvh_node32 *nodes_; const uint32_t *prim_idx_; uint32_t n_tris_;

	static float safe_inv(float d) // safe_inv_direction, glsl:137-145 == Ray::_precompute, src/core/ray.h:78-89
	{
		const float eps = 1e-9f, big = 1.0f / eps;
		return std::fabs(d) > eps ? 1.0f / d : (d >= 0.0f ? big : -big);
	}
	static float dot3(float ax, float ay, float az, float bx, float by, float bz) { return std::fmaf(ax, bx, std::fmaf(ay, by, az * bz)); }
	// ray_aabb, glsl:84-99, clamped to [t_min, best_t]
	static bool slab(const mrt_bvh_node32 &b, float ix, float iy, float iz, float nrx, float nry, float nrz, float t_min, float best_t, float &tnear)
	{
		const float x0 = std::fmaf(b.aabb_min[0], ix, nrx), x1 = std::fmaf(b.aabb_max[0], ix, nrx);
		const float y0 = std::fmaf(b.aabb_min[1], iy, nry), y1 = std::fmaf(b.aabb_max[1], iy, nry);
		const float z0 = std::fmaf(b.aabb_min[2], iz, nrz), z1 = std::fmaf(b.aabb_max[2], iz, nrz);
		tnear = std::fmax(std::fmax(std::fmin(x0, x1), std::fmin(y0, y1)), std::fmax(std::fmin(z0, z1), t_min));
		const float tfar = std::fmin(std::fmin(std::fmax(x0, x1), std::fmax(y0, y1)), std::fmin(std::fmax(z0, z1), best_t));
		return tnear <= tfar;
	}
};

// One ray against a two-level scene on the host: SceneTLAS::cast_ray / any_hit (src/accel/scene_tlas.h:198-251 =
// tinybvh::BVH::IntersectTLAS, tiny_bvh.h:3306-3380) as the DEVICE walks it (csrc/two_level_kernel.h,
// trace_two_level_kernel, restated one ray at a time): one node array for the TLAS and every BLAS, one stack; at a TLAS
// leaf the rest of the leaf is pushed, the ray goes to the instance's mesh space (o' = M o + t, d' = M d, no
// renormalisation: t stays world-parameterised), a return marker is pushed and the walk continues at the BLAS root; popping
// the marker restores the world ray.  Same fused operations, same acceptance rules, an exact tie to the lower FLAT id,
// whole instances skipped by the query mask: the records equal the device's bit for bit.  The arrays are the ones
// mrt_upload_two_level_scene uploads (mrt_two_level_prepare_host, include/mrt_hip.h).
class CpuTwoLevelWalker {
public:
	explicit CpuTwoLevelWalker(const mrt_two_level_arrays &a) : a_(a) {}

	Intersection cast(const Ray &ray, uint32_t query_mask, bool any_hit, RayStats *stats) const
	{
		Intersection out;
		if (stats) stats->rays_cast++;
		if (a_.n_instances == 0 || ray.t_min >= ray.t_max) return out;
		constexpr uint32_t kEnd = 0x7FFFFFFFu, kBack = 0x7FFFFFFEu, kLeaf = 0x80000000u; // sentinel, "back to the world ray", leaf flag
		const float wox = ray.origin.x, woy = ray.origin.y, woz = ray.origin.z, wdx = ray.direction.x, wdy = ray.direction.y, wdz = ray.direction.z;
		float ox = wox, oy = woy, oz = woz, dx = wdx, dy = wdy, dz = wdz;          // the ray being walked
		float ix = safe_inv(dx), iy = safe_inv(dy), iz = safe_inv(dz);
		float nrx = -(ox * ix), nry = -(oy * iy), nrz = -(oz * iz);
		float best_t = ray.t_max, best_u = 0.0f, best_v = 0.0f;
		uint32_t best_slot = UINT32_MAX, best_id = UINT32_MAX, best_inst = 0u;
		std::vector<uint32_t> stack(a_.depth + 8u);
		uint32_t sp = 0;
		stack[sp++] = kEnd;
		uint32_t cur = 0, id_base = 0u, cur_inst = 0u;
		bool in_blas = false;
		while (cur != kEnd) {
			if (cur < kBack) { // an inner node of the TLAS or of a BLAS
				if (stats) stats->bvh_nodes_visited++;
				const mrt_bvh_node_wide64 &n = a_.nodes[cur];
				float tl, tr;
				const bool hl = slab(n.left_min, n.left_max, ix, iy, iz, nrx, nry, nrz, ray.t_min, best_t, tl);
				const bool hr = slab(n.right_min, n.right_max, ix, iy, iz, nrx, nry, nrz, ray.t_min, best_t, tr);
				if (hl && hr) {
					const bool left_near = tl < tr;
					cur = left_near ? n.left_idx : n.right_idx;
					if (sp < stack.size()) stack[sp++] = left_near ? n.right_idx : n.left_idx;
				} else if (hl) cur = n.left_idx;
				else if (hr) cur = n.right_idx;
				else cur = stack[--sp];
				continue;
			}
			if (cur == kBack) { // the BLAS is done: back to the world ray
				ox = wox; oy = woy; oz = woz; dx = wdx; dy = wdy; dz = wdz;
				ix = safe_inv(dx); iy = safe_inv(dy); iz = safe_inv(dz);
				nrx = -(ox * ix); nry = -(oy * iy); nrz = -(oz * iz);
				in_blas = false;
				cur = stack[--sp];
				continue;
			}
			const uint32_t slot0 = cur & 0x7FFFFFFFu;
			if (!in_blas) { // TLAS leaf: a run of instances, one at a time
				const float *row = a_.instances + (size_t)slot0 * 32u;
				uint32_t meta[4], flags;
				std::memcpy(meta, row + 20, sizeof(meta)); // {basis[8], root, id_base, layers}
				std::memcpy(&flags, row + 24, sizeof(flags));
				if ((flags & 1u) == 0u && sp < stack.size()) stack[sp++] = kLeaf | (slot0 + 1u); // the rest of the leaf
				if ((meta[3] & query_mask) != 0u) {
					ox = std::fmaf(row[0], wox, std::fmaf(row[1], woy, std::fmaf(row[2], woz, row[3])));
					oy = std::fmaf(row[4], wox, std::fmaf(row[5], woy, std::fmaf(row[6], woz, row[7])));
					oz = std::fmaf(row[8], wox, std::fmaf(row[9], woy, std::fmaf(row[10], woz, row[11])));
					dx = std::fmaf(row[0], wdx, std::fmaf(row[1], wdy, row[2] * wdz));
					dy = std::fmaf(row[4], wdx, std::fmaf(row[5], wdy, row[6] * wdz));
					dz = std::fmaf(row[8], wdx, std::fmaf(row[9], wdy, row[10] * wdz));
					ix = safe_inv(dx); iy = safe_inv(dy); iz = safe_inv(dz);
					nrx = -(ox * ix); nry = -(oy * iy); nrz = -(oz * iz);
					if (sp < stack.size()) stack[sp++] = kBack;
					in_blas = true; cur_inst = slot0; id_base = meta[2];
					cur = meta[1];
				} else cur = stack[--sp];
				continue;
			}
			// BLAS leaf: its triangles, on the mesh-space ray
			uint32_t slot = slot0;
			bool last;
			do {
				const float *q = a_.tri_hot + (size_t)slot * 12u; // {v0, id | e1, layers | e2, flags}
				uint32_t local_id, tflags;
				std::memcpy(&local_id, q + 3, 4); std::memcpy(&tflags, q + 11, 4);
				last = (tflags & 1u) != 0u;
				if (stats) stats->tri_tests++;
				const float pvx = std::fmaf(dy, q[10], -(dz * q[9]));
				const float pvy = std::fmaf(dz, q[8], -(dx * q[10]));
				const float pvz = std::fmaf(dx, q[9], -(dy * q[8]));
				const float det = dot3(q[4], q[5], q[6], pvx, pvy, pvz);
				if (!(std::fabs(det) < 1e-8f)) {
					const float inv_det = 1.0f / det;
					const float tvx = ox - q[0], tvy = oy - q[1], tvz = oz - q[2];
					const float u = dot3(tvx, tvy, tvz, pvx, pvy, pvz) * inv_det;
					if (!(u < 0.0f || u > 1.0f)) {
						const float qvx = std::fmaf(tvy, q[6], -(tvz * q[5]));
						const float qvy = std::fmaf(tvz, q[4], -(tvx * q[6]));
						const float qvz = std::fmaf(tvx, q[5], -(tvy * q[4]));
						const float v = dot3(dx, dy, dz, qvx, qvy, qvz) * inv_det;
						if (!(v < 0.0f || u + v > 1.0f)) {
							const float t = dot3(q[8], q[9], q[10], qvx, qvy, qvz) * inv_det;
							const uint32_t id = id_base + local_id;
							if (!(t < ray.t_min) && (t < best_t || (t == best_t && best_slot != UINT32_MAX && id < best_id))) {
								best_t = t; best_u = u; best_v = v; best_slot = slot; best_id = id; best_inst = cur_inst;
								if (any_hit) last = true;
							}
						}
					}
				}
				slot++;
			} while (!last);
			if (any_hit && best_slot != UINT32_MAX) break;
			cur = stack[--sp];
		}
		if (best_slot != UINT32_MAX) { // SceneTLAS::cast_ray, scene_tlas.h:217-244, with the flat id
			const float *row = a_.instances + (size_t)best_inst * 32u, *b = row + 12, *no = a_.tri_cold + (size_t)best_slot * 4u;
			uint32_t layers;
			std::memcpy(&layers, row + 23, 4);
			float nx = std::fmaf(b[0], no[0], std::fmaf(b[1], no[1], b[2] * no[2]));
			float ny = std::fmaf(b[3], no[0], std::fmaf(b[4], no[1], b[5] * no[2]));
			float nz = std::fmaf(b[6], no[0], std::fmaf(b[7], no[1], b[8] * no[2]));
			const float l2 = std::fmaf(nx, nx, std::fmaf(ny, ny, nz * nz));
			if (l2 == 0.0f) { nx = ny = nz = 0.0f; }
			else { const float l = std::sqrt(l2); nx /= l; ny /= l; nz /= l; }
			out.t = best_t;
			out.position = Vector3(wox + wdx * best_t, woy + wdy * best_t, woz + wdz * best_t);
			out.normal = Vector3(nx, ny, nz); out.u = best_u; out.v = best_v; out.prim_id = best_id; out.hit_layers = layers;
			if (stats) stats->hits++;
		}
		return out;
	}

private:
	mrt_two_level_arrays a_;
	static float safe_inv(float d)
	{
		const float eps = 1e-9f, big = 1.0f / eps;
		return std::fabs(d) > eps ? 1.0f / d : (d >= 0.0f ? big : -big);
	}
	static float dot3(float ax, float ay, float az, float bx, float by, float bz) { return std::fmaf(ax, bx, std::fmaf(ay, by, az * bz)); }
	static bool slab(const float mn[3], const float mx[3], float ix, float iy, float iz, float nrx, float nry, float nrz, float t_min, float best_t, float &tnear)
	{
		const float x0 = std::fmaf(mn[0], ix, nrx), x1 = std::fmaf(mx[0], ix, nrx);
		const float y0 = std::fmaf(mn[1], iy, nry), y1 = std::fmaf(mx[1], iy, nry);
		const float z0 = std::fmaf(mn[2], iz, nrz), z1 = std::fmaf(mx[2], iz, nrz);
		tnear = std::fmax(std::fmax(std::fmin(x0, x1), std::fmin(y0, y1)), std::fmax(std::fmin(z0, z1), t_min));
		const float tfar = std::fmin(std::fmin(std::fmax(x0, x1), std::fmax(y0, y1)), std::fmin(std::fmax(z0, z1), best_t));
		return tnear <= tfar;
	}
};

} // namespace mrt
