// ref_harness.cpp — C entry points around the REFERENCE's own CPU arithmetic.
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; the code it drives is the
// reference's vendored TinyBVH 1.6.7, compiled where it lies:
//   /root/reference/thirdparty/tinybvh/tiny_bvh.h      (declarations, here)
//   /root/reference/src/accel/tinybvh_impl.cpp          (the implementation TU)
// by oracle/Makefile into oracle/_ref/libmrt_ref.so (git-ignored, not
// gpurun-ignored).  No reference source is copied into this repository.
//
// The glue above TinyBVH (RayScene / tinybvh_adapter / ThreadPool) needs
// godot-cpp headers, which are an empty submodule in the reference
// (.gitmodules:1-4) and absent from the image, so those files are unbuildable
// here; the ~40 lines of call sequence they contain are restated below with
// file:line citations.
#ifndef TINYBVH_INST_IDX_BITS
#define TINYBVH_INST_IDX_BITS 32 // src/accel/tinybvh_impl.cpp:14, src/accel/ray_scene.h:29-31
#endif
#include "thirdparty/tinybvh/tiny_bvh.h"

#include <atomic>
#include <cfloat>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Ray32 { float o[3]; float t_max; float d[3]; float t_min; };
struct Hit32 { float t; int32_t prim; float u, v; float n[3]; uint32_t layers; };
struct Tri64 { float v0[3]; uint32_t id; float e1[3]; uint32_t layers; float e2[3]; float p2; float n[3]; float p3; };
struct Node32 { float mn[3]; uint32_t left_first; float mx[3]; uint32_t tri_count; };

struct RefScene {
	std::vector<tinybvh::bvhvec4> verts; // RayScene::vertices, src/accel/ray_scene.h:48
	std::vector<Tri64> tris;             // RayScene::triangles (normal / layers lookup)
	tinybvh::BVH bvh2;
	tinybvh::BVH4_CPU bvh4;
	tinybvh::BVH8_CPU bvh8;
	bool use_avx2 = false, has4 = false, has8 = false;
};

// The reference's pool contract (src/dispatch/thread_pool.h:41-55,77-133): PERSISTENT helper threads (created once,
// parked on a condition variable between dispatches), chunks = helpers + 1, chunk = ceil(count / chunks), the caller
// runs chunk 0, the helpers claim the remaining chunks through one atomic counter, the dispatch returns when every
// chunk has run.  One pool per thread count, kept for the life of the library (the timed baseline pays no thread
// start; round 2's harness started fresh std::threads per call).
class Pool {
public:
	explicit Pool(int helpers) : n_(helpers) { for (int i = 0; i < n_; i++) th_.emplace_back([this] { loop(); }); }
	~Pool() { { std::lock_guard<std::mutex> g(m_); quit_ = true; } wake_.notify_all(); for (auto &t : th_) t.join(); }
	int helpers() const { return n_; }
	template <class F> void run(int64_t count, int64_t chunk, F &fn)
	{
		const int64_t chunks = (count + chunk - 1) / chunk;
		{
			std::lock_guard<std::mutex> g(m_);
			call_ = [&fn](int64_t a, int64_t b) { fn(a, b); };
			count_ = count; chunk_ = chunk; left_.store(chunks); next_.store((uint64_t)(++epoch_) << 32 | 1u);
		}
		wake_.notify_all();
		one(0);
		std::unique_lock<std::mutex> g(m_);
		done_.wait(g, [this] { return left_.load() == 0; });
	}
private:
	int n_; std::vector<std::thread> th_;
	std::mutex m_; std::condition_variable wake_, done_;
	bool quit_ = false; uint32_t epoch_ = 0;
	std::function<void(int64_t, int64_t)> call_;
	int64_t count_ = 0, chunk_ = 1;
	std::atomic<uint64_t> next_{0}; std::atomic<int64_t> left_{0};
	void one(int64_t c)
	{
		const int64_t s = c * chunk_;
		call_(s, std::min(s + chunk_, count_));
		if (left_.fetch_sub(1) == 1) { std::lock_guard<std::mutex> g(m_); done_.notify_all(); }
	}
	void loop()
	{
		uint32_t seen = 0;
		for (;;) {
			uint32_t e; int64_t chunks;
			{
				std::unique_lock<std::mutex> g(m_);
				wake_.wait(g, [&] { return quit_ || epoch_ != seen; });
				if (quit_) return;
				seen = e = epoch_; chunks = (count_ + chunk_ - 1) / chunk_;
			}
			uint64_t t = next_.load();
			while ((uint32_t)(t >> 32) == e && (int64_t)(uint32_t)t < chunks)
				if (next_.compare_exchange_weak(t, t + 1)) { one((int64_t)(uint32_t)t); t = next_.load(); }
		}
	}
};

template <class F> void range_split(int64_t count, int n_threads, int min_batch, F fn)
{
	if (count <= 0) return;
	const int helpers = n_threads - 1;
	if (count <= min_batch || helpers <= 0) { fn(0, count); return; }
	static std::mutex pools_m;
	static std::map<int, std::unique_ptr<Pool>> pools;
	Pool *pool;
	{
		std::lock_guard<std::mutex> g(pools_m);
		auto &slot = pools[helpers];
		if (!slot) slot.reset(new Pool(helpers));
		pool = slot.get();
	}
	const int64_t chunks = helpers + 1, chunk = (count + chunks - 1) / chunks;
	pool->run(count, chunk, fn);
}

} // namespace

extern "C" {

int ref_has_avx2() { return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"); } // src/dispatch/cpu_feature_detect.h:27-55

// RayScene::build, src/accel/ray_scene.h:62-86 (CWBVH omitted: GPU-only layout).
// variants: bit0 = BVH2 only, bit1 = also BVH4_CPU, bit2 = also BVH8_CPU, 0 = what the reference picks.
void *ref_scene_create(const float *verts4, const void *tris64, uint32_t n_tris, uint32_t variants)
{
	RefScene *s = new RefScene();
	s->verts.resize((size_t)n_tris * 3);
	std::memcpy(s->verts.data(), verts4, (size_t)n_tris * 48);
	s->tris.resize(n_tris);
	if (tris64) std::memcpy(s->tris.data(), tris64, (size_t)n_tris * 64);
	s->bvh2.Build(s->verts.data(), n_tris);
	s->use_avx2 = ref_has_avx2() != 0;
	bool want8 = variants ? (variants & 4u) != 0 : s->use_avx2;
	bool want4 = variants ? (variants & 2u) != 0 : !s->use_avx2;
	if (want8 && s->use_avx2) { s->bvh8.Build(s->verts.data(), n_tris); s->has8 = true; }
	if (want4) { s->bvh4.Build(s->verts.data(), n_tris); s->has4 = true; }
	return s;
}
void ref_scene_destroy(void *p) { delete static_cast<RefScene *>(p); }

uint32_t ref_bvh2_used_nodes(void *p) { return static_cast<RefScene *>(p)->bvh2.usedNodes; }
void ref_bvh2_copy(void *p, void *nodes32, uint32_t *prim_idx)
{
	RefScene *s = static_cast<RefScene *>(p);
	static_assert(sizeof(tinybvh::BVH::BVHNode) == sizeof(Node32), "BVHNode must be 32 bytes");
	std::memcpy(nodes32, s->bvh2.bvhNode, (size_t)s->bvh2.usedNodes * 32);
	std::memcpy(prim_idx, s->bvh2.primIdx, (size_t)s->bvh2.idxCount * 4);
}
void ref_bvh2_info(void *p, int32_t *node_count, int32_t *leaf_count, float *sah_cost)
{
	RefScene *s = static_cast<RefScene *>(p);
	*node_count = s->bvh2.NodeCount();
	*leaf_count = s->bvh2.LeafCount();
	*sah_cost = s->bvh2.SAHCost();
}

// RayScene::cast_ray, src/accel/ray_scene.h:90-118, through
// to_tinybvh_ray (src/accel/tinybvh_adapter.h:71-84) and
// from_tinybvh_intersection (:97-116).  variant: 2 = BVH::Intersect,
// 4 = BVH4_CPU, 8 = BVH8_CPU, 0 = reference choice (AVX2 ? 8 : 4).
// prim is the primitive index (== Triangle::id for flat scenes, raytracer_server.cpp:700-711).
void ref_cast_rays(void *p, const void *rays32, void *hits32, int64_t count, uint32_t query_mask, int variant, int n_threads)
{
	RefScene *s = static_cast<RefScene *>(p);
	const Ray32 *rays = static_cast<const Ray32 *>(rays32);
	Hit32 *hits = static_cast<Hit32 *>(hits32);
	if (variant == 0) variant = s->has8 ? 8 : (s->has4 ? 4 : 2);
	range_split(count, n_threads, 128 /* MIN_BATCH_FOR_THREADING, ray_dispatcher.h:423 */, [&](int64_t a, int64_t b) {
		for (int64_t i = a; i < b; i++) {
			const Ray32 &r = rays[i];
			tinybvh::Ray tr(tinybvh::bvhvec3(r.o[0], r.o[1], r.o[2]), tinybvh::bvhvec3(r.d[0], r.d[1], r.d[2]), r.t_max);
			if (variant == 8) s->bvh8.Intersect(tr);
			else if (variant == 4) s->bvh4.Intersect(tr);
			else s->bvh2.Intersect(tr);
			Hit32 &h = hits[i];
			bool hit = tr.hit.t < 1e30f && tr.hit.prim != 0xFFFFFFFFu;
			if (hit && tr.hit.prim < s->tris.size()) { // query-mask post filter, ray_scene.h:105-112
				const Tri64 &t = s->tris[tr.hit.prim];
				if ((t.layers & query_mask) == 0) hit = false;
				else {
					h.t = tr.hit.t; h.prim = (int32_t)tr.hit.prim; h.u = tr.hit.u; h.v = tr.hit.v;
					h.n[0] = t.n[0]; h.n[1] = t.n[1]; h.n[2] = t.n[2]; h.layers = t.layers;
				}
			}
			if (!hit) { h.t = FLT_MAX; h.prim = -1; h.u = h.v = 0; h.n[0] = h.n[1] = h.n[2] = 0; h.layers = 0; } // Intersection{}, intersection.h:44-45
		}
	});
}

// RayScene::any_hit, src/accel/ray_scene.h:136-149 (unmasked path only).
void ref_any_hit(void *p, const void *rays32, uint8_t *out, int64_t count, int variant, int n_threads)
{
	RefScene *s = static_cast<RefScene *>(p);
	const Ray32 *rays = static_cast<const Ray32 *>(rays32);
	if (variant == 0) variant = s->has8 ? 8 : (s->has4 ? 4 : 2);
	range_split(count, n_threads, 128, [&](int64_t a, int64_t b) {
		for (int64_t i = a; i < b; i++) {
			const Ray32 &r = rays[i];
			tinybvh::Ray tr(tinybvh::bvhvec3(r.o[0], r.o[1], r.o[2]), tinybvh::bvhvec3(r.d[0], r.d[1], r.d[2]), r.t_max);
			bool occ = variant == 8 ? s->bvh8.IsOccluded(tr) : (variant == 4 ? s->bvh4.IsOccluded(tr) : s->bvh2.IsOccluded(tr));
			out[i] = occ ? 1 : 0;
		}
	});
}

// sort_rays_by_direction keys, src/dispatch/ray_sort.h:64-76 is Vector3-typed
// (godot-cpp) and cannot be compiled here; no entry point for it.

} // extern "C"
