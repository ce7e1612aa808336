// host_cpu_test.cpp — the host-side mirror WITHOUT a device: RayTracerServer semantics over the
// router's CPU backend (a8, a12), driven the way the reference's callers drive theirs.  Runs in the CPU test tier
// (tests/test_host_cpu.py), which holds the records to the oracle.
//
// The same driver runs the GPU tier (tests/test_host_server_gpu.py): with backend = gpu the server's set_backend(BACKEND_GPU)
// initialises the device lazily and build() uploads, and every call below goes through the C-ABI.
//
// usage: host_cpu_test <in.bin> <out.bin> [cpu | gpu | auto | gpu-fallback | auto-fallback]   (default cpu)
//   in : u32 n_meshes, per mesh { u32 n_tris, f32 basis[9], f32 origin[3], u32 layer_mask, f32 verts[n_tris*9] },
//        u32 n_rays, Ray rays[n_rays] (60 B each), u32 query_mask
//   out: i32 header[16], Intersection[n] (submit NEAREST, stats), u8[n] (submit ANY_HIT), Intersection[n] (cast_rays_batch,
//        no stats), RayHit-as-Intersection[min(n,64)] from cast_ray(origin, 3 * direction), u8[min(n,64)] any_hit(max_distance = 5),
//        Intersection[n] (submit NEAREST with the coherent hint)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ray_tracer_server.hpp"

using namespace mrt;

int main(int argc, char **argv)
{
	if (argc != 3 && argc != 4) { std::fprintf(stderr, "usage: %s in.bin out.bin [cpu|gpu|auto|gpu-fallback|auto-fallback]\n", argv[0]); return 2; }
	const char *mode = argc == 4 ? argv[3] : "cpu";
	const bool want_gpu = !std::strcmp(mode, "gpu") || !std::strcmp(mode, "gpu-fallback");
	const bool want_auto = !std::strcmp(mode, "auto") || !std::strcmp(mode, "auto-fallback");
	const bool fallback = std::strstr(mode, "fallback") != nullptr;
	FILE *f = std::fopen(argv[1], "rb");
	if (!f) return 2;
	RayTracerServer server;
	uint32_t n_meshes = 0;
	if (std::fread(&n_meshes, 4, 1, f) != 1) return 2;
	for (uint32_t m = 0; m < n_meshes; m++) {
		uint32_t n_tris = 0, mask = 0; Transform3D xf; float o[3];
		if (std::fread(&n_tris, 4, 1, f) != 1 || std::fread(xf.basis, 4, 9, f) != 9 || std::fread(o, 4, 3, f) != 3 || std::fread(&mask, 4, 1, f) != 1) return 2;
		xf.origin = Vector3(o[0], o[1], o[2]);
		std::vector<float> v((size_t)n_tris * 9);
		if (std::fread(v.data(), 4, v.size(), f) != v.size()) return 2;
		std::vector<Vector3> verts;
		for (size_t i = 0; i < v.size(); i += 3) verts.push_back(Vector3(v[i], v[i + 1], v[i + 2]));
		server.register_mesh(verts, xf, mask);
	}
	uint32_t n_rays = 0, query_mask = 0;
	if (std::fread(&n_rays, 4, 1, f) != 1) return 2;
	std::vector<Ray> rays(n_rays);
	if (std::fread((void *)rays.data(), sizeof(Ray), n_rays, f) != n_rays) return 2;
	if (std::fread(&query_mask, 4, 1, f) != 1) return 2;
	std::fclose(f);

	int header[16] = {0};
	// the reference's default backend is the CPU (ray_dispatcher.h:404, raytracer_server.h backend_mode_)
	header[0] = server.get_backend();
	if (fallback) server.set_cpu_fallback(true); // opt-in: the reference's degradation to the CPU pool
	if (want_gpu) server.set_backend(RayTracerServer::BACKEND_GPU);   // lazy initialisation (raytracer_server.cpp:334-366)
	if (want_auto) server.set_backend(RayTracerServer::BACKEND_AUTO);
	server.build();                                                   // ... and build() uploads when a device backend is active
	header[1] = server.get_triangle_count(); header[2] = server.get_mesh_count();
	header[3] = server.get_bvh_node_count(); header[4] = server.get_bvh_depth(); header[5] = server.get_thread_count();

	std::vector<Intersection> nearest(n_rays), batch(n_rays);
	std::vector<uint8_t> any(n_rays, 0);
	RayQuery q = RayQuery::nearest(rays.data(), (int)n_rays, query_mask);
	q.collect_stats = true;
	RayQueryResult res; res.hits = nearest.data();
	header[6] = server.submit(q, res);
	header[7] = res.count; header[8] = (int)res.stats.rays_cast; header[9] = (int)res.stats.hits;
	// CPU pool: all four counters; device: rays_cast only (ray_scene.h:94,115 after the TinyBVH migration keep as little)
	const bool on_device = server.dispatcher().using_gpu();
	header[10] = res.elapsed_ms > 0.0f && (on_device || (res.stats.bvh_nodes_visited > 0 && res.stats.tri_tests > 0)) ? 1 : 0;
	RayQuery qa = RayQuery::any_hit(rays.data(), (int)n_rays, query_mask);
	RayQueryResult resa; resa.hit_flags = reinterpret_cast<bool *>(any.data());
	header[11] = server.submit(qa, resa);
	header[12] = server.cast_rays_batch(rays.data(), batch.data(), (int)n_rays, nullptr, query_mask);
	// cast_ray normalises the direction (3 * d here) and any_hit limits the ray to max_distance
	const uint32_t n_single = n_rays < 64u ? n_rays : 64u;
	std::vector<Intersection> single(n_single);
	std::vector<uint8_t> single_any(n_single, 0);
	for (uint32_t i = 0; i < n_single; i++) {
		const RayHit h = server.cast_ray(rays[i].origin, rays[i].direction * 3.0f, (int)(query_mask & 0x7FFFFFFFu));
		Intersection &r = single[i];
		r.t = h.distance; r.position = h.position; r.normal = h.normal; r.prim_id = (uint32_t)h.prim_id; r.hit_layers = (uint32_t)h.hit_layers;
		single_any[i] = server.any_hit(rays[i].origin, rays[i].direction * 3.0f, 5.0f, (int)(query_mask & 0x7FFFFFFFu)) ? 1 : 0;
	}
	// the coherent hint (RayQuery::coherent, ray_query.h:69-76): same records, no sort on the device path
	std::vector<Intersection> coherent(n_rays);
	RayQuery qc = RayQuery::nearest(rays.data(), (int)n_rays, query_mask);
	qc.coherent = true;
	RayQueryResult resc; resc.hits = coherent.data();
	const int rc_coherent = server.submit(qc, resc);
	if (want_gpu || want_auto) {
		header[13] = server.is_gpu_available() ? 1 : 0;
		header[14] = on_device ? 1 : 0;
		header[15] = (server.dispatcher().used_cpu_fallback() ? 1 : 0) | (rc_coherent << 8) | (server.get_backend() << 16);
	} else {
		// no device selected: the GPU and AUTO backends must say so, not hand the batch to the CPU pool
		server.set_backend(RayTracerServer::BACKEND_GPU);
		header[13] = server.cast_rays_batch(rays.data(), batch.data(), 0, nullptr, query_mask) == MRT_OK ? 0 : 1; // count 0: silent no-op (cpp:419)
		std::vector<Intersection> tmp(n_rays);
		header[14] = server.is_gpu_available() ? -1 : server.cast_rays_batch(rays.data(), tmp.data(), (int)n_rays);
		server.set_backend(RayTracerServer::BACKEND_AUTO);
		header[15] = server.is_gpu_available() ? -1 : server.cast_rays_batch(rays.data(), tmp.data(), (int)n_rays);
	}

	FILE *o = std::fopen(argv[2], "wb");
	if (!o) return 2;
	std::fwrite(header, 4, 16, o);
	std::fwrite((const void *)nearest.data(), sizeof(Intersection), n_rays, o);
	std::fwrite(any.data(), 1, n_rays, o);
	std::fwrite((const void *)batch.data(), sizeof(Intersection), n_rays, o);
	std::fwrite((const void *)single.data(), sizeof(Intersection), n_single, o);
	std::fwrite(single_any.data(), 1, n_single, o);
	std::fwrite((const void *)coherent.data(), sizeof(Intersection), n_rays, o);
	std::fclose(o);
	return 0;
}
