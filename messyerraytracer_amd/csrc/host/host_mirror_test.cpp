// host_mirror_test.cpp — drives the RayDispatcher / GPURayCaster mirrors the way
// the reference's callers do (RayTracerServer::build / cast_rays_batch / submit,
// src/godot/raytracer_server.cpp:161-181,285-328) and dumps the results for
// tests/test_host_mirror_gpu.py, which checks them against the oracle.
//
// usage: host_mirror_test <in.bin> <out.bin>
//   in : u32 n_tris, f32 verts[n_tris*9], u32 n_rays, Ray rays[n_rays] (60 B each)
//   out: i32 header[8], then Intersection[n] coherent, Intersection[n] sorted,
//        u8[n] any-hit, Intersection[n] async, Intersection single, Intersection[n] after a device-side rebuild,
//        Intersection[n] from the CPU backend (Backend::CPU, the router's default, selected explicitly)
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ray_dispatcher.hpp"

using namespace mrt;

int main(int argc, char **argv)
{
	if (argc != 3) { std::fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
	FILE *f = std::fopen(argv[1], "rb");
	if (!f) return 2;
	uint32_t n_tris = 0, n_rays = 0;
	if (std::fread(&n_tris, 4, 1, f) != 1) return 2;
	std::vector<float> verts((size_t)n_tris * 9);
	if (std::fread(verts.data(), 4, verts.size(), f) != verts.size()) return 2;
	if (std::fread(&n_rays, 4, 1, f) != 1) return 2;
	std::vector<Ray> rays(n_rays);
	if (std::fread((void *)rays.data(), sizeof(Ray), n_rays, f) != n_rays) return 2;
	std::fclose(f);

	RayDispatcher disp;
	int header[8] = {0};
	// 1. Backend::GPU / AUTO before any device is initialised: must fail loudly (MRT_ERR_NO_DEVICE), not fall back to the CPU.
	{
		std::vector<Intersection> tmp(n_rays);
		disp.set_backend(RayDispatcher::Backend::AUTO);
		header[0] = disp.cast_rays(rays.data(), tmp.data(), (int)n_rays);
		disp.set_backend(RayDispatcher::Backend::CPU);
	}
	// 2. flatten + id rule of _rebuild_scene (raytracer_server.cpp:700-711): running ids, layers all ones.
	for (uint32_t i = 0; i < n_tris; i++) {
		const float *v = &verts[(size_t)i * 9];
		disp.scene().triangles.push_back(Triangle(Vector3(v[0], v[1], v[2]), Vector3(v[3], v[4], v[5]), Vector3(v[6], v[7], v[8]), i));
	}
	// 3. set_backend(GPU) lazily initialises, then build() uploads (raytracer_server.cpp:334-366,161-181).
	disp.set_backend(RayDispatcher::Backend::GPU);
	if (!disp.initialize_gpu(0)) { std::fprintf(stderr, "no GPU\n"); return 3; }
	disp.build();
	header[1] = disp.using_gpu() ? 1 : 0;
	header[2] = disp.triangle_count();
	header[3] = disp.bvh_node_count();
	header[4] = disp.bvh_depth();

	std::vector<Intersection> coherent(n_rays), sorted(n_rays), async_res(n_rays);
	std::vector<uint8_t> any(n_rays, 0);
	RayStats stats;
	header[5] = disp.cast_rays(rays.data(), coherent.data(), (int)n_rays, &stats, 0xFFFFFFFF, true);
	header[6] = disp.cast_rays(rays.data(), sorted.data(), (int)n_rays, &stats, 0xFFFFFFFF, false);
	header[7] = disp.any_hit_rays(rays.data(), reinterpret_cast<bool *>(any.data()), (int)n_rays, &stats);
	disp.submit_gpu_async(rays.data(), (int)n_rays);
	const bool was_pending = disp.has_gpu_pending();
	disp.collect_gpu_nearest(async_res.data(), (int)n_rays);
	if (!was_pending || disp.has_gpu_pending()) { std::fprintf(stderr, "async bookkeeping broken\n"); return 4; }
	Intersection single = disp.cast_ray(rays[0]);
	if (stats.rays_cast != 3ull * n_rays) { std::fprintf(stderr, "stats broken\n"); return 5; }
	// 4. the same scene rebuilt on the device from the triangles alone: the same records
	std::vector<Intersection> device_built(n_rays);
	disp.gpu_caster().build_scene_on_device(disp.scene().triangles);
	if (!disp.using_gpu()) { std::fprintf(stderr, "device build left no scene\n"); return 6; }
	disp.cast_rays(rays.data(), device_built.data(), (int)n_rays, nullptr, 0xFFFFFFFF, false);

	// 5. the router's other backend, selected explicitly: the same records from the CPU pool
	std::vector<Intersection> cpu(n_rays);
	disp.set_backend(RayDispatcher::Backend::CPU);
	RayStats cpu_stats;
	if (disp.cast_rays(rays.data(), cpu.data(), (int)n_rays, &cpu_stats) != MRT_OK || cpu_stats.rays_cast != n_rays) { std::fprintf(stderr, "CPU backend broken\n"); return 7; }

	FILE *o = std::fopen(argv[2], "wb");
	if (!o) return 2;
	std::fwrite(header, 4, 8, o);
	std::fwrite((const void *)coherent.data(), sizeof(Intersection), n_rays, o);
	std::fwrite((const void *)sorted.data(), sizeof(Intersection), n_rays, o);
	std::fwrite(any.data(), 1, n_rays, o);
	std::fwrite((const void *)async_res.data(), sizeof(Intersection), n_rays, o);
	std::fwrite((const void *)&single, sizeof(Intersection), 1, o);
	std::fwrite((const void *)device_built.data(), sizeof(Intersection), n_rays, o);
	std::fwrite((const void *)cpu.data(), sizeof(Intersection), n_rays, o);
	std::fclose(o);
	return 0;
}
