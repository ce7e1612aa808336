// host_tlas_test.cpp — the router with a TLAS set (RayDispatcher::set_tlas, src/dispatch/ray_dispatcher.h:82-88,443-452):
// both backends walk the two-level scene, nothing flattened.  tests/test_host_cpu.py (CPU backend, CPU tier) and
// tests/test_host_server_gpu.py (device backend) hold the records to the oracle's restatement of SceneTLAS.
//
// usage: host_tlas_test <in.bin> <out.bin> [cpu | gpu]
//   in : u32 n_mesh_tris, f32 verts[n_mesh_tris * 9], u32 n_instances, mrt_instance[n_instances] (64 B each),
//        u32 n_rays, Ray rays[n_rays] (60 B each), u32 query_mask
//   out: i32 header[8], Intersection[n] (cast_rays, stats), u8[n] (any_hit_rays), Intersection[min(n, 32)] (cast_ray), u8[min(n, 32)] (any_hit)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ray_dispatcher.hpp"

using namespace mrt;

int main(int argc, char **argv)
{
	if (argc != 3 && argc != 4) { std::fprintf(stderr, "usage: %s in.bin out.bin [cpu|gpu]\n", argv[0]); return 2; }
	const bool gpu = argc == 4 && !std::strcmp(argv[3], "gpu");
	FILE *f = std::fopen(argv[1], "rb");
	if (!f) return 2;
	SceneTLAS tlas;
	uint32_t n_mesh_tris = 0, n_inst = 0, n_rays = 0, query_mask = 0;
	if (std::fread(&n_mesh_tris, 4, 1, f) != 1) return 2;
	tlas.mesh_vertices.resize((size_t)n_mesh_tris * 9);
	if (std::fread(tlas.mesh_vertices.data(), 4, tlas.mesh_vertices.size(), f) != tlas.mesh_vertices.size()) return 2;
	if (std::fread(&n_inst, 4, 1, f) != 1) return 2;
	tlas.instances.resize(n_inst);
	static_assert(sizeof(mrt_instance) == 64, "mrt_instance is 64 bytes");
	if (std::fread(tlas.instances.data(), sizeof(mrt_instance), n_inst, f) != n_inst) return 2;
	if (std::fread(&n_rays, 4, 1, f) != 1) return 2;
	std::vector<Ray> rays(n_rays);
	if (std::fread((void *)rays.data(), sizeof(Ray), n_rays, f) != n_rays) return 2;
	if (std::fread(&query_mask, 4, 1, f) != 1) return 2;
	std::fclose(f);

	int header[8] = {0};
	header[0] = tlas.build_tlas() ? 1 : 0;
	RayDispatcher disp;
	disp.set_tlas(&tlas);
	header[1] = disp.has_tlas() ? 1 : 0;
	if (gpu) {
		disp.set_backend(RayDispatcher::Backend::GPU);
		if (!disp.initialize_gpu(0)) { std::fprintf(stderr, "no GPU\n"); return 3; }
		disp.upload_to_gpu();
	}
	header[2] = disp.using_gpu() ? 1 : 0;
	std::vector<Intersection> nearest(n_rays);
	std::vector<uint8_t> any(n_rays, 0);
	RayStats stats;
	header[3] = disp.cast_rays(rays.data(), nearest.data(), (int)n_rays, &stats, query_mask);
	header[4] = disp.any_hit_rays(rays.data(), reinterpret_cast<bool *>(any.data()), (int)n_rays, nullptr, query_mask);
	header[5] = (int)stats.rays_cast;
	header[6] = (int)stats.hits;
	const uint32_t n_single = n_rays < 32u ? n_rays : 32u;
	std::vector<Intersection> single(n_single);
	std::vector<uint8_t> single_any(n_single, 0);
	for (uint32_t i = 0; i < n_single; i++) {
		single[i] = disp.cast_ray(rays[i], nullptr, query_mask);
		single_any[i] = disp.any_hit(rays[i], nullptr, query_mask) ? 1 : 0;
	}
	FILE *o = std::fopen(argv[2], "wb");
	if (!o) return 2;
	std::fwrite(header, 4, 8, o);
	std::fwrite((const void *)nearest.data(), sizeof(Intersection), n_rays, o);
	std::fwrite(any.data(), 1, n_rays, o);
	std::fwrite((const void *)single.data(), sizeof(Intersection), n_single, o);
	std::fwrite(single_any.data(), 1, n_single, o);
	std::fclose(o);
	return 0;
}
