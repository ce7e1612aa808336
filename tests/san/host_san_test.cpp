// host_san_test.cpp — the host-side C++ of libmrt_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer (and, in a
// second build, ThreadSanitizer): the SAH builder on several thread counts, the wide-node conversions (2-, 4-, 8-wide),
// the two-level preparation and refit, the BVH cache file incl. every truncation and a sweep of corrupted bytes, the
// router's CPU backend over its thread pool.  No device, no HIP: compiled by tests/test_sanitizers_cpu.py with g++ from
// the sources where they lie (csrc/host/*.cpp).  Exit code 0 and an empty sanitizer report = pass.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../messyerraytracer_amd/csrc/mrt_internal.h"
#include "../../messyerraytracer_amd/csrc/host/cpu_backend.hpp"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float frand()
{
	rng_state += 0x9E3779B97F4A7C15ull;
	uint64_t z = rng_state;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
	return (float)(z >> 40) * (1.0f / 16777216.0f);
}

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); return 1; } } while (0)

static std::vector<float> soup9(uint32_t n, float s)
{
	std::vector<float> v((size_t)n * 9);
	for (uint32_t i = 0; i < n; i++) {
		const float c[3] = { frand() * 10.0f - 5.0f, frand() * 10.0f - 5.0f, frand() * 10.0f - 5.0f };
		for (int k = 0; k < 9; k++) v[(size_t)i * 9 + k] = c[k % 3] + (frand() * 2.0f - 1.0f) * s;
	}
	return v;
}

int main(int argc, char **argv)
{
	const char *tmp = argc > 1 ? argv[1] : "/tmp/mrt_san_cache.bin";
	char err[256];
	for (uint32_t n : { 1u, 2u, 3u, 17u, 1000u, 60000u }) { // 60 000 > 50 000: the threaded build (tiny_bvh.h:2335-2345)
		std::vector<float> v9 = soup9(n, 0.3f);
		std::vector<mrt_tri64> tris(n);
		CHECK(mrt_make_triangles(v9.data(), nullptr, nullptr, n, tris.data()) == MRT_OK);
		std::vector<float> v4((size_t)n * 12, 0.0f);
		for (uint32_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) v4[(size_t)i * 12 + k * 4 + c] = v9[(size_t)i * 9 + k * 3 + c];
		std::vector<mrt_bvh_node32> nodes((size_t)2 * n + 2), nodes_b((size_t)2 * n + 2);
		std::vector<uint32_t> prim(n), prim_b(n);
		uint32_t used = 0, used_b = 0;
		CHECK(mrt_bvh2_build(v4.data(), n, nodes.data(), prim.data(), &used, 1) == MRT_OK);
		CHECK(mrt_bvh2_build(v4.data(), n, nodes_b.data(), prim_b.data(), &used_b, 5) == MRT_OK);
		CHECK(used == used_b && std::memcmp(nodes.data(), nodes_b.data(), (size_t)used * sizeof(mrt_bvh_node32)) == 0); // deterministic for any thread count
		CHECK(std::memcmp(prim.data(), prim_b.data(), (size_t)n * 4) == 0);
		// the device layouts (2-wide, 4-wide, 8-wide compressed, exact leaf boxes, hot / cold triangles)
		mrt::DeviceSceneHost h;
		h.want8 = true;
		CHECK(mrt::prepare_scene(tris.data(), n, nodes.data(), used, prim.data(), &h, err, sizeof(err)) == MRT_OK);
		CHECK(h.n_tris == n && h.nodes && h.hot && h.cold && h.n_nodes >= 1);
		std::free(h.nodes); std::free(h.nodes4); std::free(h.nodes8); std::free(h.leaf_box); std::free(h.hot); std::free(h.cold);
		// a damaged BVH must be refused, not followed
		if (used > 3) {
			std::vector<mrt_bvh_node32> bad(nodes.begin(), nodes.begin() + used);
			bad[0].left_first = used + 7;
			mrt::DeviceSceneHost hb;
			CHECK(mrt::prepare_scene(tris.data(), n, bad.data(), used, prim.data(), &hb, err, sizeof(err)) != MRT_OK);
			std::vector<uint32_t> bad_prim(prim);
			bad_prim[0] = n + 3;
			CHECK(mrt::prepare_scene(tris.data(), n, nodes.data(), used, bad_prim.data(), &hb, err, sizeof(err)) != MRT_OK);
		}
		// cache file: round trip, then every truncation and a sweep of single corrupted bytes
		CHECK(mrt_bvh2_save(tmp, nodes.data(), used, prim.data(), n) == MRT_OK);
		std::vector<mrt_bvh_node32> nl((size_t)2 * n + 2); std::vector<uint32_t> pl(n); uint32_t ul = 0;
		CHECK(mrt_bvh2_load(tmp, n, nl.data(), pl.data(), &ul) == MRT_OK && ul == used);
		CHECK(std::memcmp(nl.data(), nodes.data(), (size_t)used * sizeof(mrt_bvh_node32)) == 0 && std::memcmp(pl.data(), prim.data(), (size_t)n * 4) == 0);
		CHECK(mrt_bvh2_load(tmp, n + 1, nl.data(), pl.data(), &ul) != MRT_OK); // saved for another triangle count
		if (n <= 1000) {
			FILE *f = std::fopen(tmp, "rb"); CHECK(f);
			std::vector<unsigned char> bytes;
			for (int c; (c = std::fgetc(f)) != EOF;) bytes.push_back((unsigned char)c);
			std::fclose(f);
			const size_t step_t = bytes.size() > 4000 ? 97 : 1, step_c = bytes.size() > 4000 ? 131 : 3;
			for (size_t cut = 0; cut < bytes.size(); cut += step_t) {
				f = std::fopen(tmp, "wb"); CHECK(f); if (cut) std::fwrite(bytes.data(), 1, cut, f); std::fclose(f);
				CHECK(mrt_bvh2_load(tmp, n, nl.data(), pl.data(), &ul) != MRT_OK);
			}
			for (size_t at = 0; at < bytes.size(); at += step_c) {
				std::vector<unsigned char> c(bytes);
				c[at] ^= 0x5A;
				f = std::fopen(tmp, "wb"); CHECK(f); std::fwrite(c.data(), 1, c.size(), f); std::fclose(f);
				CHECK(mrt_bvh2_load(tmp, n, nl.data(), pl.data(), &ul) != MRT_OK); // the checksum catches it
			}
		}
		std::remove(tmp);
		// the router's CPU backend over its pool: same records from 1 chunk and from many
		if (n >= 17) {
			std::vector<mrt::Triangle> ht(n);
			for (uint32_t i = 0; i < n; i++) {
				const float *p = &v9[(size_t)i * 9];
				ht[i] = mrt::Triangle(mrt::Vector3(p[0], p[1], p[2]), mrt::Vector3(p[3], p[4], p[5]), mrt::Vector3(p[6], p[7], p[8]), i);
			}
			const int nr = 5000;
			std::vector<mrt::Ray> rays;
			for (int i = 0; i < nr; i++) {
				mrt::Vector3 d(frand() - 0.5f, frand() - 0.5f, frand() - 0.5f);
				rays.push_back(mrt::Ray(mrt::Vector3(frand() * 12.0f - 6.0f, frand() * 12.0f - 6.0f, -12.0f), d.normalized()));
			}
			mrt::CpuWalker w(ht.data(), nodes.data(), prim.data(), n);
			std::vector<mrt::Intersection> serial(nr), pooled(nr);
			for (int i = 0; i < nr; i++) serial[i] = w.cast(rays[i], 0xFFFFFFFFu, false, nullptr);
			mrt::ThreadPool pool(3);
			std::vector<mrt::RayStats> slots(pool.thread_count() + 1);
			std::atomic<uint32_t> slot{0};
			for (int rep = 0; rep < 20; rep++) { // many generations through the same pool
				slot = 0; for (auto &s : slots) s.reset();
				pool.dispatch_and_wait(nr, 128, [&](int a, int b) {
					mrt::RayStats &local = slots[slot.fetch_add(1)];
					for (int i = a; i < b; i++) pooled[i] = w.cast(rays[i], 0xFFFFFFFFu, false, &local);
				});
				mrt::RayStats sum; for (auto &s : slots) sum += s;
				CHECK(sum.rays_cast == (uint64_t)nr);
				CHECK(std::memcmp((const void *)serial.data(), (const void *)pooled.data(), sizeof(mrt::Intersection) * nr) == 0);
			}
		}
	}
	// two-level preparation + refit (SceneTLAS / MeshBLAS / BLASInstance on the host side)
	{
		const uint32_t per = 400, n_mesh = 3, n_inst = 7;
		std::vector<float> v9 = soup9(per * n_mesh, 0.2f);
		std::vector<mrt_instance> inst(n_inst);
		for (uint32_t i = 0; i < n_inst; i++) {
			mrt_instance &in = inst[i];
			std::memset(&in, 0, sizeof(in));
			in.first_tri = (i % n_mesh) * per; in.n_tris = per; in.layers = 1u << (i % 5);
			in.basis[0] = in.basis[4] = in.basis[8] = 1.0f + 0.1f * (float)i;
			in.origin[0] = (float)i * 1.5f - 4.0f; in.origin[1] = frand(); in.origin[2] = frand();
		}
		mrt::TwoLevelHost h;
		CHECK(mrt::prepare_two_level(v9.data(), per * n_mesh, inst.data(), n_inst, 3, true, &h, err, sizeof(err)) == MRT_OK);
		CHECK(h.n_inst == n_inst && h.n_blas == n_mesh && h.nodes && h.inst);
		for (uint32_t i = 0; i < n_inst; i++) inst[i].origin[2] += 0.75f;
		CHECK(mrt::refit_two_level(&h, inst.data(), n_inst, err, sizeof(err)) == MRT_OK);
		inst[2].basis[0] = inst[2].basis[4] = inst[2].basis[8] = 0.0f; // singular transform: refused
		CHECK(mrt::refit_two_level(&h, inst.data(), n_inst, err, sizeof(err)) != MRT_OK);
		mrt::free_two_level(&h);
		inst[0].first_tri = per * n_mesh; // outside the mesh array
		mrt::TwoLevelHost h2;
		CHECK(mrt::prepare_two_level(v9.data(), per * n_mesh, inst.data(), n_inst, 1, true, &h2, err, sizeof(err)) != MRT_OK);
	}
	std::puts("host_san_test ok");
	return 0;
}
