// ubench/exec_mask.hip — does a wave64 vector instruction cost less when half (or most) of EXEC is off?  gfx950, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define R10(X) X X X X X X X X X X
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k(uint32_t iters, unsigned long long mask, unsigned long long *cycles, float *sink, float sa, float sb)
{
	float a = 1.0001f + threadIdx.x * 1e-7f, b = 0.5f;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	asm volatile("s_mov_b64 exec, %0\n" : : "s"(mask));
	for (uint32_t i = 0; i < iters; i++) {
		asm volatile(R10("v_fma_f32 v40, %[sa], %[a], %[b]\n v_fma_f32 v41, %[sb], %[a], %[b]\n v_max3_f32 v42, v40, %[a], %[b]\n v_min_f32 v43, %[b], %[a]\n")
				: : [a] "v"(a), [b] "v"(b), [sa] "s"(sa), [sb] "s"(sb) : "v40", "v41", "v42", "v43");
	}
	asm volatile("s_mov_b64 exec, -1\n" ::);
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if ((threadIdx.x & 63u) == 0) atomicAdd(cycles, t1 - t0);
	float r; asm volatile("v_add_f32 %0, v40, v41\n" : "=v"(r));
	sink[(blockIdx.x * 256u + threadIdx.x) & 1023u] = r;
}
int main()
{
	unsigned long long *d_cyc; float *d_sink;
	CHECK(hipMalloc(&d_cyc, 8)); CHECK(hipMalloc(&d_sink, 4096));
	const uint32_t iters = 2000, blocks = 256 * 8;
	const unsigned long long masks[] = { ~0ull, 0x00000000FFFFFFFFull, 0xFFFFFFFF00000000ull, 0x000000000000FFFFull, 0x00000000000000FFull, 0x0000000000000001ull, 0x0000000100000001ull, 0x00FF00FF00FF00FFull };
	for (unsigned long long m : masks) {
		hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		float ms = 0;
		for (int rep = 0; rep < 2; rep++) {
			CHECK(hipMemset(d_cyc, 0, 8));
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, iters, m, d_cyc, d_sink, 1.5f, 2.5f);
			CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
			CHECK(hipEventElapsedTime(&ms, e0, e1));
		}
		unsigned long long cyc = 0; CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
		std::printf("exec %016llx  ticks/iteration/wave %8.1f  kernel %.3f ms\n", m, (double)cyc / (blocks * 4.0) / iters, ms);
	}
	return 0;
}
