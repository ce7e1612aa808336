// Microbenchmark: VALU floor of the packet node step on gfx950 (12 v_fma with an SGPR operand,
// 4 v_max/v_min, 4 v_max3/v_min3, 2 v_cmp), no memory traffic, as a function of waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/slab_rate.hip -o tools/ubench/slab_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 20000
__global__ __launch_bounds__(256) void k(float *out, float s0, float s1, float s2, float s3)
{
	float ix = 1.0f + threadIdx.x * 1e-3f, iy = 0.9f, iz = 1.1f, nrx = -0.5f, nry = 0.25f, nrz = 0.125f, tmin = 0.001f, lim = 1e30f;
	unsigned acc = 0;
	for (int i = 0; i < ITER; i++) {
		unsigned long long m0, m1;
		asm volatile(
			"v_fma_f32 v40, %4, %8, %11\n v_fma_f32 v41, %5, %9, %12\n v_fma_f32 v42, %6, %10, %13\n"
			"v_fma_f32 v43, %7, %8, %11\n v_fma_f32 v44, %4, %9, %12\n v_fma_f32 v45, %5, %10, %13\n"
			"v_fma_f32 v46, %6, %8, %11\n v_fma_f32 v47, %7, %9, %12\n v_fma_f32 v48, %4, %10, %13\n"
			"v_fma_f32 v49, %5, %8, %11\n v_fma_f32 v50, %6, %9, %12\n v_fma_f32 v51, %7, %10, %13\n"
			"v_max_f32 v42, v42, %14\n v_max3_f32 v40, v40, v41, v42\n v_min_f32 v45, v45, %15\n v_min3_f32 v43, v43, v44, v45\n"
			"v_max_f32 v48, v48, %14\n v_max3_f32 v46, v46, v47, v48\n v_min_f32 v51, v51, %15\n v_min3_f32 v49, v49, v50, v51\n"
			"v_cmp_le_f32 vcc, v40, v43\n v_cmp_le_f32_e64 %1, v46, v49\n s_mov_b64 %0, vcc\n"
			: "=s"(m0), "=s"(m1), "+v"(ix), "+v"(iy)
			: "s"(s0), "s"(s1), "s"(s2), "s"(s3), "v"(ix), "v"(iy), "v"(iz), "v"(nrx), "v"(nry), "v"(nrz), "v"(tmin), "v"(lim)
			: "vcc", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
		acc += (unsigned)m0 + (unsigned)m1;
	}
	if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
int main()
{
	float *d; (void)hipMalloc(&d, 1 << 22);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for (int wps : {1, 2, 4, 8}) {
		const int blocks = 256 * wps; // 256-thread blocks: wps blocks per CU = wps waves per SIMD
		float best = 1e9;
		for (int rep = 0; rep < 3; rep++) {
			(void)hipEventRecord(e0);
			hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 2.0f, 3.0f, 4.0f);
			(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
			float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
		}
		// per SIMD: wps waves x ITER steps
		const double cyc_per_step = best * 1e-3 * 2.4e9 / ((double)wps * ITER);
		printf("waves/SIMD %d: %.3f ms -> %.1f cycles of SIMD time per 22-VALU node step (%.2f cycles per VALU instr) @2.4GHz\n",
			wps, best, cyc_per_step, cyc_per_step / 22.0);
	}
	return 0;
}
