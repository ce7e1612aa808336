// Microbenchmark: is v_pk_fma_f32 (2 fp32 FMAs per lane) issued at the rate of v_fma_f32 on gfx950?
// hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_fma.hip -o /tmp/pk_fma && /tmp/pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float s)
{
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
	const f2 m = {s, s + 1e-7f}, c = {1e-3f, 2e-3f};
	for (int i = 0; i < ITER; i++) {
		if (MODE == 0) { // 8 independent scalar FMAs
			a0 = __builtin_fmaf(a0, s, 1e-3f); a1 = __builtin_fmaf(a1, s, 1e-3f); a2 = __builtin_fmaf(a2, s, 1e-3f); a3 = __builtin_fmaf(a3, s, 1e-3f);
			a4 = __builtin_fmaf(a4, s, 1e-3f); a5 = __builtin_fmaf(a5, s, 1e-3f); a6 = __builtin_fmaf(a6, s, 1e-3f); a7 = __builtin_fmaf(a7, s, 1e-3f);
		} else { // 4 packed FMAs = the same 8 FMAs
			p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c);
			p2 = __builtin_elementwise_fma(p2, m, c); p3 = __builtin_elementwise_fma(p3, m, c);
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = MODE == 0 ? a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 : p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
int main()
{
	float *d; hipMalloc(&d, 256 * 2048 * 4 * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int mode = 0; mode < 2; mode++) for (int rep = 0; rep < 3; rep++) {
		hipEventRecord(e0);
		if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(8192), dim3(256), 0, 0, d, 0.999f);
		else hipLaunchKernelGGL(k<1>, dim3(8192), dim3(256), 0, 0, d, 0.999f);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		const double fma = 8192.0 * 256 * ITER * 8;
		printf("mode %s rep %d: %.3f ms  %.1f TFLOP/s (2 flop per fma)\n", mode ? "v_pk_fma_f32" : "v_fma_f32   ", rep, ms, fma * 2 / ms / 1e9);
	}
	return 0;
}
