// Measured HBM ceilings on the box ("confirm the vendor 8 TB/s and report the measured
// copy / triad ceiling"): read-only sum, copy and triad over 4 GiB arrays (far beyond the 256 MiB
// Infinity Cache), 16 bytes per lane, grid-stride, and hipMemcpyDtoD for comparison.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_stream.hip -o tools/ubench/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_read(const float4 *a, size_t n, float *out)
{
	float acc = 0.0f;
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) { const float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
	if (acc == 12345.678f) out[0] = acc; // never true: keeps the loads
}
__global__ __launch_bounds__(256) void k_copy(const float4 *a, float4 *b, size_t n)
{
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_triad(const float4 *a, const float4 *b, float4 *c, size_t n, float s)
{
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
		const float4 x = a[i], y = b[i];
		c[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
	}
}

template <class F> static float best_ms(F f)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	float best = 1e9f;
	for (int r = 0; r < 5; r++) {
		(void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
	}
	return best;
}

int main()
{
	const size_t bytes = 4ull << 30, n = bytes / 16;
	float4 *a, *b, *c; float *out;
	if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&c, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void)hipMalloc(&out, 16);
	(void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes); (void)hipMemset(c, 0, bytes);
	for (int blocks : {256 * 8, 256 * 16, 256 * 32}) {
		const float r = best_ms([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, out); });
		const float cp = best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
		const float tr = best_ms([&] { hipLaunchKernelGGL(k_triad, dim3(blocks), dim3(256), 0, 0, a, b, c, n, 0.5f); });
		printf("blocks %5d: read %.0f GB/s   copy %.0f GB/s (r+w)   triad %.0f GB/s (2r+w)\n", blocks,
			bytes / r / 1e6, 2.0 * bytes / cp / 1e6, 3.0 * bytes / tr / 1e6);
	}
	const float m = best_ms([&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
	printf("hipMemcpyDtoD: %.0f GB/s (r+w)\n", 2.0 * bytes / m / 1e6);
	return 0;
}
