// Microbenchmark: what does a divergent BVH-node fetch cost on gfx950, per wave, as a function of
// where the nodes live (working-set size) and of how the 64 bytes of a node are spread over lanes?
//   NODE4 : every lane loads ITS node as 4 x global_load_dwordx4   (the lane kernel: 64 lines / instruction)
//   QUAD  : 4 neighbouring lanes load the 4 quarters of one node, 4 rounds for 4 nodes (16 lines / instruction)
//   ONE   : every lane loads one random 16-byte piece (64 lines / instruction, 1 instruction)
// CU-cycles per wave-level node step = time x 2.4 GHz x 256 CUs / (waves x steps).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/tcp_rate.hip -o tools/ubench/tcp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 1000

enum { NODE4 = 0, QUAD = 1, ONE = 2 };

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4 *__restrict__ buf, float *out, uint32_t seed, uint32_t mask)
{
	const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
	uint32_t h = tid * 2654435761u + seed;
	float acc = 0.0f;
	for (int i = 0; i < ITER; i++) {
		h = h * 1664525u + 1013904223u;
		const uint32_t rec = (h >> 8) & mask; // this lane's random node
		if (MODE == NODE4) {
			const float4 *n = buf + (size_t)rec * 4u;
			const float4 a = n[0], b = n[1], c = n[2], d = n[3];
			acc += a.x + b.y + c.z + d.w;
		} else if (MODE == QUAD) {
			const uint32_t part = threadIdx.x & 3u;
#pragma unroll
			for (int j = 0; j < 4; j++) { // round j: the quad fetches the node of its lane j
				const uint32_t r = __shfl(rec, (threadIdx.x & ~3u) | j);
				const float4 v = buf[(size_t)r * 4u + part];
				acc += v.x + v.w;
			}
		} else {
			const float4 v = buf[(size_t)rec * 4u + (h & 3u)];
			acc += v.x + v.w;
		}
	}
	out[tid] = acc;
}

template <int MODE>
static void run(const char *name, const float4 *buf, float *out, uint32_t records)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * 8; // 8 waves per SIMD
	float best = 1e9f;
	for (int rep = 0; rep < 3; rep++) {
		(void)hipEventRecord(e0);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, out, 12345u + rep, records - 1u);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
	}
	const double wave_steps = (double)blocks * 4 * ITER;
	printf("  %-5s %8.3f ms  %7.1f CU-cycles per wave step\n", name, best, best * 1e-3 * 2.4e9 * 256 / wave_steps);
}

int main()
{
	float4 *buf; float *out;
	(void)hipMalloc(&buf, (size_t)(1u << 22) * 64); (void)hipMemset(buf, 0, (size_t)(1u << 22) * 64);
	(void)hipMalloc(&out, 256u * 8 * 256 * 4);
	for (uint32_t records : {1u << 7, 1u << 12, 1u << 15, 1u << 18, 1u << 20, 1u << 22}) {
		printf("working set %u nodes = %.2f MB\n", records, records * 64.0 / 1048576.0);
		run<NODE4>("NODE4", buf, out, records);
		run<QUAD>("QUAD", buf, out, records);
		run<ONE>("ONE", buf, out, records);
	}
	return 0;
}
