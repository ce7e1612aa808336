// Microbenchmark: scalar-ALU issue rate per CU on gfx950 as a function of waves per CU.
// Each wave runs ITER x 16 independent s_add/s_xor-style scalar instructions.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/salu_rate.hip -o tools/ubench/salu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 20000
__global__ void k(unsigned *out, unsigned seed)
{
	unsigned a = seed, b = seed + 1, c = seed + 2, d = seed + 3, e = seed + 4, f = seed + 5, g = seed + 6, h = seed + 7;
	for (int i = 0; i < ITER; i++) {
		asm volatile(
			"s_add_u32 %0, %0, %8\n s_add_u32 %1, %1, %8\n s_add_u32 %2, %2, %8\n s_add_u32 %3, %3, %8\n"
			"s_add_u32 %4, %4, %8\n s_add_u32 %5, %5, %8\n s_add_u32 %6, %6, %8\n s_add_u32 %7, %7, %8\n"
			"s_xor_b32 %0, %0, %1\n s_xor_b32 %2, %2, %3\n s_xor_b32 %4, %4, %5\n s_xor_b32 %6, %6, %7\n"
			"s_add_u32 %1, %1, %8\n s_add_u32 %3, %3, %8\n s_add_u32 %5, %5, %8\n s_add_u32 %7, %7, %8\n"
			: "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h) : "s"(seed) : "scc");
	}
	if (threadIdx.x == 0) out[blockIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
int main()
{
	unsigned *d; hipMalloc(&d, 1 << 20);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	int dev_cus = 256;
	for (int waves_per_cu : {1, 2, 4, 8, 16, 32}) {
		// one wave per block; blocks = CUs * waves_per_cu (all resident at once)
		const int blocks = dev_cus * waves_per_cu;
		float best = 1e9;
		for (int rep = 0; rep < 3; rep++) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d, 12345u);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
		}
		const double instr = (double)blocks * ITER * 16;
		printf("waves/CU %2d: %.3f ms  -> %.2f SALU instr per ns chip-wide = %.3f per CU per cycle @2.4GHz\n",
			waves_per_cu, best, instr / best / 1e6, instr / best / 1e6 / 256 / 2.4);
	}
	return 0;
}
