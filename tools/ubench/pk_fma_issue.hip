// ubench/pk_fma_issue.hip — issue cost of v_pk_fma_f32 against v_fma_f32 on gfx950 (8 waves per SIMD, every CU).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define R10(X) X X X X X X X X X X
template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k(uint32_t iters, unsigned long long *cycles, float *sink, float sa, float sb)
{
	float a = 1.0001f + threadIdx.x * 1e-7f, b = 0.5f;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (uint32_t i = 0; i < iters; i++) {
		// 40 scalar fmas with an SGPR operand (the slab form: plane in an SGPR)
		if (V == 0) asm volatile(R10("v_fma_f32 v40, %[sa], %[a], %[b]\n v_fma_f32 v41, %[sb], %[a], %[b]\n v_fma_f32 v42, %[sa], %[a], %[b]\n v_fma_f32 v43, %[sb], %[a], %[b]\n")
				: : [a] "v"(a), [b] "v"(b), [sa] "s"(sa), [sb] "s"(sb) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
		// 20 packed fmas = the same 40 fmas: src0 an SGPR pair, src1 / src2 VGPR pairs read with op_sel (low half twice)
		if (V == 1) asm volatile(R10("v_pk_fma_f32 v[40:41], s[40:41], v[44:45], v[46:47] op_sel_hi:[1,0,0]\n v_pk_fma_f32 v[42:43], s[42:43], v[44:45], v[46:47] op_sel_hi:[1,0,0]\n")
				: : [a] "v"(a), [b] "v"(b) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s40", "s41", "s42", "s43");
		// 20 packed fmas, all VGPR
		if (V == 2) asm volatile(R10("v_pk_fma_f32 v[40:41], v[44:45], v[46:47], v[46:47]\n v_pk_fma_f32 v[42:43], v[44:45], v[46:47], v[44:45]\n")
				: : [a] "v"(a), [b] "v"(b) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
		// the op_sel patterns of the slab test: high halves broadcast
		if (V == 6) asm volatile(R10("v_pk_fma_f32 v[40:41], s[40:41], v[44:45], v[46:47] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[42:43], s[42:43], v[44:45], v[46:47] op_sel:[0,1,0] op_sel_hi:[1,1,0]\n")
				: : [a] "v"(a), [b] "v"(b) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s40", "s41", "s42", "s43");
		// a slab test as packet_quad_kernel.h writes it, x 5 (40 instructions, 15 of them packed)
		if (V == 7) asm volatile(R10("v_pk_fma_f32 v[50:51], s[40:41], v[44:45], v[46:47] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n v_pk_fma_f32 v[52:53], s[42:43], v[44:45], v[48:49] op_sel:[0,1,0] op_sel_hi:[1,1,0]\n"
				" v_pk_fma_f32 v[54:55], s[40:41], v[46:47], v[48:49] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n v_max_f32 v56, v54, %[a]\n")
				: : [a] "v"(a), [b] "v"(b) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "s40", "s41", "s42", "s43");
		// 40 v_max3 / v_min3
		if (V == 3) asm volatile(R10("v_max3_f32 v40, %[a], %[b], v44\n v_min3_f32 v41, %[a], %[b], v45\n v_max3_f32 v42, %[a], %[b], v46\n v_min3_f32 v43, %[a], %[b], v47\n")
				: : [a] "v"(a), [b] "v"(b) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
		// 40 v_cmp_le_f32_e64 into SGPR pairs
		if (V == 4) asm volatile(R10("v_cmp_le_f32_e64 s[40:41], %[a], %[b]\n v_cmp_le_f32_e64 s[42:43], %[a], %[b]\n v_cmp_le_f32_e64 s[44:45], %[b], %[a]\n v_cmp_le_f32_e64 s[46:47], %[b], %[a]\n")
				: : [a] "v"(a), [b] "v"(b) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
		// 40 v_max_f32 (VOP2)
		if (V == 5) asm volatile(R10("v_max_f32 v40, %[a], %[b]\n v_min_f32 v41, %[a], %[b]\n v_max_f32 v42, %[b], %[a]\n v_min_f32 v43, %[b], %[a]\n")
				: : [a] "v"(a), [b] "v"(b) : "v40", "v41", "v42", "v43");
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if ((threadIdx.x & 63u) == 0) atomicAdd(cycles, t1 - t0);
	float r; asm volatile("v_add_f32 %0, v40, v41\n" : "=v"(r));
	sink[(blockIdx.x * 256u + threadIdx.x) & 1023u] = r;
}
template <int V> static void run(const char *name, uint32_t blocks, unsigned long long *d_cyc, float *d_sink)
{
	const uint32_t iters = 2000;
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int rep = 0; rep < 2; rep++) {
		CHECK(hipMemset(d_cyc, 0, 8));
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, iters, d_cyc, d_sink, 1.5f, 2.5f);
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
	}
	float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
	unsigned long long cyc = 0; CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
	std::printf("%-12s ticks/iteration/wave %8.1f   kernel %.3f ms  (= %.2f ns per iteration per SIMD-round)\n", name, (double)cyc / (blocks * 4.0) / iters, ms, ms * 1e6 / iters / (blocks / 256.0 / 8.0 > 1 ? blocks / 256.0 / 8.0 : 1));
	std::fflush(stdout);
}
int main()
{
	unsigned long long *d_cyc; float *d_sink;
	CHECK(hipMalloc(&d_cyc, 8)); CHECK(hipMalloc(&d_sink, 4096));
	for (uint32_t blocks : { 256u, 256u * 8 }) {
		std::printf("---- %u waves per SIMD ----\n", blocks / 256);
		run<0>("fma40_sgpr", blocks, d_cyc, d_sink);
		run<1>("pk20_sgpr", blocks, d_cyc, d_sink);
		run<2>("pk20_vgpr", blocks, d_cyc, d_sink);
		run<6>("pk20_opsel", blocks, d_cyc, d_sink);
		run<7>("pk30+max10", blocks, d_cyc, d_sink);
		run<3>("minmax3_40", blocks, d_cyc, d_sink);
		run<4>("cmp64_40", blocks, d_cyc, d_sink);
		run<5>("minmax_40", blocks, d_cyc, d_sink);
	}
	return 0;
}
