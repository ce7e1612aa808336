// ubench/issue_cost.hip — what a wave pays per instruction kind inside a VALU-heavy loop on gfx950, at 1 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_cost.hip -o tools/_bin/ubench_issue && tools/_bin/ubench_issue
// Loop body = 40 independent v_fma_f32 (the slab arithmetic of one node step) + 10 copies of a probe:
//   none      nothing
//   salu      s_add_u32
//   br_nt     s_cmp + s_cbranch never taken
//   br_t      s_cmp + s_cbranch always taken (over one s_nop)
//   br_back   s_branch to a label 64 instructions away and back (two taken branches, another cache line)
//   cmp_salu  v_cmp_e64 -> s_and_b64 reading it (vector -> scalar dependency)
//   cmp_br    v_cmp_e64 -> s_cbranch_vccz never taken
//   rfl       v_readfirstlane -> s_add reading it
// Reported: cycles per loop iteration per wave, and per probe (minus `none`, / 10).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

#define FMA4 "v_fma_f32 v40, v40, %[a], %[b]\n v_fma_f32 v41, v41, %[a], %[b]\n v_fma_f32 v42, v42, %[a], %[b]\n v_fma_f32 v43, v43, %[a], %[b]\n"
#define R10(X) X X X X X X X X X X

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void issue(uint32_t iters, unsigned long long *cycles, float *sink)
{
	__shared__ uint32_t lds[256];
	const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&lds[threadIdx.x];
	float a = 1.0001f + threadIdx.x * 1e-7f, b = 0.5f;
	uint32_t zero = __builtin_amdgcn_readfirstlane(iters >> 31);
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (uint32_t i = 0; i < iters; i++) {
#define BODY(PROBE) asm volatile(R10(FMA4 PROBE) : : [a] "v"(a), [b] "v"(b), [z] "s"(zero), [l] "v"(lds_addr) : "v40", "v41", "v42", "v43", "v44", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "vcc", "scc")
		if (V == 0) BODY("");
		if (V == 1) BODY("s_add_u32 s40, s40, 1\n");
		if (V == 24) asm volatile("s_mov_b32 s41, 0\n" ::: "s41");
		if (V == 2) BODY("s_cmp_eq_u32 %[z], 1\n s_cbranch_scc1 1\n s_nop 0\n");
		if (V == 3) BODY("s_cmp_eq_u32 %[z], 0\n s_cbranch_scc1 1\n s_nop 0\n");
		if (V == 4) BODY("v_cmp_lt_f32_e64 s[40:41], v40, %[b]\n s_and_b64 s[42:43], s[40:41], exec\n");
		if (V == 5) BODY("v_cmp_lt_f32_e32 vcc, 0x7f800000, v40\n s_cbranch_vccnz 1\n s_nop 0\n");
		if (V == 6) BODY("v_readfirstlane_b32 s40, v41\n s_add_u32 s41, s40, 1\n");
		if (V == 7) BODY("s_cmp_eq_u32 %[z], 0\n s_cbranch_scc1 5\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");
		if (V == 9) BODY("v_cmp_lt_f32_e64 s[40:41], v40, %[b]\n v_cmp_lt_f32_e64 s[42:43], v41, %[b]\n v_cmp_lt_f32_e64 s[44:45], v42, %[b]\n v_cmp_lt_f32_e64 s[46:47], v43, %[b]\n"
				" s_and_b64 s[40:41], s[40:41], exec\n s_and_b64 s[42:43], s[42:43], exec\n s_and_b64 s[44:45], s[44:45], exec\n s_and_b64 s[46:47], s[46:47], exec\n");
		if (V == 10) BODY("v_readfirstlane_b32 s40, v40\n v_readfirstlane_b32 s41, v41\n v_readfirstlane_b32 s42, v42\n v_readfirstlane_b32 s43, v43\n"
				" s_add_u32 s44, s40, 1\n s_add_u32 s45, s41, 1\n s_add_u32 s46, s42, 1\n s_add_u32 s47, s43, 1\n");
		if (V == 11) BODY("s_mov_b64 exec, 0\n ds_write_b32 %[l], v40\n s_mov_b64 exec, -1\n");
		if (V == 12) BODY("ds_write_b32 %[l], v40\n");
		if (V == 13) BODY("v_cmp_lt_f32_e64 s[40:41], v40, %[b]\n");
		if (V == 14) BODY("v_cmp_lt_f32_e64 s[40:41], v40, %[b]\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_and_b64 s[42:43], s[40:41], exec\n");
		if (V == 15) BODY("v_cmp_lt_f32_e32 vcc, 0x7f800000, v40\n v_cndmask_b32 v44, v44, v40, vcc\n");
		if (V == 16) BODY("s_cselect_b64 s[40:41], s[42:43], s[44:45]\n s_cselect_b32 s46, s47, s40\n s_cmp_lg_u64 s[40:41], 0\n s_cselect_b64 s[42:43], s[40:41], s[44:45]\n");
		if (V == 17) BODY("s_cmp_eq_u32 %[z], 1\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_cbranch_scc1 1\n s_nop 0\n");
		if (V == 20) BODY("s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n");
		if (V == 21) BODY("s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1\n");
		if (V == 22) BODY("s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1\n s_add_u32 s44, s44, 1\n s_add_u32 s45, s45, 1\n s_add_u32 s46, s46, 1\n s_add_u32 s47, s47, 1\n");
		if (V == 23) BODY("s_add_u32 s40, s40, 1\n s_add_u32 s40, s40, 1\n s_add_u32 s40, s40, 1\n s_add_u32 s40, s40, 1\n");
		if (V == 24) BODY("s_mov_b32 m0, s41\n s_nop 0\n s_movrels_b32 s40, s42\n s_movrels_b64 s[44:45], s[42:43]\n");
		if (V == 25) BODY("s_or_b64 s[40:41], s[42:43], s[44:45]\n s_addc_u32 s46, s46, s46\n s_or_b64 s[40:41], s[42:43], s[44:45]\n s_addc_u32 s46, s46, s46\n");
		if (V == 8) BODY("s_cmp_eq_u32 %[z], 0\n s_cbranch_scc1 17\n" R10("s_nop 0\n") "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n");
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if ((threadIdx.x & 63u) == 0) atomicAdd(cycles, t1 - t0);
	float r; asm volatile("v_add_f32 %0, v40, v41\n" : "=v"(r));
	sink[(blockIdx.x * 256u + threadIdx.x) & 1023u] = r;
}

template <int V>
static double run(const char *name, uint32_t blocks, unsigned long long *d_cyc, float *d_sink, double base)
{
	const uint32_t iters = 2000;
	for (int rep = 0; rep < 2; rep++) {
		CHECK(hipMemset(d_cyc, 0, 8));
		hipLaunchKernelGGL(issue<V>, dim3(blocks), dim3(256), 0, 0, iters, d_cyc, d_sink);
		CHECK(hipDeviceSynchronize());
	}
	unsigned long long cyc = 0; CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
	const double per = (double)cyc / (blocks * 4.0) / iters;
	std::printf("%-9s cycles/iteration/wave %8.1f   per probe %6.1f\n", name, per, (per - base) / 10.0);
	std::fflush(stdout);
	return per;
}

int main()
{
	unsigned long long *d_cyc; float *d_sink;
	CHECK(hipMalloc(&d_cyc, 8)); CHECK(hipMalloc(&d_sink, 4096));
	for (uint32_t blocks : { 256u, 256u * 8 }) {
		std::printf("---- %u waves per SIMD (40 v_fma per iteration: %u waves x 80 cycles of vector issue) ----\n", blocks / 256, blocks / 256);
		const double base = run<0>("none", blocks, d_cyc, d_sink, 0.0);
		run<1>("salu", blocks, d_cyc, d_sink, base);
		run<2>("br_nt", blocks, d_cyc, d_sink, base);
		run<3>("br_t1", blocks, d_cyc, d_sink, base);
		run<7>("br_t5", blocks, d_cyc, d_sink, base);
		run<8>("br_t17", blocks, d_cyc, d_sink, base);
		run<4>("cmp_salu", blocks, d_cyc, d_sink, base);
		run<5>("cmp_br", blocks, d_cyc, d_sink, base);
		run<6>("rfl", blocks, d_cyc, d_sink, base);
		run<20>("salu2", blocks, d_cyc, d_sink, base);
		run<21>("salu4", blocks, d_cyc, d_sink, base);
		run<22>("salu8", blocks, d_cyc, d_sink, base);
		run<23>("salu4dep", blocks, d_cyc, d_sink, base);
		run<24>("movrels", blocks, d_cyc, d_sink, base);
		run<25>("or_addc", blocks, d_cyc, d_sink, base);
		run<9>("cmp4+and4", blocks, d_cyc, d_sink, base);
		run<10>("rfl4+add4", blocks, d_cyc, d_sink, base);
		run<11>("ds_exec0", blocks, d_cyc, d_sink, base);
		run<12>("ds_write", blocks, d_cyc, d_sink, base);
		run<13>("cmp_only", blocks, d_cyc, d_sink, base);
		run<14>("cmp4nop", blocks, d_cyc, d_sink, base);
		run<15>("cmp_cnd", blocks, d_cyc, d_sink, base);
		run<16>("csel4", blocks, d_cyc, d_sink, base);
		run<17>("br_nt_far", blocks, d_cyc, d_sink, base);
	}
	return 0;
}
