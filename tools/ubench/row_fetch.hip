// ubench/row_fetch.hip — what a wave pays for a dependent chain of row fetches on gfx950, by fetch path.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/row_fetch.hip -o tools/_bin/ubench_fetch && tools/_bin/ubench_fetch   (through gpurun)
// Every wave chases pointers through a table of 64-byte rows (row r holds the index of the next row in dword 0 and
// of a second, unrelated row in dword 1); the table is far larger than any cache below the L2 or small enough to sit
// in it.  Variants:
//   s1      one s_load_dwordx16 per step
//   s2adj   two s_load_dwordx16 per step on the two halves of one 128-byte aligned pair, one s_waitcnt
//   s2far   two s_load_dwordx16 per step on two unrelated rows, one s_waitcnt
//   v1      one global_load_dword per step by lanes 0..15 (64 bytes), v_readlane of the index
//   v2adj   one global_load_dword per step by lanes 0..31 (128 bytes, aligned pair), v_readlane
//   v2far   two global_load_dword per step (two unrelated rows), one s_waitcnt, v_readlane
// Reported: shader cycles per step per wave (s_memtime), and the aggregate rows per microsecond, at 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void chase(const uint32_t *rows, uint32_t n_rows, uint32_t steps,
		unsigned long long *cycles, uint32_t *sink)
{
	const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) / 64u, lane = threadIdx.x & 63u;
	uint32_t cur = __builtin_amdgcn_readfirstlane((wave * 2654435761u) % n_rows);
	uint32_t acc = 0;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (uint32_t i = 0; i < steps; i++) {
		if (VARIANT == 0) { // s1
			uint32_t nxt;
			asm volatile("s_lshl_b32 s52, %1, 6\n s_load_dwordx16 s[20:35], %2, s52\n s_waitcnt lgkmcnt(0)\n s_mov_b32 %0, s20\n"
					: "=s"(nxt) : "s"(cur), "s"(rows) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s52");
			cur = (nxt ^ (i * 2654435761u)) % n_rows;
		} else if (VARIANT == 1) { // s2adj: the pair of rows (cur & ~1), (cur | 1)
			uint32_t nxt;
			asm volatile("s_andn2_b32 s52, %1, 1\n s_lshl_b32 s52, s52, 6\n s_load_dwordx16 s[20:35], %2, s52\n s_load_dwordx16 s[36:51], %2, s52 offset:64\n"
					" s_waitcnt lgkmcnt(0)\n s_xor_b32 %0, s20, s37\n"
					: "=s"(nxt) : "s"(cur), "s"(rows) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35",
					  "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52");
			cur = (nxt ^ (i * 2654435761u)) % n_rows;
		} else if (VARIANT == 2) { // s2far: row cur and the unrelated row (cur * 40503 + 7) % n
			uint32_t nxt; const uint32_t other = (cur * 40503u + 7u) % n_rows;
			asm volatile("s_lshl_b32 s52, %1, 6\n s_lshl_b32 s53, %3, 6\n s_load_dwordx16 s[20:35], %2, s52\n s_load_dwordx16 s[36:51], %2, s53\n"
					" s_waitcnt lgkmcnt(0)\n s_xor_b32 %0, s20, s37\n"
					: "=s"(nxt) : "s"(cur), "s"(rows), "s"(other) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35",
					  "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53");
			cur = (nxt ^ (i * 2654435761u)) % n_rows;
		} else if (VARIANT == 3) { // v1: lanes 0..15 load the row's 16 dwords
			uint32_t v = 0;
			if (lane < 16u) v = rows[(size_t)cur * 16u + lane];
			cur = (__builtin_amdgcn_readlane(v, 0) ^ (i * 2654435761u)) % n_rows; acc += __builtin_amdgcn_readlane(v, 5);
		} else if (VARIANT == 4) { // v2adj: lanes 0..31 load the aligned pair
			uint32_t v = 0;
			if (lane < 32u) v = rows[(size_t)(cur & ~1u) * 16u + lane];
			cur = (__builtin_amdgcn_readlane(v, 0) ^ __builtin_amdgcn_readlane(v, 17) ^ (i * 2654435761u)) % n_rows;
		} else if (VARIANT == 5) { // v2far
			const uint32_t other = (cur * 40503u + 7u) % n_rows;
			uint32_t v = 0, w = 0;
			if (lane < 16u) { v = rows[(size_t)cur * 16u + lane]; w = rows[(size_t)other * 16u + lane]; }
			cur = (__builtin_amdgcn_readlane(v, 0) ^ __builtin_amdgcn_readlane(w, 1) ^ (i * 2654435761u)) % n_rows;
		} else if (VARIANT == 6) { // v1 all lanes the same 64 B (every lane loads dword lane & 15)
			const uint32_t v = rows[(size_t)cur * 16u + (lane & 15u)];
			cur = (__builtin_amdgcn_readlane(v, 0) ^ (i * 2654435761u)) % n_rows;
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	if (lane == 0) { atomicAdd(cycles, t1 - t0); sink[wave & 1023u] = cur + acc; }
}

template <int V>
static void run(const char *name, const uint32_t *d_rows, uint32_t n_rows, uint32_t steps, uint32_t blocks, unsigned long long *d_cyc, uint32_t *d_sink, int lines_per_step)
{
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int rep = 0; rep < 2; rep++) {
		CHECK(hipMemset(d_cyc, 0, 8));
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(chase<V>, dim3(blocks), dim3(256), 0, 0, d_rows, n_rows, steps, d_cyc, d_sink);
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
	}
	float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
	unsigned long long cyc = 0; CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
	const double waves = (double)blocks * 4.0;
	std::printf("%-8s rows %9u  %8.3f ms  cycles/step/wave %7.1f  steps/us (all waves) %9.1f  64B-lines/us %9.1f\n", name, n_rows, ms,
			(double)cyc / waves / steps, waves * steps / (ms * 1e3), waves * steps * lines_per_step / (ms * 1e3));
	std::fflush(stdout);
}

int main(int argc, char **argv)
{
	const uint32_t steps = 2000;
	for (uint32_t blocks : { 64u, 256u, 256u * 2, 256u * 8 }) // a quarter of the CUs with one wave per SIMD ... 8 waves per SIMD on every CU
	for (uint32_t n_rows : { 1u << 14, 1u << 21 }) { // 1 MB (L2 of one XCD: 4 MB), 128 MB
		std::printf("---- %u blocks of 4 waves, table of %u rows ----\n", blocks, n_rows);
		std::vector<uint32_t> h((size_t)n_rows * 16u);
		uint64_t s = 88172645463325252ull;
		for (size_t i = 0; i < h.size(); i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 11) % n_rows; }
		uint32_t *d_rows, *d_sink; unsigned long long *d_cyc;
		CHECK(hipMalloc(&d_rows, h.size() * 4)); CHECK(hipMalloc(&d_sink, 4096)); CHECK(hipMalloc(&d_cyc, 8));
		CHECK(hipMemcpy(d_rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
		run<0>("s1", d_rows, n_rows, steps, blocks, d_cyc, d_sink, 1);
		run<1>("s2adj", d_rows, n_rows, steps, blocks, d_cyc, d_sink, 2);
		run<2>("s2far", d_rows, n_rows, steps, blocks, d_cyc, d_sink, 2);
		run<3>("v1", d_rows, n_rows, steps, blocks, d_cyc, d_sink, 1);
		run<6>("v1all", d_rows, n_rows, steps, blocks, d_cyc, d_sink, 1);
		run<4>("v2adj", d_rows, n_rows, steps, blocks, d_cyc, d_sink, 2);
		run<5>("v2far", d_rows, n_rows, steps, blocks, d_cyc, d_sink, 2);
		CHECK(hipFree(d_rows)); CHECK(hipFree(d_sink)); CHECK(hipFree(d_cyc));
	}
	return 0;
}
