// packet_asm_kernel.h — the packet kernel with a hand-written gfx950 node loop.
// Included by kernels.hip (inside namespace mrt, after packet_kernel.h).
//
// Why: the scalar ALU is ONE unit per CU, ~1 instruction per cycle for all 32 waves
// (tools/ubench/salu_rate.hip: 0.97 per CU per cycle from 4 waves up).  The compiler's
// packet loop spends ~30 scalar instructions per node step (structurised control flow,
// popcount votes, 64-bit address arithmetic): 5.2 k per wave, 5.4 M per CU per launch,
// i.e. 2.3 ms of a 3.3 ms launch at C3 — the walk was scalar-ALU bound, which is also
// why the 4-wide and two-packets-per-wave variants, both heavier on scalar work, lost.
//
// This loop keeps scalar work to ~5 instructions per step:
//   * the two child masks come straight out of v_cmp (VCC and an SGPR pair) and are
//     consumed by s_cbranch_vccz / one s_cmp_eq_u64 — no s_cselect/s_and chains;
//   * near/far order is lane 0's `tl < tr` (v_cmp + s_bitcmp1 + 2 s_cselect) instead
//     of two popcount votes: order only affects speed, never results;
//   * the per-wave stack is addressed through a VGPR (uniform LDS byte address), with a
//     sentinel at the bottom, so push/pop cost no scalar instruction and there is no
//     empty-stack test; an entry also carries the mask of the lanes whose own ray hit the
//     pushed child's box (needed when it is a leaf: see packet_kernel.h);
//   * one s_load_dwordx16 with an SGPR offset fetches the 64-byte node;
//   * a pushed inner node is prefetched by a vector load (its own counter, vmcnt: a scalar-load prefetch
//     would sit in front of the next node's s_waitcnt lgkmcnt(0)); the data is dropped, the line is in the
//     L2 when the node is popped (1.7 % at C3 and C5).
// The slab test is the octant-specialised one of packet_kernel.h: 12 v_fma + 8
// v_max/v_min(3) + 2 v_cmp, bit-identical values.  Leaves (triangle tests) stay in C++.
// Packets whose rays do not share one octant, and counting builds, use packet_traverse.
#pragma once

// node registers after the s_load_dwordx16 into s[36:51]:
//   s36..s38 lmin.xyz  s39 left_ref | s40..s42 lmax.xyz  s43 right_ref
//   s44..s46 rmin.xyz  s47 -        | s48..s50 rmax.xyz  s51 -
// scratch: s52 byte offset, s53 far ref, s[54:55] right-child mask, s[56:57] "left is nearer" flags,
// s[58:59] mask of the pushed (far) child, s[60:61] mask of the entered child, v50..v55 slab values (one box
// at a time: six temporaries), v58 the never-read destination of the far-child prefetch.
//
// Stack entries are 16 bytes: {ref, -, lane mask (64 bit)}.  The mask holds the lanes whose OWN ray hit the
// box of the pushed child; it matters only when that child is a leaf (packet_kernel.h: a lane accepts
// triangle hits only in leaves its own ray entered).  The block ends with `node` a leaf or the sentinel and
// the leaf's mask at LDS [sp + 8]: left there by the pop, or written on the way out when the leaf was
// entered straight from its parent.  Inner steps pay nothing for this; a push pays one s_cselect_b64, two
// v_mov and a ds_write_b64.
// The rows of both children towards the scalar cache while the 22 vector instructions of the two slab tests run: the fetch of
// whichever is entered next then waits less (a leaf child's ref indexes triangles, not node rows: skipped).  Nothing waits for the
// two dwords; s62 / s63 are never read.
// It pays where a launch is about one round of waves and a walk's time is its chain of fetches (small grids, grids cast in
// pieces: C3 scene 16x12 0.149 -> 0.138 ms, 128^2 0.183 -> 0.164, 384^2 0.246 -> 0.231, 640x360 0.232 -> 0.221, 1024x576 0.275 ->
// 0.258) and costs where the chip is full and the scalar unit is what walks share (1280x720 0.274 -> 0.282, 1920x1080 0.447 ->
// 0.469, 4096^2 2.16 -> 2.29): a template parameter of the kernel, chosen by the launch (launch_trace).
#define MRT_ASM_KPREFETCH_ON                                                                                    \
		"s_bitcmp1_b32 s39, 31\n"                                                                           \
		"s_cbranch_scc1 L_kl_%=\n"                                                                          \
		"s_lshl_b32 s62, s39, 6\n"                                                                          \
		"s_load_dword s62, %[base], s62\n"                                                                  \
		"L_kl_%=:\n"                                                                                        \
		"s_bitcmp1_b32 s43, 31\n"                                                                           \
		"s_cbranch_scc1 L_kr_%=\n"                                                                          \
		"s_lshl_b32 s63, s43, 6\n"                                                                          \
		"s_load_dword s63, %[base], s63\n"                                                                  \
		"L_kr_%=:\n"
#define MRT_ASM_NODE_LOOP(CNT, KPF, LNX, LFX, LNY, LFY, LNZ, LFZ, RNX, RFX, RNY, RFY, RNZ, RFZ)                      \
	asm volatile(                                                                                           \
		"s_cmp_eq_u32 %[dopop], 1\n"                                                                        \
		"s_cbranch_scc1 L_pop_%=\n"                                                                         \
		"s_branch L_check_%=\n"                                                                             \
		"L_node_%=:\n"                                                                                      \
		CNT                                 /* counting build: one more node step of this packet */         \
		"s_lshl_b32 s52, %[node], 6\n"                                                                      \
		"s_load_dwordx16 s[36:51], %[base], s52\n"                                                          \
		"s_waitcnt lgkmcnt(0)\n"                                                                            \
		KPF                                                                                                 \
		"v_fma_f32 v50, " LNX ", %[ix], %[nrx]\n"                                                           \
		"v_fma_f32 v51, " LNY ", %[iy], %[nry]\n"                                                           \
		"v_fma_f32 v52, " LNZ ", %[iz], %[nrz]\n"                                                           \
		"v_max_f32 v52, v52, %[tmin]\n"                                                                     \
		"v_max3_f32 v50, v50, v51, v52\n"   /* v50 = tl  = entry of the left box, clamped to t_min */        \
		"v_fma_f32 v51, " LFX ", %[ix], %[nrx]\n"                                                           \
		"v_fma_f32 v52, " LFY ", %[iy], %[nry]\n"                                                           \
		"v_fma_f32 v53, " LFZ ", %[iz], %[nrz]\n"                                                           \
		"v_min_f32 v53, v53, %[lim]\n"                                                                      \
		"v_min3_f32 v51, v51, v52, v53\n"   /* v51 = tlx = exit of the left box, clamped to best_t */        \
		"v_fma_f32 v52, " RNX ", %[ix], %[nrx]\n"                                                           \
		"v_fma_f32 v53, " RNY ", %[iy], %[nry]\n"                                                           \
		"v_fma_f32 v54, " RNZ ", %[iz], %[nrz]\n"                                                           \
		"v_max_f32 v54, v54, %[tmin]\n"                                                                     \
		"v_max3_f32 v52, v52, v53, v54\n"   /* v52 = tr  */                                                  \
		"v_fma_f32 v53, " RFX ", %[ix], %[nrx]\n"                                                           \
		"v_fma_f32 v54, " RFY ", %[iy], %[nry]\n"                                                           \
		"v_fma_f32 v55, " RFZ ", %[iz], %[nrz]\n"                                                           \
		"v_min_f32 v55, v55, %[lim]\n"                                                                      \
		"v_min3_f32 v53, v53, v54, v55\n"   /* v53 = trx */                                                  \
		"v_cmp_le_f32 vcc, v50, v51\n"      /* lanes that hit the left child  */                         \
		"v_cmp_le_f32_e64 s[54:55], v52, v53\n" /* lanes that hit the right child */                         \
		"s_cbranch_vccz L_lmiss_%=\n"                                                                       \
		"s_cmp_eq_u64 s[54:55], 0\n"                                                                        \
		"s_cbranch_scc1 L_onlyl_%=\n"                                                                       \
		"v_cmp_lt_f32_e64 s[56:57], v50, v52\n" /* both hit: lane 0 decides which is nearer */              \
		"s_bitcmp1_b32 s56, 0\n"                                                                            \
		"s_cselect_b32 s53, s43, s39\n"     /* far  */                                                   \
		"s_cselect_b32 %[node], s39, s43\n" /* near */                                                   \
		"s_cselect_b64 s[58:59], s[54:55], vcc\n" /* lanes that hit the far child  */                     \
		"s_cselect_b64 s[60:61], vcc, s[54:55]\n" /* lanes that hit the near child */                     \
		"v_mov_b32 v51, s53\n"                                                                              \
		"v_mov_b32 v52, s58\n"                                                                              \
		"v_mov_b32 v53, s59\n"                                                                              \
		"ds_write_b32 %[sp], v51\n"                                                                         \
		"ds_write_b64 %[sp], v[52:53] offset:8\n"                                                           \
		"v_add_u32 %[sp], 16, %[sp]\n"                                                                      \
		"s_bitcmp1_b32 s53, 31\n"           /* the pushed child is an inner node: start pulling it towards */ \
		"s_cbranch_scc1 L_entered_%=\n"     /* the L2 now; it is popped after the near subtree is done     */ \
		"v_lshlrev_b32 v54, 6, v51\n"                                                                       \
		"global_load_dword v58, v54, %[base]\n" /* v58 is never read; vmcnt is drained at the exit         */ \
		"s_branch L_entered_%=\n"                                                                           \
		"L_onlyl_%=:\n"                                                                                     \
		"s_mov_b32 %[node], s39\n"                                                                          \
		"s_mov_b64 s[60:61], vcc\n"                                                                         \
		"s_branch L_entered_%=\n"                                                                           \
		"L_lmiss_%=:\n"                                                                                     \
		"s_cmp_eq_u64 s[54:55], 0\n"                                                                        \
		"s_cbranch_scc1 L_pop_%=\n"                                                                         \
		"s_mov_b32 %[node], s43\n"                                                                          \
		"s_mov_b64 s[60:61], s[54:55]\n"                                                                    \
		"L_entered_%=:\n"                   /* a child entered straight from its parent */               \
		"s_cmp_lt_u32 %[node], 0x7fffffff\n"                                                                \
		"s_cbranch_scc1 L_node_%=\n"                                                                        \
		"v_mov_b32 v52, s60\n"              /* a leaf: its lane mask goes where a pop would leave it */  \
		"v_mov_b32 v53, s61\n"                                                                              \
		"ds_write_b64 %[sp], v[52:53] offset:8\n"                                                           \
		"s_branch L_exit_%=\n"                                                                              \
		"L_pop_%=:\n"                                                                                       \
		"v_add_u32 %[sp], -16, %[sp]\n"                                                                     \
		"ds_read_b32 v51, %[sp]\n"                                                                          \
		"s_waitcnt lgkmcnt(0)\n"                                                                            \
		"v_readfirstlane_b32 %[node], v51\n"                                                                \
		"L_check_%=:\n"                                                                                     \
		"s_cmp_lt_u32 %[node], 0x7fffffff\n"                                                                \
		"s_cbranch_scc1 L_node_%=\n"                                                                        \
		"L_exit_%=:\n"                                                                                      \
		"s_waitcnt vmcnt(0) lgkmcnt(0)\n"   /* no prefetch may land in v58 / s62 / s63 once the compiler owns them again */ \
		: [node] "+s"(node), [sp] "+v"(sp), [cnt] "+s"(cnt)                                                 \
		: [base] "s"(base), [dopop] "s"(dopop), [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [nrx] "v"(nrx),   \
		  [nry] "v"(nry), [nrz] "v"(nrz), [tmin] "v"(tmin), [lim] "v"(lim)                                  \
		: "vcc", "scc", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45",      \
		  "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", \
		  "v50", "v51", "v52", "v53", "v54", "v55", "v58")

// Walks inner nodes until `node` is a leaf reference (>= 0x80000000) or the sentinel
// (0x7FFFFFFF = the stack ran empty).  dopop = 1: start by popping (after a leaf).
#define MRT_ASM_COUNT_STEP "s_add_u32 %[cnt], %[cnt], 1\n"
template <int OCT, bool COUNT, bool KPF = false>
__device__ __forceinline__ void packet_node_loop_asm(const DevNode *base, uint32_t &node, uint32_t &sp, uint32_t dopop, uint32_t &cnt,
		float ix, float iy, float iz, float nrx, float nry, float nrz, float tmin, float lim)
{
	// per axis: inv >= 0 -> near plane = min, far plane = max; inv < 0 -> swapped
#define MRT_ASM_OCT(O, ...) \
	if (OCT == O) { if (COUNT) MRT_ASM_NODE_LOOP(MRT_ASM_COUNT_STEP, "", __VA_ARGS__); else if (KPF) MRT_ASM_NODE_LOOP("", MRT_ASM_KPREFETCH_ON, __VA_ARGS__); else MRT_ASM_NODE_LOOP("", "", __VA_ARGS__); }
	MRT_ASM_OCT(0, "s36", "s40", "s37", "s41", "s38", "s42", "s44", "s48", "s45", "s49", "s46", "s50")
	MRT_ASM_OCT(1, "s40", "s36", "s37", "s41", "s38", "s42", "s48", "s44", "s45", "s49", "s46", "s50")
	MRT_ASM_OCT(2, "s36", "s40", "s41", "s37", "s38", "s42", "s44", "s48", "s49", "s45", "s46", "s50")
	MRT_ASM_OCT(3, "s40", "s36", "s41", "s37", "s38", "s42", "s48", "s44", "s49", "s45", "s46", "s50")
	MRT_ASM_OCT(4, "s36", "s40", "s37", "s41", "s42", "s38", "s44", "s48", "s45", "s49", "s50", "s46")
	MRT_ASM_OCT(5, "s40", "s36", "s37", "s41", "s42", "s38", "s48", "s44", "s45", "s49", "s50", "s46")
	MRT_ASM_OCT(6, "s36", "s40", "s41", "s37", "s42", "s38", "s44", "s48", "s49", "s45", "s50", "s46")
	MRT_ASM_OCT(7, "s40", "s36", "s41", "s37", "s42", "s38", "s48", "s44", "s49", "s45", "s50", "s46")
#undef MRT_ASM_OCT
}

// A leaf of a packet walk: the lanes whose own ray hit its box (`own`) test its triangles (glsl:166-192); the
// others ride along with an empty interval.  Triangle rows come through the scalar cache (uniform address).
// lim_t: this lane's far limit in box tests (best_t, or -FLT_MAX once it takes no part); updated on a hit.
template <bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void packet_leaf(const TraceParams &p, const float4 *hot, const RayRegs &r, uint32_t first_slot, bool own,
		float &lim_t, float &best_t, float &best_u, float &best_v, uint32_t &best_slot, uint32_t &best_id, uint32_t id_base, uint32_t &n_tris)
{
	float lim_leaf = own ? lim_t : -FLT_MAX;
	uint32_t slot = first_slot;
	bool last;
	do {
		const float4 *t3 = hot + (size_t)slot * 3u; // uniform address
		const float4 q0 = t3[0], q1 = t3[1], q2 = t3[2];
		last = (__float_as_uint(q2.w) & kLastInLeaf) != 0u;
		if (COUNT) n_tris++;
		if ((__float_as_uint(q1.w) & p.query_mask) != 0u) {
			// ray_triangle, glsl:105-131 == Triangle::intersect, src/core/triangle.h:56-105
			const float pvx = fma_(r.dy, q2.z, -(r.dz * q2.y));
			const float pvy = fma_(r.dz, q2.x, -(r.dx * q2.z));
			const float pvz = fma_(r.dx, q2.y, -(r.dy * q2.x));
			const float det = dot3(q1.x, q1.y, q1.z, pvx, pvy, pvz);
			if (!(__builtin_fabsf(det) < 1e-8f)) {
				const float inv_det = 1.0f / det;
				const float tvx = r.ox - q0.x, tvy = r.oy - q0.y, tvz = r.oz - q0.z;
				const float u = dot3(tvx, tvy, tvz, pvx, pvy, pvz) * inv_det;
				if (!(u < 0.0f || u > 1.0f)) {
					const float qvx = fma_(tvy, q1.z, -(tvz * q1.y));
					const float qvy = fma_(tvz, q1.x, -(tvx * q1.z));
					const float qvz = fma_(tvx, q1.y, -(tvy * q1.x));
					const float v = dot3(r.dx, r.dy, r.dz, qvx, qvy, qvz) * inv_det;
					if (!(v < 0.0f || u + v > 1.0f)) {
						const float t = dot3(q2.x, q2.y, q2.z, qvx, qvy, qvz) * inv_det;
						const uint32_t id = id_base + __float_as_uint(q0.w);
						if (!(t < r.t_min) && (t < lim_leaf || (t == lim_leaf && best_slot != 0xFFFFFFFFu && id < best_id))) {
							best_t = t; best_u = u; best_v = v; best_slot = slot; best_id = id;
							lim_t = lim_leaf = ANY_HIT ? -FLT_MAX : t;
						}
					}
				}
			}
		}
		slot++;
	} while (!last);
}

template <int OCT, bool ANY_HIT, bool COUNT = false, bool KPF = false>
__device__ __forceinline__ void packet_traverse_asm(const TraceParams &p, const RayRegs &r, uint32_t sp,
		float &best_t, float &best_u, float &best_v, uint32_t &best_slot,
		uint32_t &n_nodes, uint32_t &n_tris, // COUNT: wave-uniform numbers of node steps and triangle rows fetched
		// a BLAS of a two-level scene (two_level_kernel.h): its root, the instance's flat id base, the id of the
		// best hit so far (in / out), and whether this lane sits the walk out (its world ray missed the instance)
		uint32_t root = 0u, uint32_t id_base = 0u, uint32_t *best_id_io = nullptr, bool dead = false)
{
	const bool degenerate = r.t_min >= r.t_max;
	float lim_t = (degenerate || dead) ? -FLT_MAX : best_t;
	const float ix = safe_inv(r.dx), iy = safe_inv(r.dy), iz = safe_inv(r.dz);
	const float nrx = -(r.ox * ix), nry = -(r.oy * iy), nrz = -(r.oz * iz);
	const float4 *hot = reinterpret_cast<const float4 *>(p.tri_hot);
	uint32_t best_id = best_id_io ? *best_id_io : 0xFFFFFFFFu;
	uint32_t cur = root; // always a wide node
	uint32_t dopop = 0;
	uint32_t steps = 0;  // COUNT: node steps of this walk (an SGPR inside the asm block)
	for (;;) {
		packet_node_loop_asm<OCT, COUNT, KPF>(p.nodes, cur, sp, dopop, steps, ix, iy, iz, nrx, nry, nrz, r.t_min, lim_t);
		cur = __builtin_amdgcn_readfirstlane(cur);
		steps = __builtin_amdgcn_readfirstlane(steps); // (tells the compiler the asm operand stays wave-uniform)
		if (cur == kSentinel) break;
		// leaf: the lanes whose own ray hit its box test its triangles; their mask lies at [sp + 8] (volatile: written
		// by the asm block, which the compiler does not know to write memory)
		const unsigned long long own_mask =
				*(volatile __attribute__((address_space(3))) unsigned long long *)(uintptr_t)(sp + 8u);
		const bool own = ((own_mask >> (threadIdx.x & (MRT_WAVE - 1))) & 1ull) != 0ull;
		packet_leaf<ANY_HIT, COUNT>(p, hot, r, cur & 0x7FFFFFFFu, own, lim_t, best_t, best_u, best_v, best_slot, best_id, id_base, n_tris);
		if (ANY_HIT && __ballot(lim_t != -FLT_MAX) == 0ull) break;
		dopop = 1;
	}
	if (COUNT) n_nodes += steps;
	if (best_id_io) *best_id_io = best_id;
}

template <bool ANY_HIT, bool COUNT = false, bool KPF = false>
__global__ __launch_bounds__(MRT_WG) __attribute__((amdgpu_num_sgpr(80))) void trace_packet_asm_kernel(const TraceParams p) // (80 scalar registers: 8 waves per SIMD; 81-96 would be 7)
{
	// 16-byte stack entries {ref, -, lane mask}; entry 0 holds the sentinel
	__shared__ __attribute__((aligned(16))) uint32_t wave_stack[MRT_WG / MRT_WAVE][(MRT_PACKET_STACK + 1) * 4];
	if (skip_launch(p)) return;
	uint32_t block = blockIdx.x;
	if (p.xcd_swizzle) {
		const uint32_t per = gridDim.x >> 3;
		if (block < (per << 3)) block = (block & 7u) * per + (block >> 3);
	}
	if (p.tile_cost != nullptr && (threadIdx.x & (MRT_WAVE - 1)) == 0u) note_tile_start(p, ((uint64_t)block * MRT_WG + threadIdx.x) & ~63ull);
	uint64_t ray_idx = 0; uint32_t px = 0, py = 0;
	if (!lane_ray_index(p, block, ray_idx, px, py)) return; // exited lanes drop out of every mask
	RayRegs r;
	load_ray(p, ray_idx, px, py, r);

	float best_t = r.t_max, best_u = 0.0f, best_v = 0.0f;
	uint32_t best_slot = 0xFFFFFFFFu;
	uint32_t *stack = wave_stack[threadIdx.x / MRT_WAVE];

	const unsigned long long live = __ballot(true);
	const unsigned long long sx = __ballot(safe_inv(r.dx) < 0.0f), sy = __ballot(safe_inv(r.dy) < 0.0f),
			sz = __ballot(safe_inv(r.dz) < 0.0f);
	const bool uniform = (sx == 0ull || sx == live) && (sy == 0ull || sy == live) && (sz == 0ull || sz == live);
	const int oct = uniform ? ((sx ? 1 : 0) | (sy ? 2 : 0) | (sz ? 4 : 0)) : 8;
	uint32_t n_nodes = 0, n_tris = 0; // COUNT: wave-uniform in the asm walk, per lane (equal in all lanes) in the generic one
	if (oct == 8) { // mixed directions in one packet: the compiler-scheduled generic walk
		uint32_t nd = 0;
		packet_traverse<8, ANY_HIT, COUNT>(p, r, stack, best_t, best_u, best_v, best_slot, n_nodes, n_tris, nd);
		if (COUNT) { n_nodes = __builtin_amdgcn_readfirstlane(n_nodes); n_tris = __builtin_amdgcn_readfirstlane(n_tris); }
	} else {
		// sentinel at the bottom of the per-wave stack; sp = LDS byte address of the next free entry
		// (volatile: the asm block has no "memory" clobber — it only reads read-only scene data and this
		// private stack — so that the compiler keeps the triangle fetches of the leaf code scalar)
		*(volatile uint32_t *)&stack[0] = kSentinel;
		const uint32_t sp = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(stack + 4);
#define MRT_PKTA(O) case O: packet_traverse_asm<O, ANY_HIT, COUNT, KPF>(p, r, sp, best_t, best_u, best_v, best_slot, n_nodes, n_tris); break;
		switch (oct) { MRT_PKTA(0) MRT_PKTA(1) MRT_PKTA(2) MRT_PKTA(3) MRT_PKTA(4) MRT_PKTA(5) MRT_PKTA(6) MRT_PKTA(7) }
#undef MRT_PKTA
	}

	finish_ray(p, ray_idx, r, best_t, best_u, best_v, best_slot);
	if (p.tile_cost != nullptr && (threadIdx.x & (MRT_WAVE - 1)) == 0u) // (lane 0 = the tile's first pixel: it has a ray whenever the tile has one)
		note_tile_cost(p, ((uint64_t)block * MRT_WG + threadIdx.x) & ~63ull);

	if (COUNT) packet_count(p, n_nodes, n_tris, 0u, best_slot != 0xFFFFFFFFu, live);
}
