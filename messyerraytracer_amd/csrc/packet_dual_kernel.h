// packet_dual_kernel.h — TWO 64-ray packets per wave, walked in lockstep by one hand-written gfx950 node loop.
// Included by kernels.hip (inside namespace mrt, after packet_asm_kernel.h).
//
// Why: the packet walk is bound by the latency of its dependent node fetches, not by issue.  Occupancy sweep of
// trace_packet_asm_kernel (tools/exp_occupancy.py, profiles/r02a_occupancy.json): 8 / 4 / 2 waves per SIMD run
// C3 in 2.29 / 4.46 / 8.77 ms and C5 in 23.2 / 38.7 / 72.4 ms — time ~ 1 / (packets in flight), with the vector
// ALUs at 0.55 (C3) and 0.61 (C5) of their issue rate.  Eight waves per SIMD is the hardware's limit, so the only
// way to more node fetches in flight is more packets per wave: this kernel gives every wave two 8x8 tiles (A, B),
// fetches both packets' nodes with two s_load_dwordx16 behind ONE s_waitcnt, and runs the two 22-instruction slab
// tests back to back — 16 packets in flight per SIMD at the same 8 waves (<= 64 VGPRs, <= 80 SGPRs).
//
// Control: both packets step in every iteration of the fast path.  Anything else — a child that was not hit
// (pop), a leaf, a finished packet — sets a bit in an event register; the slow path resolves pops for both
// packets together (one LDS round trip) and leaves the block when a packet stands at a leaf or is finished.
// Leaves stay in C++ (the exact Moller-Trumbore of packet_leaf, shared with the single-packet walk); when one
// packet has finished, the other continues in the single-packet loop of packet_asm_kernel.h from where it
// stands (same stack layout).  The two packets of a wave must share one direction octant (adjacent tiles almost
// always do); otherwise they are walked one after the other by the single-packet code.
//
// Per-lane results are those of the one-ray walk, as in the other packet kernels: ownership masks (the lanes
// whose OWN ray passed a child's box test) travel with every stack entry, here read back as an SGPR pair.
#pragma once

// Fixed registers of the block:
//   packet A: node s[36:51] (s39 / s43 = left / right ref), mask of the lanes that own its current node s[60:61]
//   packet B: node s[64:79] (s67 / s71),                     mask s[62:63]
//   shared:   s52, s53 byte offsets, then s53 the far ref; s[54:55] right-child mask; s[56:57] order flags;
//             s[58:59] mask of the pushed child; v50..v55 slab values (one box at a time: six temporaries),
//             v50..v57 the popped entries; v58 the never-read destination of the far-child prefetches
// Event bits (%[ev]): 1 = A must pop, 2 = B must pop, 4 = A entered a leaf, 8 = B entered a leaf.
#define MRT_DUAL_STEP(P, NODE, SP, LREF, RREF, MASK, POPBIT, LEAFBIT, IX, IY, IZ, NRX, NRY, NRZ, TMIN, LIM,           \
		LNX, LFX, LNY, LFY, LNZ, LFZ, RNX, RFX, RNY, RFY, RNZ, RFZ)                                                   \
	"v_fma_f32 v50, " LNX ", " IX ", " NRX "\n"                                                                     \
	"v_fma_f32 v51, " LNY ", " IY ", " NRY "\n"                                                                     \
	"v_fma_f32 v52, " LNZ ", " IZ ", " NRZ "\n"                                                                     \
	"v_max_f32 v52, v52, " TMIN "\n"                                                                                \
	"v_max3_f32 v50, v50, v51, v52\n"       /* v50 = tl  = entry of the left box, clamped to t_min */              \
	"v_fma_f32 v51, " LFX ", " IX ", " NRX "\n"                                                                     \
	"v_fma_f32 v52, " LFY ", " IY ", " NRY "\n"                                                                     \
	"v_fma_f32 v53, " LFZ ", " IZ ", " NRZ "\n"                                                                     \
	"v_min_f32 v53, v53, " LIM "\n"                                                                                 \
	"v_min3_f32 v51, v51, v52, v53\n"       /* v51 = tlx = exit of the left box, clamped to best_t */              \
	"v_fma_f32 v52, " RNX ", " IX ", " NRX "\n"                                                                     \
	"v_fma_f32 v53, " RNY ", " IY ", " NRY "\n"                                                                     \
	"v_fma_f32 v54, " RNZ ", " IZ ", " NRZ "\n"                                                                     \
	"v_max_f32 v54, v54, " TMIN "\n"                                                                                \
	"v_max3_f32 v52, v52, v53, v54\n"       /* v52 = tr  */                                                        \
	"v_fma_f32 v53, " RFX ", " IX ", " NRX "\n"                                                                     \
	"v_fma_f32 v54, " RFY ", " IY ", " NRY "\n"                                                                     \
	"v_fma_f32 v55, " RFZ ", " IZ ", " NRZ "\n"                                                                     \
	"v_min_f32 v55, v55, " LIM "\n"                                                                                 \
	"v_min3_f32 v53, v53, v54, v55\n"       /* v53 = trx */                                                        \
	"v_cmp_le_f32 vcc, v50, v51\n"          /* lanes that hit the left child  */                                   \
	"v_cmp_le_f32_e64 s[54:55], v52, v53\n" /* lanes that hit the right child */                                   \
	"s_cbranch_vccz L_" P "lmiss_%=\n"                                                                              \
	"s_cmp_eq_u64 s[54:55], 0\n"                                                                                    \
	"s_cbranch_scc1 L_" P "onlyl_%=\n"                                                                              \
	"v_cmp_lt_f32_e64 s[56:57], v50, v52\n" /* both hit: lane 0 decides which is nearer (order = speed only) */    \
	"s_bitcmp1_b32 s56, 0\n"                                                                                        \
	"s_cselect_b32 s53, " RREF ", " LREF "\n"       /* far  */                                                     \
	"s_cselect_b32 " NODE ", " LREF ", " RREF "\n"  /* near */                                                     \
	"s_cselect_b64 s[58:59], s[54:55], vcc\n"       /* lanes that hit the far child  */                            \
	"s_cselect_b64 " MASK ", vcc, s[54:55]\n"       /* lanes that hit the near child */                            \
	"v_mov_b32 v51, s53\n"                                                                                          \
	"v_mov_b32 v52, s58\n"                                                                                          \
	"v_mov_b32 v53, s59\n"                                                                                          \
	"ds_write_b32 " SP ", v51\n"                                                                                    \
	"ds_write_b64 " SP ", v[52:53] offset:8\n"                                                                      \
	"v_add_u32 " SP ", 16, " SP "\n"                                                                                \
	"s_bitcmp1_b32 s53, 31\n"               /* the pushed child is an inner node: pull it towards the L2 now */    \
	"s_cbranch_scc1 L_" P "entered_%=\n"                                                                            \
	"v_lshlrev_b32 v54, 6, v51\n"                                                                                   \
	"global_load_dword v58, v54, %[base]\n" /* v58 is never read; vmcnt is drained at the exit */                  \
	"s_branch L_" P "entered_%=\n"                                                                                  \
	"L_" P "onlyl_%=:\n"                                                                                            \
	"s_mov_b32 " NODE ", " LREF "\n"                                                                                \
	"s_mov_b64 " MASK ", vcc\n"                                                                                     \
	"s_branch L_" P "entered_%=\n"                                                                                  \
	"L_" P "lmiss_%=:\n"                                                                                            \
	"s_cmp_eq_u64 s[54:55], 0\n"                                                                                    \
	"s_cbranch_scc1 L_" P "pop_%=\n"                                                                                \
	"s_mov_b32 " NODE ", " RREF "\n"                                                                                \
	"s_mov_b64 " MASK ", s[54:55]\n"                                                                                \
	"L_" P "entered_%=:\n"                  /* a child entered straight from its parent: a leaf ends the block */  \
	"s_bitcmp1_b32 " NODE ", 31\n"                                                                                  \
	"s_cbranch_scc0 L_" P "done_%=\n"                                                                               \
	"s_or_b32 %[ev], %[ev], " LEAFBIT "\n"                                                                          \
	"s_branch L_" P "done_%=\n"                                                                                     \
	"L_" P "pop_%=:\n"                                                                                              \
	"s_or_b32 %[ev], %[ev], " POPBIT "\n"                                                                           \
	"L_" P "done_%=:\n"

// The lockstep loop.  In: both packets at an inner node, or flagged for a pop in %[ev].  Out: at least one packet
// at a leaf (ref >= 0x80000000, its ownership mask in %[mA] / %[mB]) or finished (0x7FFFFFFF); the other one at
// an inner node whose step has not been taken yet, at a leaf, or finished.  %[ev] comes back 0.
#define MRT_ASM_DUAL_LOOP(CNT, LNX, LFX, LNY, LFY, LNZ, LFZ, RNX, RFX, RNY, RFY, RNZ, RFZ,                              \
		BLNX, BLFX, BLNY, BLFY, BLNZ, BLFZ, BRNX, BRFX, BRNY, BRFY, BRNZ, BRFZ)                                        \
	asm volatile(                                                                                                   \
		"s_branch L_top_%=\n"                                                                                       \
		"L_loop_%=:\n"                                                                                              \
		"s_lshl_b32 s52, %[nodeA], 6\n"                                                                             \
		"s_lshl_b32 s53, %[nodeB], 6\n"                                                                             \
		"s_load_dwordx16 s[36:51], %[base], s52\n"  /* both nodes in flight behind one wait */                     \
		"s_load_dwordx16 s[64:79], %[base], s53\n"                                                                  \
		CNT                                                                                                         \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		MRT_DUAL_STEP("A", "%[nodeA]", "%[spA]", "s39", "s43", "s[60:61]", "1", "4", "%[ixA]", "%[iyA]", "%[izA]",  \
				"%[nrxA]", "%[nryA]", "%[nrzA]", "%[tminA]", "%[limA]", LNX, LFX, LNY, LFY, LNZ, LFZ, RNX, RFX, RNY, RFY, RNZ, RFZ) \
		MRT_DUAL_STEP("B", "%[nodeB]", "%[spB]", "s67", "s71", "s[62:63]", "2", "8", "%[ixB]", "%[iyB]", "%[izB]",  \
				"%[nrxB]", "%[nryB]", "%[nrzB]", "%[tminB]", "%[limB]", BLNX, BLFX, BLNY, BLFY, BLNZ, BLFZ, BRNX, BRFX, BRNY, BRFY, BRNZ, BRFZ) \
		"L_top_%=:\n"                                                                                               \
		"s_cmp_eq_u32 %[ev], 0\n"                                                                                   \
		"s_cbranch_scc1 L_loop_%=\n"                                                                                \
		/* ---- slow path: pops of both packets behind one LDS wait ---- */                                         \
		"s_bitcmp1_b32 %[ev], 0\n"                                                                                  \
		"s_cbranch_scc0 L_s1_%=\n"                                                                                  \
		"v_add_u32 %[spA], -16, %[spA]\n"                                                                           \
		"ds_read_b128 v[50:53], %[spA]\n"       /* {ref, -, mask lo, mask hi} */                                   \
		"L_s1_%=:\n"                                                                                                \
		"s_bitcmp1_b32 %[ev], 1\n"                                                                                  \
		"s_cbranch_scc0 L_s2_%=\n"                                                                                  \
		"v_add_u32 %[spB], -16, %[spB]\n"                                                                           \
		"ds_read_b128 v[54:57], %[spB]\n"                                                                           \
		"L_s2_%=:\n"                                                                                                \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		"s_bitcmp1_b32 %[ev], 0\n"                                                                                  \
		"s_cbranch_scc0 L_s3_%=\n"                                                                                  \
		"v_readfirstlane_b32 %[nodeA], v50\n"                                                                       \
		"v_readfirstlane_b32 s60, v52\n"                                                                            \
		"v_readfirstlane_b32 s61, v53\n"                                                                            \
		"L_s3_%=:\n"                                                                                                \
		"s_bitcmp1_b32 %[ev], 1\n"                                                                                  \
		"s_cbranch_scc0 L_s4_%=\n"                                                                                  \
		"v_readfirstlane_b32 %[nodeB], v54\n"                                                                       \
		"v_readfirstlane_b32 s62, v56\n"                                                                            \
		"v_readfirstlane_b32 s63, v57\n"                                                                            \
		"L_s4_%=:\n"                                                                                                \
		"s_mov_b32 %[ev], 0\n"                                                                                      \
		"s_max_u32 s52, %[nodeA], %[nodeB]\n"   /* both below the sentinel = two inner nodes: on with the loop */  \
		"s_cmp_lt_u32 s52, 0x7fffffff\n"                                                                            \
		"s_cbranch_scc1 L_loop_%=\n"                                                                                \
		"s_waitcnt vmcnt(0)\n"                  /* no prefetch may land in v58 once the compiler owns it again */  \
		"s_mov_b64 %[mA], s[60:61]\n"                                                                               \
		"s_mov_b64 %[mB], s[62:63]\n"                                                                               \
		: [nodeA] "+s"(nodeA), [nodeB] "+s"(nodeB), [spA] "+v"(spA), [spB] "+v"(spB), [ev] "+s"(ev), [cnt] "+s"(cnt), \
		  [mA] "=&s"(maskA), [mB] "=&s"(maskB)                                                                      \
		: [base] "s"(base), [ixA] "v"(a.ix), [iyA] "v"(a.iy), [izA] "v"(a.iz), [nrxA] "v"(a.nrx), [nryA] "v"(a.nry), \
		  [nrzA] "v"(a.nrz), [tminA] "v"(a.tmin), [limA] "v"(a.lim), [ixB] "v"(b.ix), [iyB] "v"(b.iy), [izB] "v"(b.iz), \
		  [nrxB] "v"(b.nrx), [nryB] "v"(b.nry), [nrzB] "v"(b.nrz), [tminB] "v"(b.tmin), [limB] "v"(b.lim)            \
		: "vcc", "scc", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", \
		  "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", \
		  "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", \
		  "s79", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58")

// what the slab test of one packet reads (VGPRs)
struct PacketSlab { float ix, iy, iz, nrx, nry, nrz, tmin, lim; };

template <int OCT, bool COUNT>
__device__ __forceinline__ void packet_dual_loop_asm(const DevNode *base, uint32_t &nodeA, uint32_t &nodeB, uint32_t &spA, uint32_t &spB,
		uint32_t &ev, uint32_t &cnt, unsigned long long &maskA, unsigned long long &maskB, const PacketSlab &a, const PacketSlab &b)
{
	// per axis: inv >= 0 -> near plane = min, far plane = max; inv < 0 -> swapped.  Packet B's node sits 28 registers up.
#define MRT_DUAL_OCT(O, ...) \
	if (OCT == O) { if (COUNT) MRT_ASM_DUAL_LOOP(MRT_ASM_COUNT_STEP, __VA_ARGS__); else MRT_ASM_DUAL_LOOP("", __VA_ARGS__); }
	MRT_DUAL_OCT(0, "s36", "s40", "s37", "s41", "s38", "s42", "s44", "s48", "s45", "s49", "s46", "s50", "s64", "s68", "s65", "s69", "s66", "s70", "s72", "s76", "s73", "s77", "s74", "s78")
	MRT_DUAL_OCT(1, "s40", "s36", "s37", "s41", "s38", "s42", "s48", "s44", "s45", "s49", "s46", "s50", "s68", "s64", "s65", "s69", "s66", "s70", "s76", "s72", "s73", "s77", "s74", "s78")
	MRT_DUAL_OCT(2, "s36", "s40", "s41", "s37", "s38", "s42", "s44", "s48", "s49", "s45", "s46", "s50", "s64", "s68", "s69", "s65", "s66", "s70", "s72", "s76", "s77", "s73", "s74", "s78")
	MRT_DUAL_OCT(3, "s40", "s36", "s41", "s37", "s38", "s42", "s48", "s44", "s49", "s45", "s46", "s50", "s68", "s64", "s69", "s65", "s66", "s70", "s76", "s72", "s77", "s73", "s74", "s78")
	MRT_DUAL_OCT(4, "s36", "s40", "s37", "s41", "s42", "s38", "s44", "s48", "s45", "s49", "s50", "s46", "s64", "s68", "s65", "s69", "s70", "s66", "s72", "s76", "s73", "s77", "s78", "s74")
	MRT_DUAL_OCT(5, "s40", "s36", "s37", "s41", "s42", "s38", "s48", "s44", "s45", "s49", "s50", "s46", "s68", "s64", "s65", "s69", "s70", "s66", "s76", "s72", "s73", "s77", "s78", "s74")
	MRT_DUAL_OCT(6, "s36", "s40", "s41", "s37", "s42", "s38", "s44", "s48", "s49", "s45", "s50", "s46", "s64", "s68", "s69", "s65", "s70", "s66", "s72", "s76", "s77", "s73", "s78", "s74")
	MRT_DUAL_OCT(7, "s40", "s36", "s41", "s37", "s42", "s38", "s48", "s44", "s49", "s45", "s50", "s46", "s68", "s64", "s69", "s65", "s70", "s66", "s76", "s72", "s77", "s73", "s78", "s74")
#undef MRT_DUAL_OCT
}

// lane's bit of a wave-uniform 64-bit mask picks between two values (one v_cndmask with the SGPR pair as selector)
__device__ __forceinline__ float select_by_mask(unsigned long long mask, float if_set, float if_clear)
{
	float out;
	asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(if_clear), "v"(if_set), "s"(mask));
	return out;
}

// One packet's state in the dual walk
struct PacketState {
	float best_t, best_u, best_v;
	uint32_t best_slot, best_id;
	float lim_t;        // best_t for live lanes, -FLT_MAX for lanes that take no part (degenerate, invalid, any-hit done)
	uint32_t sp;        // LDS byte address of the next free stack entry
};

template <int OCT, bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void packet_traverse_dual(const TraceParams &p, const RayRegs &ra, const RayRegs &rb, bool dead_a, bool dead_b,
		PacketState &A, PacketState &B, uint32_t &n_nodes_a, uint32_t &n_tris_a, uint32_t &n_nodes_b, uint32_t &n_tris_b)
{
	const float4 *hot = reinterpret_cast<const float4 *>(p.tri_hot);
	PacketSlab sa, sb;
	sa.ix = safe_inv(ra.dx); sa.iy = safe_inv(ra.dy); sa.iz = safe_inv(ra.dz);
	sa.nrx = -(ra.ox * sa.ix); sa.nry = -(ra.oy * sa.iy); sa.nrz = -(ra.oz * sa.iz); sa.tmin = ra.t_min;
	sb.ix = safe_inv(rb.dx); sb.iy = safe_inv(rb.dy); sb.iz = safe_inv(rb.dz);
	sb.nrx = -(rb.ox * sb.ix); sb.nry = -(rb.oy * sb.iy); sb.nrz = -(rb.oz * sb.iz); sb.tmin = rb.t_min;
	A.lim_t = (dead_a || ra.t_min >= ra.t_max) ? -FLT_MAX : A.best_t;
	B.lim_t = (dead_b || rb.t_min >= rb.t_max) ? -FLT_MAX : B.best_t;
	uint32_t nodeA = 0u, nodeB = 0u, ev = 0u, steps = 0u; // both packets start at the root (always a wide node)
	for (;;) {
		unsigned long long maskA, maskB;
		sa.lim = A.lim_t; sb.lim = B.lim_t;
		packet_dual_loop_asm<OCT, COUNT>(p.nodes, nodeA, nodeB, A.sp, B.sp, ev, steps, maskA, maskB, sa, sb);
		nodeA = __builtin_amdgcn_readfirstlane(nodeA); nodeB = __builtin_amdgcn_readfirstlane(nodeB);
		steps = __builtin_amdgcn_readfirstlane(steps); // (tells the compiler the asm operands stay wave-uniform)
		ev = 0u;
		if (nodeA >= kLeafBit) {
			const bool own = select_by_mask(maskA, 1.0f, 0.0f) != 0.0f;
			packet_leaf<ANY_HIT, COUNT>(p, hot, ra, nodeA & 0x7FFFFFFFu, own, A.lim_t, A.best_t, A.best_u, A.best_v, A.best_slot, A.best_id, 0u, n_tris_a);
			if (ANY_HIT && __ballot(A.lim_t != -FLT_MAX) == 0ull) nodeA = kSentinel; // every lane of A has its answer
			else ev |= 1u;
		}
		if (nodeB >= kLeafBit) {
			const bool own = select_by_mask(maskB, 1.0f, 0.0f) != 0.0f;
			packet_leaf<ANY_HIT, COUNT>(p, hot, rb, nodeB & 0x7FFFFFFFu, own, B.lim_t, B.best_t, B.best_u, B.best_v, B.best_slot, B.best_id, 0u, n_tris_b);
			if (ANY_HIT && __ballot(B.lim_t != -FLT_MAX) == 0ull) nodeB = kSentinel;
			else ev |= 2u;
		}
		if (nodeA == kSentinel || nodeB == kSentinel) break; // one packet is finished: the other goes on alone
	}
	if (COUNT) { n_nodes_a += steps; n_nodes_b += steps; } // both packets take a step in every iteration of the lockstep loop
	// the rest of the packet that is left, in the single-packet loop, from where it stands
	if (nodeA != kSentinel) {
		uint32_t nn = 0;
		packet_traverse_asm<OCT, ANY_HIT, COUNT>(p, ra, A.sp, A.best_t, A.best_u, A.best_v, A.best_slot, nn, n_tris_a,
				nodeA, 0u, &A.best_id, A.lim_t == -FLT_MAX, (ev & 1u) ? 1u : 0u);
		if (COUNT) n_nodes_a += nn;
	} else if (nodeB != kSentinel) {
		uint32_t nn = 0;
		packet_traverse_asm<OCT, ANY_HIT, COUNT>(p, rb, B.sp, B.best_t, B.best_u, B.best_v, B.best_slot, nn, n_tris_b,
				nodeB, 0u, &B.best_id, B.lim_t == -FLT_MAX, (ev & 2u) ? 1u : 0u);
		if (COUNT) n_nodes_b += nn;
	}
}

// One packet of the wave by the single-packet walkers (packets that cannot be paired)
template <bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void packet_traverse_one(const TraceParams &p, const RayRegs &r, int oct, bool dead, uint32_t *stack,
		PacketState &S, uint32_t &n_nodes, uint32_t &n_tris)
{
	if (oct == 8 || p.n_nodes >= kAsmNodeLimit) { // mixed directions (or node offsets beyond the asm loop's 32 bits): the generic walk
		uint32_t nn = 0, nt = 0, nd = 0;
		packet_traverse<8, ANY_HIT, COUNT>(p, r, stack, S.best_t, S.best_u, S.best_v, S.best_slot, nn, nt, nd, 0u, 0u, &S.best_id, dead);
		if (COUNT) { n_nodes += __builtin_amdgcn_readfirstlane(nn); n_tris += __builtin_amdgcn_readfirstlane(nt); }
		return;
	}
#define MRT_PKT1(O) case O: packet_traverse_asm<O, ANY_HIT, COUNT>(p, r, S.sp, S.best_t, S.best_u, S.best_v, S.best_slot, n_nodes, n_tris, 0u, 0u, &S.best_id, dead); break;
	switch (oct) { MRT_PKT1(0) MRT_PKT1(1) MRT_PKT1(2) MRT_PKT1(3) MRT_PKT1(4) MRT_PKT1(5) MRT_PKT1(6) MRT_PKT1(7) }
#undef MRT_PKT1
}

// wave-uniform octant of a packet's reciprocal directions over the lanes in `part` (8 = mixed)
__device__ __forceinline__ int packet_octant(const RayRegs &r, bool part, unsigned long long &part_mask)
{
	part_mask = __ballot(part);
	const unsigned long long sx = __ballot(part && safe_inv(r.dx) < 0.0f), sy = __ballot(part && safe_inv(r.dy) < 0.0f),
			sz = __ballot(part && safe_inv(r.dz) < 0.0f);
	const bool uniform = (sx == 0ull || sx == part_mask) && (sy == 0ull || sy == part_mask) && (sz == 0ull || sz == part_mask);
	return uniform ? ((sx ? 1 : 0) | (sy ? 2 : 0) | (sz ? 4 : 0)) : 8;
}

#ifndef MRT_DUAL_WPE
#define MRT_DUAL_WPE 6
#endif
template <bool ANY_HIT, bool COUNT = false>
__global__ __launch_bounds__(MRT_WG) __attribute__((amdgpu_waves_per_eu(MRT_DUAL_WPE, 8))) void trace_packet_dual_kernel(const TraceParams p)
{
	// two stacks per wave, 16-byte entries {ref, -, lane mask}; entry 0 of each holds the sentinel
	__shared__ __attribute__((aligned(16))) uint32_t wave_stack[MRT_WG / MRT_WAVE][2][(MRT_PACKET_STACK + 1) * 4];
	if (skip_launch(p)) return;
	uint32_t block = blockIdx.x;
	if (p.xcd_swizzle) {
		const uint32_t per = gridDim.x >> 3;
		if (block < (per << 3)) block = (block & 7u) * per + (block >> 3);
	}
	// virtual thread indices of this lane's two rays: packets 2w and 2w + 1 of the launch order (neighbouring tiles)
	const uint32_t wave = threadIdx.x / MRT_WAVE, lane = threadIdx.x & (MRT_WAVE - 1);
	const uint64_t g_a = (((uint64_t)block * (MRT_WG / MRT_WAVE) + wave) * 2u) * MRT_WAVE + lane, g_b = g_a + MRT_WAVE;
	uint64_t idx_a = 0, idx_b = 0; uint32_t pxa = 0, pya = 0, pxb = 0, pyb = 0;
	const bool valid_a = lane_ray_index_g(p, g_a, idx_a, pxa, pya), valid_b = lane_ray_index_g(p, g_b, idx_b, pxb, pyb);
	if (__ballot(valid_a || valid_b) == 0ull) return; // nothing for this wave (every lane stays in otherwise)
	// a lane without a ray in a packet walks along with an empty interval
	RayRegs ra = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f}, rb = ra;
	if (valid_a) load_ray(p, idx_a, pxa, pya, ra);
	if (valid_b) load_ray(p, idx_b, pxb, pyb, rb);

	PacketState A, B;
	A.best_t = ra.t_max; A.best_u = 0.0f; A.best_v = 0.0f; A.best_slot = 0xFFFFFFFFu; A.best_id = 0xFFFFFFFFu; A.lim_t = 0.0f;
	B.best_t = rb.t_max; B.best_u = 0.0f; B.best_v = 0.0f; B.best_slot = 0xFFFFFFFFu; B.best_id = 0xFFFFFFFFu; B.lim_t = 0.0f;
	uint32_t *stack_a = wave_stack[wave][0], *stack_b = wave_stack[wave][1];
	*(volatile uint32_t *)&stack_a[0] = kSentinel;
	*(volatile uint32_t *)&stack_b[0] = kSentinel;
	A.sp = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(stack_a + 4);
	B.sp = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(stack_b + 4);

	unsigned long long part_a, part_b;
	const int oct_a = packet_octant(ra, valid_a, part_a), oct_b = packet_octant(rb, valid_b, part_b);
	uint32_t nna = 0, nta = 0, nnb = 0, ntb = 0; // COUNT: wave-uniform node steps / triangle rows of each packet
	if (part_a != 0ull && part_b != 0ull && oct_a == oct_b && oct_a != 8 && p.n_nodes < kAsmNodeLimit) {
#define MRT_PKTD(O) case O: packet_traverse_dual<O, ANY_HIT, COUNT>(p, ra, rb, !valid_a, !valid_b, A, B, nna, nta, nnb, ntb); break;
		switch (oct_a) { MRT_PKTD(0) MRT_PKTD(1) MRT_PKTD(2) MRT_PKTD(3) MRT_PKTD(4) MRT_PKTD(5) MRT_PKTD(6) MRT_PKTD(7) }
#undef MRT_PKTD
	} else {
		// different octants (tiles on an image axis), mixed directions, or only one packet: one after the other
		if (part_a != 0ull) packet_traverse_one<ANY_HIT, COUNT>(p, ra, oct_a, !valid_a, stack_a, A, nna, nta);
		if (part_b != 0ull) packet_traverse_one<ANY_HIT, COUNT>(p, rb, oct_b, !valid_b, stack_b, B, nnb, ntb);
	}

	if (valid_a) finish_ray(p, idx_a, ra, A.best_t, A.best_u, A.best_v, A.best_slot);
	if (valid_b) finish_ray(p, idx_b, rb, B.best_t, B.best_u, B.best_v, B.best_slot);

	if (COUNT) {
		if (valid_a) packet_count(p, nna, nta, 0u, A.best_slot != 0xFFFFFFFFu, part_a);
		if (valid_b) packet_count(p, nnb, ntb, 0u, B.best_slot != 0xFFFFFFFFu, part_b);
	}
}
