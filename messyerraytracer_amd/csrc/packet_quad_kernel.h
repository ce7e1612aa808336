// packet_quad_kernel.h — the 128-ray shared packet walk of packet_rows_kernel.h over FOUR-wide node rows, with the
// instruction stream cut to what gfx950 issues cheaply.  Included by kernels.hip (inside namespace mrt, after
// packet_rows_kernel.h, whose triangle test and operand macros it reuses).
//
// What the second half of round 2 measured (tools/ubench/issue_cost.hip, ubench/pk_fma_issue.hip, ubench/row_fetch.hip; counters of
// trace_packet_rows_kernel<2> in profiles/r02d_*), at 8 waves per SIMD:
//   * a wave's row-fetch waits are HIDDEN by the other waves: halving the fetches per packet (four-wide rows, the
//     first version of this file) cut the waits from 68 to 44 thousand cycles per wave and the kernel time by nothing;
//   * what the SIMD is short of is ISSUE: per wave-instruction, in wave cycles with every wave doing the same,
//     a vector op 12-19, an s_add 1.3, a branch not taken 13, TAKEN 32-41, a v_cmp feeding the scalar ALU 21,
//     a v_readfirstlane feeding it 33, an LDS write 41 — and they add.  The two-wide walk spent 10.4 k vector ops,
//     1.7 k branches and 1.3 k vector->scalar hand-overs per wave: the vector ALUs at 0.57, the rest mostly branches;
//   * v_pk_fma_f32 issues like v_fma_f32 and does two: 20 of them take the time of 20, not of 40.
// So this walk
//   * keeps the four-wide rows (half the steps: half the loop / pop / fetch instructions per box tested);
//   * tests a child box with 3 v_pk_fma_f32 + 5 instead of 6 v_fma_f32 + 5 (rows store {min, max} per axis as an
//     aligned SGPR pair; the ray's reciprocal direction and -(origin * reciprocal) sit in three VGPR pairs read with
//     op_sel), same IEEE fma per component, same results;
//   * runs the eight box tests of a step (4 children x 2 groups) as one straight line and decides afterwards from
//     a 4-bit word (which children any lane hit): no hit -> pop; one (43 % of the steps) -> s_movrels picks its ref
//     and masks; several (32 %) -> a short scalar loop orders them by lane 0's entry distance and pushes;
//   * branches once per step around a group whose 64 rays all left the subtree (31 % of the steps: half the tests).
//
// The row array: units of 64 bytes.  Node i of the four-wide collapse is the 128-byte row at unit 2i:
//   child k (k = 0..3) at dwords 6k..6k+5: min.x max.x min.y max.y min.z max.z; refs at dwords 24..27 (inner child
//   -> 2 * its index, leaf -> 0x80000000 | (2 * n_nodes4 + first slot), unused -> 0x7FFFFFFF, its box the point
//   +inf, which no ray interval reaches); dword 28 the child count.
// Triangle slot s is the 64-byte row {v0,id | e1,layers | e2,flags | normal} at unit 2 * n_nodes4 + s.
//
// Results: the collapse keeps the binary tree's exact boxes and drops inner levels only; the slab arithmetic is
// monotone in the box coordinate and boxes nest, so a ray that passes a child's box passes every dropped
// ancestor's, and every leaf's own box is still tested (it is a child box of some row): the walk reaches exactly
// the leaves the two-wide walk reaches, and tests their triangles with the same arithmetic, ownership masks and
// tie rule.  Order of visits = speed only.
//
// Registers: the row s[20:51] (a triangle row: s[20:35], read through the RA_ names); child k's hit masks
// s[52+2k : 53+2k] (group A) and s[60+2k : 61+2k] (group B); own masks s[68:69] (A), s[70:71] (B).  After the box
// tests the box registers s[20:43] are scratch: s20..s23 lane 0's entry distances, s24 the hit word, s25 k,
// the candidate s26 ref s27 distance s[28:29] s[30:31] masks, the child in hand s32 ref s33 distance s[34:35]
// s[36:37] masks, the entry to push s38 ref s[40:41] s[42:43] masks.  The triangle test's scratch (s52, s[54:59],
// s[64:67]) overlaps the masks: a step is one or the other.  v50..v57 temporaries, v58..v61 the entry distances
// of children 0..3 (of the group tested last), v62 prefetch dummy.
#pragma once

#define Q0_X "s[20:21]"
#define Q0_Y "s[22:23]"
#define Q0_Z "s[24:25]"
#define Q1_X "s[26:27]"
#define Q1_Y "s[28:29]"
#define Q1_Z "s[30:31]"
#define Q2_X "s[32:33]"
#define Q2_Y "s[34:35]"
#define Q2_Z "s[36:37]"
#define Q3_X "s[38:39]"
#define Q3_Y "s[40:41]"
#define Q3_Z "s[42:43]"
#define Q_TE0 "v58"
#define Q_TE1 "v59"
#define Q_TE2 "v60"
#define Q_TE3 "v61"

// after the three packed fmas: v50 / v51 = t at min.x / max.x, v52 / v53 = y, v54 / v55 = z.
// near / far plane of an axis by the octant's sign bit
#define MRT_QN_X0 "v50"
#define MRT_QF_X0 "v51"
#define MRT_QN_X1 "v51"
#define MRT_QF_X1 "v50"
#define MRT_QN_Y0 "v52"
#define MRT_QF_Y0 "v53"
#define MRT_QN_Y1 "v53"
#define MRT_QF_Y1 "v52"
#define MRT_QN_Z0 "v54"
#define MRT_QF_Z0 "v55"
#define MRT_QN_Z1 "v55"
#define MRT_QF_Z1 "v54"

// slab test of group P against child K's box (ray_aabb, glsl:84-99, octant-specialised; per component the fma of
// MRT_ROWS_SLAB: plane * inv + (-(origin * inv)), then the same max / min order).  P0 = (ix, iy), P1 = (iz, nrx),
// P2 = (nry, nrz).  TE = entry distance (kept), v57 = exit, both clamped to [t_min, lim]; MASK = lanes with entry <= exit
#ifdef MRT_Q_DUP_PK
#define MRT_Q_DUP(P, K) \
	"v_pk_fma_f32 v[50:51], " Q##K##_X ", " MRT_OP("p0", P) ", " MRT_OP("p1", P) " op_sel:[0,0,1] op_sel_hi:[1,0,1]\n" \
	"v_pk_fma_f32 v[52:53], " Q##K##_Y ", " MRT_OP("p0", P) ", " MRT_OP("p2", P) " op_sel:[0,1,0] op_sel_hi:[1,1,0]\n" \
	"v_pk_fma_f32 v[54:55], " Q##K##_Z ", " MRT_OP("p1", P) ", " MRT_OP("p2", P) " op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
#else
#define MRT_Q_DUP(P, K)
#endif
#ifdef MRT_Q_DUP_DECIDE
#define MRT_Q_DUPD \
		"s_mov_b32 s24, 0\n s_or_b64 s[22:23], s[58:59], s[66:67]\n s_addc_u32 s24, s24, s24\n s_or_b64 s[22:23], s[56:57], s[64:65]\n s_addc_u32 s24, s24, s24\n" \
		"s_or_b64 s[22:23], s[54:55], s[62:63]\n s_addc_u32 s24, s24, s24\n s_or_b64 s[22:23], s[52:53], s[60:61]\n s_addc_u32 s24, s24, s24\n"
#else
#define MRT_Q_DUPD
#endif
#define MRT_Q_SLAB(P, K, TE, BX, BY, BZ, MASK)                                                                        \
	MRT_Q_DUP(P, K)                                                                                                 \
	"v_pk_fma_f32 v[50:51], " Q##K##_X ", " MRT_OP("p0", P) ", " MRT_OP("p1", P) " op_sel:[0,0,1] op_sel_hi:[1,0,1]\n" \
	"v_pk_fma_f32 v[52:53], " Q##K##_Y ", " MRT_OP("p0", P) ", " MRT_OP("p2", P) " op_sel:[0,1,0] op_sel_hi:[1,1,0]\n" \
	"v_pk_fma_f32 v[54:55], " Q##K##_Z ", " MRT_OP("p1", P) ", " MRT_OP("p2", P) " op_sel:[0,0,1] op_sel_hi:[1,0,1]\n" \
	"v_max_f32 v56, " MRT_QN_Z##BZ ", " MRT_OP("tmin", P) "\n"                                                      \
	"v_min_f32 v57, " MRT_QF_Z##BZ ", " MRT_OP("lim", P) "\n"                                                       \
	"v_max3_f32 " TE ", " MRT_QN_X##BX ", " MRT_QN_Y##BY ", v56\n"                                                  \
	"v_min3_f32 v57, " MRT_QF_X##BX ", " MRT_QF_Y##BY ", v57\n"                                                     \
	"v_cmp_le_f32_e64 " MASK ", " TE ", v57\n"

// the four children against one group: masks into s[M0 + 2k]
#define MRT_Q_SLAB4(P, BX, BY, BZ, M0, M1, M2, M3)                                                                    \
	MRT_Q_SLAB(P, 0, Q_TE0, BX, BY, BZ, M0)                                                                         \
	MRT_Q_SLAB(P, 1, Q_TE1, BX, BY, BZ, M1)                                                                         \
	MRT_Q_SLAB(P, 2, Q_TE2, BX, BY, BZ, M2)                                                                         \
	MRT_Q_SLAB(P, 3, Q_TE3, BX, BY, BZ, M3)
#define MRT_Q_MASKS_A "s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]"
#define MRT_Q_MASKS_B "s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]"

#ifdef MRT_Q_NO_ONLY
#define MRT_Q_ONLY_CHECKS
#else
#define MRT_Q_ONLY_CHECKS "s_cmp_eq_u64 s[68:69], 0\n s_cbranch_scc1 L_onlyB_%=\n s_cmp_eq_u64 s[70:71], 0\n s_cbranch_scc1 L_onlyA_%=\n"
#endif
#define MRT_QUAD_CLOBBERS MRT_ROWS_CLOBBERS, "s68", "s69", "s70", "s71", "v61", "v62", "m0"

// counting builds: the stack's high-water mark (s39 is free during a push)
#define MRT_QUAD_CNT_P "v_readfirstlane_b32 s39, %[spA]\n s_max_u32 %[cntp], %[cntp], s39\n"

// push {group A's mask, group B's mask, ref, -} = s[40:41], s[42:43], s38
#define MRT_Q_PUSH(CNT_P)                                                                                             \
	"v_mov_b32 v50, s40\n"                                                                                          \
	"v_mov_b32 v51, s41\n"                                                                                          \
	"v_mov_b32 v52, s42\n"                                                                                          \
	"v_mov_b32 v53, s43\n"                                                                                          \
	"v_mov_b32 v54, s38\n"                                                                                          \
	"ds_write_b128 %[spA], v[50:53]\n"                                                                              \
	"ds_write_b32 %[spA], v54 offset:16\n"                                                                          \
	"v_add_u32 %[spA], 32, %[spA]\n"                                                                                \
	CNT_P                                                                                                           \
	"v_lshlrev_b32 v55, 6, v54\n"                 /* pull the pushed row towards the L2 now */                      \
	"global_load_dword v62, v55, %[rows]\n"       /* v62 is never read (vmcnt drained at the very end) */

// In: cur (group A's operand) = a row to visit, the groups' own masks.  Out: cur = 0x7FFFFFFF.
// Stack entries are 32 bytes: {group A's mask, group B's mask, ref, -}; the sentinel entry has ref 0x7FFFFFFF.
#define MRT_QUAD_LOOP(CNT_N, CNT_T, CNT_P, T_PRE, T_POST, ANYA, ANYB, ANYDONE, BX, BY, BZ)                            \
	asm volatile(                                                                                                   \
		"s_mov_b64 s[68:69], %[maskA]\n"                                                                            \
		"s_mov_b64 s[70:71], %[maskB]\n"                                                                            \
		"L_loop_%=:\n"                                                                                              \
		"s_lshl_b32 s52, %[curA], 6\n"                                                                              \
		T_PRE                                                                                                       \
		"s_load_dwordx16 s[20:35], %[rows], s52\n"                                                                  \
		"s_bitcmp1_b32 %[curA], 31\n"                                                                               \
		"s_cbranch_scc1 L_tri_%=\n"                                                                                 \
		"s_load_dwordx16 s[36:51], %[rows], s52 offset:64\n"                                                        \
		CNT_N                                                                                                       \
		MRT_Q_ONLY_CHECKS                                                                                           \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		T_POST                                                                                                      \
		MRT_Q_SLAB4("A", BX, BY, BZ, "s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]")                                \
		MRT_Q_SLAB4("B", BX, BY, BZ, "s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]")                                \
		"L_decide_%=:\n"                                                                                            \
		/* the hit word: bit k = some lane of the 128 hit child k */                                                \
		MRT_Q_DUPD MRT_Q_DUPD                                                                                       \
		"s_mov_b32 s24, 0\n"                                                                                        \
		"s_or_b64 s[22:23], s[58:59], s[66:67]\n"                                                                   \
		"s_addc_u32 s24, s24, s24\n"                                                                                \
		"s_or_b64 s[22:23], s[56:57], s[64:65]\n"                                                                   \
		"s_addc_u32 s24, s24, s24\n"                                                                                \
		"s_or_b64 s[22:23], s[54:55], s[62:63]\n"                                                                   \
		"s_addc_u32 s24, s24, s24\n"                                                                                \
		"s_or_b64 s[22:23], s[52:53], s[60:61]\n"                                                                   \
		"s_addc_u32 s24, s24, s24\n"                                                                                \
		"s_cmp_eq_u32 s24, 0\n"                                                                                     \
		"s_cbranch_scc1 L_pop_%=\n"                                                                                 \
		"s_ff1_i32_b32 s25, s24\n"                    /* the first child hit */                                     \
		"s_bitset0_b32 s24, s25\n"                                                                                  \
		"s_cmp_lg_u32 s24, 0\n"                                                                                     \
		"s_cbranch_scc1 L_multi_%=\n"                                                                               \
		/* one child hit: it is the next row */                                                                     \
		"s_mov_b32 m0, s25\n"                                                                                       \
		"s_lshl_b32 s25, s25, 1\n"                                                                                  \
		"s_movrels_b32 %[curA], s44\n"                                                                              \
		"s_mov_b32 m0, s25\n"                                                                                       \
		"s_nop 0\n"                                                                                                 \
		"s_movrels_b64 s[68:69], s[52:53]\n"                                                                        \
		"s_movrels_b64 s[70:71], s[60:61]\n"                                                                        \
		"s_branch L_loop_%=\n"                                                                                      \
		"L_tri_%=:\n"                                                                                               \
		CNT_T                                                                                                       \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		T_POST                                                                                                      \
		"s_cmp_eq_u64 s[68:69], 0\n"                                                                                \
		"s_cbranch_scc1 L_Atnext_%=\n"                                                                              \
		MRT_ROWS_TRI_TEST("A", RA, "s[68:69]", "%[curA]", ANYA)                                                     \
		"s_cmp_eq_u64 s[70:71], 0\n"                                                                                \
		"s_cbranch_scc1 L_Btnext_%=\n"                                                                              \
		MRT_ROWS_TRI_TEST("B", RA, "s[70:71]", "%[curA]", ANYB)                                                     \
		ANYDONE                                                                                                     \
		"s_bitcmp1_b32 " RA_FLAGS ", 0\n"        /* the last triangle of its leaf? */                               \
		"s_cbranch_scc1 L_pop_%=\n"                                                                                 \
		"s_add_u32 %[curA], %[curA], 1\n"                                                                           \
		"s_branch L_loop_%=\n"                                                                                      \
		"L_pop_%=:\n"                                                                                               \
		"v_add_u32 %[spA], -32, %[spA]\n"                                                                           \
		"ds_read_b128 v[50:53], %[spA]\n"                                                                           \
		"ds_read_b32 v54, %[spA] offset:16\n"                                                                       \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		"v_readfirstlane_b32 %[curA], v54\n"                                                                        \
		"v_readfirstlane_b32 s68, v50\n"                                                                            \
		"v_readfirstlane_b32 s69, v51\n"                                                                            \
		"v_readfirstlane_b32 s70, v52\n"                                                                            \
		"v_readfirstlane_b32 s71, v53\n"                                                                            \
		"s_cmp_lg_u32 %[curA], 0x7fffffff\n"                                                                        \
		"s_cbranch_scc1 L_loop_%=\n"                                                                                \
		"s_branch L_exit_%=\n"                                                                                      \
		/* ---- out of line: a group whose rays have all left this subtree ---- */                                  \
		"L_onlyA_%=:\n"                                                                                             \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		T_POST                                                                                                      \
		MRT_Q_SLAB4("A", BX, BY, BZ, "s[52:53]", "s[54:55]", "s[56:57]", "s[58:59]")                                \
		"s_mov_b64 s[60:61], 0\n"                                                                                   \
		"s_mov_b64 s[62:63], 0\n"                                                                                   \
		"s_mov_b64 s[64:65], 0\n"                                                                                   \
		"s_mov_b64 s[66:67], 0\n"                                                                                   \
		"s_branch L_decide_%=\n"                                                                                    \
		"L_onlyB_%=:\n"                                                                                             \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		T_POST                                                                                                      \
		MRT_Q_SLAB4("B", BX, BY, BZ, "s[60:61]", "s[62:63]", "s[64:65]", "s[66:67]")                                \
		"s_mov_b64 s[52:53], 0\n"                                                                                   \
		"s_mov_b64 s[54:55], 0\n"                                                                                   \
		"s_mov_b64 s[56:57], 0\n"                                                                                   \
		"s_mov_b64 s[58:59], 0\n"                                                                                   \
		"s_branch L_decide_%=\n"                                                                                    \
		/* ---- out of line: several children hit.  s25 = the first one, s24 = the others (not 0) ---- */           \
		"L_multi_%=:\n"                                                                                             \
		"v_readfirstlane_b32 s20, " Q_TE0 "\n"        /* lane 0's entry distances, of the group tested last */      \
		"v_readfirstlane_b32 s21, " Q_TE1 "\n"                                                                      \
		"v_readfirstlane_b32 s22, " Q_TE2 "\n"                                                                      \
		"v_readfirstlane_b32 s23, " Q_TE3 "\n"                                                                      \
		"s_mov_b32 m0, s25\n"                         /* the candidate: the first child hit */                      \
		"s_lshl_b32 s25, s25, 1\n"                                                                                  \
		"s_movrels_b32 s26, s44\n"                                                                                  \
		"s_movrels_b32 s27, s20\n"                                                                                  \
		"s_mov_b32 m0, s25\n"                                                                                       \
		"s_nop 0\n"                                                                                                 \
		"s_movrels_b64 s[28:29], s[52:53]\n"                                                                        \
		"s_movrels_b64 s[30:31], s[60:61]\n"                                                                        \
		"L_more_%=:\n"                                                                                              \
		"s_ff1_i32_b32 s25, s24\n"                    /* the next child hit */                                      \
		"s_bitset0_b32 s24, s25\n"                                                                                  \
		"s_mov_b32 m0, s25\n"                                                                                       \
		"s_lshl_b32 s25, s25, 1\n"                                                                                  \
		"s_movrels_b32 s32, s44\n"                                                                                  \
		"s_movrels_b32 s33, s20\n"                                                                                  \
		"s_mov_b32 m0, s25\n"                                                                                       \
		"s_nop 0\n"                                                                                                 \
		"s_movrels_b64 s[34:35], s[52:53]\n"                                                                        \
		"s_movrels_b64 s[36:37], s[60:61]\n"                                                                        \
		"s_cmp_lt_u32 s33, s27\n"                     /* nearer than the candidate (bit patterns: distances >= t_min >= 0)? */ \
		"s_cselect_b32 s38, s26, s32\n"               /* the farther of the two is pushed ... */                    \
		"s_cselect_b64 s[40:41], s[28:29], s[34:35]\n"                                                              \
		"s_cselect_b64 s[42:43], s[30:31], s[36:37]\n"                                                              \
		"s_cselect_b32 s26, s32, s26\n"               /* ... the nearer one is the candidate */                     \
		"s_cselect_b32 s27, s33, s27\n"                                                                             \
		"s_cselect_b64 s[28:29], s[34:35], s[28:29]\n"                                                              \
		"s_cselect_b64 s[30:31], s[36:37], s[30:31]\n"                                                              \
		MRT_Q_PUSH(CNT_P)                                                                                           \
		"s_cmp_lg_u32 s24, 0\n"                                                                                     \
		"s_cbranch_scc1 L_more_%=\n"                                                                                \
		"s_mov_b32 %[curA], s26\n"                                                                                  \
		"s_mov_b64 s[68:69], s[28:29]\n"                                                                            \
		"s_mov_b64 s[70:71], s[30:31]\n"                                                                            \
		"s_branch L_loop_%=\n"                                                                                      \
		"L_exit_%=:\n"                                                                                              \
		"s_waitcnt vmcnt(0)\n"               /* no prefetch may land in v62 once the compiler owns it again */     \
		"s_mov_b64 %[maskA], s[68:69]\n"                                                                            \
		"s_mov_b64 %[maskB], s[70:71]\n"                                                                            \
		: MRT_QUAD_OUT(A, a), [limB] "+v"(b.lim), [btB] "+v"(b.bt), [buB] "+v"(b.bu), [bvB] "+v"(b.bv),             \
		  [bsB] "+v"(b.bs), [biB] "+v"(b.bi), [maskB] "+s"(b.mask), [cntn] "+s"(cnt_n), [cntt] "+s"(cnt_t), [cntp] "+s"(sp_max), [cntw] "+s"(cnt_w) \
		: MRT_QUAD_IN(A, a), MRT_QUAD_IN(B, b), [rows] "s"(rows), [qmask] "s"(qmask), [eps] "s"(eps), [vneg] "v"(vneg) \
		: MRT_QUAD_CLOBBERS)

// everything a group's walk holds in registers
struct QuadRegs {
	float ox, oy, oz, dx, dy, dz, tmin;   // the ray (t_max lives on in lim / bt)
	unsigned long long p0, p1, p2;        // (ix, iy), (iz, nrx), (nry, nrz): safe_inv(d) and -(o * inv), as VGPR pairs
	float lim;                            // far limit of the box tests: best_t, or -FLT_MAX for a lane that takes no part
	float bt, bu, bv;                     // best hit
	uint32_t bs, bi;                      // its row unit (0xFFFFFFFF = none) and triangle id
	uint32_t sp;                          // LDS byte address of the next free stack entry (group A's is the wave's)
	uint32_t cur;                         // wave-uniform: the row to visit next (bit 31: a triangle), 0x7FFFFFFF = finished
	unsigned long long mask;              // wave-uniform: the lanes that own the current row
};
#define MRT_QUAD_OUT(P, S)                                                                                            \
	[cur##P] "+s"(S.cur), [sp##P] "+v"(S.sp), [lim##P] "+v"(S.lim), [bt##P] "+v"(S.bt), [bu##P] "+v"(S.bu),         \
	[bv##P] "+v"(S.bv), [bs##P] "+v"(S.bs), [bi##P] "+v"(S.bi), [mask##P] "+s"(S.mask)
#define MRT_QUAD_IN(P, S)                                                                                             \
	[ox##P] "v"(S.ox), [oy##P] "v"(S.oy), [oz##P] "v"(S.oz), [dx##P] "v"(S.dx), [dy##P] "v"(S.dy), [dz##P] "v"(S.dz), \
	[tmin##P] "v"(S.tmin), [p0##P] "v"(S.p0), [p1##P] "v"(S.p1), [p2##P] "v"(S.p2)

__device__ __forceinline__ unsigned long long quad_pair(float lo, float hi)
{
	return ((unsigned long long)__float_as_uint(hi) << 32) | __float_as_uint(lo);
}

__device__ __forceinline__ void quad_init(QuadRegs &s, const RayRegs &r, bool takes_part, uint32_t sp)
{
	s.ox = r.ox; s.oy = r.oy; s.oz = r.oz; s.dx = r.dx; s.dy = r.dy; s.dz = r.dz; s.tmin = r.t_min;
	const float ix = safe_inv(r.dx), iy = safe_inv(r.dy), iz = safe_inv(r.dz);
	const float nrx = -(r.ox * ix), nry = -(r.oy * iy), nrz = -(r.oz * iz);
	s.p0 = quad_pair(ix, iy); s.p1 = quad_pair(iz, nrx); s.p2 = quad_pair(nry, nrz);
	s.bt = r.t_max; s.bu = 0.0f; s.bv = 0.0f; s.bs = 0xFFFFFFFFu; s.bi = 0xFFFFFFFFu;
	s.lim = (!takes_part || r.t_min >= r.t_max) ? -FLT_MAX : r.t_max; // degenerate rays are misses, glsl:214-222
	s.sp = sp; s.cur = 0u; s.mask = 0ull;
}

template <int OCT, bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void quad_walk(const float4 *rows, uint32_t qmask, QuadRegs &a, QuadRegs &b, uint32_t &cnt_n, uint32_t &cnt_t, uint32_t &cnt_w, uint32_t &sp_max)
{
	const float eps = 1e-8f, vneg = -FLT_MAX;
#define MRT_QW(O, BX, BY, BZ)                                                                                         \
	if (OCT == O) {                                                                                                 \
		if (COUNT) { if (ANY_HIT) MRT_QUAD_LOOP(MRT_ROWS_CNT_N, MRT_ROWS_CNT_T, MRT_QUAD_CNT_P, MRT_ROWS_T_PRE, MRT_ROWS_T_POST, MRT_ROWSW_ANYHIT("A"), MRT_ROWSW_ANYHIT("B"), MRT_ROWSW_ANYDONE, BX, BY, BZ); \
			else MRT_QUAD_LOOP(MRT_ROWS_CNT_N, MRT_ROWS_CNT_T, MRT_QUAD_CNT_P, MRT_ROWS_T_PRE, MRT_ROWS_T_POST, MRT_ROWS_NEAREST("A"), MRT_ROWS_NEAREST("B"), "", BX, BY, BZ); } \
		else { if (ANY_HIT) MRT_QUAD_LOOP("", "", "", "", "", MRT_ROWSW_ANYHIT("A"), MRT_ROWSW_ANYHIT("B"), MRT_ROWSW_ANYDONE, BX, BY, BZ); \
			else MRT_QUAD_LOOP("", "", "", "", "", MRT_ROWS_NEAREST("A"), MRT_ROWS_NEAREST("B"), "", BX, BY, BZ); }         \
	}
	MRT_QW(0, 0, 0, 0) MRT_QW(1, 1, 0, 0) MRT_QW(2, 0, 1, 0) MRT_QW(3, 1, 1, 0) MRT_QW(4, 0, 0, 1) MRT_QW(5, 1, 0, 1) MRT_QW(6, 0, 1, 1) MRT_QW(7, 1, 1, 1)
#undef MRT_QW
	a.cur = __builtin_amdgcn_readfirstlane(a.cur);
	cnt_n = __builtin_amdgcn_readfirstlane(cnt_n); cnt_t = __builtin_amdgcn_readfirstlane(cnt_t); cnt_w = __builtin_amdgcn_readfirstlane(cnt_w);
	sp_max = __builtin_amdgcn_readfirstlane(sp_max);
}

template <bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void quad_walk_oct(int oct, const float4 *rows, uint32_t qmask, QuadRegs &a, QuadRegs &b, uint32_t &cnt_n, uint32_t &cnt_t, uint32_t &cnt_w, uint32_t &sp_max)
{
#define MRT_QO(O) case O: quad_walk<O, ANY_HIT, COUNT>(rows, qmask, a, b, cnt_n, cnt_t, cnt_w, sp_max); break;
	switch (oct) { MRT_QO(0) MRT_QO(1) MRT_QO(2) MRT_QO(3) MRT_QO(4) MRT_QO(5) MRT_QO(6) MRT_QO(7) }
#undef MRT_QO
}

// Two packets (neighbouring tiles of the launch order) per wave.  Packets of one octant share one walk; a packet
// whose neighbour looks another way (tiles on an image axis), or has none, walks with an empty partner group
// (own mask 0: never tested); a packet of mixed directions takes the generic compiler-scheduled walk.
template <bool ANY_HIT, bool COUNT>
__global__ __launch_bounds__(MRT_WG) __attribute__((amdgpu_waves_per_eu(MRT_ROWS_WPE, 8))) void trace_packet_quad_kernel(const TraceParams p)
{
	// per wave: 32-byte stack entries {group masks, ref, -}; entry 0 holds the sentinel
	__shared__ __attribute__((aligned(16))) uint32_t wave_stack[MRT_WG / MRT_WAVE][(MRT_PACKET_STACK + 1) * 8];
	if (skip_launch(p)) return;
	uint32_t block = blockIdx.x;
	if (p.xcd_swizzle) {
		const uint32_t per = gridDim.x >> 3;
		if (block < (per << 3)) block = (block & 7u) * per + (block >> 3);
	}
	const uint32_t wave = threadIdx.x / MRT_WAVE, lane = threadIdx.x & (MRT_WAVE - 1);
	const uint64_t g_a = (((uint64_t)block * (MRT_WG / MRT_WAVE) + wave) * 2u) * MRT_WAVE + lane, g_b = g_a + MRT_WAVE;
	uint64_t idx = 0; uint32_t px = 0, py = 0;
	const bool valid_a = lane_ray_index_g(p, g_a, idx, px, py);
	// a lane without a ray in a packet walks along with an empty interval
	RayRegs ra = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f}, rb = ra;
	if (valid_a) load_ray(p, idx, px, py, ra);
	const bool valid_b = lane_ray_index_g(p, g_b, idx, px, py);
	if (valid_b) load_ray(p, idx, px, py, rb);
	if (__ballot(valid_a || valid_b) == 0ull) return; // nothing for this wave (otherwise every lane stays in)

	const float4 *rows = reinterpret_cast<const float4 *>(p.row_array4);
	uint32_t *stack = wave_stack[wave];
	// the sentinel (volatile: the asm blocks have no "memory" clobber; they touch only read-only scene data and this stack)
	*(volatile uint32_t *)&stack[4] = kSentinel;
	const uint32_t sp0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(stack + 8);
	QuadRegs A, B;
	quad_init(A, ra, valid_a, sp0);
	quad_init(B, rb, valid_b, sp0);

	unsigned long long part_a = 0ull, part_b = 0ull;
	const int oct_a = rows_octant(ra, valid_a, part_a), oct_b = rows_octant(rb, valid_b, part_b);
	uint32_t cnt_n = 0u, cnt_t = 0u, sp_max = 0u; // COUNT: node rows / triangle rows fetched by this wave, highest stack pointer
	uint32_t cnt_w = 0u;                          // COUNT: shader cycles between issuing a row fetch and having it
	const unsigned long long t_start = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
	bool done_a = part_a == 0ull, done_b = part_b == 0ull;
	// pass 0: both packets in one walk when they look the same way, else packet A with an empty partner group
	// (own mask 0: never tested; in an any-hit walk it only keeps the walk going until the stack has drained);
	// pass 1: packet B on its own.  One call site: the register allocator sees the eight loops once.
#pragma nounroll
	for (int pass = 0; pass < 2; pass++) {
		const bool paired = pass == 0 && !done_a && !done_b && oct_a == oct_b;
		const bool walk_a = pass == 0 && !done_a && oct_a != 8, walk_b = (paired || pass == 1) && !done_b && oct_b != 8;
		if (!walk_a && !walk_b) continue;
		A.cur = 0u; A.sp = sp0;
		A.mask = walk_a ? part_a : 0ull; B.mask = walk_b ? part_b : 0ull; // at the root every lane that takes part owns the row
		quad_walk_oct<ANY_HIT, COUNT>(walk_a ? oct_a : oct_b, rows, p.query_mask, A, B, cnt_n, cnt_t, cnt_w, sp_max);
		if (walk_a) done_a = true;
		if (walk_b) done_b = true;
	}
	// best hit as the other kernels keep it: leaf-order slot = row unit - first triangle unit
	uint32_t slot_a = A.bs == 0xFFFFFFFFu ? 0xFFFFFFFFu : A.bs - p.tri_unit_base4, slot_b = B.bs == 0xFFFFFFFFu ? 0xFFFFFFFFu : B.bs - p.tri_unit_base4;
	// mixed directions inside a packet: the generic walk over the two-wide nodes + triangle arrays
	if (!done_a) {
		uint32_t nn = 0, nt = 0, nd = 0;
		A.bt = ra.t_max;
		packet_traverse<8, ANY_HIT, COUNT>(p, ra, stack, A.bt, A.bu, A.bv, slot_a, nn, nt, nd, 0u, 0u, nullptr, !valid_a);
		if (COUNT) { cnt_n += __builtin_amdgcn_readfirstlane(nn); cnt_t += __builtin_amdgcn_readfirstlane(nt); }
	}
	if (!done_b) {
		uint32_t nn = 0, nt = 0, nd = 0;
		B.bt = rb.t_max;
		packet_traverse<8, ANY_HIT, COUNT>(p, rb, stack, B.bt, B.bu, B.bv, slot_b, nn, nt, nd, 0u, 0u, nullptr, !valid_b);
		if (COUNT) { cnt_n += __builtin_amdgcn_readfirstlane(nn); cnt_t += __builtin_amdgcn_readfirstlane(nt); }
	}

	// the rays' indices again (not kept across the walk: registers)
	if (valid_a) { lane_ray_index_g(p, g_a, idx, px, py); finish_ray(p, idx, ra, A.bt, A.bu, A.bv, slot_a); }
	if (valid_b) { lane_ray_index_g(p, g_b, idx, px, py); finish_ray(p, idx, rb, B.bt, B.bu, B.bv, slot_b); }

	if (COUNT && lane == 0u && (p.count_mode != 2u || (blockIdx.x & 15u) == 0u)) { // the wave's clock (count_visits = 2: that only, sampled)
		atomicAdd(&p.counters[kCntFetchWaitCycles], (unsigned long long)cnt_w);
		atomicAdd(&p.counters[kCntWaveCycles], __builtin_amdgcn_s_memtime() - t_start);
		atomicAdd(&p.counters[kCntWaves], 1ull);
	}
	if (COUNT && p.count_mode != 2u) {
		// the stack's high-water mark, in entries (the bound is the collapse's worst case, stack4 <= 64: api.hip)
		const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)stack;
		atomicMax(&p.counters[kCntMaxStack], (unsigned long long)(sp_max > base ? (sp_max - base) / 32u : 0u));
		// per-ray words: every step of the wave is charged to every ray of the wave (both packets): an upper bound
		// per packet; the fetch words (kCntWaveNodeFetch / kCntWaveTriFetch) are exact (a node row is 128 bytes here)
		if (valid_a) packet_count(p, cnt_n, cnt_t, 0u, slot_a != 0xFFFFFFFFu, part_a | part_b);
		if (valid_b) { atomicAdd(&p.counters[kCntRays], 1ull); if (slot_b != 0xFFFFFFFFu) atomicAdd(&p.counters[kCntHits], 1ull); }
	}
}
