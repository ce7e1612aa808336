// packet_rows_kernel.h — the packet walk written end to end in gfx950 assembly: 64 rays (one packet) or 128 rays
// (two 8x8 tiles sharing ONE walk) per wave.  Included by kernels.hip (inside namespace mrt, after packet_asm_kernel.h).
//
// What round 2 measured about trace_packet_asm_kernel (profiles/r02a_*, r02b_*):
//   * occupancy sweep (tools/exp_occupancy.py): 8 / 4 / 2 waves per SIMD run C3 in 2.29 / 4.46 / 8.77 ms: a wave's
//     lifetime does not depend on how many others are resident; the vector ALUs sit at 0.55 of their issue rate;
//   * s_memtime clock (tools/exp_walk_clock.py): a row fetch costs ~200 cycles and these waits are 30-40 % of a
//     wave's life at C3 (45 % at C5); the rest is the wave's own instruction stream at ~12 cycles per instruction;
//   * the triangle loads of the leaf code were never scalar (SQ_INSTS_VMEM_RD 91 per wave): three vector loads per
//     triangle, behind an s_waitcnt vmcnt(0) that also waited for the far-child prefetches;
//   * two packets per wave walked SEPARATELY in lockstep (one wait for two fetches) halved the waits and nothing
//     else: 2.19 against 2.15 ms.  Retired.
// So the lever is the instruction stream and the fetches PER RAY.  This kernel
//   * keeps the whole walk inside one assembly block: ONE row array holds the scene's wide nodes followed by its
//     triangles as 64-byte rows ({v0,id | e1,layers | e2,flags | normal}, the reference's GPUTrianglePacked row),
//     leaf refs rebased to row indices, so a packet's current item — inner node or triangle — is always
//     `rows[ref & 0x3FFFFFF]`: one s_load_dwordx16 through the scalar cache, address = one shift; the triangle test
//     (Moller-Trumbore, the correctly rounded 1/det sequence, the tie rule, per-lane ownership) is in the block too;
//   * and gives a wave TWO rays per lane (groups A and B = two neighbouring 8x8 tiles) that share one walk: one
//     fetch, one stack, one near/far decision per step for 128 rays, the slab / triangle arithmetic once per group
//     — and only for a group that has a lane owning the row (ownership masks per group travel with every stack
//     entry), so a group never tests a row its own 64-ray walk would not have reached.  Scalar work, branches,
//     stack traffic and fetch waits per ray drop by the share of rows the two tiles have in common (most of them
//     when a tile is smaller than a triangle).
//
// Arithmetic is the canonical form of DESIGN.md (the same fma / mul sequences the compiler emits for
// packet_leaf and the lane kernels: explicit fma, nothing contracted, 1/det correctly rounded by the
// v_div_scale / v_rcp / v_div_fmas / v_div_fixup sequence), so results stay bit-identical to the oracle.
// gfx950 hazards respected by construction: a lane mask written by v_cmp is consumed by the scalar ALU, or by a
// vector instruction at least two instructions later; v_rcp's result is first read two instructions later;
// v_div_scale's VCC reaches v_div_fmas six instructions later.
#pragma once

// ---- fixed registers ----------------------------------------------------------------------------------------
// the current row: s[20:35]
//   node row:     lmin.xyz lref | lmax.xyz rref | rmin.xyz - | rmax.xyz -
//   triangle row: v0.xyz id | e1.xyz layers | e2.xyz flags | normal (unused here)
// own masks (the lanes whose own ray hit the current row's box): group A s[60:61], group B s[62:63]
// scratch: s52 (byte offset), s53 (far ref); lane-mask pairs of the triangle test s[54:55] (accepting lanes),
// s[56:57], s[58:59], s[64:65], s[66:67]; of the 128-ray node step s[36:37] / s[38:39] (group A hit left / right),
// s[40:41] / s[42:43] (group B), s[44:45] (any lane hit left), s[46:47] (left-is-nearer flags),
// s[50:51] / s[58:59] (the pushed child's masks)
// v50..v59 temporaries, v60 the never-read destination of the far-child prefetches.
#define RA_LMINX "s20"
#define RA_LMINY "s21"
#define RA_LMINZ "s22"
#define RA_LREF "s23"
#define RA_LMAXX "s24"
#define RA_LMAXY "s25"
#define RA_LMAXZ "s26"
#define RA_RREF "s27"
#define RA_RMINX "s28"
#define RA_RMINY "s29"
#define RA_RMINZ "s30"
#define RA_RMAXX "s32"
#define RA_RMAXY "s33"
#define RA_RMAXZ "s34"
#define RA_MASK "s[60:61]"
#define RA_MASKLO "s60"
#define RA_MASKHI "s61"
// the same registers read as a triangle row
#define RA_V0X "s20"
#define RA_V0Y "s21"
#define RA_V0Z "s22"
#define RA_ID "s23"
#define RA_E1X "s24"
#define RA_E1Y "s25"
#define RA_E1Z "s26"
#define RA_LAYERS "s27"
#define RA_E2X "s28"
#define RA_E2Y "s29"
#define RA_E2Z "s30"
#define RA_FLAGS "s31"

// near / far plane of an axis by the sign bit of the octant (0: inv >= 0 -> near = min, far = max; 1: swapped)
#define MRT_NEAR_0(RS, AX, SIDE) RS##_##SIDE##MIN##AX
#define MRT_NEAR_1(RS, AX, SIDE) RS##_##SIDE##MAX##AX
#define MRT_FAR_0(RS, AX, SIDE) RS##_##SIDE##MAX##AX
#define MRT_FAR_1(RS, AX, SIDE) RS##_##SIDE##MIN##AX

// operand of packet P ("A" / "B"): %[oxA] ...
#define MRT_OP(NAME, P) "%[" NAME P "]"

// slab test of one 64-ray group against both child boxes of the node row (ray_aabb, glsl:84-99, octant-specialised:
// near / far plane of each axis known from the sign of the direction).  v50 = tl, v53 = tlx (entry / exit of the left
// box clamped to [t_min, lim]), v56 = tr, v52 = trx.  Ordered so that no instruction reads its predecessor's result.
#define MRT_ROWS_SLAB(P, RS, BX, BY, BZ)                                                                             \
	"v_fma_f32 v50, " MRT_NEAR_##BX(RS, X, L) ", " MRT_OP("ix", P) ", " MRT_OP("nrx", P) "\n"                        \
	"v_fma_f32 v51, " MRT_NEAR_##BY(RS, Y, L) ", " MRT_OP("iy", P) ", " MRT_OP("nry", P) "\n"                        \
	"v_fma_f32 v52, " MRT_NEAR_##BZ(RS, Z, L) ", " MRT_OP("iz", P) ", " MRT_OP("nrz", P) "\n"                        \
	"v_fma_f32 v53, " MRT_FAR_##BX(RS, X, L) ", " MRT_OP("ix", P) ", " MRT_OP("nrx", P) "\n"                         \
	"v_fma_f32 v54, " MRT_FAR_##BY(RS, Y, L) ", " MRT_OP("iy", P) ", " MRT_OP("nry", P) "\n"                         \
	"v_fma_f32 v55, " MRT_FAR_##BZ(RS, Z, L) ", " MRT_OP("iz", P) ", " MRT_OP("nrz", P) "\n"                         \
	"v_max_f32 v52, v52, " MRT_OP("tmin", P) "\n"                                                                   \
	"v_fma_f32 v56, " MRT_NEAR_##BX(RS, X, R) ", " MRT_OP("ix", P) ", " MRT_OP("nrx", P) "\n"                        \
	"v_min_f32 v55, v55, " MRT_OP("lim", P) "\n"                                                                    \
	"v_fma_f32 v57, " MRT_NEAR_##BY(RS, Y, R) ", " MRT_OP("iy", P) ", " MRT_OP("nry", P) "\n"                        \
	"v_max3_f32 v50, v50, v51, v52\n"        /* v50 = tl  */                                                       \
	"v_fma_f32 v51, " MRT_NEAR_##BZ(RS, Z, R) ", " MRT_OP("iz", P) ", " MRT_OP("nrz", P) "\n"                        \
	"v_min3_f32 v53, v53, v54, v55\n"        /* v53 = tlx */                                                       \
	"v_fma_f32 v52, " MRT_FAR_##BX(RS, X, R) ", " MRT_OP("ix", P) ", " MRT_OP("nrx", P) "\n"                         \
	"v_fma_f32 v54, " MRT_FAR_##BY(RS, Y, R) ", " MRT_OP("iy", P) ", " MRT_OP("nry", P) "\n"                         \
	"v_fma_f32 v55, " MRT_FAR_##BZ(RS, Z, R) ", " MRT_OP("iz", P) ", " MRT_OP("nrz", P) "\n"                         \
	"v_max_f32 v51, v51, " MRT_OP("tmin", P) "\n"                                                                   \
	"v_min_f32 v55, v55, " MRT_OP("lim", P) "\n"                                                                    \
	"v_max3_f32 v56, v56, v57, v51\n"        /* v56 = tr  */                                                       \
	"v_min3_f32 v52, v52, v54, v55\n"        /* v52 = trx */

// ---- one node step of a 64-ray packet (register set RS, octant bits BX BY BZ) -------------------------------------
// Ends with cur = the child to visit next (an inner node, or a leaf = its first triangle row, bit 31 set) and the
// mask of the lanes whose own ray hit that child in the packet's mask register; or with the packet's pop bit set
// in %[ev].  A pushed entry is 16 bytes {ref, -, lane mask}.
#define MRT_ROWS_NODE_STEP(P, RS, POPBIT, CNT_P, BX, BY, BZ)                                                                 \
	MRT_ROWS_SLAB(P, RS, BX, BY, BZ)                                                                                \
	"v_cmp_le_f32 vcc, v50, v53\n"           /* lanes that hit the left child  */                                  \
	"v_cmp_le_f32_e64 s[54:55], v56, v52\n"  /* lanes that hit the right child */                                  \
	"s_cbranch_vccz L_" P "lmiss_%=\n"                                                                              \
	"s_cmp_eq_u64 s[54:55], 0\n"                                                                                    \
	"s_cbranch_scc1 L_" P "onlyl_%=\n"                                                                              \
	"v_cmp_lt_f32_e64 s[56:57], v50, v56\n"  /* both hit: lane 0 decides which is nearer (order = speed only) */   \
	"s_bitcmp1_b32 s56, 0\n"                                                                                        \
	"s_cselect_b32 s53, " RS##_RREF ", " RS##_LREF "\n"                  /* far  */                                  \
	"s_cselect_b32 " MRT_OP("cur", P) ", " RS##_LREF ", " RS##_RREF "\n" /* near */                                  \
	"s_cselect_b64 s[58:59], s[54:55], vcc\n"        /* lanes that hit the far child  */                           \
	"s_cselect_b64 " RS##_MASK ", vcc, s[54:55]\n"    /* lanes that hit the near child */                           \
	"v_mov_b32 v51, s53\n"                                                                                          \
	"v_mov_b32 v52, s58\n"                                                                                          \
	"v_mov_b32 v53, s59\n"                                                                                          \
	"ds_write_b32 " MRT_OP("sp", P) ", v51\n"                                                                       \
	"ds_write_b64 " MRT_OP("sp", P) ", v[52:53] offset:8\n"                                                         \
	"v_add_u32 " MRT_OP("sp", P) ", 16, " MRT_OP("sp", P) "\n"                                                      \
	CNT_P                                                                                                           \
	"v_lshlrev_b32 v54, 6, v51\n"            /* pull the pushed row (node, or the leaf's first triangle) towards */ \
	"global_load_dword v60, v54, %[rows]\n"  /* the L2 now; v60 is never read (vmcnt drained at the very end)    */ \
	"s_branch L_" P "done_%=\n"                                                                                     \
	"L_" P "onlyl_%=:\n"                                                                                            \
	"s_mov_b32 " MRT_OP("cur", P) ", " RS##_LREF "\n"                                                                \
	"s_mov_b64 " RS##_MASK ", vcc\n"                                                                                 \
	"s_branch L_" P "done_%=\n"                                                                                     \
	"L_" P "lmiss_%=:\n"                                                                                            \
	"s_cmp_eq_u64 s[54:55], 0\n"                                                                                    \
	"s_cbranch_scc1 L_" P "npop_%=\n"                                                                               \
	"s_mov_b32 " MRT_OP("cur", P) ", " RS##_RREF "\n"                                                                \
	"s_mov_b64 " RS##_MASK ", s[54:55]\n"                                                                            \
	"s_branch L_" P "done_%=\n"                                                                                     \
	"L_" P "npop_%=:\n"                                                                                             \
	"s_or_b32 %[ev], %[ev], " POPBIT "\n"

// ---- one triangle step of packet P: ray_triangle, glsl:105-131 == Triangle::intersect, src/core/triangle.h:56-105,
// in the operation order of packet_leaf() (packet_asm_kernel.h).  The lanes that own the leaf (the packet's mask
// register) test against lim (best_t, or -FLT_MAX for a lane that takes no part); an exact tie goes to the lower id.
// P: the group (operand names, labels); RS: the row's registers; OWNMASK: the lanes of the group that own the leaf;
// CUROP: the operand holding the row ref.  ANYHIT_CODE: what an accepting lane's limit becomes (MRT_ROWS_NEAREST /
// MRT_ROWS_ANYHIT / MRT_ROWSW_ANYHIT).  Falls out at label L_<P>tnext.
#define MRT_ROWS_TRI_TEST(P, RS, OWNMASK, CUROP, ANYHIT_CODE)                                                        \
	"s_and_b32 s52, " RS##_LAYERS ", %[qmask]\n"  /* Triangle layers & query mask (wave-uniform) */                  \
	"s_cbranch_scc0 L_" P "tnext_%=\n"                                                                              \
	"v_mul_f32 v50, " RS##_E2Y ", " MRT_OP("dz", P) "\n"                                                             \
	"v_mul_f32 v51, " RS##_E2Z ", " MRT_OP("dx", P) "\n"                                                             \
	"v_mul_f32 v52, " RS##_E2X ", " MRT_OP("dy", P) "\n"                                                             \
	"v_fma_f32 v50, " MRT_OP("dy", P) ", " RS##_E2Z ", -v50\n"   /* pvx = dy e2z - dz e2y */                         \
	"v_fma_f32 v51, " MRT_OP("dz", P) ", " RS##_E2X ", -v51\n"   /* pvy = dz e2x - dx e2z */                         \
	"v_fma_f32 v52, " MRT_OP("dx", P) ", " RS##_E2Y ", -v52\n"   /* pvz = dx e2y - dy e2x */                         \
	"v_mul_f32 v53, " RS##_E1Z ", v52\n"                                                                             \
	"v_fma_f32 v53, " RS##_E1Y ", v51, v53\n"                                                                        \
	"v_fma_f32 v53, " RS##_E1X ", v50, v53\n"                    /* det = e1 . pv */                                 \
	"v_cmp_nlt_f32_e64 s[54:55], |v53|, %[eps]\n"               /* !(|det| < 1e-8) */                               \
	"s_and_b64 s[54:55], s[54:55], " OWNMASK "\n"               /* ... among the lanes that own this leaf */        \
	"s_cbranch_scc0 L_" P "tnext_%=\n"                                                                              \
	"v_div_scale_f32 v54, s[56:57], v53, v53, 1.0\n"            /* inv_det = 1.0f / det, correctly rounded */       \
	"v_rcp_f32 v55, v54\n"                                                                                          \
	"v_div_scale_f32 v56, vcc, 1.0, v53, 1.0\n"                                                                     \
	"v_subrev_f32 v59, " RS##_V0X ", " MRT_OP("ox", P) "\n"      /* tvx = ox - v0x (also keeps v_rcp's result apart from its first use) */ \
	"v_fma_f32 v57, -v54, v55, 1.0\n"                                                                               \
	"v_fma_f32 v55, v57, v55, v55\n"                                                                                \
	"v_mul_f32 v57, v56, v55\n"                                                                                     \
	"v_fma_f32 v58, -v54, v57, v56\n"                                                                               \
	"v_fma_f32 v57, v58, v55, v57\n"                                                                                \
	"v_fma_f32 v54, -v54, v57, v56\n"                                                                               \
	"v_div_fmas_f32 v54, v54, v55, v57\n"                                                                           \
	"v_div_fixup_f32 v54, v54, v53, 1.0\n"                      /* v54 = inv_det */                                 \
	"v_subrev_f32 v55, " RS##_V0Y ", " MRT_OP("oy", P) "\n"      /* tvy */                                           \
	"v_subrev_f32 v56, " RS##_V0Z ", " MRT_OP("oz", P) "\n"      /* tvz */                                           \
	"v_mul_f32 v58, v56, v52\n"                                                                                     \
	"v_fma_f32 v58, v55, v51, v58\n"                                                                                \
	"v_fma_f32 v58, v59, v50, v58\n"                                                                                \
	"v_mul_f32 v58, v58, v54\n"                                 /* v58 = u = (tv . pv) inv_det */                   \
	"v_cmp_nlt_f32_e64 s[56:57], v58, 0\n"                      /* !(u < 0) */                                      \
	"v_cmp_ngt_f32_e64 s[58:59], v58, 1.0\n"                    /* !(u > 1) */                                      \
	"s_and_b64 s[54:55], s[54:55], s[56:57]\n"                                                                      \
	"s_and_b64 s[54:55], s[54:55], s[58:59]\n"                                                                      \
	"s_cbranch_scc0 L_" P "tnext_%=\n"                                                                              \
	"v_mul_f32 v50, " RS##_E1Y ", v56\n"                                                                             \
	"v_mul_f32 v51, " RS##_E1Z ", v59\n"                                                                             \
	"v_mul_f32 v52, " RS##_E1X ", v55\n"                                                                             \
	"v_fma_f32 v50, v55, " RS##_E1Z ", -v50\n"                   /* qvx = tvy e1z - tvz e1y */                       \
	"v_fma_f32 v51, v56, " RS##_E1X ", -v51\n"                   /* qvy = tvz e1x - tvx e1z */                       \
	"v_fma_f32 v52, v59, " RS##_E1Y ", -v52\n"                   /* qvz = tvx e1y - tvy e1x */                       \
	"v_mul_f32 v59, " MRT_OP("dz", P) ", v52\n"                                                                     \
	"v_fma_f32 v59, " MRT_OP("dy", P) ", v51, v59\n"                                                                \
	"v_fma_f32 v59, " MRT_OP("dx", P) ", v50, v59\n"                                                                \
	"v_mul_f32 v59, v59, v54\n"                                 /* v59 = v = (d . qv) inv_det */                    \
	"v_add_f32 v55, v58, v59\n"                                 /* u + v */                                         \
	"v_cmp_nlt_f32_e64 s[56:57], v59, 0\n"                      /* !(v < 0) */                                      \
	"v_cmp_ngt_f32_e64 s[58:59], v55, 1.0\n"                    /* !(u + v > 1) */                                  \
	"s_and_b64 s[54:55], s[54:55], s[56:57]\n"                                                                      \
	"s_and_b64 s[54:55], s[54:55], s[58:59]\n"                                                                      \
	"s_cbranch_scc0 L_" P "tnext_%=\n"                                                                              \
	"v_mul_f32 v55, " RS##_E2Z ", v52\n"                                                                             \
	"v_fma_f32 v55, " RS##_E2Y ", v51, v55\n"                                                                        \
	"v_fma_f32 v55, " RS##_E2X ", v50, v55\n"                                                                        \
	"v_mul_f32 v55, v55, v54\n"                                 /* v55 = t = (e2 . qv) inv_det */                   \
	/* accept: !(t < t_min) && (t < lim || (t == lim && a hit is held && id < its id)) */                          \
	"v_cmp_nlt_f32_e64 s[56:57], v55, " MRT_OP("tmin", P) "\n"                                                      \
	"v_cmp_lt_f32_e64 s[58:59], v55, " MRT_OP("lim", P) "\n"                                                        \
	"v_cmp_eq_f32_e64 s[64:65], v55, " MRT_OP("lim", P) "\n"                                                        \
	"v_cmp_ne_u32_e64 s[66:67], " MRT_OP("bs", P) ", -1\n"                                                          \
	"v_cmp_gt_u32_e64 vcc, " MRT_OP("bi", P) ", " RS##_ID "\n"   /* id < best_id */                                  \
	"s_and_b64 s[64:65], s[64:65], s[66:67]\n"                                                                      \
	"s_and_b64 s[64:65], s[64:65], vcc\n"                                                                           \
	"s_or_b64 s[58:59], s[58:59], s[64:65]\n"                                                                       \
	"s_and_b64 s[54:55], s[54:55], s[56:57]\n"                                                                      \
	"s_and_b64 s[54:55], s[54:55], s[58:59]\n"                  /* the lanes that take this hit */                  \
	"s_cbranch_scc0 L_" P "tnext_%=\n"                                                                              \
	"s_and_b32 s52, " CUROP ", 0x7fffffff\n"                    /* the triangle's row */                            \
	"v_mov_b32 v50, s52\n"                                                                                          \
	"v_mov_b32 v51, " RS##_ID "\n"                                                                                   \
	"v_cndmask_b32_e64 " MRT_OP("bt", P) ", " MRT_OP("bt", P) ", v55, s[54:55]\n"                                   \
	"v_cndmask_b32_e64 " MRT_OP("bu", P) ", " MRT_OP("bu", P) ", v58, s[54:55]\n"                                   \
	"v_cndmask_b32_e64 " MRT_OP("bv", P) ", " MRT_OP("bv", P) ", v59, s[54:55]\n"                                   \
	"v_cndmask_b32_e64 " MRT_OP("bs", P) ", " MRT_OP("bs", P) ", v50, s[54:55]\n"                                   \
	"v_cndmask_b32_e64 " MRT_OP("bi", P) ", " MRT_OP("bi", P) ", v51, s[54:55]\n"                                   \
	ANYHIT_CODE                                                                                                     \
	"L_" P "tnext_%=:\n"

// a triangle step of a 64-ray packet: the test, then the next triangle of the leaf or (last one) the pop bit in %[ev]
#define MRT_ROWS_TRI_STEP(P, RS, POPBIT, CNT_T, ANYHIT_CODE)                                                         \
	CNT_T                                                                                                           \
	MRT_ROWS_TRI_TEST(P, RS, RS##_MASK, MRT_OP("cur", P), ANYHIT_CODE)                                              \
	"s_bitcmp1_b32 " RS##_FLAGS ", 0\n"                          /* the last triangle of its leaf? */                \
	"s_cbranch_scc1 L_" P "tpop_%=\n"                                                                               \
	"s_add_u32 " MRT_OP("cur", P) ", " MRT_OP("cur", P) ", 1\n"                                                     \
	"s_branch L_" P "done_%=\n"                                                                                     \
	"L_" P "tpop_%=:\n"                                                                                             \
	"s_or_b32 %[ev], %[ev], " POPBIT "\n"

// closest hit: the new limit of an accepting lane is its t
#define MRT_ROWS_NEAREST(P) \
	"v_cndmask_b32_e64 " MRT_OP("lim", P) ", " MRT_OP("lim", P) ", v55, s[54:55]\n"
// any hit: an accepting lane is finished; when no lane of the packet is left the packet ends (sentinel, via the slow path)
#define MRT_ROWS_ANYHIT(P, DONEBIT) \
	"v_cndmask_b32_e64 " MRT_OP("lim", P) ", " MRT_OP("lim", P) ", %[vneg], s[54:55]\n"                             \
	"v_cmp_neq_f32_e64 s[56:57], " MRT_OP("lim", P) ", %[vneg]\n"                                                   \
	"s_cmp_lg_u64 s[56:57], 0\n"                                                                                    \
	"s_cbranch_scc1 L_" P "tnext_%=\n"                                                                              \
	"s_mov_b32 " MRT_OP("cur", P) ", 0x7fffffff\n"                                                                  \
	"s_or_b32 %[ev], %[ev], " DONEBIT "\n"                                                                          \
	"s_branch L_" P "done_%=\n"

// a step of packet P, whatever its current row is
#define MRT_ROWS_STEP(P, RS, POPBIT, CNT_N, CNT_T, CNT_P, ANYHIT_CODE, BX, BY, BZ)                                            \
	"s_bitcmp1_b32 " MRT_OP("cur", P) ", 31\n"                                                                      \
	"s_cbranch_scc1 L_" P "tri_%=\n"                                                                                \
	CNT_N                                                                                                           \
	MRT_ROWS_NODE_STEP(P, RS, POPBIT, CNT_P, BX, BY, BZ)                                                             \
	"s_branch L_" P "done_%=\n"                                                                                     \
	"L_" P "tri_%=:\n"                                                                                              \
	MRT_ROWS_TRI_STEP(P, RS, POPBIT, CNT_T, ANYHIT_CODE)                                                             \
	"L_" P "done_%=:\n"

// pop of packet P (slow path): the entry {ref, -, mask} into V0..V3, then cur and the mask register
#define MRT_ROWS_POP_ISSUE(P, POPBITNO, V0123)                                                                      \
	"s_bitcmp1_b32 %[ev], " POPBITNO "\n"                                                                           \
	"s_cbranch_scc0 L_" P "pi_%=\n"                                                                                 \
	"v_add_u32 " MRT_OP("sp", P) ", -16, " MRT_OP("sp", P) "\n"                                                     \
	"ds_read_b128 " V0123 ", " MRT_OP("sp", P) "\n"                                                                 \
	"L_" P "pi_%=:\n"
#define MRT_ROWS_POP_TAKE(P, RS, POPBITNO, V0, V2, V3)                                                               \
	"s_bitcmp1_b32 %[ev], " POPBITNO "\n"                                                                           \
	"s_cbranch_scc0 L_" P "pt_%=\n"                                                                                 \
	"v_readfirstlane_b32 " MRT_OP("cur", P) ", " V0 "\n"                                                            \
	"v_readfirstlane_b32 " RS##_MASKLO ", " V2 "\n"                                                                  \
	"v_readfirstlane_b32 " RS##_MASKHI ", " V3 "\n"                                                                  \
	"L_" P "pt_%=:\n"

#define MRT_ROWS_CLOBBERS                                                                                             \
	"vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", \
	"s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", \
	"s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", \
	"s66", "s67", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60"

// operands of one packet (S = its PacketRegs)
#define MRT_ROWS_OUT(P, S)                                                                                            \
	[cur##P] "+s"(S.cur), [sp##P] "+v"(S.sp), [lim##P] "+v"(S.lim), [bt##P] "+v"(S.bt), [bu##P] "+v"(S.bu),         \
	[bv##P] "+v"(S.bv), [bs##P] "+v"(S.bs), [bi##P] "+v"(S.bi), [mask##P] "+s"(S.mask)
#define MRT_ROWS_IN(P, S)                                                                                             \
	[ox##P] "v"(S.ox), [oy##P] "v"(S.oy), [oz##P] "v"(S.oz), [dx##P] "v"(S.dx), [dy##P] "v"(S.dy), [dz##P] "v"(S.dz), \
	[tmin##P] "v"(S.tmin), [ix##P] "v"(S.ix), [iy##P] "v"(S.iy), [iz##P] "v"(S.iz), [nrx##P] "v"(S.nrx),            \
	[nry##P] "v"(S.nry), [nrz##P] "v"(S.nrz)

// everything a packet's walk holds in registers
struct PacketRegs {
	float ox, oy, oz, dx, dy, dz, tmin;   // the ray (t_max lives on in lim / bt)
	float ix, iy, iz, nrx, nry, nrz;      // safe_inv(d), -(o * inv)
	float lim;                            // far limit of the box tests: best_t, or -FLT_MAX for a lane that takes no part
	float bt, bu, bv;                     // best hit
	uint32_t bs, bi;                      // its row (0xFFFFFFFF = none) and triangle id
	uint32_t sp;                          // LDS byte address of the next free stack entry
	uint32_t cur;                         // wave-uniform: the row to visit next (bit 31: a triangle), 0x7FFFFFFF = finished
	unsigned long long mask;              // wave-uniform: the lanes that own the current row (matters for triangles)
};

// ---- the loops ------------------------------------------------------------------------------------------------
// One packet (register set A).  In: cur = a row to visit.  Out: cur = 0x7FFFFFFF.
#define MRT_ROWS_LOOP1(CNT_N, CNT_T, CNT_P, T_PRE, T_POST, ANYHIT_CODE, BX, BY, BZ)                                          \
	asm volatile(                                                                                                   \
		"s_mov_b64 s[60:61], %[maskA]\n"                                                                            \
		"L_loop_%=:\n"                                                                                              \
		"s_lshl_b32 s52, %[curA], 6\n"                                                                              \
		T_PRE                                                                                                       \
		"s_load_dwordx16 s[20:35], %[rows], s52\n"                                                                  \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		T_POST                                                                                                      \
		MRT_ROWS_STEP("A", RA, "1", CNT_N, CNT_T, CNT_P, ANYHIT_CODE, BX, BY, BZ)                                   \
		"s_cmp_eq_u32 %[ev], 0\n"                                                                                   \
		"s_cbranch_scc1 L_loop_%=\n"                                                                                \
		MRT_ROWS_POP_ISSUE("A", "0", "v[50:53]")                                                                    \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		MRT_ROWS_POP_TAKE("A", RA, "0", "v50", "v52", "v53")                                                        \
		"s_mov_b32 %[ev], 0\n"                                                                                      \
		"s_cmp_lg_u32 %[curA], 0x7fffffff\n"                                                                        \
		"s_cbranch_scc1 L_loop_%=\n"                                                                                \
		"s_waitcnt vmcnt(0)\n"               /* no prefetch may land in v60 once the compiler owns it again */     \
		"s_mov_b64 %[maskA], s[60:61]\n"                                                                            \
		: MRT_ROWS_OUT(A, a), [ev] "+s"(ev), [cntn] "+s"(cnt_n), [cntt] "+s"(cnt_t), [cntw] "+s"(cnt_w), [cntp] "+s"(sp_max) \
		: MRT_ROWS_IN(A, a), [rows] "s"(rows), [qmask] "s"(qmask), [eps] "s"(eps), [vneg] "v"(vneg)                 \
		: MRT_ROWS_CLOBBERS)

// ---- the 128-ray packet: groups A and B (two rays per lane) share one walk --------------------------------------
// One group's slab test at a node row, skipped (no child hit) when no lane of the group owns the node.
// LM / RM: the group's lanes that hit the left / right child.  v50 / v56 keep tl / tr of the last group tested.
#define MRT_ROWSW_GROUP_SLAB(P, OWN, LM, RM, BX, BY, BZ)                                                              \
	"s_cmp_eq_u64 " OWN ", 0\n"                                                                                     \
	"s_cbranch_scc1 L_" P "skip_%=\n"         /* (out of line: the common path falls through) */                    \
	MRT_ROWS_SLAB(P, RA, BX, BY, BZ)                                                                                \
	"v_cmp_le_f32_e64 " LM ", v50, v53\n"                                                                           \
	"v_cmp_le_f32_e64 " RM ", v56, v52\n"                                                                           \
	"L_" P "sdone_%=:\n"
#define MRT_ROWSW_GROUP_SKIP(P, LM, RM)                                                                               \
	"L_" P "skip_%=:\n"                                                                                             \
	"s_mov_b64 " LM ", 0\n"                                                                                         \
	"s_mov_b64 " RM ", 0\n"                                                                                         \
	"s_branch L_" P "sdone_%=\n"

// any hit in the 128-ray walk: an accepting lane is finished; the walk ends when no lane of either group is left
#define MRT_ROWSW_ANYHIT(P) \
	"v_cndmask_b32_e64 " MRT_OP("lim", P) ", " MRT_OP("lim", P) ", %[vneg], s[54:55]\n"
#define MRT_ROWSW_ANYDONE                                                                                             \
	"v_cmp_neq_f32_e64 s[56:57], %[limA], %[vneg]\n"                                                                \
	"v_cmp_neq_f32_e64 s[58:59], %[limB], %[vneg]\n"                                                                \
	"s_or_b64 s[56:57], s[56:57], s[58:59]\n"                                                                       \
	"s_cbranch_scc1 L_goon_%=\n"                                                                                    \
	"s_mov_b32 %[curA], 0x7fffffff\n"                                                                               \
	"s_branch L_exit_%=\n"                                                                                          \
	"L_goon_%=:\n"

// Both child rows are pulled into the scalar cache as soon as the node's own row has arrived, before the slab tests:
// the row fetch of the next step (whichever child it visits) then finds its line there instead of waiting for the L2
// (two one-dword scalar loads into the triangle test's temporaries s64 / s65, which nothing reads; they are long back
// when the next step waits on lgkmcnt).  The far-child vector prefetch of round 1 is the alternative (MRT_ROWS_KPREFETCH=0).
#ifndef MRT_ROWS_KPREFETCH
#define MRT_ROWS_KPREFETCH 1
#endif
#if MRT_ROWS_KPREFETCH
#define MRT_ROWSW_KPREFETCH                                                                                           \
	"s_lshl_b32 s48, " RA_LREF ", 6\n"                                                                               \
	"s_lshl_b32 s49, " RA_RREF ", 6\n"                                                                               \
	"s_load_dword s64, %[rows], s48\n"                                                                              \
	"s_load_dword s65, %[rows], s49\n"
#define MRT_ROWSW_FARPREFETCH ""
#else
#define MRT_ROWSW_KPREFETCH ""
#define MRT_ROWSW_FARPREFETCH                                                                                         \
	"v_lshlrev_b32 v55, 6, v54\n"           /* pull the pushed row towards the L2 now */                            \
	"global_load_dword v60, v55, %[rows]\n" /* v60 is never read (vmcnt drained at the very end) */
#endif

// In: cur (group A's operand) = a row to visit, the groups' own masks.  Out: cur = 0x7FFFFFFF.
// Stack entries are 32 bytes: {group A's mask, group B's mask, ref, -}; the sentinel entry has ref 0x7FFFFFFF.
#define MRT_ROWS_LOOPW(CNT_N, CNT_T, CNT_P, ANYA, ANYB, ANYDONE, BX, BY, BZ)                                                 \
	asm volatile(                                                                                                   \
		"s_mov_b64 s[60:61], %[maskA]\n"                                                                            \
		"s_mov_b64 s[62:63], %[maskB]\n"                                                                            \
		"L_loop_%=:\n"                                                                                              \
		"s_lshl_b32 s52, %[curA], 6\n"                                                                              \
		"s_load_dwordx16 s[20:35], %[rows], s52\n"                                                                  \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		"s_bitcmp1_b32 %[curA], 31\n"                                                                               \
		"s_cbranch_scc1 L_tri_%=\n"                                                                                 \
		CNT_N                                                                                                       \
		MRT_ROWSW_KPREFETCH                                                                                         \
		MRT_ROWSW_GROUP_SLAB("A", "s[60:61]", "s[36:37]", "s[38:39]", BX, BY, BZ)                                   \
		MRT_ROWSW_GROUP_SLAB("B", "s[62:63]", "s[40:41]", "s[42:43]", BX, BY, BZ)                                   \
		"s_or_b64 s[44:45], s[36:37], s[40:41]\n"    /* any lane of the 128 hit the left child?  */                 \
		"s_cbranch_scc0 L_lmiss_%=\n"                                                                               \
		"s_or_b64 vcc, s[38:39], s[42:43]\n"         /* ... the right child? */                                     \
		"s_cbranch_scc0 L_onlyl_%=\n"                                                                               \
		/* both: lane 0 of the group tested last (its tl, tr are still in v50, v56) decides which is nearer */      \
		"v_cmp_lt_f32_e64 s[46:47], v50, v56\n"      /* (order = speed only) */                                     \
		"s_bitcmp1_b32 s46, 0\n"                                                                                    \
		"s_cselect_b32 s53, " RA_RREF ", " RA_LREF "\n"       /* far  */                                            \
		"s_cselect_b32 %[curA], " RA_LREF ", " RA_RREF "\n"   /* near */                                            \
		"s_cselect_b64 s[50:51], s[38:39], s[36:37]\n"        /* the far child's masks: group A, group B */         \
		"s_cselect_b64 s[58:59], s[42:43], s[40:41]\n"                                                              \
		"s_cselect_b64 s[60:61], s[36:37], s[38:39]\n"        /* the near child's */                                \
		"s_cselect_b64 s[62:63], s[40:41], s[42:43]\n"                                                              \
		"v_mov_b32 v50, s50\n"                                                                                      \
		"v_mov_b32 v51, s51\n"                                                                                      \
		"v_mov_b32 v52, s58\n"                                                                                      \
		"v_mov_b32 v53, s59\n"                                                                                      \
		"v_mov_b32 v54, s53\n"                                                                                      \
		"ds_write_b128 %[spA], v[50:53]\n"                                                                          \
		"ds_write_b32 %[spA], v54 offset:16\n"                                                                      \
		"v_add_u32 %[spA], 32, %[spA]\n"                                                                            \
		CNT_P                                                                                                       \
		MRT_ROWSW_FARPREFETCH                                                                                       \
		"s_branch L_loop_%=\n"                                                                                      \
		"L_onlyl_%=:\n"                                                                                             \
		"s_mov_b32 %[curA], " RA_LREF "\n"                                                                          \
		"s_mov_b64 s[60:61], s[36:37]\n"                                                                            \
		"s_mov_b64 s[62:63], s[40:41]\n"                                                                            \
		"s_branch L_loop_%=\n"                                                                                      \
		"L_lmiss_%=:\n"                                                                                             \
		"s_or_b64 vcc, s[38:39], s[42:43]\n"                                                                        \
		"s_cbranch_scc0 L_pop_%=\n"                                                                                 \
		"s_mov_b32 %[curA], " RA_RREF "\n"                                                                          \
		"s_mov_b64 s[60:61], s[38:39]\n"                                                                            \
		"s_mov_b64 s[62:63], s[42:43]\n"                                                                            \
		"s_branch L_loop_%=\n"                                                                                      \
		"L_tri_%=:\n"                                                                                               \
		CNT_T                                                                                                       \
		"s_cmp_eq_u64 s[60:61], 0\n"                                                                                \
		"s_cbranch_scc1 L_Atnext_%=\n"                                                                              \
		MRT_ROWS_TRI_TEST("A", RA, "s[60:61]", "%[curA]", ANYA)                                                     \
		"s_cmp_eq_u64 s[62:63], 0\n"                                                                                \
		"s_cbranch_scc1 L_Btnext_%=\n"                                                                              \
		MRT_ROWS_TRI_TEST("B", RA, "s[62:63]", "%[curA]", ANYB)                                                     \
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
		"v_readfirstlane_b32 s60, v50\n"                                                                            \
		"v_readfirstlane_b32 s61, v51\n"                                                                            \
		"v_readfirstlane_b32 s62, v52\n"                                                                            \
		"v_readfirstlane_b32 s63, v53\n"                                                                            \
		"s_cmp_lg_u32 %[curA], 0x7fffffff\n"                                                                        \
		"s_cbranch_scc1 L_loop_%=\n"                                                                                \
		"s_branch L_exit_%=\n"                                                                                      \
		MRT_ROWSW_GROUP_SKIP("A", "s[36:37]", "s[38:39]")                                                           \
		MRT_ROWSW_GROUP_SKIP("B", "s[40:41]", "s[42:43]")                                                           \
		"L_exit_%=:\n"                                                                                              \
		"s_waitcnt vmcnt(0)\n"               /* no prefetch may land in v60 once the compiler owns it again */     \
		"s_mov_b64 %[maskA], s[60:61]\n"                                                                            \
		"s_mov_b64 %[maskB], s[62:63]\n"                                                                            \
		: MRT_ROWS_OUT(A, a), [limB] "+v"(b.lim), [btB] "+v"(b.bt), [buB] "+v"(b.bu), [bvB] "+v"(b.bv),             \
		  [bsB] "+v"(b.bs), [biB] "+v"(b.bi), [maskB] "+s"(b.mask), [cntn] "+s"(cnt_n), [cntt] "+s"(cnt_t), [cntp] "+s"(sp_max) \
		: MRT_ROWS_IN(A, a), MRT_ROWS_IN(B, b), [rows] "s"(rows), [qmask] "s"(qmask), [eps] "s"(eps), [vneg] "v"(vneg) \
		: MRT_ROWS_CLOBBERS)


// ---- the 128-ray walk with packet-level frustum culling ---------------------------------------------------------
// The ownership model (tools/sim_ownership.py, profiles/r03_ownership_sim_c3.json): of the (group, child box) slab
// tests a C3 packet executes, 46 % fail for all 64 rays of the group, and 87 % of those boxes lie wholly outside the
// pyramid the packet's 128 rays span: 40 % of the vector work of the walk buys nothing.  A box outside one side plane of
// the pyramid is one no ray of the packet reaches; both groups' tests of that child are skipped (masks zero).
// Round 2's form of the test (three per-lane loads per step, waited for on the spot) cost more than it saved: the
// kernel ran 8 % slower.  This form:
//   * ONE vector load per row brings the row's 16 dwords to lanes 0..15 of every 16-lane row of the wave (lane l reads
//     dword l & 15: the same 64-byte line for all four rows), so the four rows of the wave see the node's two child
//     boxes side by side: row r of the wave stands for side plane r of the pyramid;
//   * a lane multiplies its coordinate by its weight — the plane normal's component of that axis if the coordinate is
//     the one of the box's corner farthest along the normal, else 0 — and three DPP row_shr adds leave
//     dot(n', far corner) of the left child in lane 7 and of the right child in lane 15 of each row; one v_cmp against
//     the lane's constant cc = dot(n', apex) - eps (-inf in the other lanes) gives the cull bits: 6 vector
//     instructions per node step against 11 per (group, box) test skipped;
//   * the loads are issued ONE STEP AHEAD: when a node's row has arrived (scalar path) the vector copies of BOTH child
//     rows are requested, and waited for at the top of the next step, after this step's slab tests; a popped row's
//     copy is requested at the pop, beside its scalar fetch.  (The same loads pull both child rows towards the L2 for
//     the scalar fetch that follows: the far-child prefetch of MRT_ROWS_LOOPW is subsumed.)
// Soundness: see cull_setup().  v60 / v61: the vector copies of the left / right child row (or, after a pop, v60 of the
// popped row); s[66:67]: all ones when the current row's copy is the one in v61.  The weighted sum adds the products
// of eight lanes, five of them with weight 0: the row's ref / count dwords (integers < 2^26 or leaf-flagged: denormal
// bit patterns) times 0 are 0; a NaN or infinite coordinate (the padding box of a root leaf) makes the sum NaN and
// the compare false: not culled.
// One child box against one group (the arithmetic of MRT_ROWS_SLAB, one box at a time): TE = entry distance
// (v50 for the left child, v56 for the right one: what the near / far decision reads), MASK = lanes that hit
#define MRT_ROWS_SLAB1(P, RS, SIDE, TE, BX, BY, BZ, MASK)                                                             \
	"v_fma_f32 v51, " MRT_NEAR_##BX(RS, X, SIDE) ", " MRT_OP("ix", P) ", " MRT_OP("nrx", P) "\n"                      \
	"v_fma_f32 v52, " MRT_NEAR_##BY(RS, Y, SIDE) ", " MRT_OP("iy", P) ", " MRT_OP("nry", P) "\n"                      \
	"v_fma_f32 v53, " MRT_NEAR_##BZ(RS, Z, SIDE) ", " MRT_OP("iz", P) ", " MRT_OP("nrz", P) "\n"                      \
	"v_fma_f32 v54, " MRT_FAR_##BX(RS, X, SIDE) ", " MRT_OP("ix", P) ", " MRT_OP("nrx", P) "\n"                       \
	"v_fma_f32 v55, " MRT_FAR_##BY(RS, Y, SIDE) ", " MRT_OP("iy", P) ", " MRT_OP("nry", P) "\n"                       \
	"v_fma_f32 v57, " MRT_FAR_##BZ(RS, Z, SIDE) ", " MRT_OP("iz", P) ", " MRT_OP("nrz", P) "\n"                       \
	"v_max_f32 v53, v53, " MRT_OP("tmin", P) "\n"                                                                   \
	"v_min_f32 v57, v57, " MRT_OP("lim", P) "\n"                                                                    \
	"v_max3_f32 " TE ", v51, v52, v53\n"                                                                            \
	"v_min3_f32 v54, v54, v55, v57\n"                                                                               \
	"v_cmp_le_f32_e64 " MASK ", " TE ", v54\n"

// one child (SIDE = L / R, TE its entry register, CULLBITS the bits of the cull word that speak for it): both groups
#define MRT_ROWSC_CHILD(SIDE, TE, CULLBITS, MA, MB, BX, BY, BZ)                                                       \
	"s_and_b32 s48, s44, " CULLBITS "\n"          /* some plane has the whole box outside? */                       \
	"s_cbranch_scc1 L_cull" #SIDE "_%=\n"                                                                           \
	"s_cmp_eq_u64 s[60:61], 0\n"                                                                                    \
	"s_cbranch_scc1 L_" #SIDE "A0_%=\n"                                                                             \
	MRT_ROWS_SLAB1("A", RA, SIDE, TE, BX, BY, BZ, MA)                                                               \
	"L_" #SIDE "A1_%=:\n"                                                                                           \
	"s_cmp_eq_u64 s[62:63], 0\n"                                                                                    \
	"s_cbranch_scc1 L_" #SIDE "B0_%=\n"                                                                             \
	MRT_ROWS_SLAB1("B", RA, SIDE, TE, BX, BY, BZ, MB)                                                               \
	"L_" #SIDE "B1_%=:\n"
#define MRT_ROWSC_CHILD_OOL(SIDE, MA, MB)                                                                             \
	"L_cull" #SIDE "_%=:\n"                                                                                         \
	MRT_ROWSC_CNT_CULL                                                                                              \
	"s_mov_b64 " MA ", 0\n"                                                                                         \
	"s_mov_b64 " MB ", 0\n"                                                                                         \
	"s_branch L_" #SIDE "B1_%=\n"                                                                                   \
	"L_" #SIDE "A0_%=:\n"                                                                                           \
	"s_mov_b64 " MA ", 0\n"                                                                                         \
	"s_branch L_" #SIDE "A1_%=\n"                                                                                   \
	"L_" #SIDE "B0_%=:\n"                                                                                           \
	"s_mov_b64 " MB ", 0\n"                                                                                         \
	"s_branch L_" #SIDE "B1_%=\n"
#define MRT_ROWSC_CNT_CULL ""

// request the vector copy of row REF (an SGPR) into VDST: lane l gets dword l & 15 of the row.  One vector instruction for
// the address (the scalar unit is the busier one in this loop: three scalar instructions per request made it the bound).
#define MRT_ROWSC_VLOAD(VDST, REF)                                                                                   \
	"v_lshl_add_u32 v62, " REF ", 6, %[cvo]\n"                                                                      \
	"global_load_dword " VDST ", v62, %[rows]\n"

// (optional, -DMRT_ROWSC_KPF=1: also the scalar-cache prefetch of MRT_ROWS_LOOPW; the vector copies already pull both rows to the L2)
#ifndef MRT_ROWSC_KPF
#define MRT_ROWSC_KPF 0
#endif
#if MRT_ROWSC_KPF
#define MRT_ROWSC_KPREFETCH                                                                                           \
	"s_lshl_b32 s46, " RA_LREF ", 6\n"                                                                               \
	"s_lshl_b32 s47, " RA_RREF ", 6\n"                                                                               \
	"s_load_dword s64, %[rows], s46\n"                                                                              \
	"s_load_dword s65, %[rows], s47\n"
#else
#define MRT_ROWSC_KPREFETCH ""
#endif
// In: cur (group A's operand) = a row to visit, the groups' own masks, the lane's cull constants.  Out: cur = 0x7FFFFFFF.
// Stack entries as in MRT_ROWS_LOOPW.
#define MRT_ROWS_LOOPC(CNT_N, CNT_T, CNT_P, ANYA, ANYB, ANYDONE, BX, BY, BZ)                                          \
	asm volatile(                                                                                                   \
		"s_mov_b64 s[60:61], %[maskA]\n"                                                                            \
		"s_mov_b64 s[62:63], %[maskB]\n"                                                                            \
		"L_vload_%=:\n"                           /* the current row's vector copy is not on its way yet */          \
		MRT_ROWSC_VLOAD("v60", "%[curA]")                                                                           \
		"s_mov_b64 s[66:67], 0\n"                                                                                   \
		"L_loop_%=:\n"                                                                                              \
		"s_lshl_b32 s52, %[curA], 6\n"                                                                              \
		"s_load_dwordx16 s[20:35], %[rows], s52\n"                                                                  \
		"s_waitcnt lgkmcnt(0)\n"                                                                                    \
		"s_bitcmp1_b32 %[curA], 31\n"                                                                               \
		"s_cbranch_scc1 L_tri_%=\n"                                                                                 \
		CNT_N                                                                                                       \
		"s_waitcnt vmcnt(0)\n"                    /* the row's vector copy (requested a step ago, or at the pop) */  \
		"v_cndmask_b32_e64 v58, v60, v61, s[66:67]\n"                                                               \
		"v_mul_f32 v58, v58, %[cw]\n"                                                                               \
		MRT_ROWSC_VLOAD("v60", RA_LREF)           /* both children's copies for the next step (also 2+ wait states */ \
		"v_add_f32_dpp v59, v58, v58 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" /* before each DPP read) */ \
		MRT_ROWSC_VLOAD("v61", RA_RREF)                                                                             \
		"v_add_f32_dpp v58, v59, v59 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                           \
		"s_nop 1\n"                                                                                                 \
		"v_add_f32_dpp v59, v58, v58 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                           \
		"v_cmp_lt_f32_e64 s[44:45], v59, %[ccc]\n" /* bit 16 r + 7: the left box is outside plane r; 16 r + 15: the right one */ \
		MRT_ROWSC_KPREFETCH                                                                                         \
		"s_or_b32 s44, s44, s45\n"                                                                                  \
		MRT_ROWSC_CHILD(L, "v50", "0x00800080", "s[36:37]", "s[40:41]", BX, BY, BZ)                                 \
		MRT_ROWSC_CHILD(R, "v56", "0x80008000", "s[38:39]", "s[42:43]", BX, BY, BZ)                                 \
		"s_or_b64 s[44:45], s[36:37], s[40:41]\n"    /* any lane of the 128 hit the left child?  */                 \
		"s_cbranch_scc0 L_lmiss_%=\n"                                                                               \
		"s_or_b64 vcc, s[38:39], s[42:43]\n"         /* ... the right child? */                                     \
		"s_cbranch_scc0 L_onlyl_%=\n"                                                                               \
		/* both: lane 0 of the group tested last (its tl, tr are still in v50, v56) decides which is nearer */      \
		"v_cmp_lt_f32_e64 s[46:47], v50, v56\n"      /* (order = speed only) */                                     \
		"s_bitcmp1_b32 s46, 0\n"                                                                                    \
		"s_cselect_b32 s53, " RA_RREF ", " RA_LREF "\n"       /* far  */                                            \
		"s_cselect_b32 %[curA], " RA_LREF ", " RA_RREF "\n"   /* near */                                            \
		"s_cselect_b64 s[50:51], s[38:39], s[36:37]\n"        /* the far child's masks: group A, group B */         \
		"s_cselect_b64 s[58:59], s[42:43], s[40:41]\n"                                                              \
		"s_cselect_b64 s[60:61], s[36:37], s[38:39]\n"        /* the near child's */                                \
		"s_cselect_b64 s[62:63], s[40:41], s[42:43]\n"                                                              \
		"s_cselect_b64 s[66:67], 0, -1\n"                     /* the near child's vector copy: v60 (left) / v61 */  \
		"v_mov_b32 v50, s50\n"                                                                                      \
		"v_mov_b32 v51, s51\n"                                                                                      \
		"v_mov_b32 v52, s58\n"                                                                                      \
		"v_mov_b32 v53, s59\n"                                                                                      \
		"v_mov_b32 v54, s53\n"                                                                                      \
		"ds_write_b128 %[spA], v[50:53]\n"                                                                          \
		"ds_write_b32 %[spA], v54 offset:16\n"                                                                      \
		"v_add_u32 %[spA], 32, %[spA]\n"                                                                            \
		CNT_P                                                                                                       \
		"s_branch L_loop_%=\n"                                                                                      \
		"L_onlyl_%=:\n"                                                                                             \
		"s_mov_b32 %[curA], " RA_LREF "\n"                                                                          \
		"s_mov_b64 s[60:61], s[36:37]\n"                                                                            \
		"s_mov_b64 s[62:63], s[40:41]\n"                                                                            \
		"s_mov_b64 s[66:67], 0\n"                                                                                   \
		"s_branch L_loop_%=\n"                                                                                      \
		"L_lmiss_%=:\n"                                                                                             \
		"s_or_b64 vcc, s[38:39], s[42:43]\n"                                                                        \
		"s_cbranch_scc0 L_pop_%=\n"                                                                                 \
		"s_mov_b32 %[curA], " RA_RREF "\n"                                                                          \
		"s_mov_b64 s[60:61], s[38:39]\n"                                                                            \
		"s_mov_b64 s[62:63], s[42:43]\n"                                                                            \
		"s_mov_b64 s[66:67], -1\n"                                                                                  \
		"s_branch L_loop_%=\n"                                                                                      \
		"L_tri_%=:\n"                                                                                               \
		CNT_T                                                                                                       \
		"s_cmp_eq_u64 s[60:61], 0\n"                                                                                \
		"s_cbranch_scc1 L_Atnext_%=\n"                                                                              \
		MRT_ROWS_TRI_TEST("A", RA, "s[60:61]", "%[curA]", ANYA)                                                     \
		"s_cmp_eq_u64 s[62:63], 0\n"                                                                                \
		"s_cbranch_scc1 L_Btnext_%=\n"                                                                              \
		MRT_ROWS_TRI_TEST("B", RA, "s[62:63]", "%[curA]", ANYB)                                                     \
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
		"v_readfirstlane_b32 s60, v50\n"                                                                            \
		"v_readfirstlane_b32 s61, v51\n"                                                                            \
		"v_readfirstlane_b32 s62, v52\n"                                                                            \
		"v_readfirstlane_b32 s63, v53\n"                                                                            \
		"s_cmp_lg_u32 %[curA], 0x7fffffff\n"                                                                        \
		"s_cbranch_scc1 L_vload_%=\n"                                                                               \
		"s_branch L_exit_%=\n"                                                                                      \
		MRT_ROWSC_CHILD_OOL(L, "s[36:37]", "s[40:41]")                                                              \
		MRT_ROWSC_CHILD_OOL(R, "s[38:39]", "s[42:43]")                                                              \
		"L_exit_%=:\n"                                                                                              \
		"s_waitcnt vmcnt(0)\n"               /* no copy may land in v60 / v61 once the compiler owns them again */  \
		"s_mov_b64 %[maskA], s[60:61]\n"                                                                            \
		"s_mov_b64 %[maskB], s[62:63]\n"                                                                            \
		: MRT_ROWS_OUT(A, a), [limB] "+v"(b.lim), [btB] "+v"(b.bt), [buB] "+v"(b.bu), [bvB] "+v"(b.bv),             \
		  [bsB] "+v"(b.bs), [biB] "+v"(b.bi), [maskB] "+s"(b.mask), [cntn] "+s"(cnt_n), [cntt] "+s"(cnt_t), [cntp] "+s"(sp_max) \
		: MRT_ROWS_IN(A, a), MRT_ROWS_IN(B, b), [rows] "s"(rows), [qmask] "s"(qmask),                                \
		  [eps] "s"(eps), [vneg] "v"(vneg), [cw] "v"(cull.w), [ccc] "v"(cull.cc), [cvo] "v"(cull.voff)               \
		: MRT_ROWS_CLOBBERS, "v61", "v62")

// the lane's share of the packet's culling pyramid (see MRT_ROWS_LOOPC)
struct CullRegs {
	float w;          // weight of the lane's row dword: a component of the lane's plane normal, or 0
	float cc;         // lanes 7 and 15 of each 16-lane row: dot(n', apex) - eps; -inf elsewhere (and everywhere: never culls)
	uint32_t voff;    // byte offset of the lane's dword in a row
};

// counting builds of the one-packet loop also clock the row fetch: shader cycles from before the s_load to after
// its s_waitcnt (two s_memtime reads included), summed per wave in %[cntw]
#define MRT_ROWS_T_PRE "s_memtime s[64:65]\n s_waitcnt lgkmcnt(0)\n"
#define MRT_ROWS_T_POST "s_memtime s[66:67]\n s_waitcnt lgkmcnt(0)\n s_sub_u32 s64, s66, s64\n s_add_u32 %[cntw], %[cntw], s64\n"
#define MRT_ROWS_CNT_N "s_add_u32 %[cntn], %[cntn], 1\n"
#define MRT_ROWS_CNT_P "v_readfirstlane_b32 s52, %[spA]\n s_max_u32 %[cntp], %[cntp], s52\n" /* counting builds: the stack's high-water mark */
#define MRT_ROWS_CNT_T "s_add_u32 %[cntt], %[cntt], 1\n"

// wave-uniform operands come back from an asm block as SGPRs; this tells the compiler they still are
__device__ __forceinline__ void rows_uniform(PacketRegs &s)
{
	s.cur = __builtin_amdgcn_readfirstlane(s.cur);
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)s.mask), hi = __builtin_amdgcn_readfirstlane((uint32_t)(s.mask >> 32));
	s.mask = ((unsigned long long)hi << 32) | lo;
}

template <int OCT, bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void rows_walk_one(const float4 *rows, uint32_t qmask, PacketRegs &a, uint32_t &cnt_n, uint32_t &cnt_t, uint32_t &cnt_w, uint32_t &sp_max)
{
	const float eps = 1e-8f, vneg = -FLT_MAX;
	uint32_t ev = 0u;
#define MRT_W1(O, BX, BY, BZ)                                                                                         \
	if (OCT == O) {                                                                                                 \
		if (COUNT) { if (ANY_HIT) MRT_ROWS_LOOP1(MRT_ROWS_CNT_N, MRT_ROWS_CNT_T, MRT_ROWS_CNT_P, MRT_ROWS_T_PRE, MRT_ROWS_T_POST, MRT_ROWS_ANYHIT("A", "4"), BX, BY, BZ); \
			else MRT_ROWS_LOOP1(MRT_ROWS_CNT_N, MRT_ROWS_CNT_T, MRT_ROWS_CNT_P, MRT_ROWS_T_PRE, MRT_ROWS_T_POST, MRT_ROWS_NEAREST("A"), BX, BY, BZ); } \
		else { if (ANY_HIT) MRT_ROWS_LOOP1("", "", "", "", "", MRT_ROWS_ANYHIT("A", "4"), BX, BY, BZ);              \
			else MRT_ROWS_LOOP1("", "", "", "", "", MRT_ROWS_NEAREST("A"), BX, BY, BZ); }                           \
	}
	MRT_W1(0, 0, 0, 0) MRT_W1(1, 1, 0, 0) MRT_W1(2, 0, 1, 0) MRT_W1(3, 1, 1, 0) MRT_W1(4, 0, 0, 1) MRT_W1(5, 1, 0, 1) MRT_W1(6, 0, 1, 1) MRT_W1(7, 1, 1, 1)
#undef MRT_W1
	rows_uniform(a);
	cnt_n = __builtin_amdgcn_readfirstlane(cnt_n); cnt_t = __builtin_amdgcn_readfirstlane(cnt_t); cnt_w = __builtin_amdgcn_readfirstlane(cnt_w);
	sp_max = __builtin_amdgcn_readfirstlane(sp_max);
}

template <int OCT, bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void rows_walk_wide(const float4 *rows, uint32_t qmask, PacketRegs &a, PacketRegs &b, uint32_t &cnt_n, uint32_t &cnt_t, uint32_t &sp_max)
{
	const float eps = 1e-8f, vneg = -FLT_MAX;
#define MRT_WW(O, BX, BY, BZ)                                                                                         \
	if (OCT == O) {                                                                                                 \
		if (COUNT) { if (ANY_HIT) MRT_ROWS_LOOPW(MRT_ROWS_CNT_N, MRT_ROWS_CNT_T, MRT_ROWS_CNT_P, MRT_ROWSW_ANYHIT("A"), MRT_ROWSW_ANYHIT("B"), MRT_ROWSW_ANYDONE, BX, BY, BZ); \
			else MRT_ROWS_LOOPW(MRT_ROWS_CNT_N, MRT_ROWS_CNT_T, MRT_ROWS_CNT_P, MRT_ROWS_NEAREST("A"), MRT_ROWS_NEAREST("B"), "", BX, BY, BZ); } \
		else { if (ANY_HIT) MRT_ROWS_LOOPW("", "", "", MRT_ROWSW_ANYHIT("A"), MRT_ROWSW_ANYHIT("B"), MRT_ROWSW_ANYDONE, BX, BY, BZ); \
			else MRT_ROWS_LOOPW("", "", "", MRT_ROWS_NEAREST("A"), MRT_ROWS_NEAREST("B"), "", BX, BY, BZ); }        \
	}
	MRT_WW(0, 0, 0, 0) MRT_WW(1, 1, 0, 0) MRT_WW(2, 0, 1, 0) MRT_WW(3, 1, 1, 0) MRT_WW(4, 0, 0, 1) MRT_WW(5, 1, 0, 1) MRT_WW(6, 0, 1, 1) MRT_WW(7, 1, 1, 1)
#undef MRT_WW
	rows_uniform(a);
	cnt_n = __builtin_amdgcn_readfirstlane(cnt_n); cnt_t = __builtin_amdgcn_readfirstlane(cnt_t);
	sp_max = __builtin_amdgcn_readfirstlane(sp_max);
}

template <int OCT, bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void rows_walk_cull(const float4 *rows, uint32_t qmask, PacketRegs &a, PacketRegs &b, const CullRegs &cull, uint32_t &cnt_n, uint32_t &cnt_t, uint32_t &sp_max)
{
	const float eps = 1e-8f, vneg = -FLT_MAX;
#define MRT_WC(O, BX, BY, BZ)                                                                                         \
	if (OCT == O) {                                                                                                 \
		if (COUNT) { if (ANY_HIT) MRT_ROWS_LOOPC(MRT_ROWS_CNT_N, MRT_ROWS_CNT_T, MRT_ROWS_CNT_P, MRT_ROWSW_ANYHIT("A"), MRT_ROWSW_ANYHIT("B"), MRT_ROWSW_ANYDONE, BX, BY, BZ); \
			else MRT_ROWS_LOOPC(MRT_ROWS_CNT_N, MRT_ROWS_CNT_T, MRT_ROWS_CNT_P, MRT_ROWS_NEAREST("A"), MRT_ROWS_NEAREST("B"), "", BX, BY, BZ); } \
		else { if (ANY_HIT) MRT_ROWS_LOOPC("", "", "", MRT_ROWSW_ANYHIT("A"), MRT_ROWSW_ANYHIT("B"), MRT_ROWSW_ANYDONE, BX, BY, BZ); \
			else MRT_ROWS_LOOPC("", "", "", MRT_ROWS_NEAREST("A"), MRT_ROWS_NEAREST("B"), "", BX, BY, BZ); }        \
	}
	MRT_WC(0, 0, 0, 0) MRT_WC(1, 1, 0, 0) MRT_WC(2, 0, 1, 0) MRT_WC(3, 1, 1, 0) MRT_WC(4, 0, 0, 1) MRT_WC(5, 1, 0, 1) MRT_WC(6, 0, 1, 1) MRT_WC(7, 1, 1, 1)
#undef MRT_WC
	rows_uniform(a);
	cnt_n = __builtin_amdgcn_readfirstlane(cnt_n); cnt_t = __builtin_amdgcn_readfirstlane(cnt_t);
	sp_max = __builtin_amdgcn_readfirstlane(sp_max);
}

// The packet's culling pyramid.  Culling is on only if every ray of the two tiles starts at one point (bit for bit),
// has t_min >= 0, and lies inside the four planes through the corner rays (lane 0 / 56 of tile A, lane 7 / 63 of tile B
// = the corners of the 16x8 block when B is A's right-hand neighbour) pushed outwards by 5e-5 rad: checked here for
// every lane, so any other arrangement (tiles on different image rows, rays that are not a pinhole grid, a partial
// tile without its corner lanes) simply does not cull.  Soundness: a box is skipped only if S < fl(dot(n', apex)) - eps,
// S = the rounded sum of the three rounded products n'_axis * corner_axis (relative error 2^-24 each, two real additions
// of at most 2^-24 of the partial sum each: |S - dot(n', corner)| <= 3 * 2^-24 |n'|_1 max|coordinate|), the right-hand
// side off by at most 4 * 2^-24 |n'|_1 |apex|_1, and eps = 8 * 2^-24 * |n'|_1 * (largest scene coordinate + |apex|_1),
// more than both together: so the true dot(n', corner - apex) is negative, the corner being the box's farthest point
// along n': no point of the box is on the inner side of that plane, and every ray point o + t d, t >= 0, is
// (dot(n', d) >= 1e-5 |d| was checked with the same roundings to spare).  The slab test such a ray makes against such a
// box fails by a margin (>= 1e-5 of the distance) three orders above its own rounding, so the skipped tests were
// all-false masks.
// Lane l: plane r = l >> 4 (top, right, bottom, left), row dword c = l & 15 = {lmin xyz, ref, lmax xyz, ref, rmin xyz, -,
// rmax xyz, -}.
__device__ __forceinline__ CullRegs cull_setup(const RayRegs &ra, const RayRegs &rb, bool valid_a, bool valid_b, unsigned long long part_a,
		unsigned long long part_b, float scene_abs_max, uint32_t lane)
{
	CullRegs c;
	c.w = 0.0f; c.cc = -__builtin_inff(); c.voff = (lane & 15u) * 4u;
	const bool corners = (part_a & 1ull) && (part_a >> 56 & 1ull) && (part_b >> 7 & 1ull) && (part_b >> 63 & 1ull);
	if (!corners) return c; // (wave-uniform)
#define MRT_RL(v, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l))
	const float ox = MRT_RL(ra.ox, 0), oy = MRT_RL(ra.oy, 0), oz = MRT_RL(ra.oz, 0);
	const float tl[3] = { MRT_RL(ra.dx, 0), MRT_RL(ra.dy, 0), MRT_RL(ra.dz, 0) }, bl[3] = { MRT_RL(ra.dx, 56), MRT_RL(ra.dy, 56), MRT_RL(ra.dz, 56) };
	const float tr[3] = { MRT_RL(rb.dx, 7), MRT_RL(rb.dy, 7), MRT_RL(rb.dz, 7) }, br[3] = { MRT_RL(rb.dx, 63), MRT_RL(rb.dy, 63), MRT_RL(rb.dz, 63) };
	float cen[3] = { tl[0] + tr[0] + bl[0] + br[0], tl[1] + tr[1] + bl[1] + br[1], tl[2] + tr[2] + bl[2] + br[2] };
	const float cl = __builtin_sqrtf(cen[0] * cen[0] + cen[1] * cen[1] + cen[2] * cen[2]);
	if (!(cl > 0.0f)) return c;
	cen[0] /= cl; cen[1] /= cl; cen[2] /= cl;
	// the lane's plane: k = lane >> 4 -> (top, right, bottom, left) = cross of the two corner rays on that side
	const uint32_t k = lane >> 4;
	const float *pa = k == 0u ? tl : (k == 1u ? tr : (k == 2u ? br : bl)), *pb = k == 0u ? tr : (k == 1u ? br : (k == 2u ? bl : tl));
	float n[3] = { pa[1] * pb[2] - pa[2] * pb[1], pa[2] * pb[0] - pa[0] * pb[2], pa[0] * pb[1] - pa[1] * pb[0] };
	const float nl = __builtin_sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
	bool ok = nl > 0.0f;
	const float inv = ok ? 1.0f / nl : 0.0f;
	n[0] *= inv; n[1] *= inv; n[2] *= inv;
	if (n[0] * cen[0] + n[1] * cen[1] + n[2] * cen[2] < 0.0f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
	const float push = 5e-5f;
	n[0] += push * cen[0]; n[1] += push * cen[1]; n[2] += push * cen[2];
	// every lane checks its own two rays against ALL four planes (lanes 0, 16, 32, 48 hold one each)
	bool inside = true;
	for (int q = 0; q < 4; q++) {
		const float qx = MRT_RL(n[0], 16 * q), qy = MRT_RL(n[1], 16 * q), qz = MRT_RL(n[2], 16 * q);
		const float da = qx * ra.dx + qy * ra.dy + qz * ra.dz, la = __builtin_sqrtf(ra.dx * ra.dx + ra.dy * ra.dy + ra.dz * ra.dz);
		const float db = qx * rb.dx + qy * rb.dy + qz * rb.dz, lb = __builtin_sqrtf(rb.dx * rb.dx + rb.dy * rb.dy + rb.dz * rb.dz);
		if (valid_a && !(da >= 1e-5f * la)) inside = false;
		if (valid_b && !(db >= 1e-5f * lb)) inside = false;
	}
#undef MRT_RL
	if (valid_a && !(ra.ox == ox && ra.oy == oy && ra.oz == oz && ra.t_min >= 0.0f)) inside = false;
	if (valid_b && !(rb.ox == ox && rb.oy == oy && rb.oz == oz && rb.t_min >= 0.0f)) inside = false;
	ok = ok && __builtin_isfinite(n[0]) && __builtin_isfinite(n[1]) && __builtin_isfinite(n[2]);
	const bool planes_ok = (__ballot(ok) & 0x0001000100010001ull) == 0x0001000100010001ull;
	if (__ballot(!inside) != 0ull || !planes_ok) return c;
	const float n1 = __builtin_fabsf(n[0]) + __builtin_fabsf(n[1]) + __builtin_fabsf(n[2]);
	const float o1 = __builtin_fabsf(ox) + __builtin_fabsf(oy) + __builtin_fabsf(oz);
	const float err = 8.0f * 5.9604645e-8f * n1 * (scene_abs_max + o1);
	const float cc = (n[0] * ox + n[1] * oy + n[2] * oz) - err;
	if (__ballot(!__builtin_isfinite(cc)) != 0ull) return c;
	const uint32_t d = lane & 15u, axis = d & 3u;
	const bool is_max = (d >> 2 & 1u) != 0u;
	const float na = axis == 0u ? n[0] : (axis == 1u ? n[1] : n[2]);
	c.w = (axis < 3u && ((na >= 0.0f) == is_max)) ? na : 0.0f;   // the corner farthest along n': max where n' >= 0, else min
	c.cc = (d == 7u || d == 15u) ? cc : -__builtin_inff();
	return c;
}

__device__ __forceinline__ void rows_init(PacketRegs &s, const RayRegs &r, bool takes_part, uint32_t sp)
{
	s.ox = r.ox; s.oy = r.oy; s.oz = r.oz; s.dx = r.dx; s.dy = r.dy; s.dz = r.dz; s.tmin = r.t_min;
	s.ix = safe_inv(r.dx); s.iy = safe_inv(r.dy); s.iz = safe_inv(r.dz);
	s.nrx = -(r.ox * s.ix); s.nry = -(r.oy * s.iy); s.nrz = -(r.oz * s.iz);
	s.bt = r.t_max; s.bu = 0.0f; s.bv = 0.0f; s.bs = 0xFFFFFFFFu; s.bi = 0xFFFFFFFFu;
	s.lim = (!takes_part || r.t_min >= r.t_max) ? -FLT_MAX : r.t_max; // degenerate rays are misses, glsl:214-222
	s.sp = sp; s.cur = 0u; s.mask = 0ull;                              // the root is always a wide node
}

// wave-uniform octant of a packet's reciprocal directions over the lanes that take part (8 = mixed)
__device__ __forceinline__ int rows_octant(const RayRegs &r, bool part, unsigned long long &part_mask)
{
	part_mask = __ballot(part);
	const unsigned long long sx = __ballot(part && safe_inv(r.dx) < 0.0f), sy = __ballot(part && safe_inv(r.dy) < 0.0f),
			sz = __ballot(part && safe_inv(r.dz) < 0.0f);
	const bool uniform = (sx == 0ull || sx == part_mask) && (sy == 0ull || sy == part_mask) && (sz == 0ull || sz == part_mask);
	return uniform ? ((sx ? 1 : 0) | (sy ? 2 : 0) | (sz ? 4 : 0)) : 8;
}

#ifndef MRT_ROWS_WPE
#define MRT_ROWS_WPE 8
#endif
// PACKETS = 1: one packet per wave; 2: two (neighbouring tiles of the launch order).
// WG = threads per workgroup: 256 (four waves on neighbouring tiles share a CU and its scalar cache: C5 20.9 against
// 22.6 ms) or 64 (a wave slot is refilled as soon as ITS wave ends, not when a workgroup's worth of slots is free:
// C3 2.06 against 2.17 ms); api.hip picks by the size of the scene.
// CULL: paired packets walk with packet-level frustum culling (MRT_ROWS_LOOPC).
template <bool ANY_HIT, bool COUNT, int PACKETS, int WG, bool CULL = false>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(MRT_ROWS_WPE, 8))) void trace_packet_rows_kernel(const TraceParams p)
{
	// per wave and packet: 16-byte stack entries {ref, -, lane mask}; entry 0 holds the sentinel
	__shared__ __attribute__((aligned(16))) uint32_t wave_stack[WG / MRT_WAVE][PACKETS][(MRT_PACKET_STACK + 1) * 4];
	if (skip_launch(p)) return;
	uint32_t block = blockIdx.x;
	if (p.xcd_swizzle) {
		const uint32_t per = gridDim.x >> 3;
		if (block < (per << 3)) block = (block & 7u) * per + (block >> 3);
	}
	const uint32_t wave = threadIdx.x / MRT_WAVE, lane = threadIdx.x & (MRT_WAVE - 1);
	const uint32_t wave_s = __builtin_amdgcn_readfirstlane(wave), block_s = block; // (scalars: for the epilogue)
	const uint64_t g_a = (((uint64_t)block * (WG / MRT_WAVE) + wave) * PACKETS) * MRT_WAVE + lane, g_b = g_a + MRT_WAVE;
	uint64_t idx = 0; uint32_t px = 0, py = 0;
	const bool valid_a = lane_ray_index_g(p, g_a, idx, px, py);
	// a lane without a ray in a packet walks along with an empty interval
	RayRegs ra = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f}, rb = ra;
	if (valid_a) load_ray(p, idx, px, py, ra);
	bool valid_b = false;
	if (PACKETS == 2) {
		valid_b = lane_ray_index_g(p, g_b, idx, px, py);
		if (valid_b) load_ray(p, idx, px, py, rb);
	}
	if (__ballot(valid_a || valid_b) == 0ull) return; // nothing for this wave (otherwise every lane stays in)

	const float4 *rows = reinterpret_cast<const float4 *>(p.row_array);
	uint32_t *stack_a = wave_stack[wave][0], *stack_b = wave_stack[wave][PACKETS - 1];
	// sentinels (volatile: the asm blocks have no "memory" clobber; they touch only read-only scene data and this stack)
	*(volatile uint32_t *)&stack_a[0] = kSentinel;
	if (PACKETS == 2) *(volatile uint32_t *)&stack_b[0] = kSentinel;
	PacketRegs A, B;
	rows_init(A, ra, valid_a, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(stack_a + 4));
	rows_init(B, rb, valid_b, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(stack_b + 4));

	unsigned long long part_a = 0ull, part_b = 0ull;
	const int oct_a = rows_octant(ra, valid_a, part_a);
	const int oct_b = PACKETS == 2 ? rows_octant(rb, valid_b, part_b) : 8;
	uint32_t cnt_n = 0u, cnt_t = 0u; // COUNT: node rows / triangle rows fetched by this wave
	uint32_t cnt_w = 0u;             // COUNT, one-packet loop: shader cycles between issuing a row fetch and having it
	uint32_t sp_wide = 0u, sp_one = 0u; // COUNT: highest stack pointers seen (LDS byte addresses) by the 128-ray / the one-packet walk
	// the wave's start time waits in memory, not in registers (none to spare in the walk): in the cost array itself
	// (note_tile_start / note_tile_cost), and for the counting builds in the unused words of the stack's sentinel entry
	if (p.tile_cost != nullptr && lane == 0u) note_tile_start(p, g_a);
	if (COUNT) *(volatile unsigned long long *)&stack_a[2] = __builtin_amdgcn_s_memtime();
	bool done_a = part_a == 0ull, done_b = part_b == 0ull;
	if (PACKETS == 2 && !done_a && !done_b && oct_a == oct_b && oct_a != 8) {
		// one walk for both groups: 32-byte stack entries over the wave's whole stack area, sentinel at its bottom;
		// at the root every lane that takes part owns the row
		*(volatile uint32_t *)&stack_a[4] = kSentinel;
		A.sp += 16u; // (= stack base + 32)
		A.mask = part_a; B.mask = part_b;
		if (CULL) {
			const CullRegs cull = cull_setup(ra, rb, valid_a, valid_b, part_a, part_b, p.scene_abs_max, lane);
#define MRT_RWC(O) case O: rows_walk_cull<O, ANY_HIT, COUNT>(rows, p.query_mask, A, B, cull, cnt_n, cnt_t, sp_wide); break;
			switch (oct_a) { MRT_RWC(0) MRT_RWC(1) MRT_RWC(2) MRT_RWC(3) MRT_RWC(4) MRT_RWC(5) MRT_RWC(6) MRT_RWC(7) }
#undef MRT_RWC
		} else {
#define MRT_RW2(O) case O: rows_walk_wide<O, ANY_HIT, COUNT>(rows, p.query_mask, A, B, cnt_n, cnt_t, sp_wide); break;
		switch (oct_a) { MRT_RW2(0) MRT_RW2(1) MRT_RW2(2) MRT_RW2(3) MRT_RW2(4) MRT_RW2(5) MRT_RW2(6) MRT_RW2(7) }
#undef MRT_RW2
		}
		done_a = done_b = true;
	}
	// packets that could not be paired (different octants: tiles on an image axis; a single packet): one at a time
#define MRT_RW1(O, S) case O: rows_walk_one<O, ANY_HIT, COUNT>(rows, p.query_mask, S, cnt_n, cnt_t, cnt_w, sp_one); break;
	if (!done_a && oct_a != 8) { switch (oct_a) { MRT_RW1(0, A) MRT_RW1(1, A) MRT_RW1(2, A) MRT_RW1(3, A) MRT_RW1(4, A) MRT_RW1(5, A) MRT_RW1(6, A) MRT_RW1(7, A) } done_a = true; }
	if (PACKETS == 2 && !done_b && oct_b != 8) { switch (oct_b) { MRT_RW1(0, B) MRT_RW1(1, B) MRT_RW1(2, B) MRT_RW1(3, B) MRT_RW1(4, B) MRT_RW1(5, B) MRT_RW1(6, B) MRT_RW1(7, B) } done_b = true; }
#undef MRT_RW1
	// best hit as the other kernels keep it: leaf-order slot = row - number of node rows
	uint32_t slot_a = A.bs == 0xFFFFFFFFu ? 0xFFFFFFFFu : A.bs - p.n_nodes, slot_b = B.bs == 0xFFFFFFFFu ? 0xFFFFFFFFu : B.bs - p.n_nodes;
	// mixed directions inside a packet: the generic compiler-scheduled walk over nodes + triangle arrays
	if (!done_a) {
		uint32_t nn = 0, nt = 0, nd = 0;
		A.bt = ra.t_max;
		packet_traverse<8, ANY_HIT, COUNT>(p, ra, stack_a, A.bt, A.bu, A.bv, slot_a, nn, nt, nd, 0u, 0u, nullptr, !valid_a);
		if (COUNT) { cnt_n += __builtin_amdgcn_readfirstlane(nn); cnt_t += __builtin_amdgcn_readfirstlane(nt); }
	}
	if (PACKETS == 2 && !done_b) {
		uint32_t nn = 0, nt = 0, nd = 0;
		B.bt = rb.t_max;
		packet_traverse<8, ANY_HIT, COUNT>(p, rb, stack_b, B.bt, B.bu, B.bv, slot_b, nn, nt, nd, 0u, 0u, nullptr, !valid_b);
		if (COUNT) { cnt_n += __builtin_amdgcn_readfirstlane(nn); cnt_t += __builtin_amdgcn_readfirstlane(nt); }
	}

	// The rays' indices again, from nothing the prologue computed (the wave from a scalar, the lane from the exec-mask
	// count): no index, pixel or validity flag stays live across the walk, where every vector register is spoken for
	// (the compiler spilled 18 of them to scratch around the assembly block: 0.3 GB of writes per C3 launch).
	{
		const uint32_t lane2 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
		const uint64_t h_a = (((uint64_t)block_s * (WG / MRT_WAVE) + wave_s) * PACKETS) * MRT_WAVE + lane2;
		uint64_t idx2 = 0; uint32_t px2 = 0, py2 = 0;
		if (lane_ray_index_g(p, h_a, idx2, px2, py2)) finish_ray(p, idx2, ra, A.bt, A.bu, A.bv, slot_a);
		if (PACKETS == 2 && lane_ray_index_g(p, h_a + MRT_WAVE, idx2, px2, py2)) finish_ray(p, idx2, rb, B.bt, B.bu, B.bv, slot_b);
		if (p.tile_cost != nullptr && lane2 == 0u) note_tile_cost(p, h_a);
	}
	const unsigned long long t_start = COUNT ? *(volatile unsigned long long *)&stack_a[2] : 0ull;
	if (COUNT && lane == 0u && (p.count_mode != 2u || (blockIdx.x & 15u) == 0u)) { // the wave's clock: cycles in the row-fetch waits of the one-packet loop, cycles in all
		atomicAdd(&p.counters[kCntFetchWaitCycles], (unsigned long long)cnt_w);
		atomicAdd(&p.counters[kCntWaveCycles], __builtin_amdgcn_s_memtime() - t_start);
		atomicAdd(&p.counters[kCntWaves], 1ull);
	}
	if (COUNT && p.count_mode != 2u) { // the stack's high-water mark, in entries (the bound is the uploaded BVH's depth, <= 64: api.hip)
		const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)stack_a;
		const uint32_t e_wide = sp_wide > base ? (sp_wide - base) / 32u : 0u, e_one = sp_one > base ? ((sp_one - base) % 1040u) / 16u : 0u;
		atomicMax(&p.counters[kCntMaxStack], (unsigned long long)(e_wide > e_one ? e_wide : e_one));
	}
	if (COUNT && p.count_mode != 2u) { // (count_visits = 2: the clock only, sampled, so that the counting itself does not load the memory system)
		// per-ray words: every step of the wave is charged to every ray of the wave (both packets): an upper bound
		// per packet; the fetch words (kCntWaveNodeFetch / kCntWaveTriFetch) are exact
		if (valid_a) packet_count(p, cnt_n, cnt_t, 0u, slot_a != 0xFFFFFFFFu, part_a | part_b);
		if (PACKETS == 2 && valid_b) { atomicAdd(&p.counters[kCntRays], 1ull); if (slot_b != 0xFFFFFFFFu) atomicAdd(&p.counters[kCntHits], 1ull); }
	}
}
