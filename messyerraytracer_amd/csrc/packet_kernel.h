// packet_kernel.h — the packet kernel: one wave = one 64-ray packet.
// Included by kernels.hip (inside namespace mrt, after the common helpers).
//
// For coherent batches (primary-ray tiles).  The wave walks ONE path through the
// tree: the node index lives in an SGPR, the node / triangle are fetched once per
// wave through the scalar cache (s_load_dwordx8 x2) instead of 64 x 4 vector
// loads, the traversal stack is one LDS dword per level per WAVE (256 B), and the
// per-child decisions are wave ballots.  A child is visited when ANY lane hits
// its box; lanes that miss it ride along.  In a LEAF a lane accepts triangle hits
// only if its own ray passed the box test of that leaf (the bit is kept with the
// stack entry when a leaf is pushed): child boxes are nested and the slab test is
// monotone in the box coordinate, so passing a leaf's box implies passing every
// ancestor's, and each lane ends up testing exactly the leaves its own one-ray
// walk would test -- also for a ray that grazes a box face, where a slab test
// (rounded) and the triangle test (rounded otherwise) can disagree.
//
// Octant specialisation: when every ray of the packet has the same direction
// signs (true for almost every primary-ray tile), min(t0,t1) / max(t0,t1) of the
// slab test are known per axis — fma(b, inv, c) is monotone in b — so the near /
// far planes are picked at compile time and the 12 v_min/v_max disappear.  The
// values are bit-identical to the generic form.  OCT = 8 is the generic form.
#pragma once

#define MRT_PACKET_STACK 64

template <int OCT, bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void packet_traverse(const TraceParams &p, const RayRegs &r, uint32_t *stack,
		float &best_t, float &best_u, float &best_v, uint32_t &best_slot, uint32_t &n_nodes, uint32_t &n_tris, uint32_t &n_dead,
		// a BLAS of a two-level scene (two_level_kernel.h): its root, the instance's flat id base, the id of the
		// best hit so far (in / out), and whether this lane sits the walk out (its world ray missed the instance)
		uint32_t root = 0u, uint32_t id_base = 0u, uint32_t *best_id_io = nullptr, bool dead = false)
{
	const bool degenerate = r.t_min >= r.t_max; // glsl:214-222: a miss with t = t_max
	// a lane that must not take part any more gets an empty interval: no box test can pass
	float lim_t = (degenerate || dead) ? -FLT_MAX : best_t;
	const float ix = safe_inv(r.dx), iy = safe_inv(r.dy), iz = safe_inv(r.dz);
	const float nrx = -(r.ox * ix), nry = -(r.oy * iy), nrz = -(r.oz * iz);
	const float4 *nodes = reinterpret_cast<const float4 *>(p.nodes);
	const float4 *hot = reinterpret_cast<const float4 *>(p.tri_hot);
	uint32_t best_id = best_id_io ? *best_id_io : 0xFFFFFFFFu;
	constexpr bool NX = (OCT & 1) != 0, NY = (OCT & 2) != 0, NZ = (OCT & 4) != 0; // inv < 0 on that axis
	uint32_t sp = 0;   // wave-uniform
	uint32_t cur = root; // wave-uniform: always a wide node
	bool popped = false;
	unsigned long long own_bits = 0ull; // bit k: this lane's own ray hit the box of stack entry k
	bool own = true;                    // ... of the node in `cur` (used when it is a leaf)

	for (;;) {
		cur = __builtin_amdgcn_readfirstlane(cur);
		if (cur < kSentinel) {
			const float4 *n = nodes + (size_t)cur * 4u; // uniform address: scalar loads
			const float4 a = n[0], b = n[1], c = n[2], d = n[3];
			if (COUNT) n_nodes++;
			float tl, tlx, tr, trx;
			if (OCT == 8) { // ray_aabb, glsl:84-99
				const float l0x = fma_(a.x, ix, nrx), l1x = fma_(b.x, ix, nrx);
				const float l0y = fma_(a.y, iy, nry), l1y = fma_(b.y, iy, nry);
				const float l0z = fma_(a.z, iz, nrz), l1z = fma_(b.z, iz, nrz);
				const float r0x = fma_(c.x, ix, nrx), r1x = fma_(d.x, ix, nrx);
				const float r0y = fma_(c.y, iy, nry), r1y = fma_(d.y, iy, nry);
				const float r0z = fma_(c.z, iz, nrz), r1z = fma_(d.z, iz, nrz);
				tl = fmaxf(fmaxf(fminf(l0x, l1x), fminf(l0y, l1y)), fmaxf(fminf(l0z, l1z), r.t_min));
				tlx = fminf(fminf(fmaxf(l0x, l1x), fmaxf(l0y, l1y)), fminf(fmaxf(l0z, l1z), lim_t));
				tr = fmaxf(fmaxf(fminf(r0x, r1x), fminf(r0y, r1y)), fmaxf(fminf(r0z, r1z), r.t_min));
				trx = fminf(fminf(fmaxf(r0x, r1x), fmaxf(r0y, r1y)), fminf(fmaxf(r0z, r1z), lim_t));
			} else { // same values with the entry / exit plane of each axis chosen by the octant
				const float lnx = fma_(NX ? b.x : a.x, ix, nrx), lfx = fma_(NX ? a.x : b.x, ix, nrx);
				const float lny = fma_(NY ? b.y : a.y, iy, nry), lfy = fma_(NY ? a.y : b.y, iy, nry);
				const float lnz = fma_(NZ ? b.z : a.z, iz, nrz), lfz = fma_(NZ ? a.z : b.z, iz, nrz);
				const float rnx = fma_(NX ? d.x : c.x, ix, nrx), rfx = fma_(NX ? c.x : d.x, ix, nrx);
				const float rny = fma_(NY ? d.y : c.y, iy, nry), rfy = fma_(NY ? c.y : d.y, iy, nry);
				const float rnz = fma_(NZ ? d.z : c.z, iz, nrz), rfz = fma_(NZ ? c.z : d.z, iz, nrz);
				tl = fmaxf(fmaxf(lnx, lny), fmaxf(lnz, r.t_min));
				tlx = fminf(fminf(lfx, lfy), fminf(lfz, lim_t));
				tr = fmaxf(fmaxf(rnx, rny), fmaxf(rnz, r.t_min));
				trx = fminf(fminf(rfx, rfy), fminf(rfz, lim_t));
			}
			const bool hl = tl <= tlx, hr = tr <= trx;
			const unsigned long long ml = __ballot(hl), mr = __ballot(hr);
			const uint32_t lref = __float_as_uint(a.w), rref = __float_as_uint(b.w);
			if (COUNT && popped && (ml | mr) == 0ull) n_dead++;
			popped = false;
			if (ml != 0ull && mr != 0ull) {
				// order: the child that more lanes would enter first goes first
				const unsigned long long lfirst = __ballot(hl && (!hr || tl < tr));
				const bool left_near = 2 * __builtin_popcountll(lfirst) >= __builtin_popcountll(ml | mr);
				stack[sp] = left_near ? rref : lref;
				own_bits = (left_near ? hr : hl) ? (own_bits | (1ull << sp)) : (own_bits & ~(1ull << sp));
				sp++;
				cur = left_near ? lref : rref; own = left_near ? hl : hr;
				continue;
			}
			if (ml != 0ull) { cur = lref; own = hl; continue; }
			if (mr != 0ull) { cur = rref; own = hr; continue; }
		} else {
			// leaf: every lane whose own ray hit the leaf's box tests its triangles (glsl:166-192)
			float lim_leaf = own ? lim_t : -FLT_MAX;
			uint32_t slot = cur & 0x7FFFFFFFu;
			bool last;
			do {
				const float4 *t3 = hot + (size_t)slot * 3u; // uniform address
				const float4 q0 = t3[0], q1 = t3[1], q2 = t3[2];
				last = (__float_as_uint(q2.w) & kLastInLeaf) != 0u;
				if (COUNT) n_tris++; // a triangle row fetched (its test is skipped for layers outside the query mask)
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
								// lim_t == best_t for live lanes; exact ties go to the lower triangle id
								const uint32_t id = id_base + __float_as_uint(q0.w);
								if (!(t < r.t_min) && (t < lim_leaf || (t == lim_leaf && best_slot != 0xFFFFFFFFu && id < best_id))) {
									best_t = t; best_u = u; best_v = v; best_slot = slot; best_id = id;
									lim_t = lim_leaf = ANY_HIT ? -FLT_MAX : t; // any-hit: this lane is done
								}
							}
						}
					}
				}
				slot++;
			} while (!last);
			if (ANY_HIT && __ballot(lim_t != -FLT_MAX) == 0ull) break; // every lane has its answer
		}
		if (sp == 0) break;
		sp--; cur = stack[sp];
		own = ((own_bits >> sp) & 1ull) != 0ull;
		popped = true;
	}
	if (best_id_io) *best_id_io = best_id;
}

// Counters of a packet walk: node steps and triangle rows are per PACKET (wave-uniform n_nodes, n_tris); the
// per-ray words charge every wave step to every live lane (kCntNodes, kCntTris), the fetch words count it once.
__device__ __forceinline__ void packet_count(const TraceParams &p, uint32_t n_nodes, uint32_t n_tris, uint32_t n_dead, bool hit,
		unsigned long long live)
{
	atomicAdd(&p.counters[kCntRays], 1ull);
	atomicAdd(&p.counters[kCntTris], (unsigned long long)n_tris);
	atomicAdd(&p.counters[kCntNodes], (unsigned long long)n_nodes);
	if (hit) atomicAdd(&p.counters[kCntHits], 1ull);
	atomicAdd(&p.counters[kCntDeadPops], (unsigned long long)n_dead);
	if ((threadIdx.x & (MRT_WAVE - 1)) == (uint32_t)__builtin_ctzll(live)) {
		atomicAdd(&p.counters[kCntWaveNodeFetch], (unsigned long long)n_nodes);
		atomicAdd(&p.counters[kCntWaveTriFetch], (unsigned long long)n_tris);
	}
}

template <bool ANY_HIT, bool COUNT>
__global__ __launch_bounds__(MRT_WG) void trace_packet_kernel(const TraceParams p)
{
	__shared__ uint32_t wave_stack[MRT_WG / MRT_WAVE][MRT_PACKET_STACK];
	if (skip_launch(p)) return;
	uint32_t block = blockIdx.x;
	if (p.xcd_swizzle) {
		const uint32_t per = gridDim.x >> 3;
		if (block < (per << 3)) block = (block & 7u) * per + (block >> 3);
	}
	uint64_t ray_idx = 0; uint32_t px = 0, py = 0;
	if (!lane_ray_index(p, block, ray_idx, px, py)) return; // exited lanes drop out of every ballot
	RayRegs r;
	load_ray(p, ray_idx, px, py, r);

	float best_t = r.t_max, best_u = 0.0f, best_v = 0.0f;
	uint32_t best_slot = 0xFFFFFFFFu;
	uint32_t n_nodes = 0, n_tris = 0, n_dead = 0;
	uint32_t *stack = wave_stack[threadIdx.x / MRT_WAVE];

	// wave-uniform octant of the reciprocal directions (the sign safe_inv produces)
	const unsigned long long live = __ballot(true);
	const unsigned long long sx = __ballot(safe_inv(r.dx) < 0.0f), sy = __ballot(safe_inv(r.dy) < 0.0f),
			sz = __ballot(safe_inv(r.dz) < 0.0f);
	const bool uniform = (sx == 0ull || sx == live) && (sy == 0ull || sy == live) && (sz == 0ull || sz == live);
	const int oct = uniform ? ((sx ? 1 : 0) | (sy ? 2 : 0) | (sz ? 4 : 0)) : 8;
#define MRT_PKT(O) case O: packet_traverse<O, ANY_HIT, COUNT>(p, r, stack, best_t, best_u, best_v, best_slot, n_nodes, n_tris, n_dead); break;
	switch (oct) {
		MRT_PKT(0) MRT_PKT(1) MRT_PKT(2) MRT_PKT(3) MRT_PKT(4) MRT_PKT(5) MRT_PKT(6) MRT_PKT(7)
		default: packet_traverse<8, ANY_HIT, COUNT>(p, r, stack, best_t, best_u, best_v, best_slot, n_nodes, n_tris, n_dead); break;
	}
#undef MRT_PKT

	finish_ray(p, ray_idx, r, best_t, best_u, best_v, best_slot);

	if (COUNT) packet_count(p, n_nodes, n_tris, n_dead, best_slot != 0xFFFFFFFFu, live);
}
