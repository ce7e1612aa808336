// packet2_kernel.h — two packets per wave, interleaved.
// Included by kernels.hip (inside namespace mrt, after packet_kernel.h).
//
// The packet walk (packet_kernel.h) is a chain of dependent scalar fetches: per step a
// wave waits ~600 cycles for a 64-byte node and then issues ~60 instructions.  The
// occupancy sweep in profiles/ shows time inversely proportional to waves per SIMD up to
// 4 and still 1.7x from 4 to 8 — and 8 is the hardware limit.  This kernel doubles the
// independent chains per wave instead: each wave owns TWO 64-ray packets (two adjacent
// tiles, one ray of each per lane) and advances them in lockstep: both fetches are issued
// back to back, one s_waitcnt covers both, then both are processed.  Every fetch is 64
// bytes at a uniform address: a node, or one triangle (48 B + 16 B of slack), so one
// step = one node test or one triangle test for each packet.
//
// Results are those of the other kernels (same box tests, same triangle test, ties to
// the lower triangle id).
#pragma once

#define MRT_PACKET2_STACK 64

struct Pkt2 {
	RayRegs r;
	float ix, iy, iz, nrx, nry, nrz;
	float lim_t, best_t, best_u, best_v;
	uint32_t best_slot, best_id;
	uint32_t cur, sp;   // wave-uniform
	bool alive;         // wave-uniform
};

__device__ __forceinline__ void pkt2_init(Pkt2 &k, const RayRegs &r, bool valid)
{
	k.r = r;
	if (!valid) { k.r.t_min = 1.0f; k.r.t_max = 0.0f; } // dead lane: empty interval, never hits, stores nothing
	k.ix = safe_inv(r.dx); k.iy = safe_inv(r.dy); k.iz = safe_inv(r.dz);
	k.nrx = -(r.ox * k.ix); k.nry = -(r.oy * k.iy); k.nrz = -(r.oz * k.iz);
	k.best_t = k.r.t_max; k.best_u = 0.0f; k.best_v = 0.0f;
	k.best_slot = 0xFFFFFFFFu; k.best_id = 0xFFFFFFFFu;
	k.lim_t = (k.r.t_min >= k.r.t_max) ? -FLT_MAX : k.best_t;
	k.cur = 0; k.sp = 0;
}

// 64-byte uniform fetch for the packet's current reference: a wide node or one triangle.
__device__ __forceinline__ const float4 *pkt2_addr(const TraceParams &p, uint32_t cur)
{
	const uint32_t c = __builtin_amdgcn_readfirstlane(cur);
	return c >= kLeafBit ? reinterpret_cast<const float4 *>(p.tri_hot) + (size_t)(c & 0x7FFFFFFFu) * 3u
	                     : reinterpret_cast<const float4 *>(p.nodes) + (size_t)c * 4u;
}

template <int OCT, bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void pkt2_step(const TraceParams &p, Pkt2 &k, uint32_t *stack,
		const float4 d0, const float4 d1, const float4 d2, const float4 d3, uint32_t &n_nodes, uint32_t &n_tris)
{
	constexpr bool NX = (OCT & 1) != 0, NY = (OCT & 2) != 0, NZ = (OCT & 4) != 0;
	const uint32_t cur = __builtin_amdgcn_readfirstlane(k.cur);
	bool pop = false;
	if (cur < kSentinel) {
		// d0 = lmin|lref, d1 = lmax|rref, d2 = rmin, d3 = rmax
		if (COUNT) n_nodes++;
		float tl, tlx, tr, trx;
		if (OCT == 8) {
			const float l0x = fma_(d0.x, k.ix, k.nrx), l1x = fma_(d1.x, k.ix, k.nrx);
			const float l0y = fma_(d0.y, k.iy, k.nry), l1y = fma_(d1.y, k.iy, k.nry);
			const float l0z = fma_(d0.z, k.iz, k.nrz), l1z = fma_(d1.z, k.iz, k.nrz);
			const float r0x = fma_(d2.x, k.ix, k.nrx), r1x = fma_(d3.x, k.ix, k.nrx);
			const float r0y = fma_(d2.y, k.iy, k.nry), r1y = fma_(d3.y, k.iy, k.nry);
			const float r0z = fma_(d2.z, k.iz, k.nrz), r1z = fma_(d3.z, k.iz, k.nrz);
			tl = fmaxf(fmaxf(fminf(l0x, l1x), fminf(l0y, l1y)), fmaxf(fminf(l0z, l1z), k.r.t_min));
			tlx = fminf(fminf(fmaxf(l0x, l1x), fmaxf(l0y, l1y)), fminf(fmaxf(l0z, l1z), k.lim_t));
			tr = fmaxf(fmaxf(fminf(r0x, r1x), fminf(r0y, r1y)), fmaxf(fminf(r0z, r1z), k.r.t_min));
			trx = fminf(fminf(fmaxf(r0x, r1x), fmaxf(r0y, r1y)), fminf(fmaxf(r0z, r1z), k.lim_t));
		} else {
			const float lnx = fma_(NX ? d1.x : d0.x, k.ix, k.nrx), lfx = fma_(NX ? d0.x : d1.x, k.ix, k.nrx);
			const float lny = fma_(NY ? d1.y : d0.y, k.iy, k.nry), lfy = fma_(NY ? d0.y : d1.y, k.iy, k.nry);
			const float lnz = fma_(NZ ? d1.z : d0.z, k.iz, k.nrz), lfz = fma_(NZ ? d0.z : d1.z, k.iz, k.nrz);
			const float rnx = fma_(NX ? d3.x : d2.x, k.ix, k.nrx), rfx = fma_(NX ? d2.x : d3.x, k.ix, k.nrx);
			const float rny = fma_(NY ? d3.y : d2.y, k.iy, k.nry), rfy = fma_(NY ? d2.y : d3.y, k.iy, k.nry);
			const float rnz = fma_(NZ ? d3.z : d2.z, k.iz, k.nrz), rfz = fma_(NZ ? d2.z : d3.z, k.iz, k.nrz);
			tl = fmaxf(fmaxf(lnx, lny), fmaxf(lnz, k.r.t_min));
			tlx = fminf(fminf(lfx, lfy), fminf(lfz, k.lim_t));
			tr = fmaxf(fmaxf(rnx, rny), fmaxf(rnz, k.r.t_min));
			trx = fminf(fminf(rfx, rfy), fminf(rfz, k.lim_t));
		}
		const bool hl = tl <= tlx, hr = tr <= trx;
		const unsigned long long ml = __ballot(hl), mr = __ballot(hr);
		const uint32_t lref = __float_as_uint(d0.w), rref = __float_as_uint(d1.w);
		if (ml != 0ull && mr != 0ull) {
			const unsigned long long lfirst = __ballot(hl && (!hr || tl < tr));
			const bool left_near = 2 * __builtin_popcountll(lfirst) >= __builtin_popcountll(ml | mr);
			stack[k.sp] = left_near ? rref : lref; k.sp++;
			k.cur = left_near ? lref : rref;
		} else if (ml != 0ull) k.cur = lref;
		else if (mr != 0ull) k.cur = rref;
		else pop = true;
	} else {
		// one triangle of the current leaf: d0 = v0|id, d1 = e1|layers, d2 = e2|flags
		if ((__float_as_uint(d1.w) & p.query_mask) != 0u) {
			if (COUNT) n_tris++;
			const RayRegs &r = k.r;
			const float pvx = fma_(r.dy, d2.z, -(r.dz * d2.y));
			const float pvy = fma_(r.dz, d2.x, -(r.dx * d2.z));
			const float pvz = fma_(r.dx, d2.y, -(r.dy * d2.x));
			const float det = dot3(d1.x, d1.y, d1.z, pvx, pvy, pvz);
			if (!(__builtin_fabsf(det) < 1e-8f)) {
				const float inv_det = 1.0f / det;
				const float tvx = r.ox - d0.x, tvy = r.oy - d0.y, tvz = r.oz - d0.z;
				const float u = dot3(tvx, tvy, tvz, pvx, pvy, pvz) * inv_det;
				if (!(u < 0.0f || u > 1.0f)) {
					const float qvx = fma_(tvy, d1.z, -(tvz * d1.y));
					const float qvy = fma_(tvz, d1.x, -(tvx * d1.z));
					const float qvz = fma_(tvx, d1.y, -(tvy * d1.x));
					const float v = dot3(r.dx, r.dy, r.dz, qvx, qvy, qvz) * inv_det;
					if (!(v < 0.0f || u + v > 1.0f)) {
						const float t = dot3(d2.x, d2.y, d2.z, qvx, qvy, qvz) * inv_det;
						const uint32_t id = __float_as_uint(d0.w);
						if (!(t < r.t_min) && (t < k.lim_t || (t == k.lim_t && k.best_slot != 0xFFFFFFFFu && id < k.best_id))) {
							k.best_t = t; k.best_u = u; k.best_v = v; k.best_slot = cur & 0x7FFFFFFFu; k.best_id = id;
							k.lim_t = ANY_HIT ? -FLT_MAX : t;
						}
					}
				}
			}
		}
		if ((__float_as_uint(d2.w) & kLastInLeaf) != 0u) {
			if (ANY_HIT && __ballot(k.lim_t != -FLT_MAX) == 0ull) { k.alive = false; return; } // every lane has its answer
			pop = true;
		} else k.cur = cur + 1u; // next triangle of this leaf
	}
	if (pop) {
		if (k.sp == 0) k.alive = false;
		else { k.sp--; k.cur = stack[k.sp]; }
	}
}

template <int OCT, bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void pkt2_traverse(const TraceParams &p, Pkt2 &a, Pkt2 &b, uint32_t *sa, uint32_t *sb,
		uint32_t &nn_a, uint32_t &nt_a, uint32_t &nn_b, uint32_t &nt_b)
{
	while (a.alive && b.alive) { // both chains in flight: two fetches, one wait, two steps
		const float4 *pa = pkt2_addr(p, a.cur), *pb = pkt2_addr(p, b.cur);
		const float4 a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3];
		const float4 b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3];
		pkt2_step<OCT, ANY_HIT, COUNT>(p, a, sa, a0, a1, a2, a3, nn_a, nt_a);
		pkt2_step<OCT, ANY_HIT, COUNT>(p, b, sb, b0, b1, b2, b3, nn_b, nt_b);
	}
	while (a.alive) {
		const float4 *pa = pkt2_addr(p, a.cur);
		const float4 a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3];
		pkt2_step<OCT, ANY_HIT, COUNT>(p, a, sa, a0, a1, a2, a3, nn_a, nt_a);
	}
	while (b.alive) {
		const float4 *pb = pkt2_addr(p, b.cur);
		const float4 b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3];
		pkt2_step<OCT, ANY_HIT, COUNT>(p, b, sb, b0, b1, b2, b3, nn_b, nt_b);
	}
}

__device__ __forceinline__ void pkt2_finish(const TraceParams &p, const Pkt2 &k, uint64_t ray_idx, bool valid, bool count,
		uint32_t n_nodes, uint32_t n_tris)
{
	if (!valid) return;
	RayRegs r = k.r;
	finish_ray(p, ray_idx, r, k.best_t, k.best_u, k.best_v, k.best_slot);
	if (count) {
		atomicAdd(&p.counters[0], 1ull);
		atomicAdd(&p.counters[1], (unsigned long long)n_tris);
		atomicAdd(&p.counters[2], (unsigned long long)n_nodes);
		if (k.best_slot != 0xFFFFFFFFu) atomicAdd(&p.counters[3], 1ull);
	}
}

template <bool ANY_HIT, bool COUNT>
__global__ __launch_bounds__(MRT_WG) void trace_packet2_kernel(const TraceParams p)
{
	__shared__ uint32_t wave_stack[MRT_WG / MRT_WAVE][2][MRT_PACKET2_STACK];
	uint32_t block = blockIdx.x;
	if (p.xcd_swizzle) {
		const uint32_t per = gridDim.x >> 3;
		if (block < (per << 3)) block = (block & 7u) * per + (block >> 3);
	}
	const uint32_t wave = threadIdx.x / MRT_WAVE, lane = threadIdx.x & (MRT_WAVE - 1);
	// this wave owns the two consecutive 64-ray groups 2w and 2w+1
	const uint64_t g0 = (((uint64_t)block * (MRT_WG / MRT_WAVE) + wave) * 2u) * MRT_WAVE + lane;
	uint64_t idx_a = 0, idx_b = 0; uint32_t pxa = 0, pya = 0, pxb = 0, pyb = 0;
	const bool va = lane_ray_index_g(p, g0, idx_a, pxa, pya);
	const bool vb = lane_ray_index_g(p, g0 + MRT_WAVE, idx_b, pxb, pyb);
	const bool any_a = __ballot(va) != 0ull, any_b = __ballot(vb) != 0ull;
	if (!any_a && !any_b) return;
	RayRegs ra, rb;
	load_ray(p, va ? idx_a : 0, va ? pxa : 0, va ? pya : 0, ra); // dead lanes load ray 0 (always exists)
	load_ray(p, vb ? idx_b : 0, vb ? pxb : 0, vb ? pyb : 0, rb);
	Pkt2 a, b;
	pkt2_init(a, ra, va); pkt2_init(b, rb, vb);
	a.alive = any_a; b.alive = any_b;
	uint32_t nn_a = 0, nt_a = 0, nn_b = 0, nt_b = 0;
	uint32_t *sa = wave_stack[wave][0], *sb = wave_stack[wave][1];

	// one octant for both packets, or the generic slab test
	const unsigned long long sx = __ballot(a.ix < 0.0f) | 0ull, sy = __ballot(a.iy < 0.0f), sz = __ballot(a.iz < 0.0f);
	const unsigned long long tx = __ballot(b.ix < 0.0f), ty = __ballot(b.iy < 0.0f), tz = __ballot(b.iz < 0.0f);
	const unsigned long long full = __ballot(true);
	auto uni = [&](unsigned long long m, unsigned long long n) { return (m == 0ull && n == 0ull) || (m == full && n == full); };
	const bool uniform = uni(sx, tx) && uni(sy, ty) && uni(sz, tz);
	const int oct = uniform ? ((sx ? 1 : 0) | (sy ? 2 : 0) | (sz ? 4 : 0)) : 8;
#define MRT_PKT2(O) case O: pkt2_traverse<O, ANY_HIT, COUNT>(p, a, b, sa, sb, nn_a, nt_a, nn_b, nt_b); break;
	switch (oct) {
		MRT_PKT2(0) MRT_PKT2(1) MRT_PKT2(2) MRT_PKT2(3) MRT_PKT2(4) MRT_PKT2(5) MRT_PKT2(6) MRT_PKT2(7)
		default: pkt2_traverse<8, ANY_HIT, COUNT>(p, a, b, sa, sb, nn_a, nt_a, nn_b, nt_b); break;
	}
#undef MRT_PKT2
	pkt2_finish(p, a, idx_a, va, COUNT, nn_a, nt_a);
	pkt2_finish(p, b, idx_b, vb, COUNT, nn_b, nt_b);
}
