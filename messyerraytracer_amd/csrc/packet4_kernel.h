// packet4_kernel.h — the packet kernel over the 4-wide collapsed layout (Dev4Node).
// Included by kernels.hip (inside namespace mrt, after packet_kernel.h).
//
// Same contract as trace_packet_kernel (one wave = one 64-ray packet, wave-uniform
// walk, scalar fetches, per-wave LDS stack, ballots) but each step fetches FOUR child
// boxes with one 128-byte scalar fetch, so a packet walks about half as many dependent
// fetches.  The packet kernel is latency-bound on exactly that chain (profiles/r01:
// ~940 cycles per step, time inversely proportional to waves per SIMD).
//
// Child order: each hit child gets the entry distance of the first lane that hits it
// (v_readlane), the four 32-bit keys (distance bits with the child slot in the low two
// bits; entry distances are >= t_min >= 0, so float order == unsigned order) go through
// a 5-exchange sorting network of s_min_u32 / s_max_u32, the nearest child is entered,
// the others are pushed farthest first.  Order only affects speed; results are those
// of the lane kernel (exact ties go to the lower triangle id).
#pragma once

#define MRT_PACKET4_STACK 128

template <int OCT, bool ANY_HIT, bool COUNT>
__device__ __forceinline__ void packet4_traverse(const TraceParams &p, const RayRegs &r, uint32_t *stack,
		float &best_t, float &best_u, float &best_v, uint32_t &best_slot, uint32_t &n_nodes, uint32_t &n_tris, uint32_t &n_dead)
{
	const bool degenerate = r.t_min >= r.t_max;
	// finite upper bound: the point box at +inf of an unused child slot must fail `tnear <= tfar`
	float lim_t = degenerate ? -FLT_MAX : fminf(best_t, FLT_MAX);
	const uint32_t lane_id = threadIdx.x & (MRT_WAVE - 1);
	const float ix = safe_inv(r.dx), iy = safe_inv(r.dy), iz = safe_inv(r.dz);
	const float nrx = -(r.ox * ix), nry = -(r.oy * iy), nrz = -(r.oz * iz);
	const float4 *nodes = reinterpret_cast<const float4 *>(p.nodes4);
	const float4 *hot = reinterpret_cast<const float4 *>(p.tri_hot);
	uint32_t best_id = 0xFFFFFFFFu;
	constexpr bool NX = (OCT & 1) != 0, NY = (OCT & 2) != 0, NZ = (OCT & 4) != 0;
	uint32_t sp = 0;
	uint32_t cur = 0;
	bool popped = false;

	for (;;) {
		cur = __builtin_amdgcn_readfirstlane(cur);
		if (cur < kSentinel) {
			const float4 *n = nodes + (size_t)cur * 8u; // uniform address: 2 x s_load_dwordx16
			const float4 b0 = n[0], b1 = n[1], b2 = n[2], b3 = n[3], b4 = n[4], b5 = n[5], refs = n[6];
			if (COUNT) n_nodes++;
			// child c box: min = (m[6c], m[6c+1], m[6c+2]), max = (m[6c+3], m[6c+4], m[6c+5])
			const float m[24] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w,
				b3.x, b3.y, b3.z, b3.w, b4.x, b4.y, b4.z, b4.w, b5.x, b5.y, b5.z, b5.w };
			const uint32_t ref[4] = { __float_as_uint(refs.x), __float_as_uint(refs.y), __float_as_uint(refs.z), __float_as_uint(refs.w) };
			float tn[4]; unsigned long long mk[4];
#pragma unroll
			for (int c = 0; c < 4; c++) {
				const float mnx = m[6 * c], mny = m[6 * c + 1], mnz = m[6 * c + 2];
				const float mxx = m[6 * c + 3], mxy = m[6 * c + 4], mxz = m[6 * c + 5];
				float tnear, tfar;
				if (OCT == 8) {
					const float x0 = fma_(mnx, ix, nrx), x1 = fma_(mxx, ix, nrx);
					const float y0 = fma_(mny, iy, nry), y1 = fma_(mxy, iy, nry);
					const float z0 = fma_(mnz, iz, nrz), z1 = fma_(mxz, iz, nrz);
					tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), r.t_min));
					tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), lim_t));
				} else {
					const float nx_ = fma_(NX ? mxx : mnx, ix, nrx), fx_ = fma_(NX ? mnx : mxx, ix, nrx);
					const float ny_ = fma_(NY ? mxy : mny, iy, nry), fy_ = fma_(NY ? mny : mxy, iy, nry);
					const float nz_ = fma_(NZ ? mxz : mnz, iz, nrz), fz_ = fma_(NZ ? mnz : mxz, iz, nrz);
					tnear = fmaxf(fmaxf(nx_, ny_), fmaxf(nz_, r.t_min));
					tfar = fminf(fminf(fx_, fy_), fminf(fz_, lim_t));
				}
				tn[c] = tnear;
				mk[c] = __ballot(tnear <= tfar); // unused slots hold the point box at +inf: never hit (lim_t is finite)
			}
			const bool any = (mk[0] | mk[1] | mk[2] | mk[3]) != 0ull;
			if (COUNT && popped && !any) n_dead++;
			popped = false;
			if (any) {
				// key = entry distance of the first lane that hits the child, slot in the low 2 bits
				uint32_t key[4];
#pragma unroll
				for (int c = 0; c < 4; c++) {
					const uint32_t lane = mk[c] ? (uint32_t)__builtin_ctzll(mk[c]) : 0u;
					const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tn[c]), (int)lane);
					key[c] = mk[c] ? ((d & ~3u) | (uint32_t)c) : 0xFFFFFFFFu;
				}
#define MRT_CSWAP(a, b) { const uint32_t lo_ = key[a] < key[b] ? key[a] : key[b]; const uint32_t hi_ = key[a] < key[b] ? key[b] : key[a]; key[a] = lo_; key[b] = hi_; }
				MRT_CSWAP(0, 1) MRT_CSWAP(2, 3) MRT_CSWAP(0, 2) MRT_CSWAP(1, 3) MRT_CSWAP(1, 2)
#undef MRT_CSWAP
				// lane c of vref holds ref[c]: the sorted slot numbers index it with v_readlane
				// (all 64 lanes are alive here: lanes without a ray stay in the wave as dead lanes,
				// because v_readlane reads a lane's register whether or not it is in EXEC)
				const int vref = (int)(lane_id == 0u ? ref[0] : (lane_id == 1u ? ref[1] : (lane_id == 2u ? ref[2] : ref[3])));
				auto pick = [&](uint32_t k) { return (uint32_t)__builtin_amdgcn_readlane(vref, (int)(k & 3u)); };
				// farthest first, so the nearest pushed child is popped first
				if (key[3] != 0xFFFFFFFFu) { stack[sp] = pick(key[3]); sp++; }
				if (key[2] != 0xFFFFFFFFu) { stack[sp] = pick(key[2]); sp++; }
				if (key[1] != 0xFFFFFFFFu) { stack[sp] = pick(key[1]); sp++; }
				cur = pick(key[0]);
				continue;
			}
		} else {
			uint32_t slot = cur & 0x7FFFFFFFu;
			bool last;
			do {
				const float4 *t3 = hot + (size_t)slot * 3u; // uniform address
				const float4 q0 = t3[0], q1 = t3[1], q2 = t3[2];
				last = (__float_as_uint(q2.w) & kLastInLeaf) != 0u;
				if ((__float_as_uint(q1.w) & p.query_mask) != 0u) {
					if (COUNT) n_tris++;
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
								const uint32_t id = __float_as_uint(q0.w);
								if (!(t < r.t_min) && (t < lim_t || (t == lim_t && best_slot != 0xFFFFFFFFu && id < best_id))) {
									best_t = t; best_u = u; best_v = v; best_slot = slot; best_id = id;
									lim_t = ANY_HIT ? -FLT_MAX : t;
								}
							}
						}
					}
				}
				slot++;
			} while (!last);
			if (ANY_HIT && __ballot(lim_t != -FLT_MAX) == 0ull) break;
		}
		if (sp == 0) break;
		sp--; cur = stack[sp];
		popped = true;
	}
}

template <bool ANY_HIT, bool COUNT>
__global__ __launch_bounds__(MRT_WG) void trace_packet4_kernel(const TraceParams p)
{
	__shared__ uint32_t wave_stack[MRT_WG / MRT_WAVE][MRT_PACKET4_STACK];
	uint32_t block = blockIdx.x;
	if (p.xcd_swizzle) {
		const uint32_t per = gridDim.x >> 3;
		if (block < (per << 3)) block = (block & 7u) * per + (block >> 3);
	}
	uint64_t ray_idx = 0; uint32_t px = 0, py = 0;
	// Lanes without a ray (ragged batch / tile edge) stay in the wave as dead lanes: they load
	// ray 0, get an empty interval (t_min > t_max) so they never hit a box, and store nothing.
	const bool valid = lane_ray_index(p, block, ray_idx, px, py);
	if (__ballot(valid) == 0ull) return;
	if (!valid) { ray_idx = 0; px = 0; py = 0; }
	RayRegs r;
	load_ray(p, ray_idx, px, py, r);
	if (!valid) { r.t_min = 1.0f; r.t_max = 0.0f; }

	float best_t = r.t_max, best_u = 0.0f, best_v = 0.0f;
	uint32_t best_slot = 0xFFFFFFFFu;
	uint32_t n_nodes = 0, n_tris = 0, n_dead = 0;
	uint32_t *stack = wave_stack[threadIdx.x / MRT_WAVE];

	const unsigned long long live = __ballot(true);
	const unsigned long long sx = __ballot(safe_inv(r.dx) < 0.0f), sy = __ballot(safe_inv(r.dy) < 0.0f),
			sz = __ballot(safe_inv(r.dz) < 0.0f);
	const bool uniform = (sx == 0ull || sx == live) && (sy == 0ull || sy == live) && (sz == 0ull || sz == live);
	const int oct = uniform ? ((sx ? 1 : 0) | (sy ? 2 : 0) | (sz ? 4 : 0)) : 8;
#define MRT_PKT4(O) case O: packet4_traverse<O, ANY_HIT, COUNT>(p, r, stack, best_t, best_u, best_v, best_slot, n_nodes, n_tris, n_dead); break;
	switch (oct) {
		MRT_PKT4(0) MRT_PKT4(1) MRT_PKT4(2) MRT_PKT4(3) MRT_PKT4(4) MRT_PKT4(5) MRT_PKT4(6) MRT_PKT4(7)
		default: packet4_traverse<8, ANY_HIT, COUNT>(p, r, stack, best_t, best_u, best_v, best_slot, n_nodes, n_tris, n_dead); break;
	}
#undef MRT_PKT4

	if (!valid) return;
	finish_ray(p, ray_idx, r, best_t, best_u, best_v, best_slot);

	if (COUNT) {
		atomicAdd(&p.counters[0], 1ull);
		atomicAdd(&p.counters[1], (unsigned long long)n_tris);
		atomicAdd(&p.counters[2], (unsigned long long)n_nodes);
		if (best_slot != 0xFFFFFFFFu) atomicAdd(&p.counters[3], 1ull);
		atomicAdd(&p.counters[5], (unsigned long long)n_dead);
	}
}
