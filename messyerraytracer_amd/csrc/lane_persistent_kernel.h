// lane_persistent_kernel.h — one lane = one ray, persistent waves with ray refill.
// Included by kernels.hip (inside namespace mrt, after the common helpers).
//
// For incoherent batches (config C4).  profiles/r01_c4lane: with one fixed ray per lane a
// wave executes ~540 loop iterations for rays that need ~89 (16 % SIMD efficiency: the
// wave runs until its slowest ray ends) at 5 waves per SIMD (LDS stack), 74 % of the time
// waiting on the divergent node fetches.  This kernel
//   * keeps the grid resident (waves x CUs that fit) and lets every wave pull rays from
//     one global counter: when at least `refill` lanes have finished, the wave stores
//     their hits and hands them new rays (ballot + mbcnt prefix, one atomicAdd per refill);
//   * keeps only the top `lds_depth` stack entries per lane in LDS ([depth][lane] layout)
//     and spills deeper entries to a per-lane slice of an HBM scratch buffer, so LDS no
//     longer limits occupancy (16 entries = 4 KB per wave);
//   * alternates a NODE phase and a LEAF phase under wave-uniform control: the node phase
//     steps every lane that stands at an inner node and ends as soon as `leaf_wait` lanes
//     stand at a leaf (or nobody is at an inner node); the leaf phase then intersects those
//     leaves.  The plain while-while loop of trace_lane_kernel keeps a lane that reached its
//     leaf waiting until the LAST lane of the wave has reached one: PMC counters put its node
//     steps at 11 busy lanes of 64.  Node steps are 93 % of the arithmetic of a ray and leaf
//     tests 7 %, so it pays to run the cheap leaf phase often, at low occupancy, to keep the
//     expensive node phase dense.
// The arithmetic is trace_lane_kernel's (same operations, same tie rule).
//
// WIDE4: the same walk over the 4-wide collapse of the BVH (Dev4Node, one 128-byte cache line
// per node; tools/ubench/tcp_rate.hip: a divergent fetch is paid per cache line, so a 4-wide
// node costs what a 2-wide node costs and a ray needs half as many).  Children are visited
// nearest first (sorting network on packed distance|slot keys); the order never changes a
// result.
//
// WIDTH 8: the 8-wide compressed collapse (Dev8Node: 8-bit child boxes on a per-node grid, decoded
// with fma(q, step, origin) to boxes that contain the exact ones).  Still one line per step, and a
// ray needs fewer again; the price is the decode arithmetic.
#pragma once

#define MRT_RAY_CHUNK 256u // most rays a wave reserves per atomic on a ray counter (PersistParams::chunk)

struct PersistParams {
	unsigned long long *next_ray; // 8 ray counters, one per region of the batch, 16 u64 apart (zeroed before the launch)
	uint32_t *overflow;           // [depth - lds_depth][global thread] spill area
	uint32_t overflow_stride;     // = total threads of the launch
	uint32_t lds_depth;           // stack entries per lane kept in LDS
	uint32_t refill;              // refill when at least this many lanes are idle
	uint32_t leaf_wait;           // leave the node phase when this many lanes stand at a leaf
	uint32_t chunk;               // rays a wave reserves at a time: MRT_RAY_CHUNK, less for batches that would
	                              // otherwise give a wave only a few chunks (the last ones finish unevenly)
};

// Waves per SIMD the register allocator has to leave room for.  The flat forms need 75-92 VGPRs (5-6 waves) by
// themselves; the two-level forms come to 118-125 (4 waves) and gain 6 % when held to 96 (2^22 incoherent rays
// on C5: 8.35 -> 7.8 ms).  Asking for 6 costs every form (C4 8-wide: +3 %, two-level: +8 %).
#ifndef MRT_PERSIST_WPE
#define MRT_PERSIST_WPE 5
#endif
#define MRT_PERSIST_ATTR __attribute__((amdgpu_waves_per_eu(MRT_PERSIST_WPE, 8)))

// byte k of a packed word as a float (v_cvt_f32_ubyteK)
__device__ __forceinline__ float ubyte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }

// TL: a two-level scene (two_level_kernel.h): p.nodes holds the TLAS and every BLAS, a TLAS leaf is a run of
// DevInstance rows, a lane inside an instance walks with its mesh-space ray and the marker kInstanceReturn on
// its stack takes it back to the world ray.  WIDTH 2: both levels 2-wide.  WIDTH 8: the TLAS 2-wide, every
// BLAS in the 8-wide compressed layout (p.nodes8, p.leaf_box; DevInstance::root8).
// sum of a per-lane count over the wave (counting builds)
__device__ __forceinline__ unsigned long long wave_sum(uint32_t v)
{
	unsigned long long s = v;
	for (int m = 1; m < MRT_WAVE; m <<= 1) s += __shfl_xor(s, m);
	return s;
}

// COUNT: the counting build (mrt_options.count_visits): per ray node steps (= cache lines fetched: one 64- or 128-byte
// line per step whatever the width), triangle rows tested and, for the 8-wide walk, exact leaf boxes read.
template <bool ANY_HIT, int WIDTH, bool TL = false, bool COUNT = false> // WIDTH: children per node step = 2 (DevNode), 4 (Dev4Node) or 8 (Dev8Node)
__global__ __launch_bounds__(MRT_WG) MRT_PERSIST_ATTR void trace_lane_persistent_kernel(const TraceParams p, const PersistParams q)
{
	uint32_t n_rays = 0, n_hits = 0, n_nodes = 0, n_tris = 0, n_boxchk = 0; // COUNT: this lane's totals over all its rays
	static_assert(!TL || WIDTH == 2 || WIDTH == 8, "two-level scenes: 2-wide, or 8-wide inside the instances (the TLAS is always 2-wide)");
	constexpr uint32_t kNode = TL ? kInstanceReturn : kSentinel; // refs below this are inner nodes
	extern __shared__ uint32_t lds_stack[];
	if (skip_launch(p)) return;
	const uint32_t lane = threadIdx.x & (MRT_WAVE - 1), wave = threadIdx.x / MRT_WAVE;
	const uint32_t gtid = blockIdx.x * MRT_WG + threadIdx.x;
	const uint32_t lds_base = wave * (q.lds_depth * MRT_WAVE) + lane;
	const float4 *nodes = reinterpret_cast<const float4 *>(p.nodes);
	const float4 *nodes4 = reinterpret_cast<const float4 *>(p.nodes4);
	const float4 *nodes8 = reinterpret_cast<const float4 *>(p.nodes8);
	const float4 *leaf_box = reinterpret_cast<const float4 *>(p.leaf_box);
	const float4 *hot = reinterpret_cast<const float4 *>(p.tri_hot);
	// Flat scenes that have the unified row array (packet_rows_kernel.h): triangles are read from its 64-byte rows
	// {v0,id | e1,layers | e2,flags | normal} instead of the 48-byte TriHot rows, 3 of 8 of which straddle a 128-byte line
	// (this kernel is bound by L2 requests per ray, DESIGN 4.3: 1.0 instead of 1.375 per triangle test), and the
	// winner's normal comes from the same row.  Same values, same arithmetic.
	const bool tri_rows = !TL && p.row_array != nullptr;
	const float4 *tri_base = tri_rows ? reinterpret_cast<const float4 *>(p.row_array) + (size_t)p.n_nodes * 4u : hot;
	const uint32_t tri_stride = tri_rows ? 4u : 3u;

	// per-lane ray state
	RayRegs r = {};
	uint64_t ray_idx = 0;
	float ix = 0, iy = 0, iz = 0, nrx = 0, nry = 0, nrz = 0;
	float best_t = 0, best_u = 0, best_v = 0;
	uint32_t best_slot = 0xFFFFFFFFu, best_id = 0xFFFFFFFFu;
	// TL: the ray being walked (the world ray, or its image in the mesh space of the instance the lane is in)
	float cox = 0, coy = 0, coz = 0, cdx = 0, cdy = 0, cdz = 0;
	uint32_t id_base = 0u, cur_inst = 0u, best_inst = 0u;
	bool in_blas = false;
	const float4 *inst = reinterpret_cast<const float4 *>(p.instances);
	uint32_t cur = kSentinel; // kSentinel = this lane has no work
	uint32_t depth = 0;
	bool has_ray = false;
	bool exhausted = false;   // wave-uniform: the ray counter ran past the batch
	uint64_t range_next = 0, range_end = 0; // wave-uniform: rays this wave has reserved and not yet handed out
	uint32_t region, regions_tried = 0;     // wave-uniform: the region this wave draws from; regions found empty
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(region)); // the XCD this wave runs on (speed only)
	region &= 7u;

	auto push = [&](uint32_t ref) {
		if (depth < q.lds_depth) lds_stack[lds_base + depth * MRT_WAVE] = ref;
		else q.overflow[(size_t)(depth - q.lds_depth) * q.overflow_stride + gtid] = ref;
		depth++;
	};
	auto pop = [&]() -> uint32_t {
		if (depth == 0) return kSentinel;
		depth--;
		return depth < q.lds_depth ? lds_stack[lds_base + depth * MRT_WAVE]
		                           : q.overflow[(size_t)(depth - q.lds_depth) * q.overflow_stride + gtid];
	};

	for (;;) {
		// ---- retire finished lanes, hand out new rays ----
		const bool idle = cur == kSentinel;
		if (idle && has_ray) {
			// (not finish_ray(): its output-format branch around these loads measured 11 % slower here,
			// 8.6 against 7.7 ms at C4; the lookups are unconditional in this kernel)
			int32_t prim = -1; float nx = 0.0f, ny = 0.0f, nz = 0.0f; uint32_t layers = 0u;
			if (TL) finish_two_level_ray(p, ray_idx, r, best_t, best_u, best_v, best_slot, best_id, best_inst);
			else if (best_slot != 0xFFFFFFFFu) {
				const float4 *w3 = tri_base + (size_t)best_slot * tri_stride;
				prim = (int32_t)__float_as_uint(w3[0].w);
				layers = __float_as_uint(w3[1].w);
				const float4 nn = tri_rows ? w3[3] : reinterpret_cast<const float4 *>(p.tri_cold)[best_slot];
				nx = nn.x; ny = nn.y; nz = nn.z;
			}
			if (!TL) store_hit(p, ray_idx, r, best_t, prim, best_u, best_v, nx, ny, nz, layers, best_slot);
			if (COUNT) { n_rays++; if (best_slot != 0xFFFFFFFFu) n_hits++; }
			has_ray = false;
		}
		const unsigned long long idle_mask = __ballot(idle);
		if (!exhausted && idle_mask != 0ull) {
			// Rays come from a wave-private range [range_next, range_end) that is restocked MRT_RAY_CHUNK
			// rays at a time from the global counter: one device-scope atomic on ONE address costs about
			// 10 ns chip-wide, and one atomic per refill (a million of them at C4) was what bounded the kernel.
			// The batch is cut into 8 regions with one counter each, and a wave starts in the region of the
			// XCD it runs on: with sorted rays an XCD then works on one part of the scene and its 4 MB L2
			// keeps that part of the BVH (the eight L2s hold different nodes instead of the same ones).  A
			// region that runs dry sends its waves to the next one, so the load still balances.
			const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle_mask);
			const uint64_t avail = range_end - range_next; // wave-uniform
			uint64_t fresh_lo = 0, fresh_hi = 0;           // a newly reserved chunk (if one is needed and left)
			if (avail < n_idle) {
				while (regions_tried < 8u) {
					const uint64_t lo = p.count * region / 8u, hi = p.count * (region + 1u) / 8u;
					unsigned long long base = 0;
					if (lane == (uint32_t)__builtin_ctzll(idle_mask)) base = atomicAdd(q.next_ray + region * 16u, (unsigned long long)q.chunk);
					base = __shfl(base, __builtin_ctzll(idle_mask));
					if (lo + base < hi) {
						fresh_lo = lo + base;
						fresh_hi = fresh_lo + q.chunk < hi ? fresh_lo + q.chunk : hi;
						break;
					}
					region = (region + 1u) & 7u; regions_tried++; // this region is handed out completely
				}
			}
			uint64_t g = ~0ull; // this lane's new ray (none)
			if (idle) {
				const uint32_t rank = (uint32_t)__builtin_popcountll(idle_mask & ((1ull << lane) - 1ull));
				if (rank < avail) g = range_next + rank;
				else if (fresh_lo + (rank - avail) < fresh_hi) g = fresh_lo + (rank - avail);
			}
			if (avail < n_idle) {
				const uint64_t want = n_idle - avail, got = fresh_hi - fresh_lo;
				range_next = fresh_lo + (want < got ? want : got); range_end = fresh_hi;
				if (regions_tried == 8u) exhausted = true; // every region is handed out: no lane will get a ray again
			} else range_next += n_idle;
			if (idle) {
				if (g != ~0ull) {
					ray_idx = p.perm ? (uint64_t)p.perm[g] : g;
					load_ray(p, ray_idx, 0, 0, r);
					has_ray = true;
					best_t = r.t_max; best_u = 0.0f; best_v = 0.0f; best_slot = 0xFFFFFFFFu; best_id = 0xFFFFFFFFu;
					depth = 0;
					if (r.t_min >= r.t_max) cur = kSentinel; // degenerate: a miss, retired on the next round
					else {
						ix = safe_inv(r.dx); iy = safe_inv(r.dy); iz = safe_inv(r.dz);
						nrx = -(r.ox * ix); nry = -(r.oy * iy); nrz = -(r.oz * iz);
						if (TL) { cox = r.ox; coy = r.oy; coz = r.oz; cdx = r.dx; cdy = r.dy; cdz = r.dz; in_blas = false; }
						cur = 0;
					}
				}
			}
		}
		if (__ballot(has_ray) == 0ull) { // no lane holds a ray and none is left to fetch
			if (COUNT) {
				const unsigned long long a = wave_sum(n_rays), b = wave_sum(n_tris), c = wave_sum(n_nodes), d = wave_sum(n_hits), e = wave_sum(n_boxchk);
				if (lane == 0u) {
					atomicAdd(&p.counters[kCntRays], a); atomicAdd(&p.counters[kCntTris], b); atomicAdd(&p.counters[kCntNodes], c);
					atomicAdd(&p.counters[kCntHits], d); atomicAdd(&p.counters[kCntWaveNodeFetch], c); atomicAdd(&p.counters[kCntWaveTriFetch], b);
					atomicAdd(&p.counters[kCntLeafBoxChecks], e);
				}
			}
			break;
		}

		// ---- traverse until enough lanes have finished to make a refill worthwhile ----
		for (;;) {
			// NODE phase (wave-uniform loop, lanes at an inner node take the step)
			while (__ballot(cur < kSentinel) != 0ull) {
				if (TL && cur == kInstanceReturn) { // the BLAS is done: back to the world ray
					cox = r.ox; coy = r.oy; coz = r.oz; cdx = r.dx; cdy = r.dy; cdz = r.dz;
					ix = safe_inv(cdx); iy = safe_inv(cdy); iz = safe_inv(cdz);
					nrx = -(cox * ix); nry = -(coy * iy); nrz = -(coz * iz);
					in_blas = false;
					cur = pop();
				}
				if (WIDTH == 8 && (!TL || in_blas) && cur < kNode) { // 8-wide compressed node: one 128-byte line, 96 bytes read
					if (COUNT) n_nodes++;
					const float4 *n = nodes8 + (size_t)cur * 8u;
					const float4 h = n[0], qa = n[1], qb = n[2], qc = n[3], ra = n[4], rb = n[5];
					const uint32_t meta = __float_as_uint(h.w);
					const float sx = __uint_as_float((meta & 0xFFu) << 23), sy = __uint_as_float(((meta >> 8) & 0xFFu) << 23),
							sz = __uint_as_float(((meta >> 16) & 0xFFu) << 23);
					const uint32_t n_children = meta >> 24;
					// [axis][child] bytes: qlo x = qa.xy, qlo y = qa.zw, qlo z = qb.xy, qhi x = qb.zw, qhi y = qc.xy, qhi z = qc.zw
					const uint32_t lox[2] = { __float_as_uint(qa.x), __float_as_uint(qa.y) }, loy[2] = { __float_as_uint(qa.z), __float_as_uint(qa.w) };
					const uint32_t loz[2] = { __float_as_uint(qb.x), __float_as_uint(qb.y) }, hix[2] = { __float_as_uint(qb.z), __float_as_uint(qb.w) };
					const uint32_t hiy[2] = { __float_as_uint(qc.x), __float_as_uint(qc.y) }, hiz[2] = { __float_as_uint(qc.z), __float_as_uint(qc.w) };
					const uint32_t ref[8] = { __float_as_uint(ra.x), __float_as_uint(ra.y), __float_as_uint(ra.z), __float_as_uint(ra.w),
						__float_as_uint(rb.x), __float_as_uint(rb.y), __float_as_uint(rb.z), __float_as_uint(rb.w) };
					const float lim = best_t;
					// The near plane of an axis is the box's low or high coordinate by the sign of the ray's
					// direction: choose between the packed words once (4 children per select) instead of a
					// min and a max per child and axis.  Same values: fma is monotone in the box coordinate.
					const bool ngx = ix < 0.0f, ngy = iy < 0.0f, ngz = iz < 0.0f;
					const uint32_t nrw_x[2] = { ngx ? hix[0] : lox[0], ngx ? hix[1] : lox[1] }, far_x[2] = { ngx ? lox[0] : hix[0], ngx ? lox[1] : hix[1] };
					const uint32_t nrw_y[2] = { ngy ? hiy[0] : loy[0], ngy ? hiy[1] : loy[1] }, far_y[2] = { ngy ? loy[0] : hiy[0], ngy ? loy[1] : hiy[1] };
					const uint32_t nrw_z[2] = { ngz ? hiz[0] : loz[0], ngz ? hiz[1] : loz[1] }, far_z[2] = { ngz ? loz[0] : hiz[0], ngz ? loz[1] : hiz[1] };
					uint32_t key[8];
#pragma unroll
					for (int c = 0; c < 8; c++) {
						// decode first (the builder verified exactly these values), then the usual slab arithmetic
						const float tnx = fma_(fma_(ubyte_f(nrw_x[c >> 2], c & 3), sx, h.x), ix, nrx), tfx = fma_(fma_(ubyte_f(far_x[c >> 2], c & 3), sx, h.x), ix, nrx);
						const float tny = fma_(fma_(ubyte_f(nrw_y[c >> 2], c & 3), sy, h.y), iy, nry), tfy = fma_(fma_(ubyte_f(far_y[c >> 2], c & 3), sy, h.y), iy, nry);
						const float tnz = fma_(fma_(ubyte_f(nrw_z[c >> 2], c & 3), sz, h.z), iz, nrz), tfz = fma_(fma_(ubyte_f(far_z[c >> 2], c & 3), sz, h.z), iz, nrz);
						const float tnear = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, r.t_min));
						const float tfar = fminf(fminf(tfx, tfy), fminf(tfz, lim));
						key[c] = ((uint32_t)c < n_children && tnear <= tfar) ? ((__float_as_uint(tnear) & ~7u) | (uint32_t)c) : 0xFFFFFFFFu;
					}
					// front-to-back: 19-exchange sorting network on the packed keys, nearest child next, the
					// others pushed farthest first (pushing them unsorted measured 3-6 % slower at C4)
#define MRT_CS(a, b) { const uint32_t lo_ = min(key[a], key[b]), hi_ = max(key[a], key[b]); key[a] = lo_; key[b] = hi_; }
					MRT_CS(0, 1) MRT_CS(2, 3) MRT_CS(4, 5) MRT_CS(6, 7) MRT_CS(0, 2) MRT_CS(1, 3) MRT_CS(4, 6) MRT_CS(5, 7)
					MRT_CS(1, 2) MRT_CS(5, 6) MRT_CS(0, 4) MRT_CS(3, 7) MRT_CS(1, 5) MRT_CS(2, 6) MRT_CS(1, 4) MRT_CS(3, 6)
					MRT_CS(2, 4) MRT_CS(3, 5) MRT_CS(3, 4)
#undef MRT_CS
					if (key[0] == 0xFFFFFFFFu) cur = pop();
					else {
						// ref of the slot in a key's low 3 bits: a select tree on scalars (an indexed array goes to scratch)
						auto pick = [&](uint32_t k) {
							const bool b0 = (k & 1u) != 0u, b1 = (k & 2u) != 0u, b2 = (k & 4u) != 0u;
							const uint32_t p01 = b0 ? ref[1] : ref[0], p23 = b0 ? ref[3] : ref[2], p45 = b0 ? ref[5] : ref[4], p67 = b0 ? ref[7] : ref[6];
							const uint32_t lo4 = b1 ? p23 : p01, hi4 = b1 ? p67 : p45;
							return b2 ? hi4 : lo4;
						};
						if (key[7] != 0xFFFFFFFFu) push(pick(key[7]));
						if (key[6] != 0xFFFFFFFFu) push(pick(key[6]));
						if (key[5] != 0xFFFFFFFFu) push(pick(key[5]));
						if (key[4] != 0xFFFFFFFFu) push(pick(key[4]));
						if (key[3] != 0xFFFFFFFFu) push(pick(key[3]));
						if (key[2] != 0xFFFFFFFFu) push(pick(key[2]));
						if (key[1] != 0xFFFFFFFFu) push(pick(key[1]));
						cur = pick(key[0]);
					}
				}
				if (WIDTH == 4 && cur < kSentinel) { // 4-wide collapse: one 128-byte line per step
					if (COUNT) n_nodes++;
					const float4 *n = nodes4 + (size_t)cur * 8u;
					const float4 b0 = n[0], b1 = n[1], b2 = n[2], b3 = n[3], b4 = n[4], b5 = n[5], refs = n[6];
					// child c box: min = (m[6c], m[6c+1], m[6c+2]), max = (m[6c+3], m[6c+4], m[6c+5])
					const float m[24] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w,
						b3.x, b3.y, b3.z, b3.w, b4.x, b4.y, b4.z, b4.w, b5.x, b5.y, b5.z, b5.w };
					// finite upper bound: the point box at +inf of an unused child slot must fail `tnear <= tfar`
					const float lim = fminf(best_t, FLT_MAX);
					uint32_t key[4];
#pragma unroll
					for (int c = 0; c < 4; c++) {
						const float x0 = fma_(m[6 * c], ix, nrx), x1 = fma_(m[6 * c + 3], ix, nrx);
						const float y0 = fma_(m[6 * c + 1], iy, nry), y1 = fma_(m[6 * c + 4], iy, nry);
						const float z0 = fma_(m[6 * c + 2], iz, nrz), z1 = fma_(m[6 * c + 5], iz, nrz);
						const float tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), r.t_min));
						const float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), lim));
						// key: entry distance (>= t_min >= 0: float order == unsigned order) with the slot in the
						// low two bits; the order of the walk only affects speed, never the result (tie rule)
						key[c] = tnear <= tfar ? ((__float_as_uint(tnear) & ~3u) | (uint32_t)c) : 0xFFFFFFFFu;
					}
#define MRT_CSWAP(a, b) { const uint32_t lo_ = min(key[a], key[b]), hi_ = max(key[a], key[b]); key[a] = lo_; key[b] = hi_; }
					MRT_CSWAP(0, 1) MRT_CSWAP(2, 3) MRT_CSWAP(0, 2) MRT_CSWAP(1, 3) MRT_CSWAP(1, 2)
#undef MRT_CSWAP
					if (key[0] == 0xFFFFFFFFu) cur = pop();
					else {
						const uint32_t r0 = __float_as_uint(refs.x), r1 = __float_as_uint(refs.y), r2 = __float_as_uint(refs.z), r3 = __float_as_uint(refs.w);
						auto pick = [&](uint32_t k) { const uint32_t s = k & 3u; return s == 0u ? r0 : (s == 1u ? r1 : (s == 2u ? r2 : r3)); };
						// farthest first, so the nearest pushed child is popped first
						if (key[3] != 0xFFFFFFFFu) push(pick(key[3]));
						if (key[2] != 0xFFFFFFFFu) push(pick(key[2]));
						if (key[1] != 0xFFFFFFFFu) push(pick(key[1]));
						cur = pick(key[0]);
					}
				}
				if ((WIDTH == 2 || (TL && !in_blas)) && cur < kNode) { // dual-AABB node: glsl:243-318 (TL: every TLAS node)
					if (COUNT) n_nodes++;
					const float4 *n = nodes + (size_t)cur * 4u;
					const float4 a = n[0], b = n[1], c = n[2], d = n[3];
					const float l0x = fma_(a.x, ix, nrx), l1x = fma_(b.x, ix, nrx);
					const float l0y = fma_(a.y, iy, nry), l1y = fma_(b.y, iy, nry);
					const float l0z = fma_(a.z, iz, nrz), l1z = fma_(b.z, iz, nrz);
					const float r0x = fma_(c.x, ix, nrx), r1x = fma_(d.x, ix, nrx);
					const float r0y = fma_(c.y, iy, nry), r1y = fma_(d.y, iy, nry);
					const float r0z = fma_(c.z, iz, nrz), r1z = fma_(d.z, iz, nrz);
					const float tl = fmaxf(fmaxf(fminf(l0x, l1x), fminf(l0y, l1y)), fmaxf(fminf(l0z, l1z), r.t_min));
					const float tlx = fminf(fminf(fmaxf(l0x, l1x), fmaxf(l0y, l1y)), fminf(fmaxf(l0z, l1z), best_t));
					const float tr = fmaxf(fmaxf(fminf(r0x, r1x), fminf(r0y, r1y)), fmaxf(fminf(r0z, r1z), r.t_min));
					const float trx = fminf(fminf(fmaxf(r0x, r1x), fmaxf(r0y, r1y)), fminf(fmaxf(r0z, r1z), best_t));
					const bool hl = tl <= tlx, hr = tr <= trx;
					const uint32_t lref = __float_as_uint(a.w), rref = __float_as_uint(b.w);
					if (hl && hr) {
						const bool left_near = tl < tr;
						push(left_near ? rref : lref);
						cur = left_near ? lref : rref;
					} else if (hl) cur = lref;
					else if (hr) cur = rref;
					else cur = pop();
				}
				if ((uint32_t)__builtin_popcountll(__ballot(cur >= kLeafBit)) >= q.leaf_wait) break;
			}
			// LEAF phase: every lane at a leaf intersects that leaf (glsl:166-192), then pops
			if (TL && cur >= kLeafBit && !in_blas) {
				// TLAS leaf: a run of instances, one at a time (tiny_bvh.h:3320-3360); the rest of the run goes back on the stack
				const uint32_t slot0 = cur & 0x7FFFFFFFu;
				const float4 *row = inst + (size_t)slot0 * 8u;
				const float4 m0 = row[0], m1 = row[1], m2 = row[2], meta = row[5];
				if ((__float_as_uint(row[6].x) & 1u) == 0u) push(kLeafBit | (slot0 + 1u));
				if ((__float_as_uint(meta.w) & p.query_mask) != 0u) {
					cox = fma_(m0.x, r.ox, fma_(m0.y, r.oy, fma_(m0.z, r.oz, m0.w)));
					coy = fma_(m1.x, r.ox, fma_(m1.y, r.oy, fma_(m1.z, r.oz, m1.w)));
					coz = fma_(m2.x, r.ox, fma_(m2.y, r.oy, fma_(m2.z, r.oz, m2.w)));
					cdx = fma_(m0.x, r.dx, fma_(m0.y, r.dy, m0.z * r.dz));
					cdy = fma_(m1.x, r.dx, fma_(m1.y, r.dy, m1.z * r.dz));
					cdz = fma_(m2.x, r.dx, fma_(m2.y, r.dy, m2.z * r.dz));
					ix = safe_inv(cdx); iy = safe_inv(cdy); iz = safe_inv(cdz);
					nrx = -(cox * ix); nry = -(coy * iy); nrz = -(coz * iz);
					push(kInstanceReturn);
					in_blas = true; cur_inst = slot0;
					id_base = __float_as_uint(meta.z);
					cur = WIDTH == 8 ? __float_as_uint(row[6].z) : __float_as_uint(meta.y); // the BLAS root in the layout walked
				} else cur = pop();
			} else if (cur >= kLeafBit) {
				// the ray the triangles are tested with: the world ray, or (TL) the mesh-space ray of the instance
				const float tox = TL ? cox : r.ox, toy = TL ? coy : r.oy, toz = TL ? coz : r.oz;
				const float tdx = TL ? cdx : r.dx, tdy = TL ? cdy : r.dy, tdz = TL ? cdz : r.dz;
				uint32_t slot = cur & 0x7FFFFFFFu;
				const uint32_t leaf_first = slot;
				bool last;
				do {
					const float4 *t3 = tri_base + (size_t)slot * tri_stride;
					const float4 q0 = t3[0], q1 = t3[1], q2 = t3[2];
					last = (__float_as_uint(q2.w) & kLastInLeaf) != 0u;
					if (TL || (__float_as_uint(q1.w) & p.query_mask) != 0u) { // TL: the mask was applied to the instance
						if (COUNT) n_tris++;
						const float pvx = fma_(tdy, q2.z, -(tdz * q2.y));
						const float pvy = fma_(tdz, q2.x, -(tdx * q2.z));
						const float pvz = fma_(tdx, q2.y, -(tdy * q2.x));
						const float det = dot3(q1.x, q1.y, q1.z, pvx, pvy, pvz);
						if (!(__builtin_fabsf(det) < 1e-8f)) {
							const float inv_det = 1.0f / det;
							const float tvx = tox - q0.x, tvy = toy - q0.y, tvz = toz - q0.z;
							const float u = dot3(tvx, tvy, tvz, pvx, pvy, pvz) * inv_det;
							if (!(u < 0.0f || u > 1.0f)) {
								const float qvx = fma_(tvy, q1.z, -(tvz * q1.y));
								const float qvy = fma_(tvz, q1.x, -(tvx * q1.z));
								const float qvz = fma_(tvx, q1.y, -(tvy * q1.x));
								const float v = dot3(tdx, tdy, tdz, qvx, qvy, qvz) * inv_det;
								if (!(v < 0.0f || u + v > 1.0f)) {
									const float t = dot3(q2.x, q2.y, q2.z, qvx, qvy, qvz) * inv_det;
									const uint32_t id = (TL ? id_base : 0u) + __float_as_uint(q0.w); // TL: flat id
									if (!(t < r.t_min) && (t < best_t || (t == best_t && best_slot != 0xFFFFFFFFu && id < best_id))) {
										// 8-wide: the quantised boxes that led here are looser than the exact ones; accept
										// the hit only if the ray passes the slab test on the leaf's exact box, as it does
										// in the 2-wide walk (nested boxes: that is passing every ancestor's test too).
										// The far limit is the ray's own t_max, not the best hit so far: which hits are
										// accepted must not depend on the order the leaves were reached in (two hits at
										// the same t in different leaves: the lower id has to win whichever came first).
										bool entered = true;
										if (WIDTH == 8) {
											if (COUNT) n_boxchk++;
											const float4 *lb = leaf_box + (size_t)leaf_first * 2u;
											const float4 mn = lb[0], mx = lb[1];
											const float x0 = fma_(mn.x, ix, nrx), x1 = fma_(mx.x, ix, nrx);
											const float y0 = fma_(mn.y, iy, nry), y1 = fma_(mx.y, iy, nry);
											const float z0 = fma_(mn.z, iz, nrz), z1 = fma_(mx.z, iz, nrz);
											const float tnear = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), r.t_min));
											const float tfar = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), r.t_max));
											entered = tnear <= tfar;
										}
										if (entered) {
											best_t = t; best_u = u; best_v = v; best_slot = slot; best_id = id;
											if (TL) best_inst = cur_inst;
											if (ANY_HIT) last = true;
										}
									}
								}
							}
						}
					}
					slot++;
				} while (!last);
				if (ANY_HIT && best_slot != 0xFFFFFFFFu) { cur = kSentinel; depth = 0; }
				else cur = pop();
			}
			const unsigned long long busy = __ballot(cur != kSentinel);
			if (busy == 0ull) break;
			if (!exhausted && (uint32_t)__builtin_popcountll(busy) + q.refill <= MRT_WAVE) break; // enough idle lanes: refill
		}
	}
}
