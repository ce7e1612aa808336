// two_level_kernel.h -- included by kernels.hip.
// Two-level scenes: SceneTLAS::cast_ray / any_hit (src/accel/scene_tlas.h:198-251)
// = tinybvh::BVH::IntersectTLAS (thirdparty/tinybvh/tiny_bvh.h:3306-3380), one lane per ray.
//
// One walk over one node array with one per-lane LDS stack: the TLAS is a BVH2 whose leaves are runs
// of DevInstance rows; entering an instance replaces the lane's ray by the object-space ray
// (o' = M o + t, d' = M d, no renormalisation, so t, t_min and the best hit stay world-parameterised:
// blas_instance.h:56-66, tiny_bvh.h:3327-3331), pushes kInstanceReturn and continues at the BLAS
// root; popping the marker restores the world ray.  Node and triangle arithmetic are the lane kernel's
// (same fma forms as the oracle); an exact tie goes to the lower FLAT id (instance id base + mesh-
// local id).  The record carries the flat id, the instance's layer mask and normalize(basis * n_obj)
// (scene_tlas.h:236-240); the position is taken on the world ray (scene_tlas.h:224-226).
template <bool ANY_HIT>
__global__ __launch_bounds__(MRT_WG) void trace_two_level_kernel(const TraceParams p)
{
	extern __shared__ uint32_t lds_stack[];
	if (skip_launch(p)) return;
	uint64_t ray_idx = 0; uint32_t px = 0, py = 0;
	if (!lane_ray_index(p, blockIdx.x, ray_idx, px, py)) return;
	RayRegs r;
	load_ray(p, ray_idx, px, py, r);

	float best_t = r.t_max, best_u = 0.0f, best_v = 0.0f;
	uint32_t best_slot = 0xFFFFFFFFu, best_id = 0xFFFFFFFFu, best_inst = 0u;

	if (!(r.t_min >= r.t_max)) {
		float ox = r.ox, oy = r.oy, oz = r.oz, dx = r.dx, dy = r.dy, dz = r.dz; // the ray being walked
		float ix = safe_inv(dx), iy = safe_inv(dy), iz = safe_inv(dz);
		float nrx = -(ox * ix), nry = -(oy * iy), nrz = -(oz * iz);
		const uint32_t lane = threadIdx.x & (MRT_WAVE - 1);
		const uint32_t wave = threadIdx.x / MRT_WAVE;
		uint32_t sp = wave * (p.stack_depth * MRT_WAVE) + lane;
		lds_stack[sp] = kSentinel; sp += MRT_WAVE;
		uint32_t cur = 0, id_base = 0u, cur_inst = 0u;
		bool in_blas = false;
		const float4 *nodes = reinterpret_cast<const float4 *>(p.nodes);
		const float4 *hot = reinterpret_cast<const float4 *>(p.tri_hot);
		const float4 *inst = reinterpret_cast<const float4 *>(p.instances);

		while (cur != kSentinel) {
			while (cur < kInstanceReturn) { // inner node of the TLAS or of a BLAS: the lane kernel's step
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
					cur = left_near ? lref : rref;
					lds_stack[sp] = left_near ? rref : lref; sp += MRT_WAVE;
				} else if (hl) cur = lref;
				else if (hr) cur = rref;
				else { sp -= MRT_WAVE; cur = lds_stack[sp]; }
			}
			if (cur == kInstanceReturn) { // the BLAS is done: back to the world ray
				ox = r.ox; oy = r.oy; oz = r.oz; dx = r.dx; dy = r.dy; dz = r.dz;
				ix = safe_inv(dx); iy = safe_inv(dy); iz = safe_inv(dz);
				nrx = -(ox * ix); nry = -(oy * iy); nrz = -(oz * iz);
				in_blas = false;
				sp -= MRT_WAVE; cur = lds_stack[sp];
				continue;
			}
			if (cur == kSentinel) break;
			const uint32_t slot0 = cur & 0x7FFFFFFFu;
			if (!in_blas) { // TLAS leaf: a run of instances, one at a time (tiny_bvh.h:3320-3360)
				const float4 *row = inst + (size_t)slot0 * 8u;
				const float4 m0 = row[0], m1 = row[1], m2 = row[2], meta = row[5];
				// row[5] = {basis[8], root, id_base, layers}; row[6].x = flags
				const uint32_t flags = __float_as_uint(row[6].x);
				if ((flags & 1u) == 0u) { lds_stack[sp] = kLeafBit | (slot0 + 1u); sp += MRT_WAVE; } // the rest of the leaf
				if ((__float_as_uint(meta.w) & p.query_mask) != 0u) {
					ox = fma_(m0.x, r.ox, fma_(m0.y, r.oy, fma_(m0.z, r.oz, m0.w)));
					oy = fma_(m1.x, r.ox, fma_(m1.y, r.oy, fma_(m1.z, r.oz, m1.w)));
					oz = fma_(m2.x, r.ox, fma_(m2.y, r.oy, fma_(m2.z, r.oz, m2.w)));
					dx = fma_(m0.x, r.dx, fma_(m0.y, r.dy, m0.z * r.dz));
					dy = fma_(m1.x, r.dx, fma_(m1.y, r.dy, m1.z * r.dz));
					dz = fma_(m2.x, r.dx, fma_(m2.y, r.dy, m2.z * r.dz));
					ix = safe_inv(dx); iy = safe_inv(dy); iz = safe_inv(dz);
					nrx = -(ox * ix); nry = -(oy * iy); nrz = -(oz * iz);
					lds_stack[sp] = kInstanceReturn; sp += MRT_WAVE;
					in_blas = true; cur_inst = slot0;
					id_base = __float_as_uint(meta.z);
					cur = __float_as_uint(meta.y);
				} else { sp -= MRT_WAVE; cur = lds_stack[sp]; }
				continue;
			}
			// BLAS leaf: the lane kernel's triangle loop on the object-space ray
			uint32_t slot = slot0;
			bool last;
			do {
				const float4 *t3 = hot + (size_t)slot * 3u;
				const float4 q0 = t3[0], q1 = t3[1], q2 = t3[2];
				last = (__float_as_uint(q2.w) & kLastInLeaf) != 0u;
				const float pvx = fma_(dy, q2.z, -(dz * q2.y));
				const float pvy = fma_(dz, q2.x, -(dx * q2.z));
				const float pvz = fma_(dx, q2.y, -(dy * q2.x));
				const float det = dot3(q1.x, q1.y, q1.z, pvx, pvy, pvz);
				if (!(__builtin_fabsf(det) < 1e-8f)) {
					const float inv_det = 1.0f / det;
					const float tvx = ox - q0.x, tvy = oy - q0.y, tvz = oz - q0.z;
					const float u = dot3(tvx, tvy, tvz, pvx, pvy, pvz) * inv_det;
					if (!(u < 0.0f || u > 1.0f)) {
						const float qvx = fma_(tvy, q1.z, -(tvz * q1.y));
						const float qvy = fma_(tvz, q1.x, -(tvx * q1.z));
						const float qvz = fma_(tvx, q1.y, -(tvy * q1.x));
						const float v = dot3(dx, dy, dz, qvx, qvy, qvz) * inv_det;
						if (!(v < 0.0f || u + v > 1.0f)) {
							const float t = dot3(q2.x, q2.y, q2.z, qvx, qvy, qvz) * inv_det;
							const uint32_t id = id_base + __float_as_uint(q0.w);
							if (!(t < r.t_min) && (t < best_t || (t == best_t && best_slot != 0xFFFFFFFFu && id < best_id))) {
								best_t = t; best_u = u; best_v = v; best_slot = slot; best_id = id; best_inst = cur_inst;
								if (ANY_HIT) last = true;
							}
						}
					}
				}
				slot++;
			} while (!last);
			if (ANY_HIT && best_slot != 0xFFFFFFFFu) break;
			sp -= MRT_WAVE; cur = lds_stack[sp];
		}
	}

	finish_two_level_ray(p, ray_idx, r, best_t, best_u, best_v, best_slot, best_id, best_inst);
}

// ---- the same walk for coherent batches: one wave = one packet of 64 rays --------------------------
// The TLAS is walked as packet_kernel.h walks a tree (generic octant form): node index and stack
// (one LDS dword per entry per wave) wave-uniform, nodes and DevInstance rows fetched once per wave
// through the scalar cache, a child or an instance entered when ANY lane's box test passed, one
// ownership bit per lane and stack entry.  Entering an instance, every lane takes its ray to mesh
// space and the wave walks the BLAS with the flat-scene packet walkers, started at the BLAS root with
// the instance's id base: packet_traverse_asm (hand-written node loop, octant-specialised) when the
// mesh-space rays of the lanes that take part share one octant, packet_traverse<8> otherwise.  A lane
// whose world ray missed the instance's box sits that walk out with an empty interval (no box test
// can pass), so every lane reports exactly what the one-ray walk above reports.
// Registers: the world ray, the mesh-space ray and the walkers' state come to 83 VGPRs = 5 waves per SIMD, and
// the dependent scalar node fetches want more waves to hide behind: held to 72 VGPRs (7 waves, no spills) the C5
// grid goes from 43.0 to 34.1 ms; 64 VGPRs (8 waves) spills 38 registers and is no faster (any-hit: slower).
template <bool ANY_HIT>
__global__ __launch_bounds__(MRT_WG) __attribute__((amdgpu_waves_per_eu(7, 8))) void trace_two_level_packet_kernel(const TraceParams p)
{
	__shared__ __attribute__((aligned(16))) uint32_t blas_stack[MRT_WG / MRT_WAVE][(MRT_PACKET_STACK + 1) * 4]; // 16-byte entries (packet_asm_kernel.h)
	__shared__ uint32_t tlas_stack[MRT_WG / MRT_WAVE][MRT_PACKET_STACK];
	if (skip_launch(p)) return;
	uint64_t ray_idx = 0; uint32_t px = 0, py = 0;
	if (!lane_ray_index(p, blockIdx.x, ray_idx, px, py)) return; // exited lanes drop out of every ballot
	RayRegs r;
	load_ray(p, ray_idx, px, py, r);
	uint32_t *stack = tlas_stack[threadIdx.x / MRT_WAVE];
	uint32_t *bstack = blas_stack[threadIdx.x / MRT_WAVE];

	float best_t = r.t_max, best_u = 0.0f, best_v = 0.0f;
	uint32_t best_slot = 0xFFFFFFFFu, best_id = 0xFFFFFFFFu, best_inst = 0u;
	const bool degenerate = r.t_min >= r.t_max;
	float lim_t = degenerate ? -FLT_MAX : best_t; // empty interval: this lane takes no part
	const float ix = safe_inv(r.dx), iy = safe_inv(r.dy), iz = safe_inv(r.dz);
	const float nrx = -(r.ox * ix), nry = -(r.oy * iy), nrz = -(r.oz * iz);
	const float4 *nodes = reinterpret_cast<const float4 *>(p.nodes);
	const float4 *inst = reinterpret_cast<const float4 *>(p.instances);
	uint32_t sp = 0, cur = 0; // wave-uniform
	unsigned long long own_bits = 0ull;
	bool own = true;

	for (;;) {
		cur = __builtin_amdgcn_readfirstlane(cur);
		if (cur < kSentinel) {
			const float4 *n = nodes + (size_t)cur * 4u; // uniform address: scalar loads
			const float4 a = n[0], b = n[1], c = n[2], d = n[3];
			const float l0x = fma_(a.x, ix, nrx), l1x = fma_(b.x, ix, nrx);
			const float l0y = fma_(a.y, iy, nry), l1y = fma_(b.y, iy, nry);
			const float l0z = fma_(a.z, iz, nrz), l1z = fma_(b.z, iz, nrz);
			const float r0x = fma_(c.x, ix, nrx), r1x = fma_(d.x, ix, nrx);
			const float r0y = fma_(c.y, iy, nry), r1y = fma_(d.y, iy, nry);
			const float r0z = fma_(c.z, iz, nrz), r1z = fma_(d.z, iz, nrz);
			const float tl = fmaxf(fmaxf(fminf(l0x, l1x), fminf(l0y, l1y)), fmaxf(fminf(l0z, l1z), r.t_min));
			const float tlx = fminf(fminf(fmaxf(l0x, l1x), fmaxf(l0y, l1y)), fminf(fmaxf(l0z, l1z), lim_t));
			const float tr = fmaxf(fmaxf(fminf(r0x, r1x), fminf(r0y, r1y)), fmaxf(fminf(r0z, r1z), r.t_min));
			const float trx = fminf(fminf(fmaxf(r0x, r1x), fmaxf(r0y, r1y)), fminf(fmaxf(r0z, r1z), lim_t));
			const bool hl = tl <= tlx, hr = tr <= trx;
			const unsigned long long ml = __ballot(hl), mr = __ballot(hr);
			const uint32_t lref = __float_as_uint(a.w), rref = __float_as_uint(b.w);
			if (ml != 0ull && mr != 0ull) {
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
		} else { // TLAS leaf: a run of instances, one at a time
			const uint32_t slot0 = cur & 0x7FFFFFFFu;
			const float4 *row = inst + (size_t)slot0 * 8u; // uniform address
			const float4 m0 = row[0], m1 = row[1], m2 = row[2], meta = row[5];
			const uint32_t flags = __float_as_uint(row[6].x);
			if ((flags & 1u) == 0u) { // the rest of the leaf, owned by the same lanes
				stack[sp] = kLeafBit | (slot0 + 1u);
				own_bits = own ? (own_bits | (1ull << sp)) : (own_bits & ~(1ull << sp));
				sp++;
			}
			// lanes that sit this instance out: their world ray missed its box, or they are done
			const bool dead = !own || lim_t == -FLT_MAX;
			const unsigned long long alive = __ballot(!dead);
			if ((__float_as_uint(meta.w) & p.query_mask) != 0u && alive != 0ull) {
				RayRegs ro;
				ro.ox = fma_(m0.x, r.ox, fma_(m0.y, r.oy, fma_(m0.z, r.oz, m0.w)));
				ro.oy = fma_(m1.x, r.ox, fma_(m1.y, r.oy, fma_(m1.z, r.oz, m1.w)));
				ro.oz = fma_(m2.x, r.ox, fma_(m2.y, r.oy, fma_(m2.z, r.oz, m2.w)));
				ro.dx = fma_(m0.x, r.dx, fma_(m0.y, r.dy, m0.z * r.dz));
				ro.dy = fma_(m1.x, r.dx, fma_(m1.y, r.dy, m1.z * r.dz));
				ro.dz = fma_(m2.x, r.dx, fma_(m2.y, r.dy, m2.z * r.dz));
				ro.t_min = r.t_min; ro.t_max = r.t_max;
				const uint32_t root = __builtin_amdgcn_readfirstlane(__float_as_uint(meta.y));
				const uint32_t id_base = __builtin_amdgcn_readfirstlane(__float_as_uint(meta.z));
				const uint32_t before_id = best_id; const float before_t = best_t;
				// octant of the mesh-space rays that take part
				const unsigned long long sx = __ballot(!dead && safe_inv(ro.dx) < 0.0f), sy = __ballot(!dead && safe_inv(ro.dy) < 0.0f),
						sz = __ballot(!dead && safe_inv(ro.dz) < 0.0f);
				const bool uniform = (sx == 0ull || sx == alive) && (sy == 0ull || sy == alive) && (sz == 0ull || sz == alive);
				const int oct = uniform ? ((sx ? 1 : 0) | (sy ? 2 : 0) | (sz ? 4 : 0)) : 8;
				uint32_t nn = 0, nt = 0, nd = 0; // (counters of the flat counting builds: unused here)
				if (oct == 8 || p.n_nodes >= kAsmNodeLimit) { // mixed octants, or node offsets beyond the asm loop's 32 bits
					packet_traverse<8, ANY_HIT, false>(p, ro, bstack, best_t, best_u, best_v, best_slot, nn, nt, nd, root, id_base, &best_id, dead);
				} else {
					*(volatile uint32_t *)&bstack[0] = kSentinel;
					const uint32_t bsp = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(bstack + 4);
#define MRT_PKTB(O) case O: packet_traverse_asm<O, ANY_HIT>(p, ro, bsp, best_t, best_u, best_v, best_slot, nn, nt, root, id_base, &best_id, dead); break;
					switch (oct) { MRT_PKTB(0) MRT_PKTB(1) MRT_PKTB(2) MRT_PKTB(3) MRT_PKTB(4) MRT_PKTB(5) MRT_PKTB(6) MRT_PKTB(7) }
#undef MRT_PKTB
				}
				if (best_id != before_id || best_t != before_t) best_inst = slot0;
				if (!degenerate) lim_t = (ANY_HIT && best_slot != 0xFFFFFFFFu) ? -FLT_MAX : best_t;
				// any-hit: done when every lane is degenerate or has its answer
				if (ANY_HIT && __ballot(lim_t != -FLT_MAX) == 0ull) break;
			}
		}
		if (sp == 0) break;
		sp--; cur = stack[sp];
		own = ((own_bits >> sp) & 1ull) != 0ull;
	}

	finish_two_level_ray(p, ray_idx, r, best_t, best_u, best_v, best_slot, best_id, best_inst);
}
