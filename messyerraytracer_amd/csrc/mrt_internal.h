// mrt_internal.h — shared between the host-side scene preparation and the HIP
// kernels of libmrt_hip.so.  Not part of the C-ABI.
#pragma once
#include <cstdint>
#include "../../include/mrt_hip.h"

static_assert(sizeof(mrt_ray32) == 32, "GPURayPacked must be 32 bytes (src/api/gpu_types.h:70)");
static_assert(sizeof(mrt_hit32) == 32, "GPUIntersectionPacked must be 32 bytes (src/api/gpu_types.h:93)");
static_assert(sizeof(mrt_tri64) == 64, "GPUTrianglePacked must be 64 bytes (src/api/gpu_types.h:51)");
static_assert(sizeof(mrt_bvh_node32) == 32, "GPUBVHNodePacked must be 32 bytes (src/api/gpu_types.h:127)");
static_assert(sizeof(mrt_bvh_node_wide64) == 64, "GPUBVHNodeWide must be 64 bytes (src/gpu/gpu_structs.h:47)");
static_assert(sizeof(mrt_host_ray60) == 60, "Ray must be 60 bytes (src/core/ray.h:25-51, precision=single)");
static_assert(sizeof(mrt_host_hit44) == 44, "Intersection must be 44 bytes (src/core/intersection.h:16-40)");
static_assert(sizeof(mrt_host_tri80) == 80, "Triangle must be 80 bytes (src/core/triangle.h:22-39)");

namespace mrt {

// ---- device scene layout (HBM) ------------------------------------------------
// Node: the reference's 64-byte dual-AABB node with the child encoding folded
// into one 32-bit reference per child:
//   ref <  0x7FFFFFFF : index of a wide node
//   ref == 0x7FFFFFFF : sentinel (bottom of every traversal stack)
//   ref >= 0x80000000 : leaf; low 31 bits = first slot in the leaf-ordered
//                       triangle arrays; the leaf's last triangle carries
//                       kLastInLeaf in TriHot::flags.
struct alignas(16) DevNode {
	float lmin[3]; uint32_t left_ref;
	float lmax[3]; uint32_t right_ref;
	float rmin[3]; uint32_t left_count;   // counts kept for validation / stats only
	float rmax[3]; uint32_t right_count;
};
static_assert(sizeof(DevNode) == 64, "DevNode must be 64 bytes");

// Triangle, split hot/cold: traversal reads 48 B per test, the 16-byte normal
// row is read once per ray for the winning triangle.
struct alignas(16) TriHot {
	float v0[3]; uint32_t id;
	float e1[3]; uint32_t layers;
	float e2[3]; uint32_t flags;
};
static_assert(sizeof(TriHot) == 48, "TriHot must be 48 bytes");
struct alignas(16) TriCold { float normal[3]; uint32_t pad; };

// 4-wide node for the persistent lane kernel: two levels of the BVH2 collapsed into one 128-byte
// line, halving the chain of dependent fetches a ray walks.
// Built on the host from the same binned-SAH BVH2 (greedy: keep opening the child with
// the largest surface area until there are 4).  Child refs use the DevNode encoding with
// wide4 indices; an unused slot has ref == kSentinel.
struct alignas(16) Dev4Node {
	float box[4][6];      // child c: min.xyz, max.xyz
	uint32_t ref[4];
	uint32_t n_children, pad[3];
};
static_assert(sizeof(Dev4Node) == 128, "Dev4Node must be 128 bytes");

// 8-wide compressed node (Ylitie, Karras, Laine 2017, the layout idea of the
// reference's cwbvh_traverse.comp.glsl, re-cut for one 128-byte cache line and explicit child refs):
// child boxes are 8-bit coordinates on a per-node grid, lo = fma(q, 2^(exp-127), org) per axis.  The
// builder rounds outwards and checks the DECODED value in float, so a decoded box contains the exact
// one and the slab test on it (same formula as for exact boxes, monotone in the box coordinate)
// passes whenever the test on the exact box does: the walk visits a superset of what the 2-wide walk
// visits, and every triangle is still tested with the exact arithmetic.  A hit is accepted only if
// the ray also passes the slab test on the leaf's EXACT box (leaf_box[], 32 bytes per leaf, read on
// that rare path only): boxes nest and the slab arithmetic is monotone, so passing the leaf's box is
// passing every ancestor's, and the 8-wide walk reports exactly what the 2-wide walk reports -- also
// for rays that lie in a face plane of a box, where a looser box would let an edge hit through.
struct alignas(16) Dev8Node {
	float org[3];
	uint8_t exp[3];       // float exponent byte of the grid step per axis
	uint8_t n_children;
	uint8_t qlo[3][8];    // [axis][child]
	uint8_t qhi[3][8];
	uint32_t ref[8];      // DevNode encoding with wide8 indices; unused slots kSentinel
	uint32_t pad[8];
};
static_assert(sizeof(Dev8Node) == 128, "Dev8Node must be 128 bytes");

// ---- the context's counter block (unsigned long long words; mrt_ctx::d_counters) ----
// [0, kNumCounters): visit counters of the counting kernel variants (mrt_options.count_visits)
constexpr int kCntRays = 0, kCntTris = 1, kCntNodes = 2, kCntHits = 3, kCntMaxStack = 4, kCntDeadPops = 5;
constexpr int kCntWaveNodeFetch = 6; // node fetches as the hardware sees them: one per wave step (packet kernels), one per lane step
                                     // = one divergent 64-/128-byte line (lane kernels)
constexpr int kCntWaveTriFetch = 7;  // triangle rows fetched: one per wave step (packet kernels), one per lane test (lane kernels)
constexpr int kCntLeafBoxChecks = 8; // 8-wide kernel: exact leaf boxes read for candidate hits
constexpr int kCntFetchWaitCycles = 9, kCntWaveCycles = 10, kCntWaves = 11; // trace_packet_rows_kernel<.., 1>: s_memtime clock
constexpr int kNumCounters = 16;
constexpr int kAutoGridOff = kNumCounters;         // 8 words: what detect_grid_kernel found (4 x u32 used)
constexpr int kDetectScratchOff = kAutoGridOff + 8; // 1026 words: jump masks + ticket + wide-neighbour count
constexpr int kNextRayOff = kDetectScratchOff + 1026; // 128 words: 8 ray counters of the persistent kernels, 16 words apart
constexpr int kCounterWords = kNextRayOff + 128;

constexpr uint32_t kSentinel = 0x7FFFFFFFu;
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kLastInLeaf = 1u;

enum InFmt : uint32_t { IN_RAY32 = 0, IN_HOST60 = 1, IN_GRID = 2 };
// OUT_TOKEN4: one u32 per ray = leaf-order slot of the winning triangle (0xFFFFFFFF = miss); the
// full record is rebuilt from (ray, slot) by expand_tokens_kernel, bit for bit
// OUT_TOKEN8 (two-level scenes): two u32 per ray = {slot of the winning triangle in its mesh's arrays (0xFFFFFFFF =
// miss), DevInstance row of the instance it was hit in}: the record is a function of (ray, triangle, instance)
enum OutFmt : uint32_t { OUT_HIT32 = 0, OUT_HOST44 = 1, OUT_BOOL8 = 2, OUT_TOKEN4 = 3, OUT_TOKEN8 = 4 };
enum LaneMap : uint32_t { MAP_LINEAR = 0, MAP_TILE8X8 = 1, MAP_AUTO = 2 };

struct DevInstance;
struct TraceParams {
	const DevNode *nodes;
	const Dev4Node *nodes4;    // 4-wide layout of the persistent lane kernel (may be null)
	const Dev8Node *nodes8;    // 8-wide compressed layout (may be null)
	const float *leaf_box;     // with nodes8: exact box {min xyz, -, max xyz, -} of the leaf that starts at a slot
	const DevInstance *instances; // two-level scenes (kernel == MRT_KERNEL_TWO_LEVEL)
	const TriHot *tri_hot;
	const TriCold *tri_cold;
	const void *row_array;     // nodes + triangles as one array of 64-byte rows (packet_rows_kernel.h; may be null)
	const void *row_array4;    // 4-wide node rows (128 bytes) + triangle rows, in 64-byte units (packet_quad_kernel.h; may be null)
	uint32_t rows_wg;          // trace_packet_rows_kernel<.., 2>: threads per workgroup, 64 or 256
	uint32_t rows_cull;        // ... 1: packet-level frustum culling in the 128-ray walk
	float scene_abs_max;       // largest |coordinate| of the scene's bounds (error bound of the culling test)
	uint32_t tri_unit_base4;   // row_array4: the unit of triangle slot 0 (= 2 * number of 4-wide nodes)
	const void *rays;          // IN_RAY32 / IN_HOST60 (device)
	void *hits;                // device
	const uint32_t *perm;      // optional: lane g traces ray perm[g], writes hits[perm[g]]
	unsigned long long *counters; // COUNT variants: the kCnt* words above
	const uint32_t *auto_grid; // MAP_AUTO: {row width (0 = none), rows, tiles_x, incoherent} written by detect_grid_kernel
	const uint32_t *skip_flag; // optional: the whole launch returns at once when *skip_flag == skip_when
	uint32_t skip_when;        // (a "coherent" batch that is not: the packet launch yields to the lane launch)
	uint64_t count;            // number of rays
	uint32_t query_mask;
	uint32_t in_fmt, out_fmt, lane_map;
	uint32_t grid_w, grid_h, y0, rows; // IN_GRID / MAP_TILE8X8: rows [y0, y0+rows) of a grid_w x grid_h grid
	uint32_t tiles_x;          // ceil(grid_w / tile width)
	uint32_t tile_w_log2;      // lane tile: 2^k wide, 64 / 2^k high
	uint32_t tile_order;       // 0: tiles row-major, 1: Z-order inside 16x16-tile super-tiles, 2: 32x32, 3: column strips per XCD
	uint32_t tile_group;       // tile_order 3: consecutive tiles per workgroup (set by launch_trace)
	// Frame-coherent tile schedule of grid casts (api.hip, TileSchedule): launch slot u runs schedule unit tile_sched[u]
	// (a unit = tile_unit consecutive tiles: 1, or 2 for the 128-ray walk) and leaves the shader cycles it took in
	// tile_cost[unit]; the next cast of the same grid launches the units longest first.  Both may be null.
	// An entry of tile_sched: bits 0-27 an id, bits 28-31 what the slot's wave works on -- 0: schedule unit `id`; 1: the one tile
	// `id` (in the wave's first group; a second group stays empty); 2 + q: quarter q of tile `id`, 4x4 pixels in lanes 0..15.
	// The most expensive units of the last measured frame are launched in such pieces (api.hip schedule_split): a frame of a
	// million rays ends with its longest walk, and the longest one is 2-3 x the 99th percentile.  sched_hdr[2] = slots in use
	// (the launch covers n_slots_max); a piece parks its start time in tile_cost[n_units + slot] and ADDS its share to its unit.
	const uint32_t *tile_sched;
	uint32_t *tile_cost;
	uint32_t tile_unit, n_units;
	const uint32_t *sched_hdr;
	uint32_t n_slots_max;
	// Small grids (api.hip quarter_small_grid): EVERY tile is launched as its four quarter tiles (launch slot s = quarter s & 3 of
	// tile s >> 2, 16 rays in lanes 0..15) -- a grid of fewer tiles than the device has wave slots lasts as long as its longest
	// walk, and a quarter tile's walk is about half as long as its tile's.
	uint32_t quarter_all;
	uint32_t sparse_lanes;     // linear lane map of the lane kernel: rays per wave, in lanes 0 .. sparse_lanes - 1 (0 = 64): a small batch on more, emptier waves (api.hip launch_lane)
	uint32_t kernel;           // MRT_KERNEL_LANE / MRT_KERNEL_PACKET
	uint32_t stack_depth;      // LDS stack entries per lane
	uint32_t xcd_swizzle;      // 1: remap blockIdx so each XCD owns a contiguous band
	uint32_t n_tris;           // rows in tri_hot / tri_cold (token validation)
	uint32_t n_instances;      // rows in instances (two-level scenes; token validation)
	uint32_t n_nodes;          // rows in nodes (the hand-written node loop addresses them with a 32-bit byte offset)
	uint32_t count_mode;       // COUNT variants: mrt_options.count_visits (1: visit counters, 2: only the sampled clock of the rows kernel)
	uint32_t extra_lds;        // experiments: dynamic LDS bytes added per workgroup of the packet kernels (occupancy sweeps)
	mrt_camera cam;
};

// host-side preparation (scene_prep.cpp)
struct DeviceSceneHost {
	DevNode *nodes = nullptr; uint32_t n_nodes = 0;
	Dev4Node *nodes4 = nullptr; uint32_t n_nodes4 = 0;
	uint32_t stack4 = 0;        // per-wave stack entries the 4-wide walk can need
	bool want8 = false;         // in: also build the 8-wide compressed collapse
	Dev8Node *nodes8 = nullptr; uint32_t n_nodes8 = 0; uint32_t stack8 = 0;
	float *leaf_box = nullptr;  // with nodes8: 8 floats per triangle slot, filled at the first slot of every leaf
	float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0}; // scene AABB (sort key quantisation)
	TriHot *hot = nullptr; TriCold *cold = nullptr; uint32_t n_tris = 0;
	uint32_t depth = 0;         // max stack entries any traversal can need (incl. sentinel)
};
// Returns MRT_OK or an error; on success arrays are malloc'ed (free with free()).
int prepare_scene(const mrt_tri64 *tris, uint32_t n_tris, const mrt_bvh_node32 *nodes, uint32_t used_nodes,
		const uint32_t *prim_idx, DeviceSceneHost *out, char *err, size_t err_len);

// device-side build (device_build.hip): LBVH over n >= 2 device-resident triangles, written in the
// layout above.  Arrays are hipMalloc'ed and belong to the caller on success.  `stream` is a hipStream_t.
struct DeviceBuildResult {
	DevNode *nodes = nullptr; TriHot *hot = nullptr; TriCold *cold = nullptr;
	Dev4Node *nodes4 = nullptr;  // optional: one 4-wide node per binary node, at the binary node's index
	Dev8Node *nodes8 = nullptr;  // optional: likewise for the 8-wide compressed layout
	float *leaf_box = nullptr;   // with nodes8: exact leaf boxes, 8 floats per slot
	uint32_t n_nodes = 0, n_tris = 0, depth = 0, stack4 = 0, stack8 = 0;
	float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0};
};
struct BuildArena { void *ptr = nullptr; size_t cap = 0; uint32_t *pinned = nullptr; }; // the builder's temporaries: owned by the context, grown on demand (pinned: 64 host bytes the per-level counters are read back into)
// form: 0 the radix tree over the Morton keys (fastest build), 1 PLOC, 2 binned SAH (the host builder's tree; leaves of several triangles)
int device_build_lbvh(const mrt_tri64 *d_tris, uint32_t n, bool want4, bool want8, bool safe_handoff, int form, BuildArena *arena, void *stream,
		DeviceBuildResult *out, char *err, size_t err_len);

// ---- two-level scene: SceneTLAS / MeshBLAS / BLASInstance
// (src/accel/scene_tlas.h:140-251, mesh_blas.h:86-138, blas_instance.h:47-107) -------------
// One node array: TLAS nodes at [0, tlas_cap) (rebuilt in place when instances move), the BLAS
// of every distinct mesh behind them, node and leaf refs already global.  A TLAS leaf ref is the
// first slot of a run of DevInstance rows (leaf order; the last row of a leaf has flags & 1).
constexpr uint32_t kAsmNodeLimit = 1u << 26;     // packet_asm_kernel.h: node index * 64 must fit 32 bits
constexpr uint32_t kInstanceReturn = 0x7FFFFFFEu; // stack marker: back from a BLAS to the TLAS walk
// kernel ids of two-level scenes (TraceParams.kernel; the values of the public enum, include/mrt_hip.h):
// MRT_KERNEL_TWO_LEVEL one lane per ray, _PACKET one wave per 64-ray packet (coherent batches), _PERSISTENT
// resident waves with node / leaf phases (large incoherent batches), _PERSISTENT8 the same with 8-wide BLAS nodes
using ::MRT_KERNEL_TWO_LEVEL; using ::MRT_KERNEL_TWO_LEVEL_PACKET; using ::MRT_KERNEL_TWO_LEVEL_PERSISTENT;
using ::MRT_KERNEL_TWO_LEVEL_PERSISTENT8;
struct alignas(16) DevInstance {
	float inv[12];      // world -> object, rows {m00 m01 m02 tx}: o' = M o + t, d' = M d (no renormalisation: t stays world-parameterised)
	float basis[9];     // object -> world 3x3 (normals: normalize(basis n))
	uint32_t root;      // BLAS root node (global index)
	uint32_t id_base;   // flat id of the instance's first triangle (running offset in registration order)
	uint32_t layers;    // the mesh's layer mask
	uint32_t flags;     // 1 = last instance of its TLAS leaf
	uint32_t index;     // registration index
	uint32_t root8;     // BLAS root in the 8-wide layout (scenes that have it)
	uint32_t pad[5];
};
static_assert(sizeof(DevInstance) == 128, "DevInstance must be 128 bytes");

struct TwoLevelBlas { uint32_t first_tri, n_tris, root, depth; float lo[3], hi[3]; uint32_t root8, stack8; };
struct TwoLevelHost {
	DevNode *nodes = nullptr; uint32_t n_nodes = 0, tlas_cap = 0, n_tlas_nodes = 0;
	TriHot *hot = nullptr; TriCold *cold = nullptr; uint32_t n_tris = 0;   // all BLAS triangles, mesh-space
	DevInstance *inst = nullptr; uint32_t n_inst = 0;
	TwoLevelBlas *blas = nullptr; uint32_t n_blas = 0;
	uint32_t *inst_blas = nullptr;   // per registered instance: its BLAS
	uint32_t depth = 0;              // stack entries one ray can need
	// optional 8-wide compressed layout of every BLAS (the persistent kernel's walk for incoherent rays)
	Dev8Node *nodes8 = nullptr; uint32_t n_nodes8 = 0; float *leaf_box = nullptr;
	bool wide8 = false;              // every BLAS has it (set by the builder of the BLASes)
	uint32_t depth8 = 0;             // stack entries with 8-wide BLAS walks
	uint64_t flat_tris = 0;          // triangles of the flattened scene (sum over instances)
};
void free_two_level(TwoLevelHost *h);
// Builds every BLAS (binned SAH, as MeshBLAS::build) and the TLAS over the instances' world boxes.
// build_blas = false: only groups the instances by mesh and lays the BLAS roots out (device-built BLASes).
int prepare_two_level(const float *verts9, uint32_t n_mesh_tris, const mrt_instance *instances, uint32_t n_instances,
		uint32_t n_threads, bool build_blas, TwoLevelHost *out, char *err, size_t err_len);

// New transforms for the same instances: inverse, world box, TLAS rebuilt into nodes[0, tlas_cap) and inst[].
int refit_two_level(TwoLevelHost *h, const mrt_instance *instances, uint32_t n_instances, char *err, size_t err_len);

} // namespace mrt
