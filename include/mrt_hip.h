/*
 * mrt_hip.h — C-ABI of the MI355X (gfx950) batch ray-cast backend.
 *
 * This is the drop-in boundary: everything the reference's `GPURayCaster`
 * (src/gpu/gpu_ray_caster.h:50-136) does through Godot's RenderingDevice is
 * reachable through these entry points.  `RayDispatcher`
 * (src/dispatch/ray_dispatcher.h:74-79,124-356) keeps calling a caster-shaped
 * C++ object (messyerraytracer_amd/csrc/host/gpu_ray_caster.hpp) that forwards
 * here.  Plain pointers and sizes only; no C++/torch types; never throws.
 *
 * All structs are natural C layout == GLSL std430 of the reference
 * (src/api/gpu_types.h:44-126, src/gpu/gpu_structs.h:41-47) or the reference's
 * host PODs at precision=single (src/core/ray.h:25-98,
 * src/core/intersection.h:16-61, src/core/triangle.h:22-51).
 */
#ifndef MRT_HIP_H_
#define MRT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRT_VERSION_MAJOR 0
#define MRT_VERSION_MINOR 1

/* ---- status codes (no exceptions cross the boundary; reference convention:
 *      tools/lint.py rule no-exceptions, gpu_ray_caster.cpp:76-115) ---------- */
enum {
	MRT_OK = 0,
	MRT_ERR_INVALID = 1,      /* null pointer / bad argument / bad struct_size   */
	MRT_ERR_NO_DEVICE = 2,    /* no gfx950 device at that ordinal                */
	MRT_ERR_HIP = 3,          /* a HIP runtime call failed; see mrt_last_error   */
	MRT_ERR_NO_SCENE = 4,     /* cast before upload_scene (is_available()==false) */
	MRT_ERR_PENDING = 5,      /* submit while a dispatch is pending (cpp:538)    */
	MRT_ERR_NOT_PENDING = 6,  /* collect without submit                          */
	MRT_ERR_OOM = 7,
	MRT_ERR_UNSUPPORTED = 8,
	MRT_ERR_BAD_BVH = 9       /* BVH failed host-side validation before upload   */
};

/* ---- device PODs -------------------------------------------------------- */

/* GPURayPacked, src/api/gpu_types.h:65-69 */
typedef struct mrt_ray32 {
	float origin[3];    float t_max;
	float direction[3]; float t_min;
} mrt_ray32;

/* GPUIntersectionPacked, src/api/gpu_types.h:87-92; prim_id == -1 => miss */
typedef struct mrt_hit32 {
	float t;          int32_t prim_id;
	float bary_u;     float bary_v;
	float normal[3];  uint32_t hit_layers;
} mrt_hit32;

/* GPUTrianglePacked, src/api/gpu_types.h:44-50 */
typedef struct mrt_tri64 {
	float v0[3];     uint32_t id;
	float edge1[3];  uint32_t layers;
	float edge2[3];  float pad2;
	float normal[3]; float pad3;
} mrt_tri64;

/* tinybvh::BVH::BVHNode (thirdparty/tinybvh/tiny_bvh.h:857-866) ==
 * GPUBVHNodePacked (src/api/gpu_types.h:122-126).  tri_count>0 => leaf,
 * left_first = first slot in prim_idx[]; else children are the adjacent pair
 * left_first, left_first+1.  Node 0 is the root, node 1 is an unused hole. */
typedef struct mrt_bvh_node32 {
	float aabb_min[3]; uint32_t left_first;
	float aabb_max[3]; uint32_t tri_count;
} mrt_bvh_node32;

/* GPUBVHNodeWide, src/gpu/gpu_structs.h:41-47 (Aila-Laine dual-AABB node).
 * count>0 => that child is a leaf and idx is its first triangle slot in the
 * (leaf-ordered) device triangle array; count==0 => idx is a wide-node index. */
typedef struct mrt_bvh_node_wide64 {
	float left_min[3];  uint32_t left_idx;
	float left_max[3];  uint32_t right_idx;
	float right_min[3]; uint32_t left_count;
	float right_max[3]; uint32_t right_count;
} mrt_bvh_node_wide64;

/* ---- host PODs of the reference (precision=single) ---------------------- */

/* Ray, src/core/ray.h:25-51 (60 B).  Only origin/direction/t_min/t_max are
 * consumed (gpu_ray_caster.cpp:643-650). */
typedef struct mrt_host_ray60 {
	float origin[3];
	float direction[3];
	float inv_direction[3];
	int32_t dir_sign[3];
	float t_min, t_max;
	uint32_t flags;
} mrt_host_ray60;

/* Intersection, src/core/intersection.h:16-40 (44 B); miss: prim_id=UINT32_MAX,
 * t=FLT_MAX, u=v=0, hit_layers=0, position/normal untouched by set_miss(). */
typedef struct mrt_host_hit44 {
	float t;
	float position[3];
	float normal[3];
	float u, v;
	uint32_t prim_id;
	uint32_t hit_layers;
} mrt_host_hit44;

/* Triangle, src/core/triangle.h:22-39 (80 B) */
typedef struct mrt_host_tri80 {
	float v0[3], v1[3], v2[3];
	float edge1[3], edge2[3], normal[3];
	uint32_t id, layers;
} mrt_host_tri80;

/* Camera for the on-device primary-ray grids.  Three generators, chosen by `kind`:
 *   MRT_CAMERA_DEBUG_GRID   RayTracerDebug::cast_debug_rays, src/godot/raytracer_debug.cpp:572-596
 *                           (mrt_camera_look fills basis + half extents on the host exactly as :573-583):
 *                           dir = normalize(fwd + right u + up v), v grows with the row index;
 *   MRT_CAMERA_PERSPECTIVE  RayCamera::_generate_perspective, src/modules/graphics/ray_camera.h:234-251 --
 *                           the generator behind every coherent=true query of the renderer
 *                           (ray_renderer.cpp:521-537): v is FLIPPED (row 0 = top),
 *                           dir = normalize(basis.xform((u half_w, v half_h, -1)));
 *   MRT_CAMERA_ORTHOGRAPHIC RayCamera::_generate_orthographic, ray_camera.h:255-273: parallel rays,
 *                           origin = (origin + up (v half_h)) + right (u half_w), direction = -basis column 2.
 * For the two RayCamera kinds right / up / fwd hold the columns 0 / 1 / 2 of the camera basis
 * (fwd = column 2 as it stands: the camera looks along its negative). */
enum { MRT_CAMERA_DEBUG_GRID = 0, MRT_CAMERA_PERSPECTIVE = 1, MRT_CAMERA_ORTHOGRAPHIC = 2 };
typedef struct mrt_camera {
	float origin[3];
	float fwd[3], right[3], up[3];
	float half_w, half_h;
	float t_min, t_max;     /* Ray() defaults: 0.001f, FLT_MAX (ray.h:59) */
	uint32_t kind;          /* MRT_CAMERA_* */
	float inv_w, inv_h;     /* RayCamera kinds: 1.0f / width, 1.0f / height (ray_camera.h:56-57) */
	float jitter_x, jitter_y; /* RayCamera kinds: sub-pixel offset, 0.5 = pixel centre (generate_ray_jittered, :106-122) */
	uint32_t reserved[3];
} mrt_camera;

/* RayStats, src/core/stats.h:20-55, plus device timing of the last cast. */
typedef struct mrt_stats {
	uint64_t rays_cast;
	uint64_t tri_tests;          /* filled only when options.count_visits != 0 */
	uint64_t bvh_nodes_visited;  /* idem: wide-node (internal) visits          */
	uint64_t hits;               /* idem                                        */
	float last_trace_ms;         /* hipEvent time of the trace kernel(s)        */
	float last_sort_ms;          /* key+sort+gather kernels (0 if coherent)     */
	float last_h2d_ms, last_d2h_ms;
	uint32_t last_kernel_launches;
	uint32_t max_stack_depth;    /* count_visits only                           */
	uint64_t dead_pops;          /* count_visits, packet kernel: popped nodes no lane still needed */
	uint32_t detected_grid_w;    /* count_visits: row width found for the last coherent mrt_cast (0 = none) */
	uint32_t reserved;           /* 1 if the last batch declared coherent went to the one-lane-per-ray kernel by the device's verdict:
	                                judged incoherent, or fewer than 2^15 rays in which no row width was found */
	float last_build_ms;         /* device time of the last mrt_build_scene_device */
	uint32_t last_kernel;        /* MRT_KERNEL_* that did the work of the last blocking cast (a batch declared coherent is
	                                checked on the device: this is the kernel the device chose); 0 after an ASYNC cast */
	/* count_visits, memory-side view of the walk (what the roofline of bench.py prices): */
	uint64_t wave_node_fetches;  /* node fetches issued: one per wave step in the packet kernels (the node is fetched once
	                                for 64 rays), one per lane step (= one divergent cache line) in the lane kernels */
	uint64_t wave_tri_fetches;   /* 48-byte triangle rows fetched, counted the same way */
	uint64_t leaf_box_checks;    /* 8-wide kernel: exact 32-byte leaf boxes read for candidate hits */
	/* count_visits, MRT_KERNEL_PACKET_ROWS only: a clock on the walk (s_memtime, shader cycles, summed over waves) */
	uint64_t fetch_wait_cycles;  /* between issuing a row fetch and having it (two s_memtime reads included)        */
	uint64_t wave_cycles;        /* whole kernel body                                                             */
	uint64_t waves;              /* waves that were clocked                                                       */
} mrt_stats;

/* mode: RayQuery::Mode, src/api/ray_query.h:54-57 / RAY_MODE spec constant,
 * bvh_traverse.comp.glsl:78 */
enum { MRT_MODE_NEAREST = 0, MRT_MODE_ANY_HIT = 1 };

/* flags for mrt_cast / mrt_submit */
enum {
	MRT_FLAG_COHERENT       = 1u << 0, /* RayQuery::coherent: skip the Morton sort (ray_dispatcher.h:135) */
	MRT_FLAG_RAYS_ON_DEVICE = 1u << 1, /* `rays` is a device pointer (HBM-resident input)  */
	MRT_FLAG_HITS_ON_DEVICE = 1u << 2, /* `hits` is a device pointer                      */
	MRT_FLAG_HOST_LAYOUT    = 1u << 3, /* rays are mrt_host_ray60, hits are mrt_host_hit44 (conversion of
	                                      gpu_ray_caster.cpp:639-650,442-456 runs on the device)          */
	MRT_FLAG_BOOL_OUT       = 1u << 4, /* any-hit only: `hits` is uint8_t[count] (cast_rays_any_hit)     */
	MRT_FLAG_FORCE_SORT     = 1u << 5, /* sort even if count < 256 (tests)                               */
	MRT_FLAG_TOKEN_OUT      = 1u << 6, /* `hits` is uint32_t[count]: per ray a hit token (MRT_TOKEN_MISS, or an
	                                      opaque name of the winning triangle valid for this scene upload and
	                                      for identical uploads on other devices).  mrt_expand_tokens rebuilds
	                                      the full record from (ray, token) bit for bit: the multi-GPU gather
	                                      moves 4 bytes per ray instead of 32.  Not with BOOL_OUT.  Two-level
	                                      scenes: a token is TWO words per ray, {triangle, instance} (first
	                                      word MRT_TOKEN_MISS = miss): mrt_token_bytes() = 8, `hits` is
	                                      uint32_t[2 * count].                                                */
	MRT_FLAG_ASYNC          = 1u << 7  /* mrt_cast / mrt_cast_grid with device-resident rays and hits: queue the
	                                      work on the context's stream and return without waiting (no timing
	                                      stats).  Order later work on that stream, or mrt_synchronize().
	                                      Lets a frame loop keep the device busy while the host queues the
	                                      exchange of the previous frame (sharded.py).                        */
};
#define MRT_TOKEN_MISS 0xFFFFFFFFu

/* kernel variants (options.kernel); 0 picks the default for the batch */
enum {
	MRT_KERNEL_AUTO = 0,    /* by the batch: coherent batches by packets -- from 2^22 rays the 128-ray walk, below it whichever
	                           of the packet kernels measured fastest on that grid, small grids in pieces of 4 or 16 rays per wave --,
	                           everything else one lane per ray (resident waves from 2^16 rays, waves of 1 - 4 rays up to 2^15) */
	MRT_KERNEL_LANE = 1,    /* one lane = one ray, per-lane LDS stack, while-while loop         */
	MRT_KERNEL_PACKET = 2,  /* one wave = one 64-ray packet, per-wave LDS stack, scalar fetches */
	/* 3 and 4 were two packet-walk experiments of round 1 (4-wide nodes, two packets per wave); retired, ids not reused */
	MRT_KERNEL_PACKET_ASM = 5, /* packet walk with the hand-written gfx950 node loop (default for coherent batches) */
	MRT_KERNEL_LANE_PERSISTENT = 6, /* lane kernel with resident waves pulling rays from a counter, short LDS
	                                  stack + HBM spill, node / leaf phases                                */
	MRT_KERNEL_LANE4_PERSISTENT = 7, /* the same over the 4-wide collapse of the BVH, one 128-byte line per step */
	MRT_KERNEL_LANE8_PERSISTENT = 8, /* the same over an 8-wide collapse with 8-bit child boxes on a per-node grid
	                                  (compressed wide BVH, cf. the reference's cwbvh_traverse.comp.glsl), one
	                                  128-byte line per step (default for large incoherent batches)       */
	MRT_KERNEL_PACKET_DUAL = 9, /* the packet walk end to end in gfx950 assembly over ONE array of 64-byte rows (nodes +
	                               triangles), 128 rays per wave: two neighbouring 8x8 tiles share one walk (one fetch,
	                               one stack, one near / far decision per step; the box and triangle tests once per
	                               tile that owns the row); default for coherent batches of >= 2^22 rays            */
	MRT_KERNEL_PACKET_ROWS = 10, /* the same walk with one packet per wave                                           */
	MRT_KERNEL_PACKET_QUAD = 11, /* the 128-ray shared walk over FOUR-wide node rows (128 bytes: the 4-wide collapse of the
	                                same tree, exact boxes): half the row fetches for the same box tests
	                                (packet_quad_kernel.h); needs the 4-wide layout resident                     */
	/* reported in mrt_stats.last_kernel only (chosen by the library for two-level scenes, not selectable): */
	MRT_KERNEL_TWO_LEVEL = 100, MRT_KERNEL_TWO_LEVEL_PACKET = 101, MRT_KERNEL_TWO_LEVEL_PERSISTENT = 102,
	MRT_KERNEL_TWO_LEVEL_PERSISTENT8 = 103
};
/* Name of the __global__ function behind a kernel id ("trace_packet_asm_kernel", ...); "?" for an unknown id. */
const char *mrt_kernel_name(uint32_t kernel);

typedef struct mrt_options {
	uint32_t struct_size;     /* = sizeof(mrt_options) */
	uint32_t kernel;          /* MRT_KERNEL_*                                        */
	uint32_t count_visits;    /* 1: counting kernel variant fills mrt_stats counters; 2 (MRT_KERNEL_PACKET_ROWS): only its
	                             sampled s_memtime clock (fetch_wait_cycles / wave_cycles / waves), nothing else counted */
	uint32_t sort_threshold;  /* MIN_BATCH_FOR_SORTING, default 256 (ray_dispatcher.h:427) */
	uint32_t grid_tile;       /* 0: default 8x8 lane tiling for grid casts; 1: row-major */
	/* tuning knobs (0 = default); results never depend on them */
	uint32_t tile_w_log2;     /* lane tile is 2^k wide, 64/2^k high (default k = 3: 8x8)            */
	uint32_t xcd_swizzle;     /* 1: give each XCD a contiguous band of the batch (default: the hardware's
	                             round-robin placement, which balances cheap and expensive image regions) */
	uint32_t stack_override;  /* LDS stack entries per lane (lane kernel), >= what the BVH needs    */
	uint32_t tile_order;      /* 0: by scene size (Z-order once it exceeds the 256 MB Infinity Cache), 1: tiles in
	                             row-major order, 2: Z-order inside 16x16-tile super-tiles, 3: inside 32x32-tile ones,
	                             4: column strips per XCD (each XCD's waves on one compact image region; measured neutral) */
	uint32_t sort_key;        /* 0: origin cell + direction Morton key (default), 1: the reference's
	                             direction-only key (ray_sort.h:64-76); the order never changes results */
	uint32_t refill;          /* persistent lane kernel: refill a wave when this many lanes are idle (default 16) */
	uint32_t leaf_wait;       /* persistent lane kernel: leave the node phase when this many lanes stand at a
	                             leaf (default 8 for the 8-wide walk, else 16; 64 = classic while-while) */
	uint32_t extra_lds;       /* experiments: bytes of dynamic LDS added to every packet-kernel workgroup, which lowers
	                             the number of resident waves (occupancy sweeps, tools/exp_occupancy.py); <= 60000 */
	uint32_t packet_wg;       /* MRT_KERNEL_PACKET_DUAL: threads per workgroup, 64 or 256; 0 = by the size of the scene (64 up to
	                             256 MB of nodes + triangles: wave slots refill one by one; 256 above: the four waves of a
	                             workgroup walk neighbouring tiles through one scalar cache) */
	uint32_t packet_cull;     /* MRT_KERNEL_PACKET_DUAL: packet-level frustum culling (a child box wholly outside the pyramid of
	                             a packet's rays is skipped for all 128 of them; packets that are not a pinhole bundle never
	                             cull): 0 = library default (since round 3: on for rays generated in the kernel, mrt_cast_grid -- 14 % fewer
	                             vector instructions, C3 1.5 %, C5 3 % faster --, off for rays read from memory, where the walk
	                             with the scalar-cache prefetch is 2 % faster), 1 = off, 2 = on */
	uint32_t tile_schedule;   /* grid casts of 2^19 .. 2^24 rays: 0 = launch the tiles longest first by what each cost in the last cast of
	                             the same grid (every wave notes its shader cycles; a radix sort on a side stream makes the order;
	                             the first cast of a grid runs in the plain order), the few units that would end the frame alone in
	                             pieces (single tiles, quarter tiles); 1 = always the plain order; 2 = longest first, no pieces */
} mrt_options;

typedef struct mrt_ctx mrt_ctx;

/* ---- lifecycle: GPURayCaster::initialize / cleanup (gpu_ray_caster.cpp:72-183,700+) */
int mrt_create(int device_ordinal, const mrt_options *opts, mrt_ctx **out);
void mrt_destroy(mrt_ctx *ctx);
const char *mrt_last_error(const mrt_ctx *ctx);
const char *mrt_status_string(int status);
uint32_t mrt_version(void);
/* sizeof() of the boundary's structs as this library was compiled, for bindings in other languages to check
 * their own declarations against: 0 mrt_options, 1 mrt_camera, 2 mrt_stats, 3 mrt_instance (0 for anything else). */
uint32_t mrt_struct_size(uint32_t which);
/* Launch on this HIP stream (hipStream_t as void*; 0 = the context's own stream). */
int mrt_set_stream(mrt_ctx *ctx, void *hip_stream);
int mrt_synchronize(mrt_ctx *ctx);

/* ---- host-side scene preparation (replaces RayScene::build + the conversion
 *      half of upload_scene; no device needed) --------------------------------- */

/* Triangle ctor, src/core/triangle.h:41-51: edge1, edge2, normal. */
int mrt_make_triangles(const float *verts9, const uint32_t *ids, const uint32_t *layers,
		uint32_t n_tris, mrt_tri64 *out);
int mrt_pack_host_triangles(const mrt_host_tri80 *tris, uint32_t n_tris, mrt_tri64 *out);

/* 8-bin SAH BVH2 over triangle AABBs with TinyBVH's node/primIdx conventions
 * (replaces tinybvh::BVH::Build, tiny_bvh.h:2124-2136,2261-2466, called from
 * src/accel/ray_scene.h:62-86).  verts: 3*n_tris vertices, 16-byte stride
 * (bvhvec4).  nodes must hold 2*n_tris entries, prim_idx n_tris. */
int mrt_bvh2_build(const float *verts4, uint32_t n_tris, mrt_bvh_node32 *nodes,
		uint32_t *prim_idx, uint32_t *used_nodes, uint32_t n_threads);

/* BVH cache file, the counterpart of tinybvh::BVH::Save / Load (tiny_bvh.h:1747-1799): a scene
 * that did not change is not rebuilt.  The file holds used_nodes nodes and n_tris prim indices
 * behind a 32-byte header (magic, version, counts, checksum).  mrt_bvh2_load accepts a file only
 * for the triangle count it was saved for (as the reference does) and only if the checksum holds:
 * MRT_ERR_BAD_BVH otherwise, MRT_ERR_INVALID if the file cannot be opened.  nodes must hold
 * 2*n_tris entries, prim_idx n_tris.  Host-only; the triangles are not stored. */
int mrt_bvh2_save(const char *path, const mrt_bvh_node32 *nodes, uint32_t used_nodes,
		const uint32_t *prim_idx, uint32_t n_tris);
int mrt_bvh2_load(const char *path, uint32_t n_tris, mrt_bvh_node32 *nodes, uint32_t *prim_idx,
		uint32_t *used_nodes);

/* ---- scene upload: GPURayCaster::upload_scene (gpu_ray_caster.cpp:193-341) ---
 * tris are in original order (tris[i] is the triangle with prim index i in
 * prim_idx[]); the leaf -> prim_idx -> triangle indirection is resolved here
 * (reference defect: SURVEY.md section 0, item 1) and arrays are sized by
 * used_nodes (item 2).  Drains a pending async dispatch first (cpp:198-202). */
int mrt_upload_scene(mrt_ctx *ctx, const mrt_tri64 *tris, uint32_t n_tris,
		const mrt_bvh_node32 *nodes, uint32_t used_nodes, const uint32_t *prim_idx);
/* ---- build on the device: RayTracerServer::build (src/godot/raytracer_server.cpp:161-181 =
 * RayScene::build + upload_scene) for scenes that change too often to pay the host builder
 * (1.3 s per million triangles).  An LBVH (Morton sort + Karras radix tree, one triangle per
 * leaf) is built from the triangles in milliseconds, directly in device layout; casts return
 * exactly what they return against the host-built tree (results do not depend on which valid BVH
 * is walked) but walk more nodes per ray.  tris: host array, or device array with
 * MRT_BUILD_TRIS_ON_DEVICE.  mrt_stats.last_build_ms = device time of the build.
 * MRT_ERR_UNSUPPORTED if the tree comes out deeper than the traversal stack (build on the host). */
enum {
	MRT_BUILD_TRIS_ON_DEVICE = 1u << 0,
	MRT_BUILD_BLAS_ON_DEVICE = 1u << 2, /* mrt_upload_two_level_scene: every mesh's BVH built on the device (the radix tree; with
	                                       MRT_BUILD_SAH the binned-SAH tree) */
	MRT_BUILD_SAFE_HANDOFF   = 1u << 1, /* radix tree: the bottom-up pass hands boxes between threads with an
	                                       acquire-release counter from the start (3x slower).  Every build verifies its
	                                       tree afterwards and falls back to this form by itself if a hand-off was stale. */
	MRT_BUILD_PLOC           = 1u << 3, /* parallel locally-ordered clustering on the sorted keys (merges by surface area
	                                       of the union, Meister and Bittner 2018) instead of the default radix tree over
	                                       the key bits: 2.9 against 1.2 ms per million triangles; on the soup scenes of
	                                       BASELINE.md the two trees trace alike (1.05 / 1.07 x the host SAH tree) */
	MRT_BUILD_SAH            = 1u << 4  /* the binned-SAH tree of tinybvh::BVH::Build (tiny_bvh.h:2332-2466; mrt_bvh2_build on
	                                       the host) built level by level on the device: the host builder's decisions on the
	                                       same boxes, leaves of several triangles, rows in its depth-first order -- the tree
	                                       RayScene::build would upload, without the host build */
};
int mrt_build_scene_device(mrt_ctx *ctx, const mrt_tri64 *tris, uint32_t n_tris, uint32_t flags);

/* A placed mesh: MeshBLAS + BLASInstance (src/accel/mesh_blas.h:86-138, blas_instance.h:47-107).
 * Several instances may share one mesh (the same first_tri / n_tris). */
typedef struct mrt_instance {
	uint32_t first_tri;   /* the mesh: triangles [first_tri, first_tri + n_tris) of the mesh-space array */
	uint32_t n_tris;
	uint32_t layers;      /* the mesh's layer mask (raytracer_server.cpp:702-703)                    */
	uint32_t reserved;
	float basis[9];       /* Transform3D: world = basis (row-major 3x3) * v + origin                  */
	float origin[3];
} mrt_instance;

/* RayTracerServer::_rebuild_scene (src/godot/raytracer_server.cpp:669-711) on the device: every
 * instance's triangles to world space (Transform3D::xform per vertex, Triangle ctor), ids = running
 * triangle offset in instance order, layers = the mesh's mask; sum(n_tris) triangles into d_out
 * (device).  verts9: mesh-space vertices, 9 floats per triangle, host array or device array
 * (MRT_BUILD_TRIS_ON_DEVICE); instances: host array.  The reference flattens on the host every time
 * an instance moves and rebuilds; with a 288 GB device, flatten + rebuild per frame (a millisecond
 * per million triangles + mrt_build_scene_device) is one instancing path here; the two-level scene
 * below (mrt_upload_two_level_scene) is the other. */
int mrt_flatten_instances(mrt_ctx *ctx, const float *verts9, uint32_t n_mesh_tris,
		const mrt_instance *instances, uint32_t n_instances, uint32_t flags, mrt_tri64 *d_out);
/* mrt_flatten_instances into a scratch buffer + mrt_build_scene_device over it. */
int mrt_build_instanced_scene_device(mrt_ctx *ctx, const float *verts9, uint32_t n_mesh_tris,
		const mrt_instance *instances, uint32_t n_instances, uint32_t flags);

/* ---- two-level scene: SceneTLAS + MeshBLAS + BLASInstance (src/accel/scene_tlas.h:140-251,
 * mesh_blas.h:86-138, blas_instance.h:47-107; tinybvh::BVH::IntersectTLAS, tiny_bvh.h:3306-3380).
 * Nothing is flattened: one BVH (binned SAH) per distinct mesh in mesh space, one BVH over the
 * instances' world boxes; a ray entering an instance is taken to mesh space by the inverse
 * transform without renormalising its direction, so t stays world-parameterised.  Hit records:
 * prim_id = the FLAT id of raytracer_server.cpp:700-711 (the instance's running triangle offset +
 * the mesh-local index; the reference's TLAS path reports the local index, SURVEY.md section 0
 * item 4), hit_layers = the instance's mask (whole instances are skipped by the query mask),
 * normal = normalize(basis * mesh-space normal), position on the world ray.  Every cast entry
 * point works on such a scene; hit tokens are 8 bytes there ({triangle, instance}: mrt_token_bytes).
 * verts9 / instances: host arrays.  Transforms must be invertible (MRT_ERR_INVALID).
 * flags: 0, or MRT_BUILD_BLAS_ON_DEVICE to build the meshes' BVHs with the device builder of
 * mrt_build_scene_device (milliseconds instead of 0.3 s per million triangles; the same hit
 * records; meshes of one triangle and trees deeper than the stack need the host builder:
 * MRT_ERR_UNSUPPORTED).  mrt_stats.last_build_ms = device time of that build. */
int mrt_upload_two_level_scene(mrt_ctx *ctx, const float *verts9, uint32_t n_mesh_tris,
		const mrt_instance *instances, uint32_t n_instances, uint32_t flags);
/* SceneTLAS::set_instance_transform + refit_tlas (scene_tlas.h:118-134,178-196): the same
 * instances (same meshes, same order) with new transforms / masks.  Only the top level is rebuilt
 * and re-uploaded (n_instances rows + fewer than 2 n_instances nodes). */
int mrt_update_instances(mrt_ctx *ctx, const mrt_instance *instances, uint32_t n_instances);

/* The prepared two-level scene ON THE HOST: exactly the arrays mrt_upload_two_level_scene uploads, for a host whose
 * router also has a CPU backend (RayDispatcher::_cpu_cast_rays routes to the TLAS when there is one,
 * src/dispatch/ray_dispatcher.h:443-452; the mirror's walk is csrc/host/cpu_backend.hpp, CpuTwoLevelWalker).  Host-only:
 * no device, no context.  Layouts (all little-endian 32-bit words):
 *   nodes      64-byte rows {lmin xyz, left ref | lmax xyz, right ref | rmin xyz, - | rmax xyz, -}; a ref < 0x7FFFFFFF is a
 *              node index, a ref >= 0x80000000 a leaf: its low 31 bits = first row of a run of instances (TLAS,
 *              nodes [0, n_tlas_nodes)) or of triangles (a BLAS); node 0 is the TLAS root
 *   tri_hot    48-byte rows {v0 xyz, mesh-local id | e1 xyz, layers | e2 xyz, flags}; flags & 1 = last triangle of its leaf
 *   tri_cold   16-byte rows {normal xyz, -} (mesh space)
 *   instances  128-byte rows in TLAS leaf order: 12 floats inverse transform (rows {m00 m01 m02 t}), 9 floats basis,
 *              u32 root node of the BLAS, u32 flat id of the instance's first triangle, u32 layer mask,
 *              u32 flags (1 = last instance of its TLAS leaf), u32 registration index, 7 words unused */
typedef struct mrt_two_level_host mrt_two_level_host;
typedef struct mrt_two_level_arrays {
	const mrt_bvh_node_wide64 *nodes; uint32_t n_nodes, n_tlas_nodes;
	const float *tri_hot, *tri_cold; uint32_t n_tris;
	const float *instances; uint32_t n_instances;
	uint32_t depth;       /* stack entries one ray can need */
} mrt_two_level_arrays;
int mrt_two_level_prepare_host(const float *verts9, uint32_t n_mesh_tris, const mrt_instance *instances, uint32_t n_instances,
		uint32_t n_threads, mrt_two_level_host **out);
int mrt_two_level_host_arrays(const mrt_two_level_host *h, mrt_two_level_arrays *out);
void mrt_two_level_free_host(mrt_two_level_host *h);

int mrt_is_available(const mrt_ctx *ctx);      /* initialized && scene uploaded */
int mrt_scene_info(const mrt_ctx *ctx, uint32_t *n_tris, uint32_t *n_wide_nodes, uint32_t *bvh_depth);

/* ---- casting: GPURayCaster::cast_rays / cast_rays_any_hit (cpp:417-488) and
 *      RayDispatcher's sort policy (ray_dispatcher.h:135-148).  Blocking. ------- */
int mrt_cast(mrt_ctx *ctx, const void *rays, void *hits, uint64_t count,
		uint32_t query_mask, int mode, uint32_t flags);

/* ---- async: submit_async* / collect_* (cpp:536-623); one pending dispatch --- */
int mrt_submit(mrt_ctx *ctx, const void *rays, uint64_t count,
		uint32_t query_mask, int mode, uint32_t flags);
int mrt_collect(mrt_ctx *ctx, void *hits, uint64_t count);
int mrt_has_pending(const mrt_ctx *ctx);

/* ---- primary-ray grids on the device (raytracer_debug.cpp:572-596) ---------- */
int mrt_camera_look(mrt_camera *cam, const float origin[3], const float forward[3],
		uint32_t grid_w, uint32_t grid_h, float fov_degrees);
/* RayCamera::setup + _setup_perspective / _setup_orthographic (ray_camera.h:50-76,208-230) without the
 * Camera3D: origin and basis (row-major 3x3 = Godot's Basis rows) are the camera transform, width x height
 * the resolution the grid will be cast at.  Pass the camera to mrt_generate_grid / mrt_cast_grid /
 * mrt_expand_grid_tokens with the same width and height. */
int mrt_camera_perspective(mrt_camera *cam, const float origin[3], const float basis[9],
		uint32_t width, uint32_t height, float fov_degrees);
int mrt_camera_orthographic(mrt_camera *cam, const float origin[3], const float basis[9],
		uint32_t width, uint32_t height, float size);
/* rows [y0,y1) of a grid_w x grid_h grid, row-major from row y0, into d_rays. */
int mrt_generate_grid(mrt_ctx *ctx, const mrt_camera *cam, uint32_t grid_w, uint32_t grid_h,
		uint32_t y0, uint32_t y1, mrt_ray32 *d_rays);
/* Fused: generate rows [y0,y1) and trace them; hits row-major from row y0
 * (device pointer iff MRT_FLAG_HITS_ON_DEVICE). */
int mrt_cast_grid(mrt_ctx *ctx, const mrt_camera *cam, uint32_t grid_w, uint32_t grid_h,
		uint32_t y0, uint32_t y1, void *hits, uint32_t query_mask, int mode, uint32_t flags);
/* Trace device-resident rays that the caller declares to be a row-major
 * grid_w-wide grid (lets the kernel tile lanes 8x8 instead of 64x1). */
int mrt_cast_tiled(mrt_ctx *ctx, const mrt_ray32 *d_rays, mrt_hit32 *d_hits,
		uint32_t grid_w, uint32_t rows, uint32_t query_mask, int mode);

/* ---- hit tokens -> hit records (no reference counterpart: the reference is single-device).
 * The packed->Intersection readback conversion of gpu_ray_caster.cpp:442-456 applied to
 * tokens written by a cast with MRT_FLAG_TOKEN_OUT: mrt_hit32 records (mrt_host_hit44 with
 * MRT_FLAG_HOST_LAYOUT, rays then being mrt_host_ray60), identical to what the cast would
 * have written without the flag.  Device pointers only.  Asynchronous: enqueued on
 * `hip_stream` (hipStream_t as void*; 0 = the context's stream) without waiting. */
int mrt_expand_tokens(mrt_ctx *ctx, const void *d_rays, const uint32_t *d_tokens, void *d_hits,
		uint64_t count, uint32_t flags, void *hip_stream);
/* Bytes per hit token of the scene this context holds: 4 (flat scene), 8 (two-level scene); 0 for a null context. */
uint32_t mrt_token_bytes(mrt_ctx *ctx);
/* Same for rows [y0,y1) of a camera grid (tokens from mrt_cast_grid, on this or another device). */
int mrt_expand_grid_tokens(mrt_ctx *ctx, const mrt_camera *cam, uint32_t grid_w, uint32_t grid_h,
		uint32_t y0, uint32_t y1, const uint32_t *d_tokens, mrt_hit32 *d_hits, void *hip_stream);

/* ---- Morton keys (src/dispatch/ray_sort.h:41-76), exposed for parity tests ---- */
int mrt_morton_keys(mrt_ctx *ctx, const mrt_ray32 *d_rays, uint64_t count, uint32_t *d_keys);

/* ---- stats / device memory helpers ---------------------------------------- */
int mrt_get_stats(mrt_ctx *ctx, mrt_stats *out);
/* The template instantiation that did the work of the last blocking cast, spelled as rocprofv3 prints kernel names
 * (e.g. "trace_packet_rows_kernel<false, false, 2, 64, true>"); "" before the first cast and after an ASYNC one.  No
 * reference counterpart (the reference prints its pipeline choice, gpu_ray_caster.cpp:654-671); bench.py uses it to
 * accept committed counter passes only for the very kernel a run used. */
const char *mrt_last_kernel_variant(mrt_ctx *ctx);
/* 1 if this build contains the kernel.  MRT_KERNEL_PACKET_QUAD (the four-wide packet walk: held to the oracle, not faster
 * than the default) is compiled only into builds made with MRT_WITH_QUAD=1 (messyerraytracer_amd/build.py); mrt_create
 * with it returns MRT_ERR_UNSUPPORTED otherwise. */
int mrt_kernel_available(uint32_t kernel);
int mrt_device_alloc(mrt_ctx *ctx, size_t bytes, void **d_ptr);
int mrt_device_free(mrt_ctx *ctx, void *d_ptr);
int mrt_memcpy_h2d(mrt_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int mrt_memcpy_d2h(mrt_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);

/* ---- several devices of one node from ONE process (no reference counterpart: the reference is
 * single-device, and its callers — RayDispatcher, src/dispatch/ray_dispatcher.h:124-181; RayTracerServer, src/godot/
 * raytracer_server.cpp:285-328 — are C++, which the torch.distributed path of bench.py cannot serve).  A group holds one
 * context and one stream per member; the scene is replicated; a grid's rows are split into contiguous blocks
 * (mrt_group_row_block), every member traces its block with rays generated in the kernel, 4-byte hit tokens travel to
 * member 0 as peer copies (xGMI, each peer over its own link) and member 0 rebuilds the records, bit-identical to a
 * single-device cast of the whole grid.  device_ordinals: n_devices HIP ordinals, NULL = 0 .. n_devices-1; an ordinal
 * may repeat (several members on one device: how the multi-member path is exercised on a one-GPU box).
 * Externally serialised like a context. */
typedef struct mrt_group mrt_group;
int mrt_group_create(int n_devices, const int *device_ordinals, const mrt_options *opts, mrt_group **out);
void mrt_group_destroy(mrt_group *group);
int mrt_group_size(const mrt_group *group);
mrt_ctx *mrt_group_context(mrt_group *group, int member);   /* e.g. for mrt_get_stats of one member */
const char *mrt_group_last_error(const mrt_group *group);
void mrt_group_row_block(uint32_t member, uint32_t n_members, uint32_t rows, uint32_t *y0, uint32_t *y1);
int mrt_group_upload_scene(mrt_group *group, const mrt_tri64 *tris, uint32_t n_tris,
		const mrt_bvh_node32 *nodes, uint32_t used_nodes, const uint32_t *prim_idx);
int mrt_group_upload_two_level_scene(mrt_group *group, const float *verts9, uint32_t n_mesh_tris,
		const mrt_instance *instances, uint32_t n_instances, uint32_t flags);
/* hits: grid_w * grid_h records row-major (mrt_hit32; uint8 with MRT_FLAG_BOOL_OUT in any-hit mode), on the host, or in
 * member 0's device memory with MRT_FLAG_HITS_ON_DEVICE (the only other flag accepted).  Blocking. */
int mrt_group_cast_grid(mrt_group *group, const mrt_camera *cam, uint32_t grid_w, uint32_t grid_h, void *hits,
		uint32_t query_mask, int mode, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif /* MRT_HIP_H_ */
