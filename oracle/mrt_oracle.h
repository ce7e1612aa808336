/*
 * mrt_oracle.h — CPU restatement of the reference's batch ray-cast path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under messyerraytracer_amd/ may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Parity status: PINNED against the reference itself.  oracle/_ref (the
 * reference's vendored TinyBVH 1.6.7 compiled from /root/reference by
 * oracle/Makefile) produced tests/golden/ (npz files) via tests/golden/make_golden.py;
 * tests/test_oracle_golden.py checks this restatement against those vectors.
 * The reference ships no tests of its own.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference).
 */
#ifndef MRT_ORACLE_H_
#define MRT_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float origin[3]; float t_max; float direction[3]; float t_min; } orc_ray32;   /* src/api/gpu_types.h:65-69 */
typedef struct { float t; int32_t prim_id; float bary_u, bary_v; float normal[3]; uint32_t hit_layers; } orc_hit32; /* :87-92 */
typedef struct { float v0[3]; uint32_t id; float edge1[3]; uint32_t layers; float edge2[3]; float pad2; float normal[3]; float pad3; } orc_tri64; /* :44-50 */
typedef struct { float aabb_min[3]; uint32_t left_first; float aabb_max[3]; uint32_t tri_count; } orc_node32; /* tiny_bvh.h:857-866 */
typedef struct { float lmin[3]; uint32_t left_idx; float lmax[3]; uint32_t right_idx;
                 float rmin[3]; uint32_t left_count; float rmax[3]; uint32_t right_count; } orc_wide64; /* src/gpu/gpu_structs.h:41-47 */
typedef struct { float origin[3], direction[3], inv_direction[3]; int32_t dir_sign[3]; float t_min, t_max; uint32_t flags; } orc_host_ray60; /* src/core/ray.h:25-51 */
typedef struct { float t; float position[3]; float normal[3]; float u, v; uint32_t prim_id; uint32_t hit_layers; } orc_host_hit44; /* src/core/intersection.h:16-40 */

typedef struct {
	uint64_t rays, hits, node_visits, tri_tests; /* node_visits = wide (internal) node fetches */
	uint32_t max_stack;
} orc_counters;

/* src/core/triangle.h:41-51 */
void orc_make_triangles(const float *verts9, const uint32_t *ids, const uint32_t *layers, uint32_t n, orc_tri64 *out);
/* a placed mesh: MeshBLAS + BLASInstance (src/accel/mesh_blas.h:86-138, blas_instance.h:47-107) */
typedef struct { uint32_t first_tri, n_tris, layers, reserved; float basis[9]; float origin[3]; } orc_instance;
/* scene flatten of RayTracerServer::_rebuild_scene (src/godot/raytracer_server.cpp:700-711) */
void orc_flatten_instances(const float *verts9, const orc_instance *inst, uint32_t n_inst, orc_tri64 *out);
/* tiny_bvh.h:2261-2330 (PrepareBuild) + :2332-2466 (Build), single-threaded numbering */
int orc_bvh2_build(const float *verts4, uint32_t n_tris, orc_node32 *nodes, uint32_t *prim_idx, uint32_t *used_nodes);
/* tiny_bvh.h:1889-1897, :3698-3728 + depth */
void orc_bvh2_info(const orc_node32 *nodes, uint32_t *node_count, uint32_t *leaf_count, uint32_t *depth, float *sah_cost, uint32_t *max_leaf);
/* src/gpu/gpu_ray_caster.cpp:205-311 with SURVEY section 0 defects 1+2 fixed */
int orc_to_wide(const orc_tri64 *tris, uint32_t n_tris, const orc_node32 *nodes, uint32_t used_nodes,
		const uint32_t *prim_idx, orc_wide64 *wide, uint32_t *n_wide, orc_tri64 *leaf_tris);
/* src/gpu/shaders/bvh_traverse.comp.glsl:198-328 */
void orc_trace(const orc_wide64 *wide, const orc_tri64 *leaf_tris, const orc_ray32 *rays, orc_hit32 *hits,
		uint64_t count, uint32_t query_mask, int any_hit, orc_counters *ctr, int n_threads);
/* src/accel/ray_scene.h:120-131,151-162 with the acceptance rules of triangle.h:56-105 / glsl:105-131 */
void orc_trace_brute(const orc_tri64 *tris, uint32_t n_tris, const orc_ray32 *rays, orc_hit32 *hits,
		uint64_t count, uint32_t query_mask, int any_hit, int n_threads);
/* Two-level scene: src/accel/scene_tlas.h:140-251 (build_tlas, cast_ray, any_hit), mesh_blas.h:86-138,
 * blas_instance.h:47-107, tinybvh IntersectTLAS tiny_bvh.h:3306-3380.  prim_id = flat id
 * (raytracer_server.cpp:700-711), hit_layers = the instance's mask; see the comment in the .c file. */
typedef struct orc_two_level orc_two_level;
orc_two_level *orc_two_level_build(const float *verts9, uint32_t n_mesh_tris, const orc_instance *inst, uint32_t n_inst);
void orc_two_level_free(orc_two_level *s);
void orc_two_level_trace(const orc_two_level *s, const orc_ray32 *rays, orc_hit32 *hits, uint64_t count,
		uint32_t query_mask, int any_hit, int n_threads);
/* Single ray vs single triangle; returns 1 and t/u/v when accepted with best_t = ray.t_max */
int orc_tri_test(const orc_tri64 *tri, const orc_ray32 *ray, float *t, float *u, float *v);
/* src/godot/raytracer_debug.cpp:572-596 */
void orc_camera_basis(const float forward[3], uint32_t w, uint32_t h, float fov_deg,
		float fwd[3], float right[3], float up[3], float *half_w, float *half_h);
void orc_ray_camera_rays(const float origin[3], const float basis[9], uint32_t w, uint32_t h, float param, int ortho,
		float jx, float jy, uint32_t y0, uint32_t y1, orc_ray32 *out);
void orc_grid_rays(const float origin[3], const float forward[3], uint32_t w, uint32_t h, float fov_deg,
		uint32_t y0, uint32_t y1, orc_ray32 *out);
/* src/dispatch/ray_sort.h:41-76 */
uint32_t orc_morton_key(const float dir[3]);
void orc_morton_keys(const orc_ray32 *rays, uint64_t count, uint32_t *keys);
/* src/gpu/gpu_ray_caster.cpp:639-650 and :442-456 */
void orc_pack_rays(const orc_host_ray60 *rays, uint64_t count, orc_ray32 *out);
void orc_make_host_rays(const orc_ray32 *rays, uint64_t count, orc_host_ray60 *out); /* src/core/ray.h:53-96 */
void orc_unpack_hits(const orc_hit32 *hits, const orc_host_ray60 *rays, uint64_t count, orc_host_hit44 *out);

#ifdef __cplusplus
}
#endif
#endif
