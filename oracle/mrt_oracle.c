/*
 * mrt_oracle.c — plain-C restatement of the reference's batch ray-cast path.
 * TEST INFRASTRUCTURE ONLY (see mrt_oracle.h).  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off -mfma -fopenmp; contraction is OFF so every
 * rounding below is the one written; the explicit fmaf() calls are the
 * "canonical arithmetic" the HIP kernels reproduce bit for bit, DESIGN.md
 * section "Arithmetic").
 */
#include "mrt_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BVH_FAR 1e30f /* tiny_bvh.h:140 */
#define BVHBINS 8     /* tiny_bvh.h:104-106 */

/* ------------------------------------------------------------------------- */
/* small vector helpers: the canonical operation order                        */
/* ------------------------------------------------------------------------- */
static inline void v_cross(const float a[3], const float b[3], float r[3])
{
	r[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
	r[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
	r[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
static inline float v_dot(const float a[3], const float b[3])
{
	return fmaf(a[0], b[0], fmaf(a[1], b[1], a[2] * b[2]));
}
/* godot::Vector3::normalized(): zero stays zero, else divide by sqrt(len2)
 * (godot-cpp is an empty submodule in the reference, .gitmodules:1-4; this is
 * its published algorithm).  Plain (uncontracted) products, as MSVC /fp:precise. */
static inline void v_normalize(float a[3])
{
	float l2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
	if (l2 == 0.0f) { a[0] = a[1] = a[2] = 0.0f; return; }
	float l = sqrtf(l2);
	a[0] /= l; a[1] /= l; a[2] /= l;
}

/* src/core/triangle.h:41-51 — Triangle ctor: edge1, edge2, normal. */
void orc_make_triangles(const float *verts9, const uint32_t *ids, const uint32_t *layers, uint32_t n, orc_tri64 *out)
{
	for (uint32_t i = 0; i < n; i++) {
		const float *a = verts9 + 9 * (size_t)i, *b = a + 3, *c = a + 6;
		orc_tri64 *t = &out[i];
		for (int k = 0; k < 3; k++) {
			t->v0[k] = a[k];
			t->edge1[k] = b[k] - a[k];
			t->edge2[k] = c[k] - a[k];
		}
		/* Vector3::cross: plain products and differences */
		float nrm[3] = {
			t->edge1[1] * t->edge2[2] - t->edge1[2] * t->edge2[1],
			t->edge1[2] * t->edge2[0] - t->edge1[0] * t->edge2[2],
			t->edge1[0] * t->edge2[1] - t->edge1[1] * t->edge2[0] };
		v_normalize(nrm);
		t->normal[0] = nrm[0]; t->normal[1] = nrm[1]; t->normal[2] = nrm[2];
		t->id = ids ? ids[i] : i;
		t->layers = layers ? layers[i] : 0xFFFFFFFFu;
		t->pad2 = 0.0f; t->pad3 = 0.0f;
	}
}

/* ------------------------------------------------------------------------- */
/* BVH2 builder: tiny_bvh.h:2261-2330 (PrepareBuild), :2332-2466 (Build)       */
/* ------------------------------------------------------------------------- */
typedef struct { float bmin[3], bmax[3]; } frag_t;

static inline float tb_min(float a, float b) { return a < b ? a : b; } /* tiny_bvh.h:446 */
static inline float tb_max(float a, float b) { return a > b ? a : b; } /* tiny_bvh.h:447 */
static inline int tb_clampi(int x, int a, int b) { return x > a ? (x < b ? x : b) : a; } /* :459 */
/* tiny_bvh.h:460 */
static inline float tb_half_area(const float v[3])
{
	return v[0] < -BVH_FAR ? 0.0f : (v[0] * v[1] + v[1] * v[2] + v[2] * v[0]);
}
/* x86 cvttss2si semantics for (int)float, which is what the compiled reference does */
static inline int f2i(float f)
{
	if (!(f > -2147483648.0f && f < 2147483648.0f)) return (int)0x80000000u;
	return (int)f;
}

int orc_bvh2_build(const float *verts4, uint32_t n, orc_node32 *nodes, uint32_t *prim_idx, uint32_t *used_nodes)
{
	if (!verts4 || !nodes || !prim_idx || !used_nodes || n == 0) return 1;
	frag_t *frag = (frag_t *)malloc((size_t)n * sizeof(frag_t));
	if (!frag) return 7;
	/* PrepareBuild :2290-2311 */
	memset(&nodes[1], 0, sizeof(orc_node32)); /* node 1 remains unused (:2285) */
	orc_node32 *root = &nodes[0];
	root->left_first = 0; root->tri_count = n;
	for (int k = 0; k < 3; k++) root->aabb_min[k] = BVH_FAR, root->aabb_max[k] = -BVH_FAR;
	for (uint32_t i = 0; i < n; i++) {
		const float *v0 = verts4 + 12 * (size_t)i, *v1 = v0 + 4, *v2 = v0 + 8;
		for (int k = 0; k < 3; k++) {
			frag[i].bmin[k] = tb_min(v0[k], tb_min(v1[k], v2[k]));
			frag[i].bmax[k] = tb_max(v0[k], tb_max(v1[k], v2[k]));
			root->aabb_min[k] = tb_min(root->aabb_min[k], frag[i].bmin[k]);
			root->aabb_max[k] = tb_max(root->aabb_max[k], frag[i].bmax[k]);
		}
		prim_idx[i] = i;
	}
	uint32_t new_node_ptr = 2; /* :2326 */
	/* Build :2346-2445 */
	const float c_trav = 1.0f, c_int = 1.0f; /* C_TRAV, C_INT :126-131 */
	uint32_t task[256], task_count = 0, node_idx = 0;
	float min_dim[3];
	for (int k = 0; k < 3; k++) min_dim[k] = (root->aabb_max[k] - root->aabb_min[k]) * 1e-20f;
	float best_lmin[3] = {0, 0, 0}, best_lmax[3] = {0, 0, 0}, best_rmin[3] = {0, 0, 0}, best_rmax[3] = {0, 0, 0};
	for (;;) {
		for (;;) {
			orc_node32 *node = &nodes[node_idx];
			float bin_min[3][BVHBINS][3], bin_max[3][BVHBINS][3];
			uint32_t count[3][BVHBINS];
			for (int a = 0; a < 3; a++) for (int i = 0; i < BVHBINS; i++) {
				for (int k = 0; k < 3; k++) bin_min[a][i][k] = BVH_FAR, bin_max[a][i][k] = -BVH_FAR;
				count[a][i] = 0;
			}
			float rpd3[3], nmin3[3];
			for (int k = 0; k < 3; k++) {
				rpd3[k] = (float)BVHBINS / (node->aabb_max[k] - node->aabb_min[k]);
				nmin3[k] = node->aabb_min[k];
			}
			for (uint32_t i = 0; i < node->tri_count; i++) { /* :2363-2376 */
				const uint32_t fi = prim_idx[node->left_first + i];
				const frag_t *f = &frag[fi];
				for (int a = 0; a < 3; a++) {
					int bi = f2i(((f->bmin[a] + f->bmax[a]) * 0.5f - nmin3[a]) * rpd3[a]);
					bi = tb_clampi(bi, 0, BVHBINS - 1);
					for (int k = 0; k < 3; k++) {
						bin_min[a][bi][k] = tb_min(bin_min[a][bi][k], f->bmin[k]);
						bin_max[a][bi][k] = tb_max(bin_max[a][bi][k], f->bmax[k]);
					}
					count[a][bi]++;
				}
			}
			/* per-split totals :2377-2405 */
			float ext[3] = { node->aabb_max[0] - node->aabb_min[0], node->aabb_max[1] - node->aabb_min[1],
				node->aabb_max[2] - node->aabb_min[2] };
			float split_cost = BVH_FAR, rsav = 1.0f / (ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0]);
			uint32_t best_axis = 0, best_pos = 0;
			for (int a = 0; a < 3; a++) if ((node->aabb_max[a] - node->aabb_min[a]) > min_dim[a]) {
				float lbmin[BVHBINS - 1][3], rbmin[BVHBINS - 1][3], lbmax[BVHBINS - 1][3], rbmax[BVHBINS - 1][3];
				float l1[3] = { BVH_FAR, BVH_FAR, BVH_FAR }, l2[3] = { -BVH_FAR, -BVH_FAR, -BVH_FAR };
				float r1[3] = { BVH_FAR, BVH_FAR, BVH_FAR }, r2[3] = { -BVH_FAR, -BVH_FAR, -BVH_FAR };
				float anl[BVHBINS - 1], anr[BVHBINS - 1];
				uint32_t ln = 0, rn = 0;
				for (int i = 0; i < BVHBINS - 1; i++) {
					float dl[3], dr[3];
					for (int k = 0; k < 3; k++) {
						lbmin[i][k] = l1[k] = tb_min(l1[k], bin_min[a][i][k]);
						rbmin[BVHBINS - 2 - i][k] = r1[k] = tb_min(r1[k], bin_min[a][BVHBINS - 1 - i][k]);
						lbmax[i][k] = l2[k] = tb_max(l2[k], bin_max[a][i][k]);
						rbmax[BVHBINS - 2 - i][k] = r2[k] = tb_max(r2[k], bin_max[a][BVHBINS - 1 - i][k]);
						dl[k] = l2[k] - l1[k]; dr[k] = r2[k] - r1[k];
					}
					ln += count[a][i]; rn += count[a][BVHBINS - 1 - i];
					anl[i] = ln == 0 ? BVH_FAR : (tb_half_area(dl) * (float)ln);
					anr[BVHBINS - 2 - i] = rn == 0 ? BVH_FAR : (tb_half_area(dr) * (float)rn);
				}
				for (int i = 0; i < BVHBINS - 1; i++) {
					const float c = anl[i] + anr[i];
					if (c < split_cost) {
						split_cost = c; best_axis = (uint32_t)a; best_pos = (uint32_t)i;
						for (int k = 0; k < 3; k++) {
							best_lmin[k] = lbmin[i][k]; best_rmin[k] = rbmin[i][k];
							best_lmax[k] = lbmax[i][k]; best_rmax[k] = rbmax[i][k];
						}
					}
				}
			}
			split_cost = c_trav + c_int * rsav * split_cost; /* :2406 */
			float no_split_cost = (float)node->tri_count * c_int;
			if (split_cost >= no_split_cost) break; /* :2408-2412 */
			/* in-place partition :2413-2422 */
			uint32_t j = node->left_first + node->tri_count, src = node->left_first;
			const float rpd = rpd3[best_axis], nmin = nmin3[best_axis];
			for (uint32_t i = 0; i < node->tri_count; i++) {
				const uint32_t fi = prim_idx[src];
				int bi = (int)(uint32_t)(long long)(((frag[fi].bmin[best_axis] + frag[fi].bmax[best_axis]) * 0.5f - nmin) * rpd);
				bi = tb_clampi(bi, 0, BVHBINS - 1);
				if ((uint32_t)bi <= best_pos) src++;
				else { uint32_t t = prim_idx[src]; prim_idx[src] = prim_idx[--j]; prim_idx[j] = t; }
			}
			/* child nodes :2423-2432 */
			uint32_t left_count = src - node->left_first, right_count = node->tri_count - left_count;
			if (left_count == 0 || right_count == 0 || task_count == 256) break;
			uint32_t nn = new_node_ptr; new_node_ptr += 2;
			for (int k = 0; k < 3; k++) {
				nodes[nn].aabb_min[k] = best_lmin[k]; nodes[nn].aabb_max[k] = best_lmax[k];
				nodes[nn + 1].aabb_min[k] = best_rmin[k]; nodes[nn + 1].aabb_max[k] = best_rmax[k];
			}
			nodes[nn].left_first = node->left_first; nodes[nn].tri_count = left_count;
			nodes[nn + 1].left_first = j; nodes[nn + 1].tri_count = right_count;
			node->left_first = nn; node->tri_count = 0;
			task[task_count++] = nn + 1; node_idx = nn; /* :2441 */
		}
		if (task_count == 0) break; else node_idx = task[--task_count]; /* :2444 */
	}
	*used_nodes = new_node_ptr; /* :2454 */
	free(frag);
	return 0;
}

/* tiny_bvh.h:1889-1897 (SAHCost), :3698-3728 (NodeCount/LeafCount), depth */
static float sah_rec(const orc_node32 *nodes, uint32_t idx)
{
	const orc_node32 *n = &nodes[idx];
	float e[3] = { n->aabb_max[0] - n->aabb_min[0], n->aabb_max[1] - n->aabb_min[1], n->aabb_max[2] - n->aabb_min[2] };
	float sa = e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
	if (n->tri_count > 0) return 1.0f * sa * (float)n->tri_count;
	float cost = 1.0f * sa + sah_rec(nodes, n->left_first) + sah_rec(nodes, n->left_first + 1);
	return idx == 0 ? (cost / sa) : cost;
}
void orc_bvh2_info(const orc_node32 *nodes, uint32_t *node_count, uint32_t *leaf_count, uint32_t *depth, float *sah_cost, uint32_t *max_leaf)
{
	uint32_t nc = 0, lc = 0, md = 0, ml = 0, sp = 0;
	struct { uint32_t idx, d; } stack[512];
	stack[sp].idx = 0; stack[sp].d = 1; sp++;
	while (sp) {
		sp--;
		uint32_t idx = stack[sp].idx, d = stack[sp].d;
		const orc_node32 *n = &nodes[idx];
		nc++;
		if (d > md) md = d;
		if (n->tri_count > 0) { lc++; if (n->tri_count > ml) ml = n->tri_count; continue; }
		stack[sp].idx = n->left_first + 1; stack[sp].d = d + 1; sp++;
		stack[sp].idx = n->left_first; stack[sp].d = d + 1; sp++;
	}
	if (node_count) *node_count = nc;
	if (leaf_count) *leaf_count = lc;
	if (depth) *depth = md;
	if (max_leaf) *max_leaf = ml;
	if (sah_cost) *sah_cost = sah_rec(nodes, 0);
}

/* ------------------------------------------------------------------------- */
/* BVH2 -> Aila-Laine wide nodes: src/gpu/gpu_ray_caster.cpp:205-311           */
/* Fixes: leaf triangles are gathered through prim_idx (defect 1), traversal   */
/* covers every node below used_nodes (defect 2).  A root that is a leaf is    */
/* wrapped as two leaf children that split its triangle range (the reference   */
/* wraps it with a NaN right box, :255-271; IEEE fmin/fmax drop NaNs so a      */
/* plain duplicate-free split is used instead; results are identical).         */
/* ------------------------------------------------------------------------- */
int orc_to_wide(const orc_tri64 *tris, uint32_t n_tris, const orc_node32 *nodes, uint32_t used_nodes,
		const uint32_t *prim_idx, orc_wide64 *wide, uint32_t *n_wide, orc_tri64 *leaf_tris)
{
	if (!tris || !nodes || !prim_idx || !wide || !n_wide || !leaf_tris || used_nodes == 0) return 1;
	for (uint32_t k = 0; k < n_tris; k++) {
		if (prim_idx[k] >= n_tris) return 9;
		leaf_tris[k] = tris[prim_idx[k]];
	}
	const orc_node32 *root = &nodes[0];
	if (root->tri_count > 0) {
		orc_wide64 *g = &wide[0];
		uint32_t lc = (root->tri_count + 1) / 2, rc = root->tri_count - lc;
		for (int k = 0; k < 3; k++) {
			g->lmin[k] = g->rmin[k] = root->aabb_min[k];
			g->lmax[k] = g->rmax[k] = root->aabb_max[k];
		}
		g->left_idx = root->left_first; g->left_count = lc;
		if (rc == 0) { g->right_idx = root->left_first; g->right_count = lc; }
		else { g->right_idx = root->left_first + lc; g->right_count = rc; }
		*n_wide = 1;
		return 0;
	}
	/* number internal nodes in DFS preorder */
	uint32_t *map = (uint32_t *)malloc((size_t)used_nodes * sizeof(uint32_t));
	uint32_t *stack = (uint32_t *)malloc(1024 * sizeof(uint32_t));
	if (!map || !stack) { free(map); free(stack); return 7; }
	uint32_t sp = 0, nw = 0;
	stack[sp++] = 0;
	while (sp) {
		uint32_t i = stack[--sp];
		if (i >= used_nodes || sp > 1000) { free(map); free(stack); return 9; }
		map[i] = nw++;
		uint32_t l = nodes[i].left_first, r = l + 1;
		if (r >= used_nodes) { free(map); free(stack); return 9; }
		if (nodes[r].tri_count == 0) stack[sp++] = r;
		if (nodes[l].tri_count == 0) stack[sp++] = l;
	}
	sp = 0; stack[sp++] = 0;
	while (sp) {
		uint32_t i = stack[--sp];
		orc_wide64 *g = &wide[map[i]];
		uint32_t l = nodes[i].left_first, r = l + 1;
		const orc_node32 *lc = &nodes[l], *rc = &nodes[r];
		for (int k = 0; k < 3; k++) {
			g->lmin[k] = lc->aabb_min[k]; g->lmax[k] = lc->aabb_max[k];
			g->rmin[k] = rc->aabb_min[k]; g->rmax[k] = rc->aabb_max[k];
		}
		if (lc->tri_count > 0) { g->left_idx = lc->left_first; g->left_count = lc->tri_count; }
		else { g->left_idx = map[l]; g->left_count = 0; }
		if (rc->tri_count > 0) { g->right_idx = rc->left_first; g->right_count = rc->tri_count; }
		else { g->right_idx = map[r]; g->right_count = 0; }
		if (rc->tri_count == 0) stack[sp++] = r;
		if (lc->tri_count == 0) stack[sp++] = l;
	}
	*n_wide = nw;
	free(map); free(stack);
	return 0;
}

/* ------------------------------------------------------------------------- */
/* traversal: src/gpu/shaders/bvh_traverse.comp.glsl                           */
/* ------------------------------------------------------------------------- */

/* safe_inv_direction, glsl:137-145 == Ray::_precompute, src/core/ray.h:78-89 */
static inline float safe_inv(float d)
{
	const float eps = 1e-9f;
	const float big = 1.0f / eps;
	return fabsf(d) > eps ? 1.0f / d : (d >= 0.0f ? big : -big);
}

/* ray_aabb, glsl:84-99.  (box-o)*inv is evaluated as fma(box, inv, -(o*inv)):
 * the canonical form shared with the HIP kernel. */
static inline int slab(const float bmin[3], const float bmax[3], const float inv[3], const float nro[3],
		float t_min, float t_max, float *out_tmin)
{
	float t0x = fmaf(bmin[0], inv[0], nro[0]), t1x = fmaf(bmax[0], inv[0], nro[0]);
	float t0y = fmaf(bmin[1], inv[1], nro[1]), t1y = fmaf(bmax[1], inv[1], nro[1]);
	float t0z = fmaf(bmin[2], inv[2], nro[2]), t1z = fmaf(bmax[2], inv[2], nro[2]);
	float tmin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), t_min));
	float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), t_max));
	*out_tmin = tmin;
	return tmin <= tmax;
}

/* ray_triangle, glsl:105-131 == Triangle::intersect, src/core/triangle.h:56-105.
 * The reference accepts t_min <= t < best_t, so of two triangles with exactly
 * equal t the one visited first wins, and its BVH2 / BVH4 / BVH8 / GLSL paths
 * visit in different orders.  Here (and in the HIP kernels) an exact tie goes to
 * the lower triangle id: best_id is the id of the current best hit, or
 * 0xFFFFFFFF with have_hit = 0 before any hit. */
static inline int tri_hit(const orc_tri64 *tr, const float o[3], const float d[3], float t_min, float best_t,
		int have_hit, uint32_t best_id, float *ot, float *ou, float *ov)
{
	float pvec[3], tvec[3], qvec[3];
	v_cross(d, tr->edge2, pvec);
	float det = v_dot(tr->edge1, pvec);
	if (fabsf(det) < 1e-8f) return 0;
	float inv_det = 1.0f / det;
	tvec[0] = o[0] - tr->v0[0]; tvec[1] = o[1] - tr->v0[1]; tvec[2] = o[2] - tr->v0[2];
	float u = v_dot(tvec, pvec) * inv_det;
	if (u < 0.0f || u > 1.0f) return 0;
	v_cross(tvec, tr->edge1, qvec);
	float v = v_dot(d, qvec) * inv_det;
	if (v < 0.0f || u + v > 1.0f) return 0;
	float t = v_dot(tr->edge2, qvec) * inv_det;
	if (t < t_min) return 0;
	if (!(t < best_t || (t == best_t && have_hit && tr->id < best_id))) return 0;
	*ot = t; *ou = u; *ov = v;
	return 1;
}

int orc_tri_test(const orc_tri64 *tri, const orc_ray32 *ray, float *t, float *u, float *v)
{
	return tri_hit(tri, ray->origin, ray->direction, ray->t_min, ray->t_max, 0, 0xFFFFFFFFu, t, u, v);
}

static inline void write_miss(orc_hit32 *h, float t)
{
	h->t = t; h->prim_id = -1; h->bary_u = 0.0f; h->bary_v = 0.0f;
	h->normal[0] = h->normal[1] = h->normal[2] = 0.0f; h->hit_layers = 0u;
}

typedef struct { uint32_t node; float tmin; } stack_ent;

static void trace_one(const orc_wide64 *wide, const orc_tri64 *lt, const orc_ray32 *ray, orc_hit32 *out,
		uint32_t mask, int any_hit, orc_counters *c)
{
	const float *o = ray->origin, *d = ray->direction;
	const float t_min = ray->t_min, t_max = ray->t_max;
	if (t_min >= t_max) { write_miss(out, t_max); return; } /* glsl:214-222 */
	float inv[3] = { safe_inv(d[0]), safe_inv(d[1]), safe_inv(d[2]) };
	float nro[3] = { -(o[0] * inv[0]), -(o[1] * inv[1]), -(o[2] * inv[2]) };
	float best_t = t_max, best_u = 0.0f, best_v = 0.0f;
	int32_t best_slot = -1;
	stack_ent stack[256];
	uint32_t sp = 0;
	stack[sp].node = 0; stack[sp].tmin = -1e30f; sp++; /* glsl:237-240 */
	while (sp > 0) {
		sp--;
		uint32_t ni = stack[sp].node;
		if (stack[sp].tmin > best_t) continue; /* glsl:251 */
		const orc_wide64 *n = &wide[ni];
		if (c) c->node_visits++;
		float tl, tr;
		int hl = slab(n->lmin, n->lmax, inv, nro, t_min, best_t, &tl) && tl <= best_t; /* glsl:268-274 */
		int hr = slab(n->rmin, n->rmax, inv, nro, t_min, best_t, &tr) && tr <= best_t;
		for (int side = 0; side < 2; side++) { /* INTERSECT_LEAF glsl:166-192, left then right */
			int h = side ? hr : hl;
			uint32_t cnt = side ? n->right_count : n->left_count, first = side ? n->right_idx : n->left_idx;
			if (!(h && cnt > 0)) continue;
			for (uint32_t k = 0; k < cnt; k++) {
				const orc_tri64 *tri = &lt[first + k];
				if ((tri->layers & mask) == 0u) continue;
				if (c) c->tri_tests++;
				float t, u, v;
				if (tri_hit(tri, o, d, t_min, best_t, best_slot >= 0, best_slot >= 0 ? lt[best_slot].id : 0xFFFFFFFFu, &t, &u, &v)) {
					best_t = t; best_u = u; best_v = v; best_slot = (int32_t)(first + k);
					if (any_hit) goto done;
				}
			}
		}
		int pl = hl && n->left_count == 0, pr = hr && n->right_count == 0;
		if (pl && pr) { /* far first: glsl:290-305 */
			if (tl < tr) {
				stack[sp].node = n->right_idx; stack[sp].tmin = tr; sp++;
				stack[sp].node = n->left_idx; stack[sp].tmin = tl; sp++;
			} else {
				stack[sp].node = n->left_idx; stack[sp].tmin = tl; sp++;
				stack[sp].node = n->right_idx; stack[sp].tmin = tr; sp++;
			}
		} else if (pl) { stack[sp].node = n->left_idx; stack[sp].tmin = tl; sp++; }
		else if (pr) { stack[sp].node = n->right_idx; stack[sp].tmin = tr; sp++; }
		if (c && sp > c->max_stack) c->max_stack = sp;
	}
done:
	if (best_slot >= 0) { /* glsl:322-327 */
		const orc_tri64 *tri = &lt[best_slot];
		out->t = best_t; out->prim_id = (int32_t)tri->id; out->bary_u = best_u; out->bary_v = best_v;
		out->normal[0] = tri->normal[0]; out->normal[1] = tri->normal[1]; out->normal[2] = tri->normal[2];
		out->hit_layers = tri->layers;
		if (c) c->hits++;
	} else write_miss(out, best_t);
	if (c) c->rays++;
}

void orc_trace(const orc_wide64 *wide, const orc_tri64 *leaf_tris, const orc_ray32 *rays, orc_hit32 *hits,
		uint64_t count, uint32_t query_mask, int any_hit, orc_counters *ctr, int n_threads)
{
	orc_counters total; memset(&total, 0, sizeof(total));
#ifdef _OPENMP
	if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
#endif
	{
		orc_counters local; memset(&local, 0, sizeof(local));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1024)
#endif
		for (int64_t i = 0; i < (int64_t)count; i++)
			trace_one(wide, leaf_tris, &rays[i], &hits[i], query_mask, any_hit, ctr ? &local : 0);
#ifdef _OPENMP
#pragma omp critical
#endif
		{
			total.rays += local.rays; total.hits += local.hits;
			total.node_visits += local.node_visits; total.tri_tests += local.tri_tests;
			if (local.max_stack > total.max_stack) total.max_stack = local.max_stack;
		}
	}
	if (ctr) *ctr = total;
	(void)n_threads;
}

/* src/accel/ray_scene.h:120-131 (nearest), :151-162 (any) */
void orc_trace_brute(const orc_tri64 *tris, uint32_t n_tris, const orc_ray32 *rays, orc_hit32 *hits,
		uint64_t count, uint32_t query_mask, int any_hit, int n_threads)
{
#ifdef _OPENMP
	if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 64)
#endif
	for (int64_t i = 0; i < (int64_t)count; i++) {
		const orc_ray32 *r = &rays[i];
		if (r->t_min >= r->t_max) { write_miss(&hits[i], r->t_max); continue; }
		float best_t = r->t_max, bu = 0, bv = 0; int64_t best = -1;
		for (uint32_t k = 0; k < n_tris; k++) {
			if ((tris[k].layers & query_mask) == 0u) continue;
			float t, u, v;
			if (tri_hit(&tris[k], r->origin, r->direction, r->t_min, best_t, best >= 0, best >= 0 ? tris[best].id : 0xFFFFFFFFu, &t, &u, &v)) {
				best_t = t; bu = u; bv = v; best = k;
				if (any_hit) break;
			}
		}
		if (best >= 0) {
			const orc_tri64 *tri = &tris[best];
			hits[i].t = best_t; hits[i].prim_id = (int32_t)tri->id; hits[i].bary_u = bu; hits[i].bary_v = bv;
			hits[i].normal[0] = tri->normal[0]; hits[i].normal[1] = tri->normal[1]; hits[i].normal[2] = tri->normal[2];
			hits[i].hit_layers = tri->layers;
		} else write_miss(&hits[i], best_t);
	}
	(void)n_threads;
}

/* ------------------------------------------------------------------------- */
/* primary-ray grid: src/godot/raytracer_debug.cpp:572-596                     */
/* ------------------------------------------------------------------------- */
void orc_camera_basis(const float forward[3], uint32_t w, uint32_t h, float fov_deg,
		float fwd[3], float right[3], float up[3], float *half_w, float *half_h)
{
	fwd[0] = forward[0]; fwd[1] = forward[1]; fwd[2] = forward[2];
	v_normalize(fwd);
	float hint[3] = { 0.0f, 1.0f, 0.0f };
	float dp = fwd[0] * hint[0] + fwd[1] * hint[1] + fwd[2] * hint[2];
	if (fabsf(dp) > 0.99f) { hint[0] = 1.0f; hint[1] = 0.0f; hint[2] = 0.0f; }
	/* Vector3::cross: plain products */
	right[0] = fwd[1] * hint[2] - fwd[2] * hint[1];
	right[1] = fwd[2] * hint[0] - fwd[0] * hint[2];
	right[2] = fwd[0] * hint[1] - fwd[1] * hint[0];
	v_normalize(right);
	up[0] = right[1] * fwd[2] - right[2] * fwd[1];
	up[1] = right[2] * fwd[0] - right[0] * fwd[2];
	up[2] = right[0] * fwd[1] - right[1] * fwd[0];
	v_normalize(up);
	/* Math::deg_to_rad(float): p * (float)(Math_PI / 180.0) */
	float half_fov_rad = (fov_deg * 0.5f) * (float)(3.14159265358979323846 / 180.0);
	*half_w = tanf(half_fov_rad);
	*half_h = *half_w * ((float)h / (float)w);
}

void orc_grid_rays(const float origin[3], const float forward[3], uint32_t w, uint32_t h, float fov_deg,
		uint32_t y0, uint32_t y1, orc_ray32 *out)
{
	float fwd[3], right[3], up[3], half_w, half_h;
	orc_camera_basis(forward, w, h, fov_deg, fwd, right, up, &half_w, &half_h);
	for (uint32_t y = y0; y < y1; y++) for (uint32_t x = 0; x < w; x++) {
		float u = (2.0f * ((float)x + 0.5f) / (float)w - 1.0f) * half_w;
		float v = (2.0f * ((float)y + 0.5f) / (float)h - 1.0f) * half_h;
		float dir[3] = { fwd[0] + right[0] * u + up[0] * v, fwd[1] + right[1] * u + up[1] * v, fwd[2] + right[2] * u + up[2] * v };
		v_normalize(dir);
		orc_ray32 *r = &out[(size_t)(y - y0) * w + x];
		r->origin[0] = origin[0]; r->origin[1] = origin[1]; r->origin[2] = origin[2];
		r->direction[0] = dir[0]; r->direction[1] = dir[1]; r->direction[2] = dir[2];
		r->t_min = 0.001f; r->t_max = FLT_MAX; /* Ray(origin, dir): src/core/ray.h:59 */
	}
}

/* ------------------------------------------------------------------------- */
/* RayCamera grids: src/modules/graphics/ray_camera.h                          */
/*   setup :50-76 (inv_w, inv_h), _setup_perspective :208-218 (Math_PI is a    */
/*   double: tangent in double, rounded once), _setup_orthographic :220-230,   */
/*   _generate_perspective :234-251, _generate_orthographic :255-273,          */
/*   generate_ray_jittered :106-122 (jx = jy = 0.5 is the pixel centre).       */
/* basis: row-major 3x3 = Godot's Basis rows; Basis::xform(v) = one dot per    */
/* row, x*v.x + y*v.y + z*v.z summed left to right.  ortho != 0: `param` is    */
/* Camera3D::size, else the vertical field of view in degrees.                 */
/* ------------------------------------------------------------------------- */
void orc_ray_camera_rays(const float origin[3], const float basis[9], uint32_t w, uint32_t h, float param, int ortho,
		float jx, float jy, uint32_t y0, uint32_t y1, orc_ray32 *out)
{
	const float inv_w = 1.0f / (float)w, inv_h = 1.0f / (float)h;
	const float aspect = (float)w / (float)h;
	float half_w, half_h;
	if (ortho) { half_h = param * 0.5f; half_w = half_h * aspect; }
	else {
		const float tan_half = (float)tan((double)(param * 0.5f) * (3.1415926535897932384626433833 / 180.0));
		half_w = tan_half * aspect; half_h = tan_half;
	}
	const float col0[3] = { basis[0], basis[3], basis[6] }, col1[3] = { basis[1], basis[4], basis[7] };
	const float fwd[3] = { -basis[2], -basis[5], -basis[8] }; /* forward_ = -basis_.get_column(2) */
	for (uint32_t y = y0; y < y1; y++) {
		const float v = 1.0f - (2.0f * ((float)y + jy) * inv_h);
		for (uint32_t x = 0; x < w; x++) {
			const float u = (2.0f * ((float)x + jx) * inv_w) - 1.0f;
			orc_ray32 *r = &out[(size_t)(y - y0) * w + x];
			if (!ortho) {
				const float view[3] = { u * half_w, v * half_h, -1.0f };
				float dir[3];
				for (int k = 0; k < 3; k++) dir[k] = basis[3 * k] * view[0] + basis[3 * k + 1] * view[1] + basis[3 * k + 2] * view[2];
				v_normalize(dir);
				for (int k = 0; k < 3; k++) { r->origin[k] = origin[k]; r->direction[k] = dir[k]; }
			} else {
				const float sv = v * half_h, su = u * half_w;
				for (int k = 0; k < 3; k++) {
					const float row_offset = origin[k] + col1[k] * sv; /* origin_ + up * (v * ortho_half_h_) */
					r->origin[k] = row_offset + col0[k] * su;
					r->direction[k] = fwd[k];
				}
			}
			r->t_min = 0.001f; r->t_max = FLT_MAX;
		}
	}
}

/* ------------------------------------------------------------------------- */
/* Morton key: src/dispatch/ray_sort.h:41-76                                   */
/* ------------------------------------------------------------------------- */
static inline uint32_t spread10(uint32_t v)
{
	v &= 0x000003FFu;
	v = (v | (v << 16)) & 0x030000FFu;
	v = (v | (v << 8)) & 0x0300F00Fu;
	v = (v | (v << 4)) & 0x030C30C3u;
	v = (v | (v << 2)) & 0x09249249u;
	return v;
}
static inline uint32_t quant10(float v)
{
	float n = (v + 1.0f) * 0.5f;
	n = fmaxf(0.0f, fminf(1.0f, n));
	return (uint32_t)(n * 1023.0f);
}
uint32_t orc_morton_key(const float dir[3])
{
	return (spread10(quant10(dir[0])) << 2) | (spread10(quant10(dir[1])) << 1) | spread10(quant10(dir[2]));
}
void orc_morton_keys(const orc_ray32 *rays, uint64_t count, uint32_t *keys)
{
	for (uint64_t i = 0; i < count; i++) keys[i] = orc_morton_key(rays[i].direction);
}

/* ------------------------------------------------------------------------- */
/* host <-> packed conversions                                                */
/* ------------------------------------------------------------------------- */
/* src/gpu/gpu_ray_caster.cpp:639-650 */
void orc_pack_rays(const orc_host_ray60 *rays, uint64_t count, orc_ray32 *out)
{
	for (uint64_t i = 0; i < count; i++) {
		for (int k = 0; k < 3; k++) { out[i].origin[k] = rays[i].origin[k]; out[i].direction[k] = rays[i].direction[k]; }
		out[i].t_max = rays[i].t_max; out[i].t_min = rays[i].t_min;
	}
}
/* Ray(o, d, t0, t1) + _precompute: src/core/ray.h:53-96 */
void orc_make_host_rays(const orc_ray32 *rays, uint64_t count, orc_host_ray60 *out)
{
	const float eps = 1e-9f;
	for (uint64_t i = 0; i < count; i++) {
		for (int k = 0; k < 3; k++) {
			float d = rays[i].direction[k];
			out[i].origin[k] = rays[i].origin[k]; out[i].direction[k] = d;
			out[i].inv_direction[k] = (fabsf(d) < eps) ? ((d < 0.0f) ? (-1.0f / eps) : (1.0f / eps)) : (1.0f / d);
			out[i].dir_sign[k] = (d < 0.0f) ? 1 : 0;
		}
		out[i].t_min = rays[i].t_min; out[i].t_max = rays[i].t_max; out[i].flags = 0;
	}
}
/* src/gpu/gpu_ray_caster.cpp:442-456; results[] start default-constructed
 * (src/core/intersection.h:44-45), so a miss keeps position = 0. */
void orc_unpack_hits(const orc_hit32 *hits, const orc_host_ray60 *rays, uint64_t count, orc_host_hit44 *out)
{
	for (uint64_t i = 0; i < count; i++) {
		const orc_hit32 *g = &hits[i];
		orc_host_hit44 *h = &out[i];
		h->t = g->t; h->u = g->bary_u; h->v = g->bary_v;
		h->normal[0] = g->normal[0]; h->normal[1] = g->normal[1]; h->normal[2] = g->normal[2];
		if (g->prim_id >= 0) {
			h->prim_id = (uint32_t)g->prim_id; h->hit_layers = g->hit_layers;
			for (int k = 0; k < 3; k++) h->position[k] = rays[i].origin[k] + rays[i].direction[k] * g->t;
		} else {
			h->t = FLT_MAX; h->u = 0.0f; h->v = 0.0f; h->prim_id = 0xFFFFFFFFu; h->hit_layers = 0u;
			h->position[0] = h->position[1] = h->position[2] = 0.0f;
		}
	}
}

/* ---- scene flatten: RayTracerServer::_rebuild_scene, src/godot/raytracer_server.cpp:700-711 ----
 * For every instance in order, for every triangle of its mesh: a, b, c = inst.transform.xform(v0, v1, v2),
 * Triangle(a, b, c, tri_offset++, layer mask of the mesh).  Transform3D::xform(v) is
 * (basis.rows[k].dot(v) + origin[k]) for k = 0..2 and Vector3::dot is x*x' + y*y' + z*z' summed left to
 * right (godot-cpp is not vendored in the reference tree; the reference's own a*b+c style, no fma). */
void orc_flatten_instances(const float *verts9, const orc_instance *inst, uint32_t n_inst, orc_tri64 *out)
{
	uint32_t tri_offset = 0;
	for (uint32_t i = 0; i < n_inst; i++) {
		const orc_instance *in = &inst[i];
		for (uint32_t k = 0; k < in->n_tris; k++) {
			const float *src = verts9 + 9 * (size_t)(in->first_tri + k);
			float w[9];
			for (int v = 0; v < 3; v++)
				for (int r = 0; r < 3; r++)
					w[3 * v + r] = ((in->basis[3 * r] * src[3 * v] + in->basis[3 * r + 1] * src[3 * v + 1]) +
							in->basis[3 * r + 2] * src[3 * v + 2]) + in->origin[r];
			const uint32_t id = tri_offset++;
			orc_make_triangles(w, &id, &in->layers, 1, &out[id]);
		}
	}
}

/* ------------------------------------------------------------------------- */
/* Two-level scene: SceneTLAS::build_tlas / cast_ray / any_hit                   */
/* (src/accel/scene_tlas.h:140-251), MeshBLAS::build (mesh_blas.h:86-138),       */
/* BLASInstance (blas_instance.h:47-107) and tinybvh::BVH::IntersectTLAS         */
/* (thirdparty/tinybvh/tiny_bvh.h:3306-3380): a BVH per distinct mesh in mesh    */
/* space, a BVH over the instances' world boxes; per TLAS leaf the ray is taken  */
/* to mesh space by the inverse transform WITHOUT renormalising the direction    */
/* (blas_instance.h:56-66, tiny_bvh.h:3327-3331), so t, t_min and the best hit   */
/* stay world-parameterised.  Box and triangle tests are the ones above.         */
/* Deliberate differences from the reference, shared with the HIP path:          */
/*  - prim_id is the FLAT id of raytracer_server.cpp:700-711 (running triangle   */
/*    offset of the instance + mesh-local index); SceneTLAS reports the local    */
/*    index, which its callers cannot resolve;      */
/*  - hit_layers is the instance's mask (raytracer_server.cpp:702-703) and whole */
/*    instances are skipped by the query mask (tiny_bvh.h:3323-3324);            */
/*  - an exact tie in t goes to the lower flat id (as in trace_one);             */
/*  - the inverse is taken in double and rounded once; a world box is the image  */
/*    of the mesh box's corners (blas_instance.h:76-107) in double, rounded      */
/*    outwards.                                                                  */
/* ------------------------------------------------------------------------- */
typedef struct { orc_wide64 *wide; orc_tri64 *lt; uint32_t n_wide, n_tris, first_tri; float lo[3], hi[3]; } tl_blas;
typedef struct { float inv[12]; float basis[9]; uint32_t blas, id_base, layers; } tl_inst;
struct orc_two_level {
	tl_blas *blas; uint32_t n_blas;
	tl_inst *inst; uint32_t n_inst;
	orc_wide64 *tlas; orc_tri64 *tlas_lt; uint32_t n_tlas; /* tlas_lt[k].id = instance in leaf slot k */
};

static int build_wide(const float *verts9, uint32_t n, orc_wide64 **wide, uint32_t *n_wide, orc_tri64 **lt)
{
	orc_tri64 *tris = (orc_tri64 *)malloc((size_t)n * sizeof(orc_tri64));
	float *v4 = (float *)malloc((size_t)n * 12 * sizeof(float));
	orc_node32 *nodes = (orc_node32 *)malloc((size_t)2 * n * sizeof(orc_node32));
	uint32_t *prim = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
	*wide = (orc_wide64 *)malloc((size_t)2 * n * sizeof(orc_wide64));
	*lt = (orc_tri64 *)malloc((size_t)n * sizeof(orc_tri64));
	int rc = 1;
	if (tris && v4 && nodes && prim && *wide && *lt) {
		orc_make_triangles(verts9, 0, 0, n, tris);
		for (size_t t = 0; t < (size_t)n * 3; t++) { v4[4 * t] = verts9[3 * t]; v4[4 * t + 1] = verts9[3 * t + 1]; v4[4 * t + 2] = verts9[3 * t + 2]; v4[4 * t + 3] = 0.0f; }
		uint32_t used = 0;
		rc = orc_bvh2_build(v4, n, nodes, prim, &used);
		if (!rc) rc = orc_to_wide(tris, n, nodes, used, prim, *wide, n_wide, *lt);
	}
	free(tris); free(v4); free(nodes); free(prim);
	return rc;
}

void orc_two_level_free(orc_two_level *s)
{
	if (!s) return;
	for (uint32_t k = 0; k < s->n_blas; k++) { free(s->blas[k].wide); free(s->blas[k].lt); }
	free(s->blas); free(s->inst); free(s->tlas); free(s->tlas_lt); free(s);
}

orc_two_level *orc_two_level_build(const float *verts9, uint32_t n_mesh_tris, const orc_instance *inst, uint32_t n_inst)
{
	if (!verts9 || !inst || n_inst == 0) return 0;
	orc_two_level *s = (orc_two_level *)calloc(1, sizeof(*s));
	if (!s) return 0;
	s->blas = (tl_blas *)calloc(n_inst, sizeof(tl_blas));
	s->inst = (tl_inst *)calloc(n_inst, sizeof(tl_inst));
	float *boxes = (float *)malloc((size_t)n_inst * 9 * sizeof(float));
	if (!s->blas || !s->inst || !boxes) { free(boxes); orc_two_level_free(s); return 0; }
	s->n_inst = n_inst;
	uint32_t id_base = 0;
	for (uint32_t i = 0; i < n_inst; i++) {
		const orc_instance *in = &inst[i];
		if (in->n_tris == 0 || in->first_tri >= n_mesh_tris || in->n_tris > n_mesh_tris - in->first_tri) { free(boxes); orc_two_level_free(s); return 0; }
		uint32_t b = 0;
		while (b < s->n_blas && !(s->blas[b].first_tri == in->first_tri && s->blas[b].n_tris == in->n_tris)) b++;
		if (b == s->n_blas) { /* MeshBLAS::build, mesh_blas.h:86-138 */
			tl_blas *nb = &s->blas[s->n_blas];
			nb->first_tri = in->first_tri; nb->n_tris = in->n_tris;
			if (build_wide(verts9 + 9 * (size_t)in->first_tri, in->n_tris, &nb->wide, &nb->n_wide, &nb->lt)) { s->n_blas++; free(boxes); orc_two_level_free(s); return 0; }
			for (int c = 0; c < 3; c++) { /* mesh box = union of the root's child boxes */
				nb->lo[c] = fminf(nb->wide[0].lmin[c], nb->wide[0].rmin[c]);
				nb->hi[c] = fmaxf(nb->wide[0].lmax[c], nb->wide[0].rmax[c]);
			}
			s->n_blas++;
		}
		tl_inst *d = &s->inst[i];
		d->blas = b; d->id_base = id_base; d->layers = in->layers;
		id_base += in->n_tris;
		/* Transform3D::affine_inverse (blas_instance.h:43-45), cofactors in double */
		const double a = in->basis[0], bb = in->basis[1], c = in->basis[2], dd = in->basis[3], e = in->basis[4], f = in->basis[5],
				g = in->basis[6], h = in->basis[7], ii = in->basis[8];
		const double c00 = e * ii - f * h, c01 = c * h - bb * ii, c02 = bb * f - c * e;
		const double c10 = f * g - dd * ii, c11 = a * ii - c * g, c12 = c * dd - a * f;
		const double c20 = dd * h - e * g, c21 = bb * g - a * h, c22 = a * e - bb * dd;
		const double det = a * c00 + bb * c10 + c * c20;
		if (!(fabs(det) > 0.0)) { free(boxes); orc_two_level_free(s); return 0; }
		const double m[9] = { c00 / det, c01 / det, c02 / det, c10 / det, c11 / det, c12 / det, c20 / det, c21 / det, c22 / det };
		for (int r = 0; r < 3; r++) {
			const double t = -(m[3 * r] * (double)in->origin[0] + m[3 * r + 1] * (double)in->origin[1] + m[3 * r + 2] * (double)in->origin[2]);
			d->inv[4 * r] = (float)m[3 * r]; d->inv[4 * r + 1] = (float)m[3 * r + 1]; d->inv[4 * r + 2] = (float)m[3 * r + 2]; d->inv[4 * r + 3] = (float)t;
		}
		for (int k = 0; k < 9; k++) d->basis[k] = in->basis[k];
		/* BLASInstance::compute_world_bounds, blas_instance.h:76-107 (double, rounded outwards) */
		const tl_blas *mb = &s->blas[b];
		double mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
		for (int k = 0; k < 8; k++) {
			const double x = (k & 1) ? mb->hi[0] : mb->lo[0], y = (k & 2) ? mb->hi[1] : mb->lo[1], z = (k & 4) ? mb->hi[2] : mb->lo[2];
			for (int r = 0; r < 3; r++) {
				const double w = ((double)in->basis[3 * r] * x + (double)in->basis[3 * r + 1] * y) + (double)in->basis[3 * r + 2] * z + (double)in->origin[r];
				if (w < mn[r]) mn[r] = w;
				if (w > mx[r]) mx[r] = w;
			}
		}
		for (int r = 0; r < 3; r++) { /* proxy triangle {lo, hi, centre}: its AABB is the box */
			float l = (float)mn[r], u = (float)mx[r];
			if ((double)l > mn[r]) l = nextafterf(l, -INFINITY);
			if ((double)u < mx[r]) u = nextafterf(u, INFINITY);
			boxes[9 * (size_t)i + r] = l; boxes[9 * (size_t)i + 3 + r] = u; boxes[9 * (size_t)i + 6 + r] = 0.5f * l + 0.5f * u;
		}
	}
	/* SceneTLAS::build_tlas, scene_tlas.h:140-176: the same builder over the instance boxes */
	const int rc = build_wide(boxes, n_inst, &s->tlas, &s->n_tlas, &s->tlas_lt);
	free(boxes);
	if (rc) { orc_two_level_free(s); return 0; }
	return s;
}

/* BVH::Intersect of one BLAS with the mesh-space ray: trace_one's loop on shared best-hit state */
static int walk_blas(const tl_blas *b, const float o[3], const float d[3], float t_min, uint32_t id_base, int any_hit,
		float *best_t, float *best_u, float *best_v, int64_t *best_slot, uint32_t *best_id)
{
	const float inv[3] = { safe_inv(d[0]), safe_inv(d[1]), safe_inv(d[2]) };
	const float nro[3] = { -(o[0] * inv[0]), -(o[1] * inv[1]), -(o[2] * inv[2]) };
	stack_ent stack[256];
	uint32_t sp = 0;
	int found = 0;
	stack[sp].node = 0; stack[sp].tmin = -1e30f; sp++;
	while (sp > 0) {
		sp--;
		const uint32_t ni = stack[sp].node;
		if (stack[sp].tmin > *best_t) continue;
		const orc_wide64 *n = &b->wide[ni];
		float tl, tr;
		int hl = slab(n->lmin, n->lmax, inv, nro, t_min, *best_t, &tl) && tl <= *best_t;
		int hr = slab(n->rmin, n->rmax, inv, nro, t_min, *best_t, &tr) && tr <= *best_t;
		for (int side = 0; side < 2; side++) {
			const int h = side ? hr : hl;
			const uint32_t cnt = side ? n->right_count : n->left_count, first = side ? n->right_idx : n->left_idx;
			if (!(h && cnt > 0)) continue;
			for (uint32_t k = 0; k < cnt; k++) {
				orc_tri64 tri = b->lt[first + k];
				tri.id += id_base; /* flat id: the tie rule compares these */
				float t, u, v;
				if (tri_hit(&tri, o, d, t_min, *best_t, *best_slot >= 0, *best_id, &t, &u, &v)) {
					*best_t = t; *best_u = u; *best_v = v; *best_slot = (int64_t)(first + k); *best_id = tri.id;
					found = 1;
					if (any_hit) return 1;
				}
			}
		}
		const int pl = hl && n->left_count == 0, pr = hr && n->right_count == 0;
		if (pl && pr) {
			if (tl < tr) {
				stack[sp].node = n->right_idx; stack[sp].tmin = tr; sp++;
				stack[sp].node = n->left_idx; stack[sp].tmin = tl; sp++;
			} else {
				stack[sp].node = n->left_idx; stack[sp].tmin = tl; sp++;
				stack[sp].node = n->right_idx; stack[sp].tmin = tr; sp++;
			}
		} else if (pl) { stack[sp].node = n->left_idx; stack[sp].tmin = tl; sp++; }
		else if (pr) { stack[sp].node = n->right_idx; stack[sp].tmin = tr; sp++; }
	}
	return found;
}

static void two_level_one(const orc_two_level *s, const orc_ray32 *ray, orc_hit32 *out, uint32_t mask, int any_hit)
{
	const float *o = ray->origin, *d = ray->direction;
	const float t_min = ray->t_min, t_max = ray->t_max;
	if (t_min >= t_max) { write_miss(out, t_max); return; }
	const float inv[3] = { safe_inv(d[0]), safe_inv(d[1]), safe_inv(d[2]) };
	const float nro[3] = { -(o[0] * inv[0]), -(o[1] * inv[1]), -(o[2] * inv[2]) };
	float best_t = t_max, best_u = 0.0f, best_v = 0.0f;
	int64_t best_slot = -1; uint32_t best_id = 0xFFFFFFFFu, best_inst = 0;
	stack_ent stack[256];
	uint32_t sp = 0;
	stack[sp].node = 0; stack[sp].tmin = -1e30f; sp++;
	while (sp > 0) { /* tiny_bvh.h:3314-3378 */
		sp--;
		const uint32_t ni = stack[sp].node;
		if (stack[sp].tmin > best_t) continue;
		const orc_wide64 *n = &s->tlas[ni];
		float tl, tr;
		int hl = slab(n->lmin, n->lmax, inv, nro, t_min, best_t, &tl) && tl <= best_t;
		int hr = slab(n->rmin, n->rmax, inv, nro, t_min, best_t, &tr) && tr <= best_t;
		for (int side = 0; side < 2; side++) {
			const int h = side ? hr : hl;
			const uint32_t cnt = side ? n->right_count : n->left_count, first = side ? n->right_idx : n->left_idx;
			if (!(h && cnt > 0)) continue;
			for (uint32_t k = 0; k < cnt; k++) { /* tiny_bvh.h:3320-3360 */
				const uint32_t ii = s->tlas_lt[first + k].id;
				const tl_inst *in = &s->inst[ii];
				if ((in->layers & mask) == 0u) continue;
				const float *m = in->inv;
				const float oo[3] = { fmaf(m[0], o[0], fmaf(m[1], o[1], fmaf(m[2], o[2], m[3]))),
						fmaf(m[4], o[0], fmaf(m[5], o[1], fmaf(m[6], o[2], m[7]))),
						fmaf(m[8], o[0], fmaf(m[9], o[1], fmaf(m[10], o[2], m[11]))) };
				const float od[3] = { fmaf(m[0], d[0], fmaf(m[1], d[1], m[2] * d[2])),
						fmaf(m[4], d[0], fmaf(m[5], d[1], m[6] * d[2])),
						fmaf(m[8], d[0], fmaf(m[9], d[1], m[10] * d[2])) };
				const float before_t = best_t; const uint32_t before_id = best_id;
				if (walk_blas(&s->blas[in->blas], oo, od, t_min, in->id_base, any_hit, &best_t, &best_u, &best_v, &best_slot, &best_id)) {
					if (best_t != before_t || best_id != before_id) best_inst = ii;
					if (any_hit) goto done;
				}
			}
		}
		const int pl = hl && n->left_count == 0, pr = hr && n->right_count == 0;
		if (pl && pr) {
			if (tl < tr) {
				stack[sp].node = n->right_idx; stack[sp].tmin = tr; sp++;
				stack[sp].node = n->left_idx; stack[sp].tmin = tl; sp++;
			} else {
				stack[sp].node = n->left_idx; stack[sp].tmin = tl; sp++;
				stack[sp].node = n->right_idx; stack[sp].tmin = tr; sp++;
			}
		} else if (pl) { stack[sp].node = n->left_idx; stack[sp].tmin = tl; sp++; }
		else if (pr) { stack[sp].node = n->right_idx; stack[sp].tmin = tr; sp++; }
	}
done:
	if (best_slot >= 0) { /* scene_tlas.h:217-244 */
		const tl_inst *in = &s->inst[best_inst];
		const orc_tri64 *tri = &s->blas[in->blas].lt[best_slot];
		const float *b = in->basis, *no = tri->normal;
		float nx = fmaf(b[0], no[0], fmaf(b[1], no[1], b[2] * no[2]));
		float ny = fmaf(b[3], no[0], fmaf(b[4], no[1], b[5] * no[2]));
		float nz = fmaf(b[6], no[0], fmaf(b[7], no[1], b[8] * no[2]));
		const float l2 = fmaf(nx, nx, fmaf(ny, ny, nz * nz));
		if (l2 == 0.0f) { nx = ny = nz = 0.0f; }
		else { const float l = sqrtf(l2); nx /= l; ny /= l; nz /= l; }
		out->t = best_t; out->prim_id = (int32_t)best_id; out->bary_u = best_u; out->bary_v = best_v;
		out->normal[0] = nx; out->normal[1] = ny; out->normal[2] = nz;
		out->hit_layers = in->layers;
	} else write_miss(out, best_t);
}

void orc_two_level_trace(const orc_two_level *s, const orc_ray32 *rays, orc_hit32 *hits, uint64_t count,
		uint32_t query_mask, int any_hit, int n_threads)
{
#ifdef _OPENMP
	if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 256)
#endif
	for (int64_t i = 0; i < (int64_t)count; i++) two_level_one(s, &rays[i], &hits[i], query_mask, any_hit);
	(void)n_threads;
}
