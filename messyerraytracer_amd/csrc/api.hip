// api.hip — the C-ABI of include/mrt_hip.h: context, scene upload, casts.
//
// Mirrors GPURayCaster (src/gpu/gpu_ray_caster.{h,cpp}) and the GPU half of
// RayDispatcher's policy (src/dispatch/ray_dispatcher.h:124-356): grow-only
// per-dispatch buffers (cpp:776-817), one pending async dispatch (cpp:538),
// upload drains a pending dispatch (cpp:198-202), Morton sort for incoherent
// batches of >= 256 rays (ray_dispatcher.h:135,427) — the sort, the gather and
// the unshuffle all run on the device (radix sort + permuted load/store inside
// the trace kernel) instead of std::sort + three host copies.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <atomic>
#include <new>
#include <thread>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "mrt_internal.h"

namespace mrt {
hipError_t launch_trace(const TraceParams &p, bool any_hit, bool count, hipStream_t stream);
const char *last_trace_variant();
bool quad_kernel_built();
hipError_t launch_grid_rays(const TraceParams &p, mrt_ray32 *out, hipStream_t stream);
hipError_t launch_build_rows4(const Dev4Node *nodes4, const TriHot *hot, const TriCold *cold, uint32_t n_nodes4, uint32_t n_tris,
		void *rows, hipStream_t stream);
hipError_t launch_build_rows(const DevNode *nodes, const TriHot *hot, const TriCold *cold, uint32_t n_nodes, uint32_t n_tris,
		void *rows, hipStream_t stream);
hipError_t launch_expand_tokens(const TraceParams &p, const uint32_t *tokens, hipStream_t stream);
hipError_t launch_offset_refs(DevNode *dst, const DevNode *src, uint32_t n, uint32_t node_base, uint32_t tri_base, void *stream);
hipError_t launch_offset_refs8(Dev8Node *dst, const Dev8Node *src, uint32_t n, uint32_t node_base, uint32_t tri_base, void *stream);
hipError_t launch_flatten_instances(const float *d_verts9, const mrt_instance *d_instances, const uint32_t *d_first_out,
		uint32_t n_instances, uint32_t max_tris_per_instance, mrt_tri64 *d_out, void *stream);
hipError_t launch_morton_keys(const void *rays, uint32_t in_fmt, uint64_t count, uint32_t *keys, uint32_t *index, hipStream_t stream);
hipError_t launch_trace_persistent(const TraceParams &p, unsigned long long *next_ray, uint32_t *overflow,
		uint32_t lds_depth, uint32_t refill, uint32_t leaf_wait, uint32_t blocks, bool any_hit, bool count, hipStream_t stream);
hipError_t launch_origin_dir_keys(const void *rays, uint32_t in_fmt, uint64_t count, const float lo[3], const float hi[3],
		uint32_t *keys, uint32_t *index, hipStream_t stream);
hipError_t launch_detect_grid(const void *rays, uint32_t in_fmt, uint64_t count, uint32_t tile_w_log2,
		unsigned long long *scratch, uint32_t *out, uint32_t *host_out, hipStream_t stream);
}

struct DevBuf {
	void *ptr = nullptr;
	size_t cap = 0;
};

struct mrt_ctx {
	int device = 0;
	mrt_options opts{};
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;
	hipEvent_t ev[6] = {};
	char err[512] = {0};
	// scene
	mrt::DevNode *d_nodes = nullptr; mrt::TriHot *d_hot = nullptr; mrt::TriCold *d_cold = nullptr;
	mrt::Dev4Node *d_nodes4 = nullptr; uint32_t n_nodes4 = 0;
	mrt::Dev8Node *d_nodes8 = nullptr; uint32_t n_nodes8 = 0, stack8 = 0;
	float *d_leaf_box = nullptr; // exact leaf boxes that go with d_nodes8
	mrt::BuildArena build_arena; // temporaries of the device builder, kept between builds
	void *d_rows4 = nullptr;     // flat scenes with the 4-wide layout: 128-byte node rows + triangle rows (packet_quad_kernel.h)
	void *d_rows = nullptr;      // flat scenes: nodes + triangles as one array of 64-byte rows (packet_rows_kernel.h)
	// two-level scene: d_nodes = TLAS + every BLAS, d_hot / d_cold = mesh-space triangles, d_instances in TLAS leaf order
	mrt::DevInstance *d_instances = nullptr;
	mrt::TwoLevelHost *two_level = nullptr; // host copy kept for mrt_update_instances
	float bounds_lo[3] = {0, 0, 0}, bounds_hi[3] = {0, 0, 0};
	uint32_t n_nodes = 0, n_tris = 0, depth = 0, stack_depth = 0, stack4 = 0;
	bool scene = false;
	// per-dispatch buffers (grow only, x1.5: gpu_ray_caster.cpp:776-817)
	DevBuf rays, hits, keys_in, keys_out, idx_in, idx_out, sort_tmp, overflow;
	int cu_count = 256;
	unsigned long long *d_counters = nullptr;
	// what detect_grid_kernel decided, also written to this host-mapped word block {row width, rows, tiles_x, verdict}
	// so that the host knows after the stream sync which of the two queued kernels did the work (no extra copy)
	uint32_t *h_auto = nullptr, *d_auto_host = nullptr;
	// small host-array casts (RayDispatcher::cast_ray / any_hit: one ray; tiles of a few hundred rays): rays and hits go
	// through two pinned, device-mapped buffers instead of two DMA copies (mrt_cast)
	void *h_small_in = nullptr, *d_small_in = nullptr, *h_small_out = nullptr, *d_small_out = nullptr;
	// Frame-coherent tile schedule of grid casts: what every schedule unit (one 8x8 tile, or the two of a 128-ray wave)
	// cost in the last cast of this grid, and the launch order made of it (longest first); see schedule_grid().
	struct TileSchedule {
		uint32_t grid_w = 0, grid_h = 0, y0 = 0, rows = 0, unit = 0, n_units = 0, tile_w_log2 = 0; bool pieces = false;
		// two generations: frame f notes its costs in cost[f & 1] and the side stream sorts them into order[f & 1] while frame
		// f + 1 (launched in the order of frame f - 1) already runs: back-to-back frames never wait for a sort
		uint32_t frame = 0, gen = 0;       // frames of this grid, sorts issued for it
		bool measuring = false;            // this frame notes its costs (and is sorted afterwards)
		bool have_order[2] = {false, false};
		DevBuf cost[2], order[2], cost_sorted, iota, tmp;
		DevBuf slots[2], hdr[2];           // what is launched: the order with its most expensive units in pieces (schedule_split)
		uint32_t n_slots_max = 0;
		hipStream_t side = nullptr;
		hipEvent_t traced = nullptr, ready[2] = {nullptr, nullptr};
		void forget() { have_order[0] = have_order[1] = false; }
	};
	// How a mid-size grid is cast is MEASURED per grid (tune_grid_kernel): four frames with the 64-ray kernel, four with the 128-ray
	// walk and its most expensive units launched in pieces, four with the 128-ray walk and every unit whole; from frame 12 on the
	// fastest of the three, each judged by the faster of its last two frames.  What wins flips with the number of rounds a grid
	// makes on the chip (C3 scene: 1280x720 the 64-ray kernel, 1280x960 the 128-ray walk whole, 1920x1080 the 128-ray walk in pieces).
	uint64_t last_detect_count = 0;   // rays of the last cast whose row width was looked for on the device (h_auto holds what it found)
	struct GridTune { uint32_t grid_w = 0, grid_h = 0, y0 = 0, rows = 0; int mode = -1; int phase = 0; float t_asm = 0.0f, t_dual = 0.0f, t_whole = 0.0f; bool armed = false, no_pieces = false; };
	// What has been learnt about a grid (its tile schedule, how it is cast fastest) is kept per grid AND cast mode, for the last few
	// of them: a renderer that casts two views, or closest-hit and any-hit rays of one view, or the row-block chunks of a sharded
	// frame, in turn, keeps every one's state (with one state each change of grid threw the other's away, and cost a stream
	// synchronisation and an upload to start over).  select_grid_state() picks the entry of a cast; the least recently used one goes.
	struct GridState { TileSchedule sched; GridTune tune; uint32_t k_w = 0, k_h = 0, k_y0 = 0, k_rows = 0; int k_mode = -1; uint64_t stamp = 0; };
	static constexpr int kGridStates = 8;
	GridState grid_states[kGridStates];
	GridState *gs = &grid_states[0];
	uint64_t gs_clock = 0;
	char queued_variant[96] = "", queued_alt_variant[96] = "", last_variant[96] = ""; // instantiation names (mrt_last_kernel_variant)
	uint32_t queued_kernel = 0, queued_alt_kernel = 0; bool queued_detect = false; // what the last enqueue_cast put on the stream
	// host-array pipeline (cast_host_pipelined): copy streams and per-chunk events, created on first use
	hipStream_t up_stream = nullptr, dn_stream = nullptr;
	std::vector<hipEvent_t> pipe_ev;
	// async state
	bool pending = false;
	uint64_t pending_count = 0; uint32_t pending_flags = 0; int pending_mode = 0;
	const void *pending_dev_hits = nullptr;
	mrt_stats stats{};
};

namespace {

#define HIP_TRY(ctx, call)                                                                          \
	do {                                                                                            \
		hipError_t e_ = (call);                                                                     \
		if (e_ != hipSuccess) {                                                                     \
			std::snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call,           \
					hipGetErrorString(e_), __FILE__, __LINE__);                                      \
			return MRT_ERR_HIP;                                                                     \
		}                                                                                           \
	} while (0)

int fail(mrt_ctx *ctx, int code, const char *msg)
{
	if (ctx) std::snprintf(ctx->err, sizeof(ctx->err), "%s", msg);
	return code;
}

int ensure(mrt_ctx *ctx, DevBuf &b, size_t bytes)
{
	if (b.cap >= bytes) return MRT_OK;
	size_t want = bytes + bytes / 2; // grow x1.5
	if (b.ptr) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); HIP_TRY(ctx, hipFree(b.ptr)); b.ptr = nullptr; b.cap = 0; }
	hipError_t e = hipMalloc(&b.ptr, want);
	if (e != hipSuccess) { want = bytes; e = hipMalloc(&b.ptr, want); }
	if (e != hipSuccess) { b.ptr = nullptr; return fail(ctx, MRT_ERR_OOM, "device allocation failed"); }
	b.cap = want;
	return MRT_OK;
}

void release(DevBuf &b) { if (b.ptr) (void)hipFree(b.ptr); b.ptr = nullptr; b.cap = 0; }

void free_scene(mrt_ctx *ctx)
{
	if (ctx->d_nodes) (void)hipFree(ctx->d_nodes);
	if (ctx->d_hot) (void)hipFree(ctx->d_hot);
	if (ctx->d_cold) (void)hipFree(ctx->d_cold);
	if (ctx->d_nodes4) (void)hipFree(ctx->d_nodes4);
	if (ctx->d_nodes8) (void)hipFree(ctx->d_nodes8);
	if (ctx->d_leaf_box) (void)hipFree(ctx->d_leaf_box);
	ctx->d_leaf_box = nullptr;
	if (ctx->d_rows) (void)hipFree(ctx->d_rows);
	ctx->d_rows = nullptr;
	if (ctx->d_rows4) (void)hipFree(ctx->d_rows4);
	ctx->d_rows4 = nullptr;
	if (ctx->d_instances) (void)hipFree(ctx->d_instances);
	ctx->d_instances = nullptr;
	if (ctx->two_level) { mrt::free_two_level(ctx->two_level); delete ctx->two_level; ctx->two_level = nullptr; }
	ctx->d_nodes = nullptr; ctx->d_hot = nullptr; ctx->d_cold = nullptr; ctx->d_nodes4 = nullptr; ctx->d_nodes8 = nullptr;
	ctx->n_nodes8 = ctx->stack8 = 0;
	ctx->scene = false; ctx->n_nodes = ctx->n_tris = 0;
	for (auto &g : ctx->grid_states) { g.sched.forget(); g.tune.phase = 0; g.tune.mode = -1; } // what was learnt about the old scene's grids
}

size_t ray_stride(uint32_t flags) { return (flags & MRT_FLAG_HOST_LAYOUT) ? sizeof(mrt_host_ray60) : sizeof(mrt_ray32); }
// hit tokens: 4 bytes (flat scenes: the winning triangle's slot), 8 for a two-level scene ({triangle slot, instance row})
size_t token_bytes(const mrt_ctx *ctx) { return ctx->two_level ? 8 : 4; }
size_t hit_stride(const mrt_ctx *ctx, uint32_t flags, int mode)
{
	if ((flags & MRT_FLAG_BOOL_OUT) && mode == MRT_MODE_ANY_HIT) return 1;
	if (flags & MRT_FLAG_TOKEN_OUT) return token_bytes(ctx);
	return (flags & MRT_FLAG_HOST_LAYOUT) ? sizeof(mrt_host_hit44) : sizeof(mrt_hit32);
}

uint32_t out_format(const mrt_ctx *ctx, uint32_t flags, int mode)
{
	if ((flags & MRT_FLAG_BOOL_OUT) && mode == MRT_MODE_ANY_HIT) return mrt::OUT_BOOL8;
	if (flags & MRT_FLAG_TOKEN_OUT) return ctx->two_level ? mrt::OUT_TOKEN8 : mrt::OUT_TOKEN4;
	return (flags & MRT_FLAG_HOST_LAYOUT) ? mrt::OUT_HOST44 : mrt::OUT_HIT32;
}

void base_params(mrt_ctx *ctx, mrt::TraceParams &p)
{
	std::memset(&p, 0, sizeof(p));
	p.nodes = ctx->d_nodes; p.nodes4 = ctx->d_nodes4; p.nodes8 = ctx->d_nodes8; p.leaf_box = ctx->d_leaf_box; p.instances = ctx->d_instances; p.tri_hot = ctx->d_hot; p.tri_cold = ctx->d_cold; p.row_array = ctx->d_rows; p.row_array4 = ctx->d_rows4; p.tri_unit_base4 = 2u * ctx->n_nodes4;
	p.stack_depth = ctx->stack_depth; p.n_tris = ctx->n_tris; p.n_nodes = ctx->n_nodes;
	p.n_instances = ctx->two_level ? ctx->two_level->n_inst : 0u;
	if (ctx->opts.stack_override >= ctx->depth && ctx->opts.stack_override <= 64) p.stack_depth = ctx->opts.stack_override;
	p.counters = ctx->d_counters;
	p.xcd_swizzle = ctx->opts.xcd_swizzle ? 1 : 0;
	p.tile_w_log2 = (ctx->opts.tile_w_log2 >= 1 && ctx->opts.tile_w_log2 <= 6) ? ctx->opts.tile_w_log2 : 3;
	// Z-order tiles keep the packets in flight on a compact image region.  That pays once the scene no longer
	// fits the 256 MB Infinity Cache (C5, 1.3 GB of nodes + triangles: 24.6 -> 23.5 ms) and costs 3-5 % while it
	// does (C2, C3), so the default goes by the size of the scene.
	const size_t scene_bytes = (size_t)ctx->n_nodes * sizeof(mrt::DevNode) + (size_t)ctx->n_tris * (sizeof(mrt::TriHot) + sizeof(mrt::TriCold));
	p.tile_order = ctx->opts.tile_order == 2 || (ctx->opts.tile_order == 0 && scene_bytes > (size_t)256 << 20) ? 1u : 0u;
	if (ctx->opts.tile_order == 3) p.tile_order = 2u; // 32x32-tile super-tiles (C5: 23.3 against 23.5 ms; not the default)
	if (ctx->opts.tile_order == 4) p.tile_order = 3u; // column strips per XCD (kernels.hip, xcd_strips)
	p.extra_lds = ctx->opts.extra_lds <= 60000u ? ctx->opts.extra_lds : 60000u;
	p.count_mode = ctx->opts.count_visits;
	// packet-level frustum culling in the 128-ray walk (packet_rows_kernel.h): 0 off, 1 on, 2 = by where the rays come from
	// (launch_trace): on for rays generated in the kernel (C3 1.84 against 1.86 ms, C5 20.4 against 21.1), off for rays read from
	// memory, where the walk with the scalar-cache prefetch waits less (C3 1.89 against 1.93 ms; profiles/r03_cull_cast_vs_fused.txt)
	p.rows_cull = ctx->opts.packet_cull == 1u ? 0u : (ctx->opts.packet_cull == 2u ? 1u : 2u);
	p.scene_abs_max = 0.0f;
	for (int c = 0; c < 3; c++) p.scene_abs_max = std::fmax(p.scene_abs_max, std::fmax(std::fabs(ctx->bounds_lo[c]), std::fabs(ctx->bounds_hi[c])));
	p.rows_wg = ctx->opts.packet_wg == 64u || ctx->opts.packet_wg == 256u ? ctx->opts.packet_wg : (scene_bytes > (size_t)256 << 20 ? 256u : 64u);
	p.kernel = MRT_KERNEL_LANE; // callers pick per batch with pick_kernel()
}

// MRT_KERNEL_AUTO: packets for batches the caller declares coherent (RayQuery::coherent,
// primary-ray grids), one lane per ray for everything else (sorted / incoherent batches).
uint32_t pick_kernel(const mrt_ctx *ctx, bool coherent, uint64_t count)
{
	// a two-level scene has its own pair of kernels (two_level_kernel.h)
	// Packets pay once there are enough of them: a wave that walks for 64 rays is a long serial chain (0.3 - 0.7 ms on a
	// 1 M-triangle scene, the longer the wider its 8x8 tile opens), and a small batch is over when its slowest wave is.
	// Coherent grids on the C3 scene (tools/bench_small_batches.py, profiles/r02d_small_batches.txt): 64^2 rays 0.69 ms by
	// packets, 0.32 ms one lane per ray; 128^2 0.56 / 0.40; 256^2 0.40 / 0.43; 512^2 0.35 / 0.59 (C2 scene: even at 128^2).
	// (Batches whose tiling is known or found -- grids, tiled casts, mrt_cast(COHERENT) -- leave this rule from 2^11 rays on:
	// quarter_small_grid() runs them by packets of 16 rays, faster than either.)
	const bool few = ctx->opts.kernel == MRT_KERNEL_AUTO && count < (1ull << 15);
	if (ctx->two_level) return coherent && !few && ctx->opts.kernel != MRT_KERNEL_LANE ? mrt::MRT_KERNEL_TWO_LEVEL_PACKET : mrt::MRT_KERNEL_TWO_LEVEL;
	if (ctx->opts.kernel == MRT_KERNEL_PACKET_DUAL || ctx->opts.kernel == MRT_KERNEL_PACKET_ROWS)
		return !coherent ? MRT_KERNEL_LANE : (ctx->d_rows ? ctx->opts.kernel : MRT_KERNEL_PACKET_ASM);
	if (ctx->opts.kernel == MRT_KERNEL_PACKET_QUAD)
		return !coherent ? MRT_KERNEL_LANE : (ctx->d_rows4 ? MRT_KERNEL_PACKET_QUAD : MRT_KERNEL_PACKET_ASM);
	if (ctx->opts.kernel >= MRT_KERNEL_LANE && ctx->opts.kernel <= MRT_KERNEL_LANE8_PERSISTENT) return ctx->opts.kernel;
	if (!coherent || few) return MRT_KERNEL_LANE;
	// Coherent batches: the 128-ray shared walk over the row array (packet_rows_kernel.h) once the batch is large
	// enough to fill the chip with half as many waves (C3 2.16 -> 2.06 ms, C5 23.2 -> 21.2 ms; C2's 2^20 rays are
	// 7 % faster with one packet per wave: 0.188 against 0.202 ms), else the 64-ray packet kernel with the
	// hand-written node loop.
	return (ctx->d_rows && count >= (1ull << 22)) ? MRT_KERNEL_PACKET_DUAL : MRT_KERNEL_PACKET_ASM;
}

// Flat scenes: the unified row array of the assembly packet walk, built on the device from the arrays just
// uploaded / built.  Optional: a scene too large for its 26-bit row index, or a device short of memory, goes
// without (coherent batches then take trace_packet_asm_kernel).
int build_rows(mrt_ctx *ctx)
{
	// the 4-wide rows: when the 4-wide layout is resident and its worst-case stack fits the wave's 64 entries
	const bool wanted4 = ctx->opts.kernel == MRT_KERNEL_PACKET_QUAD && mrt::quad_kernel_built(); // (AUTO never picks the four-wide walk: it is no faster)
	const uint64_t n_units4 = (uint64_t)2u * ctx->n_nodes4 + ctx->n_tris;
	if (wanted4 && ctx->d_nodes4 && ctx->n_nodes4 && ctx->stack4 <= 64u && n_units4 < mrt::kAsmNodeLimit) {
		if (hipMalloc(&ctx->d_rows4, (size_t)n_units4 * 64u) != hipSuccess) { ctx->d_rows4 = nullptr; (void)hipGetLastError(); }
		else {
			HIP_TRY(ctx, mrt::launch_build_rows4(ctx->d_nodes4, ctx->d_hot, ctx->d_cold, ctx->n_nodes4, ctx->n_tris, ctx->d_rows4, ctx->stream));
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		}
	}
	const bool wanted = ctx->opts.kernel == MRT_KERNEL_AUTO || ctx->opts.kernel == MRT_KERNEL_PACKET_DUAL || ctx->opts.kernel == MRT_KERNEL_PACKET_ROWS;
	const uint64_t n_rows = (uint64_t)ctx->n_nodes + ctx->n_tris;
	if (!wanted || n_rows >= mrt::kAsmNodeLimit) return MRT_OK;
	if (hipMalloc(&ctx->d_rows, (size_t)n_rows * 64u) != hipSuccess) { ctx->d_rows = nullptr; (void)hipGetLastError(); return MRT_OK; }
	HIP_TRY(ctx, mrt::launch_build_rows(ctx->d_nodes, ctx->d_hot, ctx->d_cold, ctx->n_nodes, ctx->n_tris, ctx->d_rows, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MRT_OK;
}

int drain_pending(mrt_ctx *ctx)
{
	if (ctx->pending) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); ctx->pending = false; }
	return MRT_OK;
}

// Sorts ray indices by direction Morton key on the device; returns the permutation in idx_out.
int device_sort(mrt_ctx *ctx, const void *d_rays, uint32_t in_fmt, uint64_t count, const uint32_t **perm)
{
	if (count > 0xFFFFFFFFull) return fail(ctx, MRT_ERR_UNSUPPORTED, "sorted batches are limited to 2^32-1 rays");
	int rc;
	if ((rc = ensure(ctx, ctx->keys_in, count * 4)) || (rc = ensure(ctx, ctx->keys_out, count * 4)) ||
			(rc = ensure(ctx, ctx->idx_in, count * 4)) || (rc = ensure(ctx, ctx->idx_out, count * 4))) return rc;
	uint32_t *ki = (uint32_t *)ctx->keys_in.ptr, *ko = (uint32_t *)ctx->keys_out.ptr;
	uint32_t *ii = (uint32_t *)ctx->idx_in.ptr, *io = (uint32_t *)ctx->idx_out.ptr;
	if (ctx->opts.sort_key == 1) // the reference's direction-only key (ray_sort.h:64-76)
		HIP_TRY(ctx, mrt::launch_morton_keys(d_rays, in_fmt, count, ki, ii, ctx->stream));
	else // origin cell first, then direction: groups rays whose origins are scattered too
		HIP_TRY(ctx, mrt::launch_origin_dir_keys(d_rays, in_fmt, count, ctx->bounds_lo, ctx->bounds_hi, ki, ii, ctx->stream));
	size_t tmp_bytes = 0;
	HIP_TRY(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, ki, ko, ii, io, (size_t)count, 0, 30, ctx->stream));
	if ((rc = ensure(ctx, ctx->sort_tmp, tmp_bytes ? tmp_bytes : 16))) return rc;
	HIP_TRY(ctx, rocprim::radix_sort_pairs(ctx->sort_tmp.ptr, tmp_bytes, ki, ko, ii, io, (size_t)count, 0, 30, ctx->stream));
	*perm = io;
	return MRT_OK;
}

} // namespace
// (defined with the grid casts below)
constexpr int kTuneFrames = 4; // frames per candidate of the grid kernel tuner (tune_grid_kernel): the last two are timed
static bool schedule_applies(const mrt_ctx *ctx, const mrt::TraceParams &p);
static int schedule_grid(mrt_ctx *ctx, mrt::TraceParams &p);
static int schedule_sort(mrt_ctx *ctx);
// the state of this grid and cast mode (mrt_ctx::GridState): the entry that holds it, else the least recently used one
static void select_grid_state(mrt_ctx *ctx, uint32_t w, uint32_t h, uint32_t y0, uint32_t rows, int mode)
{
	mrt_ctx::GridState *pick = nullptr, *oldest = &ctx->grid_states[0];
	for (auto &g : ctx->grid_states) {
		if (g.k_mode == mode && g.k_w == w && g.k_h == h && g.k_y0 == y0 && g.k_rows == rows) { pick = &g; break; }
		if (g.stamp < oldest->stamp) oldest = &g;
	}
	if (!pick) { pick = oldest; pick->k_w = w; pick->k_h = h; pick->k_y0 = y0; pick->k_rows = rows; pick->k_mode = mode; }
	pick->stamp = ++ctx->gs_clock;
	ctx->gs = pick;
}
static void tune_grid_kernel(mrt_ctx *ctx, mrt::TraceParams &p, int mode, uint32_t flags);
static void tune_record(mrt_ctx *ctx);
constexpr uint64_t kQuarterMinRays = 64, kQuarterMaxTiles = 3600, kSixteenthMaxTiles = 512; // small grids in quarter / sixteenth tiles: see quarter_small_grid()
constexpr uint64_t kQuarterAllRays = 2048ull * 64ull; // up to here four quarters per tile still fit one round of waves: no schedule needed
static void quarter_small_grid(const mrt_ctx *ctx, mrt::TraceParams &p);
namespace {

// Lane kernel launch: plain (one fixed ray per lane) or persistent (resident waves pulling rays
// from a counter, short LDS stack with HBM spill).
int launch_lane(mrt_ctx *ctx, mrt::TraceParams &p, uint64_t count, bool any_hit, bool persistent)
{
	int rc;
	// The wide walks exist in persistent form only.  For large incoherent batches MRT_KERNEL_AUTO takes the
	// 8-wide compressed layout when the scene has it (6.1 ms at C4), else the 4-wide one (7.7 ms), else the
	// 2-wide persistent kernel (11.0 ms).
	const bool wide4 = ctx->d_nodes4 != nullptr && (ctx->opts.kernel == MRT_KERNEL_LANE4_PERSISTENT ||
			(ctx->opts.kernel == MRT_KERNEL_AUTO && persistent));
	const bool wide8 = ctx->d_nodes8 != nullptr && (ctx->opts.kernel == MRT_KERNEL_LANE8_PERSISTENT ||
			(ctx->opts.kernel == MRT_KERNEL_AUTO && persistent));
	// counting builds: the persistent kernels count for flat scenes; two-level scenes take the plain lane kernel
	const bool can_count = !ctx->opts.count_visits || !ctx->two_level;
	if ((wide4 || wide8) && can_count) persistent = true;
	if (!persistent || !can_count) {
		p.kernel = ctx->two_level ? mrt::MRT_KERNEL_TWO_LEVEL : MRT_KERNEL_LANE;
		// A small batch on more, emptier waves: with fewer rays than the device has wave slots (8 192) every ray gets a wave of its
		// own, up to 2^15 rays two or four share one.  A wave's walk is as long as its longest ray's and every step costs as many
		// memory requests as it has rays; a batch this small ends with its longest wave (blocking mrt_cast of incoherent rays in
		// host arrays: 256 rays 190 -> 97 us, 1 024 rays 247 -> 125, 4 096 rays 317 -> 210; 2^14 device-resident rays 503 -> 338;
		// on the C5 two-level scene 256 rays 1 649 -> 319 us, 4 096 rays 2 272 -> 646; profiles/r03_latency.txt)
		if (p.lane_map == mrt::MAP_LINEAR && count <= (8192u << 2)) {
			uint32_t lanes = 1u;
			while ((count + lanes - 1u) / lanes > 8192u) lanes <<= 1;
			p.sparse_lanes = lanes;
		}
		HIP_TRY(ctx, mrt::launch_trace(p, any_hit, ctx->opts.count_visits != 0, ctx->stream));
		ctx->queued_kernel = p.kernel; std::snprintf(ctx->queued_variant, sizeof(ctx->queued_variant), "%s", mrt::last_trace_variant());
		return MRT_OK;
	}
	const uint32_t lds_depth = ctx->opts.stack_override >= 4 && ctx->opts.stack_override <= 64 ? ctx->opts.stack_override : 16u;
	const uint32_t lds_bytes = 4u * lds_depth * 64u * 4u; // per 256-thread workgroup
	uint32_t wg_per_cu = (160u * 1024u) / lds_bytes; if (wg_per_cu > 8u) wg_per_cu = 8u;
	uint64_t blocks = (uint64_t)ctx->cu_count * wg_per_cu;
	const uint64_t needed = (count + 255u) / 256u;
	if (blocks > needed) blocks = needed;
	uint32_t *ovf = nullptr;
	const uint32_t need = wide8 ? ctx->stack8 : (wide4 ? ctx->stack4 : ctx->depth); // entries one ray can have pending
	if (need > lds_depth) { // deeper entries spill to [depth - lds_depth][thread] in HBM
		if ((rc = ensure(ctx, ctx->overflow, (size_t)(need - lds_depth) * blocks * 256u * 4u))) return rc;
		ovf = (uint32_t *)ctx->overflow.ptr;
	}
	p.kernel = wide8 ? MRT_KERNEL_LANE8_PERSISTENT : (wide4 ? MRT_KERNEL_LANE4_PERSISTENT : MRT_KERNEL_LANE_PERSISTENT);
	if (ctx->two_level) p.kernel = wide8 ? mrt::MRT_KERNEL_TWO_LEVEL_PERSISTENT8 : mrt::MRT_KERNEL_TWO_LEVEL_PERSISTENT; // need = stack8 (= depth8) / depth
	// eight ray counters (one per region of the batch), 128 bytes apart
	unsigned long long *next_ray = ctx->d_counters + mrt::kNextRayOff;
	HIP_TRY(ctx, hipMemsetAsync(next_ray, 0, 128 * sizeof(unsigned long long), ctx->stream));
	HIP_TRY(ctx, mrt::launch_trace_persistent(p, next_ray, ovf, lds_depth, ctx->opts.refill ? ctx->opts.refill : 16u,
			ctx->opts.leaf_wait ? ctx->opts.leaf_wait : (wide8 ? 8u : 16u), (uint32_t)blocks, any_hit,
			ctx->opts.count_visits != 0 && !ctx->two_level, ctx->stream));
	ctx->queued_kernel = p.kernel; std::snprintf(ctx->queued_variant, sizeof(ctx->queued_variant), "%s", mrt::last_trace_variant());
	return MRT_OK;
}

// Enqueue H2D (if needed) + optional sort + trace.  On return the kernels are queued on ctx->stream.
int enqueue_cast(mrt_ctx *ctx, const void *rays, void *hits_dev_or_null, uint64_t count, uint32_t query_mask,
		int mode, uint32_t flags, void **d_hits_out)
{
	if (!ctx->scene) return fail(ctx, MRT_ERR_NO_SCENE, "no scene uploaded (is_available() == false)");
	if (mode != MRT_MODE_NEAREST && mode != MRT_MODE_ANY_HIT) return fail(ctx, MRT_ERR_INVALID, "bad mode");
	if ((flags & MRT_FLAG_BOOL_OUT) && mode != MRT_MODE_ANY_HIT) return fail(ctx, MRT_ERR_INVALID, "BOOL_OUT needs any-hit mode");
	ctx->gs->tune.armed = false; // (the grid tuner times a cast only if THIS cast asks it to, and only mrt_cast / mrt_cast_grid record)
	if ((flags & MRT_FLAG_BOOL_OUT) && (flags & MRT_FLAG_TOKEN_OUT)) return fail(ctx, MRT_ERR_INVALID, "BOOL_OUT and TOKEN_OUT exclude each other");
	const size_t rs = ray_stride(flags), hs = hit_stride(ctx, flags, mode);
	int rc;
	const void *d_rays = rays;
	ctx->stats.last_h2d_ms = ctx->stats.last_d2h_ms = ctx->stats.last_sort_ms = 0.0f;
	if (!(flags & MRT_FLAG_RAYS_ON_DEVICE)) {
		if ((rc = ensure(ctx, ctx->rays, count * rs))) return rc;
		HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->rays.ptr, rays, count * rs, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
		d_rays = ctx->rays.ptr;
	}
	void *d_hits = hits_dev_or_null;
	if (!d_hits) {
		if ((rc = ensure(ctx, ctx->hits, count * hs))) return rc;
		d_hits = ctx->hits.ptr;
	}
	mrt::TraceParams p;
	base_params(ctx, p);
	p.rays = d_rays; p.hits = d_hits; p.count = count; p.query_mask = query_mask;
	p.in_fmt = (flags & MRT_FLAG_HOST_LAYOUT) ? mrt::IN_HOST60 : mrt::IN_RAY32;
	p.out_fmt = out_format(ctx, flags, mode);
	p.lane_map = mrt::MAP_LINEAR;
	const uint32_t thr = ctx->opts.sort_threshold ? ctx->opts.sort_threshold : 256u; // MIN_BATCH_FOR_SORTING
	// (a batch of at most 8 192 rays runs one ray per wave in the lane kernel, launch_lane: there is no wave whose rays a sort could
	// bring together, and its three launches are a third of such a cast's time)
	const bool one_ray_waves = ctx->opts.kernel == MRT_KERNEL_AUTO && count <= 8192u && !(flags & MRT_FLAG_FORCE_SORT);
	const bool sort = !(flags & MRT_FLAG_COHERENT) && !one_ray_waves && (count >= thr || (flags & MRT_FLAG_FORCE_SORT));
	HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
	p.kernel = pick_kernel(ctx, !sort && (flags & MRT_FLAG_COHERENT), count);
	if (sort) {
		const uint32_t *perm = nullptr;
		if ((rc = device_sort(ctx, d_rays, p.in_fmt, count, &perm))) return rc;
		p.perm = perm;
	}
	// Coherent batch without a declared width: look for the row width on the device and let the
	// trace kernel tile its lanes (no host round trip: the kernel reads the answer from HBM).
	// Timed with the sort as pre-processing (last_sort_ms); last_trace_ms is the trace kernel alone.
	const bool persistent_kind = p.kernel == MRT_KERNEL_LANE_PERSISTENT || p.kernel == MRT_KERNEL_LANE4_PERSISTENT ||
			p.kernel == MRT_KERNEL_LANE8_PERSISTENT;
	const bool detect = !sort && (flags & MRT_FLAG_COHERENT) && count >= 256 && ctx->opts.grid_tile != 1 && !persistent_kind;
	if (detect) {
		uint32_t *d_auto = reinterpret_cast<uint32_t *>(ctx->d_counters + mrt::kAutoGridOff);
		HIP_TRY(ctx, mrt::launch_detect_grid(d_rays, p.in_fmt, count, p.tile_w_log2, ctx->d_counters + mrt::kDetectScratchOff, d_auto, ctx->d_auto_host, ctx->stream));
		p.lane_map = mrt::MAP_AUTO; p.auto_grid = d_auto;
	}
	// a small batch whose width the device finds: sixteenth or quarter tiles (quarter_small_grid; if no width is found the lanes stay
	// linear and the waves past the batch have nothing to do)
	if (detect && ctx->opts.kernel == MRT_KERNEL_AUTO && !ctx->opts.count_visits && p.tile_w_log2 == 3u && p.n_nodes < mrt::kAsmNodeLimit &&
			count >= kQuarterMinRays && count <= kQuarterMaxTiles * 64u && (p.kernel == MRT_KERNEL_PACKET_ASM || p.kernel == MRT_KERNEL_LANE ||
				p.kernel == mrt::MRT_KERNEL_TWO_LEVEL_PACKET || p.kernel == mrt::MRT_KERNEL_TWO_LEVEL)) {
		p.kernel = ctx->two_level ? mrt::MRT_KERNEL_TWO_LEVEL_PACKET : MRT_KERNEL_PACKET_ASM; p.quarter_all = count <= kSixteenthMaxTiles * 64u ? 2u : 1u;
	}
	if (ctx->opts.count_visits) HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, mrt::kNumCounters * sizeof(unsigned long long), ctx->stream));
	const bool any = mode == MRT_MODE_ANY_HIT;
	// the tile schedule for a batch whose width the device finds: sized from what the previous cast of as many rays found
	bool scheduled = false;
	if (detect && !ctx->pending && ctx->last_detect_count == count && ctx->h_auto[0] != 0u && ctx->h_auto[3] == 0u && schedule_applies(ctx, p) &&
			!(p.quarter_all && count <= kQuarterAllRays)) {
		p.quarter_all = 0u;
		mrt::TraceParams g = p;
		g.grid_w = ctx->h_auto[0]; g.rows = ctx->h_auto[1]; g.grid_h = g.rows; g.y0 = 0; g.tiles_x = ctx->h_auto[2];
		// ... and so is the way it is cast: the grid tuner's candidates, as for a grid cast of that width (mrt_cast records the timing)
		select_grid_state(ctx, g.grid_w, g.grid_h, 0u, g.rows, mode);
		g.lane_map = mrt::MAP_TILE8X8;
		tune_grid_kernel(ctx, g, mode, flags);
		p.kernel = g.kernel; g.lane_map = p.lane_map;
		if ((rc = schedule_grid(ctx, g))) return rc;
		p.tile_sched = g.tile_sched; p.tile_cost = g.tile_cost; p.tile_unit = g.tile_unit; p.n_units = g.n_units; p.sched_hdr = g.sched_hdr; p.n_slots_max = g.n_slots_max;
		scheduled = true;
	}
	ctx->last_detect_count = detect ? count : 0;
	HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
	ctx->queued_detect = detect; ctx->queued_alt_kernel = 0;
	if (detect && ctx->opts.kernel == MRT_KERNEL_AUTO && !ctx->opts.count_visits && p.kernel != MRT_KERNEL_LANE && p.kernel != mrt::MRT_KERNEL_TWO_LEVEL) {
		// The caller said "coherent"; the device checks.  Packet launch first, lane launch behind it:
		// detect_grid_kernel's verdict (d_auto[3]) makes exactly one of them do the work.
		p.skip_flag = p.auto_grid + 3; p.skip_when = 1u;
		HIP_TRY(ctx, mrt::launch_trace(p, any, false, ctx->stream));
		const uint32_t packet_kernel = p.kernel;
		char packet_variant[96]; std::snprintf(packet_variant, sizeof(packet_variant), "%s", mrt::last_trace_variant());
		mrt::TraceParams lp = p;
		lp.kernel = MRT_KERNEL_LANE; lp.lane_map = mrt::MAP_LINEAR; lp.auto_grid = nullptr; lp.skip_when = 0u; lp.quarter_all = 0u;
		if ((rc = launch_lane(ctx, lp, count, any, count >= 65536))) return rc; // (a two-level scene: its own lane kernels)
		ctx->queued_alt_kernel = ctx->queued_kernel; // what launch_lane queued: runs if the batch is judged incoherent
		std::snprintf(ctx->queued_alt_variant, sizeof(ctx->queued_alt_variant), "%s", ctx->queued_variant);
		ctx->queued_kernel = packet_kernel; std::snprintf(ctx->queued_variant, sizeof(ctx->queued_variant), "%s", packet_variant);
	} else {
		// large incoherent batches: resident waves that pull rays from a counter (no counting variant)
		const bool persistent = p.lane_map == mrt::MAP_LINEAR &&
				(persistent_kind || (ctx->opts.kernel == MRT_KERNEL_AUTO && count >= 65536 &&
					(p.kernel == MRT_KERNEL_LANE || p.kernel == mrt::MRT_KERNEL_TWO_LEVEL)));
		if (persistent_kind || p.kernel == MRT_KERNEL_LANE || p.kernel == mrt::MRT_KERNEL_TWO_LEVEL) {
			if ((rc = launch_lane(ctx, p, count, any, persistent))) return rc;
		} else {
			HIP_TRY(ctx, mrt::launch_trace(p, any, ctx->opts.count_visits != 0, ctx->stream));
			ctx->queued_kernel = p.kernel; std::snprintf(ctx->queued_variant, sizeof(ctx->queued_variant), "%s", mrt::last_trace_variant());
		}
	}
	HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
	if (scheduled && (rc = schedule_sort(ctx))) return rc;
	ctx->stats.last_kernel_launches = sort ? 3 : (detect ? 2 : 1);
	ctx->stats.rays_cast += count;
	*d_hits_out = d_hits;
	return MRT_OK;
}

constexpr uint64_t kSmallCast = 1024; // rays: host-array casts up to this size take the mapped-memory path of mrt_cast

int finish_timing(mrt_ctx *ctx, bool h2d, bool sorted, bool d2h)
{
	float ms = 0.0f;
	if (h2d) { HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1])); ctx->stats.last_h2d_ms = ms; }
	if (sorted) { HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3])); ctx->stats.last_sort_ms = ms; }
	HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4])); ctx->stats.last_trace_ms = ms;
	if (d2h) { HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[5])); ctx->stats.last_d2h_ms = ms; }
	// which kernel did the work: the stream has been waited for, so detect_grid_kernel's words are in h_auto
	ctx->stats.detected_grid_w = ctx->queued_detect ? ctx->h_auto[0] : 0u;
	ctx->stats.reserved = ctx->queued_detect ? ctx->h_auto[3] : 0u; // 1: the "coherent" batch was judged incoherent
	const bool alt_ran = ctx->queued_detect && ctx->queued_alt_kernel && ctx->h_auto[3];
	ctx->stats.last_kernel = alt_ran ? ctx->queued_alt_kernel : ctx->queued_kernel;
	std::snprintf(ctx->last_variant, sizeof(ctx->last_variant), "%s", alt_ran ? ctx->queued_alt_variant : ctx->queued_variant);
	if (ctx->opts.count_visits) {
		unsigned long long c[mrt::kNumCounters];
		HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
		ctx->stats.tri_tests += c[mrt::kCntTris]; ctx->stats.bvh_nodes_visited += c[mrt::kCntNodes]; ctx->stats.hits += c[mrt::kCntHits];
		if ((uint32_t)c[mrt::kCntMaxStack] > ctx->stats.max_stack_depth) ctx->stats.max_stack_depth = (uint32_t)c[mrt::kCntMaxStack];
		ctx->stats.dead_pops += c[mrt::kCntDeadPops];
		ctx->stats.wave_node_fetches += c[mrt::kCntWaveNodeFetch]; ctx->stats.wave_tri_fetches += c[mrt::kCntWaveTriFetch];
		ctx->stats.leaf_box_checks += c[mrt::kCntLeafBoxChecks];
		ctx->stats.fetch_wait_cycles += c[mrt::kCntFetchWaitCycles]; ctx->stats.wave_cycles += c[mrt::kCntWaveCycles];
		ctx->stats.waves += c[mrt::kCntWaves];
	}
	return MRT_OK;
}

// Host arrays in, host arrays out (the reference's cast_rays contract), large batch: upload, trace and
// download run as a pipeline over 2^20-ray chunks.  A pageable copy occupies the host thread that
// issues it, so uploads are issued from the calling thread and downloads from a helper thread: both
// PCIe directions then move data at once, and the trace of a chunk hides between them.
constexpr uint64_t kPipeChunk = 1ull << 20;

int cast_host_pipelined(mrt_ctx *ctx, const void *rays, void *hits, uint64_t count, uint32_t query_mask, int mode, uint32_t flags)
{
	const size_t rs = ray_stride(flags), hs = hit_stride(ctx, flags, mode);
	const uint32_t n_chunks = (uint32_t)((count + kPipeChunk - 1) / kPipeChunk);
	int rc;
	if ((rc = ensure(ctx, ctx->rays, count * rs)) || (rc = ensure(ctx, ctx->hits, count * hs))) return rc;
	if (!ctx->up_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->up_stream, hipStreamNonBlocking));
	if (!ctx->dn_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->dn_stream, hipStreamNonBlocking));
	while (ctx->pipe_ev.size() < 2u * n_chunks) {
		hipEvent_t e;
		HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
		ctx->pipe_ev.push_back(e);
	}
	char *d_rays = (char *)ctx->rays.ptr, *d_hits = (char *)ctx->hits.ptr;
	std::atomic<uint32_t> traced{0};      // chunks whose trace has been queued (their event is recorded)
	std::atomic<int> stop{0}, down_err{0};
	std::thread down([&] {
		if (hipSetDevice(ctx->device) != hipSuccess) { down_err = (int)hipErrorInvalidDevice; return; }
		for (uint32_t k = 0; k < n_chunks; k++) {
			while (traced.load(std::memory_order_acquire) <= k) { if (stop.load()) return; std::this_thread::yield(); }
			const uint64_t off = (uint64_t)k * kPipeChunk, n = count - off < kPipeChunk ? count - off : kPipeChunk;
			hipError_t e = hipEventSynchronize(ctx->pipe_ev[2 * k + 1]);
			if (e == hipSuccess) e = hipMemcpyAsync((char *)hits + off * hs, d_hits + off * hs, n * hs, hipMemcpyDeviceToHost, ctx->dn_stream);
			if (e == hipSuccess) e = hipStreamSynchronize(ctx->dn_stream);
			if (e != hipSuccess) { down_err = (int)e; return; }
		}
	});
	const uint32_t dev_flags = flags | MRT_FLAG_RAYS_ON_DEVICE | MRT_FLAG_HITS_ON_DEVICE;
	hipError_t e = hipSuccess;
	uint32_t launches = 0;
	for (uint32_t k = 0; k < n_chunks && e == hipSuccess && rc == MRT_OK && !down_err.load(); k++) {
		const uint64_t off = (uint64_t)k * kPipeChunk, n = count - off < kPipeChunk ? count - off : kPipeChunk;
		e = hipMemcpyAsync(d_rays + off * rs, (const char *)rays + off * rs, n * rs, hipMemcpyHostToDevice, ctx->up_stream);
		if (e == hipSuccess) e = hipEventRecord(ctx->pipe_ev[2 * k], ctx->up_stream);
		if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->pipe_ev[2 * k], 0);
		if (e != hipSuccess) break;
		void *unused = nullptr;
		rc = enqueue_cast(ctx, d_rays + off * rs, d_hits + off * hs, n, query_mask, mode, dev_flags, &unused);
		ctx->gs->tune.armed = false; // (chunks of a pipeline are not timed one by one)
		if (rc) break;
		if (ctx->stats.last_kernel_launches > launches) launches = ctx->stats.last_kernel_launches;
		e = hipEventRecord(ctx->pipe_ev[2 * k + 1], ctx->stream);
		if (e == hipSuccess) traced.store(k + 1, std::memory_order_release);
	}
	if (e != hipSuccess || rc != MRT_OK) stop = 1;
	down.join();
	(void)hipStreamSynchronize(ctx->stream);
	if (rc) return rc;
	if (e != hipSuccess || down_err.load()) {
		std::snprintf(ctx->err, sizeof(ctx->err), "pipelined cast failed: %s", hipGetErrorString(e != hipSuccess ? e : (hipError_t)down_err.load()));
		return MRT_ERR_HIP;
	}
	ctx->stats.last_kernel_launches = launches;
	ctx->stats.last_h2d_ms = ctx->stats.last_d2h_ms = 0.0f; // overlapped: not separable (last_trace_ms is the last chunk's)
	return finish_timing(ctx, false, launches >= 2, false);
}

} // namespace

extern "C" {

uint32_t mrt_version(void) { return (MRT_VERSION_MAJOR << 16) | MRT_VERSION_MINOR; }

uint32_t mrt_struct_size(uint32_t which)
{
	switch (which) {
		case 0: return (uint32_t)sizeof(mrt_options);
		case 1: return (uint32_t)sizeof(mrt_camera);
		case 2: return (uint32_t)sizeof(mrt_stats);
		case 3: return (uint32_t)sizeof(mrt_instance);
		default: return 0u;
	}
}

const char *mrt_status_string(int s)
{
	switch (s) {
		case MRT_OK: return "ok";
		case MRT_ERR_INVALID: return "invalid argument";
		case MRT_ERR_NO_DEVICE: return "no such HIP device";
		case MRT_ERR_HIP: return "HIP runtime error";
		case MRT_ERR_NO_SCENE: return "no scene uploaded";
		case MRT_ERR_PENDING: return "an async dispatch is already pending";
		case MRT_ERR_NOT_PENDING: return "no async dispatch pending";
		case MRT_ERR_OOM: return "out of memory";
		case MRT_ERR_UNSUPPORTED: return "unsupported";
		case MRT_ERR_BAD_BVH: return "BVH failed validation";
		default: return "unknown status";
	}
}

const char *mrt_kernel_name(uint32_t kernel)
{
	switch (kernel) {
		case MRT_KERNEL_LANE: return "trace_lane_kernel";
		case MRT_KERNEL_PACKET: return "trace_packet_kernel";
		case MRT_KERNEL_PACKET_ASM: return "trace_packet_asm_kernel";
		case MRT_KERNEL_PACKET_DUAL: return "trace_packet_rows_kernel<2>";
		case MRT_KERNEL_PACKET_ROWS: return "trace_packet_rows_kernel<1>";
		case MRT_KERNEL_PACKET_QUAD: return "trace_packet_quad_kernel";
		case MRT_KERNEL_LANE_PERSISTENT: return "trace_lane_persistent_kernel<2>";
		case MRT_KERNEL_LANE4_PERSISTENT: return "trace_lane_persistent_kernel<4>";
		case MRT_KERNEL_LANE8_PERSISTENT: return "trace_lane_persistent_kernel<8>";
		case MRT_KERNEL_TWO_LEVEL: return "trace_two_level_kernel";
		case MRT_KERNEL_TWO_LEVEL_PACKET: return "trace_two_level_packet_kernel";
		case MRT_KERNEL_TWO_LEVEL_PERSISTENT: return "trace_lane_persistent_kernel<2, two-level>";
		case MRT_KERNEL_TWO_LEVEL_PERSISTENT8: return "trace_lane_persistent_kernel<8, two-level>";
		default: return "?";
	}
}

// 1 if this build of the library contains the kernel (everything but MRT_KERNEL_PACKET_QUAD always; the four-wide packet
// walk only in builds made with MRT_WITH_QUAD)
int mrt_kernel_available(uint32_t kernel)
{
	if (kernel == MRT_KERNEL_PACKET_QUAD) return mrt::quad_kernel_built() ? 1 : 0;
	return std::strcmp(mrt_kernel_name(kernel), "?") != 0 || kernel == MRT_KERNEL_AUTO ? 1 : 0;
}

const char *mrt_last_error(const mrt_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int mrt_create(int device_ordinal, const mrt_options *opts, mrt_ctx **out)
{
	if (!out) return MRT_ERR_INVALID;
	*out = nullptr;
	if (opts && opts->struct_size != sizeof(mrt_options)) return MRT_ERR_INVALID;
	if (opts && opts->kernel == MRT_KERNEL_PACKET_QUAD && !mrt::quad_kernel_built()) return MRT_ERR_UNSUPPORTED; // not in this build (MRT_WITH_QUAD)
	if (opts && ((opts->packet_wg != 0u && opts->packet_wg != 64u && opts->packet_wg != 256u) || opts->packet_cull > 2u || opts->tile_schedule > 2u || opts->kernel > MRT_KERNEL_PACKET_QUAD || opts->kernel == 3u || opts->kernel == 4u)) return MRT_ERR_INVALID; // 3, 4: retired ids
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || device_ordinal < 0 || device_ordinal >= n) return MRT_ERR_NO_DEVICE;
	mrt_ctx *ctx = new (std::nothrow) mrt_ctx();
	if (!ctx) return MRT_ERR_OOM;
	ctx->device = device_ordinal;
	if (opts) ctx->opts = *opts;
	ctx->opts.struct_size = sizeof(mrt_options);
	auto bail = [&](int code) { mrt_destroy(ctx); return code; };
	if (hipSetDevice(device_ordinal) != hipSuccess) return bail(MRT_ERR_NO_DEVICE);
	if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) return bail(MRT_ERR_HIP);
	{
		int cus = 0;
		if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) == hipSuccess && cus > 0) ctx->cu_count = cus;
	}
	ctx->stream = ctx->own_stream;
	for (auto &e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) return bail(MRT_ERR_HIP);
	// visit counters, detected grid, detect_grid_kernel scratch, ray counters: the layout is in mrt_internal.h
	if (hipMalloc(&ctx->d_counters, mrt::kCounterWords * sizeof(unsigned long long)) != hipSuccess) return bail(MRT_ERR_OOM);
	if (hipMemset(ctx->d_counters, 0, mrt::kCounterWords * sizeof(unsigned long long)) != hipSuccess) return bail(MRT_ERR_HIP);
	if (hipHostMalloc((void **)&ctx->h_auto, 64, hipHostMallocMapped) != hipSuccess) return bail(MRT_ERR_OOM);
	std::memset(ctx->h_auto, 0, 64);
	if (hipHostGetDevicePointer((void **)&ctx->d_auto_host, ctx->h_auto, 0) != hipSuccess) return bail(MRT_ERR_HIP);
	*out = ctx;
	return MRT_OK;
}

void mrt_destroy(mrt_ctx *ctx)
{
	if (!ctx) return;
	(void)hipSetDevice(ctx->device);
	if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	free_scene(ctx);
	release(ctx->rays); release(ctx->hits); release(ctx->keys_in); release(ctx->keys_out);
	release(ctx->idx_in); release(ctx->idx_out); release(ctx->sort_tmp); release(ctx->overflow);
	for (auto &g : ctx->grid_states) {
		auto &sc = g.sched;
		if (sc.side) { (void)hipStreamSynchronize(sc.side); (void)hipStreamDestroy(sc.side); }
		if (sc.traced) (void)hipEventDestroy(sc.traced);
		for (int k = 0; k < 2; k++) { if (sc.ready[k]) (void)hipEventDestroy(sc.ready[k]); release(sc.cost[k]); release(sc.order[k]); release(sc.slots[k]); release(sc.hdr[k]); }
		release(sc.cost_sorted); release(sc.iota); release(sc.tmp);
	}
	if (ctx->d_counters) (void)hipFree(ctx->d_counters);
	if (ctx->build_arena.ptr) (void)hipFree(ctx->build_arena.ptr);
	if (ctx->build_arena.pinned) (void)hipHostFree(ctx->build_arena.pinned);
	if (ctx->h_auto) (void)hipHostFree(ctx->h_auto);
	if (ctx->h_small_in) (void)hipHostFree(ctx->h_small_in);
	if (ctx->h_small_out) (void)hipHostFree(ctx->h_small_out);
	for (auto &e : ctx->ev) if (e) (void)hipEventDestroy(e);
	for (auto &e : ctx->pipe_ev) (void)hipEventDestroy(e);
	if (ctx->up_stream) (void)hipStreamDestroy(ctx->up_stream);
	if (ctx->dn_stream) (void)hipStreamDestroy(ctx->dn_stream);
	if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
	delete ctx;
}

int mrt_set_stream(mrt_ctx *ctx, void *hip_stream)
{
	if (!ctx) return MRT_ERR_INVALID;
	if (ctx->pending) return fail(ctx, MRT_ERR_PENDING, "cannot switch streams with a dispatch pending");
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
	return MRT_OK;
}

int mrt_synchronize(mrt_ctx *ctx)
{
	if (!ctx) return MRT_ERR_INVALID;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MRT_OK;
}

int mrt_upload_scene(mrt_ctx *ctx, const mrt_tri64 *tris, uint32_t n_tris,
		const mrt_bvh_node32 *nodes, uint32_t used_nodes, const uint32_t *prim_idx)
{
	if (!ctx) return MRT_ERR_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = drain_pending(ctx); // gpu_ray_caster.cpp:198-202
	if (rc) return rc;
	mrt::DeviceSceneHost h;
	h.want8 = ctx->opts.kernel == MRT_KERNEL_LANE8_PERSISTENT || ctx->opts.kernel == MRT_KERNEL_AUTO;
	rc = mrt::prepare_scene(tris, n_tris, nodes, used_nodes, prim_idx, &h, ctx->err, sizeof(ctx->err));
	if (rc) return rc;
	auto cleanup = [&] { std::free(h.nodes); std::free(h.nodes4); std::free(h.nodes8); std::free(h.leaf_box); std::free(h.hot); std::free(h.cold); };
	if (h.depth > 64 || h.stack4 > 128) { cleanup(); return fail(ctx, MRT_ERR_UNSUPPORTED, "BVH deeper than the 64-entry traversal stack"); }
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	free_scene(ctx);
	hipError_t e;
	// the 4-wide layout is resident only when a kernel that walks it is asked for
	const bool want4 = ctx->opts.kernel == MRT_KERNEL_LANE4_PERSISTENT || ctx->opts.kernel == MRT_KERNEL_PACKET_QUAD || ctx->opts.kernel == MRT_KERNEL_AUTO;
	if ((e = hipMalloc(&ctx->d_nodes, (size_t)h.n_nodes * sizeof(mrt::DevNode))) != hipSuccess ||
			// +16 B of slack: a 64-B scalar fetch at the last 48-B triangle stays inside the allocation
			(e = hipMalloc(&ctx->d_hot, (size_t)h.n_tris * sizeof(mrt::TriHot) + 16)) != hipSuccess ||
			(e = hipMalloc(&ctx->d_cold, (size_t)h.n_tris * sizeof(mrt::TriCold))) != hipSuccess ||
			(want4 && (e = hipMalloc(&ctx->d_nodes4, (size_t)h.n_nodes4 * sizeof(mrt::Dev4Node))) != hipSuccess) ||
			(h.nodes8 && ((e = hipMalloc(&ctx->d_nodes8, (size_t)h.n_nodes8 * sizeof(mrt::Dev8Node))) != hipSuccess ||
					(e = hipMalloc(&ctx->d_leaf_box, (size_t)h.n_tris * 32)) != hipSuccess))) {
		cleanup(); free_scene(ctx);
		return fail(ctx, MRT_ERR_OOM, "scene does not fit in device memory");
	}
	e = hipMemcpy(ctx->d_nodes, h.nodes, (size_t)h.n_nodes * sizeof(mrt::DevNode), hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(ctx->d_hot, h.hot, (size_t)h.n_tris * sizeof(mrt::TriHot), hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(ctx->d_cold, h.cold, (size_t)h.n_tris * sizeof(mrt::TriCold), hipMemcpyHostToDevice);
	if (e == hipSuccess && want4) e = hipMemcpy(ctx->d_nodes4, h.nodes4, (size_t)h.n_nodes4 * sizeof(mrt::Dev4Node), hipMemcpyHostToDevice);
	if (e == hipSuccess && h.nodes8) e = hipMemcpy(ctx->d_nodes8, h.nodes8, (size_t)h.n_nodes8 * sizeof(mrt::Dev8Node), hipMemcpyHostToDevice);
	if (e == hipSuccess && h.nodes8) e = hipMemcpy(ctx->d_leaf_box, h.leaf_box, (size_t)h.n_tris * 32, hipMemcpyHostToDevice);
	ctx->n_nodes4 = want4 ? h.n_nodes4 : 0;
	ctx->n_nodes8 = h.nodes8 ? h.n_nodes8 : 0; ctx->stack8 = h.stack8;
	for (int c = 0; c < 3; c++) { ctx->bounds_lo[c] = h.bounds_lo[c]; ctx->bounds_hi[c] = h.bounds_hi[c]; }
	cleanup();
	if (e != hipSuccess) { free_scene(ctx); std::snprintf(ctx->err, sizeof(ctx->err), "scene upload failed: %s", hipGetErrorString(e)); return MRT_ERR_HIP; }
	ctx->n_nodes = h.n_nodes; ctx->n_tris = h.n_tris; ctx->depth = h.depth; ctx->stack4 = h.stack4;
	// LDS stack entries per lane: what this BVH can need, rounded up to 8, at most 64.
	ctx->stack_depth = ((h.depth + 7u) / 8u) * 8u;
	if (ctx->stack_depth < 8) ctx->stack_depth = 8;
	if ((rc = build_rows(ctx))) { free_scene(ctx); return rc; }
	ctx->scene = true;
	return MRT_OK;
}

int mrt_build_scene_device(mrt_ctx *ctx, const mrt_tri64 *tris, uint32_t n_tris, uint32_t flags)
{
	if (!ctx) return MRT_ERR_INVALID;
	if (!tris || n_tris == 0) return fail(ctx, MRT_ERR_INVALID, "build_scene_device: no triangles");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = drain_pending(ctx);
	if (rc) return rc;
	const bool on_device = (flags & MRT_BUILD_TRIS_ON_DEVICE) != 0;
	if (n_tris < 2) {
		// a one-triangle scene has no radix tree: the host path wraps the root leaf (scene_prep.cpp)
		mrt_tri64 t;
		if (on_device) HIP_TRY(ctx, hipMemcpy(&t, tris, sizeof(t), hipMemcpyDeviceToHost)); else t = tris[0];
		float verts[12] = { t.v0[0], t.v0[1], t.v0[2], 0.0f, t.v0[0] + t.edge1[0], t.v0[1] + t.edge1[1], t.v0[2] + t.edge1[2], 0.0f,
			t.v0[0] + t.edge2[0], t.v0[1] + t.edge2[1], t.v0[2] + t.edge2[2], 0.0f };
		mrt_bvh_node32 nodes[2]; uint32_t prim = 0, used = 0;
		if ((rc = mrt_bvh2_build(verts, 1, nodes, &prim, &used, 1))) return fail(ctx, rc, "build_scene_device: host build failed");
		return mrt_upload_scene(ctx, &t, 1, nodes, used, &prim);
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	hipEvent_t e0 = ctx->ev[0], e1 = ctx->ev[1];
	HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
	const mrt_tri64 *d_tris = tris;
	void *staged = nullptr;
	if (!on_device) {
		if (hipMalloc(&staged, (size_t)n_tris * sizeof(mrt_tri64)) != hipSuccess) return fail(ctx, MRT_ERR_OOM, "scene does not fit in device memory");
		hipError_t e = hipMemcpyAsync(staged, tris, (size_t)n_tris * sizeof(mrt_tri64), hipMemcpyHostToDevice, ctx->stream);
		if (e != hipSuccess) { (void)hipFree(staged); return fail(ctx, MRT_ERR_HIP, hipGetErrorString(e)); }
		d_tris = (const mrt_tri64 *)staged;
	}
	mrt::DeviceBuildResult b;
	const bool want4 = ctx->opts.kernel == MRT_KERNEL_LANE4_PERSISTENT || ctx->opts.kernel == MRT_KERNEL_PACKET_QUAD || ctx->opts.kernel == MRT_KERNEL_AUTO;
	const bool want8 = ctx->opts.kernel == MRT_KERNEL_LANE8_PERSISTENT || ctx->opts.kernel == MRT_KERNEL_AUTO;
	rc = mrt::device_build_lbvh(d_tris, n_tris, want4, want8, (flags & MRT_BUILD_SAFE_HANDOFF) != 0, (flags & MRT_BUILD_SAH) ? 2 : (flags & MRT_BUILD_PLOC) ? 1 : 0, &ctx->build_arena, (void *)ctx->stream, &b,
			ctx->err, sizeof(ctx->err));
	if (staged) (void)hipFree(staged);
	if (rc) return rc;
	auto drop_build = [&] { // the build's arrays are ours until the context takes them over below
		(void)hipFree(b.nodes); (void)hipFree(b.hot); (void)hipFree(b.cold);
		if (b.nodes4) (void)hipFree(b.nodes4);
		if (b.nodes8) (void)hipFree(b.nodes8);
		if (b.leaf_box) (void)hipFree(b.leaf_box);
	};
	if (b.depth > 64) { // the packet kernels keep 64 stack entries per wave
		drop_build();
		return fail(ctx, MRT_ERR_UNSUPPORTED, "device-built BVH deeper than the 64-entry traversal stack: build on the host");
	}
	float ms = 0.0f;
	hipError_t te = hipEventRecord(e1, ctx->stream);
	if (te == hipSuccess) te = hipEventSynchronize(e1);
	if (te == hipSuccess) te = hipEventElapsedTime(&ms, e0, e1);
	if (te != hipSuccess) { drop_build(); return fail(ctx, MRT_ERR_HIP, hipGetErrorString(te)); }
	free_scene(ctx);
	ctx->d_nodes = b.nodes; ctx->d_hot = b.hot; ctx->d_cold = b.cold;
	ctx->d_nodes4 = b.nodes4; ctx->n_nodes4 = b.nodes4 ? b.n_nodes : 0; ctx->stack4 = b.stack4;
	ctx->d_nodes8 = b.nodes8; ctx->d_leaf_box = b.leaf_box; ctx->n_nodes8 = b.nodes8 ? b.n_nodes : 0; ctx->stack8 = b.stack8;
	ctx->n_nodes = b.n_nodes; ctx->n_tris = b.n_tris; ctx->depth = b.depth;
	for (int c = 0; c < 3; c++) { ctx->bounds_lo[c] = b.bounds_lo[c]; ctx->bounds_hi[c] = b.bounds_hi[c]; }
	ctx->stack_depth = ((b.depth + 7u) / 8u) * 8u;
	if (ctx->stack_depth < 8) ctx->stack_depth = 8;
	ctx->stats.last_build_ms = ms;
	if ((rc = build_rows(ctx))) { free_scene(ctx); return rc; }
	ctx->scene = true;
	return MRT_OK;
}

// validates the instances, uploads what is on the host, flattens into d_out (device); *total = sum(n_tris)
static int flatten_instances(mrt_ctx *ctx, const float *verts9, uint32_t n_mesh_tris, const mrt_instance *instances,
		uint32_t n_instances, uint32_t flags, mrt_tri64 *d_out_or_null, mrt_tri64 **d_out_alloc, uint32_t *total)
{
	if (!verts9 || !instances || n_instances == 0) return fail(ctx, MRT_ERR_INVALID, "flatten_instances: no instances");
	std::vector<uint32_t> first(n_instances);
	uint64_t sum = 0; uint32_t max_tris = 0;
	for (uint32_t i = 0; i < n_instances; i++) {
		const mrt_instance &in = instances[i];
		if ((uint64_t)in.first_tri + in.n_tris > n_mesh_tris) return fail(ctx, MRT_ERR_INVALID, "flatten_instances: instance outside the mesh array");
		first[i] = (uint32_t)sum; sum += in.n_tris;
		if (in.n_tris > max_tris) max_tris = in.n_tris;
	}
	if (sum == 0 || sum > 0x7FFFFFFFull) return fail(ctx, MRT_ERR_INVALID, "flatten_instances: triangle count out of range");
	*total = (uint32_t)sum;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	void *d_inst = nullptr, *d_first = nullptr, *d_verts = nullptr, *d_out = d_out_or_null;
	auto drop = [&] {
		if (d_inst) (void)hipFree(d_inst);
		if (d_first) (void)hipFree(d_first);
		if (d_verts) (void)hipFree(d_verts);
	};
	const bool on_device = (flags & MRT_BUILD_TRIS_ON_DEVICE) != 0;
	hipError_t e = hipMalloc(&d_inst, (size_t)n_instances * sizeof(mrt_instance));
	if (e == hipSuccess) e = hipMalloc(&d_first, (size_t)n_instances * 4);
	if (e == hipSuccess && !on_device) e = hipMalloc(&d_verts, (size_t)n_mesh_tris * 36);
	if (e == hipSuccess && !d_out) { e = hipMalloc(&d_out, (size_t)sum * sizeof(mrt_tri64)); if (e == hipSuccess) *d_out_alloc = (mrt_tri64 *)d_out; }
	if (e != hipSuccess) { drop(); return fail(ctx, MRT_ERR_OOM, "flatten_instances: out of device memory"); }
	e = hipMemcpyAsync(d_inst, instances, (size_t)n_instances * sizeof(mrt_instance), hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_first, first.data(), (size_t)n_instances * 4, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess && !on_device) e = hipMemcpyAsync(d_verts, verts9, (size_t)n_mesh_tris * 36, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = mrt::launch_flatten_instances(on_device ? verts9 : (const float *)d_verts, (const mrt_instance *)d_inst,
			(const uint32_t *)d_first, n_instances, max_tris, (mrt_tri64 *)d_out, (void *)ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream); // `first` and the staging buffers go out of scope
	drop();
	if (e != hipSuccess) {
		if (*d_out_alloc) { (void)hipFree(*d_out_alloc); *d_out_alloc = nullptr; }
		std::snprintf(ctx->err, sizeof(ctx->err), "flatten_instances failed: %s", hipGetErrorString(e));
		return MRT_ERR_HIP;
	}
	return MRT_OK;
}

int mrt_flatten_instances(mrt_ctx *ctx, const float *verts9, uint32_t n_mesh_tris, const mrt_instance *instances,
		uint32_t n_instances, uint32_t flags, mrt_tri64 *d_out)
{
	if (!ctx) return MRT_ERR_INVALID;
	if (!d_out) return fail(ctx, MRT_ERR_INVALID, "flatten_instances: null output");
	mrt_tri64 *unused = nullptr; uint32_t total = 0;
	return flatten_instances(ctx, verts9, n_mesh_tris, instances, n_instances, flags, d_out, &unused, &total);
}

int mrt_build_instanced_scene_device(mrt_ctx *ctx, const float *verts9, uint32_t n_mesh_tris, const mrt_instance *instances,
		uint32_t n_instances, uint32_t flags)
{
	if (!ctx) return MRT_ERR_INVALID;
	mrt_tri64 *d_world = nullptr; uint32_t total = 0;
	int rc = flatten_instances(ctx, verts9, n_mesh_tris, instances, n_instances, flags, nullptr, &d_world, &total);
	if (rc) return rc;
	rc = mrt_build_scene_device(ctx, d_world, total, MRT_BUILD_TRIS_ON_DEVICE | (flags & (MRT_BUILD_PLOC | MRT_BUILD_SAH | MRT_BUILD_SAFE_HANDOFF)));
	(void)hipFree(d_world);
	return rc;
}

// Device-built BLASes for a two-level scene: every distinct mesh goes through device_build_lbvh on its own and
// is moved to its place in the scene's arrays (node refs + node_base, leaf slots + tri_base).  h comes from
// prepare_two_level(build_blas = false); on success the device arrays of ctx hold every BLAS and h knows
// their boxes and depths.
static int build_blases_on_device(mrt_ctx *ctx, mrt::TwoLevelHost *h, const float *verts9, int form)
{
	uint32_t max_tris = 0;
	for (uint32_t k = 0; k < h->n_blas; k++) if (h->blas[k].n_tris > max_tris) max_tris = h->blas[k].n_tris;
	mrt_tri64 *staged = nullptr;
	if (hipMalloc(&staged, (size_t)max_tris * sizeof(mrt_tri64)) != hipSuccess) return fail(ctx, MRT_ERR_OOM, "scene does not fit in device memory");
	std::vector<mrt_tri64> tris(max_tris);
	uint32_t tri_base = 0;
	int rc = MRT_OK;
	for (uint32_t k = 0; k < h->n_blas && rc == MRT_OK; k++) {
		mrt::TwoLevelBlas &bl = h->blas[k];
		rc = mrt_make_triangles(verts9 + (size_t)9 * bl.first_tri, nullptr, nullptr, bl.n_tris, tris.data()); // mesh-local ids, all layers
		if (rc) { fail(ctx, rc, "two-level scene: bad mesh triangles"); break; }
		hipError_t e = hipMemcpy(staged, tris.data(), (size_t)bl.n_tris * sizeof(mrt_tri64), hipMemcpyHostToDevice);
		if (e != hipSuccess) { rc = fail(ctx, MRT_ERR_HIP, hipGetErrorString(e)); break; }
		mrt::DeviceBuildResult b;
		rc = mrt::device_build_lbvh(staged, bl.n_tris, false, h->wide8, false, form, &ctx->build_arena, (void *)ctx->stream, &b, ctx->err, sizeof(ctx->err));
		if (rc) break;
		e = mrt::launch_offset_refs(ctx->d_nodes + bl.root, b.nodes, b.n_nodes, bl.root, tri_base, (void *)ctx->stream);
		if (h->wide8 && !(b.nodes8 && b.leaf_box)) h->wide8 = false; // a mesh whose boxes fit no grid: the scene goes without the 8-wide layout
		if (h->wide8) {
			if (e == hipSuccess) e = mrt::launch_offset_refs8(ctx->d_nodes8 + bl.root8, b.nodes8, b.n_nodes, bl.root8, tri_base, (void *)ctx->stream);
			if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_leaf_box + (size_t)tri_base * 8, b.leaf_box, (size_t)bl.n_tris * 32, hipMemcpyDeviceToDevice, ctx->stream);
			bl.stack8 = b.stack8;
		}
		if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_hot + tri_base, b.hot, (size_t)bl.n_tris * sizeof(mrt::TriHot), hipMemcpyDeviceToDevice, ctx->stream);
		if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_cold + tri_base, b.cold, (size_t)bl.n_tris * sizeof(mrt::TriCold), hipMemcpyDeviceToDevice, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		(void)hipFree(b.nodes); (void)hipFree(b.hot); (void)hipFree(b.cold);
		if (b.nodes8) (void)hipFree(b.nodes8);
		if (b.leaf_box) (void)hipFree(b.leaf_box);
		if (e != hipSuccess) { rc = fail(ctx, MRT_ERR_HIP, hipGetErrorString(e)); break; }
		// (the scene's node array holds n_tris - 1 rows per mesh: what the radix tree fills; the SAH form, with leaves of several triangles, fewer)
		if (b.n_nodes == 0u || b.n_nodes > bl.n_tris - 1u) { rc = fail(ctx, MRT_ERR_BAD_BVH, "two-level scene: unexpected BLAS size"); break; }
		bl.depth = b.depth;
		for (int c = 0; c < 3; c++) { bl.lo[c] = b.bounds_lo[c]; bl.hi[c] = b.bounds_hi[c]; }
		tri_base += bl.n_tris;
	}
	(void)hipFree(staged);
	return rc;
}

// SceneTLAS::build_tlas + every MeshBLAS::build (scene_tlas.h:140-176, mesh_blas.h:86-138): nothing is flattened
int mrt_upload_two_level_scene(mrt_ctx *ctx, const float *verts9, uint32_t n_mesh_tris, const mrt_instance *instances,
		uint32_t n_instances, uint32_t flags)
{
	if (!ctx) return MRT_ERR_INVALID;
	if (!verts9 || !instances || n_instances == 0 || n_mesh_tris == 0) return fail(ctx, MRT_ERR_INVALID, "two-level scene: null or empty argument");
	if (flags & ~(uint32_t)(MRT_BUILD_BLAS_ON_DEVICE | MRT_BUILD_SAH)) return fail(ctx, MRT_ERR_INVALID, "two-level scene: unknown flag");
	if ((flags & MRT_BUILD_SAH) && !(flags & MRT_BUILD_BLAS_ON_DEVICE)) return fail(ctx, MRT_ERR_INVALID, "two-level scene: MRT_BUILD_SAH goes with MRT_BUILD_BLAS_ON_DEVICE (the host builder's trees are SAH trees)");
	const bool on_device = (flags & MRT_BUILD_BLAS_ON_DEVICE) != 0;
	int rc = drain_pending(ctx);
	if (rc) return rc;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	mrt::TwoLevelHost *h = new (std::nothrow) mrt::TwoLevelHost();
	if (!h) return fail(ctx, MRT_ERR_OOM, "two-level scene: out of host memory");
	auto drop = [&] { mrt::free_two_level(h); delete h; };
	unsigned n_thr = std::thread::hardware_concurrency();
	rc = mrt::prepare_two_level(verts9, n_mesh_tris, instances, n_instances, n_thr ? n_thr : 1u, !on_device, h, ctx->err, sizeof(ctx->err));
	if (rc) { drop(); return rc; }
	if (!on_device && h->depth > 64u) { drop(); return fail(ctx, MRT_ERR_UNSUPPORTED, "two-level scene: trees too deep for the traversal stack"); }
	hipError_t e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess) { drop(); return fail(ctx, MRT_ERR_HIP, hipGetErrorString(e)); }
	free_scene(ctx);
	if ((e = hipMalloc(&ctx->d_nodes, (size_t)h->n_nodes * sizeof(mrt::DevNode))) != hipSuccess ||
			(e = hipMalloc(&ctx->d_hot, (size_t)h->n_tris * sizeof(mrt::TriHot) + 16)) != hipSuccess ||
			(e = hipMalloc(&ctx->d_cold, (size_t)h->n_tris * sizeof(mrt::TriCold))) != hipSuccess ||
			(e = hipMalloc(&ctx->d_instances, (size_t)h->n_inst * sizeof(mrt::DevInstance))) != hipSuccess) {
		drop(); free_scene(ctx);
		return fail(ctx, MRT_ERR_OOM, "scene does not fit in device memory");
	}
	if (on_device) { h->wide8 = true; h->n_nodes8 = h->n_nodes - h->tlas_cap; } // wanted; build_blases_on_device takes it back if a mesh has none
	if (h->wide8 && (hipMalloc(&ctx->d_nodes8, (size_t)h->n_nodes8 * sizeof(mrt::Dev8Node)) != hipSuccess ||
			hipMalloc(&ctx->d_leaf_box, (size_t)h->n_tris * 32) != hipSuccess)) { // an optional layout: go without it
		if (ctx->d_nodes8) (void)hipFree(ctx->d_nodes8);
		ctx->d_nodes8 = nullptr; ctx->d_leaf_box = nullptr; h->wide8 = false;
	}
	if (on_device) {
		hipEvent_t e0 = ctx->ev[0], e1 = ctx->ev[1]; // the context's own events: nothing to create or to leak here
		(void)hipEventRecord(e0, ctx->stream);
		rc = build_blases_on_device(ctx, h, verts9, (flags & MRT_BUILD_SAH) ? 2 : 0);
		if (!rc) rc = mrt::refit_two_level(h, instances, n_instances, ctx->err, sizeof(ctx->err));
		if (!rc && h->depth > 64u) rc = fail(ctx, MRT_ERR_UNSUPPORTED, "two-level scene: device-built trees too deep for the traversal stack: build on the host");
		if (rc) { drop(); free_scene(ctx); return rc; }
		e = hipMemcpy(ctx->d_nodes, h->nodes, (size_t)h->n_tlas_nodes * sizeof(mrt::DevNode), hipMemcpyHostToDevice);
		(void)hipEventRecord(e1, ctx->stream);
		(void)hipEventSynchronize(e1);
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ctx->stats.last_build_ms = ms;
	} else {
		e = hipMemcpy(ctx->d_nodes, h->nodes, (size_t)h->n_nodes * sizeof(mrt::DevNode), hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMemcpy(ctx->d_hot, h->hot, (size_t)h->n_tris * sizeof(mrt::TriHot), hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMemcpy(ctx->d_cold, h->cold, (size_t)h->n_tris * sizeof(mrt::TriCold), hipMemcpyHostToDevice);
		if (e == hipSuccess && h->wide8) e = hipMemcpy(ctx->d_nodes8, h->nodes8, (size_t)h->n_nodes8 * sizeof(mrt::Dev8Node), hipMemcpyHostToDevice);
		if (e == hipSuccess && h->wide8) e = hipMemcpy(ctx->d_leaf_box, h->leaf_box, (size_t)h->n_tris * 32, hipMemcpyHostToDevice);
	}
	if (!h->wide8 && ctx->d_nodes8) { // taken back during the device builds
		(void)hipFree(ctx->d_nodes8); (void)hipFree(ctx->d_leaf_box);
		ctx->d_nodes8 = nullptr; ctx->d_leaf_box = nullptr;
	}
	if (e == hipSuccess) e = hipMemcpy(ctx->d_instances, h->inst, (size_t)h->n_inst * sizeof(mrt::DevInstance), hipMemcpyHostToDevice);
	if (e != hipSuccess) { drop(); free_scene(ctx); return fail(ctx, MRT_ERR_HIP, hipGetErrorString(e)); }
	std::free(h->hot); std::free(h->cold); h->hot = nullptr; h->cold = nullptr; // the device has them; a refit only needs nodes + instances
	std::free(h->nodes8); std::free(h->leaf_box); h->nodes8 = nullptr; h->leaf_box = nullptr;
	ctx->n_nodes8 = h->wide8 ? h->n_nodes8 : 0; ctx->stack8 = h->wide8 ? h->depth8 : 0;
	ctx->two_level = h;
	ctx->n_nodes = h->n_nodes; ctx->n_tris = h->n_tris; ctx->depth = h->depth;
	ctx->stack_depth = ((h->depth + 7u) / 8u) * 8u;
	if (ctx->stack_depth < 8) ctx->stack_depth = 8;
	ctx->n_nodes4 = 0; ctx->stack4 = 0;
	// sort keys (origin Morton) are quantised on the scene box: the union of the TLAS root's children
	for (int c = 0; c < 3; c++) {
		ctx->bounds_lo[c] = std::fmin(h->nodes[0].lmin[c], h->nodes[0].rmin[c]);
		ctx->bounds_hi[c] = std::fmax(h->nodes[0].lmax[c], h->nodes[0].rmax[c]);
	}
	ctx->scene = true;
	return MRT_OK;
}

// SceneTLAS::set_instance_transform + refit_tlas (scene_tlas.h:118-134,178-196): instances moved, meshes unchanged
int mrt_update_instances(mrt_ctx *ctx, const mrt_instance *instances, uint32_t n_instances)
{
	if (!ctx) return MRT_ERR_INVALID;
	if (!ctx->scene || !ctx->two_level) return fail(ctx, MRT_ERR_NO_SCENE, "no two-level scene uploaded");
	if (!instances) return fail(ctx, MRT_ERR_INVALID, "null instances");
	int rc = drain_pending(ctx);
	if (rc) return rc;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	mrt::TwoLevelHost *h = ctx->two_level;
	if ((rc = mrt::refit_two_level(h, instances, n_instances, ctx->err, sizeof(ctx->err)))) return rc;
	if (h->depth > 64u) return fail(ctx, MRT_ERR_UNSUPPORTED, "two-level scene: trees too deep for the per-lane stack");
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	HIP_TRY(ctx, hipMemcpy(ctx->d_nodes, h->nodes, (size_t)h->n_tlas_nodes * sizeof(mrt::DevNode), hipMemcpyHostToDevice));
	HIP_TRY(ctx, hipMemcpy(ctx->d_instances, h->inst, (size_t)h->n_inst * sizeof(mrt::DevInstance), hipMemcpyHostToDevice));
	ctx->depth = h->depth; ctx->stack8 = h->wide8 ? h->depth8 : 0;
	ctx->stack_depth = ((h->depth + 7u) / 8u) * 8u;
	if (ctx->stack_depth < 8) ctx->stack_depth = 8;
	for (int c = 0; c < 3; c++) {
		ctx->bounds_lo[c] = std::fmin(h->nodes[0].lmin[c], h->nodes[0].rmin[c]);
		ctx->bounds_hi[c] = std::fmax(h->nodes[0].lmax[c], h->nodes[0].rmax[c]);
	}
	return MRT_OK;
}

int mrt_is_available(const mrt_ctx *ctx) { return ctx && ctx->scene ? 1 : 0; }

int mrt_scene_info(const mrt_ctx *ctx, uint32_t *n_tris, uint32_t *n_wide_nodes, uint32_t *bvh_depth)
{
	if (!ctx || !ctx->scene) return MRT_ERR_NO_SCENE;
	if (n_tris) *n_tris = ctx->n_tris;
	if (n_wide_nodes) *n_wide_nodes = ctx->n_nodes;
	if (bvh_depth) *bvh_depth = ctx->depth;
	return MRT_OK;
}

int mrt_cast(mrt_ctx *ctx, const void *rays, void *hits, uint64_t count, uint32_t query_mask, int mode, uint32_t flags)
{
	if (!ctx) return MRT_ERR_INVALID;
	if (count == 0) return ctx->scene ? MRT_OK : fail(ctx, MRT_ERR_NO_SCENE, "no scene uploaded"); // cpp:419: silent no-op
	if (!rays || !hits) return fail(ctx, MRT_ERR_INVALID, "null rays / hits");
	if (ctx->pending) return fail(ctx, MRT_ERR_PENDING, "collect the pending dispatch first");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	void *d_hits = nullptr;
	const bool hits_dev = (flags & MRT_FLAG_HITS_ON_DEVICE) != 0;
	if ((flags & MRT_FLAG_ASYNC) && !(hits_dev && (flags & MRT_FLAG_RAYS_ON_DEVICE)))
		return fail(ctx, MRT_ERR_INVALID, "ASYNC needs device-resident rays and hits");
	if (!hits_dev && !(flags & MRT_FLAG_RAYS_ON_DEVICE) && count >= 2 * kPipeChunk && !ctx->opts.count_visits) {
		if (!ctx->scene) return fail(ctx, MRT_ERR_NO_SCENE, "no scene uploaded (is_available() == false)");
		return cast_host_pipelined(ctx, rays, hits, count, query_mask, mode, flags);
	}
	// Latency path: host arrays of at most kSmallCast rays skip both DMA copies: the kernel reads the rays from, and
	// writes the records to, pinned host memory mapped into the device (one ray: 85 -> 55 us per blocking call,
	// tools/bench_latency.py).  The same kernels, the same records.
	if (!hits_dev && !(flags & (MRT_FLAG_RAYS_ON_DEVICE | MRT_FLAG_ASYNC)) && count <= kSmallCast && !ctx->opts.count_visits) {
		const size_t rs_ = ray_stride(flags), hs_ = hit_stride(ctx, flags, mode);
		if (!ctx->h_small_in) {
			if (hipHostMalloc(&ctx->h_small_in, kSmallCast * 64, hipHostMallocMapped) != hipSuccess ||
					hipHostMalloc(&ctx->h_small_out, kSmallCast * 64, hipHostMallocMapped) != hipSuccess ||
					hipHostGetDevicePointer(&ctx->d_small_in, ctx->h_small_in, 0) != hipSuccess ||
					hipHostGetDevicePointer(&ctx->d_small_out, ctx->h_small_out, 0) != hipSuccess) {
				if (ctx->h_small_in) (void)hipHostFree(ctx->h_small_in);
				if (ctx->h_small_out) (void)hipHostFree(ctx->h_small_out);
				ctx->h_small_in = ctx->h_small_out = ctx->d_small_in = ctx->d_small_out = nullptr;
				(void)hipGetLastError();
			}
		}
		if (ctx->h_small_in) {
			std::memcpy(ctx->h_small_in, rays, count * rs_);
			int rc2 = enqueue_cast(ctx, ctx->d_small_in, ctx->d_small_out, count, query_mask, mode, flags | MRT_FLAG_RAYS_ON_DEVICE, &d_hits);
			if (rc2) return rc2;
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
			std::memcpy(hits, ctx->h_small_out, count * hs_);
			return finish_timing(ctx, false, ctx->stats.last_kernel_launches >= 2, false);
		}
	}
	int rc = enqueue_cast(ctx, rays, hits_dev ? hits : nullptr, count, query_mask, mode, flags, &d_hits);
	if (rc) return rc;
	if (flags & MRT_FLAG_ASYNC) { ctx->stats.last_kernel = 0; return MRT_OK; } // queued on the context's stream; no timing
	if (!hits_dev) {
		HIP_TRY(ctx, hipMemcpyAsync(hits, d_hits, count * hit_stride(ctx, flags, mode), hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	rc = finish_timing(ctx, !(flags & MRT_FLAG_RAYS_ON_DEVICE), ctx->stats.last_kernel_launches >= 2, !hits_dev);
	if (rc == MRT_OK) tune_record(ctx);
	return rc;
}

int mrt_submit(mrt_ctx *ctx, const void *rays, uint64_t count, uint32_t query_mask, int mode, uint32_t flags)
{
	if (!ctx) return MRT_ERR_INVALID;
	if (ctx->pending) return fail(ctx, MRT_ERR_PENDING, "submit while a dispatch is pending (gpu_ray_caster.cpp:538)");
	if (!ctx->scene) return fail(ctx, MRT_ERR_NO_SCENE, "no scene uploaded");
	if (count == 0) return MRT_OK;
	if (!rays) return fail(ctx, MRT_ERR_INVALID, "null rays");
	if (flags & (MRT_FLAG_HITS_ON_DEVICE | MRT_FLAG_ASYNC)) return fail(ctx, MRT_ERR_INVALID, "submit keeps results in the context; use mrt_cast for device outputs");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	void *d_hits = nullptr;
	int rc = enqueue_cast(ctx, rays, nullptr, count, query_mask, mode, flags, &d_hits);
	ctx->gs->tune.armed = false; // (collected later: no timing of this cast alone)
	if (rc) return rc;
	ctx->pending = true; ctx->pending_count = count; ctx->pending_flags = flags; ctx->pending_mode = mode;
	ctx->pending_dev_hits = d_hits;
	return MRT_OK;
}

int mrt_collect(mrt_ctx *ctx, void *hits, uint64_t count)
{
	if (!ctx) return MRT_ERR_INVALID;
	if (!ctx->pending) return fail(ctx, MRT_ERR_NOT_PENDING, "collect without a pending dispatch");
	if (!hits) return fail(ctx, MRT_ERR_INVALID, "null hits");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const uint64_t n = count < ctx->pending_count ? count : ctx->pending_count; // cpp:573
	HIP_TRY(ctx, hipMemcpyAsync(hits, ctx->pending_dev_hits, n * hit_stride(ctx, ctx->pending_flags, ctx->pending_mode),
			hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	ctx->pending = false;
	return finish_timing(ctx, !(ctx->pending_flags & MRT_FLAG_RAYS_ON_DEVICE), ctx->stats.last_kernel_launches >= 2, true);
}

int mrt_has_pending(const mrt_ctx *ctx) { return ctx && ctx->pending ? 1 : 0; }

// ---- frame-coherent tile schedule -------------------------------------------------------------------------------------
// A grid cast ends with its slowest wave: packets differ 25-fold in cost (27 .. 667 rows at C3), and at renderer sizes
// (1 - 2 M rays = 2 - 4 rounds of waves) the last round's long walks leave most of the chip idle: 1920x1080 ran at 3.8 Grays/s
// against 9 at 4096^2.  Which tiles are expensive barely changes from one frame to the next, so every wave notes the
// shader cycles its tile(s) took (TraceParams::tile_cost), a radix sort on a side stream turns that into a launch order,
// longest first, and the next cast of the same grid (same size and rows; any camera: the order is only a permutation,
// results never depend on it) launches in that order: the long walks start first and the short ones fill the gaps behind
// them (longest-processing-time-first).  The first cast of a grid runs in the plain order.  mrt_options.tile_schedule = 1
// turns it off.  For batches of 2^19 up to (not including) 2^24 rays: 1280x720 -17 %, 1920x1080 -18 %, 3840x2160 -13 %; at 4096^2
// and above the gain is 2-4 % in kernel time and less than what the bookkeeping costs a blocking call; 640x360 measured 7 %
// slower with it (too few tiles to reorder).  The order is renewed every kScheduleRenew-th frame, not every frame: which
// tiles are expensive changes slowly, and a sort that runs beside the start of the next frame delays exactly the long walks
// that frame launches first (1920x1080: 0.52 against 0.44 ms with a sort per frame).  A batch whose row width is found on the device
// (mrt_cast with MRT_FLAG_COHERENT) is scheduled from the width the previous cast of the same size found.
#ifndef MRT_SCHEDULE_MAX_LOG2
#define MRT_SCHEDULE_MAX_LOG2 24
#endif
constexpr uint64_t kScheduleMaxRays = 1ull << MRT_SCHEDULE_MAX_LOG2;
// From 2^17 rays (2 048 tiles: below, all tiles go in quarter tiles anyway, quarter_small_grid).  With the order alone 640x360
// measured 7 % slower scheduled than not; with the most expensive tiles in quarter tiles until the chip is full
// (schedule_plan_kernel) it is 20 % faster.  MRT_SCHEDULE_MIN_LOG2 moves the bound (the tests schedule smaller grids)
static uint64_t schedule_min_rays()
{
	const char *e = std::getenv("MRT_SCHEDULE_MIN_LOG2"); // (read per cast: a test sets it for its own contexts)
	const int k = e ? std::atoi(e) : 17;
	return 1ull << (k >= 12 && k <= 24 ? k : 17);
}
// the grid tuner: from 2^19 rays on the 128-ray walk is a candidate (below, the 64-ray kernel won every measurement); a test that
// moves the schedule's bound moves this one with it
static uint64_t tune_min_rays() { return std::getenv("MRT_SCHEDULE_MIN_LOG2") ? schedule_min_rays() : (1ull << 19); }
#define kTuneMinRays tune_min_rays()
#define kScheduleMinRays schedule_min_rays()
// The launch list of a generation: the sorted order, with the units whose cost says they would end the frame alone launched in
// pieces (TraceParams::tile_sched).  A frame of 1-2 M rays is one or two rounds of waves, so it lasts as long as its longest
// walk, and the cost arrays say the longest walks are few and far out: at 1920x1080 on the C3 scene one pair of tiles takes
// 1.4 M cycles, the 99th percentile 0.57 M, and all pairs together 0.68 M per wave slot.  A unit above BOTH the work per wave
// slot and the cost of rank n / 100 goes in quarter tiles (4x4 pixels in 16 lanes; eight of them for a pair: each takes
// 0.18 of the pair, all eight 1.5 x the pair); pieces first, so the longest things still start first.  Whether that pays
// depends on how many rounds of waves the frame is: 1920x1080 (two rounds of pairs) 0.60 -> 0.37 ms, 1280x960 (1.2 rounds)
// 0.323 -> 0.337 ms -- every extra wave pushes a whole unit into the second round, and the pieces' work is half again their
// unit's; a deeper cut (the work per wave slot alone as the bound) 0.378 ms, only far outliers (1.25 x the 99th percentile)
// nothing at 1280x960 and 0.43 ms at 1920x1080.  So the rule stays simple and the kernel tuner MEASURES it: a grid's frames
// 3-5 run the 128-ray walk with pieces, 6-8 without, and the faster way is kept (tune_grid_kernel).
// hdr = {units in quarters, units in single tiles (unused: 0), slots}.  MRT_SCHED_SPLIT_PCT: the rank, in percent (default 1).
constexpr uint32_t kWaveSlots = 256u * 4u * 8u; // wave slots of the device (CUs x SIMDs x waves): what a frame's work is spread over
__global__ __launch_bounds__(1024) void schedule_plan_kernel(const uint32_t *cost_sorted, uint32_t n_units, uint32_t unit, uint32_t n_extra, uint32_t rank, uint32_t *hdr)
{
	__shared__ unsigned long long part[16];
	unsigned long long sum = 0ull;
	for (uint32_t i = threadIdx.x; i < n_units; i += 1024u) sum += cost_sorted[i];
	for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
	if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = sum;
	__syncthreads();
	if (threadIdx.x != 0u) return;
	sum = 0ull;
	for (int w = 0; w < 16; w++) sum += part[w];
	unsigned long long thr = sum / kWaveSlots;
	if (rank < n_units && (unsigned long long)cost_sorted[rank] > thr) thr = cost_sorted[rank];
	// cost_sorted is descending: how many lie above the bound
	uint32_t lo = 0u, hi = n_units;
	while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((unsigned long long)cost_sorted[mid] > thr) lo = mid + 1u; else hi = mid; }
	const uint32_t per_quartered = unit == 2u ? 7u : 3u; // extra slots of a unit in quarters
	uint32_t quartered = sum == 0ull ? 0u : lo;
	// fewer units than wave slots: the frame is one round of waves and lasts as long as its longest walk; the most expensive units
	// go in quarters until the round is full (C3 scene, 64-ray kernel: 512^2 0.305 -> 0.248 ms, 640x360 0.284 -> 0.227, 960x540
	// 0.378 -> 0.309; filling to 1.25 or 1.5 rounds instead: 0.293 / 0.317 at 512^2)
	if (sum != 0ull && n_units < kWaveSlots && (kWaveSlots - n_units) / per_quartered > quartered) quartered = (kWaveSlots - n_units) / per_quartered;
	if (quartered > n_units) quartered = n_units;
	if ((unsigned long long)quartered * per_quartered > n_extra) quartered = n_extra / per_quartered;
	hdr[0] = quartered; hdr[1] = 0u; hdr[2] = n_units + quartered * per_quartered;
}
__global__ __launch_bounds__(256) void schedule_fill_kernel(const uint32_t *order, uint32_t n_units, uint32_t unit, const uint32_t *hdr, uint32_t *slots)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_units) return;
	const uint32_t quartered = hdr[0], halved = hdr[1], u = order[i], pieces = unit == 2u ? 8u : 4u;
	if (i < quartered) {
		for (uint32_t k = 0; k < pieces; k++) slots[i * pieces + k] = ((2u + (k & 3u)) << 28) | (u * unit + (k >> 2));
	} else if (i < quartered + halved) {
		const uint32_t at = quartered * pieces + (i - quartered) * 2u;
		slots[at] = (1u << 28) | (u * 2u); slots[at + 1u] = (1u << 28) | (u * 2u + 1u);
	} else slots[quartered * pieces + halved * 2u + (i - quartered - halved)] = u;
}

static bool schedule_applies(const mrt_ctx *ctx, const mrt::TraceParams &p)
{
	if (ctx->opts.tile_schedule == 1u || ctx->opts.count_visits) return false;
	if ((p.lane_map != mrt::MAP_TILE8X8 && p.lane_map != mrt::MAP_AUTO) || p.count < kScheduleMinRays || p.count >= kScheduleMaxRays) return false;
	return p.kernel == MRT_KERNEL_PACKET_ASM || (p.kernel == MRT_KERNEL_PACKET_DUAL && p.row_array != nullptr);
}

// Before the launch: the units of this grid, the newest finished order of the same grid, and -- on a measuring frame -- a
// zeroed cost array.  Two generations of (cost, order): generation g is sorted on the side stream while later frames already
// run in the order of generation g - 1; no frame waits for a running sort.
constexpr uint32_t kScheduleRenew = 8;
static int schedule_grid(mrt_ctx *ctx, mrt::TraceParams &p)
{
	auto &s = ctx->gs->sched;
	const uint32_t th = 64u >> p.tile_w_log2;
	const uint32_t tiles_y = (p.rows + th - 1u) / th;
	const uint32_t unit = p.kernel == MRT_KERNEL_PACKET_DUAL ? 2u : 1u;
	const uint64_t tiles = (uint64_t)p.tiles_x * tiles_y;
	const uint32_t n_units = (uint32_t)((tiles + unit - 1u) / unit);
	// pieces: 8x8 tiles only, ids within the entry's 28 bits, not while the kernel tuner tries (or has chosen) the frames without
	const bool pieces = p.tile_w_log2 == 3u && ctx->opts.tile_schedule != 2u && tiles < (1ull << 28) && !ctx->gs->tune.no_pieces;
	const bool same = s.grid_w == p.grid_w && s.grid_h == p.grid_h && s.y0 == p.y0 && s.rows == p.rows && s.unit == unit &&
			s.n_units == n_units && s.tile_w_log2 == p.tile_w_log2 && s.pieces == pieces;
	int rc;
	if (!s.side) {
		HIP_TRY(ctx, hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking));
		HIP_TRY(ctx, hipEventCreateWithFlags(&s.traced, hipEventDisableTiming));
		for (int k = 0; k < 2; k++) HIP_TRY(ctx, hipEventCreateWithFlags(&s.ready[k], hipEventDisableTiming));
	}
	if (!same) {
		HIP_TRY(ctx, hipStreamSynchronize(s.side)); // no sort of the old grid may still use the arrays
		// room for pieces: half as many extra slots as there are units, or what fills one round of waves (schedule_plan_kernel)
		s.n_slots_max = pieces ? (n_units + n_units / 2u > kWaveSlots ? n_units + n_units / 2u : kWaveSlots) : n_units;
		for (int k = 0; k < 2; k++)
			if ((rc = ensure(ctx, s.cost[k], ((size_t)n_units + s.n_slots_max) * 4)) || (rc = ensure(ctx, s.order[k], (size_t)n_units * 4)) ||
					(rc = ensure(ctx, s.slots[k], (size_t)s.n_slots_max * 4)) || (rc = ensure(ctx, s.hdr[k], 16))) return rc;
		if ((rc = ensure(ctx, s.cost_sorted, (size_t)n_units * 4)) || (rc = ensure(ctx, s.iota, (size_t)n_units * 4))) return rc;
		std::vector<uint32_t> iota(n_units);
		for (uint32_t i = 0; i < n_units; i++) iota[i] = i;
		HIP_TRY(ctx, hipMemcpyAsync(s.iota.ptr, iota.data(), (size_t)n_units * 4, hipMemcpyHostToDevice, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // (the host vector goes out of scope)
		s.forget();
		s.frame = 0; s.gen = 0;
	}
	s.grid_w = p.grid_w; s.grid_h = p.grid_h; s.y0 = p.y0; s.rows = p.rows; s.unit = unit; s.n_units = n_units; s.tile_w_log2 = p.tile_w_log2; s.pieces = pieces;
	const uint32_t cur = s.gen & 1u, newest = cur ^ 1u;     // the slot the next generation goes to, the slot of the last one
	// the order to launch in: the last generation's if its sort is done, else the one before (still intact in slot `cur`:
	// that slot's ORDER array is rewritten only by the next sort, which runs after this frame's trace)
	// the frame the kernel tuner times (the third of a kernel) waits for the sorts behind it: it is launched the way later frames will be
	if (ctx->gs->tune.armed && ctx->gs->tune.phase % kTuneFrames >= kTuneFrames - 2) HIP_TRY(ctx, hipStreamSynchronize(s.side));
	const uint32_t *order = nullptr, *hdr = nullptr;
	if (s.have_order[newest] && hipEventQuery(s.ready[newest]) == hipSuccess) { order = (const uint32_t *)s.slots[newest].ptr; hdr = (const uint32_t *)s.hdr[newest].ptr; }
	(void)hipGetLastError(); // (hipErrorNotReady is not an error)
	if (!order && s.have_order[cur] && hipEventQuery(s.ready[cur]) == hipSuccess) { order = (const uint32_t *)s.slots[cur].ptr; hdr = (const uint32_t *)s.hdr[cur].ptr; }
	(void)hipGetLastError();
	// a measuring frame: the first two of a grid, then every kScheduleRenew-th -- if the slot's previous sort is done
	s.measuring = (s.gen < 2u || s.frame % kScheduleRenew == 0u) && (!s.have_order[cur] || hipEventQuery(s.ready[cur]) == hipSuccess);
	(void)hipGetLastError();
	if (s.measuring) HIP_TRY(ctx, hipMemsetAsync(s.cost[cur].ptr, 0, (size_t)n_units * 4, ctx->stream));
	p.tile_sched = order; p.sched_hdr = hdr; p.n_slots_max = order ? s.n_slots_max : 0u;
	p.tile_cost = s.measuring ? (uint32_t *)s.cost[cur].ptr : nullptr;
	p.tile_unit = unit; p.n_units = n_units;
	return MRT_OK;
}

// After the launch (ev[4] recorded on the context's stream): on a measuring frame, sort its units by cost, descending, on the
// side stream.
static int schedule_sort(mrt_ctx *ctx)
{
	auto &s = ctx->gs->sched;
	s.frame++;
	if (!s.measuring) return MRT_OK;
	const uint32_t cur = s.gen & 1u;
	HIP_TRY(ctx, hipEventRecord(s.traced, ctx->stream));
	HIP_TRY(ctx, hipStreamWaitEvent(s.side, s.traced, 0));
	size_t tmp_bytes = 0;
	uint32_t *ki = (uint32_t *)s.cost[cur].ptr, *ko = (uint32_t *)s.cost_sorted.ptr, *vi = (uint32_t *)s.iota.ptr, *vo = (uint32_t *)s.order[cur].ptr;
	HIP_TRY(ctx, rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, ki, ko, vi, vo, (size_t)s.n_units, 0, 32, s.side));
	int rc;
	if (s.tmp.cap < tmp_bytes) { HIP_TRY(ctx, hipStreamSynchronize(s.side)); if ((rc = ensure(ctx, s.tmp, tmp_bytes))) return rc; }
	HIP_TRY(ctx, rocprim::radix_sort_pairs_desc(s.tmp.ptr, tmp_bytes, ki, ko, vi, vo, (size_t)s.n_units, 0, 32, s.side));
	uint32_t split_pct = 1u;
	if (const char *e = std::getenv("MRT_SCHED_SPLIT_PCT")) { const int v = std::atoi(e); if (v >= 0 && v <= 50) split_pct = (uint32_t)v; } // tuning knob (tools/bench_resolutions.py)
	hipLaunchKernelGGL(schedule_plan_kernel, dim3(1), dim3(1024), 0, s.side, ko, s.n_units, s.unit, s.n_slots_max - s.n_units, (uint32_t)((uint64_t)s.n_units * split_pct / 100u),
			(uint32_t *)s.hdr[cur].ptr);
	hipLaunchKernelGGL(schedule_fill_kernel, dim3((s.n_units + 255u) / 256u), dim3(256), 0, s.side, vo, s.n_units, s.unit, (const uint32_t *)s.hdr[cur].ptr, (uint32_t *)s.slots[cur].ptr);
	HIP_TRY(ctx, hipGetLastError());
	HIP_TRY(ctx, hipEventRecord(s.ready[cur], s.side));
	s.have_order[cur] = true;
	s.gen++;
	if (std::getenv("MRT_SCHED_DUMP")) { // diagnosis: what the schedule was made of (tools/bench_resolutions.py with MRT_SCHED_DUMP=1)
		std::vector<uint32_t> c(s.n_units);
		uint32_t hdr[3] = {0, 0, 0};
		HIP_TRY(ctx, hipStreamSynchronize(s.side));
		HIP_TRY(ctx, hipMemcpy(c.data(), s.cost_sorted.ptr, (size_t)s.n_units * 4, hipMemcpyDeviceToHost));
		HIP_TRY(ctx, hipMemcpy(hdr, s.hdr[cur].ptr, sizeof(hdr), hipMemcpyDeviceToHost));
		unsigned long long sum = 0; for (uint32_t v : c) sum += v;
		std::fprintf(stderr, "[mrt schedule] %ux%u unit %u: %u units, cycles sum %llu, max %u, p99 %u, median %u, min %u; next launch: %u units in quarter tiles, %u slots\n", s.grid_w, s.rows, s.unit,
				s.n_units, sum, c.empty() ? 0u : c[0], c.empty() ? 0u : c[s.n_units / 100], c.empty() ? 0u : c[s.n_units / 2], c.empty() ? 0u : c[s.n_units - 1], hdr[0], hdr[2]);
	}
	return MRT_OK;
}

// Small grids of known width (mrt_cast_grid, mrt_cast_tiled; flat scenes, MRT_KERNEL_AUTO; mrt_cast(COHERENT) does the same for a
// width found on the device): the 64-ray packet kernel with EVERY tile launched in pieces (TraceParams::quarter_all) -- up to 512
// tiles as its sixteen 2x2-pixel sixteenths (4 rays in lanes 0..3 of a wave), up to 3 600 tiles as its four 4x4-pixel quarters (16
// rays).  Such a grid has fewer tiles than the device has wave slots (8 192), so it lasts as long as its longest walk whatever the
// order, and a walk for 16 rays is about half as long as its tile's, one for 4 rays a third.  C3 scene, kernel time in ms (whole
// tiles by packets / one lane per ray / quarters / sixteenths): 16x12 - / 0.23 / - / 0.15, 32^2 1.22 / 0.34 / 0.36 / 0.15, 64^2 0.67 /
// 0.31 / 0.26 / 0.12, 128^2 0.53 / 0.37 / 0.20 / 0.18, 192^2 0.46 / 0.45 / 0.25 / 0.22, 256^2 0.37 / 0.40 / 0.24 / 0.25, 384^2 0.34 / 0.47 /
// 0.25, 640x360 0.38 / 0.55 / 0.29; sixteen times as many waves are two rounds of them from 1 024 tiles on, four times as many from
// 4 096 (512^2: 0.32 / 0.55 / 0.32), and nothing is gained.  The C2 scene draws the same lines (64^2 0.38 / 0.27 / 0.18 / 0.09;
// 192^2 - / - / 0.10 / 0.10; 256^2 - / - / 0.09 / 0.15).  Between 2 048 and 8 192 tiles the cost history picks the tiles (schedule_plan_kernel).
// Two-level scenes go the same way with their own packet kernel (whose walks are longer still: a ray crosses several instances):
// C5 as a two-level scene, 64^2 2.56 -> 0.95 ms, 128^2 2.78 -> 0.83, 256^2 4.14 -> 1.73, 640x360 2.10 -> 1.79.
static void quarter_small_grid(const mrt_ctx *ctx, mrt::TraceParams &p)
{
	if (ctx->opts.kernel != MRT_KERNEL_AUTO || ctx->opts.count_visits || p.lane_map != mrt::MAP_TILE8X8 || p.tile_w_log2 != 3u) return;
	if (p.n_nodes >= mrt::kAsmNodeLimit || p.count < kQuarterMinRays) return;
	if ((uint64_t)p.tiles_x * ((p.rows + 7u) / 8u) > kQuarterMaxTiles) return;
	p.kernel = ctx->two_level ? mrt::MRT_KERNEL_TWO_LEVEL_PACKET : MRT_KERNEL_PACKET_ASM; // (a two-level scene: its packet kernel maps lanes the same way)
	p.quarter_all = (uint64_t)p.tiles_x * ((p.rows + 7u) / 8u) <= kSixteenthMaxTiles ? 2u : 1u;
}

// The kernel of a mid-size grid cast, by measurement (mrt_ctx::GridTune).  Only for MRT_KERNEL_AUTO on flat scenes, blocking
// casts (a timing is needed), grids the schedule applies to.
static void tune_grid_kernel(mrt_ctx *ctx, mrt::TraceParams &p, int mode, uint32_t flags)
{
	auto &t = ctx->gs->tune;
	t.armed = false; t.no_pieces = false;
	if (ctx->opts.kernel != MRT_KERNEL_AUTO || ctx->two_level || !ctx->d_rows || ctx->opts.count_visits || ctx->opts.tile_schedule == 1u) return;
	// (from 2^22 rays on the 128-ray walk won every measurement -- 2560x1440 .. 7680x4320, C5's row blocks --: no frames are spent on the other one)
	if (p.lane_map != mrt::MAP_TILE8X8 || p.quarter_all || p.count < kScheduleMinRays || p.count < kTuneMinRays || p.count >= kScheduleMaxRays || p.count >= (1ull << 22)) return;
	if (p.kernel != MRT_KERNEL_PACKET_ASM && p.kernel != MRT_KERNEL_PACKET_DUAL) return;
	const bool same = t.grid_w == p.grid_w && t.grid_h == p.grid_h && t.y0 == p.y0 && t.rows == p.rows && t.mode == mode;
	if (!same) { t.grid_w = p.grid_w; t.grid_h = p.grid_h; t.y0 = p.y0; t.rows = p.rows; t.mode = mode; t.phase = 0; t.t_asm = t.t_dual = t.t_whole = 0.0f; }
	// frames 0-3: the 64-ray kernel; 4-7: the 128-ray walk, its most expensive units in pieces (schedule_plan_kernel); 8-11: the
	// same with every unit whole; then the fastest of the three (each by the faster of its last two frames)
	if (t.phase < kTuneFrames) p.kernel = MRT_KERNEL_PACKET_ASM;
	else if (t.phase < 2 * kTuneFrames) p.kernel = MRT_KERNEL_PACKET_DUAL;
	else if (t.phase < 3 * kTuneFrames) { p.kernel = MRT_KERNEL_PACKET_DUAL; t.no_pieces = true; }
	else {
		const bool whole = t.t_whole <= t.t_dual * 1.03f; // (pieces must win by more than the noise of two timings)
		const float best_dual = whole ? t.t_whole : t.t_dual;
		p.kernel = best_dual < t.t_asm ? MRT_KERNEL_PACKET_DUAL : MRT_KERNEL_PACKET_ASM;
		t.no_pieces = p.kernel == MRT_KERNEL_PACKET_DUAL && whole;
	}
	t.armed = t.phase < 3 * kTuneFrames && !(flags & MRT_FLAG_ASYNC);   // an ASYNC cast has no timing: the phase waits for a blocking one
}
static void tune_record(mrt_ctx *ctx)
{
	auto &t = ctx->gs->tune;
	if (!t.armed) return;
	// the faster of a candidate's last two frames (both launched in a measured order: schedule_grid waits for the sorts behind them)
	const int cand = t.phase / kTuneFrames, at = t.phase % kTuneFrames;
	float &slot = cand == 0 ? t.t_asm : (cand == 1 ? t.t_dual : t.t_whole);
	if (at == kTuneFrames - 2) slot = ctx->stats.last_trace_ms;
	if (at == kTuneFrames - 1 && ctx->stats.last_trace_ms < slot) slot = ctx->stats.last_trace_ms;
	t.phase++;
	t.armed = false;
}

static int grid_params(mrt_ctx *ctx, const mrt_camera *cam, uint32_t grid_w, uint32_t grid_h, uint32_t y0, uint32_t y1, mrt::TraceParams &p)
{
	if (!cam || grid_w == 0 || grid_h == 0 || y0 > y1 || y1 > grid_h) return fail(ctx, MRT_ERR_INVALID, "bad grid");
	if (cam->kind > MRT_CAMERA_ORTHOGRAPHIC) return fail(ctx, MRT_ERR_INVALID, "bad camera kind");
	// RayCamera::generate_rays asserts the resolution the camera was set up for (ray_camera.h:150-152)
	if (cam->kind != MRT_CAMERA_DEBUG_GRID && (cam->inv_w != 1.0f / (float)grid_w || cam->inv_h != 1.0f / (float)grid_h))
		return fail(ctx, MRT_ERR_INVALID, "camera was set up for another resolution");
	base_params(ctx, p);
	p.cam = *cam; p.grid_w = grid_w; p.grid_h = grid_h; p.y0 = y0; p.rows = y1 - y0;
	p.tiles_x = (grid_w + (1u << p.tile_w_log2) - 1u) >> p.tile_w_log2;
	p.count = (uint64_t)grid_w * (y1 - y0);
	p.in_fmt = mrt::IN_GRID;
	return MRT_OK;
}

int mrt_generate_grid(mrt_ctx *ctx, const mrt_camera *cam, uint32_t grid_w, uint32_t grid_h,
		uint32_t y0, uint32_t y1, mrt_ray32 *d_rays)
{
	if (!ctx || !d_rays) return MRT_ERR_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	mrt::TraceParams p;
	int rc = grid_params(ctx, cam, grid_w, grid_h, y0, y1, p);
	if (rc) return rc;
	HIP_TRY(ctx, mrt::launch_grid_rays(p, d_rays, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MRT_OK;
}

int mrt_cast_grid(mrt_ctx *ctx, const mrt_camera *cam, uint32_t grid_w, uint32_t grid_h,
		uint32_t y0, uint32_t y1, void *hits, uint32_t query_mask, int mode, uint32_t flags)
{
	if (!ctx || !hits) return MRT_ERR_INVALID;
	if (!ctx->scene) return fail(ctx, MRT_ERR_NO_SCENE, "no scene uploaded");
	if (ctx->pending) return fail(ctx, MRT_ERR_PENDING, "collect the pending dispatch first");
	if (flags & MRT_FLAG_HOST_LAYOUT) return fail(ctx, MRT_ERR_UNSUPPORTED, "grid casts write packed hits");
	if ((flags & MRT_FLAG_BOOL_OUT) && (flags & MRT_FLAG_TOKEN_OUT)) return fail(ctx, MRT_ERR_INVALID, "BOOL_OUT and TOKEN_OUT exclude each other");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	mrt::TraceParams p;
	int rc = grid_params(ctx, cam, grid_w, grid_h, y0, y1, p);
	if (rc) return rc;
	if (p.count == 0) return MRT_OK;
	const size_t hs = hit_stride(ctx, flags, mode);
	const bool hits_dev = (flags & MRT_FLAG_HITS_ON_DEVICE) != 0;
	if ((flags & MRT_FLAG_ASYNC) && !hits_dev) return fail(ctx, MRT_ERR_INVALID, "ASYNC needs device-resident hits");
	void *d_hits = hits;
	if (!hits_dev) { if ((rc = ensure(ctx, ctx->hits, p.count * hs))) return rc; d_hits = ctx->hits.ptr; }
	p.hits = d_hits; p.query_mask = query_mask;
	p.out_fmt = out_format(ctx, flags, mode);
	p.lane_map = ctx->opts.grid_tile == 1 ? mrt::MAP_LINEAR : mrt::MAP_TILE8X8;
	p.kernel = pick_kernel(ctx, true, p.count);
	quarter_small_grid(ctx, p);
	if (p.count >= kScheduleMinRays) select_grid_state(ctx, p.grid_w, p.grid_h, p.y0, p.rows, mode);
	tune_grid_kernel(ctx, p, mode, flags);
	const bool scheduled = schedule_applies(ctx, p) && !(p.quarter_all && p.count <= kQuarterAllRays);
	if (scheduled) p.quarter_all = 0u; // (from 2 048 tiles on the cost history says WHICH tiles go in quarters)
	if (scheduled && (rc = schedule_grid(ctx, p))) return rc;
	HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
	if (ctx->opts.count_visits) HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, mrt::kNumCounters * sizeof(unsigned long long), ctx->stream));
	HIP_TRY(ctx, mrt::launch_trace(p, mode == MRT_MODE_ANY_HIT, ctx->opts.count_visits != 0, ctx->stream));
	ctx->queued_kernel = p.kernel; std::snprintf(ctx->queued_variant, sizeof(ctx->queued_variant), "%s", mrt::last_trace_variant()); ctx->queued_alt_kernel = 0; ctx->queued_detect = false;
	HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
	if (scheduled && (rc = schedule_sort(ctx))) return rc;
	if (flags & MRT_FLAG_ASYNC) { ctx->stats.rays_cast += p.count; ctx->stats.last_kernel = 0; return MRT_OK; }
	if (!hits_dev) {
		HIP_TRY(ctx, hipMemcpyAsync(hits, d_hits, p.count * hs, hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	ctx->stats.last_kernel_launches = 1; ctx->stats.rays_cast += p.count;
	ctx->stats.last_h2d_ms = ctx->stats.last_sort_ms = ctx->stats.last_d2h_ms = 0.0f;
	rc = finish_timing(ctx, false, false, !hits_dev);
	if (rc == MRT_OK) tune_record(ctx);
	return rc;
}

int mrt_cast_tiled(mrt_ctx *ctx, const mrt_ray32 *d_rays, mrt_hit32 *d_hits,
		uint32_t grid_w, uint32_t rows, uint32_t query_mask, int mode)
{
	if (!ctx || !d_rays || !d_hits || grid_w == 0) return MRT_ERR_INVALID;
	if (!ctx->scene) return fail(ctx, MRT_ERR_NO_SCENE, "no scene uploaded");
	if (ctx->pending) return fail(ctx, MRT_ERR_PENDING, "collect the pending dispatch first");
	if (rows == 0) return MRT_OK;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	mrt::TraceParams p;
	base_params(ctx, p);
	p.rays = d_rays; p.hits = d_hits; p.count = (uint64_t)grid_w * rows; p.query_mask = query_mask;
	p.in_fmt = mrt::IN_RAY32; p.out_fmt = mrt::OUT_HIT32;
	p.lane_map = ctx->opts.grid_tile == 1 ? mrt::MAP_LINEAR : mrt::MAP_TILE8X8;
	p.kernel = pick_kernel(ctx, true, p.count);
	p.grid_w = grid_w; p.grid_h = rows; p.y0 = 0; p.rows = rows;
	p.tiles_x = (grid_w + (1u << p.tile_w_log2) - 1u) >> p.tile_w_log2;
	quarter_small_grid(ctx, p);
	HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
	if (ctx->opts.count_visits) HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, mrt::kNumCounters * sizeof(unsigned long long), ctx->stream));
	HIP_TRY(ctx, mrt::launch_trace(p, mode == MRT_MODE_ANY_HIT, ctx->opts.count_visits != 0, ctx->stream));
	ctx->queued_kernel = p.kernel; std::snprintf(ctx->queued_variant, sizeof(ctx->queued_variant), "%s", mrt::last_trace_variant()); ctx->queued_alt_kernel = 0; ctx->queued_detect = false;
	HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	ctx->stats.last_kernel_launches = 1; ctx->stats.rays_cast += p.count;
	ctx->stats.last_h2d_ms = ctx->stats.last_sort_ms = ctx->stats.last_d2h_ms = 0.0f;
	return finish_timing(ctx, false, false, false);
}

int mrt_expand_tokens(mrt_ctx *ctx, const void *d_rays, const uint32_t *d_tokens, void *d_hits, uint64_t count,
		uint32_t flags, void *hip_stream)
{
	if (!ctx) return MRT_ERR_INVALID;
	if (!ctx->scene) return fail(ctx, MRT_ERR_NO_SCENE, "no scene uploaded");
	if (count == 0) return MRT_OK;
	if (!d_rays || !d_tokens || !d_hits) return fail(ctx, MRT_ERR_INVALID, "null rays / tokens / hits");
	if (flags & (MRT_FLAG_BOOL_OUT | MRT_FLAG_TOKEN_OUT)) return fail(ctx, MRT_ERR_INVALID, "tokens expand to hit records only");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	mrt::TraceParams p;
	base_params(ctx, p);
	p.rays = d_rays; p.hits = d_hits; p.count = count;
	p.in_fmt = (flags & MRT_FLAG_HOST_LAYOUT) ? mrt::IN_HOST60 : mrt::IN_RAY32;
	p.out_fmt = (flags & MRT_FLAG_HOST_LAYOUT) ? mrt::OUT_HOST44 : mrt::OUT_HIT32;
	HIP_TRY(ctx, mrt::launch_expand_tokens(p, d_tokens, hip_stream ? (hipStream_t)hip_stream : ctx->stream));
	return MRT_OK;
}

int mrt_expand_grid_tokens(mrt_ctx *ctx, const mrt_camera *cam, uint32_t grid_w, uint32_t grid_h,
		uint32_t y0, uint32_t y1, const uint32_t *d_tokens, mrt_hit32 *d_hits, void *hip_stream)
{
	if (!ctx) return MRT_ERR_INVALID;
	if (!ctx->scene) return fail(ctx, MRT_ERR_NO_SCENE, "no scene uploaded");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	mrt::TraceParams p;
	int rc = grid_params(ctx, cam, grid_w, grid_h, y0, y1, p);
	if (rc) return rc;
	if (p.count == 0) return MRT_OK;
	if (!d_tokens || !d_hits) return fail(ctx, MRT_ERR_INVALID, "null tokens / hits");
	p.hits = d_hits; p.out_fmt = mrt::OUT_HIT32;
	HIP_TRY(ctx, mrt::launch_expand_tokens(p, d_tokens, hip_stream ? (hipStream_t)hip_stream : ctx->stream));
	return MRT_OK;
}

int mrt_morton_keys(mrt_ctx *ctx, const mrt_ray32 *d_rays, uint64_t count, uint32_t *d_keys)
{
	if (!ctx || !d_rays || !d_keys) return MRT_ERR_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, mrt::launch_morton_keys(d_rays, mrt::IN_RAY32, count, d_keys, nullptr, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MRT_OK;
}

uint32_t mrt_token_bytes(mrt_ctx *ctx)
{
	return ctx ? (uint32_t)token_bytes(ctx) : 0u;
}

const char *mrt_last_kernel_variant(mrt_ctx *ctx)
{
	return ctx ? ctx->last_variant : "";
}

int mrt_get_stats(mrt_ctx *ctx, mrt_stats *out)
{
	if (!ctx || !out) return MRT_ERR_INVALID;
	*out = ctx->stats;
	return MRT_OK;
}

int mrt_device_alloc(mrt_ctx *ctx, size_t bytes, void **d_ptr)
{
	if (!ctx || !d_ptr) return MRT_ERR_INVALID;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (hipMalloc(d_ptr, bytes ? bytes : 16) != hipSuccess) { *d_ptr = nullptr; return fail(ctx, MRT_ERR_OOM, "device allocation failed"); }
	return MRT_OK;
}
int mrt_device_free(mrt_ctx *ctx, void *d_ptr)
{
	if (!ctx) return MRT_ERR_INVALID;
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	if (d_ptr) HIP_TRY(ctx, hipFree(d_ptr));
	return MRT_OK;
}
int mrt_memcpy_h2d(mrt_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
	if (!ctx || (bytes && (!d_dst || !h_src))) return MRT_ERR_INVALID;
	HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MRT_OK;
}
int mrt_memcpy_d2h(mrt_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
	if (!ctx || (bytes && (!h_dst || !d_src))) return MRT_ERR_INVALID;
	HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return MRT_OK;
}

} // extern "C"
