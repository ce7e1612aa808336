// group.hip — mrt_group_*: one process driving several MI355X of a node through the C-ABI ("Multi-GPU:
// mrt_group_create(n_dev, ...), mrt_group_cast_grid(...)"; 8(e): rows split into contiguous blocks, BVH replicated,
// rays generated on each device, ONE exchange at the end).
//
// The reference is single-device: its only caller of the backend is C++ (RayDispatcher, src/dispatch/
// ray_dispatcher.h:124-181; RayTracerServer::cast_rays_batch / submit, src/godot/raytracer_server.cpp:285-328), and a C++
// host cannot use the torch.distributed path of messyerraytracer_amd/sharded.py.  This is the same sharding for
// such a host: one mrt_ctx and one HIP stream per device, every device traces its row block of the grid (fused ray
// generation, mrt_cast_grid) and writes 4-byte hit tokens; the tokens travel to device 0 as peer copies (xGMI:
// point-to-point, each peer over its own link, no ring), and device 0 rebuilds the 32-byte records with
// mrt_expand_grid_tokens from its own copy of the scene — bit-identical to what the casts would have stored
// (tests/test_parity_gpu.py::test_hit_tokens_expand_to_identical_records).  Two-level scenes travel as 8-byte tokens
// {triangle, instance} (mrt_token_bytes); any-hit bool output sends what the casts wrote.  The host only queues work: casts are MRT_FLAG_ASYNC, copies are
// hipMemcpyPeerAsync on the sending device's stream, device 0's stream waits on one event per peer.
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>
#include <hip/hip_runtime.h>
#include "mrt_internal.h"

struct mrt_group {
	std::vector<mrt_ctx *> ctx;
	std::vector<int> device;
	std::vector<hipStream_t> stream;
	std::vector<hipEvent_t> done;          // per member (spare; the chunked cast uses the per-chunk events below)
	std::vector<hipStream_t> copy_stream;  // per member > 0: its peer copies (beside its next trace)
	std::vector<std::vector<hipEvent_t>> traced, copied; // per member and chunk: the chunk is traced / has arrived on device 0
	hipStream_t xstream = nullptr;         // device 0: record rebuilds + the final copy, beside member 0's tracing
	std::vector<void *> block;             // per member: what its cast writes (tokens, records or bools), on its device
	std::vector<size_t> block_cap;
	void *staged = nullptr; size_t staged_cap = 0;   // device 0: everybody's tokens, row-major
	void *image = nullptr; size_t image_cap = 0;     // device 0: the assembled output when the caller wants it on the host
	bool two_level = false;
	char err[512] = {0};
};

namespace {

int gfail(mrt_group *g, int code, const char *msg)
{
	if (g) std::snprintf(g->err, sizeof(g->err), "%s", msg);
	return code;
}

int gfail_ctx(mrt_group *g, int code, int member)
{
	std::snprintf(g->err, sizeof(g->err), "device %d: %s", g->device[member], mrt_last_error(g->ctx[member]));
	return code;
}

#define GHIP(g, call)                                                                                        \
	do {                                                                                                     \
		hipError_t e_ = (call);                                                                              \
		if (e_ != hipSuccess) {                                                                              \
			std::snprintf((g)->err, sizeof((g)->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
					__FILE__, __LINE__);                                                                     \
			return MRT_ERR_HIP;                                                                              \
		}                                                                                                    \
	} while (0)

int grow(mrt_group *g, int device, void **buf, size_t *cap, size_t bytes)
{
	if (*cap >= bytes) return MRT_OK;
	GHIP(g, hipSetDevice(device));
	if (*buf) { GHIP(g, hipDeviceSynchronize()); GHIP(g, hipFree(*buf)); *buf = nullptr; *cap = 0; }
	const size_t want = bytes + bytes / 2;   // grow-only x1.5, as the per-dispatch buffers of a context
	if (hipMalloc(buf, want) != hipSuccess) { *buf = nullptr; return gfail(g, MRT_ERR_OOM, "group: device allocation failed"); }
	*cap = want;
	return MRT_OK;
}

} // namespace

extern "C" {

// Contiguous row block of member `rank` of `n`: y in [rank * rows / n, (rank + 1) * rows / n)  (SURVEY.md 8(e);
// the same split as sharded.row_block).  Host-only.
void mrt_group_row_block(uint32_t rank, uint32_t n, uint32_t rows, uint32_t *y0, uint32_t *y1)
{
	if (n == 0) n = 1;
	if (y0) *y0 = (uint32_t)((uint64_t)rank * rows / n);
	if (y1) *y1 = (uint32_t)(((uint64_t)rank + 1u) * rows / n);
}

int mrt_group_create(int n_devices, const int *device_ordinals, const mrt_options *opts, mrt_group **out)
{
	if (!out) return MRT_ERR_INVALID;
	*out = nullptr;
	if (n_devices <= 0 || n_devices > 64) return MRT_ERR_INVALID;
	int present = 0;
	if (hipGetDeviceCount(&present) != hipSuccess || present <= 0) return MRT_ERR_NO_DEVICE;
	mrt_group *g = new (std::nothrow) mrt_group();
	if (!g) return MRT_ERR_OOM;
	for (int i = 0; i < n_devices; i++) {
		const int dev = device_ordinals ? device_ordinals[i] : i;
		if (dev < 0 || dev >= present) { mrt_group_destroy(g); return MRT_ERR_NO_DEVICE; }
		mrt_ctx *c = nullptr;
		int rc = mrt_create(dev, opts, &c);
		if (rc) { mrt_group_destroy(g); return rc; }
		hipStream_t s = nullptr; hipEvent_t e = nullptr;
		if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
				hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || mrt_set_stream(c, (void *)s) != MRT_OK) {
			if (e) (void)hipEventDestroy(e);
			if (s) (void)hipStreamDestroy(s);
			mrt_destroy(c); mrt_group_destroy(g);
			return MRT_ERR_HIP;
		}
		g->ctx.push_back(c); g->device.push_back(dev); g->stream.push_back(s); g->done.push_back(e);
		g->block.push_back(nullptr); g->block_cap.push_back(0);
		g->copy_stream.push_back(nullptr); g->traced.emplace_back(); g->copied.emplace_back();
	}
	// peers may write device 0's memory directly (xGMI); without it the copies are staged by the runtime
	for (size_t i = 1; i < g->device.size(); i++) {
		if (g->device[i] == g->device[0]) continue;
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, g->device[i], g->device[0]) == hipSuccess && can && hipSetDevice(g->device[i]) == hipSuccess)
			(void)hipDeviceEnablePeerAccess(g->device[0], 0);
		(void)hipGetLastError(); // "already enabled" is fine
	}
	*out = g;
	return MRT_OK;
}

void mrt_group_destroy(mrt_group *g)
{
	if (!g) return;
	for (size_t i = 0; i < g->ctx.size(); i++) {
		(void)hipSetDevice(g->device[i]);
		if (g->stream[i]) (void)hipStreamSynchronize(g->stream[i]);
		if (g->block[i]) (void)hipFree(g->block[i]);
		mrt_destroy(g->ctx[i]);            // (waits for its stream, which it does not own)
		if (g->done[i]) (void)hipEventDestroy(g->done[i]);
		for (hipEvent_t e : g->traced[i]) (void)hipEventDestroy(e);
		for (hipEvent_t e : g->copied[i]) (void)hipEventDestroy(e);
		if (g->copy_stream[i]) { (void)hipStreamSynchronize(g->copy_stream[i]); (void)hipStreamDestroy(g->copy_stream[i]); }
		if (g->stream[i]) (void)hipStreamDestroy(g->stream[i]);
	}
	if (!g->device.empty()) {
		(void)hipSetDevice(g->device[0]);
		if (g->xstream) { (void)hipStreamSynchronize(g->xstream); (void)hipStreamDestroy(g->xstream); }
		if (g->staged) (void)hipFree(g->staged);
		if (g->image) (void)hipFree(g->image);
	}
	delete g;
}

int mrt_group_size(const mrt_group *g) { return g ? (int)g->ctx.size() : 0; }
mrt_ctx *mrt_group_context(mrt_group *g, int member) { return (g && member >= 0 && member < (int)g->ctx.size()) ? g->ctx[member] : nullptr; }
const char *mrt_group_last_error(const mrt_group *g) { return g ? g->err : "null group"; }

// The read-only scene, replicated: every member uploads the same arrays (the conversion is deterministic, so every
// device holds the same rows and hit tokens mean the same triangle everywhere).
int mrt_group_upload_scene(mrt_group *g, const mrt_tri64 *tris, uint32_t n_tris, const mrt_bvh_node32 *nodes, uint32_t used_nodes,
		const uint32_t *prim_idx)
{
	if (!g) return MRT_ERR_INVALID;
	for (size_t i = 0; i < g->ctx.size(); i++) {
		int rc = mrt_upload_scene(g->ctx[i], tris, n_tris, nodes, used_nodes, prim_idx);
		if (rc) return gfail_ctx(g, rc, (int)i);
	}
	g->two_level = false;
	return MRT_OK;
}

int mrt_group_upload_two_level_scene(mrt_group *g, const float *verts9, uint32_t n_mesh_tris, const mrt_instance *instances,
		uint32_t n_instances, uint32_t flags)
{
	if (!g) return MRT_ERR_INVALID;
	for (size_t i = 0; i < g->ctx.size(); i++) {
		int rc = mrt_upload_two_level_scene(g->ctx[i], verts9, n_mesh_tris, instances, n_instances, flags);
		if (rc) return gfail_ctx(g, rc, (int)i);
	}
	g->two_level = true;
	return MRT_OK;
}

// One grid_w x grid_h primary-ray grid, rows sharded over the members, results assembled in `hits`: grid_w * grid_h
// records (mrt_hit32; uint8 with MRT_FLAG_BOOL_OUT in any-hit mode), row-major, on the host or — with
// MRT_FLAG_HITS_ON_DEVICE — in device 0's memory.  Blocking.  Equals mrt_cast_grid of the whole grid on one device,
// byte for byte.
//
// Every member's row block is cut into kGroupChunks chunks, and three things run beside each other: the member traces
// chunk c + 1 on its own stream while chunk c travels to device 0 on the member's copy stream (hipMemcpyPeerAsync, each
// peer over its own xGMI link), and device 0 rebuilds the records of every chunk that has arrived on a side stream of
// its own beside its own tracing (as sharded.ShardedGrid does per rank).  The host only queues work and waits once.
static constexpr uint32_t kGroupChunks = 4;

int mrt_group_cast_grid(mrt_group *g, const mrt_camera *cam, uint32_t grid_w, uint32_t grid_h, void *hits, uint32_t query_mask,
		int mode, uint32_t flags)
{
	if (!g || !cam || !hits) return MRT_ERR_INVALID;
	if (grid_w == 0 || grid_h == 0) return gfail(g, MRT_ERR_INVALID, "group: bad grid");
	if (flags & ~(uint32_t)(MRT_FLAG_HITS_ON_DEVICE | MRT_FLAG_BOOL_OUT)) return gfail(g, MRT_ERR_INVALID, "group: unsupported flag");
	if ((flags & MRT_FLAG_BOOL_OUT) && mode != MRT_MODE_ANY_HIT) return gfail(g, MRT_ERR_INVALID, "BOOL_OUT needs any-hit mode");
	const uint32_t n = (uint32_t)g->ctx.size();
	const bool bools = (flags & MRT_FLAG_BOOL_OUT) != 0;
	const bool tokens = !bools && n > 1;                        // what travels: hit tokens (4 bytes; 8 for a two-level scene), else the bools
	const size_t out_stride = bools ? 1 : sizeof(mrt_hit32);
	const size_t wire_stride = tokens ? mrt_token_bytes(g->ctx[0]) : out_stride;
	const size_t total = (size_t)grid_w * grid_h;
	const bool on_device = (flags & MRT_FLAG_HITS_ON_DEVICE) != 0;
	int rc;
	void *d_out = hits;                                         // device 0: where the assembled output goes
	if (!on_device) {
		if ((rc = grow(g, g->device[0], &g->image, &g->image_cap, total * out_stride))) return rc;
		d_out = g->image;
	}
	if (tokens && (rc = grow(g, g->device[0], &g->staged, &g->staged_cap, total * wire_stride))) return rc;
	char *dst0 = (char *)(tokens ? g->staged : d_out);          // device 0: where the blocks land
	const uint32_t cast_flags = MRT_FLAG_HITS_ON_DEVICE | MRT_FLAG_ASYNC | (tokens ? MRT_FLAG_TOKEN_OUT : 0u) | (bools ? MRT_FLAG_BOOL_OUT : 0u);
	const uint32_t chunks = n > 1 ? kGroupChunks : 1u;
	// device 0's side stream: record rebuilds and the final copy (created on first use)
	GHIP(g, hipSetDevice(g->device[0]));
	if (!g->xstream) GHIP(g, hipStreamCreateWithFlags(&g->xstream, hipStreamNonBlocking));
	for (uint32_t r = 0; r < n; r++) {
		uint32_t y0, y1;
		mrt_group_row_block(r, n, grid_h, &y0, &y1);
		if (y1 == y0) continue;
		char *block = dst0 + (size_t)y0 * grid_w * wire_stride;  // member 0 writes in place
		if (r > 0) {
			if ((rc = grow(g, g->device[r], &g->block[r], &g->block_cap[r], (size_t)(y1 - y0) * grid_w * wire_stride))) return rc;
			block = (char *)g->block[r];
		}
		GHIP(g, hipSetDevice(g->device[r]));
		if (r > 0 && !g->copy_stream[r]) GHIP(g, hipStreamCreateWithFlags(&g->copy_stream[r], hipStreamNonBlocking));
		while (g->traced[r].size() < chunks) { hipEvent_t e; GHIP(g, hipEventCreateWithFlags(&e, hipEventDisableTiming)); g->traced[r].push_back(e); }
		while (g->copied[r].size() < chunks) { hipEvent_t e; GHIP(g, hipEventCreateWithFlags(&e, hipEventDisableTiming)); g->copied[r].push_back(e); }
		for (uint32_t c = 0; c < chunks; c++) {
			uint32_t c0, c1;
			mrt_group_row_block(c, chunks, y1 - y0, &c0, &c1);
			if (c1 == c0) continue;
			const size_t off = (size_t)c0 * grid_w * wire_stride, bytes = (size_t)(c1 - c0) * grid_w * wire_stride;
			// 1. the member traces the chunk (queued on its stream; the host does not wait)
			rc = mrt_cast_grid(g->ctx[r], cam, grid_w, grid_h, y0 + c0, y0 + c1, block + off, query_mask, mode, cast_flags);
			if (rc) return gfail_ctx(g, rc, (int)r);
			GHIP(g, hipSetDevice(g->device[r]));
			GHIP(g, hipEventRecord(g->traced[r][c], g->stream[r]));
			hipEvent_t arrived = g->traced[r][c];
			if (r > 0) { // 2. the chunk to device 0 on the copy stream, beside the member's next trace
				GHIP(g, hipStreamWaitEvent(g->copy_stream[r], g->traced[r][c], 0));
				GHIP(g, hipMemcpyPeerAsync(dst0 + (size_t)(y0 + c0) * grid_w * wire_stride, g->device[0], block + off, g->device[r], bytes, g->copy_stream[r]));
				GHIP(g, hipEventRecord(g->copied[r][c], g->copy_stream[r]));
				arrived = g->copied[r][c];
			}
			// 3. device 0: once the chunk is there, rebuild its records on the side stream
			GHIP(g, hipSetDevice(g->device[0]));
			GHIP(g, hipStreamWaitEvent(g->xstream, arrived, 0));
			if (tokens) {
				rc = mrt_expand_grid_tokens(g->ctx[0], cam, grid_w, grid_h, y0 + c0, y0 + c1,
						(const uint32_t *)(g->staged ? (char *)g->staged + (size_t)(y0 + c0) * grid_w * wire_stride : nullptr),
						(mrt_hit32 *)((char *)d_out + (size_t)(y0 + c0) * grid_w * out_stride), (void *)g->xstream);
				if (rc) return gfail_ctx(g, rc, 0);
			}
		}
	}
	GHIP(g, hipSetDevice(g->device[0]));
	if (!on_device) GHIP(g, hipMemcpyAsync(hits, d_out, total * out_stride, hipMemcpyDeviceToHost, g->xstream));
	GHIP(g, hipStreamSynchronize(g->xstream));
	for (uint32_t r = 0; r < n; r++) { // (the members' buffers may be reused by the next call)
		GHIP(g, hipSetDevice(g->device[r]));
		GHIP(g, hipStreamSynchronize(g->stream[r]));
		if (g->copy_stream[r]) GHIP(g, hipStreamSynchronize(g->copy_stream[r]));
	}
	return MRT_OK;
}

} // extern "C"
