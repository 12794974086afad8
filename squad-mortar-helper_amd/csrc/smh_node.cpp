// smh_node.cpp -- the multi-GPU entry points of the C ABI (include/smh_vision_hip.h, smhv_node_*): ONE process drives
// every GPU of a node.  Frames are independent, so a batch is block-sharded over the devices with no data-path collective;
// each device runs the single-GPU pipeline on its resident shard, and the only exchange is one ncclGather (rccl.h:745) of
// the fixed-size per-frame result records to the root device over xGMI: every peer uses its own
// link to the root, the payload (1216 B per frame) is latency-bound.
// Nothing like this exists in the reference (device 0 only, vision-gpu/src/cuda.rs:34).
//
// RCCL is resolved with dlopen at smhv_node_create: the library itself keeps depending on libamdhip64 only, and a host
// that already has an RCCL loaded (PyTorch bundles one under the same SONAME) gets that one.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/smh_vision_hip.h"
#include "../../include/smh_vision_hip_debug.h"

extern "C" int smhv_internal_fail(int code, const char *fmt, ...);          // smh_runtime.cpp: sets smhv_last_error

namespace {
struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

int load_rccl(Rccl &r) {
	for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
		r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
		if (r.lib) break;
	}
	if (!r.lib) return smhv_internal_fail(SMHV_E_NO_DEVICE, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
#define SYM(field, sym)                                                                                   \
	do {                                                                                                  \
		*(void **)&r.field = dlsym(r.lib, sym);                                                           \
		if (!r.field) return smhv_internal_fail(SMHV_E_NO_DEVICE, "RCCL symbol %s missing", sym);         \
	} while (0)
	SYM(CommInitAll, "ncclCommInitAll");
	SYM(CommDestroy, "ncclCommDestroy");
	SYM(Gather, "ncclGather");
	SYM(GroupStart, "ncclGroupStart");
	SYM(GroupEnd, "ncclGroupEnd");
	SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
	return SMHV_OK;
}
}  // namespace

struct smhv_node {
	Rccl rccl;
	std::vector<int> devices;
	std::vector<smhv_ctx *> ctx;
	std::vector<smhv_pipeline *> pipe;
	std::vector<ncclComm_t> comm;
	std::vector<hipStream_t> gstream;       // one stream per device for the gather
	std::vector<hipEvent_t> gdone;
	std::vector<uint32_t> n_last, slot_last;
	uint32_t max_frames = 0;
	smhv_frame_result *d_gather = nullptr;  // root: n_devices x max_frames records
	smhv_frame_result *h_gather = nullptr;  // pinned
};

#define NCCLCHK(node, expr)                                                                                           \
	do {                                                                                                              \
		ncclResult_t _r = (expr);                                                                                     \
		if (_r != ncclSuccess) return smhv_internal_fail(SMHV_E_HIP, "%s failed: %s", #expr, (node)->rccl.GetErrorString(_r)); \
	} while (0)
#define HIPCHK(expr)                                                                                                  \
	do {                                                                                                              \
		hipError_t _e = (expr);                                                                                       \
		if (_e != hipSuccess) return smhv_internal_fail(SMHV_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
	} while (0)

extern "C" SMHV_API void smhv_shard_range(uint64_t n_total, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi) {
	// contiguous block shard: rank r owns frames [r * n / world, (r + 1) * n / world)
	if (!world) world = 1;
	if (lo) *lo = n_total * rank / world;
	if (hi) *hi = n_total * (rank + 1ull) / world;
}

extern "C" SMHV_API void smhv_node_destroy(smhv_node *nd) {
	if (!nd) return;
	for (size_t i = 0; i < nd->devices.size(); ++i) {
		(void)hipSetDevice(nd->devices[i]);
		(void)hipDeviceSynchronize();
		if (i < nd->comm.size() && nd->comm[i] && nd->rccl.CommDestroy) (void)nd->rccl.CommDestroy(nd->comm[i]);
		if (i < nd->gdone.size() && nd->gdone[i]) (void)hipEventDestroy(nd->gdone[i]);
		if (i < nd->gstream.size() && nd->gstream[i]) (void)hipStreamDestroy(nd->gstream[i]);
		if (i < nd->pipe.size() && nd->pipe[i]) smhv_pipeline_destroy(nd->pipe[i]);
	}
	if (!nd->devices.empty()) (void)hipSetDevice(nd->devices[0]);
	if (nd->d_gather) (void)hipFree(nd->d_gather);
	if (nd->h_gather) (void)hipHostFree(nd->h_gather);
	for (size_t i = 0; i < nd->ctx.size(); ++i) if (nd->ctx[i]) smhv_shutdown(nd->ctx[i]);
	delete nd;
}

extern "C" SMHV_API int smhv_node_create(const int *devices, uint32_t n_devices, uint32_t frame_w, uint32_t frame_h, uint32_t max_frames_per_device,
                                         uint32_t depth, smhv_log_fn log, smhv_node **out) {
	if (!devices || !out || n_devices == 0 || n_devices > 64 || max_frames_per_device == 0) return smhv_internal_fail(SMHV_E_INVALID, "node_create: bad arguments");
	*out = nullptr;
	smhv_node *nd = new (std::nothrow) smhv_node();
	if (!nd) return smhv_internal_fail(SMHV_E_INVALID, "out of host memory");
	int rc = load_rccl(nd->rccl);
	if (rc) { delete nd; return rc; }
	nd->devices.assign(devices, devices + n_devices);
	nd->max_frames = max_frames_per_device;
	nd->ctx.assign(n_devices, nullptr); nd->pipe.assign(n_devices, nullptr); nd->comm.assign(n_devices, nullptr);
	nd->gstream.assign(n_devices, nullptr); nd->gdone.assign(n_devices, nullptr);
	nd->n_last.assign(n_devices, 0); nd->slot_last.assign(n_devices, 0);
	for (uint32_t i = 0; i < n_devices && !rc; ++i) {
		rc = smhv_init(devices[i], log, &nd->ctx[i]);
		if (!rc) {
			// depth 0: what bench.py runs a single GPU with (12 slots, frame-granular search); the node's gather is an RCCL kernel, which
			// needs a CU without a search workgroup to run on (smhv_pipeline_options::room_for_others)
			smhv_pipeline_options opt;
			memset(&opt, 0, sizeof opt);
			opt.size = sizeof opt;
			opt.room_for_others = 1u;
			rc = smhv_pipeline_create_ex(nd->ctx[i], frame_w, frame_h, max_frames_per_device, depth ? depth : 12u, &opt, &nd->pipe[i]);
		}
		if (!rc) {
			hipError_t e = hipSetDevice(devices[i]);
			if (e == hipSuccess) e = hipStreamCreateWithFlags(&nd->gstream[i], hipStreamNonBlocking);
			if (e == hipSuccess) e = hipEventCreateWithFlags(&nd->gdone[i], hipEventDisableTiming);
			if (e != hipSuccess) rc = smhv_internal_fail(SMHV_E_HIP, "node stream: %s", hipGetErrorString(e));
		}
	}
	if (!rc) {
		ncclResult_t r = nd->rccl.CommInitAll(nd->comm.data(), (int)n_devices, nd->devices.data());
		if (r != ncclSuccess) rc = smhv_internal_fail(SMHV_E_HIP, "ncclCommInitAll over %u devices: %s", n_devices, nd->rccl.GetErrorString(r));
	}
	if (!rc) {
		hipError_t e = hipSetDevice(devices[0]);
		const size_t bytes = sizeof(smhv_frame_result) * (size_t)n_devices * max_frames_per_device;
		if (e == hipSuccess) e = hipMalloc((void **)&nd->d_gather, bytes);
		if (e == hipSuccess) e = hipHostMalloc((void **)&nd->h_gather, bytes);
		if (e != hipSuccess) rc = smhv_internal_fail(SMHV_E_HIP, "node gather buffers: %s", hipGetErrorString(e));
	}
	if (rc) { smhv_node_destroy(nd); return rc; }
	*out = nd;
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_node_ctx(smhv_node *nd, uint32_t i, smhv_ctx **ctx, smhv_pipeline **pipe) {
	if (!nd || i >= nd->devices.size()) return smhv_internal_fail(SMHV_E_INVALID, "node_ctx: bad arguments");
	if (ctx) *ctx = nd->ctx[i];
	if (pipe) *pipe = nd->pipe[i];
	return SMHV_OK;
}

extern "C" SMHV_API int smhv_node_run(smhv_node *nd, const void *const *d_frames, const uint32_t *n, uint32_t stages, int grayscale, uint32_t max_gap,
                                      const smhv_anchors *const *anchors) {
	if (!nd || !d_frames || !n) return smhv_internal_fail(SMHV_E_INVALID, "node_run: null argument");
	const size_t nd_n = nd->devices.size();
	for (size_t i = 0; i < nd_n; ++i) {                            // nothing is submitted unless every shard is acceptable
		if (n[i] > nd->max_frames) return smhv_internal_fail(SMHV_E_INVALID, "node_run: %u frames on device %zu, capacity %u", n[i], i, nd->max_frames);
		if (n[i] && !d_frames[i]) return smhv_internal_fail(SMHV_E_INVALID, "node_run: null frame pointer for device %zu", i);
	}
	std::vector<uint32_t> slot(nd->slot_last);
	for (size_t i = 0; i < nd_n; ++i) {
		if (n[i] == 0) continue;
		int rc = smhv_pipeline_submit(nd->pipe[i], d_frames[i], n[i], stages, grayscale, max_gap, anchors ? anchors[i] : nullptr, nullptr, &slot[i]);
		if (rc) return rc;                                         // the run is void: smhv_node_gather keeps returning the previous complete run
	}
	// committed only now: a gather never mixes the counts of a failed run with the slots of the one before
	for (size_t i = 0; i < nd_n; ++i) { nd->n_last[i] = n[i]; nd->slot_last[i] = slot[i]; }
	return SMHV_OK;
}

extern "C" int smhv_internal_batch_check(smhv_batch *b, const char *what);     // smh_runtime.cpp: frames that failed (SMHV_FRAME_*)

extern "C" SMHV_API int smhv_node_gather(smhv_node *nd, smhv_frame_result *out, uint32_t *n_total) {
	if (!nd || !out) return smhv_internal_fail(SMHV_E_INVALID, "node_gather: null argument");
	const size_t nd_n = nd->devices.size();
	// ncclGather moves the same count from every rank: the largest shard of the run (the records behind a shorter shard's
	// own are stale padding and are dropped below), not the capacity
	uint32_t n_max = 0;
	for (size_t i = 0; i < nd_n; ++i) n_max = nd->n_last[i] > n_max ? nd->n_last[i] : n_max;
	if (n_total) *n_total = 0;
	if (n_max == 0) return SMHV_OK;
	const size_t per = sizeof(smhv_frame_result) * (size_t)n_max;
	// everything that can fail without RCCL comes first: each device's gather stream is ordered after that device's most
	// recent pass, and the send pointers are looked up
	std::vector<void *> src(nd_n, nullptr);
	std::vector<smhv_batch *> batch(nd_n, nullptr);
	for (size_t i = 0; i < nd_n; ++i) {
		HIPCHK(hipSetDevice(nd->devices[i]));
		void *st = nullptr;
		int rc = smhv_pipeline_slot(nd->pipe[i], nd->slot_last[i], &batch[i], &st);
		if (rc) return rc;
		rc = smhv_batch_device_ptrs(batch[i], &src[i], nullptr, nullptr, nullptr, nullptr, nullptr);
		if (rc) return rc;
		HIPCHK(hipEventRecord(nd->gdone[i], (hipStream_t)st));
		HIPCHK(hipStreamWaitEvent(nd->gstream[i], nd->gdone[i], 0));
	}
	// one ncclGather over all devices (group: a single thread issues every rank's call).  The group is closed on every
	// path: an open group would swallow every later RCCL call of this thread (including those of a host framework that
	// shares the library).
	ncclResult_t first = ncclSuccess;
	hipError_t hfirst = hipSuccess;
	NCCLCHK(nd, nd->rccl.GroupStart());
	for (size_t i = 0; i < nd_n && first == ncclSuccess && hfirst == hipSuccess; ++i) {
		hfirst = hipSetDevice(nd->devices[i]);
		if (hfirst == hipSuccess) first = nd->rccl.Gather(src[i], i == 0 ? (void *)nd->d_gather : nullptr, per, ncclUint8, 0, nd->comm[i], nd->gstream[i]);
	}
	const ncclResult_t end = nd->rccl.GroupEnd();
	if (hfirst != hipSuccess) return smhv_internal_fail(SMHV_E_HIP, "node_gather: hipSetDevice failed: %s", hipGetErrorString(hfirst));
	if (first != ncclSuccess) return smhv_internal_fail(SMHV_E_HIP, "node_gather: ncclGather failed: %s", nd->rccl.GetErrorString(first));
	if (end != ncclSuccess) return smhv_internal_fail(SMHV_E_HIP, "node_gather: ncclGroupEnd failed: %s", nd->rccl.GetErrorString(end));
	for (size_t i = 0; i < nd_n; ++i) {                             // each slot's next pass waits for the gather that reads its records
		int rc = smhv_pipeline_hold(nd->pipe[i], nd->slot_last[i], nd->gstream[i]);
		if (rc) return rc;
	}
	HIPCHK(hipSetDevice(nd->devices[0]));
	HIPCHK(hipMemcpyAsync(nd->h_gather, nd->d_gather, per * nd_n, hipMemcpyDeviceToHost, nd->gstream[0]));
	HIPCHK(hipStreamSynchronize(nd->gstream[0]));
	uint32_t k = 0;
	for (size_t i = 0; i < nd_n; ++i) {                             // compact: device i contributed n_last[i] records
		memcpy(out + k, nd->h_gather + i * (size_t)n_max, sizeof(smhv_frame_result) * nd->n_last[i]);
		k += nd->n_last[i];
	}
	if (n_total) *n_total = k;
	// frames that failed on any device (their records carry the status; the root's copy above is complete either way).  The
	// gather on the root's stream finished after every peer's send, i.e. after every device's pass.
	for (size_t i = 0; i < nd_n; ++i) {
		if (nd->n_last[i] == 0) continue;
		const int r = smhv_internal_batch_check(batch[i], "node_gather");
		if (r) return r;                                           // (a later device's failures are reported by the next call)
	}
	return SMHV_OK;
}
