/*
 * node.hip -- multi-device front of libfsm_hip.so, for C hosts (rx, retest, re): one struct fsm_hip_node
 * = one replica of the DFA's table per GPU of the node.  A batch is split into contiguous index shards of
 * whole bitmap words, one per device; one host thread per device drives its shard (its own device
 * context, its own stream: no serial generator, no shared launch path).  There is no exchange on the
 * data path -- inputs are independent, the table is replicated:
 *   - host-pointer calls: each device's thread stages its slice and copies its results straight into the
 *     caller's arrays at the shard's offset: nothing to gather;
 *   - device-resident calls: each device writes its slice of the accept bitmap into its copy of the
 *     whole-batch bitmap, and ONE ncclAllGather (in place, RCCL over xGMI) gives every device the whole
 *     bitmap; the match count is one ncclAllReduce of a u64.
 * RCCL is bound with dlopen("librccl.so.1") on first use, so libfsm_hip.so carries no link dependency on
 * it; when it is absent, or when the device list repeats a device (a test rig with fewer GPUs than
 * replicas), the same exchange is done with peer-to-peer copies of the slices.
 * See include/fsm_hip.h for the contract of every entry point.
 */
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/fsm_hip.h"

/* the few RCCL entry points used (rccl/rccl.h:236, :260, :611, :678, :923, :933); ncclUint64 = 5, ncclSum = 0 */
typedef struct ncclComm *ncclComm_t;
struct rccl_api {
	int state = 0;   /* 0 unresolved, 1 usable, -1 missing */
	int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	int (*CommDestroy)(ncclComm_t) = nullptr;
	int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
	int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
	int (*GroupStart)(void) = nullptr;
	int (*GroupEnd)(void) = nullptr;
};
static rccl_api R;
static std::once_flag rccl_once;
static char rccl_path[512] = "";   /* the shared object ncclAllGather was bound from (dladdr): a torch process has its own bundle mapped */

static void rccl_resolve()
{
	void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
	if (h == nullptr) { R.state = -1; return; }
	*(void **)&R.CommInitAll = dlsym(h, "ncclCommInitAll");
	*(void **)&R.CommDestroy = dlsym(h, "ncclCommDestroy");
	*(void **)&R.AllGather = dlsym(h, "ncclAllGather");
	*(void **)&R.AllReduce = dlsym(h, "ncclAllReduce");
	*(void **)&R.GroupStart = dlsym(h, "ncclGroupStart");
	*(void **)&R.GroupEnd = dlsym(h, "ncclGroupEnd");
	R.state = R.CommInitAll && R.CommDestroy && R.AllGather && R.AllReduce && R.GroupStart && R.GroupEnd ? 1 : -1;
	Dl_info di;
	if (R.AllGather != nullptr && dladdr((void *)R.AllGather, &di) != 0 && di.dli_fname != nullptr) snprintf(rccl_path, sizeof rccl_path, "%s", di.dli_fname);
}

/* one parked host thread per device (but device 0's shard, which the calling thread drives): created with the node,
 * woken per call -- a batch of a few thousand inputs is a 20 us walk, a thread spawn + join per call was longer */
struct node_worker {
	std::thread th;
	std::mutex m;
	std::condition_variable cv;
	const std::function<int(int)> *job = nullptr;
	int err = 0;
	bool done = true, quit = false;
};

struct fsm_hip_node {
	std::vector<int> dev;
	std::vector<fsm_hip_dfa *> dfa;
	std::vector<hipStream_t> stream;             /* per device: the walk */
	std::vector<hipStream_t> cstream;            /* per device: the collective of an asynchronous call */
	std::vector<hipEvent_t> walked;              /* per device: walk done -> collective may start */
	std::vector<hipEvent_t> gathered;            /* per device and slot [2k + slot]: that call's collective is done */
	std::vector<unsigned long long *> d_count;   /* one u64 per device */
	std::vector<ncclComm_t> comm;                /* empty: exchange by peer copies */
	std::vector<std::unique_ptr<node_worker>> workers;   /* [k] drives device k, k >= 1 */
	int aslot = 0;                               /* which of the two count slots / gathered events the next call uses */
	bool async_pending = false;                  /* an asynchronous call's collective may still be running */
	bool async_count = false;                    /* ... and it reduces a match count into d_count[] */
	std::mutex mu;                               /* one batch at a time per node */
};

static void worker_main(fsm_hip_node *nd, size_t k)
{
	node_worker &w = *nd->workers[k];
	(void)hipSetDevice(nd->dev[k]);
	std::unique_lock<std::mutex> lk(w.m);
	for (;;) {
		w.cv.wait(lk, [&] { return w.quit || w.job != nullptr; });
		if (w.quit) return;
		const std::function<int(int)> *job = w.job;
		lk.unlock();
		errno = 0;
		const int e = (*job)((int)k) != 0 ? (errno ? errno : EIO) : 0;
		lk.lock();
		w.err = e;
		w.job = nullptr;
		w.done = true;
		w.cv.notify_all();
	}
}

/* clearing by a KERNEL, not hipMemsetAsync: as the first node of a replayed graph a memset node was seen to run before the work
 * that precedes the graph launch on the same stream had finished (walk_aux.h zero_async, round 4) */
__global__ void __launch_bounds__(256) node_zero_kernel(uint32_t *p, uint64_t n32)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (uint64_t)gridDim.x * blockDim.x) p[i] = 0u;
}

static hipError_t node_zero_async(void *p, uint64_t bytes, hipStream_t s)
{
	const uint64_t n32 = bytes / 4u;
	if (n32 == 0) return hipSuccess;
	uint64_t blocks = (n32 + 255u) / 256u;
	if (blocks > 4096u) blocks = 4096u;
	hipLaunchKernelGGL(node_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint32_t *>(p), n32);
	return hipGetLastError();
}

/* accepted inputs of a bitmap slice: one atomic per wavefront */
__global__ void __launch_bounds__(256)
count_bits_kernel(const uint64_t *words, uint64_t nwords, unsigned long long *out)
{
	unsigned long long c = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x)
		c += (unsigned long long)__popcll(words[i]);
	for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
	if ((threadIdx.x & 63u) == 0 && c != 0) atomicAdd(out, c);
}

extern "C" void fsm_hip_node_free(struct fsm_hip_node *nd)
{
	if (nd == nullptr) return;
	int prev = -1;
	(void)hipGetDevice(&prev);
	for (auto &w : nd->workers) {
		if (!w) continue;
		{ std::lock_guard<std::mutex> lk(w->m); w->quit = true; }
		w->cv.notify_all();
		if (w->th.joinable()) w->th.join();
	}
	for (size_t k = 0; k < nd->dev.size(); k++) {          /* an asynchronous call may still be in flight */
		(void)hipSetDevice(nd->dev[k]);
		if (k < nd->stream.size() && nd->stream[k]) (void)hipStreamSynchronize(nd->stream[k]);
		if (k < nd->cstream.size() && nd->cstream[k]) (void)hipStreamSynchronize(nd->cstream[k]);
	}
	for (size_t k = 0; k < nd->comm.size(); k++)
		if (nd->comm[k] != nullptr) (void)R.CommDestroy(nd->comm[k]);
	for (size_t k = 0; k < nd->dev.size(); k++) {
		(void)hipSetDevice(nd->dev[k]);
		if (k < nd->cstream.size() && nd->cstream[k]) (void)hipStreamDestroy(nd->cstream[k]);
		if (k < nd->walked.size() && nd->walked[k]) (void)hipEventDestroy(nd->walked[k]);
		for (size_t q = 2 * k; q < 2 * k + 2 && q < nd->gathered.size(); q++) if (nd->gathered[q]) (void)hipEventDestroy(nd->gathered[q]);
		if (k < nd->stream.size() && nd->stream[k]) (void)hipStreamDestroy(nd->stream[k]);
		if (k < nd->d_count.size() && nd->d_count[k]) (void)hipFree(nd->d_count[k]);
		if (k < nd->dfa.size()) fsm_hip_dfa_free(nd->dfa[k]);
	}
	if (prev >= 0) (void)hipSetDevice(prev);
	delete nd;
}

extern "C" struct fsm_hip_node *fsm_hip_node_create(const struct fsm_hip_dfa_desc *desc, unsigned flags, const int *devices, int ndev)
{
	int have = 0, prev = -1;
	if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) { errno = ENODEV; return nullptr; }
	if (desc == nullptr || ndev < 0 || ndev > 64) { errno = EINVAL; return nullptr; }
	fsm_hip_node *nd = new (std::nothrow) fsm_hip_node();
	if (nd == nullptr) { errno = ENOMEM; return nullptr; }
	if (ndev == 0 || devices == nullptr) {          /* every device of the node */
		for (int k = 0; k < have; k++) nd->dev.push_back(k);
	} else {
		for (int k = 0; k < ndev; k++) {
			if (devices[k] < 0 || devices[k] >= have) { delete nd; errno = EINVAL; return nullptr; }
			nd->dev.push_back(devices[k]);
		}
	}
	(void)hipGetDevice(&prev);
	int err = 0;
	for (size_t k = 0; k < nd->dev.size() && err == 0; k++) {
		hipStream_t s = nullptr;
		unsigned long long *c = nullptr;
		if (hipSetDevice(nd->dev[k]) != hipSuccess) { err = ENODEV; break; }
		fsm_hip_dfa *d = fsm_hip_dfa_create(desc, flags);   /* the table lands on the current device */
		if (d == nullptr) { err = errno ? errno : EIO; break; }
		nd->dfa.push_back(d);
		if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { err = EIO; break; }
		nd->stream.push_back(s);
		hipStream_t cs = nullptr;
		hipEvent_t e0 = nullptr, e1 = nullptr;
		if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { err = EIO; break; }
		nd->cstream.push_back(cs);
		if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess) { err = EIO; break; }
		nd->walked.push_back(e0);
		if (hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) { err = EIO; break; }
		nd->gathered.push_back(e1);
		e1 = nullptr;
		if (hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) { err = EIO; break; }
		nd->gathered.push_back(e1);
		if (hipMalloc((void **)&c, 16) != hipSuccess) { err = ENOMEM; break; }
		nd->d_count.push_back(c);
	}
	if (prev >= 0) (void)hipSetDevice(prev);
	if (err != 0) { fsm_hip_node_free(nd); errno = err; return nullptr; }
	nd->workers.resize(nd->dev.size());
	for (size_t k = 1; k < nd->dev.size(); k++) {
		nd->workers[k].reset(new (std::nothrow) node_worker());
		if (!nd->workers[k]) { fsm_hip_node_free(nd); errno = ENOMEM; return nullptr; }
		nd->workers[k]->th = std::thread(worker_main, nd, k);
	}
	/* RCCL communicators: only for a list of distinct devices */
	bool distinct = true;
	for (size_t i = 0; i < nd->dev.size(); i++)
		for (size_t j = i + 1; j < nd->dev.size(); j++) distinct = distinct && nd->dev[i] != nd->dev[j];
	if (distinct && getenv("FSM_HIP_NO_RCCL") == nullptr) {
		std::call_once(rccl_once, rccl_resolve);
		if (R.state > 0) {
			nd->comm.assign(nd->dev.size(), nullptr);
			if (R.CommInitAll(nd->comm.data(), (int)nd->dev.size(), nd->dev.data()) != 0) nd->comm.clear();
			if (prev >= 0) (void)hipSetDevice(prev);
		}
	}
	return nd;
}

extern "C" int fsm_hip_node_ndev(const struct fsm_hip_node *nd) { return nd == nullptr ? 0 : (int)nd->dev.size(); }

extern "C" int fsm_hip_node_uses_rccl(const struct fsm_hip_node *nd) { return nd != nullptr && !nd->comm.empty(); }

extern "C" const char *fsm_hip_node_rccl_path(void) { return rccl_path; }

extern "C" struct fsm_hip_dfa *fsm_hip_node_dfa(struct fsm_hip_node *nd, int k)
{
	if (nd == nullptr || k < 0 || (size_t)k >= nd->dfa.size()) { errno = EINVAL; return nullptr; }
	return nd->dfa[(size_t)k];
}

/* words of 64 inputs per device: the batch's words split evenly, the last shards may be short or empty */
static size_t words_per_dev(const fsm_hip_node *nd, size_t n)
{
	const size_t words = (n + 63) / 64, g = nd->dev.size();
	return (words + g - 1) / g;
}

extern "C" size_t fsm_hip_node_bitmap_words(const struct fsm_hip_node *nd, size_t n)
{
	return nd == nullptr ? 0 : words_per_dev(nd, n) * nd->dev.size();
}

extern "C" void fsm_hip_node_shard(const struct fsm_hip_node *nd, size_t n, int k, size_t *first, size_t *count)
{
	size_t f = 0, c = 0;
	if (nd != nullptr && k >= 0 && (size_t)k < nd->dev.size()) {
		const size_t per = words_per_dev(nd, n) * 64;
		f = (size_t)k * per < n ? (size_t)k * per : n;
		c = f + per < n ? per : n - f;
	}
	if (first) *first = f;
	if (count) *count = c;
}

/* run fn(k) on device k's parked thread (k >= 1) and fn(0) on the calling thread; returns 0 or the first errno */
static int per_device(fsm_hip_node *nd, const std::function<int(int)> &fn)
{
	const size_t g = nd->dev.size();
	for (size_t k = 1; k < g; k++) {
		node_worker &w = *nd->workers[k];
		{ std::lock_guard<std::mutex> lk(w.m); w.err = 0; w.done = false; w.job = &fn; }
		w.cv.notify_all();
	}
	errno = 0;
	int first = fn(0) != 0 ? (errno ? errno : EIO) : 0;   /* shard 0 on the calling thread */
	for (size_t k = 1; k < g; k++) {
		node_worker &w = *nd->workers[k];
		std::unique_lock<std::mutex> lk(w.m);
		w.cv.wait(lk, [&] { return w.done; });
		if (first == 0 && w.err != 0) first = w.err;
	}
	if (first != 0) { errno = first; return -1; }
	return 0;
}

extern "C" int fsm_hip_node_exec_batch(struct fsm_hip_node *nd,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap)
{
	if (nd == nullptr || (n != 0 && base == nullptr && stride != 0)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	std::lock_guard<std::mutex> lk(nd->mu);
	return per_device(nd, [&](int k) {
		size_t first, count;
		fsm_hip_node_shard(nd, n, k, &first, &count);
		if (count == 0) return 0;
		/* the replica's own host front: H2D of the slice, the walk, D2H into the caller's arrays in place */
		return fsm_hip_exec_batch(nd->dfa[(size_t)k], base + first * stride, stride, len ? len + first : nullptr, count,
		                          end_out ? end_out + first : nullptr, accept_bitmap ? accept_bitmap + first / 64 : nullptr);
	});
}

extern "C" int fsm_hip_node_exec_batch_offsets(struct fsm_hip_node *nd,
	const unsigned char *base, const uint64_t *off, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap)
{
	if (nd == nullptr || (n != 0 && off == nullptr)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	std::lock_guard<std::mutex> lk(nd->mu);
	return per_device(nd, [&](int k) {
		size_t first, count;
		fsm_hip_node_shard(nd, n, k, &first, &count);
		if (count == 0) return 0;
		/* the shard's offsets rebased to its first byte */
		std::vector<uint64_t> o(count + 1);
		for (size_t i = 0; i <= count; i++) {
			if (off[first + i] < off[first]) { errno = EINVAL; return -1; }
			o[i] = off[first + i] - off[first];
		}
		return fsm_hip_exec_batch_offsets(nd->dfa[(size_t)k], base ? base + off[first] : nullptr, o.data(), count,
		                                  end_out ? end_out + first : nullptr, accept_bitmap ? accept_bitmap + first / 64 : nullptr);
	});
}

/* the compact packed forms (u32 offsets below 4 GiB; lengths alone): each shard goes to its replica's own front */
extern "C" int fsm_hip_node_exec_batch_offsets32(struct fsm_hip_node *nd,
	const unsigned char *base, const uint32_t *off32, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap)
{
	if (nd == nullptr || (n != 0 && off32 == nullptr)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	std::lock_guard<std::mutex> lk(nd->mu);
	return per_device(nd, [&](int k) {
		size_t first, count;
		fsm_hip_node_shard(nd, n, k, &first, &count);
		if (count == 0) return 0;
		std::vector<uint32_t> o(count + 1);
		for (size_t i = 0; i <= count; i++) {
			if (off32[first + i] < off32[first]) { errno = EINVAL; return -1; }
			o[i] = off32[first + i] - off32[first];
		}
		return fsm_hip_exec_batch_offsets32(nd->dfa[(size_t)k], base ? base + off32[first] : nullptr, o.data(), count,
		                                    end_out ? end_out + first : nullptr, accept_bitmap ? accept_bitmap + first / 64 : nullptr);
	});
}

extern "C" int fsm_hip_node_exec_batch_lengths(struct fsm_hip_node *nd,
	const unsigned char *base, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap)
{
	if (nd == nullptr || (n != 0 && len == nullptr)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	std::lock_guard<std::mutex> lk(nd->mu);
	/* where each shard's bytes start: one pass over the lengths */
	const size_t g = nd->dev.size();
	std::vector<uint64_t> sbeg(g + 1, 0);
	{
		size_t k = 0, first = 0, count = 0;
		uint64_t run = 0;
		fsm_hip_node_shard(nd, n, 0, &first, &count);
		for (size_t i = 0; i < n; i++) {
			while (k + 1 < g && i >= first + count) { k++; fsm_hip_node_shard(nd, n, (int)k, &first, &count); sbeg[k] = run; }
			run += len[i];
		}
		for (k++; k <= g; k++) sbeg[k] = run;
	}
	return per_device(nd, [&](int k) {
		size_t first, count;
		fsm_hip_node_shard(nd, n, k, &first, &count);
		if (count == 0) return 0;
		return fsm_hip_exec_batch_lengths(nd->dfa[(size_t)k], base ? base + sbeg[(size_t)k] : nullptr, len + first, count,
		                                  end_out ? end_out + first : nullptr, accept_bitmap ? accept_bitmap + first / 64 : nullptr);
	});
}

/* accepted inputs of the whole batch after an exchange: device 0's reduced count (RCCL) or the sum of the devices' */
static int read_count(fsm_hip_node *nd, int slot, unsigned long long *total)
{
	const size_t g = nd->dev.size();
	*total = 0;
	for (size_t k = 0; k < (nd->comm.empty() ? g : 1); k++) {
		unsigned long long c = 0;
		(void)hipSetDevice(nd->dev[k]);
		if (hipMemcpy(&c, nd->d_count[k] + slot, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return -1;
		*total += c;
	}
	return 0;
}

static int sync_all(fsm_hip_node *nd)
{
	bool ok = true;
	for (size_t k = 0; k < nd->dev.size(); k++) {
		(void)hipSetDevice(nd->dev[k]);
		ok = hipStreamSynchronize(nd->stream[k]) == hipSuccess && ok;
		ok = hipStreamSynchronize(nd->cstream[k]) == hipSuccess && ok;
	}
	return ok ? 0 : -1;
}

extern "C" int fsm_hip_node_wait(struct fsm_hip_node *nd, uint64_t *match_count)
{
	if (nd == nullptr) { errno = EINVAL; return -1; }
	std::lock_guard<std::mutex> lk(nd->mu);
	int prev = -1, rc = 0;
	(void)hipGetDevice(&prev);
	if (sync_all(nd) != 0) { errno = EIO; rc = -1; }
	if (rc == 0 && match_count != nullptr) {
		unsigned long long t = 0;
		if (!nd->async_pending || !nd->async_count) { errno = EINVAL; rc = -1; }       /* nothing counted */
		else if (read_count(nd, nd->aslot ^ 1, &t) != 0) { errno = EIO; rc = -1; }
		else *match_count = t;
	}
	nd->async_pending = false;
	if (prev >= 0) (void)hipSetDevice(prev);
	return rc;
}

extern "C" int fsm_hip_node_exec_device(struct fsm_hip_node *nd, const struct fsm_hip_node_batch *b, size_t n,
	uint64_t *match_count, int async)
{
	if (nd == nullptr || b == nullptr || b->d_base == nullptr || (b->d_off == nullptr && b->stride == 0) ||
	    (b->d_off != nullptr && b->d_len != nullptr) || ((match_count != nullptr || b->want_count) && b->d_bitmap_all == nullptr) ||
	    (b->d_id_out != nullptr && b->ids_mode != FSM_HIP_IDS_EARLIEST && b->ids_mode != FSM_HIP_IDS_RET && b->ids_mode != FSM_HIP_IDS_ERROR) ||
	    (async && match_count != nullptr)) { errno = EINVAL; return -1; }
	const size_t g = nd->dev.size();
	if (b->d_bitmap_all != nullptr)
		for (size_t k = 0; k < g; k++)
			if (b->d_bitmap_all[k] == nullptr) { errno = EINVAL; return -1; }   /* every device takes part in the exchange */
	if (n == 0) { if (match_count) *match_count = 0; return 0; }
	std::lock_guard<std::mutex> lk(nd->mu);
	const size_t wpd = words_per_dev(nd, n);
	const bool count = match_count != nullptr || b->want_count;
	const int slot = nd->aslot;
	int prev = -1;
	(void)hipGetDevice(&prev);
	/* 1. every device walks its shard on its own stream, driven by its own host thread */
	int rc = per_device(nd, [&](int k) -> int {
		size_t first, cnt;
		fsm_hip_node_shard(nd, n, k, &first, &cnt);
		if (hipSetDevice(nd->dev[(size_t)k]) != hipSuccess) { errno = ENODEV; return -1; }
		hipStream_t s = nd->stream[(size_t)k];
		/* this slot's buffers were last used two asynchronous calls ago: their collective must be over */
		if (hipStreamWaitEvent(s, nd->gathered[2 * (size_t)k + (size_t)slot], 0) != hipSuccess) { errno = EIO; return -1; }
		uint64_t *slice = b->d_bitmap_all ? b->d_bitmap_all[k] + (size_t)k * wpd : nullptr;
		if (slice != nullptr && cnt < wpd * 64 &&
		    node_zero_async(slice, wpd * sizeof(uint64_t), s) != hipSuccess) { errno = EIO; return -1; }
		fsm_hip_dfa *d = nd->dfa[(size_t)k];
		uint32_t *e_out = b->d_end_out ? b->d_end_out[k] : nullptr;
		if (cnt != 0) {
			/* one walk writes every output asked for (round 3 launched one per output) */
			const int r = fsm_hip_exec_batch_all_device(d, b->d_base[k], b->stride, b->d_off == nullptr && b->d_len ? b->d_len[k] : nullptr,
			                                            b->d_off ? b->d_off[k] : nullptr, cnt, e_out, slice, b->ids_mode,
			                                            b->d_id_out ? b->d_id_out[k] : nullptr, b->d_eager_out ? b->d_eager_out[k] : nullptr, s);
			if (r != 0) return -1;
		}
		if (count) {
			if (node_zero_async(nd->d_count[(size_t)k] + slot, sizeof(unsigned long long), s) != hipSuccess) { errno = EIO; return -1; }
			hipLaunchKernelGGL(count_bits_kernel, dim3(256), dim3(256), 0, s, slice, (uint64_t)wpd, nd->d_count[(size_t)k] + slot);
			if (hipGetLastError() != hipSuccess) { errno = EIO; return -1; }
		}
		if (hipEventRecord(nd->walked[(size_t)k], s) != hipSuccess) { errno = EIO; return -1; }
		return 0;
	});
	const int rc_errno = errno;
	/* 2. the only exchange, on the devices' second streams (an asynchronous call's exchange runs under the next call's
	 * walk): every device gets every slice of the bitmap; the counts are summed */
	bool ok = rc == 0;
	for (size_t k = 0; k < g && ok; k++) {
		(void)hipSetDevice(nd->dev[k]);
		ok = hipStreamWaitEvent(nd->cstream[k], nd->walked[k], 0) == hipSuccess;
	}
	if (ok && !nd->comm.empty()) {
		if (b->d_bitmap_all != nullptr) {
			ok = ok && R.GroupStart() == 0;
			for (size_t k = 0; k < g && ok; k++)
				ok = R.AllGather(b->d_bitmap_all[k] + k * wpd, b->d_bitmap_all[k], wpd, 5 /* ncclUint64 */, nd->comm[k], nd->cstream[k]) == 0;
			ok = R.GroupEnd() == 0 && ok;
		}
		if (ok && count) {
			ok = ok && R.GroupStart() == 0;
			for (size_t k = 0; k < g && ok; k++)
				ok = R.AllReduce(nd->d_count[k] + slot, nd->d_count[k] + slot, 1, 5 /* ncclUint64 */, 0 /* ncclSum */, nd->comm[k], nd->cstream[k]) == 0;
			ok = R.GroupEnd() == 0 && ok;
		}
	} else if (ok && b->d_bitmap_all != nullptr) {
		/* no RCCL (or a device list with repeats): the same exchange as peer-to-peer copies of the slices */
		for (size_t k = 0; k < g && ok; k++) {
			(void)hipSetDevice(nd->dev[k]);
			for (size_t j = 0; j < g && ok; j++) {
				if (j == k || b->d_bitmap_all[j] == b->d_bitmap_all[k]) continue;
				ok = hipMemcpyPeerAsync(b->d_bitmap_all[j] + k * wpd, nd->dev[j], b->d_bitmap_all[k] + k * wpd, nd->dev[k], wpd * sizeof(uint64_t), nd->cstream[k]) == hipSuccess;
			}
		}
	}
	for (size_t k = 0; k < g && ok; k++) {
		(void)hipSetDevice(nd->dev[k]);
		ok = hipEventRecord(nd->gathered[2 * k + (size_t)slot], nd->cstream[k]) == hipSuccess;
	}
	nd->aslot ^= 1;
	nd->async_pending = true;
	nd->async_count = count;
	unsigned long long total = 0;
	if (!ok || !async) {
		/* a failed call, like a synchronous one, returns with nothing in flight: the caller may free its buffers */
		if (sync_all(nd) != 0) ok = false;
		nd->async_pending = false;
		if (ok && match_count != nullptr && read_count(nd, slot, &total) != 0) ok = false;
	}
	if (prev >= 0) (void)hipSetDevice(prev);
	if (!ok) { errno = rc != 0 && rc_errno ? rc_errno : EIO; return -1; }
	if (match_count != nullptr) *match_count = total;
	return 0;
}

extern "C" int fsm_hip_node_exec_batch_device(struct fsm_hip_node *nd,
	const void *const *d_base, size_t stride, size_t n,
	uint32_t *const *d_end_out, uint64_t *const *d_bitmap_all, uint64_t *match_count)
{
	struct fsm_hip_node_batch b;
	memset(&b, 0, sizeof b);
	b.d_base = d_base;
	b.stride = stride;
	b.d_end_out = d_end_out;
	b.d_bitmap_all = d_bitmap_all;
	if (nd == nullptr || d_base == nullptr || stride == 0) { errno = EINVAL; return -1; }
	return fsm_hip_node_exec_device(nd, &b, n, match_count, 0);
}

/* host-pointer ids / eager over the whole node: every device's own host front on its shard, results in place */
extern "C" int fsm_hip_node_exec_batch_ids(struct fsm_hip_node *nd,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n, int mode, uint32_t *id_out)
{
	if (nd == nullptr || id_out == nullptr || (n != 0 && base == nullptr && stride != 0)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	std::lock_guard<std::mutex> lk(nd->mu);
	return per_device(nd, [&](int k) -> int {
		size_t first, count;
		fsm_hip_node_shard(nd, n, k, &first, &count);
		if (count == 0) return 0;
		return fsm_hip_exec_batch_ids(nd->dfa[(size_t)k], base + first * stride, stride, len ? len + first : nullptr, count, mode, id_out + first);
	});
}

extern "C" int fsm_hip_node_exec_batch_eager(struct fsm_hip_node *nd,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n, uint32_t *end_out, uint64_t *eager_out)
{
	if (nd == nullptr || eager_out == nullptr || (n != 0 && base == nullptr && stride != 0)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	std::lock_guard<std::mutex> lk(nd->mu);
	const size_t W = fsm_hip_eager_words(nd->dfa[0]);
	return per_device(nd, [&](int k) -> int {
		size_t first, count;
		fsm_hip_node_shard(nd, n, k, &first, &count);
		if (count == 0) return 0;
		return fsm_hip_exec_batch_eager(nd->dfa[(size_t)k], base + first * stride, stride, len ? len + first : nullptr, count,
		                                end_out ? end_out + first : nullptr, eager_out + first * W);
	});
}

/* many DFAs, sharded by DFA (include/fsm_hip.h) */
extern "C" int fsm_hip_node_exec_multi(struct fsm_hip_node *const *nodes, const struct fsm_hip_multi_batch *b, size_t k)
{
	if (k == 0) return 0;
	if (nodes == nullptr || b == nullptr || nodes[0] == nullptr) { errno = EINVAL; return -1; }
	fsm_hip_node *nd0 = nodes[0];
	const size_t g = nd0->dev.size();
	std::vector<uint64_t> cost(k);
	for (size_t q = 0; q < k; q++) {
		if (nodes[q] == nullptr || nodes[q]->dev != nd0->dev || (b[q].n != 0 && b[q].off == nullptr)) { errno = EINVAL; return -1; }
		cost[q] = b[q].n ? b[q].off[b[q].n] + 64u * (uint64_t)b[q].n : 0u;
	}
	std::vector<int> dev_of(k);
	if (fsm_hip_multi_assign(cost.data(), k, (int)g, dev_of.data()) != 0) return -1;
	std::lock_guard<std::mutex> lk(nd0->mu);
	return per_device(nd0, [&](int dv) {
		std::vector<const fsm_hip_dfa *> dl;
		std::vector<fsm_hip_multi_batch> bl;
		for (size_t q = 0; q < k; q++)
			if (dev_of[q] == dv) { dl.push_back(nodes[q]->dfa[(size_t)dv]); bl.push_back(b[q]); }
		if (dl.empty()) return 0;
		return fsm_hip_exec_multi(dl.data(), bl.data(), dl.size());
	});
}
