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
}

struct fsm_hip_node {
	std::vector<int> dev;
	std::vector<fsm_hip_dfa *> dfa;
	std::vector<hipStream_t> stream;
	std::vector<unsigned long long *> d_count;   /* one u64 per device */
	std::vector<ncclComm_t> comm;                /* empty: exchange by peer copies */
	std::mutex mu;                               /* one batch at a time per node */
};

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
	for (size_t k = 0; k < nd->comm.size(); k++)
		if (nd->comm[k] != nullptr) (void)R.CommDestroy(nd->comm[k]);
	for (size_t k = 0; k < nd->dev.size(); k++) {
		(void)hipSetDevice(nd->dev[k]);
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
		if (hipMalloc((void **)&c, 16) != hipSuccess) { err = ENOMEM; break; }
		nd->d_count.push_back(c);
	}
	if (prev >= 0) (void)hipSetDevice(prev);
	if (err != 0) { fsm_hip_node_free(nd); errno = err; return nullptr; }
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

/* run fn(k) on one host thread per device; returns 0 or the first errno */
template <class F>
static int per_device(fsm_hip_node *nd, F fn)
{
	const size_t g = nd->dev.size();
	std::vector<int> err(g, 0);
	std::vector<std::thread> th;
	th.reserve(g);
	for (size_t k = 1; k < g; k++)
		th.emplace_back([&, k] { errno = 0; if (fn((int)k) != 0) err[k] = errno ? errno : EIO; });
	errno = 0;
	if (fn(0) != 0) err[0] = errno ? errno : EIO;   /* shard 0 on the calling thread */
	for (auto &t : th) t.join();
	for (size_t k = 0; k < g; k++)
		if (err[k] != 0) { errno = err[k]; return -1; }
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

extern "C" int fsm_hip_node_exec_batch_device(struct fsm_hip_node *nd,
	const void *const *d_base, size_t stride, size_t n,
	uint32_t *const *d_end_out, uint64_t *const *d_bitmap_all, uint64_t *match_count)
{
	if (nd == nullptr || d_base == nullptr || stride == 0 || (match_count != nullptr && d_bitmap_all == nullptr)) { errno = EINVAL; return -1; }
	if (n == 0) { if (match_count) *match_count = 0; return 0; }
	std::lock_guard<std::mutex> lk(nd->mu);
	const size_t g = nd->dev.size(), wpd = words_per_dev(nd, n);
	int prev = -1;
	(void)hipGetDevice(&prev);
	/* 1. every device walks its shard on its own stream, driven by its own host thread */
	int rc = per_device(nd, [&](int k) {
		size_t first, count;
		fsm_hip_node_shard(nd, n, k, &first, &count);
		if (hipSetDevice(nd->dev[(size_t)k]) != hipSuccess) { errno = ENODEV; return -1; }
		hipStream_t s = nd->stream[(size_t)k];
		uint64_t *slice = d_bitmap_all ? d_bitmap_all[k] + (size_t)k * wpd : nullptr;
		if (slice != nullptr && count < wpd * 64 &&
		    hipMemsetAsync(slice, 0, wpd * sizeof(uint64_t), s) != hipSuccess) { errno = EIO; return -1; }
		if (count != 0 &&
		    fsm_hip_exec_batch_device(nd->dfa[(size_t)k], d_base[k], stride, nullptr, count,
		                              d_end_out ? d_end_out[k] : nullptr, slice, s) != 0) return -1;
		if (match_count != nullptr) {
			if (hipMemsetAsync(nd->d_count[(size_t)k], 0, sizeof(unsigned long long), s) != hipSuccess) { errno = EIO; return -1; }
			hipLaunchKernelGGL(count_bits_kernel, dim3(256), dim3(256), 0, s, slice, (uint64_t)wpd, nd->d_count[(size_t)k]);
			if (hipGetLastError() != hipSuccess) { errno = EIO; return -1; }
		}
		return 0;
	});
	/* 2. the only exchange: every device gets every slice of the bitmap; the counts are summed */
	unsigned long long total = 0;
	if (rc == 0 && !nd->comm.empty()) {
		bool ok = true;
		if (d_bitmap_all != nullptr) {
			ok = ok && R.GroupStart() == 0;
			for (size_t k = 0; k < g && ok; k++)
				ok = R.AllGather(d_bitmap_all[k] + k * wpd, d_bitmap_all[k], wpd, 5 /* ncclUint64 */, nd->comm[k], nd->stream[k]) == 0;
			ok = R.GroupEnd() == 0 && ok;
		}
		if (ok && match_count != nullptr) {
			ok = ok && R.GroupStart() == 0;
			for (size_t k = 0; k < g && ok; k++)
				ok = R.AllReduce(nd->d_count[k], nd->d_count[k], 1, 5 /* ncclUint64 */, 0 /* ncclSum */, nd->comm[k], nd->stream[k]) == 0;
			ok = R.GroupEnd() == 0 && ok;
		}
		for (size_t k = 0; k < g; k++) {
			(void)hipSetDevice(nd->dev[k]);
			ok = hipStreamSynchronize(nd->stream[k]) == hipSuccess && ok;
		}
		if (ok && match_count != nullptr) {
			(void)hipSetDevice(nd->dev[0]);
			ok = hipMemcpy(&total, nd->d_count[0], sizeof total, hipMemcpyDeviceToHost) == hipSuccess;
		}
		if (!ok) { errno = EIO; rc = -1; }
	} else if (rc == 0) {
		/* no RCCL (or a device list with repeats): the same exchange as peer-to-peer copies of the slices */
		bool ok = true;
		for (size_t k = 0; k < g; k++) {
			(void)hipSetDevice(nd->dev[k]);
			ok = hipStreamSynchronize(nd->stream[k]) == hipSuccess && ok;
		}
		for (size_t k = 0; k < g && ok; k++) {
			if (match_count != nullptr) {
				unsigned long long c = 0;
				(void)hipSetDevice(nd->dev[k]);
				ok = hipMemcpy(&c, nd->d_count[k], sizeof c, hipMemcpyDeviceToHost) == hipSuccess && ok;
				total += c;
			}
			for (size_t j = 0; j < g && ok && d_bitmap_all != nullptr; j++) {
				if (j == k || d_bitmap_all[j] == d_bitmap_all[k]) continue;
				ok = hipMemcpyPeer(d_bitmap_all[j] + k * wpd, nd->dev[j], d_bitmap_all[k] + k * wpd, nd->dev[k], wpd * sizeof(uint64_t)) == hipSuccess;
			}
		}
		if (!ok) { errno = EIO; rc = -1; }
	}
	if (prev >= 0) (void)hipSetDevice(prev);
	if (rc == 0 && match_count != nullptr) *match_count = total;
	return rc;
}
