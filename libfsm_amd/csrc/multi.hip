/*
 * multi.hip -- the many-DFA front: K automata, each with its own packed lines and outputs, ONE submission.
 *
 * The reference's batched driver compiles a DFA per record and runs a handful of lines through it
 * (src/retest/main.c:1056-1058 fsm_runner_initialize + fsm_free, :1114 fsm_runner_run; tests/retest/ *.tst:
 * 37 DFAs x ~3 lines).  One table upload + one launch per DFA is 37 x (a dozen synchronous copies + >= 23 us
 * of launch latency) for microseconds of walking.  Here every job that is small rides in ONE host-to-device
 * copy -- descriptors, the automata's plain tables (Plan::dense, the dfa_table form of src/libfsm/vm/ir.c:649-750:
 * next state per byte class, missing edge = DEAD), offsets and lines -- ONE kernel whose workgroups map to
 * (dfa, tile of 64 lines), and ONE copy back.  A job too big for that (more than MULTI_FUSE_LINES lines or
 * MULTI_FUSE_BYTES bytes) goes through its dfa's own walk kernels, enqueued beside the fused launch.
 *
 * The walk is fsm_exec's (src/libfsm/exec.c:132-151): state = table[state][class(byte)] from the start state;
 * a lane stops at an absorbing state (DEAD = the missing edge, exec.c:133-138, or an accept-everything state),
 * end_out = fin[state] (the caller's state id, or NO_MATCH when the final state is not an end state, :153-155).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <vector>

#include "../../include/fsm_hip.h"
#include "dfa_access.h"

using namespace fsmhip;

namespace {

constexpr uint32_t MULTI_LDS_ENTRIES = 16384;          /* a table of up to this many (state, class) entries is walked from LDS (u16 row offsets) */
constexpr size_t MULTI_FUSE_LINES = 65536, MULTI_FUSE_BYTES = (size_t)1 << 20, MULTI_FUSE_TABLE = (size_t)1 << 20;
constexpr size_t MULTI_STAGE_CAP = (size_t)64 << 20;   /* one submission's staging; what does not fit goes the per-dfa way */

struct MultiJob {
	const uint32_t *dense;   /* [S1][C] next (renumbered) state */
	const uint32_t *cls4;    /* [64] byte -> class, four to a word */
	const uint32_t *fin;     /* [S1] caller's end state id or NO_MATCH */
	const uint8_t  *base;
	const uint64_t *off;     /* n + 1 */
	uint32_t *end_out;       /* or null */
	uint64_t *bitmap;        /* or null */
	uint64_t n;
	uint64_t limit;          /* bytes of this job's text that may be read (>= off[n]); 0: off[n] */
	uint32_t C, S1, start, abs_min;
	uint32_t tile0;          /* first workgroup of this job */
	uint32_t lds_table;
	uint32_t pad[4];
};
static_assert(sizeof(MultiJob) % 16 == 0, "descriptors are read as aligned records");

typedef uint32_t u32x4m __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t byte_at(const u32x4m &w, int k)
{
	const uint32_t d = (k < 4) ? w.x : (k < 8) ? w.y : (k < 12) ? w.z : w.w;
	return (d >> (8 * (k & 3))) & 0xffu;
}

/* one workgroup = one wavefront = 64 consecutive lines of ONE job */
__global__ void __launch_bounds__(64)
walk_multi(const MultiJob *jobs, const uint32_t *tile_job)
{
	__shared__ uint32_t cls4[64];
	__shared__ uint16_t tab[MULTI_LDS_ENTRIES];
	const uint32_t ji = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_job[blockIdx.x]);
	const MultiJob &j = jobs[ji];
	const uint32_t lane = threadIdx.x;
	const uint32_t C = j.C, S1 = j.S1;
	const bool in_lds = j.lds_table != 0u;
	cls4[lane] = j.cls4[lane];
	if (in_lds)
		for (uint32_t e = lane; e < S1 * C; e += 64u) tab[e] = (uint16_t)(j.dense[e] * C);   /* row offset of the next state */
	__syncthreads();

	const uint64_t tile = blockIdx.x - j.tile0, i = tile * 64u + lane;
	const bool valid = i < j.n;
	uint64_t beg = 0, len = 0;
	if (valid) { beg = j.off[i]; len = j.off[i + 1] - beg; }
	const uint64_t limit = j.limit != 0u ? j.limit : j.off[j.n];
	const uint32_t unit = in_lds ? C : 1u;            /* the walk's state: row offset (LDS) or state index (global table) */
	const uint32_t absorbing = j.abs_min * unit;
	uint32_t s = j.start * unit;
	const uint8_t *p = j.base + beg;
	typedef u32x4m __attribute__((aligned(1))) u32x4_any;

	for (uint64_t t = 0; __any(t < len && s < absorbing); t += 16u) {
		if (!(t < len && s < absorbing)) continue;
		const uint32_t cnt = len - t < 16u ? (uint32_t)(len - t) : 16u;
		u32x4m w = {0u, 0u, 0u, 0u};
		if (beg + t + 16u <= limit) {
			w = *reinterpret_cast<const u32x4_any *>(p + t);
		} else {
			uint32_t d[4] = {0u, 0u, 0u, 0u};
			for (uint32_t k = 0; k < cnt; k++) d[k >> 2] |= (uint32_t)p[t + k] << ((k & 3u) * 8u);
			w = u32x4m{d[0], d[1], d[2], d[3]};
		}
#pragma unroll
		for (int k = 0; k < 16; k++) {
			if ((uint32_t)k < cnt) {
				const uint32_t b = byte_at(w, k);
				const uint32_t c = (cls4[b >> 2] >> ((b & 3u) * 8u)) & 0xffu;
				s = in_lds ? (uint32_t)tab[s + c] : j.dense[(uint64_t)s * C + c];
			}
		}
	}
	uint32_t end = FSM_HIP_NO_MATCH;
	if (valid) end = j.fin[in_lds ? s / C : s];
	if (valid && j.end_out != nullptr) j.end_out[i] = end;
	const uint64_t m = __ballot(valid && end != FSM_HIP_NO_MATCH);
	if (j.bitmap != nullptr && lane == 0u) j.bitmap[tile] = m;
}

/* per-device staging of the fused launch: one pinned block, one device block of the same layout */
struct MultiCtx {
	std::mutex mu;
	hipStream_t s = nullptr;
	unsigned char *pin = nullptr, *dev = nullptr;
	size_t pin_bytes = 0, dev_bytes = 0;
	hipEvent_t ev = nullptr;
	bool busy = false;           /* a device-pointer call's launch may still read the blocks */
};
constexpr int MAXDEV = 64;
MultiCtx g_ctx[MAXDEV];
std::atomic<unsigned> g_last_launches{0}, g_last_fused_jobs{0};

int hip_errno_(hipError_t e)
{
	switch (e) {
	case hipSuccess: return 0;
	case hipErrorOutOfMemory: return ENOMEM;
	case hipErrorNoDevice:
	case hipErrorInvalidDevice: return ENODEV;
	case hipErrorInvalidValue: return EINVAL;
	default: return EIO;
	}
}

#define MTRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	if (getenv("FSM_HIP_DEBUG")) fprintf(stderr, "fsm_hip multi: %s -> %s\n", #expr, hipGetErrorString(e_)); \
	errno = hip_errno_(e_); return -1; } } while (0)

size_t up16(size_t x) { return (x + 15u) & ~(size_t)15u; }

int ctx_reserve(MultiCtx &cx, size_t bytes)
{
	if (cx.s == nullptr) MTRY(hipStreamCreateWithFlags(&cx.s, hipStreamNonBlocking));
	if (cx.ev == nullptr) MTRY(hipEventCreateWithFlags(&cx.ev, hipEventDisableTiming));
	if (cx.busy) { MTRY(hipEventSynchronize(cx.ev)); cx.busy = false; }
	if (bytes > cx.pin_bytes) {
		size_t want = (size_t)1 << 16;
		while (want < bytes) want *= 2;
		if (cx.pin) (void)hipHostFree(cx.pin);
		if (cx.dev) (void)hipFree(cx.dev);
		cx.pin = cx.dev = nullptr;
		cx.pin_bytes = cx.dev_bytes = 0;
		MTRY(hipHostMalloc((void **)&cx.pin, want, hipHostMallocDefault));
		MTRY(hipMalloc((void **)&cx.dev, want));
		cx.pin_bytes = cx.dev_bytes = want;
	}
	return 0;
}

/* where everything of a fused submission lies in the staging block (the same offsets on both sides) */
struct Lay {
	std::vector<size_t> dense, cls, fin, off, text, end, bm;   /* per fused job; (size_t)-1: not staged */
	size_t jobs = 0, tiles = 0, in_end = 0, total = 0;
	uint32_t ntiles = 0;
};

bool fusable(const Plan *p, size_t n, uint64_t bytes)
{
	return n != 0 && n <= MULTI_FUSE_LINES && bytes <= MULTI_FUSE_BYTES && p->S1 != 0 && p->dense.size() == (size_t)p->S1 * p->C &&
	       p->dense.size() * 4u <= MULTI_FUSE_TABLE;
}

/* Run the jobs idx[] (all on device `device`) of a submission.  host = true: b[] holds host pointers (lines and results are
 * staged); false: device pointers, launched on `stream` and not waited for. */
int run_device_group(int device, const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch *b, const std::vector<size_t> &idx,
	bool host, hipStream_t stream, unsigned *launches, unsigned *fused_jobs)
{
	if (device < 0 || device >= MAXDEV) { errno = ENODEV; return -1; }
	int prev = -1;
	(void)hipGetDevice(&prev);
	if (prev != device) MTRY(hipSetDevice(device));
	struct Restore { int prev, dev; ~Restore() { if (prev >= 0 && prev != dev) { int e = errno; (void)hipSetDevice(prev); errno = e; } } } restore{prev, device};

	MultiCtx &cx = g_ctx[device];
	std::lock_guard<std::mutex> lk(cx.mu);

	/* which jobs ride in the fused launch */
	std::vector<size_t> fj, single;
	Lay L;
	size_t o = 0;
	{
		size_t staged = 0;
		for (size_t q : idx) {
			const Plan *p = dfa_plan(dfa[q]);
			const uint64_t bytes = b[q].n ? (host ? b[q].off[b[q].n] : 0) : 0;
			if (b[q].n == 0) continue;
			const size_t need = up16(p->dense.size() * 4u) + 256u + up16((size_t)p->S1 * 4u) +
				(host ? up16((b[q].n + 1) * 8u) + up16((size_t)bytes + 16u) + up16(b[q].n * 4u) + up16(((b[q].n + 63u) / 64u) * 8u) : 0u);
			if (fusable(p, b[q].n, bytes) && staged + need <= MULTI_STAGE_CAP) { fj.push_back(q); staged += need; }
			else single.push_back(q);
		}
	}
	if (!fj.empty()) {
		const size_t kf = fj.size();
		uint64_t tiles = 0;
		for (size_t q : fj) tiles += (b[q].n + 63u) / 64u;
		if (tiles > 0x7FFFFFFFu) { errno = EINVAL; return -1; }
		L.ntiles = (uint32_t)tiles;
		L.jobs = o; o += up16(kf * sizeof(MultiJob));
		L.tiles = o; o += up16((size_t)tiles * 4u);
		L.dense.resize(kf); L.cls.resize(kf); L.fin.resize(kf); L.off.assign(kf, (size_t)-1); L.text.assign(kf, (size_t)-1);
		L.end.assign(kf, (size_t)-1); L.bm.assign(kf, (size_t)-1);
		for (size_t f = 0; f < kf; f++) {
			const Plan *p = dfa_plan(dfa[fj[f]]);
			L.dense[f] = o; o += up16(p->dense.size() * 4u);
			L.cls[f] = o; o += 256u;
			L.fin[f] = o; o += up16((size_t)p->S1 * 4u);
			if (host) {
				const size_t n = b[fj[f]].n;
				L.off[f] = o; o += up16((n + 1) * 8u);
				L.text[f] = o; o += up16((size_t)b[fj[f]].off[n] + 16u);
			}
		}
		L.in_end = o;
		if (host)
			for (size_t f = 0; f < kf; f++) {
				const size_t n = b[fj[f]].n;
				if (b[fj[f]].end_out) { L.end[f] = o; o += up16(n * 4u); }
				if (b[fj[f]].accept_bitmap) { L.bm[f] = o; o += up16(((n + 63u) / 64u) * 8u); }
			}
		L.total = o;
		if (ctx_reserve(cx, L.total) != 0) return -1;

		/* fill the pinned block */
		MultiJob *jobs = reinterpret_cast<MultiJob *>(cx.pin + L.jobs);
		uint32_t *tile_job = reinterpret_cast<uint32_t *>(cx.pin + L.tiles);
		uint32_t t0 = 0;
		for (size_t f = 0; f < kf; f++) {
			const size_t q = fj[f];
			const Plan *p = dfa_plan(dfa[q]);
			const size_t n = b[q].n;
			memcpy(cx.pin + L.dense[f], p->dense.data(), p->dense.size() * 4u);
			uint32_t *c4 = reinterpret_cast<uint32_t *>(cx.pin + L.cls[f]);
			for (unsigned w = 0; w < 64; w++)
				c4[w] = (uint32_t)p->cls[4 * w] | ((uint32_t)p->cls[4 * w + 1] << 8) | ((uint32_t)p->cls[4 * w + 2] << 16) | ((uint32_t)p->cls[4 * w + 3] << 24);
			memcpy(cx.pin + L.fin[f], p->fin.data(), (size_t)p->S1 * 4u);
			MultiJob &j = jobs[f];
			memset(&j, 0, sizeof j);
			j.dense = reinterpret_cast<const uint32_t *>(cx.dev + L.dense[f]);
			j.cls4 = reinterpret_cast<const uint32_t *>(cx.dev + L.cls[f]);
			j.fin = reinterpret_cast<const uint32_t *>(cx.dev + L.fin[f]);
			if (host) {
				const size_t bytes = (size_t)b[q].off[n];
				memcpy(cx.pin + L.off[f], b[q].off, (n + 1) * 8u);
				if (bytes) memcpy(cx.pin + L.text[f], b[q].base, bytes);
				memset(cx.pin + L.text[f] + bytes, 0, 16);
				j.base = cx.dev + L.text[f];
				j.off = reinterpret_cast<const uint64_t *>(cx.dev + L.off[f]);
				j.end_out = L.end[f] != (size_t)-1 ? reinterpret_cast<uint32_t *>(cx.dev + L.end[f]) : nullptr;
				j.bitmap = L.bm[f] != (size_t)-1 ? reinterpret_cast<uint64_t *>(cx.dev + L.bm[f]) : nullptr;
				j.limit = bytes + 16u;    /* the staged text is padded: whole 16-byte loads everywhere */
			} else {
				j.base = b[q].base;
				j.off = b[q].off;
				j.end_out = b[q].end_out;
				j.bitmap = b[q].accept_bitmap;
				j.limit = 0;              /* the kernel reads off[n] */
			}
			j.n = n;
			j.C = p->C; j.S1 = p->S1; j.start = p->start; j.abs_min = p->abs_min;
			j.tile0 = t0;
			j.lds_table = p->dense.size() <= MULTI_LDS_ENTRIES ? 1u : 0u;
			const uint32_t nt = (uint32_t)((n + 63u) / 64u);
			for (uint32_t t = 0; t < nt; t++) tile_job[t0 + t] = (uint32_t)f;
			t0 += nt;
		}
		hipStream_t s = host ? cx.s : stream;
		MTRY(hipMemcpyAsync(cx.dev, cx.pin, L.in_end, hipMemcpyHostToDevice, s));
		hipLaunchKernelGGL(walk_multi, dim3(L.ntiles), dim3(64), 0, s, reinterpret_cast<const MultiJob *>(cx.dev + L.jobs),
		                   reinterpret_cast<const uint32_t *>(cx.dev + L.tiles));
		MTRY(hipGetLastError());
		(*launches)++;
		*fused_jobs += (unsigned)kf;
		if (host) {
			if (L.total > L.in_end) MTRY(hipMemcpyAsync(cx.pin + L.in_end, cx.dev + L.in_end, L.total - L.in_end, hipMemcpyDeviceToHost, s));
		} else {
			MTRY(hipEventRecord(cx.ev, s));
			cx.busy = true;
		}
	}
	/* the big ones: each dfa's own walk, beside the fused launch (host: its synchronous front; device: enqueued on the stream) */
	for (size_t q : single) {
		const int r = host ? fsm_hip_exec_batch_offsets(dfa[q], b[q].base, b[q].off, b[q].n, b[q].end_out, b[q].accept_bitmap)
		                   : fsm_hip_exec_batch_offsets_device(dfa[q], b[q].base, b[q].off, b[q].n, b[q].end_out, b[q].accept_bitmap, stream);
		if (r != 0) { if (host && !fj.empty()) (void)hipStreamSynchronize(cx.s); return -1; }
		(*launches)++;
	}
	if (host && !fj.empty()) {
		MTRY(hipStreamSynchronize(cx.s));
		for (size_t f = 0; f < fj.size(); f++) {
			const size_t q = fj[f], n = b[q].n;
			if (L.end[f] != (size_t)-1) memcpy(b[q].end_out, cx.pin + L.end[f], n * 4u);
			if (L.bm[f] != (size_t)-1) memcpy(b[q].accept_bitmap, cx.pin + L.bm[f], ((n + 63u) / 64u) * 8u);
		}
	}
	return 0;
}

int exec_multi(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch *b, size_t k, bool host, hipStream_t stream)
{
	if (k == 0) { g_last_launches = 0; g_last_fused_jobs = 0; return 0; }
	if (dfa == nullptr || b == nullptr) { errno = EINVAL; return -1; }
	for (size_t q = 0; q < k; q++) {
		if (dfa[q] == nullptr || (b[q].n != 0 && b[q].off == nullptr)) { errno = EINVAL; return -1; }
		if (host && b[q].n != 0) {
			for (size_t i = 0; i < b[q].n; i++)
				if (b[q].off[i + 1] < b[q].off[i]) { errno = EINVAL; return -1; }
			if (b[q].off[b[q].n] != 0 && b[q].base == nullptr) { errno = EINVAL; return -1; }
		}
	}
	/* jobs by device (a submission usually has one) */
	std::vector<int> devs;
	for (size_t q = 0; q < k; q++) {
		const int dv = dfa_device(dfa[q]);
		if (std::find(devs.begin(), devs.end(), dv) == devs.end()) devs.push_back(dv);
	}
	unsigned launches = 0, fused = 0;
	for (int dv : devs) {
		std::vector<size_t> idx;
		for (size_t q = 0; q < k; q++) if (dfa_device(dfa[q]) == dv) idx.push_back(q);
		if (run_device_group(dv, dfa, b, idx, host, stream, &launches, &fused) != 0) return -1;
	}
	g_last_launches = launches;
	g_last_fused_jobs = fused;
	return 0;
}

} // namespace

extern "C" int fsm_hip_exec_multi(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch *b, size_t k)
{
	return exec_multi(dfa, b, k, true, nullptr);
}

extern "C" int fsm_hip_exec_multi_device(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch *b, size_t k, void *hip_stream)
{
	return exec_multi(dfa, b, k, false, static_cast<hipStream_t>(hip_stream));
}

extern "C" unsigned fsm_hip_multi_last_launches(void) { return g_last_launches.load(); }
extern "C" unsigned fsm_hip_multi_last_fused_jobs(void) { return g_last_fused_jobs.load(); }

/* Which device runs which job of a many-DFA submission ("multi-DFA batches shard by DFA"): largest first,
 * each to the device with the least work so far; ties go to the lower device, equal costs keep their order.  Pure host
 * arithmetic: every rank of a multi-process run computes the same split. */
extern "C" int fsm_hip_multi_assign(const uint64_t *cost, size_t k, int ndev, int *dev_of)
{
	if (ndev <= 0 || (k != 0 && (cost == nullptr || dev_of == nullptr))) { errno = EINVAL; return -1; }
	std::vector<size_t> order(k);
	std::iota(order.begin(), order.end(), (size_t)0);
	std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return cost[x] > cost[y]; });
	std::vector<uint64_t> load((size_t)ndev, 0);
	for (size_t q : order) {
		int best = 0;
		for (int g = 1; g < ndev; g++) if (load[(size_t)g] < load[(size_t)best]) best = g;
		dev_of[q] = best;
		load[(size_t)best] += cost[q] ? cost[q] : 1u;
	}
	return 0;
}
