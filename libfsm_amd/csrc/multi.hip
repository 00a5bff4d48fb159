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
#include <new>
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
	const uint32_t *fid;     /* [S1] what fsm_hip_exec_batch_ids writes for an input ending there, or null */
	const uint8_t  *base;
	const uint64_t *off;     /* n + 1 */
	uint32_t *end_out;       /* or null */
	uint32_t *id_out;        /* or null */
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
constexpr uint32_t MULTI_WAVES = 4;                    /* wavefronts per workgroup: MULTI_WAVES * 64 consecutive lines of ONE job share a table copy */

__device__ __forceinline__ uint32_t byte_at(const u32x4m &w, int k)
{
	const uint32_t d = (k < 4) ? w.x : (k < 8) ? w.y : (k < 12) ? w.z : w.w;
	return (d >> (8 * (k & 3))) & 0xffu;
}

/* one workgroup = MULTI_WAVES wavefronts = 256 consecutive lines of ONE job.  (Round 5: one wavefront per workgroup -- sized for
 * retest's three lines a record; a job of 1e5 lines then copied its table into LDS once per 64 of them.)  Every pointer of the
 * descriptor is device memory: the loads name that address space (no FLAT instruction: tests/test_abi.py). */
__global__ void __launch_bounds__(MULTI_WAVES * 64)
walk_multi(const MultiJob *jobs, const uint32_t *tile_job)
{
	typedef const uint32_t __attribute__((address_space(1))) *g_u32p;
	typedef const uint64_t __attribute__((address_space(1))) *g_u64p;
	typedef const uint8_t __attribute__((address_space(1))) *g_u8p;
	typedef u32x4m __attribute__((aligned(1))) u32x4_any;
	typedef const u32x4_any __attribute__((address_space(1))) *g_chunkp;
	__shared__ uint32_t cls4[64];
	__shared__ uint16_t cls2[256];                  /* LDS tables: 2 * class of a byte -- the byte offset of its column in a row of u16 */
	__shared__ uint16_t tab[MULTI_LDS_ENTRIES];     /* ... and per (state, class) the BYTE offset of the next state's row */
	const uint32_t ji = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_job[blockIdx.x]);
	const MultiJob &j = jobs[ji];
	const uint32_t tid = threadIdx.x, lane = tid & 63u;
	const uint32_t C = j.C, S1 = j.S1;
	const bool in_lds = j.lds_table != 0u;
	const g_u32p dense = (g_u32p)(uintptr_t)j.dense, fin = (g_u32p)(uintptr_t)j.fin, fid = (g_u32p)(uintptr_t)j.fid;
	const g_u64p off = (g_u64p)(uintptr_t)j.off;
	if (tid < 64u) {
		const uint32_t w4 = ((g_u32p)(uintptr_t)j.cls4)[tid];
		cls4[tid] = w4;
#pragma unroll
		for (int q = 0; q < 4; q++) cls2[tid * 4u + (uint32_t)q] = (uint16_t)(((w4 >> (8 * q)) & 0xffu) * 2u);
	}
	if (in_lds)
		for (uint32_t e = tid; e < S1 * C; e += MULTI_WAVES * 64u) tab[e] = (uint16_t)(dense[e] * C * 2u);   /* (S1 * C <= 16 384 entries: < 2^16 bytes) */
	__syncthreads();

	const uint64_t tile = blockIdx.x - j.tile0, i = tile * (MULTI_WAVES * 64u) + tid;
	const bool valid = i < j.n;
	uint64_t beg = 0, len = 0;
	if (valid) { beg = off[i]; len = off[i + 1] - beg; }
	const uint64_t limit = j.limit != 0u ? j.limit : off[j.n];
	const uint32_t unit = in_lds ? C * 2u : 1u;       /* the walk's state: byte offset of its row (LDS) or state index (global table) */
	const uint32_t absorbing = j.abs_min * unit;
	uint32_t s = j.start * unit;
	const uint64_t p = reinterpret_cast<uint64_t>(j.base) + beg;
	typedef const uint16_t __attribute__((address_space(3))) *l_u16p;
	const uint32_t cls2_at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)cls2;
	const uint32_t tab_at = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)tab;

	for (uint64_t t = 0;; t += 16u) {
		const bool live = t < len && s < absorbing;
		if (!__any(live)) break;
		const uint32_t cnt = !live ? 0u : len - t < 16u ? (uint32_t)(len - t) : 16u;
		u32x4m w = {0u, 0u, 0u, 0u};
		if (live) {
			if (beg + t + 16u <= limit) {
				w = *(g_chunkp)(p + t);
			} else {
				uint32_t d[4] = {0u, 0u, 0u, 0u};
				for (uint32_t k = 0; k < cnt; k++) d[k >> 2] |= (uint32_t)((g_u8p)(p + t))[k] << ((k & 3u) * 8u);
				w = u32x4m{d[0], d[1], d[2], d[3]};
			}
		}
		if (in_lds) {
			/* two LDS reads per byte: the byte's column offset (state-independent: all sixteen asked for at once), then the row */
			uint32_t c2[16];
#pragma unroll
			for (int k = 0; k < 16; k++) c2[k] = *(l_u16p)(uintptr_t)(cls2_at + byte_at(w, k) * 2u);
			if (__all(cnt == 16u || cnt == 0u)) {          /* whole chunks everywhere (lines of one length, the middle of long ones) */
				uint32_t sn = s;
#pragma unroll
				for (int k = 0; k < 16; k++) sn = *(l_u16p)(uintptr_t)(tab_at + sn + c2[k]);
				s = cnt != 0u ? sn : s;
			} else {
#pragma unroll
				for (int k = 0; k < 16; k++) {
					const uint32_t sn = *(l_u16p)(uintptr_t)(tab_at + s + c2[k]);
					s = (uint32_t)k < cnt ? sn : s;
				}
			}
		} else {
#pragma unroll
			for (int k = 0; k < 16; k++) {
				if ((uint32_t)k < cnt) {
					const uint32_t b = byte_at(w, k);
					const uint32_t c = (cls4[b >> 2] >> ((b & 3u) * 8u)) & 0xffu;
					s = dense[(uint64_t)s * C + c];
				}
			}
		}
	}
	const uint32_t fs = in_lds ? s / (C * 2u) : s;
	uint32_t end = FSM_HIP_NO_MATCH;
	if (valid) end = fin[fs];
	typedef uint32_t __attribute__((address_space(1))) *g_u32w;
	typedef uint64_t __attribute__((address_space(1))) *g_u64w;
	if (valid && j.end_out != nullptr) ((g_u32w)(uintptr_t)j.end_out)[i] = end;
	if (valid && j.id_out != nullptr) ((g_u32w)(uintptr_t)j.id_out)[i] = fid[fs];
	const uint64_t m = __ballot(valid && end != FSM_HIP_NO_MATCH);
	if (j.bitmap != nullptr && lane == 0u && (tile * MULTI_WAVES + (tid >> 6)) * 64u < j.n) ((g_u64w)(uintptr_t)j.bitmap)[tile * MULTI_WAVES + (tid >> 6)] = m;
}

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
	std::vector<size_t> dense, cls, fin, fid, off, text, end, bm, ids;   /* per fused job; (size_t)-1: not staged */
	size_t jobs = 0, tiles = 0, in_end = 0, total = 0;
	uint32_t ntiles = 0;
};

/* host front: a job small enough to stage (its lines ride in the one copy); device front: any job whose plain table fits the
 * kernel's LDS copy -- nothing of it is staged but the table, and MULTI_WAVES * 64 lines share each copy (round 5 fused on the
 * line count alone there and sent 1e5-line jobs, one launch each, through their dfa's own walk; a job with a bigger table
 * still goes that way: its planned layout beats a plain table in L2) */
bool fusable(const Plan *p, size_t n, uint64_t bytes, bool host)
{
	if (n == 0 || p->S1 == 0 || p->dense.size() != (size_t)p->S1 * p->C || p->dense.size() * 4u > MULTI_FUSE_TABLE) return false;
	if (host) return n <= MULTI_FUSE_LINES && bytes <= MULTI_FUSE_BYTES;
	return p->dense.size() <= MULTI_LDS_ENTRIES && n < ((size_t)1 << 31);
}

/* the automaton's side of fused job f: its plain table, byte classes, end states and (asked for) ids into the host block `pin`
 * at the layout's offsets, the descriptor pointing at the same offsets of the device block `dev` */
void fill_tables(unsigned char *pin, unsigned char *dev, const Lay &L, size_t f, const Plan *p, const std::vector<uint32_t> &fid, MultiJob &j)
{
	memcpy(pin + L.dense[f], p->dense.data(), p->dense.size() * 4u);
	uint32_t *c4 = reinterpret_cast<uint32_t *>(pin + L.cls[f]);
	for (unsigned w = 0; w < 64; w++)
		c4[w] = (uint32_t)p->cls[4 * w] | ((uint32_t)p->cls[4 * w + 1] << 8) | ((uint32_t)p->cls[4 * w + 2] << 16) | ((uint32_t)p->cls[4 * w + 3] << 24);
	memcpy(pin + L.fin[f], p->fin.data(), (size_t)p->S1 * 4u);
	if (L.fid[f] != (size_t)-1) memcpy(pin + L.fid[f], fid.data(), (size_t)p->S1 * 4u);
	memset(&j, 0, sizeof j);
	j.fid = L.fid[f] != (size_t)-1 ? reinterpret_cast<const uint32_t *>(dev + L.fid[f]) : nullptr;
	j.dense = reinterpret_cast<const uint32_t *>(dev + L.dense[f]);
	j.cls4 = reinterpret_cast<const uint32_t *>(dev + L.cls[f]);
	j.fin = reinterpret_cast<const uint32_t *>(dev + L.fin[f]);
	j.C = p->C; j.S1 = p->S1; j.start = p->start; j.abs_min = p->abs_min;
	j.lds_table = p->dense.size() <= MULTI_LDS_ENTRIES ? 1u : 0u;
}

/* one job as the entry points hand it over (ids: optional) */
struct JobView {
	const unsigned char *base;
	const uint64_t *off;
	size_t n;
	uint32_t *end_out;
	uint64_t *accept_bitmap;
	uint32_t *id_out;
};

/* Run the jobs idx[] (all on device `device`) of a submission.  host = true: b[] holds host pointers (lines and results are
 * staged); false: device pointers, launched on `stream` and not waited for. */
int run_device_group(int device, const struct fsm_hip_dfa *const *dfa, const JobView *b, int ids_mode, const std::vector<size_t> &idx,
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
	std::vector<std::vector<uint32_t>> fids;      /* per fused job: ids by renumbered state (empty: none asked for) */
	Lay L;
	size_t o = 0;
	{
		size_t staged = 0;
		for (size_t q : idx) {
			const Plan *p = dfa_plan(dfa[q]);
			const uint64_t bytes = b[q].n ? (host ? b[q].off[b[q].n] : 0) : 0;
			if (b[q].n == 0) continue;
			const size_t need = up16(p->dense.size() * 4u) + 256u + 2u * up16((size_t)p->S1 * 4u) +
				(host ? up16((b[q].n + 1) * 8u) + up16((size_t)bytes + 16u) + 2u * up16(b[q].n * 4u) + up16(((b[q].n + 63u) / 64u) * 8u) : 0u);
			if (fusable(p, b[q].n, bytes, host) && staged + need <= MULTI_STAGE_CAP) { fj.push_back(q); staged += need; }
			else single.push_back(q);
		}
	}
	if (!fj.empty()) {
		const size_t kf = fj.size();
		uint64_t tiles = 0;
		for (size_t q : fj) tiles += (b[q].n + MULTI_WAVES * 64u - 1u) / (MULTI_WAVES * 64u);
		if (tiles > 0x7FFFFFFFu) { errno = EINVAL; return -1; }
		fids.resize(kf);
		for (size_t f = 0; f < kf; f++)
			if (b[fj[f]].id_out != nullptr && dfa_ids_by_state(dfa[fj[f]], ids_mode, fids[f], nullptr) != 0) return -1;
		L.ntiles = (uint32_t)tiles;
		L.jobs = o; o += up16(kf * sizeof(MultiJob));
		L.tiles = o; o += up16((size_t)tiles * 4u);
		L.dense.resize(kf); L.cls.resize(kf); L.fin.resize(kf); L.fid.assign(kf, (size_t)-1); L.off.assign(kf, (size_t)-1); L.text.assign(kf, (size_t)-1);
		L.end.assign(kf, (size_t)-1); L.bm.assign(kf, (size_t)-1); L.ids.assign(kf, (size_t)-1);
		for (size_t f = 0; f < kf; f++) {
			const Plan *p = dfa_plan(dfa[fj[f]]);
			L.dense[f] = o; o += up16(p->dense.size() * 4u);
			L.cls[f] = o; o += 256u;
			L.fin[f] = o; o += up16((size_t)p->S1 * 4u);
			if (!fids[f].empty()) { L.fid[f] = o; o += up16((size_t)p->S1 * 4u); }
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
				if (b[fj[f]].id_out) { L.ids[f] = o; o += up16(n * 4u); }
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
			MultiJob &j = jobs[f];
			fill_tables(cx.pin, cx.dev, L, f, p, fids[f], j);
			if (host) {
				const size_t bytes = (size_t)b[q].off[n];
				memcpy(cx.pin + L.off[f], b[q].off, (n + 1) * 8u);
				if (bytes) memcpy(cx.pin + L.text[f], b[q].base, bytes);
				memset(cx.pin + L.text[f] + bytes, 0, 16);
				j.base = cx.dev + L.text[f];
				j.off = reinterpret_cast<const uint64_t *>(cx.dev + L.off[f]);
				j.end_out = L.end[f] != (size_t)-1 ? reinterpret_cast<uint32_t *>(cx.dev + L.end[f]) : nullptr;
				j.id_out = L.ids[f] != (size_t)-1 ? reinterpret_cast<uint32_t *>(cx.dev + L.ids[f]) : nullptr;
				j.bitmap = L.bm[f] != (size_t)-1 ? reinterpret_cast<uint64_t *>(cx.dev + L.bm[f]) : nullptr;
				j.limit = bytes + 16u;    /* the staged text is padded: whole 16-byte loads everywhere */
			} else {
				j.base = b[q].base;
				j.off = b[q].off;
				j.end_out = b[q].end_out;
				j.id_out = b[q].id_out;
				j.bitmap = b[q].accept_bitmap;
				j.limit = 0;              /* the kernel reads off[n] */
			}
			j.n = n;
			j.tile0 = t0;
			const uint32_t nt = (uint32_t)((n + MULTI_WAVES * 64u - 1u) / (MULTI_WAVES * 64u));
			for (uint32_t t = 0; t < nt; t++) tile_job[t0 + t] = (uint32_t)f;
			t0 += nt;
		}
		hipStream_t s = host ? cx.s : stream;
		MTRY(hipMemcpyAsync(cx.dev, cx.pin, L.in_end, hipMemcpyHostToDevice, s));
		hipLaunchKernelGGL(walk_multi, dim3(L.ntiles), dim3(MULTI_WAVES * 64u), 0, s, reinterpret_cast<const MultiJob *>(cx.dev + L.jobs),
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
		int r;
		if (b[q].id_out == nullptr)
			r = host ? fsm_hip_exec_batch_offsets(dfa[q], b[q].base, b[q].off, b[q].n, b[q].end_out, b[q].accept_bitmap)
			         : fsm_hip_exec_batch_offsets_device(dfa[q], b[q].base, b[q].off, b[q].n, b[q].end_out, b[q].accept_bitmap, stream);
		else   /* every output of the job from ONE walk (fsm_hip_exec_batch_packed_all*) */
			r = host ? fsm_hip_exec_batch_packed_all(dfa[q], b[q].base, FSM_HIP_META_OFF64, b[q].off, b[q].n, b[q].end_out, b[q].accept_bitmap, ids_mode, b[q].id_out, nullptr)
			         : fsm_hip_exec_batch_packed_all_device(dfa[q], b[q].base, FSM_HIP_META_OFF64, b[q].off, b[q].n, b[q].end_out, b[q].accept_bitmap, ids_mode, b[q].id_out, nullptr, stream);
		if (r != 0) { if (host && !fj.empty()) (void)hipStreamSynchronize(cx.s); return -1; }
		(*launches)++;
	}
	if (host && !fj.empty()) {
		MTRY(hipStreamSynchronize(cx.s));
		for (size_t f = 0; f < fj.size(); f++) {
			const size_t q = fj[f], n = b[q].n;
			if (L.end[f] != (size_t)-1) memcpy(b[q].end_out, cx.pin + L.end[f], n * 4u);
			if (L.ids[f] != (size_t)-1) memcpy(b[q].id_out, cx.pin + L.ids[f], n * 4u);
			if (L.bm[f] != (size_t)-1) memcpy(b[q].accept_bitmap, cx.pin + L.bm[f], ((n + 63u) / 64u) * 8u);
		}
	}
	return 0;
}

/* what every form of a submission checks before anything is launched; *ids_mode: ERROR -> EARLIEST once no job is ambiguous */
int check_jobs(const struct fsm_hip_dfa *const *dfa, const JobView *b, int *ids_mode_io, size_t k, bool host)
{
	int ids_mode = *ids_mode_io;
	if (dfa == nullptr || b == nullptr) { errno = EINVAL; return -1; }
	bool want_ids = false;
	for (size_t q = 0; q < k; q++) want_ids = want_ids || b[q].id_out != nullptr;
	if (want_ids) {
		if (ids_mode != FSM_HIP_IDS_EARLIEST && ids_mode != FSM_HIP_IDS_RET && ids_mode != FSM_HIP_IDS_ERROR) { errno = EINVAL; return -1; }
		if (ids_mode == FSM_HIP_IDS_ERROR) {
			/* AMBIG_ERROR: an end state with more than one id is refused before anything is launched (as fsm_hip_exec_batch_ids) */
			std::vector<uint32_t> tmp;
			for (size_t q = 0; q < k; q++) {
				uint32_t cf = FSM_HIP_NO_MATCH;
				if (dfa[q] == nullptr) { errno = EINVAL; return -1; }
				if (b[q].id_out != nullptr && (dfa_ids_by_state(dfa[q], FSM_HIP_IDS_EARLIEST, tmp, &cf) != 0 || cf != FSM_HIP_NO_MATCH)) { errno = EINVAL; return -1; }
			}
			ids_mode = FSM_HIP_IDS_EARLIEST;
		}
	}
	for (size_t q = 0; q < k; q++) {
		if (dfa[q] == nullptr || (b[q].n != 0 && b[q].off == nullptr)) { errno = EINVAL; return -1; }
		if (host && b[q].n != 0) {
			for (size_t i = 0; i < b[q].n; i++)
				if (b[q].off[i + 1] < b[q].off[i]) { errno = EINVAL; return -1; }
			if (b[q].off[b[q].n] != 0 && b[q].base == nullptr) { errno = EINVAL; return -1; }
		}
	}
	*ids_mode_io = ids_mode;
	return 0;
}

int exec_multi(const struct fsm_hip_dfa *const *dfa, const JobView *b, int ids_mode, size_t k, bool host, hipStream_t stream)
{
	if (k == 0) { g_last_launches = 0; g_last_fused_jobs = 0; return 0; }
	if (check_jobs(dfa, b, &ids_mode, k, host) != 0) return -1;
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
		if (run_device_group(dv, dfa, b, ids_mode, idx, host, stream, &launches, &fused) != 0) return -1;
	}
	g_last_launches = launches;
	g_last_fused_jobs = fused;
	return 0;
}

} // namespace

static std::vector<JobView> views(const struct fsm_hip_multi_batch *b, size_t k)
{
	std::vector<JobView> v(b ? k : 0);
	for (size_t q = 0; q < v.size(); q++) v[q] = JobView{b[q].base, b[q].off, b[q].n, b[q].end_out, b[q].accept_bitmap, nullptr};
	return v;
}
static std::vector<JobView> views(const struct fsm_hip_multi_batch_ids *b, size_t k)
{
	std::vector<JobView> v(b ? k : 0);
	for (size_t q = 0; q < v.size(); q++) v[q] = JobView{b[q].base, b[q].off, b[q].n, b[q].end_out, b[q].accept_bitmap, b[q].id_out};
	return v;
}

extern "C" int fsm_hip_exec_multi(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch *b, size_t k)
{
	const std::vector<JobView> v = views(b, k);
	return exec_multi(dfa, k && b ? v.data() : nullptr, 0, k, true, nullptr);
}

extern "C" int fsm_hip_exec_multi_device(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch *b, size_t k, void *hip_stream)
{
	const std::vector<JobView> v = views(b, k);
	return exec_multi(dfa, k && b ? v.data() : nullptr, 0, k, false, static_cast<hipStream_t>(hip_stream));
}

extern "C" int fsm_hip_exec_multi_ids(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch_ids *b, size_t k, int ids_mode)
{
	const std::vector<JobView> v = views(b, k);
	return exec_multi(dfa, k && b ? v.data() : nullptr, ids_mode, k, true, nullptr);
}

extern "C" int fsm_hip_exec_multi_ids_device(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch_ids *b, size_t k, int ids_mode, void *hip_stream)
{
	const std::vector<JobView> v = views(b, k);
	return exec_multi(dfa, k && b ? v.data() : nullptr, ids_mode, k, false, static_cast<hipStream_t>(hip_stream));
}

/*
 * The PREPARED form: a submission of device-resident jobs whose descriptors, tile map and tables are put on the device ONCE.
 * fsm_hip_multi_launch is then one kernel launch (plus one per job whose table is too big to fuse: its dfa's own device front)
 * on the caller's stream -- no copy, no allocation, no wait: it can be captured into a HIP graph and replayed on whatever the
 * jobs' buffers hold by then (reperf's loop over the same matcher, src/retest/reperf.c:772-784, for K matchers at once).
 */
struct fsm_hip_multi_prepared {
	int device = 0, ids_mode = FSM_HIP_IDS_EARLIEST;
	unsigned char *dev = nullptr;
	size_t jobs_off = 0, tiles_off = 0;
	uint32_t ntiles = 0;
	unsigned fused = 0;
	struct Single { const struct fsm_hip_dfa *dfa; JobView v; };
	std::vector<Single> singles;
};

extern "C" int fsm_hip_multi_prepare(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch_ids *b, size_t k, int ids_mode,
	struct fsm_hip_multi_prepared **out)
{
	if (out == nullptr) { errno = EINVAL; return -1; }
	*out = nullptr;
	std::vector<JobView> v = views(b, k);
	if (k != 0 && check_jobs(dfa, v.data(), &ids_mode, k, false) != 0) return -1;
	for (size_t q = 1; q < k; q++)
		if (dfa_device(dfa[q]) != dfa_device(dfa[0])) { errno = EINVAL; return -1; }      /* one stream, one device (fsm_hip_node_exec_multi shards) */
	fsm_hip_multi_prepared *pp = new (std::nothrow) fsm_hip_multi_prepared;
	if (pp == nullptr) { errno = ENOMEM; return -1; }
	pp->ids_mode = ids_mode;
	pp->device = k != 0 ? dfa_device(dfa[0]) : 0;
	std::vector<size_t> fj;
	for (size_t q = 0; q < k; q++) {
		if (v[q].n == 0) continue;
		if (fusable(dfa_plan(dfa[q]), v[q].n, 0, false)) fj.push_back(q);
		else pp->singles.push_back({dfa[q], v[q]});
	}
	if (!fj.empty()) {
		const size_t kf = fj.size();
		uint64_t tiles = 0;
		for (size_t q : fj) tiles += (v[q].n + MULTI_WAVES * 64u - 1u) / (MULTI_WAVES * 64u);
		if (tiles > 0x7FFFFFFFu) { delete pp; errno = EINVAL; return -1; }
		std::vector<std::vector<uint32_t>> fids(kf);
		for (size_t f = 0; f < kf; f++)
			if (v[fj[f]].id_out != nullptr && dfa_ids_by_state(dfa[fj[f]], ids_mode, fids[f], nullptr) != 0) { delete pp; return -1; }
		Lay L;
		size_t o = 0;
		L.jobs = o; o += up16(kf * sizeof(MultiJob));
		L.tiles = o; o += up16((size_t)tiles * 4u);
		L.dense.resize(kf); L.cls.resize(kf); L.fin.resize(kf); L.fid.assign(kf, (size_t)-1);
		for (size_t f = 0; f < kf; f++) {
			const Plan *p = dfa_plan(dfa[fj[f]]);
			L.dense[f] = o; o += up16(p->dense.size() * 4u);
			L.cls[f] = o; o += 256u;
			L.fin[f] = o; o += up16((size_t)p->S1 * 4u);
			if (!fids[f].empty()) { L.fid[f] = o; o += up16((size_t)p->S1 * 4u); }
		}
		int prev = -1;
		(void)hipGetDevice(&prev);
		struct Restore { int prev, dev; ~Restore() { if (prev >= 0 && prev != dev) { int e = errno; (void)hipSetDevice(prev); errno = e; } } } restore{prev, pp->device};
		hipError_t e = prev != pp->device ? hipSetDevice(pp->device) : hipSuccess;
		std::vector<unsigned char> host(o);
		if (e == hipSuccess) e = hipMalloc((void **)&pp->dev, o);
		if (e != hipSuccess) { delete pp; errno = hip_errno_(e); return -1; }
		MultiJob *jobs = reinterpret_cast<MultiJob *>(host.data() + L.jobs);
		uint32_t *tile_job = reinterpret_cast<uint32_t *>(host.data() + L.tiles);
		uint32_t t0 = 0;
		for (size_t f = 0; f < kf; f++) {
			const size_t q = fj[f];
			MultiJob &j = jobs[f];
			fill_tables(host.data(), pp->dev, L, f, dfa_plan(dfa[q]), fids[f], j);
			j.base = v[q].base; j.off = v[q].off;
			j.end_out = v[q].end_out; j.id_out = v[q].id_out; j.bitmap = v[q].accept_bitmap;
			j.limit = 0;
			j.n = v[q].n;
			j.tile0 = t0;
			const uint32_t nt = (uint32_t)((v[q].n + MULTI_WAVES * 64u - 1u) / (MULTI_WAVES * 64u));
			for (uint32_t t = 0; t < nt; t++) tile_job[t0 + t] = (uint32_t)f;
			t0 += nt;
		}
		e = hipMemcpy(pp->dev, host.data(), o, hipMemcpyHostToDevice);
		if (e != hipSuccess) { (void)hipFree(pp->dev); delete pp; errno = hip_errno_(e); return -1; }
		pp->jobs_off = L.jobs; pp->tiles_off = L.tiles; pp->ntiles = (uint32_t)tiles; pp->fused = (unsigned)kf;
	}
	*out = pp;
	return 0;
}

extern "C" int fsm_hip_multi_launch(const struct fsm_hip_multi_prepared *pp, void *hip_stream)
{
	if (pp == nullptr) { errno = EINVAL; return -1; }
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	int prev = -1;
	(void)hipGetDevice(&prev);
	if (prev != pp->device) MTRY(hipSetDevice(pp->device));
	struct Restore { int prev, dev; ~Restore() { if (prev >= 0 && prev != dev) { int e = errno; (void)hipSetDevice(prev); errno = e; } } } restore{prev, pp->device};
	unsigned launches = 0;
	if (pp->ntiles != 0) {
		hipLaunchKernelGGL(walk_multi, dim3(pp->ntiles), dim3(MULTI_WAVES * 64u), 0, s, reinterpret_cast<const MultiJob *>(pp->dev + pp->jobs_off),
		                   reinterpret_cast<const uint32_t *>(pp->dev + pp->tiles_off));
		MTRY(hipGetLastError());
		launches++;
	}
	for (const auto &sg : pp->singles) {
		const JobView &j = sg.v;
		const int r = j.id_out == nullptr
			? fsm_hip_exec_batch_offsets_device(sg.dfa, j.base, j.off, j.n, j.end_out, j.accept_bitmap, s)
			: fsm_hip_exec_batch_packed_all_device(sg.dfa, j.base, FSM_HIP_META_OFF64, j.off, j.n, j.end_out, j.accept_bitmap, pp->ids_mode, j.id_out, nullptr, s);
		if (r != 0) return -1;
		launches++;
	}
	g_last_launches = launches;
	g_last_fused_jobs = pp->fused;
	return 0;
}

extern "C" void fsm_hip_multi_prepared_free(struct fsm_hip_multi_prepared *pp)
{
	if (pp == nullptr) return;
	if (pp->dev != nullptr) {
		int prev = -1;
		(void)hipGetDevice(&prev);
		if (prev != pp->device) (void)hipSetDevice(pp->device);
		(void)hipFree(pp->dev);           /* (waits for the device: a launch still reading the block ends first) */
		if (prev >= 0 && prev != pp->device) (void)hipSetDevice(prev);
	}
	delete pp;
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
