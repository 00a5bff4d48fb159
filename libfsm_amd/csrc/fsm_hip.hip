/*
 * fsm_hip.hip -- C ABI of libfsm_hip.so (core layer): table upload, kernel
 * selection/launch, host staging, end-id lookup, synthetic generator.
 * See include/fsm_hip.h for the contract of every entry point and the
 * reference interface (file:line) each one replaces.
 */
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cxxabi.h>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "../../include/fsm_hip.h"
#include "../../include/fsm_hip_plan.h"
#include "plan.h"
#include "dfa_access.h"
#include "launch.h"
#include "gen_kernels.h"
#include "walk_aux.h"
#include "walk_lazy.h"   /* FSMHIP_LAZY_PIECE (the kernels themselves are instantiated in kern_glob.hip) */
#include "trace_kernel.h"

using namespace fsmhip;

/* ------------------------------------------------------------------ */

struct fsm_hip_dfa {
	Plan plan;
	int device = 0;
	int ncu = 256;
	uint32_t lds_limit = 160u * 1024u;
	void *d_tab = nullptr;
	uint32_t *d_fin = nullptr;
	uint32_t *d_btab = nullptr;
	uint32_t *d_lazy = nullptr;                      /* sparse layout: the lazy image (plan.cpp build_lazy), if the automaton has one */
	uint32_t *d_lazy_ctr = nullptr;                  /* ... and a ring of tile counters: a launch zeroes and uses the next one (launches on
	                                                  * several streams may be in flight; LAZY_CTRS of them never are) */
	unsigned lazy_ctr_next = 0;
	int knob_lazy_lines = 1;                         /* the lazy walk also serves the variable-length fronts and resumed walks (0: walk_ragged / walk_generic over the records, for A/B runs) */
	int knob_lazy_dyn = 1;                           /* the lazy walk's wavefronts claim their tiles from a counter (0: static striding) */
	/* device end-id delivery (built on first use) */
	std::vector<uint32_t> fin_host;                 /* copy of the fin table uploaded to d_fin */
	uint32_t *d_fin_earliest = nullptr, *d_fin_ret = nullptr;
	std::vector<uint32_t> ret_off, ret_ids;          /* de-duplicated id sets, CSR */
	bool ids_ready = false;
	uint32_t ids_conflict = FSM_HIP_NO_MATCH;        /* lowest end state carrying more than one id */
	/* resume tables (built on first use) */
	uint32_t *d_enc_of = nullptr, *d_orig_of = nullptr;
	std::vector<uint32_t> enc_host;                  /* [S1] encoded state per renumbered state */
	bool resume_ready = false;
	uint64_t *d_emask = nullptr;                     /* eager-output masks, indexed like fin */
	uint32_t *d_ew_off = nullptr, *d_ew_word = nullptr; /* wide eager sets (> 64 ids) */
	uint64_t *d_ew_mask = nullptr;
	/* the eager-output stream (trace_kernel.h; built on first use): plain renumbered table, byte classes, id lists (CSR), fin */
	uint32_t *d_tr_dense = nullptr, *d_tr_cls4 = nullptr, *d_tr_eoff = nullptr, *d_tr_eids = nullptr, *d_tr_fin = nullptr;
	bool trace_ready = false;
	/* device-side choice between walk_generic and walk_ragged: a ring of flags, one per launch (launches on several streams
	 * may be in flight; PICK_FLAGS of them never are), allocated with the dfa */
	uint32_t *d_pick = nullptr;
	unsigned pick_next = 0;
	/* the lengths-only front: tile bases (u64 per 64 inputs) + block totals: one grow-only block per dfa.  Calls are enqueued
	 * under the dfa's lock; every call waits for the block's last user's event, so the block is never shared by two
	 * launches in flight (waiting on one's own stream costs nothing) */
	unsigned char *tb_scratch = nullptr;
	size_t tb_scratch_bytes = 0;
	hipEvent_t tb_scratch_ev = nullptr;
	bool tb_scratch_busy = false;
	std::vector<void *> tb_scratch_old;              /* outgrown blocks: freed with the dfa */
	unsigned char *arena = nullptr;                  /* device scratch of the host-pointer front */
	size_t arena_bytes = 0;
	unsigned char *stage = nullptr;                  /* pinned host staging for small calls */
	hipStream_t hs = nullptr;                        /* private stream of the host-pointer fronts */
	/* guards everything an exec call mutates: arena / stage, the timing events, the lazily built
	 * end-id and resume tables.  Host-pointer fronts hold it for the whole call (they share the
	 * arena), device-pointer fronts only while they enqueue. */
	std::recursive_mutex mu;
	WalkArgs proto;
	uint32_t table_lds = 0;      /* LDS bytes of the policy's tables */
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	bool timed = false;
	std::string last_kernel;     /* demangled name of the walk kernel of the last launch */
	std::string last_kernel_pick[3];   /* a device-side pick launched several (indexed by PICK_*): which one RAN is read from the flag on demand */
	const uint32_t *last_pick_flag = nullptr;
	/* tuning knobs (fsm_hip_dfa_tune) */
	int knob_input_mode = -1;    /* -1 auto */
	int knob_nb = 0;             /* 0 auto */
	uint32_t glob_row_bytes = 4;
	bool glob16 = false;         /* GLOBAL layout, <= 65 535 states: 2-byte entries (Glob16Pol) */
	uint64_t glob_tab_bytes = 0;
	int knob_seg = 0;            /* 0 auto (128) */
	int knob_prefetch = -1;      /* -1 auto (on) */
	int knob_nt = -1;            /* -1 auto */
	int knob_rows = 0;           /* the lazy walk's inputs per lane; 0 auto */
	int knob_waves = 0;          /* 0 auto */
	int knob_blocks_per_cu = 0;  /* 0 auto */
	int knob_early = -1;         /* -1: from flags */
	int knob_noskip = 0;         /* 1: chunk skip off (measurement) */
	int knob_sparse_fast = 1;    /* sparse layout: entry-as-state walk (0: the id-as-state chain loop, for A/B runs) */
	bool sparse_fast_ok = true;  /* the record array sits inside one 4 GiB window (SparseFastPol::enter) */
	int knob_pick_mean = -1;     /* variable-length batches whose mean input length is below this many bytes go to a per-lane kernel;
	                              * -1: 128 where walk_lines32 is the per-lane kernel, 96 for walk_generic (pick_mean_of()) */
	unsigned flags = 0;
	/* The pair table (lds2) wins on fixed-stride rows but leaves no LDS for the ragged kernel's tiles: a dfa planned that way keeps a
	 * SECOND automaton image (lds / combself / ...) for its variable-length, unaligned and resumed batches */
	fsm_hip_dfa *alt = nullptr;
	std::atomic<const fsm_hip_dfa *> last_used{nullptr};   /* which of the two the last launch went to (timing / kernel name); written by concurrent callers */
	std::atomic<bool> uploaded{false};   /* the layout's tables are on the device (FSM_HIP_DEFER_UPLOAD: not before the first single-dfa call);
	                                      * release-stored under mu once everything is in place, acquire-loaded without it */
	int upload_errno = 0;        /* a failed upload is not retried over its partial allocations (fsm_hip_dfa_free releases them): every later call fails with this */
};

/* GLOBAL layout: how much of the table head (rows nearest the start state) every workgroup
 * keeps in LDS.  The walk is bound by L2 gather requests and every lookup served from LDS is one
 * less: on a 6.7 MB table 0 / 40 / 80 / 120 KiB of hot rows measured 265 / 314 / 350 / 360 GB/s
 * (profiles/r01_c5_global_hot.txt), so the default takes what LDS offers. */
static void set_hot_bytes(fsm_hip_dfa *d, uint32_t want)
{
	uint64_t hot = want;
	if (hot > d->glob_tab_bytes) hot = d->glob_tab_bytes;
	/* (the 4-byte table leaves room for eight wavefronts' LDS-DMA tiles; the 2-byte one is walked with per-lane loads and takes
	 * all of LDS: every row there is a step that stays out of L2) */
	const uint32_t keep = d->glob16 ? 64u : 8u * 4096u;
	if (hot + lds_bytes_btab() + keep > d->lds_limit) hot = d->lds_limit - lds_bytes_btab() - keep;
	hot -= hot % d->glob_row_bytes;
	d->proto.tab_bytes = (uint32_t)hot;
	d->table_lds = GlobPol::lds_bytes((uint32_t)hot);
}

static const unsigned LAZY_CTRS = 64, PICK_FLAGS = 64;

static int hip_errno(hipError_t e)
{
	switch (e) {
	case hipSuccess: return 0;
	case hipErrorOutOfMemory: return ENOMEM;
	case hipErrorNoDevice:
	case hipErrorInvalidDevice:
	case hipErrorInsufficientDriver: return ENODEV;
	case hipErrorInvalidValue: return EINVAL;
	default: return EIO;
	}
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	if (getenv("FSM_HIP_DEBUG")) fprintf(stderr, "fsm_hip: %s -> %s\n", #expr, hipGetErrorString(e_)); \
	errno = hip_errno(e_); goto fail; } } while (0)

/* make the dfa's device current for the duration of a call and give the caller's device back */
struct DevGuard {
	int prev = -1;
	bool good = true;
	explicit DevGuard(int dev)
	{
		if (hipGetDevice(&prev) != hipSuccess) prev = -1;
		if (prev != dev && hipSetDevice(dev) != hipSuccess) good = false;
		if (prev == dev) prev = -1;
	}
	~DevGuard() { if (prev >= 0 && good) { int e = errno; (void)hipSetDevice(prev); errno = e; } }
	bool ok() const { return good; }
};

typedef std::lock_guard<std::recursive_mutex> DfaLock;

/* the automaton image a batch goes to: the second one (fsm_hip_dfa::alt) for everything the fixed-stride kernels do not take */
static const fsm_hip_dfa *route(const fsm_hip_dfa *d, bool fixed_stride_fast)
{
	const fsm_hip_dfa *t = d->alt != nullptr && !fixed_stride_fast ? d->alt : d;
	const_cast<fsm_hip_dfa *>(d)->last_used.store(t, std::memory_order_relaxed);
	return t;
}

extern "C" int fsm_hip_version(void) { return 210; }

/* for multi.hip (dfa_access.h): the host-side plan and the device of a dfa */
namespace fsmhip {
const Plan *dfa_plan(const fsm_hip_dfa *d) { return &d->plan; }
int dfa_device(const fsm_hip_dfa *d) { return d->device; }
int dfa_ncu(const fsm_hip_dfa *d) { return d->ncu; }
}

/* ------------------------------------------------------------------ */
/* create / free / info                                               */
/* ------------------------------------------------------------------ */

template <class T>
static hipError_t upload(T **dst, const std::vector<T> &src)
{
	*dst = nullptr;
	size_t bytes = src.size() * sizeof(T);
	if (bytes == 0) bytes = sizeof(T);
	hipError_t e = hipMalloc((void **)dst, (bytes + 15) & ~(size_t)15);
	if (e != hipSuccess) return e;
	if (!src.empty()) e = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
	return e;
}

/* the layout's device image + the per-dfa launch resources (events, private stream).  fsm_hip_dfa_create does this at once
 * unless FSM_HIP_DEFER_UPLOAD asks to wait for the first call that needs it: a dfa that is only ever used through
 * fsm_hip_exec_multi (retest: a new DFA per record, a few lines each) never needs it -- its plain table rides in that call's
 * one host-to-device copy */
static int dfa_upload(fsm_hip_dfa *d)
{
	const unsigned flags = d->flags;
	{
		Plan &p = d->plan;
		WalkArgs &a = d->proto;
		memset(&a, 0, sizeof a);
		std::vector<uint32_t> btab(256);
		switch (p.layout) {
		case FSM_HIP_LAYOUT_TINY: {
			if (!p.tiny5_col.empty()) {   /* <= 6 states: Tiny5Pol, state code = 5 * state */
				uint32_t *t = nullptr;
				HIP_TRY(upload(&t, p.tiny5_col));
				d->d_tab = t;
				HIP_TRY(upload(&d->d_fin, p.fin));
				a.tab_bytes = 256 * 4;
				a.start = p.start * 5u;
				a.abs_min = p.abs_min * 5u;
				a.fin_div = 5;
				d->table_lds = Tiny5Pol::lds_bytes(0);
				break;
			}
			uint64_t *t = nullptr;
			HIP_TRY(upload(&t, p.tiny_col));
			d->d_tab = t;
			HIP_TRY(upload(&d->d_fin, p.fin));
			a.tab_bytes = 256 * 8;
			a.start = p.start;
			a.abs_min = p.abs_min;
			a.fin_div = 1;
			/* 7..16 states: 64-bit columns; a 32-bit format for <= 8 states measured slower on the same box
			 * (4.95-5.06 vs 5.10-5.61 TB/s, profiles/r01_tiny_by_states.txt) and was dropped */
			d->table_lds = TinyPol<uint64_t>::lds_bytes(0);
			break;
		}
		case FSM_HIP_LAYOUT_COMBSELF: {
			/* image: comb64[n] = {entry, smask(next)}, dsm[32] (mask of each class's default state), rng16[n]
			 * (self-loop byte range by row offset) -- these three parts go to LDS (tab_bytes) -- then smask[n]
			 * by row offset (global only: seeds a walk) */
			uint32_t *t = nullptr;
			const size_t n = p.comb.size(), rw = (n + 1) / 2;   /* rng16[n] in u32 words */
			std::vector<uint32_t> img(2 * n + 64 + rw + n, 0);
			for (size_t k = 0; k < n; k++) {
				img[2 * k] = p.comb[k];
				const uint32_t nxt = p.comb[k] & 0xffffu;
				img[2 * k + 1] = (p.comb[k] >> 16) != 0xFFFFu && nxt < n ? p.comb_smask[nxt] : 0u;
			}
			for (uint32_t c = 0; c < p.C; c++) {
				img[2 * n + 2 * c] = p.comb_dflt[c];
				img[2 * n + 2 * c + 1] = p.comb_smask[p.comb_dflt[c]];
			}
			for (size_t k = 0; k < n; k++) img[2 * n + 64 + k / 2] |= (uint32_t)p.comb_rng[k] << (16 * (k & 1));
			for (size_t k = 0; k < n; k++) img[2 * n + 64 + rw + k] = p.comb_smask[k];
			HIP_TRY(upload(&t, img));
			d->d_tab = t;
			HIP_TRY(upload(&d->d_fin, p.comb_fin));
			for (int b = 0; b < 256; b++) btab[b] = p.cls[b];
			a.tab_bytes = (uint32_t)((2 * n + 64 + rw) * 4);
			a.dflt = (uint32_t)n;   /* entries: tells the kernel where dsm[] and rng16[] start */
			a.start = p.comb_off[p.start];
			a.abs_min = p.comb_abs_min_off;
			a.fin_div = 1;
			d->table_lds = CombSelfPol::lds_bytes(a.tab_bytes);
			break;
		}
		case FSM_HIP_LAYOUT_COMB256: {
			uint32_t *t = nullptr;
			HIP_TRY(upload(&t, p.comb256));
			d->d_tab = t;
			HIP_TRY(upload(&d->d_fin, p.comb256_fin));
			a.tab_bytes = (uint32_t)(p.comb256.size() * 4);
			a.start = p.comb256_off[p.start];
			a.abs_min = p.comb256_abs_min_off;
			a.dflt = p.comb256_dflt;
			a.fin_div = 1;
			d->table_lds = Comb256Pol::lds_bytes(a.tab_bytes);
			break;
		}
		case FSM_HIP_LAYOUT_LDS: {
			uint16_t *t = nullptr;
			HIP_TRY(upload(&t, p.lds_tab));
			d->d_tab = t;
			HIP_TRY(upload(&d->d_fin, p.fin));
			for (int b = 0; b < 256; b++) btab[b] = p.cls[b];
			a.tab_bytes = (uint32_t)(p.lds_tab.size() * 2);
			a.start = p.start * p.row_bytes;
			a.abs_min = p.abs_min * p.row_bytes;
			a.fin_div = p.row_bytes;
			d->table_lds = lds_bytes_btab() + ((a.tab_bytes + 15u) & ~15u);
			break;
		}
		case FSM_HIP_LAYOUT_LDS2: {
			uint16_t *t = nullptr;
			HIP_TRY(upload(&t, p.lds_tab));
			d->d_tab = t;
			HIP_TRY(upload(&d->d_fin, p.fin));
			for (int b = 0; b < 256; b++) btab[b] = p.cls[b];
			const uint32_t row = p.lds2_c1 * p.lds2_c1;      /* entries per state: the walk's state is state * row */
			a.tab_bytes = (uint32_t)(p.lds_tab.size() * 2);
			a.start = p.start * row;
			a.abs_min = p.abs_min * row;
			a.fin_div = row;
			a.dflt = p.lds2_c1;
			d->table_lds = Lds2Pol::lds_bytes(a.tab_bytes);
			break;
		}
		case FSM_HIP_LAYOUT_LDSSELF: {
			uint16_t *t = nullptr;
			HIP_TRY(upload(&t, p.lds_tab));
			d->d_tab = t;
			HIP_TRY(upload(&d->d_fin, p.fin));
			for (int b = 0; b < 256; b++) btab[b] = p.cls[b];
			a.tab_bytes = (uint32_t)(p.lds_tab.size() * 2);
			a.start = p.start * p.row_bytes;
			a.abs_min = p.abs_min * p.row_bytes;
			a.fin_div = p.row_bytes;
			d->table_lds = LdsSelfPol::lds_bytes(a.tab_bytes);
			break;
		}
		case FSM_HIP_LAYOUT_COMB: {
			uint32_t *t = nullptr;
			std::vector<uint32_t> img(p.comb);               /* image = comb[n], dflt[256] */
			img.resize(p.comb.size() + 256, 0);
			for (uint32_t c = 0; c < p.C; c++) img[p.comb.size() + c] = p.comb_dflt[c];
			HIP_TRY(upload(&t, img));
			d->d_tab = t;
			HIP_TRY(upload(&d->d_fin, p.comb_fin));
			for (int b = 0; b < 256; b++) btab[b] = p.cls[b];
			a.tab_bytes = (uint32_t)(img.size() * 4);
			a.start = p.comb_off[p.start];
			a.abs_min = p.comb_abs_min_off;
			a.fin_div = 1;
			d->table_lds = lds_bytes_btab() + ((a.tab_bytes + 15u) & ~15u);
			break;
		}
		case FSM_HIP_LAYOUT_GLOBAL: {
			if (!p.glob_tab16.empty()) {
				/* <= 65 535 states: 2-byte entries = the next state's index (Glob16Pol) */
				uint16_t *t16 = nullptr;
				HIP_TRY(upload(&t16, p.glob_tab16));
				d->d_tab = t16;
				HIP_TRY(upload(&d->d_fin, p.glob16_fin));      /* rows in visit-frequency order (plan.cpp): fin follows */
				for (int b = 0; b < 256; b++) btab[b] = p.cls[b];
				a.start = p.glob16_rank.empty() ? p.start : p.glob16_rank[p.start];
				a.abs_min = p.abs_min;
				a.fin_div = 1;
				a.dflt = p.C * 2u;          /* bytes per row */
				d->glob16 = true;
				d->glob_row_bytes = p.C * 2u;
				d->glob_tab_bytes = (uint64_t)p.glob_tab16.size() * 2u;
				set_hot_bytes(d, 160u * 1024u);
				break;
			}
			uint32_t *t = nullptr;
			HIP_TRY(upload(&t, p.glob_tab));
			d->d_tab = t;
			HIP_TRY(upload(&d->d_fin, p.fin));
			for (int b = 0; b < 256; b++) btab[b] = p.cls[b];
			a.start = p.start * p.C * 4u;
			a.abs_min = p.abs_min * p.C * 4u;
			a.fin_div = p.C * 4u;
			d->glob_row_bytes = p.C * 4u;
			d->glob_tab_bytes = (uint64_t)p.glob_tab.size() * 4u;
			set_hot_bytes(d, 120u * 1024u);
			break;
		}
		case FSM_HIP_LAYOUT_SPARSE: {
			uint32_t *t = nullptr;
			HIP_TRY(upload(&t, p.sparse_img));
			d->d_tab = t;
			HIP_TRY(upload(&d->d_fin, p.fin));
			a.tab_bytes = p.sparse_lds_bytes;
			a.start = p.start;
			a.abs_min = p.abs_min;
			a.fin_div = 1;
			d->table_lds = SparsePol::lds_bytes(a.tab_bytes);
			{
				/* SparseFastPol::enter builds a record's address from a 32-bit low half: the record array must not cross
				 * a 4 GiB boundary (hipMalloc hands out 2 MiB-aligned blocks, the array is a few MB: practically never) */
				const uint64_t g = reinterpret_cast<uint64_t>(t) + p.sparse_img[5], bytes = (uint64_t)p.S1 * 16u;
				if ((g >> 32) != ((g + bytes) >> 32)) d->sparse_fast_ok = false;
			}
			if (p.lazy_lds_bytes != 0 && ((p.lazy_lds_bytes + 15u) & ~15u) + FSMHIP_LAZY_QBYTES <= d->lds_limit) {
				HIP_TRY(upload(&d->d_lazy, p.lazy_img));
				HIP_TRY(hipMalloc((void **)&d->d_lazy_ctr, LAZY_CTRS * sizeof(uint32_t)));
				a.lazy = d->d_lazy;
				/* the default where it pays: few states beyond the LDS set whose own record sends hits the exact way
				 * (img[10]) or that carry nothing (img[9]) -- on the 1e5-literal automaton 4.5 % of them, deep in the trie */
				const uint64_t deep = p.abs_min > p.lazy_img[1] ? p.abs_min - p.lazy_img[1] : 0;
				if (((uint64_t)p.lazy_img[9] + p.lazy_img[10]) * 10u <= deep) d->knob_sparse_fast = 3;
			}
			break;
		}
		default:
			errno = EINVAL;
			goto fail;
		}
		HIP_TRY(upload(&d->d_btab, btab));
		d->enc_host.resize(p.S1);
		for (uint32_t n2 = 0; n2 < p.S1; n2++) {
			switch (p.layout) {
			case FSM_HIP_LAYOUT_TINY: d->enc_host[n2] = p.tiny5_col.empty() ? n2 : n2 * 5u; break;
			case FSM_HIP_LAYOUT_SPARSE: d->enc_host[n2] = n2; break;
			case FSM_HIP_LAYOUT_LDS:
			case FSM_HIP_LAYOUT_LDSSELF: d->enc_host[n2] = n2 * p.row_bytes; break;
			case FSM_HIP_LAYOUT_LDS2: d->enc_host[n2] = n2 * p.lds2_c1 * p.lds2_c1; break;
			case FSM_HIP_LAYOUT_COMB:
			case FSM_HIP_LAYOUT_COMBSELF: d->enc_host[n2] = p.comb_off[n2]; break;
			case FSM_HIP_LAYOUT_COMB256: d->enc_host[n2] = p.comb256_off[n2]; break;
			default: d->enc_host[n2] = !p.glob_tab16.empty() ? (p.glob16_rank.empty() ? n2 : p.glob16_rank[n2]) : n2 * p.C * 4u; break;   /* global: the row's byte offset, or (2-byte entries) its row */
			}
		}
		d->fin_host = (p.layout == FSM_HIP_LAYOUT_COMB || p.layout == FSM_HIP_LAYOUT_COMBSELF) ? p.comb_fin
			: p.layout == FSM_HIP_LAYOUT_COMB256 ? p.comb256_fin : (p.layout == FSM_HIP_LAYOUT_GLOBAL && !p.glob_tab16.empty()) ? p.glob16_fin : p.fin;
		if (!p.emask.empty()) {
			/* eager masks indexed like fin; thresholds in encoded-state units */
			const size_t F = d->fin_host.size();
			std::vector<uint64_t> em(F, 0);
			std::vector<uint32_t> state_of(F, 0xFFFFFFFFu);   /* fin index -> renumbered state */
			for (uint32_t n2 = 0; n2 < p.S1; n2++) {
				em[d->enc_host[n2] / a.fin_div] = p.emask[n2];
				state_of[d->enc_host[n2] / a.fin_div] = n2;
			}
			HIP_TRY(upload(&d->d_emask, em));
			a.emask = d->d_emask;
			switch (p.layout) {
			case FSM_HIP_LAYOUT_COMB:
			case FSM_HIP_LAYOUT_COMBSELF:   /* rows placed region by region: thresholds on row offsets (plan.cpp build_comb) */
				a.eager_lo_end = p.comb_eager_lo_off;
				a.eager_hi_begin = p.comb_eager_hi_off;
				break;
			case FSM_HIP_LAYOUT_COMB256:
				a.eager_lo_end = p.comb256_eager_lo_off;
				a.eager_hi_begin = p.comb256_eager_hi_off;
				break;
			default:
				a.eager_lo_end = p.eager_lo_end < p.S1 ? d->enc_host[p.eager_lo_end] : 0xFFFFFFFFu;
				a.eager_hi_begin = p.eager_hi_begin < p.S1 ? d->enc_host[p.eager_hi_begin] : 0xFFFFFFFFu;
				break;
			}
			a.eager_words = p.eager_words;
			if (p.eager_words > 1) {
				/* wide sets: the (word, mask) runs re-laid in fin-index order (= renumbered state except in
				 * the comb layouts, whose fin index is the row offset) */
				std::vector<uint32_t> off(F + 1, 0), w;
				std::vector<uint64_t> m;
				for (size_t i = 0; i < F; i++) {
					off[i] = (uint32_t)w.size();
					const uint32_t n2 = state_of[i];
					if (n2 == 0xFFFFFFFFu) continue;
					w.insert(w.end(), p.ew_word.begin() + p.ew_off[n2], p.ew_word.begin() + p.ew_off[n2 + 1]);
					m.insert(m.end(), p.ew_mask.begin() + p.ew_off[n2], p.ew_mask.begin() + p.ew_off[n2 + 1]);
				}
				off[F] = (uint32_t)w.size();
				if (w.empty()) { w.push_back(0); m.push_back(0); }
				HIP_TRY(upload(&d->d_ew_off, off));
				HIP_TRY(upload(&d->d_ew_word, w));
				HIP_TRY(upload(&d->d_ew_mask, m));
				a.ew_off = d->d_ew_off;
				a.ew_word = d->d_ew_word;
				a.ew_mask = d->d_ew_mask;
			}
		}
		/* fin_index(): code / fin_div as a multiply where that is exact for every code the walk can produce */
		a.fin_mul = 0;
		if (a.fin_div > 1u) {
			uint32_t maxcode = 0;
			for (uint32_t n2 = 0; n2 < p.S1; n2++) maxcode = d->enc_host[n2] > maxcode ? d->enc_host[n2] : maxcode;
			if ((uint64_t)maxcode * a.fin_div < ((uint64_t)1 << 32)) a.fin_mul = (uint32_t)((((uint64_t)1 << 32) / a.fin_div) + 1u);
		}
		/* the spare class of the self-loop-mask layouts (plan.cpp sets bit 31 in every mask when there are <= 31 classes) */
		a.ident_class = ((p.layout == FSM_HIP_LAYOUT_COMBSELF || p.layout == FSM_HIP_LAYOUT_LDSSELF) && p.C <= 31u) ? 31u : 0xFFFFFFFFu;
		a.tab = d->d_tab;
		a.fin = d->d_fin;
		a.btab = d->d_btab;
		a.early = (flags & FSM_HIP_NO_EARLY_RETIRE) ? 0u : 1u;
	}
	HIP_TRY(hipMalloc((void **)&d->d_pick, PICK_FLAGS * sizeof(uint32_t)));
	HIP_TRY(hipEventCreateWithFlags(&d->tb_scratch_ev, hipEventDisableTiming));
	HIP_TRY(hipEventCreate(&d->ev0));
	HIP_TRY(hipEventCreate(&d->ev1));
	HIP_TRY(hipStreamCreateWithFlags(&d->hs, hipStreamNonBlocking));
	d->uploaded.store(true, std::memory_order_release);
	return 0;
fail:
	d->upload_errno = errno != 0 ? errno : EIO;
	return -1;
}

static int ensure_uploaded(const fsm_hip_dfa *cd)
{
	if (cd->uploaded.load(std::memory_order_acquire)) return 0;
	fsm_hip_dfa *d = const_cast<fsm_hip_dfa *>(cd);
	DfaLock lk(d->mu);
	if (d->uploaded.load(std::memory_order_relaxed)) return 0;
	if (d->upload_errno != 0) { errno = d->upload_errno; return -1; }
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	return dfa_upload(d);
}

extern "C" struct fsm_hip_dfa *fsm_hip_dfa_create(const struct fsm_hip_dfa_desc *desc, unsigned flags)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		errno = ENODEV; /* no CPU fallback by design */
		return nullptr;
	}
	fsm_hip_dfa *d = new (std::nothrow) fsm_hip_dfa();
	if (d == nullptr) { errno = ENOMEM; return nullptr; }
	d->flags = flags;
	{
		int v = 0;
		HIP_TRY(hipGetDevice(&d->device));
		if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d->device) == hipSuccess && v > 0) d->ncu = v;
		if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, d->device) == hipSuccess && v > 0) d->lds_limit = (uint32_t)v;
		if (d->lds_limit > 160u * 1024u) d->lds_limit = 160u * 1024u;
	}
	{
		int r = build_plan(desc, flags, d->lds_limit, d->plan);
		if (r != 0) { errno = r; goto fail; }
	}
	if (!(flags & FSM_HIP_DEFER_UPLOAD) && dfa_upload(d) != 0) goto fail;
	if (d->plan.layout == FSM_HIP_LAYOUT_LDS2 && (flags & FSM_HIP_LAYOUT_MASK) == FSM_HIP_LAYOUT_AUTO && !(flags & FSM_HIP_PLAN_NO_LDS2)) {
		/* fewer than eight ragged wavefronts fit beside the pair table: the variable-length fronts get a table of their own */
		const uint32_t pair_lds = Lds2Pol::lds_bytes((uint32_t)(d->plan.lds_tab.size() * 2));
		if (pair_lds + 8u * FSMHIP_RAGGED_WAVE_LDS > d->lds_limit) {
			d->alt = fsm_hip_dfa_create(desc, flags | FSM_HIP_PLAN_NO_LDS2);
			if (d->alt == nullptr) goto fail;
		}
	}
	return d;
fail:
	{
		int e = errno;
		fsm_hip_dfa_free(d);
		errno = e;
	}
	return nullptr;
}

extern "C" void fsm_hip_dfa_free(struct fsm_hip_dfa *d)
{
	if (d == nullptr) return;
	if (d->alt) fsm_hip_dfa_free(d->alt);
	if (d->d_tab) (void)hipFree(d->d_tab);
	if (d->d_fin) (void)hipFree(d->d_fin);
	if (d->d_btab) (void)hipFree(d->d_btab);
	if (d->d_lazy) (void)hipFree(d->d_lazy);
	if (d->d_lazy_ctr) (void)hipFree(d->d_lazy_ctr);
	if (d->d_fin_earliest) (void)hipFree(d->d_fin_earliest);
	if (d->d_fin_ret) (void)hipFree(d->d_fin_ret);
	if (d->d_enc_of) (void)hipFree(d->d_enc_of);
	if (d->d_orig_of) (void)hipFree(d->d_orig_of);
	if (d->d_emask) (void)hipFree(d->d_emask);
	if (d->d_ew_off) (void)hipFree(d->d_ew_off);
	if (d->d_ew_word) (void)hipFree(d->d_ew_word);
	if (d->d_ew_mask) (void)hipFree(d->d_ew_mask);
	for (uint32_t *q : { d->d_tr_dense, d->d_tr_cls4, d->d_tr_eoff, d->d_tr_eids, d->d_tr_fin }) if (q) (void)hipFree(q);
	if (d->d_pick) (void)hipFree(d->d_pick);
	if (d->tb_scratch) (void)hipFree(d->tb_scratch);
	for (void *q : d->tb_scratch_old) (void)hipFree(q);
	if (d->tb_scratch_ev) (void)hipEventDestroy(d->tb_scratch_ev);
	if (d->arena) (void)hipFree(d->arena);
	if (d->stage) (void)hipHostFree(d->stage);
	if (d->hs) (void)hipStreamDestroy(d->hs);
	if (d->ev0) (void)hipEventDestroy(d->ev0);
	if (d->ev1) (void)hipEventDestroy(d->ev1);
	delete d;
}

/* ------------------------------------------------------------------ */
/* launch                                                             */
/* ------------------------------------------------------------------ */

/* Where the per-lane kernels hand over to walk_ragged (mean input length, bytes).  walk_generic: 96 (round 3).  walk_lines32 is
 * faster than that kernel and still wins at a mean of 96 (64-128 byte lines: 2.40 vs 1.86 TB/s on the column table, 2.35 vs 1.40
 * on the C3 table), walk_ragged from ~160 on (64-256 bytes: 1.81 vs 2.69, 1.86 vs 2.09): 128 (profiles/r08r_*). */
static int pick_mean_of(const fsm_hip_dfa *d, bool lines32)
{
	return d->knob_pick_mean >= 0 ? d->knob_pick_mean : lines32 ? 128 : 96;
}

/* per_lane: take walk_generic unless a knob says otherwise -- the inputs average < 96 bytes (walk_ragged works in
 * 128-byte segments: 0.7-1.2 vs 1.5-2.1 TB/s at 8-64 bytes, profiles/r03t_*); huge: the batch may hold an input the
 * ragged kernel's 32-bit piece count cannot (>= 2^36 bytes) */
static LaunchCfg pick_cfg(const fsm_hip_dfa *d, bool fast_ok, uint64_t stride, int eager, bool per_lane = false, bool huge = false, bool resumed = false, bool many = false)
{
	LaunchCfg c = LaunchCfg();    /* kfn = nullptr until a launch sets it */
	const uint32_t layout = d->plan.layout;
	c.nb = 8;
	c.seg = 128;
	/* every input line is consumed by exactly one DMA instruction (SEG = 128): nontemporal loads
	 * measured +7.5 % on the HBM-bound tiny layout (profiles/r01_sweep8*), neutral elsewhere */
	c.nt = d->knob_nt >= 0 ? (d->knob_nt != 0) : 1;
	/* CombSelfPol's branchy chain is latency-bound: drop the register double-buffer (<= 64 VGPRs)
	 * so two 16-wave workgroups share a CU (profiles/r01_sweep5*: 4.52 vs 4.32 TB/s) */
	c.prefetch = d->knob_prefetch >= 0 ? (d->knob_prefetch != 0) : (layout == FSM_HIP_LAYOUT_COMBSELF || layout == FSM_HIP_LAYOUT_SPARSE ? 0 : 1);
	/* ragged / packed / unaligned inputs: the coalesced, lane-refilling kernel whenever at least four
	 * waves' tiles and rings fit next to the table, else per-lane loads (walk_generic) */
	/* the 5-bit column table keeps the ragged kernel's ring and row records in its own holes (walk_kernels.h ragged_aux_in_holes) */
	const bool ragged_holes = layout == FSM_HIP_LAYOUT_TINY && !d->plan.tiny5_col.empty() && !eager;
	const uint32_t ragged_wave_lds = ragged_holes ? 8192u : FSMHIP_RAGGED_WAVE_LDS;
	const bool ragged_fits = d->table_lds + 4u * ragged_wave_lds <= d->lds_limit;
	int mode = ragged_fits && !per_lane && !huge ? IN_RAGGED : IN_GENERIC;
	if (d->knob_input_mode == IN_GENERIC || (d->knob_input_mode == IN_RAGGED && ragged_fits && !huge)) mode = d->knob_input_mode;
	else if (fast_ok) {
		/* measured (profiles/r01_sweep*.txt): LDS-DMA staging (8 KiB tile per wave) is the better input
		 * path whenever at least 12 waves of tiles fit next to the table (lds layout 5.1 vs 4.8 TB/s;
		 * combself, once whole self-loop chunks are skipped, 6.1 vs 5.0 TB/s: profiles/r01_sweep11*);
		 * per-lane loads with 8 chunks in flight next to a bigger LDS table */
		const bool dma_fits = d->table_lds + 12u * 8192u <= d->lds_limit;
		int m = (layout == FSM_HIP_LAYOUT_TINY || dma_fits) ? IN_LDSDMA : IN_DIRECT;
		if (d->knob_input_mode == IN_DIRECT || d->knob_input_mode == IN_LDSDMA) m = d->knob_input_mode;
		if (m == IN_LDSDMA && stride % 64u != 0) m = IN_DIRECT;
		if (m == IN_LDSDMA) {
			c.seg = d->knob_seg == 64 ? 64 : 128;
			if (stride % 128u != 0) c.seg = 64;
			/* the eager kernels exist for 128-byte segments and register-held sets only (2.3 vs 1.1 TB/s for
			 * those; wide sets measured 1.8 behind LDS-DMA vs 2.1 with per-lane loads in round 2 and 2.53 vs 2.71 in round 3:
			 * profiles/r04x_eager_wide_dma.txt) */
			if (eager && (c.seg != 128 || eager == 2)) m = IN_DIRECT;
		}
		if (m == IN_DIRECT) {
			/* the sparse layout waits on gathers, not on its input: 4 chunks keep it under 64 VGPRs;
			 * the eager kernel is instantiated for 4 chunks only */
			c.nb = d->knob_nb == 4 || d->knob_nb == 8 ? d->knob_nb : (layout == FSM_HIP_LAYOUT_SPARSE || d->glob16 ? 4 : 8);   /* (the 2-byte global table: 4 chunks x TWO inputs per lane, kern_glob16.hip) */
			if (eager || !c.prefetch) c.nb = 4;
			if ((stride / 16u) % 8u != 0) c.nb = 4;
			if ((stride / 16u) % 4u != 0) m = -1;   /* rows shorter than / not a multiple of 64 bytes */
		}
		if (m >= 0) mode = m;
	}
	c.sparse_fast = d->sparse_fast_ok ? d->knob_sparse_fast : (d->knob_sparse_fast == 3 ? 3 : 0);
	c.lazy_abs = 0;
	if (mode == IN_DIRECT && layout == FSM_HIP_LAYOUT_SPARSE && !eager && !resumed && d->d_lazy != nullptr && d->knob_sparse_fast == 3 &&
	    (stride / 16u) % 4u == 0) {
		/* the lazy walk: one 16-wave workgroup per CU beside its 131 KiB of tables, three inputs per lane (walk_lazy.h) */
		c.mode = IN_LAZY;
		c.nt = d->knob_nt > 0;      /* nontemporal input loads: A/B knob (FSM_HIP_KNOB_NT) */
		c.lazy_abs = d->plan.lazy_img[11] != 0;
		/* inputs per lane x chunks in flight: 3 x 4 (kern_glob.hip: 1 036 GB/s on the 1e5-literal automaton; 2 x 4: 1 000, 3 x 2: 970,
		 * 4 x 2: 758 -- profiles/r09j_*); FSM_HIP_KNOB_ROWS = 2: round 5's shape (A/B) */
		c.lazy_rows = d->knob_rows == 2 ? 2 : 3;
		c.nb = 4;
		c.waves = 16;
		c.lds = d->plan.lazy_lds_bytes;
		c.blocks_per_cu = d->knob_blocks_per_cu > 0 ? d->knob_blocks_per_cu : 1;
		return c;
	}
	if (layout == FSM_HIP_LAYOUT_SPARSE && !eager && d->d_lazy != nullptr && d->knob_sparse_fast == 3 && d->knob_input_mode < 0 && d->knob_lazy_lines && !many) {   /* (many: 2^32 inputs or more -- its queue entries hold 32-bit indexes) */
		/* ... and on everything else such an automaton is asked (round 5): inputs of any length and metadata form, strides that
		 * are not a multiple of 64, unaligned rows, resumed walks -- one input per lane slot with lane refill (walk_lazy_lines) */
		c.mode = IN_LAZY_LINES;
		c.lazy_abs = d->plan.lazy_img[11] != 0 || resumed;   /* a resumed input may start in DEAD */
		/* THREE slots per lane, three whole chunks per slot and turn (kern_glob.hip has the numbers).  FSM_HIP_KNOB_ROWS = 2 brings the
		 * two-slot forms back (FSM_HIP_KNOB_NB = 2 / 4 chunks), FSM_HIP_KNOB_NB = 2 / 4 at three slots the two- and four-chunk turns: A/B */
		c.lazy_rows = d->knob_rows == 2 ? 2 : 3;
		c.nb = c.lazy_rows == 3 ? (d->knob_nb == 2 || d->knob_nb == 4 ? d->knob_nb : 3) : d->knob_nb == 2 ? 2 : 4;
		c.waves = 16;
		/* the table + the wavefronts' queues: all the LDS there is (plan.cpp leaves at least FSMHIP_LAZY_QBYTES: 112 entries per
		 * wavefront; the kernel uses up to 128) */
		c.lds = ((d->plan.lazy_lds_bytes + 15u) & ~15u) + 16u * 128u * 16u;
		if (c.lds > d->lds_limit) c.lds = d->lds_limit;
		c.blocks_per_cu = d->knob_blocks_per_cu > 0 ? d->knob_blocks_per_cu : 1;
		return c;
	}
	if (c.sparse_fast == 3) c.sparse_fast = d->sparse_fast_ok ? 1 : 0;
	/* a staging path asked for by knob beside a table that leaves no LDS for one wavefront's tile (the 2-byte global table takes
	 * all of it): the per-lane kernel of the same class of input */
	if ((mode == IN_LDSDMA && d->table_lds + 64u * (uint32_t)c.seg > d->lds_limit) || (mode == IN_RAGGED && d->table_lds + ragged_wave_lds > d->lds_limit)) {
		mode = mode == IN_LDSDMA ? IN_DIRECT : IN_GENERIC;
		if (mode == IN_DIRECT) {
			c.nb = d->knob_nb == 8 && (stride / 16u) % 8u == 0 ? 8 : 4;
			if ((stride / 16u) % 4u != 0) mode = IN_GENERIC;
		}
	}
	c.mode = mode;
	const uint32_t per_wave = mode == IN_LDSDMA ? 64u * (uint32_t)c.seg : mode == IN_RAGGED ? ragged_wave_lds : 0u;
	/* waves per block: as many behind one table copy as LDS holds, 16 at most: the tiny layouts keep a
	 * 64 KiB column table (one private copy per lane / bank), which leaves 12 x 8 KiB tiles of the 160 KiB.
	 * combself behind LDS-DMA: 12 waves measured best at 10^8 x 1 KiB (6.09 TB/s; 14: 5.82, 10: 5.80, 8: 5.72).
	 * Kernels compiled for fewer threads (their register budget): ragged 12 waves, eager LDS-DMA on the 64-bit column
	 * table 12 (launch.h eager_dma_threads), eager ragged / generic 8. */
	int wmax = 16;
	if (mode == IN_RAGGED) wmax = eager ? 8 : ragged_holes ? (int)FSMHIP_RAGGED_HOLE_WAVES : 12;
	else if (eager && mode == IN_GENERIC) wmax = 8;
	else if (eager && mode == IN_LDSDMA) wmax = ((layout == FSM_HIP_LAYOUT_TINY && d->plan.tiny5_col.empty()) || layout == FSM_HIP_LAYOUT_COMB ||
		                                            layout == FSM_HIP_LAYOUT_COMBSELF || layout == FSM_HIP_LAYOUT_LDSSELF) ? 12 : 16;   /* launch.h eager_dma_threads */
	else if (layout == FSM_HIP_LAYOUT_COMBSELF && mode == IN_LDSDMA) wmax = 12;
	/* the plain row table behind LDS-DMA: its walk keeps the LDS array ~85 % busy (two reads per byte, the table read
	 * replayed for bank conflicts: profiles/r04f_pmc_eager_lds.txt); 14 waves measured 4.24 / 4.01 TB/s where 16 gave
	 * 3.77 / 3.71 and 12 4.01 / 3.99 (354- and 1132-state tables, 8e6 x 1 KiB: profiles/r04h_eager_probe_8M.txt).  The
	 * eager form of the same walk wants 16 (3.53 vs 3.33 at 14, 3.12 at 12). */
	else if (layout == FSM_HIP_LAYOUT_LDS && mode == IN_LDSDMA && !eager) wmax = 14;
	/* short inputs on the column tables: 12-wave workgroups (two per CU) measured 2.29 / 1.61 / 2.52 TB/s at 8-64 / 8-16 /
	 * 32-128 bytes where 16 gave 2.05 / 1.32 / 2.26 (6e6 packed inputs, profiles/r04t_generic_waves.txt); the other
	 * layouts keep 16 (combself: 1.47 vs 1.25) */
	else if (layout == FSM_HIP_LAYOUT_TINY && mode == IN_GENERIC && !eager) wmax = 12;
	int waves = d->knob_waves > 0 && d->knob_waves < wmax ? d->knob_waves : wmax;
	if (!eager && mode != IN_RAGGED && d->knob_waves > wmax && d->knob_waves <= 16 &&
	    !(layout == FSM_HIP_LAYOUT_COMBSELF && mode == IN_LDSDMA)) waves = d->knob_waves;   /* that kernel is compiled for 12 */
	while (waves > 1 && d->table_lds + (uint32_t)waves * per_wave > d->lds_limit) waves -= (waves > 8 && mode != IN_RAGGED ? 2 : 1);
	c.waves = waves;
	c.lds = d->table_lds + (uint32_t)waves * per_wave;
	int bpc = (int)(d->lds_limit / (c.lds ? c.lds : 1u));
	if (bpc * waves > 32) bpc = 32 / waves;
	if (bpc < 1) bpc = 1;
	/* twice the resident workgroups: the tail of the persistent grid balances better (the ragged kernel
	 * partitions statically: one range per resident wave) */
	if (mode == IN_DIRECT || mode == IN_LDSDMA) bpc *= 2;
	if (d->knob_blocks_per_cu > 0) bpc = d->knob_blocks_per_cu;
	c.blocks_per_cu = bpc;
	return c;
}

/* what a front knows about its batch beyond the arguments (host fronts: everything; device fronts: nothing) */
struct BatchHint {
	uint64_t bytes = 0;        /* total input bytes, 0 = unknown */
	bool short_mean = false;   /* the inputs average < 96 bytes */
};

static hipError_t launch_layout(const fsm_hip_dfa *d, int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	const Plan &p = d->plan;
	switch (p.layout) {
	case FSM_HIP_LAYOUT_TINY:     return launch_tiny(p.tiny5_col.empty() ? POL_TINY64 : POL_TINY5, eager, c, a, grid, block, s);
	case FSM_HIP_LAYOUT_LDS:      return launch_lds(POL_LDS, eager, c, a, grid, block, s);
	case FSM_HIP_LAYOUT_LDSSELF:  return launch_lds(POL_LDSSELF, eager, c, a, grid, block, s);
	case FSM_HIP_LAYOUT_LDS2:     return launch_lds(POL_LDS2, eager, c, a, grid, block, s);
	case FSM_HIP_LAYOUT_COMB:     return launch_comb(POL_COMB, eager, c, a, grid, block, s);
	case FSM_HIP_LAYOUT_COMB256:  return launch_comb(POL_COMB256, eager, c, a, grid, block, s);
	case FSM_HIP_LAYOUT_COMBSELF: return launch_comb(POL_COMBSELF, eager, c, a, grid, block, s);
	case FSM_HIP_LAYOUT_SPARSE:   return launch_glob(POL_SPARSE, eager, c, a, grid, block, s);
	default:                      return d->glob16 ? launch_glob16(eager, c, a, grid, block, s) : launch_glob(POL_GLOB, eager, c, a, grid, block, s);
	}
}

/* FSM_HIP_DEBUG=2: synchronise after every stage of a launch and say which one it was (a GPU fault aborts the process) */
/* the launched kernel's own name, as the profiler shows it */
static std::string kernel_name(const void *kfn, hipStream_t s)
{
	if (kfn == nullptr) return "?";
	/* resolved once per kernel: the lookup + demangling is ~10 us, a small batch's walk is 20 */
	static std::mutex cache_mu;
	static std::vector<std::pair<const void *, std::string>> cache;
	{
		std::lock_guard<std::mutex> lk(cache_mu);
		for (const auto &e : cache) if (e.first == kfn) return e.second;
	}
	const char *m = hipKernelNameRefByPtr(kfn, s);
	if (m == nullptr) return "?";
	int st = 0;
	char *dm = abi::__cxa_demangle(m, nullptr, nullptr, &st);
	std::string r = st == 0 && dm != nullptr ? dm : m;
	free(dm);
	const size_t p = r.find("(fsmhip::WalkArgs");
	if (p != std::string::npos) r.resize(p);
	if (r.compare(0, 5, "void ") == 0) r.erase(0, 5);
	std::lock_guard<std::mutex> lk(cache_mu);
	cache.emplace_back(kfn, r);
	return r;
}

/* vector registers of a kernel (cached per kernel: the query is a driver call) */
static int kernel_vgprs(const void *kfn)
{
	if (kfn == nullptr) return 0;
	static std::mutex cache_mu;
	static std::vector<std::pair<const void *, int>> cache;
	{
		std::lock_guard<std::mutex> lk(cache_mu);
		for (const auto &e : cache) if (e.first == kfn) return e.second;
	}
	hipFuncAttributes at;
	if (hipFuncGetAttributes(&at, kfn) != hipSuccess) {   /* (not remembered: asked again at the next launch) */
		(void)hipGetLastError();
		return 0;
	}
	std::lock_guard<std::mutex> lk(cache_mu);
	cache.emplace_back(kfn, at.numRegs);
	return at.numRegs;
}

/* The per-lane kernels on an LDS table are bound by latency (a tile's chunks are asked for and waited for at once): what
 * counts is how many wavefronts a CU holds.  The table's LDS allows `by_lds` workgroups per CU; a SIMD holds 512 / registers
 * wavefronts (8 at most), a workgroup of W wavefronts puts ceil(W / 4) on each.  walk_lines32<CombSelfPol> (76 registers: 6 per
 * SIMD): one 16-wavefront workgroup fits (4 per SIMD), or TWO of 12 (6 per SIMD) -- 24000000 lines of 8-64 bytes on the C3 table:
 * 0.485 -> 0.418 ms, 8-16 bytes 0.294 -> 0.234 (profiles/r08d_lines32_waves.txt).  Returns the workgroup size (wavefronts)
 * that keeps most wavefronts resident, the larger one on a tie. */
static int waves_by_occupancy(int vgprs, int by_lds, int wmax)
{
	if (vgprs <= 0) return wmax;
	const int alloc = (vgprs + 7) / 8 * 8;
	int per_simd = 512 / alloc;
	if (per_simd > 8) per_simd = 8;
	int best = wmax, best_res = 0;
	for (int w = wmax; w >= 8; w -= 4) {
		const int each = (w + 3) / 4;
		int blocks = per_simd / each;
		if (blocks > by_lds) blocks = by_lds;
		if (blocks * w > 32) blocks = 32 / w;
		if (blocks * w > best_res) { best_res = blocks * w; best = w; }
	}
	return best;
}

extern "C" int fsm_hip_waves_by_occupancy(int vgprs, int workgroups_by_lds, int max_waves)
{
	if (max_waves < 8) return max_waves;
	return waves_by_occupancy(vgprs, workgroups_by_lds < 1 ? 1 : workgroups_by_lds, max_waves > 16 ? 16 : max_waves);
}

static void debug_stage(hipStream_t s, const char *what)
{
	static const int lvl = getenv("FSM_HIP_DEBUG") ? atoi(getenv("FSM_HIP_DEBUG")) : 0;
	if (lvl < 2) return;
	fprintf(stderr, "fsm_hip: %s ...", what);
	fflush(stderr);
	const hipError_t e = hipStreamSynchronize(s);
	fprintf(stderr, " %s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
	fflush(stderr);
}

static int launch_walk(const fsm_hip_dfa *d, WalkArgs a, bool fast_ok, hipStream_t s, const BatchHint &hint = BatchHint())
{
	if (a.n == 0) return 0;
	const int eager = a.eager_out == nullptr ? 0 : a.eager_words > 1 ? 2 : 1;
	const bool varlen = a.off != nullptr || a.off32 != nullptr || a.len != nullptr;
	const uint64_t known_bytes = hint.bytes != 0 ? hint.bytes : !varlen ? (uint64_t)a.n * a.stride : 0;
	const LaunchCfg c = pick_cfg(d, fast_ok, a.stride, eager, hint.short_mean, known_bytes >= ((uint64_t)1 << 36), a.state_io != nullptr, a.n >= ((uint64_t)1 << 32));
	const uint64_t ntiles = (a.n + 63u) / 64u;
	const uint64_t lazy_tile = 64u * (uint64_t)(c.lazy_rows > 0 ? c.lazy_rows : 2);      /* inputs per wavefront of the fixed-stride lazy walk */
	uint64_t nblocks = c.mode == IN_LAZY ? ((a.n + lazy_tile - 1u) / lazy_tile + c.waves - 1) / c.waves
		: c.mode == IN_LAZY_LINES ? ((a.n + FSMHIP_LAZY_PIECE - 1u) / FSMHIP_LAZY_PIECE + c.waves - 1) / c.waves : (ntiles + c.waves - 1) / c.waves;
	const uint64_t cap = (uint64_t)d->ncu * c.blocks_per_cu;
	if (nblocks > cap) nblocks = cap;
	if (d->knob_early >= 0) a.early = (uint32_t)d->knob_early; /* bit 0 wave retire, bit 1 per-lane load skip */
	if (d->knob_noskip > 0) a.early |= 4u;

	/* Variable-length inputs (the retest / rx front).  Short ones (mean < 96 bytes) walk fastest one per lane with per-lane
	 * loads (walk_generic; a plain walk of a packed batch below 4 GiB: its 32-bit form walk_lines32), long ones in 128-byte
	 * segments with lane refill (walk_ragged).  A host-pointer front knows the mean and the size; a device-pointer front cannot
	 * know them without a synchronising copy, so the candidates are ALL launched and a small kernel decides on the device
	 * which of them runs (offsets_pick, walk_aux.h); the others return at once. */
	const bool pick_len = varlen && known_bytes == 0 && !hint.short_mean && d->knob_input_mode < 0 && c.mode == IN_RAGGED;
	/* the 32-bit lines kernel: plain outputs, a packed front, < 2^29 inputs; the batch's size: known (1 / 0) or not (-1) */
	const bool lines_cand = !eager && a.out2 == nullptr && a.state_io == nullptr && (a.off != nullptr || a.off32 != nullptr || a.tbase != nullptr) &&
		a.n < 0x1FFFFFF0ull && !(a.early & 32u) && (c.mode == IN_GENERIC || pick_len);
	int fits32 = !lines_cand ? 0 : a.off32 != nullptr ? 1 : known_bytes != 0 ? (known_bytes < ((uint64_t)1 << 32) ? 1 : 0) : -1;
	bool skip_generic = false;
	if (fits32 < 0 && pick_len) {
		/* a device front with u64 offsets / lengths alone: the batch's size is on the device -- but it should not end beyond the
		 * ALLOCATION its base points into.  Fewer than 4 GiB from the base to the allocation's end (every batch but a huge one in
		 * a huge block): walk_generic's own body is not needed and is not launched.  The reported range is a HINT, not a proof
		 * (memory mapped in pieces may report one piece): fits32 stays "unknown", offsets_pick still looks at the batch's last
		 * offset, and is told that walk_generic was not launched (cand bit 2 clear) -- a batch of 4 GiB and more then goes to
		 * walk_ragged, which is launched and takes any batch; where walk_ragged is no candidate walk_generic is always launched.
		 * (Not asked during a stream capture: the query is not a stream operation, and an API that is not may end a capture.) */
		hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
		if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
			hipDeviceptr_t ab = nullptr;
			size_t asz = 0;
			if (hipMemGetAddressRange(&ab, &asz, (hipDeviceptr_t)const_cast<uint8_t *>(a.base)) == hipSuccess && ab != nullptr) {
				const uint64_t room = reinterpret_cast<uint64_t>(ab) + asz - reinterpret_cast<uint64_t>(a.base);
				if (room < ((uint64_t)1 << 32)) skip_generic = true;
			} else (void)hipGetLastError();
		} else (void)hipGetLastError();
	}
	const bool pick = pick_len || fits32 < 0;

	fsm_hip_dfa *md = const_cast<fsm_hip_dfa *>(d);
	DfaLock lk(md->mu);   /* the timing events, the flag ring and the tile-base block are per dfa */
	hipError_t e = hipSuccess;
	/* the ragged kernel sets bitmap bits one input at a time */
	if ((c.mode == IN_RAGGED || c.mode == IN_LAZY_LINES) && a.bitmap != nullptr) e = zero_async(a.bitmap, ntiles * sizeof(uint64_t), s);
	if (e == hipSuccess && ((c.mode == IN_LAZY && d->knob_lazy_dyn) || c.mode == IN_LAZY_LINES) && d->d_lazy_ctr != nullptr) {
		a.tile_ctr = d->d_lazy_ctr + (md->lazy_ctr_next++ % LAZY_CTRS);
		e = zero_async(a.tile_ctr, sizeof(uint32_t), s);
	}
	if (e == hipSuccess) e = hipEventRecord(md->ev0, s);
	md->last_pick_flag = nullptr;
	for (auto &nm : md->last_kernel_pick) nm.clear();
	if (e == hipSuccess && pick) {
		a.pick_flag = md->d_pick + (md->pick_next++ % PICK_FLAGS);
		a.skip_flag = a.pick_flag;
		/* bit 0: walk_ragged is a candidate (else every batch counts as short); bit 1: so is walk_lines32 */
		hipLaunchKernelGGL(offsets_pick, dim3(1), dim3(256), 0, s, a, (uint32_t)pick_mean_of(d, false), (uint32_t)pick_mean_of(d, true),
		                   (pick_len ? 1u : 0u) | (fits32 != 0 ? 2u : 0u) | (fits32 != 1 && !skip_generic ? 4u : 0u));
		e = hipGetLastError();
	}
	/* the per-lane kernels (short inputs): walk_generic unless the batch is known to fit 32 bits, walk_lines32 unless known not to */
	const bool per_lane = c.mode == IN_GENERIC || pick_len;
	if (e == hipSuccess && per_lane) {
		const LaunchCfg g0 = c.mode == IN_GENERIC ? c : pick_cfg(d, false, a.stride, eager, true, false);
		for (int w32 = 0; w32 < 2 && e == hipSuccess; w32++) {
			if (w32 ? fits32 == 0 : (fits32 == 1 || skip_generic)) continue;
			WalkArgs ag = a;
			ag.run_when = w32 ? PICK_LINES32 : PICK_GENERIC;
			LaunchCfg g = g0;
			g.lines32 = w32;
			/* the workgroup size that keeps most wavefronts on a CU, from the kernel's own register count (LDS tables, plain walks,
			 * no knob) */
			if (!eager && d->knob_waves <= 0 && d->knob_blocks_per_cu <= 0 && d->table_lds != 0 && d->plan.layout != FSM_HIP_LAYOUT_TINY) {
				g.probe = 1;
				g.kfn = nullptr;
				if (launch_layout(d, eager, g, ag, dim3(1), dim3(64), s) == hipSuccess && g.kfn != nullptr) {
					const int by_lds = (int)(d->lds_limit / d->table_lds);
					g.waves = waves_by_occupancy(kernel_vgprs(g.kfn), by_lds < 1 ? 1 : by_lds, g0.waves);
					int bpc = by_lds < 1 ? 1 : by_lds;
					if (bpc * g.waves > 32) bpc = 32 / g.waves;
					g.blocks_per_cu = bpc < 1 ? 1 : bpc;
				}
				g.probe = 0;
			}
			const uint64_t gb0 = (ntiles + g.waves - 1) / g.waves, gcap = (uint64_t)d->ncu * g.blocks_per_cu;
			const dim3 ggrid((unsigned)(gb0 < gcap ? gb0 : gcap)), gblock((unsigned)g.waves * 64u);
			g.kfn = nullptr;
			e = launch_layout(d, eager, g, ag, ggrid, gblock, s);
			if (e == hipSuccess) {
				md->last_kernel = kernel_name(g.kfn, s);
				md->last_kernel_pick[ag.run_when] = md->last_kernel;
			}
		}
		debug_stage(s, "walk (per-lane)");
	}
	if (e == hipSuccess && c.mode != IN_GENERIC) {
		a.run_when = PICK_RAGGED;
		c.kfn = nullptr;
		e = launch_layout(d, eager, c, a, dim3((unsigned)nblocks), dim3((unsigned)c.waves * 64u), s);
		if (e == hipSuccess) {
			md->last_kernel = kernel_name(c.kfn, s);
			md->last_kernel_pick[PICK_RAGGED] = md->last_kernel;
		}
		debug_stage(s, "walk");
	}
	if (e == hipSuccess && pick) {   /* several were launched, one ran: fsm_hip_last_kernel_name() asks the flag which */
		md->last_pick_flag = a.pick_flag;
		std::string all;
		for (const auto &nm : md->last_kernel_pick) if (!nm.empty()) all += (all.empty() ? "" : " | ") + nm;
		md->last_kernel = all;
	}
	if (e == hipSuccess) e = hipEventRecord(md->ev1, s);
	if (e != hipSuccess) {
		if (getenv("FSM_HIP_DEBUG")) fprintf(stderr, "fsm_hip: launch -> %s\n", hipGetErrorString(e));
		errno = hip_errno(e);
		return -1;
	}
	md->timed = true;
	return 0;
}

/* The lengths-only front: byte offset of every 64th input of a batch packed back to back (walk_aux.h tile_bases_*), into the
 * dfa's grow-only block; the walk that follows on the same stream reads it.  The block's previous user (any stream) is
 * waited for first; `tbase` = the array the walk kernels take (T + 1 entries, the last one the batch's size). */
static int ensure_ids(fsm_hip_dfa *d);
static int ensure_resume(fsm_hip_dfa *d);

static size_t tb_bytes_for(size_t n)
{
	const uint64_t T1 = (n + 63u) / 64u + 1u, nb = (T1 + 1023u) / 1024u;
	return (size_t)(T1 + nb) * sizeof(uint64_t);
}

/* (grows by doubling: a handful of blocking hipMalloc calls over a dfa's life; launches in flight may still use the old block) */
static hipError_t tb_grow(fsm_hip_dfa *d, size_t want)
{
	if (want <= d->tb_scratch_bytes) return hipSuccess;
	if (d->tb_scratch) d->tb_scratch_old.push_back(d->tb_scratch);
	d->tb_scratch = nullptr;
	d->tb_scratch_bytes = 0;
	size_t cap = (size_t)1 << 16;
	while (cap < want) cap *= 2;
	const hipError_t e = hipMalloc((void **)&d->tb_scratch, cap);
	if (e == hipSuccess) d->tb_scratch_bytes = cap;
	return e;
}

/* Everything a later call on batches of up to n inputs would allocate, now: the tile-base block of the lengths-only front
 * (the layout's tables too, for a dfa created with FSM_HIP_DEFER_UPLOAD).  After it the device-pointer fronts make no
 * allocation for such batches and can be captured into a HIP graph from the first launch. */
extern "C" int fsm_hip_reserve(struct fsm_hip_dfa *d, size_t n)
{
	if (d == nullptr) { errno = EINVAL; return -1; }
	if (ensure_uploaded(d) != 0) return -1;
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	if (d->alt != nullptr && fsm_hip_reserve(d->alt, n) != 0) return -1;   /* the variable-length fronts run on the second image */
	DfaLock lk(d->mu);
	const hipError_t e = tb_grow(d, tb_bytes_for(n));
	if (e != hipSuccess) { errno = hip_errno(e); return -1; }
	/* ... and the tables the end-id and resume fronts build at their first call */
	(void)ensure_ids(d);
	(void)ensure_resume(d);
	errno = 0;
	return 0;
}

static int tile_bases(fsm_hip_dfa *d, const uint32_t *d_len, size_t n, hipStream_t s, const uint64_t **tbase)
{
	DfaLock lk(d->mu);
	const uint64_t T1 = (n + 63u) / 64u + 1u, nb = (T1 + 1023u) / 1024u;
	const size_t want = tb_bytes_for(n);
	hipError_t e = hipSuccess;
	if (d->tb_scratch_busy) e = hipStreamWaitEvent(s, d->tb_scratch_ev, 0);
	if (e == hipSuccess && want > d->tb_scratch_bytes) {
		/* a blocking hipMalloc: illegal while the stream is being captured into a graph -- fsm_hip_reserve() sizes the block ahead */
		hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
		if (s != nullptr && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
			if (getenv("FSM_HIP_DEBUG")) fprintf(stderr, "fsm_hip: the lengths front needs %zu bytes of scratch during a stream capture: call fsm_hip_reserve() first\n", want);
			errno = ENOMEM;
			return -1;
		}
		e = tb_grow(d, want);
	}
	if (e == hipSuccess) {
		uint64_t *tb = reinterpret_cast<uint64_t *>(d->tb_scratch), *bt = tb + T1;
		hipLaunchKernelGGL(tile_bases_pass1, dim3((unsigned)nb), dim3(1024), 0, s, d_len, (uint64_t)n, T1, tb, bt);
		hipLaunchKernelGGL(tile_bases_pass2, dim3(1), dim3(1024), 0, s, bt, nb);
		uint64_t g3 = (T1 + 255u) / 256u;
		if (g3 > (uint64_t)d->ncu * 8u) g3 = (uint64_t)d->ncu * 8u;
		hipLaunchKernelGGL(tile_bases_pass3, dim3((unsigned)g3), dim3(256), 0, s, tb, T1, (const uint64_t *)bt);
		e = hipGetLastError();
		*tbase = tb;
	}
	if (e != hipSuccess) { errno = hip_errno(e); return -1; }
	return 0;
}

/* ... and once the walk is enqueued: the block is busy until this point of the stream */
static void tile_bases_done(fsm_hip_dfa *d, hipStream_t s)
{
	DfaLock lk(d->mu);
	if (hipEventRecord(d->tb_scratch_ev, s) == hipSuccess) d->tb_scratch_busy = true;
}

static int exec_stride_device(const struct fsm_hip_dfa *d,
	const void *d_base, size_t stride, const uint32_t *d_len, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream, const BatchHint &hint)
{
	if (d == nullptr || (n != 0 && d_base == nullptr && stride != 0)) { errno = EINVAL; return -1; }
	{
		const fsm_hip_dfa *t = route(d, d_len == nullptr && stride != 0 && stride % 16u == 0 && (reinterpret_cast<uintptr_t>(d_base) % 16u) == 0);
		if (t != d) return exec_stride_device(t, d_base, stride, d_len, n, d_end_out, d_accept_bitmap, hip_stream, hint);
	}
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	if (ensure_uploaded(d) != 0) return -1;
	WalkArgs a = d->proto;
	a.base = static_cast<const uint8_t *>(d_base);
	a.stride = stride;
	a.len = d_len;
	a.off = nullptr;
	a.n = n;
	a.end_out = d_end_out;
	a.bitmap = d_accept_bitmap;
	const bool fast = d_len == nullptr && stride != 0 && stride % 16u == 0 &&
		(reinterpret_cast<uintptr_t>(d_base) % 16u) == 0 && d->knob_input_mode != IN_GENERIC;
	return launch_walk(d, a, fast, static_cast<hipStream_t>(hip_stream), hint);
}

extern "C" int fsm_hip_exec_batch_device(const struct fsm_hip_dfa *d,
	const void *d_base, size_t stride, const uint32_t *d_len, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream)
{
	return exec_stride_device(d, d_base, stride, d_len, n, d_end_out, d_accept_bitmap, hip_stream, BatchHint());
}

/* inputs packed back to back, located by u64 offsets, u32 offsets, or their lengths alone (exactly one of the three) */
static int exec_packed_device(const struct fsm_hip_dfa *d,
	const void *d_base, const uint64_t *d_off, const uint32_t *d_off32, const uint32_t *d_len, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream, const BatchHint &hint)
{
	if (d == nullptr || (n != 0 && d_off == nullptr && d_off32 == nullptr && d_len == nullptr)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	{
		const fsm_hip_dfa *t = route(d, false);
		if (t != d) return exec_packed_device(t, d_base, d_off, d_off32, d_len, n, d_end_out, d_accept_bitmap, hip_stream, hint);
	}
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	if (ensure_uploaded(d) != 0) return -1;
	WalkArgs a = d->proto;
	a.base = static_cast<const uint8_t *>(d_base);
	a.stride = 0;
	a.len = nullptr;
	a.off = d_off;
	a.off32 = d_off == nullptr ? d_off32 : nullptr;
	a.n = n;
	a.end_out = d_end_out;
	a.bitmap = d_accept_bitmap;
	const bool lenonly = d_off == nullptr && d_off32 == nullptr;
	/* ONE critical section over the pre-pass, the walk and the "block busy until here" event: another host thread's call on
	 * this dfa must not refill the tile-base block between them (the mutex is recursive: the three take it again) */
	DfaLock lk(const_cast<fsm_hip_dfa *>(d)->mu);
	if (lenonly) {
		a.len = d_len;
		if (tile_bases(const_cast<fsm_hip_dfa *>(d), d_len, n, s, &a.tbase) != 0) return -1;
	}
	const int r = launch_walk(d, a, false, s, hint);
	if (lenonly) tile_bases_done(const_cast<fsm_hip_dfa *>(d), s);
	return r;
}

static int exec_offsets_device(const struct fsm_hip_dfa *d,
	const void *d_base, const uint64_t *d_off, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream, const BatchHint &hint)
{
	if (n != 0 && d_off == nullptr) { errno = EINVAL; return -1; }
	return exec_packed_device(d, d_base, d_off, nullptr, nullptr, n, d_end_out, d_accept_bitmap, hip_stream, hint);
}

extern "C" int fsm_hip_exec_batch_offsets_device(const struct fsm_hip_dfa *d,
	const void *d_base, const uint64_t *d_off, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream)
{
	return exec_offsets_device(d, d_base, d_off, n, d_end_out, d_accept_bitmap, hip_stream, BatchHint());
}

extern "C" int fsm_hip_exec_batch_offsets32_device(const struct fsm_hip_dfa *d,
	const void *d_base, const uint32_t *d_off32, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream)
{
	if (n != 0 && d_off32 == nullptr) { errno = EINVAL; return -1; }
	return exec_packed_device(d, d_base, nullptr, d_off32, nullptr, n, d_end_out, d_accept_bitmap, hip_stream, BatchHint());
}

extern "C" int fsm_hip_exec_batch_lengths_device(const struct fsm_hip_dfa *d,
	const void *d_base, const uint32_t *d_len, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream)
{
	if (n != 0 && d_len == nullptr) { errno = EINVAL; return -1; }
	return exec_packed_device(d, d_base, nullptr, nullptr, d_len, n, d_end_out, d_accept_bitmap, hip_stream, BatchHint());
}

extern "C" const char *fsm_hip_last_kernel_name(const struct fsm_hip_dfa *d)
{
	if (d == nullptr) return "";
	if (const fsm_hip_dfa *lu = d->last_used.load(std::memory_order_relaxed); lu != nullptr && lu != d) return fsm_hip_last_kernel_name(lu);   /* the launch went to the second image */
	fsm_hip_dfa *md = const_cast<fsm_hip_dfa *>(d);
	DfaLock lk(md->mu);
	if (md->last_pick_flag != nullptr) {
		/* a device-pointer variable-length call: the choice between the two kernels was made on the device (offsets_pick): read it
		 * back (a 4-byte copy; waits for the launch) and name the kernel that walked the batch, as rocprofv3 would show it busy */
		uint32_t flag = 3;
		DevGuard dg(d->device);
		if (dg.ok() && md->timed && hipEventSynchronize(md->ev1) == hipSuccess &&
		    hipMemcpy(&flag, md->last_pick_flag, sizeof flag, hipMemcpyDeviceToHost) == hipSuccess && flag <= 2u && !md->last_kernel_pick[flag].empty()) {
			int cands = 0;
			for (const auto &nm : md->last_kernel_pick) cands += nm.empty() ? 0 : 1;
			md->last_kernel = md->last_kernel_pick[flag];
			md->last_kernel += flag == PICK_RAGGED ? " (decided on the device" : flag == PICK_LINES32 ? " (picked on the device: short lines, a batch below 4 GiB"
			                                                                  : " (mean length below the pick threshold, decided on the device";
			md->last_kernel += cands == 2 ? ", 1 of 2 launched)" : cands == 3 ? ", 1 of 3 launched)" : ")";
		}
		md->last_pick_flag = nullptr;
	}
	return d->last_kernel.c_str();
}

extern "C" double fsm_hip_last_kernel_ms(const struct fsm_hip_dfa *d)
{
	if (d == nullptr) return -1.0;
	if (const fsm_hip_dfa *lu = d->last_used.load(std::memory_order_relaxed); lu != nullptr && lu != d) return fsm_hip_last_kernel_ms(lu);
	DfaLock lk(const_cast<fsm_hip_dfa *>(d)->mu);
	if (!d->timed) return -1.0;
	float ms = 0.f;
	if (hipEventSynchronize(d->ev1) != hipSuccess) return -1.0;
	if (hipEventElapsedTime(&ms, d->ev0, d->ev1) != hipSuccess) return -1.0;
	return (double)ms;
}

/* ------------------------------------------------------------------ */
/* host-buffer front: stage through HBM                                */
/* ------------------------------------------------------------------ */

/*
 * Host-pointer fronts.  One device arena per dfa (grow-only up to ARENA_KEEP, so repeated calls --
 * retest / re(1) issue one per input line -- do no hipMalloc/hipFree), laid out
 *     [inputs ...] [in/out arrays ...] [outputs ...]
 * Small calls gather their host arrays in one pinned buffer: one H2D copy (inputs + in/out), the
 * kernel, one D2H copy (in/out + outputs), one stream synchronise.  Large calls copy each array
 * straight from / to the caller's pages.
 */
static const size_t ARENA_KEEP = (size_t)256 << 20, STAGE_BYTES = (size_t)1 << 20;

static size_t up256(size_t x) { return (x + 255u) & ~(size_t)255u; }

struct HostCall {
	enum Kind { IN = 0, INOUT = 1, OUT = 2 };
	struct Part { const void *src; void *dst; size_t bytes, pad, off; int kind; };
	fsm_hip_dfa *d;
	DfaLock lk;       /* arena and staging buffer are shared by the calls on one dfa */
	Part parts[8];
	int np = 0;
	unsigned char *arena = nullptr;
	bool temp = false, small = false;
	size_t h2d_end = 0, d2h_begin = 0, total = 0;

	explicit HostCall(const fsm_hip_dfa *cd) : d(const_cast<fsm_hip_dfa *>(cd)), lk(d->mu) {}
	~HostCall() { if (temp && arena) { int e = errno; (void)hipFree(arena); errno = e; } }
	/* declare the arrays in the order IN..., INOUT..., OUT...; returns the part's index (or -1 for a NULL array) */
	int add(int kind, const void *src, void *dst, size_t bytes, size_t pad = 0)
	{
		if (kind != IN && dst == nullptr) return -1;
		if (kind == IN && src == nullptr && bytes != 0) return -1;
		Part &p = parts[np];
		p.src = src; p.dst = dst; p.bytes = bytes; p.pad = pad; p.kind = kind; p.off = 0;
		return np++;
	}
	template <class T> T *dev(int part) const { return part < 0 ? nullptr : reinterpret_cast<T *>(arena + parts[part].off); }

	int begin()
	{
		if (ensure_uploaded(d) != 0) return -1;   /* FSM_HIP_DEFER_UPLOAD: the private stream and the tables come with the first call */
		size_t o = 0;
		h2d_end = 0;
		d2h_begin = (size_t)-1;
		for (int i = 0; i < np; i++) {   /* declared in the order IN, INOUT, OUT */
			Part &p = parts[i];
			p.off = o;
			o += up256(p.bytes + p.pad);
			if (p.kind != OUT) h2d_end = o;
			if (p.kind != IN && d2h_begin == (size_t)-1) d2h_begin = p.off;
		}
		total = o;
		if (d2h_begin == (size_t)-1) d2h_begin = total;
		if (total <= d->arena_bytes) {
			arena = d->arena;
		} else if (total <= ARENA_KEEP) {
			if (d->arena) { (void)hipFree(d->arena); d->arena = nullptr; d->arena_bytes = 0; }
			size_t want = (size_t)2 << 20;
			while (want < total) want *= 2;
			HIP_TRY(hipMalloc((void **)&d->arena, want));
			d->arena_bytes = want;
			arena = d->arena;
		} else {
			HIP_TRY(hipMalloc((void **)&arena, total));
			temp = true;
		}
		small = total <= STAGE_BYTES;
		if (small && d->stage == nullptr) HIP_TRY(hipHostMalloc((void **)&d->stage, STAGE_BYTES, hipHostMallocDefault));
		if (small) {
			for (int i = 0; i < np; i++)
				if (parts[i].kind != OUT && parts[i].bytes) memcpy(d->stage + parts[i].off, parts[i].src, parts[i].bytes);
			if (h2d_end) HIP_TRY(hipMemcpyAsync(arena, d->stage, h2d_end, hipMemcpyHostToDevice, d->hs));
		} else {
			for (int i = 0; i < np; i++)
				if (parts[i].kind != OUT && parts[i].bytes)
					HIP_TRY(hipMemcpyAsync(arena + parts[i].off, parts[i].src, parts[i].bytes, hipMemcpyHostToDevice, d->hs));
		}
		return 0;
	fail:
		return -1;
	}

	int end()
	{
		if (small) {
			if (total > d2h_begin) HIP_TRY(hipMemcpyAsync(d->stage + d2h_begin, arena + d2h_begin, total - d2h_begin, hipMemcpyDeviceToHost, d->hs));
			HIP_TRY(hipStreamSynchronize(d->hs));
			for (int i = 0; i < np; i++)
				if (parts[i].kind != IN && parts[i].bytes) memcpy(parts[i].dst, d->stage + parts[i].off, parts[i].bytes);
		} else {
			for (int i = 0; i < np; i++)
				if (parts[i].kind != IN && parts[i].bytes)
					HIP_TRY(hipMemcpyAsync(parts[i].dst, arena + parts[i].off, parts[i].bytes, hipMemcpyDeviceToHost, d->hs));
			HIP_TRY(hipStreamSynchronize(d->hs));
		}
		return 0;
	fail:
		return -1;
	}
};

static int exec_host(const struct fsm_hip_dfa *d,
	const unsigned char *base, size_t in_bytes, size_t stride,
	const uint32_t *len, const uint64_t *off, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap, const uint32_t *off32 = nullptr, bool packed_len = false)
{
	if (d == nullptr) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	HostCall hc(d);
	/* +32: the generic kernel reads whole aligned 16-byte chunks */
	const int p_in = hc.add(HostCall::IN, base, nullptr, in_bytes, 32);
	const int p_len = len ? hc.add(HostCall::IN, len, nullptr, n * sizeof(uint32_t)) : -1;
	const int p_off = off ? hc.add(HostCall::IN, off, nullptr, (n + 1) * sizeof(uint64_t)) : -1;
	const int p_off32 = off32 ? hc.add(HostCall::IN, off32, nullptr, (n + 1) * sizeof(uint32_t)) : -1;
	const int p_end = hc.add(HostCall::OUT, nullptr, end_out, n * sizeof(uint32_t));
	const int p_bm = hc.add(HostCall::OUT, nullptr, accept_bitmap, ((n + 63) / 64) * sizeof(uint64_t));
	if (hc.begin() != 0) return -1;
	BatchHint hint;
	hint.bytes = in_bytes;
	if (off != nullptr || off32 != nullptr || packed_len) {
		/* a plain walk of packed lines: walk_lines32 is the per-lane kernel (below 4 GiB and 2^29 lines, not the record walk) */
		const bool l32 = in_bytes < ((uint64_t)1 << 32) && n < 0x1FFFFFF0ull && d->plan.layout != FSM_HIP_LAYOUT_SPARSE;
		hint.short_mean = in_bytes / n < (size_t)pick_mean_of(d, l32);
	} else if (len != nullptr) {   /* the average of the lengths, not of the rows they sit in */
		uint64_t sum = 0;
		for (size_t i = 0; i < n; i++) sum += len[i];
		hint.short_mean = sum / n < (uint64_t)pick_mean_of(d, false);
	}
	if (off || off32 || packed_len) {
		if (exec_packed_device(d, hc.dev<unsigned char>(p_in), hc.dev<uint64_t>(p_off), hc.dev<uint32_t>(p_off32), packed_len ? hc.dev<uint32_t>(p_len) : nullptr, n,
		                       hc.dev<uint32_t>(p_end), hc.dev<uint64_t>(p_bm), hc.d->hs, hint) != 0) return -1;
	} else {
		if (exec_stride_device(d, hc.dev<unsigned char>(p_in), stride, hc.dev<uint32_t>(p_len), n,
		                       hc.dev<uint32_t>(p_end), hc.dev<uint64_t>(p_bm), hc.d->hs, hint) != 0) return -1;
	}
	return hc.end();
}

extern "C" int fsm_hip_exec_batch(const struct fsm_hip_dfa *d,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap)
{
	if (n != 0 && base == nullptr && stride != 0) { errno = EINVAL; return -1; }
	if (len != nullptr)
		for (size_t i = 0; i < n; i++)
			if (len[i] > stride) { errno = EINVAL; return -1; }
	return exec_host(d, base, n * stride, stride, len, nullptr, n, end_out, accept_bitmap);
}

extern "C" int fsm_hip_exec_batch_offsets(const struct fsm_hip_dfa *d,
	const unsigned char *base, const uint64_t *off, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap)
{
	if (n != 0 && off == nullptr) { errno = EINVAL; return -1; }
	for (size_t i = 0; i < n; i++)
		if (off[i + 1] < off[i]) { errno = EINVAL; return -1; }
	const size_t total = n ? (size_t)off[n] : 0;
	if (total != 0 && base == nullptr) { errno = EINVAL; return -1; }
	return exec_host(d, base, total, 0, nullptr, off, n, end_out, accept_bitmap);
}

extern "C" int fsm_hip_exec_batch_offsets32(const struct fsm_hip_dfa *d,
	const unsigned char *base, const uint32_t *off32, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap)
{
	if (n != 0 && off32 == nullptr) { errno = EINVAL; return -1; }
	for (size_t i = 0; i < n; i++)
		if (off32[i + 1] < off32[i]) { errno = EINVAL; return -1; }
	const size_t total = n ? (size_t)off32[n] : 0;
	if (total != 0 && base == nullptr) { errno = EINVAL; return -1; }
	return exec_host(d, base, total, 0, nullptr, nullptr, n, end_out, accept_bitmap, off32);
}

extern "C" int fsm_hip_exec_batch_lengths(const struct fsm_hip_dfa *d,
	const unsigned char *base, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap)
{
	if (n != 0 && len == nullptr) { errno = EINVAL; return -1; }
	size_t total = 0;
	for (size_t i = 0; i < n; i++) total += len[i];
	if (total != 0 && base == nullptr) { errno = EINVAL; return -1; }
	return exec_host(d, base, total, 0, len, nullptr, n, end_out, accept_bitmap, nullptr, true);
}

extern "C" double fsm_hip_lds_chain_probe_gbps(size_t table_bytes, int waves, int blocks_per_cu, size_t steps, void *d_scratch4, void *hip_stream)
{
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	hipEvent_t e0 = nullptr, e1 = nullptr;
	float ms = -1.f;
	double gbps = -1.0;
	if (d_scratch4 == nullptr || table_bytes < 2048 || table_bytes > 160u * 1024u || waves < 1 || waves > 16 || blocks_per_cu < 1 || steps < 16) { errno = EINVAL; return -1.0; }
	int dev = 0, ncu = 256;
	(void)hipGetDevice(&dev);
	(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
	const uint32_t words = (uint32_t)(table_bytes / 4u);
	const unsigned grid = (unsigned)(ncu * blocks_per_cu);
	HIP_TRY(hipFuncSetAttribute((const void *)lds_chain_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)table_bytes));
	for (int rep = 0; rep < 2; rep++) {   /* the second launch is the timed one */
		if (e0 == nullptr) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); }
		HIP_TRY(hipEventRecord(e0, s));
		hipLaunchKernelGGL(lds_chain_probe_kernel, dim3(grid), dim3((unsigned)waves * 64u), table_bytes, s, words, (uint32_t)(steps / 16u * 16u), static_cast<uint32_t *>(d_scratch4));
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(e1, s));
		HIP_TRY(hipEventSynchronize(e1));
		HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	}
	gbps = (double)grid * waves * 64.0 * (double)(steps / 16u * 16u) / ((double)ms * 1e-3) / 1e9;
fail:
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	return gbps;
}

/* ------------------------------------------------------------------ */
/* info / tuning / end-ids                                            */
/* ------------------------------------------------------------------ */

extern "C" int fsm_hip_dfa_info(const struct fsm_hip_dfa *d, struct fsm_hip_dfa_info *out)
{
	if (d == nullptr || out == nullptr) { errno = EINVAL; return -1; }
	if (ensure_uploaded(d) != 0) return -1;
	memset(out, 0, sizeof *out);
	const Plan &p = d->plan;
	out->nstates = p.nstates;
	out->nclasses = p.C;
	out->layout = p.layout;
	out->nabsorbing = p.nabsorbing;
	switch (p.layout) {
	case FSM_HIP_LAYOUT_TINY: out->table_bytes = p.tiny5_col.empty() ? 256 * 8 : 256 * 4; break;
	case FSM_HIP_LAYOUT_LDS:
	case FSM_HIP_LAYOUT_LDS2:
	case FSM_HIP_LAYOUT_LDSSELF: out->table_bytes = p.lds_tab.size() * 2; break;
	case FSM_HIP_LAYOUT_COMB: out->table_bytes = p.comb.size() * 4 + 1024; break;
	case FSM_HIP_LAYOUT_COMB256: out->table_bytes = p.comb256.size() * 4; break;
	case FSM_HIP_LAYOUT_COMBSELF: out->table_bytes = p.comb.size() * 10 + 256; break;
	case FSM_HIP_LAYOUT_SPARSE: out->table_bytes = p.sparse_img.size() * 4; break;
	default: out->table_bytes = p.glob_tab16.empty() ? p.glob_tab.size() * 4 : p.glob_tab16.size() * 2; break;
	}
	LaunchCfg c = pick_cfg(d, true, 1024, 0);
	out->lds_bytes = c.lds;
	out->waves_per_block = (uint32_t)c.waves;
	out->device = (uint32_t)d->device;
	return 0;
}

extern "C" int fsm_hip_dfa_tune(struct fsm_hip_dfa *d, int knob, int value)
{
	if (d == nullptr) { errno = EINVAL; return -1; }
	if (ensure_uploaded(d) != 0) return -1;
	if (d->alt != nullptr) (void)fsm_hip_dfa_tune(d->alt, knob, value);   /* the knobs of the variable-length kernels live there */
	switch (knob) {
	case FSM_HIP_KNOB_INPUT_MODE: d->knob_input_mode = value; break;
	case FSM_HIP_KNOB_NB: d->knob_nb = value; break;
	case FSM_HIP_KNOB_ROWS: d->knob_rows = value; break;   /* the lazy walk's inputs per lane (2 / 3 / 4: A/B); the table walks have one */
	case FSM_HIP_KNOB_LAZY_DYN: d->knob_lazy_dyn = value != 0; break;
	case FSM_HIP_KNOB_LAZY_LINES: d->knob_lazy_lines = value != 0; break;
	case FSM_HIP_KNOB_MASK: break;   /* retired: exec-masking absorbing lanes cost more than it saved */
	case FSM_HIP_KNOB_SEG: d->knob_seg = value; break;
	case FSM_HIP_KNOB_PREFETCH: d->knob_prefetch = value; break;
	case FSM_HIP_KNOB_NT: d->knob_nt = value; break;
	case FSM_HIP_KNOB_HOT_BYTES:
		if (d->plan.layout != FSM_HIP_LAYOUT_GLOBAL || value < 0) { errno = EINVAL; return -1; }
		set_hot_bytes(d, (uint32_t)value);
		break;
	case FSM_HIP_KNOB_WAVES: d->knob_waves = value; break;
	case FSM_HIP_KNOB_BLOCKS_PER_CU: d->knob_blocks_per_cu = value; break;
	case FSM_HIP_KNOB_EARLY_RETIRE:
		/* bit 128 is round 4's resource bound, kept for the test that pins the range rule: it LOSES bytes.  Not for a stray value */
		if (value >= 0 && (value & 128) && getenv("FSM_HIP_TEST_KNOBS") == nullptr) { errno = EINVAL; return -1; }
		d->knob_early = value;
		break;
	case FSM_HIP_KNOB_NOSKIP: d->knob_noskip = value; break;
	case FSM_HIP_KNOB_RAGGED_ALIGN: break;   /* retired: segments start at the input's own first byte now */
	case FSM_HIP_KNOB_PK_RMIN:
	case FSM_HIP_KNOB_PK_RMAX:
	case FSM_HIP_KNOB_PK_DEBUG: break;   /* retired with walk_packed (round 4): accepted, ignored */
	case FSM_HIP_KNOB_SPARSE_FAST: d->knob_sparse_fast = value < 0 || value > 3 ? (d->d_lazy ? 3 : 1) : value; break;
	case FSM_HIP_KNOB_PICK_MEAN: if (value < -1) { errno = EINVAL; return -1; } d->knob_pick_mean = value; break;   /* -1: the defaults */
	case FSM_HIP_KNOB_DMA_BUFS: break;   /* retired: two DMA tiles per wave measured slower (profiles/r02m_ab_one_vs_two_dma_tiles.txt) */
	default: errno = EINVAL; return -1;
	}
	return 0;
}

extern "C" size_t fsm_hip_endid_count(const struct fsm_hip_dfa *d, uint32_t end_state)
{
	if (d == nullptr || end_state >= d->plan.nstates) return 0;
	return d->plan.endid_off[end_state + 1] - d->plan.endid_off[end_state];
}

extern "C" int fsm_hip_endid_get(const struct fsm_hip_dfa *d, uint32_t end_state,
	size_t id_buf_count, uint32_t *id_buf)
{
	if (d == nullptr || end_state >= d->plan.nstates) return 0;
	const uint32_t a = d->plan.endid_off[end_state], b = d->plan.endid_off[end_state + 1];
	if (b - a > id_buf_count) return 0; /* fsm_endid_get: 0 = buffer too small */
	for (uint32_t k = a; k < b; k++) id_buf[k - a] = d->plan.endids[k];
	return 1;
}

/* ------------------------------------------------------------------ */
/* plan inspection (no device needed; used by the CPU-side tests)      */
/* ------------------------------------------------------------------ */

struct fsm_hip_plan { Plan p; };

extern "C" struct fsm_hip_plan *fsm_hip_plan_create(const struct fsm_hip_dfa_desc *desc, unsigned flags, uint32_t lds_limit)
{
	fsm_hip_plan *pl = new (std::nothrow) fsm_hip_plan();
	if (pl == nullptr) { errno = ENOMEM; return nullptr; }
	int r = build_plan(desc, flags, lds_limit ? lds_limit : 160u * 1024u, pl->p);
	if (r != 0) { delete pl; errno = r; return nullptr; }
	return pl;
}

extern "C" void fsm_hip_plan_free(struct fsm_hip_plan *pl) { delete pl; }

extern "C" int fsm_hip_plan_get(const struct fsm_hip_plan *pl, int what, const void **data, size_t *count)
{
	if (pl == nullptr || data == nullptr || count == nullptr) { errno = EINVAL; return -1; }
	const Plan &p = pl->p;
	static thread_local uint32_t scalars[20];
	switch (what) {
	case FSM_HIP_PLAN_SCALARS:
		scalars[0] = p.nstates; scalars[1] = p.S1; scalars[2] = p.start; scalars[3] = p.C;
		scalars[4] = p.abs_min; scalars[5] = p.nabsorbing; scalars[6] = p.layout; scalars[7] = p.row_bytes;
		scalars[8] = p.comb_abs_min_off; scalars[9] = p.comb256_abs_min_off; scalars[10] = p.comb256_dflt;
		scalars[11] = p.eager_lo_end; scalars[12] = p.eager_hi_begin;
		scalars[13] = p.comb_eager_lo_off; scalars[14] = p.comb_eager_hi_off;
		scalars[15] = p.comb256_eager_lo_off; scalars[16] = p.comb256_eager_hi_off;
		*data = scalars; *count = 17; return 0;
	case FSM_HIP_PLAN_CLS: *data = p.cls; *count = 256; return 0;
	case FSM_HIP_PLAN_NEW2OLD: *data = p.new2old.data(); *count = p.new2old.size(); return 0;
	case FSM_HIP_PLAN_FIN: *data = p.fin.data(); *count = p.fin.size(); return 0;
	case FSM_HIP_PLAN_DENSE: *data = p.dense.data(); *count = p.dense.size(); return 0;
	case FSM_HIP_PLAN_TINY_COL: *data = p.tiny_col.data(); *count = p.tiny_col.size(); return 0;
	case FSM_HIP_PLAN_TINY5_COL: *data = p.tiny5_col.data(); *count = p.tiny5_col.size(); return 0;
	case FSM_HIP_PLAN_LDS_TAB: *data = p.lds_tab.data(); *count = p.lds_tab.size(); return 0;
	case FSM_HIP_PLAN_COMB: *data = p.comb.data(); *count = p.comb.size(); return 0;
	case FSM_HIP_PLAN_COMB_DFLT: *data = p.comb_dflt.data(); *count = p.comb_dflt.size(); return 0;
	case FSM_HIP_PLAN_COMB_OFF: *data = p.comb_off.data(); *count = p.comb_off.size(); return 0;
	case FSM_HIP_PLAN_COMB_FIN: *data = p.comb_fin.data(); *count = p.comb_fin.size(); return 0;
	case FSM_HIP_PLAN_GLOB_TAB: *data = p.glob_tab.data(); *count = p.glob_tab.size(); return 0;
	case FSM_HIP_PLAN_GLOB_TAB16: *data = p.glob_tab16.data(); *count = p.glob_tab16.size(); return 0;
	case FSM_HIP_PLAN_GLOB16_RANK: *data = p.glob16_rank.data(); *count = p.glob16_rank.size(); return 0;
	case FSM_HIP_PLAN_SPARSE: *data = p.sparse_img.data(); *count = p.sparse_img.size(); return 0;
	case FSM_HIP_PLAN_LAZY: *data = p.lazy_img.data(); *count = p.lazy_img.size(); return 0;
	case FSM_HIP_PLAN_COMB256: *data = p.comb256.data(); *count = p.comb256.size(); return 0;
	case FSM_HIP_PLAN_COMB256_OFF: *data = p.comb256_off.data(); *count = p.comb256_off.size(); return 0;
	case FSM_HIP_PLAN_COMB256_FIN: *data = p.comb256_fin.data(); *count = p.comb256_fin.size(); return 0;
	case FSM_HIP_PLAN_COMB_SMASK: *data = p.comb_smask.data(); *count = p.comb_smask.size(); return 0;
	case FSM_HIP_PLAN_COMB_RNG: *data = p.comb_rng.data(); *count = p.comb_rng.size(); return 0;
	case FSM_HIP_PLAN_EMASK: *data = p.emask.data(); *count = p.emask.size(); return 0;
	case FSM_HIP_PLAN_EAGER_IDS: *data = p.eager_ids.data(); *count = p.eager_ids.size(); return 0;
	case FSM_HIP_PLAN_EW_OFF: *data = p.ew_off.data(); *count = p.ew_off.size(); return 0;
	case FSM_HIP_PLAN_EW_WORD: *data = p.ew_word.data(); *count = p.ew_word.size(); return 0;
	case FSM_HIP_PLAN_EW_MASK: *data = p.ew_mask.data(); *count = p.ew_mask.size(); return 0;
	default: errno = EINVAL; return -1;
	}
}

/* ------------------------------------------------------------------ */
/* synthetic generator                                                */
/* ------------------------------------------------------------------ */

static int fill_gen(GenArgs &g, void *base, size_t stride, size_t n, uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *plant, unsigned plant_len, unsigned plant_every)
{
	memset(&g, 0, sizeof g);
	if (nalpha > 256 || plant_len > sizeof g.plant || (plant_len && (plant_every == 0 || plant_len > stride))) return -1;
	g.base = static_cast<unsigned char *>(base);
	g.stride = stride; g.n = n; g.first_index = first_index; g.seed = seed;
	g.nalpha = alphabet ? nalpha : 0;
	if (g.nalpha) memcpy(g.alphabet, alphabet, g.nalpha);
	g.plant_len = plant ? plant_len : 0;
	g.plant_every = plant_every;
	if (g.plant_len) memcpy(g.plant, plant, g.plant_len);
	return 0;
}

extern "C" int fsm_hip_gen_inputs_device(void *d_base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *plant, unsigned plant_len, unsigned plant_every,
	void *hip_stream)
{
	GenArgs g;
	if (stride == 0 || stride % 8u != 0 || (reinterpret_cast<uintptr_t>(d_base) % 8u) != 0 ||
	    fill_gen(g, d_base, stride, n, first_index, seed, alphabet, nalpha, plant, plant_len, plant_every) != 0) {
		errno = EINVAL;
		return -1;
	}
	if (n == 0) return 0;
	uint64_t total = (uint64_t)n * (stride / 8u);
	uint64_t blocks = (total + 255) / 256;
	if (blocks > 256u * 64u) blocks = 256u * 64u;
	hipLaunchKernelGGL(gen_inputs_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), g);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) { errno = hip_errno(e); return -1; }
	return 0;
}

extern "C" void fsm_hip_gen_inputs_host(unsigned char *base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *plant, unsigned plant_len, unsigned plant_every)
{
	GenArgs g;
	if (fill_gen(g, base, stride, n, first_index, seed, alphabet, nalpha, plant, plant_len, plant_every) != 0) return;
	for (size_t row = 0; row < n; row++) {
		const uint64_t gi = first_index + row;
		unsigned char *p = base + row * stride;
		for (size_t t = 0; t < stride; t += 8) {
			uint64_t v = gen_word(g, gi, t / 8);
			for (size_t k = 0; k < 8 && t + k < stride; k++) p[t + k] = (unsigned char)(v >> (8 * k));
		}
		if (g.plant_len != 0 && gi % g.plant_every == 0)
			memcpy(p + plant_offset(g, gi), g.plant, g.plant_len);
	}
}

extern "C" int fsm_hip_gen_pack_rows_device(const void *d_rows, size_t stride, const uint32_t *d_len, const uint64_t *d_off, size_t n,
	size_t max_len, void *d_out, void *hip_stream)
{
	if (n == 0) return 0;
	if (d_rows == nullptr || d_len == nullptr || d_off == nullptr || d_out == nullptr || max_len > stride) { errno = EINVAL; return -1; }
	const uint64_t wpr = (max_len + 7u) / 8u;
	if (wpr == 0) return 0;
	uint64_t blocks = ((uint64_t)n * wpr + 255u) / 256u;
	if (blocks > 256u * 64u) blocks = 256u * 64u;
	hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
	                   static_cast<const unsigned char *>(d_rows), (uint64_t)stride, d_len, d_off, (uint64_t)n, wpr, static_cast<unsigned char *>(d_out));
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) { errno = hip_errno(e); return -1; }
	return 0;
}

/* ---- affix generator (see walk_kernels.h) ---- */

static int fill_affix(AffixArgs &x, const unsigned char *body, unsigned nbody, const unsigned char *body2, unsigned nbody2,
	unsigned npfx, unsigned nsfx, unsigned every)
{
	memset(&x, 0, sizeof x);
	if (body == nullptr || nbody == 0 || nbody > 256 || npfx == 0 || nsfx == 0 || every == 0 ||
	    nbody2 > sizeof x.body2 || (nbody2 != 0 && body2 == nullptr)) return -1;
	x.npfx = npfx; x.nsfx = nsfx; x.every = every; x.nbody = nbody; x.nbody2 = nbody2;
	memcpy(x.body, body, nbody);
	if (nbody2) memcpy(x.body2, body2, nbody2);
	return 0;
}

static int check_affixes(const unsigned char *t, unsigned n, size_t stride)
{
	if (t == nullptr) return -1;
	for (unsigned i = 0; i < n; i++) if (t[8 * i] > 7 || t[8 * i] > stride / 2) return -1;
	return 0;
}

extern "C" int fsm_hip_gen_affix2_inputs_device(void *d_base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *body, unsigned nbody,
	const unsigned char *body2, unsigned nbody2,
	const unsigned char *prefixes, unsigned npfx,
	const unsigned char *suffixes, unsigned nsfx,
	unsigned every, void *hip_stream)
{
	GenArgs g;
	AffixArgs x;
	unsigned char *d_tab = nullptr;
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	int rc = -1;
	if (stride == 0 || stride % 8u != 0 || (reinterpret_cast<uintptr_t>(d_base) % 8u) != 0 ||
	    fill_gen(g, d_base, stride, n, first_index, seed, alphabet, nalpha, nullptr, 0, 0) != 0 ||
	    fill_affix(x, body, nbody, body2, nbody2, npfx, nsfx, every) != 0 ||
	    check_affixes(prefixes, npfx, stride) != 0 || check_affixes(suffixes, nsfx, stride) != 0) {
		errno = EINVAL;
		return -1;
	}
	if (n == 0) return 0;
	HIP_TRY(hipMalloc((void **)&d_tab, 8u * ((size_t)npfx + nsfx)));
	HIP_TRY(hipMemcpy(d_tab, prefixes, 8u * (size_t)npfx, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(d_tab + 8u * (size_t)npfx, suffixes, 8u * (size_t)nsfx, hipMemcpyHostToDevice));
	x.pfx = d_tab;
	x.sfx = d_tab + 8u * (size_t)npfx;
	{
		uint64_t total = (uint64_t)n * (stride / 8u);
		uint64_t blocks = (total + 255) / 256;
		if (blocks > 256u * 64u) blocks = 256u * 64u;
		hipLaunchKernelGGL(gen_affix_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, x);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(s)); /* the table is freed below */
	}
	rc = 0;
fail:
	{
		int e = errno;
		if (d_tab) (void)hipFree(d_tab);
		errno = e;
	}
	return rc;
}

extern "C" int fsm_hip_gen_affix_inputs_device(void *d_base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *body, unsigned nbody,
	const unsigned char *prefixes, unsigned npfx,
	const unsigned char *suffixes, unsigned nsfx,
	unsigned every, void *hip_stream)
{
	return fsm_hip_gen_affix2_inputs_device(d_base, stride, n, first_index, seed, alphabet, nalpha, body, nbody, nullptr, 0,
	                                        prefixes, npfx, suffixes, nsfx, every, hip_stream);
}

extern "C" void fsm_hip_gen_affix2_inputs_host(unsigned char *base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *body, unsigned nbody,
	const unsigned char *body2, unsigned nbody2,
	const unsigned char *prefixes, unsigned npfx,
	const unsigned char *suffixes, unsigned nsfx,
	unsigned every)
{
	GenArgs g;
	AffixArgs x;
	if (fill_gen(g, base, stride, n, first_index, seed, alphabet, nalpha, nullptr, 0, 0) != 0 ||
	    fill_affix(x, body, nbody, body2, nbody2, npfx, nsfx, every) != 0 ||
	    check_affixes(prefixes, npfx, stride) != 0 || check_affixes(suffixes, nsfx, stride) != 0) return;
	x.pfx = prefixes;
	x.sfx = suffixes;
	for (size_t row = 0; row < n; row++) {
		unsigned char *p = base + row * stride;
		for (size_t t = 0; t < stride; t += 8) {
			uint64_t v = affix_word(g, x, first_index + row, t / 8);
			for (size_t k = 0; k < 8 && t + k < stride; k++) p[t + k] = (unsigned char)(v >> (8 * k));
		}
	}
}

extern "C" void fsm_hip_gen_affix_inputs_host(unsigned char *base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *body, unsigned nbody,
	const unsigned char *prefixes, unsigned npfx,
	const unsigned char *suffixes, unsigned nsfx,
	unsigned every)
{
	fsm_hip_gen_affix2_inputs_host(base, stride, n, first_index, seed, alphabet, nalpha, body, nbody, nullptr, 0,
	                               prefixes, npfx, suffixes, nsfx, every);
}

/* ---- HBM read-stream probe ---- */

extern "C" double fsm_hip_stream_read_probe_ms(const void *d_base, size_t bytes, void *d_scratch4, int reps, void *hip_stream)
{
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	hipEvent_t e0 = nullptr, e1 = nullptr;
	float ms = -1.f;
	if (d_base == nullptr || d_scratch4 == nullptr || bytes < 16 || reps <= 0 ||
	    (reinterpret_cast<uintptr_t>(d_base) % 16u) != 0) { errno = EINVAL; return -1.0; }
	int dev = 0, ncu = 256;
	(void)hipGetDevice(&dev);
	(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	for (int nt = 0; nt < 2; nt++) {   /* plain and nontemporal loads: the faster of the two is reported */
		void (*k)(const u32x4 *, uint64_t, uint32_t *) = nt ? stream_read_kernel<true> : stream_read_kernel<false>;
		float t = -1.f;
		hipLaunchKernelGGL(k, dim3((unsigned)ncu * 8u), dim3(256), 0, s,
		                   static_cast<const u32x4 *>(d_base), (uint64_t)(bytes / 16u), static_cast<uint32_t *>(d_scratch4));
		HIP_TRY(hipEventRecord(e0, s));
		for (int r = 0; r < reps; r++)
			hipLaunchKernelGGL(k, dim3((unsigned)ncu * 8u), dim3(256), 0, s,
			                   static_cast<const u32x4 *>(d_base), (uint64_t)(bytes / 16u), static_cast<uint32_t *>(d_scratch4));
		HIP_TRY(hipEventRecord(e1, s));
		HIP_TRY(hipEventSynchronize(e1));
		HIP_TRY(hipEventElapsedTime(&t, e0, e1));
		t /= (float)reps;
		if (ms < 0.f || t < ms) ms = t;
	}
	if (bytes >= ((size_t)64 << 20)) {
		/* third candidate: the walk's own input path without the walk -- LDS-DMA of 128-byte segments of
		 * 1 KiB rows into per-wave tiles (6.5-7.06 TB/s; a few waves per CU already saturate it) */
		const uint64_t stride = 1024, nrows = (bytes / stride) / 64u * 64u;
		for (int waves = 3; waves <= 12; waves *= 2) {   /* 3, 6, 12 waves per workgroup, two workgroups per CU */
			const size_t ldsb = (size_t)waves * 8192u;
			float t = -1.f;
			HIP_TRY(hipFuncSetAttribute((const void *)dma_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
			hipLaunchKernelGGL(dma_stream_kernel, dim3((unsigned)ncu * 2u), dim3((unsigned)waves * 64u), ldsb, s,
			                   static_cast<const uint8_t *>(d_base), nrows, stride, static_cast<uint32_t *>(d_scratch4));
			HIP_TRY(hipEventRecord(e0, s));
			for (int r = 0; r < reps; r++)
				hipLaunchKernelGGL(dma_stream_kernel, dim3((unsigned)ncu * 2u), dim3((unsigned)waves * 64u), ldsb, s,
				                   static_cast<const uint8_t *>(d_base), nrows, stride, static_cast<uint32_t *>(d_scratch4));
			HIP_TRY(hipEventRecord(e1, s));
			HIP_TRY(hipEventSynchronize(e1));
			HIP_TRY(hipEventElapsedTime(&t, e0, e1));
			t = t / (float)reps * (float)((double)bytes / (double)(nrows * stride));   /* normalise to `bytes` */
			if (t > 0.f && t < ms) ms = t;
		}
	}
fail:
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	return (double)ms;
}

extern "C" double fsm_hip_gather_probe_ms(const void *d_base, size_t bytes, size_t ngathers, int vec_bytes, void *d_scratch4, void *hip_stream)
{
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	hipEvent_t e0 = nullptr, e1 = nullptr;
	float ms = -1.f;
	if (d_base == nullptr || d_scratch4 == nullptr || (vec_bytes != 4 && vec_bytes != 16) || bytes < 16 || ngathers == 0 ||
	    (reinterpret_cast<uintptr_t>(d_base) % 16u) != 0) { errno = EINVAL; return -1.0; }
	int dev = 0, ncu = 256;
	(void)hipGetDevice(&dev);
	(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
	HIP_TRY(hipEventCreate(&e0));
	HIP_TRY(hipEventCreate(&e1));
	HIP_TRY(hipEventRecord(e0, s));
	if (vec_bytes == 16)
		hipLaunchKernelGGL(gather_probe_kernel<16>, dim3((unsigned)ncu * 8u), dim3(256), 0, s, static_cast<const unsigned char *>(d_base),
		                   (uint64_t)(bytes / 16u), (uint64_t)ngathers, static_cast<uint32_t *>(d_scratch4));
	else
		hipLaunchKernelGGL(gather_probe_kernel<4>, dim3((unsigned)ncu * 8u), dim3(256), 0, s, static_cast<const unsigned char *>(d_base),
		                   (uint64_t)(bytes / 4u), (uint64_t)ngathers, static_cast<uint32_t *>(d_scratch4));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(e1, s));
	HIP_TRY(hipEventSynchronize(e1));
	HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
fail:
	if (e0) (void)hipEventDestroy(e0);
	if (e1) (void)hipEventDestroy(e1);
	return (double)ms;
}

/* ------------------------------------------------------------------ */
/* end-ids delivered by the device                                    */
/* ------------------------------------------------------------------ */

#include <algorithm>
#include <map>

/* Build, once, the per-encoded-state tables the kernel copies into id_out[]:
 *   earliest: lowest end-id of the state (AMBIG_EARLIEST, src/libfsm/print/c.c:67-85)
 *   ret:      index of the state's id set in the de-duplicated list of sets, ordered by
 *             count, then memcmp of the id arrays -- the order build_retlist() produces
 *             (src/libfsm/vm/retlist.c:93-138, cmp_ret). */
typedef std::vector<uint32_t> IdSet;
/* the de-duplicated id sets of the end states in build_retlist()'s order (src/libfsm/vm/retlist.c:93-138), from the plan alone */
static void build_ret_sets(const Plan &p, std::vector<uint32_t> &ret_off, std::vector<uint32_t> &ret_ids, std::map<IdSet, uint32_t> &index)
{
	typedef IdSet Set;
	std::vector<Set> sets;
	/* end states = those appearing in fin */
	std::vector<uint8_t> is_end(p.nstates, 0);
	for (uint32_t v : p.fin) if (v != FSM_HIP_NO_MATCH) is_end[v] = 1;
	for (uint32_t s = 0; s < p.nstates; s++)
		if (is_end[s]) sets.emplace_back(p.endids.begin() + p.endid_off[s], p.endids.begin() + p.endid_off[s + 1]);
	/* cmp_ret (src/libfsm/vm/retlist.c:63-79): by count, then memcmp over the raw id array -- byte-wise,
	 * i.e. NOT numeric for ids >= 256 on a little-endian host ({256} sorts before {1}) */
	std::sort(sets.begin(), sets.end(), [](const Set &a, const Set &b) {
		if (a.size() != b.size()) return a.size() < b.size();
		return !a.empty() && memcmp(a.data(), b.data(), a.size() * sizeof(uint32_t)) < 0;
	});
	sets.erase(std::unique(sets.begin(), sets.end()), sets.end());
	ret_off.assign(1, 0);
	ret_ids.clear();
	index.clear();
	for (uint32_t k = 0; k < sets.size(); k++) {
		index[sets[k]] = k;
		ret_ids.insert(ret_ids.end(), sets[k].begin(), sets[k].end());
		ret_off.push_back((uint32_t)ret_ids.size());
	}
}

/* for multi.hip (dfa_access.h): what fsm_hip_exec_batch_ids writes for an input that ends in renumbered state n (Plan::dense's
 * numbering), mode EARLIEST or RET, from the plan alone -- a dfa created with FSM_HIP_DEFER_UPLOAD stays without tables of its
 * own.  *conflict: the lowest caller's end state with more than one id, or NO_MATCH (what FSM_HIP_IDS_ERROR refuses). */
namespace fsmhip {
int dfa_ids_by_state(const fsm_hip_dfa *d, int mode, std::vector<uint32_t> &out, uint32_t *conflict)
{
	const Plan &p = d->plan;
	std::vector<uint32_t> ro, ri;
	std::map<IdSet, uint32_t> index;
	if (mode == FSM_HIP_IDS_RET) build_ret_sets(p, ro, ri, index);
	out.assign(p.S1, FSM_HIP_NO_MATCH);
	uint32_t cf = FSM_HIP_NO_MATCH;
	for (uint32_t n = 0; n < p.S1; n++) {
		const uint32_t s = p.fin[n];
		if (s == FSM_HIP_NO_MATCH) continue;
		const uint32_t a = p.endid_off[s], b = p.endid_off[s + 1];
		if (b - a > 1 && s < cf) cf = s;
		out[n] = mode == FSM_HIP_IDS_RET ? index[IdSet(p.endids.begin() + a, p.endids.begin() + b)] : (b > a ? p.endids[a] : FSM_HIP_NO_ID);
	}
	if (conflict) *conflict = cf;
	return 0;
}
}

static int ensure_ids(fsm_hip_dfa *d)
{
	if (ensure_uploaded(d) != 0) return -1;
	DfaLock lk(d->mu);
	if (d->ids_ready) return 0;
	const Plan &p = d->plan;
	typedef IdSet Set;
	std::map<Set, uint32_t> index;
	build_ret_sets(p, d->ret_off, d->ret_ids, index);
	std::vector<uint32_t> fe(d->fin_host.size(), FSM_HIP_NO_MATCH), fr(d->fin_host.size(), FSM_HIP_NO_MATCH);
	for (size_t i = 0; i < d->fin_host.size(); i++) {
		const uint32_t s = d->fin_host[i];
		if (s == FSM_HIP_NO_MATCH) continue;
		const uint32_t a = p.endid_off[s], b = p.endid_off[s + 1];
		fe[i] = b > a ? p.endids[a] : FSM_HIP_NO_ID;
		if (b - a > 1 && s < d->ids_conflict) d->ids_conflict = s;
		fr[i] = index[Set(p.endids.begin() + a, p.endids.begin() + b)];
	}
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	HIP_TRY(upload(&d->d_fin_earliest, fe));
	HIP_TRY(upload(&d->d_fin_ret, fr));
	d->ids_ready = true;
	return 0;
fail:
	return -1;
}

static int ids_device(fsm_hip_dfa *d, const void *d_base, size_t stride, const uint32_t *d_len, const uint64_t *d_off, size_t n,
	int mode, uint32_t *d_id_out, void *hip_stream, const BatchHint &hint)
{
	if (d == nullptr || d_id_out == nullptr || (mode != FSM_HIP_IDS_EARLIEST && mode != FSM_HIP_IDS_RET && mode != FSM_HIP_IDS_ERROR) ||
	    (n != 0 && d_off == nullptr && d_base == nullptr && stride != 0)) { errno = EINVAL; return -1; }
	{
		const fsm_hip_dfa *t = route(d, d_off == nullptr && d_len == nullptr && stride != 0 && stride % 16u == 0 && (reinterpret_cast<uintptr_t>(d_base) % 16u) == 0);
		if (t != d) return ids_device(const_cast<fsm_hip_dfa *>(t), d_base, stride, d_len, d_off, n, mode, d_id_out, hip_stream, hint);
	}
	if (ensure_ids(d) != 0) return -1;
	if (mode == FSM_HIP_IDS_ERROR) {
		/* AMBIG_ERROR: an end state with more than one id is refused (print/c.c:67-72 fails the
		 * print with EINVAL); without such a state it is AMBIG_EARLIEST */
		if (d->ids_conflict != FSM_HIP_NO_MATCH) { errno = EINVAL; return -1; }
		mode = FSM_HIP_IDS_EARLIEST;
	}
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	if (ensure_uploaded(d) != 0) return -1;
	WalkArgs a = d->proto;
	a.base = static_cast<const uint8_t *>(d_base);
	a.stride = d_off ? 0 : stride;
	a.len = d_off ? nullptr : d_len;
	a.off = d_off;
	a.n = n;
	a.fin2 = mode == FSM_HIP_IDS_EARLIEST ? d->d_fin_earliest : d->d_fin_ret;
	a.out2 = d_id_out;
	const bool fast = d_off == nullptr && d_len == nullptr && stride != 0 && stride % 16u == 0 &&
		(reinterpret_cast<uintptr_t>(d_base) % 16u) == 0 && d->knob_input_mode != IN_GENERIC;
	return launch_walk(d, a, fast, static_cast<hipStream_t>(hip_stream), hint);
}

extern "C" int fsm_hip_exec_batch_ids_device(const struct fsm_hip_dfa *dc,
	const void *d_base, size_t stride, const uint32_t *d_len, size_t n,
	int mode, uint32_t *d_id_out, void *hip_stream)
{
	return ids_device(const_cast<fsm_hip_dfa *>(dc), d_base, stride, d_len, nullptr, n, mode, d_id_out, hip_stream, BatchHint());
}

extern "C" int fsm_hip_exec_batch_ids_offsets_device(const struct fsm_hip_dfa *dc,
	const void *d_base, const uint64_t *d_off, size_t n,
	int mode, uint32_t *d_id_out, void *hip_stream)
{
	if (n != 0 && d_off == nullptr) { errno = EINVAL; return -1; }
	return ids_device(const_cast<fsm_hip_dfa *>(dc), d_base, 0, nullptr, d_off, n, mode, d_id_out, hip_stream, BatchHint());
}

/* what the host fronts know about their batch (launch_walk picks the kernel from it) */
static BatchHint host_hint(size_t in_bytes, const uint32_t *len, const uint64_t *off, size_t n, int pick_mean)
{
	BatchHint hint;
	hint.bytes = in_bytes;
	if (n == 0) return hint;
	if (len != nullptr) {   /* the average of the lengths, not of the rows they sit in */
		uint64_t sum = 0;
		for (size_t i = 0; i < n; i++) sum += len[i];
		hint.short_mean = sum / n < (uint64_t)pick_mean;
	} else if (off != nullptr) {
		hint.short_mean = in_bytes / n < (size_t)pick_mean;
	}
	return hint;
}

static int check_host_batch(const unsigned char *base, size_t stride, const uint32_t *len, const uint64_t *off, size_t n, size_t *in_bytes)
{
	if (off != nullptr) {
		for (size_t i = 0; i < n; i++)
			if (off[i + 1] < off[i]) return -1;
		*in_bytes = n ? (size_t)off[n] : 0;
		if (*in_bytes != 0 && base == nullptr) return -1;
		return 0;
	}
	if (n != 0 && base == nullptr && stride != 0) return -1;
	if (len != nullptr)
		for (size_t i = 0; i < n; i++)
			if (len[i] > stride) return -1;
	*in_bytes = n * stride;
	return 0;
}

static int ids_host(const struct fsm_hip_dfa *d, const unsigned char *base, size_t stride, const uint32_t *len, const uint64_t *off, size_t n,
	int mode, uint32_t *id_out)
{
	size_t in_bytes = 0;
	if (d == nullptr || id_out == nullptr || check_host_batch(base, stride, len, off, n, &in_bytes) != 0) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	HostCall hc(d);
	const int p_in = hc.add(HostCall::IN, base, nullptr, in_bytes, 32);
	const int p_len = len ? hc.add(HostCall::IN, len, nullptr, n * sizeof(uint32_t)) : -1;
	const int p_off = off ? hc.add(HostCall::IN, off, nullptr, (n + 1) * sizeof(uint64_t)) : -1;
	const int p_out = hc.add(HostCall::OUT, nullptr, id_out, n * sizeof(uint32_t));
	if (hc.begin() != 0) return -1;
	if (ids_device(hc.d, hc.dev<unsigned char>(p_in), stride, hc.dev<uint32_t>(p_len), hc.dev<uint64_t>(p_off), n, mode,
	               hc.dev<uint32_t>(p_out), hc.d->hs, host_hint(in_bytes, len, off, n, pick_mean_of(d, false))) != 0) return -1;
	return hc.end();
}

extern "C" int fsm_hip_exec_batch_ids(const struct fsm_hip_dfa *d,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	int mode, uint32_t *id_out)
{
	return ids_host(d, base, stride, len, nullptr, n, mode, id_out);
}

extern "C" int fsm_hip_exec_batch_ids_offsets(const struct fsm_hip_dfa *d,
	const unsigned char *base, const uint64_t *off, size_t n,
	int mode, uint32_t *id_out)
{
	if (n != 0 && off == nullptr) { errno = EINVAL; return -1; }
	return ids_host(d, base, 0, nullptr, off, n, mode, id_out);
}

extern "C" int fsm_hip_ids_conflict(const struct fsm_hip_dfa *dc, fsm_state_t *state)
{
	fsm_hip_dfa *d = const_cast<fsm_hip_dfa *>(dc);
	if (d == nullptr) { errno = EINVAL; return -1; }
	if (ensure_ids(d) != 0) return -1;
	if (d->ids_conflict == FSM_HIP_NO_MATCH) return 0;
	if (state != nullptr) *state = d->ids_conflict;
	return 1;
}

extern "C" size_t fsm_hip_ret_count(const struct fsm_hip_dfa *dc)
{
	fsm_hip_dfa *d = const_cast<fsm_hip_dfa *>(dc);
	if (d == nullptr || ensure_ids(d) != 0) return 0;
	return d->ret_off.size() - 1;
}

extern "C" int fsm_hip_ret_get(const struct fsm_hip_dfa *dc, uint32_t ret_index, const uint32_t **ids, size_t *count)
{
	fsm_hip_dfa *d = const_cast<fsm_hip_dfa *>(dc);
	if (d == nullptr || ids == nullptr || count == nullptr || ensure_ids(d) != 0 || ret_index + 1 >= d->ret_off.size()) {
		errno = EINVAL;
		return -1;
	}
	*ids = d->ret_ids.data() + d->ret_off[ret_index];
	*count = d->ret_off[ret_index + 1] - d->ret_off[ret_index];
	return 0;
}

/* ------------------------------------------------------------------ */
/* resume: start every input from a given state, return the state reached */
/* ------------------------------------------------------------------ */

static int ensure_resume(fsm_hip_dfa *d)
{
	if (ensure_uploaded(d) != 0) return -1;
	DfaLock lk(d->mu);
	if (d->resume_ready) return 0;
	const Plan &p = d->plan;
	/* caller id (+ nstates = DEAD) -> encoded state */
	std::vector<uint32_t> enc(p.nstates + 1);
	for (uint32_t o = 0; o <= p.nstates; o++) enc[o] = d->enc_host[p.old2new[o]];
	/* encoded index (as used for fin) -> caller id, DEAD marker for the synthetic state */
	std::vector<uint32_t> orig(d->fin_host.size(), FSMHIP_STATE_DEAD);
	for (uint32_t n2 = 0; n2 + 1 < p.S1; n2++) orig[d->enc_host[n2] / d->proto.fin_div] = p.new2old[n2];
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	HIP_TRY(upload(&d->d_enc_of, enc));
	HIP_TRY(upload(&d->d_orig_of, orig));
	d->resume_ready = true;
	return 0;
fail:
	return -1;
}

extern "C" int fsm_hip_state_is_absorbing(const struct fsm_hip_dfa *d, uint32_t state)
{
	if (d == nullptr) { errno = EINVAL; return -1; }
	if (state == FSMHIP_STATE_DEAD) return 1;
	if (state == FSMHIP_STATE_START) state = d->plan.new2old[d->plan.start];
	if (state >= d->plan.nstates) { errno = EINVAL; return -1; }
	return d->plan.old2new[state] >= d->plan.abs_min ? 1 : 0;
}

static int resume_device(fsm_hip_dfa *d, const void *d_base, size_t stride, const uint32_t *d_len, const uint64_t *d_off, size_t n,
	uint32_t *d_state_io, uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream, const BatchHint &hint,
	const uint32_t *d_off32 = nullptr, bool lens_only = false)
{
	if (d == nullptr || d_state_io == nullptr || (n != 0 && d_off == nullptr && d_off32 == nullptr && !lens_only && d_base == nullptr && stride != 0) ||
	    (lens_only && n != 0 && d_len == nullptr)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	{
		const fsm_hip_dfa *t = route(d, false);     /* (a resumed walk never takes the fixed-stride kernels of the pair table) */
		if (t != d) return resume_device(const_cast<fsm_hip_dfa *>(t), d_base, stride, d_len, d_off, n, d_state_io, d_end_out, d_accept_bitmap, hip_stream, hint, d_off32, lens_only);
	}
	if (ensure_resume(d) != 0) return -1;
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	if (ensure_uploaded(d) != 0) return -1;
	WalkArgs a = d->proto;
	const bool packed = d_off != nullptr || d_off32 != nullptr || lens_only;
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	a.base = static_cast<const uint8_t *>(d_base);
	a.stride = packed ? 0 : stride;
	a.len = d_off != nullptr || d_off32 != nullptr ? nullptr : d_len;
	a.off = d_off;
	a.off32 = d_off == nullptr ? d_off32 : nullptr;
	a.n = n;
	a.end_out = d_end_out;
	a.bitmap = d_accept_bitmap;
	a.state_io = d_state_io;
	a.enc_of = d->d_enc_of;
	a.orig_of = d->d_orig_of;
	a.nstates = d->plan.nstates;
	const bool lo = lens_only && d_off == nullptr && d_off32 == nullptr;
	DfaLock lk(d->mu);   /* pre-pass, walk and the block's event in one critical section (see exec_packed_device) */
	if (lo && tile_bases(d, d_len, n, s, &a.tbase) != 0) return -1;
	const bool fast = !packed && d_len == nullptr && stride != 0 && stride % 16u == 0 &&
		(reinterpret_cast<uintptr_t>(d_base) % 16u) == 0 && d->knob_input_mode != IN_GENERIC;
	const int r = launch_walk(d, a, fast, s, hint);
	if (lo) tile_bases_done(d, s);
	return r;
}

/* resume over packed inputs whose metadata is u64 offsets, u32 offsets or lengths alone: the carry of fsm_vm_match_file
 * (src/libfsm/vm.c:188-216: the state survives from one buffer to the next) for batches in the compact forms */
extern "C" int fsm_hip_exec_batch_resume_packed_device(const struct fsm_hip_dfa *dc,
	const void *d_base, int meta_form, const void *d_meta, size_t n,
	uint32_t *d_state_io, uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream)
{
	fsm_hip_dfa *d = const_cast<fsm_hip_dfa *>(dc);
	if (n != 0 && d_meta == nullptr) { errno = EINVAL; return -1; }
	switch (meta_form) {
	case FSM_HIP_META_OFF64:
		return resume_device(d, d_base, 0, nullptr, static_cast<const uint64_t *>(d_meta), n, d_state_io, d_end_out, d_accept_bitmap, hip_stream, BatchHint());
	case FSM_HIP_META_OFF32:
		return resume_device(d, d_base, 0, nullptr, nullptr, n, d_state_io, d_end_out, d_accept_bitmap, hip_stream, BatchHint(), static_cast<const uint32_t *>(d_meta), false);
	case FSM_HIP_META_LENGTHS:
		return resume_device(d, d_base, 0, static_cast<const uint32_t *>(d_meta), nullptr, n, d_state_io, d_end_out, d_accept_bitmap, hip_stream, BatchHint(), nullptr, true);
	default:
		errno = EINVAL;
		return -1;
	}
}

extern "C" int fsm_hip_exec_batch_resume_packed(const struct fsm_hip_dfa *d,
	const unsigned char *base, int meta_form, const void *meta, size_t n,
	uint32_t *state_io, uint32_t *end_out)
{
	if (d == nullptr || state_io == nullptr || (n != 0 && meta == nullptr) ||
	    (meta_form != FSM_HIP_META_OFF64 && meta_form != FSM_HIP_META_OFF32 && meta_form != FSM_HIP_META_LENGTHS)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	size_t in_bytes = 0, meta_bytes = 0;
	if (meta_form == FSM_HIP_META_OFF64) {
		const uint64_t *o = static_cast<const uint64_t *>(meta);
		for (size_t i = 0; i < n; i++) if (o[i + 1] < o[i]) { errno = EINVAL; return -1; }
		in_bytes = (size_t)o[n]; meta_bytes = (n + 1) * sizeof(uint64_t);
	} else if (meta_form == FSM_HIP_META_OFF32) {
		const uint32_t *o = static_cast<const uint32_t *>(meta);
		for (size_t i = 0; i < n; i++) if (o[i + 1] < o[i]) { errno = EINVAL; return -1; }
		in_bytes = o[n]; meta_bytes = (n + 1) * sizeof(uint32_t);
	} else {
		const uint32_t *l = static_cast<const uint32_t *>(meta);
		uint64_t sum = 0;
		for (size_t i = 0; i < n; i++) sum += l[i];
		in_bytes = (size_t)sum; meta_bytes = n * sizeof(uint32_t);
	}
	if (in_bytes != 0 && base == nullptr) { errno = EINVAL; return -1; }
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	HostCall hc(d);
	const int p_in = hc.add(HostCall::IN, base, nullptr, in_bytes, 32);
	const int p_meta = hc.add(HostCall::IN, meta, nullptr, meta_bytes);
	const int p_st = hc.add(HostCall::INOUT, state_io, state_io, n * sizeof(uint32_t));
	const int p_end = hc.add(HostCall::OUT, nullptr, end_out, n * sizeof(uint32_t));
	if (hc.begin() != 0) return -1;
	BatchHint hint;
	hint.bytes = in_bytes;
	hint.short_mean = in_bytes / n < (size_t)pick_mean_of(d, false);
	const void *dm = hc.dev<unsigned char>(p_meta);
	if (resume_device(hc.d, hc.dev<unsigned char>(p_in), 0, meta_form == FSM_HIP_META_LENGTHS ? static_cast<const uint32_t *>(dm) : nullptr,
	                  meta_form == FSM_HIP_META_OFF64 ? static_cast<const uint64_t *>(dm) : nullptr, n,
	                  hc.dev<uint32_t>(p_st), hc.dev<uint32_t>(p_end), nullptr, hc.d->hs, hint,
	                  meta_form == FSM_HIP_META_OFF32 ? static_cast<const uint32_t *>(dm) : nullptr, meta_form == FSM_HIP_META_LENGTHS) != 0) return -1;
	return hc.end();
}

extern "C" int fsm_hip_exec_batch_resume_device(const struct fsm_hip_dfa *dc,
	const void *d_base, size_t stride, const uint32_t *d_len, size_t n,
	uint32_t *d_state_io, uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream)
{
	return resume_device(const_cast<fsm_hip_dfa *>(dc), d_base, stride, d_len, nullptr, n, d_state_io, d_end_out, d_accept_bitmap, hip_stream, BatchHint());
}

extern "C" int fsm_hip_exec_batch_resume_offsets_device(const struct fsm_hip_dfa *dc,
	const void *d_base, const uint64_t *d_off, size_t n,
	uint32_t *d_state_io, uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream)
{
	if (n != 0 && d_off == nullptr) { errno = EINVAL; return -1; }
	return resume_device(const_cast<fsm_hip_dfa *>(dc), d_base, 0, nullptr, d_off, n, d_state_io, d_end_out, d_accept_bitmap, hip_stream, BatchHint());
}

static int resume_host(const struct fsm_hip_dfa *d, const unsigned char *base, size_t stride, const uint32_t *len, const uint64_t *off, size_t n,
	uint32_t *state_io, uint32_t *end_out)
{
	size_t in_bytes = 0;
	if (d == nullptr || state_io == nullptr || check_host_batch(base, stride, len, off, n, &in_bytes) != 0) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	HostCall hc(d);
	const int p_in = hc.add(HostCall::IN, base, nullptr, in_bytes, 32);
	const int p_len = len ? hc.add(HostCall::IN, len, nullptr, n * sizeof(uint32_t)) : -1;
	const int p_off = off ? hc.add(HostCall::IN, off, nullptr, (n + 1) * sizeof(uint64_t)) : -1;
	const int p_st = hc.add(HostCall::INOUT, state_io, state_io, n * sizeof(uint32_t));
	const int p_end = hc.add(HostCall::OUT, nullptr, end_out, n * sizeof(uint32_t));
	if (hc.begin() != 0) return -1;
	if (resume_device(hc.d, hc.dev<unsigned char>(p_in), stride, hc.dev<uint32_t>(p_len), hc.dev<uint64_t>(p_off), n,
	                  hc.dev<uint32_t>(p_st), hc.dev<uint32_t>(p_end), nullptr, hc.d->hs, host_hint(in_bytes, len, off, n, pick_mean_of(d, false))) != 0) return -1;
	return hc.end();
}

extern "C" int fsm_hip_exec_batch_resume(const struct fsm_hip_dfa *d,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	uint32_t *state_io, uint32_t *end_out)
{
	return resume_host(d, base, stride, len, nullptr, n, state_io, end_out);
}

extern "C" int fsm_hip_exec_batch_resume_offsets(const struct fsm_hip_dfa *d,
	const unsigned char *base, const uint64_t *off, size_t n,
	uint32_t *state_io, uint32_t *end_out)
{
	if (n != 0 && off == nullptr) { errno = EINVAL; return -1; }
	return resume_host(d, base, 0, nullptr, off, n, state_io, end_out);
}

/* ------------------------------------------------------------------ */
/* eager outputs                                                      */
/* ------------------------------------------------------------------ */

extern "C" size_t fsm_hip_eager_id_count(const struct fsm_hip_dfa *d)
{
	return d == nullptr ? 0 : d->plan.eager_ids.size();
}

extern "C" size_t fsm_hip_eager_words(const struct fsm_hip_dfa *d)
{
	return d == nullptr || d->plan.eager_words == 0 ? 1 : d->plan.eager_words;
}

extern "C" uint32_t fsm_hip_eager_id(const struct fsm_hip_dfa *d, unsigned bit)
{
	if (d == nullptr || bit >= d->plan.eager_ids.size()) return FSM_HIP_NO_MATCH;
	return d->plan.eager_ids[bit];
}

static int eager_device(const struct fsm_hip_dfa *d, const void *d_base, size_t stride, const uint32_t *d_len, const uint64_t *d_off, size_t n,
	uint32_t *d_end_out, uint64_t *d_eager_out, void *hip_stream, const BatchHint &hint)
{
	if (d == nullptr || d_eager_out == nullptr || (n != 0 && d_off == nullptr && d_base == nullptr && stride != 0)) { errno = EINVAL; return -1; }
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	if (d->plan.emask.empty()) {
		/* no state emits anything: the answer is all zeros, the walk is the plain one */
		hipError_t e = zero_async(d_eager_out, n * sizeof(uint64_t), static_cast<hipStream_t>(hip_stream));
		if (e != hipSuccess) { errno = hip_errno(e); return -1; }
		if (d_off != nullptr) return exec_offsets_device(d, d_base, d_off, n, d_end_out, nullptr, hip_stream, hint);
		return exec_stride_device(d, d_base, stride, d_len, n, d_end_out, nullptr, hip_stream, hint);
	}
	if (d->plan.eager_words > 1) {
		/* wide sets are OR-ed in place by the kernel: start from zero */
		hipError_t e = zero_async(d_eager_out, n * d->plan.eager_words * sizeof(uint64_t), static_cast<hipStream_t>(hip_stream));
		if (e != hipSuccess) { errno = hip_errno(e); return -1; }
	}
	if (ensure_uploaded(d) != 0) return -1;
	WalkArgs a = d->proto;
	a.base = static_cast<const uint8_t *>(d_base);
	a.stride = d_off ? 0 : stride;
	a.len = d_off ? nullptr : d_len;
	a.off = d_off;
	a.n = n;
	a.end_out = d_end_out;
	a.eager_out = d_eager_out;
	const bool fast = d_off == nullptr && d_len == nullptr && stride != 0 && stride % 16u == 0 &&
		(reinterpret_cast<uintptr_t>(d_base) % 16u) == 0 && d->knob_input_mode != IN_GENERIC;
	return launch_walk(d, a, fast, static_cast<hipStream_t>(hip_stream), hint);
}

extern "C" int fsm_hip_exec_batch_eager_device(const struct fsm_hip_dfa *d,
	const void *d_base, size_t stride, const uint32_t *d_len, size_t n,
	uint32_t *d_end_out, uint64_t *d_eager_out, void *hip_stream)
{
	return eager_device(d, d_base, stride, d_len, nullptr, n, d_end_out, d_eager_out, hip_stream, BatchHint());
}

extern "C" int fsm_hip_exec_batch_eager_offsets_device(const struct fsm_hip_dfa *d,
	const void *d_base, const uint64_t *d_off, size_t n,
	uint32_t *d_end_out, uint64_t *d_eager_out, void *hip_stream)
{
	if (n != 0 && d_off == nullptr) { errno = EINVAL; return -1; }
	return eager_device(d, d_base, 0, nullptr, d_off, n, d_end_out, d_eager_out, hip_stream, BatchHint());
}

/* Every output of one batch from ONE walk: end states and / or the accept bitmap, device-side end-ids (ids_mode, d_id_out) and
 * eager sets (d_eager_out), whichever are asked for -- the walk kernels write all of them in one pass (the multi-device front used
 * to launch one walk per output) */
static int all_device(const struct fsm_hip_dfa *dc,
	const void *d_base, size_t stride, const uint32_t *d_len, const uint64_t *d_off, const uint32_t *d_off32, bool lens_only, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, int ids_mode, uint32_t *d_id_out, uint64_t *d_eager_out, void *hip_stream, const BatchHint &hint)
{
	fsm_hip_dfa *d = const_cast<fsm_hip_dfa *>(dc);
	const bool packed = d_off != nullptr || d_off32 != nullptr || lens_only;
	if (d == nullptr || (n != 0 && !packed && d_base == nullptr && stride != 0) || (d_off != nullptr && d_len != nullptr) ||
	    (lens_only && n != 0 && d_len == nullptr)) { errno = EINVAL; return -1; }
	{
		const fsm_hip_dfa *t = route(d, !packed && d_len == nullptr && stride != 0 && stride % 16u == 0 && (reinterpret_cast<uintptr_t>(d_base) % 16u) == 0);
		if (t != d) return all_device(t, d_base, stride, d_len, d_off, d_off32, lens_only, n, d_end_out, d_accept_bitmap, ids_mode, d_id_out, d_eager_out, hip_stream, hint);
	}
	if (d_id_out != nullptr) {
		if (ids_mode != FSM_HIP_IDS_EARLIEST && ids_mode != FSM_HIP_IDS_RET && ids_mode != FSM_HIP_IDS_ERROR) { errno = EINVAL; return -1; }
		if (ensure_ids(d) != 0) return -1;
		if (ids_mode == FSM_HIP_IDS_ERROR) {
			if (d->ids_conflict != FSM_HIP_NO_MATCH) { errno = EINVAL; return -1; }
			ids_mode = FSM_HIP_IDS_EARLIEST;
		}
	}
	if (n == 0) return 0;
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	if (ensure_uploaded(d) != 0) return -1;
	WalkArgs a = d->proto;
	a.base = static_cast<const uint8_t *>(d_base);
	a.stride = packed ? 0 : stride;
	a.len = d_off != nullptr || d_off32 != nullptr ? nullptr : d_len;
	a.off = d_off;
	a.off32 = d_off == nullptr ? d_off32 : nullptr;
	a.n = n;
	a.end_out = d_end_out;
	a.bitmap = d_accept_bitmap;
	if (d_id_out != nullptr) {
		a.fin2 = ids_mode == FSM_HIP_IDS_EARLIEST ? d->d_fin_earliest : d->d_fin_ret;
		a.out2 = d_id_out;
	}
	if (d_eager_out != nullptr) {
		if (d->plan.emask.empty() || d->plan.eager_words > 1) {
			/* no state emits anything: all zeros; wide sets are OR-ed in place: start from zero */
			const size_t w = d->plan.emask.empty() ? 1 : d->plan.eager_words;
			hipError_t e = zero_async(d_eager_out, n * w * sizeof(uint64_t), s);
			if (e != hipSuccess) { errno = hip_errno(e); return -1; }
		}
		if (!d->plan.emask.empty()) a.eager_out = d_eager_out;
	}
	const bool lo = lens_only && d_off == nullptr && d_off32 == nullptr;
	DfaLock lk(d->mu);   /* pre-pass, walk and the block's event in one critical section (see exec_packed_device) */
	if (lo && tile_bases(d, d_len, n, s, &a.tbase) != 0) return -1;
	const bool fast = !packed && d_len == nullptr && stride != 0 && stride % 16u == 0 &&
		(reinterpret_cast<uintptr_t>(d_base) % 16u) == 0 && d->knob_input_mode != IN_GENERIC;
	const int r = launch_walk(d, a, fast, s, hint);
	if (lo) tile_bases_done(d, s);
	return r;
}

extern "C" int fsm_hip_exec_batch_all_device(const struct fsm_hip_dfa *d,
	const void *d_base, size_t stride, const uint32_t *d_len, const uint64_t *d_off, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, int ids_mode, uint32_t *d_id_out, uint64_t *d_eager_out, void *hip_stream)
{
	return all_device(d, d_base, stride, d_len, d_off, nullptr, false, n, d_end_out, d_accept_bitmap, ids_mode, d_id_out, d_eager_out, hip_stream, BatchHint());
}

/* the same over packed inputs whose metadata is u64 offsets, u32 offsets or lengths alone */
extern "C" int fsm_hip_exec_batch_packed_all_device(const struct fsm_hip_dfa *d,
	const void *d_base, int meta_form, const void *d_meta, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, int ids_mode, uint32_t *d_id_out, uint64_t *d_eager_out, void *hip_stream)
{
	if (n != 0 && d_meta == nullptr) { errno = EINVAL; return -1; }
	switch (meta_form) {
	case FSM_HIP_META_OFF64:
		return all_device(d, d_base, 0, nullptr, static_cast<const uint64_t *>(d_meta), nullptr, false, n, d_end_out, d_accept_bitmap, ids_mode, d_id_out, d_eager_out, hip_stream, BatchHint());
	case FSM_HIP_META_OFF32:
		return all_device(d, d_base, 0, nullptr, nullptr, static_cast<const uint32_t *>(d_meta), false, n, d_end_out, d_accept_bitmap, ids_mode, d_id_out, d_eager_out, hip_stream, BatchHint());
	case FSM_HIP_META_LENGTHS:
		return all_device(d, d_base, 0, static_cast<const uint32_t *>(d_meta), nullptr, nullptr, true, n, d_end_out, d_accept_bitmap, ids_mode, d_id_out, d_eager_out, hip_stream, BatchHint());
	default:
		errno = EINVAL;
		return -1;
	}
}

/* host pointers: the metadata is checked (non-decreasing offsets, the lengths' sum is the batch) */
extern "C" int fsm_hip_exec_batch_packed_all(const struct fsm_hip_dfa *d,
	const unsigned char *base, int meta_form, const void *meta, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap, int ids_mode, uint32_t *id_out, uint64_t *eager_out)
{
	if (d == nullptr || (n != 0 && meta == nullptr) || (meta_form != FSM_HIP_META_OFF64 && meta_form != FSM_HIP_META_OFF32 && meta_form != FSM_HIP_META_LENGTHS)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	size_t in_bytes = 0, meta_bytes = 0;
	uint64_t sum = 0;
	if (meta_form == FSM_HIP_META_OFF64) {
		const uint64_t *o = static_cast<const uint64_t *>(meta);
		for (size_t i = 0; i < n; i++) if (o[i + 1] < o[i]) { errno = EINVAL; return -1; }
		in_bytes = (size_t)o[n]; meta_bytes = (n + 1) * sizeof(uint64_t);
	} else if (meta_form == FSM_HIP_META_OFF32) {
		const uint32_t *o = static_cast<const uint32_t *>(meta);
		for (size_t i = 0; i < n; i++) if (o[i + 1] < o[i]) { errno = EINVAL; return -1; }
		in_bytes = o[n]; meta_bytes = (n + 1) * sizeof(uint32_t);
	} else {
		const uint32_t *l = static_cast<const uint32_t *>(meta);
		for (size_t i = 0; i < n; i++) sum += l[i];
		in_bytes = (size_t)sum; meta_bytes = n * sizeof(uint32_t);
	}
	if (in_bytes != 0 && base == nullptr) { errno = EINVAL; return -1; }
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	HostCall hc(d);
	const int p_in = hc.add(HostCall::IN, base, nullptr, in_bytes, 32);
	const int p_meta = hc.add(HostCall::IN, meta, nullptr, meta_bytes);
	const int p_end = hc.add(HostCall::OUT, nullptr, end_out, n * sizeof(uint32_t));
	const int p_bm = hc.add(HostCall::OUT, nullptr, accept_bitmap, ((n + 63) / 64) * sizeof(uint64_t));
	const int p_id = hc.add(HostCall::OUT, nullptr, id_out, n * sizeof(uint32_t));
	const int p_eo = hc.add(HostCall::OUT, nullptr, eager_out, n * fsm_hip_eager_words(d) * sizeof(uint64_t));
	if (hc.begin() != 0) return -1;
	BatchHint hint;
	hint.bytes = in_bytes;
	hint.short_mean = in_bytes / n < (size_t)pick_mean_of(d, false);
	const void *dm = hc.dev<unsigned char>(p_meta);
	if (all_device(d, hc.dev<unsigned char>(p_in), 0, meta_form == FSM_HIP_META_LENGTHS ? static_cast<const uint32_t *>(dm) : nullptr,
	               meta_form == FSM_HIP_META_OFF64 ? static_cast<const uint64_t *>(dm) : nullptr,
	               meta_form == FSM_HIP_META_OFF32 ? static_cast<const uint32_t *>(dm) : nullptr, meta_form == FSM_HIP_META_LENGTHS, n,
	               hc.dev<uint32_t>(p_end), hc.dev<uint64_t>(p_bm), ids_mode, hc.dev<uint32_t>(p_id), hc.dev<uint64_t>(p_eo), hc.d->hs, hint) != 0) return -1;
	return hc.end();
}

static int eager_host(const struct fsm_hip_dfa *d, const unsigned char *base, size_t stride, const uint32_t *len, const uint64_t *off, size_t n,
	uint32_t *end_out, uint64_t *eager_out)
{
	size_t in_bytes = 0;
	if (d == nullptr || eager_out == nullptr || check_host_batch(base, stride, len, off, n, &in_bytes) != 0) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	HostCall hc(d);
	const int p_in = hc.add(HostCall::IN, base, nullptr, in_bytes, 32);
	const int p_len = len ? hc.add(HostCall::IN, len, nullptr, n * sizeof(uint32_t)) : -1;
	const int p_off = off ? hc.add(HostCall::IN, off, nullptr, (n + 1) * sizeof(uint64_t)) : -1;
	const int p_end = hc.add(HostCall::OUT, nullptr, end_out, n * sizeof(uint32_t));
	const int p_eo = hc.add(HostCall::OUT, nullptr, eager_out, n * fsm_hip_eager_words(d) * sizeof(uint64_t));
	if (hc.begin() != 0) return -1;
	if (eager_device(d, hc.dev<unsigned char>(p_in), stride, hc.dev<uint32_t>(p_len), hc.dev<uint64_t>(p_off), n,
	                 hc.dev<uint32_t>(p_end), hc.dev<uint64_t>(p_eo), hc.d->hs, host_hint(in_bytes, len, off, n, pick_mean_of(d, false))) != 0) return -1;
	return hc.end();
}

extern "C" int fsm_hip_exec_batch_eager(const struct fsm_hip_dfa *d,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *eager_out)
{
	return eager_host(d, base, stride, len, nullptr, n, end_out, eager_out);
}

extern "C" int fsm_hip_exec_batch_eager_offsets(const struct fsm_hip_dfa *d,
	const unsigned char *base, const uint64_t *off, size_t n,
	uint32_t *end_out, uint64_t *eager_out)
{
	if (n != 0 && off == nullptr) { errno = EINVAL; return -1; }
	return eager_host(d, base, 0, nullptr, off, n, end_out, eager_out);
}

/* ---- the emission stream (order and repeats kept): trace_kernel.h ---- */

static int ensure_trace(fsm_hip_dfa *d)
{
	if (ensure_uploaded(d) != 0) return -1;
	DfaLock lk(d->mu);
	if (d->trace_ready) return 0;
	const Plan &p = d->plan;
	if (p.dense.size() != (size_t)p.S1 * p.C) { errno = ENOTSUP; return -1; }
	/* id lists per renumbered state, ascending (fsm_eager_output_get order: eager_output.c:317-329) */
	std::vector<uint32_t> eoff(p.S1 + 1, 0), eids;
	for (uint32_t n = 0; n < p.S1; n++) {
		if (!p.emask.empty() && p.emask[n] != 0) {
			if (p.eager_words <= 1) {
				for (unsigned b = 0; b < 64; b++) if (p.emask[n] >> b & 1u) eids.push_back(p.eager_ids[b]);
			} else {
				const size_t first = eids.size();
				for (uint32_t k = p.ew_off[n]; k < p.ew_off[n + 1]; k++)
					for (unsigned b = 0; b < 64; b++) if (p.ew_mask[k] >> b & 1u) eids.push_back(p.eager_ids[(size_t)p.ew_word[k] * 64u + b]);
				std::sort(eids.begin() + (ptrdiff_t)first, eids.end());
			}
		}
		eoff[n + 1] = (uint32_t)eids.size();
	}
	std::vector<uint32_t> cls4(64, 0);
	for (unsigned b = 0; b < 256; b++) cls4[b >> 2] |= (uint32_t)p.cls[b] << ((b & 3u) * 8u);
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	HIP_TRY(upload(&d->d_tr_dense, p.dense));
	HIP_TRY(upload(&d->d_tr_cls4, cls4));
	HIP_TRY(upload(&d->d_tr_eoff, eoff));
	HIP_TRY(upload(&d->d_tr_eids, eids));
	HIP_TRY(upload(&d->d_tr_fin, p.fin));
	d->trace_ready = true;
	return 0;
fail:
	return -1;
}

extern "C" int fsm_hip_exec_batch_eager_trace_device(const struct fsm_hip_dfa *cd,
	const void *d_base, size_t stride, const uint32_t *d_len, const uint64_t *d_off, size_t n, size_t cap,
	uint32_t *d_end_out, uint32_t *d_count_out, uint32_t *d_ids_out, uint32_t *d_pos_out, void *hip_stream)
{
	fsm_hip_dfa *d = const_cast<fsm_hip_dfa *>(cd);
	if (d == nullptr || d_count_out == nullptr || (cap != 0 && d_ids_out == nullptr) || cap > 0xFFFFFFFFu ||
	    (n != 0 && d_off == nullptr && d_base == nullptr && stride != 0)) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	if (ensure_trace(d) != 0) return -1;
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	TraceArgs a;
	a.base = static_cast<const uint8_t *>(d_base);
	a.stride = d_off ? 0 : stride;
	a.len = d_off ? nullptr : d_len;
	a.off = d_off;
	a.n = n;
	a.dense = d->d_tr_dense; a.cls4 = d->d_tr_cls4; a.eoff = d->d_tr_eoff; a.eids = d->d_tr_eids; a.fin = d->d_tr_fin;
	a.C = d->plan.C; a.start = d->plan.start; a.dead = d->plan.S1 - 1u;
	a.cap = (uint32_t)cap;
	a.end_out = d_end_out; a.count_out = d_count_out; a.ids_out = d_ids_out; a.pos_out = d_pos_out;
	const uint64_t blocks = (n + 255u) / 256u;
	if (blocks > 0x7FFFFFFFull) { errno = EINVAL; return -1; }
	hipLaunchKernelGGL(eager_trace_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) { errno = hip_errno(e); return -1; }
	return 0;
}

extern "C" int fsm_hip_exec_batch_eager_trace(const struct fsm_hip_dfa *d,
	const unsigned char *base, size_t stride, const uint32_t *len, const uint64_t *off, size_t n, size_t cap,
	uint32_t *end_out, uint32_t *count_out, uint32_t *ids_out, uint32_t *pos_out)
{
	size_t in_bytes = 0;
	if (d == nullptr || count_out == nullptr || (cap != 0 && ids_out == nullptr) || cap > 0xFFFFFFFFu ||
	    check_host_batch(base, stride, len, off, n, &in_bytes) != 0) { errno = EINVAL; return -1; }
	if (n == 0) return 0;
	if (n > ((size_t)1 << 40) / (cap ? cap : 1)) { errno = EINVAL; return -1; }
	DevGuard dg(d->device);
	if (!dg.ok()) { errno = ENODEV; return -1; }
	HostCall hc(d);
	const int p_in = hc.add(HostCall::IN, base, nullptr, in_bytes, 32);
	const int p_len = len ? hc.add(HostCall::IN, len, nullptr, n * sizeof(uint32_t)) : -1;
	const int p_off = off ? hc.add(HostCall::IN, off, nullptr, (n + 1) * sizeof(uint64_t)) : -1;
	const int p_end = hc.add(HostCall::OUT, nullptr, end_out, n * sizeof(uint32_t));
	const int p_cnt = hc.add(HostCall::OUT, nullptr, count_out, n * sizeof(uint32_t));
	const int p_ids = hc.add(HostCall::OUT, nullptr, ids_out, n * cap * sizeof(uint32_t));
	const int p_pos = hc.add(HostCall::OUT, nullptr, pos_out, n * cap * sizeof(uint32_t));
	if (hc.begin() != 0) return -1;
	if (fsm_hip_exec_batch_eager_trace_device(d, hc.dev<unsigned char>(p_in), stride, hc.dev<uint32_t>(p_len), hc.dev<uint64_t>(p_off), n, cap,
	                                          hc.dev<uint32_t>(p_end), hc.dev<uint32_t>(p_cnt), hc.dev<uint32_t>(p_ids), hc.dev<uint32_t>(p_pos), hc.d->hs) != 0) return -1;
	return hc.end();
}
