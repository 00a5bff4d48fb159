/*
 * fsm_hip_plan.h -- tuning knobs and table-plan inspection for libfsm_hip.so.
 *
 * Not part of the drop-in surface (that is fsm_hip.h).  The plan functions run
 * entirely on the host and exist so that the table builder -- the counterpart
 * of the reference's DFA -> dfa_table expansion, src/libfsm/vm/ir.c:649-750 --
 * can be checked against the oracle on machines without a GPU.  They never
 * execute an input: there is no CPU matching path in this library.
 */
#ifndef FSM_HIP_PLAN_H
#define FSM_HIP_PLAN_H

#include "fsm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
	FSM_HIP_KNOB_INPUT_MODE    = 1,  /* 0 direct per-lane loads, 1 LDS-DMA tiles (both: uniform 16-byte-aligned rows only),
	                                  * 2 generic (per-lane loads, any input), 3 ragged (coalesced + lane refill, any input);
	                                  * -1 auto.  (4 was walk_packed -- a lane owning a byte range across input boundaries:
	                                  * correct, slower than walk_generic everywhere, removed in round 4; the record is
	                                  * profiles/r03s_packed_* and tools/packed_model.py) */
	FSM_HIP_KNOB_NB            = 2,  /* direct mode: 16-byte chunks in flight per lane (4 or 8); the lazy lines kernel (walk_lazy.h): whole chunks per
	                                  * slot and turn, 2 or 4 instead of the default 3 (A/B aid) */
	FSM_HIP_KNOB_ROWS          = 3,  /* the lazy walk (walk_lazy.h): inputs / slots per lane -- 0 auto (3), 2 = round 5's shape (A/B aid; ignored elsewhere) */
	FSM_HIP_KNOB_WAVES         = 4,  /* wavefronts per workgroup (1..16)                              */
	FSM_HIP_KNOB_BLOCKS_PER_CU = 5,  /* persistent grid = CUs * this                                  */
	FSM_HIP_KNOB_EARLY_RETIRE  = 6,  /* -1: from the dfa's flags; else a bit set -- 1: retire a wavefront whose lanes are all absorbing
	                                  * (0 = FSM_HIP_NO_EARLY_RETIRE), 2: absorbing lanes stop loading, 4: no chunk skips, 8: no
	                                  * absorbing-lane masking, 16: walk_generic always asks for four chunks; measurement / test aids
	                                  * of the lines kernels: 32: never walk_lines32 (walk_generic's own body, what batches of 4 GiB
	                                  * and more run), 64: walk_lines32 keeps the first chunk's skip tests, 128: round 4's resource
	                                  * bound (total - 8: loses bytes; for the test that pins the range rule: EINVAL unless FSM_HIP_TEST_KNOBS is set in the environment) */
	FSM_HIP_KNOB_MASK          = 7,  /* retired (accepted, ignored)                                   */
	FSM_HIP_KNOB_HOT_BYTES     = 8,  /* global layout: bytes of the table head mirrored in LDS        */
	FSM_HIP_KNOB_SEG           = 9,  /* LDS-DMA mode: bytes of each row per tile, 64 or 128 (0 auto)  */
	FSM_HIP_KNOB_PREFETCH      = 10, /* direct mode: 0 = no register double-buffer (<= 64 VGPRs)      */
	FSM_HIP_KNOB_NT            = 11, /* LDS-DMA mode, 128-byte segments: nontemporal input loads       */
	FSM_HIP_KNOB_NOSKIP        = 13, /* 1: self-loop layouts never skip a whole chunk (measurement aid: every byte pays its test) */
	FSM_HIP_KNOB_PK_RMIN       = 16, /* retired with walk_packed (accepted, ignored)                 */
	FSM_HIP_KNOB_PK_RMAX       = 17, /* retired (accepted, ignored)                                   */
	FSM_HIP_KNOB_PICK_MEAN     = 18, /* variable-length batches whose mean input length (bytes) is below this go to a per-lane
	                                  * kernel, the others to walk_ragged; -1 (default): 128 where the per-lane kernel is
	                                  * walk_lines32 (plain walks of packed lines), 96 where it is walk_generic (the host fronts
	                                  * know the mean, the device fronts ask a small kernel: walk_aux.h offsets_pick) */
	FSM_HIP_KNOB_PK_DEBUG      = 19, /* retired (accepted, ignored)                                   */
	FSM_HIP_KNOB_SPARSE_FAST   = 20, /* sparse layout, fixed-stride rows: 3 (default where the automaton has a lazy form) states beyond the
	                                  * LDS set are entered without their record, a Bloom filter in LDS says when to fetch it
	                                  * (walk_lazy.h); 1 (default otherwise) the record is the walk state and a byte is three
	                                  * straight-line record probes; 0 the chain loop over state ids (A/B measurement)   */
	FSM_HIP_KNOB_LAZY_DYN      = 21, /* the lazy walk: 1 (default) wavefronts claim their tiles from a device counter, 0 static striding */
	FSM_HIP_KNOB_LAZY_LINES    = 22, /* the lazy walk also serves variable-length / unaligned / resumed batches (walk_lazy_lines): 1 (default), 0 off (A/B) */
	FSM_HIP_KNOB_DMA_BUFS      = 15, /* retired (accepted, ignored): two DMA tiles per wave measured slower than one */
	FSM_HIP_KNOB_RAGGED_ALIGN  = 14  /* retired (accepted, ignored): the ragged kernel fetches from the inputs' own byte addresses */
};

int fsm_hip_dfa_tune(struct fsm_hip_dfa *dfa, int knob, int value);

struct fsm_hip_plan;

enum {
	FSM_HIP_PLAN_SCALARS   = 0,  /* u32[17]: nstates,S1,start,C,abs_min,nabsorbing,layout,row_bytes,comb_abs_min_off,
	                              *          comb256_abs_min_off,comb256_dflt,eager_lo_end,eager_hi_begin,
	                              *          comb_eager_lo_off,comb_eager_hi_off,comb256_eager_lo_off,comb256_eager_hi_off */
	FSM_HIP_PLAN_CLS       = 1,  /* u8[256]  */
	FSM_HIP_PLAN_NEW2OLD   = 2,  /* u32[S1]  */
	FSM_HIP_PLAN_FIN       = 3,  /* u32[S1]  */
	FSM_HIP_PLAN_DENSE     = 4,  /* u32[S1*C] */
	FSM_HIP_PLAN_TINY_COL  = 5,  /* u64[256] */
	FSM_HIP_PLAN_LDS_TAB   = 6,  /* u16[S1*Cpad] */
	FSM_HIP_PLAN_COMB      = 7,  /* u32[] */
	FSM_HIP_PLAN_COMB_DFLT = 8,  /* u32[C] */
	FSM_HIP_PLAN_COMB_OFF  = 9,  /* u32[S1] */
	FSM_HIP_PLAN_COMB_FIN  = 10, /* u32[] */
	FSM_HIP_PLAN_GLOB_TAB  = 11, /* u32[S1*C] */
	FSM_HIP_PLAN_COMB256     = 12, /* u32[] */
	FSM_HIP_PLAN_COMB256_OFF = 13, /* u32[S1] */
	FSM_HIP_PLAN_COMB256_FIN = 14, /* u32[] */
	FSM_HIP_PLAN_COMB_SMASK  = 15, /* u32[] by comb row offset */
	FSM_HIP_PLAN_EMASK       = 16, /* u64[S1] eager-output masks by renumbered state (empty: none) */
	FSM_HIP_PLAN_EAGER_IDS   = 17, /* u32[K] ids in ascending order: bit k of a mask */
	FSM_HIP_PLAN_SPARSE      = 18, /* u32[] image of the sparse layout (header + tables, plan.cpp build_sparse) */
	FSM_HIP_PLAN_EW_OFF      = 19, /* u32[S1+1] wide eager sets (> 64 ids): per state a run of (word, mask) pairs */
	FSM_HIP_PLAN_EW_WORD     = 20, /* u32[] */
	FSM_HIP_PLAN_EW_MASK     = 21, /* u64[] */
	FSM_HIP_PLAN_TINY5_COL   = 22, /* u32[256], <= 6 states: 5-bit fields of 5 * next state */
	FSM_HIP_PLAN_COMB_RNG    = 23, /* u16[] by comb row offset: self-loop byte range lo | hi << 8 (0x0080: none) */
	FSM_HIP_PLAN_LAZY        = 24, /* u32[] image of the sparse layout's lazy form (plan.cpp build_lazy); empty: the automaton has none */
	FSM_HIP_PLAN_GLOB_TAB16  = 25, /* u16[S1*C], global layout of an automaton of <= 65 535 states: the next state's ROW (empty otherwise) */
	FSM_HIP_PLAN_GLOB16_RANK = 26  /* u32[S1]: the row of every renumbered state in that table (rows in visit-frequency order); empty: row = state */
};

/* lds_limit 0 = 160 KiB (gfx950).  NULL + errno on failure. */
struct fsm_hip_plan *fsm_hip_plan_create(const struct fsm_hip_dfa_desc *desc, unsigned flags, uint32_t lds_limit);
void fsm_hip_plan_free(struct fsm_hip_plan *plan);
/* Borrowed pointer into the plan (count = number of elements). */
int fsm_hip_plan_get(const struct fsm_hip_plan *plan, int what, const void **data, size_t *count);

/* Measurement aid: average milliseconds of the fastest of three read-only passes over `bytes`
 * device bytes -- 16 B per lane grid-stride with plain loads, the same with nontemporal loads, and
 * LDS-DMA of 128-byte segments of 1 KiB rows into per-wave tiles (the walk's own input path
 * without the walk) -- i.e. the HBM read rate this device actually sustains.
 * d_scratch4 = 4 writable device bytes.  <0 on error. */
double fsm_hip_stream_read_probe_ms(const void *d_base, size_t bytes, void *d_scratch4, int reps, void *hip_stream);

/* Counter calibration aid: one launch of `ngathers` independent vec_bytes-byte (4 or 16) loads at pseudo-random
 * aligned offsets of the `bytes`-byte device buffer at d_base (make it far larger than the caches).  Returns the
 * kernel's milliseconds (< 0 on error).  tools/fetch_calib.py runs it under rocprofv3 --pmc to see what
 * FETCH_SIZE tallies per gather. */
double fsm_hip_gather_probe_ms(const void *d_base, size_t bytes, size_t ngathers, int vec_bytes, void *d_scratch4, void *hip_stream);

/* The LDS-chain ceiling of the lookup layouts (comb256 / lds / lds2: one random LDS read per input byte on the dependent
 * chain): the same chain -- ds_read_b32, bit-field extract, add, compare, select -- on `waves` wavefronts per workgroup,
 * `blocks_per_cu` workgroups per CU, beside a random table of table_bytes, the input bytes made in registers.  Returns GB/s
 * of "bytes" walked by the whole device (every lane walks `steps`), -1 on error.  d_scratch4: 4 writable device bytes. */
double fsm_hip_lds_chain_probe_gbps(size_t table_bytes, int waves, int blocks_per_cu, size_t steps, void *d_scratch4, void *hip_stream);

/* The workgroup size (wavefronts: max_waves, max_waves - 4, ... >= 8) the library gives a latency-bound per-lane kernel
 * (walk_lines32, walk_generic on an LDS table) that uses `vgprs` vector registers beside a table whose LDS lets
 * `workgroups_by_lds` workgroups share a CU: the one that keeps most wavefronts resident (a SIMD holds 512 / registers of
 * them, 8 at most, in steps of 8 registers; a workgroup of W puts ceil(W / 4) on each SIMD; 32 per CU at most), the larger
 * one on a tie.  76 registers, 2 by LDS, 16 -> 12 (two workgroups of 12 where one of 16 fits: -20 % on 8-64 byte lines).
 * Pure arithmetic (no device needed): exported so that the rule is testable and visible. */
int fsm_hip_waves_by_occupancy(int vgprs, int workgroups_by_lds, int max_waves);

#ifdef __cplusplus
}
#endif

#endif
