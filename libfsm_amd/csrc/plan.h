/*
 * plan.h -- host-side table planner: flat DFA description -> device table images.
 *
 * This is the from-scratch counterpart of the reference's "IR -> 256-entry
 * scratch table" expansion (struct dfa_table, src/libfsm/vm/ir.c:105-130,
 * dfa_table_init/group_to_table/error_to_table :649-691): every state becomes a
 * dense row over the 256 byte values with "no edge" mapped to one synthetic,
 * absorbing, non-accepting DEAD state, which makes the walk branch-free while
 * keeping fsm_exec's result (a missing edge returns 0, src/libfsm/exec.c:133-138).
 * The preconditions fsm_exec re-checks on every call (exec.c:106-114) are
 * checked once, here.
 */
#ifndef FSMHIP_CSRC_PLAN_H
#define FSMHIP_CSRC_PLAN_H

#include <cstdint>
#include <vector>
#include "../../include/fsm_hip.h"

namespace fsmhip {

struct Plan {
	uint32_t nstates = 0;     /* caller's state count                          */
	uint32_t S1 = 0;          /* nstates + 1 (synthetic dead state, last)      */
	uint32_t start = 0;       /* renumbered start state                        */
	uint32_t C = 0;           /* number of byte equivalence classes            */
	uint8_t  cls[256];        /* byte -> class                                 */
	uint32_t abs_min = 0;     /* renumbered states >= abs_min are absorbing    */
	uint32_t nabsorbing = 0;
	std::vector<uint32_t> old2new, new2old;   /* new2old[S1-1] = NO_MATCH (dead) */
	std::vector<uint32_t> fin;    /* [S1] caller's state id if end state else FSM_HIP_NO_MATCH */
	std::vector<uint32_t> dense;  /* [S1][C] next (renumbered) state per class */
	/* end-ids by ORIGINAL state id (CSR) */
	std::vector<uint32_t> endid_off, endids;

	/* eager outputs: distinct ids ascending (bit k of a mask = eager_ids[k]); per renumbered
	 * state the mask of ids it emits.  Renumbering puts eager non-absorbing states first and
	 * eager absorbing states just below DEAD, so "has eager outputs" is
	 *   state < eager_lo_end || state >= eager_hi_begin.                               */
	std::vector<uint32_t> eager_ids;
	std::vector<uint64_t> emask;      /* [S1], empty if the DFA has no eager outputs */
	uint32_t eager_lo_end = 0, eager_hi_begin = 0xFFFFFFFFu;
	/* more than 64 ids: emask[] only flags the emitting states; state n ORs ew_mask[k] into word
	 * ew_word[k] of the input's id set for k in [ew_off[n], ew_off[n+1]) */
	uint32_t eager_words = 0;         /* 64-bit words per id set: ceil(#ids / 64) */
	std::vector<uint32_t> ew_off, ew_word;
	std::vector<uint64_t> ew_mask;

	uint32_t layout = 0;          /* FSM_HIP_LAYOUT_* chosen */

	/* ---- device images (host copies) ---- */
	/* TINY: col[256] = 16 nibbles, nibble s = next state of s on that byte */
	std::vector<uint64_t> tiny_col;
	/* <= 6 states: col5[256], field s (5 bits at bit 5 s) = 5 * next state of s on that byte */
	std::vector<uint32_t> tiny5_col;
	/* LDS dense: u16 entries, row stride row_bytes (multiple of 4);
	 * entry = next_state * (row_bytes/4)                                   */
	std::vector<uint16_t> lds_tab;
	uint32_t row_bytes = 0;
	/* LDS2 (stride 2): lds_tab[state][c1 * C1 + c2] = row index of delta(delta(state, c1), c2) in ENTRIES (state * C1 * C1),
	 * C1 = C + 1: class C is "no byte", the identity of every state -- T[s][c][C] = delta(s, c) serves an odd byte, the bytes
	 * beyond an input's end are given class C.  row_bytes = C1 * C1 * 2. */
	uint32_t lds2_c1 = 0;
	/* COMB: column default + exception comb.
	 * comb entry = (owner_off << 16) | next_off  where *_off are the comb
	 * row offsets (in entries) of the owning / destination state;
	 * dflt[c] = row offset of the default destination of class c.          */
	std::vector<uint32_t> comb;
	std::vector<uint32_t> comb_dflt;      /* [C] */
	std::vector<uint32_t> comb_off;       /* [S1] row offset of each renumbered state */
	std::vector<uint32_t> comb_fin;       /* [comb.size()] fin by row offset (NO_MATCH elsewhere) */
	uint32_t comb_abs_min_off = 0;        /* row offsets >= this are absorbing */
	/* eager outputs in the comb layouts: rows are placed region by region (emitting non-absorbing,
	 * other non-absorbing, absorbing in index order), so the same two thresholds work on row offsets:
	 * a state emits iff off < eager_lo_off || off >= eager_hi_off */
	uint32_t comb_eager_lo_off = 0, comb_eager_hi_off = 0xFFFFFFFFu;
	uint32_t comb256_eager_lo_off = 0, comb256_eager_hi_off = 0xFFFFFFFFu;
	/* COMBSELF: comb + per-row-offset self-loop masks (bit c: class c loops to the same state;
	 * absorbing states: all ones).  Device image = comb[] followed by comb_smask[].   */
	std::vector<uint32_t> comb_smask;
	/* COMBSELF: per row offset the state's self-loop BYTE set when it is one contiguous range lo..hi that a
	 * SWAR test can check on raw input dwords (lo <= 128 and hi <= 127 or hi == 255): lo | hi << 8; 0x0080
	 * (lo 128, hi 0: no byte passes) otherwise.  Absorbing states: 0xFF00 (every byte). */
	std::vector<uint16_t> comb_rng;
	double selfloop_fraction = 0.0;   /* non-absorbing states owning at least one self-loop class */
	/* COMB256: the same over raw bytes (256-wide rows), one default state for
	 * every column: no byte->class lookup at all in the walk.                */
	std::vector<uint32_t> comb256, comb256_off, comb256_fin;
	uint32_t comb256_dflt = 0, comb256_abs_min_off = 0;
	/* GLOBAL: u32 entries [S1][C], entry = next_state * C * 4 (byte offset) */
	std::vector<uint32_t> glob_tab;
	/* GLOBAL, <= 65 535 states: the same table with 2-byte entries (next state's INDEX): twice the rows in the LDS copy of
	 * its head, half the L2 footprint (Glob16Pol, walk_kernels.h) */
	std::vector<uint16_t> glob_tab16;
	/* ... its rows in the order of how often a walk over TEXT is in them (glob16_rank[renumbered state] = row; empty: the
	 * renumbered order): the LDS copy of the table's head then holds the rows that matter, not the ones nearest the start */
	std::vector<uint32_t> glob16_rank, glob16_fin;
	/* SPARSE: base-row records (see build_sparse in plan.cpp); states are plain renumbered ids */
	std::vector<uint32_t> sparse_img;
	uint32_t sparse_lds_bytes = 0;    /* leading part of the image that the kernel mirrors in LDS */
	/* SPARSE, lazy form (plan.cpp build_lazy, walk_lazy.h): a second image for the fixed-stride plain walk in which a
	 * state beyond the LDS set is entered without fetching its record; empty when the automaton has none */
	std::vector<uint32_t> lazy_img;
	uint32_t lazy_lds_bytes = 0;
};

/* Returns 0 or an errno value (EINVAL, ENOMEM, ENOTSUP for a forced layout
 * that cannot hold this DFA).  lds_limit = usable LDS bytes per workgroup. */
/* internal flag (above the public ones of fsm_hip.h): AUTO skips the pair table -- the second plan a dfa keeps for its variable-length fronts */
#define FSM_HIP_PLAN_NO_LDS2 0x4000u

int build_plan(const fsm_hip_dfa_desc *desc, unsigned flags, uint32_t lds_limit, Plan &out);

/* LDS bytes needed by each layout's kernel for `waves` wavefronts per block
 * (tables only; the kernels add their own staging on top). */
uint32_t lds_bytes_tiny();
uint32_t lds_bytes_btab();

} // namespace fsmhip

#endif
