/*
 * plan.cpp -- host-side table planner (see plan.h).
 *
 * Steps:
 *  1. validate the description (a DFA: ranges of one state do not overlap);
 *  2. expand to a dense [S1][256] next-state table, "no edge" -> DEAD
 *     (recipe of src/libfsm/vm/ir.c:649-750 in the reference);
 *  3. merge identical byte columns into equivalence classes;
 *  4. renumber states breadth-first from the start state (rows touched early
 *     in a walk end up adjacent), absorbing states last, DEAD very last, so a
 *     single compare `state >= abs_min` identifies lanes that can never change
 *     state again (the reference VM's STOP shortcut, vm/ir.c:763-766);
 *  5. emit the device image for the chosen layout.
 */
#include "plan.h"

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <unordered_map>

namespace fsmhip {

static const uint32_t NOEDGE = 0xFFFFFFFFu;

uint32_t lds_bytes_tiny() { return 256u * 32u * 8u; }
uint32_t lds_bytes_btab() { return 256u; }

static int build_comb(Plan &p, uint32_t max_entries, bool bytewise);

static int build_sparse(Plan &p, uint32_t lds_limit);
static void build_lazy(Plan &p, uint32_t lds_limit, const uint8_t *bit_of, const std::vector<uint32_t> &base);

int build_plan(const fsm_hip_dfa_desc *d, unsigned flags, uint32_t lds_limit, Plan &p)
try {
	if (d == nullptr || d->nstates == 0 || d->start >= d->nstates ||
	    d->edge_off == nullptr || d->is_end == nullptr) {
		return EINVAL;
	}
	if (d->nstates >= 0x00FFFFFFu) {
		return EINVAL; /* fsm_edge.state is 24 bit, src/libfsm/internal.h:48 */
	}
	const uint32_t S = d->nstates, S1 = S + 1, DEAD = S;
	p.nstates = S;
	p.S1 = S1;

	/* 1+2: dense expansion in the caller's numbering */
	std::vector<uint32_t> nx((size_t)S1 * 256, DEAD);
	for (uint32_t s = 0; s < S; s++) {
		uint32_t a = d->edge_off[s], b = d->edge_off[s + 1];
		if (b < a || (b > a && d->ranges == nullptr)) return EINVAL;
		uint32_t *row = &nx[(size_t)s * 256];
		bool seen[256];
		memset(seen, 0, sizeof seen);
		for (uint32_t k = a; k < b; k++) {
			const fsm_hip_range &r = d->ranges[k];
			if (r.lo > r.hi || r.to >= S) return EINVAL;
			for (unsigned c = r.lo; c <= r.hi; c++) {
				if (seen[c]) return EINVAL; /* two edges on one byte: not a DFA */
				seen[c] = true;
				row[c] = r.to;
			}
		}
	}

	/* 3: byte classes by partition refinement over the rows */
	{
		uint32_t cur[256];
		uint32_t ncls = 1;
		memset(cur, 0, sizeof cur);
		std::unordered_map<uint64_t, uint32_t> m;
		for (uint32_t s = 0; s < S && ncls < 256; s++) {
			const uint32_t *row = &nx[(size_t)s * 256];
			/* cheap skip: a row constant on every current class cannot split */
			m.clear();
			uint32_t nn = 0;
			uint32_t nxt[256];
			for (unsigned c = 0; c < 256; c++) {
				uint64_t key = ((uint64_t)cur[c] << 32) | row[c];
				auto it = m.find(key);
				if (it == m.end()) {
					it = m.emplace(key, nn++).first;
				}
				nxt[c] = it->second;
			}
			if (nn != ncls) {
				memcpy(cur, nxt, sizeof cur);
				ncls = nn;
			}
		}
		/* canonical numbering: by first byte of each class */
		uint32_t remap[256];
		for (unsigned i = 0; i < 256; i++) remap[i] = NOEDGE;
		uint32_t k = 0;
		for (unsigned c = 0; c < 256; c++) {
			if (remap[cur[c]] == NOEDGE) remap[cur[c]] = k++;
			p.cls[c] = (uint8_t)remap[cur[c]];
		}
		p.C = k;
	}
	const uint32_t C = p.C;
	uint8_t rep[256]; /* representative byte per class */
	for (int c = 255; c >= 0; c--) rep[p.cls[c]] = (uint8_t)c;

	/* eager outputs: id -> bit, per original state a mask */
	std::vector<uint64_t> emask_old(S1, 0);
	p.eager_ids.clear();
	if (d->eager_off != nullptr && d->eager_off[S] > 0) {
		if (d->eager_ids == nullptr) return EINVAL;
		p.eager_ids.assign(d->eager_ids, d->eager_ids + d->eager_off[S]);
		std::sort(p.eager_ids.begin(), p.eager_ids.end());
		p.eager_ids.erase(std::unique(p.eager_ids.begin(), p.eager_ids.end()), p.eager_ids.end());
		/* up to 64 ids: one mask per state, carried in a register pair by the walk; more: the mask
		 * only says "emits something" and the ids go into per-state (word, mask) lists (below) */
		const bool wide = p.eager_ids.size() > 64;
		for (uint32_t s = 0; s < S; s++) {
			if (d->eager_off[s + 1] < d->eager_off[s]) return EINVAL;
			for (uint32_t k = d->eager_off[s]; k < d->eager_off[s + 1]; k++) {
				size_t bit = std::lower_bound(p.eager_ids.begin(), p.eager_ids.end(), d->eager_ids[k]) - p.eager_ids.begin();
				emask_old[s] |= wide ? (uint64_t)1 : (uint64_t)1 << bit;
			}
		}
	}
	const bool has_eager = !p.eager_ids.empty();
	p.eager_words = (uint32_t)((p.eager_ids.size() + 63u) / 64u);

	/* 4: renumber */
	std::vector<uint8_t> absorbing(S1, 0);
	for (uint32_t s = 0; s < S1; s++) {
		const uint32_t *row = &nx[(size_t)s * 256];
		bool ab = true;
		for (uint32_t c = 0; c < C && ab; c++) ab = (row[rep[c]] == s);
		absorbing[s] = ab;
	}
	p.old2new.assign(S1, NOEDGE);
	p.new2old.clear();
	p.new2old.reserve(S1);
	{
		std::vector<uint32_t> q;
		q.reserve(S1);
		std::vector<uint8_t> vis(S1, 0);
		vis[d->start] = 1;
		q.push_back(d->start);
		std::vector<uint32_t> lvl(S1, 0);
		for (size_t h = 0; h < q.size(); h++) {
			uint32_t s = q[h];
			const uint32_t *row = &nx[(size_t)s * 256];
			for (uint32_t c = 0; c < C; c++) {
				uint32_t t = row[rep[c]];
				if (!vis[t]) { vis[t] = 1; lvl[t] = lvl[s] + 1u; q.push_back(t); }
			}
		}
		/* Big automata (a literal set's trie below its first levels): breadth-first numbering scatters a path over the levels --
		 * the records of the states one word walks through lie megabytes apart, and a walk that has to fetch the record of every
		 * deep state it enters (walk_lazy.h: the states from F up) pays a cold miss per byte: 8-64 byte lines of which every 8th
		 * ends in a literal ran at 260 GB/s where lines without one ran at 410 (tests/tools/c5_lines_probe.py --plant 8).  Below
		 * the levels that can matter for LDS residency (the ones holding the first 16 Ki states, and the level after them: the
		 * first one a resident record leads to) the states are numbered depth-first by SIBLING BLOCKS: a state's children (in
		 * the breadth-first tree, class order) get consecutive ids when the state is visited -- what the CONSEC records and the
		 * lazy walk's first + rank arithmetic need, exactly as before -- and then the first child's subtree is numbered before
		 * the second child's: along a chain of single-child states the ids are consecutive, eight 16-byte records to a memory
		 * line. */
		if (q.size() > 65536u && getenv("FSM_HIP_PLAN_BFS") == nullptr) {   /* (the variable: an A/B aid, breadth-first all the way down) */
			std::vector<uint32_t> cnt;
			for (uint32_t s : q) { if (lvl[s] >= cnt.size()) cnt.resize(lvl[s] + 1u, 0); cnt[lvl[s]]++; }
			uint32_t keep = 0;          /* levels [0, keep) stay breadth-first */
			uint64_t cum = 0;
			while (keep < cnt.size() && cum + cnt[keep] <= 16384u) cum += cnt[keep++];
			if (keep < cnt.size()) keep++;
			if (keep < cnt.size()) {
				/* children lists of the breadth-first tree, in discovery (= class) order: q itself, level by level, is that order */
				std::vector<uint32_t> parent(S1, NOEDGE), first(S1, NOEDGE), nchild(S1, 0);
				{
					std::vector<uint8_t> v2(S1, 0);
					v2[d->start] = 1;
					for (size_t h = 0; h < q.size(); h++) {
						const uint32_t s = q[h];
						const uint32_t *row = &nx[(size_t)s * 256];
						for (uint32_t c = 0; c < C; c++) {
							const uint32_t t = row[rep[c]];
							if (!v2[t]) { v2[t] = 1; parent[t] = s; }
						}
					}
				}
				/* the children of s are a contiguous run of q (breadth-first order): first[s] = where it starts */
				for (size_t h = 1; h < q.size(); h++) {
					const uint32_t t = q[h], ps = parent[t];
					if (first[ps] == NOEDGE) first[ps] = (uint32_t)h;
					nchild[ps]++;
				}
				std::vector<uint32_t> order;
				order.reserve(q.size());
				size_t h0 = 0;
				while (h0 < q.size() && lvl[q[h0]] < keep) order.push_back(q[h0++]);
				/* the states of level keep - 1, in order: each one's block of children, then depth-first below */
				std::vector<uint32_t> stack;
				for (size_t h = 0; h < h0; h++) {
					if (lvl[q[h]] != keep - 1u) continue;
					stack.push_back(q[h]);
					while (!stack.empty()) {
						const uint32_t s = stack.back();
						stack.pop_back();
						if (nchild[s] == 0) continue;
						for (uint32_t k = 0; k < nchild[s]; k++) order.push_back(q[first[s] + k]);
						for (uint32_t k = nchild[s]; k-- > 0;) stack.push_back(q[first[s] + k]);   /* the first child's subtree first */
					}
				}
				if (order.size() == q.size()) q.swap(order);
			}
		}
		for (uint32_t s = 0; s < S; s++) if (!vis[s]) q.push_back(s); /* unreachable */
		if (!vis[DEAD]) q.push_back(DEAD);
		/* non-absorbing first (BFS order; those with eager outputs ahead of the rest),
		 * absorbing after (those with eager outputs last), DEAD very last */
		for (uint32_t s : q) if (!absorbing[s] && emask_old[s] != 0) p.new2old.push_back(s);
		p.eager_lo_end = (uint32_t)p.new2old.size();
		for (uint32_t s : q) if (!absorbing[s] && emask_old[s] == 0) p.new2old.push_back(s);
		p.abs_min = (uint32_t)p.new2old.size();
		for (uint32_t s : q) if (absorbing[s] && s != DEAD && emask_old[s] == 0) p.new2old.push_back(s);
		p.eager_hi_begin = (uint32_t)p.new2old.size();
		for (uint32_t s : q) if (absorbing[s] && s != DEAD && emask_old[s] != 0) p.new2old.push_back(s);
		p.new2old.push_back(DEAD);
		p.nabsorbing = S1 - p.abs_min;
	}
	for (uint32_t n = 0; n < S1; n++) p.old2new[p.new2old[n]] = n;
	p.start = p.old2new[d->start];
	p.fin.assign(S1, FSM_HIP_NO_MATCH);
	for (uint32_t n = 0; n + 1 < S1; n++) {
		uint32_t o = p.new2old[n];
		if (d->is_end[o]) p.fin[n] = o;
	}
	p.emask.clear();
	p.ew_off.clear();
	p.ew_word.clear();
	p.ew_mask.clear();
	if (has_eager) {
		p.emask.assign(S1, 0);
		for (uint32_t n = 0; n + 1 < S1; n++) p.emask[n] = emask_old[p.new2old[n]];
		if (p.eager_words > 1) {
			/* wide sets: per renumbered state the 64-bit words of the id set it touches */
			p.ew_off.assign((size_t)S1 + 1, 0);
			for (uint32_t n = 0; n + 1 < S1; n++) {
				const uint32_t o = p.new2old[n];
				p.ew_off[n] = (uint32_t)p.ew_word.size();
				for (uint32_t k = d->eager_off[o]; k < d->eager_off[o + 1]; k++) {   /* ids of a state are sorted */
					const size_t bit = std::lower_bound(p.eager_ids.begin(), p.eager_ids.end(), d->eager_ids[k]) - p.eager_ids.begin();
					const uint32_t w = (uint32_t)(bit / 64u);
					if (p.ew_word.size() == p.ew_off[n] || p.ew_word.back() != w) {
						p.ew_word.push_back(w);
						p.ew_mask.push_back(0);
					}
					p.ew_mask.back() |= (uint64_t)1 << (bit % 64u);
				}
			}
			p.ew_off[S1 - 1] = p.ew_off[S1] = (uint32_t)p.ew_word.size();
		}
	} else {
		p.eager_lo_end = 0;
		p.eager_hi_begin = 0xFFFFFFFFu;
	}
	p.new2old[S1 - 1] = FSM_HIP_NO_MATCH;
	p.dense.assign((size_t)S1 * C, 0);
	for (uint32_t n = 0; n < S1; n++) {
		uint32_t o = (n == S1 - 1) ? DEAD : p.new2old[n];
		const uint32_t *row = &nx[(size_t)o * 256];
		for (uint32_t c = 0; c < C; c++) p.dense[(size_t)n * C + c] = p.old2new[row[rep[c]]];
	}
	std::vector<uint32_t>().swap(nx);

	/* end-ids stay keyed by the caller's state ids */
	p.endid_off.assign((size_t)S + 1, 0);
	p.endids.clear();
	if (d->endid_off != nullptr) {
		for (uint32_t s = 0; s < S; s++) {
			uint32_t a = d->endid_off[s], b = d->endid_off[s + 1];
			if (b < a || (b > a && d->endids == nullptr)) return EINVAL;
			p.endids.insert(p.endids.end(), d->endids + a, d->endids + b);
			p.endid_off[s + 1] = (uint32_t)p.endids.size();
		}
	}

	/* 5: layout */
	const uint32_t want = flags & FSM_HIP_LAYOUT_MASK;
	const uint32_t Cpad = (C + 1u) & ~1u;
	const uint64_t dense_lds = (uint64_t)S1 * Cpad * 2u;
	/* the walk kernels need the byte->class table (32 KiB) and room for at
	 * least 8 wavefronts of input staging next to the table */
	const uint64_t lds_room = lds_limit > lds_bytes_btab() + 8u * 4096u
		? lds_limit - lds_bytes_btab() - 8u * 4096u : 0;

	auto emit_tiny = [&]() -> int {
		if (S1 > 16) return ENOTSUP;
		p.tiny_col.assign(256, 0);
		for (unsigned b = 0; b < 256; b++) {
			uint64_t v = 0;
			for (uint32_t s = 0; s < 16; s++) {
				uint32_t t = s < S1 ? p.dense[(size_t)s * C + p.cls[b]] : s;
				v |= (uint64_t)t << (4 * s);
			}
			p.tiny_col[b] = v;
		}
		p.tiny5_col.clear();
		if (S1 <= 6) {   /* 5-bit fields holding 5 * next(s): Tiny5Pol */
			p.tiny5_col.assign(256, 0);
			for (unsigned b = 0; b < 256; b++) {
				uint32_t v = 0;
				for (uint32_t s = 0; s < 6; s++) {
					uint32_t t = s < S1 ? p.dense[(size_t)s * C + p.cls[b]] : s;
					v |= (5u * t) << (5 * s);
				}
				p.tiny5_col[b] = v;
			}
		}
		p.layout = FSM_HIP_LAYOUT_TINY;
		return 0;
	};
	auto emit_lds = [&]() -> int {
		if (dense_lds > lds_room || (uint64_t)S1 * (Cpad / 2u) > 65536u) return ENOTSUP;
		p.row_bytes = Cpad * 2u;
		p.lds_tab.assign((size_t)S1 * Cpad, 0);
		for (uint32_t n = 0; n < S1; n++)
			for (uint32_t c = 0; c < Cpad; c++) {
				uint32_t t = c < C ? p.dense[(size_t)n * C + c] : n;
				p.lds_tab[(size_t)n * Cpad + c] = (uint16_t)(t * (p.row_bytes / 4u));
			}
		p.layout = FSM_HIP_LAYOUT_LDS;
		return 0;
	};
	/* LDS2: the dense table over PAIRS of classes -- one LDS lookup per TWO input bytes.  The lookup layouts are bound by
	 * the LDS array (a wave's 64 random table reads are replayed for every bank conflict: 62-81 % busy at one read per
	 * byte, profiles/r04f_*): halving the reads is the one lever that moves them.  (S + 1) * (C + 1)^2 entries of 16 bits:
	 * up to ~60 Ki entries (300 states x 13 classes, 1000 x 7) beside 16 waves of per-lane input loads.  Plain walks only
	 * (an eager walk must see every state entered). */
	auto emit_lds2 = [&]() -> int {
		const uint64_t C1 = (uint64_t)C + 1u, entries = (uint64_t)S1 * C1 * C1;
		if (has_eager || entries > 61440u || entries * 2u + 256u + 16u * 1024u > lds_limit) return ENOTSUP;
		p.lds2_c1 = (uint32_t)C1;
		p.row_bytes = (uint32_t)(C1 * C1 * 2u);
		p.lds_tab.assign((size_t)entries, 0);
		auto d1 = [&](uint32_t n, uint32_t c) { return c < C ? p.dense[(size_t)n * C + c] : n; };
		for (uint32_t n = 0; n < S1; n++)
			for (uint32_t c1 = 0; c1 < C1; c1++) {
				const uint32_t m = d1(n, c1);
				for (uint32_t c2 = 0; c2 < C1; c2++)
					p.lds_tab[((size_t)n * C1 + c1) * C1 + c2] = (uint16_t)(d1(m, c2) * (uint32_t)(C1 * C1));
			}
		p.layout = FSM_HIP_LAYOUT_LDS2;
		return 0;
	};
	auto emit_comb = [&]() -> int {
		uint32_t max_entries = lds_room > 1024u ? (uint32_t)std::min<uint64_t>((lds_room - 1024u) / 4u, 65535u) : 0u;
		int r = build_comb(p, max_entries, false);
		if (r) return r;
		p.layout = FSM_HIP_LAYOUT_COMB;
		return 0;
	};
	auto emit_combself = [&]() -> int {
		if (C > 32) return ENOTSUP;
		/* LDS image = 8 bytes per comb entry (entry + mask of its target) + 256 */
		uint32_t max_entries = lds_room > 256u ? (uint32_t)std::min<uint64_t>((lds_room - 256u) / 8u, 65535u) : 0u;
		int r = build_comb(p, max_entries, false);
		if (r) return r;
		p.comb_smask.assign(p.comb.size(), 0u);
		p.comb_rng.assign(p.comb.size(), (uint16_t)0x0080u);
		for (uint32_t n = 0; n < S1; n++) {
			uint32_t m = 0;
			for (uint32_t c = 0; c < C; c++)
				if (p.dense[(size_t)n * C + c] == n) m |= 1u << c;
			if (n >= p.abs_min) m = 0xFFFFFFFFu;
			if (C <= 31u) m |= 0x80000000u;   /* class 31 = "no byte": a self-loop of every state (walk_kernels.h step16_part) */
			p.comb_smask[p.comb_off[n]] = m;
			/* the self-loop bytes as one range, if they are one ([0-9]+, [a-z]*, .*, an absorbing state) */
			int lo = -1, hi = -1, runs = 0;
			for (int b = 0; b < 256; b++) {
				const bool in = n >= p.abs_min || ((m >> p.cls[b]) & 1u);
				if (in && (b == 0 || !(n >= p.abs_min || ((m >> p.cls[b - 1]) & 1u)))) { runs++; if (lo < 0) lo = b; }
				if (in) hi = b;
			}
			if (runs == 1 && lo <= 128 && (hi <= 127 || hi == 255))
				p.comb_rng[p.comb_off[n]] = (uint16_t)(lo | (hi << 8));
		}
		p.layout = FSM_HIP_LAYOUT_COMBSELF;
		return 0;
	};
	{
		uint32_t with_loop = 0;
		for (uint32_t n = 0; n < p.abs_min; n++) {
			bool any = false;
			for (uint32_t c = 0; c < C && !any; c++) any = p.dense[(size_t)n * C + c] == n;
			with_loop += any;
		}
		p.selfloop_fraction = p.abs_min ? (double)with_loop / p.abs_min : 0.0;
	}
	/* LDSSELF: the dense table with each row followed by the state's self-loop mask (bit c: class c
	 * maps the state to itself; absorbing states: all ones).  The walk keeps the mask of the current
	 * state in a register (LdsSelfPol), so bytes that do not change the state cost no table lookup and
	 * whole 16-byte chunks can be skipped with one wave vote -- as CombSelfPol, but the states keep
	 * their order, so eager-output DFAs can use it. */
	auto emit_ldsself = [&]() -> int {
		const uint32_t rb = Cpad * 2u + 4u;
		if (C > 32u || (uint64_t)S1 * rb > lds_room || (uint64_t)S1 * (rb / 4u) > 65536u) return ENOTSUP;
		p.row_bytes = rb;
		const uint32_t rw = rb / 2u;   /* u16 slots per row */
		p.lds_tab.assign((size_t)S1 * rw, 0);
		for (uint32_t n = 0; n < S1; n++) {
			uint32_t m = 0;
			for (uint32_t c = 0; c < Cpad; c++) {
				uint32_t t = c < C ? p.dense[(size_t)n * C + c] : n;
				p.lds_tab[(size_t)n * rw + c] = (uint16_t)(t * (rb / 4u));
				if (c < C && t == n) m |= 1u << c;
			}
			if (n >= p.abs_min) m = 0xFFFFFFFFu;
			if (C <= 31u) m |= 0x80000000u;   /* class 31 = "no byte": a self-loop of every state */
			p.lds_tab[(size_t)n * rw + Cpad] = (uint16_t)(m & 0xffffu);
			p.lds_tab[(size_t)n * rw + Cpad + 1] = (uint16_t)(m >> 16);
		}
		p.layout = FSM_HIP_LAYOUT_LDSSELF;
		return 0;
	};
	auto emit_comb256 = [&]() -> int {
		/* no byte->class table in LDS for this layout */
		uint32_t max_entries = (uint32_t)std::min<uint64_t>((lds_room + lds_bytes_btab()) / 4u, 65535u);
		int r = build_comb(p, max_entries, true);
		if (r) return r;
		p.layout = FSM_HIP_LAYOUT_COMB256;
		return 0;
	};
	auto emit_glob = [&]() -> int {
		if ((uint64_t)S1 * C * 4u >= 0xFFFFFFFFull) return ENOTSUP;
		p.glob_tab.resize((size_t)S1 * C);
		for (size_t i = 0; i < p.glob_tab.size(); i++) p.glob_tab[i] = p.dense[i] * C * 4u;
		p.glob_tab16.clear();
		p.glob16_rank.clear();
		p.glob16_fin.clear();
		if (S1 <= 65535u) {
			/* Row order.  Only the head of the table is in LDS, and a step that leaves it costs the whole wavefront an L2 round
			 * trip: which rows are there decides the rate.  Breadth-first order is a proxy (near the start = visited often) that
			 * is wrong where it matters -- the 4 133-state union of rx-style patterns <letters>[0-9]$: the end states (three
			 * letters, then one of TEN digits) are ten times likelier than the four-letter prefix states of the same depth.
			 * So: the occupancy of every state under a memoryless byte source over printable ASCII (every other byte at 1 % of
			 * a printable one's weight), by power iteration from the start state, averaged over the iterations (a periodic
			 * automaton has no limit); the non-absorbing states sorted by it.  The absorbing ones keep their places at the end
			 * (codes >= abs_min: what the kernels' retire test means).  Eager-output ranges are in renumbered order: no re-ordering. */
			std::vector<uint32_t> order(S1);
			for (uint32_t n = 0; n < S1; n++) order[n] = n;
			if (p.emask.empty() && (uint64_t)S1 * C <= (4u << 20)) {
				std::vector<double> w(C, 0.0);
				double W = 0;
				for (unsigned b = 0; b < 256; b++) { const double x = (b >= 0x20 && b <= 0x7e) ? 1.0 : 0.01; w[p.cls[b]] += x; W += x; }
				std::vector<double> pi(S1, 0.0), nx2(S1, 0.0), acc(S1, 0.0);
				pi[p.start] = 1.0;
				const int iters = 48;
				for (int it = 0; it < iters; it++) {
					std::fill(nx2.begin(), nx2.end(), 0.0);
					for (uint32_t n = 0; n < S1; n++) {
						if (pi[n] == 0.0) continue;
						const double m = pi[n] / W;
						const uint32_t *row = &p.dense[(size_t)n * C];
						for (uint32_t c = 0; c < C; c++) nx2[row[c]] += m * w[c];
					}
					pi.swap(nx2);
					if (it >= 8) for (uint32_t n = 0; n < S1; n++) acc[n] += pi[n];
				}
				std::stable_sort(order.begin(), order.begin() + p.abs_min, [&](uint32_t x, uint32_t y) { return acc[x] > acc[y]; });
				p.glob16_rank.assign(S1, 0);
				for (uint32_t r = 0; r < S1; r++) p.glob16_rank[order[r]] = r;
			}
			p.glob_tab16.resize((size_t)S1 * C);
			p.glob16_fin.resize(S1);
			for (uint32_t r = 0; r < S1; r++) {
				const uint32_t n = order[r];
				p.glob16_fin[r] = p.fin[n];
				for (uint32_t c = 0; c < C; c++) {
					const uint32_t t = p.dense[(size_t)n * C + c];
					p.glob_tab16[(size_t)r * C + c] = (uint16_t)(p.glob16_rank.empty() ? t : p.glob16_rank[t]);
				}
			}
		}
		p.layout = FSM_HIP_LAYOUT_GLOBAL;
		return 0;
	};

	auto emit_sparse = [&]() -> int {
		return build_sparse(p, lds_limit);   /* states keep their renumbered ids: eager ranges hold as they are */
	};

	switch (want) {
	case FSM_HIP_LAYOUT_SPARSE: return emit_sparse();
	case FSM_HIP_LAYOUT_LDSSELF: return emit_ldsself();
	case FSM_HIP_LAYOUT_TINY:   return emit_tiny();
	case FSM_HIP_LAYOUT_LDS:    return emit_lds();
	case FSM_HIP_LAYOUT_LDS2:   return emit_lds2();
	case FSM_HIP_LAYOUT_COMB:   return emit_comb();
	case FSM_HIP_LAYOUT_GLOBAL: return emit_glob();
	case FSM_HIP_LAYOUT_COMB256: return emit_comb256();
	case FSM_HIP_LAYOUT_COMBSELF: return emit_combself();
	case FSM_HIP_LAYOUT_AUTO:
		if (emit_tiny() == 0) return 0;
		/* one lookup per two bytes where the pair table fits: ahead of every one-lookup-per-byte layout, the self-loop
		 * ones included -- those walk self-loop runs at 6 TB/s but a transition-dense input at 1.2 (profiles/
		 * r06f_lds2_probe.txt: 107 states, 21 classes), the pair table 4.6-4.75 whatever the input */
		if (!(flags & FSM_HIP_PLAN_NO_LDS2) && emit_lds2() == 0) return 0;
		/* many states sit in self-loops ([0-9]+, .*): bytes that do not change the state then
		 * cost one conflict-free lookup (CombSelfPol) */
		if (p.selfloop_fraction >= 0.15 && emit_combself() == 0) return 0;
		if (emit_comb256() == 0) return 0;
		if (p.selfloop_fraction >= 0.15 && emit_ldsself() == 0) return 0;
		if (emit_lds() == 0) return 0;
		if (emit_comb() == 0) return 0;
		/* too big for LDS.  A plain table that still fits the L2 (<= 8 MB) walks fastest with as much
		 * of its head mirrored in LDS as fits (360 vs 302 GB/s on a 6.7 MB literal-set table); beyond
		 * that, base-row records when they shrink the table at least 4x (Aho-Corasick and other DFAs
		 * whose rows repeat their predecessors': 256 MB -> 20 MB, 180 -> 302 GB/s), else the plain table */
		if ((uint64_t)S1 * C * 4u > ((uint64_t)8 << 20) && emit_sparse() == 0) {
			if ((uint64_t)p.sparse_img.size() * 4u * 4u <= (uint64_t)S1 * C * 4u) return 0;
			std::vector<uint32_t>().swap(p.sparse_img);
		}
		return emit_glob();
	default:
		return EINVAL;
	}
} catch (const std::bad_alloc &) {
	return ENOMEM;
}

/*
 * Base-row records ("sparse" layout) for tables that do not fit LDS.
 *
 * Every non-absorbing state n is either DENSE (a full row of C next states) or SPARSE: a base
 * state b(n), numbered before n, plus the classes on which row(n) differs from row(b(n)) -- a
 * bitmap over (up to 64) classes and the list of next states for the set bits:
 *
 *     delta(n, c) = exc[off(n) + popcount(bits(n) below bit(c))]   if bit(c) is set in bits(n)
 *                 = delta(b(n), c)                                 otherwise (follow the chain)
 *
 * The base is found without knowing where the DFA came from: for a state first reached from its
 * breadth-first parent p on class c, the candidate is delta(b(p), c) (the start state when p is
 * dense).  On an Aho-Corasick DFA this reconstructs the failure links (fail(n) = delta(fail(p), c),
 * src/libre/ac.c:229-241) and the exceptions are the trie children; on other automata it is just a
 * row that is often similar.  A state whose row differs from the candidate's (and from the start
 * state's) on more than half the classes stays dense, so the form is never much larger than the
 * table and the result is the same for any choice.
 *
 * Records are 16 bytes {bits lo, bits hi, base (28 bits) | DENSE | CONSEC | FULLBASE | FAST, offset}; those of the states nearest the
 * start state, and the dense rows among them, are mirrored in LDS (breadth-first numbering puts
 * the states a walk visits most first), the rest stays in HBM/L2.
 *
 * Image (u32 words): hdr[16] | pmap u16[256] (byte -> class | bit << 8, bit 0xff = unmapped)
 *   | LDS dense rows | LDS records || all records (one per state) | all dense rows | exceptions.
 * Everything before the || is copied to LDS by the kernel (Plan::sparse_lds_bytes).
 */
static int build_sparse(Plan &p, uint32_t lds_limit)
{
	const uint32_t S1 = p.S1, C = p.C, N = p.abs_min;   /* records for the non-absorbing states only */
	const uint32_t NONE = 0xFFFFFFFFu, DENSE = 0x80000000u;
	if (N >= DENSE || p.start >= S1) return ENOTSUP;
	auto row = [&](uint32_t n) { return &p.dense[(size_t)n * C]; };

	/* breadth-first parents over the renumbered graph */
	std::vector<uint32_t> parent(N, NONE), order;
	std::vector<uint8_t> pcls(N, 0), seen(N, 0);
	order.reserve(N);
	if (p.start < N) { seen[p.start] = 1; order.push_back(p.start); }
	for (size_t h = 0; h < order.size(); h++) {
		const uint32_t n = order[h];
		for (uint32_t c = 0; c < C; c++) {
			const uint32_t t = row(n)[c];
			if (t < N && !seen[t]) { seen[t] = 1; parent[t] = n; pcls[t] = (uint8_t)c; order.push_back(t); }
		}
	}
	for (uint32_t n = 0; n < N; n++) if (!seen[n]) order.push_back(n);   /* unreachable: dense */

	/* vb[n]: the candidate of n (its "failure state"), kept whether or not n ends up sparse, so
	 * that the candidates of n's children derive from it even across dense states */
	std::vector<uint32_t> base(N, NONE), vb(N, NONE), nexc(N, 0);
	std::vector<uint8_t> done(N, 0), chain(N, 0);
	auto diff = [&](uint32_t a, uint32_t b) {
		uint32_t k = 0;
		const uint32_t *ra = row(a), *rb = row(b);
		for (uint32_t c = 0; c < C; c++) k += ra[c] != rb[c];
		return k;
	};
	const uint32_t thr = C / 2u;
	auto consec_vs = [&](uint32_t a, uint32_t b) {
		const uint32_t *ra = row(a), *rb = row(b);
		uint32_t first = NONE, k = 0;
		for (uint32_t c = 0; c < C; c++) {
			if (ra[c] == rb[c]) continue;
			if (first == NONE) first = ra[c];
			if (ra[c] != first + k) return false;
			k++;
		}
		return k != 0;
	};
	for (uint32_t n : order) {
		done[n] = 1;
		if (n == p.start || parent[n] == NONE || p.start >= N) continue;
		const uint32_t pp = parent[n];
		uint32_t cand = vb[pp] == NONE ? p.start : row(vb[pp])[pcls[n]];
		if (cand >= N || cand == n || !done[cand]) cand = p.start;
		vb[n] = cand;
		if (chain[cand] >= 6) cand = p.start;
		uint32_t k = diff(n, cand);
		/* a row that differs from its base on many classes still makes a 16-byte record when the
		 * differing targets are consecutive ids (CONSEC, below): a trie node with all 64 children costs
		 * a record, not a 260-byte dense row -- which is what lets every depth-2 record of the 1e5-literal
		 * automaton stay in LDS */
		if (k > thr && !consec_vs(n, cand) && cand != p.start) { cand = p.start; k = diff(n, cand); }
		if (k > thr && !consec_vs(n, cand)) continue;
		base[n] = cand;
		nexc[n] = k;
		chain[n] = (uint8_t)(chain[cand] + 1);
	}

	/* classes that get a bit: the 64 most often excepted; a state excepted elsewhere goes dense */
	uint8_t bit_of[256];
	memset(bit_of, 0xff, sizeof bit_of);
	{
		std::vector<uint64_t> cnt(C, 0);
		for (uint32_t n = 0; n < N; n++) {
			if (base[n] == NONE) continue;
			const uint32_t *ra = row(n), *rb = row(base[n]);
			for (uint32_t c = 0; c < C; c++) cnt[c] += ra[c] != rb[c];
		}
		std::vector<uint32_t> byc(C);
		for (uint32_t c = 0; c < C; c++) byc[c] = c;
		std::stable_sort(byc.begin(), byc.end(), [&](uint32_t a, uint32_t b) { return cnt[a] > cnt[b]; });
		/* only classes that are excepted somewhere need a bit: a class on which every row equals its
		 * base row (bytes outside a literal set's alphabet, say) is answered by the chain's dense end */
		uint32_t npick = 0;
		while (npick < C && npick < 64u && cnt[byc[npick]] != 0) npick++;
		std::vector<uint32_t> pick(byc.begin(), byc.begin() + npick);
		std::sort(pick.begin(), pick.end());
		for (size_t k = 0; k < pick.size(); k++) bit_of[pick[k]] = (uint8_t)k;
		if (C > 64u) {
			for (uint32_t n = 0; n < N; n++) {
				if (base[n] == NONE) continue;
				const uint32_t *ra = row(n), *rb = row(base[n]);
				for (uint32_t c = 0; c < C; c++) {
					if (ra[c] != rb[c] && bit_of[c] == 0xff) { base[n] = NONE; nexc[n] = 0; break; }
				}
			}
		}
	}

	/* which states live in LDS: the H nearest the start state (breadth-first numbering).  Half the LDS, so that two
	 * 16-wave workgroups share a CU: the walk waits on gathers, and 32 resident waves measured 283 vs 220 GB/s with
	 * all of LDS and 16 (profiles/r01_c5_sparse.txt) */
	const uint32_t budget = lds_limit / 2u > 4096u ? lds_limit / 2u - 2048u : 0;
	uint32_t H = 0, HD = 0;                                              /* records / dense rows in LDS */
	{
		uint64_t used = 64u + 512u;
		for (uint32_t n = 0; n < N; n++) {
			const uint64_t need = 16u + (base[n] == NONE ? (uint64_t)C * 4u : 0u);
			if (used + need > budget) break;
			used += need;
			H++;
			if (base[n] == NONE) HD++;
		}
	}
	/* Short chains through LDS.  The walk evaluates a byte against a state's own record and, on a miss, against its
	 * base's and that one's base's -- straight-line, from LDS (SparseFastPol, walk_kernels.h).  That is exact when the
	 * third record owns every class (a full trie node: level 0), i.e. when the own record is at level <= 2 with
	 *     level(full record) = 0,   level(n) = 1 + level(base(n)),   both bases among the H LDS records.
	 * A deep state's natural base (its failure state) is itself deep: such a record is re-based onto the first record of
	 * its base chain that is LDS-resident and at level <= 1, and the classes on which the skipped records differed from
	 * that one become exceptions of its own (the result is the same for any choice of base).  A literal-set state then
	 * costs an exception or two more, and a miss on it is three probes, never a walk through HBM. */
	{
		uint32_t nbits = 0;
		for (uint32_t c = 0; c < C; c++) nbits += bit_of[c] != 0xff;
		auto is_full = [&](uint32_t n, uint32_t b) {          /* n differs from b on every class that owns a bit */
			const uint32_t *ra = row(n), *rb = row(b);
			uint32_t k = 0;
			for (uint32_t c = 0; c < C; c++) k += bit_of[c] != 0xff && ra[c] != rb[c];
			return nbits != 0 && k == nbits;
		};
		std::vector<uint8_t> lvl(N, 99);                      /* dense rows and whatever hangs off them: never through */
		for (uint32_t n : order) {
			if (base[n] == NONE) continue;
			if (!is_full(n, base[n])) {
				uint32_t cand = base[n];
				while (cand != NONE && (cand >= H || lvl[cand] > 1)) cand = base[cand];
				if (cand != NONE && cand != n && cand != base[n]) {
					const uint32_t k = diff(n, cand);
					/* (a handful of exceptions at most: the list costs table bytes -- 4 per exception -- and a hit on a listed
					 * class takes the general loop) */
					bool ok = k <= 4u || consec_vs(n, cand);
					const uint32_t *ra = row(n), *rb = row(cand);
					for (uint32_t c = 0; c < C && ok; c++) ok = ra[c] == rb[c] || bit_of[c] != 0xff;   /* every exception owns a bit */
					if (ok) { base[n] = cand; nexc[n] = k; }
				}
			}
			if (is_full(n, base[n])) lvl[n] = 0;
			else if (base[n] < H && lvl[base[n]] < 98) lvl[n] = (uint8_t)(lvl[base[n]] + 1);
		}
	}

	/* CONSEC records: the targets of the exceptions are consecutive state ids in class (= bit) order.
	 * Breadth-first numbering hands the children of a trie node consecutive ids, so on an Aho-Corasick
	 * DFA every record with at least one exception qualifies: the walk then computes the next state as
	 * first + rank(bit) and the exception list -- one dependent gather per hit -- is not stored at all. */
	const uint32_t CONSEC = 0x40000000u, FULLBASE = 0x20000000u, FAST = 0x10000000u;
	if (N >= FAST) return ENOTSUP;
	std::vector<uint8_t> consec(N, 0);
	for (uint32_t n = 0; n < N; n++) {
		if (base[n] == NONE || nexc[n] == 0) continue;
		const uint32_t *ra = row(n), *rb = row(base[n]);
		uint32_t first = NONE, k = 0;
		bool ok = true;
		for (uint32_t c = 0; c < C && ok; c++) {
			if (ra[c] == rb[c]) continue;
			if (first == NONE) first = ra[c];
			ok = ra[c] == first + k;
			k++;
		}
		consec[n] = ok;
	}

	/* sizes */
	uint64_t ndense = 0, ntot_exc = 0;
	for (uint32_t n = 0; n < N; n++) { if (base[n] == NONE) ndense++; else if (!consec[n]) ntot_exc += nexc[n]; }
	const uint64_t words = 16u + 128u + (uint64_t)S1 * 4u + ndense * C + ntot_exc;
	if (words * 4u + 160u * 1024u >= 0xFFFFFFFFull) return ENOTSUP;

	std::vector<uint32_t> &img = p.sparse_img;
	const uint32_t lds_dense_w = 16u + 128u, lds_rec_w = (lds_dense_w + HD * C + 3u) & ~3u;
	const uint32_t lds_words = lds_rec_w + H * 4u;
	/* one record per state, the absorbing ones included (no bits, FAST: never read by the chain loop, which tests
	 * `state < abs_min` first; SparseFastPol enters a state by loading its record without that test) */
	const uint32_t grec_w = lds_words, gdense_w = grec_w + S1 * 4u;
	const uint64_t exc_w64 = (uint64_t)gdense_w + ndense * C;
	const uint32_t exc_w = (uint32_t)exc_w64;
	img.assign((size_t)(exc_w64 + ntot_exc), 0);
	{
		uint16_t *pm = reinterpret_cast<uint16_t *>(&img[16]);
		for (unsigned b = 0; b < 256; b++) pm[b] = (uint16_t)(p.cls[b] | (bit_of[p.cls[b]] << 8));
	}
	uint32_t drow = 0, eoff = 0, maxchain = 0, nconsec = 0, nfullbase = 0;
	for (uint32_t n = 0; n < N; n++) {
		uint32_t *r = &img[grec_w + (size_t)n * 4u];
		if (base[n] == NONE) {
			r[0] = r[1] = 0;
			r[2] = DENSE;
			r[3] = drow * C;
			memcpy(&img[gdense_w + (size_t)drow * C], row(n), (size_t)C * 4u);
			if (drow < HD) memcpy(&img[lds_dense_w + (size_t)drow * C], row(n), (size_t)C * 4u);
			drow++;
		} else {
			uint64_t bits = 0;
			const uint32_t *ra = row(n), *rb = row(base[n]);
			r[3] = eoff;
			bool first = true;
			for (uint32_t c = 0; c < C; c++) {          /* bit order = class order (bit_of is monotone) */
				if (ra[c] == rb[c]) continue;
				bits |= (uint64_t)1 << bit_of[c];
				if (!consec[n]) img[exc_w + eoff++] = ra[c];
				else if (first) r[3] = ra[c];           /* the k-th exception goes to r[3] + k */
				first = false;
			}
			r[0] = (uint32_t)bits;
			r[1] = (uint32_t)(bits >> 32);
			r[2] = base[n] | (consec[n] ? CONSEC : 0u);
			nconsec += consec[n];
			if (chain[n] > maxchain) maxchain = chain[n];
		}
	}
	/* FULLBASE: the base is an LDS-resident CONSEC record with a bit set for EVERY class that owns a bit
	 * (a trie node with all children, e.g. every depth-1 node of a literal set over its whole alphabet).
	 * A miss on such a record's child need not visit the base: the answer is first(base) + bit(class),
	 * one 4-byte LDS read instead of a whole chain step. */
	{
		uint32_t nbits = 0;
		for (uint32_t c = 0; c < C; c++) nbits += bit_of[c] != 0xff;
		const uint64_t all = nbits >= 64u ? ~(uint64_t)0 : (((uint64_t)1 << nbits) - 1u);
		for (uint32_t n = 0; n < N; n++) {
			uint32_t *r = &img[grec_w + (size_t)n * 4u];
			if (base[n] == NONE || base[n] >= H || !consec[base[n]]) continue;
			const uint32_t *rb = &img[grec_w + (size_t)base[n] * 4u];
			if ((rb[0] | ((uint64_t)rb[1] << 32)) == all) { r[2] |= FULLBASE; nfullbase++; }
		}
	}
	/* FAST: every class (that owns a bit and) that this record does NOT own is answered by SparseFastPol's straight-line
	 * step (walk_kernels.h) -- a hit on the record itself is, when the record is CONSEC (first + rank); a hit on one of
	 * the few records that keep an exception list takes the general loop --: the class is answered
	 * either by its base B -- an LDS-resident CONSEC record -- or, where B does not own the class either, by B's base C,
	 * an LDS-resident CONSEC record that owns EVERY bit (rank = bit index: the answer is first(C) + bit, one 4-byte LDS
	 * read of C's `first` word and an add -- no third popcount probe).  The step checks this one flag instead of
	 * re-deriving all that per byte.  (first(base(B)) in an LDS word of its own next to each record, read beside B's
	 * record instead of after it, was tried: 4 bytes per LDS record push 210 of the 4 096 depth-2 records of the
	 * 1e5-literal automaton out of LDS and with them a third of the deep records out of FAST.) */
	uint32_t nfastmiss = 0;
	{
		uint32_t nbits = 0;
		for (uint32_t c = 0; c < C; c++) nbits += bit_of[c] != 0xff;
		const uint64_t all = nbits >= 64u ? ~(uint64_t)0 : (((uint64_t)1 << nbits) - 1u);
		auto bits_of = [&](uint32_t n) { const uint32_t *r = &img[grec_w + (size_t)n * 4u]; return r[0] | ((uint64_t)r[1] << 32); };
		auto full_consec = [&](uint32_t n) { return n < H && base[n] != NONE && consec[n] && bits_of(n) == all; };
		for (uint32_t n = 0; n < N; n++) {
			if (base[n] == NONE) continue;
			const uint64_t need = all & ~bits_of(n);
			bool ok = need == 0;
			if (!ok) {
				const uint32_t B = base[n];
				if (B < H && base[B] != NONE) {
					const uint64_t inB = need & bits_of(B), rest = need & ~bits_of(B);
					ok = inB == 0 || consec[B];
					if (ok && rest != 0) ok = full_consec(base[B]);
				}
			}
			if (ok) { img[grec_w + (size_t)n * 4u + 2u] |= FAST; nfastmiss++; }
		}
	}
	/* the absorbing states' records: no bits, FAST (the step's one flag test lets them through; it keeps their id) */
	for (uint32_t n = N; n < S1; n++) img[grec_w + (size_t)n * 4u + 2u] = FAST;
	for (uint32_t n = 0; n < H; n++) memcpy(&img[lds_rec_w + (size_t)n * 4u], &img[grec_w + (size_t)n * 4u], 16);
	img[0] = 0x31525053u;   /* "SPR1" */
	img[1] = H;
	img[2] = HD * C;
	img[3] = lds_dense_w * 4u;
	img[4] = lds_rec_w * 4u;
	img[5] = grec_w * 4u;
	img[6] = gdense_w * 4u;
	img[7] = exc_w * 4u;
	img[8] = N;
	img[9] = C;
	img[10] = (uint32_t)ndense;
	img[11] = (uint32_t)ntot_exc;
	img[12] = maxchain;
	img[13] = nconsec;
	img[14] = nfullbase;
	img[15] = nfastmiss;
	p.sparse_lds_bytes = lds_words * 4u;
	p.layout = FSM_HIP_LAYOUT_SPARSE;
	build_lazy(p, lds_limit, bit_of, base);
	return 0;
}

/*
 * The LAZY form of the sparse layout (walk_lazy.h): a second, self-contained image for the fixed-stride plain walk.
 *
 * build_sparse's walk fetches the 16-byte record of every state it enters; on a literal-set automaton two thirds of
 * those fetches come from LDS (the H records nearest the start state) and one third -- the first level that no longer
 * fits -- is an L2 gather whose only use, 98 % of the time, is to learn "no exception on this byte": 0.33 L2 requests
 * per input byte, which is what bounds that walk (profiles/r04p_c5_memory_pipeline.txt).  Here a state beyond the LDS
 * set is entered WITHOUT its record.  The walk carries (id, E): E = id for an LDS-resident state, else the LDS-resident
 * record "behind" it (on an Aho-Corasick automaton: the deepest LDS-resident node on its failure chain,
 * fail(n) = delta(fail(p), c), src/libre/ac.c:229-241), derived by arithmetic from the step that entered the state.
 * A byte is answered by E's record unless the state's own record excepts it; whether it may is asked of a bit array
 * in LDS keyed by (id, class) -- a one-hash Bloom filter over the exceptions of the states [H, F) -- and only a set
 * bit (a true exception, 2 % of the steps of such a state on the 1e5-literal automaton, or a false positive) costs
 * the gather of the state's own record.  States from F up are always fetched (they are entered rarely: one has to
 * take a true exception to get there).
 *
 * LDS fast record of state e < H, 16 bytes {bits, fm1, cf63}: for a class with bit b
 *     delta(e, class) = fm1 + popcount(bits << (63 - b))     if bit b of bits is set (targets are consecutive ids),
 *                     = cf63 - (63 - b)  (= X + b)           otherwise,
 * X being the most frequent value of `target - b` over the state's row (a trie node whose failure state owns every
 * child: X = first child of that state).  A result with bit 31 set is a SENTINEL: the kernel then takes that byte
 * through build_sparse's exact records (fm1 = 0x80000000 when the excepted targets are not consecutive).
 * What the walk carries into a state m >= H entered from (id, E) on class c is rep = ev >= H ? cf(E) + b : ev, with
 * ev = E's answer on c.  The planner follows every reachable state with exactly that rule (car[]); a state reached
 * with two different values, or through the exact path, is served by the exact path only (its own record says
 * "except everything": bits = ~0, fm1 = sentinel), and the own record of every other state m >= H lists the classes
 * on which its row differs from car[m]'s answers -- so the result is delta for ANY automaton, by construction, and
 * tests/test_plan.py re-derives that for every reachable (state, carried record, byte).
 *
 * Image (u32 words): hdr[16] | LDS part: sh[256] at LDS address 0 (byte -> 63 - bit, or 0x80000000 for bytes whose
 * class owns no bit), filter[fwords], records[H + 1] (the last one all-sentinel: Z)
 * | own records, 16 bytes per state {bits, base, stride}, then one per CLONE (round 6: img[12] of them, ids S1 ..) | car[S1] | cloneof[].
 */
static void build_lazy(Plan &p, uint32_t lds_limit, const uint8_t *bit_of, const std::vector<uint32_t> &base)
{
	p.lazy_img.clear();
	p.lazy_lds_bytes = 0;
	const uint32_t S1 = p.S1, C = p.C, N = p.abs_min, SENT = 0x80000000u, NONE = 0xFFFFFFFFu;
	if (!p.emask.empty() || S1 >= (1u << 24)) return;          /* plain walks only */
	auto row = [&](uint32_t n) { return &p.dense[(size_t)n * C]; };
	uint32_t nbits = 0, cls_of_bit[64];
	for (uint32_t c = 0; c < C; c++)
		if (bit_of[c] != 0xff) { cls_of_bit[bit_of[c]] = c; nbits++; }
	if (nbits == 0) return;
	struct FRec { uint64_t bits; uint32_t fm1, cf63; };
	auto popc = [](uint64_t v) { return (uint32_t)__builtin_popcountll(v); };

	/* 1. the LDS set: the longest prefix of states (breadth-first from the start state) whose excepted targets are
	 * consecutive ids, and whose X + b stays inside the set for every bit b */
	/* the variable-length kernel keeps its queue of input tails behind the table (walk_lazy.h FSMHIP_LAZY_QBYTES: 16 wavefronts x
	 * 112 entries x 16 bytes): the image is planned into what that leaves */
	const uint32_t queue_bytes = 16u * 112u * 16u;
	if (lds_limit <= queue_bytes) return;
	lds_limit -= queue_bytes;
	const uint32_t room = lds_limit > 1024u + 16384u + 32u ? lds_limit - 1024u - 16384u - 32u : 0u;
	const uint32_t Hcap = room / 16u;
	std::vector<FRec> LR;
	std::vector<uint64_t> need;
	std::vector<uint32_t> Xof;
	for (uint32_t n = 0; n < N && n < Hcap; n++) {
		const uint32_t *rn = row(n);
		/* candidates for X, best first: what the state's base (on a literal set: its failure state f) answers on the class
		 * of bit 0 -- when f's row is linear (a trie node with every child) X + b is then delta(f, b), the failure state of
		 * the child on b: exactly what that child should carry --, then the values of target - b by how many classes vote
		 * for them.  The first one whose exceptions have consecutive targets is taken. */
		uint32_t cand[65], cnt[65], nc = 0;
		if (n < base.size() && base[n] != NONE && base[n] < N) { cand[0] = row(base[n])[cls_of_bit[0]]; cnt[0] = 0xFFFFFFFFu; nc = 1; }
		for (uint32_t b = 0; b < nbits; b++) {
			const uint32_t v = rn[cls_of_bit[b]] - b;
			uint32_t k = 0;
			while (k < nc && cand[k] != v) k++;
			if (k == nc) { cand[nc] = v; cnt[nc++] = 0; }
			if (cnt[k] != 0xFFFFFFFFu) cnt[k]++;
		}
		bool found = false;
		FRec r = { 0, 0, 0 };
		uint32_t X = 0;
		for (uint32_t round = 0; round < nc && !found; round++) {
			uint32_t best = 0;
			for (uint32_t k = 1; k < nc; k++) if (cnt[k] > cnt[best]) best = k;
			if (cnt[best] == 0) break;
			X = cand[best];
			cnt[best] = 0;
			r.bits = 0;
			r.cf63 = X + 63u;
			uint32_t first = NONE, k = 0;
			bool consec = true;
			for (uint32_t b = 0; b < nbits; b++) {
				const uint32_t t = rn[cls_of_bit[b]];
				if (t == X + b) continue;
				r.bits |= (uint64_t)1 << b;
				if (first == NONE) first = t;
				if (t != first + k) consec = false;
				k++;
			}
			r.fm1 = k == 0 ? 0u : first - 1u;
			found = consec && !(r.fm1 & SENT) && !(r.cf63 & SENT) && X + nbits >= X;
		}
		if (!found) break;
		LR.push_back(r);
		Xof.push_back(X);
		need.push_back((uint64_t)X + nbits);   /* X + b < H for every b */
	}
	uint32_t H = 0;
	{
		uint64_t mx = 0;
		std::vector<uint64_t> pm(LR.size() + 1, 0);
		for (size_t n = 0; n < LR.size(); n++) { mx = need[n] > mx ? need[n] : mx; pm[n + 1] = mx; }
		for (size_t h = LR.size(); h > 0; h--)
			if (pm[h] <= h) { H = (uint32_t)h; break; }
	}
	if (getenv("FSM_HIP_DEBUG")) fprintf(stderr, "build_lazy: nbits=%u LR=%zu H=%u start=%u N=%u Hcap=%u\n", nbits, LR.size(), H, p.start, N, Hcap);
	if (H == 0 || p.start >= H) return;
	LR.resize(H);
	uint32_t fwords = 4096;
	while ((uint64_t)fwords * 2u * 4u + ((uint64_t)H + 1u) * 16u + 1024u <= lds_limit) fwords *= 2u;
	if ((uint64_t)fwords * 4u + ((uint64_t)H + 1u) * 16u + 1024u > lds_limit) return;
	const uint32_t Z = H;

	auto evalF = [&](uint32_t e, uint32_t b) -> uint32_t {
		const FRec &r = LR[e];
		if ((r.bits >> b) & 1u) return r.fm1 + popc(r.bits & (((uint64_t)2 << b) - 1u));
		return r.cf63 - (63u - b);
	};
	auto cfbF = [&](uint32_t e, uint32_t b) -> uint32_t { return LR[e].cf63 - (63u - b); };
	/* the own record of a state s >= H that carries e: the classes on which its row differs from e's answers */
	/* ... as {bits, base, stride}: the k-th exception (in bit order, k from 1) leads to base + k * stride.  That covers the
	 * children of a trie node (consecutive ids: stride 1) and ANY record with two exceptions -- the usual shape of a deep
	 * literal-set node whose failure state is itself beyond the LDS set: one child of its own, one inherited --; base =
	 * sentinel where the targets are no such progression (a hit then takes the exact path) */
	/* ... and where they are not (round 6): CLONES.  44 816 of the 982 000 deep states of the 1e5-literal automaton have three
	 * exceptions that are no progression -- their own child and two inherited ones -- and a hit on such a record used to be a
	 * sentinel: the whole chunk re-walked the exact way, the wavefront waiting.  A word planted in a line passes ~9 deep states
	 * and met one in a third of the cases: 8-64 byte lines of which every 8th ends in a literal ran at 260 GB/s where lines
	 * without one ran at 410.  Such a state now gets k fresh consecutive ids behind the last real state, one per exception in
	 * class order -- clone j stands for the state its j-th exception leads to: same own record, same carried record (the
	 * carried record of a state entered from (state, class) does not depend on the target's id), cloneof[] maps it back
	 * wherever an id leaves the fast path (results, the exact re-walk) -- and its record is the progression {bits, first clone
	 * - 1, 1}.  `clonable`: every target lies beyond the LDS set, at most 8 of them. */
	const uint32_t CLONE_MAX = 8;
	auto make_deep = [&](uint32_t s, uint32_t e, uint64_t &bits, uint32_t &fm1, uint32_t &stride, bool *clonable = nullptr) {
		stride = 0;
		if (clonable) *clonable = false;
		if (e == Z) { bits = ~(uint64_t)0; fm1 = SENT; return; }
		bits = 0;
		uint32_t t0 = 0, k = 0;
		int64_t d = 1;
		bool ok = true, deep = true;
		const uint32_t *rs = row(s);
		for (uint32_t b = 0; b < nbits; b++) {
			const uint32_t f = evalF(e, b);
			if (f & SENT) continue;                 /* the step sees the sentinel and takes the exact path whatever this record says */
			const uint32_t t = rs[cls_of_bit[b]];
			if (t == f) continue;
			bits |= (uint64_t)1 << b;
			if (k == 0) t0 = t;
			else if (k == 1) d = (int64_t)t - (int64_t)t0;
			if ((int64_t)t != (int64_t)t0 + (int64_t)k * d || t < H) ok = false;    /* (a target inside the LDS set would have to carry itself) */
			if (t < H || t >= N) deep = false;
			k++;
		}
		if (d == 0 || d >= (1 << 22) || d <= -(1 << 22)) ok = false;
		if (k == 0) { fm1 = 0; return; }
		if (!ok) { fm1 = SENT; if (clonable) *clonable = deep && k <= CLONE_MAX; return; }
		stride = (uint32_t)(int32_t)d;
		fm1 = t0 - stride;
	};

	/* 2. what each state beyond the LDS set carries, by the kernel's own rule, over everything reachable.  A state entered
	 * through the exact path (a sentinel step) is GIVEN its record by the kernel -- word 3 of its own record, car[] -- so
	 * only the straight-line entries constrain car[]; a state they reach with two different values gets Z (its own record
	 * then excepts everything: all its steps are exact ones). */
	std::vector<uint32_t> car(S1, NONE), sugg(S1, NONE);
	std::vector<uint8_t> seen(S1, 0), inq(S1, 0);
	std::vector<uint32_t> q, deferred;
	bool abs_reach = false;
	seen[p.start] = 1; inq[p.start] = 1; q.push_back(p.start);
	for (;;) {
		while (!q.empty()) {
			const uint32_t s = q.back();
			q.pop_back();
			inq[s] = 0;
			const uint32_t es = s < H ? s : car[s];
			uint64_t sb = 0;
			uint32_t sfm1 = 0, sstr = 0;
			bool sclon = false;
			if (s >= H) make_deep(s, es, sb, sfm1, sstr, &sclon);
			const uint32_t *rs = row(s);
			for (uint32_t c = 0; c < C; c++) {
				const uint32_t m = rs[c], b = bit_of[c];
				if (m >= N) { abs_reach = true; continue; }   /* absorbing: the kernel keeps the id, nothing is carried */
				if (m < H) {
					if (!seen[m]) { seen[m] = 1; inq[m] = 1; q.push_back(m); }
					continue;
				}
				bool exact = true;
				uint32_t cr = Z;
				if (b != 0xff && es != Z) {
					const uint32_t f = evalF(es, b);
					const bool own = s >= H && ((sb >> b) & 1u);
					if (!(f & SENT)) {
						cr = f >= H ? cfbF(es, b) : f;
						exact = own && sfm1 == SENT && !sclon;      /* (a clonable record's hit enters a clone of m straight-line: m must carry cr) */
					}
				}
				if (exact) {
					/* entered through the exact path: takes whatever car[m] turns out to be; failing a straight-line entry,
					 * what one would have carried here (so that the states below a record with scattered exceptions
					 * still walk the fast way) */
					if (cr != Z && sugg[m] == NONE) sugg[m] = cr;
					if (!seen[m]) { seen[m] = 1; deferred.push_back(m); }
					else if (car[m] == NONE) deferred.push_back(m);
					continue;
				}
				seen[m] = 1;
				if (car[m] == NONE) { car[m] = cr; inq[m] = 1; q.push_back(m); }
				else if (car[m] != cr && car[m] != Z) { car[m] = Z; if (!inq[m]) { inq[m] = 1; q.push_back(m); } }
			}
		}
		/* states only ever entered through the exact path so far */
		bool any = false;
		for (uint32_t m : deferred)
			if (car[m] == NONE) { car[m] = sugg[m] != NONE ? sugg[m] : Z; inq[m] = 1; q.push_back(m); any = true; }
		deferred.clear();
		if (!any) break;
	}

	/* 3. the image */
	const uint32_t sh_off = 0, filt_off = 1024u, rec_off = filt_off + fwords * 4u, lds_bytes = rec_off + (H + 1u) * 16u;
	const uint32_t lds_w = lds_bytes / 4u, grec_w = (16u + lds_w + 3u) & ~3u;
	std::vector<uint32_t> &img = p.lazy_img;
	/* the clones: decided now (the carried records are final), numbered from S1 up */
	std::vector<uint32_t> cloneof;                 /* clone S1 + j stands for state cloneof[j] */
	std::vector<uint32_t> clone_first(S1, 0);      /* a cloned state's first clone */
	for (uint32_t s2 = H; s2 < N; s2++) {
		uint64_t bits;
		uint32_t fm1, stride;
		bool clon = false;
		make_deep(s2, car[s2] == NONE ? Z : car[s2], bits, fm1, stride, &clon);
		if (!clon || (uint64_t)S1 + cloneof.size() + CLONE_MAX >= (1u << 24)) continue;
		clone_first[s2] = S1 + (uint32_t)cloneof.size();
		for (uint32_t b = 0; b < nbits; b++)
			if ((bits >> b) & 1u) cloneof.push_back(row(s2)[cls_of_bit[b]]);
	}
	const uint32_t NCL = (uint32_t)cloneof.size();
	const size_t car_w = (size_t)grec_w + ((size_t)S1 + NCL) * 4u;
	img.assign(car_w + S1 + NCL, 0);
	uint32_t *L = &img[16];
	for (uint32_t n = 0; n < H; n++) {
		uint32_t *r = L + rec_off / 4u + n * 4u;
		r[0] = (uint32_t)LR[n].bits; r[1] = (uint32_t)(LR[n].bits >> 32); r[2] = LR[n].fm1; r[3] = LR[n].cf63;
	}
	{
		uint32_t *r = L + rec_off / 4u + Z * 4u;     /* Z: no bits, every answer a sentinel */
		r[0] = r[1] = 0; r[2] = SENT; r[3] = SENT + 63u;
	}
	for (unsigned v = 0; v < 256; v++) {
		const uint32_t b = bit_of[p.cls[v]];
		L[sh_off / 4u + v] = b != 0xff ? 63u - b : SENT;   /* bit 31: the kernel ORs a chunk's 16 entries into its sentinel test */
	}
	uint32_t nzstates = 0, nsent = 0;
	for (uint32_t s = H; s < N; s++) {
		uint64_t bits;
		uint32_t fm1, stride;
		make_deep(s, car[s] == NONE ? Z : car[s], bits, fm1, stride);
		if (clone_first[s] != 0) { fm1 = clone_first[s] - 1u; stride = 1u; }     /* the k-th exception leads to its k-th clone */
		uint32_t *r = &img[grec_w + (size_t)s * 4u];
		r[0] = (uint32_t)bits; r[1] = (uint32_t)(bits >> 32); r[2] = fm1; r[3] = stride;
		img[car_w + s] = car[s] == NONE ? Z : car[s];   /* what the state carries: read by the kernel after an exact step into it */
		if (car[s] == Z) nzstates++;
		else if (fm1 == SENT) nsent++;
	}
	/* a clone's own record is the record of the state it stands for (written above: clones of clones' targets included) */
	for (uint32_t j = 0; j < NCL; j++) {
		memcpy(&img[grec_w + ((size_t)S1 + j) * 4u], &img[grec_w + (size_t)cloneof[j] * 4u], 16);
		img[car_w + S1 + j] = cloneof[j];
	}
	/* the filter: the exceptions of the states [H, F), as many states as keep it under ~0.22 keys per bit */
	uint64_t nkeys = 0;
	uint32_t F = H;
	const uint64_t fbits = (uint64_t)fwords * 32u, budget = fbits * 22u / 100u;
	for (uint32_t s = H; s < N; s++) {
		const uint32_t *r = &img[grec_w + (size_t)s * 4u];
		const uint64_t bits = r[0] | ((uint64_t)r[1] << 32);
		const uint32_t k = popc(bits);
		if (nkeys + k > budget) break;
		for (uint32_t b = 0; b < 64; b++) {
			if (!((bits >> b) & 1u)) continue;
			L[filt_off / 4u + (s & (fwords - 1u))] |= 1u << ((63u - b) & 31u);   /* one word per state id, bit sh % 32 (walk_lazy.h) */
		}
		nkeys += k;
		F = s + 1u;
	}
	img[0] = 0x31595a4cu;   /* "LZY1" */
	img[1] = H;
	img[2] = F;
	img[3] = fwords;
	img[4] = rec_off;
	img[5] = filt_off;
	img[6] = lds_bytes;
	img[7] = grec_w * 4u;
	img[8] = (uint32_t)nkeys;
	img[9] = nzstates;
	img[10] = nsent;
	img[11] = abs_reach ? 1u : 0u;
	img[12] = NCL;                         /* clones: own records S1 .. S1 + NCL - 1, cloneof[] right behind car[] */
	img[13] = S1;
	img[14] = (uint32_t)(car_w * 4u);
	{
		/* does any LDS record answer with a state from F up?  (Those are fetched whatever the filter says; where none
		 * does the kernel is spared the compare) */
		uint32_t mx = 0;
		for (uint32_t n = 0; n < H; n++)
			for (uint32_t b = 0; b < nbits; b++) { const uint32_t t = row(n)[cls_of_bit[b]]; if (t < N && t > mx) mx = t; }
		img[15] = mx >= F ? 1u : 0u;
	}
	p.lazy_lds_bytes = lds_bytes;
}

/*
 * Column-default + comb ("row displacement") compression.
 *
 * dflt[c] = the most common destination of class c over all states.  A state
 * only stores the classes where it differs from dflt (its exceptions); rows
 * are overlaid into one array so that no two exceptions share a slot, and
 * every state gets a distinct row offset which doubles as its id on the
 * device.  delta(s, c) = comb[off(s)+c] if that slot is owned by off(s),
 * else dflt[c].  Regex unions are mostly "no edge" (98 % dead in SURVEY.md
 * section 6's 4 609-state union) and Aho-Corasick rows mostly equal the root
 * row, so both shrink by an order of magnitude and fit LDS.
 */
static int build_comb(Plan &p, uint32_t max_entries, bool bytewise)
{
	/* W columns: byte classes (CombPol) or raw bytes (Comb256Pol) */
	const uint32_t S1 = p.S1, C = p.C, W = bytewise ? 256u : C;
	if (max_entries < W + 1) return ENOTSUP;
	auto cell = [&](uint32_t n, uint32_t w) -> uint32_t {
		return p.dense[(size_t)n * C + (bytewise ? p.cls[w] : w)];
	};

	std::vector<uint32_t> dflt(W);
	{
		std::vector<uint32_t> cdf(C), cnt(S1);
		for (uint32_t c = 0; c < C; c++) {
			std::fill(cnt.begin(), cnt.end(), 0);
			uint32_t best = 0;
			for (uint32_t n = 0; n < S1; n++) {
				uint32_t t = p.dense[(size_t)n * C + c];
				if (++cnt[t] > cnt[best]) best = t;
			}
			cdf[c] = best;
		}
		for (uint32_t w = 0; w < W; w++) dflt[w] = cdf[bytewise ? p.cls[w] : w];
	}
	if (bytewise) {
		/* one default for every column, else the walk would need a per-byte lookup */
		for (uint32_t w = 1; w < W; w++) if (dflt[w] != dflt[0]) return ENOTSUP;
	}
	std::vector<uint32_t> nexc(S1, 0);
	uint64_t total = 0;
	for (uint32_t n = 0; n < S1; n++) {
		for (uint32_t w = 0; w < W; w++)
			if (cell(n, w) != dflt[w]) nexc[n]++;
		total += nexc[n];
	}
	if (total + W > max_entries || S1 > max_entries) return ENOTSUP;

	/* place non-absorbing states densest first -- those with eager outputs (renumbered ids below
	 * eager_lo_end) as a block of their own below the others --, then absorbing ones above, in index
	 * order (eager absorbing states just below DEAD, as in the renumbering) */
	const uint32_t lo_end = p.emask.empty() ? 0u : p.eager_lo_end;
	auto region = [&](uint32_t n) { return n >= p.abs_min ? 2 : n < lo_end ? 0 : 1; };
	std::vector<uint32_t> order(S1);
	for (uint32_t n = 0; n < S1; n++) order[n] = n;
	std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
		const int ra = region(a), rb = region(b);
		if (ra != rb) return ra < rb;
		if (ra == 2) return a < b;            /* absorbing: keep order, DEAD last */
		return nexc[a] > nexc[b];
	});

	/* exception column lists (CSR) so the fit test touches only real entries */
	std::vector<uint32_t> exo(S1 + 1, 0);
	for (uint32_t n = 0; n < S1; n++) exo[n + 1] = exo[n] + nexc[n];
	std::vector<uint16_t> exc(exo[S1]);
	for (uint32_t n = 0; n < S1; n++) {
		uint32_t k = exo[n];
		for (uint32_t w = 0; w < W; w++)
			if (cell(n, w) != dflt[w]) exc[k++] = (uint16_t)w;
	}

	std::vector<uint8_t> slot_used(max_entries + W, 0), off_used(max_entries + 1, 0);
	std::vector<uint32_t> &off = bytewise ? p.comb256_off : p.comb_off;
	off.assign(S1, 0);
	uint32_t hi = 0;         /* 1 + highest offset handed out so far */
	uint32_t first_free = 0; /* lowest offset not yet handed out */
	uint32_t abs_min_off = 0, eager_lo_off = 0, eager_hi_off = 0xFFFFFFFFu, floor = 0;
	bool in_abs = false, in_plain = false;
	for (uint32_t idx = 0; idx < S1; idx++) {
		uint32_t n = order[idx];
		if (!in_plain && !in_abs && region(n) == 1) {
			in_plain = true;
			eager_lo_off = floor = hi; /* rows without eager outputs sit above every row that has them */
		}
		if (!in_abs && n >= p.abs_min) {
			in_abs = true;
			abs_min_off = hi; /* absorbing states get offsets >= every other state's */
			if (!in_plain) eager_lo_off = hi;
		}
		if (!p.emask.empty() && n == p.eager_hi_begin) eager_hi_off = hi;
		/* absorbing offsets keep increasing so one compare identifies them */
		uint32_t o = in_abs ? hi : (first_free > floor ? first_free : floor);
		for (;; o++) {
			if (o + W > max_entries) return ENOTSUP;
			if (off_used[o]) continue;
			bool ok = true;
			for (uint32_t k = exo[n]; k < exo[n + 1] && ok; k++)
				if (slot_used[o + exc[k]]) ok = false;
			if (ok) break;
		}
		off_used[o] = 1;
		for (uint32_t k = exo[n]; k < exo[n + 1]; k++) slot_used[o + exc[k]] = 1;
		off[n] = o;
		if (o + 1 > hi) hi = o + 1;
		while (first_free < max_entries && off_used[first_free]) first_free++;
	}
	uint32_t size = hi + W;
	if (size > max_entries || size > 0xFFFFu) return ENOTSUP;

	std::vector<uint32_t> &comb = bytewise ? p.comb256 : p.comb;
	std::vector<uint32_t> &cfin = bytewise ? p.comb256_fin : p.comb_fin;
	/* owner 0xFFFF never matches a real offset.  CombPol / CombSelfPol entries are owner << 16 | next; the
	 * bytewise form (Comb256Pol) keeps the NEXT state in the high half, next << 16 | owner: the walk then
	 * carries the raw entry as its state and needs one 16-bit compare and one select per byte */
	comb.assign(size, bytewise ? 0x0000FFFFu : 0xFFFF0000u);
	cfin.assign(size, FSM_HIP_NO_MATCH);
	for (uint32_t n = 0; n < S1; n++) {
		uint32_t o = off[n];
		cfin[o] = p.fin[n];
		for (uint32_t w = 0; w < W; w++) {
			uint32_t t = cell(n, w);
			if (t != dflt[w]) comb[o + w] = bytewise ? (off[t] << 16) | o : (o << 16) | off[t];
		}
	}
	if (p.emask.empty()) { eager_lo_off = 0; eager_hi_off = 0xFFFFFFFFu; }
	(bytewise ? p.comb256_eager_lo_off : p.comb_eager_lo_off) = eager_lo_off;
	(bytewise ? p.comb256_eager_hi_off : p.comb_eager_hi_off) = eager_hi_off;
	if (bytewise) {
		p.comb256_dflt = off[dflt[0]];
		p.comb256_abs_min_off = abs_min_off;
	} else {
		p.comb_dflt.resize(C);
		for (uint32_t c = 0; c < C; c++) p.comb_dflt[c] = off[dflt[c]];
		p.comb_abs_min_off = abs_min_off;
	}
	return 0;
}

} // namespace fsmhip
