/*
 * walk_packed.h -- gfx950 device code: the packed front for SHORT inputs (the lines retest / rx feed:
 * /root/reference/src/retest/main.c:1114 hands fsm_runner_run one line at a time, reperf.c:772-784 one
 * string per run; include/fsm_hip.h batches them as base + off[n + 1]).
 *
 * walk_ragged gives every input a 128-byte lane slot per segment, so inputs of 8-64 bytes use a sixth of
 * the lanes' byte steps and pay the claim / record / result code once per input.  Here a lane owns BYTES,
 * not inputs:
 *  - the batch's byte range is cut into rows of R = 2^rshift bytes (128 B .. 1 KiB, 128-byte aligned
 *    addresses); a tile is 64 adjacent rows, one per lane, and every lane streams its row 128 bytes a
 *    round with eight 16-byte loads of one cache line: the access pattern of the fixed-stride kernels,
 *    whatever the inputs' lengths.  Tiles are dealt round-robin to the resident wavefronts;
 *  - the inputs that start in a tile are a contiguous index range, so their offsets are read coalesced,
 *    64 per instruction, and each sets one bit -- "an input starts at this byte" -- in a per-wave LDS
 *    bitmask of the tile, laid out [round][lane] so that a lane's 128 bits of a round are one
 *    conflict-free 16-byte piece.  All 64 lanes share this work evenly, whatever the lengths;
 *  - a lane walks every input that STARTS in its row, one after the other, across input boundaries: in a
 *    16-byte chunk whose mask bits are not all zero the chain is 16 x { bit k ? restart from the start
 *    state : keep, step } in straight-line code, the 16 states entered go to a lane-private LDS record,
 *    and every set bit k ends the lane's current input with the state before byte k and begins its next
 *    one.  Chunks without a bit in any lane take the policy's plain step16 (chunk skips included);
 *  - the bytes before the row's first bit belong to an input that started in an earlier row: the lane
 *    walks them without an input of its own, and the lane that does own that input runs on past the end
 *    of its row -- through its neighbours' mask pieces, and past the tile's end towards the one boundary
 *    that can still lie there -- until the input ends;
 *  - results leave as raw state codes (one fire-and-forget store per input: no lookup whose latency the
 *    walk would wait for); packed_finish maps them through fin[] / fin2[] and builds the bitmap with
 *    wave votes.  Empty inputs never reach the walk: packed_first writes their result (the start state's)
 *    and a bitmap of them, which a lane consults to step over their indices -- only if the batch has any.
 * packed_first (walk_packed_aux.h), run before the walk, also finds for every row the first input that
 * starts in it (one pass over off[], no search), picks R from the batch's size and decides from the mean
 * length whether this kernel or walk_ragged takes the batch -- on the device, so the device-pointer
 * fronts stay asynchronous; the kernel that is not chosen returns at once.
 *
 * Semantics are fsm_exec's (src/libfsm/exec.c:85-167) per input: start state, delta per byte, end
 * state -> fin[]; an empty input ends in the start state.
 */
#ifndef FSM_HIP_WALK_PACKED_H
#define FSM_HIP_WALK_PACKED_H

#include "walk_kernels.h"

namespace fsmhip {

struct PackedParams {
	uint64_t a0;        /* address of row 0's first byte: 128-byte aligned, <= base + off[0]                */
	uint64_t aend;      /* end of the 128-byte line that holds the batch's last byte: nothing at or beyond it is read */
	uint64_t nrows;
	uint32_t rshift;    /* rows are 1 << rshift bytes                                                        */
	uint32_t use;       /* 1: walk_packed takes the batch (walk_ragged / walk_generic return at once), 0: the reverse */
	uint32_t has_empty; /* some input is empty: result indices step over them (kbits)                        */
	uint32_t pad[7];
};
#define FSMHIP_PK_FIRST_OFF 16u   /* first[] starts at this u32 index of the scratch block (after the parameters) */
#define FSMHIP_PK_RMAX 10u        /* rows of at most 1 KiB: the tile's bitmask is 8 KiB of LDS */

/* how a policy's state code is kept in the lane's 16-state record: 16 bits (after a shift) where it fits */
template <class Pol> struct packed_code { static constexpr bool c16 = true; static constexpr uint32_t shift = 0u; };
template <> struct packed_code<LdsPol> { static constexpr bool c16 = true; static constexpr uint32_t shift = 2u; };     /* row byte offsets, multiples of 4, < 2^18 */
template <> struct packed_code<LdsSelfPol> { static constexpr bool c16 = true; static constexpr uint32_t shift = 2u; };
template <> struct packed_code<GlobPol> { static constexpr bool c16 = false; static constexpr uint32_t shift = 0u; };
template <> struct packed_code<SparsePol> { static constexpr bool c16 = false; static constexpr uint32_t shift = 0u; };

/* per-wave LDS: the tile's bitmask (64 rows x 2^rmax / 8 bytes) + per lane 16 codes and, just before them, the carry-in code */
__host__ __device__ constexpr uint32_t packed_wave_lds(uint32_t rmax, bool c16) { return (8u << rmax) + 64u * (c16 ? 48u : 80u); }

typedef uint32_t u32x32 __attribute__((ext_vector_type(32)));

/* m = all ones: restart from the start state; m = 0: keep (one v_bfi_b32 per 32-bit field: no compare, no VCC) */
__device__ __forceinline__ uint32_t pk_reset(uint32_t m, uint32_t start, uint32_t st) { return (start & m) | (st & ~m); }
__device__ __forceinline__ LdsSelfState pk_reset(uint32_t m, const LdsSelfState &start, const LdsSelfState &st)
{
	LdsSelfState r = { pk_reset(m, start.st, st.st), pk_reset(m, start.sm, st.sm) };
	return r;
}
__device__ __forceinline__ CombSelfState pk_reset(uint32_t m, const CombSelfState &start, const CombSelfState &st)
{
	CombSelfState r = { pk_reset(m, start.st, st.st), pk_reset(m, start.sm, st.sm), pk_reset(m, start.rng, st.rng) };
	return r;
}

/* the index of the first non-empty input at or after i (kbits: bit j set = input j is empty; bits at and beyond n are clear) */
__device__ __forceinline__ uint32_t packed_skip_empty(const uint64_t *kbits, uint32_t i)
{
	for (;;) {
		const uint64_t w = ~(kbits[i >> 6] >> (i & 63u));   /* ones: non-empty inputs (and the zeros shifted in) */
		const uint32_t room = 64u - (i & 63u);
		const uint64_t live = room == 64u ? w : w & (((uint64_t)1 << room) - 1u);
		if (live != 0u) return i + (uint32_t)__builtin_ctzll(live);
		i += room;
	}
}

template <class Pol, int MAXT>
__global__ void __launch_bounds__(MAXT)
walk_packed(const WalkArgs a)
{
	constexpr bool C16 = packed_code<Pol>::c16;
	constexpr uint32_t CSH = packed_code<Pol>::shift;
	constexpr uint32_t CW = C16 ? 48u : 80u, NONE = 0xFFFFFFFFu;   /* record: the carry-in code just before offset 16, the codes after bytes 0..15 from offset 16 */
	typedef typename Pol::S S;
	typedef const u32x4 __attribute__((address_space(1))) *glb_chunk_p;

	const PackedParams *pp = reinterpret_cast<const PackedParams *>(a.pk);
	if (pp->use == 0u) return;

	extern __shared__ __align__(16) unsigned char lds[];
	Pol pol;
	pol.setup(lds, a);
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const uint32_t mask_bytes = 8u << a.pk_rmax;
	unsigned char *mk = lds + Pol::lds_bytes(a.tab_bytes) + wave * (mask_bytes + 64u * CW);   /* [round][lane][16 bytes] */
	unsigned char *cs = mk + mask_bytes + lane * CW;                                            /* this lane's saved codes: C16: code k at cs + 14 + 2 k, else cs + 12 + 4 k (k = 0: the carry-in code) */

	const S start_s = init_state(pol, a.start, a, 0, false, 0);
	const uint64_t A0 = pp->a0, Aend = pp->aend, nrows = pp->nrows;
	const uint32_t rsh = pp->rshift, rpr = 1u << (rsh - 7u);      /* rounds per row */
	const bool emp = pp->has_empty != 0u;
	const uint32_t *first_tab = a.pk + FSMHIP_PK_FIRST_OFF;
	const uint64_t *kbits = a.pk_kbits;
	uint32_t *codes_out = a.pk_codes;
	const uint64_t ntiles = (nrows + 63u) / 64u, base = reinterpret_cast<uint64_t>(a.base);
	const uint32_t tile_bytes = 64u << rsh;

	/* tiles are dealt round-robin to the resident wavefronts; a tile's first[] entries are asked for one tile ahead */
	const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6), gw = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
	uint32_t fst_n = 0, lim_n = 0;
	{
		const uint64_t v = gw * 64u + lane;
		fst_n = first_tab[v < nrows ? v : nrows];
		lim_n = first_tab[v + 1u < nrows ? v + 1u : nrows];
	}
	for (uint64_t tile = gw; tile < ntiles; tile += nwaves) {
		const uint32_t fst = fst_n, lim = lim_n;
		{
			const uint64_t v = (tile + nwaves) * 64u + lane;     /* (unconditional loads of clamped indices: nothing to wait for here) */
			fst_n = first_tab[v < nrows ? v : nrows];
			lim_n = first_tab[v + 1u < nrows ? v + 1u : nrows];
		}
		const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)fst), e1 = (uint32_t)__builtin_amdgcn_readlane((int)lim, 63);
		const uint64_t tileaddr = A0 + (tile << (rsh + 6u)), tileoff = tileaddr - base;   /* offsets relative to the tile (wraps for tile 0: fine) */
		const uint64_t rowaddr = tileaddr + ((uint64_t)lane << rsh);


		/* the tile's bitmask: bit p = an input starts at byte p of the tile (the end of the batch's last input counts as one) */
		for (uint32_t k = 0; k < rpr; k++) *reinterpret_cast<u32x4 *>(mk + ((k << 6) + lane) * 16u) = u32x4{0u, 0u, 0u, 0u};
		/* the only boundary that can lie beyond the tile and still matter: the end of the last input that starts in it */
		const uint64_t tail = a.off[e1] - tileoff;
		for (uint32_t e = e0; e <= e1; e += 512u) {   /* 8 x 64 offsets in flight (e1 < 2^32 - 4096: the host checks n) */
			uint64_t o[8];
#pragma unroll
			for (uint32_t q = 0; q < 8; q++) {
				const uint32_t ei = e + 64u * q + lane;
				o[q] = a.off[ei <= e1 ? ei : e1];                 /* unconditional: a load under a lane mask is waited for on the spot */
			}
#pragma unroll
			for (uint32_t q = 0; q < 8; q++) {
				const uint64_t rel = o[q] - tileoff;
				if (e + 64u * q + lane <= e1 && rel < tile_bytes) {
					const uint32_t p = (uint32_t)rel;
					uint32_t *word = reinterpret_cast<uint32_t *>(mk + ((((p >> 7) & (rpr - 1u)) << 6) + (p >> rsh)) * 16u + ((p >> 5) & 3u) * 4u);
					atomicOr(word, 1u << (p & 31u));      /* ds_or_b32, no return value */
				}
			}
		}
		__builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0) */
		__asm__ volatile("" ::: "memory");
		__builtin_amdgcn_wave_barrier();

		/* Which input is the lane in?  rel = how many of its own inputs have begun (0: it is in the bytes before its row's
		 * first input), nl = how many it owns; the r-th set bit met ends own input number rel + r (if 1 <= rel + r <= nl: index
		 * fst + rel + r - 1) and begins the next one; the lane is done once an input beyond its own has begun.  With empty
		 * inputs in the batch the indices are not consecutive: cur / nxt step over them through kbits instead. */
		bool act = fst < lim;
		const uint32_t nl = lim - fst;
		uint32_t rel = 0;
		uint32_t cur = NONE, nxt = fst;
		if (emp && act) nxt = packed_skip_empty(kbits, nxt);
		S st = start_s;

		for (uint64_t rpos = 0; __any(act); rpos += 128u) {
			const uint64_t ra = rowaddr + rpos;
			act = act && ra <= Aend;                  /* (only malformed offsets get here: every input ends at or before the batch's end) */
			u32x32 w;
#pragma unroll
			for (uint32_t j = 0; j < 8; j++) {
				u32x4 x = {0u, 0u, 0u, 0u};
				if (act && ra + 16u * j < Aend && !(a.pk_debug & 4u)) x = *(glb_chunk_p)(ra + 16u * j);
				w[4 * j] = x.x; w[4 * j + 1] = x.y; w[4 * j + 2] = x.z; w[4 * j + 3] = x.w;
			}
			__builtin_amdgcn_s_waitcnt(0x0F70);   /* vmcnt(0), once per round: the chunk loop below must not wait (for its own result stores) on every chunk */
			__asm__ volatile("" ::: "memory");
			/* this round's mask piece: the lane's own row, a neighbour's once it runs past its row's end, none beyond the tile */
			const uint64_t rr = lane + (rpos >> rsh);
			const bool inside = rr < 64u;
			const unsigned char *piece = mk + (((((uint32_t)rpos >> 7) & (rpr - 1u)) << 6) + (inside ? (uint32_t)rr : 0u)) * 16u;
			/* beyond the tile: the distance from this round's first byte to the tail boundary (saturated) */
			const bool beyond = __any(act && !inside);
			uint32_t tb = 0xFFFFFFFFu;
			if (beyond) {
				const uint64_t d = tail - (((uint64_t)lane << rsh) + rpos);
				if (!inside && d < 0xFFFFFFFFull) tb = (uint32_t)d;
			}
			for (uint32_t c = 0; c < 8u; c++) {
				const u32x4 wc = {w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]};
				uint32_t bm = inside ? (uint32_t)*reinterpret_cast<const uint16_t *>(piece + 2u * c) : 0u;
				if (beyond) {
					const uint32_t tx = tb - 16u * c;     /* wraps (no bit) once the tail lies before this chunk */
					if (tx < 16u) bm |= 1u << tx;
				}
				if (!act) bm = 0u;
				if (!__any(bm != 0u)) {
					/* no input ends or starts in this chunk in any lane: the policy's plain chunk step */
					S s1[1] = { st };
					const u32x4 w1[1] = { wc };
					step16<Pol, 1>(pol, s1, w1);
					st = s1[0];
					continue;
				}
				typename Pol::P pre[16];
#pragma unroll
				for (int k = 0; k < 16; k++) pre[k] = pre_of(pol, wc, k, 0);
				/* the state before the chunk's first byte is what an input that ends there ends in: it sits just before the 16 codes */
				const uint32_t prevc = Pol::code(st) >> CSH;
				if (C16) *reinterpret_cast<uint16_t *>(cs + 14) = (uint16_t)prevc;
				else *reinterpret_cast<uint32_t *>(cs + 12) = prevc;
				uint32_t cd[16];
#pragma unroll
				for (int k = 0; k < 16; k++) {
					uint32_t rm = (uint32_t)__builtin_amdgcn_sbfe((int)bm, k, 1);   /* bit k of bm, sign-extended: v_bfe_i32 */
					__asm__("" : "+v"(rm));   /* (no instruction: it only hides that rm is 0 or ~0, which turns the v_bfi_b32 below into a shift, a compare and a select) */
					st = pk_reset(rm, start_s, st);
					st = pol.next(st, pre[k]);
					cd[k] = Pol::code(st) >> CSH;
				}
				if (C16) {
					const u32x4 lo = {cd[0] | (cd[1] << 16), cd[2] | (cd[3] << 16), cd[4] | (cd[5] << 16), cd[6] | (cd[7] << 16)};
					const u32x4 hi = {cd[8] | (cd[9] << 16), cd[10] | (cd[11] << 16), cd[12] | (cd[13] << 16), cd[14] | (cd[15] << 16)};
					*reinterpret_cast<u32x4 *>(cs + 16) = lo;
					*reinterpret_cast<u32x4 *>(cs + 32) = hi;
				} else {
#pragma unroll
					for (int q = 0; q < 4; q++) {
						const u32x4 x = {cd[4 * q], cd[4 * q + 1], cd[4 * q + 2], cd[4 * q + 3]};
						*reinterpret_cast<u32x4 *>(cs + 16 + 16 * q) = x;
					}
				}
				/* every set bit k: the lane's input (if it is in one) ends with the state before byte k; its next one begins */
				uint32_t m = bm;
				if (!emp) {
					const uint32_t rm1 = rel - 1u;        /* bit r ends own input number rel + r, at index fst + rm1 + r */
					for (uint32_t r = 0; __any(m != 0u); r++) {
						if (m != 0u) {
							const uint32_t k = (uint32_t)__builtin_ctz(m);
							m &= m - 1u;
							const uint32_t code = C16 ? (uint32_t)*reinterpret_cast<const uint16_t *>(cs + 14u + k * 2u)
							                          : *reinterpret_cast<const uint32_t *>(cs + 12u + k * 4u);
							if (rm1 + r < nl && !(a.pk_debug & 1u)) codes_out[fst + rm1 + r] = code << CSH;   /* (rel + r = 0: the bytes before the first own input) */
						}
					}
					rel += (uint32_t)__builtin_popcount(bm);
					act = act && rel <= nl;
				} else {
					while (__any(m != 0u)) {
						if (m != 0u) {
							const uint32_t k = (uint32_t)__builtin_ctz(m);
							m &= m - 1u;
							const uint32_t code = C16 ? (uint32_t)*reinterpret_cast<const uint16_t *>(cs + 14u + k * 2u)
							                          : *reinterpret_cast<const uint32_t *>(cs + 12u + k * 4u);
							if (cur != NONE && !(a.pk_debug & 1u)) codes_out[cur] = code << CSH;
							cur = nxt;
							nxt = nxt + 1u;
							if (cur >= lim) {             /* that input starts in a later row: this lane is done */
								cur = NONE;
								act = false;
								m = 0u;
							} else {
								nxt = packed_skip_empty(kbits, nxt);
							}
						}
					}
				}
				if (!__any(act)) break;
			}
		}
	}
}

} // namespace fsmhip

#endif
