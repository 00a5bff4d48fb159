/*
 * walk_kernels.h -- gfx950 device code: the batched DFA walk.
 *
 * One lane = one input string, 64 inputs per wavefront advancing in lockstep
 * against the same DFA (the data-parallel form of the reference's per-input
 * loop `while ((c = getc()) != EOF) state = delta(state, c)`,
 * src/libfsm/exec.c:132-151).  No MFMA: there is no contraction here; the
 * kernels are bound by HBM input bandwidth (tiny / combself), by the latency of
 * the dependent lookup chain (the generic LDS layouts) or by the instructions of
 * a divergent chain walk (sparse).
 *
 * Table policies (how delta(state, byte) is evaluated).  Each policy splits a
 * step into  pre(byte)  -- independent of the state, so all 16 of a 16-byte
 * chunk are issued together --  and  next(state, pre)  -- the dependent chain:
 *   TinyPol<uint64_t> 7..16 states.  LDS holds, per byte value, the whole transition COLUMN (4-bit next
 *              states packed in one word), one copy per LDS bank: conflict-free by construction.
 *              pre = the column, next = one 64-bit shift: the state chain never touches memory.
 *   Tiny5Pol   <= 6 states: 5-bit fields of 5 * next state, one private column copy per lane at LDS
 *              address (byte << 8) | (lane << 2): one v_perm_b32 + one v_bfe_u32 per input byte.
 *   LdsPol     class-compressed dense table T[state][class] (u16) in LDS plus a
 *              256-byte byte->class map (conflict-free for 7-bit text, <= 2-way otherwise).
 *   CombPol    column-default + comb exceptions over byte CLASSES, see plan.cpp.
 *   Comb256Pol comb exceptions over raw BYTES with one default state for every column (typically
 *              DEAD): one LDS lookup per input byte; the state is the raw entry (next << 16 | owner).
 *   CombSelfPol CombPol + the current state's self-loop class mask and self-loop byte range in
 *              registers: whole chunks of self-loops are skipped with one wave vote.
 *   LdsSelfPol LdsPol + the current state's self-loop mask in a register (rows carry their mask).
 *   GlobPol    T[state][class] (u32) in HBM/L2, class map in LDS (+ LDS mirror of the table's head).
 *   SparsePol  per-state record {exception bitmap, base state | flags, offset} in HBM/L2, the
 *              records nearest the start state in LDS (failure-link form of big tables).
 * Wrappers: EagerPol (<= 64 eager-output ids, set in registers), EagerWidePol (more: set in memory).
 * Optional policy hooks, found by SFINAE: pre_dw (lookup from the raw input dword), skip16_raw (skip a
 * chunk after a test on its raw bytes), skip16 (the same after the class lookups), walk16 (a whole chunk
 * at once), init_at / finish_at (input index).
 *
 * Kernels.  Uniform-length, 16-byte aligned rows:
 *   walk_ldsdma  rows are fetched as whole 64- or 128-byte segments (4 / 8 adjacent lanes per row) by
 *                global_load_lds_dwordx4 straight into a 4 / 8 KiB per-wave LDS tile, piece-rotated so
 *                that the row-per-lane ds_read_b128 that follows is bank-conflict-free; no VGPR
 *                staging, no ds_write.
 *   walk_direct  every lane reads its own row 16 bytes at a time, NB = 4 or 8 chunks in flight;
 *   walk_direct_np  the same without the register double-buffer (<= 64 VGPRs, occupancy instead).
 * Any length / alignment / packed offsets:
 *   walk_ragged  the LDS-DMA input path with per-lane source addresses + lane refill per segment;
 *   walk_generic per-lane 16-byte loads (fallback, and the better one for very short inputs);
 *   walk_lines32 the same walk in 32 bits for plain walks of packed batches below 4 GiB (round 5: the short-lines kernel).
 * Where an input lies: fixed stride (+ lengths), u64 or u32 offsets, or lengths alone (packed back to back: per-tile
 * bases from a small pre-pass + a wavefront prefix sum).
 */
#ifndef FSM_HIP_WALK_KERNELS_H
#define FSM_HIP_WALK_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace fsmhip {

struct WalkArgs {
	const uint8_t  *base;     /* device input bytes                                   */
	uint64_t        stride;   /* bytes between inputs (fixed-stride modes)            */
	const uint32_t *len;      /* per-input lengths or NULL (= stride)                 */
	const uint64_t *off;      /* packed mode: n+1 offsets, or NULL                    */
	const uint32_t *off32;    /* packed mode, batches below 4 GiB: n+1 32-bit offsets, or NULL */
	const uint64_t *tbase;    /* packed mode, lengths only (len != NULL, stride == 0): byte offset of input 64 t for every
	                           * tile t of 64 inputs (tile_bases_* below); the offsets inside a tile are a wavefront prefix sum */
	uint64_t        n;        /* number of inputs                                     */
	uint32_t       *end_out;  /* n entries or NULL                                    */
	uint64_t       *bitmap;   /* ceil(n/64) words or NULL                             */
	const void     *tab;      /* policy-specific table image in device memory         */
	const uint32_t *fin;      /* policy-specific: encoded state -> caller id/NO_MATCH */
	const uint32_t *btab;     /* [256] byte -> B-table entry (policy-specific)        */
	uint32_t        tab_bytes;
	uint32_t        start;    /* encoded start state                                  */
	uint32_t        abs_min;  /* encoded states >= abs_min are absorbing              */
	uint32_t        fin_div;  /* fin index = encoded state / fin_div                  */
	uint32_t        fin_mul;  /* 0, or floor(2^32 / fin_div) + 1: then encoded state * fin_mul >> 32 is that quotient (fin_index()) */
	uint32_t        ident_class; /* self-loop-mask layouts with <= 31 classes: 31, the class that is a self-loop of EVERY state
	                           * (what the bytes beyond an input's end are given: step16_part); else >= 32 */
	uint32_t        early;    /* bit 0: retire a wavefront once every lane is absorbing;
	                           * bit 1: absorbing lanes stop loading their input;
	                           * bit 2: never skip a chunk (skip16 off: measurement aid);
	                           * bit 3: no absorbing-lane masking; bit 4: walk_generic always asks for four chunks;
	                           * bit 5 (32): never the 32-bit lines kernel (walk_lines32): walk_generic's own body, as batches
	                           *   of 4 GiB and more take it; bit 6 (64): walk_lines32 keeps the skip tests on an input's
	                           *   first chunk; bit 7 (128): the per-lane loads' resource ends at total - 8 (round 4's bound:
	                           *   loses bytes, kept for the test that pins the hardware's range rule) */
	uint32_t        dflt;     /* Comb256Pol: encoded default state                    */
	const uint32_t *fin2;     /* optional second per-state table (end-id / ret index) */
	uint32_t       *out2;     /* n entries, written from fin2, or NULL                */
	/* resume (streaming): state_io[i] = caller's state id to start from (or START / DEAD), and
	 * receives the state reached; enc_of[nstates+1] maps caller ids (+DEAD) to encoded states,
	 * orig_of[] (indexed like fin) maps back. */
	uint32_t       *state_io;
	const uint32_t *enc_of;
	const uint32_t *orig_of;
	uint32_t        nstates;
	/* eager outputs: emask indexed like fin; has-eager test on the encoded state */
	const uint64_t *emask;
	uint64_t       *eager_out;
	uint32_t        eager_lo_end, eager_hi_begin;
	/* more than 64 eager ids: eager_out holds eager_words u64 per input; a state's ids are the
	 * (word, mask) pairs ew_word/ew_mask[ew_off[idx] .. ew_off[idx+1]) with idx as for fin */
	uint32_t        eager_words;
	const uint32_t *ew_off;
	const uint32_t *ew_word;
	const uint64_t *ew_mask;
	/* several kernels launched for one batch, the choice made on the device: return at once unless *skip_flag == run_when */
	const uint32_t *skip_flag;
	uint32_t        run_when;
	uint32_t       *pick_flag;   /* offsets_pick writes PICK_RAGGED, PICK_GENERIC or PICK_LINES32 here */
	/* sparse layout, lazy form (walk_lazy.h): the image of plan.cpp build_lazy; a zeroed tile counter, or NULL */
	const void     *lazy;
	uint32_t       *tile_ctr;
	uint32_t        lds_bytes;   /* dynamic LDS of this launch (launch.h launch_fn fills it in): walk_lazy_lines sizes its queues by what lies behind the table */
};

#define FSMHIP_STATE_START 0xFFFFFFFDu
#define FSMHIP_STATE_DEAD  0xFFFFFFFCu

/* encoded state input i starts from */
__device__ __forceinline__ uint32_t start_code(const WalkArgs &a, uint64_t i, bool valid)
{
	if (a.state_io == nullptr || !valid) return a.start;
	const uint32_t sid = a.state_io[i];
	if (sid == FSMHIP_STATE_START) return a.start;
	return a.enc_of[sid == FSMHIP_STATE_DEAD || sid >= a.nstates ? a.nstates : sid];
}

/* encoded state -> index of the per-state tables.  A division by a run-time value is ~30 vector instructions: nothing next
 * to a 1 KiB input, 8 % of a 36-byte one.  Most layouts index by the code itself; the others by a multiply the host prepared
 * (exact while code * fin_div < 2^32, which it checks) */
__device__ __forceinline__ uint32_t fin_index(const WalkArgs &a, uint32_t code)
{
	if (a.fin_div == 1u) return code;
	if (a.fin_mul != 0u) return __umulhi(code, a.fin_mul);
	return code / a.fin_div;
}

/* one past the last byte of a batch, relative to its base */
__device__ __forceinline__ uint64_t batch_bytes(const WalkArgs &a)
{
	if (a.off != nullptr) return a.off[a.n];
	if (a.off32 != nullptr) return a.off32[a.n];
	if (a.tbase != nullptr) return a.tbase[(a.n + 63u) / 64u];
	return a.n * a.stride;
}

/* exclusive prefix sum over the 64 lanes of a wavefront (the lengths-only front: where an input starts inside its tile) */
__device__ __forceinline__ uint64_t wave_excl_prefix(uint32_t v, uint32_t lane)
{
	uint64_t x = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint64_t y = __shfl_up(x, d, 64);
		if (lane >= (uint32_t)d) x += y;
	}
	return x - v;
}

/* the same in 32 bits by data-parallel primitives: seven adds whose second operand comes from another lane of the row / the
 * row before (row_shr 1, 2, 3; 4 and 8 on the banks that have such a lane; the last lane of row 0 / 2 broadcast into row 1 / 3;
 * lane 31 into rows 2 and 3) -- the shuffle form above is six LDS-path permutes of 64 bits each.  (walk_lines32's lengths front:
 * a batch below 4 GiB.) */
__device__ __forceinline__ uint32_t wave_excl_prefix32(uint32_t v)
{
	uint32_t s = v;
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   /* row_shr:1 */
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   /* row_shr:2 */
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, false);   /* row_shr:3 */
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xe, false);   /* row_shr:4, lanes 4..15 of a row */
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xc, false);   /* row_shr:8, lanes 8..15 */
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, false);   /* row_bcast:15 into rows 1 and 3 */
	s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, false);   /* row_bcast:31 into rows 2 and 3 */
	return s - v;
}

enum { PICK_RAGGED = 0, PICK_GENERIC = 1, PICK_LINES32 = 2 };
enum { IN_DIRECT = 0, IN_LDSDMA = 1, IN_GENERIC = 2, IN_RAGGED = 3, IN_LAZY = 5, IN_LAZY_LINES = 6 };   /* (4 was walk_packed: removed in round 4) */

#define FSMHIP_NO_MATCH 0xFFFFFFFFu
#define FSMHIP_BTAB_BYTES 256u

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t byte_of(const u32x4 &w, int k)
{
	const uint32_t d = (k < 4) ? w.x : (k < 8) ? w.y : (k < 12) ? w.z : w.w;
	return (d >> (8 * (k & 3))) & 0xffu;
}

/* ------------------------------------------------------------------ */
/* table policies                                                     */
/* ------------------------------------------------------------------ */

template <class W>
struct TinyPol {
	static constexpr bool heavy_next = false;
	static_assert(sizeof(W) == 8, "16 states x 4 bits per column");
	typedef W P;
	typedef uint32_t S;   /* carried unmasked: only bits 3:0 are the state (see next) */
	__device__ __forceinline__ S init(uint32_t code) const { return code; }
	__device__ __forceinline__ static uint32_t code(S s) { return s & 15u; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, S) {}
	const W *colp; /* LDS column table, already offset by lane%32 */

	__host__ __device__ static uint32_t lds_bytes(uint32_t) { return 256u * 32u * (uint32_t)sizeof(W); }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		W *col = reinterpret_cast<W *>(lds);
		const uint64_t *src = static_cast<const uint64_t *>(a.tab);
		for (uint32_t i = threadIdx.x; i < 256u * 32u; i += blockDim.x) col[i] = (W)src[i >> 5];
		colp = col + (threadIdx.x & 31u);
		/* pre_dw forms LDS addresses by permutation: the table must start at LDS address 0 */
		if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds != 0u) __builtin_trap();
	}
	__device__ __forceinline__ P pre(uint32_t b) const { return colp[b * 32u]; }
	/* a column row is 32 x 8 = 256 bytes, so -- the table starting at LDS address 0 -- the address is
	 * the byte permutation [0, 0, input byte, (lane % 32) << 3]: one v_perm_b32 */
	__device__ __forceinline__ P pre_dw(uint32_t d, int k) const
	{
		typedef const uint64_t __attribute__((address_space(3))) *lds_u64p;
		const uint32_t sel = 0x0c0c0400u + ((uint32_t)k << 8);
		return *(lds_u64p)(uintptr_t)__builtin_amdgcn_perm(d, (threadIdx.x & 31u) << 3, sel);
	}
	__device__ __forceinline__ uint32_t next(uint32_t st, P v) const
	{
		/* One 64-bit shift; it uses bits 5:0 of its amount, so the state is carried unmasked (code() masks it):
		 * 2 operations per byte.
		 * History of this line.  Round 1 saw wrong states in ~45 % of 16-wavefront launches, blamed the compiler's
		 * register allocation (destination overlapping the shift-amount register) and replaced the C++ shift by an
		 * inline-asm v_lshrrev_b64 with an early-clobber destination.  Round 2 found that asm form returning a wrong
		 * state in ~3 % of launches of one build of walk_ragged<EagerPol<TinyPol>> (only with a lane-divergent branch
		 * around it, only in the second wavefront of a SIMD), while the SAME build with the C++ shift below passed --
		 * with its destination overlapping the shift amount in 875 of 1 360 instances -- as did round 1's reproducer
		 * and 91 000 stress launches.  An inline-asm VALU instruction is opaque to the compiler
		 * (no hazard model, no EXEC dependence); the plain shift is not.  The asm is gone. */
		const uint32_t sh = st << 2;
		const uint64_t t = (uint64_t)v >> (sh & 63u);
		return (uint32_t)t;
	}
};

/*
 * byte -> class map: 256 u8 entries = 64 LDS dwords over 32 banks.  Lanes reading different bytes of
 * one dword are served by a broadcast, and only dwords d and d+32 (bytes b and b+128) share a bank,
 * so a lookup is conflict-free for 7-bit text and at most 2-way for arbitrary bytes -- a 256-byte
 * table does what a bank-private [256][32] copy (32 KiB) was first used for.
 */
__device__ __forceinline__ const uint8_t *setup_btab(unsigned char *lds, const WalkArgs &a)
{
	for (uint32_t i = threadIdx.x; i < 64u; i += blockDim.x) {
		const uint32_t k = 4u * i;
		reinterpret_cast<uint32_t *>(lds)[i] = (a.btab[k] & 0xffu) | ((a.btab[k + 1] & 0xffu) << 8) |
			((a.btab[k + 2] & 0xffu) << 16) | ((a.btab[k + 3] & 0xffu) << 24);
	}
	return lds;
}

__device__ __forceinline__ void copy_table(unsigned char *dst, const WalkArgs &a)
{
	uint32_t *T = reinterpret_cast<uint32_t *>(dst);
	const uint32_t *src = static_cast<const uint32_t *>(a.tab);
	for (uint32_t i = threadIdx.x; i < (a.tab_bytes + 3u) / 4u; i += blockDim.x) T[i] = src[i];
}

/*
 * Tiny5Pol: <= 6 states (C1/C2: 5 + DEAD).  Two VALU operations per input byte instead of four:
 *  - the column of byte b sits at LDS byte address (b << 8) | (lane << 2) (one private copy per
 *    lane: conflict-free), so ONE v_perm_b32 builds the address from the raw input dword;
 *  - a column packs, for every state s, 5 * next(s) in the 5-bit field at bit 5 * s, and the state
 *    is carried as 5 * s: the dependent chain is ONE v_bfe_u32 per byte.
 */
struct Tiny5Pol {
	static constexpr bool heavy_next = false;
	typedef uint32_t P;
	typedef uint32_t S;
	__device__ __forceinline__ S init(uint32_t code) const { return code; }
	__device__ __forceinline__ static uint32_t code(S s) { return s; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, S) {}
	typedef const uint32_t __attribute__((address_space(3))) *lds_u32p;
	uint32_t lanebase;         /* LDS byte address of this lane's copy of column 0: table base + (lane << 2) */

	__host__ __device__ static uint32_t lds_bytes(uint32_t) { return 256u * 64u * 4u; }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		uint32_t *col = reinterpret_cast<uint32_t *>(lds);
		const uint32_t *src = static_cast<const uint32_t *>(a.tab);
		for (uint32_t i = threadIdx.x; i < 256u * 64u; i += blockDim.x) col[i] = src[i >> 6];
		/* addresses are formed by byte permutation, not addition: the table must start at LDS
		 * address 0 (these kernels have no static LDS, the dynamic segment is the whole of it) */
		const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
		if (base != 0u) __builtin_trap();
		lanebase = (threadIdx.x & 63u) << 2;
	}
	/* lanes 32-63 read copies 0-31 (a ds_read_b32 wave is served in two 32-lane groups: no conflict between l and l + 32):
	 * the upper half of every 256-byte row is then never read -- walk_ragged keeps its ring there (ragged_aux_in_holes) */
	__device__ __forceinline__ void half_copies() { lanebase &= 0x7Cu; }
	__device__ __forceinline__ P pre(uint32_t b) const { return *(lds_u32p)(uintptr_t)((b << 8) | lanebase); }
	/* byte k of the input dword d -> address bytes [0, 0, d.byte[k], lanebase.byte[0]] */
	__device__ __forceinline__ P pre_dw(uint32_t d, int k) const
	{
		const uint32_t sel = 0x0c0c0400u + ((uint32_t)k << 8);
		return *(lds_u32p)(uintptr_t)__builtin_amdgcn_perm(d, lanebase, sel);
	}
	__device__ __forceinline__ uint32_t next(uint32_t st, P v) const { return __builtin_amdgcn_ubfe(v, st, 5u); }
};

/*
 * The lookup layouts without a self-loop mask (LdsPol, CombPol, Comb256Pol) are bound by the LDS array, not by latency:
 * a wave's table read has 64 random addresses and is replayed for every bank conflict (profiles/r03r_pmc_c3t_comb256.txt:
 * 6.3 LDS cycles per wave read where 2 are the conflict-free cost; SQ_LDS_IDX_ACTIVE = 81 % of the kernel's cycles per CU).
 * A lane that has reached an absorbing state can no longer change state (fsm_exec stops pulling bytes at a missing edge,
 * exec.c:133-138) but its reads still take part in those replays.  mask_absorbing(): such a lane sees its chunk as 16
 * zero bytes -- delta(absorbing, b) is the state itself for every b -- so all the absorbing lanes of a wave read ONE
 * address per state (a broadcast, no conflict).  One compare + four ANDs per chunk, nothing on the per-byte chain.  (A
 * per-byte `if (absorbing) skip the read` was measured first: the exec-mask branches cost 27 %, profiles/r04d_absorbing_branch_per_byte_slower.txt.)
 * (a.early & 8) switches the masking off for A/B runs.
 */
__device__ __forceinline__ uint32_t absorbing_limit(const WalkArgs &a) { return (a.early & 8u) ? 0xFFFFFFFFu : a.abs_min; }

struct LdsPol {
	static constexpr bool heavy_next = false;
	typedef uint32_t P;
	typedef uint32_t S;
	__device__ __forceinline__ S init(uint32_t code) const { return code; }
	__device__ __forceinline__ static uint32_t code(S s) { return s; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, S) {}
	const uint8_t *bp;         /* LDS byte -> class map                           */
	const unsigned char *tab;  /* LDS table; state is a byte offset into it      */
	uint32_t abs_min;

	__host__ __device__ static uint32_t lds_bytes(uint32_t tab_bytes) { return FSMHIP_BTAB_BYTES + ((tab_bytes + 15u) & ~15u); }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		bp = setup_btab(lds, a);
		copy_table(lds + FSMHIP_BTAB_BYTES, a);
		tab = lds + FSMHIP_BTAB_BYTES;
		abs_min = absorbing_limit(a);
	}
	__device__ __forceinline__ bool absorbing(S s) const { return s >= abs_min; }
	__device__ __forceinline__ P pre(uint32_t b) const { return bp[b]; }
	__device__ __forceinline__ uint32_t next(uint32_t st, P c) const
	{
		return (uint32_t)(*reinterpret_cast<const uint16_t *>(tab + st + c * 2u)) << 2;
	}
};

/*
 * Lds2Pol: the dense table over PAIRS of byte classes (plan.cpp emit_lds2): T[state][c1 * C1 + c2] = the state two bytes on,
 * as a row index in entries.  A 16-byte chunk is 8 lookups instead of 16 -- the lookup layouts are bound by the LDS array,
 * whose 64 random reads per wave are replayed for every bank conflict -- and the dependent chain per TWO bytes is one
 * add-shift and one ds_read_u16.  Class C ("no byte") is the identity: it serves a lone byte (next()) and the bytes beyond
 * an input's end (step16_part_ident).
 */
struct Lds2Pol {
	static constexpr bool heavy_next = false;
	typedef uint32_t P;          /* byte class */
	typedef uint32_t S;          /* row index in entries: state * C1 * C1 */
	__device__ __forceinline__ S init(uint32_t code) const { return code; }
	__device__ __forceinline__ static uint32_t code(S s) { return s; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, S) {}
	const uint8_t *bp;           /* LDS byte -> class map */
	const unsigned char *tab;    /* LDS table */
	uint32_t C1, ident, abs_min;

	__host__ __device__ static uint32_t lds_bytes(uint32_t tab_bytes) { return FSMHIP_BTAB_BYTES + ((tab_bytes + 15u) & ~15u); }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		bp = setup_btab(lds, a);
		copy_table(lds + FSMHIP_BTAB_BYTES, a);
		tab = lds + FSMHIP_BTAB_BYTES;
		C1 = a.dflt;             /* C + 1 */
		ident = C1 - 1u;
		abs_min = absorbing_limit(a);
	}
	__device__ __forceinline__ bool ident_ok() const { return true; }
	__device__ __forceinline__ bool absorbing(S s) const { return s >= abs_min; }
	__device__ __forceinline__ P pre(uint32_t b) const { return bp[b]; }
	__device__ __forceinline__ S pair(S st, P c1, P c2) const
	{
		return *reinterpret_cast<const uint16_t *>(tab + ((st + (c1 * C1 + c2)) << 1));
	}
	__device__ __forceinline__ S next(S st, P c) const { return pair(st, c, ident); }
	__device__ __forceinline__ void walk16(S &st, const P (&pre)[16]) const
	{
#pragma unroll
		for (int k = 0; k < 16; k += 2) st = pair(st, pre[k], pre[k + 1]);
	}
};

/*
 * LdsSelfPol: LdsPol plus the self-loop mask of the current state in a register (the planner stores it
 * after each row).  Bytes whose class is a self-loop of the state cost no table lookup, whole chunks of
 * them are skipped with one wave vote (skip16), and -- unlike the comb layouts -- states keep their
 * order, so EagerPol can wrap it.  <= 32 classes.
 */
struct LdsSelfState {
	uint32_t st;   /* byte offset of the row */
	uint32_t sm;   /* bit c: class c maps the state to itself */
};

struct LdsSelfPol {
	static constexpr bool heavy_next = false;
	typedef uint32_t P;
	typedef LdsSelfState S;
	const uint8_t *bp;
	const unsigned char *tab;
	uint32_t smoff;   /* offset of the mask inside a row */
	uint32_t ident;   /* WalkArgs::ident_class */
	uint32_t start_sm; /* the start state's mask */
	bool skip_on;

	__host__ __device__ static uint32_t lds_bytes(uint32_t tab_bytes) { return FSMHIP_BTAB_BYTES + ((tab_bytes + 15u) & ~15u); }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		bp = setup_btab(lds, a);
		copy_table(lds + FSMHIP_BTAB_BYTES, a);
		tab = lds + FSMHIP_BTAB_BYTES;
		smoff = a.fin_div - 4u;   /* fin_div = row bytes */
		ident = a.ident_class;
		skip_on = !(a.early & 4u);
		start_sm = *reinterpret_cast<const uint32_t *>(static_cast<const unsigned char *>(a.tab) + a.start + smoff);
	}
	__device__ __forceinline__ bool ident_ok() const { return ident < 32u; }
	__device__ __forceinline__ bool first_noskip() const { return ident == 31u && (start_sm & 0x7FFFFFFFu) == 0u; }
	__device__ __forceinline__ uint32_t mask_of(uint32_t st) const { return *reinterpret_cast<const uint32_t *>(tab + st + smoff); }
	__device__ __forceinline__ S init(uint32_t code) const { S s = { code, mask_of(code) }; return s; }
	__device__ __forceinline__ static uint32_t code(const S &s) { return s.st; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, const S &) {}
	__device__ __forceinline__ P pre(uint32_t b) const { return bp[b]; }
	__device__ __forceinline__ bool skip16(const S &s, const P (&c)[16]) const
	{
		uint32_t m = 0;
#pragma unroll
		for (int k = 0; k < 16; k++) m |= 1u << c[k];
		return skip_on && __all((m & ~s.sm) == 0u);
	}
	__device__ __forceinline__ S next(S s, P c) const
	{
		if (!((s.sm >> c) & 1u)) {
			s.st = (uint32_t)(*reinterpret_cast<const uint16_t *>(tab + s.st + c * 2u)) << 2;
			s.sm = mask_of(s.st);
		}
		return s;
	}
};

struct CombPol {
	static constexpr bool heavy_next = false;
	typedef uint32_t P;
	typedef uint32_t S;
	__device__ __forceinline__ S init(uint32_t code) const { return code; }
	__device__ __forceinline__ static uint32_t code(S s) { return s; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, S) {}
	const uint8_t *bp;      /* LDS byte -> class map                                    */
	const uint32_t *comb;   /* LDS comb array; state = row offset in entries            */
	const uint32_t *dfl;    /* LDS [C]: row offset of each class's default state        */
	uint32_t abs_min;

	__host__ __device__ static uint32_t lds_bytes(uint32_t tab_bytes) { return FSMHIP_BTAB_BYTES + ((tab_bytes + 15u) & ~15u); }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		bp = setup_btab(lds, a);
		copy_table(lds + FSMHIP_BTAB_BYTES, a);
		comb = reinterpret_cast<const uint32_t *>(lds + FSMHIP_BTAB_BYTES);
		dfl = comb + a.tab_bytes / 4u - 256u; /* image = comb[n], dflt[256] */
		abs_min = absorbing_limit(a);
	}
	__device__ __forceinline__ bool absorbing(S s) const { return s >= abs_min; }
	__device__ __forceinline__ P pre(uint32_t b) const { return bp[b]; }
	__device__ __forceinline__ uint32_t next(uint32_t st, P c) const
	{
		const uint32_t x = comb[st + c] ^ (st << 16);
		return x < 0x10000u ? x : dfl[c];
	}
};

struct Comb256Pol {
	static constexpr bool heavy_next = false;
	typedef uint32_t P;
	/* the state is carried as the raw comb entry that led to it: next state (a row offset) in the high half,
	 * whatever owner tag the entry had in the low half.  That shortens the dependent chain per byte to
	 * bfe (state) -> add_lshl (address) -> ds_read -> 16-bit compare (owner tag == state) -> select:
	 * 4 vector operations where the owner << 16 | next form needed 7 (the walk is bound by the latency of
	 * this chain times the 16 waves a 91 KB table leaves room for, not by LDS bandwidth) */
	typedef uint32_t S;
	__device__ __forceinline__ S init(uint32_t code) const { return code << 16; }
	__device__ __forceinline__ static uint32_t code(S s) { return s >> 16; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, S) {}
	const uint32_t *comb;   /* LDS comb array indexed by row offset + byte: next << 16 | owner */
	uint32_t dflt_e;        /* the default state in entry form */
	uint32_t abs_e;         /* entries >= this lead to absorbing states (absorbing_limit() << 16) */

	__host__ __device__ static uint32_t lds_bytes(uint32_t tab_bytes) { return (tab_bytes + 15u) & ~15u; }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		copy_table(lds, a);
		comb = reinterpret_cast<const uint32_t *>(lds);
		dflt_e = a.dflt << 16;
		const uint32_t lim = absorbing_limit(a);
		abs_e = lim > 0xFFFFu ? 0xFFFFFFFFu : lim << 16;
	}
	__device__ __forceinline__ bool absorbing(S s) const { return s >= abs_e; }
	__device__ __forceinline__ P pre(uint32_t b) const { return b; }
	__device__ __forceinline__ S next(S s, P b) const
	{
		const uint32_t st = s >> 16;
		const uint32_t e = comb[st + b];
		return (uint16_t)e == (uint16_t)st ? e : dflt_e;
	}
};

/*
 * CombSelfPol: CombPol plus a per-state SELF-LOOP MASK carried in a register next to the state:
 * bit c of sm says delta(state, class c) == state.  Regex DFAs spend most bytes in self-loops
 * ([0-9]+, .*, the absorbing DEAD/accept states), and for those bytes the walk needs only the
 * conflict-free byte->class lookup: the comb lookup (random banks, ~3.5-way conflicts) and the
 * reload of the mask run under an exec mask for the few lanes that really change state, and are
 * skipped by the whole wavefront when none does.  Absorbing states have every bit set, so
 * retired lanes drop out of the LDS traffic for free.  Needs <= 32 byte classes.
 */
struct CombSelfState { uint32_t st, sm, rng; };

struct CombSelfPol {
	static constexpr bool heavy_next = false;
	typedef uint32_t P;
	typedef CombSelfState S;
	const uint8_t *bp;      /* LDS byte -> class map                                    */
	const uint2 *comb;      /* LDS comb array of {owner_off << 16 | next_off, smask(next)}:
	                         * one ds_read_b64 brings the next state AND its self-loop mask */
	const uint2 *dsm;       /* LDS [32]: {row offset, self-loop mask} of each class's default state */
	const uint16_t *rng16;  /* LDS: self-loop byte range lo | hi << 8 by row offset      */
	const uint32_t *smask0; /* global: smask by row offset (only to seed a walk)        */
	uint32_t start, start_sm;
	uint32_t ident;         /* WalkArgs::ident_class */
	bool skip_on;

	__host__ __device__ static uint32_t lds_bytes(uint32_t tab_bytes) { return FSMHIP_BTAB_BYTES + ((tab_bytes + 15u) & ~15u); }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		/* device image = comb64[n] (8 B each), dsm[32] (8 B each), rng16[n]; then smask[n] (global only) */
		bp = setup_btab(lds, a);
		copy_table(lds + FSMHIP_BTAB_BYTES, a);
		comb = reinterpret_cast<const uint2 *>(lds + FSMHIP_BTAB_BYTES);
		dsm = comb + a.dflt;   /* a.dflt = number of comb entries */
		rng16 = reinterpret_cast<const uint16_t *>(dsm + 32);
		smask0 = reinterpret_cast<const uint32_t *>(static_cast<const unsigned char *>(a.tab) + a.tab_bytes);
		start = a.start;
		start_sm = smask0[a.start];
		ident = a.ident_class;
		skip_on = !(a.early & 4u);
	}
	__device__ __forceinline__ bool ident_ok() const { return ident < 32u; }
	/* no byte is a self-loop of the start state (anchored patterns): a chunk-level skip test on an input's FIRST chunk cannot pass */
	__device__ __forceinline__ bool first_noskip() const { return ident == 31u && (start_sm & 0x7FFFFFFFu) == 0u; }
	/* every input starts from the start state unless it is resumed: its mask is fetched once per
	 * workgroup, not once per input (the ragged kernel seeds a lane every time an input ends) */
	__device__ __forceinline__ S init(uint32_t code) const
	{
		S s = { code, code == start ? start_sm : smask0[code], rng16[code] };
		return s;
	}
	__device__ __forceinline__ static uint32_t code(S s) { return s.st; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, S) {}
	__device__ __forceinline__ P pre(uint32_t b) const { return bp[b]; }
	/*
	 * The cheapest test first, on the raw input, before any class lookup: when a state's self-loop bytes
	 * are one contiguous range ([0-9]+, [a-z]*, .* and every absorbing state: the planner stores lo | hi << 8
	 * per state), "all 16 bytes of the chunk lie in lo..hi" is a SWAR test on the four dwords -- exists a
	 * byte < lo: (x - lo*0x01010101) & ~x & 0x80808080; exists a byte > hi: ((x + (127-hi)*0x01010101) | x)
	 * & 0x80808080 (exact for the existence question for lo <= 128, hi <= 127; hi = 255 switches the upper
	 * test off) -- about 25 vector operations and NO LDS traffic, where the class-mask form below costs 16
	 * LDS lookups + 32 operations.  On C3's inputs every chunk but a row's first and last passes it, and the
	 * LDS pipe, which the DMA tiles also land in, is left to the tiles.
	 */
	__device__ __forceinline__ bool skip16_raw(const S &s, const u32x4 &w) const
	{
		const uint32_t lo = s.rng & 0xffu, hi = s.rng >> 8;
		const uint32_t LO = lo * 0x01010101u, AD = (127u - hi) * 0x01010101u;
		const uint32_t MM = hi == 255u ? 0u : 0x80808080u;
		const uint32_t aL = ((w.x - LO) & ~w.x) | ((w.y - LO) & ~w.y) | ((w.z - LO) & ~w.z) | ((w.w - LO) & ~w.w);
		const uint32_t aM = ((w.x + AD) | w.x) | ((w.y + AD) | w.y) | ((w.z + AD) | w.z) | ((w.w + AD) | w.w);
		return skip_on && __all(((aL | (aM & MM)) & 0x80808080u) == 0u);
	}
	/* all 16 classes of the chunk are self-loops of every lane's state (digit runs, dead lanes):
	 * 16 shift-ORs and one wave vote replace 16 test-and-branch steps */
	__device__ __forceinline__ bool skip16(const S &s, const P (&c)[16]) const
	{
		uint32_t m = 0;
#pragma unroll
		for (int k = 0; k < 16; k++) m |= 1u << c[k];
		return skip_on && __all((m & ~s.sm) == 0u);
	}
	__device__ __forceinline__ S next(S s, P c) const
	{
		if (!((s.sm >> c) & 1u)) {
			const uint2 e = comb[s.st + c];
			const uint32_t x = e.x ^ (s.st << 16);
			if (x < 0x10000u) {
				s.st = x;
				s.sm = e.y;
			} else { /* no exception here: the class's default state (rare: mostly dying lanes) */
				const uint2 d = dsm[c];
				s.st = d.x;
				s.sm = d.y;
			}
			s.rng = rng16[s.st];
		}
		return s;
	}
};

struct GlobPol {
	static constexpr bool heavy_next = false;
	typedef uint32_t P;
	typedef uint32_t S;
	__device__ __forceinline__ S init(uint32_t code) const { return code; }
	__device__ __forceinline__ static uint32_t code(S s) { return s; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, S) {}
	const uint8_t *bp;         /* LDS byte -> class map                                    */
	const unsigned char *tab;  /* device table; state is a byte offset into it             */
	const unsigned char *hot;  /* LDS copy of the first hot_bytes of the table: the rows   */
	uint32_t hot_bytes;        /* nearest the start state (breadth-first numbering)        */
	uint32_t abs_min;

	__host__ __device__ static uint32_t lds_bytes(uint32_t tab_bytes) { return FSMHIP_BTAB_BYTES + ((tab_bytes + 15u) & ~15u); }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		bp = setup_btab(lds, a);
		copy_table(lds + FSMHIP_BTAB_BYTES, a);
		hot = lds + FSMHIP_BTAB_BYTES;
		hot_bytes = a.tab_bytes;
		tab = static_cast<const unsigned char *>(a.tab);
		abs_min = a.abs_min;
	}
	__device__ __forceinline__ P pre(uint32_t b) const { return bp[b]; }
	__device__ __forceinline__ uint32_t next(uint32_t st, P c) const
	{
		/* explicit address spaces (see SparsePol::next_t: no FLAT load of a selected address) */
		typedef const uint32_t __attribute__((address_space(3))) *lds_u32p;
		typedef const uint32_t __attribute__((address_space(1))) *glb_u32p;
		const uint32_t ad = st + c * 4u;
		const bool cold = ad >= hot_bytes;
		uint32_t v = *(lds_u32p)(uintptr_t)((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)hot + (cold ? 0u : ad));
		if (cold) v = *(glb_u32p)(uintptr_t)(tab + ad);
		return v;
	}
};

/*
 * Glob16Pol: GlobPol for automata of <= 65 535 states -- 2-byte entries (the next state's INDEX; the row address is one
 * v_mad_u32_u24), so the LDS copy of the table's head holds twice the rows and the rest half the L2 lines.  The regime: the
 * COMPLETE DFA of an unanchored pattern list as rx builds one (src/rx/main.c:487-566, :1338-1385: every pattern carries the
 * implicit leading .*, the union has no DEAD default and changes state on almost every byte), a few thousand states whose
 * dense table is 2-4 x LDS: column defaults do not compress it (every (state, letter) pair leads to its own bigram state) and
 * GlobPol kept 1 059 of its 4 133 rows in LDS -- with 64 inputs per wavefront some lane left them in most steps and every
 * step paid the L2 round trip: 0.20 of the HBM peak.  Here breadth-first numbering + 58-byte rows put every state a random
 * input reaches with probability > 1e-3 in LDS, the lookup is an explicit ds_read (not a flat load of a selected address),
 * and the L2 path sits behind a wave-uniform branch that is rarely taken.
 */
struct Glob16Pol {
	static constexpr bool heavy_next = false;
	typedef uint32_t P;        /* class * 2 */
	typedef uint32_t S;        /* state index */
	__device__ __forceinline__ S init(uint32_t code) const { return code; }
	__device__ __forceinline__ static uint32_t code(S s) { return s; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, S) {}
	typedef const uint16_t __attribute__((address_space(3))) *lds_u16p;
	typedef const uint16_t __attribute__((address_space(1))) *glb_u16p;
	const uint8_t *bp;         /* LDS byte -> class map */
	const unsigned char *tab;  /* device table */
	uint32_t hot_lds;          /* LDS byte address of the copy of the table's first hot_bytes */
	uint32_t hot_bytes, row_bytes;

	__host__ __device__ static uint32_t lds_bytes(uint32_t tab_bytes) { return FSMHIP_BTAB_BYTES + ((tab_bytes + 15u) & ~15u); }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		bp = setup_btab(lds, a);
		copy_table(lds + FSMHIP_BTAB_BYTES, a);
		hot_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(lds + FSMHIP_BTAB_BYTES);
		hot_bytes = a.tab_bytes;
		row_bytes = a.dflt;        /* bytes per row: classes * 2 */
		tab = static_cast<const unsigned char *>(a.tab);
	}
	__device__ __forceinline__ P pre(uint32_t b) const { return (uint32_t)bp[b] * 2u; }
	__device__ __forceinline__ uint32_t next(uint32_t st, P c2) const
	{
		const uint32_t ad = __umul24(st, row_bytes) + c2;
		const bool cold = ad >= hot_bytes;
		uint32_t v = *(lds_u16p)(uintptr_t)(hot_lds + (cold ? 0u : ad));
		if (__any(cold)) {
			if (cold) v = *(glb_u16p)(uintptr_t)(tab + ad);
		}
		return v;
	}
	/* two inputs per lane (step16's ROWS == 2 form): both LDS reads in flight together, ONE vote for the L2 path.  A 128 KiB
	 * table leaves room for one 16-wavefront workgroup per CU -- four dependent chains per SIMD where the small LDS layouts
	 * have eight; the second input per lane gives the missing ones back (digits-only rows, which never leave the first rows:
	 * 2.2 TB/s with one input per lane). */
	__device__ __forceinline__ void next2(uint32_t &s0, uint32_t &s1, P c0, P c1) const
	{
		const uint32_t ad0 = __umul24(s0, row_bytes) + c0, ad1 = __umul24(s1, row_bytes) + c1;
		const bool cold0 = ad0 >= hot_bytes, cold1 = ad1 >= hot_bytes;
		uint32_t v0 = *(lds_u16p)(uintptr_t)(hot_lds + (cold0 ? 0u : ad0));
		uint32_t v1 = *(lds_u16p)(uintptr_t)(hot_lds + (cold1 ? 0u : ad1));
		if (__any(cold0 | cold1)) {
			if (cold0) v0 = *(glb_u16p)(uintptr_t)(tab + ad0);
			if (cold1) v1 = *(glb_u16p)(uintptr_t)(tab + ad1);
		}
		s0 = v0;
		s1 = v1;
	}
};

/*
 * SparsePol: base-row records (plan.cpp build_sparse).  A state is its renumbered id; its 16-byte
 * record {bits lo, bits hi, base | DENSE | CONSEC | FULLBASE, offset} comes from LDS for the H states nearest the
 * start state and from HBM/L2 for the rest.  A lane follows base links until a record has the class's
 * bit set -- next state = offset + rank of the bit when the record's targets are consecutive ids
 * (CONSEC: pure arithmetic), else one gather from the exception list -- or is dense (next state from
 * the dense row, LDS for the first rows).  The loop is lane-divergent; chains are bounded by the planner.
 * The LDS / global choice of a record is left to generic pointers on purpose: the compiler turns
 * `st < H ? lrec[st] : grec[st]` into ONE flat_load of a selected address, and one load instruction per
 * turn of the loop measured faster than a ds_read + a global_load under complementary exec masks
 * (422 vs 385 GB/s, profiles/r02_c5_steps.txt): the walk is bound by instructions per turn, every path
 * being live in some lane of a 64-lane wave.
 */
struct SparsePol {
	static constexpr bool heavy_next = true;   /* next() is a divergent loop: EagerPol keeps its per-byte form */
	typedef uint32_t P;   /* class | bit << 8 */
	typedef uint32_t S;
	__device__ __forceinline__ S init(uint32_t code) const { return code; }
	__device__ __forceinline__ static uint32_t code(S s) { return s; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, S) {}
	typedef const u32x4 __attribute__((address_space(3))) *lds_rec_p;
	typedef const uint32_t __attribute__((address_space(3))) *lds_u32_p;
	typedef const u32x4 __attribute__((address_space(1))) *glb_rec_p;
	typedef const uint32_t __attribute__((address_space(1))) *glb_u32_p;
	const uint16_t *pm;        /* LDS: byte -> class | bit << 8 */
	uint32_t ldense_lds;       /* LDS byte address of the first dense rows */
	const uint32_t *ldense;    /* LDS: first dense rows         */
	const u32x4 *lrec;         /* LDS: first H records (generic pointer: see the note on flat loads above) */
	uint32_t lrec_lds;         /* the same as an LDS byte address, for the turns that are known to stay in LDS */
	const u32x4 *grec;
	const uint32_t *gdense, *exc;
	uint32_t H, HDE, abs_min;

	__host__ __device__ static uint32_t lds_bytes(uint32_t tab_bytes) { return (tab_bytes + 15u) & ~15u; }
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		copy_table(lds, a);
		const uint32_t *hdr = static_cast<const uint32_t *>(a.tab);
		const unsigned char *g = static_cast<const unsigned char *>(a.tab);
		H = hdr[1];
		HDE = hdr[2];
		pm = reinterpret_cast<const uint16_t *>(lds + 64);
		ldense = reinterpret_cast<const uint32_t *>(lds + hdr[3]);
		ldense_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(lds + hdr[3]);
		lrec = reinterpret_cast<const u32x4 *>(lds + hdr[4]);
		lrec_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(lds + hdr[4]);
		grec = reinterpret_cast<const u32x4 *>(g + hdr[5]);
		gdense = reinterpret_cast<const uint32_t *>(g + hdr[6]);
		exc = reinterpret_cast<const uint32_t *>(g + hdr[7]);
		abs_min = a.abs_min;
	}
	__device__ __forceinline__ P pre(uint32_t b) const { return pm[b]; }
	__device__ __forceinline__ uint32_t next(uint32_t st, P p) const { return next_t<false>(st, p); }
	/* A chunk whose 16 bytes all belong to classes that own a bit (on a literal set: every byte of its
	 * alphabet) is walked with the `has a bit` tests compiled out: one wave vote per chunk saves five
	 * operations per byte */
	__device__ __forceinline__ void walk16(S &st, const P (&pre)[16]) const
	{
		uint32_t mx = 0;
#pragma unroll
		for (int k = 0; k < 16; k++) mx = pre[k] > mx ? pre[k] : mx;
		if (__all((mx >> 8) < 64u)) {
#pragma unroll
			for (int k = 0; k < 16; k++) st = next_t<true>(st, pre[k]);
		} else {
#pragma unroll
			for (int k = 0; k < 16; k++) st = next_t<false>(st, pre[k]);
		}
	}
	template <bool ALLBITS>
	__device__ __forceinline__ uint32_t next_t(uint32_t st, P p) const
	{
		const uint32_t cls = p & 0xffu, bit = p >> 8;            /* bit 0xff: the class owns no bit */
		const bool hasbit = ALLBITS || bit < 64u;
		const uint64_t sel = hasbit ? (uint64_t)1 << bit : 0u, below = hasbit ? sel - 1u : 0u;
		uint32_t res = st;
		bool live = st < abs_min;
		/* One turn per record of the chain.  The common outcomes -- a hit on a CONSEC record (first + rank,
		 * pure arithmetic), a miss on a FULLBASE record (first(base) + bit, one 4-byte LDS read), a miss
		 * that moves on to the base -- are computed straight-line and selected; only the rare ones (a
		 * dense row, a hit on a record that still owns an exception list) branch.  The walk is bound by
		 * the instructions of this loop: every path is live in some lane of a 64-lane wave. */
		/* first turn: the record of the state itself, LDS or global (one flat load of a selected address) */
		bool first = true;
		while (live) {
			/* The record: LDS for the H states nearest the start state, device memory for the rest -- by an explicit ds_read_b128
			 * (of record 0 for the lanes beyond H: an LDS read cannot fault) and an explicit global_load under the other lanes'
			 * mask.  Rounds 2-5 left the choice to a generic pointer: ONE flat_load of a selected address, measured 10 % faster
			 * on the record walk (422 vs 385 GB/s, profiles/r02_c5_steps.txt) -- but a FLAT instruction whose lanes split between
			 * LDS and memory comes back over two paths, counts on both wait counters and may complete out of order, and the two
			 * intermittent wrong-answer builds this project has met (round 2, round 5: profiles/r08i_*) were both kernels with
			 * such loads inside lane-divergent loops.  Neither could be pinned on the instruction; the record walk is no longer
			 * the fast path of anything (the lazy walk is), so the product now contains no FLAT load at all (tests/test_abi.py
			 * checks the code objects). */
			u32x4 r;
			const bool inl = st < H;
			(void)first;
			r = *(lds_rec_p)(uintptr_t)(lrec_lds + (inl ? st : 0u) * 16u);
			if (!inl) r = *(glb_rec_p)(uintptr_t)(grec + st);
			first = false;
			const uint64_t bits = (uint64_t)r.x | ((uint64_t)r.y << 32);
			const bool hit = (bits & sel) != 0u;
			const uint32_t id = r.z & 0x0FFFFFFFu;
			const bool dense = (r.z & 0x80000000u) != 0u;            /* a dense record has no bits */
			const bool fb = !hit && hasbit && (r.z & 0x20000000u) != 0u;
			uint32_t v = r.w + (uint32_t)__popcll(bits & below);
			if (fb) v = *(lds_u32_p)(uintptr_t)(lrec_lds + id * 16u + 12u) + bit;
			if (dense || (hit && !(r.z & 0x40000000u))) {
				if (dense) {
					const uint32_t o = r.w + cls;
					if (o < HDE) v = *(lds_u32_p)(uintptr_t)(ldense_lds + o * 4u); else v = *(glb_u32_p)(uintptr_t)(gdense + o);
				} else {
					v = *(glb_u32_p)(uintptr_t)(exc + v);
				}
			}
			const bool done = hit || fb || dense;
			res = done ? v : res;
			st = id;
			live = !done;
		}
		return res;
	}
};

/*
 * SparseFastPol: SparsePol with the ENTRY AS THE STATE.  SparsePol carries a state id and begins every turn of its
 * chain loop by loading the record of the id it holds: two or three dependent load -> wait -> ~30 vector + ~25 scalar
 * instruction turns per input byte on a literal-set automaton (own record: a deep trie node, nearly always a miss;
 * its base: the failure state; that one's base: a full shallow node), and the loop's control flow is scalar work of
 * its own.  Here the walk state IS the current state's record {bits, base | flags, first} plus its id, fetched once
 * when the state is entered, and a byte is evaluated in straight-line code against the own record (registers), its
 * base B's (one LDS read: bases are nearly always among the H records nearest the start state, which live in LDS) and,
 * without reading it, B's base:
 *     x = bits << (63 - bit)          bit 63 of x: the class's bit; popcount(x) - 1: its rank among the set bits below
 *     hit   <=> x < 0 (signed);       next = first + popcount(x) - 1        (CONSEC records: children are consecutive ids)
 * i.e. one 64-bit shift, two v_bcnt (the second adds `first - 1`) and one sign test for each of the two records; where
 * neither owns the class the answer is first(B's base) + bit -- B's base owns EVERY bit (a full trie node: rank = bit
 * index), so it costs a 4-byte LDS read of its `first` word and an add.  (Round 3's first form evaluated the third record
 * like the other two: a second 16-byte LDS read, a 64-bit shift, two v_bcnt and a sign test more per byte.)
 * Whether that is the whole story for a state is known when the table is planned (plan.cpp: records are re-based onto
 * LDS-resident ancestors; FAST says that every class the record does not own is answered as above, CONSEC that a hit
 * on it is first + rank), so one flag test and one wave vote per byte decide it; the lanes where it fails (a hit on a record
 * that keeps an exception list, dense rows, classes without a bit) take SparsePol's general loop for that byte.  Used by
 * the fixed-stride kernels on plain (non-eager) walks.
 */
struct SparseFastState {
	uint32_t b0, b1;   /* the state's record: class bits */
	uint32_t meta;     /* base | DENSE | CONSEC | FULLBASE | FAST */
	uint32_t off;      /* first child / exception offset / dense row offset */
	uint32_t id;
};

struct SparseFastPol : SparsePol {
	typedef SparseFastState S;
	typedef SparsePol::P P;
	typedef const uint32_t __attribute__((address_space(3))) *lds_w_p;
	uint32_t lrec_lo, lrec_hi, grec_lo, grec_hi;   /* the two record arrays as flat addresses, halves apart (enter()) */
	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		SparsePol::setup(lds, a);
		const uint64_t l = reinterpret_cast<uint64_t>(lrec), g = reinterpret_cast<uint64_t>(grec);
		lrec_lo = (uint32_t)l; lrec_hi = (uint32_t)(l >> 32);
		grec_lo = (uint32_t)g; grec_hi = (uint32_t)(g >> 32);
	}
	/* every state has a record (the absorbing ones an all-zero one: plan.cpp), so a state is entered by ONE flat load of
	 * `its array's base + 16 * id`.  The host guarantees that neither array crosses a 4 GiB boundary (fsm_hip.hip: the
	 * knob falls back to SparsePol otherwise), so only the LOW half of the address depends on the id: a select + shift-add
	 * for it, a select for the high half -- where a 64-bit select + 64-bit shift-add cost 8 vector operations per byte */
	__device__ __forceinline__ S enter(uint32_t id) const
	{
		/* (explicit address spaces, as in SparsePol::next_t -- and for its reason; rounds 3-5 made this ONE flat load of a
		 * selected address) */
		const bool in = id < H;
		u32x4 r = *(lds_rec_p)(uintptr_t)(lrec_lds + (in ? id : 0u) * 16u);
		if (!in) r = *(glb_rec_p)(uintptr_t)(grec + id);
		S s = { r.x, r.y, r.z, r.w, id };
		return s;
	}
	__device__ __forceinline__ S init(uint32_t code) const { return enter(code); }
	__device__ __forceinline__ static uint32_t code(const S &s) { return s.id; }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, const S &) {}
	__device__ __forceinline__ S next(const S &s, P p) const { return enter(SparsePol::next_t<false>(s.id, p)); }

	__device__ __forceinline__ S step_fast(const S &s, P p) const
	{
		const uint32_t bit = p >> 8, sh = 63u - bit;             /* (every class of the chunk owns a bit: walk16 checked) */
		/* own record: x = bits << (63 - bit); bit 63 of x is the class's bit, popcount(x) - 1 its rank */
		const uint64_t xA = (((uint64_t)s.b1 << 32) | s.b0) << sh;
		const bool hA = (int32_t)(xA >> 32) < 0;
		const uint32_t nA = (uint32_t)__builtin_popcount((uint32_t)xA) + ((uint32_t)__builtin_popcount((uint32_t)(xA >> 32)) + (s.off - 1u));
		/* its base B, then first(B's base): unguarded -- where the planner's FAST flag is clear they may read anything (an
		 * LDS read cannot fault) and the lane takes the general loop below */
		const uint32_t B = s.meta & 0x0FFFFFFFu;
		const u32x4 rb = *(lds_rec_p)(uintptr_t)(lrec_lds + B * 16u);
		const uint32_t cf = *(lds_w_p)(uintptr_t)(lrec_lds + (rb.z & 0x0FFFFFFFu) * 16u + 12u);
		const uint64_t xB = (((uint64_t)rb.y << 32) | rb.x) << sh;
		const bool hB = (int32_t)(xB >> 32) < 0;
		const uint32_t nB = (uint32_t)__builtin_popcount((uint32_t)xB) + ((uint32_t)__builtin_popcount((uint32_t)(xB >> 32)) + (rb.w - 1u));
		uint32_t n = hA ? nA : hB ? nB : cf + bit;              /* B's base owns every bit: rank = bit index */
		/* the record owns the class: its children must be consecutive ids (CONSEC); it does not: the planner vouches for
		 * the rest (FAST).  (An absorbing state's record has no bits and FAST set: plan.cpp.) */
		const uint32_t need = s.meta & (hA ? 0x40000000u : 0x10000000u);
		if (__builtin_amdgcn_ballot_w64(need == 0u) != 0u) {
			if (need == 0u) n = SparsePol::next_t<true>(s.id, p);   /* the general chain loop, for these lanes only */
		}
		if (s.id >= abs_min) n = s.id;
		return enter(n);
	}
	__device__ __forceinline__ void walk16(S &st, const P (&pre)[16]) const
	{
		uint32_t mx = 0;
#pragma unroll
		for (int k = 0; k < 16; k++) mx = pre[k] > mx ? pre[k] : mx;
		if (__builtin_amdgcn_ballot_w64((mx >> 8) >= 64u) == 0u) {
#pragma unroll
			for (int k = 0; k < 16; k++) st = step_fast(st, pre[k]);
		} else {
#pragma unroll
			for (int k = 0; k < 16; k++) st = next(st, pre[k]);
		}
	}
};

/*
 * EagerPol<Pol>: any policy plus the eager-output side channel of fsm_exec (exec.c:126-144): the
 * ids attached to the start state and to every state entered are OR-ed into a 64-bit set carried
 * next to the state.  States with outputs are numbered so that one range test on the encoded state
 * finds them; only lanes entering such a state do the (exec-masked) mask lookup.
 */
template <class Pol>
struct EagerState {
	typename Pol::S s;
	uint64_t acc;
};

template <class Pol>
struct EagerPol : Pol {
	typedef EagerState<Pol> S;
	typedef typename Pol::P P;
	const uint64_t *emask;
	uint32_t lo_end, hi_begin, span, fin_div, abs_min_code;

	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		Pol::setup(lds, a);
		abs_min_code = a.abs_min;
		emask = a.emask;
		lo_end = a.eager_lo_end;
		hi_begin = a.eager_hi_begin;
		span = hi_begin - lo_end;
		fin_div = a.fin_div;
	}
	__device__ __forceinline__ uint64_t outputs_of(uint32_t c) const
	{
		uint64_t m = 0;
		if (c < lo_end || c >= hi_begin) m = emask[c / fin_div];
		return m;
	}
	__device__ __forceinline__ S init(uint32_t code) const
	{
		S st;
		st.s = Pol::init(code);
		st.acc = outputs_of(code); /* the start state emits before any input (exec.c:126-130) */
		return st;
	}
	__device__ __forceinline__ static uint32_t code(const S &st) { return Pol::code(st.s); }
	template <class Q = Pol>
	__device__ __forceinline__ auto absorbing(const S &st) const -> decltype(static_cast<const Q *>(nullptr)->absorbing(st.s)) { return Pol::absorbing(st.s); }
	/* a chunk that changes no state emits nothing either */
	template <class Q = Pol>
	__device__ __forceinline__ auto skip16(const S &st, const P (&pre)[16]) const
		-> decltype(static_cast<const Q *>(nullptr)->skip16(st.s, pre))
	{
		return Pol::skip16(st.s, pre);
	}
	template <class Q = Pol>
	__device__ __forceinline__ auto skip16_raw(const S &st, const u32x4 &w) const
		-> decltype(static_cast<const Q *>(nullptr)->skip16_raw(st.s, w))
	{
		return Pol::skip16_raw(st.s, w);
	}
	__device__ __forceinline__ S next(S st, P p) const
	{
		const uint32_t before = Pol::code(st.s);
		st.s = Pol::next(st.s, p);
		const uint32_t c = Pol::code(st.s);
		/* OR is idempotent: staying in the same state emits nothing new */
		/* outside [lo_end, hi_begin) in one unsigned compare */
		if (c != before && (c - lo_end) >= span) st.acc |= emask[c / fin_div];
		return st;
	}
	/* A whole 16-byte chunk at once.  Entering a state with outputs is rare (a pattern has just completed),
	 * so the chunk is first walked as a plain chunk while one running MINIMUM of the states entered notes
	 * whether any of them lies below lo_end (the non-absorbing states with outputs; v_min3: half an
	 * operation per byte, where next() has 4 + a branch and the first form of this test -- a running maximum
	 * of state - lo_end -- had 2); the states from hi_begin up are absorbing, so one of those was entered in
	 * this chunk iff the chunk ENDS in it.  Only the lanes that did enter one re-walk the chunk byte by byte from
	 * its first state with the exact per-byte rule.  A lane that starts the chunk in an absorbing state
	 * cannot enter anything: its outputs were collected when it got there. */
	template <class Q = Pol, class = typename std::enable_if<!Q::heavy_next>::type>
	__device__ __forceinline__ void walk16(S &st, const P (&pre)[16]) const
	{
		typename Pol::S s = st.s;
		uint32_t m = 0xFFFFFFFFu;
#pragma unroll
		for (int k = 0; k < 16; k++) {
			s = Pol::next(s, pre[k]);
			const uint32_t c = Pol::code(s);
			m = c < m ? c : m;
		}
		if ((m < lo_end || Pol::code(s) >= hi_begin) && Pol::code(st.s) < abs_min_code) {
			S t = st;
#pragma unroll
			for (int k = 0; k < 16; k++) t = next(t, pre[k]);
			st = t;
		} else {
			st.s = s;
		}
	}
	__device__ __forceinline__ static void finish(const WalkArgs &a, uint64_t i, bool valid, const S &st)
	{
		if (valid && a.eager_out != nullptr) a.eager_out[i] = st.acc;
	}
};

/*
 * EagerWidePol<Pol>: the same side channel for more than 64 ids.  The id set of input i is
 * eager_words u64 in device memory, owned by the lane that walks i: entering a state with outputs
 * ORs that state's (word, mask) pairs into it (plain read-modify-write, no atomics; rare).  The
 * buffer is zeroed on the launch stream before the kernel.
 */
template <class Pol>
struct EagerWideState {
	typename Pol::S s;
	uint64_t *row;     /* NULL for lanes without an input */
	uint32_t pend;     /* encoded state whose outputs are not written yet, or NONE */
};

template <class Pol>
struct EagerWidePol : Pol {
	typedef EagerWideState<Pol> S;
	typedef typename Pol::P P;
	const uint32_t *ew_off, *ew_word;
	const uint64_t *ew_mask;
	uint32_t lo_end, hi_begin, fin_div, abs_min_code;

	__device__ __forceinline__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		Pol::setup(lds, a);
		abs_min_code = a.abs_min;
		ew_off = a.ew_off;
		ew_word = a.ew_word;
		ew_mask = a.ew_mask;
		lo_end = a.eager_lo_end;
		hi_begin = a.eager_hi_begin;
		fin_div = a.fin_div;
	}
	__device__ __forceinline__ bool emits(uint32_t c) const { return (c - lo_end) >= (hi_begin - lo_end); }   /* outside [lo_end, hi_begin) */
	__device__ __forceinline__ void emit(uint32_t c, uint64_t *row) const
	{
		if (c != 0xFFFFFFFFu && row != nullptr) {
			const uint32_t idx = c / fin_div;
			for (uint32_t k = ew_off[idx]; k < ew_off[idx + 1]; k++) row[ew_word[k]] |= ew_mask[k];
		}
	}
	__device__ __forceinline__ S init_at(uint32_t code, const WalkArgs &a, uint64_t i, bool valid) const
	{
		S st;
		st.s = Pol::init(code);
		st.row = valid ? a.eager_out + i * a.eager_words : nullptr;
		st.pend = emits(code) ? code : 0xFFFFFFFFu;   /* the start state emits before any input (exec.c:126-130) */
		return st;
	}
	__device__ __forceinline__ static uint32_t code(const S &st) { return Pol::code(st.s); }
	template <class Q = Pol>
	__device__ __forceinline__ auto absorbing(const S &st) const -> decltype(static_cast<const Q *>(nullptr)->absorbing(st.s)) { return Pol::absorbing(st.s); }
	template <class Q = Pol>
	__device__ __forceinline__ auto skip16(const S &st, const P (&pre)[16]) const
		-> decltype(static_cast<const Q *>(nullptr)->skip16(st.s, pre))
	{
		return Pol::skip16(st.s, pre);   /* pending outputs stay pending: the state is unchanged */
	}
	template <class Q = Pol>
	__device__ __forceinline__ auto skip16_raw(const S &st, const u32x4 &w) const
		-> decltype(static_cast<const Q *>(nullptr)->skip16_raw(st.s, w))
	{
		return Pol::skip16_raw(st.s, w);
	}
	/* The kernels may compute next() for a byte past the end of a ragged input and drop the result,
	 * so a step must not write.  The state handed IN is committed: its pending outputs are written
	 * here, the new state's are left pending (the last one is written by finish()). */
	__device__ __forceinline__ S next(S st, P p) const
	{
		emit(st.pend, st.row);
		const uint32_t before = Pol::code(st.s);
		st.s = Pol::next(st.s, p);
		const uint32_t c = Pol::code(st.s);
		/* fsm_exec emits on every transition INTO a state, self-loops included (exec.c:139-144); OR is
		 * idempotent, so re-entering the same state need not write again */
		st.pend = (c != before && emits(c)) ? c : 0xFFFFFFFFu;
		return st;
	}
	/* chunk-level form, as EagerPol::walk16: the chunk is walked as a plain chunk unless a running maximum
	 * says that some state entered in it emits; the pending outputs of the state the chunk starts in are
	 * written first (that state is committed) */
	template <class Q = Pol, class = typename std::enable_if<!Q::heavy_next>::type>
	__device__ __forceinline__ void walk16(S &st, const P (&pre)[16]) const
	{
		typename Pol::S s = st.s;
		uint32_t m = 0xFFFFFFFFu;
#pragma unroll
		for (int k = 0; k < 16; k++) {
			s = Pol::next(s, pre[k]);
			const uint32_t c = Pol::code(s);
			m = c < m ? c : m;
		}
		if ((m < lo_end || Pol::code(s) >= hi_begin) && Pol::code(st.s) < abs_min_code) {
			S t = st;
#pragma unroll
			for (int k = 0; k < 16; k++) t = next(t, pre[k]);
			st = t;
		} else {
			emit(st.pend, st.row);
			st.pend = 0xFFFFFFFFu;
			st.s = s;
		}
	}
	__device__ __forceinline__ void finish_at(const S &st) const { emit(st.pend, st.row); }
	__device__ __forceinline__ static void finish(const WalkArgs &, uint64_t, bool, const S &) {}
};

/* member-wise select of a walk state (a ternary on the aggregates themselves makes the compiler take
 * their addresses: 48-64 bytes of scratch per lane in the first version of the ragged / generic kernels) */
__device__ __forceinline__ uint32_t pick(bool c, uint32_t x, uint32_t y) { return c ? x : y; }
__device__ __forceinline__ LdsSelfState pick(bool c, const LdsSelfState &x, const LdsSelfState &y)
{
	LdsSelfState r = { c ? x.st : y.st, c ? x.sm : y.sm };
	return r;
}
__device__ __forceinline__ CombSelfState pick(bool c, const CombSelfState &x, const CombSelfState &y)
{
	CombSelfState r = { c ? x.st : y.st, c ? x.sm : y.sm, c ? x.rng : y.rng };
	return r;
}
__device__ __forceinline__ SparseFastState pick(bool c, const SparseFastState &x, const SparseFastState &y)
{
	SparseFastState r = { c ? x.b0 : y.b0, c ? x.b1 : y.b1, c ? x.meta : y.meta, c ? x.off : y.off, c ? x.id : y.id };
	return r;
}
template <class Pol>
__device__ __forceinline__ EagerState<Pol> pick(bool c, const EagerState<Pol> &x, const EagerState<Pol> &y)
{
	EagerState<Pol> r;
	r.s = pick(c, x.s, y.s);
	r.acc = c ? x.acc : y.acc;
	return r;
}
template <class Pol>
__device__ __forceinline__ EagerWideState<Pol> pick(bool c, const EagerWideState<Pol> &x, const EagerWideState<Pol> &y)
{
	EagerWideState<Pol> r;
	r.s = pick(c, x.s, y.s);
	r.row = c ? x.row : y.row;
	r.pend = c ? x.pend : y.pend;
	return r;
}

/* the state an input starts from: policies that need the input index define init_at() */
template <class Pol>
__device__ __forceinline__ auto init_state(const Pol &pol, uint32_t code, const WalkArgs &a, uint64_t i, bool valid, int)
	-> decltype(pol.init_at(code, a, i, valid))
{
	return pol.init_at(code, a, i, valid);
}
template <class Pol>
__device__ __forceinline__ typename Pol::S init_state(const Pol &pol, uint32_t code, const WalkArgs &, uint64_t, bool, long)
{
	return pol.init(code);
}

/* end of an input: policies with deferred side effects define finish_at() */
template <class Pol>
__device__ __forceinline__ auto finish_state(const Pol &pol, const WalkArgs &a, uint64_t i, bool valid, const typename Pol::S &st, int)
	-> decltype(pol.finish_at(st))
{
	pol.finish_at(st);
	Pol::finish(a, i, valid, st);
}
template <class Pol>
__device__ __forceinline__ void finish_state(const Pol &, const WalkArgs &a, uint64_t i, bool valid, const typename Pol::S &st, long)
{
	Pol::finish(a, i, valid, st);
}

/* 16 input bytes of ROWS independent rows: all state-independent lookups
 * first, then the ROWS state chains interleaved byte by byte. */
/* the state-independent lookup of byte k of a 16-byte chunk; a policy may want the raw dword (pre_dw) */
template <class Pol>
__device__ __forceinline__ auto pre_of(const Pol &pol, const u32x4 &w, int k, int) -> decltype(pol.pre_dw(0u, 0))
{
	const uint32_t d = (k < 4) ? w.x : (k < 8) ? w.y : (k < 12) ? w.z : w.w;
	return pol.pre_dw(d, k & 3);
}
template <class Pol>
__device__ __forceinline__ typename Pol::P pre_of(const Pol &pol, const u32x4 &w, int k, long)
{
	return pol.pre(byte_of(w, k));
}

/* a policy may know cheaply that none of 16 bytes changes the state of ANY lane (skip16) */
template <class Pol>
__device__ __forceinline__ auto skip_chunk(const Pol &pol, const typename Pol::S &st, const typename Pol::P (&pre)[16], int)
	-> decltype(pol.skip16(st, pre))
{
	return pol.skip16(st, pre);
}
template <class Pol>
__device__ __forceinline__ bool skip_chunk(const Pol &, const typename Pol::S &, const typename Pol::P (&)[16], long)
{
	return false;
}

/* ... or, cheaper still, from the raw 16 input bytes before any lookup (skip16_raw) */
template <class Pol>
__device__ __forceinline__ auto skip_chunk_raw(const Pol &pol, const typename Pol::S &st, const u32x4 &w, int)
	-> decltype(pol.skip16_raw(st, w))
{
	return pol.skip16_raw(st, w);
}
template <class Pol>
__device__ __forceinline__ bool skip_chunk_raw(const Pol &, const typename Pol::S &, const u32x4 &, long)
{
	return false;
}

/* a policy may walk a whole chunk itself (walk16: EagerPol's rare-event form) */
template <class Pol>
__device__ __forceinline__ auto walk_chunk(const Pol &pol, typename Pol::S &st, const typename Pol::P (&pre)[16], int)
	-> decltype(pol.walk16(st, pre))
{
	pol.walk16(st, pre);
}
template <class Pol>
__device__ __forceinline__ void walk_chunk(const Pol &pol, typename Pol::S &st, const typename Pol::P (&pre)[16], long)
{
#pragma unroll
	for (int k = 0; k < 16; k++) st = pol.next(st, pre[k]);
}

/* mask_absorbing (see absorbing_limit()): the chunk as a lane in an absorbing state gets to see it */
template <class Pol>
__device__ __forceinline__ auto mask_absorbing(const Pol &pol, const typename Pol::S &st, const u32x4 &w, int)
	-> decltype(pol.absorbing(st), u32x4())
{
	const uint32_t m = pol.absorbing(st) ? 0u : 0xFFFFFFFFu;
	return u32x4{w.x & m, w.y & m, w.z & m, w.w & m};
}
template <class Pol>
__device__ __forceinline__ u32x4 mask_absorbing(const Pol &, const typename Pol::S &, const u32x4 &w, long) { return w; }

/* the dependent chains of ROWS inputs per lane over a chunk, interleaved byte by byte; a policy with next2 walks two inputs'
 * bytes in one call (Glob16Pol: one vote for its L2 path instead of two) */
template <class Pol, int ROWS>
__device__ __forceinline__ auto step16_rows_chain(const Pol &pol, typename Pol::S (&st)[ROWS], const typename Pol::P (&pre)[ROWS][16], int)
	-> typename std::enable_if<ROWS == 2, decltype(pol.next2(st[0], st[0], pre[0][0], pre[0][0]), void())>::type
{
#pragma unroll
	for (int k = 0; k < 16; k++) pol.next2(st[0], st[ROWS - 1], pre[0][k], pre[ROWS - 1][k]);
}
template <class Pol, int ROWS>
__device__ __forceinline__ void step16_rows_chain(const Pol &pol, typename Pol::S (&st)[ROWS], const typename Pol::P (&pre)[ROWS][16], long)
{
#pragma unroll
	for (int k = 0; k < 16; k++)
#pragma unroll
		for (int r = 0; r < ROWS; r++) st[r] = pol.next(st[r], pre[r][k]);
}

template <class Pol, int ROWS>
__device__ __forceinline__ void step16(const Pol &pol, typename Pol::S (&st)[ROWS], const u32x4 (&w)[ROWS])
{
	if (ROWS == 1 && skip_chunk_raw(pol, st[0], w[0], 0)) return;
	typename Pol::P pre[ROWS][16];
#pragma unroll
	for (int r = 0; r < ROWS; r++) {
		const u32x4 wm = mask_absorbing(pol, st[r], w[r], 0);
#pragma unroll
		for (int k = 0; k < 16; k++) pre[r][k] = pre_of(pol, wm, k, 0);
	}
	if (ROWS == 1) {
		if (skip_chunk(pol, st[0], pre[0], 0)) return;
		walk_chunk(pol, st[0], pre[0], 0);
		return;
	}
	step16_rows_chain<Pol, ROWS>(pol, st, pre, 0);
}

/* ------------------------------------------------------------------ */
/* result write-back                                                  */
/* ------------------------------------------------------------------ */

__device__ __forceinline__ void write_result(const WalkArgs &a, uint64_t word, uint64_t i, bool valid, uint32_t st)
{
	uint32_t end = FSMHIP_NO_MATCH;
	const uint32_t idx = fin_index(a, st);
	if (valid) end = a.fin[idx];
	if (valid && a.end_out != nullptr) a.end_out[i] = end;
	if (valid && a.out2 != nullptr) a.out2[i] = a.fin2[idx];
	if (valid && a.state_io != nullptr) a.state_io[i] = a.orig_of[idx];
	const uint64_t m = __ballot(valid && end != FSMHIP_NO_MATCH);
	if (a.bitmap != nullptr && (threadIdx.x & 63u) == 0 && word * 64u < a.n) a.bitmap[word] = m;
}

/* ------------------------------------------------------------------ */
/* walk_direct: per-lane 16-byte loads, NB chunks in flight, ROWS rows */
/* ------------------------------------------------------------------ */

template <class Pol, int NB, int ROWS>
__global__ void __launch_bounds__(1024)
walk_direct(const WalkArgs a)
{
	extern __shared__ __align__(16) unsigned char lds[];
	Pol pol;
	pol.setup(lds, a);
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const uint64_t ntiles = (a.n + 64u * ROWS - 1u) / (64u * ROWS);
	const uint32_t nchunks = (uint32_t)(a.stride / 16u);
	const uint32_t ngroups = nchunks / NB; /* host guarantees nchunks % NB == 0 */

	for (uint64_t tile = (uint64_t)blockIdx.x * nw + wave; tile < ntiles; tile += (uint64_t)gridDim.x * nw) {
		uint64_t i[ROWS];
		const u32x4 *q[ROWS];
		typename Pol::S st[ROWS];
		u32x4 cur[NB][ROWS], nxt[NB][ROWS];
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			i[r] = (tile * ROWS + r) * 64u + lane;
			q[r] = reinterpret_cast<const u32x4 *>(a.base + (i[r] < a.n ? i[r] : a.n - 1) * a.stride);
			st[r] = init_state(pol, start_code(a, i[r], i[r] < a.n), a, i[r], i[r] < a.n, 0);
		}
#pragma unroll
		for (int j = 0; j < NB; j++)
#pragma unroll
			for (int r = 0; r < ROWS; r++) cur[j][r] = q[r][j];
		for (uint32_t g = 0; g < ngroups; g++) {
			if (g + 1 < ngroups) {
#pragma unroll
				for (int j = 0; j < NB; j++)
#pragma unroll
					for (int r = 0; r < ROWS; r++) nxt[j][r] = q[r][(g + 1) * NB + j];
			}
#pragma unroll
			for (int j = 0; j < NB; j++) step16<Pol, ROWS>(pol, st, cur[j]);
			if (a.early & 1u) {
				bool done = true;
#pragma unroll
				for (int r = 0; r < ROWS; r++) done = done && Pol::code(st[r]) >= a.abs_min;
				if (__all(done)) break;
			}
#pragma unroll
			for (int j = 0; j < NB; j++)
#pragma unroll
				for (int r = 0; r < ROWS; r++) cur[j][r] = nxt[j][r];
		}
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			write_result(a, tile * ROWS + r, i[r], i[r] < a.n, Pol::code(st[r]));
			finish_state(pol, a, i[r], i[r] < a.n, st[r], 0);
		}
	}
}

/* walk_direct_np: same loads, NO register double-buffer: 8 chunks (one full 128-byte line per
 * lane) are loaded, waited for and walked; latency is hidden by occupancy instead (<= 64 VGPRs so
 * two 16-wave workgroups share a CU when their LDS tables fit twice). */
template <class Pol, int NB>
__global__ void __launch_bounds__(1024, 8)
walk_direct_np(const WalkArgs a)
{
	extern __shared__ __align__(16) unsigned char lds[];
	Pol pol;
	pol.setup(lds, a);
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const uint64_t ntiles = (a.n + 63u) / 64u;
	const uint32_t ngroups = (uint32_t)(a.stride / 16u) / NB; /* host guarantees divisibility */

	for (uint64_t tile = (uint64_t)blockIdx.x * nw + wave; tile < ntiles; tile += (uint64_t)gridDim.x * nw) {
		const uint64_t i = tile * 64u + lane;
		const u32x4 *q = reinterpret_cast<const u32x4 *>(a.base + (i < a.n ? i : a.n - 1) * a.stride);
		typename Pol::S st[1] = { init_state(pol, start_code(a, i, i < a.n), a, i, i < a.n, 0) };
		for (uint32_t g = 0; g < ngroups; g++) {
			u32x4 cur[NB][1];
			/* (a.early & 2): a lane whose input can no longer change state stops reading it, as
			 * fsm_exec stops pulling bytes at a missing edge (exec.c:133-138) */
			if (!(a.early & 2u) || Pol::code(st[0]) < a.abs_min) {
#pragma unroll
				for (int j = 0; j < NB; j++) cur[j][0] = q[g * NB + j];
			} else {
#pragma unroll
				for (int j = 0; j < NB; j++) cur[j][0] = u32x4{0u, 0u, 0u, 0u};
			}
#pragma unroll
			for (int j = 0; j < NB; j++) step16<Pol, 1>(pol, st, cur[j]);
			if ((a.early & 1u) && __all(Pol::code(st[0]) >= a.abs_min)) break;
		}
		write_result(a, tile, i, i < a.n, Pol::code(st[0]));
		finish_state(pol, a, i, i < a.n, st[0], 0);
	}
}

/* ------------------------------------------------------------------ */
/* walk_ldsdma: coalesced SEG-byte row segments DMA'd into a per-wave  */
/* LDS tile, read back row-per-lane                                   */
/* ------------------------------------------------------------------ */

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

/*
 * SEG = 64 or 128 bytes of every row per tile (tile = 64 rows x SEG = 4 / 8 KiB per wave).
 * One global_load_lds_dwordx4 moves 1 KiB: 64/PIECES rows x PIECES 16-byte pieces, PIECES = SEG/16
 * adjacent lanes per row, so every request is a whole 64- or 128-byte run of one row.  With SEG = 128
 * each 128-byte line is fetched by exactly one instruction (SEG = 64 splits a line over two
 * instructions a whole tile apart; rocprofv3 FETCH_SIZE showed 16 % re-fetch, profiles/r01a_c2*).
 *
 * LDS placement is fixed by the hardware (M0 base + lane*16), so the loader picks WHICH piece each
 * lane fetches: loader lane (row r, slot q) of DMA instruction j fetches piece (q - rot(i)) mod PIECES
 * of row i = j*RPI + r, and reader lane i finds piece p of its row in slot (p + rot(i)) mod PIECES,
 * rot(i) = (i >> ROTSH) mod PIECES.  For both SEG values every ds_read_b128 lane group
 * ({0-3,12-15,20-27}, ...) then touches 16 distinct 16-byte slots: conflict-free.
 * AUX = 2 marks the DMA loads nontemporal: every input line is used exactly once (SEG = 128), so
 * it need not displace the transition table from L2.
 */
template <class Pol, int SEG, int AUX, int MAXT = 1024>
__global__ void __launch_bounds__(MAXT)
walk_ldsdma(const WalkArgs a)
{
	constexpr uint32_t PIECES = SEG / 16u;      /* 4 | 8 */
	constexpr uint32_t RPI = 64u / PIECES;      /* rows per DMA instruction: 16 | 8 */
	constexpr uint32_t NDMA = PIECES;           /* DMA instructions per tile: 4 | 8 */
	constexpr uint32_t ROTSH = SEG == 64 ? 2u : 1u;
	constexpr uint32_t TILE = 64u * SEG;

	extern __shared__ __align__(16) unsigned char lds[];
	Pol pol;
	pol.setup(lds, a);
	__syncthreads();

	/* the wavefront's index as a SCALAR: everything derived from it (tile slot, input range, ring cursors) is then
	 * wave-uniform to the compiler too and lives in SGPRs / on the scalar unit */
	const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
	unsigned char *stg = lds + Pol::lds_bytes(a.tab_bytes) + wave * TILE;
	const uint64_t ntiles = (a.n + 63u) / 64u;
	const uint32_t nseg = (uint32_t)(a.stride / SEG); /* host guarantees stride % SEG == 0 */

	const uint32_t lr = lane / PIECES, lq = lane % PIECES;          /* loader role */
	const uint32_t rj = lane / RPI, rr = lane % RPI;                /* reader role */
	const unsigned char *rd = stg + rj * 1024u + rr * SEG;
	const uint32_t rot = (lane >> ROTSH) & (PIECES - 1u);

	for (uint64_t tile = (uint64_t)blockIdx.x * nw + wave; tile < ntiles; tile += (uint64_t)gridDim.x * nw) {
		const uint64_t i = tile * 64u + lane;
		const bool valid = i < a.n;
		const uint64_t row0 = tile * 64u;
		const unsigned char *src[NDMA];
#pragma unroll
		for (uint32_t j = 0; j < NDMA; j++) {
			const uint32_t ri = j * RPI + lr; /* the reader lane this row belongs to */
			uint64_t row = row0 + ri;
			if (row >= a.n) row = a.n - 1;
			const uint32_t piece = (lq - ((ri >> ROTSH) & (PIECES - 1u))) & (PIECES - 1u);
			src[j] = a.base + row * a.stride + piece * 16u;
		}
		typename Pol::S st[1] = { init_state(pol, start_code(a, i, valid), a, i, valid, 0) };
#pragma unroll
		for (uint32_t j = 0; j < NDMA; j++)
			__builtin_amdgcn_global_load_lds((glb_void_t *)(src[j]), (lds_void_t *)(stg + j * 1024u), 16, 0, AUX);
		for (uint32_t s = 0; s < nseg; s++) {
			__builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): the tile has landed */
			__asm__ volatile("" ::: "memory");
			u32x4 w[PIECES][1];
#pragma unroll
			for (uint32_t p = 0; p < PIECES; p++)
				w[p][0] = *reinterpret_cast<const u32x4 *>(rd + ((p + rot) & (PIECES - 1u)) * 16u);
			__builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): tile is in registers, slot reusable */
			__asm__ volatile("" ::: "memory");
			if (s + 1 < nseg) {
				if (a.early & 2u) {
					/* A lane in an absorbing state stops reading its row, as fsm_exec stops pulling bytes at a
					 * missing edge (exec.c:133-138): the rows of the lanes that are absorbing NOW are left out of
					 * the next tile (their slots keep stale bytes, which an absorbing state ignores -- and which
					 * pass every chunk-level skip test, an absorbing state looping on all 256 bytes). */
					const uint64_t mine = __ballot(Pol::code(st[0]) >= a.abs_min) >> lr;   /* bit j * RPI: row j * RPI + lr */
#pragma unroll
					for (uint32_t j = 0; j < NDMA; j++)
						if (!((mine >> (j * RPI)) & 1u))
							__builtin_amdgcn_global_load_lds((glb_void_t *)(src[j] + (uint64_t)(s + 1) * SEG),
							                                 (lds_void_t *)(stg + j * 1024u), 16, 0, AUX);
				} else {
#pragma unroll
					for (uint32_t j = 0; j < NDMA; j++)
						__builtin_amdgcn_global_load_lds((glb_void_t *)(src[j] + (uint64_t)(s + 1) * SEG),
						                                 (lds_void_t *)(stg + j * 1024u), 16, 0, AUX);
				}
			}
#pragma unroll
			for (uint32_t p = 0; p < PIECES; p++) step16<Pol, 1>(pol, st, w[p]);
			if ((a.early & 1u) && __all(Pol::code(st[0]) >= a.abs_min)) {
				__builtin_amdgcn_s_waitcnt(0x0F70); /* drain the prefetch before the tile is reused */
				break;
			}
		}
		write_result(a, tile, i, valid, Pol::code(st[0]));
		finish_state(pol, a, i, valid, st[0], 0);
	}
}

/* does the policy have a chunk-level skip test (on raw bytes or on classes)? */
template <class Pol>
constexpr auto has_skip_raw(int) -> decltype(static_cast<const Pol *>(nullptr)->skip16_raw(*static_cast<const typename Pol::S *>(nullptr), *static_cast<const u32x4 *>(nullptr)), true) { return true; }
template <class Pol> constexpr bool has_skip_raw(long) { return false; }
template <class Pol>
constexpr auto has_skip_cls(int) -> decltype(static_cast<const Pol *>(nullptr)->skip16(*static_cast<const typename Pol::S *>(nullptr), *static_cast<const typename Pol::P (*)[16]>(nullptr)), true) { return true; }
template <class Pol> constexpr bool has_skip_cls(long) { return false; }

/* A chunk of which only bytes [lo, lo + cnt) belong to the input (cnt >= 1): the others are replaced by a
 * copy of byte lo, so that a chunk-level skip test -- "does any of these 16 bytes leave the state's
 * self-loop set" -- answers for the input's own bytes alone. */
__device__ __forceinline__ u32x4 fill_invalid(const u32x4 &w, uint32_t lo, uint32_t cnt)
{
	const uint32_t vm = (0xffffu >> (16u - cnt)) << lo;                  /* bit k: byte k is the input's */
	const uint32_t d = lo < 8u ? (lo < 4u ? w.x : w.y) : (lo < 12u ? w.z : w.w);
	const uint32_t rep = ((d >> ((lo & 3u) * 8u)) & 0xffu) * 0x01010101u;
	/* four mask bits -> four mask bytes: n * 0x00204081 puts bit t of n at bit 8t */
	const uint32_t m0 = ((((vm      ) & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu;
	const uint32_t m1 = ((((vm >>  4) & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu;
	const uint32_t m2 = ((((vm >>  8) & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu;
	const uint32_t m3 = ((((vm >> 12) & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu;
	u32x4 r;
	r.x = (w.x & m0) | (rep & ~m0);
	r.y = (w.y & m1) | (rep & ~m1);
	r.z = (w.z & m2) | (rep & ~m2);
	r.w = (w.w & m3) | (rep & ~m3);
	return r;
}

/* 16 bytes of which only [lo, lo + cnt) belong to the input: the policy's chunk-level skip tests first
 * (on the filled chunk), then 16 predicated steps */
/* The self-loop-mask layouts with a spare class (ident_class = 31, bit 31 set in every state's mask): the bytes that do not
 * belong to the input are given THAT class -- a self-loop of every state -- and the chunk is then an ordinary one: the
 * class-level skip vote, 16 unpredicated steps.  One select per byte in the state-independent part instead of a compare +
 * a select per state register per byte on the dependent chain.  (The raw-byte range test runs first, on the filled chunk,
 * as before: on C3's inputs it skips most tails outright.) */
template <class Pol>
__device__ __forceinline__ auto step16_part_ident(const Pol &pol, typename Pol::S &st, const u32x4 &w, uint32_t lo, uint32_t cnt, int)
	-> decltype(pol.ident, bool())
{
	if (!pol.ident_ok()) return false;
	typename Pol::P pre[16];
#pragma unroll
	for (int k = 0; k < 16; k++) {
		const typename Pol::P c = pre_of(pol, w, k, 0);
		pre[k] = ((uint32_t)k - lo) < cnt ? c : (typename Pol::P)pol.ident;
	}
	if (skip_chunk(pol, st, pre, 0)) return true;
	walk_chunk(pol, st, pre, 0);      /* the policy's own chunk walk if it has one (Lds2Pol: pairs), else 16 x next() */
	return true;
}
template <class Pol>
__device__ __forceinline__ bool step16_part_ident(const Pol &, typename Pol::S &, const u32x4 &, uint32_t, uint32_t, long) { return false; }

template <class Pol>
__device__ __forceinline__ void step16_part(const Pol &pol, typename Pol::S &st, const u32x4 &w0, uint32_t lo, uint32_t cnt)
{
	u32x4 w = w0;
	if (has_skip_raw<Pol>(0) || has_skip_cls<Pol>(0)) {
		w = fill_invalid(w0, lo, cnt);
		if (skip_chunk_raw(pol, st, w, 0)) return;
	}
	if (step16_part_ident(pol, st, w0, lo, cnt, 0)) return;
	typename Pol::P pre[16];
#pragma unroll
	for (int k = 0; k < 16; k++) pre[k] = pre_of(pol, w, k, 0);
	if (skip_chunk(pol, st, pre, 0)) return;
#pragma unroll
	for (int k = 0; k < 16; k++) {
		const typename Pol::S nx = pol.next(st, pre[k]);
		st = pick(((uint32_t)k - lo) < cnt, nx, st); /* k < lo wraps: fails the test */
	}
}

/* An input's FIRST chunk when the policy says no byte is a self-loop of the start state (first_noskip(): anchored patterns --
 * C3): the chunk-level skip tests (fill + raw-range test + class-mask vote: ~80 vector instructions on the self-loop-mask
 * layouts) cannot pass, so the chunk goes straight to its lookups and the walk.  Plain walks only: every lane is in the start state. */
template <class Pol>
__device__ __forceinline__ auto first_noskip(const Pol &pol, int) -> decltype(pol.first_noskip()) { return pol.first_noskip(); }
template <class Pol>
__device__ __forceinline__ bool first_noskip(const Pol &, long) { return false; }

template <class Pol>
__device__ __forceinline__ auto step16_noskip(const Pol &pol, typename Pol::S &st, const u32x4 &w, uint32_t cnt, int) -> decltype(pol.ident, void())
{
	/* the bytes beyond the input's end get the spare class: one compare + one select per byte.  The byte count passes through an
	 * empty asm per byte: without it the compiler hoists the sixteen compares to the head of the tile (they depend on the
	 * input's length alone) and keeps their 32 scalar registers live across the walk -- registers this kernel does not have:
	 * they went to vector lanes and back, 2 + 2 more instructions per byte. */
	typename Pol::P pre[16];
#pragma unroll
	for (int k = 0; k < 16; k++) {
		uint32_t ck = cnt;
		asm volatile("" : "+v"(ck));
		const typename Pol::P c = pre_of(pol, w, k, 0);
		pre[k] = (uint32_t)k < ck ? c : (typename Pol::P)pol.ident;
	}
	walk_chunk(pol, st, pre, 0);
}
template <class Pol>
__device__ __forceinline__ void step16_noskip(const Pol &pol, typename Pol::S &st, const u32x4 &w, uint32_t cnt, long) { step16_part(pol, st, w, 0u, cnt); }

/* does the policy give the bytes beyond an input's end a class of their own (step16_part_ident)?  Then a partial chunk is
 * as cheap as a whole one */
template <class Pol>
constexpr auto tail_in_step(int) -> decltype(static_cast<const Pol *>(nullptr)->ident, true) { return true; }
template <class Pol> constexpr bool tail_in_step(long) { return false; }

/* ------------------------------------------------------------------ */
/* walk_generic: ragged lengths, any alignment, fixed stride or packed */
/* ------------------------------------------------------------------ */

/* 16 bytes of which [addr, limit) exist (out of line: the tiles that touch the batch's last bytes, and tiles whose 64
 * inputs span 4 GiB or more) */
__device__ __noinline__ u32x4 load_chunk_edge(uint64_t addr, bool want, uint64_t limit, uint64_t safe)
{
	typedef u32x4 __attribute__((aligned(1))) u32x4_any;      /* an input starts at any byte */
	typedef const u32x4_any __attribute__((address_space(1))) *glb_chunk_p;
	if (!want) return *(glb_chunk_p)safe;
	if (addr + 16u <= limit) return *(glb_chunk_p)addr;
	uint32_t d[4] = {0u, 0u, 0u, 0u};
	typedef const unsigned char __attribute__((address_space(1))) *glb_u8p;
	for (uint32_t k = 0; k < 15u && addr + k < limit; k++)
		d[k >> 2] |= (uint32_t)((glb_u8p)addr)[k] << ((k & 3u) * 8u);
	return u32x4{d[0], d[1], d[2], d[3]};
}

/* result write-back of the plain walk (no second per-state table, no resume): what the PLAIN instantiation of walk_generic
 * keeps live across its loop is then the two output pointers and the fin table */
__device__ __forceinline__ void write_result_plain(const WalkArgs &a, uint64_t word, uint64_t i, bool valid, uint32_t st)
{
	uint32_t end = FSMHIP_NO_MATCH;
	const uint32_t idx = fin_index(a, st);
	if (valid) end = a.fin[idx];
	if (valid && a.end_out != nullptr) a.end_out[i] = end;
	const uint64_t m = __ballot(valid && end != FSMHIP_NO_MATCH);
	if (a.bitmap != nullptr && (threadIdx.x & 63u) == 0 && word * 64u < a.n) a.bitmap[word] = m;
}

/*
 * One input per lane, 64 consecutive inputs per wavefront and step of a persistent loop: the kernel of the lines retest /
 * rx feed (a few dozen bytes each) and the fallback for everything else.  A step has ONE wait:
 *  - the offsets (or lengths) of the NEXT step's inputs are asked for at the top of a step;
 *  - an input's first NC = 4 chunks are loaded together from its own byte address (no partial chunk at the head).  Round 4:
 *    BUFFER loads through a resource re-based, per step, on the first input of the tile and bounded 4 bytes short of the
 *    batch's last byte (8 in round 4, which lost the last bytes of an input ending at total - 8 .. - 11: see generic_body32): the address of a chunk is one 32-bit offset + an immediate (the first version added 64-bit
 *    addresses and selected a safe address for the lanes without that chunk: ~20 vector instructions a step), a chunk that
 *    lies beyond the batch reads zeros instead of faulting, and the only tiles that take the out-of-line byte assembly are
 *    the ones that really touch the batch's last 8 bytes (or span 4 GiB);
 *  - the previous step's results are looked up (fin[]) and written under the same wait.
 * The walk itself, round 4: every lane walks its WHOLE chunks first -- unpredicated steps under the lane's own condition
 * -- and the one partial chunk an input can have (its last len % 16 bytes) takes ONE predicated step for all lanes at the
 * end.  (The first version ran chunk c for all lanes at once: with 64 lengths per wavefront some lane's input ended in every
 * chunk, so all four steps were predicated ones -- twice the instructions of a plain step for the column tables.)
 * Inputs longer than NC chunks go on one chunk at a time with the next one in flight.
 * PLAIN: no second output table, no resume state (launch.h picks it whenever that is so): fewer live scalars.
 */
/* FRONT: which metadata form the instantiation is for (FR_ANY: decided at run time from the arguments: every pointer of
 * every form then stays live across the loop) */
enum { FR_ANY = 0, FR_OFF64 = 1, FR_OFF32 = 2, FR_LENS = 3, FR_STRIDE = 4 };

/*
 * walk_lines32 (round 5, second half): walk_generic's walk for the batches retest / rx actually make -- a plain walk (end
 * states / accept bitmap) of lines packed back to back, the batch below 4 GiB and 2^29 lines -- with everything in 32 bits.
 * The host front knows a batch's size; a device front launches this kernel AND walk_generic (AND walk_ragged) and offsets_pick
 * says on the device which one runs.  What was measured on the way (24e6 lines of 8-64 bytes on the C3 table, profiles/r08*):
 *  - the per-tile prologue.  The ISA accounting of walk_generic had ~170 instructions of prologue
 *    and ~60 of loop control around a walk of 32-70: 64-bit offsets, a window re-based per tile with its readfirstlanes, a
 *    three-way slow test, clamped metadata indices, a run-time resume test.  Here ONE buffer resource spans the batch and one
 *    the metadata (a chunk's address is the line's 32-bit byte offset + an immediate; an index beyond n reads zeros), u64
 *    offsets are read as their low halves by one 12-byte load, the edge test is one compare, nothing of resume / ids / eager
 *    sets is looked at.  Alone that was worth 0-10 % (0.485 -> 0.485 / 0.44 ms by front): the kernel was not instruction-bound;
 *  - it is LATENCY-bound (a tile's chunks are asked for and waited for; SQ_WAIT_ANY 0.6 of the wave cycles, VALU 58 % busy), so
 *    what counts is wavefronts per CU: 76 registers let ONE 16-wavefront workgroup on a CU (4 per SIMD) where two of 12 fit
 *    (6 per SIMD): fsm_hip.hip sizes the workgroup from the kernel's register count -- 0.485 -> 0.386 ms; <= 80 registers held
 *    by __launch_bounds__(1024, 6);
 *  - the previous tile's fin[] lookup is asked for first and its stores come after the walk (loads return in order: nothing that
 *    is waited for may sit behind the chunks); a lane that lacks chunk j asks beyond the resource (zeros, no memory request:
 *    the texture addresser was 55-64 % busy with every lane asking for four chunks): 0.386 -> 0.350 ms = 0.41 of HBM peak on
 *    sum(len) + 12 bytes a line (walk_generic: 0.32); 8-16 byte lines 0.294 -> 0.199 ms;
 *  - the light policies (column tables, lds, comb ...) have the registers for a SECOND tile in flight (PF): tile t + 1's chunks
 *    are asked for before tile t is walked: + 10 % on 8-16 byte lines, nothing on 8-64 (the addresser again);
 *  - an input's FIRST chunk skips the chunk-level skip tests when no byte is a self-loop of the start state (step16_noskip);
 *  - the lengths front's prefix sum is seven DPP adds (wave_excl_prefix32), its tile base a scalar load.
 * The resource ends 4 bytes short of the batch and the <= 8 lines that end in its last 8 bytes walk their last <= 23 bytes
 * byte by byte (see the note at `nrec` below: round 4's bound lost bytes).
 */
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

/* a tile's results in two halves: the fin[] lookup, and -- once it is back -- the stores */
struct ResPend32 { uint32_t end; bool valid; };
__device__ __forceinline__ ResPend32 result_load32(const WalkArgs &a, uint32_t tile, uint32_t st, uint32_t n)
{
	ResPend32 r;
	r.valid = tile * 64u + (threadIdx.x & 63u) < n;
	r.end = FSMHIP_NO_MATCH;
	const uint32_t idx = fin_index(a, st);
	if (r.valid) r.end = a.fin[idx];
	return r;
}
__device__ __forceinline__ void result_store32(const WalkArgs &a, uint32_t tile, const ResPend32 &r)
{
	const uint32_t lane = threadIdx.x & 63u, i = tile * 64u + lane;
	if (r.valid && a.end_out != nullptr) a.end_out[i] = r.end;
	const uint64_t m = __ballot(r.valid && r.end != FSMHIP_NO_MATCH);
	if (a.bitmap != nullptr && lane == 0) a.bitmap[tile] = m;
}

template <class Pol, int FRONT, bool PF>
__device__ __forceinline__ void generic_body32(const WalkArgs &a, const Pol &pol, const uint32_t total_v)
{
	constexpr uint32_t NC = 4;
	/* (a value loaded from global memory sits in a vector register even when every lane loaded the same word, and a buffer
	 * resource built from it makes every load through it a waterfall loop) */
	const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total_v);
	/* (the workgroup's size comes out of the dispatch packet by a vector load: said to be uniform, or the tile counter, the
	 * stride and everything indexed by them sit in vector registers -- the lengths front's build did that and spilled) */
	const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const uint32_t nw = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6));
	const uint32_t n = (uint32_t)a.n, ntiles = (n + 63u) >> 6, tstride = (uint32_t)__builtin_amdgcn_readfirstlane((int)(gridDim.x * nw));
	/* (the saturating subtraction is a vector instruction: back to a scalar register, or the resource is a vector one) */
	const uint32_t lim8 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(total >= 8u ? total - 8u : 0u));
	/* The resource ends 4 bytes short of the batch, the edge test is at 8: gfx950 returns a dword of a buffer load only when the
	 * WHOLE dword lies inside the resource (offset + 4 <= num_records; tests/test_gpu_round5.py pins it: with the resource
	 * ending at total - 8, as round 4 had it, an input that ended at total - 8 .. total - 11 lost its last bytes), and a dword
	 * that starts at any byte alignment inside [.., total - 4) never reaches past the batch's last byte.
	 * (a.early & 128: round 4's bound, for that test) */
	const uint32_t nrec = (a.early & 128u) ? lim8 : (uint32_t)__builtin_amdgcn_readfirstlane((int)(total >= 4u ? total - 4u : 0u));
	const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.base), 0, (int)nrec, 0x00020000);
	const void *mp = FRONT == FR_OFF64 ? static_cast<const void *>(a.off) : FRONT == FR_OFF32 ? static_cast<const void *>(a.off32) : static_cast<const void *>(a.len);
	const uint32_t mbytes = FRONT == FR_OFF64 ? (n + 1u) * 8u : FRONT == FR_OFF32 ? (n + 1u) * 4u : n * 4u;
	const __amdgpu_buffer_rsrc_t meta = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(mp), 0, (int)mbytes, 0x00020000);

	const bool noskip0 = first_noskip(pol, 0) && !(a.early & 64u);    /* (a.early & 64: off, for A/B runs) */
	uint32_t nb = 0, ne = 0, ntb = 0, nhi = 0;
	auto fetch = [&](uint32_t tile) {
		const uint32_t i = tile * 64u + lane;            /* beyond n: zeros come back (lengths 0; offsets: masked below) */
		if (FRONT == FR_OFF64) {
			/* (the offset's high half is looked at -- it is 0 in a batch below 4 GiB; an input whose is not counts as empty -- so that
			 * its register stays the load's until the load is back: the compiler took a dead middle register of a load in flight
			 * for a temporary, and the write-after-write wait that costs took a prefetched chunk with it) */
			const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(meta, (int)(i * 8u), 0, 0);
			nb = v.x; nhi = v.y; ne = v.z;
		} else if (FRONT == FR_OFF32) {
			const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(meta, (int)(i * 4u), 0, 0);
			nb = v.x; ne = v.y;
		} else {
			ne = __builtin_amdgcn_raw_buffer_load_b32(meta, (int)(i * 4u), 0, 0);
			/* (read as constant memory -- the pre-pass that wrote it is an earlier kernel -- so that it is a scalar load into a
			 * scalar register: as ordinary global memory it was a vector load of one address and a vector register per lane) */
			typedef const uint64_t __attribute__((address_space(4))) *const_u64p;
			ntb = (uint32_t)reinterpret_cast<const_u64p>(reinterpret_cast<uintptr_t>(a.tbase))[tile < ntiles ? tile : ntiles];
		}
	};
	/* where the inputs of the tile whose metadata was fetched last lie.  lenw = the bytes that are walked in chunks: all of them,
	 * except for an input that ends within 8 bytes of the batch's end (at most eight inputs of a batch): its whole chunks that lie
	 * inside the resource are walked with everybody's, its last <= 23 bytes afterwards, fetched a byte at a time through a
	 * resource that ends where the batch does */
	struct Where { uint32_t beg, len, lenw; };
	auto prepare = [&](uint32_t tile) -> Where {
		const uint32_t i = tile * 64u + lane;
		Where t;
		if (FRONT == FR_LENS) { t.len = ne; t.beg = ntb + wave_excl_prefix32(t.len); }
		else { t.beg = nb; t.len = i < n && (FRONT != FR_OFF64 || nhi == 0u) ? ne - nb : 0u; }
		const bool edge = t.len != 0u && t.beg + t.len > lim8;
		t.lenw = edge ? (t.beg < lim8 ? (lim8 - t.beg) & ~15u : 0u) : t.len;
		return t;
	};
	/* an input's first NC chunks, asked for together.  A chunk NO input of the tile has is skipped by a wave-uniform branch (8-16
	 * byte lines: one load, not four); in a chunk that some have, an input that does not asks at an offset beyond the resource:
	 * zeros, no memory request (a.early & 4096: every lane asks at its own offset, as the first version did -- A/B runs) */
	auto issue = [&](const Where &t, u32x4 (&w)[NC]) {
		const uint32_t nch = (t.lenw + 15u) >> 4;
		const bool all_ask = (a.early & 4096u) != 0u;
#pragma unroll
		for (uint32_t j = 0; j < NC; j++) {
			w[j] = u32x4{0u, 0u, 0u, 0u};
			if (j == 0 || __any(j < nch))
				w[j] = __builtin_amdgcn_raw_buffer_load_b128(win, (int)(j < nch || all_ask ? t.beg + 16u * j : 0xFFFFFFF0u), 0, 0);
		}
	};

	uint32_t tile = blockIdx.x * nw + wave;
	if (tile >= ntiles) return;
	fetch(tile);
	Where cur = {0u, 0u, 0u}, nxt = {0u, 0u, 0u};
	u32x4 wq[NC], wn[NC];
	if (PF) {
		/* two tiles in flight: the chunks of tile t + 1 are asked for before tile t is walked (light policies: the registers are there) */
		cur = prepare(tile);
		fetch(tile + tstride);
		issue(cur, wq);
	}
	bool pend = false;
	uint32_t ptile = 0, pcode = 0;
	for (; tile < ntiles; tile += tstride) {
		const uint32_t i = tile * 64u + lane;
		const bool valid = i < n;
		/* the previous tile's results: the fin[] lookup is asked for FIRST (loads come back in order: what is waited for must not
		 * sit behind the next tile's chunks) */
		ResPend32 rp = {FSMHIP_NO_MATCH, false};
		if (pend) rp = result_load32(a, ptile, pcode, n);
		if (PF) {
			nxt = prepare(tile + tstride);
			fetch(tile + 2u * tstride);
			issue(nxt, wn);
		} else {
			cur = prepare(tile);
			fetch(tile + tstride);
			issue(cur, wq);
		}
		const uint32_t beg = cur.beg, len = cur.len, lenw = cur.lenw;
		const bool edge = lenw != len;
		const bool any_edge = __any(edge);
		const uint32_t nfull = lenw >> 4, tail = lenw & 15u, nchunks = (lenw + 15u) >> 4;
		typename Pol::S st[1] = { init_state(pol, a.start, a, (uint64_t)i, valid, 0) };
		auto load_chunk = [&](uint32_t c) -> u32x4 {
			return __builtin_amdgcn_raw_buffer_load_b128(win, (int)(beg + 16u * c), 0, 0);
		};
		if (tail_in_step<Pol>(0)) {
#pragma unroll 1
			for (uint32_t c = 0; c < NC; c++) {
				if (!__any(c < nchunks)) break;
				if (c < nchunks) {
					if (c == 0u && noskip0) {
						step16_noskip(pol, st[0], wq[0], nfull != 0u ? 16u : tail, 0);
					} else if (__all(c < nfull)) {
						const u32x4 w1[1] = { wq[0] };
						step16<Pol, 1>(pol, st, w1);
					} else {
						step16_part(pol, st[0], wq[0], 0u, c < nfull ? 16u : tail);
					}
				}
				wq[0] = wq[1]; wq[1] = wq[2]; wq[2] = wq[3];
			}
			if (__any(nchunks > NC)) {
				u32x4 w[1] = { load_chunk(NC) };
				for (uint32_t c = NC; __any(c < nchunks); c++) {
					if (c < nchunks) {
						const u32x4 wn = load_chunk(c + 1u);
						if (__all(c < nfull)) step16<Pol, 1>(pol, st, w);
						else step16_part(pol, st[0], w[0], 0u, c < nfull ? 16u : tail);
						w[0] = wn;
					}
					if ((a.early & 1u) && __all(Pol::code(st[0]) >= a.abs_min || c + 1u >= nchunks)) break;
				}
			}
		} else {
			u32x4 tw = wq[0];
#pragma unroll
			for (uint32_t c = 0; c < NC; c++) {
				if (!__any(c < nfull)) break;
				if (c < nfull) {
					const u32x4 w1[1] = { wq[c] };
					step16<Pol, 1>(pol, st, w1);
				}
				if (c + 1u < NC && nfull == c + 1u) tw = wq[c + 1u];
			}
			if (__any(nchunks > NC)) {
				u32x4 w[1] = { load_chunk(NC) };
				for (uint32_t c = NC; __any(c < nchunks); c++) {
					if (c < nchunks) {
						const u32x4 wn = load_chunk(c + 1u);
						if (c < nfull) step16<Pol, 1>(pol, st, w);
						else tw = w[0];
						w[0] = wn;
					}
					if ((a.early & 1u) && __all(Pol::code(st[0]) >= a.abs_min || c + 1u >= nchunks)) break;
				}
			}
			if (__any(tail != 0u)) {
				if (tail != 0u) step16_part(pol, st[0], tw, 0u, tail);
			}
		}
		if (any_edge) {
			/* (four bytes a turn, the turns not unrolled: this runs once per launch, and the registers of an unrolled form -- 24
			 * offsets + 24 bytes -- would set the whole kernel's count) */
			const __amdgpu_buffer_rsrc_t wint = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.base), 0, (int)total, 0x00020000);
			const uint32_t rem = edge ? len - lenw : 0u, from = beg + lenw;
#pragma unroll 1
			for (uint32_t k0 = 0; __any(k0 < rem); k0 += 4u) {
				uint32_t b[4];
#pragma unroll
				for (uint32_t k = 0; k < 4u; k++)     /* out of range (0xFFFFFFFF): zero, no memory request */
					b[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(wint, (int)(k0 + k < rem ? from + k0 + k : 0xFFFFFFFFu), 0, 0);
#pragma unroll
				for (uint32_t k = 0; k < 4u; k++) {
					const typename Pol::S nx = pol.next(st[0], pol.pre(b[k] & 0xffu));
					st[0] = pick(k0 + k < rem, nx, st[0]);
				}
			}
		}
		finish_state(pol, a, (uint64_t)i, valid, st[0], 0);
		/* ... and their stores come after the walk: the oldest operations still counted when the NEXT walk waits for its chunks are
		 * then these stores, a whole walk old (a store asked for before the walk would sit between the chunks and the prefetch in
		 * the in-order count, and the wait for the chunks would take it, or a prefetched chunk, along) */
		if (pend) result_store32(a, ptile, rp);
		pend = true;
		ptile = tile;
		pcode = Pol::code(st[0]);
		if (PF) {
			cur = nxt;
#pragma unroll
			for (uint32_t j = 0; j < NC; j++) wq[j] = wn[j];
		}
	}
	if (pend) {
		const ResPend32 rp = result_load32(a, ptile, pcode, n);
		result_store32(a, ptile, rp);
	}
}

/* the policies whose walk leaves the registers for a second tile in flight (the self-loop-mask layouts and the record walk sit at
 * 70-80 registers without it) */
template <class Pol> struct lines_prefetch { static constexpr bool value = true; };
template <> struct lines_prefetch<CombSelfPol> { static constexpr bool value = false; };
template <> struct lines_prefetch<LdsSelfPol> { static constexpr bool value = false; };
template <> struct lines_prefetch<SparsePol> { static constexpr bool value = false; };
template <> struct lines_prefetch<TinyPol<uint64_t>> { static constexpr bool value = false; };   /* sixteen 64-bit columns a chunk */

/* Which policies this kernel takes: all of them.  Round 5 kept the record walk (SparsePol) on walk_generic: one build of
 * walk_lines32<SparsePol> lost the state of a lane between an input's first and second chunk in ~45 % of its launches (8+
 * wavefronts per workgroup; inputs whose match straddles byte 16), the builds before and after it -- same walk source, a
 * different order of the loads around it -- in none of 300 (profiles/r08i_*).  Its per-byte loop mixed FLAT loads that land in
 * LDS or in memory (a record fetched through a generic pointer) with global loads under complementary exec masks.  Round 6
 * rewrote every such fetch with explicit address spaces (SparsePol::next_t: no FLAT instruction is left in any walk kernel) and
 * runs the harness that showed the loss as a test (tests/test_gpu_round6.py: 1 200 launches of this instantiation by workgroup
 * size); see DESIGN.md section 4 for what is and is not known about the cause. */
template <class Pol> struct lines32_ok { static constexpr bool value = true; };

/* the kernel around it: a packed front (FR_OFF64 / FR_OFF32 / FR_LENS), plain outputs, a batch below 4 GiB and 2^29 inputs --
 * the host front knows that, a device front launches this kernel AND walk_generic and offsets_pick says which one runs */
/* (six wavefronts per SIMD = two 12-wavefront workgroups per CU beside a table of up to 80 KB: <= 80 vector registers.  The
 * self-loop-mask layouts sit at 75-82 on their own; every other layout is far below) */
template <class Pol, int FRONT>
__global__ void __launch_bounds__(1024, 6)
walk_lines32(const WalkArgs a)
{
	if (a.skip_flag != nullptr && *a.skip_flag != a.run_when) return;   /* another kernel took the batch */
	extern __shared__ __align__(16) unsigned char lds[];
	Pol pol;
	pol.setup(lds, a);
	__syncthreads();
	const uint32_t total = FRONT == FR_OFF64 ? (uint32_t)a.off[a.n] : FRONT == FR_OFF32 ? a.off32[a.n] : (uint32_t)a.tbase[(a.n + 63u) / 64u];
	if (lines_prefetch<Pol>::value && !(a.early & 256u)) generic_body32<Pol, FRONT, true>(a, pol, total);   /* (a.early & 256: off, for A/B runs) */
	else generic_body32<Pol, FRONT, false>(a, pol, total);
}

template <class Pol, int MAXT = 1024, bool PLAIN = false, int FRONT = FR_ANY>
__global__ void __launch_bounds__(MAXT)
walk_generic(const WalkArgs a)
{
	if (a.skip_flag != nullptr && *a.skip_flag != a.run_when) return;   /* another kernel took the batch */
	extern __shared__ __align__(16) unsigned char lds[];
	Pol pol;
	pol.setup(lds, a);
	__syncthreads();

	constexpr uint32_t NC = 4;
	const bool f_off = FRONT == FR_ANY ? a.off != nullptr : FRONT == FR_OFF64;
	const bool f_off32 = FRONT == FR_ANY ? a.off == nullptr && a.off32 != nullptr : FRONT == FR_OFF32;
	const bool f_lens = FRONT == FR_ANY ? a.off == nullptr && a.off32 == nullptr && a.tbase != nullptr : FRONT == FR_LENS;
	const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
	const uint64_t ntiles = (a.n + 63u) / 64u, tstride = (uint64_t)gridDim.x * nw;
	const uint64_t base = reinterpret_cast<uint64_t>(a.base);
	/* one past the batch's last byte: no load may reach beyond it */
	const uint64_t total = f_off ? a.off[a.n] : f_off32 ? a.off32[a.n] : f_lens ? a.tbase[(a.n + 63u) / 64u] : a.n * a.stride, limit = base + total;
	const uint64_t safe = reinterpret_cast<uint64_t>(a.btab);   /* 1 KiB that is always there: what a lane without a chunk reads on the slow path */

	/* the offsets / lengths of a step's inputs, asked for one step ahead (clamped indices: the loads are unconditional) */
	uint64_t nb = 0, ne = 0, ntb = 0;
	uint32_t nl = 0, nb32 = 0, ne32 = 0;
	auto fetch = [&](uint64_t tile) {
		const uint64_t i = tile * 64u + lane, ic = i < a.n ? i : a.n - 1u;
		if (f_off) { nb = a.off[ic]; ne = a.off[ic + 1u]; }
		else if (f_off32) { nb32 = a.off32[ic]; ne32 = a.off32[ic + 1u]; }
		else if (f_lens) { nl = a.len[ic]; ntb = a.tbase[tile < ntiles ? tile : ntiles]; }
		else if (a.len != nullptr) nl = a.len[ic];
	};
	auto result = [&](uint64_t word, uint64_t i, bool valid, uint32_t code) {
		if (PLAIN) write_result_plain(a, word, i, valid, code);
		else write_result(a, word, i, valid, code);
	};

	uint64_t tile = (uint64_t)blockIdx.x * nw + wave;
	fetch(tile);
	bool pend = false;                 /* wave-uniform: the previous step's results are not written yet */
	uint64_t ptile = 0, pi = 0;
	bool pvalid = false;
	uint32_t pcode = 0;
	for (; tile < ntiles; tile += tstride) {
		const uint64_t i = tile * 64u + lane;
		const bool valid = i < a.n;
		uint64_t beg = 0, len = 0;
		if (f_off) { beg = nb; len = ne - nb; }
		else if (f_off32) { beg = nb32; len = ne32 - nb32; }
		else if (f_lens) { len = valid ? nl : 0u; beg = ntb + wave_excl_prefix((uint32_t)len, lane); }
		else { beg = i * a.stride; len = a.len != nullptr ? nl : a.stride; }
		if (!valid) len = 0;
		fetch(tile + tstride);
		/* the tile's window: from its first input's first byte (inputs lie in index order: lane 0's) */
		const uint64_t tb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(beg >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)beg);
		if (!valid) beg = tb;
		const uint64_t rel64 = beg - tb, nfull = len >> 4;
		const uint32_t tail = (uint32_t)len & 15u;
		const uint64_t nchunks = nfull + (tail != 0u ? 1u : 0u);
		typename Pol::S st[1] = { init_state(pol, start_code(a, i, valid), a, i, valid, 0) };
		/* slow: an input of this step ends within 8 bytes of the batch's end, or the step's inputs reach 4 GiB beyond its first byte
		 * (then 32-bit offsets do not do), or lie out of order */
		const bool slow = __any(nchunks != 0 && (beg + len + 8u > total || rel64 + 16u * nchunks >= 0xFFFFFF00ull || beg < tb));
		/* (4 bytes short of the batch, the slow test above at 8: a dword comes back only when all of it lies inside the resource --
		 * see generic_body32; a.early & 128: round 4's bound of 8) */
		const uint64_t wcut = (a.early & 128u) ? 8u : 4u;
		const uint64_t wbytes = total - tb >= wcut ? total - tb - wcut : 0u;
		const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base + tb), 0, (int)(wbytes < 0xFFFFFFF0ull ? (uint32_t)wbytes : 0xFFFFFFF0u), 0x00020000);
		const uint32_t rel = (uint32_t)rel64;
		const uint64_t p0 = base + beg;
		auto load_chunk = [&](uint64_t c) -> u32x4 {
			if (slow) return load_chunk_edge(p0 + 16u * c, c < nchunks, limit, safe);
			return __builtin_amdgcn_raw_buffer_load_b128(win, (int)(rel + 16u * (uint32_t)c), 0, 0);
		};
		u32x4 wq[NC];
#pragma unroll
		for (uint32_t j = 0; j < NC; j++) {
			/* a chunk that NO input of this step has is not asked for (wave-uniform: 8-16 byte lines need one load, not four) */
			wq[j] = u32x4{0u, 0u, 0u, 0u};
			if (j == 0 || (a.early & 16u) || __any(j < nchunks)) wq[j] = load_chunk(j);   /* (a.early & 16: always, for A/B runs) */
		}
		/* the previous step's results: their fin[] lookup is in flight with this step's chunks */
		if (pend) result(ptile, pi, pvalid, pcode);
		if (tail_in_step<Pol>(0)) {
			/* self-loop-mask layouts with a spare class: the bytes beyond an input's end become that class in the state-
			 * independent part of a step, so a partial chunk costs what a whole one does -- chunk c of every lane in one step */
			/* (one copy of the step code, the four chunks rotated through wq[0]: unrolled, the two forms of a step times four
			 * were 55 KB of a 64 KB instruction cache that two CUs share) */
#pragma unroll 1
			for (uint32_t c = 0; c < NC; c++) {
				if (!__any(c < nchunks)) break;
				if (c < nchunks) {
					if (__all(c < nfull)) {
						const u32x4 w1[1] = { wq[0] };
						step16<Pol, 1>(pol, st, w1);
					} else {
						step16_part(pol, st[0], wq[0], 0u, c < nfull ? 16u : tail);
					}
				}
				wq[0] = wq[1]; wq[1] = wq[2]; wq[2] = wq[3];
			}
			if (__any(nchunks > NC)) {
				u32x4 w[1] = { load_chunk(NC) };
				for (uint64_t c = NC; __any(c < nchunks); c++) {
					if (c < nchunks) {
						const u32x4 wn = load_chunk(c + 1u);   /* next chunk in flight */
						if (__all(c < nfull)) step16<Pol, 1>(pol, st, w);
						else step16_part(pol, st[0], w[0], 0u, c < nfull ? 16u : tail);
						w[0] = wn;
					}
					if ((a.early & 1u) && __all(Pol::code(st[0]) >= a.abs_min || c + 1 >= nchunks)) break;
				}
			}
		} else {
			/* whole chunks, each lane its own (unpredicated steps under the lane's condition); the partial one is kept for the end */
			u32x4 tw = wq[0];
#pragma unroll
			for (uint32_t c = 0; c < NC; c++) {
				if (!__any(c < nfull)) break;
				if (c < nfull) {
					const u32x4 w1[1] = { wq[c] };
					step16<Pol, 1>(pol, st, w1);
				}
				if (c + 1u < NC && nfull == c + 1u) tw = wq[c + 1u];
			}
			if (__any(nchunks > NC)) {
				u32x4 w[1] = { load_chunk(NC) };
				for (uint64_t c = NC; __any(c < nchunks); c++) {
					if (c < nchunks) {
						const u32x4 wn = load_chunk(c + 1u);   /* next chunk in flight */
						if (c < nfull) step16<Pol, 1>(pol, st, w);
						else tw = w[0];
						w[0] = wn;
					}
					if ((a.early & 1u) && __all(Pol::code(st[0]) >= a.abs_min || c + 1 >= nchunks)) break;
				}
			}
			if (__any(tail != 0u)) {
				if (tail != 0u) step16_part(pol, st[0], tw, 0u, tail);
			}
		}
		finish_state(pol, a, i, valid, st[0], 0);
		pend = true;
		ptile = tile;
		pi = i;
		pvalid = valid;
		pcode = Pol::code(st[0]);
	}
	if (pend) result(ptile, pi, pvalid, pcode);
}

/* ------------------------------------------------------------------ */
/* walk_ragged: ragged lengths / packed offsets, coalesced + refilled   */
/* ------------------------------------------------------------------ */

/* per-lane result write (lanes finish at different times: no wavefront-wide ballot).  The bitmap
 * is filled with atomic ORs and must have been cleared on the launch stream. */
__device__ __forceinline__ void write_result_lane(const WalkArgs &a, uint64_t i, uint32_t st)
{
	const uint32_t idx = fin_index(a, st);
	const uint32_t end = a.fin[idx];
	if (a.end_out != nullptr) a.end_out[i] = end;
	if (a.out2 != nullptr) a.out2[i] = a.fin2[idx];
	if (a.state_io != nullptr) a.state_io[i] = a.orig_of[idx];
	if (a.bitmap != nullptr && end != FSMHIP_NO_MATCH)
		atomicOr(reinterpret_cast<unsigned long long *>(a.bitmap + (i >> 6)), 1ull << (i & 63u));
}

#define FSMHIP_RAGGED_RING 64u                                   /* staged (offset, length) pairs per wave (round 4: 64, topped up 32 at a time: 1 KiB less per wave) */
#define FSMHIP_RAGGED_WAVE_LDS (8192u + FSMHIP_RAGGED_RING * 16u + 1024u) /* 8 KiB tile + the ring + one 16-byte row record per lane */

/* Tiny5Pol's column table is 256 rows of 256 bytes (64 dword copies: the lookup address is formed by one byte
 * permutation, (byte << 8) | lane * 4) -- but a ds_read_b32 wave is served in two 32-lane groups over 32 banks
 * (MI355X_MICROARCH.md, LDS), so lanes l and l + 32 can share a copy: lanes 32-63 read copies 0-31 and the upper
 * 128 bytes of every row are HOLES that no lookup touches.  walk_ragged keeps its ring and row records there
 * (a 64-entry ring there: 128 16-byte entries = 16 holes per wavefront), which leaves 8 KiB of LDS per wavefront next to the
 * 64 KiB table: 12 wavefronts per workgroup instead of 8 -- exactly the 160 KiB. */
template <class Pol> struct ragged_aux_in_holes { static constexpr bool value = false; };
template <> struct ragged_aux_in_holes<Tiny5Pol> { static constexpr bool value = true; };
#define FSMHIP_RAGGED_HOLE_WAVES 12u

/*
 * The retest / rx front: inputs of any length at any byte offset (packed back to back with an offsets
 * array, or fixed stride + lengths).  walk_generic gives every lane its own 16-byte loads with two
 * chunks in flight (latency-bound, 1.1-1.4 TB/s) and runs a wavefront until its longest input ends
 * (half the lane-steps idle at uniform 0..1024 B).  Here
 *  - input bytes arrive as in walk_ldsdma: per 128-byte segment of a lane's input, 8 adjacent loader
 *    lanes fetch its 8 16-byte pieces with ONE global_load_lds_dwordx4 (each row's source address is
 *    the owner lane's current position, left for the loaders in a per-lane LDS record; pieces beyond
 *    the input's last one are masked off), piece-rotated so the row-per-lane ds_read_b128 that
 *    follows is conflict-free; the next segment is in flight while the current one is walked;
 *  - the source of a piece is the input's own byte address + 16 * piece: global_load_lds_dwordx4 takes
 *    any byte alignment on gfx950 (tools/probes/unaligned_dma.hip), so an input's first byte is byte 0 of
 *    its piece 0 wherever the input starts and no chunk is partial at the head.  The last piece of an
 *    input whose length is not a multiple of 16 is fetched from (end - 16): it overlaps bytes already
 *    walked, which its one predicated step skips, and nothing outside the input is ever read.  (Inputs
 *    shorter than 16 bytes read [start, start + 16) when that stays inside the batch; the handful at the
 *    very end of the buffer are assembled from byte loads.)
 *  - a wavefront owns a contiguous range of inputs and REFILLS its lanes at every segment boundary:
 *    a lane whose input ends with the segment in hand (or sits in an absorbing state: fsm_exec's own
 *    early exit, exec.c:133-138) claims the next unclaimed input of the range -- ballot, popcount
 *    rank, no atomics -- so lanes stay busy whatever the length distribution.  The (offset, length)
 *    pairs of the next <= 64 inputs wait in an LDS ring that is topped up 32 at a time, one
 *    iteration ahead of their use (round 4: 64 and 32 -- a kilobyte less per wavefront).  With lengths
 *    alone, the byte offset of the next input to stage is carried along from the tile base the range starts at.
 * Results are written per lane when its input ends.  (Holding them back one iteration, so that the stores
 * go out right after the wait for the tile, measured no faster -- profiles/r02t_ragged_variants.txt -- and
 * that build of walk_ragged<EagerPol<TinyPol<u64>>> returned a wrong result in ~3 % of launches with two
 * wavefronts per SIMD: see the note in TinyPol::next.)
 */
/* FRONT as for walk_generic: an instantiation per metadata form keeps only that form's pointers live */
template <class Pol, int MAXT, int FRONT = FR_ANY>
__global__ void __launch_bounds__(MAXT)
walk_ragged(const WalkArgs a)
{
	const bool f_off = FRONT == FR_ANY ? a.off != nullptr : FRONT == FR_OFF64;
	const bool f_off32 = FRONT == FR_ANY ? a.off == nullptr && a.off32 != nullptr : FRONT == FR_OFF32;
	const bool f_lens = FRONT == FR_ANY ? a.off == nullptr && a.off32 == nullptr && a.tbase != nullptr : FRONT == FR_LENS;
	constexpr uint32_t RING = FSMHIP_RAGGED_RING, TOP = RING / 2u;   /* top-up granularity */
	if (a.skip_flag != nullptr && *a.skip_flag != a.run_when) return;   /* another kernel took the batch */
	extern __shared__ __align__(16) unsigned char lds[];
	constexpr bool HOLES = ragged_aux_in_holes<Pol>::value;
	Pol pol;
	pol.setup(lds, a);
	if constexpr (HOLES) pol.half_copies();
	__syncthreads();

	/* the wavefront's index as a SCALAR: everything derived from it (tile slot, input range, ring cursors) is then
	 * wave-uniform to the compiler too and lives in SGPRs / on the scalar unit */
	const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
	unsigned char *stg = lds + Pol::lds_bytes(a.tab_bytes) + wave * (HOLES ? 8192u : FSMHIP_RAGGED_WAVE_LDS);
	/* the wavefront's 16-byte auxiliary entries: [0, RING) the ring of (byte offset, length) pairs, [RING, RING + 64) the
	 * row records for the loaders -- behind the tile, or in the table's holes (8 entries per hole) */
	auto aux = [&](uint32_t e) -> unsigned char * {
		if (HOLES) return lds + (wave * ((RING + 64u) / 8u) + (e >> 3)) * 256u + 128u + (e & 7u) * 16u;
		return stg + 8192u + e * 16u;
	};

	/* contiguous range of whole bitmap words per wavefront */
	const uint64_t nwaves = (uint64_t)gridDim.x * nw, gw = (uint64_t)blockIdx.x * nw + wave;
	const uint64_t words = (a.n + 63u) / 64u, per = ((words + nwaves - 1u) / nwaves) * 64u;
	const uint64_t w_lo = gw * per < a.n ? gw * per : a.n;
	const uint64_t w_hi = w_lo + per < a.n ? w_lo + per : a.n;
	if (w_lo >= w_hi) return;
	/* one past the last byte of the batch: no 16-byte fetch may reach beyond it */
	const uint64_t limit = reinterpret_cast<uint64_t>(a.base) + (f_off ? a.off[a.n] : f_off32 ? a.off32[a.n] : f_lens ? a.tbase[(a.n + 63u) / 64u] : a.n * a.stride);

	const uint32_t lr = lane / 8u, lq = lane % 8u;                   /* loader role */
	unsigned char *rd = stg + (lane / 8u) * 1024u + (lane % 8u) * 128u;   /* reader role */
	const uint32_t rot = (lane >> 1) & 7u;
	const uint64_t lt = (1ull << lane) - 1ull;

	uint64_t staged = w_lo, next = w_lo;      /* wave-uniform: ring holds [next, staged) */
	u32x4 soff = {0u, 0u, 0u, 0u};            /* staging loads in flight: this lane's off[i], off[i + 1] ... */
	uint32_t slen = 0, s32a = 0, s32b = 0;    /* ... or its len[i], or its off32[i], off32[i + 1] */
	uint64_t stb = f_lens ? a.tbase[w_lo >> 6] : 0;   /* lengths only: the byte offset of the next input to be staged, carried along (w_lo is a multiple of 64) */
	uint32_t spend = 0;                       /* wave-uniform: how many pairs they are */

	bool have = false;                        /* this lane holds an input whose segment is in the tile */
	uint64_t ci = 0, csrc = 0;                /* its index, address of its piece kpos */
	uint32_t nfull = 0, tail = 0, kpos = 0;   /* whole 16-byte pieces, bytes after them, pieces walked so far */
	typename Pol::S st = init_state(pol, a.start, a, 0, false, 0);
	bool tile = false;                        /* wave-uniform: a segment is in flight */

	for (;;) {
		u32x4 w[8];
		if (tile || spend != 0) {
			__builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): tile and staged pairs have landed */
			__asm__ volatile("" ::: "memory");
		}
		if (spend != 0) {
			/* lengths only: where each of the staged inputs starts (every lane takes part in the prefix sum) */
			uint64_t pfx = 0;
			if (f_lens) {
				const uint32_t sl = lane < spend ? slen : 0u;
				pfx = stb + wave_excl_prefix(sl, lane);
				const uint64_t after = pfx + sl;      /* lane 63's: one past the last staged input */
				stb = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(after >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)after, 63);
			}
			if (lane < spend) {
				const uint64_t i = staged + lane;
				uint64_t b, l;
				if (f_off) {
					b = ((uint64_t)soff.y << 32) | soff.x;
					l = (((uint64_t)soff.w << 32) | soff.z) - b;
				} else if (f_off32) {
					b = s32a;
					l = s32b - s32a;
				} else if (f_lens) {
					b = pfx;
					l = slen;
				} else {
					b = i * a.stride;
					l = a.len != nullptr ? slen : a.stride;
				}
				uint64_t *re = reinterpret_cast<uint64_t *>(aux((uint32_t)i & (RING - 1u)));
				re[0] = b;
				re[1] = l;
			}
			staged += spend;
			spend = 0;
		}
		/* the piece after an input's last whole one holds its final (len % 16) bytes: it is read by its
		 * (lane-varying) index for the one predicated step below */
		u32x4 tw = {0u, 0u, 0u, 0u};
		if (tile) {
#pragma unroll
			for (uint32_t p = 0; p < 8; p++)
				w[p] = *reinterpret_cast<const u32x4 *>(rd + ((p + rot) & 7u) * 16u);
			tw = *reinterpret_cast<const u32x4 *>(rd + ((((nfull - kpos) & 7u) + rot) & 7u) * 16u);
		}
		__builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): tile in registers (slot reusable), ring written */
		__asm__ volatile("" ::: "memory");
		__builtin_amdgcn_wave_barrier();

		/* lanes whose input ends with the segment in hand */
		const uint32_t nch = nfull + (tail != 0u ? 1u : 0u);
		const bool fin = have && (nch - kpos <= 8u || ((a.early & 1u) && Pol::code(st) >= a.abs_min));
		const bool cont = have && !fin;
		/* refill: every lane that will be idle claims the next unclaimed input, in lane order */
		bool got = false, direct = false;
		uint64_t ni = 0, np0 = 0;
		uint32_t nnfull = 0, ntail = 0;
		uint64_t need = __ballot(!cont);
		while (need != 0 && next < staged) {   /* wave-uniform; repeats only over empty inputs */
			const uint64_t idx = next + (uint64_t)__builtin_popcountll(need & lt);
			const bool take = !cont && !got && idx < staged;
			const uint64_t want = (uint64_t)__builtin_popcountll(need);
			next = next + want < staged ? next + want : staged;
			if (take) {
				const uint64_t *re = reinterpret_cast<const uint64_t *>(aux((uint32_t)idx & (RING - 1u)));
				const uint64_t beg = re[0], len = re[1];
				np0 = reinterpret_cast<uint64_t>(a.base) + beg;
				nnfull = (uint32_t)(len >> 4);
				ntail = (uint32_t)len & 15u;
				ni = idx;
				if (len == 0) {
					/* empty input: accepted iff the start state is an end state; no bytes to fetch */
					const typename Pol::S e = init_state(pol, start_code(a, idx, true), a, idx, true, 0);
					write_result_lane(a, idx, Pol::code(e));
					finish_state(pol, a, idx, true, e, 0);
				} else {
					got = true;
					if (nnfull == 0u && np0 + 16u > limit) {
						/* fewer than 16 bytes, less than 16 bytes before the end of the batch: byte loads, placed
						 * where the DMA would have put piece 0 of this lane's row */
						uint32_t d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
						for (uint32_t k = 0; k < 15; k++)
							if (k < ntail) d[k >> 2] |= (uint32_t)((const unsigned char __attribute__((address_space(1))) *)np0)[k] << ((k & 3u) * 8u);
						const u32x4 dv = {d[0], d[1], d[2], d[3]};
						*reinterpret_cast<u32x4 *>(rd + (rot & 7u) * 16u) = dv;
						direct = true;
					}
				}
			}
			need = __ballot(!cont && !got);
		}

		/* top the ring up, one iteration ahead of the claims that will read it -- and BEFORE this iteration's tile
		 * requests go out: whatever wait the compiler attaches to these loads then has nothing of the tile to wait for */
		if (staged - next < TOP && staged < w_hi) {
			const uint64_t c = w_hi - staged < TOP ? w_hi - staged : TOP;
			/* The loads land in registers of their own and nothing is computed from them here: any arithmetic
			 * (or a copy into a variable shared by the two fronts) makes the compiler wait for them -- and with
			 * them for the tile requests issued just above -- on the spot. */
			if (lane < c) {
				const uint64_t i = staged + lane;
				const u32x4 *po = reinterpret_cast<const u32x4 *>(a.off + i);   /* off[i] and off[i + 1] */
				const uint32_t *pl = a.len + i;                                  /* (both addresses first: a temporary formed
				                                                                 * after one load would be ordered behind it) */
				const uint32_t *p32 = a.off32 + i;
				if (f_off) soff = *po;
				else if (f_off32) { s32a = p32[0]; s32b = p32[1]; }
				else if (a.len != nullptr) slen = *pl;
			}
			spend = (uint32_t)c;
		}

		/* source, piece budget and last-piece pull-back of every row's next segment: each lane leaves a
		 * 16-byte record for the 8 loader lanes of its row (one ds_write + 8 broadcast ds_reads per lane and
		 * ONE wait, where cross-lane shuffles cost 3 per row and a wait each).  The record holds the source
		 * moved back by the pull-back, so that every piece but the last adds it again: offsets stay >= 0. */
		const uint64_t lsrc = cont ? csrc + 128u : np0;
		const uint32_t lrem = cont ? nch - (kpos + 8u) : (got && !direct ? nnfull + (ntail != 0u ? 1u : 0u) : 0u);
		const uint32_t ltail = cont ? tail : ntail, lfull = cont ? nfull : nnfull;
		const uint32_t ladj = (ltail != 0u && lfull != 0u) ? 16u - ltail : 0u;
		const bool more = __any(lrem != 0u || direct);
		if (more) {
			const uint64_t lbase = lsrc - ladj;
			const u32x4 rec = {(uint32_t)lbase, (uint32_t)(lbase >> 32), lrem, ladj};
			*reinterpret_cast<u32x4 *>(aux(RING + lane)) = rec;
			u32x4 rr[8];
#pragma unroll
			for (uint32_t j = 0; j < 8; j++)
				rr[j] = *reinterpret_cast<const u32x4 *>(aux(RING + j * 8u + lr));
#pragma unroll
			for (uint32_t j = 0; j < 8; j++) {
				const uint32_t piece = (lq - ((j * 4u + (lr >> 1)) & 7u)) & 7u;   /* rotation of row j * 8 + lr: (row >> 1) & 7 */
				if (piece < rr[j].z) {
					const uint32_t o = piece * 16u + (piece + 1u == rr[j].z ? 0u : rr[j].w);
					const unsigned char *src = reinterpret_cast<const unsigned char *>(((uint64_t)rr[j].y << 32) | rr[j].x) + o;
					__builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(stg + j * 1024u), 16, 0, 0);
				}
			}
		}
		/* walk the segment in hand: its whole pieces, then -- for the lanes whose input ends here with a
		 * partial piece -- one predicated step, for all of them at once (predicating every byte of every
		 * chunk instead costs 7 VALU operations per byte: with 64 ragged lanes some lane is nearly always
		 * in a partial chunk) */
		if (tile && have) {
			const uint32_t m = nfull - kpos;
#pragma unroll
			for (uint32_t p = 0; p < 8; p++) {
				if (p < m) {
					typename Pol::S s1[1] = { st };
					const u32x4 w1[1] = { w[p] };
					step16<Pol, 1>(pol, s1, w1);
					st = s1[0];
				}
			}
			if (tail != 0u && m < 8u)
				step16_part(pol, st, tw, nfull != 0u ? 16u - tail : 0u, tail);
		}
		if (fin) {
			write_result_lane(a, ci, Pol::code(st));
			finish_state(pol, a, ci, true, st, 0);
		}
		if (cont) {
			kpos += 8u;
			csrc += 128u;
		} else if (got) {
			ci = ni; csrc = np0; nfull = nnfull; tail = ntail; kpos = 0;
			st = init_state(pol, start_code(a, ni, true), a, ni, true, 0);
		}
		have = cont || got;
		tile = more;
		if (!more && spend == 0 && next >= w_hi) break;
	}
}

} // namespace fsmhip


#endif
