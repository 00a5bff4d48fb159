/*
 * walk_lazy.h -- gfx950 device code: the LAZY walk of the sparse layout (plan.cpp build_lazy).
 *
 * The table of a 1e5-literal Aho-Corasick automaton (BASELINE configs[4]: ~1e6 states) does not fit LDS, and
 * SparseFastPol (walk_kernels.h) -- the record of every state entered is fetched, from LDS for the ~4 000 states
 * nearest the start state, by a 16-byte L2 gather for the rest -- is bound by those gathers: 0.33 L2 requests per
 * input byte keep the vector memory pipe 80 % busy (profiles/r04p_c5_memory_pipeline.txt), and 98 % of them only
 * confirm that the state has no exception on the next byte.  Here a state beyond the LDS set is entered WITHOUT its
 * record:
 *   - the walk state is (id, E): E = id for an LDS-resident state, else the LDS-resident record whose answers the
 *     state inherits (on a literal set: the deepest LDS-resident node on its failure chain, ac.c:229-241), obtained
 *     by arithmetic in the step that entered the state;
 *   - a byte is answered by E's 16-byte LDS record {bits, fm1, cf63}: exception (bit set) -> fm1 + rank, else
 *     cf63 - sh (= X + bit: the planner's most frequent target of the row); one 64-bit shift, two v_bcnt, a select;
 *   - whether the state's OWN record excepts the byte is asked of a one-hash Bloom filter in LDS keyed by
 *     (id, class) -- 64 KiB of bits for the ~1e5 exceptions of the ~8e4 depth-3 nodes: 17 % false positives -- and
 *     only a set bit costs the gather of the own record (a buffer load whose offset is pushed out of range for the
 *     other lanes: they get zeros, i.e. "no exception", and make no memory request);
 *   - any answer with bit 31 set is a SENTINEL (non-consecutive targets, a class without a bit, a state reached in
 *     two ways...): the 16-byte chunk is then re-walked for those lanes with the exact records of build_sparse in
 *     device memory.  The planner proves the rest (plan.cpp, tests/test_plan.py).
 * One workgroup of 16 waves per CU next to a 131 KiB table (64 KiB filter + 4 162 records + the byte map), several inputs
 * per lane for the instruction-level parallelism the second workgroup used to give (two in rounds 4-5, THREE since round
 * 6); per-lane 16-byte input loads, four chunks per row in flight (as walk_direct_np).  Fixed-stride rows, plain
 * (non-eager, non-resumed) walks.
 * Round 6: 888 -> 1 036 GB/s on the 1e5-literal automaton (1e7 x 1 KiB rows), in three steps that needed one another:
 *   - plan.cpp's CLONES: no own record sends a hit the exact way any more (44 816 of them did: a literal planted in a row's
 *     tail re-walked its chunk with the wavefront waiting) -- 934;
 *   - the byte -> shift lookups eight bytes at a time, their OR into the sentinel word taken at once: the compiler kept all
 *     sixteen shifts of every input alive to the block's end for that OR -- 1 000 with two inputs, and registers for
 *   - a third input per lane (128 registers, no scratch) -- 1 036.  (Four: 758.  profiles/r09i_*, r09j_*.)
 * What bounds it there: 25.2 vector instructions per byte-step at 4 cycles each = 1.54 TB/s at full VALU issue; the walk
 * reaches two thirds of that with 16 wavefronts per CU, the most a 131 KiB table leaves room for.
 */
#ifndef FSM_HIP_WALK_LAZY_H
#define FSM_HIP_WALK_LAZY_H

#include "walk_kernels.h"

namespace fsmhip {

#define FSMHIP_LAZY_SH_BYTES 1024u      /* sh[256] (u32: a byte-wide table measured slower, profiles/r06c_*) sits at LDS address 0, the filter right behind it */

/* one exact step through build_sparse's records in DEVICE memory (no LDS mirror, the FULLBASE shortcut not taken:
 * the chain simply goes on to the base) */
__device__ __forceinline__ uint32_t sparse_next_global(const uint32_t *img, uint32_t st, uint32_t byte, uint32_t abs_min)
{
	const uint16_t *pm = reinterpret_cast<const uint16_t *>(img + 16);
	const unsigned char *g = reinterpret_cast<const unsigned char *>(img);
	const u32x4 *grec = reinterpret_cast<const u32x4 *>(g + img[5]);
	const uint32_t *gdense = reinterpret_cast<const uint32_t *>(g + img[6]);
	const uint32_t *exc = reinterpret_cast<const uint32_t *>(g + img[7]);
	const uint32_t p = pm[byte], cls = p & 0xffu, bit = p >> 8;
	const bool hasbit = bit < 64u;
	const uint64_t sel = hasbit ? (uint64_t)1 << (bit & 63u) : 0u, below = hasbit ? sel - 1u : 0u;
	uint32_t res = st;
	bool live = st < abs_min;
	while (live) {
		const u32x4 r = grec[st];
		const uint64_t bits = (uint64_t)r.x | ((uint64_t)r.y << 32);
		const bool hit = (bits & sel) != 0u, dense = (r.z & 0x80000000u) != 0u;
		uint32_t v = r.w + (uint32_t)__popcll(bits & below);
		if (dense) v = gdense[r.w + cls];
		else if (hit && !(r.z & 0x40000000u)) v = exc[v];
		const bool done = hit || dense;
		res = done ? v : res;
		st = r.z & 0x0FFFFFFFu;
		live = !done;
	}
	return res;
}

struct LazyCtx {
	uint32_t H, F, fmask, recbase, abs_min;
	uint32_t S1, nabs;              /* real states (ids from S1 up are CLONES: plan.cpp build_lazy); absorbing ones = [abs_min, abs_min + nabs) */
	const uint32_t *cloneof;        /* clone S1 + j stands for state cloneof[j] */
	__amdgpu_buffer_rsrc_t own;     /* the own records, 16 bytes per state and clone: out-of-range offsets read zeros */
};

/* the state an id stands for: itself, or -- a clone, an id the fast path alone knows -- the one it copies.  Wherever an id
 * leaves the fast path: results, the exact re-walk. */
__device__ __forceinline__ uint32_t lazy_unclone(const LazyCtx &cx, uint32_t id)
{
	if (id >= cx.S1) id = ((const uint32_t __attribute__((address_space(1))) *)(uintptr_t)cx.cloneof)[id - cx.S1];
	return id;
}

/* the header words every lazy kernel needs (wave-uniform, and known to be: read through the scalar unit's eyes) */
__device__ __forceinline__ void lazy_ctx_init(LazyCtx &cx, const uint32_t *lz, const WalkArgs &a)
{
	cx.H = (uint32_t)__builtin_amdgcn_readfirstlane((int)lz[1]);
	cx.F = (uint32_t)__builtin_amdgcn_readfirstlane((int)lz[2]);
	cx.fmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)((lz[3] - 1u) << 2));
	cx.recbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)lz[4]);
	cx.abs_min = a.abs_min;
	cx.S1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lz[13]);
	cx.nabs = cx.S1 - (a.abs_min < cx.S1 ? a.abs_min : cx.S1);
	cx.cloneof = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(lz) + lz[14]) + lz[13];   /* right behind car[S1] */
	/* a buffer resource in VGPRs costs a waterfall loop per load: make every word of it a scalar */
	const uint64_t ob = reinterpret_cast<uint64_t>(lz) + lz[7];
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ob), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ob >> 32));
	const uint32_t nrec = (uint32_t)__builtin_amdgcn_readfirstlane((int)((lz[13] + lz[12]) * 16u));
	cx.own = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, (int)nrec, 0x00020000);
}

typedef const u32x4 __attribute__((address_space(3))) *lazy_rec_p;
typedef const uint32_t __attribute__((address_space(3))) *lazy_u32_p;

/* popcount(x) + acc as ONE v_bcnt_u32_b32 (its second operand is an addend); without the barrier the compiler
 * re-associates two of them into v_bcnt, v_bcnt, v_add3 */
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
	uint32_t r = (uint32_t)__builtin_popcount(x) + acc;
	__asm__("" : "+v"(r));
	return r;
}

/* rank + first - 1 of the class's bit in a record {bits, fm1}, and whether the bit is set.  bits << sh moves the class's bit
 * to bit 63 and the bits below it above it: the hit is a sign test of the high word, the rank a popcount.  (The barrier keeps
 * the sign test a 32-bit compare: as the 64-bit compare the compiler prefers it drags wait states behind it.) */
__device__ __forceinline__ bool lazy_probe(uint32_t b0, uint32_t b1, uint32_t fm1, uint32_t sh, uint32_t &n)
{
	const uint64_t x = (((uint64_t)b1 << 32) | b0) << (sh & 63u);
	uint32_t hi = (uint32_t)(x >> 32);
	__asm__("" : "+v"(hi));
	n = bcnt_acc((uint32_t)x, bcnt_acc(hi, fm1));
	return (int32_t)hi < 0;
}

/* The walk state of one input: the state's id, the LDS record E that answers for it, and -- prepared by the step that
 * entered the state, off the critical path -- what the own-record question needs: the filter word of the state the LDS
 * answer led to, whether that state lies beyond the LDS set (D), and whether the own record must be fetched whatever the
 * filter says (A: ids from F up, and every state entered through an own-record exception, whose id was not known when the
 * filter word was read -- its own record is the authority either way). */
struct LazyState {
	uint32_t id, E, fw;
	bool A, D;
};

__device__ __forceinline__ LazyState lazy_enter(const LazyCtx &cx, uint32_t id, uint32_t E)
{
	LazyState s;
	s.id = id;
	s.E = E;
	s.fw = *(lazy_u32_p)(uintptr_t)(((id << 2) & cx.fmask) + FSMHIP_LAZY_SH_BYTES);
	s.A = id >= cx.F;
	s.D = id >= cx.H;
	return s;
}

/* One byte of one input.  All boolean logic is between compares (lane masks in SGPRs, combined by the scalar unit);
 * written over 0 / 1 integers the compiler does it with vector selects.  The dependent chain from one own-record gather to
 * the next is: shift, sign test, (scalar or), select of the offset -- the filter word of the next state was read for the
 * LDS answer `ev` while the gather was in flight. */
/* TAIL (the variable-length fronts, walk_lazy_lines): `live` = the byte belongs to the lane's input.  A byte beyond the input's
 * end keeps the state (as the ABS form keeps an absorbing one) and asks for no own record; every later byte of that lane is
 * beyond the end too, so what happens to E / fw / A / D no longer matters. */
template <bool ABS, bool TAIL = false>
__device__ __forceinline__ void lazy_step(const LazyCtx &cx, LazyState &s, uint32_t sh, uint32_t &bacc, bool live = true)
{
	const u32x4 rb = *(lazy_rec_p)(uintptr_t)(cx.recbase + s.E * 16u);
	/* the own record, for the lanes that may have an exception here: the others' offset is out of range (zeros, no request).
	 * The filter: one 32-bit word per state id (modulo the filter's size), bit sh % 32 of it */
	bool pos = s.A | (s.D & (__builtin_amdgcn_ubfe(s.fw, sh, 1u) != 0u));   /* (no short circuit) */
	if (TAIL) pos = pos & live;
	const uint32_t off = pos ? s.id << 4 : 0xFFFFFFF0u;
	const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(cx.own, (int)off, 0, 0);
	/* E's answer */
	uint32_t nB, nA;
	const bool hB = lazy_probe(rb.x, rb.y, rb.z, sh, nB);
	const uint32_t cfb = rb.w - sh;
	const uint32_t ev = hB ? nB : cfb;
	const bool evD = (int32_t)ev >= (int32_t)cx.H;
	uint32_t rep = evD ? cfb : ev;
	const uint32_t fwn = *(lazy_u32_p)(uintptr_t)(((ev << 2) & cx.fmask) + FSMHIP_LAZY_SH_BYTES);
	const bool evA = ev >= cx.F;     /* (compiling this compare out where no LDS record answers with a state from F up -- plan.cpp img[15] --
	                                  * made the compiler keep A in a VGPR: 10 % more vector instructions, profiles/r06c_c5_lazy_u8_eva_slower.txt) */
	/* the own record's {bits, base, stride}: the k-th exception leads to base + k * stride (an exception of a state beyond
	 * the LDS set leads beyond the LDS set: plan.cpp) */
	const bool hA = lazy_probe(g.x, g.y, 0u, sh, nA);
	nA = (uint32_t)(__mul24((int)nA, (int)g.w) + (int)g.z);
	uint32_t m = hA ? nA : ev;
	if (ABS || TAIL) {
		const bool ab = (ABS && (s.id - cx.abs_min) < cx.nabs) | (TAIL && !live);      /* (clone ids lie beyond the absorbing range) */
		m = ab ? s.id : m;
		rep = ab ? s.E : rep;
	}
	bacc |= m | rep;
	s.id = m;
	s.E = rep;
	s.fw = fwn;
	s.A = evA | hA;
	s.D = evD;
}

/* byte k (run-time) of a chunk */
__device__ __forceinline__ uint32_t byte_dyn(const u32x4 &w, uint32_t k)
{
	const uint32_t d = k < 8u ? (k < 4u ? w.x : w.y) : (k < 12u ? w.z : w.w);
	return (d >> ((k & 3u) * 8u)) & 0xffu;
}

/* the exact re-walk of a chunk, from the state (id, E) it began in, for the lanes that met a sentinel in it */
template <bool ABS>
__device__ __forceinline__ void lazy_careful(const LazyCtx &cx, const uint32_t *simg, const uint32_t *car, uint32_t &id, uint32_t &E, const u32x4 &w, uint32_t cnt = 16u)
{
#pragma unroll 1
	for (uint32_t k = 0; k < cnt; k++) {
		const uint32_t byte = byte_dyn(w, k);
		const uint32_t sh = *(lazy_u32_p)(uintptr_t)(byte * 4u);
		LazyState c = lazy_enter(cx, id, E);
		uint32_t b = 0;
		lazy_step<ABS>(cx, c, sh, b);
		if ((int32_t)(b | sh) < 0) {
			/* the exact step; what a state beyond the LDS set carries comes from plan.cpp's car[] */
			c.id = sparse_next_global(simg, lazy_unclone(cx, id), byte, cx.abs_min);
			c.E = c.id < cx.H ? c.id : car[c.id];
		}
		id = c.id;
		E = c.E;
	}
}

/* ROWS inputs per lane (independent chains: the instruction-level parallelism a second workgroup per CU would give),
 * NB 16-byte chunks per row in flight.  a.tile_ctr != NULL: the wavefronts claim their tiles of 64 * ROWS inputs from that
 * counter (zeroed on the launch stream) instead of striding: the tail of the persistent grid balances to one tile. */
template <bool ABS, int ROWS, int NB, bool NT = false, int SHQ = 8>
__global__ void __launch_bounds__(1024)
walk_lazy(const WalkArgs a)
{
	extern __shared__ __align__(16) unsigned char lds[];
	const uint32_t *lz = static_cast<const uint32_t *>(a.lazy);
	const uint32_t *simg = static_cast<const uint32_t *>(a.tab);
	const uint32_t *car = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(lz) + lz[14]);   /* what a state beyond the LDS set carries */
	LazyCtx cx;
	lazy_ctx_init(cx, lz, a);
	{
		const u32x4 *src = reinterpret_cast<const u32x4 *>(lz + 16);
		u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
		const uint32_t nv = lz[6] / 16u;
		for (uint32_t i = threadIdx.x; i < nv; i += blockDim.x) dst[i] = src[i];
	}
	/* LDS addresses are formed from table-relative offsets: the dynamic segment must start at LDS address 0 */
	if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds != 0u) __builtin_trap();
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
	constexpr uint32_t TILE = 64u * ROWS;
	const uint64_t ntiles = (a.n + TILE - 1u) / TILE;
	const uint32_t ngroups = (uint32_t)(a.stride / 16u) / NB;   /* host guarantees divisibility */

	auto claim = [&](uint64_t prev) -> uint64_t {
		if (a.tile_ctr == nullptr) return prev + (uint64_t)gridDim.x * nw;
		uint32_t t = 0;
		if (lane == 0) t = atomicAdd(a.tile_ctr, 1u);
		return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
	};
	uint64_t tile = a.tile_ctr == nullptr ? (uint64_t)blockIdx.x * nw + wave : claim(0);
	for (; tile < ntiles; tile = claim(tile)) {
		uint64_t i[ROWS];
		const u32x4 *q[ROWS];
		LazyState st[ROWS];
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			i[r] = tile * TILE + (uint32_t)r * 64u + lane;
			q[r] = reinterpret_cast<const u32x4 *>(a.base + (i[r] < a.n ? i[r] : a.n - 1) * a.stride);
			st[r] = lazy_enter(cx, a.start, a.start);      /* the start state is LDS-resident (plan.cpp) */
		}
		for (uint32_t g = 0; g < ngroups; g++) {
			u32x4 cur[NB][ROWS];
#pragma unroll
			for (int j = 0; j < NB; j++)
#pragma unroll
				for (int r = 0; r < ROWS; r++) cur[j][r] = NT ? __builtin_nontemporal_load(&q[r][g * NB + j]) : q[r][g * NB + j];
#pragma unroll
			for (int j = 0; j < NB; j++) {
				uint32_t sid[ROWS], sE[ROWS], bacc[ROWS];
#pragma unroll
				for (int r = 0; r < ROWS; r++) {
					sid[r] = st[r].id;
					sE[r] = st[r].E;
					bacc[r] = 0u;
				}
				/* the byte -> shift lookups eight bytes at a time, their OR into the sentinel word taken at once (a byte whose class
				 * owns no bit has bit 31 set in its sh entry: the chunk then goes the exact way too).  Left to itself the compiler
				 * ORs the sixteen shifts in at the block's END and keeps them alive for it: the registers of a third input per lane. */
#pragma unroll
				for (int h = 0; h < 16 / SHQ; h++) {
					uint32_t sh[ROWS][SHQ];
#pragma unroll
					for (int r = 0; r < ROWS; r++)
#pragma unroll
						for (int k = 0; k < SHQ; k++) sh[r][k] = *(lazy_u32_p)(uintptr_t)(byte_of(cur[j][r], SHQ * h + k) * 4u);
#pragma unroll
					for (int r = 0; r < ROWS; r++) {
#pragma unroll
						for (int k = 0; k < SHQ; k++) bacc[r] |= sh[r][k];
						__asm__ volatile("" : "+v"(bacc[r]));
					}
#pragma unroll
					for (int k = 0; k < SHQ; k++)
#pragma unroll
						for (int r = 0; r < ROWS; r++) lazy_step<ABS>(cx, st[r], sh[r][k], bacc[r]);
				}
				uint32_t ball = 0;
#pragma unroll
				for (int r = 0; r < ROWS; r++) ball |= bacc[r];
				if (__builtin_amdgcn_ballot_w64((int32_t)ball < 0) != 0u) {
#pragma unroll
					for (int r = 0; r < ROWS; r++) {
						if ((int32_t)bacc[r] < 0) {
							uint32_t cid = sid[r], cE = sE[r];
							lazy_careful<ABS>(cx, simg, car, cid, cE, cur[j][r]);
							st[r] = lazy_enter(cx, cid, cE);
						}
					}
				}
			}
			if (ABS && (a.early & 1u)) {
				bool done = true;
#pragma unroll
				for (int r = 0; r < ROWS; r++) done = done && (st[r].id - cx.abs_min) < cx.nabs;
				if (__all(done)) break;
			}
		}
#pragma unroll
		for (int r = 0; r < ROWS; r++) write_result(a, tile * ROWS + (uint32_t)r, i[r], i[r] < a.n, lazy_unclone(cx, st[r].id));
	}
}

/*
 * walk_lazy_lines: the lazy walk on the fronts retest / rx drive (src/retest/main.c:1114 a line at a time, rx's literal path
 * src/rx/main.c:405-434): inputs of ANY length -- packed lines located by u64 offsets, u32 offsets or their lengths alone, fixed
 * stride + lengths, strides that are not a multiple of 64 --, and resumed walks (state_io: a state beyond the LDS set re-enters
 * with what plan.cpp's car[] says it carries).
 *
 * One input per lane SLOT, ROWS slots per lane, and a slot whose input has ended takes the next one (lane refill): the
 * wavefronts claim pieces of LAZY_PIECE consecutive inputs from the launch's counter, a slot in need takes the piece's next
 * input by its rank among the needy lanes (one ballot + mbcnt; the lengths-only front adds a wavefront prefix sum over the
 * takers' lengths to a running byte offset), so the 64 x ROWS inputs in flight stay within a few KB of one another whatever
 * their lengths.
 *
 * Round 6: WHOLE chunks and TAILS part ways.  Round 5's kernel walked every slot's next NB chunks in lockstep and masked the
 * bytes beyond an input's end inside the step -- and with 64 x ROWS inputs per wavefront some input ends in 98 % of the chunk
 * steps, so nearly every step was the masked one (30.75 vector instructions per byte against 25.2, 40.6 per USEFUL byte:
 * profiles/r07d_*).  Now
 *   - a turn walks, per slot, the next min(NB, rem / 16) WHOLE 16-byte chunks of its input (all asked for at once at the turn's
 *     head, from the input's own byte address: whole chunks lie inside the input, no edge case) with the UNMASKED step of the
 *     fixed-stride kernel.  A slot with fewer than NB whole chunks left walks zeros after them; its state is snapshotted at
 *     the chunk boundary (one compare + two selects per CHUNK, not per byte) and restored at the turn's end;
 *   - the last rem % 16 bytes of an input -- its TAIL -- are not walked by the slot: (input index, state, byte offset, count)
 *     goes into a per-wavefront queue in LDS (16 bytes an entry behind the table: plan.cpp leaves the room), the slot takes the
 *     next input at once, and when the queue cannot take a row's tails the wavefront walks up to 64 queued tails at once, one
 *     per lane, with the masked form of the step: one masked step per 64 inputs instead of one per chunk.  Inputs shorter
 *     than 16 bytes are all tail: they go from the piece to the queue.
 * Memory traffic other than the own-record gathers happens at the turn's head only: the vector memory counter retires in
 * order, so any load in flight delays the first gather wait behind it (a chunk asked for per step instead of NB per turn
 * measured 12 % slower on 1 KiB lines).
 * Measured, 4 GB of lines on the 1e5-literal automaton (tests/tools/c5_lines_probe.py; round 5's kernel: 523 / 243 GB/s on the
 * first two): two slots per lane, four chunks a turn: 0-1024 bytes 559, 8-64 bytes 363 (410 at two chunks a turn); all 1024 bytes
 * 754, all 64 bytes 770.  THREE slots, two chunks a turn (32 bytes of scratch outside the step block): 607 / 427 / 773 / 863.
 * SHQ = the byte -> shift lookups taken at once per slot (8: the fixed-stride kernel's; it keeps 8 x ROWS registers busy).  Two at
 * a time gives 24 registers back, a THIRD chunk per turn fits without scratch, and a turn's head and chunk wait are paid per 48
 * bytes of an input: 664 / 486 / 843 / 955 (shipped: <ABS, 3, 3, 2>; profiles/r09u_*).
 * Built and dropped: a slot changing inputs in MID-turn (its next input chosen at the turn's head, the old input's queue entry
 * pre-written and the state ORed in at the step its whole chunks end): no idle slots, but the bookkeeping costs more than
 * they did -- 565 / 308 on the same two mixes, 618 on 64-byte lines.
 * Results are written by the lane that finishes an input (the accept bitmap by atomic OR: cleared on the launch stream).
 */
#define FSMHIP_LAZY_PIECE 256u
#define FSMHIP_LAZY_QMIN 112u                                          /* queue entries per wavefront that plan.cpp guarantees */
#define FSMHIP_LAZY_QBYTES (16u * FSMHIP_LAZY_QMIN * 16u)              /* 16 wavefronts x 112 entries x 16 bytes = 28 KiB of LDS behind the table */

template <bool ABS, int ROWS, int NB, int SHQ = 8>
__global__ void __launch_bounds__(1024)
walk_lazy_lines(const WalkArgs a)
{
	extern __shared__ __align__(16) unsigned char lds[];
	const uint32_t *lz = static_cast<const uint32_t *>(a.lazy);
	const uint32_t *simg = static_cast<const uint32_t *>(a.tab);
	const uint32_t *car = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(lz) + lz[14]);
	LazyCtx cx;
	lazy_ctx_init(cx, lz, a);
	{
		const u32x4 *src = reinterpret_cast<const u32x4 *>(lz + 16);
		u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
		const uint32_t nv = lz[6] / 16u;
		for (uint32_t i = threadIdx.x; i < nv; i += blockDim.x) dst[i] = src[i];
	}
	if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds != 0u) __builtin_trap();
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u;
	const bool f_off = a.off != nullptr, f_off32 = !f_off && a.off32 != nullptr, f_lens = !f_off && !f_off32 && a.tbase != nullptr;
	const uint64_t base = reinterpret_cast<uint64_t>(a.base), limit = base + batch_bytes(a);
	const uint64_t safe = reinterpret_cast<uint64_t>(a.btab);
	typedef u32x4 __attribute__((aligned(1))) u32x4_any;
	typedef const u32x4_any __attribute__((address_space(1))) *glb_chunk_p;
	typedef u32x4 __attribute__((address_space(3))) *lds_q_p;

	/* the wavefront's tail queue: entry = {input index, tail offset lo, state id | tail length << 24, E | tail offset hi << 16}
	 * (state ids < 2^24: plan.cpp; E <= H < 2^16; the offset is relative to the batch's base: < 2^48; fewer than 2^32 inputs: the
	 * host checks).  As many entries as the LDS behind the table holds (a.lds_bytes: this launch's), 128 at most. */
	const uint32_t qoff = (lz[6] + 15u) & ~15u, nwv = blockDim.x >> 6;
	uint32_t qcap = (a.lds_bytes - qoff) / (16u * nwv);
	qcap = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qcap < 128u ? qcap : 128u));
	const uint32_t qbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qoff + (threadIdx.x >> 6) * (qcap * 16u)));
	uint32_t qcount = 0;

	/* up to 64 queued tails, one per lane, the masked step; the results are final.  (Two per lane -- up to 128 a pass -- measured
	 * SLOWER wherever tails are many: 329 against 363 GB/s on 8-64 byte lines, 453 against 544 on 37-byte ones.) */
	auto flush = [&]() {
		const uint32_t m = qcount < 64u ? qcount : 64u;
		const bool have = lane < m;
		qcount -= m;
		const u32x4 e = *(lds_q_p)(uintptr_t)(qbase + (qcount + (have ? lane : 0u)) * 16u);
		const uint32_t fli = e.x, id0 = e.z & 0xFFFFFFu, E0 = e.w & 0xFFFFu;
		const uint32_t cnt = have ? (e.z >> 24) & 15u : 0u;
		const uint64_t ad = base + (e.y | ((uint64_t)(e.w >> 16) << 32));
		u32x4 w = u32x4{0u, 0u, 0u, 0u};
		if (have) {
			if (ad + 16u <= limit) w = *(glb_chunk_p)ad;
			else w = load_chunk_edge(ad, true, limit, safe);
		}
		LazyState s = lazy_enter(cx, id0, E0);
		uint32_t sh[16], bacc = 0u;
#pragma unroll
		for (int k = 0; k < 16; k++) sh[k] = *(lazy_u32_p)(uintptr_t)(byte_of(w, k) * 4u);
#pragma unroll
		for (int k = 0; k < 16; k++) bacc |= (uint32_t)k < cnt ? sh[k] : 0u;
#pragma unroll
		for (int k = 0; k < 16; k++) lazy_step<ABS, true>(cx, s, sh[k], bacc, (uint32_t)k < cnt);
		if (__builtin_amdgcn_ballot_w64((int32_t)bacc < 0) != 0u) {
			if ((int32_t)bacc < 0) {
				uint32_t cid = id0, cE = E0;
				lazy_careful<ABS>(cx, simg, car, cid, cE, w, cnt);
				s.id = cid;
			}
		}
		if (have) write_result_lane(a, fli, lazy_unclone(cx, s.id));
	};
	/* the lanes with p set queue their tails (cnt in 1..15 bytes at cur); the queue is walked first when they do not fit */
	auto push = [&](bool p, uint32_t li, uint32_t id, uint32_t E, uint64_t cur, uint32_t cnt) {
		const uint64_t pm = __ballot(p);
		if (pm == 0u) return;
		const uint32_t k = (uint32_t)__popcll(pm);
		if (qcount + k > qcap) flush();       /* (more than qcap - 64 >= 48 queued: 64 of them walked, or all -- and k <= 64 fit) */
		const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
		if (p) {
			const uint64_t rel = cur - base;
			*(lds_q_p)(uintptr_t)(qbase + (qcount + rank) * 16u) = u32x4{li, (uint32_t)rel, id | (cnt << 24), E | ((uint32_t)(rel >> 32) << 16)};
		}
		qcount += k;
	};

	/* the wavefront's piece: inputs [pnext, pend); run = byte offset of input pnext (lengths-only front) */
	uint64_t pnext = 0, pend = 0, run = 0;
	bool more = true;

	uint32_t li[ROWS], rem[ROWS];     /* rem: bytes left, or 0xFFFFFFF0 while an input has that many and more (the retire step asks again) */
	uint64_t cur[ROWS];
	bool act[ROWS];
	LazyState st[ROWS];
#pragma unroll
	for (int r = 0; r < ROWS; r++) {
		li[r] = 0; cur[r] = base; rem[r] = 0; act[r] = false;
		st[r] = lazy_enter(cx, a.start, a.start);
	}

	for (;;) {
		/* ---- retire + refill: an input with fewer than 16 bytes left leaves its slot (no bytes left: its result is written;
		 * else its tail is queued), free slots take the next inputs of the piece ---- */
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			for (;;) {
				bool fin = act[r] && rem[r] < 16u;
				if (__any(fin)) {
					/* a piece of an input of 4 GiB and more has ended where the input has not (u64 offsets and fixed strides can say so;
					 * lengths and u32 offsets cannot): it goes on with what is really left */
					if (fin && rem[r] == 0u && (f_off || (!f_off32 && !f_lens && a.len == nullptr)) && !(ABS && (a.early & 1u) && (st[r].id - cx.abs_min) < cx.nabs)) {
						const uint64_t endb = f_off ? a.off[(uint64_t)li[r] + 1u] : ((uint64_t)li[r] + 1u) * a.stride;
						const uint64_t left = base + endb - cur[r];
						rem[r] = left > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)left;
						fin = rem[r] < 16u;
					}
					if (fin && rem[r] == 0u) write_result_lane(a, li[r], lazy_unclone(cx, st[r].id));
					push(fin && rem[r] != 0u, li[r], st[r].id, st[r].E, cur[r], rem[r]);
					act[r] = act[r] && !fin;
					rem[r] = fin ? 0u : rem[r];
				}
				const uint64_t needm = __ballot(!act[r]);
				if (needm == 0u) break;
				if (pnext == pend) {
					if (!more) break;
					uint32_t t = 0;
					if (lane == 0) t = atomicAdd(a.tile_ctr, 1u);
					t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
					const uint64_t p0 = (uint64_t)t * FSMHIP_LAZY_PIECE;
					if (p0 >= a.n) { more = false; break; }
					pnext = p0;
					pend = p0 + FSMHIP_LAZY_PIECE < a.n ? p0 + FSMHIP_LAZY_PIECE : a.n;
					if (f_lens) run = a.tbase[p0 / 64u];
				}
				const uint32_t avail = (uint32_t)(pend - pnext);
				const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(needm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needm, 0u));
				const bool take = !act[r] && rank < avail;
				const uint64_t i = pnext + rank;
				uint64_t beg = 0, len = 0;
				if (f_lens) {
					const uint32_t l = take ? a.len[i] : 0u;
					const uint64_t ex = wave_excl_prefix(l, lane);
					beg = run + ex;
					len = l;
					const uint64_t tot = ex + l;       /* lane 63 holds the takers' total */
					run += ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(tot >> 32), 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)tot, 63);
				} else if (take) {
					if (f_off) { beg = a.off[i]; len = a.off[i + 1u] - beg; }
					else if (f_off32) { const uint32_t b32 = a.off32[i]; beg = b32; len = a.off32[i + 1u] - b32; }
					else { beg = i * a.stride; len = a.len != nullptr ? a.len[i] : a.stride; }
				}
				if (take) {
					li[r] = (uint32_t)i;
					cur[r] = base + beg;
					rem[r] = len > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)len;
					act[r] = true;
					const uint32_t code = start_code(a, i, true);
					st[r] = lazy_enter(cx, code, code < cx.H ? code : car[code < lz[13] ? code : 0u]);
					/* an input that starts in an absorbing state is done (fsm_exec stops pulling bytes at a missing edge, exec.c:133-138) */
					if (ABS && (a.early & 1u) && code >= a.abs_min) rem[r] = 0u;
				}
				const uint32_t k = (uint32_t)__popcll(needm);
				pnext += k < avail ? k : avail;
			}
		}
		bool anyact = false;
#pragma unroll
		for (int r = 0; r < ROWS; r++) anyact = anyact || act[r];
		if (!__any(anyact)) break;

		/* ---- the next whole chunks of every slot (every active slot has at least one; a free slot has rem = 0: none) ---- */
		u32x4 w[NB][ROWS];
#pragma unroll
		for (int j = 0; j < NB; j++)
#pragma unroll
			for (int r = 0; r < ROWS; r++) {
				w[j][r] = u32x4{0u, 0u, 0u, 0u};
				if (rem[r] >= 16u * (uint32_t)(j + 1)) w[j][r] = *(glb_chunk_p)(cur[r] + 16u * (uint32_t)j);
			}

		/* (snid, snE): the state at the start of the chunk a slot is in while it has whole chunks, frozen from its last whole
		 * chunk's end on -- what the exact re-walk of a chunk starts from, and what the slot goes back to at the turn's end */
		uint32_t snid[ROWS], snE[ROWS];
#pragma unroll
		for (int r = 0; r < ROWS; r++) { snid[r] = st[r].id; snE[r] = st[r].E; }
#pragma unroll 1
		for (uint32_t j = 0; j < (uint32_t)NB; j++) {
			bool some = false;
#pragma unroll
			for (int r = 0; r < ROWS; r++) {
				const bool upto = rem[r] >= 16u * j;          /* chunk j, or the boundary right behind the slot's last whole chunk */
				snid[r] = upto ? st[r].id : snid[r];
				snE[r] = upto ? st[r].E : snE[r];
				some = some || rem[r] >= 16u * j + 16u;
			}
			if (!__any(some)) break;
			/* (the byte -> shift lookups eight bytes at a time, and their OR taken at once: left to itself the compiler ORs the
			 * sixteen shifts into the sentinel word at the block's END and keeps them alive for it -- 40 registers, scratch) */
			uint32_t bacc[ROWS];
#pragma unroll
			for (int r = 0; r < ROWS; r++) bacc[r] = 0u;
#pragma unroll
			for (int h = 0; h < 16 / SHQ; h++) {
				uint32_t sh[ROWS][SHQ];
#pragma unroll
				for (int r = 0; r < ROWS; r++)
#pragma unroll
					for (int k = 0; k < SHQ; k++) sh[r][k] = *(lazy_u32_p)(uintptr_t)(byte_of(w[0][r], SHQ * h + k) * 4u);
#pragma unroll
				for (int r = 0; r < ROWS; r++) {
#pragma unroll
					for (int k = 0; k < SHQ; k++) bacc[r] |= sh[r][k];
					__asm__ volatile("" : "+v"(bacc[r]));
				}
#pragma unroll
				for (int k = 0; k < SHQ; k++)
#pragma unroll
					for (int r = 0; r < ROWS; r++) lazy_step<ABS, false>(cx, st[r], sh[r][k], bacc[r]);
			}
			uint32_t ball = 0;
#pragma unroll
			for (int r = 0; r < ROWS; r++) {
				bacc[r] = rem[r] >= 16u * j + 16u ? bacc[r] : 0u;       /* (whatever the zeros past a slot's chunks met is nobody's business) */
				ball |= bacc[r];
			}
			if (__builtin_amdgcn_ballot_w64((int32_t)ball < 0) != 0u) {
#pragma unroll
				for (int r = 0; r < ROWS; r++) {
					if ((int32_t)bacc[r] < 0) {
						uint32_t cid = snid[r], cE = snE[r];
						lazy_careful<ABS>(cx, simg, car, cid, cE, w[0][r]);
						st[r] = lazy_enter(cx, cid, cE);
					}
				}
			}
			/* the chunks rotate through w[0]: ONE copy of the step code (unrolled over j it would be four) */
#pragma unroll
			for (int q = 0; q + 1 < NB; q++)
#pragma unroll
				for (int r = 0; r < ROWS; r++) w[q][r] = w[q + 1][r];
		}
#pragma unroll
		for (int r = 0; r < ROWS; r++) {
			/* a slot that ran out of whole chunks inside the turn goes back to where they ended (the filter word / A / D of the state
			 * it is in are re-read either way: a state entered through an own-record exception carries A for a word it never read) */
			const bool back = rem[r] < 16u * (uint32_t)NB;
			st[r] = lazy_enter(cx, back ? snid[r] : st[r].id, back ? snE[r] : st[r].E);
			const uint32_t adv = back ? rem[r] & ~15u : 16u * (uint32_t)NB;
			cur[r] += adv;
			rem[r] -= adv;
			/* an absorbing state ends the input early (fsm_exec stops pulling bytes at a missing edge, exec.c:133-138) */
			if (ABS && (a.early & 1u) && act[r] && (st[r].id - cx.abs_min) < cx.nabs) rem[r] = 0u;
		}
	}
	while (qcount != 0u) flush();
}

} // namespace fsmhip

#endif
