/*
 * walk_kernels.h -- gfx950 device code: the batched DFA walk.
 *
 * One lane = one input string, 64 inputs per wavefront advancing in lockstep
 * against the same DFA (the data-parallel form of the reference's per-input
 * loop `while ((c = getc()) != EOF) state = delta(state, c)`,
 * src/libfsm/exec.c:132-151).  No MFMA: there is no contraction here; the
 * kernel is bound by HBM input bandwidth and by LDS lookup rate.
 *
 * Table policies (how delta(state, byte) is evaluated):
 *   TinyPol   <=16 states.  LDS holds, per byte value, the whole transition
 *             COLUMN (16 x 4-bit next states in one 64-bit word), replicated
 *             once per LDS bank pair so lane l always reads bank 2*(l%32):
 *             conflict-free by construction.  The lookup depends only on the
 *             input byte, never on the state, so all 16 lookups of a 16-byte
 *             chunk are in flight together and the state chain is pure VALU.
 *   LdsPol    class-compressed dense table T[state][class] (u16) in LDS plus a
 *             bank-private byte->class table B[256][32] (conflict-free).
 *   CombPol   column-default + comb exceptions in LDS (see plan.cpp).
 *   GlobPol   T[state][class] (u32) in HBM/L2, B in LDS.
 *
 * Input staging modes (uniform-length, 16-byte aligned rows):
 *   IN_DIRECT  every lane reads its own row 16 bytes at a time, NB chunks in
 *              flight (global_load_dwordx4 at row stride).
 *   IN_LDSDMA  rows are fetched as whole 64-byte segments (4 adjacent lanes per
 *              row) by global_load_lds_dwordx4 straight into a 4 KiB per-wave
 *              LDS tile, piece-rotated so the row-per-lane ds_read_b128 that
 *              follows is bank-conflict-free; no VGPR staging, no ds_write.
 * plus IN_GENERIC for ragged lengths / arbitrary alignment / packed offsets.
 */
#ifndef FSM_HIP_WALK_KERNELS_H
#define FSM_HIP_WALK_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsmhip {

struct WalkArgs {
	const uint8_t  *base;     /* device input bytes                                   */
	uint64_t        stride;   /* bytes between inputs (fixed-stride modes)            */
	const uint32_t *len;      /* per-input lengths or NULL (= stride)                 */
	const uint64_t *off;      /* packed mode: n+1 offsets, or NULL                    */
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
	uint32_t        early;    /* retire a wavefront once every lane is absorbing      */
};

enum { IN_DIRECT = 0, IN_LDSDMA = 1, IN_GENERIC = 2 };

#define FSMHIP_NO_MATCH 0xFFFFFFFFu

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t byte_of(const u32x4 &w, int k)
{
	const uint32_t d = (k < 4) ? w.x : (k < 8) ? w.y : (k < 12) ? w.z : w.w;
	return (d >> (8 * (k & 3))) & 0xffu;
}

/* ------------------------------------------------------------------ */
/* table policies                                                     */
/* ------------------------------------------------------------------ */

struct TinyPol {
	static constexpr uint32_t kLdsBytes = 256u * 32u * 8u;
	const uint64_t *colp; /* LDS column table, already offset by lane%32 */

	__device__ static uint32_t lds_bytes(const WalkArgs &) { return kLdsBytes; }
	__device__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		uint64_t *col = reinterpret_cast<uint64_t *>(lds);
		const uint64_t *src = static_cast<const uint64_t *>(a.tab);
		for (uint32_t i = threadIdx.x; i < 256u * 32u; i += blockDim.x) col[i] = src[i >> 5];
		colp = col + (threadIdx.x & 31u);
	}
	__device__ __forceinline__ uint32_t step1(uint32_t st, uint32_t b) const
	{
		const uint64_t v = colp[b * 32u];
		return (uint32_t)(v >> (st * 4u)) & 15u;
	}
	__device__ __forceinline__ void step16(uint32_t &st, const u32x4 &w) const
	{
		uint64_t v[16];
#pragma unroll
		for (int k = 0; k < 16; k++) v[k] = colp[byte_of(w, k) * 32u];
#pragma unroll
		for (int k = 0; k < 16; k++) st = (uint32_t)(v[k] >> (st * 4u)) & 15u;
	}
};

struct LdsPol {
	const uint32_t *bp;        /* LDS B table + lane%32 ; entry = class * 2      */
	const unsigned char *tab;  /* LDS table; state is a byte offset into it      */

	__device__ static uint32_t lds_bytes(const WalkArgs &a) { return 256u * 32u * 4u + ((a.tab_bytes + 15u) & ~15u); }
	__device__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		uint32_t *B = reinterpret_cast<uint32_t *>(lds);
		for (uint32_t i = threadIdx.x; i < 256u * 32u; i += blockDim.x) B[i] = a.btab[i >> 5];
		uint32_t *T = reinterpret_cast<uint32_t *>(lds + 256u * 32u * 4u);
		const uint32_t *src = static_cast<const uint32_t *>(a.tab);
		for (uint32_t i = threadIdx.x; i < (a.tab_bytes + 3u) / 4u; i += blockDim.x) T[i] = src[i];
		bp = B + (threadIdx.x & 31u);
		tab = lds + 256u * 32u * 4u;
	}
	__device__ __forceinline__ uint32_t step1(uint32_t st, uint32_t b) const
	{
		const uint32_t ca = bp[b * 32u];
		return (uint32_t)(*reinterpret_cast<const uint16_t *>(tab + st + ca)) << 2;
	}
	__device__ __forceinline__ void step16(uint32_t &st, const u32x4 &w) const
	{
		uint32_t ca[16];
#pragma unroll
		for (int k = 0; k < 16; k++) ca[k] = bp[byte_of(w, k) * 32u];
#pragma unroll
		for (int k = 0; k < 16; k++)
			st = (uint32_t)(*reinterpret_cast<const uint16_t *>(tab + st + ca[k])) << 2;
	}
};

struct CombPol {
	const uint32_t *bp;     /* LDS B table + lane%32 ; entry = (dflt_off << 16) | class */
	const uint32_t *comb;   /* LDS comb array; state = row offset in entries            */

	__device__ static uint32_t lds_bytes(const WalkArgs &a) { return 256u * 32u * 4u + ((a.tab_bytes + 15u) & ~15u); }
	__device__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		uint32_t *B = reinterpret_cast<uint32_t *>(lds);
		for (uint32_t i = threadIdx.x; i < 256u * 32u; i += blockDim.x) B[i] = a.btab[i >> 5];
		uint32_t *T = reinterpret_cast<uint32_t *>(lds + 256u * 32u * 4u);
		const uint32_t *src = static_cast<const uint32_t *>(a.tab);
		for (uint32_t i = threadIdx.x; i < a.tab_bytes / 4u; i += blockDim.x) T[i] = src[i];
		bp = B + (threadIdx.x & 31u);
		comb = T;
	}
	__device__ __forceinline__ uint32_t step1(uint32_t st, uint32_t b) const
	{
		const uint32_t be = bp[b * 32u];
		const uint32_t x = comb[st + (be & 0xffffu)] ^ (st << 16);
		return x < 0x10000u ? x : (be >> 16);
	}
	__device__ __forceinline__ void step16(uint32_t &st, const u32x4 &w) const
	{
		uint32_t be[16];
#pragma unroll
		for (int k = 0; k < 16; k++) be[k] = bp[byte_of(w, k) * 32u];
#pragma unroll
		for (int k = 0; k < 16; k++) {
			const uint32_t x = comb[st + (be[k] & 0xffffu)] ^ (st << 16);
			st = x < 0x10000u ? x : (be[k] >> 16);
		}
	}
};

struct GlobPol {
	const uint32_t *bp;        /* LDS B table + lane%32 ; entry = class * 4       */
	const unsigned char *tab;  /* device table; state is a byte offset into it    */

	__device__ static uint32_t lds_bytes(const WalkArgs &) { return 256u * 32u * 4u; }
	__device__ void setup(unsigned char *lds, const WalkArgs &a)
	{
		uint32_t *B = reinterpret_cast<uint32_t *>(lds);
		for (uint32_t i = threadIdx.x; i < 256u * 32u; i += blockDim.x) B[i] = a.btab[i >> 5];
		bp = B + (threadIdx.x & 31u);
		tab = static_cast<const unsigned char *>(a.tab);
	}
	__device__ __forceinline__ uint32_t step1(uint32_t st, uint32_t b) const
	{
		const uint32_t ca = bp[b * 32u];
		return *reinterpret_cast<const uint32_t *>(tab + st + ca);
	}
	__device__ __forceinline__ void step16(uint32_t &st, const u32x4 &w) const
	{
		uint32_t ca[16];
#pragma unroll
		for (int k = 0; k < 16; k++) ca[k] = bp[byte_of(w, k) * 32u];
#pragma unroll
		for (int k = 0; k < 16; k++) st = *reinterpret_cast<const uint32_t *>(tab + st + ca[k]);
	}
};

/* ------------------------------------------------------------------ */
/* result write-back                                                  */
/* ------------------------------------------------------------------ */

__device__ __forceinline__ void write_result(const WalkArgs &a, uint64_t tile, uint64_t i, bool valid, uint32_t st)
{
	uint32_t end = FSMHIP_NO_MATCH;
	if (valid) end = a.fin[st / a.fin_div];
	if (valid && a.end_out != nullptr) a.end_out[i] = end;
	const uint64_t m = __ballot(valid && end != FSMHIP_NO_MATCH);
	if (a.bitmap != nullptr && (threadIdx.x & 63u) == 0) a.bitmap[tile] = m;
}

/* ------------------------------------------------------------------ */
/* IN_DIRECT: per-lane 16-byte loads, NB chunks in flight             */
/* ------------------------------------------------------------------ */

template <class Pol, int NB, bool NT>
__global__ void __launch_bounds__(1024)
walk_direct(const WalkArgs a)
{
	extern __shared__ __align__(16) unsigned char lds[];
	Pol pol;
	pol.setup(lds, a);
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const uint64_t ntiles = (a.n + 63u) / 64u;
	const uint32_t nchunks = (uint32_t)(a.stride / 16u);
	const uint32_t ngroups = nchunks / NB; /* host guarantees nchunks % NB == 0 */

	for (uint64_t tile = (uint64_t)blockIdx.x * nw + wave; tile < ntiles; tile += (uint64_t)gridDim.x * nw) {
		const uint64_t i = tile * 64u + lane;
		const bool valid = i < a.n;
		const u32x4 *q = reinterpret_cast<const u32x4 *>(a.base + (valid ? i : a.n - 1) * a.stride);
		uint32_t st = a.start;
		u32x4 cur[NB], nxt[NB];
#pragma unroll
		for (int j = 0; j < NB; j++) cur[j] = NT ? __builtin_nontemporal_load(q + j) : q[j];
		for (uint32_t g = 0; g < ngroups; g++) {
			if (g + 1 < ngroups) {
#pragma unroll
				for (int j = 0; j < NB; j++)
					nxt[j] = NT ? __builtin_nontemporal_load(q + (g + 1) * NB + j) : q[(g + 1) * NB + j];
			}
#pragma unroll
			for (int j = 0; j < NB; j++) pol.step16(st, cur[j]);
			if (a.early && __all(st >= a.abs_min)) break;
#pragma unroll
			for (int j = 0; j < NB; j++) cur[j] = nxt[j];
		}
		write_result(a, tile, i, valid, st);
	}
}

/* ------------------------------------------------------------------ */
/* IN_LDSDMA: coalesced 64-byte row segments DMA'd into a per-wave     */
/* LDS tile, read back row-per-lane                                   */
/* ------------------------------------------------------------------ */

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

template <class Pol>
__global__ void __launch_bounds__(1024)
walk_ldsdma(const WalkArgs a)
{
	extern __shared__ __align__(16) unsigned char lds[];
	Pol pol;
	pol.setup(lds, a);
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
	unsigned char *stg = lds + ((Pol::lds_bytes(a) + 15u) & ~15u) + wave * 4096u;
	const uint64_t ntiles = (a.n + 63u) / 64u;
	const uint32_t nseg = (uint32_t)(a.stride / 64u); /* host guarantees stride % 64 == 0 */

	/* loader role: lane = 4*r + q fetches, in DMA instruction j, the piece
	 * p = (q - (r>>2)) & 3 of row 16*j + r, landing at stg + j*1024 + lane*16.
	 * reader role: lane i = 16*j + r finds piece p of its own row at
	 * stg + j*1024 + (4*r + ((p + (r>>2)) & 3)) * 16: within every
	 * ds_read_b128 lane group the 16-byte slots are all distinct. */
	const uint32_t lr = lane >> 2, lq = lane & 3u;
	const uint32_t lpiece = (lq - (lr >> 2)) & 3u;
	const uint32_t rj = lane >> 4, rr = lane & 15u;
	const unsigned char *rd = stg + rj * 1024u + rr * 64u;
	const uint32_t rot = rr >> 2;

	for (uint64_t tile = (uint64_t)blockIdx.x * nw + wave; tile < ntiles; tile += (uint64_t)gridDim.x * nw) {
		const uint64_t i = tile * 64u + lane;
		const bool valid = i < a.n;
		const uint64_t row0 = tile * 64u;
		const unsigned char *src[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			uint64_t row = row0 + 16u * j + lr;
			if (row >= a.n) row = a.n - 1;
			src[j] = a.base + row * a.stride + lpiece * 16u;
		}
		uint32_t st = a.start;
#pragma unroll
		for (int j = 0; j < 4; j++)
			__builtin_amdgcn_global_load_lds((glb_void_t *)(src[j]), (lds_void_t *)(stg + j * 1024u), 16, 0, 0);
		for (uint32_t s = 0; s < nseg; s++) {
			__builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): the tile has landed */
			__asm__ volatile("" ::: "memory");
			u32x4 w[4];
#pragma unroll
			for (int p = 0; p < 4; p++)
				w[p] = *reinterpret_cast<const u32x4 *>(rd + (((uint32_t)p + rot) & 3u) * 16u);
			__builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0): tile is in registers, slot reusable */
			__asm__ volatile("" ::: "memory");
			if (s + 1 < nseg) {
#pragma unroll
				for (int j = 0; j < 4; j++)
					__builtin_amdgcn_global_load_lds((glb_void_t *)(src[j] + (uint64_t)(s + 1) * 64u),
					                                 (lds_void_t *)(stg + j * 1024u), 16, 0, 0);
			}
#pragma unroll
			for (int p = 0; p < 4; p++) pol.step16(st, w[p]);
			if (a.early && __all(st >= a.abs_min)) {
				__builtin_amdgcn_s_waitcnt(0x0F70); /* drain the prefetch before the tile is reused */
				break;
			}
		}
		write_result(a, tile, i, valid, st);
	}
}

/* ------------------------------------------------------------------ */
/* IN_GENERIC: ragged lengths, any alignment, fixed stride or packed  */
/* ------------------------------------------------------------------ */

template <class Pol>
__global__ void __launch_bounds__(1024)
walk_generic(const WalkArgs a)
{
	extern __shared__ __align__(16) unsigned char lds[];
	Pol pol;
	pol.setup(lds, a);
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
	const uint64_t ntiles = (a.n + 63u) / 64u;

	for (uint64_t tile = (uint64_t)blockIdx.x * nw + wave; tile < ntiles; tile += (uint64_t)gridDim.x * nw) {
		const uint64_t i = tile * 64u + lane;
		const bool valid = i < a.n;
		uint64_t beg = 0, len = 0;
		if (valid) {
			if (a.off != nullptr) { beg = a.off[i]; len = a.off[i + 1] - beg; }
			else { beg = i * a.stride; len = a.len != nullptr ? a.len[i] : a.stride; }
		}
		/* every 16-byte aligned chunk that contains at least one byte of the
		 * input is read whole; bytes outside [beg, beg+len) are masked.  An
		 * aligned 16-byte chunk never crosses a page, so this cannot fault. */
		const uint64_t p0 = reinterpret_cast<uint64_t>(a.base) + beg;
		const uint64_t q0 = p0 & ~(uint64_t)15;
		const uint32_t head = (uint32_t)(p0 - q0);
		const uint64_t span = len ? head + len : 0;
		const uint64_t nchunks = (span + 15u) / 16u;
		uint32_t st = a.start;
		for (uint64_t c = 0; __any(c < nchunks); c++) {
			if (c < nchunks) {
				const u32x4 w = *reinterpret_cast<const u32x4 *>(q0 + c * 16u);
#pragma unroll
				for (int k = 0; k < 16; k++) {
					const uint64_t pos = c * 16u + k - head; /* wraps below head: huge, fails the test */
					const uint32_t nx = pol.step1(st, byte_of(w, k));
					st = pos < len ? nx : st;
				}
			}
			if (a.early && __all(st >= a.abs_min || c + 1 >= nchunks)) break;
		}
		write_result(a, tile, i, valid, st);
	}
}

/* ------------------------------------------------------------------ */
/* synthetic input generator                                          */
/* ------------------------------------------------------------------ */

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z)
{
	z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
	z ^= z >> 27; z *= 0x94D049BB133111EBull;
	z ^= z >> 31;
	return z;
}

struct GenArgs {
	unsigned char *base;
	uint64_t stride, n, first_index, seed;
	uint32_t nalpha, plant_len, plant_every;
	unsigned char alphabet[256];
	unsigned char plant[64];
};

__host__ __device__ __forceinline__ uint64_t gen_word(const GenArgs &g, uint64_t gi, uint64_t wi)
{
	uint64_t r = mix64(g.seed ^ (gi * 0x9E3779B97F4A7C15ull) ^ wi);
	if (g.nalpha != 0) {
		uint64_t o = 0;
		for (int k = 0; k < 8; k++)
			o |= (uint64_t)g.alphabet[((r >> (8 * k)) & 0xff) % g.nalpha] << (8 * k);
		r = o;
	}
	return r;
}

__host__ __device__ __forceinline__ uint64_t plant_offset(const GenArgs &g, uint64_t gi)
{
	return mix64(g.seed ^ gi ^ 0xA5A5A5A5A5A5A5A5ull) % (g.stride - g.plant_len + 1);
}

/* one thread = one 8-byte word of one row; rows are stride/8 words */
__global__ void __launch_bounds__(256)
gen_inputs_kernel(const GenArgs g)
{
	const uint64_t wpr = g.stride / 8u;
	const uint64_t total = g.n * wpr;
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t row = t / wpr, wi = t - row * wpr;
		const uint64_t gi = g.first_index + row;
		uint64_t v = gen_word(g, gi, wi);
		if (g.plant_len != 0 && gi % g.plant_every == 0) {
			const uint64_t po = plant_offset(g, gi);
			for (int k = 0; k < 8; k++) {
				const uint64_t pos = wi * 8u + k;
				if (pos >= po && pos < po + g.plant_len)
					v = (v & ~(0xffull << (8 * k))) | ((uint64_t)g.plant[pos - po] << (8 * k));
			}
		}
		*reinterpret_cast<uint64_t *>(g.base + row * g.stride + wi * 8u) = v;
	}
}

/* Affix generator (rx-style workload, BASELINE config 3): rows whose global
 * index is a multiple of `every` are  prefix + body alphabet + suffix  (exactly
 * stride bytes, so they can match ^<prefix>[0-9]+(x|yz)$-like patterns); all
 * other rows are random over the plain alphabet.  affix entries are 8 bytes:
 * [len, b0..b6]. */
struct AffixArgs {
	const unsigned char *pfx, *sfx; /* npfx / nsfx entries of 8 bytes */
	uint32_t npfx, nsfx, every, nbody;
	unsigned char body[256];
};

__host__ __device__ __forceinline__ uint64_t affix_word(const GenArgs &g, const AffixArgs &x, uint64_t gi, uint64_t wi)
{
	if (gi % x.every != 0) return gen_word(g, gi, wi);
	const uint64_t r = mix64(g.seed ^ (gi * 0x9E3779B97F4A7C15ull) ^ wi);
	const uint64_t h = mix64(g.seed ^ gi ^ 0x5A5A5A5A5A5A5A5Aull);
	const unsigned char *pe = x.pfx + 8u * (uint32_t)((h & 0xffffffffu) % x.npfx);
	const unsigned char *se = x.sfx + 8u * (uint32_t)((h >> 32) % x.nsfx);
	const uint32_t pl = pe[0], sl = se[0];
	uint64_t o = 0;
	for (int k = 0; k < 8; k++) {
		const uint64_t pos = wi * 8u + k;
		unsigned char b = x.body[((r >> (8 * k)) & 0xff) % x.nbody];
		if (pos < pl) b = pe[1 + pos];
		else if (pos >= g.stride - sl) b = se[1 + (pos - (g.stride - sl))];
		o |= (uint64_t)b << (8 * k);
	}
	return o;
}

__global__ void __launch_bounds__(256)
gen_affix_kernel(const GenArgs g, const AffixArgs x)
{
	const uint64_t wpr = g.stride / 8u;
	const uint64_t total = g.n * wpr;
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t row = t / wpr, wi = t - row * wpr;
		*reinterpret_cast<uint64_t *>(g.base + row * g.stride + wi * 8u) = affix_word(g, x, g.first_index + row, wi);
	}
}

} // namespace fsmhip


#endif
