/*
 * walk_lines.h -- walk_generic for batches of SHORT lines of mixed lengths: the lines of a wavefront sorted by chunk count.
 *
 * walk_generic gives each lane one of 64 consecutive lines and runs a step until the longest of them ends: with the 8-64
 * byte lines retest / rx feed, every step takes four chunk-steps although a line has 2.7 chunks on average, and nearly
 * every chunk-step sees a partial chunk in SOME lane, so all of them take the predicated form.  Measured on fixed lengths
 * (24e6 lines, C2 / C3 table; 16, 32, 48, 64 bytes: 0.160 / 0.228 / 0.295 / 0.403 ms and 0.285 / 0.326 / 0.379 / 0.448) a step
 * costs a fixed part plus 0.068 / 0.05 ms per chunk-step -- and the mixed batch 0.351 / 0.479: what all-64-byte lines cost.
 *
 * Here a wavefront takes G x 64 CONSECUTIVE lines (locality kept: their bytes are one contiguous run), sorts them by chunk
 * count with ballots (a counting sort over four buckets: <= 1, 2, 3, >= 4 chunks; no workgroup barrier -- everything is
 * wave-private), leaves (offset, length | index) in LDS slots and walks them in G passes of 64 lines of (nearly) equal chunk
 * count: a pass runs as many chunk-steps as ITS lines need, the whole-chunk steps of a pass take the unpredicated form,
 * and chunk loads that no lane of the pass needs are not issued.  Results go straight to end_out[index] (the 256 results
 * of a wavefront's lines lie within 1 KiB); the accept bitmap is assembled in four LDS words with ds_or and stored whole.
 * Lines that do not fit a slot (a super-tile spanning 4 GiB, a line of 16 MiB) take an out-of-line chunk loop.
 */
#ifndef FSMHIP_CSRC_WALK_LINES_H
#define FSMHIP_CSRC_WALK_LINES_H

namespace fsmhip {

#define FSMHIP_LINES_G 4u
#define FSMHIP_LINES_SLOTS (FSMHIP_LINES_G * 64u + 4u)                  /* 8-byte slots: the lines + four bitmap words */
#define FSMHIP_LINES_WAVE_LDS (((FSMHIP_LINES_SLOTS * 8u) + 15u) & ~15u)

/* Tiny5Pol: the slots live in the column table's unread upper row halves (walk_kernels.h ragged_aux_in_holes): 16 per hole,
 * 17 holes per wavefront -- room for 15 wavefronts, no LDS beyond the table */
template <class Pol> struct lines_slots_in_holes { static constexpr bool value = false; };
template <> struct lines_slots_in_holes<Tiny5Pol> { static constexpr bool value = true; };
#define FSMHIP_LINES_HOLE_WAVES 15u

template <class Pol, int MAXT, int FRONT>
__global__ void __launch_bounds__(MAXT)
walk_lines(const WalkArgs a)
{
	if (a.skip_flag != nullptr && *a.skip_flag == a.skip_when) return;   /* the other kernel took the batch */
	extern __shared__ __align__(16) unsigned char lds[];
	constexpr bool HOLES = lines_slots_in_holes<Pol>::value;
	Pol pol;
	pol.setup(lds, a);
	if constexpr (HOLES) pol.half_copies();
	__syncthreads();

	constexpr uint32_t NC = 4, G = FSMHIP_LINES_G;
	static_assert(FRONT != FR_ANY, "walk_lines is instantiated per metadata form");
	constexpr bool f_off = FRONT == FR_OFF64, f_off32 = FRONT == FR_OFF32, f_lens = FRONT == FR_LENS;
	const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = blockDim.x >> 6;
	const uint64_t nst = (a.n + (G * 64u - 1u)) / (G * 64u), sstride = (uint64_t)gridDim.x * nw, ntiles = (a.n + 63u) / 64u;
	const uint64_t base = reinterpret_cast<uint64_t>(a.base);
	const uint64_t total = f_off ? a.off[a.n] : f_off32 ? a.off32[a.n] : f_lens ? a.tbase[ntiles] : a.n * a.stride, limit = base + total;
	const uint64_t safe = reinterpret_cast<uint64_t>(a.btab);
	auto slot = [&](uint32_t e) -> unsigned char * {
		if (HOLES) return lds + (wave * 17u + (e >> 4)) * 256u + 128u + (e & 15u) * 8u;
		return lds + Pol::lds_bytes(a.tab_bytes) + wave * FSMHIP_LINES_WAVE_LDS + e * 8u;
	};

	/* the metadata of a super-tile's lines, asked for one super-tile ahead (clamped indices: the loads are unconditional) */
	uint64_t nb[G], ne[G], ntb[G];
	uint32_t nl[G], nb32[G], ne32[G];
	auto fetch = [&](uint64_t st) {
#pragma unroll
		for (uint32_t g = 0; g < G; g++) {
			const uint64_t i = (st * G + g) * 64u + lane, ic = i < a.n ? i : a.n - 1u;
			if (f_off) { nb[g] = a.off[ic]; ne[g] = a.off[ic + 1u]; }
			else if (f_off32) { nb32[g] = a.off32[ic]; ne32[g] = a.off32[ic + 1u]; }
			else if (f_lens) { nl[g] = a.len[ic]; const uint64_t t = st * G + g; ntb[g] = a.tbase[t < ntiles ? t : ntiles]; }
			else if (a.len != nullptr) nl[g] = a.len[ic];
		}
	};
	auto store = [&](uint64_t i, bool valid, uint32_t code, uint32_t idx) {
		uint32_t end = FSMHIP_NO_MATCH;
		if (valid) end = a.fin[fin_index(a, code)];
		if (valid && a.end_out != nullptr) a.end_out[i] = end;
		if (a.bitmap != nullptr && valid && end != FSMHIP_NO_MATCH)
			__hip_atomic_fetch_or(reinterpret_cast<uint64_t *>(slot(G * 64u + (idx >> 6))), 1ull << (idx & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
	};

	uint64_t st = (uint64_t)blockIdx.x * nw + wave;
	if (st >= nst) return;
	fetch(st);
	for (; st < nst; st += sstride) {
		uint64_t beg[G];
		uint32_t len32[G];
		bool fits = true;
#pragma unroll
		for (uint32_t g = 0; g < G; g++) {
			const uint64_t i = (st * G + g) * 64u + lane;
			const bool valid = i < a.n;
			uint64_t b, l;
			if (f_off) { b = nb[g]; l = ne[g] - nb[g]; }
			else if (f_off32) { b = nb32[g]; l = ne32[g] - nb32[g]; }
			else if (f_lens) { l = valid ? nl[g] : 0u; b = ntb[g] + wave_excl_prefix((uint32_t)l, lane); }
			else { b = i * a.stride; l = a.len != nullptr ? nl[g] : a.stride; }
			if (!valid) l = 0;
			beg[g] = b;
			fits = fits && l < (1u << 24);
			len32[g] = (uint32_t)l;
		}
		fetch(st + sstride);
		/* the super-tile's window: from its first line's first byte (lines lie in index order) */
		const uint64_t tb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(beg[0] >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)beg[0]);
		bool edge = false;
#pragma unroll
		for (uint32_t g = 0; g < G; g++) {
			const bool valid = (st * G + g) * 64u + lane < a.n;
			if (!valid) beg[g] = tb;
			fits = fits && beg[g] >= tb && beg[g] - tb + len32[g] < 0xFFFFFF00ull;
			edge = edge || (len32[g] != 0u && beg[g] + len32[g] + 8u > total);
		}
		const uint64_t word0 = st * G;
		if (!__all(fits)) {
			/* a line of 16 MiB or more, a super-tile spanning 4 GiB, offsets out of order: no slots -- every lane walks its own G
			 * lines chunk by chunk (out of line loads that never reach beyond the batch) */
#pragma unroll 1
			for (uint32_t g = 0; g < G; g++) {
				const uint64_t i = (word0 + g) * 64u + lane;
				const bool valid = i < a.n;
				uint64_t l = 0, b = 0;
				/* (the 32-bit copies above may be truncated: the lengths again, in full) */
				if (f_off) { const uint64_t ic = valid ? i : a.n - 1u; b = a.off[ic]; l = a.off[ic + 1u] - b; }
				else if (f_off32 || f_lens || a.len != nullptr) { b = beg[g]; l = len32[g]; }      /* u32 offsets, u32 lengths: nothing was truncated */
				else { b = beg[g]; l = a.stride; }
				if (!valid) l = 0;
				typename Pol::S s1 = init_state(pol, start_code(a, i, valid), a, i, valid, 0);
				const uint64_t nch = (l + 15u) >> 4;
				for (uint64_t c = 0; __any(c < nch); c++) {
					if (c < nch) {
						const u32x4 w = load_chunk_edge(base + b + 16u * c, true, limit, safe);
						const uint64_t left = l - 16u * c;
						step16_part(pol, s1, w, 0u, left < 16u ? (uint32_t)left : 16u);
					}
				}
				finish_state(pol, a, i, valid, s1, 0);
				uint32_t end = FSMHIP_NO_MATCH;
				if (valid) end = a.fin[fin_index(a, Pol::code(s1))];
				if (valid && a.end_out != nullptr) a.end_out[i] = end;
				const uint64_t m = __ballot(valid && end != FSMHIP_NO_MATCH);
				if (a.bitmap != nullptr && lane == 0 && (word0 + g) * 64u < a.n) a.bitmap[word0 + g] = m;
			}
			continue;
		}
		/* counting sort by chunk count, most chunks first: four buckets (>= 4, 3, 2, <= 1 chunks) */
		uint32_t key[G], pos[G];
#pragma unroll
		for (uint32_t g = 0; g < G; g++) {
			const uint32_t nch = (len32[g] + 15u) >> 4;
			key[g] = nch >= 4u ? 0u : nch == 3u ? 1u : nch == 2u ? 2u : 3u;
			pos[g] = 0;
		}
		uint32_t run = 0;
#pragma unroll
		for (uint32_t k = 0; k < 4; k++) {
#pragma unroll
			for (uint32_t g = 0; g < G; g++) {
				const uint64_t m = __ballot(key[g] == k);
				const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
				if (key[g] == k) pos[g] = run + below;
				run += (uint32_t)__builtin_popcountll(m);
			}
		}
#pragma unroll
		for (uint32_t g = 0; g < G; g++) {
			const uint2 e = { (uint32_t)(beg[g] - tb), len32[g] | ((g * 64u + lane) << 24) };
			*reinterpret_cast<uint2 *>(slot(pos[g])) = e;
		}
		if (lane < G) *reinterpret_cast<uint64_t *>(slot(G * 64u + lane)) = 0ull;
		__builtin_amdgcn_s_waitcnt(0xC07F);   /* lgkmcnt(0) */
		__asm__ volatile("" ::: "memory");
		__builtin_amdgcn_wave_barrier();

		const uint64_t wbytes = total - tb >= 8u ? total - tb - 8u : 0u;
		const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(base + tb), 0, (int)(wbytes < 0xFFFFFFF0ull ? (uint32_t)wbytes : 0xFFFFFFF0u), 0x00020000);
		const bool slow = __any(edge);      /* a line of this super-tile ends within 8 bytes of the batch's end */
		bool pend = false;
		uint64_t pi = 0;
		bool pvalid = false;
		uint32_t pcode = 0, pidx = 0;
#pragma unroll 1
		for (uint32_t p = 0; p < G; p++) {
			const uint2 e = *reinterpret_cast<const uint2 *>(slot(p * 64u + lane));
			const uint32_t rel = e.x, len = e.y & 0xFFFFFFu, idx = e.y >> 24;
			const uint64_t i = word0 * 64u + idx;
			const bool valid = i < a.n;
			const uint32_t nfull = len >> 4, tail = len & 15u, nchunks = nfull + (tail != 0u ? 1u : 0u);
			const uint64_t p0 = base + tb + rel;
			auto load_chunk = [&](uint32_t c) -> u32x4 {
				if (slow) return load_chunk_edge(p0 + 16u * c, c < nchunks, limit, safe);
				return __builtin_amdgcn_raw_buffer_load_b128(win, (int)(rel + 16u * c), 0, 0);
			};
			u32x4 wq[NC];
#pragma unroll
			for (uint32_t j = 0; j < NC; j++) {
				wq[j] = u32x4{0u, 0u, 0u, 0u};
				if (__any(j < nchunks)) wq[j] = load_chunk(j);      /* sorted: the later chunks of a pass of short lines are never asked for */
			}
			if (pend) store(pi, pvalid, pcode, pidx);
			typename Pol::S sv[1] = { init_state(pol, start_code(a, i, valid), a, i, valid, 0) };
			if (tail_in_step<Pol>(0)) {
#pragma unroll 1
				for (uint32_t c = 0; c < NC; c++) {
					if (!__any(c < nchunks)) break;
					if (c < nchunks) {
						if (__all(c < nfull || c >= nchunks)) {
							const u32x4 w1[1] = { wq[0] };
							step16<Pol, 1>(pol, sv, w1);
						} else {
							step16_part(pol, sv[0], wq[0], 0u, c < nfull ? 16u : tail);
						}
					}
					wq[0] = wq[1]; wq[1] = wq[2]; wq[2] = wq[3];
				}
				if (__any(nchunks > NC)) {
					u32x4 w[1] = { load_chunk(NC) };
					for (uint32_t c = NC; __any(c < nchunks); c++) {
						if (c < nchunks) {
							const u32x4 wn = load_chunk(c + 1u);
							if (__all(c < nfull || c >= nchunks)) step16<Pol, 1>(pol, sv, w);
							else step16_part(pol, sv[0], w[0], 0u, c < nfull ? 16u : tail);
							w[0] = wn;
						}
						if ((a.early & 1u) && __all(Pol::code(sv[0]) >= a.abs_min || c + 1 >= nchunks)) break;
					}
				}
			} else {
				u32x4 tw = wq[0];
#pragma unroll
				for (uint32_t c = 0; c < NC; c++) {
					if (!__any(c < nfull)) break;
					if (c < nfull) {
						const u32x4 w1[1] = { wq[c] };
						step16<Pol, 1>(pol, sv, w1);
					}
					if (c + 1u < NC && nfull == c + 1u) tw = wq[c + 1u];
				}
				if (__any(nchunks > NC)) {
					u32x4 w[1] = { load_chunk(NC) };
					for (uint32_t c = NC; __any(c < nchunks); c++) {
						if (c < nchunks) {
							const u32x4 wn = load_chunk(c + 1u);
							if (c < nfull) step16<Pol, 1>(pol, sv, w);
							else tw = w[0];
							w[0] = wn;
						}
						if ((a.early & 1u) && __all(Pol::code(sv[0]) >= a.abs_min || c + 1 >= nchunks)) break;
					}
				}
				if (__any(tail != 0u)) {
					if (tail != 0u) step16_part(pol, sv[0], tw, 0u, tail);
				}
			}
			finish_state(pol, a, i, valid, sv[0], 0);
			pend = true;
			pi = i;
			pvalid = valid;
			pcode = Pol::code(sv[0]);
			pidx = idx;
		}
		store(pi, pvalid, pcode, pidx);
		if (a.bitmap != nullptr) {
			__builtin_amdgcn_s_waitcnt(0xC07F);
			__asm__ volatile("" ::: "memory");
			__builtin_amdgcn_wave_barrier();
			if (lane < G && (word0 + lane) * 64u < a.n) a.bitmap[word0 + lane] = *reinterpret_cast<const uint64_t *>(slot(G * 64u + lane));
		}
		/* the next super-tile's slots are written only after this one's were read: same wavefront, program order */
		__builtin_amdgcn_wave_barrier();
	}
}

} // namespace fsmhip

#endif
