/*
 * walk_packed_aux.h -- the two helper kernels around walk_packed (walk_packed.h): packed_first, which
 * cuts the batch into rows and decides which walk kernel takes it, and packed_finish, which maps raw
 * state codes to the caller's results.  Included by fsm_hip.hip only (plain __global__ functions).
 */
#ifndef FSM_HIP_WALK_PACKED_AUX_H
#define FSM_HIP_WALK_PACKED_AUX_H

#include "walk_packed.h"

namespace fsmhip {

/* the batch cut into rows; which kernel takes it (use) */
__device__ __forceinline__ PackedParams packed_params(const WalkArgs &a)
{
	PackedParams p;
	const uint64_t b0 = reinterpret_cast<uint64_t>(a.base) + a.off[0], b1 = reinterpret_cast<uint64_t>(a.base) + a.off[a.n];
	p.a0 = b0 & ~(uint64_t)127;
	p.aend = (b1 + 127u) & ~(uint64_t)127;
	const uint64_t span = b1 - p.a0, mean = (b1 - b0) / a.n;
	/* rows: about four tiles per resident wavefront (a batch of 200 MB is 500 bytes per resident lane: the tail of the
	 * grid must stay short), at least four mean lengths (a lane runs about one past its row's end), within the
	 * knobs' bounds and what first[] (nvmax) and the LDS bitmask (pk_rmax) hold */
	uint32_t rs = 7u;
	while (rs < a.pk_rmax && (((uint64_t)1 << rs) < a.pk_rmin_bytes || (span >> rs) > 4u * a.pk_lanes || ((uint64_t)1 << rs) < 4u * mean)) rs++;
	while (rs < a.pk_rmax && (span >> rs) + 1u > a.pk_nvmax) rs++;
	p.rshift = rs;
	p.nrows = (span >> rs) + 1u;
	p.use = mean <= a.pk_mean_max && p.nrows <= a.pk_nvmax ? 1u : 0u;
	p.has_empty = 0;
	for (int k = 0; k < 7; k++) p.pad[k] = 0;
	return p;
}

/* first[v] = the first input whose first byte lies at or after row v's start (n if none), v = 0 .. nrows:
 * input j is that input for every row that starts in (start of j - 1, start of j].  Empty inputs get their result
 * here (the start state's code) and a bit in kbits[].  The scratch block's head (has_empty) was zeroed
 * on the stream before the launch. */
__global__ void __launch_bounds__(256)
packed_first(const WalkArgs a)
{
	const PackedParams p = packed_params(a);
	PackedParams *out = reinterpret_cast<PackedParams *>(a.pk);
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		out->a0 = p.a0; out->aend = p.aend; out->nrows = p.nrows; out->rshift = p.rshift; out->use = p.use;
	}
	if (p.use == 0u) return;
	uint32_t *first = a.pk + FSMHIP_PK_FIRST_OFF;
	const uint64_t base = reinterpret_cast<uint64_t>(a.base);
	const uint32_t lane = threadIdx.x & 63u;
	/* whole wavefronts over 64 consecutive inputs: one vote gives the 64 kbits of a word */
	for (uint64_t j0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(uint64_t)63; j0 <= a.n; j0 += (uint64_t)gridDim.x * blockDim.x) {
		const uint64_t j = j0 + lane;
		bool empty = false;
		if (j <= a.n) {
			const uint64_t oj = a.off[j];
			const uint64_t hi = j == a.n ? p.nrows : (base + oj - p.a0) >> p.rshift;
			const uint64_t lo = j == 0 ? 0 : ((base + a.off[j - 1] - p.a0) >> p.rshift) + 1u;
			for (uint64_t v = lo; v <= hi; v++) first[v] = (uint32_t)j;
			if (j < a.n && a.off[j + 1] == oj) {
				empty = true;
				a.pk_codes[j] = a.start;      /* fsm_exec on no bytes: the start state decides (exec.c:153-165) */
			}
		}
		const uint64_t kb = __ballot(empty);
		if (lane == 0) a.pk_kbits[j0 >> 6] = kb;   /* (n >> 6) + 1 words: a lane's cursor may rest on index n */
		if (lane == 0 && kb != 0u) out->has_empty = 1u;
	}
}

/* the packed-offsets front without walk_packed: is the batch short (mean length below `threshold` bytes: per-lane
 * loads, walk_generic) or long (128-byte segments with lane refill, walk_ragged)?  One thread; writes PackedParams::use
 * (1 = short).  A device-pointer front cannot know off[n] without a synchronising copy. */
__global__ void offsets_pick(const WalkArgs a, uint32_t threshold)
{
	PackedParams *out = reinterpret_cast<PackedParams *>(a.pk);
	out->use = (a.off[a.n] - a.off[0]) / a.n < threshold ? 1u : 0u;
}

/* raw state codes -> the caller's results */
__global__ void __launch_bounds__(256)
packed_finish(const WalkArgs a)
{
	if (reinterpret_cast<const PackedParams *>(a.pk)->use == 0u) return;
	const uint32_t lane = threadIdx.x & 63u;
	const uint64_t nwords = (a.n + 63u) / 64u, nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
	for (uint64_t word = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); word < nwords; word += nwaves) {
		const uint64_t i = word * 64u + lane;
		const bool valid = i < a.n;
		const uint32_t idx = fin_index(a, valid ? a.pk_codes[i] : 0u);
		const uint32_t end = valid ? a.fin[idx] : FSMHIP_NO_MATCH;
		if (valid && a.end_out != nullptr) a.end_out[i] = end;
		if (valid && a.out2 != nullptr) a.out2[i] = a.fin2[idx];
		const uint64_t m = __ballot(end != FSMHIP_NO_MATCH);
		if (a.bitmap != nullptr && lane == 0) a.bitmap[word] = m;
	}
}

} // namespace fsmhip

#endif
