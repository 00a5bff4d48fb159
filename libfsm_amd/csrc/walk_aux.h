/*
 * walk_aux.h -- the small kernels around the walk (included by fsm_hip.hip only):
 *   offsets_pick   which walk kernel takes a batch of variable-length inputs: walk_generic / walk_lines32 (mean length below a
 *                  threshold; the batch below 4 GiB) or walk_ragged; decided on the device, a device-pointer front cannot know
 *                  the lengths;
 *   tile_bases_*   the lengths-only front (inputs packed back to back, u32 len[n] and nothing else -- what a caller
 *                  holding (b, e) pairs has, src/libfsm/print/c.c:569-619): byte offset of every 64th input.  The walk
 *                  kernels add a wavefront prefix sum of the 64 lengths they load anyway, so the n-entry offsets array
 *                  is never materialised: 4 B of metadata per input in this pass + 4 B in the walk + 8 B per 64 inputs.
 */
#ifndef FSM_HIP_WALK_AUX_H
#define FSM_HIP_WALK_AUX_H

#include "walk_kernels.h"

namespace fsmhip {

/* cand bit 0: walk_ragged is a candidate (else every batch goes to a per-lane kernel); bit 1: walk_lines32 is one -- taken for a
 * short-lines batch that ends below 4 GiB (its last offset, not its size: the kernel's offsets are relative to the base);
 * bit 2: walk_generic was launched (the host leaves it out where the allocation behind the base suggests that no batch can reach
 * 4 GiB -- a hint: a short-lines batch that does reach it then goes to walk_ragged, which the host launches in that case).
 * Only a kernel that was launched is ever picked: a flag naming another would leave the outputs unwritten. */
__global__ void __launch_bounds__(256)
offsets_pick(const WalkArgs a, uint32_t threshold, uint32_t threshold32, uint32_t cand)
{
	__shared__ uint64_t part[256];
	uint64_t bytes = 0, cnt = a.n, last = ~(uint64_t)0;
	if (a.off != nullptr) { last = a.off[a.n]; bytes = last - a.off[0]; }
	else if (a.off32 != nullptr) { last = a.off32[a.n]; bytes = last - a.off32[0]; }
	else if (a.tbase != nullptr) bytes = last = a.tbase[(a.n + 63u) / 64u];
	else if (a.len != nullptr) {
		/* fixed stride + lengths: a strided sample of (at most) 4096 of them */
		const uint64_t ns = a.n < 4096u ? a.n : 4096u, step = a.n / ns;
		uint64_t s = 0;
		for (uint64_t k = threadIdx.x; k < ns; k += blockDim.x) s += a.len[k * step];
		part[threadIdx.x] = s;
		__syncthreads();
		for (uint32_t w = 128; w != 0; w >>= 1) {
			if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
			__syncthreads();
		}
		bytes = part[0];
		cnt = ns;
	} else bytes = a.n * a.stride;
	if (threadIdx.x == 0) {
		/* (threshold32: where walk_lines32 hands over to walk_ragged -- later than walk_generic does: fsm_hip.hip pick_mean_of) */
		const uint64_t mean = bytes / (cnt ? cnt : 1u);
		const bool l32 = (cand & 2u) && last < ((uint64_t)1 << 32);
		const bool shrt = !(cand & 1u) || mean < (l32 ? threshold32 : threshold);
		*a.pick_flag = !shrt ? (uint32_t)PICK_RAGGED : l32 ? (uint32_t)PICK_LINES32 : (cand & 4u) || !(cand & 1u) ? (uint32_t)PICK_GENERIC : (uint32_t)PICK_RAGGED;
	}
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}

/* pass 1: one workgroup of 16 waves per 1024 tiles (a tile = 64 inputs; tile T = ceil(n / 64) is an empty one whose
 * base is the batch's size).  tbase[t] = bytes of the block's tiles before t, btot[block] = the block's bytes. */
__global__ void __launch_bounds__(1024)
tile_bases_pass1(const uint32_t *len, uint64_t n, uint64_t ntiles1, uint64_t *tbase, uint64_t *btot)
{
	__shared__ uint64_t wtot[16];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint64_t t0 = (uint64_t)blockIdx.x * 1024u + wave * 64u;
	uint64_t mine = 0;
	for (uint32_t k = 0; k < 64u; k++) {
		const uint64_t i = (t0 + k) * 64u + lane;
		const uint64_t s = wave_sum_u64(i < n ? len[i] : 0u);
		if (lane == k) mine = s;
	}
	/* exclusive scan of the 1024 tile sums: inside the wave, then over the 16 wave totals */
	uint64_t x = mine;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint64_t y = __shfl_up(x, d, 64);
		if (lane >= (uint32_t)d) x += y;
	}
	if (lane == 63u) wtot[wave] = x;
	__syncthreads();
	uint64_t before = 0;
	for (uint32_t w = 0; w < wave; w++) before += wtot[w];
	const uint64_t t = t0 + lane;
	if (t < ntiles1) tbase[t] = before + x - mine;
	if (threadIdx.x == 1023u) btot[blockIdx.x] = before + x;
}

/* pass 2: exclusive scan of the block totals in place (one workgroup) */
__global__ void __launch_bounds__(1024)
tile_bases_pass2(uint64_t *btot, uint64_t nb)
{
	__shared__ uint64_t wtot[16];
	__shared__ uint64_t carry;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (uint64_t b0 = 0; b0 < nb; b0 += 1024u) {
		const uint64_t b = b0 + threadIdx.x;
		const uint64_t mine = b < nb ? btot[b] : 0u;
		uint64_t x = mine;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const uint64_t y = __shfl_up(x, d, 64);
			if (lane >= (uint32_t)d) x += y;
		}
		if (lane == 63u) wtot[wave] = x;
		__syncthreads();
		uint64_t before = carry;
		for (uint32_t w = 0; w < wave; w++) before += wtot[w];
		if (b < nb) btot[b] = before + x - mine;
		__syncthreads();
		if (threadIdx.x == 1023u) carry = before + x;
		__syncthreads();
	}
}

/* pass 3: add each block's base */
__global__ void __launch_bounds__(256)
tile_bases_pass3(uint64_t *tbase, uint64_t ntiles1, const uint64_t *btot)
{
	for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntiles1; t += (uint64_t)gridDim.x * blockDim.x)
		tbase[t] += btot[t >> 10];
}

/* Clearing an output before a kernel that ORs into it (the ragged kernel's accept bitmap, wide eager sets, the lazy walk's tile
 * counter).  A kernel, not hipMemsetAsync: as the first node of a captured graph a memset node was seen to run before the work
 * that precedes the graph launch on the same stream had finished (stale bits survived the clear in tests/test_gpu_round4.py's
 * graph test, in a long session only): a kernel node is ordered like every other kernel. */
__global__ void __launch_bounds__(256) zero_u32_kernel(uint32_t *p, uint64_t n32)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n32; i += (uint64_t)gridDim.x * blockDim.x) p[i] = 0u;
}

static inline hipError_t zero_async(void *p, uint64_t bytes, hipStream_t s)
{
	const uint64_t n32 = bytes / 4u;
	if (n32 == 0) return hipSuccess;
	uint64_t blocks = (n32 + 255u) / 256u;
	if (blocks > 4096u) blocks = 4096u;
	hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint32_t *>(p), n32);
	return hipGetLastError();
}

} // namespace fsmhip

#endif
