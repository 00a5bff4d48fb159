/*
 * trace_kernel.h -- fsm_exec's eager-output callback STREAM, input by input.
 *
 * The reference calls the callback for every output id of the start state before it reads a byte, then for every
 * id of every state it enters, repeats included, and stops at the first missing edge
 * (src/libfsm/exec.c:126-144, match_eager_outputs_for_state :62-78).  The walk kernels deliver the SET of ids per
 * input (one register-held mask, fsm_hip_exec_batch_eager); a caller that needs the order of the emissions or how
 * often an id fired asks for the stream, and gets this kernel: one input per lane over the plain renumbered table in
 * global memory (every layout keeps Plan::dense on the host; it is uploaded the first time a trace is asked for),
 * writing (id, position) records.  It is an exact-semantics path, not a fast one: emissions are lane-divergent
 * stores by nature.
 */
#ifndef FSMHIP_CSRC_TRACE_KERNEL_H
#define FSMHIP_CSRC_TRACE_KERNEL_H

#include <hip/hip_runtime.h>
#include <cstdint>

namespace fsmhip {

struct TraceArgs {
	const uint8_t *base;
	uint64_t stride;            /* fixed-stride fronts (off == nullptr) */
	const uint32_t *len;        /* or nullptr: every input is `stride` bytes */
	const uint64_t *off;        /* packed inputs: n + 1 offsets */
	uint64_t n;
	const uint32_t *dense;      /* [S1][C] renumbered next state */
	const uint32_t *cls4;       /* [64] byte -> class, four to a word */
	const uint32_t *eoff;       /* [S1 + 1] CSR into eids */
	const uint32_t *eids;
	const uint32_t *fin;        /* [S1] caller's end state id or NO_MATCH */
	uint32_t C, start, dead;
	uint32_t cap;               /* records kept per input */
	uint32_t *end_out;          /* may be null */
	uint32_t *count_out;        /* emissions per input (may exceed cap: the stream was cut) */
	uint32_t *ids_out;          /* [n][cap] */
	uint32_t *pos_out;          /* [n][cap] bytes consumed when the id fired (0 = the start state's), may be null */
};

__device__ __forceinline__ uint32_t trace_emit(const TraceArgs &a, uint64_t i, uint32_t s, uint32_t pos, uint32_t cnt)
{
	const uint32_t lo = a.eoff[s], hi = a.eoff[s + 1];
	for (uint32_t k = lo; k < hi; k++, cnt++) {
		if (cnt < a.cap) {
			a.ids_out[i * a.cap + cnt] = a.eids[k];
			if (a.pos_out) a.pos_out[i * a.cap + cnt] = pos;
		}
	}
	return cnt;
}

__global__ void __launch_bounds__(256) eager_trace_kernel(TraceArgs a)
{
	__shared__ uint32_t cls4[64];
	if (threadIdx.x < 64) cls4[threadIdx.x] = a.cls4[threadIdx.x];
	__syncthreads();
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.n) return;
	const uint8_t *p;
	uint64_t l;
	if (a.off) { p = a.base + a.off[i]; l = a.off[i + 1] - a.off[i]; }
	else       { p = a.base + i * a.stride; l = a.len ? a.len[i] : a.stride; }
	uint32_t s = a.start;
	uint32_t cnt = trace_emit(a, i, s, 0, 0);
	for (uint64_t t = 0; t < l; t++) {
		const uint32_t b = p[t];
		const uint32_t c = (cls4[b >> 2] >> ((b & 3u) * 8u)) & 0xFFu;
		s = a.dense[(uint64_t)s * a.C + c];
		if (s == a.dead) break;                 /* the missing edge: fsm_exec returns 0 here, nothing fires after it */
		if (a.eoff[s + 1] != a.eoff[s]) cnt = trace_emit(a, i, s, (uint32_t)(t + 1 > 0xFFFFFFFFull ? 0xFFFFFFFFull : t + 1), cnt);
	}
	if (a.end_out) a.end_out[i] = a.fin[s];
	a.count_out[i] = cnt;
}

} // namespace fsmhip

#endif
