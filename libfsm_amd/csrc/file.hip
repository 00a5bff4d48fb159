/*
 * file.hip -- the large-input front: ONE input (a file, a big buffer) walked by the whole device.
 *
 * fsm_vm_match_file() keeps a struct vm_state across 4 KiB fread()s and stops reading as soon as the VM has decided
 * (src/libfsm/vm.c:188-216); re(1) -x hands it every file named on the command line (src/re/main.c:1106-1181).  A DFA walk
 * over one input is a dependent chain -- one lane, ~100 ns a byte: rounds 2-5 carried the state through
 * fsm_hip_exec_batch_resume() 64 KiB at a time and managed ~10 MB/s, seventy times slower than the reference's VM on one
 * host core.  Here the input is cut into CHUNK-byte pieces that are walked AT ONCE, one per lane, each from a GUESSED
 * state, and the guesses are corrected until they stand:
 *     in[0] = the state carried into the window, in[i] = START            (first pass)
 *     out[i] = delta*(in[i], piece i)                                      one fsm_hip_exec_batch_resume_device over all pieces
 *     in'[i] = out[i - 1]; stop when in' == in                             (the host compares two small arrays)
 * At the fixed point in[i] is delta*(carry, pieces 0 .. i-1) for every i -- by induction over i, whatever the guesses were
 * -- so the answer is exactly the sequential walk's.  Every pass makes at least one more piece right (piece 0 is right from
 * the start), so there are at most n passes: the bound is the sequential cost.  What makes it fast is that a DFA built from a
 * pattern FORGETS: walking a KiB of text from START and from the true state nearly always ends in the same state, so the
 * second pass already stands (a counter such as (aa)* does not forget and pays pass after pass; it is no slower than before).
 * The window's bytes cross PCIe once; the passes re-read them from HBM at the fixed-stride kernels' rate.
 * Reading stops once the state can no longer change -- DEAD (a missing edge: the VM's STOP fail) or an absorbing state (its
 * STOP success shortcut, vm/ir.c:763-766).  A read error gives 0, as the reference's ferror() check does.
 */
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/fsm_hip.h"
#include "dfa_access.h"

using namespace fsmhip;

namespace {

constexpr size_t CHUNK = 1024;                 /* bytes a lane walks per pass */
constexpr size_t WINDOW = (size_t)32 << 20;    /* bytes staged per window (two of them in flight: read k + 1 while k is walked) */
constexpr size_t SMALL = (size_t)256 << 10;    /* an input up to this size is one plain call */

int herr(hipError_t e)
{
	switch (e) {
	case hipSuccess: return 0;
	case hipErrorOutOfMemory: return ENOMEM;
	case hipErrorNoDevice:
	case hipErrorInvalidDevice: return ENODEV;
	default: return EIO;
	}
}
#define FTRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { errno = herr(e_); return -1; } } while (0)

struct Engine {
	const fsm_hip_dfa *d = nullptr;
	int dev = 0, prev = -1;
	hipStream_t s = nullptr;
	unsigned char *pin[2] = {nullptr, nullptr}, *dbuf[2] = {nullptr, nullptr};
	uint32_t *d_st = nullptr, *h_in = nullptr, *h_out = nullptr;   /* states: device array, pinned host copies */
	unsigned passes = 0, windows = 0;

	int open(const fsm_hip_dfa *dfa)
	{
		d = dfa;
		dev = dfa_device(dfa);
		(void)hipGetDevice(&prev);
		if (prev != dev) FTRY(hipSetDevice(dev));
		FTRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
		for (int k = 0; k < 2; k++) {
			FTRY(hipHostMalloc((void **)&pin[k], WINDOW, hipHostMallocDefault));
			FTRY(hipMalloc((void **)&dbuf[k], WINDOW));
		}
		const size_t n = WINDOW / CHUNK;
		FTRY(hipMalloc((void **)&d_st, n * 4u));
		FTRY(hipHostMalloc((void **)&h_in, n * 4u, hipHostMallocDefault));
		FTRY(hipHostMalloc((void **)&h_out, n * 4u, hipHostMallocDefault));
		return 0;
	}
	~Engine()
	{
		const int e = errno;
		if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
		for (int k = 0; k < 2; k++) { if (pin[k]) (void)hipHostFree(pin[k]); if (dbuf[k]) (void)hipFree(dbuf[k]); }
		if (d_st) (void)hipFree(d_st);
		if (h_in) (void)hipHostFree(h_in);
		if (h_out) (void)hipHostFree(h_out);
		if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
		errno = e;
	}
	/* the window's bytes on their way to the device (returns at once) */
	int upload(int k, size_t bytes)
	{
		FTRY(hipMemcpyAsync(dbuf[k], pin[k], bytes, hipMemcpyHostToDevice, s));
		return 0;
	}
	/* n whole pieces of window k, from `carry`: the state after them */
	int walk(int k, size_t n, uint32_t carry, uint32_t *out_state)
	{
		windows++;
		h_in[0] = carry;
		for (size_t i = 1; i < n; i++) h_in[i] = FSM_HIP_STATE_START;
		for (;;) {
			passes++;
			FTRY(hipMemcpyAsync(d_st, h_in, n * 4u, hipMemcpyHostToDevice, s));
			if (fsm_hip_exec_batch_resume_device(d, dbuf[k], CHUNK, nullptr, n, d_st, nullptr, nullptr, s) != 0) return -1;
			FTRY(hipMemcpyAsync(h_out, d_st, n * 4u, hipMemcpyDeviceToHost, s));
			FTRY(hipStreamSynchronize(s));
			bool same = true;
			for (size_t i = 1; i < n; i++) {
				if (h_in[i] != h_out[i - 1]) { h_in[i] = h_out[i - 1]; same = false; }
			}
			if (same) break;
		}
		*out_state = h_out[n - 1];
		return 0;
	}
};

bool settled(const fsm_hip_dfa *d, uint32_t st)
{
	return st == FSM_HIP_STATE_DEAD || fsm_hip_state_is_absorbing(d, st) == 1;
}

/* the last bytes of an input (fewer than a piece), or a small input: one plain call; *end = the caller's end state or NO_MATCH */
int tail_call(const fsm_hip_dfa *d, const unsigned char *p, size_t n, uint32_t *st, uint32_t *end)
{
	const uint32_t len = (uint32_t)n;
	unsigned char none = 0;
	return fsm_hip_exec_batch_resume(d, n ? p : &none, n ? n : 1u, &len, 1, st, end);
}

/* read: fills up to `cap` bytes, returns how many (0: the end), or (size_t)-1 on error */
template <class Read>
int match_stream(const fsm_hip_dfa *dfa, Read read, uint32_t *end_out, unsigned *passes, unsigned *windows)
{
	uint32_t st = FSM_HIP_STATE_START, end = FSM_HIP_NO_MATCH;
	/* the first SMALL bytes into plain memory: most inputs end there */
	std::vector<unsigned char> head(SMALL);
	size_t got = 0;
	while (got < SMALL) {
		const size_t r = read(head.data() + got, SMALL - got);
		if (r == (size_t)-1) return -2;
		if (r == 0) break;
		got += r;
	}
	if (got < SMALL) {
		if (tail_call(dfa, head.data(), got, &st, &end) != 0) return -1;
		*end_out = end;
		return 0;
	}
	Engine en;
	if (en.open(dfa) != 0) return -1;
	memcpy(en.pin[0], head.data(), SMALL);
	size_t have = SMALL;       /* bytes in the window being filled */
	int k = 0;
	bool eof = false;
	std::vector<unsigned char> rest;      /* the input's last bytes that are no whole piece */
	for (;;) {
		while (!eof && have < WINDOW) {
			const size_t r = read(en.pin[k] + have, WINDOW - have);
			if (r == (size_t)-1) return -2;
			if (r == 0) eof = true;
			else have += r;
		}
		const size_t n = have / CHUNK, whole = n * CHUNK;
		if (n != 0) {
			if (en.upload(k, whole) != 0) return -1;
			if (en.walk(k, n, st, &st) != 0) return -1;
		}
		if (eof || settled(dfa, st)) {
			rest.assign(en.pin[k] + whole, en.pin[k] + have);
			break;
		}
		/* (a full window is a whole number of pieces: nothing is carried over) */
		k ^= 1;
		have = 0;
	}
	if (passes) *passes = en.passes;
	if (windows) *windows = en.windows;
	if (settled(dfa, st) && !eof) rest.clear();      /* nothing that follows can change the state */
	if (tail_call(dfa, rest.data(), rest.size(), &st, &end) != 0) return -1;
	*end_out = end;
	return 0;
}

unsigned g_last_passes = 0, g_last_windows = 0;

} // namespace

extern "C" int fsm_hip_match_file(const struct fsm_hip_dfa *dfa, FILE *f)
{
	if (dfa == nullptr || f == nullptr) { errno = EINVAL; return -1; }
	uint32_t end = FSM_HIP_NO_MATCH;
	unsigned passes = 0, windows = 0;
	const int r = match_stream(dfa, [&](unsigned char *p, size_t cap) -> size_t {
		const size_t got = fread(p, 1, cap, f);
		if (got == 0 && ferror(f)) return (size_t)-1;
		return got;
	}, &end, &passes, &windows);
	g_last_passes = passes;
	g_last_windows = windows;
	if (r == -2 || ferror(f)) return 0;      /* a read error: no match, as vm.c:205-208 */
	if (r != 0) return -1;
	return end != FSM_HIP_NO_MATCH;
}

/* the same engine over memory: fsm_vm_match_buffer() for inputs worth the whole device (shim.c sends the small ones the plain way) */
extern "C" int fsm_hip_match_buffer_big(const struct fsm_hip_dfa *dfa, const char *buf, size_t n, uint32_t *end_state)
{
	if (dfa == nullptr || (n != 0 && buf == nullptr)) { errno = EINVAL; return -1; }
	size_t pos = 0;
	uint32_t end = FSM_HIP_NO_MATCH;
	unsigned passes = 0, windows = 0;
	const int r = match_stream(dfa, [&](unsigned char *p, size_t cap) -> size_t {
		const size_t take = n - pos < cap ? n - pos : cap;
		if (take) memcpy(p, buf + pos, take);
		pos += take;
		return take;
	}, &end, &passes, &windows);
	g_last_passes = passes;
	g_last_windows = windows;
	if (r != 0) return -1;
	if (end_state) *end_state = end;
	return end != FSM_HIP_NO_MATCH;
}

/* how the last fsm_hip_match_file / _buffer_big of this process went: windows walked and passes over them (2 per window when
 * every guess stood after the first correction; 0 / 0: the input was small and took one plain call) */
extern "C" void fsm_hip_match_last_passes(unsigned *windows, unsigned *passes)
{
	if (windows) *windows = g_last_windows;
	if (passes) *passes = g_last_passes;
}
