/*
 * oracle/dfa_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement, in plain C, of the reference's DFA execution path, used as
 * the parity checker for the HIP kernels (tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg only; the product never links or calls this).
 *
 * Parity pinning: this restatement is checked (tests/test_oracle.py) against
 *   - the reference's own retest fixtures tests/retest/*.tst (37 regexes,
 *     115 +/- cases) and the endids / re_strings known-answer programs,
 *     frozen as tests/golden/**.npz by tests/golden/make_golden.py
 *     from outputs of the REAL reference fsm_exec (oracle/_ref), and
 *   - live, against oracle/_ref's fsm_exec on seeded random inputs.
 *
 * What is restated (reference file:line):
 *   struct edge_group {uint64 symbols[4]; fsm_state_t to}   src/adt/edgeset.c:34-41
 *   edge_set_find: linear scan, first group with the bit    src/adt/edgeset.c:393-418
 *   edge_set_transition                                     src/adt/edgeset.c:564-579
 *   fsm_exec: -1/EINVAL without start; per byte transition,
 *     missing edge -> return 0 at once; at EOF return 0 unless
 *     fsm_isend(state); else *end = state, return 1          src/libfsm/exec.c:85-167
 *   fsm_endid_count / fsm_endid_get (sorted unique ids,
 *     0 = buffer too small, 1 otherwise)                     src/libfsm/endids.c:653-755
 *   eager outputs: the ids attached to the start state and to every
 *     state entered are delivered through the callback, in order
 *     (oracle_exec_eager_stride below)                      src/libfsm/exec.c:126-144
 * Captures (exec.c:41-44) are outside the accelerated path and are not
 * restated.
 */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

struct oracle_range {
	uint8_t lo, hi;
	uint16_t reserved;
	uint32_t to;
};

struct edge_group {
	uint64_t symbols[4];
	uint32_t to;
};

struct ostate {
	struct edge_group *groups; /* sorted by .to, like the reference keeps them */
	size_t count;
	unsigned end:1;
};

struct oracle_dfa {
	struct ostate *states;
	uint32_t statecount;
	uint32_t start;
	int hasstart;
	uint32_t *endid_off; /* statecount+1, may be NULL */
	uint32_t *endids;
	uint32_t *dense;     /* lazily built [statecount+1][256] for the table walker */
};

#define SYMBOLS_GET(S, C) (((S)[(C) / 64] >> ((C) & 63)) & 1u)
#define SYMBOLS_SET(S, C) ((S)[(C) / 64] |= (uint64_t) 1 << ((C) & 63))

void
oracle_dfa_free(struct oracle_dfa *d)
{
	uint32_t s;
	if (d == NULL) {
		return;
	}
	if (d->states != NULL) {
		for (s = 0; s < d->statecount; s++) {
			free(d->states[s].groups);
		}
	}
	free(d->states);
	free(d->endid_off);
	free(d->endids);
	free(d->dense);
	free(d);
}

static int
cmp_group(const void *a, const void *b)
{
	const struct edge_group *x = a, *y = b;
	return (x->to > y->to) - (x->to < y->to);
}

struct oracle_dfa *
oracle_dfa_new(uint32_t nstates, uint32_t start, int hasstart,
	const uint32_t *edge_off, const struct oracle_range *ranges, const uint8_t *is_end,
	const uint32_t *endid_off, const uint32_t *endids)
{
	struct oracle_dfa *d;
	uint32_t s, k;

	d = calloc(1, sizeof *d);
	if (d == NULL) {
		return NULL;
	}
	d->statecount = nstates;
	d->start = start;
	d->hasstart = hasstart;
	d->states = calloc(nstates ? nstates : 1, sizeof *d->states);
	if (d->states == NULL) {
		goto fail;
	}
	for (s = 0; s < nstates; s++) {
		struct ostate *st = &d->states[s];
		uint32_t a = edge_off[s], b = edge_off[s + 1];
		st->end = is_end[s] ? 1 : 0;
		st->groups = calloc((b - a) ? (b - a) : 1, sizeof *st->groups);
		if (st->groups == NULL) {
			goto fail;
		}
		for (k = a; k < b; k++) {
			size_t g;
			unsigned c;
			for (g = 0; g < st->count; g++) {
				if (st->groups[g].to == ranges[k].to) {
					break;
				}
			}
			if (g == st->count) {
				st->groups[g].to = ranges[k].to;
				st->count++;
			}
			for (c = ranges[k].lo; c <= ranges[k].hi; c++) {
				SYMBOLS_SET(st->groups[g].symbols, c);
			}
		}
		qsort(st->groups, st->count, sizeof *st->groups, cmp_group);
	}
	if (endid_off != NULL) {
		size_t total = endid_off[nstates];
		d->endid_off = malloc(((size_t) nstates + 1) * sizeof *d->endid_off);
		d->endids = malloc((total ? total : 1) * sizeof *d->endids);
		if (d->endid_off == NULL || d->endids == NULL) {
			goto fail;
		}
		memcpy(d->endid_off, endid_off, ((size_t) nstates + 1) * sizeof *d->endid_off);
		if (total) {
			memcpy(d->endids, endids, total * sizeof *d->endids);
		}
	}
	return d;
fail:
	oracle_dfa_free(d);
	return NULL;
}

/* edge_set_find + edge_set_transition */
static int
edge_set_transition(const struct ostate *st, unsigned char symbol, uint32_t *state)
{
	size_t i;
	for (i = 0; i < st->count; i++) {
		const struct edge_group *eg = &st->groups[i];
		if (SYMBOLS_GET(eg->symbols, symbol)) {
			*state = eg->to;
			return 1;
		}
	}
	return 0;
}

/* fsm_exec over a (ptr,len) stream */
int
oracle_exec(const struct oracle_dfa *d, const unsigned char *buf, size_t len, uint32_t *end)
{
	uint32_t state;
	size_t i;

	if (!d->hasstart) {
		errno = EINVAL;
		return -1;
	}
	state = d->start;
	for (i = 0; i < len; i++) {
		if (!edge_set_transition(&d->states[state], buf[i], &state)) {
			return 0;
		}
	}
	if (!d->states[state].end) {
		return 0;
	}
	*end = state;
	return 1;
}

static double
now_s(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

/* packed batch; end[i] = 0xFFFFFFFF unless ret[i] == 1; returns loop seconds */
double
oracle_exec_batch(const struct oracle_dfa *d, const unsigned char *base, const uint64_t *off, size_t n,
	int8_t *ret, uint32_t *end)
{
	double t0 = now_s();
	size_t i;
	for (i = 0; i < n; i++) {
		uint32_t e = 0xFFFFFFFFu;
		int r = oracle_exec(d, base + off[i], (size_t) (off[i + 1] - off[i]), &e);
		if (ret != NULL) {
			ret[i] = (int8_t) r;
		}
		if (end != NULL) {
			end[i] = r == 1 ? e : 0xFFFFFFFFu;
		}
	}
	return now_s() - t0;
}

double
oracle_exec_batch_stride(const struct oracle_dfa *d, const unsigned char *base, size_t stride,
	const uint32_t *len, size_t n, int8_t *ret, uint32_t *end)
{
	double t0 = now_s();
	size_t i;
	for (i = 0; i < n; i++) {
		uint32_t e = 0xFFFFFFFFu;
		int r = oracle_exec(d, base + i * stride, len ? len[i] : stride, &e);
		if (ret != NULL) {
			ret[i] = (int8_t) r;
		}
		if (end != NULL) {
			end[i] = r == 1 ? e : 0xFFFFFFFFu;
		}
	}
	return now_s() - t0;
}

/*
 * Dense-table walker: same function, O(1) per byte, for full-size parity
 * sweeps where the group scan above would take hours.  It is itself verified
 * against oracle_exec in tests/test_oracle.py.  Row `statecount` is the
 * "missing edge" sink.
 */
static int
build_dense(struct oracle_dfa *d)
{
	uint32_t s, S = d->statecount;
	if (d->dense != NULL) {
		return 1;
	}
	d->dense = malloc(((size_t) S + 1) * 256 * sizeof *d->dense);
	if (d->dense == NULL) {
		return 0;
	}
	for (s = 0; s <= S; s++) {
		unsigned c;
		for (c = 0; c < 256; c++) {
			uint32_t t = S;
			if (s < S && !edge_set_transition(&d->states[s], (unsigned char) c, &t)) {
				t = S;
			}
			d->dense[(size_t) s * 256 + c] = t;
		}
	}
	return 1;
}

double
oracle_table_walk_stride(struct oracle_dfa *d, const unsigned char *base, size_t stride,
	const uint32_t *len, size_t n, uint32_t *end)
{
	double t0;
	size_t i;
	uint32_t S = d->statecount;

	if (!d->hasstart || !build_dense(d)) {
		return -1.0;
	}
	t0 = now_s();
	for (i = 0; i < n; i++) {
		const unsigned char *p = base + i * stride;
		size_t l = len ? len[i] : stride, t;
		uint32_t st = d->start;
		for (t = 0; t < l; t++) {
			st = d->dense[(size_t) st * 256 + p[t]];
		}
		end[i] = (st < S && d->states[st].end) ? st : 0xFFFFFFFFu;
	}
	return now_s() - t0;
}

/*
 * The same dense-table walk on `nthreads` host threads over contiguous slices of the rows: the checker of
 * bench.py's --full-parity leg ("full N compare GPU vs CPU table-walker"), where all N
 * inputs are streamed back from the device and walked here.  The table walker is itself proven equal to
 * the edge-group restatement (and so to fsm_exec) by tests/test_oracle.py.  Returns wall seconds.
 */
struct walk_job {
	const struct oracle_dfa *d;
	const unsigned char *base;
	size_t stride, first, count;
	const uint32_t *len;      /* or NULL: every input is `stride` bytes */
	uint32_t *end;
};

static void *
walk_worker(void *opaque)
{
	struct walk_job *j = opaque;
	const struct oracle_dfa *d = j->d;
	const uint32_t S = d->statecount;
	size_t i, t;
	for (i = j->first; i < j->first + j->count; i++) {
		const unsigned char *p = j->base + i * j->stride;
		const size_t l = j->len ? j->len[i] : j->stride;
		uint32_t st = d->start;
		for (t = 0; t < l; t++) {
			st = d->dense[(size_t) st * 256 + p[t]];
		}
		j->end[i] = (st < S && d->states[st].end) ? st : 0xFFFFFFFFu;
	}
	return NULL;
}

double oracle_table_walk_lens_mt(struct oracle_dfa *d, const unsigned char *base, size_t stride, const uint32_t *len, size_t n, uint32_t *end, int nthreads);

double
oracle_table_walk_stride_mt(struct oracle_dfa *d, const unsigned char *base, size_t stride, size_t n, uint32_t *end, int nthreads)
{
	return oracle_table_walk_lens_mt(d, base, stride, NULL, n, end, nthreads);
}

/* the same with per-input lengths (len[i] <= stride; NULL: whole rows): the lines of the packed-lines bench legs */
double
oracle_table_walk_lens_mt(struct oracle_dfa *d, const unsigned char *base, size_t stride, const uint32_t *len, size_t n, uint32_t *end, int nthreads)
{
	pthread_t th[256];
	struct walk_job jobs[256];
	double t0;
	int t, started = 0;

	if (!d->hasstart || !build_dense(d)) {
		return -1.0;
	}
	if (nthreads < 1) {
		nthreads = 1;
	}
	if (nthreads > 256) {
		nthreads = 256;
	}
	t0 = now_s();
	for (t = 0; t < nthreads; t++) {
		jobs[t].d = d;
		jobs[t].base = base;
		jobs[t].stride = stride;
		jobs[t].first = n * (size_t) t / (size_t) nthreads;
		jobs[t].count = n * (size_t) (t + 1) / (size_t) nthreads - jobs[t].first;
		jobs[t].len = len;
		jobs[t].end = end;
		if (pthread_create(&th[t], NULL, walk_worker, &jobs[t]) != 0) {
			walk_worker(&jobs[t]);   /* no thread to be had: walk the slice here */
			th[t] = (pthread_t) 0;
			continue;
		}
		started |= 1;
	}
	for (t = 0; t < nthreads; t++) {
		if (th[t] != (pthread_t) 0) {
			pthread_join(th[t], NULL);
		}
	}
	(void) started;
	return now_s() - t0;
}

/* ... and over inputs packed back to back (input i = base[off[i] .. off[i+1]): the packed-lines front's full-parity checker;
 * off[] is relative to `base`, threads take contiguous ranges of inputs */
struct packed_job {
	const struct oracle_dfa *d;
	const unsigned char *base;
	const uint64_t *off;
	size_t first, count;
	uint32_t *end;
};

static void *
packed_worker(void *opaque)
{
	struct packed_job *j = opaque;
	const struct oracle_dfa *d = j->d;
	const uint32_t S = d->statecount;
	size_t i;
	for (i = j->first; i < j->first + j->count; i++) {
		const unsigned char *p = j->base + j->off[i], *e = j->base + j->off[i + 1];
		uint32_t st = d->start;
		for (; p < e; p++) {
			st = d->dense[(size_t) st * 256 + *p];
		}
		j->end[i] = (st < S && d->states[st].end) ? st : 0xFFFFFFFFu;
	}
	return NULL;
}

double
oracle_table_walk_packed_mt(struct oracle_dfa *d, const unsigned char *base, const uint64_t *off, size_t n, uint32_t *end, int nthreads)
{
	pthread_t th[256];
	struct packed_job jobs[256];
	double t0;
	int t;

	if (!d->hasstart || !build_dense(d)) {
		return -1.0;
	}
	if (nthreads < 1) nthreads = 1;
	if (nthreads > 256) nthreads = 256;
	t0 = now_s();
	for (t = 0; t < nthreads; t++) {
		jobs[t].d = d;
		jobs[t].base = base;
		jobs[t].off = off;
		jobs[t].first = n * (size_t) t / (size_t) nthreads;
		jobs[t].count = n * (size_t) (t + 1) / (size_t) nthreads - jobs[t].first;
		jobs[t].end = end;
		if (pthread_create(&th[t], NULL, packed_worker, &jobs[t]) != 0) {
			packed_worker(&jobs[t]);
			th[t] = (pthread_t) 0;
		}
	}
	for (t = 0; t < nthreads; t++) {
		if (th[t] != (pthread_t) 0) {
			pthread_join(th[t], NULL);
		}
	}
	return now_s() - t0;
}

size_t
oracle_endid_count(const struct oracle_dfa *d, uint32_t state)
{
	if (d->endid_off == NULL || state >= d->statecount) {
		return 0;
	}
	return d->endid_off[state + 1] - d->endid_off[state];
}

int
oracle_endid_get(const struct oracle_dfa *d, uint32_t state, size_t n, uint32_t *buf)
{
	size_t cnt = oracle_endid_count(d, state), k;
	if (cnt > n) {
		return 0;
	}
	for (k = 0; k < cnt; k++) {
		buf[k] = d->endids[d->endid_off[state] + k];
	}
	return 1;
}

/*
 * Streaming form: start from state_io[i] (0xFFFFFFFD = start state, 0xFFFFFFFC = already dead),
 * consume the bytes, leave the state reached (or 0xFFFFFFFC after a missing edge) in state_io[i].
 * This is fsm_exec's loop (exec.c:132-151) with the state exposed, the way fsm_vm_match_file
 * carries struct vm_state across chunks (vm.c:188-216).
 */
void
oracle_state_walk_stride(const struct oracle_dfa *d, const unsigned char *base, size_t stride,
	const uint32_t *len, size_t n, uint32_t *state_io)
{
	size_t i;
	for (i = 0; i < n; i++) {
		const unsigned char *p = base + i * stride;
		size_t l = len ? len[i] : stride, t;
		uint32_t st = state_io[i];
		if (st == 0xFFFFFFFDu) {
			st = d->start;
		}
		if (st == 0xFFFFFFFCu) {
			continue;
		}
		for (t = 0; t < l; t++) {
			if (!edge_set_transition(&d->states[st], p[t], &st)) {
				st = 0xFFFFFFFCu;
				break;
			}
		}
		state_io[i] = st;
	}
}

int
oracle_isend(const struct oracle_dfa *d, uint32_t state)
{
	return state < d->statecount && d->states[state].end;
}

/*
 * fsm_exec with the eager-output side channel (exec.c:126-144): the ids attached to the start
 * state are emitted before any input, then those of every state entered; a missing edge stops the
 * walk (return 0) but what was emitted stays emitted.  ids[i*cap ...] receives the distinct ids in
 * order of first emission, counts[i] how many.  eager_off/eager_ids: CSR by state.
 */
static void
emit_state(const uint32_t *eager_off, const uint32_t *eager_ids, uint32_t state,
	uint32_t *ids, uint32_t *used, uint32_t cap)
{
	uint32_t k, j;
	for (k = eager_off[state]; k < eager_off[state + 1]; k++) {
		for (j = 0; j < *used; j++) {
			if (ids[j] == eager_ids[k]) {
				break;
			}
		}
		if (j == *used && *used < cap) {
			ids[(*used)++] = eager_ids[k];
		}
	}
}

void
oracle_exec_eager_stride(const struct oracle_dfa *d, const uint32_t *eager_off, const uint32_t *eager_ids,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	int8_t *ret, uint32_t *end, uint32_t *ids, uint32_t *counts, uint32_t cap)
{
	size_t i;
	for (i = 0; i < n; i++) {
		const unsigned char *p = base + i * stride;
		size_t l = len ? len[i] : stride, t;
		uint32_t st = d->start, used = 0;
		int r = 1;
		emit_state(eager_off, eager_ids, st, ids + i * cap, &used, cap);
		for (t = 0; t < l; t++) {
			if (!edge_set_transition(&d->states[st], p[t], &st)) {
				r = 0;
				break;
			}
			emit_state(eager_off, eager_ids, st, ids + i * cap, &used, cap);
		}
		if (r && !d->states[st].end) {
			r = 0;
		}
		ret[i] = (int8_t) r;
		end[i] = r ? st : 0xFFFFFFFFu;
		counts[i] = used;
	}
}

/*
 * The callback STREAM of the same walk, nothing folded: one record per callback call of exec.c:62-78, in call
 * order, repeats included -- ids[i*cap + k] the id, pos[i*cap + k] the bytes consumed when it fired (0 for the
 * start state's, exec.c:126-130; t + 1 for the state entered on byte t, :140-144).  counts[i] = number of calls
 * (may exceed cap: only the first cap are recorded).  Within one state the ids come in eager_ids[] order.
 */
void
oracle_exec_eager_trace(const struct oracle_dfa *d, const uint32_t *eager_off, const uint32_t *eager_ids,
	const unsigned char *base, const uint64_t *off, size_t stride, const uint32_t *len, size_t n,
	int8_t *ret, uint32_t *end, uint32_t *ids, uint32_t *pos, uint32_t *counts, uint32_t cap)
{
	size_t i;
	for (i = 0; i < n; i++) {
		const unsigned char *p = off ? base + off[i] : base + i * stride;
		size_t l = off ? (size_t) (off[i + 1] - off[i]) : len ? len[i] : stride, t;
		uint32_t st = d->start, used = 0, k;
		int r = 1;
		for (k = eager_off[st]; k < eager_off[st + 1]; k++, used++) {
			if (used < cap) { ids[i * cap + used] = eager_ids[k]; pos[i * cap + used] = 0; }
		}
		for (t = 0; t < l; t++) {
			if (!edge_set_transition(&d->states[st], p[t], &st)) {
				r = 0;
				break;
			}
			for (k = eager_off[st]; k < eager_off[st + 1]; k++, used++) {
				if (used < cap) { ids[i * cap + used] = eager_ids[k]; pos[i * cap + used] = (uint32_t) (t + 1); }
			}
		}
		if (r && !d->states[st].end) {
			r = 0;
		}
		ret[i] = (int8_t) r;
		end[i] = r ? st : 0xFFFFFFFFu;
		counts[i] = used;
	}
}
