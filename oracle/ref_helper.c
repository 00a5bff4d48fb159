/*
 * oracle/ref_helper.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Thin C wrappers over the REAL reference libfsm/libre (built by
 * oracle/build_ref.sh from /root/reference into oracle/_ref/), so that Python
 * tests, the golden-vector generator and bench.py's cpu_baseline leg can
 *   - compile regexes / unions / Aho-Corasick string sets into DFAs,
 *   - run the literal reference fsm_exec() (src/libfsm/exec.c:85-167) and the
 *     DFAVM interpreters (src/libfsm/vm/v1.c:321-432, v2.c:248-333) over
 *     batches of (ptr,len) inputs.
 * Built into oracle/_ref/libref_helper.so, linked against libfsm_ref.so.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <fsm/fsm.h>
#include <fsm/bool.h>
#include <fsm/options.h>
#include <fsm/print.h>
#include <fsm/vm.h>
#include <re/re.h>
#include <re/strings.h>

/* (ptr,len) getc: fsm_sgetc would stop at the first NUL (src/libfsm/getc.c:25-29);
 * same shape as theft/wrap.c:19-36 in the reference. */
struct span {
	const unsigned char *p, *e;
};

static int
span_getc(void *opaque)
{
	struct span *s = opaque;
	if (s->p == s->e) {
		return EOF;
	}
	return *s->p++;
}

struct fsm *
rh_re_comp(int dialect, const char *re, int flags, int determinise, int minimise, long endid)
{
	const char *s = re;
	struct re_err err;
	struct fsm *fsm;

	fsm = re_comp((enum re_dialect) dialect, fsm_sgetc, &s, NULL, (enum re_flags) flags, &err);
	if (fsm == NULL) {
		return NULL;
	}
	if (determinise && !fsm_determinise(fsm)) {
		fsm_free(fsm);
		return NULL;
	}
	if (minimise && !fsm_minimise(fsm)) {
		fsm_free(fsm);
		return NULL;
	}
	if (endid >= 0 && !fsm_setendid(fsm, (fsm_end_id_t) endid)) {
		fsm_free(fsm);
		return NULL;
	}
	return fsm;
}

/* rx-style multi-pattern DFA: each pattern det+min+fsm_setendid(i), then
 * fsm_union_array + fsm_determinise, NOT minimised so end-ids stay distinct
 * (src/rx/main.c:1338-1385, src/re/main.c:857-987). */
struct fsm *
rh_union_res(int dialect, const char *const *res, size_t n, int flags)
{
	struct fsm **a;
	struct fsm *u;
	size_t i;

	a = calloc(n ? n : 1, sizeof *a);
	if (a == NULL) {
		return NULL;
	}
	for (i = 0; i < n; i++) {
		a[i] = rh_re_comp(dialect, res[i], flags, 1, 1, (long) i);
		if (a[i] == NULL) {
			size_t j;
			for (j = 0; j < i; j++) {
				fsm_free(a[j]);
			}
			free(a);
			return NULL;
		}
	}
	u = fsm_union_array(n, a, NULL);
	free(a);
	if (u == NULL) {
		return NULL;
	}
	if (!fsm_determinise(u)) {
		fsm_free(u);
		return NULL;
	}
	return u;
}

/* Aho-Corasick DFA over a word list (src/libre/re_strings.c:83-136);
 * with_endids: word i carries end-id i. */
struct fsm *
rh_re_strings(const char *const *words, const uint32_t *lens, size_t n, int flags, int with_endids)
{
	struct re_strings *g;
	struct fsm *fsm;
	size_t i;

	g = re_strings_new();
	if (g == NULL) {
		return NULL;
	}
	for (i = 0; i < n; i++) {
		fsm_end_id_t id = (fsm_end_id_t) i;
		if (!re_strings_add_raw(g, words[i], lens[i], with_endids ? &id : NULL)) {
			re_strings_free(g);
			return NULL;
		}
	}
	fsm = re_strings_build(g, NULL, (enum re_strings_flags) flags);
	re_strings_free(g);
	return fsm;
}

void
rh_fsm_free(struct fsm *fsm)
{
	fsm_free(fsm);
}

unsigned
rh_countstates(const struct fsm *fsm)
{
	return fsm_countstates(fsm);
}

int
rh_shuffle(struct fsm *fsm, unsigned seed)
{
	return fsm_shuffle(fsm, seed);
}

/* literal reference fsm_exec on one (ptr,len) input */
int
rh_exec(const struct fsm *fsm, const unsigned char *buf, size_t len, unsigned *end)
{
	struct span s;
	fsm_state_t st = 0;
	int r;

	s.p = buf;
	s.e = buf + len;
	r = fsm_exec(fsm, span_getc, &s, &st, NULL);
	if (r == 1 && end != NULL) {
		*end = st;
	}
	return r;
}

/* packed batch: input i = base[off[i] .. off[i+1]); end[i] = 0xFFFFFFFF unless ret[i]==1.
 * Returns seconds spent inside the fsm_exec loop (CLOCK_MONOTONIC, as reperf.c:614-648). */
double
rh_exec_batch(const struct fsm *fsm, const unsigned char *base, const uint64_t *off, size_t n,
	int8_t *ret, uint32_t *end)
{
	struct timespec t0, t1;
	size_t i;

	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (i = 0; i < n; i++) {
		unsigned e = 0xFFFFFFFFu;
		int r = rh_exec(fsm, base + off[i], (size_t) (off[i + 1] - off[i]), &e);
		if (ret != NULL) {
			ret[i] = (int8_t) r;
		}
		if (end != NULL) {
			end[i] = r == 1 ? e : 0xFFFFFFFFu;
		}
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

double
rh_exec_batch_stride(const struct fsm *fsm, const unsigned char *base, size_t stride, size_t n,
	int8_t *ret, uint32_t *end)
{
	struct timespec t0, t1;
	size_t i;

	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (i = 0; i < n; i++) {
		unsigned e = 0xFFFFFFFFu;
		int r = rh_exec(fsm, base + i * stride, stride, &e);
		if (ret != NULL) {
			ret[i] = (int8_t) r;
		}
		if (end != NULL) {
			end[i] = r == 1 ? e : 0xFFFFFFFFu;
		}
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

size_t
rh_endid_count(const struct fsm *fsm, unsigned state)
{
	return fsm_endid_count(fsm, state);
}

int
rh_endid_get(const struct fsm *fsm, unsigned state, size_t n, unsigned *buf)
{
	return fsm_endid_get(fsm, state, n, buf);
}

/* reference DFAVM: version 1 or 2 bytecode (src/libfsm/vm.c:88-131) */
struct fsm_dfavm *
rh_vm_compile(const struct fsm *fsm, int version)
{
	static const struct fsm_options opt_zero;
	struct fsm_options opt = opt_zero;
	struct fsm_vm_compile_opts vo;

	vo.flags = FSM_VM_COMPILE_DEFAULT_FLAGS;
	vo.output = version == 1 ? FSM_VM_COMPILE_VM_V1 : FSM_VM_COMPILE_VM_V2;
	vo.log = NULL;
	return fsm_vm_compile_with_options(fsm, &opt, vo);
}

void
rh_vm_free(struct fsm_dfavm *vm)
{
	fsm_vm_free(vm);
}

double
rh_vm_match_batch_stride(const struct fsm_dfavm *vm, const unsigned char *base, size_t stride, size_t n,
	int8_t *ret)
{
	struct timespec t0, t1;
	size_t i;

	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (i = 0; i < n; i++) {
		int r = fsm_vm_match_buffer(vm, (const char *) (base + i * stride), stride);
		if (ret != NULL) {
			ret[i] = (int8_t) r;
		}
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

double
rh_vm_match_batch(const struct fsm_dfavm *vm, const unsigned char *base, const uint64_t *off, size_t n,
	int8_t *ret)
{
	struct timespec t0, t1;
	size_t i;

	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (i = 0; i < n; i++) {
		int r = fsm_vm_match_buffer(vm, (const char *) (base + off[i]), (size_t) (off[i + 1] - off[i]));
		if (ret != NULL) {
			ret[i] = (int8_t) r;
		}
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

/* ---- eager outputs (src/libfsm/eager_output.c, exec.c:126-144) ------------------ */

/* The construction of tests/eager_output/utils.c:57-131: every pattern compiled with
 * RE_SAVE_LINKAGE_INFO, combined with fsm_union_repeated_pattern_group (eager output id =
 * id_base + index) or, with force_endids, fsm_setendid + fsm_union_array; then determinised
 * and minimised. */
struct fsm *
rh_union_repeated(int dialect, const char *const *res, size_t n, unsigned id_base, int force_endids)
{
	struct fsm **a;
	struct fsm *u;
	size_t i;

	a = calloc(n ? n : 1, sizeof *a);
	if (a == NULL) {
		return NULL;
	}
	for (i = 0; i < n; i++) {
		const char *s = res[i];
		a[i] = re_comp((enum re_dialect) dialect, fsm_sgetc, &s, NULL, RE_SAVE_LINKAGE_INFO, NULL);
		if (a[i] == NULL) {
			free(a);
			return NULL;
		}
		if (force_endids && !fsm_setendid(a[i], (fsm_end_id_t) (i + id_base))) {
			free(a);
			return NULL;
		}
	}
	u = force_endids ? fsm_union_array(n, a, NULL)
	                 : fsm_union_repeated_pattern_group(n, a, NULL, id_base);
	free(a);
	if (u == NULL) {
		return NULL;
	}
	if (!fsm_determinise(u) || !fsm_minimise(u)) {
		fsm_free(u);
		return NULL;
	}
	return u;
}

struct eager_acc {
	uint32_t *ids;
	uint32_t used, cap;
};

static void
eager_cb(fsm_output_id_t id, void *opaque)
{
	struct eager_acc *acc = opaque;
	uint32_t i;
	for (i = 0; i < acc->used; i++) {
		if (acc->ids[i] == id) {
			return;
		}
	}
	if (acc->used < acc->cap) {
		acc->ids[acc->used++] = id;
	}
}

/* literal fsm_exec with the eager-output callback installed: ids[i*cap .. i*cap+counts[i]) are the
 * distinct ids emitted while walking input i, in order of first emission -- emitted whether or not
 * the input finally matches (exec.c:126-144; the reference's tests drop them on reject). */
void
rh_exec_eager_batch(struct fsm *fsm, const unsigned char *base, const uint64_t *off, size_t n,
	int8_t *ret, uint32_t *end, uint32_t *ids, uint32_t *counts, uint32_t cap)
{
	size_t i;
	for (i = 0; i < n; i++) {
		struct eager_acc acc;
		unsigned e = 0xFFFFFFFFu;
		int r;
		acc.ids = ids + i * cap;
		acc.used = 0;
		acc.cap = cap;
		fsm_eager_output_set_cb(fsm, eager_cb, &acc);
		r = rh_exec(fsm, base + off[i], (size_t) (off[i + 1] - off[i]), &e);
		ret[i] = (int8_t) r;
		end[i] = r == 1 ? e : 0xFFFFFFFFu;
		counts[i] = acc.used;
	}
	fsm_eager_output_set_cb(fsm, NULL, NULL);
}

struct eager_stream {
	uint32_t *ids;
	uint32_t used, cap;
};

static void
eager_stream_cb(fsm_output_id_t id, void *opaque)
{
	struct eager_stream *acc = opaque;
	if (acc->used < acc->cap) {
		acc->ids[acc->used] = id;
	}
	acc->used++;
}

/* the raw callback stream of literal fsm_exec: every call, in call order (ids[i*cap .. ], counts[i] = calls made) */
void
rh_exec_eager_stream_batch(struct fsm *fsm, const unsigned char *base, const uint64_t *off, size_t n,
	int8_t *ret, uint32_t *end, uint32_t *ids, uint32_t *counts, uint32_t cap)
{
	size_t i;
	for (i = 0; i < n; i++) {
		struct eager_stream acc;
		unsigned e = 0xFFFFFFFFu;
		int r;
		acc.ids = ids + i * cap;
		acc.used = 0;
		acc.cap = cap;
		fsm_eager_output_set_cb(fsm, eager_stream_cb, &acc);
		r = rh_exec(fsm, base + off[i], (size_t) (off[i + 1] - off[i]), &e);
		ret[i] = (int8_t) r;
		end[i] = r == 1 ? e : 0xFFFFFFFFu;
		counts[i] = acc.used;
	}
	fsm_eager_output_set_cb(fsm, NULL, NULL);
}

size_t
rh_eager_output_count(const struct fsm *fsm, unsigned state)
{
	return fsm_eager_output_count(fsm, state);
}

/* ---- golden .fsm corpus (tests/{pcre,native,glob,...}/out*.fsm) ------------------ */

#include <fsm/parser.h>
#include <fsm/walk.h>

/* fsm_parse + determinise + minimise: the checked-in expected automata of the reference's
 * golden-DFA tests (tests/pcre/Makefile:40-63) as executable DFAs. */
struct fsm *
rh_parse_fsm_file(const char *path)
{
	FILE *f = fopen(path, "r");
	struct fsm *fsm;
	if (f == NULL) {
		return NULL;
	}
	fsm = fsm_parse(f, NULL);
	fclose(f);
	if (fsm == NULL) {
		return NULL;
	}
	if (!fsm_determinise(fsm) || !fsm_minimise(fsm)) {
		fsm_free(fsm);
		return NULL;
	}
	return fsm;
}

struct gen_env {
	unsigned char *buf;   /* [cap][maxlen] */
	uint32_t *lens;
	size_t cap, used, maxlen;
};

static enum fsm_generate_matches_cb_res
gen_cb(const struct fsm *fsm, size_t depth, size_t match_count, size_t steps,
	const char *input, size_t input_length, fsm_state_t end_state, void *opaque)
{
	struct gen_env *env = opaque;
	(void) fsm; (void) depth; (void) match_count; (void) end_state;
	if (steps > 200000) {
		return FSM_GENERATE_MATCHES_CB_RES_HALT;
	}
	if (input_length <= env->maxlen) {
		memcpy(env->buf + env->used * env->maxlen, input, input_length);
		env->lens[env->used++] = (uint32_t) input_length;
	}
	return env->used >= env->cap ? FSM_GENERATE_MATCHES_CB_RES_HALT : FSM_GENERATE_MATCHES_CB_RES_CONTINUE;
}

/* up to cap accepted inputs from fsm_generate_matches (src/libfsm/gen.c:143) on a CLONE (it trims) */
size_t
rh_generate_matches(const struct fsm *fsm, size_t maxlen, unsigned seed, unsigned char *buf, uint32_t *lens, size_t cap)
{
	struct gen_env env;
	struct fsm *c = fsm_clone(fsm);
	if (c == NULL) {
		return 0;
	}
	env.buf = buf;
	env.lens = lens;
	env.cap = cap;
	env.used = 0;
	env.maxlen = maxlen;
	srand(seed);
	(void) fsm_generate_matches(c, maxlen, 1, gen_cb, &env);
	fsm_free(c);
	return env.used;
}

/* ---- all-cores CPU baseline ------------------------------------------------------ */

#include <pthread.h>

struct mt_job {
	const struct fsm_dfavm *vm;
	const struct fsm *fsm;
	const unsigned char *base;
	size_t stride, first, count;
	int reps;
	unsigned long matched;
};

static void *
mt_worker(void *p)
{
	struct mt_job *j = p;
	unsigned long m = 0;
	int r;
	size_t i;
	for (r = 0; r < j->reps; r++) {
		for (i = 0; i < j->count; i++) {
			const unsigned char *row = j->base + (j->first + i) * j->stride;
			if (j->vm != NULL) {
				m += (unsigned long) fsm_vm_match_buffer(j->vm, (const char *) row, j->stride);
			} else {
				unsigned e;
				m += rh_exec(j->fsm, row, j->stride, &e) == 1;
			}
		}
	}
	j->matched = m;
	return NULL;
}

/* The reference's matcher on nthreads host threads: the n rows are split into contiguous slices,
 * one per thread, each walked `reps` times (the compiled VM / the fsm are shared read-only; the
 * reference has no threading of its own, src/libfsm is single-threaded).  vm != NULL: DFAVM
 * fsm_vm_match_buffer; else literal fsm_exec.  Returns wall seconds; *matched = accepts of one pass. */
double
rh_match_threads(const struct fsm_dfavm *vm, const struct fsm *fsm, const unsigned char *base, size_t stride, size_t n,
	int nthreads, int reps, unsigned long *matched)
{
	pthread_t *th;
	struct mt_job *jobs;
	struct timespec t0, t1;
	int t;
	unsigned long total = 0;

	if (nthreads < 1) {
		nthreads = 1;
	}
	th = calloc((size_t) nthreads, sizeof *th);
	jobs = calloc((size_t) nthreads, sizeof *jobs);
	if (th == NULL || jobs == NULL) {
		free(th);
		free(jobs);
		return -1.0;
	}
	for (t = 0; t < nthreads; t++) {
		jobs[t].vm = vm;
		jobs[t].fsm = fsm;
		jobs[t].base = base;
		jobs[t].stride = stride;
		jobs[t].first = n * (size_t) t / (size_t) nthreads;
		jobs[t].count = n * (size_t) (t + 1) / (size_t) nthreads - jobs[t].first;
		jobs[t].reps = reps;
	}
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (t = 0; t < nthreads; t++) {
		if (pthread_create(&th[t], NULL, mt_worker, &jobs[t]) != 0) {
			nthreads = t;
			break;
		}
	}
	for (t = 0; t < nthreads; t++) {
		pthread_join(th[t], NULL);
		total += jobs[t].matched;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if (matched != NULL) {
		*matched = reps > 0 ? total / (unsigned long) reps : 0;
	}
	free(th);
	free(jobs);
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}


/* ---- CPU baseline lines SURVEY.md section 8(d) asks for beside the literal fsm_exec ---------------- */

/*
 * (2) fsm_exec with its per-call precondition hoisted.  NOT the reference: build_ref.sh derives
 * fsm_exec_hoisted() from the reference's src/libfsm/exec.c at build time (a sed of a scratch copy under
 * oracle/_ref/obj/: the function renamed, the `if (!fsm_all(fsm, fsm_isdfa))` block of exec.c:106-109
 * removed), so that the loop itself -- indirect getc, edge_set_find's linear scan -- is timed without the
 * O(states) sweep that dominates big DFAs.  Same signature and results as fsm_exec on a DFA.
 */
int fsm_exec_hoisted(const struct fsm *fsm, int (*fsm_getc)(void *opaque), void *opaque,
	fsm_state_t *end, struct fsm_capture *captures);

double
rh_exec_hoisted_batch_stride(const struct fsm *fsm, const unsigned char *base, size_t stride, size_t n,
	int8_t *ret, uint32_t *end)
{
	struct timespec t0, t1;
	size_t i;

	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (i = 0; i < n; i++) {
		struct span s;
		fsm_state_t st = 0;
		int r;
		s.p = base + i * stride;
		s.e = s.p + stride;
		r = fsm_exec_hoisted(fsm, span_getc, &s, &st, NULL);
		if (ret != NULL) {
			ret[i] = (int8_t) r;
		}
		if (end != NULL) {
			end[i] = r == 1 ? st : 0xFFFFFFFFu;
		}
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}

/*
 * (4) the generated matcher `retest -l vmc|c` compiles and dlopens (src/retest/runner.c:63-135, :290-404):
 * fsm_print() with retest's own options (anonymous states, consolidated edges, comments, io = PAIR,
 * src/retest/main.c:1225-1229) writes `int fsm_main(const char *b, const char *e)` to `path`; the
 * caller compiles it with the C compiler and hands the dlsym'd function to rh_codegen_match_batch_stride.
 * lang: 0 = FSM_PRINT_VMC, 1 = FSM_PRINT_C; comments: retest's opt.comments (example strings in the
 * generated code; no effect on the matcher).  Returns 0, or -1 + errno.
 */
int
rh_print_matcher(const struct fsm *fsm, int lang, int comments, const char *path)
{
	struct fsm_options opt;
	FILE *f;
	int e;

	memset(&opt, 0, sizeof opt);
	opt.anonymous_states = 1;
	opt.consolidate_edges = 1;
	opt.comments = comments != 0;   /* retest sets 1; the per-state example strings take minutes on a 4k-state DFA */
	opt.io = FSM_IO_PAIR;
	f = fopen(path, "w");
	if (f == NULL) {
		return -1;
	}
	if (lang == 0) {
		fprintf(f, "#include <string.h>\n\n");   /* the vmc codegen may emit memcmp/strncmp (runner.c:81-84) */
	}
	e = fsm_print(f, fsm, &opt, NULL, lang == 0 ? FSM_PRINT_VMC : FSM_PRINT_C);
	if (fclose(f) == EOF || e == -1) {
		return -1;
	}
	return 0;
}

double
rh_codegen_match_batch_stride(int (*fsm_main)(const char *, const char *), const unsigned char *base, size_t stride, size_t n,
	int8_t *ret)
{
	struct timespec t0, t1;
	size_t i;

	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (i = 0; i < n; i++) {
		const char *b = (const char *) base + i * stride;
		ret[i] = (int8_t) (fsm_main(b, b + stride) != 0);
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	return (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
}
