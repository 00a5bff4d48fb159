/* tests/c/stub_fsm_hip.c -- TEST DOUBLE, not part of the product and never shipped.
 *
 * The patched retest (integration/retest) links libfsm_hip.so.  This file builds a stand-in with the
 * same soname for the CPU test suite (tests/test_retest_patch.py), so the control flow the patch adds to
 * the reference's retest -- read ahead to the end of a record, one fsm_hip_exec_batch_offsets() call for
 * its test lines, results handed out by fsm_runner_run() -- can be checked on a box without a GPU.  The
 * entry points the patch calls are answered by the reference's own DFAVM (fsm_vm_*, resolved from
 * the retest executable, which contains libfsm), and each call is counted: the counts are printed when
 * the process exits.  The GPU suite runs the same retest binary against the real library. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

struct fsm;
struct fsm_dfavm;
struct fsm_capture;
int fsm_exec(const struct fsm *fsm, int (*fsm_getc)(void *opaque), void *opaque, unsigned *end, struct fsm_capture *captures);
int fsm_vm_match_file(const struct fsm_dfavm *vm, FILE *f);
struct fsm_dfavm *fsm_vm_compile(const struct fsm *fsm);
int fsm_vm_match_buffer(const struct fsm_dfavm *vm, const char *buf, size_t n);
void fsm_vm_free(struct fsm_dfavm *vm);

struct fsm_hip_dfa { struct fsm_dfavm *vm; const struct fsm *fsm; /* valid while the caller keeps it (re(1) does; retest does not) */ };

struct span { const unsigned char *p, *e; };
static int span_getc(void *o) { struct span *s = o; return s->p == s->e ? -1 : *s->p++; }

static unsigned long n_compile, n_batch, n_batch_inputs, n_single, n_stride, n_stride_inputs, n_multi, n_multi_jobs, n_multi_inputs;

static void
report(void)
{
	fprintf(stderr, "stub_fsm_hip: compile=%lu batch_calls=%lu batch_inputs=%lu single_calls=%lu stride_calls=%lu stride_inputs=%lu multi_calls=%lu multi_jobs=%lu multi_inputs=%lu\n",
		n_compile, n_batch, n_batch_inputs, n_single, n_stride, n_stride_inputs, n_multi, n_multi_jobs, n_multi_inputs);
}

struct fsm_hip_dfa *
fsm_hip_compile(const struct fsm *fsm, unsigned flags)
{
	struct fsm_hip_dfa *d;
	(void) flags;
	if (n_compile++ == 0) {
		atexit(report);
	}
	d = malloc(sizeof *d);
	if (d == NULL) {
		return NULL;
	}
	d->vm = fsm_vm_compile(fsm);
	d->fsm = fsm;
	if (d->vm == NULL) {
		free(d);
		return NULL;
	}
	return d;
}

void
fsm_hip_dfa_free(struct fsm_hip_dfa *d)
{
	if (d != NULL) {
		fsm_vm_free(d->vm);
		free(d);
	}
}

int
fsm_hip_match_buffer(const struct fsm_hip_dfa *d, const char *buf, size_t n)
{
	n_single++;
	return fsm_vm_match_buffer(d->vm, buf, n);
}

int
fsm_hip_exec_batch_offsets(const struct fsm_hip_dfa *d, const unsigned char *base, const uint64_t *off, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap)
{
	size_t i;
	n_batch++;
	n_batch_inputs += n;
	for (i = 0; i < n; i++) {
		if (accept_bitmap != NULL && fsm_vm_match_buffer(d->vm, (const char *) base + off[i], (size_t) (off[i + 1] - off[i]))) {
			accept_bitmap[i / 64] |= (uint64_t) 1 << (i % 64);
		}
		if (end_out != NULL) {   /* the state fsm_exec returns (re -H: the caller's fsm is still alive) */
			struct span sp = { base + off[i], base + off[i + 1] };
			unsigned end = 0;
			end_out[i] = fsm_exec(d->fsm, span_getc, &sp, &end, NULL) == 1 ? end : 0xFFFFFFFFu;
		}
	}
	return 0;
}

int
fsm_hip_exec_batch(const struct fsm_hip_dfa *d, const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap)
{
	size_t i;
	(void) end_out;
	n_stride++;
	n_stride_inputs += n;
	for (i = 0; i < n; i++) {
		if (fsm_vm_match_buffer(d->vm, (const char *) base + i * stride, len != NULL ? len[i] : stride)) {
			accept_bitmap[i / 64] |= (uint64_t) 1 << (i % 64);
		}
	}
	return 0;
}

/* the many-DFA front (`retest -l hip`: a whole .tst file per submission): job by job through the reference's VM */
struct fsm_hip_multi_batch {
	const unsigned char *base;
	const uint64_t *off;
	size_t n;
	uint32_t *end_out;
	uint64_t *accept_bitmap;
};

int
fsm_hip_exec_multi(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch *b, size_t k)
{
	size_t q, i;
	n_multi++;
	n_multi_jobs += k;
	for (q = 0; q < k; q++) {
		n_multi_inputs += b[q].n;
		for (i = 0; i < b[q].n; i++) {
			if (b[q].accept_bitmap != NULL && fsm_vm_match_buffer(dfa[q]->vm, (const char *) b[q].base + b[q].off[i], (size_t) (b[q].off[i + 1] - b[q].off[i]))) {
				b[q].accept_bitmap[i / 64] |= (uint64_t) 1 << (i % 64);
			}
		}
	}
	return 0;
}

unsigned fsm_hip_multi_last_launches(void) { return 1; }

int
fsm_hip_match_file(const struct fsm_hip_dfa *d, FILE *f)
{
	n_single++;
	return fsm_vm_match_file(d->vm, f);
}

int
fsm_hip_exec(const struct fsm_hip_dfa *d, int (*fsm_getc)(void *opaque), void *opaque, unsigned *end, struct fsm_capture *captures)
{
	n_single++;
	return fsm_exec(d->fsm, fsm_getc, opaque, end, captures);
}

/* the stand-in has no eager front: exec_via_hip.c then takes the plain path */
size_t fsm_hip_eager_id_count(const struct fsm_hip_dfa *d) { (void) d; return 0; }
size_t fsm_hip_eager_words(const struct fsm_hip_dfa *d) { (void) d; return 1; }
uint32_t fsm_hip_eager_id(const struct fsm_hip_dfa *d, unsigned bit) { (void) d; (void) bit; return 0xFFFFFFFFu; }
int
fsm_hip_exec_batch_eager(const struct fsm_hip_dfa *d, const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *eager_out)
{
	(void) d; (void) base; (void) stride; (void) len; (void) n; (void) end_out; (void) eager_out;
	return -1;
}
