/*
 * shim.c -- libfsm-facing layer of libfsm_hip.so (plain C, like the host code
 * it sits next to).
 *
 * Reads a `const struct fsm *` ONLY through libfsm's public API, resolved with
 * dlsym(RTLD_DEFAULT, ...) on first use so that this library neither links
 * libfsm nor needs its headers at build time.  The prototypes below restate
 * the public declarations they bind (reference file:line in each comment).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fsm_hip.h"
#include "flat.h"

struct api {
	int resolved;
	/* include/fsm/fsm.h:388 */
	unsigned int (*countstates)(const struct fsm *);
	/* include/fsm/fsm.h:382 */
	int (*getstart)(const struct fsm *, fsm_state_t *);
	/* include/fsm/pred.h:21, :24 */
	int (*isend)(const struct fsm *, fsm_state_t);
	int (*isdfa)(const struct fsm *, fsm_state_t);
	/* include/fsm/walk.h:35 */
	int (*all)(const struct fsm *, int (*)(const struct fsm *, fsm_state_t));
	/* include/fsm/walk.h:75-78 */
	int (*walk_edges)(const struct fsm *, void *,
		int (*)(const struct fsm *, fsm_state_t, fsm_state_t, char, void *),
		int (*)(const struct fsm *, fsm_state_t, fsm_state_t, void *));
	/* include/fsm/capture.h:28 */
	unsigned (*countcaptures)(const struct fsm *);
	/* include/fsm/fsm.h:223-228 */
	int (*endid_get)(const struct fsm *, fsm_state_t, size_t, fsm_end_id_t *);
	size_t (*endid_count)(const struct fsm *, fsm_state_t);
	/* include/fsm/fsm.h:327 */
	size_t (*eager_output_count)(const struct fsm *, fsm_state_t);
	/* include/fsm/fsm.h:343-345 */
	int (*eager_output_get)(const struct fsm *, fsm_state_t, size_t, unsigned int *);
};

static struct api A;
static pthread_once_t api_once = PTHREAD_ONCE_INIT;

static void
resolve_once(void)
{
#define SYM(field, name) do { \
		*(void **)(&A.field) = dlsym(RTLD_DEFAULT, name); \
		if (A.field == NULL) { ok = 0; } \
	} while (0)
	int ok = 1;
	SYM(countstates, "fsm_countstates");
	SYM(getstart, "fsm_getstart");
	SYM(isend, "fsm_isend");
	SYM(isdfa, "fsm_isdfa");
	SYM(all, "fsm_all");
	SYM(walk_edges, "fsm_walk_edges");
	SYM(countcaptures, "fsm_countcaptures");
	SYM(endid_get, "fsm_endid_get");
	SYM(endid_count, "fsm_endid_count");
	SYM(eager_output_count, "fsm_eager_output_count");
	SYM(eager_output_get, "fsm_eager_output_get");
#undef SYM
	A.resolved = ok ? 1 : -1;
}

/* first use from any thread resolves the table exactly once */
static int
resolve(void)
{
	(void) pthread_once(&api_once, resolve_once);
	return A.resolved > 0;
}

/* ---- flatten ------------------------------------------------------- */

struct walk_env {
	uint32_t *next;   /* [nstates][256], 0xFFFFFFFF = no edge */
	uint32_t nstates;
	int nondet;
};

static int
lit_cb(const struct fsm *fsm, fsm_state_t from, fsm_state_t to, char c, void *opaque)
{
	struct walk_env *env = opaque;
	uint32_t *slot;
	(void) fsm;
	slot = &env->next[(size_t) from * 256 + (unsigned char) c];
	if (*slot != 0xFFFFFFFFu && *slot != to) {
		env->nondet = 1;
		return 0;
	}
	*slot = to;
	return 1;
}

static int
eps_cb(const struct fsm *fsm, fsm_state_t from, fsm_state_t to, void *opaque)
{
	struct walk_env *env = opaque;
	(void) fsm; (void) from; (void) to;
	env->nondet = 1;
	return 0;
}

void
fsm_hip_desc_free(struct fsm_hip_dfa_desc *desc)
{
	struct flat *f = (struct flat *) desc;
	if (f == NULL) {
		return;
	}
	free(f->edge_off);
	free(f->ranges);
	free(f->is_end);
	free(f->endid_off);
	free(f->endids);
	free(f->eager_off);
	free(f->eager_ids);
	free(f);
}

struct fsm_hip_dfa_desc *
fsm_hip_flatten(const struct fsm *fsm)
{
	struct flat *f = NULL;
	struct walk_env env;
	fsm_state_t start;
	uint32_t n, s;
	size_t nr, cap, nid;

	memset(&env, 0, sizeof env);
	if (fsm == NULL) {
		errno = EINVAL;
		return NULL;
	}
	if (!resolve()) {
		errno = ENOSYS;
		return NULL;
	}
	/* what fsm_exec checks per call (src/libfsm/exec.c:106-114), once */
	if (!A.all(fsm, A.isdfa) || !A.getstart(fsm, &start)) {
		errno = EINVAL;
		return NULL;
	}
	/* captures are per-byte host callbacks with positions (exec.c:41-44): not
	 * representable in a table walk.  Eager outputs (exec.c:126-144) are carried:
	 * flattened below, delivered by fsm_hip_exec_batch_eager*(). */
	if (A.countcaptures(fsm) > 0) {
		errno = ENOTSUP;
		return NULL;
	}
	n = A.countstates(fsm);
	if (n == 0) {
		errno = EINVAL;
		return NULL;
	}

	f = calloc(1, sizeof *f);
	env.next = malloc((size_t) n * 256 * sizeof *env.next);
	if (f == NULL || env.next == NULL) {
		goto oom;
	}
	memset(env.next, 0xFF, (size_t) n * 256 * sizeof *env.next);
	env.nstates = n;
	if (!A.walk_edges(fsm, &env, lit_cb, eps_cb) || env.nondet) {
		free(env.next);
		free(f);
		errno = EINVAL;
		return NULL;
	}

	f->edge_off = malloc(((size_t) n + 1) * sizeof *f->edge_off);
	f->is_end = malloc(n);
	f->endid_off = malloc(((size_t) n + 1) * sizeof *f->endid_off);
	cap = 1024;
	f->ranges = malloc(cap * sizeof *f->ranges);
	if (f->edge_off == NULL || f->is_end == NULL || f->endid_off == NULL || f->ranges == NULL) {
		goto oom;
	}
	nr = 0;
	for (s = 0; s < n; s++) {
		const uint32_t *row = &env.next[(size_t) s * 256];
		unsigned c = 0;
		f->edge_off[s] = (uint32_t) nr;
		f->is_end[s] = A.isend(fsm, s) ? 1 : 0;
		while (c < 256) {
			unsigned lo;
			if (row[c] == 0xFFFFFFFFu) {
				c++;
				continue;
			}
			lo = c;
			while (c + 1 < 256 && row[c + 1] == row[lo]) {
				c++;
			}
			if (nr == cap) {
				struct fsm_hip_range *t;
				cap *= 2;
				t = realloc(f->ranges, cap * sizeof *t);
				if (t == NULL) {
					goto oom;
				}
				f->ranges = t;
			}
			f->ranges[nr].lo = (uint8_t) lo;
			f->ranges[nr].hi = (uint8_t) c;
			f->ranges[nr].reserved = 0;
			f->ranges[nr].to = row[lo];
			nr++;
			c++;
		}
	}
	f->edge_off[n] = (uint32_t) nr;
	free(env.next);
	env.next = NULL;

	/* end-ids: sorted unique per end state, as fsm_endid_get returns them
	 * (src/libfsm/endids.c:686-755) */
	nid = 0;
	for (s = 0; s < n; s++) {
		f->endid_off[s] = (uint32_t) nid;
		if (f->is_end[s]) {
			nid += A.endid_count(fsm, s);
		}
	}
	f->endid_off[n] = (uint32_t) nid;
	f->endids = malloc((nid ? nid : 1) * sizeof *f->endids);
	if (f->endids == NULL) {
		goto oom;
	}
	for (s = 0; s < n; s++) {
		size_t cnt = f->endid_off[s + 1] - f->endid_off[s];
		if (cnt > 0 && !A.endid_get(fsm, s, cnt, f->endids + f->endid_off[s])) {
			fsm_hip_desc_free(&f->d);
			errno = EINVAL;
			return NULL;
		}
	}

	/* eager outputs: sorted unique per state, as fsm_eager_output_get returns them
	 * (src/libfsm/eager_output.c:240ff; on any state, end or not) */
	nid = 0;
	for (s = 0; s < n; s++) {
		nid += A.eager_output_count(fsm, s);
	}
	if (nid > 0) {
		size_t k = 0;
		f->eager_off = malloc(((size_t) n + 1) * sizeof *f->eager_off);
		f->eager_ids = malloc(nid * sizeof *f->eager_ids);
		if (f->eager_off == NULL || f->eager_ids == NULL) {
			goto oom;
		}
		for (s = 0; s < n; s++) {
			size_t cnt = A.eager_output_count(fsm, s);
			f->eager_off[s] = (uint32_t) k;
			if (cnt > 0 && !A.eager_output_get(fsm, s, cnt, f->eager_ids + k)) {
				fsm_hip_desc_free(&f->d);
				errno = EINVAL;
				return NULL;
			}
			k += cnt;
		}
		f->eager_off[n] = (uint32_t) k;
	}

	f->d.nstates = n;
	f->d.start = start;
	f->d.edge_off = f->edge_off;
	f->d.ranges = f->ranges;
	f->d.is_end = f->is_end;
	f->d.endid_off = f->endid_off;
	f->d.endids = f->endids;
	f->d.eager_off = f->eager_off;
	f->d.eager_ids = f->eager_ids;
	return &f->d;

oom:
	free(env.next);
	if (f != NULL) {
		fsm_hip_desc_free(&f->d);
	}
	errno = ENOMEM;
	return NULL;
}

struct fsm_hip_dfa *
fsm_hip_compile(const struct fsm *fsm, unsigned flags)
{
	struct fsm_hip_dfa_desc *desc;
	struct fsm_hip_dfa *dfa;
	int e;

	desc = fsm_hip_flatten(fsm);
	if (desc == NULL) {
		return NULL;
	}
	dfa = fsm_hip_dfa_create(desc, flags);
	e = errno;
	fsm_hip_desc_free(desc);
	errno = e;
	return dfa;
}

struct fsm_hip_node *
fsm_hip_node_compile(const struct fsm *fsm, unsigned flags, const int *devices, int ndev)
{
	struct fsm_hip_dfa_desc *desc;
	struct fsm_hip_node *node;
	int e;

	desc = fsm_hip_flatten(fsm);
	if (desc == NULL) {
		return NULL;
	}
	node = fsm_hip_node_create(desc, flags, devices, ndev);
	e = errno;
	fsm_hip_desc_free(desc);
	errno = e;
	return node;
}

/* ---- single-input fronts (batch of one on the GPU) ------------------ */

int
fsm_hip_match_buffer(const struct fsm_hip_dfa *dfa, const char *buf, size_t n)
{
	uint32_t end = FSM_HIP_NO_MATCH;
	uint64_t off[2];
	if (n >= ((size_t) 1 << 20)) {
		/* worth the whole device: pieces walked at once from guessed states, corrected until they stand (file.hip) */
		return fsm_hip_match_buffer_big(dfa, buf, n, NULL);
	}
	off[0] = 0;
	off[1] = n;
	if (fsm_hip_exec_batch_offsets(dfa, (const unsigned char *) buf, off, 1, &end, NULL) != 0) {
		return -1;
	}
	return end != FSM_HIP_NO_MATCH;
}

int
fsm_hip_exec(const struct fsm_hip_dfa *dfa,
	int (*fsm_getc)(void *opaque), void *opaque,
	fsm_state_t *end, struct fsm_capture *captures)
{
	unsigned char *buf = NULL, *t;
	size_t n = 0, cap = 0;
	uint32_t e = FSM_HIP_NO_MATCH;
	uint64_t off[2];
	int c, r;

	if (dfa == NULL || fsm_getc == NULL || end == NULL || captures != NULL) {
		errno = EINVAL;
		return -1;
	}
	while (c = fsm_getc(opaque), c != EOF) {
		if (n == cap) {
			cap = cap ? cap * 2 : 4096;
			t = realloc(buf, cap);
			if (t == NULL) {
				free(buf);
				errno = ENOMEM;
				return -1;
			}
			buf = t;
		}
		buf[n++] = (unsigned char) c;
	}
	off[0] = 0;
	off[1] = n;
	r = fsm_hip_exec_batch_offsets(dfa, buf, off, 1, &e, NULL);
	free(buf);
	if (r != 0) {
		return -1;
	}
	if (e == FSM_HIP_NO_MATCH) {
		return 0; /* *end untouched, as exec.c:133-138 / :153-155 */
	}
	*end = e;
	return 1;
}

/* fsm_hip_match_file(): file.hip (the whole device on one input) */

/* ---- on-disk form of a flat DFA description -------------------------- */
/*
 * Little-endian, self-describing, no pointers:
 *   char     magic[8] = "FSMHIP01"
 *   uint32   nstates, start, nranges, nendids
 *   uint32   edge_off[nstates + 1]
 *   range    ranges[nranges]            (lo u8, hi u8, reserved u16, to u32)
 *   uint8    is_end[nstates], zero padded to a multiple of 4
 *   uint32   endid_off[nstates + 1]
 *   uint32   endids[nendids]
 * "FSMHIP02" appends the eager outputs:
 *   uint32   neager
 *   uint32   eager_off[nstates + 1]
 *   uint32   eager_ids[neager]
 * It serialises what fsm_hip_flatten() extracts from a struct fsm, i.e. the
 * role the reference's DFAVM save/load has for its bytecode
 * (fsm_dfavm_save/load, src/libfsm/vm.c:39-71, src/libfsm/vm/v1.c:19-82).
 */
static const char desc_magic[8] = { 'F', 'S', 'M', 'H', 'I', 'P', '0', '1' };
static const char desc_magic2[8] = { 'F', 'S', 'M', 'H', 'I', 'P', '0', '2' };

int
fsm_hip_desc_write(const struct fsm_hip_dfa_desc *d, FILE *f)
{
	uint32_t hdr[4], n, nr, nid, pad = 0;
	static const uint32_t zero_off[1] = { 0 };

	if (d == NULL || f == NULL || d->nstates == 0 || d->edge_off == NULL || d->is_end == NULL) {
		errno = EINVAL;
		return -1;
	}
	n = d->nstates;
	nr = d->edge_off[n];
	nid = d->endid_off != NULL ? d->endid_off[n] : 0;
	hdr[0] = n;
	hdr[1] = d->start;
	hdr[2] = nr;
	hdr[3] = nid;
	if (fwrite(d->eager_off != NULL ? desc_magic2 : desc_magic, 1, 8, f) != 8 || fwrite(hdr, 4, 4, f) != 4 ||
	    fwrite(d->edge_off, 4, (size_t) n + 1, f) != (size_t) n + 1 ||
	    (nr > 0 && fwrite(d->ranges, sizeof *d->ranges, nr, f) != nr) ||
	    fwrite(d->is_end, 1, n, f) != n ||
	    ((n & 3u) != 0 && fwrite(&pad, 1, 4 - (n & 3u), f) != 4 - (n & 3u))) {
		return -1;
	}
	if (d->endid_off != NULL) {
		if (fwrite(d->endid_off, 4, (size_t) n + 1, f) != (size_t) n + 1 ||
		    (nid > 0 && fwrite(d->endids, 4, nid, f) != nid)) {
			return -1;
		}
	} else {
		uint32_t i;
		for (i = 0; i <= n; i++) {
			if (fwrite(zero_off, 4, 1, f) != 1) {
				return -1;
			}
		}
	}
	if (d->eager_off != NULL) {
		uint32_t ne = d->eager_off[n];
		if (fwrite(&ne, 4, 1, f) != 1 || fwrite(d->eager_off, 4, (size_t) n + 1, f) != (size_t) n + 1 ||
		    (ne > 0 && fwrite(d->eager_ids, 4, ne, f) != ne)) {
			return -1;
		}
	}
	return 0;
}

/* Read `count` elements of `elem` bytes into a fresh buffer that grows with the bytes actually
 * present, so that a corrupt header cannot make the reader allocate more than the file holds
 * (plus one 4 MiB step).  NULL + errno (EINVAL: truncated, ENOMEM). */
static void *
read_array(FILE *f, size_t count, size_t elem)
{
	const size_t step = (size_t) 4 << 20;
	size_t want, have = 0, cap;
	unsigned char *p, *t;

	if (elem != 0 && count > (size_t) -1 / elem) {
		errno = EINVAL;
		return NULL;
	}
	want = count * elem;
	cap = want < step ? (want ? want : 1) : step;
	p = malloc(cap);
	if (p == NULL) {
		errno = ENOMEM;
		return NULL;
	}
	while (have < want) {
		size_t n = want - have < cap - have ? want - have : cap - have;
		if (n == 0) {
			cap = cap * 2 < want ? cap * 2 : want;
			t = realloc(p, cap);
			if (t == NULL) {
				free(p);
				errno = ENOMEM;
				return NULL;
			}
			p = t;
			continue;
		}
		if (fread(p + have, 1, n, f) != n) {
			free(p);
			errno = EINVAL; /* truncated */
			return NULL;
		}
		have += n;
	}
	return p;
}

struct fsm_hip_dfa_desc *
fsm_hip_desc_read(FILE *f)
{
	struct flat *fl = NULL;
	char magic[8];
	uint32_t hdr[4], n, nr, nid, s, ne = 0;
	unsigned char pad[4];
	int v2;

	if (f == NULL) {
		errno = EINVAL;
		return NULL;
	}
	if (fread(magic, 1, 8, f) != 8 || (memcmp(magic, desc_magic, 8) != 0 && memcmp(magic, desc_magic2, 8) != 0) ||
	    fread(hdr, 4, 4, f) != 4) {
		errno = EINVAL;
		return NULL;
	}
	v2 = memcmp(magic, desc_magic2, 8) == 0;
	n = hdr[0];
	nr = hdr[2];
	nid = hdr[3];
	/* a DFA state has at most 256 byte ranges */
	if (n == 0 || n >= 0x00FFFFFFu || hdr[1] >= n || (uint64_t) nr > (uint64_t) n * 256u) {
		errno = EINVAL;
		return NULL;
	}
	fl = calloc(1, sizeof *fl);
	if (fl == NULL) {
		errno = ENOMEM;
		return NULL;
	}
	if ((fl->edge_off = read_array(f, (size_t) n + 1, 4)) == NULL ||
	    (fl->ranges = read_array(f, nr, sizeof *fl->ranges)) == NULL ||
	    (fl->is_end = read_array(f, n, 1)) == NULL ||
	    ((n & 3u) != 0 && fread(pad, 1, 4 - (n & 3u), f) != 4 - (n & 3u) && (errno = EINVAL, 1)) ||
	    (fl->endid_off = read_array(f, (size_t) n + 1, 4)) == NULL ||
	    (fl->endids = read_array(f, nid, 4)) == NULL) {
		int e = errno;
		fsm_hip_desc_free(&fl->d);
		errno = e;
		return NULL;
	}
	/* structural checks: offsets monotone and consistent with the header; ranges of one state
	 * ascending and disjoint (so edge_off[] cannot claim more ranges than were supplied) */
	if (fl->edge_off[0] != 0 || fl->edge_off[n] != nr || fl->endid_off[0] != 0 || fl->endid_off[n] != nid) {
		goto bad;
	}
	for (s = 0; s < n; s++) {
		if (fl->edge_off[s + 1] < fl->edge_off[s] || fl->edge_off[s + 1] > nr ||
		    fl->edge_off[s + 1] - fl->edge_off[s] > 256u ||
		    fl->endid_off[s + 1] < fl->endid_off[s] || fl->endid_off[s + 1] > nid) {
			goto bad;
		}
	}
	for (s = 0; s < nr; s++) {
		if (fl->ranges[s].lo > fl->ranges[s].hi || fl->ranges[s].to >= n) {
			goto bad;
		}
	}
	fl->d.nstates = n;
	fl->d.start = hdr[1];
	fl->d.edge_off = fl->edge_off;
	fl->d.ranges = fl->ranges;
	fl->d.is_end = fl->is_end;
	fl->d.endid_off = fl->endid_off;
	fl->d.endids = fl->endids;
	if (v2) {
		if (fread(&ne, 4, 1, f) != 1) {
			goto bad;
		}
		if ((fl->eager_off = read_array(f, (size_t) n + 1, 4)) == NULL ||
		    (fl->eager_ids = read_array(f, ne, 4)) == NULL) {
			int e = errno;
			fsm_hip_desc_free(&fl->d);
			errno = e;
			return NULL;
		}
		if (fl->eager_off[0] != 0 || fl->eager_off[n] != ne) {
			goto bad;
		}
		for (s = 0; s < n; s++) {
			if (fl->eager_off[s + 1] < fl->eager_off[s] || fl->eager_off[s + 1] > ne) {
				goto bad;
			}
		}
		fl->d.eager_off = fl->eager_off;
		fl->d.eager_ids = fl->eager_ids;
	}
	return &fl->d;
bad:
	fsm_hip_desc_free(&fl->d);
	errno = EINVAL;
	return NULL;
}

/* ---- printer: struct fsm * -> on-disk table -------------------------- */

/* The shape of one more fsm_print() language (src/libfsm/print.c:242-416 dispatches
 * FSM_PRINT_C, _VMC, ... to printers taking (FILE *, fsm)): writes the executable table of `fsm`
 * in the FSMHIP on-disk form, ready for fsm_hip_desc_read() + fsm_hip_dfa_create() in a process
 * that has no libfsm at all.  0 on success, -1 + errno (EINVAL: not a DFA / no start). */
int
fsm_hip_print(FILE *f, const struct fsm *fsm)
{
	struct fsm_hip_dfa_desc *desc;
	int r, e;

	if (f == NULL) {
		errno = EINVAL;
		return -1;
	}
	desc = fsm_hip_flatten(fsm);
	if (desc == NULL) {
		return -1;
	}
	r = fsm_hip_desc_write(desc, f);
	e = errno;
	fsm_hip_desc_free(desc);
	errno = e;
	return r;
}
