/*
 * tests/golden/record_exec.c -- fixture recorder (test infrastructure, not product).
 *
 * The reference's own C test programs (tests/endids/*.c, tests/re_strings/*.c, ...) are compiled
 * where they lie under /root/reference with `-Dfsm_exec=rec_fsm_exec` and linked with this file
 * and oracle/_ref/libfsm_ref.so.  Every fsm_exec() call those programs make (src/libfsm/exec.c:85)
 * is forwarded to the real fsm_exec and logged: the automaton in the product's on-disk table form
 * (fsm_hip_flatten + fsm_hip_desc_write), the whole input, the return code, the end state and
 * fsm_endid_get's answer for it.  make_golden.py turns the log into tests/golden/recorded/ *.npz.
 *
 * The stream is drained up front and replayed from memory so the log holds the complete input even
 * when fsm_exec stops at a failing byte (exec.c:133-138).
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fsm/fsm.h>

#include "fsm_hip.h"

#undef fsm_exec
int fsm_exec(const struct fsm *fsm, int (*fsm_getc)(void *opaque), void *opaque, fsm_state_t *end, struct fsm_capture *captures);

struct replay {
	const unsigned char *p, *e;
};

static int
replay_getc(void *opaque)
{
	struct replay *r = opaque;
	return r->p == r->e ? EOF : *r->p++;
}

static void
put32(FILE *f, uint32_t v)
{
	unsigned char b[4] = { v & 0xff, (v >> 8) & 0xff, (v >> 16) & 0xff, (v >> 24) & 0xff };
	fwrite(b, 1, 4, f);
}

int
rec_fsm_exec(const struct fsm *fsm, int (*fsm_getc)(void *opaque), void *opaque, fsm_state_t *end, struct fsm_capture *captures)
{
	unsigned char *buf = NULL;
	size_t n = 0, cap = 0;
	struct replay rp;
	fsm_state_t e = 0;
	int c, ret, saved;
	const char *out = getenv("REC_OUT");
	FILE *f;

	while ((c = fsm_getc(opaque)) != EOF) {
		if (n == cap) {
			cap = cap ? cap * 2 : 64;
			buf = realloc(buf, cap);
			if (buf == NULL) { abort(); }
		}
		buf[n++] = (unsigned char) c;
	}
	rp.p = buf;
	rp.e = buf + n;
	errno = 0;
	ret = fsm_exec(fsm, replay_getc, &rp, &e, captures);
	saved = errno;
	if (ret == 1 && end != NULL) {
		*end = e;
	}

	if (out != NULL && (f = fopen(out, "ab")) != NULL) {
		struct fsm_hip_dfa_desc *d;
		char *blob = NULL;
		size_t blen = 0;
		int ferr = 0;

		errno = 0;
		d = fsm_hip_flatten(fsm);
		if (d != NULL) {
			FILE *m = open_memstream(&blob, &blen);
			if (fsm_hip_desc_write(d, m) != 0) { ferr = errno ? errno : EIO; }
			fclose(m);
			fsm_hip_desc_free(d);
		} else {
			ferr = errno ? errno : EINVAL;
		}
		put32(f, 0x43455852u);                 /* "RXEC" */
		put32(f, ferr ? 0 : (uint32_t) blen);
		put32(f, (uint32_t) ferr);
		if (!ferr) { fwrite(blob, 1, blen, f); }
		free(blob);
		put32(f, (uint32_t) n);
		fwrite(buf, 1, n, f);
		put32(f, (uint32_t) ret);
		put32(f, ret == 1 ? e : 0xFFFFFFFFu);
		if (ret == 1) {
			size_t cnt = fsm_endid_count(fsm, e), i;
			fsm_end_id_t *ids = calloc(cnt ? cnt : 1, sizeof *ids);
			if (cnt > 0 && !fsm_endid_get(fsm, e, cnt, ids)) { abort(); }
			put32(f, (uint32_t) cnt);
			for (i = 0; i < cnt; i++) { put32(f, ids[i]); }
			free(ids);
		} else {
			put32(f, 0);
		}
		fclose(f);
	}
	free(buf);
	errno = saved;
	return ret;
}
