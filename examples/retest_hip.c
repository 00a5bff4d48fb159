/*
 * examples/retest_hip.c -- retest(1)'s job with the batched front: run .tst files through libre +
 * libfsm_hip.so, ONE fsm_hip_exec_batch_offsets() call per regex block instead of one
 * fsm_runner_run() per test line (src/retest/main.c:1083-1147, src/retest/runner.c:494-531).
 *
 *   retest_hip [-v] file.tst ...
 *
 * .tst format as read by the reference's process_test_file (src/retest/main.c:738-1177), restated:
 *   blank line        ends a regex block          # ...   comment
 *   R [dialect]       dialect for what follows (default pcre: like literal glob native sql pcre)
 *   M flags           regex flags for the next block (i t m r s z a x, 0 clears)
 *   O +e|-e|=e|&      escape processing of the regex line on/off; "O &" restores it at each blank line
 *   first other line  the regex; a leading '~' is dropped
 *   +text / -text     an input that must / must not match (always escape-processed)
 * Escapes (main.c:299-441): \a \b \e \f \n \r \t \v \" \\  \ooo  \xHH  \x{HH}.
 *
 * The host side is the reference's own: re_comp, fsm_determinise, fsm_minimise from the libfsm/libre
 * this program is linked with (prototypes restated from include/re/re.h:13-37,137-140 and
 * include/fsm/fsm.h, because those headers are not part of this repository).  The struct fsm is freed
 * right after fsm_hip_compile(), like retest frees it after fsm_runner_initialize (main.c:1056-1058).
 * Exit status: 0 if every test passed, 1 otherwise, 2 on usage / IO / compile errors.
 */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fsm_hip.h"

enum re_dialect { RE_LIKE, RE_LITERAL, RE_GLOB, RE_NATIVE, RE_SQL, RE_PCRE };
struct re_err { int e; char buf[256]; };
struct fsm *re_comp(enum re_dialect, int (*)(void *), void *, const void *alloc, int flags, struct re_err *);
int fsm_determinise(struct fsm *);
int fsm_minimise(struct fsm *);
void fsm_free(struct fsm *);

struct bytes { unsigned char *p; size_t n, cap; };
struct tcase { size_t line; int expect; size_t off, len; };

struct block {
	int active;
	size_t line;
	enum re_dialect dialect;
	int flags;
	struct bytes regex, text;     /* text: all case inputs back to back */
	struct tcase *cases;
	size_t ncases, capcases;
};

struct cursor { const unsigned char *p, *e; };

static int
cursor_getc(void *opaque)
{
	struct cursor *c = opaque;
	return c->p == c->e ? EOF : *c->p++;
}

static void
push(struct bytes *b, unsigned c)
{
	if (b->n == b->cap) {
		b->cap = b->cap ? 2 * b->cap : 64;
		b->p = realloc(b->p, b->cap);
		if (b->p == NULL) { perror("realloc"); exit(2); }
	}
	b->p[b->n++] = (unsigned char) c;
}

static int
hexval(int c)
{
	if (c >= '0' && c <= '9') return c - '0';
	if (c >= 'a' && c <= 'f') return c - 'a' + 10;
	if (c >= 'A' && c <= 'F') return c - 'A' + 10;
	return -1;
}

/* append s[0..n) to out, escape-processed or verbatim; 0 on a malformed escape */
static int
append_text(struct bytes *out, const char *s, size_t n, int escapes)
{
	size_t i = 0;
	while (i < n) {
		unsigned c = (unsigned char) s[i++];
		if (!escapes || c != '\\') { push(out, c); continue; }
		if (i == n) return 0;
		c = (unsigned char) s[i++];
		switch (c) {
		case 'a': push(out, 7); break;
		case 'b': push(out, 8); break;
		case 'e': push(out, 27); break;
		case 'f': push(out, 12); break;
		case 'n': push(out, 10); break;
		case 'r': push(out, 13); break;
		case 't': push(out, 9); break;
		case 'v': push(out, 11); break;
		case '"': case '\\': push(out, c); break;
		case 'x': {
			unsigned v = 0, nd = 0;
			int curly = i < n && s[i] == '{';
			if (curly) i++;
			while (i < n && nd < 2 && hexval((unsigned char) s[i]) >= 0) { v = v * 16 + (unsigned) hexval((unsigned char) s[i++]); nd++; }
			if (curly) { if (i == n || s[i] != '}') return 0; i++; }
			push(out, v & 0xff);
			break;
		}
		default:
			if (c >= '0' && c <= '7') {
				unsigned v = c - '0', nd = 1;
				while (i < n && nd < 3 && s[i] >= '0' && s[i] <= '7') { v = v * 8 + (unsigned) (s[i++] - '0'); nd++; }
				push(out, v & 0xff);
				break;
			}
			return 0;
		}
	}
	return 1;
}

static int
dialect_of(const char *name, enum re_dialect *out)
{
	static const char *names[] = { "like", "literal", "glob", "native", "sql", "pcre" };
	for (int i = 0; i < 6; i++) if (strcmp(name, names[i]) == 0) { *out = (enum re_dialect) i; return 1; }
	return 0;
}

static int
flag_of(int ch)
{
	/* enum re_flags, include/re/re.h:22-37 */
	switch (ch) {
	case 'i': return 1; case 't': return 2; case 'm': return 4; case 'r': return 8;
	case 's': return 16; case 'z': return 32; case 'a': return 64; case 'x': return 128;
	default: return 0;
	}
}

struct totals { size_t blocks, tests, failed, errors; };

static void
run_block(const char *file, struct block *b, struct totals *t, int verbose)
{
	struct cursor cur = { b->regex.p, b->regex.p + b->regex.n };
	struct re_err err;
	struct fsm *fsm;
	struct fsm_hip_dfa *dfa;
	uint64_t *off;
	uint32_t *end;

	if (!b->active) return;
	t->blocks++;
	memset(&err, 0, sizeof err);
	fsm = re_comp(b->dialect, cursor_getc, &cur, NULL, b->flags, &err);
	if (fsm == NULL || !fsm_determinise(fsm) || !fsm_minimise(fsm)) {
		fprintf(stderr, "[%s:%zu] regex does not compile\n", file, b->line);
		if (fsm != NULL) fsm_free(fsm);
		t->errors++;
		goto done;
	}
	dfa = fsm_hip_compile(fsm, 0);
	fsm_free(fsm);
	if (dfa == NULL) {
		fprintf(stderr, "[%s:%zu] fsm_hip_compile: %s\n", file, b->line, strerror(errno));
		t->errors++;
		goto done;
	}
	off = malloc((b->ncases + 1) * sizeof *off);
	end = malloc((b->ncases ? b->ncases : 1) * sizeof *end);
	if (off == NULL || end == NULL) { perror("malloc"); exit(2); }
	for (size_t i = 0; i < b->ncases; i++) off[i] = b->cases[i].off;
	off[b->ncases] = b->text.n;
	/* the whole block in one launch */
	if (fsm_hip_exec_batch_offsets(dfa, b->text.p, off, b->ncases, end, NULL) != 0) {
		fprintf(stderr, "[%s:%zu] fsm_hip_exec_batch_offsets: %s\n", file, b->line, strerror(errno));
		t->errors++;
	} else {
		for (size_t i = 0; i < b->ncases; i++) {
			const int matched = end[i] != FSM_HIP_NO_MATCH;
			t->tests++;
			if (matched != b->cases[i].expect) {
				t->failed++;
				printf("[%s:%zu] FAIL regexp /%.*s/ expected to %smatch the input of line %zu\n", file, b->line,
					(int) b->regex.n, (const char *) b->regex.p, b->cases[i].expect ? "" : "not ", b->cases[i].line);
			} else if (verbose) {
				printf("[%s:%zu] ok line %zu\n", file, b->line, b->cases[i].line);
			}
		}
	}
	free(off);
	free(end);
	fsm_hip_dfa_free(dfa);
done:
	b->active = 0;
	b->regex.n = b->text.n = 0;
	b->ncases = 0;
}

static int
process_file(const char *path, struct totals *t, int verbose)
{
	FILE *f = fopen(path, "rb");
	char *line = NULL;
	size_t cap = 0, lineno = 0;
	ssize_t len;
	struct block b;
	enum re_dialect dialect = RE_PCRE;
	int flags = 0, opt_e = 0, restore = 0, saved_e = 0;

	if (f == NULL) { perror(path); return 0; }
	memset(&b, 0, sizeof b);
	while ((len = getline(&line, &cap, f)) != -1) {
		size_t n = (size_t) len;
		lineno++;
		if (n > 0 && line[n - 1] == '\n') n--;
		if (n == 0) {                         /* blank: the block is complete */
			run_block(path, &b, t, verbose);
			flags = 0;
			if (restore) opt_e = saved_e;
			continue;
		}
		if (line[0] == '#') continue;
		if (line[0] == 'R' && (n == 1 || line[1] == ' ')) {
			line[n] = '\0';
			if (n == 1) dialect = RE_PCRE;
			else if (!dialect_of(line + 2, &dialect)) { fprintf(stderr, "[%s:%zu] unknown dialect\n", path, lineno); t->errors++; }
			continue;
		}
		if (n >= 2 && line[0] == 'O' && line[1] == ' ') {
			int arg = 0;
			if (n >= 3 && line[2] == '&') { restore = 1; saved_e = opt_e; continue; }
			for (size_t i = 3; i < n; i++) if (line[i] == 'e') arg = 1;
			if (n >= 3 && line[2] == '=') opt_e = arg;
			else if (n >= 3 && line[2] == '+') opt_e = opt_e || arg;
			else if (n >= 3 && line[2] == '-') opt_e = opt_e && !arg;
			continue;
		}
		if (n >= 2 && line[0] == 'M' && line[1] == ' ') {
			for (size_t i = 2; i < n; i++) { if (line[i] == '0') flags = 0; else flags |= flag_of((unsigned char) line[i]); }
			continue;
		}
		{
			const char *s = line;
			if (s[0] == '~') { s++; n--; }
			if (!b.active) {
				b.active = 1;
				b.line = lineno;
				b.dialect = dialect;
				b.flags = flags;
				if (!append_text(&b.regex, s, n, opt_e)) { fprintf(stderr, "[%s:%zu] bad escape\n", path, lineno); t->errors++; }
			} else if (s[0] == '+' || s[0] == '-') {
				struct tcase c;
				c.line = lineno;
				c.expect = s[0] == '+';
				c.off = b.text.n;
				if (!append_text(&b.text, s + 1, n - 1, 1)) { fprintf(stderr, "[%s:%zu] bad escape\n", path, lineno); t->errors++; continue; }
				c.len = b.text.n - c.off;
				if (b.ncases == b.capcases) {
					b.capcases = b.capcases ? 2 * b.capcases : 16;
					b.cases = realloc(b.cases, b.capcases * sizeof *b.cases);
					if (b.cases == NULL) { perror("realloc"); exit(2); }
				}
				b.cases[b.ncases++] = c;
			} else {
				fprintf(stderr, "[%s:%zu] unrecognised line\n", path, lineno);
				t->errors++;
			}
		}
	}
	run_block(path, &b, t, verbose);
	free(line);
	free(b.regex.p);
	free(b.text.p);
	free(b.cases);
	fclose(f);
	return 1;
}

int
main(int argc, char **argv)
{
	struct totals t = { 0, 0, 0, 0 };
	int verbose = 0, i = 1;

	if (i < argc && strcmp(argv[i], "-v") == 0) { verbose = 1; i++; }
	if (i >= argc) { fprintf(stderr, "usage: retest_hip [-v] file.tst ...\n"); return 2; }
	for (; i < argc; i++) if (!process_file(argv[i], &t, verbose)) return 2;
	printf("%zu regexps, %zu tests, %zu failed, %zu errors\n", t.blocks, t.tests, t.failed, t.errors);
	if (t.errors) return 2;
	return t.failed ? 1 : 0;
}
