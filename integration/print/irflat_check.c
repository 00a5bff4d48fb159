/*
 * irflat_check.c -- CPU check of the IR consumer (print_hip_ir.c): for every .fsm file named on the command line, the
 * description flattened from libfsm's codegen IR (make_ir -> fsm_hip_desc_from_ir: groups of ranges + mode / error
 * ranges -> rows, the recipe of src/libfsm/vm/ir.c:649-750) must describe the same automaton as the one the shim
 * flattens through the public API (fsm_hip_flatten: fsm_walk_edges): same states, start, end states, next state on all
 * 256 bytes of every state, end-id sets.  Prints one line per file and a summary; exit status 1 on any difference.
 * Built by integration/print/build.sh against the reference tree (test infrastructure: tests/test_print_patch.py).
 *   irflat_check [-d] file.fsm...      -d: determinise + minimise first (for NFA fixtures)
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fsm/fsm.h>
#include <fsm/bool.h>
#include <fsm/options.h>
#include <fsm/pred.h>
#include <fsm/walk.h>
#include <fsm/print.h>
#include <fsm/parser.h>

#include "libfsm/print/ir.h"

#include <fsm_hip.h>

struct fsm_hip_dfa_desc *fsm_hip_desc_from_ir(const struct ir *ir);
void fsm_hip_desc_from_ir_free(struct fsm_hip_dfa_desc *d);

static void
expand(const struct fsm_hip_dfa_desc *d, uint32_t s, uint32_t row[256])
{
	uint32_t k;
	int c;
	for (c = 0; c < 256; c++) {
		row[c] = 0xFFFFFFFFu;
	}
	for (k = d->edge_off[s]; k < d->edge_off[s + 1]; k++) {
		for (c = d->ranges[k].lo; c <= d->ranges[k].hi; c++) {
			row[c] = d->ranges[k].to;
		}
	}
}

static int
same(const struct fsm_hip_dfa_desc *a, const struct fsm_hip_dfa_desc *b, const char **why)
{
	uint32_t s;
	if (a->nstates != b->nstates) { *why = "state count"; return 0; }
	if (a->start != b->start) { *why = "start state"; return 0; }
	for (s = 0; s < a->nstates; s++) {
		uint32_t ra[256], rb[256], na, nb;
		if ((a->is_end[s] != 0) != (b->is_end[s] != 0)) { *why = "end states"; return 0; }
		expand(a, s, ra);
		expand(b, s, rb);
		if (memcmp(ra, rb, sizeof ra) != 0) { *why = "transitions"; return 0; }
		na = a->endid_off ? a->endid_off[s + 1] - a->endid_off[s] : 0;
		nb = b->endid_off ? b->endid_off[s + 1] - b->endid_off[s] : 0;
		if (na != nb || (na != 0 && memcmp(a->endids + a->endid_off[s], b->endids + b->endid_off[s], na * sizeof(uint32_t)) != 0)) {
			*why = "end-ids";
			return 0;
		}
	}
	return 1;
}

int
main(int argc, char **argv)
{
	static const struct fsm_options zero;
	struct fsm_options opt = zero;
	int i, det = 0, nok = 0, nbad = 0, nskip = 0;

	opt.comments = 0;
	opt.anonymous_states = 1;
	opt.consolidate_edges = 1;
	opt.io = FSM_IO_GETC;
	for (i = 1; i < argc; i++) {
		FILE *f;
		struct fsm *fsm;
		struct ir *ir;
		struct fsm_hip_dfa_desc *d1, *d2;
		const char *why = "";

		if (strcmp(argv[i], "-d") == 0) {
			det = 1;
			continue;
		}
		f = fopen(argv[i], "r");
		if (f == NULL) {
			perror(argv[i]);
			nbad++;
			continue;
		}
		fsm = fsm_parse(f, NULL);
		fclose(f);
		if (fsm == NULL) {
			printf("%s: skipped (fsm_parse)\n", argv[i]);
			nskip++;
			continue;
		}
		if (det && (!fsm_determinise(fsm) || !fsm_minimise(fsm))) {
			printf("%s: skipped (determinise)\n", argv[i]);
			fsm_free(fsm);
			nskip++;
			continue;
		}
		if (!fsm_all(fsm, fsm_isdfa) || fsm_countstates(fsm) == 0) {
			printf("%s: skipped (not a DFA)\n", argv[i]);
			fsm_free(fsm);
			nskip++;
			continue;
		}
		{
			fsm_state_t st;
			if (!fsm_getstart(fsm, &st)) {
				printf("%s: skipped (no start state)\n", argv[i]);
				fsm_free(fsm);
				nskip++;
				continue;
			}
		}
		d1 = fsm_hip_flatten(fsm);
		ir = make_ir(fsm, &opt);
		d2 = ir != NULL ? fsm_hip_desc_from_ir(ir) : NULL;
		if (d1 == NULL || d2 == NULL) {
			printf("%s: FAILED to flatten (%s: %s)\n", argv[i], d1 == NULL ? "fsm_walk_edges" : "ir", strerror(errno));
			nbad++;
		} else if (!same(d1, d2, &why)) {
			printf("%s: DIFFERENT (%s)\n", argv[i], why);
			nbad++;
		} else {
			printf("%s: ok (%u states)\n", argv[i], d1->nstates);
			nok++;
		}
		if (d1 != NULL) fsm_hip_desc_free(d1);
		fsm_hip_desc_from_ir_free(d2);
		if (ir != NULL) free_ir(fsm, ir);
		fsm_free(fsm);
	}
	printf("# %d ok, %d different or failed, %d skipped\n", nok, nbad, nskip);
	return nbad != 0;
}
