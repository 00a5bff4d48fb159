/*
 * print_hip_ir.c -- fsm_print_hip(): the FSM_PRINT_HIP printer as a consumer of libfsm's codegen IR.
 *
 * An ir_print_f (src/libfsm/print.h:64-67), entered from fsm_print()'s print_ir tier (src/libfsm/print.c:347-369) like
 * fsm_print_c / fsm_print_ir: fsm_print() has already run make_ir() (src/libfsm/print/ir.c:509-637), refused
 * ambiguous end-ids (opt->ambig), and will check ferror(f) and free the IR after us (the `done:` tail, print.c:401-414).
 *
 * The IR (src/libfsm/print/ir.h:23-108) gives per state a strategy, groups of inclusive byte ranges per destination,
 * a mode (the dominant destination) and error ranges; it is expanded to a 256-entry row exactly as the VM compiler's
 * dfa_table does it (src/libfsm/vm/ir.c:649-750): default = mode or "no edge", error ranges -> "no edge", groups
 * overwrite.  The row is re-compressed into the ranges of struct fsm_hip_dfa_desc and written by the library's own
 * fsm_hip_desc_write() ("FSMHIP02") -- the form fsm_hip_desc_read() / examples/hipgrep.c / fsm_hip_dfa_create() take.
 *
 * Compiled by integration/print/build.sh against the reference's headers where they lie (nothing of the reference is
 * stored here) and linked into the rebuilt rx / re next to the patched print.c.
 */
#include <assert.h>
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fsm/fsm.h>
#include <fsm/options.h>
#include <fsm/print.h>

#include "libfsm/print.h"       /* ir_print_f */
#include "libfsm/print/ir.h"    /* struct ir, ir_state, ir_group, ir_range */

#include <fsm_hip.h>

#define NOEDGE (-1L)

static void
ranges_into(long row[256], const struct ir_range *r, size_t n, long to)
{
	size_t i;
	for (i = 0; i < n; i++) {
		int c;
		for (c = r[i].start; c <= (int) r[i].end; c++) {
			row[c] = to;
		}
	}
}

static void
groups_into(long row[256], const struct ir_group *g, size_t n)
{
	size_t i;
	for (i = 0; i < n; i++) {
		ranges_into(row, g[i].ranges, g[i].n, (long) g[i].to);
	}
}

/* one state of the IR as a 256-entry row: the recipe of vm/ir.c:649-750 */
static int
state_row(const struct ir_state *st, long row[256])
{
	int c;
	long dflt = NOEDGE;

	switch (st->strategy) {
	case IR_NONE:     break;
	case IR_SAME:     dflt = (long) st->u.same.to; break;
	case IR_COMPLETE: break;
	case IR_PARTIAL:  break;
	case IR_DOMINANT: dflt = (long) st->u.dominant.mode; break;
	case IR_ERROR:    dflt = (long) st->u.error.mode; break;
	default:          errno = ENOTSUP; return 0;       /* IR_TABLE: "not yet implemented" upstream */
	}
	for (c = 0; c < 256; c++) {
		row[c] = dflt;
	}
	switch (st->strategy) {
	case IR_COMPLETE: groups_into(row, st->u.complete.groups, st->u.complete.n); break;
	case IR_PARTIAL:  groups_into(row, st->u.partial.groups, st->u.partial.n); break;
	case IR_DOMINANT: groups_into(row, st->u.dominant.groups, st->u.dominant.n); break;
	case IR_ERROR:
		ranges_into(row, st->u.error.error.ranges, st->u.error.error.n, NOEDGE);
		groups_into(row, st->u.error.groups, st->u.error.n);
		break;
	default: break;
	}
	return 1;
}

static int
cmp_u32(const void *a, const void *b)
{
	const uint32_t x = *(const uint32_t *) a, y = *(const uint32_t *) b;
	return x < y ? -1 : x > y;
}

void
fsm_hip_desc_from_ir_free(struct fsm_hip_dfa_desc *d)
{
	if (d == NULL) {
		return;
	}
	free((void *) d->edge_off);
	free((void *) d->ranges);
	free((void *) d->is_end);
	free((void *) d->endid_off);
	free((void *) d->endids);
	free((void *) d->eager_off);
	free((void *) d->eager_ids);
	free(d);
}

/* struct ir -> flat description (every array malloc'd; fsm_hip_desc_from_ir_free) */
struct fsm_hip_dfa_desc *
fsm_hip_desc_from_ir(const struct ir *ir)
{
	struct fsm_hip_dfa_desc *d;
	uint32_t *edge_off, *endid_off, *eager_off, *endids = NULL, *eager_ids = NULL;
	struct fsm_hip_range *ranges = NULL;
	uint8_t *is_end;
	size_t s, nr = 0, cap = 0, nid = 0, neg = 0, k;
	int any_eager = 0;

	if (ir == NULL || ir->n == 0 || ir->n >= 0x00FFFFFFu) {
		errno = EINVAL;
		return NULL;
	}
	d = calloc(1, sizeof *d);
	edge_off = calloc(ir->n + 1, sizeof *edge_off);
	endid_off = calloc(ir->n + 1, sizeof *endid_off);
	eager_off = calloc(ir->n + 1, sizeof *eager_off);
	is_end = calloc(ir->n, 1);
	if (d == NULL || edge_off == NULL || endid_off == NULL || eager_off == NULL || is_end == NULL) {
		goto oom;
	}
	for (s = 0; s < ir->n; s++) {
		nid += ir->states[s].endids.count;
		if (ir->states[s].eager_outputs != NULL) {
			neg += ir->states[s].eager_outputs->count;
		}
	}
	endids = calloc(nid ? nid : 1, sizeof *endids);
	eager_ids = calloc(neg ? neg : 1, sizeof *eager_ids);
	if (endids == NULL || eager_ids == NULL) {
		goto oom;
	}
	nid = neg = 0;
	for (s = 0; s < ir->n; s++) {
		const struct ir_state *st = &ir->states[s];
		long row[256];
		int c;

		if (!state_row(st, row)) {
			goto fail;
		}
		for (c = 0; c < 256; ) {
			int e = c;
			if (row[c] == NOEDGE) {
				c++;
				continue;
			}
			if ((size_t) row[c] >= ir->n) {
				errno = EINVAL;
				goto fail;
			}
			while (e + 1 < 256 && row[e + 1] == row[c]) {
				e++;
			}
			if (nr == cap) {
				struct fsm_hip_range *nw;
				cap = cap ? cap * 2 : 1024;
				nw = realloc(ranges, cap * sizeof *ranges);
				if (nw == NULL) {
					goto oom;
				}
				ranges = nw;
			}
			memset(&ranges[nr], 0, sizeof ranges[nr]);
			ranges[nr].lo = (unsigned char) c;
			ranges[nr].hi = (unsigned char) e;
			ranges[nr].to = (uint32_t) row[c];
			nr++;
			c = e + 1;
		}
		edge_off[s + 1] = (uint32_t) nr;
		is_end[s] = st->isend ? 1 : 0;
		/* end-ids (sorted, unique: fsm_endid_get order, as make_ir copied them) */
		for (k = 0; k < st->endids.count; k++) {
			endids[nid + k] = (uint32_t) st->endids.ids[k];
		}
		qsort(endids + nid, st->endids.count, sizeof *endids, cmp_u32);
		nid += st->endids.count;
		endid_off[s + 1] = (uint32_t) nid;
		if (st->eager_outputs != NULL) {
			for (k = 0; k < st->eager_outputs->count; k++) {
				eager_ids[neg + k] = (uint32_t) st->eager_outputs->ids[k];
			}
			qsort(eager_ids + neg, st->eager_outputs->count, sizeof *eager_ids, cmp_u32);
			neg += st->eager_outputs->count;
			any_eager = any_eager || st->eager_outputs->count != 0;
		}
		eager_off[s + 1] = (uint32_t) neg;
	}
	d->nstates = (uint32_t) ir->n;
	d->start = ir->start;
	d->edge_off = edge_off;
	d->ranges = ranges;
	d->is_end = is_end;
	d->endid_off = endid_off;
	d->endids = endids;
	if (any_eager) {
		d->eager_off = eager_off;
		d->eager_ids = eager_ids;
	} else {
		free(eager_off);
		free(eager_ids);
	}
	return d;

oom:
	errno = ENOMEM;
fail:
	free(d);
	free(edge_off);
	free(endid_off);
	free(eager_off);
	free(is_end);
	free(endids);
	free(eager_ids);
	free(ranges);
	return NULL;
}

int
fsm_print_hip(FILE *f, const struct fsm_options *opt, const struct fsm_hooks *hooks, const struct ir *ir)
{
	struct fsm_hip_dfa_desc *d;
	int r;

	assert(f != NULL);
	(void) opt;
	(void) hooks;

	d = fsm_hip_desc_from_ir(ir);
	if (d == NULL) {
		return -1;
	}
	r = fsm_hip_desc_write(d, f);
	fsm_hip_desc_from_ir_free(d);
	return r;
}
