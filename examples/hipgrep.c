/*
 * examples/hipgrep.c -- a caller of the path that needs no libfsm at run time:
 *
 *     hipgrep TABLE.fsmhip < lines.txt
 *
 * loads a DFA table written by fsm_hip_print() (e.g. from an rx-style union of patterns), reads
 * newline-separated records from stdin, matches every record in ONE batched launch
 * (fsm_hip_exec_batch_offsets) and prints "<line-number>:<end-id>[,<end-id>...]" for each record the
 * DFA accepts -- what `re -z` prints per argument (src/re/main.c:1152-1166), for a whole file.
 * Plain C against include/fsm_hip.h only.
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fsm_hip.h"

int
main(int argc, char **argv)
{
	struct fsm_hip_dfa_desc *desc;
	struct fsm_hip_dfa *dfa;
	unsigned char *buf = NULL;
	uint64_t *off = NULL;
	uint32_t *end = NULL;
	size_t cap = 0, len = 0, n = 0, ocap = 0, i, got, start;
	FILE *tf;

	if (argc != 2) {
		fprintf(stderr, "usage: hipgrep TABLE.fsmhip < records\n");
		return 2;
	}
	tf = fopen(argv[1], "rb");
	if (tf == NULL) {
		perror(argv[1]);
		return 2;
	}
	desc = fsm_hip_desc_read(tf);
	fclose(tf);
	if (desc == NULL) {
		perror("fsm_hip_desc_read");
		return 2;
	}
	dfa = fsm_hip_dfa_create(desc, 0);
	fsm_hip_desc_free(desc);
	if (dfa == NULL) {
		perror("fsm_hip_dfa_create");
		return 2;
	}

	/* slurp stdin */
	for (;;) {
		if (cap - len < 65536) {
			cap = cap ? cap * 2 : 1 << 20;
			buf = realloc(buf, cap);
			if (buf == NULL) {
				perror("realloc");
				return 2;
			}
		}
		got = fread(buf + len, 1, cap - len, stdin);
		if (got == 0) {
			break;
		}
		len += got;
	}
	/* records = lines without their '\n': offsets[i] .. offsets[i+1]-1 would include the newline,
	 * so the batch is built over a copy with the newlines squeezed out */
	start = 0;
	for (i = 0; i <= len; i++) {
		if (i == len ? start < len : buf[i] == '\n') {
			if (n + 2 > ocap) {
				ocap = ocap ? ocap * 2 : 1024;
				off = realloc(off, ocap * sizeof *off);
				if (off == NULL) {
					perror("realloc");
					return 2;
				}
			}
			if (n == 0) {
				off[0] = 0;
			}
			memmove(buf + off[n], buf + start, i - start);
			off[n + 1] = off[n] + (i - start);
			n++;
			start = i + 1;
		}
	}
	if (n == 0) {
		return 1;
	}
	end = malloc(n * sizeof *end);
	if (end == NULL || fsm_hip_exec_batch_offsets(dfa, buf, off, n, end, NULL) != 0) {
		perror("fsm_hip_exec_batch_offsets");
		return 2;
	}
	got = 0;
	for (i = 0; i < n; i++) {
		uint32_t ids[64];
		size_t c, k;
		if (end[i] == FSM_HIP_NO_MATCH) {
			continue;
		}
		got++;
		c = fsm_hip_endid_count(dfa, end[i]);
		printf("%zu:", i + 1);
		if (c <= 64 && fsm_hip_endid_get(dfa, end[i], c, ids)) {
			for (k = 0; k < c; k++) {
				printf(k ? ",%u" : "%u", ids[k]);
			}
		}
		putchar('\n');
	}
	fsm_hip_dfa_free(dfa);
	free(end);
	free(off);
	free(buf);
	return got ? 0 : 1;
}
