/* integration/reftests/exec_via_hip.c -- fsm_exec() for the reference's test programs, through libfsm_hip.so:
 * the automaton handed in is compiled for the GPU as it stands (the tests transform it between calls, so nothing
 * is cached), the input is walked there (fsm_hip_exec: same arguments and return convention as fsm_exec), and the
 * table is freed again.  One launch per call: plumbing, not speed.  A count of the calls is printed at exit; an
 * automaton the boundary does not take (capture paths: ENOTSUP) would be counted as a fallback to libfsm's own
 * fsm_exec -- the endids / re_strings / eager_output programs have none. */
#include <errno.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#include <fsm/fsm.h>   /* the real fsm_exec: this file is compiled WITHOUT exec_via_hip.h */

#include <fsm_hip.h>

static unsigned long n_hip, n_fallback, n_eager;

static void
report(void)
{
	fprintf(stderr, "exec_via_hip: %lu fsm_exec calls answered by the HIP path, %lu fallbacks (%lu with eager outputs delivered through the automaton's callback)\n",
		n_hip, n_fallback, n_eager);
}

int
fsm_exec_via_hip(const struct fsm *fsm, int (*fsm_getc)(void *opaque), void *opaque,
	fsm_state_t *end, struct fsm_capture *captures)
{
	struct fsm_hip_dfa *dfa;
	int r;

	if (n_hip + n_fallback == 0) {
		atexit(report);
	}

	dfa = fsm_hip_compile(fsm, 0);
	if (dfa == NULL) {
		if (errno == ENOTSUP) {
			n_fallback++;
			return fsm_exec(fsm, fsm_getc, opaque, end, captures);
		}
		n_hip++;
		return -1;   /* EINVAL: not a DFA / no start state, as fsm_exec itself answers (exec.c:106-114) */
	}

	n_hip++;
	{
		/* eager outputs (exec.c:126-144): fsm_exec calls the automaton's callback for the ids of the start state and
		 * of every state entered.  The device returns the SET of ids emitted along the walk; the callback is called
		 * once per member, in ascending order (the tests' callback ignores repeats, tests/eager_output/utils.c:10-22). */
		fsm_eager_output_cb *cb = NULL;
		void *cb_opaque = NULL;

		fsm_eager_output_get_cb(fsm, &cb, &cb_opaque);
		if (cb != NULL && fsm_hip_eager_id_count(dfa) > 0) {
			size_t n = 0, cap = 256, w, words = fsm_hip_eager_words(dfa);
			unsigned char *buf = malloc(cap);
			uint64_t *set = calloc(words, sizeof *set);
			uint32_t len32, e = FSM_HIP_NO_MATCH;
			int c;

			if (buf == NULL || set == NULL) { free(buf); free(set); fsm_hip_dfa_free(dfa); return -1; }
			while (c = fsm_getc(opaque), c != EOF) {
				if (n == cap) {
					unsigned char *nb = realloc(buf, cap *= 2);
					if (nb == NULL) { free(buf); free(set); fsm_hip_dfa_free(dfa); return -1; }
					buf = nb;
				}
				buf[n++] = (unsigned char) c;
			}
			len32 = (uint32_t) n;
			r = fsm_hip_exec_batch_eager(dfa, buf, n, &len32, 1, &e, set);
			if (r == 0) {
				n_eager++;
				for (w = 0; w < words * 64; w++) {
					if (set[w / 64] >> (w % 64) & 1) {
						cb(fsm_hip_eager_id(dfa, (unsigned) w), cb_opaque);
					}
				}
				r = e != FSM_HIP_NO_MATCH;
				if (r == 1 && end != NULL) {
					*end = e;
				}
			}
			free(buf);
			free(set);
			fsm_hip_dfa_free(dfa);
			return r;
		}
	}
	r = fsm_hip_exec(dfa, fsm_getc, opaque, end, captures);
	fsm_hip_dfa_free(dfa);
	return r;
}
