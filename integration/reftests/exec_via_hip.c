/* integration/reftests/exec_via_hip.c -- fsm_exec() for the reference's test programs, through libfsm_hip.so:
 * the automaton handed in is compiled for the GPU as it stands (the tests transform it between calls, so nothing
 * is cached), the input is walked there (fsm_hip_exec: same arguments and return convention as fsm_exec), and the
 * table is freed again.  One launch per call: plumbing, not speed.  A count of the calls is printed at exit; an
 * automaton the boundary does not take (capture paths: ENOTSUP) would be counted as a fallback to libfsm's own
 * fsm_exec -- the endids / re_strings programs have none. */
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>

#include <fsm/fsm.h>   /* the real fsm_exec: this file is compiled WITHOUT exec_via_hip.h */

#include <fsm_hip.h>

static unsigned long n_hip, n_fallback;

static void
report(void)
{
	fprintf(stderr, "exec_via_hip: %lu fsm_exec calls answered by the HIP path, %lu fallbacks\n", n_hip, n_fallback);
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
	r = fsm_hip_exec(dfa, fsm_getc, opaque, end, captures);
	fsm_hip_dfa_free(dfa);
	return r;
}
