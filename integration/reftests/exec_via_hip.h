/* integration/reftests/exec_via_hip.h -- force-included (gcc -include) into the reference's own test programs:
 * every fsm_exec() call they make is answered by the HIP path instead ("reference C tests
 * re-linked against the shim").  The prototype stays libfsm's; only the callee changes. */
#ifndef EXEC_VIA_HIP_H
#define EXEC_VIA_HIP_H

#include <fsm/fsm.h>

int fsm_exec_via_hip(const struct fsm *fsm, int (*fsm_getc)(void *opaque), void *opaque,
	fsm_state_t *end, struct fsm_capture *captures);

#define fsm_exec fsm_exec_via_hip

#endif
