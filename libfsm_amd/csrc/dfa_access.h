/*
 * dfa_access.h -- what the other translation units of libfsm_hip.so may know about a struct fsm_hip_dfa
 * (defined in fsm_hip.hip): its host-side plan and its device.  Not part of the C ABI.
 */
#ifndef FSMHIP_CSRC_DFA_ACCESS_H
#define FSMHIP_CSRC_DFA_ACCESS_H

#include <vector>

#include "plan.h"

struct fsm_hip_dfa;

namespace fsmhip {
__attribute__((visibility("hidden"))) const Plan *dfa_plan(const fsm_hip_dfa *d);
__attribute__((visibility("hidden"))) int dfa_device(const fsm_hip_dfa *d);
__attribute__((visibility("hidden"))) int dfa_ncu(const fsm_hip_dfa *d);
/* what fsm_hip_exec_batch_ids (mode EARLIEST or RET) writes for an input ending in renumbered state n; *conflict: the lowest
 * caller's end state that carries more than one id, or FSM_HIP_NO_MATCH */
__attribute__((visibility("hidden"))) int dfa_ids_by_state(const fsm_hip_dfa *d, int mode, std::vector<uint32_t> &out, uint32_t *conflict);
}

#endif
