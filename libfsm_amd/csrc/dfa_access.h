/*
 * dfa_access.h -- what the other translation units of libfsm_hip.so may know about a struct fsm_hip_dfa
 * (defined in fsm_hip.hip): its host-side plan and its device.  Not part of the C ABI.
 */
#ifndef FSMHIP_CSRC_DFA_ACCESS_H
#define FSMHIP_CSRC_DFA_ACCESS_H

#include "plan.h"

struct fsm_hip_dfa;

namespace fsmhip {
__attribute__((visibility("hidden"))) const Plan *dfa_plan(const fsm_hip_dfa *d);
__attribute__((visibility("hidden"))) int dfa_device(const fsm_hip_dfa *d);
__attribute__((visibility("hidden"))) int dfa_ncu(const fsm_hip_dfa *d);
}

#endif
