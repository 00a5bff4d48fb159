/* flat.h -- owner of the arrays behind a malloc'd struct fsm_hip_dfa_desc (shim.c, strings.cpp);
 * fsm_hip_desc_free() releases every member with free(). */
#ifndef FSMHIP_CSRC_FLAT_H
#define FSMHIP_CSRC_FLAT_H

#include "../../include/fsm_hip.h"

struct flat {
	struct fsm_hip_dfa_desc d;
	uint32_t *edge_off;
	struct fsm_hip_range *ranges;
	uint8_t *is_end;
	uint32_t *endid_off;
	uint32_t *endids;
	uint32_t *eager_off;
	uint32_t *eager_ids;
};

#endif
