/* kern_lds.hip -- walk kernels of the dense LDS-table policies; see launch.h */
#include "launch.h"

namespace fsmhip {

hipError_t launch_lds(int pol, int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	if (pol == POL_LDSSELF) return launch_family<LdsSelfPol>(eager, c, a, grid, block, s);
	if (pol == POL_LDS2) return launch_pol<Lds2Pol>(c, a, grid, block, s);   /* plain walks only (plan.cpp emit_lds2) */
	return launch_family<LdsPol>(eager, c, a, grid, block, s);
}

} // namespace fsmhip
