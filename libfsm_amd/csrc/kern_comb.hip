/* kern_comb.hip -- walk kernels of the comb (row-displacement) policies; see launch.h */
#include "launch.h"

namespace fsmhip {

hipError_t launch_comb(int pol, int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	if (pol == POL_COMBSELF) return launch_family<CombSelfPol>(eager, c, a, grid, block, s);
	if (pol == POL_COMB256) return launch_family<Comb256Pol>(eager, c, a, grid, block, s);
	return launch_family<CombPol>(eager, c, a, grid, block, s);
}

} // namespace fsmhip
