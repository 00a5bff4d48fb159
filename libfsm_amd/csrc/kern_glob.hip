/* kern_glob.hip -- walk kernels of the policies whose table stays in HBM / L2; see launch.h */
#include "launch.h"

namespace fsmhip {

hipError_t launch_glob(int pol, int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	if (pol == POL_SPARSE) return launch_family<SparsePol>(eager, c, a, grid, block, s);
	return launch_family<GlobPol>(eager, c, a, grid, block, s);
}

} // namespace fsmhip
