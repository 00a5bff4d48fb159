/* kern_tiny.hip -- walk kernels of the column-table policies (<= 16 states); see launch.h */
#include "launch.h"

namespace fsmhip {

hipError_t launch_tiny(int pol, int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	if (pol == POL_TINY5) return launch_family<Tiny5Pol>(eager, c, a, grid, block, s);
	return launch_family<TinyPol<uint64_t>>(eager, c, a, grid, block, s);
}

} // namespace fsmhip
