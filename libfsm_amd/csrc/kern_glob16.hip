/* kern_glob16.hip -- walk kernels of the 2-byte-entry table in HBM / L2 with its head in LDS (Glob16Pol); see launch.h */
#include "launch.h"

namespace fsmhip {

hipError_t launch_glob16(int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	/* fixed-stride rows, plain walk: two inputs per lane (Glob16Pol::next2) */
	if (eager == 0 && c.mode == IN_DIRECT && c.nb != 8) return launch_fn(walk_direct<Glob16Pol, 4, 2>, c, a, grid, block, s);
	return launch_family<Glob16Pol>(eager, c, a, grid, block, s);
}

} // namespace fsmhip
