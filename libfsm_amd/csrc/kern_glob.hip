/* kern_glob.hip -- walk kernels of the policies whose table stays in HBM / L2; see launch.h */
#include "launch.h"
#include "walk_lazy.h"

namespace fsmhip {

hipError_t launch_glob(int pol, int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	if (pol == POL_SPARSE && eager == 0 && c.mode == IN_LAZY) {
		/* fixed-stride rows, plain walk, an automaton with a lazy form: states beyond the LDS set are entered without their record */
		/* (three inputs per lane with two chunks in flight, and two with two, measured no faster than <2, 4>:
		 * profiles/r06b_c5_lazy_variants_1e7.txt) */
		walk_fn k = c.nt ? (c.lazy_abs ? walk_lazy<true, 2, 4, true> : walk_lazy<false, 2, 4, true>) : (c.lazy_abs ? walk_lazy<true, 2, 4> : walk_lazy<false, 2, 4>);
		return launch_fn(k, c, a, grid, block, s);
	}
	if (pol == POL_SPARSE && eager == 0 && c.mode == IN_LAZY_LINES) {
		/* the same walk on inputs of any length / metadata form, and resumed walks: one input per lane slot, lane refill */
		walk_fn k = c.nb == 2 ? (c.lazy_abs ? walk_lazy_lines<true, 2, 2> : walk_lazy_lines<false, 2, 2>) : (c.lazy_abs ? walk_lazy_lines<true, 2, 4> : walk_lazy_lines<false, 2, 4>);
		return launch_fn(k, c, a, grid, block, s);
	}
	if (pol == POL_SPARSE && eager == 0 && c.sparse_fast && c.mode == IN_DIRECT) {
		/* fixed-stride rows, plain walk: the record is the state (SparseFastPol) */
		walk_fn k = !c.prefetch && c.nb == 4 ? (c.sparse_fast == 2 ? walk_direct_np<SparseFastPol, 2> : walk_direct_np<SparseFastPol, 4>) : c.nb == 4 ? walk_direct<SparseFastPol, 4, 1> : walk_direct<SparseFastPol, 8, 1>;
		return launch_fn(k, c, a, grid, block, s);
	}
	if (pol == POL_SPARSE) return launch_family<SparsePol>(eager, c, a, grid, block, s);
	return launch_family<GlobPol>(eager, c, a, grid, block, s);
}

} // namespace fsmhip
