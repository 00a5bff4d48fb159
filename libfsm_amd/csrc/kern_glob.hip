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
		/* THREE inputs per lane, four chunks each in flight (round 6: 1 036 GB/s on the 1e5-literal automaton where round 5's two
		 * inputs gave 888).  What made room for the third: no record sends a hit the exact way any more (plan.cpp's clones), and
		 * the byte -> shift lookups are taken eight at a time with their OR at once (walk_lazy.h) -- 128 registers, no scratch.
		 * FSM_HIP_KNOB_ROWS = 2: round 5's shape (A/B; the 3 x 2 and 4 x 2 shapes of profiles/r09j_* were measured and are not built). */
		/* (the byte -> shift lookups eight at a time here: four and two measured slower on this kernel, 858 and 906 against 959 --
		 * profiles/r09u_c5_fixed_stride_lookup_batch.txt -- where the lines kernel below gains from two) */
		walk_fn k = c.lazy_abs ? walk_lazy<true, 3, 4> : walk_lazy<false, 3, 4>;
		if (c.lazy_rows == 2) k = c.nt ? (c.lazy_abs ? walk_lazy<true, 2, 4, true> : walk_lazy<false, 2, 4, true>) : (c.lazy_abs ? walk_lazy<true, 2, 4> : walk_lazy<false, 2, 4>);
		return launch_fn(k, c, a, grid, block, s);
	}
	if (pol == POL_SPARSE && eager == 0 && c.mode == IN_LAZY_LINES) {
		/* the same walk on inputs of any length / metadata form, and resumed walks: one input per lane slot, lane refill */
		/* three slots per lane, THREE whole chunks per slot and turn, the byte -> shift lookups TWO at a time: taking them eight at
		 * a time (the fixed-stride kernel's way) cost 24 registers, and with those back a third chunk per turn fits without scratch
		 * (109 registers) -- a turn's head and its wait for the chunks are paid once per 48 bytes of every input instead of 32.
		 * 0-1024 B lines 605 -> 664 GB/s, 8-64 B 424 -> 486, all 64 B 866 -> 955, all 1 KiB 766 -> 843 (four chunks: 668 / 439 / 990 /
		 * 865; four slots x two chunks, one lookup at a time: 642 / 486 / 989 / 902: profiles/r09u_*).
		 * By knob (A/B): FSM_HIP_KNOB_NB = 2 the first half of round 6's shape (two chunks, lookups eight at a time), 4 four chunks; ROWS = 2 round 5's */
		walk_fn k = c.lazy_abs ? walk_lazy_lines<true, 3, 3, 2> : walk_lazy_lines<false, 3, 3, 2>;
		if (c.lazy_rows == 3 && c.nb == 2) k = c.lazy_abs ? walk_lazy_lines<true, 3, 2, 8> : walk_lazy_lines<false, 3, 2, 8>;
		if (c.lazy_rows == 3 && c.nb == 4) k = c.lazy_abs ? walk_lazy_lines<true, 3, 4, 2> : walk_lazy_lines<false, 3, 4, 2>;
		if (c.lazy_rows == 2) k = c.nb == 2 ? (c.lazy_abs ? walk_lazy_lines<true, 2, 2> : walk_lazy_lines<false, 2, 2>) : (c.lazy_abs ? walk_lazy_lines<true, 2, 4> : walk_lazy_lines<false, 2, 4>);
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
