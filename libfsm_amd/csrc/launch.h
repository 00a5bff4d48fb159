/*
 * launch.h -- kernel selection shared by the per-policy translation units.
 *
 * The walk kernels are templates over the table policy; instantiating every (policy, kernel) pair in
 * one translation unit made the build serial (two minutes).  Each kern_*.hip instantiates the kernels
 * of one policy family through launch_family<>() below and exports one plain function; build.sh
 * compiles the units in parallel.  fsm_hip.hip holds the host side only.
 */
#ifndef FSM_HIP_LAUNCH_H
#define FSM_HIP_LAUNCH_H

#include "walk_kernels.h"

namespace fsmhip {

struct LaunchCfg {
	int mode;            /* IN_DIRECT | IN_LDSDMA | IN_GENERIC | IN_RAGGED | IN_LAZY */
	int nb;              /* direct: 16-byte chunks in flight per lane (4 or 8) */
	int waves, blocks_per_cu;
	int seg;             /* LDS-DMA: 64 or 128 */
	int prefetch;        /* direct: register double-buffer (0: <= 64 VGPRs, occupancy instead) */
	int nt;              /* LDS-DMA, 128-byte segments: nontemporal loads */
	int sparse_fast;     /* sparse layout, per-lane loads, plain walk: the entry-as-state policy (SparseFastPol) */
	int lazy_abs;        /* IN_LAZY: an absorbing state is reachable (the kernel variant that tests for one) */
	int lazy_rows;       /* IN_LAZY: inputs per lane (3, or 2 / 4 by FSM_HIP_KNOB_ROWS: A/B) */
	int lines32;         /* IN_GENERIC, plain walk of a packed batch below 4 GiB / 2^29 inputs: the 32-bit kernel (walk_lines32) */
	uint32_t lds;        /* dynamic LDS bytes per workgroup */
	int probe;           /* 1: do not launch, only say (kfn) which kernel it would be (fsm_hip.hip sizes a workgroup by the kernel's registers) */
	mutable const void *kfn;   /* out: the kernel launch_fn launched (its name goes into fsm_hip_last_kernel_name) */
};

/* which policy of a family */
enum {
	POL_TINY5 = 0, POL_TINY64 = 1,
	POL_LDS = 0, POL_LDSSELF = 1, POL_LDS2 = 2,
	POL_COMB = 0, POL_COMB256 = 1, POL_COMBSELF = 2,
	POL_GLOB = 0, POL_SPARSE = 1
};
/* eager: 0 = plain walk, 1 = EagerPol (<= 64 ids, set in registers), 2 = EagerWidePol */
hipError_t launch_tiny(int pol, int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s);
hipError_t launch_lds(int pol, int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s);
hipError_t launch_comb(int pol, int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s);
hipError_t launch_glob(int pol, int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s);
hipError_t launch_glob16(int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s);   /* Glob16Pol: kern_glob16.hip */

typedef void (*walk_fn)(const WalkArgs);

static inline hipError_t launch_fn(walk_fn k, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	if (c.probe) {
		c.kfn = (const void *)k;
		return hipSuccess;
	}
	hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds);
	if (e != hipSuccess) return e;
	WalkArgs b = a;
	b.lds_bytes = c.lds;
	hipLaunchKernelGGL(k, grid, block, c.lds, s, b);
	c.kfn = (const void *)k;
	return hipGetLastError();
}

/* thread cap of the plain LDS-DMA kernel: CombSelfPol runs 12 waves behind LDS-DMA (measured best, and its
 * range-skip state + the 32-VGPR tile want more than the 128 registers of a 16-wave workgroup) */
template <class Pol> struct ldsdma_threads { static constexpr int value = 1024; };
template <> struct ldsdma_threads<CombSelfPol> { static constexpr int value = 768; };

/* plain walk: every input path */
template <class Pol>
static hipError_t launch_pol(const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	walk_fn k = nullptr;
	switch (c.mode) {
	case IN_RAGGED:
		/* (one instantiation per metadata form, as for walk_generic, takes this kernel's SGPR spills from 21-44 to 0-15 -- and
		 * its rate on the column tables from 2.95 to 2.71 TB/s, on the C3 table from 3.49 to 3.52: a same-box A/B,
		 * profiles/r06i_ragged_per_form_ab.txt.  The staging loads' wait moves.  The form-generic kernel stays.) */
		k = walk_ragged<Pol, 768>;
		break;
	case IN_GENERIC:
		/* the plain walk (no second output table, no resume) has an instantiation per metadata form: with the form decided at
		 * run time every pointer of every form stays live across the loop -- 40-56 scalar registers spilled to vector lanes
		 * against 7-19 (tools/kernel_resources.py) */
		if (lines32_ok<Pol>::value && c.lines32 && a.out2 == nullptr && a.state_io == nullptr && (a.off != nullptr || a.off32 != nullptr || a.tbase != nullptr)) {
			if constexpr (lines32_ok<Pol>::value)
				k = a.off != nullptr ? walk_lines32<Pol, FR_OFF64> : a.off32 != nullptr ? walk_lines32<Pol, FR_OFF32> : walk_lines32<Pol, FR_LENS>;
		} else if (a.out2 == nullptr && a.state_io == nullptr) {
			k = a.off != nullptr ? walk_generic<Pol, 1024, true, FR_OFF64> : a.off32 != nullptr ? walk_generic<Pol, 1024, true, FR_OFF32>
			  : a.tbase != nullptr ? walk_generic<Pol, 1024, true, FR_LENS> : walk_generic<Pol, 1024, true, FR_STRIDE>;
		} else k = walk_generic<Pol>;
		break;
	case IN_LDSDMA:
		if (c.seg == 128) k = c.nt ? walk_ldsdma<Pol, 128, 2, ldsdma_threads<Pol>::value> : walk_ldsdma<Pol, 128, 0, ldsdma_threads<Pol>::value>;
		else k = walk_ldsdma<Pol, 64, 0, ldsdma_threads<Pol>::value>;
		break;
	default:
		if (!c.prefetch && c.nb == 4) k = walk_direct_np<Pol, 4>;
		else k = c.nb == 4 ? walk_direct<Pol, 4, 1> : walk_direct<Pol, 8, 1>;
		break;
	}
	return launch_fn(k, c, a, grid, block, s);
}

/* thread cap of the eager LDS-DMA kernel.  The chunk-level eager walk (EagerPol::walk16) keeps a chunk's 16 lookups live
 * across its two passes next to the tile and the 64-bit id set: every layout now fits the 128 VGPRs of a 16-wave
 * workgroup without spilling (96-127, tools/kernel_resources.py) except the 64-bit column table, the class comb and the two
 * self-loop-mask layouts (a third register per state), which spill 10 there and stay at 12 waves (<= 170 VGPRs).  Round 2 ran all of them at 12 waves: 0.71 of the plain walk's rate
 * on the lds layout, next to 12 / 16 = 0.75 (profiles/r02f_eager_probe.txt). */
template <class Pol> struct eager_dma_threads { static constexpr int value = 1024; };
template <> struct eager_dma_threads<TinyPol<uint64_t>> { static constexpr int value = 768; };
template <> struct eager_dma_threads<CombPol> { static constexpr int value = 768; };       /* 10 spills at 1024, so do the next two */
template <> struct eager_dma_threads<CombSelfPol> { static constexpr int value = 768; };
template <> struct eager_dma_threads<LdsSelfPol> { static constexpr int value = 768; };

/* eager-output walks: the policy wrapped; the register-set form also behind LDS-DMA (128-byte segments) */
template <class EP, bool DMA, int DMAT>
static hipError_t launch_eager_pol(const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	walk_fn k = nullptr;
	switch (c.mode) {
	case IN_RAGGED:  k = walk_ragged<EP, 512>; break;
	case IN_GENERIC: k = walk_generic<EP, 512>; break;
	case IN_LDSDMA:
		if constexpr (DMA) { k = walk_ldsdma<EP, 128, 2, DMAT>; break; }   /* wide sets: per-lane loads (pick_cfg never asks) */
		/* fallthrough */
	default: k = walk_direct<EP, 4, 1>; break;
	}
	return launch_fn(k, c, a, grid, block, s);
}

template <class Pol>
static hipError_t launch_family(int eager, const LaunchCfg &c, const WalkArgs &a, dim3 grid, dim3 block, hipStream_t s)
{
	if (eager == 0) return launch_pol<Pol>(c, a, grid, block, s);
	if (eager == 1) return launch_eager_pol<EagerPol<Pol>, true, eager_dma_threads<Pol>::value>(c, a, grid, block, s);
	return launch_eager_pol<EagerWidePol<Pol>, false, 1024>(c, a, grid, block, s);
}

} // namespace fsmhip

#endif
