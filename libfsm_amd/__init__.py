"""libfsm_amd -- MI355X-native batched DFA execution behind libfsm's C API.

The product is libfsm_hip.so (C ABI in include/fsm_hip.h, HIP kernels in
libfsm_amd/csrc/).  This package only binds it for the Python harness.
"""
from .capi import (  # noqa: F401
    NO_MATCH, NO_ID, STATE_START, STATE_DEAD, FlatDfa, HipDfa, HipNode, Plan, load_library, gen_inputs_host, gen_inputs_device,
    LAYOUT_AUTO, LAYOUT_TINY, LAYOUT_LDS, LAYOUT_COMB, LAYOUT_GLOBAL, LAYOUT_COMB256, LAYOUT_COMBSELF, LAYOUT_SPARSE, LAYOUT_LDSSELF, LAYOUT_LDS2, META_OFF64, META_OFF32, META_LENGTHS, ALL_LAYOUTS, NO_EARLY_RETIRE,
    KNOB_INPUT_MODE, KNOB_NB, KNOB_ROWS, KNOB_WAVES, KNOB_BLOCKS_PER_CU, KNOB_EARLY_RETIRE, KNOB_MASK, KNOB_HOT_BYTES, KNOB_SEG, KNOB_PREFETCH, KNOB_NT, KNOB_NOSKIP, KNOB_RAGGED_ALIGN, KNOB_DMA_BUFS,
    IN_DIRECT, IN_LDSDMA, IN_GENERIC, IN_RAGGED, KNOB_PICK_MEAN, KNOB_SPARSE_FAST, KNOB_LAZY_DYN, KNOB_LAZY_LINES, LIB_PATH, pack_affixes, gen_pack_rows_device, gather_probe_ms, gen_affix_inputs_host, gen_affix_inputs_device, stream_read_probe_gbps,
    DEFER_UPLOAD, MultiBatch, MultiBatchIds, exec_multi, exec_multi_device, exec_multi_ids, exec_multi_ids_device, MultiPrepared, multi_last_launches, multi_last_fused_jobs, multi_assign, lds_chain_probe_gbps, waves_by_occupancy,
)
