/*
 * fsm_hip.h -- C ABI of libfsm_hip.so: batched DFA execution on MI355X (gfx950).
 *
 * Drop-in boundary for ONE path of katef/libfsm: "walk a finished DFA over
 * input bytes and report accept/reject + the end state".  Every entry point
 * cites the reference interface it replaces (paths relative to the reference
 * tree).  Plain pointers and sizes only; no C++/torch types.
 *
 * Two layers live in the one shared object:
 *
 *   1. core   -- takes a flat DFA description (struct fsm_hip_dfa_desc) and
 *                raw byte buffers.  No dependency on libfsm at all.
 *   2. shim   -- takes libfsm's own `const struct fsm *` and reads it through
 *                libfsm's PUBLIC API only (fsm_countstates, fsm_getstart,
 *                fsm_isend, fsm_walk_edges, fsm_all/fsm_isdfa,
 *                fsm_countcaptures, fsm_eager_output_count, fsm_endid_count/get).
 *                Those symbols are resolved with dlsym(RTLD_DEFAULT) on first
 *                use, i.e. from whichever libfsm the host program already
 *                links, so libfsm_hip.so itself loads without libfsm.
 *
 * Error convention follows the reference: functions returning int give
 * -1 + errno on failure (cf. fsm_exec, src/libfsm/exec.c:106-114), pointer
 * returning functions give NULL + errno (cf. fsm_vm_compile, src/libfsm/vm.c:88-131).
 * There is NO CPU fallback: without a usable HIP device every exec call fails
 * with -1/ENODEV.
 *
 * Threading: the table of a struct fsm_hip_dfa is immutable after creation.  What
 * exec calls do mutate -- the staging arena of the host-pointer fronts, the
 * timing events, the lazily built end-id / resume tables -- sits behind a
 * per-dfa mutex, so several host threads may share one dfa: host-pointer calls
 * on it run one after the other, device-pointer calls only serialise their
 * enqueue (tuning knobs are not locked: set them before sharing).  Distinct dfa
 * objects are fully independent.  A call makes the dfa's device current while
 * it runs and restores the caller's device before it returns; host-pointer
 * fronts run on a private non-blocking stream, never on the NULL stream.
 */
#ifndef FSM_HIP_H
#define FSM_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libfsm's own opaque/public types, forward-declared exactly as the reference
 * does (include/fsm/fsm.h:13, :24, :28; include/fsm/capture.h:21-24). */
struct fsm;
struct fsm_capture;
typedef unsigned int fsm_state_t;   /* include/fsm/fsm.h:24 */
typedef unsigned int fsm_end_id_t;  /* include/fsm/fsm.h:28 */

/* Value written to end_out[i] for a rejected input.  The reference leaves *end
 * untouched on reject (src/libfsm/exec.c:133-138, :153-155); a batch needs a
 * sentinel instead.  fsm_edge.state is 24 bit (src/libfsm/internal.h:48) so
 * no real state id can collide with it. */
#define FSM_HIP_NO_MATCH 0xFFFFFFFFu

/* ------------------------------------------------------------------ */
/* flat DFA description (core layer input)                            */
/* ------------------------------------------------------------------ */

/* One labelled edge group: bytes lo..hi (inclusive) go to state `to`.
 * This is the byte-range form of the reference's struct edge_group
 * {uint64 symbols[4]; fsm_state_t to} (src/adt/edgeset.c:34-41) and of the
 * codegen IR's struct ir_range (src/libfsm/print/ir.h:23-108). */
struct fsm_hip_range {
	uint8_t  lo, hi;
	uint16_t reserved;
	uint32_t to;
};

struct fsm_hip_dfa_desc {
	uint32_t nstates;                    /* fsm_countstates() */
	uint32_t start;                      /* fsm_getstart()    */
	const uint32_t *edge_off;            /* nstates+1 CSR offsets into ranges[] */
	const struct fsm_hip_range *ranges;  /* ranges of one state must not overlap (DFA) */
	const uint8_t  *is_end;              /* nstates; fsm_isend() */
	const uint32_t *endid_off;           /* nstates+1 CSR offsets into endids[], or NULL */
	const uint32_t *endids;              /* sorted unique per state (fsm_endid_get order) */
	/* eager outputs (src/libfsm/eager_output.c): ids emitted whenever the state is entered,
	 * start state included (exec.c:126-144).  NULL = none. */
	const uint32_t *eager_off;           /* nstates+1 CSR offsets into eager_ids[], or NULL */
	const uint32_t *eager_ids;           /* sorted unique per state (fsm_eager_output_get order) */
};

/* flags for fsm_hip_dfa_create / fsm_hip_compile */
enum {
	FSM_HIP_LAYOUT_AUTO   = 0,   /* planner picks by table size            */
	FSM_HIP_LAYOUT_TINY   = 1,   /* <=16 states: column-in-register walk   */
	FSM_HIP_LAYOUT_LDS    = 2,   /* class-compressed dense table in LDS    */
	FSM_HIP_LAYOUT_COMB   = 3,   /* column-default + comb exceptions, LDS  */
	FSM_HIP_LAYOUT_GLOBAL = 4,   /* class-compressed table in HBM/L2       */
	FSM_HIP_LAYOUT_COMB256 = 5,  /* byte-indexed comb, one default state, LDS */
	FSM_HIP_LAYOUT_COMBSELF = 6, /* comb + self-loop mask per state (<= 32 classes) */
	FSM_HIP_LAYOUT_SPARSE = 7,   /* base-row records (failure-link form) in HBM/L2, hottest in LDS */
	FSM_HIP_LAYOUT_LDSSELF = 8,  /* dense table in LDS + self-loop mask per state (<= 32 classes) */
	FSM_HIP_LAYOUT_LDS2   = 9,   /* dense table over PAIRS of byte classes in LDS: one lookup per two input bytes (plain walks) */
	FSM_HIP_LAYOUT_MASK   = 0xf,
	FSM_HIP_NO_EARLY_RETIRE = 0x10, /* never stop a wavefront early on absorbing states */
	FSM_HIP_DEFER_UPLOAD  = 0x20    /* plan now, upload the layout's device image at the first call that needs it: a dfa used only
	                                 * through fsm_hip_exec_multi (a new DFA per retest record, src/retest/main.c:1056) never does */
};

struct fsm_hip_dfa;   /* opaque: device-resident transition table + host-side end-id table */

struct fsm_hip_dfa_info {
	uint32_t nstates;        /* as given (without the synthetic dead state) */
	uint32_t nclasses;       /* byte equivalence classes */
	uint32_t layout;         /* FSM_HIP_LAYOUT_* actually chosen */
	uint32_t nabsorbing;     /* states whose every byte loops to itself (+ dead state) */
	uint64_t table_bytes;    /* bytes of the device transition table */
	uint32_t lds_bytes;      /* LDS bytes per workgroup */
	uint32_t waves_per_block;
	uint32_t device;         /* HIP device ordinal the table lives on */
	uint32_t reserved;
};

/* Build the device table from a flat description.  Replaces, for this path,
 * fsm_vm_compile_with_options() (src/libfsm/vm.c:88-131): DFA -> executable
 * form, self-contained after return (the caller may free `desc`, like retest
 * frees the fsm right after fsm_runner_initialize, src/retest/main.c:1056-1058).
 * NULL + errno=EINVAL if desc is not a DFA (overlapping ranges, bad ids),
 * ENODEV without a HIP device, ENOMEM. */
struct fsm_hip_dfa *fsm_hip_dfa_create(const struct fsm_hip_dfa_desc *desc, unsigned flags);

/* cf. fsm_vm_free() include/fsm/vm.h:56 */
void fsm_hip_dfa_free(struct fsm_hip_dfa *dfa);

int fsm_hip_dfa_info(const struct fsm_hip_dfa *dfa, struct fsm_hip_dfa_info *out);

/* Everything a later call on batches of up to n inputs would allocate, now: the tile-base block of the lengths-only front (and
 * the layout's tables, for a dfa created with FSM_HIP_DEFER_UPLOAD).  After it the device-pointer fronts make no allocation
 * for such batches: they can be captured into a HIP graph from the first launch (an allocation that would be needed during a
 * stream capture fails with ENOMEM instead of breaking the capture). */
int fsm_hip_reserve(struct fsm_hip_dfa *dfa, size_t n);

/* ------------------------------------------------------------------ */
/* batched execution (the hot path)                                   */
/* ------------------------------------------------------------------ */

/* Run n independent inputs through the DFA.  Input i is the len[i] bytes at
 * base + i*stride (len == NULL: every input is exactly `stride` bytes).
 * For each input this computes exactly what
 *     fsm_exec(fsm, getc_over(ptr,len), &opaque, &end, NULL)
 * computes (src/libfsm/exec.c:85-167): end_out[i] = end state if it returned 1,
 * FSM_HIP_NO_MATCH if it returned 0.  accept_bitmap (optional, may be NULL)
 * gets bit (i%64) of word (i/64) set iff input i matched; it must hold
 * ceil(n/64) words.  end_out may be NULL if only the bitmap is wanted.
 * Host pointers; data is staged through HBM (PCIe-inclusive).
 * Replaces the per-input loop over fsm_runner_run() in retest/reperf
 * (src/retest/main.c:1114, src/retest/reperf.c:772-784).
 * Returns 0, or -1 + errno.  One input may be up to 2^36 - 1 bytes long (the kernels count its 16-byte
 * chunks in 32 bits); longer ones go through fsm_hip_exec_batch_resume() in pieces. */
int fsm_hip_exec_batch(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap);

/* Same, inputs packed back to back: input i is bytes [off[i], off[i+1]) of base. */
int fsm_hip_exec_batch_offsets(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, const uint64_t *off, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap);

/* Same as fsm_hip_exec_batch but every pointer is a DEVICE pointer on the
 * dfa's device and the launch is asynchronous on `hip_stream` (a hipStream_t
 * passed as void *, NULL = default stream).  Nothing crosses PCIe.
 * CONTRACT of every *_device entry point: the metadata is the caller's to get right -- the host fronts check
 * len[i] <= stride and off[i] <= off[i + 1], a device front cannot without a synchronising copy.  d_len[i] <= stride,
 * d_off[] non-decreasing.  The kernels read an input 16 bytes at a time from its own first byte: up to 15 bytes past
 * an input's end are READ (never past the batch's last byte, base + n * stride or base + d_off[n]), and the bytes
 * between inputs (stride > len) must therefore be addressable.  Bad metadata is undefined behaviour, not EINVAL.
 * HIP GRAPHS: once a dfa is warm (one call of the same front and size class has run: lazily built tables and scratch
 * blocks exist) a *_device call allocates nothing and synchronises nothing, so it can be captured into a HIP graph on
 * the capturing stream and replayed on new bytes / metadata in the same buffers (tests/test_gpu_round4.py; small batches
 * are launch-bound: 13 us per replay against 23 us per launch at 4 096 inputs).  A captured launch keeps the slot of the
 * per-dfa rings (device-side kernel choice, tile counters: 64 slots) it was captured with: replay one graph at a time per
 * dfa, not concurrently with other launches on the same dfa.  fsm_hip_last_kernel_ms is meaningless for replays. */
int fsm_hip_exec_batch_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, size_t stride, const uint32_t *d_len, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream);

/* Device pointers, packed inputs with a device-resident offsets array. */
int fsm_hip_exec_batch_offsets_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, const uint64_t *d_off, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream);

/* Every output of one batch from ONE walk (device pointers, asynchronous): end states and / or the accept bitmap (as
 * fsm_hip_exec_batch_device), device-side end-ids (ids_mode = FSM_HIP_IDS_*, d_id_out: as fsm_hip_exec_batch_ids_device) and
 * eager-output sets (d_eager_out: as fsm_hip_exec_batch_eager_device), whichever pointers are not NULL.  Rows with a
 * stride (+ d_len) or packed with d_off (then stride is ignored and d_len must be NULL). */
int fsm_hip_exec_batch_all_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, size_t stride, const uint32_t *d_len, const uint64_t *d_off, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, int ids_mode, uint32_t *d_id_out, uint64_t *d_eager_out, void *hip_stream);

/* Compact metadata for packed inputs (round 4).  The generated matchers of the reference take a line as a (b, e)
 * pointer pair (src/libfsm/print/c.c:569-619) and retest / reperf hand lines over one by one (src/retest/main.c:1114,
 * src/retest/reperf.c:772-784): a batch of them is bytes packed back to back plus, per line, 8 bytes of u64 offsets
 * (above), or
 *   4 bytes: u32 offsets off32[n + 1], for batches below 4 GiB;
 *   4 bytes: the lengths alone, len[n] -- input i is the len[i] bytes after input i - 1.  The offsets are never
 *            materialised: a pre-pass leaves the byte offset of every 64th input (8 bytes per 64 inputs) and the walk
 *            kernels add a wavefront prefix sum of the 64 lengths they read anyway.
 * Results as for fsm_hip_exec_batch (end_out and/or the 1-bit-per-input accept_bitmap, either may be NULL).  The host
 * forms check their metadata (off32 non-decreasing); the device forms follow the *_device contract above.  A device
 * call of the lengths form uses a per-dfa scratch block that grows (a blocking allocation) the first time a batch
 * needs more than its predecessors. */
int fsm_hip_exec_batch_offsets32(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, const uint32_t *off32, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap);
int fsm_hip_exec_batch_offsets32_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, const uint32_t *d_off32, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream);
int fsm_hip_exec_batch_lengths(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap);
int fsm_hip_exec_batch_lengths_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, const uint32_t *d_len, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream);

/* Every output of one PACKED batch from one walk, whatever the metadata form: what the generated matchers with ids return
 * (`int fsm_main(const char *b, const char *e, unsigned *id)` and the ids / count form, src/libfsm/print/c.c:569-619) for a
 * batch of (b, e) lines, without widening their offsets first.  meta_form says what `meta` is: u64 offsets [n + 1], u32 offsets
 * [n + 1], or u32 lengths [n].  end_out / accept_bitmap / id_out (ids_mode = FSM_HIP_IDS_*) / eager_out as in
 * fsm_hip_exec_batch_all_device, whichever are not NULL. */
enum { FSM_HIP_META_OFF64 = 0, FSM_HIP_META_OFF32 = 1, FSM_HIP_META_LENGTHS = 2 };
int fsm_hip_exec_batch_packed_all(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, int meta_form, const void *meta, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap, int ids_mode, uint32_t *id_out, uint64_t *eager_out);
int fsm_hip_exec_batch_packed_all_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, int meta_form, const void *d_meta, size_t n,
	uint32_t *d_end_out, uint64_t *d_accept_bitmap, int ids_mode, uint32_t *d_id_out, uint64_t *d_eager_out, void *hip_stream);

/* Time of the most recent *_device launch on this dfa, measured with HIP
 * events recorded on the launch stream around the walk kernel only.
 * Blocks until that launch finished.  Returns milliseconds, or <0 on error. */
double fsm_hip_last_kernel_ms(const struct fsm_hip_dfa *dfa);

/* The walk kernel of the most recent launch on this dfa by its own (demangled) name, as rocprofv3 lists it -- e.g.
 * "fsmhip::walk_ldsdma<fsmhip::CombSelfPol, 128, 2, 768>"; two names separated by " | " when the choice between them was
 * made on the device.  Valid until the next launch on this dfa. */
const char *fsm_hip_last_kernel_name(const struct fsm_hip_dfa *dfa);

/* ------------------------------------------------------------------ */
/* end-ids (host side, by end state)                                  */
/* ------------------------------------------------------------------ */

/* cf. fsm_endid_count() src/libfsm/endids.c:653-684 */
size_t fsm_hip_endid_count(const struct fsm_hip_dfa *dfa, uint32_t end_state);

/* cf. fsm_endid_get() src/libfsm/endids.c:686-755: ids sorted ascending,
 * unique; returns 0 if id_buf_count is too small, 1 otherwise. */
int fsm_hip_endid_get(const struct fsm_hip_dfa *dfa, uint32_t end_state,
	size_t id_buf_count, uint32_t *id_buf);

/* ------------------------------------------------------------------ */
/* libfsm-facing shim                                                 */
/* ------------------------------------------------------------------ */

/* struct fsm * -> device table.  Checks ONCE what fsm_exec checks on every
 * call (fsm_all(fsm, fsm_isdfa) and fsm_getstart, src/libfsm/exec.c:106-114).
 * NULL + errno=EINVAL if not a DFA / no start; errno=ENOTSUP if the fsm uses
 * captures (fsm_countcaptures > 0: per-byte host callbacks); errno=ENOSYS if
 * libfsm's symbols are not present in the process.  Eager outputs are carried
 * into the table and delivered by fsm_hip_exec_batch_eager*(); the plain exec
 * calls ignore them. */
struct fsm_hip_dfa *fsm_hip_compile(const struct fsm *fsm, unsigned flags);

/* Same signature and result as fsm_exec() (include/fsm/fsm.h:560-562):
 * 1 and *end on match, 0 on no match (*end untouched), -1 + errno on error.
 * Drains fsm_getc to EOF (the reference stops pulling at the first missing
 * edge, src/libfsm/exec.c:133-138; the result is the same).  captures must
 * be NULL.  One input = one GPU launch: for plumbing/compat, not speed. */
int fsm_hip_exec(const struct fsm_hip_dfa *dfa,
	int (*fsm_getc)(void *opaque), void *opaque,
	fsm_state_t *end, struct fsm_capture *captures);

/* cf. fsm_vm_match_buffer() src/libfsm/vm.c:218-229: 1 / 0, -1 on error.  (1 MiB and more: the engine below.) */
int fsm_hip_match_buffer(const struct fsm_hip_dfa *dfa, const char *buf, size_t n);

/* cf. fsm_vm_match_file() src/libfsm/vm.c:188-216 (what re(1) -x runs on every file it is given, src/re/main.c:1106-1181):
 * 1 / 0, -1 + errno on error; a read error gives 0 as the reference's does.  A file of up to 256 KiB is one plain call.  A
 * bigger one is read in 32 MiB windows (two pinned buffers: window k + 1 is read while k is walked); a window crosses PCIe
 * once and is walked as 1 KiB pieces AT ONCE, one per lane, each from a guessed state (START), the guesses then corrected
 * from the previous piece's result until none changes -- at that fixed point every piece has started where the sequential
 * walk would have, so the result is exactly fsm_exec's over the whole file (file.hip).  A DFA built from patterns forgets
 * within a piece and the second pass already stands; an automaton that counts pays one pass per piece, the sequential cost.
 * Reading stops once the state can no longer change (DEAD or absorbing), as the VM's STOP does. */
int fsm_hip_match_file(const struct fsm_hip_dfa *dfa, FILE *f);
/* the same engine over memory; *end_state (optional) receives the caller's end state id or FSM_HIP_NO_MATCH */
int fsm_hip_match_buffer_big(const struct fsm_hip_dfa *dfa, const char *buf, size_t n, uint32_t *end_state);
/* how the last such call of this process went: windows walked and passes over them (2 per window where every guess stood
 * after its first correction; 0 / 0: a small input, one plain call) */
void fsm_hip_match_last_passes(unsigned *windows, unsigned *passes);

/* Flatten a struct fsm * into a malloc'd description (free with
 * fsm_hip_desc_free).  Exposed so callers can serialise the table. */
struct fsm_hip_dfa_desc *fsm_hip_flatten(const struct fsm *fsm);
void fsm_hip_desc_free(struct fsm_hip_dfa_desc *desc);

/* On-disk form of a flat description ("FSMHIP01", little-endian, layout in
 * libfsm_amd/csrc/shim.c): write returns 0 / -1; read returns a malloc'd
 * description (fsm_hip_desc_free) or NULL + errno=EINVAL on a bad or truncated
 * file.  Plays the role of the reference's DFAVM save/load
 * (src/libfsm/vm.c:39-71) for this path's executable form. */
int fsm_hip_desc_write(const struct fsm_hip_dfa_desc *desc, FILE *f);
struct fsm_hip_dfa_desc *fsm_hip_desc_read(FILE *f);

/* One more "language" in the shape of fsm_print()'s printers (src/libfsm/print.c:242-416):
 * flatten `fsm` and write its table in the on-disk form.  0, or -1 + errno. */
int fsm_hip_print(FILE *f, const struct fsm *fsm);

/* ------------------------------------------------------------------ */
/* streaming: inputs that arrive in pieces                            */
/* ------------------------------------------------------------------ */

/* The batched form of fsm_vm_match_file()'s chunk carry (struct vm_state kept
 * across 4 KiB fread chunks, src/libfsm/vm.c:188-216, src/libfsm/vm/vm.h:177-181):
 * input i starts from state_io[i] -- FSM_HIP_STATE_START, FSM_HIP_STATE_DEAD
 * (an earlier piece already hit a missing edge), or a state id of the caller's
 * fsm returned by a previous call -- consumes its bytes, and state_io[i]
 * receives the state reached.  end_out / accept_bitmap (optional) say whether
 * that state is an end state, i.e. what fsm_exec would return if the input
 * ended here. */
#define FSM_HIP_STATE_START 0xFFFFFFFDu
#define FSM_HIP_STATE_DEAD  0xFFFFFFFCu

/* 1 if no input byte can move the DFA out of `state` (a caller's state id, FSM_HIP_STATE_START or
 * FSM_HIP_STATE_DEAD) -- the condition under which the reference VM stops reading (STOP success on an
 * absorbing end state, src/libfsm/vm/ir.c:763-766; STOP fail on a missing edge); 0 otherwise;
 * -1 + errno=EINVAL for a bad id.  A streaming caller may stop feeding such an input. */
int fsm_hip_state_is_absorbing(const struct fsm_hip_dfa *dfa, uint32_t state);

int fsm_hip_exec_batch_resume(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	uint32_t *state_io, uint32_t *end_out);

int fsm_hip_exec_batch_resume_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, size_t stride, const uint32_t *d_len, size_t n,
	uint32_t *d_state_io, uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream);

/* the same over packed inputs (base + off[n + 1], as fsm_hip_exec_batch_offsets): the form retest / rx lines take
 * (src/retest/main.c:1114) when a line arrives in pieces */
int fsm_hip_exec_batch_resume_offsets(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, const uint64_t *off, size_t n,
	uint32_t *state_io, uint32_t *end_out);
int fsm_hip_exec_batch_resume_offsets_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, const uint64_t *d_off, size_t n,
	uint32_t *d_state_io, uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream);
/* ... and over the compact forms (meta_form = FSM_HIP_META_OFF64 / _OFF32 / _LENGTHS, as fsm_hip_exec_batch_packed_all): the carry
 * of fsm_vm_match_file (src/libfsm/vm.c:188-216) for batches whose metadata is u32 offsets or lengths alone (round 5) */
int fsm_hip_exec_batch_resume_packed(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, int meta_form, const void *meta, size_t n,
	uint32_t *state_io, uint32_t *end_out);
int fsm_hip_exec_batch_resume_packed_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, int meta_form, const void *d_meta, size_t n,
	uint32_t *d_state_io, uint32_t *d_end_out, uint64_t *d_accept_bitmap, void *hip_stream);

/* ------------------------------------------------------------------ */
/* end-ids delivered by the device (no host lookup per input)         */
/* ------------------------------------------------------------------ */

/* What id_out[i] holds, after enum fsm_ambig (include/fsm/options.h:35-43):
 *   FSM_HIP_IDS_EARLIEST  the lowest end-id of the end state (AMBIG_EARLIEST,
 *                         src/libfsm/print/c.c:67-85);
 *   FSM_HIP_IDS_RET       the index of the end state's id SET in the
 *                         de-duplicated, sorted list of sets (AMBIG_MULTIPLE; the
 *                         list is built like build_retlist, src/libfsm/vm/retlist.c:93-138:
 *                         ordered by count, then by memcmp of the id arrays, cmp_ret :63-79 --
 *                         so the index equals the reference's ret index); resolve it
 *                         with fsm_hip_ret_get().
 * Rejected inputs get FSM_HIP_NO_MATCH, accepted inputs whose end state carries
 * no id get FSM_HIP_NO_ID (EARLIEST) or the index of the empty set (RET). */
#define FSM_HIP_NO_ID 0xFFFFFFFEu
enum { FSM_HIP_IDS_EARLIEST = 1, FSM_HIP_IDS_RET = 2, FSM_HIP_IDS_ERROR = 3 };

/* AMBIG_ERROR (include/fsm/options.h:35-43): "emit a single endid", refusing DFAs in which some
 * end state carries more than one -- fsm_print fails with EINVAL there (src/libfsm/print/c.c:67-72)
 * after calling hook.conflict().  FSM_HIP_IDS_ERROR makes fsm_hip_exec_batch_ids*() fail the same
 * way (-1, errno = EINVAL, nothing launched); on a conflict-free DFA it is FSM_HIP_IDS_EARLIEST.
 * fsm_hip_ids_conflict() plays the hook: 1 and *state = the lowest such end state, 0 if none,
 * -1 + errno on error. */
int fsm_hip_ids_conflict(const struct fsm_hip_dfa *dfa, fsm_state_t *state);

int fsm_hip_exec_batch_ids(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	int mode, uint32_t *id_out);

int fsm_hip_exec_batch_ids_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, size_t stride, const uint32_t *d_len, size_t n,
	int mode, uint32_t *d_id_out, void *hip_stream);

/* the same over packed inputs: what a generated rx matcher hands its caller per line -- the id(s) of the pattern(s)
 * that matched (src/libfsm/print/c.c:569-619) -- for a whole file of lines in one launch */
int fsm_hip_exec_batch_ids_offsets(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, const uint64_t *off, size_t n,
	int mode, uint32_t *id_out);
int fsm_hip_exec_batch_ids_offsets_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, const uint64_t *d_off, size_t n,
	int mode, uint32_t *d_id_out, void *hip_stream);

size_t fsm_hip_ret_count(const struct fsm_hip_dfa *dfa);

/* Borrowed pointer to the sorted unique ids of set `ret_index` (valid until
 * fsm_hip_dfa_free).  0 on success, -1 + errno=EINVAL for a bad index. */
int fsm_hip_ret_get(const struct fsm_hip_dfa *dfa, uint32_t ret_index,
	const uint32_t **ids, size_t *count);

/* ------------------------------------------------------------------ */
/* eager outputs                                                      */
/* ------------------------------------------------------------------ */

/* fsm_exec with an eager-output callback installed (fsm_eager_output_set_cb,
 * include/fsm/fsm.h:311-312; exec.c:126-144) calls it with every output id of
 * the start state and of every state entered, also on inputs that finally do
 * not match.  The batch form returns, per input, the SET of ids emitted as a
 * bit set of W = fsm_hip_eager_words(dfa) 64-bit words (W = 1 for up to 64
 * distinct ids): bit k%64 of eager_out[i*W + k/64] <=> id fsm_hip_eager_id(dfa, k)
 * was emitted (ids numbered in ascending order).  eager_out holds n*W words.
 * end_out is as in fsm_hip_exec_batch.
 * The result is a SET: fsm_exec calls the callback once per id per state entered, in stream order, repeats
 * included (exec.c:126-144); these calls keep neither the order nor the multiplicity of the emissions (the
 * reference's own tests compare sets: tests/eager_output/utils.c:227-231).  A caller that needs the callback
 * stream itself uses fsm_hip_exec_batch_eager_trace below. */
int fsm_hip_exec_batch_eager(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *eager_out);

int fsm_hip_exec_batch_eager_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, size_t stride, const uint32_t *d_len, size_t n,
	uint32_t *d_end_out, uint64_t *d_eager_out, void *hip_stream);

/* the same over packed inputs */
int fsm_hip_exec_batch_eager_offsets(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, const uint64_t *off, size_t n,
	uint32_t *end_out, uint64_t *eager_out);
int fsm_hip_exec_batch_eager_offsets_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, const uint64_t *d_off, size_t n,
	uint32_t *d_end_out, uint64_t *d_eager_out, void *hip_stream);

/* The callback STREAM of fsm_exec (exec.c:120-144, match_eager_outputs_for_state :62-78), order and repeats kept:
 * for input i, count_out[i] = how many times the reference would have called the eager-output callback, and the
 * first min(count_out[i], cap) calls are recorded as ids_out[i*cap + k] = the id passed, pos_out[i*cap + k] = the
 * number of input bytes consumed when it fired (0: the start state's outputs, which fire before the first byte;
 * t + 1: the state entered on byte t).  Nothing fires after a missing edge.  count_out[i] > cap: the stream of
 * input i was cut at cap records -- call again with a larger cap.  Within ONE state's set the ids come in ascending
 * order (fsm_eager_output_get's order, eager_output.c:317-329); the reference's callback order inside one state is
 * the insertion order of its internal table (eager_output.c:264-266), which no public getter exposes.
 * Inputs: packed (off != NULL: n + 1 offsets, stride/len ignored) or stride (+ len, may be NULL).  end_out and
 * pos_out may be NULL.  This is an exact-semantics path over the plain table in device memory (uploaded on first
 * use: nstates * classes * 4 bytes), one input per lane; the set-valued calls above are the fast ones. */
int fsm_hip_exec_batch_eager_trace(const struct fsm_hip_dfa *dfa,
	const unsigned char *base, size_t stride, const uint32_t *len, const uint64_t *off, size_t n, size_t cap,
	uint32_t *end_out, uint32_t *count_out, uint32_t *ids_out, uint32_t *pos_out);
int fsm_hip_exec_batch_eager_trace_device(const struct fsm_hip_dfa *dfa,
	const void *d_base, size_t stride, const uint32_t *d_len, const uint64_t *d_off, size_t n, size_t cap,
	uint32_t *d_end_out, uint32_t *d_count_out, uint32_t *d_ids_out, uint32_t *d_pos_out, void *hip_stream);

size_t fsm_hip_eager_id_count(const struct fsm_hip_dfa *dfa);
size_t fsm_hip_eager_words(const struct fsm_hip_dfa *dfa);   /* ceil(id_count / 64), at least 1 */
uint32_t fsm_hip_eager_id(const struct fsm_hip_dfa *dfa, unsigned bit);

/* ------------------------------------------------------------------ */
/* many-DFA front: K automata x their own lines, ONE submission        */
/* ------------------------------------------------------------------ */

/* retest compiles a DFA per record and runs a handful of lines through it (src/retest/main.c:1056-1058
 * fsm_runner_initialize + fsm_free, :1114 fsm_runner_run; tests/retest/ *.tst: 37 DFAs x ~3 lines): one table upload and
 * one launch per DFA is all overhead.  fsm_hip_exec_multi takes K (dfa, lines, outputs) jobs at once.  Every small job
 * (<= 65 536 lines, <= 1 MiB of text, a plain table <= 1 MiB) rides in ONE host-to-device copy -- descriptors, the
 * automaton's plain next-state table, offsets, lines -- ONE kernel whose workgroups map to (dfa, tile of 64 lines), and
 * ONE copy back; a bigger job goes through its dfa's own walk (fsm_hip_exec_batch_offsets).  A dfa created with
 * FSM_HIP_DEFER_UPLOAD and used only here never uploads a table of its own.
 * Job q: input i is bytes off[i]..off[i+1] of base (as fsm_hip_exec_batch_offsets); end_out (n entries) and
 * accept_bitmap (ceil(n/64) words) are optional and get exactly what fsm_hip_exec_batch_offsets gives.
 * 0 / -1 + errno (EINVAL: NULL dfa, decreasing offsets; ENODEV; ENOMEM).  The dfas may live on several devices. */
struct fsm_hip_multi_batch {
	const unsigned char *base;
	const uint64_t *off;          /* n + 1 */
	size_t n;
	uint32_t *end_out;
	uint64_t *accept_bitmap;
};
int fsm_hip_exec_multi(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch *b, size_t k);
/* the same with DEVICE pointers inside b[] (the array itself is host memory); enqueued on hip_stream, not waited for.
 * Not capturable into a HIP graph (the descriptors ride in a staging block that the next call reuses): fsm_hip_multi_prepare
 * + fsm_hip_multi_launch below are. */
int fsm_hip_exec_multi_device(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch *b, size_t k, void *hip_stream);
/* ... with end-ids delivered by the device, per job (what the reference's multi-pattern consumers want beside accept / reject:
 * re(1) -z src/re/main.c:1152-1166, the generated matchers' `unsigned *id` src/libfsm/print/c.c:569-619): id_out (n entries,
 * optional) gets what fsm_hip_exec_batch_ids writes under ids_mode -- the lowest id of the end state (FSM_HIP_IDS_EARLIEST),
 * the index of its id set (FSM_HIP_IDS_RET, the dfa's own fsm_hip_ret_* tables), FSM_HIP_NO_ID for an end state without ids,
 * FSM_HIP_NO_MATCH for a rejected input; FSM_HIP_IDS_ERROR refuses the whole submission (EINVAL, nothing launched) when a dfa
 * that is asked for ids has an end state with more than one.  The id tables ride in the same copy as the automata's: a dfa
 * created with FSM_HIP_DEFER_UPLOAD still uploads nothing of its own.  Jobs without id_out cost what they cost above. */
struct fsm_hip_multi_batch_ids {
	const unsigned char *base;
	const uint64_t *off;          /* n + 1 */
	size_t n;
	uint32_t *end_out;
	uint64_t *accept_bitmap;
	uint32_t *id_out;
};
int fsm_hip_exec_multi_ids(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch_ids *b, size_t k, int ids_mode);
int fsm_hip_exec_multi_ids_device(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch_ids *b, size_t k, int ids_mode, void *hip_stream);
/* The device-pointer forms fuse every job whose plain next-state table fits the kernel's LDS copy (16 384 entries), whatever
 * its line count: workgroups of four wavefronts map to (dfa, 256 consecutive lines) and share one copy of that dfa's table --
 * 1 024 small automata x 1e5 lines each are ONE launch (round 5 sent every job above 65 536 lines through its dfa's own walk,
 * one launch each).  A job of few but very long lines is fused too (correct, one lane per line): hand such a job to its dfa's
 * own fronts.  Every dfa of a device-pointer submission is launched on the one hip_stream: they must live on that stream's
 * device (fsm_hip_node_exec_multi shards a submission by DFA over several). */
/* The PREPARED form of a device-pointer submission: descriptors, tile map and the automata's tables go to the device ONCE
 * (fsm_hip_multi_prepare: allocates, copies, waits); fsm_hip_multi_launch is then one kernel launch on hip_stream (plus one
 * per job whose table is too big to fuse) -- no copy, no allocation, no wait, so it can be captured into a HIP graph and
 * replayed on whatever the jobs' buffers hold by then (reperf runs the same matcher over and over, src/retest/reperf.c:772-784;
 * this is that loop for K matchers at once).  All dfas on one device (EINVAL otherwise); they and the buffers named in b[]
 * must outlive the handle; b[] itself is copied.  ids_mode as fsm_hip_exec_multi_ids (ignored when no job has id_out). */
struct fsm_hip_multi_prepared;
int fsm_hip_multi_prepare(const struct fsm_hip_dfa *const *dfa, const struct fsm_hip_multi_batch_ids *b, size_t k, int ids_mode,
	struct fsm_hip_multi_prepared **out);
int fsm_hip_multi_launch(const struct fsm_hip_multi_prepared *p, void *hip_stream);
void fsm_hip_multi_prepared_free(struct fsm_hip_multi_prepared *p);
/* kernels the last fsm_hip_exec_multi* / fsm_hip_multi_launch call of this process launched (1 when every job was small), and how many jobs rode
 * in the fused one */
unsigned fsm_hip_multi_last_launches(void);
unsigned fsm_hip_multi_last_fused_jobs(void);
/* Which device takes which job when a submission is sharded BY DFA over ndev devices: largest cost first,
 * each to the device with the least work so far (ties: the lower device; equal costs keep their order).  Pure host
 * arithmetic -- every rank of a multi-process run computes the same split.  dev_of[k] receives 0..ndev-1. */
int fsm_hip_multi_assign(const uint64_t *cost, size_t k, int ndev, int *dev_of);

/* ------------------------------------------------------------------ */
/* multi-device front: one table replica per GPU of the node           */
/* ------------------------------------------------------------------ */

/* The path shards by input: inputs are independent and the table is small enough
 * to replicate, so a node handle = one struct fsm_hip_dfa per device, and a batch of n inputs is split into
 * contiguous index shards of whole bitmap words -- shard k = inputs [k*per, min(n, (k+1)*per)) with
 * per = 64 * ceil(ceil(n/64) / ndev) -- each driven by its own host thread on its own device and stream.
 * This is what lets a C host (rx, retest, re(1)) use all 8 GPUs without torch.distributed.
 *
 * devices == NULL or ndev == 0: every device of the node.  The list may repeat a device (several replicas on
 * one GPU: a test rig).  NULL + errno as fsm_hip_dfa_create / fsm_hip_compile. */
struct fsm_hip_node;

struct fsm_hip_node *fsm_hip_node_create(const struct fsm_hip_dfa_desc *desc, unsigned flags, const int *devices, int ndev);
struct fsm_hip_node *fsm_hip_node_compile(const struct fsm *fsm, unsigned flags, const int *devices, int ndev);
void fsm_hip_node_free(struct fsm_hip_node *node);
int fsm_hip_node_ndev(const struct fsm_hip_node *node);
/* the k-th replica (borrowed: end-ids, ret sets, knobs, info are per replica and identical) */
struct fsm_hip_dfa *fsm_hip_node_dfa(struct fsm_hip_node *node, int k);
/* shard k of a batch of n inputs */
void fsm_hip_node_shard(const struct fsm_hip_node *node, size_t n, int k, size_t *first, size_t *count);
/* 1 if the device-resident exchange below runs over RCCL (distinct devices, librccl present), 0 if it is done
 * with peer-to-peer copies */
int fsm_hip_node_uses_rccl(const struct fsm_hip_node *node);
/* The shared object the node front's RCCL entry points were bound from (dlopen + dladdr; "" before the first node that
 * wanted RCCL, or when there is none): a process that has torch's RCCL bundle mapped binds that one, a plain C host
 * /opt/rocm/lib's. */
const char *fsm_hip_node_rccl_path(void);

/* fsm_hip_exec_batch / _offsets over the whole node: same arguments, same results.  Every device stages its
 * own slice from the caller's pages and copies its results straight into the caller's arrays at the shard's
 * offset -- no collective is needed for host-side results. */
int fsm_hip_node_exec_batch(struct fsm_hip_node *node,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap);
int fsm_hip_node_exec_batch_offsets(struct fsm_hip_node *node,
	const unsigned char *base, const uint64_t *off, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap);
/* ... and the compact packed forms of fsm_hip_exec_batch_offsets32 / _lengths, sharded the same way (round 5) */
int fsm_hip_node_exec_batch_offsets32(struct fsm_hip_node *node,
	const unsigned char *base, const uint32_t *off32, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap);
int fsm_hip_node_exec_batch_lengths(struct fsm_hip_node *node,
	const unsigned char *base, const uint32_t *len, size_t n,
	uint32_t *end_out, uint64_t *accept_bitmap);

/* Device-resident shards (inputs generated or loaded on the GPUs): d_base[k] = shard k's rows on device k,
 * uniform `stride`-byte inputs, 16-byte aligned.  d_end_out[k] (optional array, optional entries) receives
 * shard k's end states on device k.  d_bitmap_all[k] (optional) is a buffer of
 * fsm_hip_node_bitmap_words(node, n) words on device k: device k writes its slice at word k * (words / ndev)
 * and ONE in-place ncclAllGather over RCCL/xGMI (the path's only collective) leaves the whole batch's accept
 * bitmap on every device; *match_count (optional, needs d_bitmap_all) = accepted inputs of the whole batch,
 * one ncclAllReduce of a u64.  Returns after every device has finished. */
size_t fsm_hip_node_bitmap_words(const struct fsm_hip_node *node, size_t n);
int fsm_hip_node_exec_batch_device(struct fsm_hip_node *node,
	const void *const *d_base, size_t stride, size_t n,
	uint32_t *const *d_end_out, uint64_t *const *d_bitmap_all, uint64_t *match_count);

/* The general device-resident form: every per-device array is indexed by replica k and lives on device k; unused
 * members are NULL / 0 (memset the struct first).
 *   d_base[k]            shard k's input bytes;
 *   stride, d_len        fixed stride (+ optional lengths, d_len[k] = shard k's) -- or
 *   d_off[k]             shard k's packed offsets, count + 1 of them, relative to d_base[k] (stride ignored);
 *   d_end_out[k]         end states;  d_id_out[k] + ids_mode: end-ids as fsm_hip_exec_batch_ids;
 *   d_eager_out[k]       eager-output sets as fsm_hip_exec_batch_eager;
 *   d_bitmap_all[k]      the whole batch's accept bitmap on every device (all entries non-NULL), exchanged as above;
 *   want_count           also reduce the number of accepted inputs (read by fsm_hip_node_wait).
 * async = 0: returns after every device has finished, *match_count (optional) = accepted inputs.
 * async = 1: returns once the work is enqueued (match_count must be NULL).  The exchange runs on a second stream per
 * device, so the NEXT call's walk overlaps it -- provided that call uses other output buffers: alternate between two
 * sets.  fsm_hip_node_wait() returns when everything enqueued has finished (*match_count, optional: the last call's,
 * if it asked for one).  A failed call returns with nothing of it in flight. */
struct fsm_hip_node_batch {
	const void *const *d_base;
	size_t stride;
	const uint32_t *const *d_len;
	const uint64_t *const *d_off;
	uint32_t *const *d_end_out;
	uint32_t *const *d_id_out;
	int ids_mode;
	uint64_t *const *d_eager_out;
	uint64_t *const *d_bitmap_all;
	int want_count;
};
int fsm_hip_node_exec_device(struct fsm_hip_node *node, const struct fsm_hip_node_batch *batch, size_t n,
	uint64_t *match_count, int async);
int fsm_hip_node_wait(struct fsm_hip_node *node, uint64_t *match_count);

/* fsm_hip_exec_batch_ids / _eager over the whole node (host pointers; results land in the caller's arrays in place) */
int fsm_hip_node_exec_batch_ids(struct fsm_hip_node *node,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n, int mode, uint32_t *id_out);
int fsm_hip_node_exec_batch_eager(struct fsm_hip_node *node,
	const unsigned char *base, size_t stride, const uint32_t *len, size_t n, uint32_t *end_out, uint64_t *eager_out);

/* A many-DFA submission sharded BY DFA over the node's devices: nodes[q] holds job q's automaton (every node over the same
 * device list), fsm_hip_multi_assign(cost = text bytes + 64 per line) picks the device, each device runs ONE
 * fsm_hip_exec_multi over its jobs on its replicas (its own host thread), results land in the caller's arrays. */
int fsm_hip_node_exec_multi(struct fsm_hip_node *const *nodes, const struct fsm_hip_multi_batch *b, size_t k);

/* ------------------------------------------------------------------ */
/* synthetic input generator (benchmarks and parity tests)            */
/* ------------------------------------------------------------------ */

/* Counter-based generator, identical on host and device:
 *   r(i, t)    = (mix64(seed ^ (i * 0x9E3779B97F4A7C15) ^ (t >> 3)) >> (8 * (t & 7))) & 0xff
 *   byte(i, t) = alphabet ? alphabet[r % nalpha] : r
 * with mix64 the splitmix64 finaliser and i the GLOBAL input index
 * (first_index + local row).  If plant_len > 0 (<= 64), every input with
 * i % plant_every == 0 has `plant` copied at offset
 * mix64(seed ^ i ^ 0xA5A5A5A5A5A5A5A5) % (stride - plant_len + 1).
 * d_base is a device pointer to n rows of `stride` bytes. */
int fsm_hip_gen_inputs_device(void *d_base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *plant, unsigned plant_len, unsigned plant_every,
	void *hip_stream);

/* Lines out of rows, for the benchmarks of the variable-length fronts: input i = the first d_len[i] (<= max_len <= stride)
 * bytes of row i of d_rows, copied to d_out + d_off[i] (d_off[i] = sum of the lengths before i: the caller's prefix sum).
 * Device pointers, asynchronous on hip_stream. */
int fsm_hip_gen_pack_rows_device(const void *d_rows, size_t stride, const uint32_t *d_len, const uint64_t *d_off, size_t n,
	size_t max_len, void *d_out, void *hip_stream);

/* Host twin of the generator (same bytes). */
void fsm_hip_gen_inputs_host(unsigned char *base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *plant, unsigned plant_len, unsigned plant_every);

/* Affix variant for rx-style multi-pattern workloads: rows whose global index
 * is a multiple of `every` are  prefix + random bytes of `body` + suffix,
 * exactly stride bytes long (prefix = prefixes[h % npfx], suffix =
 * suffixes[(h>>32) % nsfx], h = mix64(seed ^ i ^ 0x5A5A5A5A5A5A5A5A)); every other
 * row is random over `alphabet` as above.  prefixes/suffixes are HOST arrays of
 * 8-byte entries [len<=7, b0..b6].  Synchronises the stream before returning. */
int fsm_hip_gen_affix_inputs_device(void *d_base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *body, unsigned nbody,
	const unsigned char *prefixes, unsigned npfx,
	const unsigned char *suffixes, unsigned nsfx,
	unsigned every, void *hip_stream);

void fsm_hip_gen_affix_inputs_host(unsigned char *base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *body, unsigned nbody,
	const unsigned char *prefixes, unsigned npfx,
	const unsigned char *suffixes, unsigned nsfx,
	unsigned every);

/* The same with an ALTERNATING body: positions after the prefix take bytes of `body` and `body2` in turn
 * (a transition-dense stream for patterns like ^ab([0-9][a-f])+(x|yz)$: every byte changes the state), and the
 * suffix is the first one, from the hashed choice on, that leaves a whole number of pairs.  nbody2 <= 64. */
int fsm_hip_gen_affix2_inputs_device(void *d_base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *body, unsigned nbody,
	const unsigned char *body2, unsigned nbody2,
	const unsigned char *prefixes, unsigned npfx,
	const unsigned char *suffixes, unsigned nsfx,
	unsigned every, void *hip_stream);

void fsm_hip_gen_affix2_inputs_host(unsigned char *base, size_t stride, size_t n,
	uint64_t first_index, uint64_t seed,
	const unsigned char *alphabet, unsigned nalpha,
	const unsigned char *body, unsigned nbody,
	const unsigned char *body2, unsigned nbody2,
	const unsigned char *prefixes, unsigned npfx,
	const unsigned char *suffixes, unsigned nsfx,
	unsigned every);

/* ------------------------------------------------------------------ */
/* literal sets: the Aho-Corasick caller of the path                   */
/* ------------------------------------------------------------------ */

/* Word list -> DFA, in the shape of libre's re_strings interface
 * (include/re/strings.h:15-55, src/libre/re_strings.c:21-137, src/libre/ac.c): the same trie,
 * failure edges, output propagation and state numbering (depth-first from the root in byte
 * order, ac.c:277-346; state 0 is the shared absorbing end state when neither ANCHOR_RIGHT nor
 * AC_AUTOMATON is given, re_strings.c:105-117), so the description equals
 * fsm_hip_flatten(re_strings_build(...)) state for state -- but built iteratively, straight into
 * the flat form, without a struct fsm (the reference recurses once per trie level and keeps 2 KiB
 * per trie node).  add_* return 1 / 0 like re_strings_add_*; build returns a malloc'd description
 * (fsm_hip_desc_free) or NULL + errno.  The builder may be reused after build. */
enum {
	FSM_HIP_STRINGS_ANCHOR_LEFT  = 1 << 0,   /* RE_STRINGS_ANCHOR_LEFT  */
	FSM_HIP_STRINGS_ANCHOR_RIGHT = 1 << 1,   /* RE_STRINGS_ANCHOR_RIGHT */
	FSM_HIP_STRINGS_AC_AUTOMATON = 1 << 2    /* RE_STRINGS_AC_AUTOMATON */
};

struct fsm_hip_strings;   /* plays struct re_strings */

struct fsm_hip_strings *fsm_hip_strings_new(void);
void fsm_hip_strings_free(struct fsm_hip_strings *g);
int fsm_hip_strings_add_raw(struct fsm_hip_strings *g, const void *p, size_t n, const fsm_end_id_t *endid);
int fsm_hip_strings_add_str(struct fsm_hip_strings *g, const char *s, const fsm_end_id_t *endid);
struct fsm_hip_dfa_desc *fsm_hip_strings_build(struct fsm_hip_strings *g, unsigned flags);

/* re_strings() in one call (re_strings.c:21-52): NUL-terminated words, no end-ids. */
struct fsm_hip_dfa_desc *fsm_hip_strings(const char *const a[], size_t n, unsigned flags);

/* Library/ABI version: major*10000 + minor*100 + patch. */
int fsm_hip_version(void);

#ifdef __cplusplus
}
#endif

#endif /* FSM_HIP_H */
