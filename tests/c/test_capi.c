/*
 * tests/c/test_capi.c -- the drop-in boundary exercised from plain C, written
 * the way the reference's own unit tests are (cf. match_string() in
 * tests/endids/utils.c:6-48 and run_test() in tests/re_strings/testutil.c:15-70):
 * build an fsm with libre/libfsm, run every input through BOTH fsm_exec() and
 * the HIP path (fsm_hip_compile + fsm_hip_exec / fsm_hip_exec_batch_offsets),
 * assert identical return codes, end states and end-id sets.
 *
 * Links the real reference (oracle/_ref/libfsm_ref.so) -- test code may -- and
 * libfsm_hip.so.  Prototypes of the reference's public API are restated here
 * (include/fsm/fsm.h, include/re/re.h) because its headers do not travel to
 * the GPU box.  Exit status 0 = PASS, like the reference's tests/ *.c programs.
 */
#include <assert.h>
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fsm_hip.h"

/* include/re/re.h:13-20, :137-140; include/fsm/fsm.h:65,199,223-228,473,503,560-562,579 */
enum re_dialect { RE_LIKE, RE_LITERAL, RE_GLOB, RE_NATIVE, RE_SQL, RE_PCRE };
struct re_err { int e; char buf[256]; };
struct fsm *re_comp(enum re_dialect, int (*)(void *), void *, const void *alloc, int flags, struct re_err *);
int fsm_sgetc(void *opaque);
int fsm_determinise(struct fsm *);
int fsm_minimise(struct fsm *);
int fsm_setendid(struct fsm *, fsm_end_id_t);
struct fsm *fsm_union(struct fsm *, struct fsm *, void *);
int fsm_exec(const struct fsm *, int (*)(void *), void *, fsm_state_t *, struct fsm_capture *);
size_t fsm_endid_count(const struct fsm *, fsm_state_t);
int fsm_endid_get(const struct fsm *, fsm_state_t, size_t, fsm_end_id_t *);
void fsm_free(struct fsm *);

static struct fsm *
compile(const char *re, fsm_end_id_t id)
{
	const char *s = re;
	struct fsm *fsm = re_comp(RE_PCRE, fsm_sgetc, &s, NULL, 0, NULL);
	assert(fsm != NULL);
	assert(fsm_determinise(fsm));
	assert(fsm_minimise(fsm));
	assert(fsm_setendid(fsm, id));
	return fsm;
}

int
main(void)
{
	static const char *patterns[] = { "^abc$", "^ab*c$", "^a.c$", "[Ll]ibf+(sm)*", "^x(yz)+$" };
	static const char *inputs[] = { "abc", "ac", "abbbc", "axc", "libfsm", "xxLibffsm", "xyzyz", "xy", "", "abcd", "zzz" };
	enum { NP = sizeof patterns / sizeof *patterns, NI = sizeof inputs / sizeof *inputs };
	struct fsm *fsm = NULL;
	struct fsm_hip_dfa *dfa;
	uint64_t off[NI + 1];
	unsigned char buf[1024];
	uint32_t end[NI];
	uint64_t bitmap[(NI + 63) / 64];
	size_t i, total = 0;
	int npass = 0;

	for (i = 0; i < NP; i++) {
		struct fsm *f = compile(patterns[i], (fsm_end_id_t) (100 + i));
		fsm = fsm == NULL ? f : fsm_union(fsm, f, NULL);
		assert(fsm != NULL);
	}
	assert(fsm_determinise(fsm)); /* rx-style: union + determinise, end-ids kept (src/rx/main.c:1338-1385) */

	dfa = fsm_hip_compile(fsm, 0);
	if (dfa == NULL) {
		perror("fsm_hip_compile");
		return EXIT_FAILURE;
	}

	/* one input at a time: fsm_hip_exec has fsm_exec's signature and contract */
	for (i = 0; i < NI; i++) {
		const char *s1 = inputs[i], *s2 = inputs[i];
		fsm_state_t e1 = 0xDEAD, e2 = 0xDEAD;
		int r1 = fsm_exec(fsm, fsm_sgetc, &s1, &e1, NULL);
		int r2 = fsm_hip_exec(dfa, fsm_sgetc, &s2, &e2, NULL);
		assert(r1 == r2);
		assert(e1 == e2); /* both untouched (0xDEAD) on reject */
		assert(fsm_hip_match_buffer(dfa, inputs[i], strlen(inputs[i])) == r1);
		off[i] = total;
		memcpy(buf + total, inputs[i], strlen(inputs[i]));
		total += strlen(inputs[i]);
		npass += r1;
	}
	off[NI] = total;

	/* the whole set as one batch: what retest's per-line loop becomes */
	assert(fsm_hip_exec_batch_offsets(dfa, buf, off, NI, end, bitmap) == 0);
	for (i = 0; i < NI; i++) {
		const char *s = inputs[i];
		fsm_state_t e = 0;
		int r = fsm_exec(fsm, fsm_sgetc, &s, &e, NULL);
		assert((end[i] != FSM_HIP_NO_MATCH) == (r == 1));
		assert(((bitmap[i / 64] >> (i % 64)) & 1) == (uint64_t) (r == 1));
		if (r == 1) {
			fsm_end_id_t a[16], b[16];
			size_t n = fsm_endid_count(fsm, e), k;
			assert(end[i] == e);
			assert(fsm_hip_endid_count(dfa, end[i]) == n && n <= 16);
			assert(fsm_endid_get(fsm, e, n, a) == 1);
			assert(fsm_hip_endid_get(dfa, end[i], n, b) == 1);
			assert(n == 0 || fsm_hip_endid_get(dfa, end[i], n - 1, b) == 0); /* 0 = buffer too small */
			for (k = 0; k < n; k++) {
				assert(a[k] == b[k]);
			}
		}
	}
	assert(npass >= 6);

	/* round 4: the same lines with u32 offsets, with their lengths alone, and every output from one walk (the `*id` the
	 * generated matchers return with AMBIG_EARLIEST, print/c.c:67-85: the lowest end-id of the end state) */
	{
		uint32_t off32[NI + 1], len[NI], e32[NI], el[NI], ea[NI], ids[NI];
		uint64_t bm2[(NI + 63) / 64];
		for (i = 0; i <= NI; i++) {
			off32[i] = (uint32_t) off[i];
		}
		for (i = 0; i < NI; i++) {
			len[i] = (uint32_t) (off[i + 1] - off[i]);
		}
		assert(fsm_hip_exec_batch_offsets32(dfa, buf, off32, NI, e32, NULL) == 0);
		assert(fsm_hip_exec_batch_lengths(dfa, buf, len, NI, el, bm2) == 0);
		assert(memcmp(end, e32, sizeof end) == 0 && memcmp(end, el, sizeof end) == 0 && bm2[0] == bitmap[0]);
		assert(fsm_hip_exec_batch_packed_all(dfa, buf, FSM_HIP_META_LENGTHS, len, NI, ea, NULL, FSM_HIP_IDS_EARLIEST, ids, NULL) == 0);
		assert(memcmp(end, ea, sizeof end) == 0);
		for (i = 0; i < NI; i++) {
			if (end[i] == FSM_HIP_NO_MATCH) {
				assert(ids[i] == FSM_HIP_NO_MATCH);
			} else {
				fsm_end_id_t a[16];
				size_t n = fsm_endid_count(fsm, end[i]);
				assert(n >= 1 && n <= 16 && fsm_endid_get(fsm, end[i], n, a) == 1);
				assert(ids[i] == a[0]); /* fsm_endid_get sorts ascending */
			}
		}
		errno = 0;
		assert(fsm_hip_exec_batch_packed_all(dfa, buf, 7, len, NI, ea, NULL, 0, NULL, NULL) == -1 && errno == EINVAL);
	}

	/* round 5: retest's shape -- a DFA per record and a few lines through each (src/retest/main.c:1056-1058, :1114) -- as ONE
	 * submission: every pattern its own automaton (no table upload of its own: FSM_HIP_DEFER_UPLOAD), every automaton all the
	 * lines, one fsm_hip_exec_multi; each answer against fsm_exec on that pattern's own fsm.  Then the same split over a node's
	 * devices by DFA, resume over the compact metadata forms, and the allocation-free promise of fsm_hip_reserve. */
	{
		struct fsm *pf[NP];
		struct fsm_hip_dfa *pd[NP];
		const struct fsm_hip_dfa *cpd[NP];
		struct fsm_hip_node *pn[NP];
		struct fsm_hip_multi_batch mb[NP];
		static uint32_t mend[NP][NI];
		static uint64_t mbm[NP][(NI + 63) / 64];
		static const int devs2[2] = { 0, 0 };
		uint64_t cost[4] = { 10, 1000, 10, 500 };
		int dev_of[4];
		size_t q;

		for (q = 0; q < NP; q++) {
			pf[q] = compile(patterns[q], (fsm_end_id_t) q);
			pd[q] = fsm_hip_compile(pf[q], FSM_HIP_DEFER_UPLOAD);
			assert(pd[q] != NULL);
			cpd[q] = pd[q];
			mb[q].base = buf;
			mb[q].off = off;
			mb[q].n = NI;
			mb[q].end_out = mend[q];
			mb[q].accept_bitmap = mbm[q];
		}
		assert(fsm_hip_exec_multi(cpd, mb, NP) == 0);
		assert(fsm_hip_multi_last_launches() == 1 && fsm_hip_multi_last_fused_jobs() == NP);
		for (q = 0; q < NP; q++) {
			for (i = 0; i < NI; i++) {
				const char *sq = inputs[i];
				fsm_state_t e = 0;
				int r = fsm_exec(pf[q], fsm_sgetc, &sq, &e, NULL);
				assert((mend[q][i] != FSM_HIP_NO_MATCH) == (r == 1));
				assert(r != 1 || mend[q][i] == e);
				assert(((mbm[q][i / 64] >> (i % 64)) & 1) == (uint64_t) (r == 1));
			}
		}
		/* sharded by DFA over two replicas per automaton */
		for (q = 0; q < NP; q++) {
			pn[q] = fsm_hip_node_compile(pf[q], FSM_HIP_DEFER_UPLOAD, devs2, 2);
			assert(pn[q] != NULL);
			memset(mend[q], 0x55, sizeof mend[q]);
		}
		{
			uint32_t keep[NP][NI];
			assert(fsm_hip_exec_multi(cpd, mb, NP) == 0);
			memcpy(keep, mend, sizeof keep);
			for (q = 0; q < NP; q++) memset(mend[q], 0x55, sizeof mend[q]);
			assert(fsm_hip_node_exec_multi(pn, mb, NP) == 0);
			assert(memcmp(keep, mend, sizeof keep) == 0);
		}
		assert(fsm_hip_multi_assign(cost, 4, 2, dev_of) == 0);
		assert(dev_of[1] == 0 && dev_of[3] == 1 && dev_of[0] == 1 && dev_of[2] == 1);   /* largest first, least loaded device, ties low */
		/* resume over the compact forms: every line in two pieces (its first byte, the rest), u32 offsets then lengths alone */
		{
			uint32_t o32[NI + 1], l2[NI], st[NI], e2[NI];
			unsigned char b1[NI + 1], b2[1024];
			size_t t2 = 0;
			o32[0] = 0;
			for (i = 0; i < NI; i++) {
				const size_t len = strlen(inputs[i]);
				b1[o32[i]] = len ? (unsigned char) inputs[i][0] : 0;
				o32[i + 1] = o32[i] + (len ? 1 : 0);
				l2[i] = len ? (uint32_t) (len - 1) : 0;
				memcpy(b2 + t2, inputs[i] + (len ? 1 : 0), l2[i]);
				t2 += l2[i];
				st[i] = FSM_HIP_STATE_START;
			}
			assert(fsm_hip_reserve(dfa, NI) == 0);
			assert(fsm_hip_exec_batch_resume_packed(dfa, b1, FSM_HIP_META_OFF32, o32, NI, st, NULL) == 0);
			assert(fsm_hip_exec_batch_resume_packed(dfa, b2, FSM_HIP_META_LENGTHS, l2, NI, st, e2) == 0);
			assert(memcmp(end, e2, sizeof end) == 0);
		}
		for (q = 0; q < NP; q++) {
			fsm_hip_node_free(pn[q]);
			fsm_hip_dfa_free(pd[q]);
			fsm_free(pf[q]);
		}
	}

	/* the node front: one replica per listed device (this rig lists its one GPU twice), the batch sharded
	 * over them, results in the caller's arrays: identical to the single-dfa batch */
	{
		static const int devs[2] = { 0, 0 };
		struct fsm_hip_node *node = fsm_hip_node_compile(fsm, 0, devs, 2);
		uint32_t end2[NI];
		uint64_t bitmap2[(NI + 63) / 64];
		size_t first, count;
		fsm_state_t conflict = 0;
		assert(node != NULL && fsm_hip_node_ndev(node) == 2);
		fsm_hip_node_shard(node, NI, 0, &first, &count);
		assert(first == 0 && count == NI);           /* fewer than 64 inputs: one shard holds them all */
		assert(fsm_hip_node_exec_batch_offsets(node, buf, off, NI, end2, bitmap2) == 0);
		assert(memcmp(end, end2, sizeof end) == 0 && bitmap[0] == bitmap2[0]);
		/* AMBIG_ERROR: "abc" is accepted by ^abc$, ^ab*c$ and ^a.c$ -- an end state with three ids */
		assert(fsm_hip_ids_conflict(fsm_hip_node_dfa(node, 1), &conflict) == 1);
		assert(fsm_hip_endid_count(dfa, conflict) > 1);
		fsm_hip_node_free(node);
	}

	fsm_free(fsm); /* the device table is self-contained (retest frees the fsm early, main.c:1056-1058) */
	assert(fsm_hip_match_buffer(dfa, "abc", 3) == 1);
	assert(fsm_hip_state_is_absorbing(dfa, FSM_HIP_STATE_DEAD) == 1);
	fsm_hip_dfa_free(dfa);
	printf("PASS %d/%d inputs matched, fsm_exec == fsm_hip on all\n", npass, (int) NI);
	return EXIT_SUCCESS;
}
