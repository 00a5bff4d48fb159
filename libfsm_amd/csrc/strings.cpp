/*
 * strings.cpp -- literal sets -> flat DFA description (host side, no GPU work).
 *
 * The Aho-Corasick caller of the path: what libre's re_strings interface builds as a struct fsm
 * (include/re/strings.h:15-55, src/libre/re_strings.c:21-137, src/libre/ac.c:141-346) is built
 * here directly as a struct fsm_hip_dfa_desc, state for state and id for id the same automaton:
 *
 *   trie              ac.c:141-181   one node per distinct prefix; the last node of a word is an
 *                                    "output" node and collects the word's end-id (a set);
 *   failure edges     ac.c:183-252   breadth-first, children in byte order; a node's output flag
 *                                    is OR-ed with that of the node the failure search stopped AT
 *                                    (`fs`, the parent of the failure target, ac.c:243) -- kept
 *                                    exactly so, results must equal the reference's;
 *   next state        ac.c:254-288   child, else follow failure links, else the root;
 *   numbering         ac.c:290-360   depth-first pre-order from the root, bytes ascending; output
 *                                    nodes without end-ids collapse into the one absorbing end
 *                                    state (state 0) unless ANCHOR_RIGHT / AC_AUTOMATON
 *                                    (re_strings.c:105-121), and are not expanded.
 *
 * Differences in HOW: no recursion (the reference recurses once per state on the C stack), the
 * trie keeps a hash of edges instead of 256 pointers per node, and next-state is a dense
 * [node][used byte] table filled in breadth-first order instead of a failure-chain walk per edge.
 */
#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <unordered_map>
#include <utility>
#include <vector>

#include "flat.h"

namespace {
const uint32_t NONE = 0xFFFFFFFFu;
}

struct fsm_hip_strings {
    std::unordered_map<uint64_t, uint32_t> edge;   /* (node << 8 | byte) -> child */
    std::vector<uint8_t> output;                   /* per node */
    std::vector<std::pair<uint32_t, uint32_t>> ids; /* (node, end-id) */
};
typedef struct fsm_hip_strings strings_t;   /* the one-call function of the same name hides the tag in C++ */

extern "C" struct fsm_hip_strings *fsm_hip_strings_new(void) {
    strings_t *g = new (std::nothrow) strings_t;
    if (g == nullptr) { errno = ENOMEM; return nullptr; }
    try {
        g->output.push_back(0);   /* the root, trie_create ac.c:114-139 */
    } catch (...) { delete g; errno = ENOMEM; return nullptr; }
    return g;
}

extern "C" void fsm_hip_strings_free(struct fsm_hip_strings *g) { delete g; }

extern "C" int fsm_hip_strings_add_raw(struct fsm_hip_strings *g, const void *p, size_t n, const fsm_end_id_t *endid) {
    if (g == nullptr || (p == nullptr && n > 0)) { errno = EINVAL; return 0; }
    try {
        const unsigned char *w = static_cast<const unsigned char *>(p);
        uint32_t st = 0;
        for (size_t i = 0; i < n; i++) {
            uint64_t key = (uint64_t) st << 8 | w[i];
            auto it = g->edge.find(key);
            if (it == g->edge.end()) {
                if (g->output.size() >= 0xFFFFFFF0u) { errno = ENOMEM; return 0; }
                uint32_t nx = (uint32_t) g->output.size();
                g->output.push_back(0);
                g->edge.emplace(key, nx);
                st = nx;
            } else {
                st = it->second;
            }
        }
        g->output[st] = 1;
        if (endid != nullptr) { g->ids.emplace_back(st, (uint32_t) *endid); }
    } catch (...) { errno = ENOMEM; return 0; }
    return 1;
}

extern "C" int fsm_hip_strings_add_str(struct fsm_hip_strings *g, const char *s, const fsm_end_id_t *endid) {
    if (s == nullptr) { errno = EINVAL; return 0; }
    return fsm_hip_strings_add_raw(g, s, strlen(s), endid);
}

static struct fsm_hip_dfa_desc *build(const strings_t *g, unsigned flags) {
    const bool left = (flags & FSM_HIP_STRINGS_ANCHOR_LEFT) != 0;
    const bool have_end = (flags & (FSM_HIP_STRINGS_AC_AUTOMATON | FSM_HIP_STRINGS_ANCHOR_RIGHT)) == 0;
    const uint32_t N = (uint32_t) g->output.size();

    /* children in byte order, CSR */
    std::vector<std::pair<uint64_t, uint32_t>> es(g->edge.begin(), g->edge.end());
    std::sort(es.begin(), es.end());
    std::vector<uint32_t> coff(N + 1, 0), cnode(es.size());
    std::vector<uint8_t> csym(es.size());
    int cls[256];
    uint32_t A = 0;
    {
        bool used[256] = {false};
        for (size_t i = 0; i < es.size(); i++) {
            coff[(es[i].first >> 8) + 1]++;
            csym[i] = (uint8_t) (es[i].first & 0xff);
            cnode[i] = es[i].second;
            used[csym[i]] = true;
        }
        for (uint32_t n = 0; n < N; n++) { coff[n + 1] += coff[n]; }
        for (int c = 0; c < 256; c++) { cls[c] = used[c] ? (int) A++ : -1; }
    }
    std::vector<std::pair<uint64_t, uint32_t>>().swap(es);

    /* end-id sets: sorted unique per node (state_set semantics, ac.c:176-178) */
    std::vector<std::pair<uint32_t, uint32_t>> ids(g->ids);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    std::vector<uint8_t> has_ids(N, 0);
    for (auto &pr : ids) { has_ids[pr.first] = 1; }

    std::vector<uint8_t> out(g->output);
    std::vector<uint32_t> delta;   /* [N][A] next node, unanchored-left only */
    if (!left) {
        if ((uint64_t) N * (A ? A : 1) > (uint64_t) 3 << 30) { errno = ENOMEM; return nullptr; }
        std::vector<uint32_t> fail(N, 0), q;
        q.reserve(N);
        delta.assign((size_t) N * A, 0);   /* row of the root: child, else the root itself */
        for (uint32_t i = coff[0]; i < coff[1]; i++) {
            fail[cnode[i]] = 0;
            q.push_back(cnode[i]);
            delta[cls[csym[i]]] = cnode[i];
        }
        for (size_t bot = 0; bot < q.size(); bot++) {
            const uint32_t st = q[bot];
            /* row(st) = row(fail(st)) overlaid with st's own children; fail(st) is shallower, so done */
            uint32_t *row = delta.data() + (size_t) st * A;
            memcpy(row, delta.data() + (size_t) fail[st] * A, (size_t) A * sizeof *row);
            for (uint32_t i = coff[st]; i < coff[st + 1]; i++) {
                const uint32_t nx = cnode[i];
                const int k = cls[csym[i]];
                /* the failure search of ac.c:229-241, from st's failure node; what the reference ORs
                 * into the output flag is that of the node the search stopped AT (`fs`) */
                uint32_t fs = fail[st], target = 0;
                for (;;) {
                    bool has = false;
                    for (uint32_t j = coff[fs]; j < coff[fs + 1]; j++) {
                        if (csym[j] == csym[i]) { has = true; target = cnode[j]; break; }
                    }
                    if (has || fs == 0) { break; }
                    fs = fail[fs];
                }
                fail[nx] = target;          /* fs->children[sym], or the root when there is none */
                out[nx] = out[nx] | out[fs];
                row[k] = nx;
                q.push_back(nx);
            }
        }
    }

    auto collapsed = [&](uint32_t n) { return out[n] && have_end && !has_ids[n]; };

    /* numbering: iterative pre-order, bytes ascending (trie_to_fsm_state ac.c:290-346) */
    const uint32_t first = have_end ? 1 : 0;
    std::vector<uint32_t> id_of(N, NONE), node_of;   /* node_of[id - first] */
    struct frame { uint32_t node; uint32_t pos; };
    std::vector<frame> stack;
    auto visit = [&](uint32_t n) -> uint32_t {
        if (collapsed(n)) { return 0; }
        if (id_of[n] == NONE) {
            id_of[n] = first + (uint32_t) node_of.size();
            node_of.push_back(n);
            stack.push_back(frame{n, 0});
        }
        return id_of[n];
    };
    const uint32_t start = visit(0);
    while (!stack.empty()) {
        frame &f = stack.back();
        if (left) {                       /* only real children exist */
            if (f.pos == coff[f.node + 1] - coff[f.node]) { stack.pop_back(); continue; }
            const uint32_t nx = cnode[coff[f.node] + f.pos++];
            visit(nx);
        } else {
            if (f.pos == 256) { stack.pop_back(); continue; }
            const int k = cls[f.pos++];
            visit(k < 0 ? 0u : delta[(size_t) f.node * A + k]);   /* bytes no word uses lead to the root */
        }
    }

    const uint32_t S = first + (uint32_t) node_of.size();
    struct flat *fl = static_cast<struct flat *>(calloc(1, sizeof *fl));
    if (fl == nullptr) { errno = ENOMEM; return nullptr; }
    std::vector<struct fsm_hip_range> ranges;
    fl->edge_off = static_cast<uint32_t *>(malloc(((size_t) S + 1) * sizeof *fl->edge_off));
    fl->is_end = static_cast<uint8_t *>(calloc(S ? S : 1, 1));
    fl->endid_off = static_cast<uint32_t *>(malloc(((size_t) S + 1) * sizeof *fl->endid_off));
    fl->endids = static_cast<uint32_t *>(malloc((ids.size() ? ids.size() : 1) * sizeof *fl->endids));
    if (!fl->edge_off || !fl->is_end || !fl->endid_off || !fl->endids) {
        fsm_hip_desc_free(&fl->d);
        errno = ENOMEM;
        return nullptr;
    }
    auto dst_id = [&](uint32_t n) { return collapsed(n) ? 0u : id_of[n]; };
    auto push = [&](unsigned lo, unsigned hi, uint32_t to) {
        struct fsm_hip_range r;
        r.lo = (uint8_t) lo; r.hi = (uint8_t) hi; r.reserved = 0; r.to = to;
        ranges.push_back(r);
    };
    size_t nid = 0;
    if (have_end) {                       /* re_strings.c:106-117: the end state loops on every byte */
        fl->edge_off[0] = 0;
        push(0, 255, 0);
        fl->is_end[0] = 1;
        fl->endid_off[0] = 0;
    }
    for (uint32_t s = first; s < S; s++) {
        const uint32_t n = node_of[s - first];
        fl->edge_off[s] = (uint32_t) ranges.size();
        if (left) {
            for (uint32_t i = coff[n]; i < coff[n + 1]; i++) {
                const uint32_t to = dst_id(cnode[i]);
                if (ranges.size() > fl->edge_off[s] && ranges.back().to == to && ranges.back().hi + 1u == csym[i]) {
                    ranges.back().hi = csym[i];
                } else {
                    push(csym[i], csym[i], to);
                }
            }
        } else {
            const uint32_t root_to = dst_id(0);
            const uint32_t *row = delta.data() + (size_t) n * A;
            for (unsigned c = 0; c < 256; c++) {
                const uint32_t to = cls[c] < 0 ? root_to : dst_id(row[cls[c]]);
                if (ranges.size() > fl->edge_off[s] && ranges.back().to == to) {
                    ranges.back().hi = (uint8_t) c;
                } else {
                    push(c, c, to);
                }
            }
        }
        fl->is_end[s] = out[n];
        fl->endid_off[s] = (uint32_t) nid;
        if (out[n] && has_ids[n]) {
            auto it = std::lower_bound(ids.begin(), ids.end(), std::make_pair(n, 0u));
            for (; it != ids.end() && it->first == n; ++it) { fl->endids[nid++] = it->second; }
        }
    }
    fl->edge_off[S] = (uint32_t) ranges.size();
    fl->endid_off[S] = (uint32_t) nid;
    fl->ranges = static_cast<struct fsm_hip_range *>(malloc((ranges.size() ? ranges.size() : 1) * sizeof *fl->ranges));
    if (fl->ranges == nullptr) { fsm_hip_desc_free(&fl->d); errno = ENOMEM; return nullptr; }
    if (!ranges.empty()) { memcpy(fl->ranges, ranges.data(), ranges.size() * sizeof *fl->ranges); }

    fl->d.nstates = S;
    fl->d.start = start;
    fl->d.edge_off = fl->edge_off;
    fl->d.ranges = fl->ranges;
    fl->d.is_end = fl->is_end;
    fl->d.endid_off = fl->endid_off;
    fl->d.endids = fl->endids;
    fl->d.eager_off = nullptr;
    fl->d.eager_ids = nullptr;
    return &fl->d;
}

extern "C" struct fsm_hip_dfa_desc *fsm_hip_strings_build(struct fsm_hip_strings *g, unsigned flags) {
    if (g == nullptr || (flags & ~7u) != 0) { errno = EINVAL; return nullptr; }
    try {
        return build(g, flags);
    } catch (const std::bad_alloc &) {
        errno = ENOMEM;
        return nullptr;
    }
}

extern "C" struct fsm_hip_dfa_desc *fsm_hip_strings(const char *const a[], size_t n, unsigned flags) {
    struct fsm_hip_strings *g = fsm_hip_strings_new();
    struct fsm_hip_dfa_desc *d = nullptr;
    if (g == nullptr) { return nullptr; }
    for (size_t i = 0; i < n; i++) {
        if (!fsm_hip_strings_add_str(g, a[i], nullptr)) { fsm_hip_strings_free(g); return nullptr; }
    }
    d = fsm_hip_strings_build(g, flags);
    fsm_hip_strings_free(g);
    return d;
}
