#!/usr/bin/env python3
"""tests/tools/eager_tiny64_stress.py -- repeated launches of the eager walks whose automaton has 7..16 states
(64-bit transition columns, TinyPol<u64>) in the TINY layout, every input mode, 4 / 8 / 12 wavefronts per workgroup.

Two row sets alternate, so a result left over from the previous launch is a wrong one.  One build of the ragged
kernel gave an intermittent wrong result here (walk_kernels.h, note in TinyPol::next): this is the run that
showed it (61-86 bad launches of 2700) and that the shipped kernels pass.  REPS = launches per configuration.
Prints one line per (mode, waves) and exits non-zero if any launch was wrong."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(reps, modes=None, waves=(4, 8, 12), out=print, layouts=("TINY",), states=(7, 16), fronts=("eager",)):
    import libfsm_amd as hip
    from common import Golden, eager_golden_paths
    from oracle.pyoracle import Oracle
    hip.load_library()
    rng = np.random.RandomState(77)
    modes = modes or (hip.IN_LDSDMA, hip.IN_DIRECT, hip.IN_GENERIC, hip.IN_RAGGED)
    stats = {}
    for path in eager_golden_paths():
        g = Golden(path)
        if not states[0] <= g.flat.nstates + 1 <= states[1]:
            continue
        pats = [p.encode("latin1").strip(b"^$") for p in g.meta["patterns"]]
        alpha = np.frombuffer((" ".join(g.meta["patterns"]) + " xyz").encode("latin1"), np.uint8)
        sets_ = []
        for v in range(2):
            rows = alpha[rng.randint(0, len(alpha), (330, 256))]
            for i in range(v, 330, 3):
                p = pats[rng.randint(len(pats))]
                if 0 < len(p) <= 60 and not any(c in p for c in b"[]()*+?|\\."):
                    at = rng.randint(0, 256 - len(p))
                    rows[i, at:at + len(p)] = np.frombuffer(p, np.uint8)
            _, wend, wsets = Oracle(g.flat).exec_eager(rows)
            sets_.append((rows, wend, wsets, Oracle(g.flat).exec_stride(rows)[1]))
        for lname in layouts:
            try:
                dfa = hip.HipDfa(g.flat, getattr(hip, "LAYOUT_" + lname))
            except OSError:
                continue
            for front in fronts:
                for mode in modes:
                    for w in waves:
                        dfa.tune(hip.KNOB_INPUT_MODE, mode)
                        dfa.tune(hip.KNOB_WAVES, w)
                        t = stats.setdefault((lname, front, mode, w), [0, 0, set()])
                        for rep in range(reps):
                            rows, wend, wsets, wplain = sets_[rep & 1]
                            if front == "eager":
                                end, sets = dfa.exec_batch_eager(rows)
                                bad = [i for i in range(len(rows)) if end[i] != wend[i] or (os.environ.get("CHECK") != "end" and not np.array_equal(sets[i], wsets[i]))]
                            else:
                                end, _ = dfa.exec_batch(rows)
                                bad = list(np.nonzero(end != wplain)[0])
                            t[0] += 1
                            if bad:
                                t[1] += 1
                                t[2].update(int(b) // 64 for b in bad)
            dfa.close()
    total_bad = 0
    for (lname, front, mode, w), t in sorted(stats.items()):
        out(f"{lname:8s} {front:5s} input mode {mode}, {w:2d} wavefronts: {t[0]} launches, {t[1]} wrong (bitmap words {sorted(t[2])})")
        total_bad += t[1]
    return sum(t[0] for t in stats.values()), total_bad


def run_packed(reps, out=print):
    """The ragged kernel on packed inputs, plain walks: two different batches (uniform 0..300-byte inputs, some
    accepted) alternate on every layout that holds the C1 / C3 automaton; every launch is compared with the oracle."""
    import libfsm_amd as hip
    from common import GOLDEN, Golden
    from oracle.pyoracle import Oracle
    hip.load_library()
    from common import eager_golden_paths
    total = bad_total = 0
    sources = [("c1.npz", b"Llibfsmx\0", None), ("c3.npz", b"abcdwxyz0123456789", None)]
    # 7..16-state automata (64-bit transition columns: TinyPol<u64>, whose step is inline asm) under ragged lengths
    for path in eager_golden_paths():
        g = Golden(path)
        if 7 <= g.flat.nstates + 1 <= 16 and len(sources) < 8:
            sources.append((os.path.basename(path), (" ".join(g.meta["patterns"]) + " xyz").encode("latin1"), g))
    for name, alpha, gg in sources:
        g = gg if gg is not None else Golden(os.path.join(GOLDEN, name))
        o = Oracle(g.flat)
        rng = np.random.RandomState(len(alpha))
        a = np.frombuffer(alpha, np.uint8)
        pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n") if name == "c3.npz" else None

        lits = None
        if gg is not None:
            lits = [p.encode("latin1").strip(b"^$") for p in gg.meta["patterns"]]
            lits = [p for p in lits if 0 < len(p) <= 60 and not any(c in p for c in b"[]()*+?|\\.")]

        def one(k):
            if lits and k >= 8 and rng.randint(3) == 0:
                s = bytearray(a[rng.randint(0, len(a), max(k, 64))])
                p = lits[rng.randint(len(lits))]
                at = rng.randint(0, len(s) - len(p) + 1)
                s[at:at + len(p)] = p
                return bytes(s)
            if gg is not None:
                return bytes(a[rng.randint(0, len(a), k)])
            if k >= 8 and rng.randint(3) == 0:
                if pats is None:
                    s = bytearray(a[rng.randint(0, len(a), k)])
                    at = rng.randint(0, k - 5)
                    s[at:at + 6] = b"Libfsm"
                    return bytes(s)
                p = pats[rng.randint(len(pats))]
                return p[1:p.index(b"[")] + bytes(rng.randint(48, 58, max(1, k - 6)).astype(np.uint8)) + b"yz"
            return bytes(a[rng.randint(0, len(a), k)])

        batches = []
        for v in range(2):
            strings = [one(rng.randint(0, 301)) for _ in range(700 + 37 * v)]
            off = np.zeros(len(strings) + 1, np.uint64)
            off[1:] = np.cumsum([len(s) for s in strings])
            batches.append((np.frombuffer(b"".join(strings), np.uint8), off, o.exec_strings(strings)[1]))
        for layout in hip.ALL_LAYOUTS:
            try:
                dfa = hip.HipDfa(g.flat, layout)
            except OSError:
                continue
            dfa.tune(hip.KNOB_INPUT_MODE, hip.IN_RAGGED)
            for w in (0, 6):
                dfa.tune(hip.KNOB_WAVES, w)
                n = bad = 0
                for rep in range(reps):
                    base, off, want = batches[rep & 1]
                    end, _ = dfa.exec_batch_offsets(base, off)
                    n += 1
                    bad += not np.array_equal(end, want)
                out(f"{name} {dfa.info()['layout_name']:8s} packed, ragged kernel, waves {w:2d}: {n} launches, {bad} wrong")
                total += n
                bad_total += bad
            dfa.close()
    return total, bad_total


if __name__ == "__main__":
    if os.environ.get("PACKED"):
        n, bad = run_packed(int(os.environ.get("REPS", 100)))
    elif os.environ.get("ALL_LAYOUTS"):      # every eager-capable layout, every automaton, eager and plain fronts
        n, bad = run(int(os.environ.get("REPS", 40)), modes=(1, 3), waves=(8, 12), states=(1, 1 << 30), fronts=("eager", "plain"),
                     layouts=("TINY", "COMBSELF", "COMB256", "LDSSELF", "LDS", "COMB", "SPARSE", "GLOBAL"))
    else:
        n, bad = run(int(os.environ.get("REPS", 100)))
    sys.exit(1 if bad else 0)

