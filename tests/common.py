"""Shared helpers for the test-suite: golden fixture loading."""
import glob
import json
import os

import numpy as np

from libfsm_amd import FlatDfa, gen_inputs_host

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, path):
        self.path = path
        self.name = os.path.relpath(path, GOLDEN)
        z = np.load(path)
        self.flat = FlatDfa.load(z)
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.ret = z["ret"]
        self.end = z["end"]
        self.expect = z["expect"] if "expect" in z else None
        self.ids_off = z["ids_off"] if "ids_off" in z else None
        self.ids = z["ids"] if "ids" in z else None
        self.eager_out_off = z["eager_out_off"] if "eager_out_off" in z else None
        self.eager_out = z["eager_out"] if "eager_out" in z else None
        if "in_off" in z:                      # packed variable-length inputs
            self.base, self.off = z["in_bytes"], z["in_off"]
            self.rows = None
        elif "in_rows" in z:                   # fixed-stride rows stored verbatim
            self.rows = z["in_rows"]
            self.base = self.off = None
        else:                                  # rows defined by the generator parameters
            g = self.meta["gen"]
            self.rows = gen_inputs_host(g["n"], g["stride"], 0, g["seed"], None, g["plant"].encode(), g["plant_every"])
            assert int(self.rows.astype(np.uint64).sum()) == int(z["in_sum"]), "generator drifted from the golden inputs"
            self.base = self.off = None

    def strings(self):
        if self.off is not None:
            return [bytes(self.base[int(self.off[i]):int(self.off[i + 1])]) for i in range(len(self.off) - 1)]
        return [bytes(r) for r in self.rows]

    def packed(self):
        if self.off is not None:
            return self.base, self.off
        n, L = self.rows.shape
        return self.rows.reshape(-1), (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))

    def eager_of(self, i):
        return self.eager_out[int(self.eager_out_off[i]):int(self.eager_out_off[i + 1])]

    def padded_rows(self):
        """Packed inputs as fixed-stride rows + lengths (for the fixed-stride fronts)."""
        strs = self.strings()
        stride = max(16, (max([len(s) for s in strs] + [1]) + 15) // 16 * 16)
        rows = np.zeros((len(strs), stride), np.uint8)
        lens = np.array([len(s) for s in strs], np.uint32)
        for i, s in enumerate(strs):
            rows[i, :len(s)] = np.frombuffer(s, np.uint8)
        return rows, lens

    def ids_of(self, i):
        return self.ids[int(self.ids_off[i]):int(self.ids_off[i + 1])]


def all_golden_paths():
    return (sorted(glob.glob(os.path.join(GOLDEN, "retest", "*.npz"))) + sorted(glob.glob(os.path.join(GOLDEN, "eager", "*.npz")))
            + sorted(glob.glob(os.path.join(GOLDEN, "recorded", "*.npz")))
            + sorted(glob.glob(os.path.join(GOLDEN, "reperf", "*.npz")))
            + [p for p in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))) if not p.endswith("fsm_corpus.npz")])


def ac_golden_paths():
    return sorted(glob.glob(os.path.join(GOLDEN, "ac", "*.npz")))


def eager_golden_paths():
    return sorted(glob.glob(os.path.join(GOLDEN, "eager", "*.npz")))


def golden_id(p):
    return os.path.relpath(p, GOLDEN).replace(".npz", "")


class Corpus:
    """tests/golden/fsm_corpus.npz: the reference's checked-in out*.fsm automata with fsm_exec answers."""

    def __init__(self):
        self.z = np.load(os.path.join(GOLDEN, "fsm_corpus.npz"))
        self.names = bytes(self.z["names"]).decode().split("\n")

    def __len__(self):
        return len(self.names)

    def get(self, k):
        z = self.z
        from libfsm_amd.capi import RANGE_DTYPE
        r = np.zeros(len(z[f"d{k}_r_lo"]), dtype=RANGE_DTYPE)
        r["lo"], r["hi"], r["to"] = z[f"d{k}_r_lo"], z[f"d{k}_r_hi"], z[f"d{k}_r_to"]
        flat = FlatDfa(int(z[f"d{k}_nstates"]), int(z[f"d{k}_start"]), z[f"d{k}_edge_off"], r, z[f"d{k}_is_end"],
                       z[f"d{k}_endid_off"], z[f"d{k}_endids"])
        return flat, z[f"d{k}_in_bytes"], z[f"d{k}_in_off"], z[f"d{k}_ret"], z[f"d{k}_end"]


def retest_tst_lines():
    """The reference's tests/retest/*.tst (37 regexps, 115 +/- cases), re-emitted in retest's own file format
    (src/retest/main.c:738-1177: R dialect, M flags, regexp, +/- lines, blank line) from the frozen goldens
    tests/golden/retest/*.npz -- regex, dialect, flags, inputs and the fixture's expectations.
    Returns (lines, index of one '+' line that a test may flip)."""
    def esc(b):
        return "".join("\\\\" if c == 0x5C else chr(c) if 32 <= c < 127 else "\\x%02x" % c for c in b)

    letters = {1: "i", 2: "t", 4: "m", 8: "r", 16: "s", 32: "z", 64: "a", 128: "x"}
    lines, ncases, nre = ["# regenerated from tests/golden/retest/*.npz", "O +e"], 0, 0
    flip = None
    for path in [q for q in all_golden_paths() if "/retest/" in q]:
        g = Golden(path)
        regex = g.meta["regex"].encode("latin1").split(b"\0")[0]
        lines.append("R " + g.meta["dialect"])
        fl = "".join(v for k, v in letters.items() if g.meta["flags"] & k)
        if fl:
            lines.append("M " + fl)
        lines.append(("~" if regex[:1] in (b"#", b"~", b"R", b"O", b"M", b"+", b"-") or not regex else "") + esc(regex))
        if not regex:
            lines[-1] = "~"
        for inp, r in zip(g.strings(), g.ret):
            lines.append(("+" if r == 1 else "-") + esc(inp))
            if flip is None and r == 1:
                flip = len(lines) - 1
            ncases += 1
        lines.append("")
        nre += 1
    assert (nre, ncases) == (37, 115)
    return lines, flip


def reperf_scr_lines(count, flip=None):
    """reperf/boost.scr (5 string cases, R 1 each) re-emitted in reperf's script format (src/retest/reperf.c:338-560:
    '- name', M regexp, D dialect, S string, N runs, R expected matches, X = run) from the frozen goldens
    tests/golden/reperf/*.npz, with N = count.  flip = index of a case whose expectation is inverted (R 0)."""
    lines = ["# regenerated from tests/golden/reperf/*.npz"]
    paths = sorted(glob.glob(os.path.join(GOLDEN, "reperf", "*.npz")))
    assert len(paths) == 5
    for k, path in enumerate(paths):
        g = Golden(path)
        lines += ["", "- " + g.meta["name"], "M " + g.meta["regex"], "D " + g.meta["dialect"], "S " + g.strings()[0].decode("latin1"),
                  "N %d" % count, "R %d" % (g.meta["expected_matches"] if k != flip else 1 - g.meta["expected_matches"]), "X"]
    return lines
