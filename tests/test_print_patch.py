"""integration/print/print_hip.patch without a GPU: the reference's own rx(1) and re(1), rebuilt with FSM_PRINT_HIP in
fsm_print()'s language switch (src/libfsm/print.c:308-338, include/fsm/print.h, the -l tables of src/rx/main.c and
src/re/main.c), write a DFA in the FSMHIP on-disk form.  Printing needs no device: the real libfsm_hip.so flattens the
struct fsm through the public fsm(3) API.  The file is read back (fsm_hip_desc_read) and the oracle walks it: every
pattern's own example must come out with that pattern's id (rx numbers patterns by line), and `re -l hip` must describe
the automaton the reference builds in process.  The GPU suite feeds the same kind of file to examples/hipgrep
(tests/test_gpu_round3.py::test_rx_l_hip_into_hipgrep)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RX = os.path.join(ROOT, "integration", "_build", "print", "rx")
RE = os.path.join(ROOT, "integration", "_build", "print", "re")


@pytest.fixture(scope="module")
def printers(built):
    if not os.path.exists(RX) or os.path.isdir("/root/reference/src/rx"):
        subprocess.run(["sh", os.path.join(ROOT, "integration", "print", "build.sh")], capture_output=True, text=True)
    if not (os.path.exists(RX) and os.path.exists(RE)):
        if os.environ.get("FSM_REQUIRE_INTEGRATION"):
            pytest.fail("integration/_build/print/{rx,re} missing and FSM_REQUIRE_INTEGRATION is set")
        pytest.skip("integration/_build/print not built (needs /root/reference at build time)")
    return RX, RE


def test_rx_l_hip_table_carries_the_patterns_ids(printers, tmp_path):
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle
    rng = np.random.RandomState(2)
    words = sorted(set("".join(rng.choice(list("abcdefgh"), rng.randint(3, 7))) for _ in range(60)))
    pats = ["^%s[0-9]+$" % w for w in words] + ["^%s(x|yz)$" % w for w in words[:10]] + ["needle"]
    pf = tmp_path / "patterns"
    pf.write_text("\n".join(pats) + "\n")
    out = subprocess.run([printers[0], "-u", "-l", "hip", str(pf)], capture_output=True, timeout=300)
    assert out.returncode == 0 and out.stdout[:6] == b"FSMHIP", out.stderr[-500:]
    table = tmp_path / "t.fsmhip"
    table.write_bytes(out.stdout)
    flat = hip.FlatDfa.read_c(str(table))
    o = Oracle(flat)
    lines = [w.encode() + b"42" for w in words] + [w.encode() + b"yz" for w in words[:10]] + [b"needle", b"nothing", words[0].encode(), b""]
    ret, end = o.exec_strings(lines)
    assert list(ret[:len(words) + 11]) == [1] * (len(words) + 11) and list(ret[-3:]) == [0, 0, 0]
    for i in range(len(words) + 11):
        assert list(o.endids(int(end[i]))) == [i], (i, lines[i])
    # without -u / -t an ambiguous set is refused before anything is printed, as for every other language
    amb = tmp_path / "amb"
    amb.write_text("ab+c\nhello\n")
    out = subprocess.run([printers[0], "-l", "hip", str(amb)], capture_output=True, timeout=120)
    assert out.stdout == b"" and b"ambiguous" in out.stderr


def test_re_l_hip_is_the_automaton_the_reference_builds(printers, tmp_path):
    import libfsm_amd as hip
    from oracle import pyoracle
    if not pyoracle.have_ref():
        pytest.skip("oracle/_ref not built")
    for dialect, regex in (("pcre", b"^ab+c$"), ("pcre", b"[Ll]ibf+(sm)*"), ("glob", b"*.c"), ("literal", b"abc")):
        out = subprocess.run([printers[1], "-r", dialect, "-l", "hip", regex.decode()], capture_output=True, timeout=120)
        assert out.returncode == 0 and out.stdout[:6] == b"FSMHIP", (regex, out.stderr[-300:])
        table = tmp_path / "r.fsmhip"
        table.write_bytes(out.stdout)
        flat = hip.FlatDfa.read_c(str(table))
        f = pyoracle.RefFsm.re_comp(dialect, regex, 0, True, True)
        assert flat.nstates == f.nstates
        lines = [b"abc", b"abbbc", b"ac", b"libfsm", b"xLibffsmsm.", b"main.c", b"main.h", b"", b"abcabc"]
        ret, _ = pyoracle.Oracle(flat).exec_strings(lines)
        rret, _ = f.exec_strings(lines)
        assert np.array_equal(ret, rret), regex
    # an NFA cannot be flattened: the printer fails like any other that needs a DFA
    out = subprocess.run([printers[1], "-r", "pcre", "-n", "-l", "hip", "a|ab"], capture_output=True, timeout=120)
    assert out.returncode != 0 and out.stdout[:6] != b"FSMHIP"


def test_ir_flattening_equals_walk_edges_flattening(printers, tmp_path):
    """The IR consumer (integration/print/print_hip_ir.c: libfsm's codegen IR -- strategy, groups of ranges, mode, error
    ranges -- expanded the way the VM compiler's dfa_table does it, vm/ir.c:649-750) against the shim's flattening through
    fsm_walk_edges, on every DFA of the reference's regex-dialect corpus (tests/{pcre,...}/out*.fsm, determinised and
    minimised first where the fixture is an NFA) and on the 1 024-pattern union of configs[2]."""
    import glob
    check = os.path.join(os.path.dirname(printers[0]), "irflat_check")
    if not os.path.exists(check):
        pytest.skip("integration/_build/print/irflat_check not built")
    ref = "/root/reference/tests"
    files = sorted(f for d in ("pcre", "pcre-anchor", "pcre-flags", "pcre-repeat", "native", "glob", "like", "literal", "sql")
                   for f in glob.glob(os.path.join(ref, d, "out*.fsm")))
    if len(files) < 200:
        pytest.skip("the reference's fixture corpus is not on this machine")
    out = subprocess.run([check, "-d"] + files, capture_output=True, text=True, timeout=600)
    tail = out.stdout.strip().splitlines()[-1]
    nok, nbad, nskip = (int(x) for x in __import__("re").findall(r"\d+", tail))
    assert out.returncode == 0 and nbad == 0 and nok >= 240 and nok + nskip == len(files), tail
    # configs[2]: rx unions the 1 024 patterns and prints the DFA through the IR consumer; the table read back must accept
    # exactly what the golden automaton (flattened from the reference's own union by tests/golden/make_golden.py through
    # fsm_walk_edges) accepts, with the same end-id sets
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle
    from common import Golden, GOLDEN
    g = Golden(os.path.join(GOLDEN, "c3.npz"))
    pats = bytes(np.load(os.path.join(GOLDEN, "c3.npz"))["patterns"]).split(b"\n")
    pf = tmp_path / "patterns"
    pf.write_bytes(b"\n".join(pats) + b"\n")
    out = subprocess.run([printers[0], "-u", "-l", "hip", str(pf)], capture_output=True, timeout=600)
    assert out.returncode == 0 and out.stdout[:6] == b"FSMHIP", out.stderr[-300:]
    (tmp_path / "c3.fsmhip").write_bytes(out.stdout)
    flat = hip.FlatDfa.read_c(str(tmp_path / "c3.fsmhip"))
    assert abs(flat.nstates - g.flat.nstates) <= 2      # (rx keeps a state the golden's minimisation dropped)
    rng = np.random.RandomState(8)
    a = np.frombuffer(b"abcdwxyz0123456789", np.uint8)
    lines = []
    for i in range(6000):
        if i % 2:
            p = pats[rng.randint(len(pats))]
            lines.append(p[1:p.index(b"[")] + bytes(rng.randint(48, 58, rng.randint(1, 40)).astype(np.uint8)) + (b"x" if rng.randint(2) else b"yz"))
        else:
            lines.append(bytes(a[rng.randint(0, len(a), rng.randint(0, 50))]))
    o1, o2 = Oracle(flat), Oracle(g.flat)
    r1, e1 = o1.exec_strings(lines)
    r2, e2 = o2.exec_strings(lines)
    assert np.array_equal(r1, r2) and (r1 == 1).sum() > 2500
    for i in np.nonzero(r1 == 1)[0][:1500]:
        assert list(o1.endids(int(e1[i]))) == list(o2.endids(int(e2[i])))
