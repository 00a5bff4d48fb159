"""integration/retest/impl_hip.patch, control flow only (no GPU): the patched copy of the reference's retest
is run against a stand-in libfsm_hip.so (tests/c/stub_fsm_hip.c: the four entry points the patch calls,
answered by the reference's DFAVM and counted).  What is checked here is what the patch adds to main.c /
runner.c: `-l hip` (round 5) reads the WHOLE file ahead -- the main loop itself, run once with the runner collecting --
and matches every line of every record with ONE fsm_hip_exec_multi() call, the ordinary pass then gets its automata and
answers from the held results; `-l hip-record` (round 2's form) reads ahead to the end of each record and matches its
'+' / '-' lines with one fsm_hip_exec_batch_offsets() call; `-l hip-line` keeps one call per line.  The GPU suite (tests/test_gpu_round2.py::test_retest_l_hip) runs the same binary
against the real library."""
import os
import re
import subprocess

import pytest

from common import reperf_scr_lines, retest_tst_lines

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EXE = os.path.join(ROOT, "integration", "_build", "retest")


@pytest.fixture(scope="module")
def stub_dir(tmp_path_factory):
    if not os.path.isdir("/root/reference/src/retest") and not os.path.exists(EXE):
        pytest.skip("patched retest not built and no reference tree to build it from")
    sh = subprocess.run(["sh", os.path.join(ROOT, "integration", "retest", "build.sh")], capture_output=True, text=True)
    subprocess.run(["sh", os.path.join(ROOT, "integration", "re", "build.sh")], capture_output=True, text=True)
    subprocess.run(["sh", os.path.join(ROOT, "integration", "reftests", "build.sh")], capture_output=True, text=True)
    subprocess.run(["sh", os.path.join(ROOT, "integration", "fsm", "build.sh")], capture_output=True, text=True)
    if not os.path.exists(EXE):
        pytest.skip("integration/_build/retest not built: " + sh.stderr[-300:])
    d = tmp_path_factory.mktemp("stub")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-shared", "-fPIC", os.path.join(ROOT, "tests", "c", "stub_fsm_hip.c"),
                           "-Wl,-soname,libfsm_hip.so", "-o", str(d / "libfsm_hip.so")])
    return str(d)


def run(stub_dir, impl, path):
    env = dict(os.environ, LD_LIBRARY_PATH=stub_dir)      # searched before the binary's RUNPATH
    out = subprocess.run([EXE, "-l", impl, str(path)], capture_output=True, text=True, errors="replace", env=env, timeout=300)
    m = re.findall(r"stub_fsm_hip: compile=(\d+) batch_calls=(\d+) batch_inputs=(\d+) single_calls=(\d+) stride_calls=\d+ stride_inputs=\d+ multi_calls=(\d+) multi_jobs=(\d+) multi_inputs=(\d+)", out.stderr)
    assert m, out.stderr[-500:]
    # retest forks one child per file: one report
    run.multi = tuple(int(x) for x in m[-1][4:])
    return out, tuple(int(x) for x in m[-1][:4])


def test_record_lines_go_out_in_one_call(stub_dir, tmp_path):
    lines, flip = retest_tst_lines()
    tst = tmp_path / "all.tst"
    tst.write_bytes(("\n".join(lines) + "\n").encode("latin1"))
    records_with_cases = 0
    in_rec = False
    for i, l in enumerate(lines):
        if l == "":
            in_rec = False
        elif l[0] in "+-" and not in_rec:
            in_rec = True
            records_with_cases += 1

    # the whole file in ONE submission: 37 automata compiled once, one fsm_hip_exec_multi of 37 jobs / 115 lines, nothing else
    out, (ncomp, nbatch, nin, nsingle) = run(stub_dir, "hip", tst)
    tail = out.stdout.strip().splitlines()[-2:]
    assert out.returncode == 0, (out.stdout[-800:], out.stderr[-800:])
    assert tail[0].endswith("37 regexps, 115 test cases") and tail[1].endswith("0 re errors, 0 errors")
    assert out.stdout.count("[OK    ]") == 115 and "[NOT OK]" not in out.stdout
    assert (ncomp, nbatch, nin, nsingle) == (37, 0, 0, 0) and run.multi == (1, 37, 115)
    assert out.stdout.count("[BATCH ]") == 1 and "37 records, 115 test lines matched in 1 launch" in out.stdout
    # ... and what it prints is, but for that line, what the reference's own VM prints
    ref = subprocess.run([EXE, "-l", "vm", str(tst)], capture_output=True, text=True, errors="replace", timeout=300)
    strip_b = lambda s: [l for l in s.splitlines() if not l.startswith("[BATCH ]")]
    assert strip_b(out.stdout) == strip_b(ref.stdout)

    # a record per launch (round 2's form)
    out, (ncomp, nbatch, nin, nsingle) = run(stub_dir, "hip-record", tst)
    assert out.returncode == 0 and out.stdout.count("[OK    ]") == 115 and "[NOT OK]" not in out.stdout
    assert (ncomp, nbatch, nin, nsingle) == (37, records_with_cases, 115, 0) and run.multi == (0, 0, 0)
    assert out.stdout.count("[BATCH ]") == records_with_cases

    out, (ncomp, nbatch, nin, nsingle) = run(stub_dir, "hip-line", tst)
    assert out.returncode == 0 and out.stdout.count("[OK    ]") == 115
    assert (ncomp, nbatch, nin, nsingle) == (37, 0, 0, 115)

    # a wrong expectation is still reported, on its own line number
    lines[flip] = "-" + lines[flip][1:]
    bad = tmp_path / "bad.tst"
    bad.write_bytes(("\n".join(lines) + "\n").encode("latin1"))
    out, counts = run(stub_dir, "hip", bad)
    assert out.returncode == 1 and out.stdout.count("[NOT OK]") == 1
    assert "[NOT OK] line %d:" % (flip + 1) in out.stdout
    assert counts[3] == 0


def test_lines_the_read_ahead_did_not_see_fall_back(stub_dir, tmp_path):
    """Comment lines, a line with a bad escape (reported by the main loop, not matched), an 'M'-prefixed line inside
    a record (the main loop takes it as a flags line), an over-long line (fgets splits it the same way in both
    passes), an empty input, and a last record that ends at EOF without a blank line."""
    long_in = "a" * 5000
    text = "\n".join([
        "O +e",
        "R pcre",
        "^a*$",
        "# comment",
        "+",
        "+aaa",
        "-aab",
        "+\\x61\\x61",
        "+bad\\xZZ",
        "M i",
        "+" + long_in,
        "",
        "R literal",
        "abc",
        "+abc",
        "-abd",
    ]) + "\n"
    tst = tmp_path / "odd.tst"
    tst.write_text(text)
    out, (ncomp, nbatch, nin, nsingle) = run(stub_dir, "hip", tst)
    ref = subprocess.run([EXE, "-l", "vm", str(tst)], capture_output=True, text=True, errors="replace", timeout=300)
    strip = lambda s: [l for l in s.splitlines() if not l.startswith("[BATCH ]")]
    assert strip(out.stdout) == strip(ref.stdout)          # line for line what the reference's own VM reports
    assert out.returncode == ref.returncode
    assert ncomp == 2 and nbatch == 0 and nsingle == 0 and run.multi[:2] == (1, 2), (ncomp, nbatch, nin, nsingle, run.multi)
    out, (ncomp, nbatch, nin, nsingle) = run(stub_dir, "hip-record", tst)
    assert strip(out.stdout) == strip(ref.stdout) and out.returncode == ref.returncode
    assert ncomp == 2 and nbatch == 2 and nsingle == 0, (ncomp, nbatch, nin, nsingle)


def test_reperf_runs_go_out_as_batches_of_copies(stub_dir, tmp_path):
    """reperf(1) with the same patch: `-l hip` turns the N runs of a test (src/retest/reperf.c:772-784: N calls of
    fsm_runner_run on the same string) into fsm_runner_run_repeat() -- batches of copies through fsm_hip_exec_batch --
    `-l hip-line` keeps N calls; a wrong expectation (R 0 where the string matches) is reported."""
    exe = os.path.join(ROOT, "integration", "_build", "reperf")
    if not os.path.exists(exe):
        pytest.skip("integration/_build/reperf not built")
    env = dict(os.environ, LD_LIBRARY_PATH=stub_dir)

    def go(impl, lines):
        scr = tmp_path / "t.scr"
        scr.write_text("\n".join(lines) + "\n")
        out = subprocess.run([exe, "-C", "-l", impl, str(scr)], capture_output=True, text=True, errors="replace", env=env, timeout=300)
        m = re.findall(r"stub_fsm_hip: compile=(\d+) batch_calls=\d+ batch_inputs=\d+ single_calls=(\d+) stride_calls=(\d+) stride_inputs=(\d+)", out.stderr)
        assert m, out.stderr[-400:]
        return out, tuple(int(x) for x in m[-1])

    N = 3_000_000          # more than one block of 2^20 copies
    out, (ncomp, nsingle, nstride, ninputs) = go("hip", reperf_scr_lines(N))
    assert out.returncode == 0, out.stdout[-600:]
    assert out.stdout.count("execute %d iterations took" % N) == 5 and "ERROR" not in out.stdout.upper().replace("ERROR_NONE", "")
    assert (ncomp, nsingle, ninputs) == (5, 0, 5 * N) and nstride == 5 * 3
    out, (ncomp, nsingle, nstride, ninputs) = go("hip-line", reperf_scr_lines(2000))
    assert out.returncode == 0 and (ncomp, nsingle, nstride) == (5, 5 * 2000, 0)
    out, counts = go("hip", reperf_scr_lines(1000, flip=2))
    ref = subprocess.run([exe, "-C", "-l", "vm", str(tmp_path / "t.scr")], capture_output=True, text=True, errors="replace", timeout=300)
    strip = lambda t: [l for l in t.splitlines() if "iterations took" not in l]
    assert strip(out.stdout) == strip(ref.stdout) and out.returncode == ref.returncode     # the reference's VM reports the same failure
    assert "should not match" in (out.stdout + out.stderr).lower() or out.returncode != 0


RE_CASES = [
    (["-r", "pcre", "^ab+c$", "--", "abc", "abbc", "xyz", "ab", ""], 1),
    (["-r", "pcre", "^ab+c$", "--", "abc", "abbbbbc"], 0),
    (["-r", "pcre", "-z", "^ab+c$", "^x+$", "--", "abc", "xx"], 0),
    (["-r", "pcre", "-z", "^ab+c$", "^x+$", "--", "abc", "xx", "q"], 1),
    (["-r", "literal", "a.c", "--", "a.c"], 0),
    (["-r", "glob", "-i", "*.TXT", "--", "notes.txt", "x.txt.bak"], 1),
]


def test_re_H_matches_all_arguments_in_one_call(stub_dir, tmp_path):
    """re(1) with integration/re/hip_exec.patch: `re -H ... -- text...` answers like plain `re` (exit status, -z's
    pattern lines) from ONE fsm_hip_exec_batch_offsets call over all the arguments; -x files go through
    fsm_hip_match_file."""
    exe = os.path.join(ROOT, "integration", "_build", "re")
    if not os.path.exists(exe):
        pytest.skip("integration/_build/re not built")
    env = dict(os.environ, LD_LIBRARY_PATH=stub_dir)
    for args, rc in RE_CASES:
        ref = subprocess.run([exe] + args, capture_output=True, text=True, env=env, timeout=60)
        got = subprocess.run([exe, "-H"] + args, capture_output=True, text=True, env=env, timeout=60)
        assert ref.returncode == rc and got.returncode == rc, (args, ref.returncode, got.returncode, got.stderr[-300:])
        assert got.stdout == ref.stdout, args
        ntext = len(args) - args.index("--") - 1
        assert "stub_fsm_hip: compile=1 batch_calls=1 batch_inputs=%d single_calls=0" % ntext in got.stderr, got.stderr[-300:]
    f1, f2 = tmp_path / "a.txt", tmp_path / "b.txt"
    f1.write_bytes(b"abbbc")
    f2.write_bytes(b"abd" * 50000)
    for files, rc in (([f1], 0), ([f1, f2], 1)):
        args = ["-r", "pcre", "-x", "^ab+c$", "--"] + [str(f) for f in files]
        ref = subprocess.run([exe] + args, capture_output=True, text=True, env=env, timeout=60)
        got = subprocess.run([exe, "-H"] + args, capture_output=True, text=True, env=env, timeout=60)
        assert ref.returncode == rc == got.returncode
        assert "single_calls=%d" % len(files) in got.stderr
    bad = subprocess.run([exe, "-H", "-M", "-r", "pcre", "a", "--", "a"], capture_output=True, text=True, env=env)
    assert bad.returncode != 0 and "-H cannot be used" in bad.stderr


def reference_test_programs():
    d = os.path.join(ROOT, "integration", "_build", "reftests")
    return sorted(os.path.join(d, f) for f in os.listdir(d)) if os.path.isdir(d) else []


def test_reference_test_programs_route_fsm_exec(stub_dir):
    """integration/reftests: the reference's tests/endids (16) and tests/re_strings (4) programs, every fsm_exec() call
    routed through exec_via_hip.c.  Against the stand-in library this checks the harness only: all 20 exit 0 and report
    their calls as taken by the HIP entry points, none falling back.  (The 22 tests/eager_output programs need the real
    library's eager front: GPU suite.)"""
    progs = [p for p in reference_test_programs() if not os.path.basename(p).startswith("eager_")]   # the stand-in has no eager front
    if len(progs) != 20:
        pytest.skip("integration/_build/reftests not built")
    env = dict(os.environ, LD_LIBRARY_PATH=stub_dir)
    total = 0
    for exe in progs:
        out = subprocess.run([exe], capture_output=True, text=True, errors="replace", env=env, timeout=300)
        assert out.returncode == 0, (os.path.basename(exe), out.stdout[-300:], out.stderr[-300:])
        m = re.search(r"exec_via_hip: (\d+) fsm_exec calls answered by the HIP path, (\d+) fallbacks", out.stderr)
        if m is None:          # a program that builds and inspects automata without executing them (endids6)
            continue
        assert int(m.group(1)) > 0 and int(m.group(2)) == 0, (os.path.basename(exe), out.stderr[-300:])
        total += int(m.group(1))
    assert total > 300


FSM_DFA = '0 -> 1 "a";\n1 -> 1 "b";\n1 -> 2 "c";\n2 -> 2 "c";\nstart: 0;\nend: 2;\n'
FSM_NFA = '0 -> 1 "a";\n0 -> 2 "a";\nstart: 0;\nend: 2;\n'
FSM_CASES = [(["abc", "abbcc"], 0), (["abc", "abd"], 1), (["", "a"], 1), ([], 0)]


def test_fsm_H_matches_all_arguments_in_one_call(stub_dir, tmp_path):
    """fsm(1) with integration/fsm/hip_exec.patch: `fsm -H file.fsm text...` answers like plain `fsm` from one
    fsm_hip_exec_batch_offsets call; an NFA is refused where fsm_exec refuses it."""
    exe = os.path.join(ROOT, "integration", "_build", "fsm")
    if not os.path.exists(exe):
        pytest.skip("integration/_build/fsm not built")
    env = dict(os.environ, LD_LIBRARY_PATH=stub_dir)
    dfa, nfa = tmp_path / "d.fsm", tmp_path / "n.fsm"
    dfa.write_text(FSM_DFA)
    nfa.write_text(FSM_NFA)
    for texts, rc in FSM_CASES:
        ref = subprocess.run([exe, str(dfa)] + texts, capture_output=True, text=True, env=env, timeout=60)
        got = subprocess.run([exe, "-H", str(dfa)] + texts, capture_output=True, text=True, env=env, timeout=60)
        assert ref.returncode == rc == got.returncode and got.stdout == ref.stdout, (texts, got.stderr[-300:])
        if texts:
            assert "compile=1 batch_calls=1 batch_inputs=%d single_calls=0" % len(texts) in got.stderr
    ref = subprocess.run([exe, str(nfa), "a"], capture_output=True, text=True, env=env, timeout=60)
    got = subprocess.run([exe, "-H", str(nfa), "a"], capture_output=True, text=True, env=env, timeout=60)
    assert ref.returncode != 0 and got.returncode != 0 and "fsm_hip_compile" in got.stderr
