#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerate the committed golden vectors.

Runs ONLY where /root/reference exists (the build container): it drives the
REAL reference (oracle/_ref, built by oracle/build_ref.sh from the reference's
own sources) and freezes its answers as small .npz fixtures:

  retest/NNN.npz   every regex block of the reference's tests/retest/*.tst
                   (37 regexes, 115 +/- cases): flattened DFA, the test inputs,
                   fsm_exec's return code and end state, and the fixture's own
                   +/- expectation (checked here to agree with fsm_exec).
  endids_*.npz     the constructions of tests/endids/endids2_union_many_endids.c
                   (6 patterns x 5 ids, union, determinise[, minimise]) with
                   fsm_exec + fsm_endid_get answers.
  re_strings_*.npz tests/re_strings/re_strings{1,2,3,4}.c word lists.
  c1.npz           BASELINE config 1/2 DFA: PCRE [Ll]ibf+(sm)* det+min, end-id 0,
                   with fsm_exec answers on generator inputs (parameters stored).
  c3.npz           BASELINE config 3: 1024 anchored PCRE unioned + determinised
  c3u.npz          its unanchored (rx-style) twin: 1024 patterns <letters>[0-9]$ with the implicit leading .*, a complete ~4100-state DFA
                   (rx-style, not minimised), end-id = pattern index, with
                   fsm_exec + end-id answers on 512 x 1 KiB inputs.

  ac/in<n><m>.npz  tests/aho_corasick: the regex side of the reference's re_strings == regex check
                   (4 anchor modes x 3 word lists) with fsm_exec on every short string.
  reperf/NNN.npz   reperf/boost.scr: the 5 string cases (regex DFA, string, fsm_exec answer = the script's R).
  recorded/*.npz   every fsm_exec call made by the reference's own tests/endids, tests/re_strings and
                   tests/capture programs (compiled in place with fsm_exec routed through
                   record_exec.c), grouped per distinct automaton.

Each fixture stores the flat DFA (libfsm_amd.FlatDfa.save) produced by the
product's own fsm_hip_flatten() from the reference `struct fsm *`.
"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from libfsm_amd import FlatDfa, gen_inputs_host  # noqa: E402
from oracle.pyoracle import RE_FLAGS, RefFsm, build_ref  # noqa: E402

REF = os.environ.get("FSM_REF", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def parse_escapes(s: bytes):
    """retest's escape grammar for test strings (src/retest/main.c:299-441):
    \\a \\b \\e \\f \\n \\r \\t \\v \\" \\\\, \\ooo (<=3 octal digits), \\xHH, \\x{HH}."""
    out = bytearray()
    i, n = 0, len(s)
    simple = {ord("a"): 7, ord("b"): 8, ord("e"): 27, ord("f"): 12, ord("n"): 10, ord("r"): 13, ord("t"): 9,
              ord("v"): 11, ord('"'): ord('"'), ord("\\"): ord("\\")}
    while i < n:
        c = s[i]
        if c != 0x5C:
            out.append(c)
            i += 1
            continue
        i += 1
        if i >= n:
            raise ValueError("dangling backslash")
        c = s[i]
        if c in simple:
            out.append(simple[c])
            i += 1
        elif ord("0") <= c <= ord("7"):
            v, nd = 0, 0
            while i < n and nd < 3 and ord("0") <= s[i] <= ord("7"):
                v = v * 8 + (s[i] - ord("0"))
                nd += 1
                i += 1
            out.append(v & 0xFF)
        elif c == ord("x"):
            i += 1
            curly = i < n and s[i] == ord("{")
            if curly:
                i += 1
            v, nd = 0, 0
            while i < n and nd < 2 and chr(s[i]) in "0123456789abcdefABCDEF":
                v = v * 16 + int(chr(s[i]), 16)
                nd += 1
                i += 1
            if curly:
                if i >= n or s[i] != ord("}"):
                    raise ValueError("incomplete \\x{")
                i += 1
            out.append(v & 0xFF)
        else:
            raise ValueError("invalid escape")
    return bytes(out)


def parse_tst(path):
    """Yield (line, dialect, flags, regex, [(line, expect_match, raw, bytes)...]) per regex block,
    following process_test_file (src/retest/main.c:738-1177)."""
    dialect, default = "pcre", "pcre"
    flags, opt_e = 0, False
    restore, saved_e = False, False  # "O &": main.c:861-865, restored at every blank line :832-834
    regex, cases, rline = None, [], 0
    with open(path, "rb") as f:
        for ln, raw in enumerate(f.read().split(b"\n"), 1):
            s = raw
            if len(s) == 0:
                if regex is not None:
                    yield rline, dialect_at, rflags, regex, cases
                regex, cases, flags = None, [], 0
                if restore:
                    opt_e = saved_e
                continue
            if s[:1] == b"#":
                continue
            if s[:1] == b"R" and (len(s) == 1 or s[1:2] == b" "):
                dialect = default if len(s) == 1 else s[2:].decode()
                continue
            if s[:2] == b"O ":
                if s[2:3] == b"&":
                    restore, saved_e = True, opt_e
                    continue
                arg = b"e" in s[3:]
                if s[2:3] == b"=":
                    opt_e = arg
                elif s[2:3] == b"+":
                    opt_e = opt_e or arg
                elif s[2:3] == b"-":
                    opt_e = opt_e and not arg
                continue
            if s[:2] == b"M ":
                for ch in s[2:].decode():
                    if ch == "0":
                        flags = 0
                    elif ch in RE_FLAGS:
                        flags |= RE_FLAGS[ch]
                continue
            if s[:1] == b"~":
                s = s[1:]
            if regex is None:
                regex = parse_escapes(s) if opt_e else s
                rline, dialect_at, rflags = ln, dialect, flags
            else:
                assert s[:1] in (b"+", b"-"), (path, ln, s)
                cases.append((ln, s[:1] == b"+", s[1:], parse_escapes(s[1:])))
    if regex is not None:
        yield rline, dialect_at, rflags, regex, cases


def pack(strings):
    off = np.zeros(len(strings) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in strings])
    return np.frombuffer(b"".join(strings), np.uint8), off


def endid_csr(fsm, ends):
    """fsm_endid_get for every matched end state, as CSR aligned with the inputs."""
    off, ids = [0], []
    for e in ends:
        if e != 0xFFFFFFFF:
            ids.extend(int(x) for x in fsm.endids(int(e)))
        off.append(len(ids))
    return np.array(off, np.uint32), np.array(ids, np.uint32)


def save_case(path, fsm, strings, meta, expect=None):
    flat = fsm.flatten()
    base, off = pack(strings)
    ret, end = fsm.exec_offsets(base, off)
    io, ii = endid_csr(fsm, end)
    extra = dict(in_bytes=base, in_off=off, ret=ret, end=end, ids_off=io, ids=ii,
                 meta=np.frombuffer(json.dumps(meta).encode(), np.uint8))
    if expect is not None:
        extra["expect"] = np.array(expect, np.int8)
        assert (ret == extra["expect"]).all(), (meta, ret, expect)
    flat.save(path, **extra)
    return flat, ret, end


def gen_retest():
    d = os.path.join(OUT, "retest")
    os.makedirs(d, exist_ok=True)
    k = nre = ncase = 0
    for fn in sorted(os.listdir(os.path.join(REF, "tests/retest"))):
        if not fn.endswith(".tst"):
            continue
        for rline, dialect, flags, regex, cases in parse_tst(os.path.join(REF, "tests/retest", fn)):
            nre += 1
            fsm = RefFsm.re_comp(dialect, regex.split(b"\0")[0], flags, True, True)
            meta = dict(file=f"tests/retest/{fn}", line=rline, dialect=dialect, flags=flags,
                        regex=regex.decode("latin1"), case_lines=[c[0] for c in cases])
            save_case(os.path.join(d, f"{k:03d}.npz"), fsm, [c[3] for c in cases], meta, [int(c[1]) for c in cases])
            ncase += len(cases)
            k += 1
    print(f"retest: {nre} regexes, {ncase} cases")
    assert (nre, ncase) == (37, 115), "SURVEY section 8c counts"


def gen_endids():
    # tests/endids/endids2_union_many_endids.c:25-33, :137-170
    patterns = [b"abc", b"def", b"abc.def", b"abc_def", b"foo", b"bar"]
    u = None
    for i, p in enumerate(patterns):
        f = RefFsm.re_comp("native", p, 0, True, True)
        for j in range(5):
            f.setendid(5 * i + j + 1)
        if u is None:
            u = f
        else:
            u.union_with(f)
    u.determinise()
    inputs = [b"abc", b"def", b"abcXdef", b"abc_def", b"foo", b"bar", b"", b"ab", b"xxabcyy", b"abc_def foo bar",
              b"foobar", b"barfoo", b"abcdef", b"de", b"fo\0o", b"abc\0def", b"zzz", b"abc.def", b"ABC", b"barabcfoo_def"]
    meta = dict(source="tests/endids/endids2_union_many_endids.c", patterns=[p.decode() for p in patterns], stage="determinised")
    save_case(os.path.join(OUT, "endids_union_det.npz"), u, inputs, meta)
    u.minimise()
    meta["stage"] = "minimised"
    save_case(os.path.join(OUT, "endids_union_min.npz"), u, inputs, meta)


def gen_re_strings():
    # tests/re_strings/re_strings{1,2,3,4}.c + testutil.c:15-70 (flags = 0, end-id = index)
    sets = {
        1: [b"aa", b"ab", b"ac", b"ba", b"bb", b"bc", b"ca", b"cb", b"cc"],
        2: [b"first", b"duplicate", b"duplicate", b"duplicate", b"last"],
        3: [b"duplicate", b"duplicate", b"duplicate"],
    }
    for k, words in sets.items():
        f = RefFsm.re_strings(words, 0, True)
        inputs = list(words) + [b"", b"x", b"xx" + words[0], words[-1] + b"yy", words[0] + words[-1], b"a", b"dup"]
        meta = dict(source=f"tests/re_strings/re_strings{k}.c", words=[w.decode() for w in words], flags=0)
        flat, ret, end = save_case(os.path.join(OUT, f"re_strings_{k}.npz"), f, inputs, meta)
        for i, w in enumerate(words):  # the reference test's own assertion (testutil.c:45-60)
            assert ret[i] == 1 and i in f.endids(int(end[i]))


C3_SEED = 20260923


def c3_patterns(n=1024, seed=C3_SEED):
    """1 024 patterns ^<2-3 lowercase>[0-9]+(x|yz)$.  Prefixes may repeat:
    two patterns with one prefix share their end states, which then carry two end-ids."""
    rng = random.Random(seed)
    pats = []
    while len(pats) < n:
        k = 2 if rng.random() < 0.62 else 3
        pre = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(k))
        pats.append(f"^{pre}[0-9]+(x|yz)$".encode())
    return pats


def c3_inputs(pats, n, L=1024, seed=C3_SEED + 1):
    """50 % derived from a pattern (prefix + digits + suffix, exactly L bytes), 50 % random over [a-z0-9]."""
    rng = np.random.RandomState(seed)
    alnum = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", np.uint8)
    digits = np.frombuffer(b"0123456789", np.uint8)
    out = np.empty((n, L), np.uint8)
    for i in range(n):
        if i % 2 == 0:
            p = pats[rng.randint(len(pats))]
            pre = p[1:p.index(b"[")]
            suf = b"x" if rng.randint(2) else b"yz"
            row = digits[rng.randint(0, 10, L)]
            row[:len(pre)] = np.frombuffer(pre, np.uint8)
            row[L - len(suf):] = np.frombuffer(suf, np.uint8)
            if rng.randint(8) == 0:  # a few near-misses: break one byte
                row[rng.randint(L)] = ord("!")
            out[i] = row
        else:
            out[i] = alnum[rng.randint(0, 36, L)]
    return out


def gen_c1():
    f = RefFsm.re_comp("pcre", b"[Ll]ibf+(sm)*", 0, True, True, endid=0)
    flat = f.flatten()
    assert flat.nstates == 5
    params = dict(n=4096, stride=256, seed=0x5EEDF5A1, plant="Libfsm", plant_every=8)
    data = gen_inputs_host(params["n"], params["stride"], 0, params["seed"], None, params["plant"].encode(), params["plant_every"])
    ret, end = f.exec_stride(data)
    flat.save(os.path.join(OUT, "c1.npz"), ret=ret, end=end, in_sum=np.uint64(int(data.astype(np.uint64).sum())),
              meta=np.frombuffer(json.dumps(dict(regex="[Ll]ibf+(sm)*", dialect="pcre", gen=params)).encode(), np.uint8))
    print("c1: accepts", int((ret == 1).sum()), "of", len(ret))


def gen_c3():
    pats = c3_patterns()
    f = RefFsm.union_res("pcre", pats, 0)
    flat = f.flatten()
    print("c3: states", flat.nstates, "end states", int(flat.is_end.sum()))
    data = c3_inputs(pats, 512)
    ret, end = f.exec_stride(data)
    io, ii = endid_csr(f, end)
    flat.save(os.path.join(OUT, "c3.npz"), in_rows=data, ret=ret, end=end, ids_off=io, ids=ii,
              patterns=np.frombuffer(b"\n".join(pats), np.uint8),
              meta=np.frombuffer(json.dumps(dict(source="BASELINE.json configs[2]", seed=C3_SEED, npatterns=len(pats))).encode(), np.uint8))
    print("c3: accepts", int((ret == 1).sum()), "of", len(ret), "multi-id ends", int(sum(1 for k in range(len(ret)) if io[k + 1] - io[k] > 1)))


def c3t_patterns(n=1024, seed=C3_SEED + 7):
    """The transition-dense twin of C3: 1 024 patterns ^<2-3 lowercase>([0-9][a-f])+(x|yz)$ -- a row that stays alive
    alternates between two states on every byte, so no 16-byte chunk is ever a run of self-loops."""
    rng = random.Random(seed)
    pats = []
    while len(pats) < n:
        k = 2 if rng.random() < 0.62 else 3
        pre = "".join(rng.choice("ghijklmnopqrstuvwxyz") for _ in range(k))   # prefixes avoid [a-f], the pair's second class
        pats.append(f"^{pre}([0-9][a-f])+(x|yz)$".encode())
    return pats


def gen_c3t():
    from libfsm_amd import gen_affix_inputs_host
    pats = c3t_patterns()
    f = RefFsm.union_res("pcre", pats, 0)
    flat = f.flatten()
    print("c3t: states", flat.nstates, "end states", int(flat.is_end.sum()))
    pf = [p[1:p.index(b"(")] for p in pats]
    data = gen_affix_inputs_host(512, 1024, 0, 0x5EEDF5A1, b"abcdefghijklmnopqrstuvwxyz0123456789", b"0123456789", pf, [b"x", b"yz"], 2, body2=b"abcdef")
    ret, end = f.exec_stride(data)
    io, ii = endid_csr(f, end)
    flat.save(os.path.join(OUT, "c3t.npz"), in_rows=data, ret=ret, end=end, ids_off=io, ids=ii,
              patterns=np.frombuffer(b"\n".join(pats), np.uint8),
              meta=np.frombuffer(json.dumps(dict(source="transition-dense twin of BASELINE.json configs[2]", seed=C3_SEED + 7, npatterns=len(pats))).encode(), np.uint8))
    print("c3t: accepts", int((ret == 1).sum()), "of", len(ret))


def c3u_patterns(n=1024, seed=C3_SEED + 13):
    """The UNANCHORED twin of C3, as rx builds a pattern list (src/rx/main.c:487-566 re_comp without ^, determinise, minimise,
    fsm_setendid(line); :1338-1385 fsm_union_array + fsm_determinise): 1 024 patterns <3-4 lowercase>[0-9]$ -- no left anchor, so
    every one carries the implicit leading .* -- whose union is a COMPLETE DFA of ~4 100 states: no DEAD default, no run of
    self-loops, a state change on almost every byte, end-id = pattern.  (Unanchored on the right as well, a pattern's accept state
    absorbs and the union has to remember which subset of the 1 024 has matched so far: it does not determinise in any budget.)"""
    rng = random.Random(seed)
    pats = []
    while len(pats) < n:
        k = rng.randint(3, 4)
        pats.append(("".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(k)) + "[0-9]$").encode())
    return pats


def c3u_suffixes(pats):
    """what a row must end with to be accepted by pattern i: its letters and one digit (i % 10)"""
    return [p[:p.index(b"[")] + str(i % 10).encode() for i, p in enumerate(pats)]


def gen_c3u():
    from libfsm_amd import gen_affix_inputs_host
    pats = c3u_patterns()
    f = RefFsm.union_res("pcre", pats, 0)
    flat = f.flatten()
    print("c3u: states", flat.nstates, "end states", int(flat.is_end.sum()))
    alnum = b"abcdefghijklmnopqrstuvwxyz0123456789"
    data = gen_affix_inputs_host(512, 1024, 0, 0x5EEDF5A1, alnum, alnum, [b""], c3u_suffixes(pats), 2)
    ret, end = f.exec_stride(data)
    io, ii = endid_csr(f, end)
    flat.save(os.path.join(OUT, "c3u.npz"), in_rows=data, ret=ret, end=end, ids_off=io, ids=ii,
              patterns=np.frombuffer(b"\n".join(pats), np.uint8),
              meta=np.frombuffer(json.dumps(dict(source="unanchored (rx-style) twin of BASELINE.json configs[2]", seed=C3_SEED + 13, npatterns=len(pats))).encode(), np.uint8))
    print("c3u: accepts", int((ret == 1).sum()), "of", len(ret), "multi-id ends", int(sum(1 for k in range(len(ret)) if io[k + 1] - io[k] > 1)))


def c_unescape(lit: str) -> bytes:
    """A C string literal body -> bytes (the escapes the reference's test sources use)."""
    out, i = bytearray(), 0
    simple = {"n": 10, "t": 9, "r": 13, "0": 0, "\\": 92, '"': 34, "'": 39, "a": 7, "b": 8, "f": 12, "v": 11}
    while i < len(lit):
        ch = lit[i]
        if ch != "\\":
            out += ch.encode("latin1")
            i += 1
            continue
        nx = lit[i + 1]
        if nx == "x":
            j = i + 2
            while j < len(lit) and lit[j] in "0123456789abcdefABCDEF":
                j += 1
            out.append(int(lit[i + 2:j], 16) & 0xFF)
            i = j
        elif nx in simple:
            out.append(simple[nx])
            i += 2
        else:
            out += nx.encode("latin1")
            i += 2
    return bytes(out)


def parse_eager_test(path):
    """Pull .patterns / .inputs out of a tests/eager_output/*.c program (struct eager_output_test,
    tests/eager_output/utils.h:28-39)."""
    import re
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    m = re.search(r"\.patterns\s*=\s*\{(.*?)\}\s*,\s*\.inputs", src, flags=re.S)
    if not m:
        return None
    STR = r'"((?:[^"\\]|\\.)*)"'
    pats = [c_unescape(x) for x in re.findall(STR, m.group(1))]
    body = src[m.end():]
    inputs = []
    MULTI = r'((?:"(?:[^"\\]|\\.)*"\s*)+)'          # adjacent literals concatenate
    for im in re.finditer(r"\{\s*\.input\s*=\s*" + MULTI, body):
        depth, pos = 1, im.end()
        while depth and pos < len(body):          # scan to the brace closing this entry
            ch = body[pos]
            if ch == '"':                          # skip string literals
                pos += 1
                while body[pos] != '"':
                    pos += 2 if body[pos] == "\\" else 1
            elif ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
            pos += 1
        rest = body[im.end():pos]
        ids = re.search(r"\.expected_ids\s*=\s*\{([^}]*)\}", rest)
        want = [int(x) for x in re.findall(r"\d+", ids.group(1))] if ids else []
        text = b"".join(c_unescape(x) for x in re.findall(STR, im.group(1)))
        inputs.append((text, "expect_fail" in rest, want))
    return pats, inputs


def gen_eager():
    """tests/eager_output/*.c: 22 programs; the combined DFA carries eager outputs (and sometimes
    end-ids).  Frozen: fsm_exec's return/end state, the end state's end-ids, and the ids the
    eager-output callback received; checked against each program's own expected_ids the way
    run_test() does (utils.c:170-251)."""
    d = os.path.join(OUT, "eager")
    os.makedirs(d, exist_ok=True)
    k = total = 0
    for fn in sorted(os.listdir(os.path.join(REF, "tests/eager_output"))):
        if not fn.startswith("eager_output") or not fn.endswith(".c"):
            continue
        parsed = parse_eager_test(os.path.join(REF, "tests/eager_output", fn))
        assert parsed, fn
        pats, inputs = parsed
        fsm = RefFsm.union_repeated("pcre", pats, 1, False)
        strings = [x[0] for x in inputs]
        ret, end, eager = fsm.exec_eager_strings(strings)
        flat = fsm.flatten()
        for i, (s_, fail, want) in enumerate(inputs):
            got = sorted(set(eager[i].tolist()) | set(fsm.endids(int(end[i])).tolist())) if ret[i] == 1 else []
            if fail or not want:
                assert ret[i] == 0 or not got, (fn, s_)
            else:
                assert ret[i] == 1 and got == want, (fn, s_, got, want)
        base, off = pack(strings)
        io, ii = endid_csr(fsm, end)
        eo = np.zeros(len(strings) + 1, np.uint32)
        eo[1:] = np.cumsum([len(e) for e in eager])
        flat.save(os.path.join(d, f"{k:03d}.npz"), in_bytes=base, in_off=off, ret=ret, end=end, ids_off=io, ids=ii,
                  eager_out_off=eo, eager_out=np.concatenate(eager).astype(np.uint32) if eo[-1] else np.zeros(0, np.uint32),
                  meta=np.frombuffer(json.dumps(dict(source=f"tests/eager_output/{fn}", patterns=[p.decode("latin1") for p in pats],
                                                     expected=[x[2] for x in inputs], expect_fail=[x[1] for x in inputs])).encode(), np.uint8))
        k += 1
        total += len(inputs)
    print(f"eager: {k} programs, {total} inputs")


def gen_fsm_corpus():
    """The reference's golden-DFA directories (tests/{pcre,pcre-anchor,pcre-classes,pcre-flags,
    pcre-repeat,native,glob,like,literal,sql,...}/out*.fsm): every checked-in expected automaton,
    parsed, determinised and minimised by the reference, with inputs from the reference's own
    fsm_generate_matches plus mutations and random strings, and fsm_exec's answers.  One container
    file: keys d<k>_<field>."""
    import glob as _glob
    rng = np.random.RandomState(77)
    store, k, ncase, skipped = {}, 0, 0, 0
    names = []
    for path in sorted(_glob.glob(os.path.join(REF, "tests", "*", "out*.fsm"))):
        if "/eclosure/" in path:  # not .fsm syntax (the reference's parser exits on it)
            continue
        f = RefFsm.parse_file(path)
        if f is None or f.nstates == 0 or f.nstates > 3000:
            skipped += 1
            continue
        try:
            flat = f.flatten()
        except OSError:
            skipped += 1          # no start state etc.: fsm_exec would return -1 as well
            continue
        pos = f.generate_matches(24, 8, seed=k + 1)
        strings = list(pos)
        for s_ in pos:            # mutations: truncate, extend, flip one byte
            if s_:
                strings.append(s_[:-1])
                t = bytearray(s_)
                t[rng.randint(len(t))] ^= 1 << rng.randint(8)
                strings.append(bytes(t))
            strings.append(s_ + bytes([rng.randint(256)]))
        used = sorted(set(int(r["lo"]) for r in flat.ranges) | {0x61, 0x0a, 0x00})
        for _ in range(8):
            strings.append(bytes(rng.choice(used, rng.randint(0, 12)).astype(np.uint8)))
        strings.append(b"")
        ret, end = f.exec_strings(strings)
        base, off = pack(strings)
        io, ii = endid_csr(f, end)
        rel = os.path.relpath(path, REF)
        for key, val in dict(nstates=np.uint32(flat.nstates), start=np.uint32(flat.start), edge_off=flat.edge_off,
                             r_lo=flat.ranges["lo"], r_hi=flat.ranges["hi"], r_to=flat.ranges["to"], is_end=flat.is_end,
                             endid_off=flat.endid_off, endids=flat.endids, in_bytes=base, in_off=off, ret=ret, end=end,
                             ids_off=io, ids=ii).items():
            store[f"d{k}_{key}"] = val
        names.append(rel)
        ncase += len(strings)
        k += 1
    store["names"] = np.frombuffer("\n".join(names).encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "fsm_corpus.npz"), **store)
    print(f"fsm corpus: {k} automata, {ncase} inputs, {skipped} skipped")


def gen_recorded():
    """Compile the reference's own C test programs (tests/endids, tests/re_strings, tests/capture)
    where they lie, with fsm_exec routed through record_exec.c, run them, and freeze every
    (automaton, input, fsm_exec answer, end-ids) they produce: recorded/<dir>_<program>_<k>.npz.
    Automata with capture paths flatten to ENOTSUP; those calls are frozen as `unsupported` counts."""
    import struct
    import subprocess
    import tempfile
    d = os.path.join(OUT, "recorded")
    os.makedirs(d, exist_ok=True)
    for old in os.listdir(d):
        os.unlink(os.path.join(d, old))
    refdir = os.path.join(ROOT, "oracle", "_ref")
    libdir = os.path.join(ROOT, "libfsm_amd")
    groups = {
        "endids": (["utils.c"], lambda fn: fn.startswith("endids") and fn.endswith(".c")),
        "re_strings": (["testutil.c"], lambda fn: fn.startswith("re_strings") and fn.endswith(".c")),
        "capture": (["captest.c"], lambda fn: fn.startswith("capture") and fn.endswith(".c")),
    }
    nprog = ncalls = nfsm = nunsup = 0
    summary = {}
    with tempfile.TemporaryDirectory() as tmp:
        for g, (common, pick) in groups.items():
            src = os.path.join(REF, "tests", g)
            for fn in sorted(os.listdir(src)):
                if not pick(fn):
                    continue
                exe, log = os.path.join(tmp, "prog"), os.path.join(tmp, "log")
                cmd = ["gcc", "-std=gnu99", "-O1", "-UNDEBUG", "-w", "-Dfsm_exec=rec_fsm_exec", f"-I{REF}/include", f"-I{REF}/src",
                       f"-I{REF}/src/adt", f"-I{ROOT}/include", os.path.join(src, fn)] + [os.path.join(src, c) for c in common] + [
                       os.path.join(OUT, "record_exec.c"), "-o", exe, f"-L{refdir}", "-lfsm_ref", f"-L{libdir}", "-lfsm_hip",
                       f"-Wl,-rpath,{refdir}", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
                subprocess.check_call(cmd)
                if os.path.exists(log):
                    os.unlink(log)
                r = subprocess.run([exe], env=dict(os.environ, REC_OUT=log), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                assert r.returncode == 0, (fn, r.returncode)      # the reference's own assertions hold
                nprog += 1
                raw = open(log, "rb").read() if os.path.exists(log) else b""
                pos, by_fsm, unsup = 0, {}, 0
                while pos < len(raw):
                    magic, blen, ferr = struct.unpack_from("<III", raw, pos)
                    assert magic == 0x43455852
                    pos += 12
                    blob = raw[pos:pos + blen]
                    pos += blen
                    (n,) = struct.unpack_from("<I", raw, pos)
                    data = raw[pos + 4:pos + 4 + n]
                    pos += 4 + n
                    ret, end, cnt = struct.unpack_from("<iII", raw, pos)
                    ids = struct.unpack_from(f"<{cnt}I", raw, pos + 12)
                    pos += 12 + 4 * cnt
                    ncalls += 1
                    if ferr:
                        assert ferr == 95, (fn, ferr)           # ENOTSUP: capture paths
                        unsup += 1
                        continue
                    by_fsm.setdefault(blob, []).append((data, ret, end, ids))
                nunsup += unsup
                summary[f"{g}/{fn}"] = dict(calls=sum(len(v) for v in by_fsm.values()) + unsup, automata=len(by_fsm), unsupported=unsup)
                for k, (blob, calls) in enumerate(by_fsm.items()):
                    tf = os.path.join(tmp, "t.fsmhip")
                    open(tf, "wb").write(blob)
                    flat = FlatDfa.read_c(tf)
                    seen, uniq = set(), []
                    for c in calls:                              # the same input twice must give the same answer
                        if c[0] in seen:
                            assert [u for u in uniq if u[0] == c[0]][0] == c
                            continue
                        seen.add(c[0])
                        uniq.append(c)
                    base, off = pack([c[0] for c in uniq])
                    io = np.zeros(len(uniq) + 1, np.uint32)
                    io[1:] = np.cumsum([len(c[3]) for c in uniq])
                    ii = np.array([x for c in uniq for x in c[3]], np.uint32)
                    meta = dict(source=f"tests/{g}/{fn}", recorded_by="tests/golden/record_exec.c", automaton=k)
                    flat.save(os.path.join(d, f"{g}_{fn[:-2]}_{k}.npz"), in_bytes=base, in_off=off,
                              ret=np.array([c[1] for c in uniq], np.int32), end=np.array([c[2] for c in uniq], np.uint32),
                              ids_off=io, ids=ii, meta=np.frombuffer(json.dumps(meta).encode(), np.uint8))
                    nfsm += 1
    json.dump(summary, open(os.path.join(d, "SUMMARY.json"), "w"), indent=1, sort_keys=True)
    print(f"recorded: {nprog} reference test programs, {ncalls} fsm_exec calls, {nfsm} automata, {nunsup} calls on capture automata (ENOTSUP)")


def gen_aho_corasick():
    """tests/aho_corasick: the reference checks that re_strings over in<n>.txt equals, as a language, the regex
    ^(w1|..)$ / ^(w1|..).* / .*(w1|..)$ / .*(w1|..).* (Makefile:28-92, modes a l r u).  Frozen here: the REGEX side
    (native dialect, determinised + minimised) with fsm_exec's answer on EVERY string up to a length bound over
    the words' letters plus one foreign letter; the tests hold the literal-set builder against it."""
    import itertools
    d = os.path.join(OUT, "ac")
    os.makedirs(d, exist_ok=True)
    forms = {"a": (b"^(", b")$", 3), "l": (b"^(", b").*", 1), "r": (b".*(", b")$", 2), "u": (b".*(", b").*", 0)}
    total = 0
    for k in (1, 2, 3):
        words = open(os.path.join(REF, "tests/aho_corasick", f"in{k}.txt"), "rb").read().split()
        alpha = sorted(set(b"".join(words))) + [ord("x")]
        maxlen = {1: 5, 2: 5, 3: 7}[k]
        strings = [bytes(t) for n in range(maxlen + 1) for t in itertools.product(alpha, repeat=n)]
        for mode, (pre, post, flags) in forms.items():
            regex = pre + b"|".join(words) + post
            f = RefFsm.re_comp("native", regex, 0, True, True)
            meta = dict(source=f"tests/aho_corasick/in{k}.txt", mode=mode, regex=regex.decode(), words=[w.decode() for w in words],
                        strings_flags=flags, maxlen=maxlen)
            save_case(os.path.join(d, f"in{k}{mode}.npz"), f, strings, meta)
            total += len(strings)
    print(f"aho_corasick: 12 automata, {total} inputs")


def gen_reperf():
    """reperf/boost.scr (format: src/retest/reperf.c:400-560): "- name", M regex, D dialect, S string (repeated S
    lines join with a newline), N iterations, R expected matches, X run.  The string cases are frozen: DFA,
    string, fsm_exec's answer, the script's own R.  (F cases name files that are not in the tree.)"""
    d = os.path.join(OUT, "reperf")
    os.makedirs(d, exist_ok=True)
    cur, last, k = {}, "", 0
    for ln, raw in enumerate(open(os.path.join(REF, "reperf", "boost.scr"), "rb").read().split(b"\n"), 1):
        if not raw or raw[:1] == b"#":
            continue
        op, arg = raw[:1], raw[1:]
        if op == b"-":
            cur = dict(name=arg.strip().decode(), line=ln)
        elif op == b"M":
            cur["regex"] = arg[1:] if arg[:1] == b" " else arg
        elif op == b"D":
            cur["dialect"] = arg.strip().decode()
        elif op == b"S":
            b_ = arg[1:] if arg[:1] == b" " else arg
            cur["string"] = b_ if last != b"S" else cur["string"] + b"\n" + b_
        elif op == b"F":
            cur.pop("string", None)
        elif op in (b"N", b"R"):
            cur["count" if op == b"N" else "expected"] = int(arg.strip())
        elif op == b"X" and "string" in cur:
            f = RefFsm.re_comp(cur["dialect"], cur["regex"], 0, True, True)
            meta = dict(source="reperf/boost.scr", line=cur["line"], name=cur["name"], dialect=cur["dialect"],
                        regex=cur["regex"].decode("latin1"), count=cur["count"], expected_matches=cur["expected"])
            save_case(os.path.join(d, f"{k:03d}.npz"), f, [cur["string"]], meta, [cur["expected"]])
            k += 1
        last = op
    print(f"reperf: {k} string cases")
    assert k == 5, "SURVEY section 8c count"


if __name__ == "__main__":
    assert build_ref(), "needs /root/reference to build oracle/_ref"
    gen_retest()
    gen_endids()
    gen_re_strings()
    gen_c1()
    gen_c3()
    gen_c3t()
    gen_c3u()
    gen_eager()
    gen_fsm_corpus()
    gen_recorded()
    gen_aho_corasick()
    gen_reperf()
