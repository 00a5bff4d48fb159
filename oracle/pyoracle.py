"""oracle/pyoracle.py -- TEST INFRASTRUCTURE, not product code.

ctypes bindings of
  * oracle/liboracle.so      -- the plain-C restatement (dfa_oracle.c), and
  * oracle/_ref/*.so         -- the REAL reference libfsm/libre + ref_helper.c.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REF_SO = os.path.join(REF_DIR, "libfsm_ref.so")
HELPER_SO = REF_SO  # ref_helper.c is linked into the same object

# enum re_dialect, include/re/re.h:13-20
DIALECTS = {"like": 0, "literal": 1, "glob": 2, "native": 3, "sql": 4, "pcre": 5}
# enum re_flags, include/re/re.h:22-37
RE_FLAGS = {"i": 1, "t": 2, "m": 4, "r": 8, "s": 16, "z": 32, "a": 64, "x": 128}
RE_STRINGS_ANCHOR_LEFT, RE_STRINGS_ANCHOR_RIGHT = 1, 2


def build_oracle(force: bool = False) -> str:
    src = os.path.join(HERE, "dfa_oracle.c")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", src, "-o", ORACLE_SO, "-lpthread"])
    return ORACLE_SO


def build_ref() -> bool:
    """(Re)build oracle/_ref from /root/reference when it exists; keep prebuilt files otherwise."""
    if os.path.isdir(os.environ.get("FSM_REF", "/root/reference")):
        subprocess.check_call(["sh", os.path.join(HERE, "build_ref.sh")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.exists(REF_SO) and os.path.exists(HELPER_SO)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """The plain-C restatement of fsm_exec over a flat DFA description."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL(build_oracle())
            vp, sz = C.c_void_p, C.c_size_t
            L.oracle_dfa_new.restype = vp
            L.oracle_dfa_new.argtypes = [C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp, vp, vp]
            L.oracle_dfa_free.argtypes = [vp]
            L.oracle_exec.argtypes = [vp, vp, sz, C.POINTER(C.c_uint32)]
            for f in ("oracle_exec_batch",):
                getattr(L, f).restype = C.c_double
                getattr(L, f).argtypes = [vp, vp, vp, sz, vp, vp]
            L.oracle_exec_batch_stride.restype = C.c_double
            L.oracle_exec_batch_stride.argtypes = [vp, vp, sz, vp, sz, vp, vp]
            L.oracle_table_walk_stride.restype = C.c_double
            L.oracle_table_walk_stride.argtypes = [vp, vp, sz, vp, sz, vp]
            L.oracle_state_walk_stride.restype = None
            L.oracle_state_walk_stride.argtypes = [vp, vp, sz, vp, sz, vp]
            L.oracle_isend.argtypes = [vp, C.c_uint32]
            L.oracle_exec_eager_stride.restype = None
            L.oracle_exec_eager_stride.argtypes = [vp, vp, vp, vp, sz, vp, sz, vp, vp, vp, vp, C.c_uint32]
            L.oracle_exec_eager_trace.restype = None
            L.oracle_exec_eager_trace.argtypes = [vp, vp, vp, vp, vp, sz, vp, sz, vp, vp, vp, vp, vp, C.c_uint32]
            L.oracle_endid_count.restype = sz
            L.oracle_endid_count.argtypes = [vp, C.c_uint32]
            L.oracle_endid_get.argtypes = [vp, C.c_uint32, sz, vp]
            cls._lib = L
        return cls._lib

    def __init__(self, flat, hasstart: bool = True):
        L = self.lib()
        self.flat = flat
        self._h = L.oracle_dfa_new(flat.nstates, flat.start, int(hasstart), _p(flat.edge_off),
                                   _p(flat.ranges) if len(flat.ranges) else None, _p(flat.is_end),
                                   _p(flat.endid_off), _p(flat.endids) if len(flat.endids) else None)
        if not self._h:
            raise MemoryError("oracle_dfa_new")
        self.last_seconds = 0.0

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib().oracle_dfa_free(self._h)
            self._h = None

    def exec_one(self, s: bytes):
        e = C.c_uint32(0xFFFFFFFF)
        b = np.frombuffer(s, dtype=np.uint8) if len(s) else np.zeros(1, np.uint8)
        r = self.lib().oracle_exec(self._h, _p(b), len(s), C.byref(e))
        return r, (e.value if r == 1 else 0xFFFFFFFF)

    def exec_offsets(self, base: np.ndarray, off: np.ndarray):
        base = np.ascontiguousarray(base, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        ret, end = np.zeros(n, np.int8), np.zeros(n, np.uint32)
        b = base if len(base) else np.zeros(1, np.uint8)
        self.last_seconds = self.lib().oracle_exec_batch(self._h, _p(b), _p(off), n, _p(ret), _p(end))
        return ret, end

    def exec_strings(self, strings):
        off = np.zeros(len(strings) + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in strings])
        return self.exec_offsets(np.frombuffer(b"".join(strings), np.uint8), off)

    def exec_stride(self, data: np.ndarray, lens=None):
        data = np.ascontiguousarray(data, np.uint8)
        n, stride = data.shape
        ret, end = np.zeros(n, np.int8), np.zeros(n, np.uint32)
        if lens is not None:
            lens = np.ascontiguousarray(lens, np.uint32)
        self.last_seconds = self.lib().oracle_exec_batch_stride(self._h, _p(data), stride, _p(lens), n, _p(ret), _p(end))
        return ret, end

    def table_walk(self, data: np.ndarray, lens=None):
        data = np.ascontiguousarray(data, np.uint8)
        n, stride = data.shape
        end = np.zeros(n, np.uint32)
        if lens is not None:
            lens = np.ascontiguousarray(lens, np.uint32)
        self.last_seconds = self.lib().oracle_table_walk_stride(self._h, _p(data), stride, _p(lens), n, _p(end))
        if self.last_seconds < 0:
            raise RuntimeError("oracle_table_walk_stride")
        return end

    def table_walk_mt(self, data: np.ndarray, nthreads: int, lens=None) -> np.ndarray:
        """table_walk on nthreads host threads (contiguous slices); whole rows, or the first lens[i] bytes of row i."""
        lib = self.lib()
        lib.oracle_table_walk_lens_mt.restype = C.c_double
        lib.oracle_table_walk_lens_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        data = np.ascontiguousarray(data, np.uint8)
        n, stride = data.shape
        if lens is not None:
            lens = np.ascontiguousarray(lens, np.uint32)
            assert len(lens) == n and (n == 0 or int(lens.max()) <= stride)
        end = np.zeros(n, np.uint32)
        self.last_seconds = lib.oracle_table_walk_lens_mt(self._h, _p(data), stride, _p(lens), n, _p(end), int(nthreads))
        if self.last_seconds < 0:
            raise RuntimeError("oracle_table_walk_lens_mt")
        return end

    def table_walk_packed_mt(self, base: np.ndarray, off: np.ndarray, nthreads: int = 1) -> np.ndarray:
        """the dense-table walk over inputs packed back to back (input i = base[off[i]:off[i+1]]), on nthreads host threads"""
        lib = self.lib()
        lib.oracle_table_walk_packed_mt.restype = C.c_double
        lib.oracle_table_walk_packed_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        base = np.ascontiguousarray(base, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        assert n >= 0 and (n == 0 or int(off[-1]) <= len(base))
        end = np.zeros(n, np.uint32)
        self.last_seconds = lib.oracle_table_walk_packed_mt(self._h, _p(base) if len(base) else None, _p(off), n, _p(end), int(nthreads))
        if self.last_seconds < 0:
            raise RuntimeError("oracle_table_walk_packed_mt")
        return end

    def state_walk(self, data: np.ndarray, state_io: np.ndarray, lens=None) -> np.ndarray:
        """Streaming walk: returns the states reached (0xFFFFFFFC = dead)."""
        data = np.ascontiguousarray(data, np.uint8)
        n, stride = data.shape
        st = np.ascontiguousarray(state_io, np.uint32).copy()
        if lens is not None:
            lens = np.ascontiguousarray(lens, np.uint32)
        self.lib().oracle_state_walk_stride(self._h, _p(data) if data.size else None, stride, _p(lens), n, _p(st))
        return st

    def exec_eager(self, data: np.ndarray, lens=None, cap: int = 64):
        """fsm_exec + eager outputs: (ret, end, [sorted emitted ids per input])."""
        flat = self.flat
        data = np.ascontiguousarray(data, np.uint8)
        n, stride = data.shape
        eo = flat.eager_off if flat.eager_off is not None else np.zeros(flat.nstates + 1, np.uint32)
        ei = flat.eager_ids if flat.eager_off is not None and len(flat.eager_ids) else np.zeros(1, np.uint32)
        ret, end = np.zeros(n, np.int8), np.zeros(n, np.uint32)
        ids, cnt = np.zeros((n, cap), np.uint32), np.zeros(n, np.uint32)
        if lens is not None:
            lens = np.ascontiguousarray(lens, np.uint32)
        self.lib().oracle_exec_eager_stride(self._h, _p(eo), _p(ei), _p(data) if data.size else None, stride, _p(lens), n,
                                            _p(ret), _p(end), _p(ids), _p(cnt), cap)
        return ret, end, [np.sort(ids[i, :cnt[i]]) for i in range(n)]

    def exec_eager_trace(self, data: np.ndarray, lens=None, off=None, cap: int = 64):
        """fsm_exec's eager-output callback stream: (ret, end, counts, [(ids, positions) in call order, repeats kept])."""
        flat = self.flat
        data = np.ascontiguousarray(data, np.uint8)
        if off is not None:
            off = np.ascontiguousarray(off, np.uint64)
            n, stride = len(off) - 1, 0
        else:
            n, stride = data.shape
        eo = flat.eager_off if flat.eager_off is not None else np.zeros(flat.nstates + 1, np.uint32)
        ei = flat.eager_ids if flat.eager_off is not None and len(flat.eager_ids) else np.zeros(1, np.uint32)
        ret, end = np.zeros(n, np.int8), np.zeros(n, np.uint32)
        ids, pos, cnt = np.zeros((n, cap), np.uint32), np.zeros((n, cap), np.uint32), np.zeros(n, np.uint32)
        if lens is not None:
            lens = np.ascontiguousarray(lens, np.uint32)
        self.lib().oracle_exec_eager_trace(self._h, _p(eo), _p(ei), _p(data) if data.size else None, _p(off), stride, _p(lens), n,
                                           _p(ret), _p(end), _p(ids), _p(pos), _p(cnt), cap)
        k = np.minimum(cnt, cap)
        return ret, end, cnt, [(ids[i, :k[i]].copy(), pos[i, :k[i]].copy()) for i in range(n)]

    def isend(self, state: int) -> bool:
        return bool(self.lib().oracle_isend(self._h, int(state)))

    def endids(self, state: int) -> np.ndarray:
        n = self.lib().oracle_endid_count(self._h, state)
        buf = np.zeros(max(n, 1), np.uint32)
        assert self.lib().oracle_endid_get(self._h, state, n, _p(buf)) == 1
        return buf[:n]


def have_ref() -> bool:
    return os.path.exists(REF_SO) and os.path.exists(HELPER_SO)


class Ref:
    """The real reference (oracle/_ref): regex -> DFA, literal fsm_exec, DFAVM."""

    _libs = None

    @classmethod
    def libs(cls):
        if cls._libs is None:
            if not have_ref():
                raise RuntimeError("oracle/_ref not built (needs /root/reference once; prebuilt files travel to the GPU box)")
            ref = C.CDLL(REF_SO, mode=C.RTLD_GLOBAL)  # RTLD_GLOBAL: the product shim binds libfsm by dlsym
            H = ref
            vp, sz = C.c_void_p, C.c_size_t
            H.rh_re_comp.restype = vp
            H.rh_re_comp.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_long]
            H.rh_union_res.restype = vp
            H.rh_union_res.argtypes = [C.c_int, vp, sz, C.c_int]
            H.rh_re_strings.restype = vp
            H.rh_re_strings.argtypes = [vp, vp, sz, C.c_int, C.c_int]
            H.rh_fsm_free.argtypes = [vp]
            H.rh_countstates.restype = C.c_uint
            H.rh_countstates.argtypes = [vp]
            H.rh_shuffle.argtypes = [vp, C.c_uint]
            H.rh_exec.argtypes = [vp, vp, sz, C.POINTER(C.c_uint)]
            H.rh_exec_batch.restype = C.c_double
            H.rh_exec_batch.argtypes = [vp, vp, vp, sz, vp, vp]
            H.rh_exec_batch_stride.restype = C.c_double
            H.rh_exec_batch_stride.argtypes = [vp, vp, sz, sz, vp, vp]
            H.rh_endid_count.restype = sz
            H.rh_endid_count.argtypes = [vp, C.c_uint]
            H.rh_endid_get.argtypes = [vp, C.c_uint, sz, vp]
            H.rh_vm_compile.restype = vp
            H.rh_vm_compile.argtypes = [vp, C.c_int]
            H.rh_vm_free.argtypes = [vp]
            H.rh_vm_match_batch_stride.restype = C.c_double
            H.rh_vm_match_batch_stride.argtypes = [vp, vp, sz, sz, vp]
            H.rh_vm_match_batch.restype = C.c_double
            H.rh_vm_match_batch.argtypes = [vp, vp, vp, sz, vp]
            H.rh_match_threads.restype = C.c_double
            H.rh_match_threads.argtypes = [vp, vp, vp, sz, sz, C.c_int, C.c_int, C.POINTER(C.c_ulong)]
            H.rh_parse_fsm_file.restype = vp
            H.rh_parse_fsm_file.argtypes = [C.c_char_p]
            H.rh_generate_matches.restype = sz
            H.rh_generate_matches.argtypes = [vp, sz, C.c_uint, vp, vp, sz]
            H.rh_union_repeated.restype = vp
            H.rh_union_repeated.argtypes = [C.c_int, vp, sz, C.c_uint, C.c_int]
            H.rh_exec_eager_batch.restype = None
            H.rh_exec_eager_batch.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, C.c_uint]
            H.rh_exec_eager_stream_batch.restype = None
            H.rh_exec_eager_stream_batch.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp, C.c_uint]
            ref.fsm_setendid.argtypes = [vp, C.c_uint]
            ref.fsm_union.restype = vp
            ref.fsm_union.argtypes = [vp, vp, vp]
            ref.fsm_determinise.argtypes = [vp]
            ref.fsm_minimise.argtypes = [vp]
            cls._libs = (ref, H)
        return cls._libs


class RefFsm:
    """Owns a `struct fsm *` of the real reference."""

    def __init__(self, ptr: int):
        if not ptr:
            raise ValueError("reference returned NULL fsm")
        self.ptr = ptr
        self.last_seconds = 0.0

    def __del__(self):
        if getattr(self, "ptr", None):
            Ref.libs()[1].rh_fsm_free(self.ptr)
            self.ptr = None

    def release(self) -> int:
        p, self.ptr = self.ptr, None
        return p

    # -- constructors ---------------------------------------------------------
    @classmethod
    def re_comp(cls, dialect: str, regex: bytes, flags: int = 0, determinise=True, minimise=True, endid: int = -1):
        _, H = Ref.libs()
        return cls(H.rh_re_comp(DIALECTS[dialect], regex, flags, int(determinise), int(minimise), endid))

    @classmethod
    def union_res(cls, dialect: str, regexes, flags: int = 0):
        _, H = Ref.libs()
        arr = (C.c_char_p * len(regexes))(*regexes)
        return cls(H.rh_union_res(DIALECTS[dialect], arr, len(regexes), flags))

    @classmethod
    def re_strings(cls, words, flags: int = 0, with_endids: bool = True):
        _, H = Ref.libs()
        arr = (C.c_char_p * len(words))(*words)
        lens = np.array([len(w) for w in words], np.uint32)
        return cls(H.rh_re_strings(arr, _p(lens) if len(words) else None, len(words), flags, int(with_endids)))

    @classmethod
    def union_repeated(cls, dialect: str, regexes, id_base: int = 1, force_endids: bool = False):
        """tests/eager_output/utils.c:57-131: RE_SAVE_LINKAGE_INFO + fsm_union_repeated_pattern_group
        (eager output id = id_base + index), determinise, minimise."""
        _, H = Ref.libs()
        arr = (C.c_char_p * len(regexes))(*regexes)
        return cls(H.rh_union_repeated(DIALECTS[dialect], arr, len(regexes), id_base, int(force_endids)))

    @classmethod
    def parse_file(cls, path: str):
        """fsm_parse + determinise + minimise of a .fsm file; None if the reference rejects it."""
        _, H = Ref.libs()
        p = H.rh_parse_fsm_file(path.encode())
        return cls(p) if p else None

    def generate_matches(self, maxlen: int = 24, cap: int = 8, seed: int = 1):
        """Accepted inputs found by the reference's fsm_generate_matches (randomised labels)."""
        _, H = Ref.libs()
        buf = np.zeros((cap, maxlen), np.uint8)
        lens = np.zeros(cap, np.uint32)
        n = H.rh_generate_matches(self.ptr, maxlen, seed, _p(buf), _p(lens), cap)
        return [bytes(buf[i, :lens[i]]) for i in range(n)]

    def exec_eager_strings(self, strings, cap: int = 64):
        """Literal fsm_exec with the eager-output callback: (ret, end, [sorted emitted ids])."""
        _, H = Ref.libs()
        off = np.zeros(len(strings) + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in strings])
        base = np.frombuffer(b"".join(strings) + b"\0", np.uint8)
        n = len(strings)
        ret, end = np.zeros(n, np.int8), np.zeros(n, np.uint32)
        ids, cnt = np.zeros((n, cap), np.uint32), np.zeros(n, np.uint32)
        H.rh_exec_eager_batch(self.ptr, _p(base), _p(off), n, _p(ret), _p(end), _p(ids), _p(cnt), cap)
        return ret, end, [np.sort(ids[i, :cnt[i]]) for i in range(n)]

    def exec_eager_stream_strings(self, strings, cap: int = 64):
        """Literal fsm_exec, the raw callback stream: (ret, end, counts, [ids in call order, repeats kept])."""
        _, H = Ref.libs()
        off = np.zeros(len(strings) + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in strings])
        base = np.frombuffer(b"".join(strings) + b"\0", np.uint8)
        n = len(strings)
        ret, end = np.zeros(n, np.int8), np.zeros(n, np.uint32)
        ids, cnt = np.zeros((n, cap), np.uint32), np.zeros(n, np.uint32)
        H.rh_exec_eager_stream_batch(self.ptr, _p(base), _p(off), n, _p(ret), _p(end), _p(ids), _p(cnt), cap)
        return ret, end, cnt, [ids[i, :min(cnt[i], cap)].copy() for i in range(n)]

    # -- queries ----------------------------------------------------------------
    @property
    def nstates(self) -> int:
        return Ref.libs()[1].rh_countstates(self.ptr)

    def setendid(self, i: int):
        assert Ref.libs()[0].fsm_setendid(self.ptr, i) == 1

    def union_with(self, other: "RefFsm"):
        p = Ref.libs()[0].fsm_union(self.ptr, other.release(), None)
        assert p
        self.ptr = p

    def determinise(self):
        assert Ref.libs()[0].fsm_determinise(self.ptr) == 1

    def minimise(self):
        assert Ref.libs()[0].fsm_minimise(self.ptr) == 1

    def shuffle(self, seed: int):
        assert Ref.libs()[1].rh_shuffle(self.ptr, seed)

    def exec_one(self, s: bytes):
        e = C.c_uint(0)
        b = np.frombuffer(s, np.uint8) if len(s) else np.zeros(1, np.uint8)
        r = Ref.libs()[1].rh_exec(self.ptr, _p(b), len(s), C.byref(e))
        return r, (e.value if r == 1 else 0xFFFFFFFF)

    def exec_offsets(self, base: np.ndarray, off: np.ndarray):
        base = np.ascontiguousarray(base, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        ret, end = np.zeros(n, np.int8), np.zeros(n, np.uint32)
        b = base if len(base) else np.zeros(1, np.uint8)
        self.last_seconds = Ref.libs()[1].rh_exec_batch(self.ptr, _p(b), _p(off), n, _p(ret), _p(end))
        return ret, end

    def exec_strings(self, strings):
        off = np.zeros(len(strings) + 1, np.uint64)
        off[1:] = np.cumsum([len(s) for s in strings])
        return self.exec_offsets(np.frombuffer(b"".join(strings), np.uint8), off)

    def exec_stride(self, data: np.ndarray):
        data = np.ascontiguousarray(data, np.uint8)
        n, stride = data.shape
        ret, end = np.zeros(n, np.int8), np.zeros(n, np.uint32)
        self.last_seconds = Ref.libs()[1].rh_exec_batch_stride(self.ptr, _p(data), stride, n, _p(ret), _p(end))
        return ret, end

    def exec_hoisted_stride(self, data: np.ndarray):
        """fsm_exec with its per-call fsm_all(fsm_isdfa) check removed (NOT the reference: derived from
        exec.c by oracle/build_ref.sh; SURVEY.md 8(d) CPU line 2)."""
        _, H = Ref.libs()
        H.rh_exec_hoisted_batch_stride.restype = C.c_double
        H.rh_exec_hoisted_batch_stride.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        data = np.ascontiguousarray(data, np.uint8)
        n, stride = data.shape
        ret, end = np.zeros(n, np.int8), np.zeros(n, np.uint32)
        self.last_seconds = H.rh_exec_hoisted_batch_stride(self.ptr, _p(data), stride, n, _p(ret), _p(end))
        return ret, end

    def codegen_match_stride(self, data: np.ndarray, lang: str = "vmc", cflags=("-O3",), timeout: float = 120.0, comments: bool = False):
        """What `retest -l vmc|c` runs (src/retest/runner.c:290-404): fsm_print() the DFA as C, compile it with
        the host C compiler into a shared object, dlopen it and call fsm_main(b, e) per input.  Returns the 0/1
        array (or None if printing / compiling failed or timed out); last_seconds = time inside the loop,
        last_compile_seconds = fsm_print + cc."""
        import tempfile
        import time as _t
        _, H = Ref.libs()
        H.rh_print_matcher.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        H.rh_codegen_match_batch_stride.restype = C.c_double
        H.rh_codegen_match_batch_stride.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        d = tempfile.mkdtemp(prefix="fsmcodegen")
        src, so = os.path.join(d, "m.c"), os.path.join(d, "m.so")
        t0 = _t.perf_counter()
        if H.rh_print_matcher(self.ptr, 0 if lang == "vmc" else 1, int(comments), src.encode()) != 0:
            return None
        try:
            subprocess.check_call([os.environ.get("CC", "gcc"), *cflags, "-shared", "-fPIC", "-w", src, "-o", so], timeout=timeout)
        except (subprocess.SubprocessError, OSError):
            return None
        self.last_compile_seconds = _t.perf_counter() - t0
        lib = C.CDLL(so)
        fn = C.cast(lib.fsm_main, C.c_void_p)
        data = np.ascontiguousarray(data, np.uint8)
        n, stride = data.shape
        ret = np.zeros(n, np.int8)
        self.last_seconds = H.rh_codegen_match_batch_stride(fn, _p(data), stride, n, _p(ret))
        return ret

    def endids(self, state: int) -> np.ndarray:
        _, H = Ref.libs()
        n = H.rh_endid_count(self.ptr, state)
        buf = np.zeros(max(n, 1), np.uint32)
        assert H.rh_endid_get(self.ptr, state, n, _p(buf)) == 1
        return buf[:n]

    def vm_match_stride(self, data: np.ndarray, version: int = 2):
        _, H = Ref.libs()
        vm = H.rh_vm_compile(self.ptr, version)
        assert vm
        data = np.ascontiguousarray(data, np.uint8)
        n, stride = data.shape
        ret = np.zeros(n, np.int8)
        self.last_seconds = H.rh_vm_match_batch_stride(vm, _p(data), stride, n, _p(ret))
        H.rh_vm_free(vm)
        return ret

    def match_threads(self, data: np.ndarray, nthreads: int, reps: int = 1, use_vm: int = 2):
        """The reference matcher on `nthreads` host threads over contiguous slices of the rows
        (use_vm 1/2: DFAVM bytecode version; 0: literal fsm_exec).  Returns (GB/s, accepts)."""
        _, H = Ref.libs()
        data = np.ascontiguousarray(data, np.uint8)
        n, stride = data.shape
        vm = H.rh_vm_compile(self.ptr, use_vm) if use_vm else None
        m = C.c_ulong(0)
        t = H.rh_match_threads(vm, self.ptr, _p(data), stride, n, nthreads, reps, C.byref(m))
        if vm:
            H.rh_vm_free(vm)
        self.last_seconds = t
        return data.size * reps / t / 1e9, int(m.value)

    def flatten(self):
        """struct fsm * -> FlatDfa through the PRODUCT's shim (fsm_hip_flatten)."""
        from libfsm_amd.capi import FlatDfa, load_library
        lib = load_library()
        C.set_errno(0)
        d = lib.fsm_hip_flatten(self.ptr)
        if not d:
            raise OSError(C.get_errno(), "fsm_hip_flatten")
        try:
            return FlatDfa.from_desc(d.contents)
        finally:
            lib.fsm_hip_desc_free(d)
