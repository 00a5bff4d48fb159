"""ctypes binding of libfsm_hip.so -- the C ABI declared in include/fsm_hip.h.

Python is only the harness language here (tests, bench.py); the product is the
shared library.  Nothing in this module matches inputs on the CPU: every
exec_* call goes through the HIP kernels and raises if the library or a GPU is
missing.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

NO_MATCH = 0xFFFFFFFF
NO_ID = 0xFFFFFFFE
STATE_START = 0xFFFFFFFD
STATE_DEAD = 0xFFFFFFFC

LAYOUT_AUTO, LAYOUT_TINY, LAYOUT_LDS, LAYOUT_COMB, LAYOUT_GLOBAL, LAYOUT_COMB256, LAYOUT_COMBSELF, LAYOUT_SPARSE, LAYOUT_LDSSELF, LAYOUT_LDS2 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
META_OFF64, META_OFF32, META_LENGTHS = 0, 1, 2   # fsm_hip_exec_batch_packed_all: what the metadata array is
LAYOUT_NAMES = {1: "tiny", 2: "lds", 3: "comb", 4: "global", 5: "comb256", 6: "combself", 7: "sparse", 8: "ldsself", 9: "lds2"}
ALL_LAYOUTS = (LAYOUT_TINY, LAYOUT_COMBSELF, LAYOUT_LDS2, LAYOUT_COMB256, LAYOUT_LDSSELF, LAYOUT_LDS, LAYOUT_COMB, LAYOUT_SPARSE, LAYOUT_GLOBAL)
NO_EARLY_RETIRE = 0x10
DEFER_UPLOAD = 0x20      # plan now, upload the layout's device image at the first call that needs it (fsm_hip_exec_multi never does)

KNOB_INPUT_MODE, KNOB_NB, KNOB_ROWS, KNOB_WAVES, KNOB_BLOCKS_PER_CU, KNOB_EARLY_RETIRE, KNOB_MASK, KNOB_HOT_BYTES, KNOB_SEG, KNOB_PREFETCH, KNOB_NT = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
KNOB_NOSKIP = 13
KNOB_RAGGED_ALIGN = 14
KNOB_DMA_BUFS = 15
KNOB_PICK_MEAN, KNOB_SPARSE_FAST, KNOB_LAZY_DYN, KNOB_LAZY_LINES = 18, 20, 21, 22
IN_DIRECT, IN_LDSDMA, IN_GENERIC, IN_RAGGED = 0, 1, 2, 3

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfsm_hip.so")

RANGE_DTYPE = np.dtype([("lo", "u1"), ("hi", "u1"), ("reserved", "<u2"), ("to", "<u4")])


class _Desc(C.Structure):
    _fields_ = [
        ("nstates", C.c_uint32),
        ("start", C.c_uint32),
        ("edge_off", C.c_void_p),
        ("ranges", C.c_void_p),
        ("is_end", C.c_void_p),
        ("endid_off", C.c_void_p),
        ("endids", C.c_void_p),
        ("eager_off", C.c_void_p),
        ("eager_ids", C.c_void_p),
    ]


class _Info(C.Structure):
    _fields_ = [
        ("nstates", C.c_uint32),
        ("nclasses", C.c_uint32),
        ("layout", C.c_uint32),
        ("nabsorbing", C.c_uint32),
        ("table_bytes", C.c_uint64),
        ("lds_bytes", C.c_uint32),
        ("waves_per_block", C.c_uint32),
        ("device", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load libfsm_hip.so (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"libfsm_hip.so not built at {p}: run `python __graft_entry__.py` (no CPU fallback exists)")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL, use_errno=True)
    vp, u32p, u64p, sz = C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t
    lib.fsm_hip_version.restype = C.c_int
    lib.fsm_hip_dfa_create.restype = vp
    lib.fsm_hip_dfa_create.argtypes = [C.POINTER(_Desc), C.c_uint]
    lib.fsm_hip_dfa_free.argtypes = [vp]
    lib.fsm_hip_dfa_info.argtypes = [vp, C.POINTER(_Info)]
    lib.fsm_hip_exec_batch.argtypes = [vp, vp, sz, u32p, sz, u32p, u64p]
    lib.fsm_hip_exec_batch_offsets.argtypes = [vp, vp, u64p, sz, u32p, u64p]
    lib.fsm_hip_exec_batch_device.argtypes = [vp, vp, sz, u32p, sz, u32p, u64p, vp]
    lib.fsm_hip_exec_batch_offsets_device.argtypes = [vp, vp, u64p, sz, u32p, u64p, vp]
    lib.fsm_hip_last_kernel_ms.restype = C.c_double
    lib.fsm_hip_last_kernel_ms.argtypes = [vp]
    lib.fsm_hip_endid_count.restype = sz
    lib.fsm_hip_endid_count.argtypes = [vp, C.c_uint32]
    lib.fsm_hip_endid_get.argtypes = [vp, C.c_uint32, sz, u32p]
    lib.fsm_hip_compile.restype = vp
    lib.fsm_hip_compile.argtypes = [vp, C.c_uint]
    lib.fsm_hip_exec.argtypes = [vp, vp, vp, C.POINTER(C.c_uint), vp]
    lib.fsm_hip_match_buffer.argtypes = [vp, C.c_char_p, sz]
    lib.fsm_hip_flatten.restype = C.POINTER(_Desc)
    lib.fsm_hip_flatten.argtypes = [vp]
    lib.fsm_hip_desc_free.argtypes = [C.POINTER(_Desc)]
    lib.fsm_hip_desc_read.restype = C.POINTER(_Desc)
    lib.fsm_hip_strings_new.restype = vp
    lib.fsm_hip_strings_free.argtypes = [vp]
    lib.fsm_hip_strings_add_raw.argtypes = [vp, C.c_char_p, sz, u32p]
    lib.fsm_hip_strings_build.restype = C.POINTER(_Desc)
    lib.fsm_hip_strings_build.argtypes = [vp, C.c_uint]
    lib.fsm_hip_gen_inputs_device.argtypes = [vp, sz, sz, C.c_uint64, C.c_uint64, vp, C.c_uint, vp, C.c_uint, C.c_uint, vp]
    lib.fsm_hip_gen_inputs_host.restype = None
    lib.fsm_hip_gen_inputs_host.argtypes = [vp, sz, sz, C.c_uint64, C.c_uint64, vp, C.c_uint, vp, C.c_uint, C.c_uint]
    lib.fsm_hip_dfa_tune.argtypes = [vp, C.c_int, C.c_int]
    lib.fsm_hip_plan_create.restype = vp
    lib.fsm_hip_plan_create.argtypes = [C.POINTER(_Desc), C.c_uint, C.c_uint32]
    lib.fsm_hip_plan_free.argtypes = [vp]
    lib.fsm_hip_plan_get.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(sz)]
    if path is None:
        _lib = lib
    return lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _oserr(what: str):
    e = C.get_errno()
    return OSError(e, f"{what}: {os.strerror(e)}")


@dataclass
class FlatDfa:
    """Flat DFA description = struct fsm_hip_dfa_desc (include/fsm_hip.h)."""

    nstates: int
    start: int
    edge_off: np.ndarray  # u32 [nstates+1]
    ranges: np.ndarray  # RANGE_DTYPE [nranges]
    is_end: np.ndarray  # u8 [nstates]
    endid_off: np.ndarray  # u32 [nstates+1]
    endids: np.ndarray  # u32 []
    eager_off: Optional[np.ndarray] = None  # u32 [nstates+1] or None (no eager outputs)
    eager_ids: Optional[np.ndarray] = None  # u32 []

    def __post_init__(self):
        if self.eager_off is not None:
            self.eager_off = np.ascontiguousarray(self.eager_off, dtype=np.uint32)
            self.eager_ids = np.ascontiguousarray(self.eager_ids if self.eager_ids is not None else [], dtype=np.uint32)
            if int(self.eager_off[-1]) == 0:
                self.eager_off = self.eager_ids = None
        self.edge_off = np.ascontiguousarray(self.edge_off, dtype=np.uint32)
        self.ranges = np.ascontiguousarray(self.ranges, dtype=RANGE_DTYPE)
        self.is_end = np.ascontiguousarray(self.is_end, dtype=np.uint8)
        self.endid_off = np.ascontiguousarray(self.endid_off, dtype=np.uint32)
        self.endids = np.ascontiguousarray(self.endids, dtype=np.uint32)

    def desc(self) -> _Desc:
        d = _Desc()
        d.nstates, d.start = self.nstates, self.start
        d.edge_off = self.edge_off.ctypes.data
        d.ranges = self.ranges.ctypes.data if len(self.ranges) else None
        d.is_end = self.is_end.ctypes.data
        d.endid_off = self.endid_off.ctypes.data
        d.endids = self.endids.ctypes.data if len(self.endids) else None
        d.eager_off = self.eager_off.ctypes.data if self.eager_off is not None else None
        d.eager_ids = self.eager_ids.ctypes.data if self.eager_off is not None and len(self.eager_ids) else None
        return d

    @classmethod
    def from_desc(cls, d: _Desc) -> "FlatDfa":
        n = d.nstates

        def arr(addr, count, dt):
            if count == 0 or not addr:
                return np.zeros(0, dtype=dt)
            buf = (C.c_char * (count * np.dtype(dt).itemsize)).from_address(addr)
            return np.frombuffer(buf, dtype=dt).copy()

        edge_off = arr(d.edge_off, n + 1, np.uint32)
        endid_off = arr(d.endid_off, n + 1, np.uint32) if d.endid_off else np.zeros(n + 1, np.uint32)
        eo = arr(d.eager_off, n + 1, np.uint32) if d.eager_off else None
        return cls(n, d.start, edge_off, arr(d.ranges, int(edge_off[n]), RANGE_DTYPE), arr(d.is_end, n, np.uint8),
                   endid_off, arr(d.endids, int(endid_off[n]), np.uint32),
                   eo, arr(d.eager_ids, int(eo[n]), np.uint32) if eo is not None else None)

    @classmethod
    def from_dense(cls, next_tab: np.ndarray, start: int, is_end: Sequence[int], endids=None) -> "FlatDfa":
        """next_tab: [S][256] int64/uint32, negative or 0xFFFFFFFF = no edge."""
        nt = np.asarray(next_tab, dtype=np.int64)
        S = nt.shape[0]
        rng, off = [], [0]
        for s in range(S):
            c = 0
            row = nt[s]
            while c < 256:
                t = row[c]
                if t < 0 or t == NO_MATCH:
                    c += 1
                    continue
                lo = c
                while c + 1 < 256 and row[c + 1] == t:
                    c += 1
                rng.append((lo, c, 0, int(t)))
                c += 1
            off.append(len(rng))
        eo, ei = [0], []
        for s in range(S):
            if endids is not None and s in endids:
                ei.extend(sorted(set(endids[s])))
            eo.append(len(ei))
        return cls(S, start, np.array(off, np.uint32), np.array(rng, dtype=RANGE_DTYPE) if rng else np.zeros(0, RANGE_DTYPE),
                   np.asarray(is_end, np.uint8), np.array(eo, np.uint32), np.array(ei, np.uint32))

    def dense(self) -> np.ndarray:
        """[S][256] uint32 next table, NO_MATCH = no edge."""
        t = np.full((self.nstates, 256), NO_MATCH, dtype=np.uint32)
        for s in range(self.nstates):
            for k in range(int(self.edge_off[s]), int(self.edge_off[s + 1])):
                r = self.ranges[k]
                t[s, int(r["lo"]):int(r["hi"]) + 1] = r["to"]
        return t

    def endids_of(self, state: int) -> np.ndarray:
        return self.endids[int(self.endid_off[state]):int(self.endid_off[state + 1])]

    def write_c(self, path: str):
        """fsm_hip_desc_write(): the library's own on-disk form ("FSMHIP01")."""
        lib = load_library()
        libc = C.CDLL(None, use_errno=True)
        libc.fopen.restype = C.c_void_p
        f = libc.fopen(path.encode(), b"wb")
        assert f
        d = self.desc()
        r = lib.fsm_hip_desc_write(C.byref(d), C.c_void_p(f))
        libc.fclose(C.c_void_p(f))
        if r != 0:
            raise _oserr("fsm_hip_desc_write")

    @classmethod
    def read_c(cls, path: str) -> "FlatDfa":
        lib = load_library()
        libc = C.CDLL(None, use_errno=True)
        libc.fopen.restype = C.c_void_p
        f = libc.fopen(path.encode(), b"rb")
        if not f:
            raise OSError(C.get_errno(), "fopen")
        C.set_errno(0)
        d = lib.fsm_hip_desc_read(C.c_void_p(f))
        libc.fclose(C.c_void_p(f))
        if not d:
            raise _oserr("fsm_hip_desc_read")
        try:
            return cls.from_desc(d.contents)
        finally:
            lib.fsm_hip_desc_free(d)

    @classmethod
    def from_strings(cls, words: Sequence[bytes], flags: int = 0, endids: Optional[Sequence[int]] = None) -> "FlatDfa":
        """fsm_hip_strings_*(): literal set -> DFA, the automaton libre's re_strings builds
        (flags: 1 anchor left, 2 anchor right, 4 AC automaton; endids: one per word or None)."""
        lib = load_library()
        g = lib.fsm_hip_strings_new()
        if not g:
            raise _oserr("fsm_hip_strings_new")
        try:
            for i, w in enumerate(words):
                eid = C.byref(C.c_uint32(int(endids[i]))) if endids is not None else None
                if not lib.fsm_hip_strings_add_raw(g, w, len(w), C.cast(eid, C.POINTER(C.c_uint32)) if eid is not None else None):
                    raise _oserr("fsm_hip_strings_add_raw")
            C.set_errno(0)
            d = lib.fsm_hip_strings_build(g, flags)
            if not d:
                raise _oserr("fsm_hip_strings_build")
            try:
                return cls.from_desc(d.contents)
            finally:
                lib.fsm_hip_desc_free(d)
        finally:
            lib.fsm_hip_strings_free(g)

    def canonical(self):
        """(nstates, start, dense [S][256] next table with 0xFFFFFFFF = no edge, is_end, end-id lists): order-free form."""
        nt = np.full((self.nstates, 256), NO_MATCH, np.uint32)
        for s_ in range(self.nstates):
            for r in self.ranges[int(self.edge_off[s_]):int(self.edge_off[s_ + 1])]:
                nt[s_, int(r["lo"]):int(r["hi"]) + 1] = r["to"]
        return self.nstates, self.start, nt, self.is_end.copy(), self.endid_off.copy(), self.endids.copy()

    def save(self, path: str, **extra):
        np.savez_compressed(path, nstates=np.uint32(self.nstates), start=np.uint32(self.start), edge_off=self.edge_off,
                            r_lo=self.ranges["lo"], r_hi=self.ranges["hi"], r_to=self.ranges["to"], is_end=self.is_end,
                            endid_off=self.endid_off, endids=self.endids,
                            **({"eager_off": self.eager_off, "eager_ids": self.eager_ids} if self.eager_off is not None else {}), **extra)

    @classmethod
    def load(cls, path_or_npz) -> "FlatDfa":
        z = np.load(path_or_npz) if isinstance(path_or_npz, (str, os.PathLike)) else path_or_npz
        r = np.zeros(len(z["r_lo"]), dtype=RANGE_DTYPE)
        r["lo"], r["hi"], r["to"] = z["r_lo"], z["r_hi"], z["r_to"]
        return cls(int(z["nstates"]), int(z["start"]), z["edge_off"], r, z["is_end"], z["endid_off"], z["endids"],
                   z["eager_off"] if "eager_off" in z else None, z["eager_ids"] if "eager_ids" in z else None)

    def eager_of(self, state: int) -> np.ndarray:
        if self.eager_off is None:
            return np.zeros(0, np.uint32)
        return self.eager_ids[int(self.eager_off[state]):int(self.eager_off[state + 1])]


class Plan:
    """Host-side table plan (fsm_hip_plan_*): inspection only, never executes inputs."""

    _WHAT = dict(scalars=(0, np.uint32), cls=(1, np.uint8), new2old=(2, np.uint32), fin=(3, np.uint32),
                 dense=(4, np.uint32), tiny_col=(5, np.uint64), lds_tab=(6, np.uint16), comb=(7, np.uint32),
                 comb_dflt=(8, np.uint32), comb_off=(9, np.uint32), comb_fin=(10, np.uint32), glob_tab=(11, np.uint32),
                 comb256=(12, np.uint32), comb256_off=(13, np.uint32), comb256_fin=(14, np.uint32), comb_smask=(15, np.uint32),
                 emask=(16, np.uint64), eager_ids=(17, np.uint32), sparse=(18, np.uint32),
                 ew_off=(19, np.uint32), ew_word=(20, np.uint32), ew_mask=(21, np.uint64), tiny5_col=(22, np.uint32),
                 comb_rng=(23, np.uint16), lazy=(24, np.uint32), glob_tab16=(25, np.uint16), glob16_rank=(26, np.uint32))

    def __init__(self, flat: FlatDfa, flags: int = 0, lds_limit: int = 0):
        lib = load_library()
        C.set_errno(0)
        d = flat.desc()
        self._h = lib.fsm_hip_plan_create(C.byref(d), flags, lds_limit)
        if not self._h:
            raise _oserr("fsm_hip_plan_create")
        self._lib = lib
        s = self.get("scalars")
        (self.nstates, self.S1, self.start, self.C, self.abs_min, self.nabsorbing, self.layout, self.row_bytes,
         self.comb_abs_min_off, self.comb256_abs_min_off, self.comb256_dflt, self.eager_lo_end,
         self.eager_hi_begin, self.comb_eager_lo_off, self.comb_eager_hi_off, self.comb256_eager_lo_off,
         self.comb256_eager_hi_off) = (int(x) for x in s)

    def get(self, what: str) -> np.ndarray:
        w, dt = self._WHAT[what]
        p, n = C.c_void_p(), C.c_size_t()
        if self._lib.fsm_hip_plan_get(self._h, w, C.byref(p), C.byref(n)) != 0:
            raise _oserr("fsm_hip_plan_get")
        if n.value == 0:
            return np.zeros(0, dt)
        buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt).copy()

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.fsm_hip_plan_free(self._h)
            self._h = None


class HipDfa:
    """struct fsm_hip_dfa *: a DFA resident on the GPU."""

    def __init__(self, flat: Optional[FlatDfa] = None, flags: int = 0, *, handle=None, borrowed: bool = False):
        self._lib = load_library()
        self._borrowed = borrowed      # a replica owned by a HipNode: never freed from here
        if handle is not None:
            self._h = handle
        else:
            C.set_errno(0)
            d = flat.desc()
            self._h = self._lib.fsm_hip_dfa_create(C.byref(d), flags)
            if not self._h:
                raise _oserr("fsm_hip_dfa_create")

    @classmethod
    def compile_fsm(cls, fsm_ptr: int, flags: int = 0) -> "HipDfa":
        """fsm_hip_compile(const struct fsm *): libfsm must already be loaded RTLD_GLOBAL."""
        lib = load_library()
        C.set_errno(0)
        h = lib.fsm_hip_compile(fsm_ptr, flags)
        if not h:
            raise _oserr("fsm_hip_compile")
        return cls(handle=h)

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._lib.fsm_hip_dfa_free(self._h)
            self._h = None

    __del__ = close

    @property
    def handle(self):
        return self._h

    def info(self) -> dict:
        i = _Info()
        if self._lib.fsm_hip_dfa_info(self._h, C.byref(i)) != 0:
            raise _oserr("fsm_hip_dfa_info")
        d = {k: getattr(i, k) for k, _ in _Info._fields_ if k != "reserved"}
        d["layout_name"] = LAYOUT_NAMES.get(i.layout, "?")
        return d

    def tune(self, knob: int, value: int):
        if self._lib.fsm_hip_dfa_tune(self._h, knob, value) != 0:
            raise _oserr("fsm_hip_dfa_tune")

    # ---- host-buffer fronts -------------------------------------------------
    def exec_batch(self, data: np.ndarray, lens: Optional[np.ndarray] = None, want_bitmap: bool = True):
        """data: uint8 [n, stride]; returns (end u32[n], bitmap u64[ceil(n/64)])."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n, stride = data.shape
        end = np.empty(n, dtype=np.uint32)
        bm = np.zeros((n + 63) // 64, dtype=np.uint64) if want_bitmap else None
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.uint32)
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch(self._h, _ptr(data), stride, _ptr(lens), n, _ptr(end), _ptr(bm)) != 0:
            raise _oserr("fsm_hip_exec_batch")
        return end, bm

    def exec_batch_offsets(self, base: np.ndarray, off: np.ndarray, want_bitmap: bool = True):
        base = np.ascontiguousarray(base, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        end = np.empty(n, dtype=np.uint32)
        bm = np.zeros((n + 63) // 64, dtype=np.uint64) if want_bitmap else None
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_offsets(self._h, _ptr(base) if len(base) else None, _ptr(off), n, _ptr(end), _ptr(bm)) != 0:
            raise _oserr("fsm_hip_exec_batch_offsets")
        return end, bm

    def exec_batch_offsets32(self, base: np.ndarray, off32: np.ndarray, want_bitmap: bool = True, want_end: bool = True):
        """fsm_hip_exec_batch_offsets32: u32 offsets (batches below 4 GiB)."""
        base = np.ascontiguousarray(base, dtype=np.uint8)
        off32 = np.ascontiguousarray(off32, dtype=np.uint32)
        n = len(off32) - 1
        end = np.empty(n, dtype=np.uint32) if want_end else None
        bm = np.zeros((n + 63) // 64, dtype=np.uint64) if want_bitmap else None
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_offsets32(C.c_void_p(self._h), C.c_void_p(base.ctypes.data if len(base) else None), C.c_void_p(off32.ctypes.data),
                                                  C.c_size_t(n), C.c_void_p(end.ctypes.data if want_end else None), C.c_void_p(bm.ctypes.data if want_bitmap else None)) != 0:
            raise _oserr("fsm_hip_exec_batch_offsets32")
        return end, bm

    def exec_batch_lengths(self, base: np.ndarray, lens: np.ndarray, want_bitmap: bool = True, want_end: bool = True):
        """fsm_hip_exec_batch_lengths: inputs packed back to back, their lengths and nothing else."""
        base = np.ascontiguousarray(base, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = len(lens)
        end = np.empty(n, dtype=np.uint32) if want_end else None
        bm = np.zeros((n + 63) // 64, dtype=np.uint64) if want_bitmap else None
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_lengths(C.c_void_p(self._h), C.c_void_p(base.ctypes.data if len(base) else None), C.c_void_p(lens.ctypes.data if n else None),
                                                C.c_size_t(n), C.c_void_p(end.ctypes.data if want_end else None), C.c_void_p(bm.ctypes.data if want_bitmap else None)) != 0:
            raise _oserr("fsm_hip_exec_batch_lengths")
        return end, bm

    def exec_batch_offsets32_device(self, d_base: int, d_off32: int, n: int, d_end: int = 0, d_bitmap: int = 0, stream: int = 0):
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_offsets32_device(C.c_void_p(self._h), C.c_void_p(d_base), C.c_void_p(d_off32), C.c_size_t(n), C.c_void_p(d_end or None),
                                                         C.c_void_p(d_bitmap or None), C.c_void_p(stream or None)) != 0:
            raise _oserr("fsm_hip_exec_batch_offsets32_device")

    def exec_batch_lengths_device(self, d_base: int, d_len: int, n: int, d_end: int = 0, d_bitmap: int = 0, stream: int = 0):
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_lengths_device(C.c_void_p(self._h), C.c_void_p(d_base), C.c_void_p(d_len), C.c_size_t(n), C.c_void_p(d_end or None),
                                                       C.c_void_p(d_bitmap or None), C.c_void_p(stream or None)) != 0:
            raise _oserr("fsm_hip_exec_batch_lengths_device")

    def exec_strings(self, strings: Sequence[bytes]):
        off = np.zeros(len(strings) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in strings])
        base = np.frombuffer(b"".join(strings), dtype=np.uint8)
        return self.exec_batch_offsets(base, off)

    def match_buffer(self, s: bytes) -> int:
        C.set_errno(0)
        r = self._lib.fsm_hip_match_buffer(self._h, s, len(s))
        if r < 0:
            raise _oserr("fsm_hip_match_buffer")
        return r

    # ---- device-pointer front (raw addresses, e.g. torch tensors' data_ptr()) ----
    def exec_batch_device(self, d_base: int, stride: int, n: int, d_end: int = 0, d_bitmap: int = 0, d_len: int = 0, stream: int = 0):
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_device(self._h, d_base, stride, d_len or None, n, d_end or None, d_bitmap or None, stream or None) != 0:
            raise _oserr("fsm_hip_exec_batch_device")

    def last_kernel_name(self) -> str:
        self._lib.fsm_hip_last_kernel_name.restype = C.c_char_p
        return (self._lib.fsm_hip_last_kernel_name(C.c_void_p(self._h)) or b"").decode()

    def exec_batch_eager_device(self, d_base: int, stride: int, n: int, d_end: int, d_sets: int, d_len: int = 0, stream: int = 0):
        """d_sets: n * eager_words() u64 on the device."""
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_eager_device(C.c_void_p(self._h), C.c_void_p(d_base), C.c_size_t(stride), C.c_void_p(d_len or None),
                                                     C.c_size_t(n), C.c_void_p(d_end or None), C.c_void_p(d_sets), C.c_void_p(stream or None)) != 0:
            raise _oserr("fsm_hip_exec_batch_eager_device")

    def eager_words(self) -> int:
        self._lib.fsm_hip_eager_words.restype = C.c_size_t
        return int(self._lib.fsm_hip_eager_words(C.c_void_p(self._h)))

    def exec_batch_offsets_device(self, d_base: int, d_off: int, n: int, d_end: int = 0, d_bitmap: int = 0, stream: int = 0):
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_offsets_device(self._h, d_base, d_off, n, d_end or None, d_bitmap or None, stream or None) != 0:
            raise _oserr("fsm_hip_exec_batch_offsets_device")

    def exec_batch_ids(self, data: np.ndarray, mode: int, lens: Optional[np.ndarray] = None) -> np.ndarray:
        """Device-side end-id delivery: mode 1 = lowest id (AMBIG_EARLIEST), 2 = index into ret sets, 3 = AMBIG_ERROR."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n, stride = data.shape
        out = np.empty(n, dtype=np.uint32)
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.uint32)
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_ids(C.c_void_p(self._h), _ptr(data), C.c_size_t(stride), _ptr(lens), C.c_size_t(n),
                                            C.c_int(mode), _ptr(out)) != 0:
            raise _oserr("fsm_hip_exec_batch_ids")
        return out

    def exec_batch_resume(self, data: np.ndarray, state_io: np.ndarray, lens: Optional[np.ndarray] = None):
        """Streaming: start row i from state_io[i] (START/DEAD/state id); returns (state_out, end)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n, stride = data.shape
        st = np.ascontiguousarray(state_io, dtype=np.uint32).copy()
        end = np.empty(n, dtype=np.uint32)
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.uint32)
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_resume(C.c_void_p(self._h), _ptr(data), C.c_size_t(stride), _ptr(lens), C.c_size_t(n),
                                               _ptr(st), _ptr(end)) != 0:
            raise _oserr("fsm_hip_exec_batch_resume")
        return st, end

    def exec_batch_eager(self, data: np.ndarray, lens: Optional[np.ndarray] = None):
        """Walk + eager outputs: returns (end u32[n], list of emitted-id arrays per input)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n, stride = data.shape
        end = np.empty(n, dtype=np.uint32)
        self._lib.fsm_hip_eager_words.restype = C.c_size_t
        W = self._lib.fsm_hip_eager_words(C.c_void_p(self._h))
        eo = np.zeros((n, W), dtype=np.uint64)
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.uint32)
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_eager(C.c_void_p(self._h), _ptr(data), C.c_size_t(stride), _ptr(lens), C.c_size_t(n),
                                              _ptr(end), _ptr(eo)) != 0:
            raise _oserr("fsm_hip_exec_batch_eager")
        self._lib.fsm_hip_eager_id_count.restype = C.c_size_t
        self._lib.fsm_hip_eager_id.restype = C.c_uint32
        k = self._lib.fsm_hip_eager_id_count(C.c_void_p(self._h))
        ids = np.array([self._lib.fsm_hip_eager_id(C.c_void_p(self._h), C.c_uint(b)) for b in range(k)], np.uint32)
        bits = np.unpackbits(eo.view(np.uint8).reshape(n, W * 8), axis=1, bitorder="little")[:, :k].astype(bool)
        sets = [ids[row] for row in bits]
        return end, sets

    def exec_batch_eager_trace(self, data: np.ndarray, lens: Optional[np.ndarray] = None, off: Optional[np.ndarray] = None, cap: int = 64):
        """fsm_exec's eager-output callback stream per input, order and repeats kept: (end u32[n], count u32[n],
        [(ids, positions) of the first min(count, cap) emissions]).  data: (n, stride) rows (+ lens), or a flat byte
        array with off[n + 1]."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if off is not None:
            off = np.ascontiguousarray(off, dtype=np.uint64)
            n, stride = len(off) - 1, 0
        else:
            n, stride = data.shape
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.uint32)
        end, cnt = np.empty(n, np.uint32), np.zeros(n, np.uint32)
        ids, pos = np.zeros((n, max(cap, 1)), np.uint32), np.zeros((n, max(cap, 1)), np.uint32)
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_eager_trace(C.c_void_p(self._h), _ptr(data) if data.size else None, C.c_size_t(stride), _ptr(lens), _ptr(off),
                                                    C.c_size_t(n), C.c_size_t(cap), _ptr(end), _ptr(cnt), _ptr(ids), _ptr(pos)) != 0:
            raise _oserr("fsm_hip_exec_batch_eager_trace")
        k = np.minimum(cnt, cap)
        return end, cnt, [(ids[i, :k[i]].copy(), pos[i, :k[i]].copy()) for i in range(n)]

    def exec_batch_eager_trace_device(self, d_base: int, stride: int, n: int, cap: int, d_count: int, d_ids: int, d_pos: int = 0, d_end: int = 0,
                                      d_len: int = 0, d_off: int = 0, stream: int = 0):
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_eager_trace_device(C.c_void_p(self._h), C.c_void_p(d_base or None), C.c_size_t(stride), C.c_void_p(d_len or None),
                                                           C.c_void_p(d_off or None), C.c_size_t(n), C.c_size_t(cap), C.c_void_p(d_end or None),
                                                           C.c_void_p(d_count), C.c_void_p(d_ids or None), C.c_void_p(d_pos or None), C.c_void_p(stream or None)) != 0:
            raise _oserr("fsm_hip_exec_batch_eager_trace_device")

    def exec_packed_all_form(self, base: np.ndarray, meta_form: int, meta: np.ndarray, n: int, ids_mode: int = 0, want_end=True, want_bitmap=False,
                             want_eager=False):
        """fsm_hip_exec_batch_packed_all: returns dict(end=, bitmap=, ids=, eager=) with the outputs asked for."""
        base = np.ascontiguousarray(base, dtype=np.uint8)
        meta = np.ascontiguousarray(meta, dtype=np.uint64 if meta_form == 0 else np.uint32)
        end = np.empty(n, np.uint32) if want_end else None
        bm = np.zeros((n + 63) // 64, np.uint64) if want_bitmap else None
        ids = np.empty(n, np.uint32) if ids_mode else None
        W = self.eager_words()
        eo = np.zeros((n, W), np.uint64) if want_eager else None
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_packed_all(C.c_void_p(self._h), _ptr(base) if base.size else None, C.c_int(meta_form), _ptr(meta), C.c_size_t(n),
                                                   _ptr(end), _ptr(bm), C.c_int(ids_mode), _ptr(ids), _ptr(eo)) != 0:
            raise _oserr("fsm_hip_exec_batch_packed_all")
        return {"end": end, "bitmap": bm, "ids": ids, "eager": eo}

    def exec_packed_all_device(self, d_base: int, meta_form: int, d_meta: int, n: int, d_end: int = 0, d_bitmap: int = 0, ids_mode: int = 0, d_ids: int = 0,
                               d_eager: int = 0, stream: int = 0):
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_packed_all_device(C.c_void_p(self._h), C.c_void_p(d_base or None), C.c_int(meta_form), C.c_void_p(d_meta or None), C.c_size_t(n),
                                                          C.c_void_p(d_end or None), C.c_void_p(d_bitmap or None), C.c_int(ids_mode), C.c_void_p(d_ids or None),
                                                          C.c_void_p(d_eager or None), C.c_void_p(stream or None)) != 0:
            raise _oserr("fsm_hip_exec_batch_packed_all_device")

    # ---- the same three fronts over packed inputs (base + off[n + 1]) -----------
    def exec_offsets_ids(self, base: np.ndarray, off: np.ndarray, mode: int) -> np.ndarray:
        base = np.ascontiguousarray(base, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        out = np.empty(n, dtype=np.uint32)
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_ids_offsets(C.c_void_p(self._h), _ptr(base) if len(base) else None, _ptr(off), C.c_size_t(n),
                                                    C.c_int(mode), _ptr(out)) != 0:
            raise _oserr("fsm_hip_exec_batch_ids_offsets")
        return out

    def exec_offsets_resume(self, base: np.ndarray, off: np.ndarray, state_io: np.ndarray):
        base = np.ascontiguousarray(base, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        st = np.ascontiguousarray(state_io, dtype=np.uint32).copy()
        end = np.empty(n, dtype=np.uint32)
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_resume_offsets(C.c_void_p(self._h), _ptr(base) if len(base) else None, _ptr(off), C.c_size_t(n),
                                                       _ptr(st), _ptr(end)) != 0:
            raise _oserr("fsm_hip_exec_batch_resume_offsets")
        return st, end

    def exec_packed_resume(self, base: np.ndarray, meta_form: int, meta: np.ndarray, n: int, state_io: np.ndarray):
        """fsm_hip_exec_batch_resume_packed: resume over u64 offsets / u32 offsets / lengths alone; returns (state_out, end)."""
        base = np.ascontiguousarray(base, dtype=np.uint8)
        meta = np.ascontiguousarray(meta)
        st = np.ascontiguousarray(state_io, dtype=np.uint32).copy()
        end = np.empty(n, dtype=np.uint32)
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_resume_packed(C.c_void_p(self._h), C.c_void_p(base.ctypes.data if len(base) else None), C.c_int(meta_form),
                                                      C.c_void_p(meta.ctypes.data if len(meta) else None), C.c_size_t(n), _ptr(st), _ptr(end)) != 0:
            raise _oserr("fsm_hip_exec_batch_resume_packed")
        return st, end

    def exec_packed_resume_device(self, d_base: int, meta_form: int, d_meta: int, n: int, d_state_io: int, d_end: int = 0, d_bitmap: int = 0, stream: int = 0):
        C.set_errno(0)
        vp = C.c_void_p
        if self._lib.fsm_hip_exec_batch_resume_packed_device(vp(self._h), vp(d_base or None), C.c_int(meta_form), vp(d_meta or None), C.c_size_t(n), vp(d_state_io),
                                                             vp(d_end or None), vp(d_bitmap or None), vp(stream or None)) != 0:
            raise _oserr("fsm_hip_exec_batch_resume_packed_device")

    def reserve(self, n: int):
        """fsm_hip_reserve: allocate now what batches of up to n inputs would allocate later (graph capture from the first launch)."""
        C.set_errno(0)
        if self._lib.fsm_hip_reserve(C.c_void_p(self._h), C.c_size_t(n)) != 0:
            raise _oserr("fsm_hip_reserve")

    def exec_offsets_eager(self, base: np.ndarray, off: np.ndarray):
        base = np.ascontiguousarray(base, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        end = np.empty(n, dtype=np.uint32)
        self._lib.fsm_hip_eager_words.restype = C.c_size_t
        W = self._lib.fsm_hip_eager_words(C.c_void_p(self._h))
        eo = np.zeros((n, W), dtype=np.uint64)
        C.set_errno(0)
        if self._lib.fsm_hip_exec_batch_eager_offsets(C.c_void_p(self._h), _ptr(base) if len(base) else None, _ptr(off), C.c_size_t(n),
                                                      _ptr(end), _ptr(eo)) != 0:
            raise _oserr("fsm_hip_exec_batch_eager_offsets")
        self._lib.fsm_hip_eager_id_count.restype = C.c_size_t
        self._lib.fsm_hip_eager_id.restype = C.c_uint32
        k = self._lib.fsm_hip_eager_id_count(C.c_void_p(self._h))
        ids = np.array([self._lib.fsm_hip_eager_id(C.c_void_p(self._h), C.c_uint(b)) for b in range(k)], np.uint32)
        bits = np.unpackbits(eo.view(np.uint8).reshape(n, W * 8), axis=1, bitorder="little")[:, :k].astype(bool)
        return end, [ids[row] for row in bits]

    def exec_offsets_device_front(self, what: str, d_base: int, d_off: int, n: int, d_out: int, d_aux: int = 0, mode: int = 1, stream: int = 0):
        """The *_offsets_device entry points: what = 'ids' (d_out = ids), 'resume' (d_out = state_io, d_aux = end), 'eager' (d_out = end, d_aux = sets)."""
        C.set_errno(0)
        vp = C.c_void_p
        if what == "ids":
            r = self._lib.fsm_hip_exec_batch_ids_offsets_device(vp(self._h), vp(d_base), vp(d_off), C.c_size_t(n), C.c_int(mode), vp(d_out), vp(stream or None))
        elif what == "resume":
            r = self._lib.fsm_hip_exec_batch_resume_offsets_device(vp(self._h), vp(d_base), vp(d_off), C.c_size_t(n), vp(d_out), vp(d_aux or None), None, vp(stream or None))
        else:
            r = self._lib.fsm_hip_exec_batch_eager_offsets_device(vp(self._h), vp(d_base), vp(d_off), C.c_size_t(n), vp(d_out or None), vp(d_aux), vp(stream or None))
        if r != 0:
            raise _oserr("fsm_hip_exec_batch_%s_offsets_device" % what)

    def eager_id_count(self) -> int:
        self._lib.fsm_hip_eager_id_count.restype = C.c_size_t
        return int(self._lib.fsm_hip_eager_id_count(C.c_void_p(self._h)))

    def eager_id(self, bit: int) -> int:
        self._lib.fsm_hip_eager_id.restype = C.c_uint32
        return int(self._lib.fsm_hip_eager_id(C.c_void_p(self._h), C.c_uint(bit)))

    def ret_sets(self):
        """The de-duplicated end-id sets, in retlist order."""
        self._lib.fsm_hip_ret_count.restype = C.c_size_t
        out = []
        for k in range(self._lib.fsm_hip_ret_count(C.c_void_p(self._h))):
            p, n = C.c_void_p(), C.c_size_t()
            assert self._lib.fsm_hip_ret_get(C.c_void_p(self._h), C.c_uint32(k), C.byref(p), C.byref(n)) == 0
            out.append(np.frombuffer((C.c_char * (n.value * 4)).from_address(p.value), dtype=np.uint32).copy() if n.value else np.zeros(0, np.uint32))
        return out

    def ids_conflict(self):
        """AMBIG_ERROR's check: None, or the lowest end state that carries more than one end-id."""
        st = C.c_uint(0)
        r = self._lib.fsm_hip_ids_conflict(C.c_void_p(self._h), C.byref(st))
        if r < 0:
            raise _oserr("fsm_hip_ids_conflict")
        return int(st.value) if r == 1 else None

    def state_is_absorbing(self, state: int) -> bool:
        r = self._lib.fsm_hip_state_is_absorbing(C.c_void_p(self._h), C.c_uint32(state))
        if r < 0:
            raise _oserr("fsm_hip_state_is_absorbing")
        return bool(r)

    def match_file(self, path: str) -> int:
        """fsm_hip_match_file(dfa, FILE *) on a file opened with the C library."""
        libc = C.CDLL(None, use_errno=True)
        libc.fopen.restype = C.c_void_p
        libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        libc.fclose.argtypes = [C.c_void_p]
        f = libc.fopen(path.encode(), b"rb")
        if not f:
            raise OSError(C.get_errno(), "fopen")
        try:
            C.set_errno(0)
            r = self._lib.fsm_hip_match_file(C.c_void_p(self._h), C.c_void_p(f))
        finally:
            libc.fclose(f)
        if r < 0:
            raise _oserr("fsm_hip_match_file")
        return r

    def match_buffer_big(self, data: bytes):
        """fsm_hip_match_buffer_big: (1 / 0, caller's end state or NO_MATCH) of ONE input walked by the whole device."""
        e = C.c_uint32(NO_MATCH)
        buf = (C.c_char * max(len(data), 1)).from_buffer_copy(data if len(data) else b"\0")
        C.set_errno(0)
        r = self._lib.fsm_hip_match_buffer_big(C.c_void_p(self._h), buf, C.c_size_t(len(data)), C.byref(e))
        if r < 0:
            raise _oserr("fsm_hip_match_buffer_big")
        return r, e.value

    def match_last_passes(self):
        """(windows, passes) of the last match_file / match_buffer_big call of this process"""
        w, p_ = C.c_uint(0), C.c_uint(0)
        self._lib.fsm_hip_match_last_passes(C.byref(w), C.byref(p_))
        return w.value, p_.value

    def last_kernel_ms(self) -> float:
        return float(self._lib.fsm_hip_last_kernel_ms(self._h))

    # ---- end-ids --------------------------------------------------------------
    def endids(self, end_state: int) -> np.ndarray:
        n = self._lib.fsm_hip_endid_count(self._h, end_state)
        buf = np.zeros(max(n, 1), dtype=np.uint32)
        if self._lib.fsm_hip_endid_get(self._h, end_state, n, _ptr(buf)) != 1:
            raise RuntimeError("fsm_hip_endid_get")
        return buf[:n]


class HipNode:
    """struct fsm_hip_node *: one table replica per device, a batch sharded over them (include/fsm_hip.h)."""

    def __init__(self, flat: FlatDfa, devices: Optional[Sequence[int]] = None, flags: int = 0):
        self._lib = load_library()
        lib = self._lib
        lib.fsm_hip_node_create.restype = C.c_void_p
        lib.fsm_hip_node_dfa.restype = C.c_void_p
        lib.fsm_hip_node_bitmap_words.restype = C.c_size_t
        d = flat.desc()
        devs = (C.c_int * len(devices))(*devices) if devices else None
        C.set_errno(0)
        self._h = lib.fsm_hip_node_create(C.byref(d), C.c_uint(flags), devs, C.c_int(len(devices) if devices else 0))
        if not self._h:
            raise _oserr("fsm_hip_node_create")
        self.ndev = int(lib.fsm_hip_node_ndev(C.c_void_p(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fsm_hip_node_free(C.c_void_p(self._h))
            self._h = None

    __del__ = close

    def uses_rccl(self) -> bool:
        return bool(self._lib.fsm_hip_node_uses_rccl(C.c_void_p(self._h)))

    def rccl_path(self) -> str:
        self._lib.fsm_hip_node_rccl_path.restype = C.c_char_p
        return (self._lib.fsm_hip_node_rccl_path() or b"").decode()

    def replica(self, k: int) -> "HipDfa":
        """Borrowed view of the k-th replica (do not close it)."""
        h = self._lib.fsm_hip_node_dfa(C.c_void_p(self._h), C.c_int(k))
        if not h:
            raise _oserr("fsm_hip_node_dfa")
        return HipDfa(handle=h, borrowed=True)

    def shard(self, n: int, k: int):
        f, c = C.c_size_t(), C.c_size_t()
        self._lib.fsm_hip_node_shard(C.c_void_p(self._h), C.c_size_t(n), C.c_int(k), C.byref(f), C.byref(c))
        return int(f.value), int(c.value)

    def bitmap_words(self, n: int) -> int:
        return int(self._lib.fsm_hip_node_bitmap_words(C.c_void_p(self._h), C.c_size_t(n)))

    def exec_batch(self, data: np.ndarray, lens: Optional[np.ndarray] = None):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n, stride = data.shape
        end = np.empty(n, dtype=np.uint32)
        bm = np.zeros((n + 63) // 64, dtype=np.uint64)
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.uint32)
        C.set_errno(0)
        if self._lib.fsm_hip_node_exec_batch(C.c_void_p(self._h), _ptr(data), C.c_size_t(stride), _ptr(lens), C.c_size_t(n), _ptr(end), _ptr(bm)) != 0:
            raise _oserr("fsm_hip_node_exec_batch")
        return end, bm

    def exec_batch_offsets32(self, base: np.ndarray, off32: np.ndarray, want_bitmap: bool = True, want_end: bool = True):
        """fsm_hip_node_exec_batch_offsets32: u32 offsets (batches below 4 GiB), sharded over the devices."""
        base = np.ascontiguousarray(base, dtype=np.uint8)
        off32 = np.ascontiguousarray(off32, dtype=np.uint32)
        n = len(off32) - 1
        end = np.empty(n, dtype=np.uint32) if want_end else None
        bm = np.zeros((n + 63) // 64, dtype=np.uint64) if want_bitmap else None
        C.set_errno(0)
        if self._lib.fsm_hip_node_exec_batch_offsets32(C.c_void_p(self._h), C.c_void_p(base.ctypes.data if len(base) else None), C.c_void_p(off32.ctypes.data),
                                                  C.c_size_t(n), C.c_void_p(end.ctypes.data if want_end else None), C.c_void_p(bm.ctypes.data if want_bitmap else None)) != 0:
            raise _oserr("fsm_hip_node_exec_batch_offsets32")
        return end, bm

    def exec_batch_lengths(self, base: np.ndarray, lens: np.ndarray, want_bitmap: bool = True, want_end: bool = True):
        """fsm_hip_node_exec_batch_lengths: inputs packed back to back, their lengths and nothing else, sharded over the devices."""
        base = np.ascontiguousarray(base, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        n = len(lens)
        end = np.empty(n, dtype=np.uint32) if want_end else None
        bm = np.zeros((n + 63) // 64, dtype=np.uint64) if want_bitmap else None
        C.set_errno(0)
        if self._lib.fsm_hip_node_exec_batch_lengths(C.c_void_p(self._h), C.c_void_p(base.ctypes.data if len(base) else None), C.c_void_p(lens.ctypes.data if n else None),
                                                C.c_size_t(n), C.c_void_p(end.ctypes.data if want_end else None), C.c_void_p(bm.ctypes.data if want_bitmap else None)) != 0:
            raise _oserr("fsm_hip_node_exec_batch_lengths")
        return end, bm

    def exec_strings(self, strings: Sequence[bytes]):
        off = np.zeros(len(strings) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(s) for s in strings])
        base = np.frombuffer(b"".join(strings) or b"\0", dtype=np.uint8)
        n = len(strings)
        end = np.empty(n, dtype=np.uint32)
        bm = np.zeros((n + 63) // 64, dtype=np.uint64)
        C.set_errno(0)
        if self._lib.fsm_hip_node_exec_batch_offsets(C.c_void_p(self._h), _ptr(base), _ptr(off), C.c_size_t(n), _ptr(end), _ptr(bm)) != 0:
            raise _oserr("fsm_hip_node_exec_batch_offsets")
        return end, bm

    def exec_batch_device(self, d_base: Sequence[int], stride: int, n: int, d_end: Optional[Sequence[int]] = None,
                          d_bitmap_all: Optional[Sequence[int]] = None, want_count: bool = False):
        g = self.ndev
        vp = C.c_void_p * g
        base = vp(*[C.c_void_p(x) for x in d_base])
        ends = vp(*[C.c_void_p(x or None) for x in d_end]) if d_end is not None else None
        bms = vp(*[C.c_void_p(x) for x in d_bitmap_all]) if d_bitmap_all is not None else None
        cnt = C.c_uint64(0)
        C.set_errno(0)
        if self._lib.fsm_hip_node_exec_batch_device(C.c_void_p(self._h), base, C.c_size_t(stride), C.c_size_t(n), ends, bms,
                                                    C.byref(cnt) if want_count else None) != 0:
            raise _oserr("fsm_hip_node_exec_batch_device")
        return int(cnt.value) if want_count else None


class NodeBatch(C.Structure):
    """struct fsm_hip_node_batch (include/fsm_hip.h)."""
    _fields_ = [("d_base", C.POINTER(C.c_void_p)), ("stride", C.c_size_t), ("d_len", C.POINTER(C.c_void_p)), ("d_off", C.POINTER(C.c_void_p)),
                ("d_end_out", C.POINTER(C.c_void_p)), ("d_id_out", C.POINTER(C.c_void_p)), ("ids_mode", C.c_int),
                ("d_eager_out", C.POINTER(C.c_void_p)), ("d_bitmap_all", C.POINTER(C.c_void_p)), ("want_count", C.c_int)]


def _node_exec_device(self, n: int, d_base, stride: int = 0, d_len=None, d_off=None, d_end=None, d_ids=None, ids_mode: int = 1,
                      d_eager=None, d_bitmap_all=None, want_count: bool = False, async_: bool = False):
    """fsm_hip_node_exec_device: per-device lists of device pointers (None entries allowed where the header allows them)."""
    g = self.ndev

    def arr(xs):
        if xs is None:
            return None
        a = (C.c_void_p * g)(*[C.c_void_p(x or None) for x in xs])
        keep.append(a)
        return C.cast(a, C.POINTER(C.c_void_p))

    keep = []
    b = NodeBatch(arr(d_base), stride, arr(d_len), arr(d_off), arr(d_end), arr(d_ids), ids_mode, arr(d_eager), arr(d_bitmap_all), 1 if want_count else 0)
    cnt = C.c_uint64(0)
    C.set_errno(0)
    if self._lib.fsm_hip_node_exec_device(C.c_void_p(self._h), C.byref(b), C.c_size_t(n), C.byref(cnt) if (want_count and not async_) else None,
                                          C.c_int(1 if async_ else 0)) != 0:
        raise _oserr("fsm_hip_node_exec_device")
    return int(cnt.value) if (want_count and not async_) else None


def _node_wait(self, want_count: bool = False):
    cnt = C.c_uint64(0)
    C.set_errno(0)
    if self._lib.fsm_hip_node_wait(C.c_void_p(self._h), C.byref(cnt) if want_count else None) != 0:
        raise _oserr("fsm_hip_node_wait")
    return int(cnt.value) if want_count else None


def _node_exec_batch_ids(self, data: np.ndarray, mode: int, lens: Optional[np.ndarray] = None) -> np.ndarray:
    data = np.ascontiguousarray(data, dtype=np.uint8)
    n, stride = data.shape
    out = np.empty(n, dtype=np.uint32)
    if lens is not None:
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
    C.set_errno(0)
    if self._lib.fsm_hip_node_exec_batch_ids(C.c_void_p(self._h), _ptr(data), C.c_size_t(stride), _ptr(lens), C.c_size_t(n), C.c_int(mode), _ptr(out)) != 0:
        raise _oserr("fsm_hip_node_exec_batch_ids")
    return out


HipNode.exec_device = _node_exec_device
HipNode.wait = _node_wait
HipNode.exec_batch_ids = _node_exec_batch_ids


class MultiBatch(C.Structure):
    """struct fsm_hip_multi_batch (include/fsm_hip.h)"""
    _fields_ = [("base", C.c_void_p), ("off", C.c_void_p), ("n", C.c_size_t), ("end_out", C.c_void_p), ("accept_bitmap", C.c_void_p)]


def exec_multi(dfas: Sequence["HipDfa"], jobs: Sequence[Sequence[bytes]], want_bitmap: bool = True, nodes: Optional[Sequence["HipNode"]] = None):
    """fsm_hip_exec_multi: job q = the strings jobs[q] through dfas[q]; ONE submission.  Returns [(end, bitmap), ...].
    nodes: fsm_hip_node_exec_multi instead (the list sharded by DFA over the nodes' devices)."""
    lib = load_library()
    k = len(jobs)
    keep, arr = [], (MultiBatch * max(k, 1))()
    outs = []
    for q, strs in enumerate(jobs):
        n = len(strs)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in strs])
        base = np.frombuffer(b"".join(strs) or b"\0", dtype=np.uint8)
        end = np.full(n, 0xDEADBEEF, dtype=np.uint32)
        bm = np.zeros((n + 63) // 64, dtype=np.uint64) if want_bitmap else None
        keep += [off, base, end, bm]
        arr[q].base, arr[q].off, arr[q].n = base.ctypes.data, off.ctypes.data, n
        arr[q].end_out = end.ctypes.data if n else None
        arr[q].accept_bitmap = bm.ctypes.data if (want_bitmap and n) else None
        outs.append((end, bm))
    C.set_errno(0)
    if nodes is not None:
        hs = (C.c_void_p * max(k, 1))(*[nd._h for nd in nodes])
        if lib.fsm_hip_node_exec_multi(hs, arr, C.c_size_t(k)) != 0:
            raise _oserr("fsm_hip_node_exec_multi")
    else:
        hs = (C.c_void_p * max(k, 1))(*[d._h for d in dfas])
        if lib.fsm_hip_exec_multi(hs, arr, C.c_size_t(k)) != 0:
            raise _oserr("fsm_hip_exec_multi")
    return outs


def exec_multi_device(dfas: Sequence["HipDfa"], jobs: Sequence[tuple], stream: int = 0):
    """fsm_hip_exec_multi_device: jobs[q] = (d_base, d_off, n, d_end, d_bitmap) device pointers (0 = NULL)."""
    lib = load_library()
    k = len(jobs)
    arr = (MultiBatch * max(k, 1))()
    for q, (b, o, n, e, m) in enumerate(jobs):
        arr[q].base, arr[q].off, arr[q].n, arr[q].end_out, arr[q].accept_bitmap = b or None, o or None, n, e or None, m or None
    hs = (C.c_void_p * max(k, 1))(*[d._h for d in dfas])
    C.set_errno(0)
    if lib.fsm_hip_exec_multi_device(hs, arr, C.c_size_t(k), C.c_void_p(stream or None)) != 0:
        raise _oserr("fsm_hip_exec_multi_device")


class MultiBatchIds(C.Structure):
    _fields_ = [("base", C.c_void_p), ("off", C.c_void_p), ("n", C.c_size_t), ("end_out", C.c_void_p), ("accept_bitmap", C.c_void_p), ("id_out", C.c_void_p)]


def exec_multi_ids(dfas: Sequence["HipDfa"], jobs: Sequence[Sequence[bytes]], ids_mode: int):
    """fsm_hip_exec_multi_ids: job q = the strings jobs[q] through dfas[q], ONE submission, end-ids by the device.
    Returns [(end, bitmap, ids), ...]."""
    lib = load_library()
    k = len(jobs)
    keep, arr, outs = [], (MultiBatchIds * max(k, 1))(), []
    for q, strs in enumerate(jobs):
        n = len(strs)
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in strs])
        base = np.frombuffer(b"".join(strs) or b"\0", dtype=np.uint8)
        end = np.full(n, 0xDEADBEEF, dtype=np.uint32)
        ids = np.full(n, 0xDEADBEEF, dtype=np.uint32)
        bm = np.zeros((n + 63) // 64, dtype=np.uint64)
        keep += [off, base, end, bm, ids]
        arr[q].base, arr[q].off, arr[q].n = base.ctypes.data, off.ctypes.data, n
        arr[q].end_out, arr[q].accept_bitmap, arr[q].id_out = (end.ctypes.data, bm.ctypes.data, ids.ctypes.data) if n else (None, None, None)
        outs.append((end, bm, ids))
    hs = (C.c_void_p * max(k, 1))(*[d._h for d in dfas])
    C.set_errno(0)
    if lib.fsm_hip_exec_multi_ids(hs, arr, C.c_size_t(k), C.c_int(ids_mode)) != 0:
        raise _oserr("fsm_hip_exec_multi_ids")
    return outs


def exec_multi_ids_device(dfas: Sequence["HipDfa"], jobs: Sequence[tuple], ids_mode: int, stream: int = 0):
    """fsm_hip_exec_multi_ids_device: jobs[q] = (d_base, d_off, n, d_end, d_bitmap, d_ids) device pointers (0 = NULL)."""
    lib = load_library()
    k = len(jobs)
    arr = (MultiBatchIds * max(k, 1))()
    for q, (b, o, n, e, m, i) in enumerate(jobs):
        arr[q].base, arr[q].off, arr[q].n, arr[q].end_out, arr[q].accept_bitmap, arr[q].id_out = b or None, o or None, n, e or None, m or None, i or None
    hs = (C.c_void_p * max(k, 1))(*[d._h for d in dfas])
    C.set_errno(0)
    if lib.fsm_hip_exec_multi_ids_device(hs, arr, C.c_size_t(k), C.c_int(ids_mode), C.c_void_p(stream or None)) != 0:
        raise _oserr("fsm_hip_exec_multi_ids_device")


class MultiPrepared:
    """fsm_hip_multi_prepare / _launch / _prepared_free: a device-pointer submission put on the device once; launch() is
    one kernel launch on the stream (capturable into a HIP graph).  jobs[q] = (d_base, d_off, n, d_end, d_bitmap, d_ids)."""

    def __init__(self, dfas: Sequence["HipDfa"], jobs: Sequence[tuple], ids_mode: int = 0):
        lib = load_library()
        k = len(jobs)
        arr = (MultiBatchIds * max(k, 1))()
        for q, (b, o, n, e, m, i) in enumerate(jobs):
            arr[q].base, arr[q].off, arr[q].n, arr[q].end_out, arr[q].accept_bitmap, arr[q].id_out = b or None, o or None, n, e or None, m or None, i or None
        hs = (C.c_void_p * max(k, 1))(*[d._h for d in dfas])
        self._keep = list(dfas)
        self._p = C.c_void_p()
        C.set_errno(0)
        if lib.fsm_hip_multi_prepare(hs, arr, C.c_size_t(k), C.c_int(ids_mode), C.byref(self._p)) != 0:
            raise _oserr("fsm_hip_multi_prepare")

    def launch(self, stream: int = 0) -> None:
        C.set_errno(0)
        if load_library().fsm_hip_multi_launch(self._p, C.c_void_p(stream or None)) != 0:
            raise _oserr("fsm_hip_multi_launch")

    def close(self) -> None:
        if self._p:
            lib = load_library()
            lib.fsm_hip_multi_prepared_free.restype = None
            lib.fsm_hip_multi_prepared_free(self._p)
            self._p = C.c_void_p()


def multi_last_launches() -> int:
    lib = load_library()
    lib.fsm_hip_multi_last_launches.restype = C.c_uint
    return int(lib.fsm_hip_multi_last_launches())


def multi_last_fused_jobs() -> int:
    lib = load_library()
    lib.fsm_hip_multi_last_fused_jobs.restype = C.c_uint
    return int(lib.fsm_hip_multi_last_fused_jobs())


def multi_assign(cost: Sequence[int], ndev: int) -> np.ndarray:
    """fsm_hip_multi_assign: which device takes which job of a many-DFA submission (largest first, least loaded device).  Host arithmetic only."""
    lib = load_library()
    c = np.ascontiguousarray(cost, dtype=np.uint64)
    out = np.zeros(len(c), dtype=np.int32)
    C.set_errno(0)
    if lib.fsm_hip_multi_assign(C.c_void_p(c.ctypes.data if len(c) else None), C.c_size_t(len(c)), C.c_int(ndev), C.c_void_p(out.ctypes.data if len(c) else None)) != 0:
        raise _oserr("fsm_hip_multi_assign")
    return out


def _gen_args(alphabet, plant):
    a = np.frombuffer(bytes(alphabet), dtype=np.uint8).copy() if alphabet else None
    p = np.frombuffer(bytes(plant), dtype=np.uint8).copy() if plant else None
    return a, p


def gen_inputs_host(n: int, stride: int, first_index: int = 0, seed: int = 0x5EEDF5A1, alphabet: Optional[bytes] = None,
                    plant: Optional[bytes] = None, plant_every: int = 0) -> np.ndarray:
    lib = load_library()
    out = np.empty((n, stride), dtype=np.uint8)
    a, p = _gen_args(alphabet, plant)
    lib.fsm_hip_gen_inputs_host(_ptr(out), stride, n, first_index, seed, _ptr(a), len(a) if a is not None else 0,
                                _ptr(p), len(p) if p is not None else 0, plant_every)
    return out


def gen_inputs_device(d_base: int, n: int, stride: int, first_index: int = 0, seed: int = 0x5EEDF5A1,
                      alphabet: Optional[bytes] = None, plant: Optional[bytes] = None, plant_every: int = 0, stream: int = 0):
    lib = load_library()
    a, p = _gen_args(alphabet, plant)
    C.set_errno(0)
    if lib.fsm_hip_gen_inputs_device(d_base, stride, n, first_index, seed, _ptr(a), len(a) if a is not None else 0,
                                     _ptr(p), len(p) if p is not None else 0, plant_every, stream or None) != 0:
        raise _oserr("fsm_hip_gen_inputs_device")


def gen_pack_rows_device(d_rows: int, stride: int, d_len: int, d_off: int, n: int, max_len: int, d_out: int, stream: int = 0):
    """fsm_hip_gen_pack_rows_device: input i = the first len[i] bytes of row i, packed at d_out + off[i]."""
    lib = load_library()
    C.set_errno(0)
    if lib.fsm_hip_gen_pack_rows_device(C.c_void_p(d_rows), C.c_size_t(stride), C.c_void_p(d_len), C.c_void_p(d_off), C.c_size_t(n),
                                        C.c_size_t(max_len), C.c_void_p(d_out), C.c_void_p(stream or None)) != 0:
        raise _oserr("fsm_hip_gen_pack_rows_device")


def gather_probe_ms(d_base: int, nbytes: int, ngathers: int, vec_bytes: int, d_scratch4: int, stream: int = 0) -> float:
    """fsm_hip_gather_probe_ms: one launch of `ngathers` independent vec_bytes-byte loads at pseudo-random offsets of the buffer."""
    lib = load_library()
    lib.fsm_hip_gather_probe_ms.restype = C.c_double
    ms = lib.fsm_hip_gather_probe_ms(C.c_void_p(d_base), C.c_size_t(nbytes), C.c_size_t(ngathers), C.c_int(vec_bytes), C.c_void_p(d_scratch4), C.c_void_p(stream or None))
    if ms < 0:
        raise _oserr("fsm_hip_gather_probe_ms")
    return float(ms)


def lds_chain_probe_gbps(table_bytes: int, waves: int, blocks_per_cu: int, steps: int, d_scratch4: int, stream: int = 0) -> float:
    """fsm_hip_lds_chain_probe_gbps: what a dependent chain of random LDS reads sustains (the lookup layouts' ceiling)."""
    lib = load_library()
    lib.fsm_hip_lds_chain_probe_gbps.restype = C.c_double
    C.set_errno(0)
    r = lib.fsm_hip_lds_chain_probe_gbps(C.c_size_t(table_bytes), C.c_int(waves), C.c_int(blocks_per_cu), C.c_size_t(steps), C.c_void_p(d_scratch4), C.c_void_p(stream or None))
    if r < 0:
        raise _oserr("fsm_hip_lds_chain_probe_gbps")
    return float(r)


def waves_by_occupancy(vgprs: int, workgroups_by_lds: int, max_waves: int = 16) -> int:
    """fsm_hip_waves_by_occupancy: the workgroup size (wavefronts) a latency-bound per-lane kernel gets (pure arithmetic)."""
    lib = load_library()
    lib.fsm_hip_waves_by_occupancy.restype = C.c_int
    return int(lib.fsm_hip_waves_by_occupancy(C.c_int(vgprs), C.c_int(workgroups_by_lds), C.c_int(max_waves)))


def pack_affixes(items: Sequence[bytes]) -> np.ndarray:
    """[len<=7, b0..b6] entries for the affix generator."""
    t = np.zeros((len(items), 8), dtype=np.uint8)
    for i, s in enumerate(items):
        assert len(s) <= 7
        t[i, 0] = len(s)
        t[i, 1:1 + len(s)] = np.frombuffer(s, dtype=np.uint8)
    return t


def gen_affix_inputs_host(n: int, stride: int, first_index: int, seed: int, alphabet: bytes, body: bytes,
                          prefixes: Sequence[bytes], suffixes: Sequence[bytes], every: int = 2, body2: Optional[bytes] = None) -> np.ndarray:
    """body2: alternate body / body2 byte by byte after the prefix (fsm_hip_gen_affix2_inputs_host)."""
    lib = load_library()
    lib.fsm_hip_gen_affix2_inputs_host.restype = None
    out = np.empty((n, stride), dtype=np.uint8)
    a, b = _gen_args(alphabet, body)
    b2 = np.frombuffer(bytes(body2), dtype=np.uint8).copy() if body2 else None
    p, s = pack_affixes(prefixes), pack_affixes(suffixes)
    lib.fsm_hip_gen_affix2_inputs_host(_ptr(out), C.c_size_t(stride), C.c_size_t(n), C.c_uint64(first_index), C.c_uint64(seed),
                                       _ptr(a), C.c_uint(len(a)), _ptr(b), C.c_uint(len(b)), _ptr(b2), C.c_uint(len(b2) if b2 is not None else 0),
                                       _ptr(p), C.c_uint(len(p)), _ptr(s), C.c_uint(len(s)), C.c_uint(every))
    return out


def gen_affix_inputs_device(d_base: int, n: int, stride: int, first_index: int, seed: int, alphabet: bytes, body: bytes,
                            prefixes: Sequence[bytes], suffixes: Sequence[bytes], every: int = 2, stream: int = 0, body2: Optional[bytes] = None):
    lib = load_library()
    a, b = _gen_args(alphabet, body)
    b2 = np.frombuffer(bytes(body2), dtype=np.uint8).copy() if body2 else None
    p, s = pack_affixes(prefixes), pack_affixes(suffixes)
    C.set_errno(0)
    r = lib.fsm_hip_gen_affix2_inputs_device(C.c_void_p(d_base), C.c_size_t(stride), C.c_size_t(n), C.c_uint64(first_index),
                                             C.c_uint64(seed), _ptr(a), C.c_uint(len(a)), _ptr(b), C.c_uint(len(b)),
                                             _ptr(b2), C.c_uint(len(b2) if b2 is not None else 0),
                                             _ptr(p), C.c_uint(len(p)), _ptr(s), C.c_uint(len(s)), C.c_uint(every),
                                             C.c_void_p(stream or None))
    if r != 0:
        raise _oserr("fsm_hip_gen_affix2_inputs_device")


def stream_read_probe_gbps(d_base: int, nbytes: int, d_scratch4: int, reps: int = 3, stream: int = 0) -> float:
    """GB/s of a trivially coalesced read-only kernel over the same buffer (measured HBM ceiling)."""
    lib = load_library()
    lib.fsm_hip_stream_read_probe_ms.restype = C.c_double
    ms = lib.fsm_hip_stream_read_probe_ms(C.c_void_p(d_base), C.c_size_t(nbytes), C.c_void_p(d_scratch4), C.c_int(reps), C.c_void_p(stream or None))
    if ms <= 0:
        raise _oserr("fsm_hip_stream_read_probe_ms")
    return nbytes / (ms * 1e-3) / 1e9
