"""CPU: the C-ABI library loads and exports every symbol include/*.h declares, and
fails LOUDLY (ENODEV) instead of falling back to the CPU when there is no GPU."""
import ctypes
import errno
import os
import re

import numpy as np
import pytest

from common import GOLDEN, Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for h in ("fsm_hip.h", "fsm_hip_plan.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        syms |= set(re.findall(r"\b(fsm_hip_\w+)\s*\(", txt))
    return syms


def test_exports_every_declared_symbol(built):
    from libfsm_amd import load_library
    lib = load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in sorted(syms) if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.fsm_hip_version() >= 100


def test_no_libfsm_link_dependency(built):
    """The shim binds libfsm by dlsym at run time: the .so itself must not need libfsm."""
    import subprocess
    from libfsm_amd import LIB_PATH
    out = subprocess.check_output(["readelf", "-d", LIB_PATH]).decode()
    assert "libfsm" not in out.replace("libfsm_hip", "")
    assert "liboracle" not in out and "_ref" not in out


def test_no_cpu_fallback_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from libfsm_amd import HipDfa
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    with pytest.raises(OSError) as ei:
        HipDfa(g.flat)
    assert ei.value.errno == errno.ENODEV


def test_product_sources_do_not_touch_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may use oracle/."""
    for d, _, files in os.walk(os.path.join(ROOT, "libfsm_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".cpp", ".h", ".hip", ".sh")):
                txt = open(os.path.join(d, f), errors="replace").read()
                assert "pyoracle" not in txt and "dfa_oracle" not in txt and "oracle/" not in txt.replace("see oracle/", ""), (d, f)


def test_generator_host_properties(built):
    from libfsm_amd import gen_inputs_host
    a = gen_inputs_host(64, 256, 0, 1234)
    b = gen_inputs_host(32, 256, 32, 1234)
    assert np.array_equal(a[32:], b)                      # counter based: rows depend on global index only
    assert not np.array_equal(a[0], a[1])
    h = np.bincount(gen_inputs_host(512, 1024, 0, 99).reshape(-1), minlength=256)
    assert h.min() > 1500 and h.max() < 2600              # ~2048 expected per value
    c = gen_inputs_host(64, 128, 0, 5, b"abc")
    assert set(np.unique(c)) <= set(b"abc")
    d = gen_inputs_host(64, 128, 0, 5, None, b"Libfsm", 8)
    for i in range(64):
        assert (b"Libfsm" in bytes(d[i])) or i % 8 != 0


def test_on_disk_description_roundtrip(built, tmp_path):
    """fsm_hip_desc_write / fsm_hip_desc_read ("FSMHIP01") reproduce every golden description
    exactly, and reject corrupt or truncated files with EINVAL."""
    import errno as _errno
    from common import all_golden_paths
    from libfsm_amd import FlatDfa
    for k, path in enumerate(all_golden_paths()[::4] + [os.path.join(GOLDEN, "c3.npz")]):
        flat = Golden(path).flat
        p = str(tmp_path / f"d{k}.fsmhip")
        flat.write_c(p)
        back = FlatDfa.read_c(p)
        assert (back.nstates, back.start) == (flat.nstates, flat.start)
        for f in ("edge_off", "is_end", "endid_off", "endids"):
            assert np.array_equal(getattr(back, f), getattr(flat, f)), f
        assert np.array_equal(back.ranges, flat.ranges)
    raw = open(p, "rb").read()
    assert raw[:8] == b"FSMHIP01"
    for bad in (raw[:len(raw) // 2], b"XXXXXXXX" + raw[8:], raw[:12] + b"\xff\xff\xff\x7f" + raw[16:]):
        q = str(tmp_path / "bad.fsmhip")
        open(q, "wb").write(bad)
        with pytest.raises(OSError) as ei:
            FlatDfa.read_c(q)
        assert ei.value.errno == _errno.EINVAL


def test_no_flat_instruction_in_any_walk_kernel(built):
    """Every load of the walk kernels names its address space: LDS (ds_read) or device memory (global_load / buffer_load).  A
    FLAT load whose lanes split between LDS and memory completes over two paths and two wait counters; the two intermittent
    wrong-answer builds in this project's history were kernels with such loads inside lane-divergent loops
    (profiles/r08i_*).  Round 6 removed them all; this keeps it so.  (The input generators are not the
    product: gen_affix_kernel reads its by-value argument arrays with flat loads.)"""
    import re
    import subprocess
    import tempfile
    import libfsm_amd
    so = os.path.join(os.path.dirname(libfsm_amd.__file__), "libfsm_hip.so")
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "clang-offload-bundler")):
        pytest.skip("no ROCm LLVM tools here")
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        data = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        offs = [m.start() for m in re.finditer(re.escape(magic), data)] + [len(data)]
        assert len(offs) > 4
        bad, nkernels = {}, 0
        for k in range(len(offs) - 1):
            b, e = os.path.join(td, f"b{k}.bundle"), os.path.join(td, f"b{k}.elf")
            open(b, "wb").write(data[offs[k]:offs[k + 1]])
            subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + b,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + e])
            dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", e], capture_output=True, text=True).stdout
            cur = None
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    nkernels += "walk_" in cur
                elif cur and "walk_" in cur and re.search(r"\bflat_(load|store|atomic)", line):
                    bad[cur] = bad.get(cur, 0) + 1
        assert nkernels > 150, nkernels
        assert not bad, {k[:80]: v for k, v in list(bad.items())[:6]}
