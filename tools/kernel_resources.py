#!/usr/bin/env python3
"""tools/kernel_resources.py -- register / scratch use of every walk kernel instantiation.

Compiles the kernel translation units device-only for gfx950 (no GPU needed), unbundles the code
objects and reads the AMDGPU metadata notes: VGPRs, AGPRs, SGPRs, spilled VGPRs/SGPRs, private
(scratch) bytes.  Prints one line per kernel and a summary; exits non-zero if any kernel spills
VGPRs or uses scratch (--strict).  usage: kernel_resources.py [--strict] [out.txt]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libfsm_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
UNITS = ["kern_tiny", "kern_lds", "kern_comb", "kern_glob", "kern_glob16", "multi"]
# --strict also holds the kernels the bench lines run to a bound on SGPR spills (scalar registers parked in vector lanes: an
# instruction each way, and in the short-lines kernels they sat inside the per-tile path -- 23-37 of them in round 4)
HOT_SSPILL = [("walk_lines32<", 8), ("walk_ldsdma<CombSelfPol, 128, 2, 768>", 0), ("walk_ldsdma<Tiny5Pol, 128, 2, 1024>", 0),
              ("walk_direct<Comb256Pol, 8, 1>", 0), ("walk_lazy<", 8), ("walk_ragged<Tiny5Pol, 768, 0>", 24), ("walk_ragged<CombSelfPol, 768, 0>", 48),
              ("walk_direct<Glob16Pol, 4, 2>", 0)]
# the one kernel that is allowed scratch: the A/B form of the lines kernel with the lookups eight at a time (32 bytes outside the step
# block; the shipped <., 3, 3, 2> has none)
SCRATCH_OK = [("walk_lazy_lines<false, 3, 2, 8>", 64), ("walk_lazy_lines<true, 3, 2, 8>", 64)]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main():
    strict = "--strict" in sys.argv
    outp = [a for a in sys.argv[1:] if not a.startswith("--")]
    tmp = tempfile.mkdtemp(prefix="kres")
    procs = []
    for u in UNITS:
        co = os.path.join(tmp, u + ".bundle")
        procs.append((u, co, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only",
                                               "-c", os.path.join(CSRC, u + ".hip"), "-o", co])))
    rows = []
    for u, co, p in procs:
        assert p.wait() == 0, u
        elf = co + ".elf"
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + co,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            f = dict(re.findall(r"\.(\w+):\s+(\S+)", ".agpr_count:" + blk.split("\n    - .agpr_count")[0]))
            if "name" not in f:
                continue
            rows.append((u, f["name"], int(f.get("vgpr_count", 0)), int(f.get("agpr_count", 0)), int(f.get("sgpr_count", 0)),
                         int(f.get("vgpr_spill_count", 0)), int(f.get("sgpr_spill_count", 0)), int(f.get("private_segment_fixed_size", 0))))
    names = demangle([r[1] for r in rows])
    lines = []
    bad = hot_bad = 0
    for r, nm in zip(rows, names):
        nm = nm.replace("fsmhip::", "").replace("(fsmhip::WalkArgs)", "").replace("void ", "")
        flag = ""
        if r[5] or r[7]:
            ok = [lim for pat, lim in SCRATCH_OK if pat in nm]
            if ok and r[7] <= ok[0]:
                flag = f"  (scratch {r[7]} B: allowed, see SCRATCH_OK)"
            else:
                flag = "  <-- spills/scratch"
                bad += 1
        for pat, lim in HOT_SSPILL:
            if pat in nm and r[6] > lim:
                flag += f"  <-- hot kernel: {r[6]} SGPR spills > {lim}"
                hot_bad += 1
        lines.append(f"{r[0]:10s} vgpr={r[2]:3d} agpr={r[3]:3d} sgpr={r[4]:3d} vspill={r[5]:3d} sspill={r[6]:3d} scratch={r[7]:4d}  {nm}{flag}")
    lines.sort()
    lines.append(f"# {len(rows)} kernels, {bad} with VGPR spills or scratch, {hot_bad} hot kernels over their SGPR-spill bound")
    text = "\n".join(lines)
    print(text)
    if outp:
        open(outp[0], "w").write(text + "\n")
    if strict and (bad or hot_bad):
        sys.exit(1)


if __name__ == "__main__":
    main()
