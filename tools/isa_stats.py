#!/usr/bin/env python3
"""tools/isa_stats.py FILE.s SUBSTRING [NBLOCKS] -- registers / spills / code size of the kernels whose mangled name contains
SUBSTRING in a hipcc --save-temps assembly file, and an instruction histogram of their largest basic blocks (the unrolled
hot loops).  No GPU needed."""
import re
import sys
from collections import Counter


def main():
    path, sub = sys.argv[1], sys.argv[2]
    nblocks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", lines[i])
        if not m or sub not in m.group(1):
            i += 1
            continue
        name = m.group(1)
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i + 1:j]
        k = j
        meta = {}
        while k < len(lines) and k < j + 120:
            mm = re.match(r"^;\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy|codeLenInByte|SGPRSpill|VGPRSpill|LDSByteSize)\D*(\d+)", lines[k].replace("=", ":"))
            if mm:
                meta[mm.group(1)] = int(mm.group(2))
            if re.match(r"^(_Z\w+):", lines[k]):
                break
            k += 1
        print(name[:100])
        print("   ", meta)
        blocks, cur = [], ["entry", []]
        blocks.append(cur)
        for l in body:
            if re.match(r"^\.LBB\d+_\d+:", l):
                cur = [l.split(":")[0], []]
                blocks.append(cur)
            elif l.startswith("\t") and not l.strip().startswith((".", ";")):
                cur[1].append(l.strip())
        tot = Counter()
        for b in blocks:
            tot.update(x.split()[0] for x in b[1])
        print("    instructions:", sum(tot.values()), "waitcnt:", tot.get("s_waitcnt", 0), "readlane/writelane:", tot.get("v_readlane_b32", 0) + tot.get("v_writelane_b32", 0))
        for b in sorted(blocks, key=lambda b: -len(b[1]))[:nblocks]:
            c = Counter(x.split()[0] for x in b[1])
            cls = lambda p: sum(v for kk, v in c.items() if kk.startswith(p))
            print(f"    block {b[0]}: {len(b[1])} instr: valu {cls('v_')} salu {cls('s_')} lds {cls('ds_')} vmem {cls('buffer_') + cls('global_') + cls('flat_') + cls('scratch_')}")
            print("       ", ", ".join(f"{kk} {v}" for kk, v in c.most_common(28)))
        i = j


if __name__ == "__main__":
    main()
