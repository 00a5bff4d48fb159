#!/usr/bin/env python3
"""tools/packed_model.py -- scalar model of walk_packed's per-lane algorithm (libfsm_amd/csrc/walk_packed.h), used to
check the index logic on the CPU before it is transcribed to HIP: rows of R bytes owned by one lane each, tiles of 64
rows, first[] (the first input starting in a row), the tile's start-bit mask (bits beyond the tile are not kept: the
one boundary that can matter there is the tail), the lane's cur / nxt input cursor, empties stepped over through
kbits, the run past the row's and the tile's end.  Compares against a plain per-input walk of a toy DFA.
Not part of the product or of the oracle."""
import sys

import numpy as np

NONE = -1


def plain(table, start, data, off):
    out = []
    for j in range(len(off) - 1):
        s = start
        for b in data[off[j]:off[j + 1]]:
            s = table[s][b]
        out.append(s)
    return out


def model(table, start, data, off, base_addr, rshift, lanes=64):
    n = len(off) - 1
    a_first = base_addr + off[0]
    a_last = base_addr + off[n]
    A0 = a_first & ~127
    Aend = (a_last + 127) & ~127
    nrows = ((a_last - A0) >> rshift) + 1
    first = [None] * (nrows + 1)
    out = [None] * n
    kbits = [False] * (n + 65)
    for j in range(n + 1):                      # packed_first
        hi = nrows if j == n else (base_addr + off[j] - A0) >> rshift
        lo = 0 if j == 0 else ((base_addr + off[j - 1] - A0) >> rshift) + 1
        for v in range(lo, hi + 1):
            assert first[v] is None
            first[v] = j
        if j < n and off[j + 1] == off[j]:
            kbits[j] = True
            out[j] = start
    assert all(f is not None for f in first), first
    emp = any(kbits)

    def skip(i):
        while kbits[i]:
            i += 1
        return i

    def byte_at(addr):
        assert A0 <= addr < Aend, "read outside the batch's lines"
        k = addr - base_addr
        return data[k] if 0 <= k < len(data) else 0xAA

    ntiles = (nrows + lanes - 1) // lanes
    tile_bytes = lanes << rshift
    for tile in range(ntiles):
        fl = [(first[min(tile * lanes + l, nrows)], first[min(tile * lanes + l + 1, nrows)]) for l in range(lanes)]
        e0, e1 = fl[0][0], fl[lanes - 1][1]
        tileaddr = A0 + tile * tile_bytes
        mask = set()
        for e in range(e0, e1 + 1):
            o = base_addr + off[e] - tileaddr
            assert o >= 0
            if o < tile_bytes:
                mask.add(o)
        tail = base_addr + off[e1] - tileaddr
        for l in range(lanes):
            fst, lim = fl[l]
            act = fst < lim
            cur, nxt = NONE, fst
            if emp and act:
                nxt = skip(nxt)
            st = start
            rowrel = l << rshift
            rpos = 0
            while act:
                ra = tileaddr + rowrel + rpos
                assert ra <= Aend
                for c in range(8):
                    cpos = rowrel + rpos + 16 * c             # tile-relative
                    inside = (l + (rpos >> rshift)) < lanes
                    bm = 0
                    for k in range(16):
                        if inside:
                            if cpos + k in mask:
                                bm |= 1 << k
                        elif cpos + k == tail:
                            bm |= 1 << k
                    pre = [byte_at(tileaddr + cpos + k) if tileaddr + cpos + k < Aend else 0 for k in range(16)]
                    prevc = st
                    cd = []
                    for k in range(16):
                        if (bm >> k) & 1:
                            st = start
                        st = table[st][pre[k]]
                        cd.append(st)
                    for k in range(16):
                        if not (bm >> k) & 1:
                            continue
                        code = prevc if k == 0 else cd[k - 1]
                        if cur != NONE:
                            assert out[cur] is None
                            out[cur] = code
                        cur = nxt
                        nxt += 1
                        if cur >= lim:
                            cur = NONE
                            act = False
                            break
                        elif emp:
                            nxt = skip(nxt)
                    if not act:
                        break
                rpos += 128
    assert all(o is not None for o in out), [i for i, o in enumerate(out) if o is None][:10]
    return out


def main():
    rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    S = 7
    table = rng.randint(0, S, (S, 256))
    for trial in range(300):
        kind = trial % 8
        n = rng.randint(1, 80)
        if kind == 0:
            lens = rng.randint(0, 40, n)
        elif kind == 1:
            lens = rng.randint(0, 3, n)
        elif kind == 2:
            lens = np.where(rng.randint(0, 5, n) == 0, rng.randint(0, 700, n), rng.randint(0, 20, n))
        elif kind == 3:
            lens = np.zeros(n, int)
        elif kind == 4:
            lens = rng.randint(8, 65, n)
        elif kind == 5:
            lens = np.full(n, rng.choice([1, 15, 16, 17, 128, 127, 129, 512]))
        elif kind == 6:
            lens = rng.randint(0, 1025, n)
        else:
            lens = np.where(rng.randint(0, 3, n) == 0, 0, rng.randint(0, 33, n))
        lead = rng.randint(0, 50)           # off[0] need not be 0
        off = np.zeros(n + 1, int)
        off[0] = lead
        off[1:] = lead + np.cumsum(lens)
        data = rng.randint(0, 256, int(off[-1])).astype(np.uint8)
        base_addr = 4096 * 10 + rng.randint(0, 300)
        want = plain(table, 3, data, off)
        for rshift in (7, 8, 9):
            for lanes in (4, 64):           # few lanes per tile: the tail beyond the tile is exercised
                got = model(table, 3, data, off, base_addr, rshift, lanes)
                assert got == want, (trial, kind, rshift, lanes)
    print("packed model ok")


if __name__ == "__main__":
    main()
