#!/usr/bin/env python3
"""tests/tools/round5_stress.py -- repeated launches of round 5's new paths against the oracle:
  1. walk_lazy_lines (the lazy walk on the variable-length fronts): literal sets of three shapes; u64 offsets, u32 offsets, lengths
     alone, stride + lengths, resume in two pieces; two line sets alternate in the same device buffers (a result left over from the
     previous launch is a wrong one); end states AND accept bitmap compared on the device;
  2. fsm_hip_exec_multi: random subsets of the golden vectors in random order, host front (the staging block is reused call after
     call) and device front, against the reference's frozen answers.
REPS launches per (automaton, line mix, front); exits non-zero on any mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pack(strs):
    off = np.zeros(len(strs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in strs])
    return np.frombuffer(b"".join(strs) + b"\0" * 16, np.uint8).copy(), off


def main():
    import torch
    import libfsm_amd as hip
    from common import Golden, all_golden_paths
    from oracle.pyoracle import Oracle
    hip.load_library()
    torch.cuda.set_device(0)
    reps = int(os.environ.get("REPS", 100))
    bad = launches = 0
    shapes = [(b"abcd", 400, 3, 9, 2), (b"abcdefghijklmnop", 4000, 5, 9, 0),
              (b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-", 20000, 8, 16, 2)]
    for alpha_b, nw, lo, hi, flags in shapes:
        rng = np.random.RandomState(len(alpha_b) + nw)
        al = np.frombuffer(alpha_b, np.uint8)
        words = sorted(set(bytes(al[rng.randint(0, len(al), rng.randint(lo, hi + 1))]) for _ in range(nw)))
        flat = hip.FlatDfa.from_strings(words, flags, list(range(len(words))))
        o = Oracle(flat)
        dfa = hip.HipDfa(flat, hip.LAYOUT_SPARSE)
        dfa.tune(hip.KNOB_SPARSE_FAST, 3)
        n = 9000 + 41
        for mix, (l0, l1) in (("0-600", (0, 600)), ("0-90", (0, 90)), ("8-16", (8, 16))):
            sets = []
            for v in range(2):
                strs = []
                for i in range(n):
                    L = int(rng.randint(l0, l1 + 1))
                    b = al[rng.randint(0, len(al), L)]
                    if i % 3 == 0:
                        w = words[rng.randint(len(words))]
                        if len(w) <= L:
                            b[L - len(w):] = np.frombuffer(w, np.uint8)
                    strs.append(bytes(b))
                for i in rng.randint(0, n, 40):
                    strs[i] = b""
                base, off = pack(strs)
                lens = np.diff(off).astype(np.uint32)
                Lmax = max(16, int(lens.max()))
                rows = np.zeros((n, Lmax), np.uint8)
                for i, s in enumerate(strs):
                    rows[i, :len(s)] = np.frombuffer(s, np.uint8)
                want = o.table_walk(rows, lens)
                wbm = np.zeros((n + 63) // 64 * 64, bool)
                wbm[:n] = want != 0xFFFFFFFF
                cut = (lens // 3).astype(np.uint32)
                b1, o1 = pack([s[:c] for s, c in zip(strs, cut)])
                b2, o2 = pack([s[c:] for s, c in zip(strs, cut)])
                sets.append(dict(base=base, off=off, lens=lens, rows=rows, b1=b1, o1=o1, b2=b2, o2=o2,
                                 want=torch.from_numpy(want.view(np.int32)).cuda(), wbm=torch.from_numpy(np.packbits(wbm, bitorder="little").view(np.int64)).cuda()))
            cap = max(len(s["base"]) for s in sets) + 64
            Lm = max(s["rows"].shape[1] for s in sets)
            d_base = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            d_b2 = torch.zeros(cap, dtype=torch.uint8, device="cuda")
            d_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
            d_o2 = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
            d_off32 = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
            d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
            d_rows = torch.zeros((n, Lm), dtype=torch.uint8, device="cuda")
            d_end = torch.zeros(n, dtype=torch.int32, device="cuda")
            d_st = torch.zeros(n, dtype=torch.int32, device="cuda")
            d_bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")

            def resume_two():
                d_st.fill_(int(np.int32(np.uint32(hip.STATE_START))))
                dfa.exec_offsets_device_front("resume", d_base.data_ptr(), d_off.data_ptr(), n, d_st.data_ptr(), 0)
                dfa.exec_packed_resume_device(d_b2.data_ptr(), hip.META_OFF64, d_o2.data_ptr(), n, d_st.data_ptr(), d_end.data_ptr(), d_bm.data_ptr())

            fronts = {
                "off64": lambda: dfa.exec_batch_offsets_device(d_base.data_ptr(), d_off.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr()),
                "off32": lambda: dfa.exec_batch_offsets32_device(d_base.data_ptr(), d_off32.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr()),
                "lengths": lambda: dfa.exec_batch_lengths_device(d_base.data_ptr(), d_len.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr()),
                "stride+len": lambda: dfa.exec_batch_device(d_rows.data_ptr(), Lm, n, d_end.data_ptr(), d_bm.data_ptr(), d_len=d_len.data_ptr()),
                "resume2": resume_two,
            }
            for fname, call in fronts.items():
                nbad = 0
                for r in range(reps):
                    s = sets[r & 1]
                    if fname == "resume2":
                        d_base[:len(s["b1"])] = torch.from_numpy(s["b1"]).cuda()
                        d_off.copy_(torch.from_numpy(s["o1"].view(np.int64)))
                        d_b2[:len(s["b2"])] = torch.from_numpy(s["b2"]).cuda()
                        d_o2.copy_(torch.from_numpy(s["o2"].view(np.int64)))
                    elif fname == "stride+len":
                        d_rows.zero_()
                        d_rows[:, :s["rows"].shape[1]] = torch.from_numpy(s["rows"]).cuda()
                        d_len.copy_(torch.from_numpy(s["lens"].view(np.int32)))
                    else:
                        d_base[:len(s["base"])] = torch.from_numpy(s["base"]).cuda()
                        d_off.copy_(torch.from_numpy(s["off"].view(np.int64)))
                        d_off32.copy_(torch.from_numpy(s["off"].astype(np.uint32).view(np.int32)))
                        d_len.copy_(torch.from_numpy(s["lens"].view(np.int32)))
                    call()
                    launches += 1
                    ok = bool(torch.equal(d_end, s["want"])) and bool(torch.equal(d_bm, s["wbm"]))
                    nbad += 0 if ok else 1
                assert "walk_lazy_lines" in dfa.last_kernel_name(), dfa.last_kernel_name()
                print(f"lazy lines  alpha={len(alpha_b):2d} states={flat.nstates:6d} lines={mix:6s} front={fname:10s}: {reps} launches, {nbad} wrong", flush=True)
                bad += nbad
        dfa.close()

    # ---- the many-DFA front ----
    gs = [Golden(p) for p in all_golden_paths()]
    dfas = [hip.HipDfa(g.flat, hip.DEFER_UPLOAD) for g in gs]
    jobs = [g.strings() for g in gs]
    wants = [np.where(g.ret == 1, g.end, 0xFFFFFFFF).astype(np.uint32) for g in gs]
    rng = np.random.RandomState(9)
    nbad = 0
    for r in range(reps * 3):
        k = int(rng.randint(1, len(gs) + 1))
        pick = rng.permutation(len(gs))[:k]
        outs = hip.exec_multi([dfas[q] for q in pick], [jobs[q] for q in pick], want_bitmap=bool(r & 1))
        launches += 1
        for q, (end, bm) in zip(pick, outs):
            ok = np.array_equal(end, wants[q])
            if bm is not None:
                ok = ok and np.array_equal(np.unpackbits(bm.view(np.uint8), bitorder="little")[:len(end)].astype(bool), wants[q] != 0xFFFFFFFF)
            nbad += 0 if ok else 1
    print(f"multi (host front): {reps * 3} submissions of 1..{len(gs)} jobs in random order, {nbad} jobs wrong", flush=True)
    bad += nbad
    keep, djobs = [], []
    for g in gs:
        base, off = g.packed()
        n = len(off) - 1
        tb = torch.from_numpy(np.ascontiguousarray(base)).cuda() if len(base) else torch.zeros(1, dtype=torch.uint8, device="cuda")
        to = torch.from_numpy(off.astype(np.int64)).cuda()
        te = torch.zeros(n, dtype=torch.int32, device="cuda")
        tm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        keep.append((tb, to, te, tm))
        djobs.append((tb.data_ptr(), to.data_ptr(), n, te.data_ptr(), tm.data_ptr()))
    nbad = 0
    st = torch.cuda.Stream()
    for r in range(reps * 3):
        k = int(rng.randint(1, len(gs) + 1))
        pick = rng.permutation(len(gs))[:k]
        for q in pick:
            keep[q][2].fill_(7)
        torch.cuda.synchronize()
        hip.exec_multi_device([dfas[q] for q in pick], [djobs[q] for q in pick], stream=st.cuda_stream if r & 1 else 0)
        torch.cuda.synchronize()
        launches += 1
        for q in pick:
            nbad += 0 if np.array_equal(keep[q][2].cpu().numpy().view(np.uint32), wants[q]) else 1
    print(f"multi (device front): {reps * 3} submissions, {nbad} jobs wrong", flush=True)
    bad += nbad
    print(f"TOTAL: {launches} launches, {bad} wrong")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
