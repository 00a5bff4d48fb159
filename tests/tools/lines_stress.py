#!/usr/bin/env python3
"""tests/tools/lines_stress.py -- repeated launches of the variable-length fronts (u64 offsets, u32 offsets, lengths alone,
stride + lengths; the library's device-side kernel choice, walk_generic forced, walk_ragged forced) on the C2 (5-bit column
table: ring and row records in the table's holes, 12 wavefronts) and C3 tables.  Two line sets alternate in the same device
buffers, so a result left over from the previous launch is a wrong one; every launch's end states AND accept bitmap are
compared on the device with the oracle's.  REPS launches per (table, line mix, front, mode); exits non-zero on any mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import bench
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle
    hip.load_library()
    torch.cuda.set_device(0)
    reps = int(os.environ.get("REPS", 300))
    n, L = 20000 + 37, 1024
    bad_total = launches = 0
    for wl in ("c2", "c3"):
        flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz" if wl == "c2" else "c3.npz"))
        o = Oracle(flat)
        # LAYOUTS: comma list of table layouts to force (0 = the planner's choice); a layout the automaton cannot take is skipped
        for layout in [int(x) for x in os.environ.get("LAYOUTS", "0").split(",")]:
            try:
                dfa = hip.HipDfa(flat, layout)
            except OSError:
                continue
            print(f"== {wl} layout {layout}: {dfa.info()['layout_name']}", flush=True)
            for mix, (lo, hi) in [m for m in (("0-1024", (0, 1024)), ("0-200", (0, 200)), ("8-64", (8, 64)), ("8-16", (8, 16))) if m[0] in os.environ.get("MIXES", "0-1024,0-200,8-64").split(",")]:
                sets = []
                for v in range(2):
                    rows = torch.empty((n, L), dtype=torch.uint8, device="cuda")
                    bench.generate(hip, wl, rows.data_ptr(), n, L, v * 1000003)
                    rng = np.random.RandomState(5 + v)
                    lens = rng.randint(lo, hi + 1, n).astype(np.uint32)
                    lens[rng.randint(0, n, 50)] = 0
                    hrows = rows.cpu().numpy()
                    want = o.table_walk(hrows, lens)
                    off = np.zeros(n + 1, np.uint64)
                    off[1:] = np.cumsum(lens)
                    packed = np.concatenate([hrows[i, :lens[i]] for i in range(n)] + [np.zeros(16, np.uint8)])
                    wbm = np.zeros((n + 63) // 64 * 64, bool)
                    wbm[:n] = want != 0xFFFFFFFF
                    sets.append(dict(rows=rows, lens=lens, off=off, packed=packed, want=torch.from_numpy(want.view(np.int32)).cuda(),
                                     wbm=torch.from_numpy(np.packbits(wbm, bitorder="little").view(np.int64)).cuda()))
                cap = max(len(s["packed"]) for s in sets)
                d_packed = torch.zeros(cap, dtype=torch.uint8, device="cuda")
                d_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
                d_off32 = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
                d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
                d_rows = torch.zeros((n, L), dtype=torch.uint8, device="cuda")
                d_end = torch.zeros(n, dtype=torch.int32, device="cuda")
                d_bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
                fronts = {
                    "off64": lambda: dfa.exec_batch_offsets_device(d_packed.data_ptr(), d_off.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr()),
                    "off32": lambda: dfa.exec_batch_offsets32_device(d_packed.data_ptr(), d_off32.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr()),
                    "lengths": lambda: dfa.exec_batch_lengths_device(d_packed.data_ptr(), d_len.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr()),
                    "stride+len": lambda: dfa.exec_batch_device(d_rows.data_ptr(), L, n, d_end.data_ptr(), d_bm.data_ptr(), d_len=d_len.data_ptr()),
                }
                for mode in [int(x) for x in os.environ.get("MODES", "-1,2,3").split(",")]:
                    dfa.tune(hip.KNOB_INPUT_MODE, mode)
                    for fname, call in fronts.items():
                        bad = 0
                        for r in range(reps):
                            s = sets[r & 1]
                            if r < 2 or True:       # the other set's bytes and metadata into the same buffers
                                d_packed[:len(s["packed"])] = torch.from_numpy(s["packed"]).cuda() if r < 2 else s["_dp"]
                                if r < 2:
                                    s["_dp"] = torch.from_numpy(s["packed"]).cuda()
                                    s["_do"] = torch.from_numpy(s["off"].view(np.int64)).cuda()
                                    s["_dl"] = torch.from_numpy(s["lens"].view(np.int32)).cuda()
                                d_off.copy_(s["_do"])
                                d_off32.copy_(s["_do"].to(torch.int32))
                                d_len.copy_(s["_dl"])
                                if fname == "stride+len":
                                    d_rows.copy_(s["rows"])
                            d_end.fill_(7)
                            d_bm.fill_(-1)
                            call()
                            ok = bool(torch.equal(d_end, s["want"])) and bool(torch.equal(d_bm, s["wbm"]))
                            bad += 0 if ok else 1
                        launches += reps
                        bad_total += bad
                        print(f"{wl} lines {mix:7s} front={fname:10s} mode={mode:2d} launches={reps} wrong={bad} kernel={dfa.last_kernel_name()[-60:]}", flush=True)
            dfa.close()
    print(f"TOTAL launches={launches} wrong={bad_total}")
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
