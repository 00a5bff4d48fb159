#!/usr/bin/env python3
"""tests/tools/sweep.py -- time many kernel variants in ONE process (GPU minutes are scarce).
(Lives under tests/ because it uses the oracle as its checker.)

Every variant's end states are compared with the first variant's (and the first
is sample-checked against the oracle), so a fast-but-wrong variant is flagged.
Usage: python tests/tools/sweep.py [--n 8000000] [--workloads c2,c3] [--quick]
"""
import argparse
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def c5_dfa(nwords=100000, seed=5):
    """BASELINE configs[4]: Aho-Corasick DFA over 1e5 random 4-8 letter words, built by the real
    reference (re_strings, src/libre/ac.c) -- its recursion needs a big stack, hence the thread."""
    import threading
    from oracle.pyoracle import RefFsm
    rng = np.random.RandomState(seed)
    alpha = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz", np.uint8)
    words = sorted(set(bytes(alpha[rng.randint(0, 26, rng.randint(4, 9))]) for _ in range(nwords)))
    out = {}

    def work():
        f = RefFsm.re_strings(words, 0, True)
        out["flat"] = f.flatten()
        out["f"] = f

    threading.stack_size(1 << 30)
    th = threading.Thread(target=work)
    th.start()
    th.join()
    threading.stack_size(0)
    return out["flat"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8_000_000)
    ap.add_argument("--len", type=int, default=1024)
    ap.add_argument("--workloads", default="c2,c3")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--set", default="full")
    ap.add_argument("--layouts", default="all")
    a = ap.parse_args()
    import torch
    import bench
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle
    hip.load_library()
    torch.cuda.set_device(0)
    n, L = a.n, a.len
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    ref_end = torch.empty(n, dtype=torch.int32, device="cuda")
    bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    print(f"# n={n} len={L} bytes={n * L / 1e9:.2f} GB  device={torch.cuda.get_device_name(0)}", flush=True)
    for wl in a.workloads.split(","):
        if wl == "c5":
            flat = c5_dfa()
            hip.gen_inputs_device(buf.data_ptr(), n, L, 0, 0x5EEDF5A1, b"abcdefghijklmnopqrstuvwxyz")
        else:
            flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", {"c2": "c1.npz", "c3t": "c3t.npz"}.get(wl, "c3.npz")))
            bench.generate(hip, wl, buf.data_ptr(), n, L, 0)
        torch.cuda.synchronize()
        first = True
        layouts = hip.ALL_LAYOUTS if a.layouts == "all" else [int(x) for x in a.layouts.split(",")]
        for layout in layouts:
            try:
                dfa = hip.HipDfa(flat, layout)
            except OSError:
                continue
            info = dfa.info()
            print(f"## {wl} layout={info['layout_name']} states={info['nstates']} classes={info['nclasses']} "
                  f"table_bytes={info['table_bytes']} lds={info['lds_bytes']}", flush=True)
            tiny = info["layout_name"] == "tiny"
            # (mode, nb, rows, waves, bpc, mask, early)
            variants = [(-1, 0, 0, 0, 0, -1, 1)]  # library defaults
            if a.set == "full":
                masks = (0,)
                variants += [(hip.IN_DIRECT, nb, r, 16, 0, m, 1) for nb in (4, 8) for r in (1, 2) for m in masks if not (nb == 8 and r == 2)]
                variants += [(hip.IN_DIRECT, 2, 2, 16, 0, masks[-1], 1), (hip.IN_DIRECT, 8, 1, 8, 0, masks[-1], 1)]
                variants += [(hip.IN_LDSDMA, seg, 1, w, 0, 0, 1) for w in (2, 4, 8, 16) for seg in (64, 128)]
                variants += [(hip.IN_LDSDMA, 128, 1, 4, b, 0, 1) for b in (1, 2, 3, 4)]
                variants += [(hip.IN_LDSDMA, 128, 1, 8, 0, 0, 0), (hip.IN_DIRECT, 8, 1, 16, 0, 0, 0)]
                variants += [(hip.IN_DIRECT, nb, 1, w, b, 2, 1) for nb in (4, 8) for w in (16, 8) for b in (0, 4)]  # mask=2: no-prefetch kernel
                variants += [(hip.IN_GENERIC, 0, 1, 16, 0, masks[-1], 1)]
            if a.set == "r2":  # round 2: every input path of every layout, library defaults first
                variants = [(-1, 0, 0, 0, 0, -1, 1), (hip.IN_LDSDMA, 128, 1, 16, 0, 4, 1), (hip.IN_LDSDMA, 128, 1, 12, 0, 4, 1),
                            (hip.IN_LDSDMA, 128, 1, 8, 0, 4, 1), (hip.IN_DIRECT, 8, 1, 16, 0, 0, 1), (hip.IN_DIRECT, 4, 1, 16, 0, 2, 1),
                            (hip.IN_RAGGED, 0, 1, 0, 0, 0, 1), (hip.IN_GENERIC, 0, 1, 16, 0, 0, 1)]
            if a.set == "rows":  # one vs two inputs per lane (needs walk_direct<Pol,4,2> instantiated: profiles/r03b_*; rows=2 is ignored otherwise)
                variants = [(-1, 0, 0, 0, 0, -1, 1), (hip.IN_DIRECT, 4, 1, 16, 0, 0, 1), (hip.IN_DIRECT, 4, 2, 16, 0, 0, 1), (hip.IN_DIRECT, 4, 2, 12, 0, 0, 1),
                            (hip.IN_DIRECT, 4, 2, 8, 0, 0, 1), (hip.IN_DIRECT, 8, 1, 16, 0, 0, 1), (hip.IN_LDSDMA, 128, 1, 8, 0, 4, 1)]
            if a.set == "c3t":  # round 3: which input path feeds a lookup chain that sits next to a 90 KB table
                variants = [(-1, 0, 0, 0, 0, -1, 1), (hip.IN_LDSDMA, 64, 1, 16, 0, 0, 1), (hip.IN_LDSDMA, 64, 1, 16, 0, 4, 1), (hip.IN_LDSDMA, 64, 1, 14, 0, 0, 1),
                            (hip.IN_LDSDMA, 64, 1, 12, 0, 0, 1), (hip.IN_LDSDMA, 128, 1, 8, 0, 4, 1), (hip.IN_DIRECT, 8, 1, 16, 0, 0, 1), (hip.IN_DIRECT, 8, 1, 16, 0, 0, 9),
                            (hip.IN_DIRECT, 4, 1, 16, 0, 2, 1), (hip.IN_DIRECT, 8, 1, 14, 0, 0, 1), (hip.IN_DIRECT, 8, 1, 12, 0, 0, 1), (hip.IN_DIRECT, 8, 1, 16, 0, 0, 1)]
            if a.set == "dma":  # LDS-DMA staging next to an LDS table: how many waves fit / pay
                variants = [(hip.IN_LDSDMA, 128, 1, w, b, m, 1) for w in (16, 14, 12, 10, 8) for b in (0, 1) for m in (0, 4)]
                variants += [(hip.IN_LDSDMA, 64, 1, 16, 0, 0, 1), (hip.IN_DIRECT, 8, 1, 16, 0, 2, 1), (hip.IN_DIRECT, 8, 1, 16, 0, 0, 1)]
            if a.set == "nt":  # nontemporal LDS-DMA input loads (mask bit 2)
                variants = [(hip.IN_LDSDMA, 128, 1, w, 0, m, 1) for w in (4, 8, 16) for m in (0, 4)]
                variants += [(hip.IN_DIRECT, 8, 1, 16, 0, 0, 1)]
            if a.set == "hot":  # global layout: how much of the table head to mirror in LDS
                variants = [(hip.IN_DIRECT, 8, 1, w, 0, pf, 1) for w in (16, 8) for pf in (0, 2)]
                variants += [(hip.IN_LDSDMA, 64, 1, 16, 0, 0, 1), (hip.IN_LDSDMA, 128, 1, 8, 0, 0, 1)]
            hots = (0, 16384, 40960, 98304) if (a.set == "hot" and info["layout_name"] == "global") else (None,)
            for hot in hots:
              if hot is not None:
                dfa.tune(hip.KNOB_HOT_BYTES, hot)
                print(f"### hot_bytes={hot}", flush=True)
              variants_ = variants
              for mode, nb, rows_, waves, bpc, mask, early in variants_:
                if True:
                    dfa.tune(hip.KNOB_INPUT_MODE, mode)
                    dfa.tune(hip.KNOB_NB, nb if mode != hip.IN_LDSDMA else 0)
                    dfa.tune(hip.KNOB_SEG, nb if mode == hip.IN_LDSDMA else 0)
                    dfa.tune(hip.KNOB_ROWS, rows_)
                    dfa.tune(hip.KNOB_WAVES, waves)
                    dfa.tune(hip.KNOB_BLOCKS_PER_CU, bpc)
                    dfa.tune(hip.KNOB_MASK, mask & 1)
                    dfa.tune(hip.KNOB_PREFETCH, 0 if mask & 2 else 1)
                    dfa.tune(hip.KNOB_NT, 1 if mask & 4 else 0)
                    dfa.tune(hip.KNOB_EARLY_RETIRE, early)
                    nt = rows_
                    ms = []
                    try:
                        for r in range(a.reps + 1):
                            dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), bm.data_ptr())
                            t = dfa.last_kernel_ms()
                            if r:
                                ms.append(t)
                    except OSError as e:
                        print(f"{wl} {info['layout_name']:6s} mode={mode} nb={nb} rows={rows_} waves={waves} bpc={bpc} mask={mask} early={early} ERROR {e}", flush=True)
                        continue
                    torch.cuda.synchronize()
                    if first:
                        ref_end.copy_(end)
                        idx = np.random.RandomState(0).randint(0, n, 2048)
                        rows = buf[torch.from_numpy(idx).cuda()].cpu().numpy()
                        ok = np.array_equal(Oracle(flat).table_walk(rows), end.cpu().numpy().view(np.uint32)[idx])
                        print(f"# {wl}: first variant vs oracle on 2048 sampled rows: {'OK' if ok else 'MISMATCH'}; accepts={(end != -1).sum().item()}", flush=True)
                        first = False
                        same = True
                    else:
                        same = bool(torch.equal(end, ref_end))
                    best = min(ms)
                    print(f"{wl} {info['layout_name']:6s} mode={mode:2d} nb={nb} rows={rows_} waves={waves:2d} bpc={bpc} mask={mask:2d} early={early} "
                          f"ms={best:8.3f} GB/s={n * L / best / 1e6:8.1f} frac_hbm={n * (L + 4) / best / 1e6 / 8000:.3f} {'ok' if same else 'DIFF'}", flush=True)
            dfa.close()


if __name__ == "__main__":
    main()
