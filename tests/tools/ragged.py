#!/usr/bin/env python3
"""tests/tools/ragged.py -- throughput of the ragged / packed front on device-resident inputs: fixed stride
with per-input lengths, and inputs packed back to back with an offsets array; the coalesced lane-refilling
kernel (walk_ragged, mode 3) next to per-lane loads (walk_generic, mode 2), at several workgroup sizes.
Results are checked against the oracle on a sample and kernel against kernel on every input."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle
    hip.load_library()
    torch.cuda.set_device(0)
    n, L = int(os.environ.get("RAGGED_N", 2_000_000)), 1024      # the profiles/ tables use RAGGED_N=6000000
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    only = os.environ.get("RAGGED_CASES")      # e.g. "c2:packed:3,c2:rows:1": workload:front:mode filters (profiling runs)
    for dist in os.environ.get("RAGGED_DISTS", "uniform0-1024,short8-64,short8-16,short32-128,mid64-256").split(","):
        if dist.startswith("uniform"):
            lens = torch.randint(0, L + 1, (n,), device="cuda", dtype=torch.int32, generator=g)
        else:
            lo, hi = [int(x) for x in dist.lstrip("shortmid").split("-")]
            lens = torch.randint(lo, hi + 1, (n,), device="cuda", dtype=torch.int32, generator=g)
        total = int(lens.sum().item())
        off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        off[1:] = torch.cumsum(lens.to(torch.int64), 0)
        for wl in ("c2", "c3"):
            flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz" if wl == "c2" else "c3.npz"))
            bench.generate(hip, wl, buf.data_ptr(), n, L, 0)
            torch.cuda.synchronize()
            ar = torch.arange(L, device="cuda", dtype=torch.int32)[None, :]
            packed = torch.empty(total, dtype=torch.uint8, device="cuda")      # exactly the inputs' bytes: nothing after the last one
            for r0 in range(0, n, 1 << 20):                                     # masked selects of <= 1 GiB at a time
                r1 = min(n, r0 + (1 << 20))
                packed[int(off[r0]):int(off[r1])] = buf[r0:r1][ar < lens[r0:r1, None]]
            dfa = hip.HipDfa(flat, int(os.environ.get("RAGGED_LAYOUT", "0")))
            idx = np.random.RandomState(0).randint(0, n, 1024)
            rows = buf[torch.from_numpy(idx).cuda()].cpu().numpy()
            want = Oracle(flat).table_walk(rows, lens.cpu().numpy().astype(np.uint32)[idx])
            ref = None
            off32 = off.to(torch.int32) if total < (1 << 32) else None
            bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
            for front, mode, waves, align in (("packed", -1, 0, 0), ("packed", 3, 0, 0), ("packed", 2, 0, 0), ("off32", -1, 0, 0), ("lengths", -1, 0, 0), ("len-bitmap", -1, 0, 0),
                                              ("stride+len", -1, 0, 0), ("stride+len", 3, 0, 0), ("stride+len", 2, 0, 0), ("rows", 3, 0, 0), ("rows", -1, 0, 0)):
                if only is not None and f"{wl}:{front}:{mode}" not in only.split(","):
                    continue
                if front == "rows" and dist != "uniform0-1024":
                    continue
                dfa.tune(hip.KNOB_INPUT_MODE, mode)
                dfa.tune(hip.KNOB_WAVES, waves or int(os.environ.get("PK_WAVES", 0)))
                for kn, ev in ((hip.KNOB_PICK_MEAN, "PICK_MEAN"), (hip.KNOB_NOSKIP, "RAGGED_NOSKIP")):
                    if os.environ.get(ev):
                        dfa.tune(kn, int(os.environ[ev]))
                # RAGGED_EARLY: one value of the early knob or a comma list of them, timed one after the other on the same batch
                # (33 = walk_generic's own body instead of walk_lines32, 65 = walk_lines32 with the first-chunk skip tests left in)
                for early in [int(x) for x in os.environ.get("RAGGED_EARLY", "-1").split(",")]:
                    dfa.tune(hip.KNOB_EARLY_RETIRE, early)
                    ms = []
                    for r in range(4):
                        if front == "packed":
                            dfa.exec_batch_offsets_device(packed.data_ptr(), off.data_ptr(), n, end.data_ptr(), 0)
                        elif front == "off32":
                            if off32 is None:
                                break
                            dfa.exec_batch_offsets32_device(packed.data_ptr(), off32.data_ptr(), n, end.data_ptr(), 0)
                        elif front == "lengths":
                            dfa.exec_batch_lengths_device(packed.data_ptr(), lens.data_ptr(), n, end.data_ptr(), 0)
                        elif front == "len-bitmap":       # the 1-bit-per-input answer alone (the end states of the previous front stay in `end`)
                            dfa.exec_batch_lengths_device(packed.data_ptr(), lens.data_ptr(), n, 0, bm.data_ptr())
                        elif front == "rows":       # whole 1024-byte rows: what the same kernel (3) / the LDS-DMA kernel (-1) does without raggedness
                            dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), 0)
                        else:
                            dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), 0, d_len=lens.data_ptr())
                        t = dfa.last_kernel_ms()
                        if r:
                            ms.append(t)
                    torch.cuda.synchronize()
                    if not ms:
                        continue
                    if front == "rows":
                        print(f"{wl} {dfa.info()['layout_name']:8s} lens=1024           front={front:10s} mode={mode:2d} waves={waves:2d} ms={min(ms):8.3f} "
                              f"GB/s(walked)={n * L / min(ms) / 1e6:8.1f}", flush=True)
                        continue
                    ok = np.array_equal(end.cpu().numpy().view(np.uint32)[idx], want)
                    if ref is None:
                        ref = end.clone()
                    ok = ok and bool(torch.equal(ref, end))
                    if front == "len-bitmap":
                        got = np.unpackbits(bm.cpu().numpy().view(np.uint8), bitorder="little")[:n].astype(bool)
                        ok = ok and np.array_equal(got, ref.cpu().numpy() != -1)
                    print(f"{wl} {dfa.info()['layout_name']:8s} lens={dist:14s} front={front:10s} mode={mode:2d} waves={waves:2d} early={early:3d} ms={min(ms):8.3f} "
                          f"GB/s(walked)={total / min(ms) / 1e6:8.1f} {'ok' if ok else 'MISMATCH'}  {dfa.last_kernel_name()[:60]}", flush=True)
            dfa.close()


if __name__ == "__main__":
    main()
