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
    n, L = int(os.environ.get("RAGGED_N", 2_000_000)), 1024
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(1)
    for dist in ("uniform0-1024", "short8-64"):
        if dist.startswith("uniform"):
            lens = torch.randint(0, L + 1, (n,), device="cuda", dtype=torch.int32, generator=g)
        else:
            lens = torch.randint(8, 65, (n,), device="cuda", dtype=torch.int32, generator=g)
        total = int(lens.sum().item())
        off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
        off[1:] = torch.cumsum(lens.to(torch.int64), 0)
        for wl in ("c2", "c3"):
            flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz" if wl == "c2" else "c3.npz"))
            bench.generate(hip, wl, buf.data_ptr(), n, L, 0)
            torch.cuda.synchronize()
            mask = torch.arange(L, device="cuda", dtype=torch.int32)[None, :] < lens[:, None]
            packed = torch.cat([buf[mask], torch.zeros(64, dtype=torch.uint8, device="cuda")])
            dfa = hip.HipDfa(flat)
            idx = np.random.RandomState(0).randint(0, n, 1024)
            rows = buf[torch.from_numpy(idx).cuda()].cpu().numpy()
            want = Oracle(flat).table_walk(rows, lens.cpu().numpy().astype(np.uint32)[idx])
            ref = None
            for front, mode, waves, align in (("packed", 3, 0, 0), ("packed", 3, 0, 1), ("packed", 3, 6, 0), ("packed", 2, 0, 0),
                                              ("stride+len", 3, 0, 0), ("stride+len", 3, 0, 1), ("stride+len", 2, 0, 0)):
                dfa.tune(hip.KNOB_INPUT_MODE, mode)
                dfa.tune(hip.KNOB_WAVES, waves)
                dfa.tune(hip.KNOB_RAGGED_ALIGN, align)
                ms = []
                for r in range(4):
                    if front == "packed":
                        dfa.exec_batch_offsets_device(packed.data_ptr(), off.data_ptr(), n, end.data_ptr(), 0)
                    else:
                        dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), 0, d_len=lens.data_ptr())
                    t = dfa.last_kernel_ms()
                    if r:
                        ms.append(t)
                torch.cuda.synchronize()
                ok = np.array_equal(end.cpu().numpy().view(np.uint32)[idx], want)
                if ref is None:
                    ref = end.clone()
                ok = ok and bool(torch.equal(ref, end))
                print(f"{wl} {dfa.info()['layout_name']:8s} lens={dist:14s} front={front:10s} mode={mode} waves={waves:2d} align128={align} ms={min(ms):8.3f} "
                      f"GB/s(walked)={total / min(ms) / 1e6:8.1f} {'ok' if ok else 'MISMATCH'}", flush=True)
            dfa.close()


if __name__ == "__main__":
    main()
