#!/usr/bin/env python3
"""tests/tools/ab_dma.py -- A/B of the LDS-DMA kernel with one tile per wave (12 waves) against two tiles per wave
(walk_ldsdma2, 6-8 waves) on the full-size C3 / C2 batches, alternating the two in ONE process on ONE box
(run-to-run spread between boxes and between launches is ~2 %, as large as the effect)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import libfsm_amd as hip
    hip.load_library()
    n, L = int(os.environ.get("AB_N", 100_000_000)), 1024
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    for wl in ("c3", "c2"):
        flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz" if wl == "c2" else "c3.npz"))
        bench.generate(hip, wl, buf.data_ptr(), n, L, 0)
        torch.cuda.synchronize()
        dfa = hip.HipDfa(flat)
        ref = None
        res = {}
        for rnd in range(4):
            for bufs, waves in ((1, 0), (2, 0), (2, 6), (1, 8)):
                dfa.tune(hip.KNOB_DMA_BUFS, bufs)
                dfa.tune(hip.KNOB_WAVES, waves)
                ms = []
                for _ in range(4):
                    dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), 0)
                    ms.append(dfa.last_kernel_ms())
                if ref is None:
                    ref = end.clone()
                assert torch.equal(ref, end)
                res.setdefault((bufs, waves), []).extend(ms[1:])
        for k, v in res.items():
            v = np.array(v)
            print(f"{wl} tiles/wave={k[0]} waves={k[1] or 'default'}: median {np.median(v):.3f} ms min {v.min():.3f} max {v.max():.3f}  "
                  f"-> {n * (L + 4) / np.median(v) / 1e6:.0f} GB/s (median)", flush=True)
        dfa.close()


if __name__ == "__main__":
    main()
