#!/usr/bin/env python3
"""tests/tools/extras.py -- two side measurements for DESIGN.md:
 (1) per-lane load skip (KNOB_EARLY_RETIRE bit 1) on the C3 workload, with the result diffed;
 (2) the PCIe-inclusive rate of the host-pointer front fsm_hip_exec_batch()."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import libfsm_amd as hip
    hip.load_library()
    torch.cuda.set_device(0)
    n, L = 8_000_000, 1024
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    ref = torch.empty(n, dtype=torch.int32, device="cuda")
    for wl in ("c3", "c2"):
        flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz" if wl == "c2" else "c3.npz"))
        bench.generate(hip, wl, buf.data_ptr(), n, L, 0)
        torch.cuda.synchronize()
        dfa = hip.HipDfa(flat, hip.LAYOUT_COMBSELF if wl == "c3" else hip.LAYOUT_LDS)
        dfa.tune(hip.KNOB_INPUT_MODE, hip.IN_DIRECT)
        dfa.tune(hip.KNOB_PREFETCH, 0)
        first = True
        for early in (0, 1, 0, 1, 3, 2):
            dfa.tune(hip.KNOB_EARLY_RETIRE, early)
            ms = []
            for r in range(4):
                dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), 0)
                t = dfa.last_kernel_ms()
                if r:
                    ms.append(t)
            torch.cuda.synchronize()
            if first:
                ref.copy_(end)
                first = False
            print(f"{wl} {dfa.info()['layout_name']} direct_np early={early} (bit0 wave retire, bit1 lane load skip) "
                  f"ms={min(ms):.3f} GB/s={n * L / min(ms) / 1e6:.1f} {'ok' if torch.equal(end, ref) else 'DIFF'}", flush=True)
        dfa.close()
    # host-pointer front: PCIe-inclusive
    flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c1.npz"))
    dfa = hip.HipDfa(flat)
    rows = hip.gen_inputs_host(1_000_000, 1024, 0, 1, None, b"Libfsm", 8)
    for _ in range(2):
        t0 = time.perf_counter()
        e, _bm = dfa.exec_batch(rows)
        dt = time.perf_counter() - t0
    print(f"host-pointer front fsm_hip_exec_batch: {rows.size / 1e9:.2f} GB in {dt * 1e3:.1f} ms = {rows.size / dt / 1e9:.1f} GB/s "
          f"(hipMalloc + H2D + kernel + D2H + hipFree; kernel alone {dfa.last_kernel_ms():.3f} ms)")


if __name__ == "__main__":
    main()
