#!/usr/bin/env python3
"""tests/tools/tiny_probe.py -- walk rate of the tiny layout by state count (5 / 7 / 8 / 12 / 15 states + DEAD):
Tiny5Pol (<= 6), TinyPol<u32> (<= 8), TinyPol<u64> (<= 16).  Random complete DFAs, 8e6 x 1 KiB random inputs,
checked against the oracle on a sample."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle
    hip.load_library()
    torch.cuda.set_device(0)
    n, L = int(os.environ.get("TINY_N", "8000000")), 1024
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    hip.gen_inputs_device(buf.data_ptr(), n, L, 0, 11, b"abcdefgh")
    torch.cuda.synchronize()
    rng = np.random.RandomState(4)
    for S in (5, 7, 8, 12, 15):
        nt = np.full((S, 256), -1, np.int64)
        for c in b"abcdefgh":
            nt[:, c] = rng.randint(0, S, S)       # complete on the alphabet: no lane ever dies
        flat = hip.FlatDfa.from_dense(nt, 0, (rng.rand(S) < 0.5).astype(int).tolist())
        dfa = hip.HipDfa(flat)
        if os.environ.get("TINY_WAVES"):
            dfa.tune(hip.KNOB_WAVES, int(os.environ["TINY_WAVES"]))
        ms = []
        for _ in range(5):
            dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), 0)
            ms.append(dfa.last_kernel_ms())
        torch.cuda.synchronize()
        t = min(ms[1:])
        k = 512
        ok = np.array_equal(Oracle(flat).table_walk(buf[:k].cpu().numpy()), end[:k].cpu().numpy().view(np.uint32))
        i = dfa.info()
        print(f"states={S:2d}(+dead) layout={i['layout_name']} lds={i['lds_bytes']:6d} waves={i['waves_per_block']:2d} "
              f"ms={t:7.3f} GB/s={n * L / t / 1e6:7.1f} {'ok' if ok else 'MISMATCH'}", flush=True)
        dfa.close()


if __name__ == "__main__":
    main()
