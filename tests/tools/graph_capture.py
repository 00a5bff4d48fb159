#!/usr/bin/env python3
"""tests/tools/graph_capture.py -- are the device-pointer fronts capturable into a HIP graph?  Captures one launch of each front on
torch's capture stream, replays the graph on fresh inputs and compares with the oracle; prints the replay latency next to the
plain launch's for a small batch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle
    hip.load_library()
    torch.cuda.set_device(0)
    flat = hip.FlatDfa.load(os.path.join(ROOT, "tests", "golden", "c3.npz"))
    o = Oracle(flat)
    dfa = hip.HipDfa(flat)
    rng = np.random.RandomState(3)
    n, L = 4096, 256
    alpha = np.frombuffer(b"abcdwxyz0123456789", np.uint8)
    rows = alpha[rng.randint(0, len(alpha), (n, L))]
    lens = rng.randint(0, L + 1, n).astype(np.uint32)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    packed = np.concatenate([rows[i, :lens[i]] for i in range(n)] + [np.zeros(16, np.uint8)])
    d_rows = torch.from_numpy(rows).cuda()
    d_len = torch.from_numpy(lens.view(np.int32)).cuda()
    d_off = torch.from_numpy(off.view(np.int64)).cuda()
    d_packed = torch.from_numpy(packed).cuda()
    d_end = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_bm = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    fronts = {
        "stride": (lambda s: dfa.exec_batch_device(d_rows.data_ptr(), L, n, d_end.data_ptr(), d_bm.data_ptr(), stream=s), o.table_walk(rows)),
        "stride+len": (lambda s: dfa.exec_batch_device(d_rows.data_ptr(), L, n, d_end.data_ptr(), d_bm.data_ptr(), d_len=d_len.data_ptr(), stream=s), o.table_walk(rows, lens)),
        "offsets": (lambda s: dfa.exec_batch_offsets_device(d_packed.data_ptr(), d_off.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr(), stream=s), o.table_walk(rows, lens)),
        "lengths": (lambda s: dfa.exec_batch_lengths_device(d_packed.data_ptr(), d_len.data_ptr(), n, d_end.data_ptr(), d_bm.data_ptr(), stream=s), o.table_walk(rows, lens)),
    }
    for name, (call, want) in fronts.items():
        call(torch.cuda.current_stream().cuda_stream)      # warm: lazily built tables, scratch
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                call(torch.cuda.current_stream().cuda_stream)
        except Exception as e:      # noqa: BLE001
            print(f"{name}: capture FAILED: {e!r}"[:300])
            continue
        d_end.fill_(7)
        g.replay()
        torch.cuda.synchronize()
        ok = np.array_equal(d_end.cpu().numpy().view(np.uint32), want)
        t0 = time.perf_counter()
        for _ in range(200):
            g.replay()
        torch.cuda.synchronize()
        t_g = (time.perf_counter() - t0) / 200 * 1e6
        s = torch.cuda.current_stream().cuda_stream
        t0 = time.perf_counter()
        for _ in range(200):
            call(s)
        torch.cuda.synchronize()
        t_p = (time.perf_counter() - t0) / 200 * 1e6
        print(f"{name}: captured, replay {'ok' if ok else 'MISMATCH'}; {t_g:.1f} us per replay, {t_p:.1f} us per plain launch ({n} inputs)")


if __name__ == "__main__":
    main()
