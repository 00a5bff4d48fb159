#!/usr/bin/env python3
"""tests/tools/multi_bulk_probe.py -- the many-DFA front at throughput size, outside bench.py (run on the GPU box):
K automata (the retest goldens, cycled) x NL lines each, device pointers, ONE fsm_hip_exec_multi_device; line lengths fixed
or uniform in a range.  Prints ms per call and GB/s of line bytes walked, and checks every 64th job against its dfa's own walk."""
import argparse
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--nl", type=int, default=100_000)
    ap.add_argument("--mixes", default="64-64,8-64,16-16,200-200")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    import libfsm_amd as hip
    from common import Golden
    gs = [Golden(p) for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "retest", "*.npz")))]
    K, nl = a.k, a.nl
    stream = torch.cuda.current_stream().cuda_stream
    for mix in a.mixes.split(","):
        lo, hi = (int(x) for x in mix.split("-"))
        rng = np.random.RandomState(5)
        lens = rng.randint(lo, hi + 1, nl).astype(np.int64)
        off_h = np.zeros(nl + 1, np.int64)
        off_h[1:] = np.cumsum(lens)
        per = int(off_h[-1])
        text = torch.empty(K * per + 16, dtype=torch.uint8, device="cuda")
        hip.gen_inputs_device(text.data_ptr(), (K * per) // 64, 64, 0, 77, bytes(range(32, 127)))
        off = torch.from_numpy(off_h).cuda()
        ends = torch.empty(K * nl, dtype=torch.int32, device="cuda")
        ds = [hip.HipDfa(gs[q % len(gs)].flat, hip.DEFER_UPLOAD) for q in range(K)]
        jobs = [(text.data_ptr() + q * per, off.data_ptr(), nl, ends.data_ptr() + q * nl * 4, 0) for q in range(K)]
        for _ in range(2):
            hip.exec_multi_device(ds, jobs, stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            hip.exec_multi_device(ds, jobs, stream=stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        pr = hip.MultiPrepared(ds, [j + (0,) for j in jobs], 1)
        for _ in range(2):
            pr.launch(stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            pr.launch(stream)
        torch.cuda.synchronize()
        ms_p = (time.perf_counter() - t0) / a.reps * 1e3
        pr.close()
        ok = True
        chk = torch.empty(nl, dtype=torch.int32, device="cuda")
        for q in range(0, K, 64):
            one = hip.HipDfa(gs[q % len(gs)].flat)
            one.exec_batch_offsets_device(jobs[q][0], off.data_ptr(), nl, chk.data_ptr(), 0, stream=stream)
            torch.cuda.synchronize()
            ok = ok and bool(torch.equal(chk, ends[q * nl:(q + 1) * nl]))
            one.close()
        for d in ds:
            d.close()
        print(f"lines {mix:>9}: {K} automata x {nl} lines, launches {hip.multi_last_launches()}: {ms:8.3f} ms  {K * per / ms / 1e6:8.1f} GB/s of line bytes  "
              f"{K * (per + nl * 12) / ms / 1e6:8.1f} with 8 B offset + 4 B end state per line  {'ok' if ok else 'MISMATCH'}   prepared: {ms_p:8.3f} ms {K * per / ms_p / 1e6:8.1f} / {K * (per + nl * 12) / ms_p / 1e6:8.1f}", flush=True)
        del text, ends


if __name__ == "__main__":
    main()
