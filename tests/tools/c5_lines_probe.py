#!/usr/bin/env python3
"""tests/tools/c5_lines_probe.py -- the lazy walk on packed lines of chosen length mixes (BASELINE configs[4] automaton): where the
variable-length kernel's time goes.  FIXED lengths (every line the same: all slots end together, no dead slots, a refill every
len / 64 turns) against uniform mixes; the fixed-stride kernel on the same bytes as the yardstick.  Every mix's end states are
checked against the oracle on a sample.  (Lives under tests/ because it uses the oracle as its checker.)"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.tools.c5_probe import ALPHA64, make_words  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bytes", type=int, default=4_000_000_000)
    ap.add_argument("--mixes", default="1024-1024,512-512,64-64,32-32,16-16,0-1024,8-64,100-100,37-37,8-16")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--knobs", default="", help="knob=value,...")
    ap.add_argument("--plant", type=int, default=0, help="every PLANTth line ends with one of the literals (0: none)")
    a = ap.parse_args()
    import torch
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle
    hip.load_library()
    torch.cuda.set_device(0)
    words = make_words(100000, 64, 8, 16)
    flat = hip.FlatDfa.from_strings(words, 2, list(range(len(words))))
    dfa = hip.HipDfa(flat, 7)
    for kv in [x for x in a.knobs.split(",") if x]:
        dfa.tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
    orc = Oracle(flat)
    L = 1024
    nrows = a.bytes // L
    buf = torch.empty((nrows, L), dtype=torch.uint8, device="cuda")
    hip.gen_inputs_device(buf.data_ptr(), nrows, L, 0, 0x5EEDF5A1, ALPHA64)
    end = torch.empty(nrows, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    ms = []
    for _ in range(a.reps + 1):
        dfa.exec_batch_device(buf.data_ptr(), L, nrows, end.data_ptr(), 0, stream=stream)
        ms.append(dfa.last_kernel_ms())
    torch.cuda.synchronize()
    print(f"fixed stride 1024: {min(ms[1:]):8.3f} ms {nrows * L / min(ms[1:]) / 1e6:8.1f} GB/s  {dfa.last_kernel_name()}", flush=True)
    flatbuf = buf.view(-1)
    rng = np.random.RandomState(7)
    for mix in a.mixes.split(","):
        lo, hi = (int(x) for x in mix.split("-"))
        mean = (lo + hi) / 2 or 1
        n = int(min(a.bytes * 0.98 / mean, 120_000_000))
        lens = rng.randint(lo, hi + 1, n).astype(np.int64)
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum(lens)
        total = int(off[-1])
        assert total <= flatbuf.numel()
        d_off = torch.from_numpy(off).cuda()
        if a.plant:
            # (the buffer is re-generated per mix: earlier plants would otherwise pile up)
            hip.gen_inputs_device(buf.data_ptr(), nrows, L, 0, 0x5EEDF5A1, ALPHA64)
            lw = np.array([len(w_) for w_ in words], np.int64)
            W = np.zeros((len(words), int(lw.max())), np.uint8)
            for q_, w_ in enumerate(words):
                W[q_, :len(w_)] = np.frombuffer(w_, np.uint8)
            Wd, ld = torch.from_numpy(W).cuda(), torch.from_numpy(lw).cuda()
            idx = torch.arange(0, n, a.plant, device="cuda")
            widx = (idx * 2654435761) % len(words)
            dl = torch.from_numpy(lens).cuda()
            for l_ in sorted(set(lw.tolist())):
                m_ = (ld[widx] == l_) & (dl[idx] >= l_)
                if bool(m_.any()):
                    pos = (d_off[idx[m_] + 1] - l_).unsqueeze(1) + torch.arange(l_, device="cuda").unsqueeze(0)
                    flatbuf[pos.reshape(-1)] = Wd[widx[m_], :l_].reshape(-1)
            del Wd, ld, idx, widx, dl
        e = torch.empty(n, dtype=torch.int32, device="cuda")
        ms = []
        for _ in range(a.reps + 1):
            dfa.exec_batch_offsets_device(flatbuf.data_ptr(), d_off.data_ptr(), n, e.data_ptr(), 0, stream=stream)
            ms.append(dfa.last_kernel_ms())
        torch.cuda.synchronize()
        k = min(n, 4096)
        idx = np.unique(np.concatenate([np.arange(k // 2), n - 1 - np.arange(k // 2)]))
        host = flatbuf[: int(off[k // 2 + 1]) + 1].cpu().numpy()
        tail0 = int(off[n - k // 2 - 1])
        hostt = flatbuf[tail0:total].cpu().numpy()
        got = e.cpu().numpy().view(np.uint32)
        ok = True
        rows = np.zeros((len(idx), max(hi, 1)), np.uint8)
        for q, i in enumerate(idx):
            if i <= k // 2:
                rows[q, :lens[i]] = host[off[i]:off[i + 1]]
            else:
                rows[q, :lens[i]] = hostt[off[i] - tail0:off[i + 1] - tail0]
        want = orc.table_walk(rows, lens[idx].astype(np.uint32))
        ok = np.array_equal(want, got[idx])
        t = min(ms[1:])
        print(f"lines {mix:>10s}: n={n:10d} {t:8.3f} ms {total / t / 1e6:8.1f} GB/s of line bytes  {'ok' if ok else 'MISMATCH'}  {dfa.last_kernel_name()[:60]}", flush=True)


if __name__ == "__main__":
    main()
