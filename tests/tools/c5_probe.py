#!/usr/bin/env python3
"""tests/tools/c5_probe.py -- BASELINE configs[4] regime: a literal-set (Aho-Corasick) DFA whose table
does not fit LDS.  Times the global layout while the LDS-resident hot prefix (KNOB_HOT_BYTES) and the
occupancy vary; every variant's end states are compared with the first one's and a sample with the oracle.
(Lives under tests/ because it uses the oracle as its checker.)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

ALPHA64 = b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789_-"


def make_words(nwords, nalpha, lo, hi, seed=5):
    rng = np.random.RandomState(seed)
    alpha = np.frombuffer(ALPHA64[:nalpha], np.uint8)
    return [bytes(alpha[rng.randint(0, nalpha, rng.randint(lo, hi + 1))]) for _ in range(nwords)]


def plant_tails(torch, buf, words, every=8):
    """every `every`-th input ends with one of the words (so it is accepted under ANCHOR_RIGHT)."""
    n, L = buf.shape
    lens = np.array([len(w) for w in words], np.int64)
    W = np.zeros((len(words), int(lens.max())), np.uint8)
    for i, w in enumerate(words):
        W[i, :len(w)] = np.frombuffer(w, np.uint8)
    Wd, ld = torch.from_numpy(W).to(buf.device), torch.from_numpy(lens).to(buf.device)
    rows = torch.arange(0, n, every, device=buf.device)
    widx = (rows * 2654435761) % len(words)
    for l in sorted(set(lens.tolist())):
        m = ld[widx] == l
        if bool(m.any()):
            buf[rows[m], L - l:] = Wd[widx[m], :l]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nwords", type=int, default=100000)
    ap.add_argument("--alpha", type=int, default=64)
    ap.add_argument("--minlen", type=int, default=8)
    ap.add_argument("--maxlen", type=int, default=16)
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--len", type=int, default=1024)
    ap.add_argument("--hot", default="0,40960,81920,122880")
    ap.add_argument("--waves", default="0")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--layout", type=int, default=4, help="4 global, 7 sparse, 0 auto")
    ap.add_argument("--early", type=int, default=-1)
    ap.add_argument("--variants", default="", help="semicolon list of knob=value,... sets, e.g. 10=0,2=4;10=1,2=2")
    a = ap.parse_args()
    import torch
    import libfsm_amd as hip
    from oracle.pyoracle import Oracle
    hip.load_library()
    torch.cuda.set_device(0)
    words = make_words(a.nwords, a.alpha, a.minlen, a.maxlen)
    t0 = time.time()
    flat = hip.FlatDfa.from_strings(words, 2, list(range(len(words))))
    t1 = time.time()
    dfa = hip.HipDfa(flat, a.layout)
    t2 = time.time()
    info = dfa.info()
    if a.early >= 0:
        dfa.tune(hip.KNOB_EARLY_RETIRE, a.early)
    n, L = a.n, a.len
    print(f"# words={len(words)} alpha={a.alpha} len={a.minlen}-{a.maxlen} states={flat.nstates} classes={info['nclasses']} "
          f"layout={info['layout_name']} table_bytes={info['table_bytes']} build={t1 - t0:.1f}s plan+upload={t2 - t1:.1f}s n={n} L={L}", flush=True)
    buf = torch.empty((n, L), dtype=torch.uint8, device="cuda")
    hip.gen_inputs_device(buf.data_ptr(), n, L, 0, 0x5EEDF5A1, ALPHA64[:a.alpha])
    plant_tails(torch, buf, words)
    torch.cuda.synchronize()
    end = torch.empty(n, dtype=torch.int32, device="cuda")
    ref = None
    stream = torch.cuda.current_stream().cuda_stream
    variants = a.variants.split(";") if a.variants else [""]
    hots = [int(x) for x in a.hot.split(",")] if info["layout_name"] == "global" else [-1]
    for var, waves, hot in [(v, w, h) for v in variants for w in [int(x) for x in a.waves.split(",")] for h in hots]:
        for k_v in [kv for kv in var.split(",") if kv]:
            dfa.tune(int(k_v.split("=")[0]), int(k_v.split("=")[1]))
        if hot >= 0:
            dfa.tune(hip.KNOB_HOT_BYTES, hot)
        if waves:
            dfa.tune(hip.KNOB_WAVES, waves)
        ms = []
        for _ in range(a.reps + 1):
            dfa.exec_batch_device(buf.data_ptr(), L, n, end.data_ptr(), 0, stream=stream)
            ms.append(dfa.last_kernel_ms())
        torch.cuda.synchronize()
        ms = min(ms[1:])
        if ref is None:
            ref = end.clone()
            k = 512
            want = Oracle(flat).table_walk(buf[:k].cpu().numpy())
            same = np.array_equal(want, end[:k].cpu().numpy().view(np.uint32))
            print(f"# first variant vs oracle on {k} rows: {'OK' if same else 'MISMATCH'}; accepts={int((end != -1).sum())} "
                  f"distinct end states={int(torch.unique(end).numel())}", flush=True)
        else:
            same = bool(torch.equal(ref, end))
        i2 = dfa.info()
        print(f"c5 [{var}] hot={hot:7d} waves={i2['waves_per_block']:2d} lds={i2['lds_bytes']:6d} ms={ms:9.3f} "
              f"GB/s={n * L / ms / 1e6:8.1f} {'ok' if same else 'DIFF'}", flush=True)


if __name__ == "__main__":
    main()
