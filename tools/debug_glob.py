import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import libfsm_amd as hip
from oracle.pyoracle import RefFsm, Oracle
hip.load_library()
torch.cuda.set_device(0)
f = RefFsm.re_comp("glob", b"foo*bar?", 0, True, True, endid=5); f.shuffle(1234)
flat = f.flatten()
rng = np.random.RandomState(17)
alpha = np.frombuffer(b"abcdefgxLlibsm0123456789:? \0\xff", np.uint8)
strings = [bytes(alpha[rng.randint(0, len(alpha), rng.randint(0, 40))]) for _ in range(3000)]
strings += [b"foo*bar?", b"", b"libfsm", b"abbbcdefg", b"foobar", b"fooXXbarz", b"ab:cd:ef", b"abcabdx"]
ret, want = f.exec_strings(strings)
off = np.zeros(len(strings) + 1, np.uint64); off[1:] = np.cumsum([len(s) for s in strings])
base = np.frombuffer(b"".join(strings), np.uint8)
n = len(strings)
d_base = torch.from_numpy(np.concatenate([base, np.zeros(64, np.uint8)])).cuda()
d_off = torch.from_numpy(off.view(np.int64).copy()).cuda()
d_end = torch.empty(n, dtype=torch.int32, device="cuda")
T = 30
for L in (hip.LAYOUT_TINY, hip.LAYOUT_LDS, hip.LAYOUT_COMB, hip.LAYOUT_GLOBAL):
    dfa = hip.HipDfa(flat, L)
    for waves in (16, 4, 1):
        dfa.tune(hip.KNOB_WAVES, waves)
        nbad_dev = nbad_host = 0
        first = None
        for t in range(T):
            d_end.fill_(-7)
            torch.cuda.synchronize()
            dfa.exec_batch_offsets_device(d_base.data_ptr(), d_off.data_ptr(), n, d_end.data_ptr(), 0)
            torch.cuda.synchronize()
            e = d_end.cpu().numpy().view(np.uint32)
            bad = np.nonzero(e != want)[0]
            if len(bad):
                nbad_dev += 1
                if first is None: first = [(int(i), int(e[i]), int(want[i])) for i in bad[:5]]
            e2, _ = dfa.exec_strings(strings)
            if not np.array_equal(e2, want): nbad_host += 1
        print(f"layout {dfa.info()['layout_name']:6s} waves={waves:2d}: device-front bad trials {nbad_dev}/{T}, host-front bad trials {nbad_host}/{T}", first, flush=True)
    dfa.close()
