"""CPU, world_size 2, gloo: the N>1 path of bench.py -- contiguous index-range shards,
counter-based generator keyed by GLOBAL index, all-gather of the accept bitmaps --
reproduces the single-process result bit for bit.  (The per-shard walk is done by the
oracle here; on GPUs the same plumbing wraps the HIP kernel.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from libfsm_amd.shard import bitmap_words, gather_bitmap, gather_bitmap_ragged, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 63, 64, 65, 1000, 4096, 100_000_037):
        for world in (1, 2, 3, 4, 8):
            pos = 0
            sizes = []
            for r in range(world):
                first, cnt = shard_range(n, r, world)
                assert first == pos and (first % 64 == 0 or cnt == 0)
                pos += cnt
                sizes.append(cnt)
            assert pos == n
            assert max(sizes) - min(sizes) < 128


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bitmap(end):
    n = len(end)
    bits = np.zeros(bitmap_words(n) * 64, np.uint8)
    bits[:n] = end != 0xFFFFFFFF
    return np.packbits(bits, bitorder="little").view(np.int64)


def _worker(rank, world, port, n_total, ragged, out_path):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from common import GOLDEN, Golden
    from libfsm_amd import gen_inputs_host
    from oracle.pyoracle import Oracle
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    o = Oracle(g.flat)
    if ragged:
        counts = [shard_range(n_total, r, world)[1] for r in range(world)]
        first, cnt = shard_range(n_total, rank, world)
    else:
        cnt = n_total // world
        first = rank * cnt
        counts = [cnt] * world
    rows = gen_inputs_host(cnt, 128, first, 99, None, b"Libfsm", 8)   # keyed by global index
    local = torch.from_numpy(_bitmap(o.table_walk(rows)).copy())
    got = gather_bitmap_ragged(local, counts) if ragged else gather_bitmap(local, world)
    acc = torch.tensor([int((o.table_walk(rows) != 0xFFFFFFFF).sum())])
    dist.all_reduce(acc)
    if rank == 0:
        np.save(out_path, np.concatenate([got.numpy(), acc.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,ragged", [(4096, False), (5000, True)])
def test_two_rank_gather_equals_single_process(tmp_path, built, n_total, ragged):
    from common import GOLDEN, Golden
    from libfsm_amd import gen_inputs_host
    from oracle.pyoracle import Oracle
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), n_total, ragged, out), nprocs=2, join=True)
    got = np.load(out)
    g = Golden(os.path.join(GOLDEN, "c1.npz"))
    whole = Oracle(g.flat).table_walk(gen_inputs_host(n_total, 128, 0, 99, None, b"Libfsm", 8))
    assert np.array_equal(got[:-1], _bitmap(whole))
    assert int(got[-1]) == int((whole != 0xFFFFFFFF).sum())


# ---- many-DFA submissions shard BY DFA (fsm_hip_node_exec_multi / fsm_hip_multi_assign) ---------------------

def test_assign_by_dfa_is_the_c_fronts_rule(built):
    """libfsm_amd.shard.assign_by_dfa == fsm_hip_multi_assign (largest first, least loaded device, ties low), and the split is
    balanced: no device carries more than the mean plus the largest job."""
    import libfsm_amd as hip
    from libfsm_amd.shard import assign_by_dfa
    rng = np.random.RandomState(3)
    for _ in range(300):
        k, world = rng.randint(0, 60), rng.randint(1, 9)
        cost = (rng.randint(0, 5000, k) * rng.randint(0, 2, k)).astype(np.uint64)
        a = hip.multi_assign(cost, world)
        assert list(a) == assign_by_dfa(list(cost), world)
        if k:
            load = np.bincount(a, weights=np.maximum(cost, 1).astype(np.float64), minlength=world)
            assert load.max() <= np.maximum(cost, 1).sum() / world + max(int(cost.max()), 1)


def _multi_worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from common import Golden, all_golden_paths
    from libfsm_amd.shard import assign_by_dfa, job_cost
    from oracle.pyoracle import Oracle
    gs = [Golden(p) for p in all_golden_paths() if "/retest/" in p]
    jobs = [g.strings() for g in gs]
    owner = assign_by_dfa([job_cost(len(j), sum(len(x) for x in j)) for j in jobs], world)
    nmax = max(len(j) for j in jobs)
    mine = torch.full((len(jobs), nmax), -2, dtype=torch.int64)          # -2: not mine
    for q, g in enumerate(gs):
        if owner[q] != rank:
            continue
        rows, lens = g.padded_rows()
        end = Oracle(g.flat).table_walk(rows, lens)                          # (on GPUs: this rank's fsm_hip_exec_multi over its jobs)
        mine[q, :len(end)] = torch.from_numpy(end.astype(np.int64))
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)                                             # the "trivial gather of match results"
    if rank == 0:
        full = torch.stack(parts).max(dim=0).values.numpy()                  # every job was walked by exactly one rank
        assert all(sum(int(parts[r][q, 0] != -2) for r in range(world)) == 1 for q in range(len(jobs)) if len(jobs[q]))
        np.save(out_path, np.concatenate([full.reshape(-1), np.array(owner)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_the_retest_goldens_by_dfa(tmp_path, built):
    """world_size 2, gloo: the 37 retest automata split by DFA with the C front's rule, each rank walks its own jobs, one
    all_gather of the per-job results: every line's end state equals the reference's frozen answer, both ranks got work."""
    from common import Golden, all_golden_paths
    out = str(tmp_path / "multi.npy")
    mp.spawn(_multi_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    gs = [Golden(p) for p in all_golden_paths() if "/retest/" in p]
    assert len(gs) == 37
    nmax = max(len(g.strings()) for g in gs)
    ends, owner = got[:len(gs) * nmax].reshape(len(gs), nmax), got[len(gs) * nmax:]
    assert set(owner.tolist()) == {0, 1}
    for q, g in enumerate(gs):
        want = np.where(g.ret == 1, g.end, 0xFFFFFFFF).astype(np.int64)
        assert np.array_equal(ends[q, :len(want)], want), g.name
