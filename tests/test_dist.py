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
