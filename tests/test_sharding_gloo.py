"""CPU, world_size = 2 over gloo: the multi-GPU path of bench.py / configs[4] — utterances are sharded in
contiguous blocks with no data-path collective; only the wall-time max and the audio-seconds sum are reduced."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from zerovox_cpp_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = sharding.mixed_length_batch(3, 9)
    b, e = sharding.shard_utterances(len(lens), world, rank)
    # every rank "synthesises" its own utterances: audio seconds ~ frames, wall differs per rank
    audio = float(sum(lens[b:e]))
    wall = 1.0 + rank
    xrt = sharding.aggregate_throughput(audio, wall)
    q.put((rank, b, e, xrt, float(sum(lens)) / world_max_wall(world)))
    dist.destroy_process_group()


def world_max_wall(world):
    return float(world)       # walls are 1..world


def test_two_rank_sharding_and_aggregation():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, b0, e0, x0, want0), (r1, b1, e1, x1, want1) = res
    assert (b0, e0, b1, e1) == (0, 5, 5, 9)                   # contiguous blocks cover all 9 utterances once
    assert x0 == pytest.approx(want0) and x1 == pytest.approx(want1) and x0 == x1


def test_shard_edges():
    sys.path.insert(0, ROOT)
    from zerovox_cpp_amd import sharding
    assert [sharding.shard_utterances(256, 8, r) for r in (0, 7)] == [(0, 32), (224, 256)]
    assert sharding.shard_utterances(3, 8, 5) == (3, 3)        # more ranks than utterances: empty shard
    got = [sharding.shard_utterances(10, 4, r) for r in range(4)]
    assert got == [(0, 3), (3, 6), (6, 9), (9, 10)]
    lens = sharding.mixed_length_batch(1, 1000)
    assert min(lens) >= 32 and max(lens) <= 256 and lens == sharding.mixed_length_batch(1, 1000)
