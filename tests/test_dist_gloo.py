"""World-size-2/3 gloo tests of the frame sharding layer (CPU; the compute is a stand-in since the
HIP engine needs a GPU -- the sharding logic is what is under test)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stereo_reconstruction_cv_amd import dist as D


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        D.shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_compute(l, r):
    # stands in for the engine: something frame-wise and order-sensitive
    return (l.to(torch.int16) * 3 - r.to(torch.int16)).contiguous()


def _fake_compute_pair(l, r):
    # (disparity, XYZ)-shaped result: two tensors of different dtype and rank per frame
    d = _fake_compute(l, r)
    return d, torch.stack([d.float(), d.float() * 0.5, d.float() + 1.0], dim=-1)


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        lefts = torch.randint(0, 256, (n, 6, 10), dtype=torch.uint8, generator=g)
        rights = torch.randint(0, 256, (n, 6, 10), dtype=torch.uint8, generator=g)
        out = D.run_sharded(_fake_compute, lefts if rank == 0 else None, rights if rank == 0 else None)
        lo, hi = D.shard_range(n, rank, world)
        both = D.run_sharded(_fake_compute_pair, lefts if rank == 0 else None, rights if rank == 0 else None)
        if rank == 0:
            want_d, want_x = _fake_compute_pair(lefts, rights)
            assert isinstance(both, tuple) and torch.equal(both[0], want_d) and torch.equal(both[1], want_x)
            q.put(("result", out.numpy(), _fake_compute(lefts, rights).numpy()))
        else:
            assert both is None
        q.put(("span", rank, lo, hi))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 5), (2, 4), (3, 2), (2, 1)])
def test_scatter_compute_gather(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=120) for _ in range(world + 1)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res = [m for m in msgs if m[0] == "result"][0]
    assert np.array_equal(res[1], res[2])
    spans = sorted((m[1], m[2], m[3]) for m in msgs if m[0] == "span")
    assert spans[0][1] == 0 and spans[-1][2] == n
