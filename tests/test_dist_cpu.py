"""world_size-2 gloo tests of the N>1 path (sample sharding, the single table-image broadcast,
max-over-ranks timing) on CPU tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from varigraph_amd import dist as vdist


class FakeCtx:
    """Stands in for vgmi.Context: same table-image and counter methods over CPU tensors."""

    def __init__(self, image=None, counts=None):
        self.image = image
        self.imported = None
        self.counts = counts
        self.n_keys = 0 if counts is None else counts.numel()

    def counts_export_device(self, t):
        t.copy_(self.counts)

    def counts_import_device(self, t):
        self.counts = t.clone()

    def table_image_bytes(self):
        return int(self.image.numel())

    def table_export(self, t):
        t.copy_(self.image)

    def table_import(self, t):
        self.imported = t.clone()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        g = torch.Generator().manual_seed(1234)
        image = torch.randint(0, 256, (100_003,), dtype=torch.uint8, generator=g)
        ctx = FakeCtx(image if rank == 0 else None)
        n = vdist.broadcast_table_image(ctx, dist, rank, dev)
        ok_img = n == 100_003 and (rank == 0 or bool(torch.equal(ctx.imported, image)))
        arrays = None
        if rank == 0:
            arrays = {"node_off": np.arange(11, dtype=np.uint64) * 3, "flags": np.arange(30, dtype=np.uint8) % 2,
                      "idx": np.arange(30, dtype=np.uint32)[::-1].copy()}
        got = vdist.broadcast_arrays(arrays, dist, rank, dev)
        ok_arr = (got["node_off"].dtype == np.uint64 and got["node_off"][-1] == 30 and got["idx"][0] == 29
                  and got["flags"].sum() == 15)
        # read-sharded sample: raw counters summed over ranks
        cctx = FakeCtx(counts=torch.arange(10, dtype=torch.int32) * (rank + 1))
        vdist.allreduce_counts(cctx, dist, dev)
        ok_arr = ok_arr and bool(torch.equal(cctx.counts, torch.arange(10, dtype=torch.int32) * 3))
        t = vdist.max_over_ranks(1.0 + rank, dist, dev)
        s = vdist.sum_over_ranks(10.0 * (rank + 1), dist, dev)
        q.put((rank, ok_img, ok_arr, t, s, vdist.shard_samples(8, world, rank)))
    finally:
        dist.destroy_process_group()


def test_two_rank_broadcast_and_reduction():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_img, ok_arr, t, s, shard in res:
        assert ok_img and ok_arr
        assert t == 2.0 and s == 30.0
        assert shard == list(range(rank, 8, 2))


def test_shard_samples_partition():
    for world in (1, 2, 3, 4, 8):
        allidx = sorted(i for r in range(world) for i in vdist.shard_samples(8, world, r))
        assert allidx == list(range(8))
    assert vdist.shard_samples(3, 8, 5) == []
