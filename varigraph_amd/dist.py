"""Multi-GPU plumbing: one process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm).

The hot path shards by SAMPLE: the reference already processes samples one after
another with a full state reset (src/varigraph.cpp:158-171), so sample s simply goes to rank
s mod world.  The only exchange is ONE broadcast of the read-only table image from the rank that
parsed graph.bin; there is no data-path collective.  The same functions run on CPU tensors with
the gloo backend (tests/test_dist_cpu.py)."""
import os

import numpy as np


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def _sync(device):
    """Collectives on the nccl/RCCL backend are asynchronous to the host; the vgmi context works on
    its own HIP streams, so make the collective's result visible before handing buffers over."""
    import torch
    if getattr(device, "type", str(device)) == "cuda" or str(device).startswith("cuda"):
        torch.cuda.synchronize(device)


def shard_samples(n_samples, world, rank):
    """Indices of the samples rank `rank` genotypes (round robin: sample s -> rank s mod world)."""
    return list(range(rank, n_samples, world))


def broadcast_table_image(ctx, dist, rank, device, src=0, ctx_device=None):
    """Rank `src` exports its device table image, everybody else imports it.
    `ctx` needs table_image_bytes() / table_export(tensor) / table_import(tensor).
    `device` is where the collective runs (cuda for RCCL); `ctx_device` is where the context lives,
    if different (debug runs over gloo stage through host memory)."""
    import torch
    ctx_device = ctx_device or device
    sz = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == src:
        sz[0] = ctx.table_image_bytes()
    dist.broadcast(sz, src)
    nbytes = int(sz.item())
    img = torch.empty(nbytes, dtype=torch.uint8, device=ctx_device)
    if rank == src:
        ctx.table_export(img)
    wire = img if str(ctx_device) == str(device) else img.to(device)
    dist.broadcast(wire, src)
    _sync(device)
    if rank != src:
        img = wire if wire is img else wire.to(ctx_device)
        ctx.table_import(img)
    return nbytes


def broadcast_arrays(arrays, dist, rank, device, src=0):
    """Broadcast a dict of numpy arrays (node CSR, flags ...) from `src`; returns the dict everywhere."""
    import torch
    meta = [None]
    if rank == src:
        meta = [[(k, str(v.dtype), tuple(v.shape)) for k, v in arrays.items()]]
    dist.broadcast_object_list(meta, src)
    out = {}
    for name, dtype, shape in meta[0]:
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(arrays[name]).view(np.uint8).reshape(-1)).to(device)
        else:
            t = torch.empty(int(np.prod(shape)) * np.dtype(dtype).itemsize, dtype=torch.uint8, device=device)
        dist.broadcast(t, src)
        out[name] = t.cpu().numpy().view(dtype).reshape(shape).copy()
    return out


def max_over_ranks(value, dist, device):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist, device):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allreduce_counts(ctx, dist, device, ctx_device=None):
    """Read-sharded single sample: sum the raw per-key counters over all ranks (one RCCL all-reduce)
    and write the totals back, so that counts_finish() on any rank yields min(255, global total)."""
    import torch
    ctx_device = ctx_device or device
    t = torch.empty(max(ctx.n_keys, 1), dtype=torch.int32, device=ctx_device)
    ctx.counts_export_device(t)
    wire = t if str(ctx_device) == str(device) else t.to(device)
    dist.all_reduce(wire, op=dist.ReduceOp.SUM)
    _sync(device)
    t = wire if wire is t else wire.to(ctx_device)
    ctx.counts_import_device(t)
    return t
