"""The N > 1 path on real contexts: two ranks (gloo rendezvous, both on GPU 0 -- the box has one) hand the table image
from the rank that built it to the other through varigraph_amd.dist, each counts its own sample and checks it against
the oracle; `bench.py --gpus 2` starts its own ranks; the CLI builds the table once for several devices."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, get_cohort

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    from varigraph_amd import dist as vdist
    from varigraph_amd import vgmi
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cohort = get_cohort("c1")
        keys = cohort.graph.keys
        ctx = vgmi.Context(0, buffer_mib=16)
        if rank == 0:
            ctx.table_upload(keys, 27)
        nbytes = vdist.broadcast_table_image(ctx, dist, rank, torch.device("cpu"), ctx_device=torch.device("cuda", 0))
        assert ctx.table_info()["n_keys"] == keys.size
        # every rank its own sample (sample s -> rank s mod world)
        block = vgmi.synth_reads_host(1000 + rank, 0, 20_000, 150, cohort.haplotypes())
        ctx.counts_reset()
        ctx.reads_submit(block, 20_000)
        got, _, _ = ctx.counts_finish()
        t = o.Table(keys)
        t.count_block(block, 27)
        ok = bool(np.array_equal(got, t.counts()))
        # read-sharded mode on the same contexts: both halves of rank 0's sample, raw counters all-reduced
        b0 = vgmi.synth_reads_host(1000, 0, 20_000, 150, cohort.haplotypes())
        half = 10_000 * 151
        ctx.counts_reset()
        ctx.reads_submit(b0[rank * half:(rank + 1) * half], 10_000)
        ctx.counts_finish_device(None, None, None)
        vdist.allreduce_counts(ctx, dist, torch.device("cpu"), ctx_device=torch.device("cuda", 0))
        summed, _, _ = ctx.counts_finish()
        t.reset()
        t.count_block(b0, 27)
        ok_sum = bool(np.array_equal(summed, t.counts()))
        ctx.close()
        q.put((rank, nbytes, ok, ok_sum, int(got.sum())))
    finally:
        dist.destroy_process_group()


def test_two_ranks_real_contexts_broadcast_then_count():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] > 1_000_000
    for rank, nbytes, ok, ok_sum, total in res:
        assert ok and ok_sum and total > 0
    assert res[0][4] != res[1][4]   # different samples


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` (no launcher): two ranks, one JSON line with n_gpus 2 and the broadcast it did."""
    env = dict(os.environ, VGMI_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    # (round 6: the chr20-class and whole-genome-class legs run on both ranks too, at reduced size -- graph and haplotypes built on rank 0
    # only, table image and haplotypes handed on; at their stated sizes they are tests/test_gpu_large.py's and the default bench line's)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--reads", "4000000", "--no-sample-level",
                        "--steps", "3", "--warmup", "1", "--verify-reads", "200000",
                        "--c3-genome", "4000000", "--c3-variants", "66000", "--c3-reads", "1000000", "--c3-steps", "2",
                        "--c5-genome", "8000000", "--c5-variants", "100000", "--c5-reads", "1000000", "--c5-steps", "2"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2
    assert out["table_broadcast"]["bytes"] > 1_000_000
    assert out["verify"]["oracle_match"] is True
    assert out["rccl"]["nranks"] == 2 and len(out["rccl"]["kernel_ms_min_max_over_ranks"]) == 2
    for leg in ("c3", "c5"):
        assert "error" not in out[leg] and "skipped" not in out[leg], out[leg]
        assert out[leg]["verify"]["oracle_match"] is True and out[leg]["table_broadcast"]["bytes"] > 1_000_000
        assert out[leg]["value"] > 0 and out[leg]["roofline"]["kernel_ms"] > 0
    assert out["value"] > 0 and out["scaling"] == "weak"
