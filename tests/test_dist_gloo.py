"""N > 1 path of bench.py on CPU: world_size-2 gloo.  The sharding is the whole multi-GPU story
(independent instances, contiguous id slices per rank, no data-path collective) -- the
only collectives are the barrier, the max-over-ranks of the step time and the sum of solved counts.
The per-rank work is done by the CPU oracle here (no GPU in this container)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, per_rank, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time
    import torch
    import torch.distributed as dist
    import bench
    import oracle_py as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = bench.shard_range(rank, world, per_rank)
    dist.barrier()
    t0 = time.perf_counter()
    ok, x, y, st = O.synth_batch_solve(first, last - first, 16, 20, 4, opt=O.default_options(perturbStep=0), threads=1)
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([ok], dtype=torch.int64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    ids = torch.tensor(list(range(first, last)), dtype=torch.int64)
    gathered = [torch.zeros_like(ids) for _ in range(world)]
    dist.all_gather(gathered, ids)
    if rank == 0:
        q.put((float(el.item()), int(cnt.item()), torch.cat(gathered).tolist(), x))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_ids():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8):
        ids = []
        for r in range(world):
            a, b = bench.shard_range(r, world, 1024)
            ids += list(range(a, b))
        assert ids == list(range(1024 * world))       # BASELINE config C4: 8192 = 8 x 1024, contiguous slices


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    per_rank = 3
    procs = [ctx.Process(target=_worker, args=(r, 2, port, per_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    elapsed, solved, ids, x0 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert solved == 2 * per_rank and ids == list(range(2 * per_rank)) and elapsed > 0
    ok, x, _, _ = oracle.synth_batch_solve(0, per_rank, 16, 20, 4, opt=oracle.default_options(perturbStep=0), threads=1)
    assert np.array_equal(x, x0)      # rank 0's shard is exactly instances [0, per_rank)
