"""World-size-2 (and 3) gloo test of the multi-GPU host logic: contig sharding plan + gather of the small
per-region stat rows.  The device work of each rank is independent (no data-path collective), so what
needs a multi-process test is exactly this: disjoint, complete ownership and an order-preserving merge."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
          135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
          46709983, 50818468, 156040895, 57227415, 16569]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sambamba_amd.shard import plan_contig_shards, regions_of_shard, owner_of_region, gather_region_stats
    shards = plan_contig_shards(GRCH38, world)
    mine = shards[rank]
    # a fake BED of 1000 regions spread over the contigs, in input order
    import random
    rng = random.Random(7)
    bed = [(rng.randrange(len(GRCH38)), i) for i in range(1000)]
    local = [(i, ("rank%d" % rank, ref)) for (ref, i) in bed if owner_of_region(shards, ref) == rank]
    merged = gather_region_stats(local, dist)
    # max-over-ranks reduction used by bench.py for the timing
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put((shards, [m[0] for m in merged], [m[1][1] for m in merged], float(t.item()),
               regions_of_shard(GRCH38, mine)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_contig_sharding_and_gather(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    shards, idx, refs, tmax, regs0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # complete, disjoint, consecutive ownership
    assert shards[0][0] == 0 and shards[-1][1] == len(GRCH38)
    for a, b in zip(shards[:-1], shards[1:]):
        assert a[1] == b[0]
    # balanced within the largest contig
    sizes = [sum(GRCH38[a:b]) for a, b in shards]
    assert max(sizes) - min(sizes) <= max(GRCH38) + 1
    # merge is complete and in input order
    assert idx == list(range(1000))
    assert tmax == float(world)
    assert regs0[0] == (0, 0, GRCH38[0])


def test_plan_edge_cases():
    from sambamba_amd.shard import plan_contig_shards
    assert plan_contig_shards([100], 4) == [(0, 0), (0, 0), (0, 0), (0, 1)] or sum(b - a for a, b in plan_contig_shards([100], 4)) == 1
    s = plan_contig_shards([10, 10, 10, 10], 2)
    assert s == [(0, 2), (2, 4)]
    s = plan_contig_shards([], 2)
    assert s == [(0, 0), (0, 0)]


def _worker_rows(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sambamba_amd.shard import gather_rows, plan_position_shards, send_text_to_rank0
    import numpy as np
    plan = plan_position_shards([5_000_000], world, align=1000)      # ONE contig: shards by position only
    mine = plan[rank]
    # rows = windows of 1000 starting in my slice; payload = (window index * 7, rank)
    ids = [k for (r, b, e) in mine for k in range(b // 1000, e // 1000)]
    vals = np.array([[k * 7, rank] for k in ids], dtype=np.int64).reshape(len(ids), 2)
    idx, merged = gather_rows(np.array(ids, dtype=np.int64), vals, dist)
    # text in rank order
    got = []
    send_text_to_rank0([b"r%d:" % rank, b"x" * (rank + 1), b""], dist, got.append)
    if rank == 0:
        q.put((plan, idx.tolist(), merged.tolist(), b"".join(got)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_position_sharding_row_gather_and_text_order(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29660 + world
    procs = [ctx.Process(target=_worker_rows, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    plan, idx, merged, text = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the slices partition the contig, cuts on window boundaries
    flat = [iv for part in plan for iv in part]
    assert flat[0][1] == 0 and flat[-1][2] == 5_000_000
    for a, b in zip(flat[:-1], flat[1:]):
        assert a[2] == b[1] and a[2] % 1000 == 0
    assert idx == list(range(5000))
    assert [m[0] for m in merged] == [k * 7 for k in range(5000)]
    assert sorted(set(m[1] for m in merged)) == list(range(world))
    assert text == b"".join(b"r%d:" % r + b"x" * (r + 1) for r in range(world))


def test_position_shard_plan_properties():
    from sambamba_amd.shard import clip_regions_to_shards, owner_of_position, plan_position_shards
    for lens, world, align in ((GRCH38, 8, 1000), ([248956422], 8, 1024), ([100, 0, 50], 3, 16), ([1000], 4, 1024)):
        plan = plan_position_shards(lens, world, align)
        assert len(plan) == world
        seen = {}
        for part in plan:
            for r, b, e in part:
                assert 0 <= b < e <= lens[r]
                seen.setdefault(r, []).append((b, e))
        for r, L in enumerate(lens):
            cur = 0
            for b, e in sorted(seen.get(r, [])):
                assert b == cur and (b % align == 0)
                cur = e
            assert cur == L
        sizes = [sum(e - b for _, b, e in part) for part in plan]
        if sum(lens) > world * align * 4:
            assert max(sizes) - min(sizes) <= 2 * align + 2
    plan = plan_position_shards([10000, 10000], 2, 1000)
    assert owner_of_position(plan, 0, 9999) == 0 and owner_of_position(plan, 1, 0) == 1 and owner_of_position(plan, 1, 10000) == -1
    regs = [(0, 9990, 10000), (1, 0, 50), (0, 100, 20000)]
    assert clip_regions_to_shards(regs, plan[0]) == [(0, 9990, 10000), (0, 100, 20000)]
    assert clip_regions_to_shards(regs, plan[1]) == [(1, 0, 50)]
