"""sbx_plan_shards (the C ABI's position-sharding plan, used by `sbx-depth --gpus N` and the D glue) against
shard.plan_position_shards (the torch.distributed driver's): one rule, two statements of it -- plus the properties the
rule promises: every position of every contig in exactly one shard, cuts inside contigs aligned, shares balanced."""
import ctypes as C
import random

import pytest

import sambamba_amd
from sambamba_amd.shard import plan_position_shards


class Shard(C.Structure):
    _fields_ = [("shard", C.c_uint32), ("ref_id", C.c_uint32), ("beg", C.c_uint32), ("end", C.c_uint32)]


def c_plan(lengths, world, align):
    L = sambamba_amd.lib()
    arr = (C.c_int64 * max(1, len(lengths)))(*lengths)
    n = C.c_size_t(0)
    assert L.sbx_plan_shards(arr, len(lengths), world, align, None, 0, C.byref(n)) == 0
    out = (Shard * max(1, n.value))()
    assert L.sbx_plan_shards(arr, len(lengths), world, align, out, n.value, C.byref(n)) == 0
    plan = [[] for _ in range(world)]
    for i in range(n.value):
        plan[out[i].shard].append((out[i].ref_id, out[i].beg, out[i].end))
    return plan


GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622,
          133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468,
          156040895, 57227415, 16569]


@pytest.mark.parametrize("lengths,world,align", [
    ([248956422], 8, 1024), (GRCH38, 8, 1024), (GRCH38, 8, 1000), (GRCH38, 3, 1024), ([5000, 0, 300, 0, 70000], 4, 1024),
    ([10, 20, 30], 7, 1024), ([0, 0], 2, 16), ([1 << 31 - 1], 2, 1024), ([100000] * 40, 1, 1024)])
def test_the_two_statements_of_the_rule_agree(lengths, world, align):
    assert c_plan(lengths, world, align) == [[tuple(iv) for iv in ivs] for ivs in plan_position_shards(lengths, world, align=align)]


def test_random_dictionaries_agree_and_partition_every_position():
    rng = random.Random(6)
    for _ in range(300):
        n = rng.randint(1, 30)
        lengths = [rng.choice([0, rng.randint(1, 50), rng.randint(1000, 3_000_000)]) for _ in range(n)]
        world = rng.randint(1, 9)
        align = rng.choice([1, 16, 1000, 1024])
        plan = c_plan(lengths, world, align)
        assert plan == [[tuple(iv) for iv in ivs] for ivs in plan_position_shards(lengths, world, align=align)]
        seen = {}
        flat = [iv for ivs in plan for iv in ivs]
        assert flat == sorted(flat)                                   # genome order across the shards
        for r, b, e in flat:
            assert 0 <= b < e <= lengths[r]
            assert b % align == 0 and (e % align == 0 or e == lengths[r])
            assert seen.get(r, 0) == b                                 # contiguous, no gap, no overlap
            seen[r] = e
        assert all(seen.get(r, 0) == L for r, L in enumerate(lengths))


def test_arguments_are_checked_and_small_buffers_reported():
    L = sambamba_amd.lib()
    arr = (C.c_int64 * 2)(5000, 7000)
    n = C.c_size_t(0)
    assert L.sbx_plan_shards(arr, 2, 0, 1024, None, 0, C.byref(n)) == -1          # SBX_EINVAL
    assert L.sbx_plan_shards(arr, 2, 2, 0, None, 0, C.byref(n)) == -1
    out = (Shard * 1)()
    assert L.sbx_plan_shards(arr, 2, 4, 1024, out, 1, C.byref(n)) == -8 and n.value > 1      # SBX_ENOMEM, the count it needs
    assert L.sbx_device_count() >= 0
