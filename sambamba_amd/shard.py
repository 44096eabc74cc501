"""Contig sharding of ONE BAM across ranks (BASELINE config 4: `depth region` contig-sharded over 8 GPUs).

The path shards by reference position: every rank takes a run of consecutive contigs (balanced by
length), restricts the engine to them with sbx_set_regions (the engine then inflates only the BGZF
blocks that the BAI lists for those contigs) and computes its share of the per-region / per-window
statistics.  Outputs of different ranks are disjoint, so the only exchange is a gather of the small
stat arrays (torch.distributed: RCCL on GPUs, gloo in the CPU tests) -- no per-position counters ever
cross xGMI.
"""
from typing import List, Sequence, Tuple


def plan_contig_shards(ref_lengths: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """[first_ref, last_ref) per rank: consecutive contigs, cut where the running length crosses k/world.

    Consecutive runs keep every rank's BGZF block range contiguous; the cut points are the ones closest to
    equal shares of the total length (greedy prefix cuts)."""
    n = len(ref_lengths)
    total = float(sum(ref_lengths)) or 1.0
    cuts = [0]
    acc = 0.0
    r = 0
    for k in range(1, world):
        target = total * k / world
        while r < n and acc + ref_lengths[r] / 2.0 <= target:
            acc += ref_lengths[r]
            r += 1
        cuts.append(r)
    cuts.append(n)
    return [(cuts[i], max(cuts[i], cuts[i + 1])) for i in range(world)]


def regions_of_shard(ref_lengths: Sequence[int], shard: Tuple[int, int]) -> List[Tuple[int, int, int]]:
    """Whole-contig regions (ref_id, 0, length) of a shard, for sbx_set_regions."""
    return [(r, 0, int(ref_lengths[r])) for r in range(shard[0], shard[1]) if ref_lengths[r] > 0]


def owner_of_region(shards: Sequence[Tuple[int, int]], ref_id: int) -> int:
    for rank, (a, b) in enumerate(shards):
        if a <= ref_id < b:
            return rank
    return -1


def gather_region_stats(local_rows, dist=None):
    """All-gather per-rank lists of (region_index, payload) rows and return them merged by region index.

    `dist` is torch.distributed (initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(local_rows, key=lambda r: r[0])
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, list(local_rows))
    merged = [row for part in gathered for row in part]
    return sorted(merged, key=lambda r: r[0])
