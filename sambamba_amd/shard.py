"""Sharding of ONE BAM across the ranks of a torch.distributed job.

The path shards by reference position (the reference's analogue is pileupChunks,
BioD/bio/std/hts/bam/pileup.d:1011-1015): every rank takes a contiguous slice of the concatenated reference
-- whole contigs and, where a contig is cut, position intervals inside it --, fetches the reads overlapping
its slice through the BAI (sbx_run_interval: the chunk arithmetic of randomaccessmanager.d:247-294 plus the
linear-index cut of baifile.d:75-80) and clips its contributions to the slice.  A read near a cut is seen by
both neighbours; each counts only the positions it owns, so per-position outputs need no reduction, only
concatenation.  With --fix-mate-overlaps both mates of a pair that straddles a cut are seen by both sides
(each overlaps the slice of every position where the pair is resolved).  Regions and windows are never split:
cuts are aligned to the window step, and a BED region belongs to the rank that owns its first position.
The only exchange is a gather of small stat rows / of the finished text (torch.distributed: RCCL on GPUs, gloo
in the CPU tests) -- no per-position counters ever cross xGMI.
"""
from typing import List, Sequence, Tuple

Interval = Tuple[int, int, int]     # (ref_id, beg, end), 0-based half-open


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


def plan_position_shards(ref_lengths: Sequence[int], world: int, align: int = 1024) -> List[List[Interval]]:
    """Per rank, the intervals (ref_id, beg, end) of its slice of the concatenated reference: equal shares of the total
    length, cuts inside a contig rounded to a multiple of `align` (the tile size, or the window step) so that no tile /
    window is split.  Every position of every contig belongs to exactly one rank; a rank's intervals are in genome order."""
    import bisect
    n_ref = len(ref_lengths)
    lens = [max(0, int(x)) for x in ref_lengths]
    total = sum(lens)
    starts, acc = [], 0
    for L in lens:
        starts.append(acc)
        acc += L
    out: List[List[Interval]] = [[] for _ in range(world)]
    if total == 0:
        return out

    def locate(g):      # global position 0 <= g < total -> (ref, position aligned down)
        r = bisect.bisect_right(starts, g) - 1
        return r, (g - starts[r]) // align * align

    bounds = [(0, 0)] + [locate(total * k // world) for k in range(1, world)] + [(n_ref, 0)]
    for k in range(1, world):       # keep the cuts monotone (tiny contigs, more ranks than tiles)
        if bounds[k] < bounds[k - 1]:
            bounds[k] = bounds[k - 1]
    for k in range(world):
        (r0, p0), (r1, p1) = bounds[k], bounds[k + 1]
        for r in range(r0, min(r1, n_ref - 1) + 1):
            beg = p0 if r == r0 else 0
            end = p1 if r == r1 else lens[r]
            if end > beg:
                out[k].append((r, beg, end))
    return out


def regions_of_shard(ref_lengths: Sequence[int], shard: Tuple[int, int]) -> List[Interval]:
    """Whole-contig regions (ref_id, 0, length) of a contig shard, for sbx_set_regions."""
    return [(r, 0, int(ref_lengths[r])) for r in range(shard[0], shard[1]) if ref_lengths[r] > 0]


def owner_of_region(shards: Sequence[Tuple[int, int]], ref_id: int) -> int:
    for rank, (a, b) in enumerate(shards):
        if a <= ref_id < b:
            return rank
    return -1


def owner_of_position(plan: Sequence[Sequence[Interval]], ref_id: int, pos: int) -> int:
    """Rank whose slice holds position `pos` of contig `ref_id` (-1: beyond the contig)."""
    for rank, ivs in enumerate(plan):
        for r, b, e in ivs:
            if r == ref_id and b <= pos < e:
                return rank
    return -1


def clip_regions_to_shards(regions: Sequence[Interval], mine: Sequence[Interval]) -> List[Interval]:
    """The regions a rank owns: those whose first position lies in its slice (a region is never split)."""
    out = []
    for (r, s, e) in regions:
        for (mr, mb, me) in mine:
            if r == mr and mb <= s < me:
                out.append((r, s, e))
                break
    return out


def merge_regions(regs: Sequence[Interval]) -> List[Interval]:
    """Sorted union of the regions (what sbx_set_regions takes)."""
    out: List[Interval] = []
    for r, s, e in sorted(regs):
        if out and out[-1][0] == r and out[-1][2] >= s:
            out[-1] = (r, out[-1][1], max(out[-1][2], e))
        else:
            out.append((r, s, e))
    return out


def read_bed_regions(path: str, ref_names: Sequence[str]) -> List[Interval]:
    """(ref_id, start, end) of a BED written by this harness (bench.py): three integer columns, known contigs only.
    User-supplied -L arguments go through Depth.parse_regions (the library's parser, identical to the CLI's)."""
    idx = {n: i for i, n in enumerate(ref_names)}
    out = []
    with open(path) as fh:
        for line in fh:
            f = line.split()
            if len(f) >= 3 and f[0] in idx:
                out.append((idx[f[0]], int(f[1]), int(f[2])))
    return out


# ---- exchange ---------------------------------------------------------------------------------------------------
def _dev(dist):
    import torch
    if dist is not None and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_rows(index, values, dist=None):
    """All-gather per-rank stat rows as tensors and return them merged and sorted by row index.

    index: int64 [n]; values: int64 [n, k] (same k on every rank).  Ranks may hold different numbers of rows: the
    lengths are exchanged first and the payload travels as one padded tensor per rank (RCCL all_gather on GPUs)."""
    import torch
    index = torch.as_tensor(index, dtype=torch.int64).reshape(-1)
    values = torch.as_tensor(values, dtype=torch.int64).reshape(index.numel(), -1)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        order = torch.argsort(index)
        return index[order], values[order]
    dev = _dev(dist)
    world = dist.get_world_size()
    k = torch.tensor([values.shape[1]], dtype=torch.int64, device=dev)
    dist.all_reduce(k, op=dist.ReduceOp.MAX)          # a rank without rows does not know k
    k = int(k.item())
    n = torch.tensor([index.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    cap = max(1, max(sizes))
    buf = torch.zeros((cap, k + 1), dtype=torch.int64, device=dev)
    if index.numel():
        buf[:index.numel(), 0] = index.to(dev)
        buf[:index.numel(), 1:1 + values.shape[1]] = values.to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    merged = torch.cat([p[:sizes[i]] for i, p in enumerate(parts)], dim=0).cpu()
    order = torch.argsort(merged[:, 0], stable=True)
    merged = merged[order]
    return merged[:, 0], merged[:, 1:]


def gather_region_stats(local_rows, dist=None):
    """(kept for small, irregular payloads) all-gather lists of (index, payload) rows as Python objects."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(local_rows, key=lambda r: r[0])
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, list(local_rows))
    merged = [row for part in gathered for row in part]
    return sorted(merged, key=lambda r: r[0])


def text_group(dist):
    """The process group finished text travels through: host memory.  With RCCL as the job's backend a second, gloo group
    is created for it -- text is produced in pinned host buffers and consumed by a host writer, sending it through HBM
    and xGMI would only add two copies (call on every rank, once)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    return dist.new_group(backend="gloo")


def exclusive_offset(nbytes, dist=None):
    """(sum of `nbytes` over the lower ranks, sum over all ranks): where this rank's byte range starts in a shared output."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0, int(nbytes)
    dev = _dev(dist)
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = torch.tensor([int(nbytes)], dtype=torch.int64, device=dev)
    parts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine)
    sizes = [int(x.item()) for x in parts]
    return sum(sizes[:rank]), sum(sizes)


class TextFunnel:
    """Rank-ordered concatenation of the ranks' text on rank 0, streamed: a rank calls write(chunk) as its pieces are
    produced and end() when it is done; rank 0 first drains its own writes straight to `sink`, then receives rank 1's
    pieces, rank 2's, ...  Point to point through host memory (`group`: gloo).  The other ranks send without blocking and
    keep up to `max_in_flight` pieces posted (bounded memory), so that they format and copy their text while rank 0 is still
    busy with its own instead of stalling at their first piece."""

    def __init__(self, dist, group, sink, max_in_flight=8):
        self.dist, self.group, self.sink = dist, group, sink
        self.solo = dist is None or not dist.is_initialized() or dist.get_world_size() == 1
        self.rank = 0 if self.solo else dist.get_rank()
        self.max_in_flight = max_in_flight
        self._posted = []           # (work handle, tensor kept alive) of the sends not yet known to be complete

    def _post(self, t):
        self._posted.append((self.dist.isend(t, dst=0, group=self.group), t))
        while len(self._posted) > 2 * self.max_in_flight:        # (two sends per piece: its length and its bytes)
            w, _ = self._posted.pop(0)
            w.wait()

    def begin(self):
        """Rank 0 starts RECEIVING now, on a helper thread, while it formats and writes its own text: the pieces of rank 1, 2, ... are
        taken off the wire in rank order into a bounded queue (rank 0 used to post its first receive only after its own
        text was out, so the other ranks' sends sat in their `max_in_flight` window until then); end() drains the queue into the sink."""
        if self.solo or self.rank != 0:
            return
        import queue
        import threading
        import torch
        self._queue = queue.Queue(maxsize=4 * self.max_in_flight)
        self._recv_error = None

        def receiver():
            try:
                for src in range(1, self.dist.get_world_size()):
                    while True:
                        n = torch.zeros(1, dtype=torch.int64)
                        self.dist.recv(n, src=src, group=self.group)
                        n = int(n.item())
                        if n < 0:
                            break
                        buf = torch.empty(n, dtype=torch.uint8)
                        self.dist.recv(buf, src=src, group=self.group)
                        self._queue.put(buf)
            except Exception as e:          # (reported by end(): a lost rank must not leave rank 0 waiting for the queue)
                self._recv_error = e
            self._queue.put(None)

        self._receiver = threading.Thread(target=receiver, name="sbx-text-funnel", daemon=True)
        self._receiver.start()

    def write(self, chunk):
        import torch
        if self.solo or self.rank == 0:
            self.sink(chunk)
            return
        if not len(chunk):
            return
        self._post(torch.tensor([len(chunk)], dtype=torch.int64))
        self._post(torch.frombuffer(bytearray(chunk), dtype=torch.uint8))

    def end(self):
        import torch
        if self.solo:
            return
        if self.rank != 0:
            self._post(torch.tensor([-1], dtype=torch.int64))
            for w, _ in self._posted:
                w.wait()
            self._posted = []
            return
        if getattr(self, "_receiver", None) is None:
            self.begin()
        while True:
            buf = self._queue.get()
            if buf is None:
                break
            self.sink(buf.numpy().tobytes())
        self._receiver.join()
        self._receiver = None
        if self._recv_error is not None:
            raise self._recv_error


def send_text_to_rank0(chunks, dist, write, group=None):
    """Concatenate the ranks' text in rank order on rank 0 (TextFunnel over a list of chunks)."""
    if group is None:
        group = text_group(dist)
    f = TextFunnel(dist, group, write)
    f.begin()
    for c in chunks:
        f.write(c)
    f.end()
