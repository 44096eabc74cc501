"""`depth base|region|window` of ONE BAM sharded over the ranks of a torch.distributed job.

Every rank opens the BAM on its own GPU and takes a contiguous slice of the concatenated reference
(shard.plan_position_shards: whole contigs and, where a contig is cut, position intervals inside it -- so ONE long
contig shards as well as a whole genome does).  For every interval of its slice it runs the device pipeline over the
reads that overlap the interval (sbx_run_interval: only the BGZF blocks the BAI lists for it are uploaded and inflated)
and produces its share of the output:

  base    the text of the positions it owns, formatted on the device and streamed in bounded pieces (sbx_stream_base_rows).
          With -o every rank writes its own byte range of the output file (pwrite at the offset an exclusive scan of the
          ranks' text sizes gives it -- the sizes come from the device's measuring pass, no text is held); without -o
          the pieces travel to rank 0 as they are produced, through host memory (a gloo group), and rank 0 prints them
  window  the statistics of the windows that start in its slice (cuts are aligned to the window size, so no window is split)
  region  the statistics of the BED regions whose first position it owns (a region is never split)

Shards own disjoint outputs, so the only exchange is the gather of small stat rows (as tensors: RCCL all_gather on GPUs)
or of finished text; per-position counters never leave a GPU.  Rank 0 prints byte for byte what `sbx-depth` prints on one
GPU for the supported option set (tests/test_gpu_dist.py).

`base --reduce allreduce` is the alternative BASELINE.json's north_star names: the READS are partitioned
between the ranks by their start position (sbx_run_interval_owned; the reference's analogue is pileupChunks,
BioD/bio/std/hts/bam/pileup.d:1011-1015), every rank counts all the positions its reads cover, and the per-position counter
arrays are summed with an all-reduce (RCCL over xGMI on GPUs; int32 sums are the uint32 sums modulo 2^32).  It moves
28 bytes per reference position and sample through the ring where the default form moves nothing; it exists to be measured
next to it (bench.py `allreduce_option`), and prints the same text (tests/test_gpu_dist.py).

    python -m torch.distributed.run --nproc-per-node N -m sambamba_amd.dist_depth base in.bam -o out.txt
    python -m torch.distributed.run --nproc-per-node N -m sambamba_amd.dist_depth region -L x.bed -T 10 in.bam

Not replicated here (rejected with an error instead of printing something else; the single-GPU CLI handles them):
`window --overlap > 0` (the reference's ring bookkeeping is order dependent), `base -L` together with `-c 0` (stateful
consumption of the raw BED), and `window` on a genome whose LAST contigs have no reads (the reference continues the
previous contig's window coordinates there).
"""
import argparse
import os
import sys

import numpy as np

from . import Depth, SBX_MODE_BASE, SBX_MODE_REGION, SBX_MODE_WINDOW
from .shard import exclusive_offset, gather_rows, plan_position_shards, text_group, TextFunnel


def fmt_g(x):
    """D's write(float) == C's %g of the float32 value (depth.d:859-864)."""
    return "%g" % float(np.float32(x))


def region_row(prefix, length, n_reads, n_bases, cov, thresholds, sample, combined, annotate, min_cov, max_cov):
    """printRegionStats (depth.d:847-876) for one region and sample; None when the row is suppressed."""
    mean = np.float32(n_bases) / np.float32(length)
    ok = min_cov <= float(mean) <= max_cov
    if not ok and not annotate:
        return None
    row = prefix + str(int(n_reads)) + "\t" + fmt_g(mean)
    for t, c in zip(thresholds, cov):
        pct = np.float32(100.0) if t == 0 else np.float32(c) * np.float32(100) / np.float32(length)
        row += "\t" + fmt_g(pct)
    if not combined:
        row += "\t" + sample
    if annotate:
        row += "\ty" if ok else "\tn"
    return row + "\n"


MATE_SLACK = 16384     # first guess of the positions fetched left of a slice so that overlapping mates of its reads are in the run


class Unsupported(RuntimeError):
    pass


def _min_over_ranks(dist, value):
    if dist is None:
        return value
    import torch
    from .shard import _dev
    t = torch.tensor([value], dtype=torch.int64, device=_dev(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def _max_flags(dist, flags):
    if dist is None:
        return flags
    import torch
    from .shard import _dev
    t = torch.tensor(flags, dtype=torch.int64, device=_dev(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(x) for x in t.cpu()]


def first_column_in(d, ref, beg, end):
    """Position of the first pileup column of the resident run inside [beg, end) of contig `ref`, or None.  Only tiles
    that hold admitted reads are looked at (sbx_next_active_range), a few kilobytes at a time."""
    pos = beg
    while pos < end:
        r = d.next_active_range(ref, pos)
        if r is None or r[0] >= end:
            return None
        b, e = max(r[0], pos), min(r[1], end)
        step = 1 << 13
        for x in range(b, e, step):
            cov = d.covered(ref, x, min(e, x + step))      # (`covered` is what every kind of run keeps)
            nz = np.flatnonzero(cov)
            if len(nz):
                return x + int(nz[0])
        pos = e
    return None


def run_with_mate_slack(d, ref, beg, end, fix_mate, left_only=True):
    """sbx_run_interval over [beg - slack, end (+ slack)) with the slack --fix-mate-overlaps needs: a read that overlaps the
    slice pairs with a mate that overlaps IT, so the mate overlaps some position >= beg - span(read); fetching everything
    that overlaps [beg - S, ...) with S >= the longest alignment of the run brings every such mate in.  S starts at one
    linear-index window and is raised to what the run reports (spliced RNA-seq, long reads) -- never silently too small."""
    if not fix_mate:
        return d.run_interval(ref, beg, end), 0
    slack = MATE_SLACK
    for _ in range(4):
        lo = max(0, beg - slack)
        hi = end if left_only else end + slack
        st = d.run_interval(ref, lo, hi)
        need = int(st.get("max_alignment_span", 0))
        if need <= slack:
            return st, slack
        slack = (need + 16383) // 16384 * 16384
    raise Unsupported("--fix-mate-overlaps: alignments of the run span more than %d positions; use sbx-depth on one GPU" % slack)


def owner_rank_of_region(plan, ref_lengths, ref, start):
    """Rank that reports on a region starting at `start` of contig `ref`: the owner of that position; regions that start at or
    beyond the end of their contig belong to the owner of the contig's last position, regions of zero-length contigs to
    rank 0 (the single-GPU CLI prints a row for them as well)."""
    L = ref_lengths[ref]
    if L <= 0:
        return 0
    p = min(start, L - 1)
    for rank, ivs in enumerate(plan):
        for r, b, e in ivs:
            if r == ref and b <= p < e:
                return rank
    return 0


def base_header(a):
    return ("REF\tPOS\tCOV\tA\tC\tG\tT\tDEL\tREFSKIP" + ("" if a.combined else "\tSAMPLE") + ("\tFLAG" if a.annotate else "") + "\n").encode()


def format_counter_rows(name, beg, cnt, samples, combined, annotate, lo, hi):
    """Rows of `depth base` (writeColumn, depth.d:534-555) for positions beg.. from a counter array [n][S][7] held on the
    host -- the all-reduce form has no resident tile set to format on the device.  Positions without coverage print nothing
    (min coverage >= 1 in this form)."""
    n, S = cnt.shape[0], cnt.shape[1]
    cov = cnt.sum(axis=2, dtype=np.uint64)                       # [n][S]
    ok = (cov >= lo) & (cov <= hi)
    any_col = cnt.reshape(n, -1).any(axis=1)
    out = []
    for i in np.flatnonzero(any_col):
        for s in range(S):
            if not ok[i, s] and not annotate:
                break                                             # the reference's loop returns at the first failing sample
            c = cnt[i, s]
            row = "%s\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d" % (name, beg + i, cov[i, s], c[0], c[1], c[2], c[3], c[5], c[6])
            if not combined:
                row += "\t" + samples[s]
            if annotate:
                row += "\ty" if ok[i, s] else "\tn"
            out.append(row + "\n")
    return "".join(out).encode()


def allreduce_contig_counters(d, dist, ref, length, S, piece, on_piece, dev):
    """Sum the per-position counters of the ranks' owned runs over contig `ref`: piece by piece, device to device into a
    tensor (sbx_depth_base_tile_device), all-reduce (RCCL when the tensor is on the GPU), `on_piece(beg, tensor)`.
    Returns (bytes reduced per rank, seconds inside export + all_reduce)."""
    import time
    import torch
    total_b, secs = 0, 0.0
    for a0 in range(0, length, piece):
        b0 = min(length, a0 + piece)
        t0 = time.perf_counter()
        buf = torch.empty((b0 - a0, S, 7), dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        d.base_counters_to_device(ref, a0, b0, buf.data_ptr())
        if dist is not None:
            if dev.type == "cuda":
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize()
            else:                                       # gloo (CPU tests, ranks sharing a GPU): through host memory
                h = buf.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                buf = h
        secs += time.perf_counter() - t0
        total_b += buf.numel() * 4
        if on_piece is not None:
            on_piece(a0, buf)
    return total_b, secs


def allreduce_base_counters(d, dist, world, rank, red_dev, windows=(), steps=3, piece=1 << 23):
    """bench.py's `allreduce_option`: the resident BAM of `d` (configs[1]: one contig) with the reads partitioned between the
    ranks by start position and the per-position counters summed by an all-reduce.  Returns timings, the bytes every rank
    puts through the collective and, for each (ref, beg, end) of `windows`, the reduced counters (numpy) for the caller to
    compare with an independent result."""
    import time
    import torch
    ref_lengths = d.ref_lengths
    plan = plan_position_shards(ref_lengths, world, align=1024)
    mine = plan[rank]
    S = d.n_samples_eff
    picked = {}

    def grab(ref):
        def on_piece(a0, buf):
            for (r, a, b) in windows:
                lo, hi = max(a, a0), min(b, a0 + buf.shape[0])
                if r == ref and lo < hi:
                    w = picked.setdefault((r, a, b), np.zeros((b - a, S, 7), dtype=np.uint32))
                    w[lo - a:hi - a] = buf[lo - a0:hi - a0].cpu().numpy().astype(np.uint32)
        return on_piece

    t_run = t_red = 0.0
    nbytes = 0
    for step in range(steps + 1):                 # one untimed pass first
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        per_ref = {}            # (one contig per config here; a rank's slice of a contig is one interval)
        for ref, beg, end in mine:
            per_ref.setdefault(ref, []).append((beg, end))
        run_s = red_s = 0.0
        nb = 0
        for ref in range(len(ref_lengths)):
            if ref_lengths[ref] <= 0:
                continue
            ivs = per_ref.get(ref, [])
            if len(ivs) > 1:
                raise Unsupported("allreduce option: a rank's slice of a contig must be one interval")
            if not ivs:
                # this rank owns no read of the contig: it still takes part in the collective, with zeros
                for a0 in range(0, ref_lengths[ref], piece):
                    b0 = min(ref_lengths[ref], a0 + piece)
                    z = torch.zeros((b0 - a0, S, 7), dtype=torch.int32, device=red_dev)
                    tq = time.perf_counter()
                    if dist is not None:
                        dist.all_reduce(z, op=dist.ReduceOp.SUM)
                    red_s += time.perf_counter() - tq
                    nb += z.numel() * 4
                    if step == steps:
                        grab(ref)(a0, z)
                continue
            t1 = time.perf_counter()
            d.run_interval_owned(ref, ivs[0][0], ivs[0][1])
            run_s += time.perf_counter() - t1
            b, s = allreduce_contig_counters(d, dist, ref, ref_lengths[ref], S, piece, grab(ref) if step == steps else None, red_dev)
            nb += b
            red_s += s
        if dist is not None:
            dist.barrier()
        if step > 0:
            t_run += run_s
            t_red += red_s
            nbytes = nb
    t = torch.tensor([t_run / steps, t_red / steps], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms_run, ms_red = float(t[0].item()) * 1e3, float(t[1].item()) * 1e3
    # the collective's time means something over RCCL (device tensors over xGMI) only: through gloo the counters travel through host
    # memory and a loopback socket -- the results are still checked, the time is not reported
    on_rccl = str(red_dev).startswith("cuda")
    return {"ms_owned_run": round(ms_run, 3), "ms_export_and_allreduce": round(ms_red, 3) if on_rccl else None,
            "ms_per_step": round(ms_run + ms_red, 3) if on_rccl else None,
            "allreduce_bytes_per_rank": int(nbytes),
            "allreduce_GBps_per_rank": round(nbytes / max(1e-9, ms_red * 1e-3) / 1e9, 2) if on_rccl else None,
            "collective_backend": "rccl" if on_rccl else "gloo (host memory: not timed)",
            "steps": steps, "piece_positions": piece, "windows": picked,
            "what": "reads partitioned between the ranks by start position (sbx_run_interval_owned), every rank counts all positions its "
                    "reads cover, int32 all-reduce (SUM) of the u32[L][S][7] counter array in pieces, device to device"}


def run_sharded(a, dist, device, out, out_path=None):
    """The whole job of one rank; rank 0 writes to `out` (or, for base mode with -o, every rank writes its own byte range of
    `out_path`).  Raises Unsupported for option sets this driver rejects."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    thresholds = list(a.cov_threshold)
    nt = len(thresholds)
    mode = a.mode
    # option checks first: nothing below may fail on one rank only, or divide by a zero window size
    if mode == "window":
        if a.window_size <= 0:
            raise Unsupported("positive window size must be specified")
        if a.overlap:
            raise Unsupported("window --overlap > 0 is not supported by the sharded driver (use sbx-depth on one GPU)")
    min_cov = a.min_coverage if a.min_coverage is not None else (1.0 if mode == "base" else 0.0)
    if mode == "base" and a.regions and min_cov <= 0:
        raise Unsupported("base -L with --min-coverage=0 is not supported by the sharded driver (use sbx-depth on one GPU)")
    reduce_mode = getattr(a, "reduce", "none")
    if reduce_mode == "allreduce":
        if mode != "base" or a.fix_mate_overlaps or a.regions or min_cov < 1 or a.min_base_quality:
            raise Unsupported("--reduce allreduce: base mode without -m, -L, -q and with --min-coverage >= 1 only")
    with Depth(a.bam, device=device) as d:
        if a.filter is not None:
            d.set_filter(a.filter)
        d.set_params(mode={"base": SBX_MODE_BASE, "region": SBX_MODE_REGION, "window": SBX_MODE_WINDOW}[mode], min_bq=a.min_base_quality,
                     fix_mate_overlaps=a.fix_mate_overlaps, combined=a.combined, window=a.window_size, thresholds=thresholds)
        samples = ["*"] if a.combined else d.sample_names
        S = len(samples)
        merged = raw = lines = None
        if a.regions:
            merged, raw, lines = d.parse_regions(a.regions)
            if not merged:
                raise Unsupported("Enforcement failed")
        align = a.window_size if mode == "window" else 1024
        plan = plan_position_shards(d.ref_lengths, world, align=align)
        mine = plan[rank]
        n_ref = len(d.ref_lengths)

        # ---- header line (rank 0) ----
        if mode == "base":
            header = base_header(a)
        else:
            n_before = 3 if mode == "window" else len(lines[0].split()) if lines else 3
            head = "# " + "".join(c + "\t" for c in ["chrom", "chromStart", "chromEnd"][:min(3, n_before)]) + "".join("F%d\t" % k for k in range(3, n_before))
            head += "readCount\tmeanCoverage" + "".join("\tpercentage%d" % t for t in thresholds)
            head += ("" if a.combined else "\tsampleName") + ("\tmeanCovWithinBounds" if a.annotate else "") + "\n"
            header = head.encode()

        if mode == "base" and reduce_mode == "allreduce":
            import torch
            from .shard import _dev
            dev = _dev(dist)
            if rank == 0:
                out.write(header)
            lo = int(np.ceil(min_cov))
            hi = int(min(a.max_coverage, 1.8e19))
            own = {}
            for ref, beg, end in mine:
                own.setdefault(ref, []).append((beg, end))
            for ref in range(n_ref):
                L = d.ref_lengths[ref]
                if L <= 0:
                    continue
                ivs = own.get(ref, [])
                if len(ivs) > 1:
                    raise Unsupported("--reduce allreduce: a rank's slice of a contig must be one interval")
                piece = 1 << 20
                # alignments may hang over the contig end: the reduced range includes the spare tiles the ranks laid out for them
                top = L
                if ivs:
                    d.run_interval_owned(ref, ivs[0][0], ivs[0][1])
                    top = max(L, d.active_end(ref))
                if dist is not None:
                    tt = torch.tensor([top], dtype=torch.int64, device=dev)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    top = int(tt.item())
                if ivs:

                    def emit(a0, buf):
                        if rank == 0:
                            out.write(format_counter_rows(d.ref_names[ref], a0, buf.cpu().numpy().astype(np.uint32), samples, a.combined,
                                                          a.annotate, lo, hi))
                    allreduce_contig_counters(d, dist, ref, top, S, piece, emit, dev)
                else:
                    for a0 in range(0, top, piece):
                        b0 = min(top, a0 + piece)
                        z = torch.zeros((b0 - a0, S, 7), dtype=torch.int32, device=dev)
                        if dist is not None:
                            dist.all_reduce(z, op=dist.ReduceOp.SUM)
                        if rank == 0:
                            out.write(format_counter_rows(d.ref_names[ref], a0, z.cpu().numpy().astype(np.uint32), samples, a.combined,
                                                          a.annotate, lo, hi))
            return

        if mode == "base":
            if merged is not None:
                d.set_regions(merged)

            def spans_of(ref, beg, end):
                """[(ref, a, b, min_cov for the piece)] to print for the interval whose run is resident."""
                if merged is not None:
                    return [(ref, max(s, beg), min(e, end), min_cov) for r, s, e in merged if r == ref and s < end and e > beg]
                sp = [(ref, beg, end, min_cov)]
                # alignments hanging over the end of the contig have columns beyond it: the owner of the contig's last position prints them
                if end == d.ref_lengths[ref]:
                    mc = max(min_cov, 1e-9) if min_cov <= 0 else min_cov
                    top = max(end, d.active_end(ref))
                    if top > end and d.measure_base_rows(ref, end, top, mc, a.max_coverage, a.annotate):
                        if min_cov <= 0:
                            raise Unsupported("alignments hang over the end of contig %s: not supported with --min-coverage=0 by the sharded driver" % d.ref_names[ref])
                        sp.append((ref, end, top, mc))
                return sp

            zero_fill = merged is None and min_cov <= 0
            need_sizes = out_path is not None
            resident = None
            sizes = {}            # interval index -> [bytes of its spans]
            has_cols = [0] * n_ref
            if zero_fill or need_sizes:
                # measuring pass: which contigs have pileup columns (zero rows under -c 0: push / close, depth.d:567-606) and how many
                # bytes every span prints (the device's measuring kernel; no text is produced)
                for k, (ref, beg, end) in enumerate(mine):
                    d.run_interval(ref, beg, end)
                    resident = k
                    if zero_fill and first_column_in(d, ref, beg, end) is not None:
                        has_cols[ref] = 1
                    if need_sizes:
                        sizes[k] = [d.measure_base_rows(r, s, e, mc, a.max_coverage, a.annotate) for (r, s, e, mc) in spans_of(ref, beg, end)]
            keep = lambda r: True
            if zero_fill:
                # a contig WITHOUT columns is zero-filled only before the first and after the last contig that has some
                has_cols = _max_flags(dist, has_cols)
                with_cols = [r for r in range(n_ref) if has_cols[r]]
                keep = lambda r: bool(has_cols[r]) or not with_cols or r < with_cols[0] or r > with_cols[-1]
            if need_sizes:
                my_bytes = sum(sum(v) for k, v in sizes.items() if keep(mine[k][0]))
                offset, total = exclusive_offset(my_bytes, dist)
                offset += len(header)
                fd = os.open(out_path, os.O_WRONLY | os.O_CREAT, 0o644)
                if rank == 0:
                    os.ftruncate(fd, 0)
                    os.pwrite(fd, header, 0)
                if dist is not None:
                    dist.barrier()           # nobody writes before the truncation
                pos = [offset]

                def write(chunk):
                    os.pwrite(fd, chunk, pos[0])
                    pos[0] += len(chunk)
            else:
                funnel = TextFunnel(dist, text_group(dist), out.write if rank == 0 else None)
                if rank == 0:
                    out.write(header)
                funnel.begin()
                write = funnel.write
            for k, (ref, beg, end) in enumerate(mine):
                if not keep(ref):
                    continue
                if resident != k:
                    d.run_interval(ref, beg, end)
                    resident = k
                for (r, s, e, mc) in spans_of(ref, beg, end):
                    d.stream_base_rows(r, s, e, write, mc, a.max_coverage, a.annotate)
            if need_sizes:
                if pos[0] != offset + my_bytes:
                    raise RuntimeError("internal: measured %d bytes of text, wrote %d" % (my_bytes, pos[0] - offset))
                os.close(fd)
            else:
                funnel.end()
            return

        if rank == 0:
            out.write(header)

        if mode == "region":
            # Every rank reports on the raw regions whose first position it owns.  Reads are selected against ALL merged
            # regions (a mate that reaches the pileup through a neighbour's region must still pair, depth.d:717-758), but
            # fetched only for the hull of the owned regions of a contig, widened by the mate slack on each side.
            d.set_regions(merged)
            ids_all, rows = [], []
            mine_ids = {}
            for i, g in enumerate(raw):
                if owner_rank_of_region(plan, d.ref_lengths, g[0], g[1]) == rank:
                    mine_ids.setdefault(g[0], []).append(i)
            for ref in sorted(mine_ids):
                ids = mine_ids[ref]
                lo = min(raw[i][1] for i in ids)
                hi = max(raw[i][2] for i in ids)
                run_with_mate_slack(d, ref, lo, hi, a.fix_mate_overlaps, left_only=False)
                nr, nb, cov, seen = d.region_stats([raw[i] for i in ids], nt)
                for j, i in enumerate(ids):
                    v = [int(seen[j])]
                    for s in range(S):
                        v += [int(nr[j][s]), int(nb[j][s])] + [int(x) for x in cov[j][s][:nt]]
                    ids_all.append(i)
                    rows.append(v)
            idx, vals = gather_rows(np.asarray(ids_all, dtype=np.int64), np.asarray(rows, dtype=np.int64).reshape(len(ids_all), 1 + S * (2 + nt)), dist)
            if rank == 0 and len(idx) and bool((vals[:, 0] != 0).any()):      # rows only if some column fell inside some region
                for i, v in zip(idx.tolist(), vals.tolist()):
                    r, s0, e0 = raw[i]
                    for s in range(S):
                        o = 1 + s * (2 + nt)
                        row = region_row(lines[i].rstrip() + "\t", e0 - s0, v[o], v[o + 1], v[o + 2:o + 2 + nt], thresholds, samples[s], a.combined,
                                         a.annotate, min_cov, a.max_coverage)
                        if row:
                            out.write(row.encode())
            return

        # ---- window ----
        w = a.window_size
        first_col = 1 << 62         # (ref << 32 | pos) of the first pileup column of the run
        has_cols = [0] * n_ref
        ids, rows = [], []
        win_base = np.cumsum([0] + [L // w for L in d.ref_lengths])
        # With --fix-mate-overlaps a read that lies past the overlap with its mate is counted differently from an unpaired one
        # (status `past`, depth.d:717-845), so the mate must be in the run even when it ends before the slice: run_with_mate_slack
        for ref, beg, end in mine:
            run_with_mate_slack(d, ref, beg, end, a.fix_mate_overlaps)
            fc = first_column_in(d, ref, beg, end)
            if fc is not None:
                has_cols[ref] = 1
                first_col = min(first_col, (ref << 32) | fc)
            k0 = beg // w
            k1 = d.ref_lengths[ref] // w if end >= d.ref_lengths[ref] else end // w      # only full windows are printed
            if k1 > k0:
                nr, nb, cov = d.window_stats(ref, k0, k1 - k0, nt)
                for k in range(k0, k1):
                    ids.append(int(win_base[ref]) + k)
                    v = [ref, k]
                    for s in range(S):
                        v += [int(nr[k - k0][s]), int(nb[k - k0][s])] + [int(x) for x in cov[k - k0][s][:nt]]
                    rows.append(v)
        first_col = _min_over_ranks(dist, first_col)
        has_cols = _max_flags(dist, has_cols)
        with_cols = [r for r in range(n_ref) if has_cols[r]]
        if with_cols and any(d.ref_lengths[r] // w for r in range(with_cols[-1] + 1, n_ref)):
            raise Unsupported("window mode: the contigs after %s have no reads; the reference continues the previous contig's window "
                              "coordinates there -- use sbx-depth on one GPU" % d.ref_names[with_cols[-1]])
        idx, vals = gather_rows(np.asarray(ids, dtype=np.int64), np.asarray(rows, dtype=np.int64).reshape(len(ids), 2 + S * (2 + nt)), dist)
        if rank == 0 and first_col != (1 << 62):
            f_ref, f_pos = first_col >> 32, first_col & 0xFFFFFFFF
            for v in vals.tolist():
                r, k = v[0], v[1]
                if r < f_ref or (r == f_ref and (k + 1) * w <= f_pos):
                    continue                        # windows finished before the first column of the run print nothing
                prefix = "%s\t%d\t%d\t" % (d.ref_names[r], k * w, (k + 1) * w)
                for s in range(S):
                    o = 2 + s * (2 + nt)
                    row = region_row(prefix, w, v[o], v[o + 1], v[o + 2:o + 2 + nt], thresholds, samples[s], a.combined, a.annotate, min_cov,
                                     a.max_coverage)
                    if row:
                        out.write(row.encode())


def main(argv=None):
    ap = argparse.ArgumentParser(prog="sambamba_amd.dist_depth")
    ap.add_argument("mode", choices=["base", "region", "window"])
    ap.add_argument("bam")
    ap.add_argument("-L", "--regions")
    ap.add_argument("-w", "--window-size", type=int, default=0)
    ap.add_argument("--overlap", type=int, default=0)
    ap.add_argument("-T", "--cov-threshold", type=int, action="append", default=[])
    ap.add_argument("-q", "--min-base-quality", type=int, default=0)
    ap.add_argument("-F", "--filter")
    ap.add_argument("-c", "--min-coverage", type=float, default=None)
    ap.add_argument("-C", "--max-coverage", type=float, default=1e50)
    ap.add_argument("-a", "--annotate", action="store_true")
    ap.add_argument("-m", "--fix-mate-overlaps", action="store_true")
    ap.add_argument("-o", "--output-filename")
    ap.add_argument("--combined", action="store_true")
    ap.add_argument("--reduce", choices=["none", "allreduce"], default="none",
                    help="base mode: `allreduce` partitions the READS between the ranks and sums per-position counters with an all-reduce")
    a = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    # SBX_DIST_FORCE_GROUP=1: initialise the process group even for one rank (the RCCL path at world size 1, tests)
    if world > 1 or os.environ.get("SBX_DIST_FORCE_GROUP"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29400")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("SBX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
        dd = dist
    else:
        dd = None
    rank = dist.get_rank() if dd is not None else 0
    if a.mode == "region" and not a.regions:
        sys.stderr.write("BED file or a region must be provided in region mode\n")
        sys.exit(1)
    out = None
    per_rank_file = a.mode == "base" and a.output_filename and a.reduce == "none"
    if rank == 0 and not per_rank_file:
        out = open(a.output_filename, "wb") if a.output_filename else sys.stdout.buffer
    rc = 0
    try:
        run_sharded(a, dd, device, out, out_path=a.output_filename if per_rank_file else None)
    except Exception as e:      # the failing rank must not leave the others waiting in a collective: tear the group down hard
        sys.stderr.write("sambamba-depth: %s\n" % (e.msg if hasattr(e, "msg") else e))
        sys.stderr.flush()
        if out is not None:
            out.flush()
        os._exit(1)
    if out is not None:
        out.flush()
    if dd is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
