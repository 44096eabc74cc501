"""`depth base|region|window` of ONE BAM sharded over the ranks of a torch.distributed job.

Every rank opens the BAM on its own GPU and takes a contiguous slice of the concatenated reference
(shard.plan_position_shards: whole contigs and, where a contig is cut, position intervals inside it -- so ONE long
contig shards as well as a whole genome does).  For every interval of its slice it runs the device pipeline over the
reads that overlap the interval (sbx_run_interval: only the BGZF blocks the BAI lists for it are uploaded and inflated)
and produces its share of the output:

  base    the text of the positions it owns, formatted on the device; rank 0 concatenates the ranks' text in rank order
          (point-to-point sends; with -o every rank could equally write its own byte range)
  window  the statistics of the windows that start in its slice (cuts are aligned to the window size, so no window is split)
  region  the statistics of the BED regions whose first position it owns (a region is never split)

Shards own disjoint outputs, so the only exchange is the gather of small stat rows (as tensors: RCCL all_gather on GPUs)
or of finished text; per-position counters never leave a GPU.  Rank 0 prints byte for byte what `sbx-depth` prints on one
GPU for the supported option set (tests/test_gpu_dist.py).

    python -m torch.distributed.run --nproc-per-node N -m sambamba_amd.dist_depth base in.bam
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
from .shard import gather_rows, plan_position_shards, send_text_to_rank0


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


MATE_SLACK = 16384     # positions fetched left of a slice so that overlapping mates of its reads are in the run


class Unsupported(RuntimeError):
    pass


def _all_ok(dist, ok):
    """True iff every rank is fine -- so that one failing rank cannot leave the others waiting in a collective."""
    if dist is None:
        return ok
    import torch
    from .shard import _dev
    t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=_dev(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


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
            _, cov = d.base_counters(ref, x, min(e, x + step), with_covered=True)
            nz = np.flatnonzero(cov)
            if len(nz):
                return x + int(nz[0])
        pos = e
    return None


def run_sharded(a, dist, device, out):
    """The whole job of one rank; rank 0 writes to `out`.  Raises Unsupported for option sets this driver rejects."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    thresholds = list(a.cov_threshold)
    nt = len(thresholds)
    mode = a.mode
    if mode == "window" and a.overlap:
        raise Unsupported("window --overlap > 0 is not supported by the sharded driver (use sbx-depth on one GPU)")
    min_cov = a.min_coverage if a.min_coverage is not None else (1.0 if mode == "base" else 0.0)
    if mode == "base" and a.regions and min_cov <= 0:
        raise Unsupported("base -L with --min-coverage=0 is not supported by the sharded driver (use sbx-depth on one GPU)")
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
        if rank == 0:
            if mode == "base":
                out.write(("REF\tPOS\tCOV\tA\tC\tG\tT\tDEL\tREFSKIP" + ("" if a.combined else "\tSAMPLE") + ("\tFLAG" if a.annotate else "") + "\n").encode())
            else:
                n_before = 3 if mode == "window" else len(lines[0].split()) if lines else 3
                head = "# " + "".join(c + "\t" for c in ["chrom", "chromStart", "chromEnd"][:min(3, n_before)]) + "".join("F%d\t" % k for k in range(3, n_before))
                head += "readCount\tmeanCoverage" + "".join("\tpercentage%d" % t for t in thresholds)
                head += ("" if a.combined else "\tsampleName") + ("\tmeanCovWithinBounds" if a.annotate else "") + "\n"
                out.write(head.encode())

        if mode == "base":
            # which contigs have pileup columns decides where zero rows go under -c 0 (push / close, depth.d:567-606)
            chunks = []
            has_cols = [0] * n_ref
            pieces = []         # (ref, beg, end, text or None)
            if merged is not None:
                d.set_regions(merged)
            for ref, beg, end in mine:
                d.run_interval(ref, beg, end)
                if merged is not None:
                    for r, s, e in merged:
                        if r == ref and s < end and e > beg:
                            pieces.append((ref, max(s, beg), min(e, end), d.format_base_rows(ref, max(s, beg), min(e, end), min_cov, a.max_coverage, a.annotate)))
                    continue
                if first_column_in(d, ref, beg, end) is not None:
                    has_cols[ref] = 1
                pieces.append((ref, beg, end, d.format_base_rows(ref, beg, end, min_cov, a.max_coverage, a.annotate)))
                # alignments hanging over the end of the contig have columns beyond it: the owner of the contig's last position prints them
                if end == d.ref_lengths[ref]:
                    over = d.format_base_rows(ref, end, end + 1024, max(min_cov, 1e-9) if min_cov <= 0 else min_cov, a.max_coverage, a.annotate)
                    if over:
                        if min_cov <= 0:
                            raise Unsupported("alignments hang over the end of contig %s: not supported with --min-coverage=0 by the sharded driver" % d.ref_names[ref])
                        pieces.append((ref, end, end + 1024, over))
            if merged is None and min_cov <= 0:
                # a contig WITHOUT columns is zero-filled only before the first and after the last contig that has some
                has_cols = _max_flags(dist, has_cols)
                with_cols = [r for r in range(n_ref) if has_cols[r]]
                keep = lambda r: has_cols[r] or not with_cols or r < with_cols[0] or r > with_cols[-1]
                pieces = [p for p in pieces if keep(p[0])]
            chunks = [p[3] for p in pieces if p[3]]
            send_text_to_rank0(chunks, dist, out.write if out is not None else None)
            return

        if mode == "region":
            # Every rank reports on the raw regions whose first position it owns.  Reads are selected against ALL merged
            # regions (a mate that reaches the pileup through a neighbour's region must still pair, depth.d:717-758), but
            # fetched only for the hull of the owned regions of a contig, widened by one linear-index window on each side.
            d.set_regions(merged)
            ids_all, rows = [], []
            for ref, beg, end in mine:
                ids = [i for i, g in enumerate(raw) if g[0] == ref and beg <= g[1] < end]
                if not ids:
                    continue
                lo = max(0, min(raw[i][1] for i in ids) - 16384)
                hi = max(raw[i][2] for i in ids) + 16384
                d.run_interval(ref, lo, hi)
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
        if w <= 0:
            raise Unsupported("positive window size must be specified")
        first_col = 1 << 62         # (ref << 32 | pos) of the first pileup column of the run
        has_cols = [0] * n_ref
        ids, rows = [], []
        win_base = np.cumsum([0] + [L // w for L in d.ref_lengths])
        # With --fix-mate-overlaps a read that lies past the overlap with its mate is counted differently from an unpaired one
        # (status `past`, depth.d:717-845), so the mate must be in the run even when it ends before the slice: fetch one
        # linear-index window more on the left (mates overlap, so the partner starts within one read span of the cut).
        slack = MATE_SLACK if a.fix_mate_overlaps else 0
        for ref, beg, end in mine:
            d.run_interval(ref, max(0, beg - slack), end)
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
    a = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = local % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SBX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    dd = dist if world > 1 else None
    rank = dist.get_rank() if dd is not None else 0
    if a.mode == "region" and not a.regions:
        sys.stderr.write("BED file or a region must be provided in region mode\n")
        sys.exit(1)
    out = None
    if rank == 0:
        out = open(a.output_filename, "wb") if a.output_filename else sys.stdout.buffer
    rc = 0
    try:
        run_sharded(a, dd, device, out)
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
