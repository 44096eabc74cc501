"""`depth region|window` of ONE BAM sharded over the ranks of a torch.distributed job (BASELINE config 4).

Every rank opens the BAM on its own GPU, takes a run of consecutive contigs (shard.plan_contig_shards, balanced
by length), runs the device pipeline for those contigs only (sbx_run_batch -- the engine then inflates just the
BGZF blocks the BAI lists for them) and computes the statistics of the regions / windows lying on them.  Shards
own disjoint outputs, so the only exchange is an all-gather of the small per-region rows; per-position data
never leaves a GPU.  Rank 0 prints exactly what `sbx-depth` prints on one GPU.

    python -m torch.distributed.run --nproc-per-node N -m sambamba_amd.dist_depth region -L x.bed -T 10 in.bam

Window mode here covers disjoint windows (no --overlap) of genomes whose read-less contigs, if any, are followed by a
contig with reads; the ring quirks the single-GPU CLI reproduces for the other cases are not
replicated in this driver.
"""
import argparse
import os
import sys

import numpy as np

from . import Depth, SBX_MODE_REGION, SBX_MODE_WINDOW
from .shard import gather_region_stats, plan_contig_shards


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


def read_bed(path, depth):
    """(ref_id, start, end, line) of every BED line naming a contig of the BAM, in file order."""
    out = []
    with open(path) as fh:
        for line in fh:
            f = line.split()
            if len(f) < 3 or f[0] not in depth.ref_names:
                continue
            out.append((depth.ref_names.index(f[0]), int(f[1]), int(f[2]), line.rstrip()))
    return out


def sharded_stats(bam, mode, raw=None, window=0, thresholds=(), min_bq=0, combined=False, fix_mate=False, filt=None,
                  dist=None, device=0):
    """Rows (index, payload) of every region (mode 'region') or window (mode 'window'), merged over the ranks.

    region payload: (n_reads[S], n_bases[S], cov[S][n_thr], seen); window payload: (ref, k, n_reads[S], n_bases[S], cov[S][n_thr])."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    rows = []
    with Depth(bam, device=device) as d:
        if filt is not None:
            d.set_filter(filt)
        d.set_params(mode=SBX_MODE_REGION if mode == "region" else SBX_MODE_WINDOW, min_bq=min_bq, fix_mate_overlaps=fix_mate,
                     combined=combined, window=window, thresholds=thresholds)
        first, last = plan_contig_shards(d.ref_lengths, world)[rank]
        if mode == "region":
            merged = merge_regions([(r, s, e) for (r, s, e, _) in raw])
            d.set_regions(merged)
        info = {"ref_names": d.ref_names, "ref_lengths": d.ref_lengths, "samples": d.sample_names, "first_column": None}
        if last > first:
            # (a shard larger than the device would loop over d.plan_batches() here; one batch per shard otherwise)
            d.run_batch(first, last - first)
            nt = len(thresholds)
            if mode == "region":
                ids = [i for i, g in enumerate(raw) if first <= g[0] < last]
                if ids:
                    nr, nb, cov, seen = d.region_stats([raw[i][:3] for i in ids], nt)
                    rows = [(i, (nr[j], nb[j], cov[j], int(seen[j]))) for j, i in enumerate(ids)]
            else:
                base = 0
                for r in range(len(d.ref_lengths)):
                    n_full = d.ref_lengths[r] // window
                    if first <= r < last:
                        if info["first_column"] is None:
                            info["first_column"] = first_column(d, r)
                        if n_full:
                            nr, nb, cov = d.window_stats(r, 0, n_full, nt)
                            rows += [(base + k, (r, k, nr[k], nb[k], cov[k])) for k in range(n_full)]
                    base += n_full
    firsts = gather_region_stats([(rank, info["first_column"])], dist)
    info["first_column"] = next((fc for _, fc in firsts if fc is not None), None)
    return gather_region_stats(rows, dist), info


def merge_regions(regs):
    out = []
    for r, s, e in sorted(regs):
        if out and out[-1][0] == r and out[-1][2] >= s:
            out[-1] = (r, out[-1][1], max(out[-1][2], e))
        else:
            out.append((r, s, e))
    return out


def first_column(d, ref):
    """(ref, pos) of the first pileup column on contig `ref`, or None."""
    L = d.ref_lengths[ref]
    step = 1 << 20
    for b in range(0, L, step):
        _, cov = d.base_counters(ref, b, min(L, b + step), with_covered=True)
        nz = np.flatnonzero(cov)
        if len(nz):
            return (ref, b + int(nz[0]))
    return None


def main(argv=None):
    ap = argparse.ArgumentParser(prog="sambamba_amd.dist_depth")
    ap.add_argument("mode", choices=["region", "window"])
    ap.add_argument("bam")
    ap.add_argument("-L", "--regions")
    ap.add_argument("-w", "--window-size", type=int, default=0)
    ap.add_argument("-T", "--cov-threshold", type=int, action="append", default=[])
    ap.add_argument("-q", "--min-base-quality", type=int, default=0)
    ap.add_argument("-F", "--filter")
    ap.add_argument("-c", "--min-coverage", type=float, default=None)
    ap.add_argument("-C", "--max-coverage", type=float, default=float("inf"))
    ap.add_argument("-a", "--annotate", action="store_true")
    ap.add_argument("-m", "--fix-mate-overlaps", action="store_true")
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
    min_cov = a.min_coverage if a.min_coverage is not None else 0.0     # region/window print everything by default
    raw = None
    if a.mode == "region":
        with Depth(a.bam, device=device) as d0:
            raw = read_bed(a.regions, d0)
    rows, info = sharded_stats(a.bam, a.mode, raw=raw, window=a.window_size, thresholds=a.cov_threshold, min_bq=a.min_base_quality,
                               combined=a.combined, fix_mate=a.fix_mate_overlaps, filt=a.filter, dist=dd, device=device)
    if dd is None or dist.get_rank() == 0:
        out = sys.stdout
        samples = ["*"] if a.combined else info["samples"]
        hdr_cols = ["chrom", "chromStart", "chromEnd"]
        n_before = 3 if a.mode == "window" else len(raw[0][3].split()) if raw else 3
        head = "# " + "".join(c + "\t" for c in hdr_cols[:min(3, n_before)]) + "".join("F%d\t" % k for k in range(3, n_before))
        head += "readCount\tmeanCoverage" + "".join("\tpercentage%d" % t for t in a.cov_threshold)
        head += ("" if a.combined else "\tsampleName") + ("\tmeanCovWithinBounds" if a.annotate else "") + "\n"
        out.write(head)
        if a.mode == "region":
            if any(p[3] for _, p in rows):          # rows only if some column fell inside some region
                for i, (nr, nb, cov, _seen) in rows:
                    r, s, e, line = raw[i]
                    for si, sm in enumerate(samples):
                        row = region_row(line + "\t", e - s, nr[si], nb[si], cov[si], a.cov_threshold, sm, a.combined, a.annotate,
                                         min_cov, a.max_coverage)
                        if row:
                            out.write(row)
        else:
            fc = info["first_column"]
            w = a.window_size
            for _, (r, k, nr, nb, cov) in rows:
                if fc is None or r < fc[0] or (r == fc[0] and (k + 1) * w <= fc[1]):
                    continue                        # windows finished before the first column of the run print nothing
                prefix = "%s\t%d\t%d\t" % (info["ref_names"][r], k * w, (k + 1) * w)
                for si, sm in enumerate(samples):
                    row = region_row(prefix, w, nr[si], nb[si], cov[si], a.cov_threshold, sm, a.combined, a.annotate, min_cov,
                                     a.max_coverage)
                    if row:
                        out.write(row)
        out.flush()
    if dd is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
