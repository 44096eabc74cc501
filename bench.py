#!/usr/bin/env python3
"""bench.py -- the `sambamba depth` hot path on MI355X: BGZF inflate -> BAM record index ->
CIGAR-walk per-position coverage counters, all on the device (libsbx_depth.so).

Contract (driver): python bench.py --gpus N --steps K --warmup W   (N>1: launched under
torch.distributed.run, one rank per GPU).  One "step" = one full pass of the hot path over the
synthetic coordinate-sorted BAM of BASELINE.json configs[1] (chr1, L=248,956,422, 30x, 2x150 bp
paired reads), whose compressed bytes are already resident in HBM when the timed
region starts.

N == 1: the whole BAM on one GPU.  N > 1 (default --mode auto == shard): ONE BAM with N chr1-sized contigs (N x 49.8 M
reads, the per-GPU work of configs[1] kept fixed: "scaling": "weak") is sharded over the ranks by reference position
(sambamba_amd.shard.plan_position_shards -- with equal contigs every rank owns one): every rank fetches the reads
overlapping its slice through the BAI, keeps the BGZF blocks of its slice resident in its HBM, clips its contributions to
the slice and checks windows of its slice against the oracle.  The path has no data-path collective; torch.distributed
(RCCL) carries the barrier, the max-over-ranks time and the read totals.  Side fields of the same line: `strong_one_contig`
(the single chr1 BAM cut into N position slices: strong scaling, bounded below by one K1a residency)
and `allreduce_option` (reads partitioned by start position, per-position counters summed with an RCCL all-reduce: the
alternative north_star names, measured next to the collective-free form).  --mode strong makes the one-contig cut the
headline; --mode replicas runs the full chr1 BAM on every GPU (round 2's headline).

Other BASELINE configs (builder-run lines, committed under profiles/): --config 3 (window -w 1000 on the
25-contig genome, streamed in batches), --config 4 (region -L exome BED on the same BAM), --config 5
(base --fix-mate-overlaps -q20 on a 300x contig).  --scale shrinks contig lengths for development.

Prints ONE JSON line on rank 0 with the driver's fields plus `roofline` (dominant kernel,
algorithmic bytes / HIP-event kernel time), `cpu_baseline` (the CPU oracle -- a literal port
of the reference algorithm -- timed on a bounded sample of the same BAM on this box's host),
`parity_checked` (device results of THIS run compared with the oracle on sampled windows; a mismatch
makes the process exit non-zero) and `e2e` (the product CLI, file in page cache -> text to /dev/null).
"""
import argparse
import hashlib
import json
import os
import random
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHR1_LEN = 248_956_422
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# GRCh38 primary assembly: chr1-22, X, Y, M
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717,
          133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285,
          58617616, 64444167, 46709983, 50818468, 156040895, 57227415, 16569]
GRCH38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY", "chrM"]


_T0 = time.time()


def log(msg):
    """Progress of the harness on stderr (the JSON line on stdout stays alone)."""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("[bench %7.1fs] %s\n" % (time.time() - _T0, msg))
        sys.stderr.flush()


def ensure_built():
    from sambamba_amd import build as b
    b.build()
    for d in ("oracle", "tools"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, d)])


def tmp_dir():
    return "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"


def workload(args, copies=1):
    """(contigs string, coverage, seed, extra gen_bam options, description) of the selected BASELINE config.
    copies > 1 (configs 2 and 5, N GPUs): that many contigs of the config's length in ONE BAM -- the per-GPU work of the
    config kept fixed while the BAM is sharded over the ranks by contig."""
    sc = args.scale
    if args.config == 2:
        return ",".join("chr%d:%d" % (i + 1, args.length) for i in range(copies)), args.coverage, 0x5A4D0002, [], "configs[1]"
    if args.config in (3, 4):
        contigs = ",".join("%s:%d" % (n, max(2000, int(l * sc))) for n, l in zip(GRCH38_NAMES, GRCH38))
        return contigs, args.coverage, 0x5A4D0003, [], "configs[%d]" % (args.config - 1)
    if args.config == 5:
        L = max(2000, int(50_000_000 * sc))
        return ",".join("chr%d:%d" % (i + 1, L) for i in range(copies)), 300.0 if args.coverage == 30.0 else args.coverage, 0x5A4D0005, \
            ["--insert-mean", "250", "--insert-sd", "40", "--tie-free-overlaps"], "configs[4]"
    raise SystemExit("unknown --config")


def bam_path(contigs, coverage, seed, level, codec, extra):
    key = "%s_%g_%x_%d_%s_%s" % (contigs, coverage, seed, level, codec, " ".join(extra))
    return os.path.join(tmp_dir(), "sbx_bench_%s.bam" % hashlib.sha1(key.encode()).hexdigest()[:12])


def make_room(keep, need_bytes):
    """Drop cached bench BAMs of other workloads when the scratch directory is short of space."""
    import glob
    import shutil
    try:
        if shutil.disk_usage(tmp_dir()).free > need_bytes:
            return
    except OSError:
        return
    for f in glob.glob(os.path.join(tmp_dir(), "sbx_bench_*.bam")):
        if f not in keep:
            for g in (f, f + ".bai", f + ".json", f + ".exome.bed"):
                try:
                    os.remove(g)
                except OSError:
                    pass


def generate(path, contigs, coverage, seed, level, codec, extra):
    """Seeded synthetic BAM+BAI (tools/gen_bam.cpp); cached on tmpfs across invocations on the same box."""
    meta = path + ".json"
    if os.path.exists(path) and os.path.exists(path + ".bai") and os.path.exists(meta):
        return json.load(open(meta))
    t0 = time.time()
    tmp_out = path + ".tmp%d" % os.getpid()
    env = dict(os.environ, GEN_TIMING="1")
    r = subprocess.run([os.path.join(ROOT, "tools", "gen_bam"), "--out", tmp_out, "--contigs", contigs, "--coverage", str(coverage),
                        "--seed", hex(seed), "--level", str(level), "--codec", codec] + list(extra),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, check=True)
    info = json.loads(r.stdout.decode().strip().splitlines()[-1])
    info["gen_seconds"] = time.time() - t0
    info["gen_phases"] = r.stderr.decode().strip().splitlines()[-1:] or [""]
    os.replace(tmp_out + ".bai", path + ".bai")
    os.replace(tmp_out, path)
    json.dump(info, open(meta, "w"))
    return info


def exome_bed(path, names, lengths, seed=0x5A4D0004, n_intervals=200_000, scale=1.0):
    """Exome-like BED: intervals clustered into `genes` (8-12 exons a few kb apart), lengths
    lognormal with median 150 bp, sorted, non-overlapping, seed-fixed; total ~ 35 Mbp at scale 1."""
    if os.path.exists(path):
        return
    rng = random.Random(seed)
    total = sum(lengths)
    n_intervals = max(25, int(n_intervals * scale))
    rows = []
    for name, L in zip(names, lengths):
        want = max(1, int(n_intervals * L / total))
        got, ivs = 0, []
        while got < want:
            g0 = rng.randrange(0, max(1, L - 60_000))
            p = g0
            for _ in range(rng.randint(8, 12)):
                ln = max(30, min(5000, int(rng.lognormvariate(5.01, 0.6))))
                if p + ln >= L:
                    break
                ivs.append((p, p + ln))
                got += 1
                p += ln + rng.randint(200, 6000)
        ivs.sort()
        last = -1
        for a, b in ivs:
            if a <= last:
                continue
            rows.append("%s\t%d\t%d\n" % (name, a, b))
            last = b
    with open(path + ".tmp", "w") as fh:
        fh.writelines(rows)
    os.replace(path + ".tmp", path)


def cpu_baseline(bam, sample_reads, mode_args):
    """Reference-algorithm CPU stand-in: the oracle CLI on the first `sample_reads` records of the same
    BAM, output to /dev/null.  Structured like the reference: zlib inflate on a pool of worker threads
    with in-order delivery (`-t`), then the single-threaded sweep-line pileup + text formatting.  Timed
    three times -- sambamba's default (`-t 0`: everything on one thread, depth.d:1081,1154), with 16 inflate
    workers and with nproc - 1 of them (T = nproc) -- and the fastest one is reported (the serial
    sweep bounds what more threads can buy)."""
    exe = os.path.join(ROOT, "oracle", "depth_oracle")
    env = dict(os.environ, ORC_STATS="1")
    runs = []
    ncpu = os.cpu_count() or 2
    # T = nproc.  The reference spends -t on the inflate pool only: beyond a dozen workers nothing is gained, and a pool of
    # 255 workers that hand their blocks to ONE consumer is slower than the serial run (0.11 against 0.54 Mreads/s on the 256-core box:
    # profiles/round5) -- so the nproc run is timed on a tenth of the sample and listed, the two useful settings on all of it
    for workers in sorted(set([0, max(1, min(16, ncpu - 1)), max(1, ncpu - 1)])):
        t0 = time.time()
        n_reads = sample_reads if workers <= 16 else max(1, sample_reads // 10)
        r = subprocess.run([exe] + mode_args + ["-t", str(workers), "--max-reads", str(n_reads), bam],
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        dt = time.time() - t0
        if r.returncode != 0:
            continue
        secs, seen = dt, n_reads
        for line in r.stderr.decode().splitlines():
            if line.startswith("[oracle]"):
                parts = line.split()
                secs = float(parts[1])
                seen = int(parts[3])
        runs.append({"workers": workers, "mreads_per_s": seen / secs / 1e6, "secs": secs, "seen": seen})
    if not runs:
        return None
    best = max(runs, key=lambda x: x["mreads_per_s"])
    return {"value": round(best["mreads_per_s"], 4), "unit": "Mreads/s", "cores": best["workers"] + 1, "kind": "port",
            "sample": "first %d records of the same BAM, depth %s -> /dev/null, %.1f s wall" % (best["seen"], " ".join(mode_args), best["secs"]),
            "runs": [{"inflate_workers": x["workers"], "Mreads_per_s": round(x["mreads_per_s"], 4)} for x in runs],
            "threads_note": "sambamba depth default is 0 worker threads (fully serial); the sweep-line pileup and the "
                            "text output are single-threaded in the reference whatever -t says; nproc=%d" % (os.cpu_count() or 0)}


def parity_windows(d, bam, intervals, n_windows, seed, min_bq=0, fix_mate=False):
    """Compare the device counters of THIS run with the CPU oracle (which fetches the reads through the BAI) on sampled
    50 kb windows of the processed intervals [(ref, beg, end)]: the first and the last window of the work, one just
    past position 2^27 when there is one, the rest random.  Returns (checked, mismatching windows)."""
    import numpy as np
    from tests.util import oracle_base_counters
    rng = random.Random(seed)
    W = 50_000
    picks = []
    ref, beg, end = intervals[0]
    picks.append((ref, beg, min(end, beg + W)))
    ref, beg, end = intervals[-1]
    picks.append((ref, max(beg, end - W), end))
    for ref, beg, end in intervals:
        if beg <= (1 << 27) and end >= (1 << 27) + W:
            picks.append((ref, (1 << 27) - 1000, (1 << 27) - 1000 + W))
            break
    while len(picks) < n_windows:
        ref, beg, end = intervals[rng.randrange(len(intervals))]
        if end - beg <= W:
            picks.append((ref, beg, end))
            continue
        a = rng.randrange(beg, end - W)
        picks.append((ref, a, a + W))
    bad = []
    S = d.n_samples_eff
    for ref, a, b in picks:
        got = d.base_counters(ref, a, b)
        want = oracle_base_counters(bam, ref, a, b, n_samples=S, min_bq=min_bq, fix_mate=fix_mate, ref_name=d.ref_names[ref])
        if not np.array_equal(got, want):
            bad.append([ref, a, b])
    return len(picks), bad


def parity_text(d, bam, ref_name, ref, beg, end, extra_args):
    """md5 of the device-formatted rows of [beg, end) against the oracle CLI's text for the same region."""
    got = d.format_base_rows(ref, beg, end)
    r = subprocess.run([os.path.join(ROOT, "oracle", "depth_oracle"), "base"] + extra_args +
                       ["-L", "%s:%d-%d" % (ref_name, beg + 1, end), bam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    want = r.stdout.split(b"\n", 1)[1] if r.stdout else b""      # drop the header line
    return hashlib.md5(got).hexdigest(), hashlib.md5(want).hexdigest(), len(got)


def parity_full_text(d, bam, intervals, extra_args, min_slices=64, share_of_host=1):
    """WHOLE-share parity of the timed results: every interval [(ref, beg, end)] is cut into position slices (>= `min_slices` in all,
    at least one per host core), the oracle CLI prints `depth base -L chr:a-b` for every slice on the host cores in parallel, and
    the md5 of its rows is compared with the md5 of the device-formatted rows (K6 over the counters the LAST timed pass left)
    of the same slice.  Every position of the share is covered exactly once.  Returns a dict for `parity_checked`."""
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.time()
    total = sum(e - b for _, b, e in intervals)
    if total <= 0:
        return {"coverage": 0.0, "slices": 0, "ok": True}
    ncpu = max(1, (os.cpu_count() or 8) // max(1, share_of_host))      # (the ranks of a node share its cores)
    n_slices = max(min_slices, min(ncpu, 256))
    step = max(1024, -(-total // n_slices) // 1024 * 1024 + 1024)
    slices = [(r, a, min(e, a + step)) for r, b, e in intervals for a in range(b, e, step)]
    oracle = os.path.join(ROOT, "oracle", "depth_oracle")

    def want_md5(sl):
        r, a, b = sl
        p = subprocess.Popen([oracle, "base"] + extra_args + ["-L", "%s:%d-%d" % (d.ref_names[r], a + 1, b), bam],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, bufsize=0)
        h, n, first = hashlib.md5(), 0, True
        while True:
            buf = p.stdout.read(1 << 22)
            if not buf:
                break
            if first:                       # drop the header line
                nl = buf.find(b"\n")
                if nl < 0:
                    continue
                buf, first = buf[nl + 1:], False
            h.update(buf)
            n += len(buf)
        if p.wait() != 0:
            return None, n
        return h.hexdigest(), n

    workers = max(1, min(len(slices), ncpu))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        fut = [ex.submit(want_md5, sl) for sl in slices]
        # the device side next to the oracle processes: format, hash, drop
        hpool = ThreadPoolExecutor(max_workers=4)
        got = []
        for r, a, b in slices:
            txt = d.format_base_rows(r, a, b)
            got.append((hpool.submit(lambda t=txt: hashlib.md5(t).hexdigest()), len(txt)))
        got = [(f.result(), n) for f, n in got]
        hpool.shutdown()
        want = [f.result() for f in fut]
    bad = [list(slices[i]) for i in range(len(slices)) if got[i][0] != want[i][0] or got[i][1] != want[i][1]]
    allmd5 = hashlib.md5("".join(g[0] for g in got).encode()).hexdigest()
    return {"coverage": round(sum(b - a for _, a, b in slices) / float(total), 6), "slices": len(slices), "ok": not bad,
            "mismatching_slices": bad[:4], "text_bytes": int(sum(n for _, n in got)), "md5_of_slice_md5s": allmd5,
            "host_workers": workers, "seconds": round(time.time() - t0, 1),
            "what": "md5 of ALL device-formatted rows of this rank's share (K6 over the counters of the last timed pass), slice by slice, "
                    "against `depth_oracle base -L chr:a-b` run for every slice on the host cores"}


def parity_full_regions(d, bam, regs, got_rows, share_of_host=1):
    """EVERY window / BED region of a whole contig against the oracle: `regs` = [(ref, beg, end)] in position order, `got_rows` =
    the device's (readCount, meanCoverage as %g text) per region.  The regions are cut into consecutive slices, one BED file and one
    `depth_oracle region -L slice.bed` process per slice on the host cores; every row is compared.  Returns a dict."""
    from concurrent.futures import ThreadPoolExecutor
    import tempfile
    t0 = time.time()
    if not regs:
        return {"regions": 0, "ok": True}
    ncpu = max(1, (os.cpu_count() or 8) // max(1, share_of_host))
    n_sl = max(1, min(len(regs), min(ncpu, 256)))
    per = -(-len(regs) // n_sl)
    oracle = os.path.join(ROOT, "oracle", "depth_oracle")
    tmpd = tempfile.mkdtemp(prefix="sbx_par_", dir=tmp_dir())

    def one(k):
        part = regs[k * per:(k + 1) * per]
        if not part:
            return []
        bed = os.path.join(tmpd, "s%d.bed" % k)
        with open(bed, "w") as fh:
            fh.writelines("%s\t%d\t%d\n" % (d.ref_names[r], a, b) for r, a, b in part)
        out = subprocess.run([oracle, "region", "-L", bed, bam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
        rows = [ln.split("\t") for ln in out.splitlines() if ln and not ln.startswith("#")]
        os.remove(bed)
        return [(int(f[1]), int(f[2]), int(f[3]), f[4]) for f in rows]

    with ThreadPoolExecutor(max_workers=min(n_sl, ncpu)) as ex:
        want = [row for part in ex.map(one, range(n_sl)) for row in part]
    try:
        os.rmdir(tmpd)
    except OSError:
        pass
    # (the reference prints no row for a region behind the last pileup column of its input -- PerRegionPrinter only learns of a region
    #  when a column passes it, depth.d:661-698 -- so a slice that ends in read-less regions has fewer rows than regions: a region the
    #  oracle is silent about must be all-zero on the device)
    by_pos = {(wa, wb): (wn, wm) for wa, wb, wn, wm in want}
    bad, silent = [], 0
    known = set((x[1], x[2]) for x in regs)
    if len(by_pos) != len(want) or any(k not in known for k in by_pos):
        bad.append(["rows the list does not hold, or duplicates", len(want), len(regs)])
    for i, (r, a, b) in enumerate(regs):
        w = by_pos.get((a, b))
        if w is None:
            silent += 1
            w = (0, "0")
        if got_rows[i] != w:
            bad.append([r, a, b, list(got_rows[i]), list(w)])
            if len(bad) >= 4:
                break
    return {"regions": len(regs), "ok": not bad, "mismatches": bad, "oracle_rows": len(want), "regions_without_a_row": silent,
            "oracle_processes": n_sl, "seconds": round(time.time() - t0, 1)}


def cli_e2e(bam, mode_args, reads, pause=3.0):
    """The product CLI end to end: file (page cache) -> device -> text on /dev/null, wall clock of the command as ONE process,
    teardown of the device context included (`seconds`: the like-for-like figure next to cpu_baseline).  Three runs, the best is
    reported, all are listed, with the CLI's own phase clock of the best one.  Two more runs with SBX_DETACH=1 -- the work in a
    child process, the command returns when the output is complete and leaves the teardown to the child -- are a side field
    (`detached_seconds`); a few seconds apart: a process started right behind a detached teardown waits for the driver."""
    from sambamba_amd import cli_path
    cmd = [cli_path()] + mode_args + ["-o", "/dev/null", bam]

    def once(env):
        t0 = time.time()
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        dt = time.time() - t0
        if r.returncode != 0:
            raise RuntimeError(r.stderr.decode()[-300:])
        return dt, [ln for ln in r.stderr.decode().splitlines() if ln.startswith("[sbx")]

    try:
        runs, det = [], []
        for k in range(3):
            if k:
                time.sleep(pause)    # (a process started right behind another one's teardown waits for the driver to scrub the freed memory)
            runs.append(once(dict(os.environ, SBX_TIMING="1")))
        for k in range(2):
            time.sleep(pause)
            det.append(once(dict(os.environ, SBX_TIMING="1", SBX_DETACH="1")))
        time.sleep(min(pause, 2.0))
    except RuntimeError as e:
        return {"error": str(e)}
    best = min(runs, key=lambda x: x[0])
    d = min(det, key=lambda x: x[0])
    return {"seconds": round(best[0], 3), "Mreads_per_s": round(reads / best[0] / 1e6, 2), "all_seconds": [round(x[0], 3) for x in runs],
            "detached_seconds": round(d[0], 3), "detached_Mreads_per_s": round(reads / d[0] / 1e6, 2),
            "detached_all_seconds": [round(x[0], 3) for x in det],
            "phases": best[1][-4:],
            "what": "sbx-depth %s -o /dev/null <bam> (file in page cache; process start, open, pinned H2D, device kernels, device text "
                    "formatting, pinned D2H, write, teardown of the device context) as ONE process, best of 3; `detached_seconds` = the "
                    "same with SBX_DETACH=1: slices through upload / kernels / text concurrently in a child process, the command "
                    "returns with the output complete and the child tears the context down" % " ".join(mode_args)}


class Job:
    """One BAM opened on this rank's GPU with the parameters of a BASELINE config, and the passes over it."""

    def __init__(self, args, path, info, dev_index, rank, world, dist, seed):
        import sambamba_amd
        from sambamba_amd import shard as shardmod
        self.args, self.path, self.info, self.rank, self.world, self.dist, self.seed = args, path, info, rank, world, dist, seed
        self.shardmod = shardmod
        SBX = sambamba_amd
        d = self.d = sambamba_amd.Depth(path, device=dev_index)
        self.mode_args, self.min_bq, self.fix_mate, self.regions = ["base"], 0, False, None
        if args.config == 2:
            d.set_params()             # depth base, default filter, -q 0
        elif args.config == 3:
            d.set_params(mode=SBX.SBX_MODE_WINDOW, window=1000)
            self.mode_args = ["window", "-w", "1000"]
        elif args.config == 4:
            bed = path + ".exome.bed"
            if rank == 0:
                exome_bed(bed, d.ref_names, d.ref_lengths, scale=args.scale)
            if dist:
                dist.barrier()
            d.set_params(mode=SBX.SBX_MODE_REGION)
            self.regions = shardmod.read_bed_regions(bed, d.ref_names)
            import numpy as np
            self.regions_arr = np.ascontiguousarray(np.asarray(self.regions, dtype=np.uint32).reshape(-1, 3))
            self.mode_args = ["region", "-L", bed]
        elif args.config == 5:
            self.min_bq, self.fix_mate = 20, True
            d.set_params(min_bq=20, fix_mate_overlaps=True)
            self.mode_args = ["base", "-m", "-q", "20"]
        self.ref_lengths = d.ref_lengths
        self.window_rows = 0

    def share(self, sharded):
        """(my, plan): `my` = this rank's position slice [(ref, beg, end)] of the sharded BAM, or None with `plan` = the
        batches of the whole BAM."""
        d, args, regions, shardmod = self.d, self.args, self.regions, self.shardmod
        if sharded:
            align = 1000 if args.config == 3 else 1024
            mine = shardmod.plan_position_shards(self.ref_lengths, self.world, align=align)[self.rank]
            if regions is not None:
                mine = shardmod.clip_regions_to_shards(regions, mine)
            return mine, None
        if args.config in (3, 4) and regions is None:
            d.preload()
            return None, d.plan_batches()
        if regions is not None:
            d.set_regions(shardmod.merge_regions(regions))
            return None, [None]
        d.preload()                    # compressed BAM resident in HBM before the timed region
        return None, [None]

    def one_pass(self, my, plan):
        """One pass of the hot path over this rank's share; returns the list of per-run statistics."""
        d, args, regions, ref_lengths = self.d, self.args, self.regions, self.ref_lengths
        sts = []
        if my is not None:
            if regions is not None:
                d.set_regions(self.shardmod.merge_regions(my))
                sts.append(d.run())
                if my:
                    d.region_stats(my)
            else:
                for ref, beg, end in my:
                    sts.append(d.run_interval(ref, beg, end))
                    if args.config == 3:
                        n = (min(end, ref_lengths[ref]) // 1000) - beg // 1000
                        if n > 0:
                            d.window_stats(ref, beg // 1000, n)
                            self.window_rows += n
            return sts
        for b in plan:
            if b is None:
                sts.append(d.run())
                if regions is not None:
                    d.region_stats(self.regions_arr)
            else:
                sts.append(d.run_batch(b[0], b[1]))
                if args.config == 3:
                    for r in range(b[0], b[0] + b[1]):
                        n = ref_lengths[r] // 1000
                        if n:
                            d.window_stats(r, 0, n)
                            self.window_rows += n
        return sts

    def timed(self, sharded, warmup, steps, sync, red_dev):
        """warmup + `steps` timed passes (barrier + device synchronisation on both sides, MAX over the ranks)."""
        import torch
        dist = self.dist
        my, plan = self.share(sharded)
        for _ in range(warmup):
            self.one_pass(my, plan)
        sync()
        t0 = time.perf_counter()
        ks = []
        for _ in range(steps):
            ks.append(self.one_pass(my, plan))
        sync()
        el = time.perf_counter() - t0
        lastp = ks[-1]
        reads = float(sum(s["n_records"] for s in lastp))
        adm = float(sum(s["n_admitted"] for s in lastp))
        if dist:
            t = torch.tensor([el], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            r = torch.tensor([reads, adm], dtype=torch.float64, device=red_dev)
            dist.all_reduce(r, op=dist.ReduceOp.SUM)
            reads, adm = float(r[0].item()), float(r[1].item())
        return {"my": my, "plan": plan, "kstats": ks, "elapsed": el, "sum_reads": reads, "sum_adm": adm}

    def device_text(self, steps, sync):
        """`depth base` prints text: one pass INCLUDING the formatting of every row (K6), the text left in HBM -- nothing crosses PCIe,
        nothing is skipped.  Returns ms per pass and the rate next to the counters-only `value`."""
        import hashlib as hl
        import torch
        d = self.d
        d.run()
        piece = 8 << 20
        spans = []
        for ref in range(len(self.ref_lengths)):
            at = 0
            while True:
                r = d.next_active_range(ref, at)
                if r is None:
                    break
                for b in range(r[0], r[1], piece):
                    spans.append((ref, b, min(r[1], b + piece)))
                at = r[1]
        sizes = [d.format_base_rows_to_device(ref, b, e, 0, 0) for ref, b, e in spans]
        total = sum(sizes)
        buf = torch.empty(total + 64, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
        sync()
        t0 = time.perf_counter()
        ms_run = 0.0
        for _ in range(steps):
            st = d.run()
            ms_run += st["ms_total"]
            off = 0
            for (ref, b, e), sz in zip(spans, sizes):
                got = d.format_base_rows_to_device(ref, b, e, buf.data_ptr() + off, total - off)
                if got != sz:
                    raise RuntimeError("device text: measured %d bytes, wrote %d" % (sz, got))
                off += sz
        sync()
        el = (time.perf_counter() - t0) / steps
        # the bytes are the ones the host-copy path hands out (which the parity checks hold against the oracle): first and last span
        ok = True
        for i in sorted(set([0, len(spans) - 1])) if spans else []:
            off = sum(sizes[:i])
            dev = bytes(buf[off:off + sizes[i]].cpu().numpy().tobytes())
            host = d.format_base_rows(spans[i][0], spans[i][1], spans[i][2])
            ok = ok and hl.md5(dev).digest() == hl.md5(host).digest()
        reads = float(self.info["reads"])
        return {"ms_per_step": round(el * 1e3, 3), "value": round(reads / el / 1e6, 3), "unit": "Mreads/s", "steps": steps,
                "text_bytes": int(total), "ms_kernels_of_the_pass": round(ms_run / steps, 3),
                "ms_format_and_host_calls": round(el * 1e3 - ms_run / steps, 3), "text_equals_host_copy_path": bool(ok),
                "what": "one pass of the hot path PLUS the text of every row of `depth base` (format_measure + format_write over all "
                        "positions, pieces of 8 M positions), the text left in HBM (sbx_format_base_rows_device); wall clock"}

    def parity(self, my, plan, n_windows):
        """Device results of the LAST pass against the CPU oracle (every rank checks its own share)."""
        import numpy as np
        d, args, path, seed, rank = self.d, self.args, self.path, self.seed, self.rank
        ref_lengths, regions = self.ref_lengths, self.regions
        par = {"windows": 0, "ok": True, "mismatches": []}
        if n_windows > 0 and args.config in (2, 5):
            if my is not None:
                ivs = [iv for iv in my][-1:]       # the last run of the pass left the last interval resident
            else:
                ivs = [(r, 0, ref_lengths[r]) for r in range(len(ref_lengths)) if ref_lengths[r] > 0]
            if ivs:
                n, bad = parity_windows(d, path, ivs, n_windows, seed ^ rank, min_bq=self.min_bq, fix_mate=self.fix_mate)
                par = {"windows": n, "ok": not bad, "mismatches": bad[:4]}
                ref, a, b = ivs[-1]
                a2 = a + (b - a) // 3 // 1024 * 1024
                b2 = min(b, a2 + 1_000_000)
                md_got, md_want, nbytes = parity_text(d, path, d.ref_names[ref], ref, a2, b2, self.mode_args[1:])
                par.update({"text_slab": [ref, a2, b2], "text_bytes": nbytes, "text_md5": md_got, "text_ok": md_got == md_want})
                par["ok"] = par["ok"] and md_got == md_want
                if args.full_parity:
                    full = parity_full_text(d, path, ivs, self.mode_args[1:], share_of_host=self.world)
                    par["full_text"] = full
                    par["coverage"] = full["coverage"]
                    par["ok"] = par["ok"] and full["ok"]
        if n_windows > 0 and args.config in (3, 4) and my is None:
            # window / region statistics against the oracle's `depth region -L chr:a-b` (a window is the region [k w, (k + 1) w);
            # the reads come through the BAI, seconds per call).  Config 3 streams the genome in batches and only the last one is
            # still resident after the timed region: the first and a middle batch are run once more for the check.
            rng = random.Random(seed ^ 0x33)
            checked, bad, where = 0, [], []
            oracle = os.path.join(ROOT, "oracle", "depth_oracle")
            full = None
            if args.full_parity:
                # every window / every BED region of ONE WHOLE CONTIG (the longest one that is still resident) against the oracle
                if args.config == 3:
                    batches = [b for b in plan if b is not None]
                    refs_res = range(batches[-1][0], batches[-1][0] + batches[-1][1]) if batches else range(len(ref_lengths))
                    r_full = max(refs_res, key=lambda r: ref_lengths[r])
                    nwin = ref_lengths[r_full] // 1000
                    regs = [(r_full, k * 1000, (k + 1) * 1000) for k in range(nwin)]
                    nr, nb, _cov = d.window_stats(r_full, 0, nwin) if nwin else ([], [], None)
                    rows = [(int(nr[k][0]), "%g" % float(np.float32(nb[k][0]) / np.float32(1000))) for k in range(nwin)]
                else:
                    by_ref = {}
                    for r, a, b in regions:
                        by_ref.setdefault(r, []).append((r, a, b))
                    r_full = max(by_ref, key=lambda r: len(by_ref[r]))
                    regs = by_ref[r_full]
                    nr, nb, _cov, _seen = d.region_stats(regs)
                    rows = [(int(nr[j][0]), "%g" % float(np.float32(nb[j][0]) / np.float32(b - a))) for j, (r, a, b) in enumerate(regs)]
                full = parity_full_regions(d, path, regs, rows, share_of_host=self.world)
                full["contig"] = d.ref_names[r_full]
                full["what"] = "readCount and meanCoverage of EVERY %s of contig %s against `depth_oracle region -L`" % (
                    "window" if args.config == 3 else "BED region", d.ref_names[r_full])

            def want_of(r, a, b):
                out = subprocess.run([oracle, "region", "-L", "%s:%d-%d" % (d.ref_names[r], a + 1, b), path],
                                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode().splitlines()
                f = out[-1].split("\t")
                return (int(f[3]), f[4])

            if args.config == 3:
                batches = [b for b in plan if b is not None]
                pick = sorted(set([len(batches) - 1, 0, len(batches) // 2])) if batches else [None]
                per = max(1, -(-n_windows // len(pick)))
                for bi in reversed(pick):          # the last batch is resident: check it first
                    if bi is None:
                        refs_res = range(len(ref_lengths))
                    else:
                        lastb = batches[bi]
                        refs_res = range(lastb[0], lastb[0] + lastb[1])
                        if bi != len(batches) - 1:
                            d.run_batch(lastb[0], lastb[1])
                    cands = [r for r in refs_res if ref_lengths[r] >= 4000]
                    for j in range(per):
                        if not cands:
                            break
                        r = cands[rng.randrange(len(cands))]
                        nwin = ref_lengths[r] // 1000
                        k = nwin - 1 if j == 0 else rng.randrange(1, nwin - 1)       # the last full window of a contig, then random ones
                        nr, nb, _cov = d.window_stats(r, k, 1)
                        got = (int(nr[0][0]), "%g" % float(np.float32(nb[0][0]) / np.float32(1000)))
                        want = want_of(r, k * 1000, (k + 1) * 1000)
                        checked += 1
                        where.append([int(bi) if bi is not None else -1, r, k])
                        if got != want:
                            bad.append([r, k, list(got), list(want)])
            else:
                picks = [regions[0], regions[len(regions) // 2], regions[-1]] + [regions[rng.randrange(len(regions))] for _ in range(max(0, n_windows - 3))]
                nr, nb, _cov, _seen = d.region_stats(picks)
                for j, (r, a, b) in enumerate(picks):
                    got = (int(nr[j][0]), "%g" % float(np.float32(nb[j][0]) / np.float32(b - a)))
                    want = want_of(r, a, b)
                    checked += 1
                    where.append([r, a, b])
                    if got != want:
                        bad.append([r, a, b, list(got), list(want)])
            if full is not None and not full["ok"]:
                bad.append(["whole contig", full["contig"]] + full["mismatches"][:2])
            par = {"windows": checked, "ok": not bad, "mismatches": bad[:4], "checked": where[:32], "whole_contig": full,
                   "what": "readCount and meanCoverage of sampled %s against `depth_oracle region -L`%s" % (
                       "windows" if args.config == 3 else "BED regions (first, middle, last of the BED + random ones)",
                       " (batch index, contig, window; first / middle / last batch of the streamed genome)" if args.config == 3 else "")}
        return par


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=int(os.environ.get("SBX_BENCH_STEPS", 150)),
                    help="passes in the timed region (default: ~10 s of device time on configs[1])")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=int(os.environ.get("SBX_BENCH_CONFIG", 2)), help="BASELINE.json config number (2..5)")
    ap.add_argument("--mode", choices=["auto", "shard", "strong", "replicas"], default=os.environ.get("SBX_BENCH_MODE", "auto"))
    ap.add_argument("--length", type=int, default=int(os.environ.get("SBX_BENCH_LEN", CHR1_LEN)),
                    help="contig length of config 2 (default: chr1; smaller values are for development only)")
    ap.add_argument("--scale", type=float, default=float(os.environ.get("SBX_BENCH_SCALE", 1.0)),
                    help="configs 3-5: shrink every contig by this factor (development)")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--codec", default=os.environ.get("SBX_BENCH_CODEC", "zlib"))
    ap.add_argument("--cpu-sample-reads", type=int, default=int(os.environ.get("SBX_BENCH_CPU_READS", 10_000_000)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--e2e-pause", type=float, default=3.0, help="seconds between the end-to-end CLI runs (the tests use a short one)")
    ap.add_argument("--no-side-runs", action="store_true", help="N > 1: skip the strong-scaling / all-reduce side measurements")
    ap.add_argument("--parity-windows", type=int, default=8)
    ap.add_argument("--no-full-parity", dest="full_parity", action="store_false",
                    help="skip the whole-share comparison of the device text with the oracle (configs 2 and 5)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # one rank per GPU; SBX_BENCH_BACKEND=gloo lets the multi-rank path be exercised on a box with fewer GPUs
    # than ranks (ranks then share devices and the timing reductions travel through host memory)
    backend = os.environ.get("SBX_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if world > 1 else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # rank 0 generates the synthetic BAM (minutes for N chr1-sized contigs on a CPU-quota'd box) while the others wait in a barrier
        long_wait = datetime.timedelta(minutes=90)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=long_wait)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=long_wait)

    if rank == 0:
        ensure_built()
    if dist:
        dist.barrier()

    mode = args.mode
    if mode == "auto":
        mode = "shard"
    if world == 1:
        mode = "single"
    # shard: ONE BAM holding one contig of the config's size per rank (configs 2, 5) or the genome (configs 3, 4)
    copies = world if (mode == "shard" and args.config in (2, 5)) else 1
    contigs, coverage, seed, extra, cfg_name = workload(args, copies)
    path = bam_path(contigs, coverage, seed, args.level, args.codec, extra)
    if rank == 0:
        log("built; generating %s" % path)
        make_room({path}, int(6e9) * copies * (13 if args.config in (3, 4) else 1))
        info = generate(path, contigs, coverage, seed, args.level, args.codec, extra)
        log("BAM ready: %d reads" % int(info["reads"]))
    if dist:
        dist.barrier()
    if rank != 0:
        info = json.load(open(path + ".json"))

    def sync():
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    sharded = mode in ("shard", "strong")
    job = Job(args, path, info, dev_index, rank, world, dist, seed)
    d = job.d
    ref_lengths = job.ref_lengths
    mode_args = job.mode_args
    log("context open; warmup + timed region: %d steps (%s)" % (args.steps, "one BAM sharded over the ranks" if sharded else
                                                                "whole BAM per rank"))
    if args.config == 4 and world == 1:
        # a `depth region` command runs once: the BAI query of 200 k regions, its grouping into runs and K2's read-selection table are
        # part of what it pays.  The engine keeps them between identical runs (engine.cpp RunsCache); the headline is timed with the cache
        # off, the re-run figure is a side field
        os.environ["SBX_RUNS_CACHE"] = "0"
    main_run = job.timed(sharded, args.warmup, args.steps, sync, red_dev)
    rerun_cached = None
    if args.config == 4 and world == 1:
        del os.environ["SBX_RUNS_CACHE"]
        if not args.no_side_runs:
            rr = job.timed(sharded, 1, max(2, args.steps // 2), sync, red_dev)
            n_rr = max(2, args.steps // 2)
            rerun_cached = {"ms_per_step": round(rr["elapsed"] / n_rr * 1e3, 3), "value": round(rr["sum_reads"] / (rr["elapsed"] / n_rr) / 1e6, 3),
                            "unit": "Mreads/s", "steps": n_rr,
                            "what": "the same pass repeated on the same context with the work list of the first run kept (BAI query, chain runs, "
                                    "read-selection table): what a caller that asks twice pays, not what `depth region` pays"}
    my, plan, kstats, elapsed = main_run["my"], main_run["plan"], main_run["kstats"], main_run["elapsed"]
    sum_reads, sum_adm = main_run["sum_reads"], main_run["sum_adm"]
    log("timed region done: %.1f ms per step; parity" % (elapsed / args.steps * 1e3))
    last = kstats[-1]
    # sharded: reads near a cut are seen by both neighbours, the job's reads are the file's
    total_reads = float(info["reads"]) if sharded else sum_reads
    total_admitted = sum_adm * (total_reads / sum_reads) if (sharded and sum_reads) else sum_adm

    # ---- parity of THIS run's results at scale (every rank checks its own share) --------------------------------
    par = job.parity(my, plan, args.parity_windows)
    if dist:
        okt = torch.tensor([1.0 if par["ok"] else 0.0, float(par["windows"])], dtype=torch.float64, device=red_dev)
        allok = okt.clone()
        dist.all_reduce(allok, op=dist.ReduceOp.MIN)
        dist.all_reduce(okt, op=dist.ReduceOp.SUM)
        par["ok_all_ranks"] = bool(allok[0].item() >= 1.0)
        par["windows_all_ranks"] = int(okt[1].item())
    parity_ok = par.get("ok_all_ranks", par["ok"])

    device_text = None
    if world == 1 and args.config in (2, 5) and not args.no_side_runs:
        try:
            device_text = job.device_text(max(3, args.steps // 4), sync)
        except Exception as e:      # a side measurement must not take the headline down
            device_text = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    # ---- N > 1: what the collective layer saw (so that a scaling record can be audited: RCCL with N ranks on N distinct devices) -----
    coll = None
    if dist:
        import socket
        props = torch.cuda.get_device_properties(dev_index)
        knames = {"huffman_decode": "ms_huffman", "lz77_resolve": "ms_lz77", "record_index": "ms_index", "decode_accumulate": "ms_accumulate"}
        me = {"rank": rank, "local_rank": local_rank, "host": socket.gethostname(), "device_index": dev_index, "device": props.name,
              "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
              "kernel_ms": {k: round(sum(sum(st[v] for st in p) for p in kstats) / max(1, len(kstats)), 3) for k, v in knames.items()},
              "reads_seen": int(sum(st["n_records"] for st in kstats[-1])) if kstats else 0,
              "blocks": int(sum(st["n_bgzf_blocks"] for st in kstats[-1])) if kstats else 0}
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        ones = torch.ones(1, dtype=torch.float32, device=red_dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)          # a collective over the data-path backend: must add up to the world size
        ident = set((e["host"], e["uuid"] or e["pci_bus_id"] or e["device_index"]) for e in everyone)
        coll = {"backend": str(dist.get_backend()), "is_rccl": backend == "nccl" and torch.version.hip is not None,
                "world_size": world, "unique_devices": len(ident), "allreduce_of_ones": float(ones.item()),
                "ranks": everyone,
                "what": "torch.distributed backend of this run (`nccl` on a ROCm build IS RCCL), an all-reduce of ones over it, and per rank: "
                        "host, device identity, per-kernel ms per pass, reads and BGZF blocks of its share"}

    # ---- N > 1, default mode, side measurements -------------------------------------------------------------------
    strong = None
    allred = None
    if world > 1 and mode == "shard" and args.config in (2, 5) and not args.no_side_runs:
        # (a) the config's single contig cut into N position slices: strong scaling
        contigs1, cov1, seed1, extra1, _ = workload(args, 1)
        path1 = bam_path(contigs1, cov1, seed1, args.level, args.codec, extra1)
        if rank == 0:
            info1 = generate(path1, contigs1, cov1, seed1, args.level, args.codec, extra1)
        dist.barrier()
        if rank != 0:
            info1 = json.load(open(path1 + ".json"))
        d.close()
        job1 = Job(args, path1, info1, dev_index, rank, world, dist, seed1)
        s_steps = max(3, args.steps // 4)
        log("strong scaling: one contig cut into %d slices, %d steps" % (world, s_steps))
        sh = job1.timed(True, 1, s_steps, sync, red_dev)
        ok_s = True
        if args.parity_windows > 0 and sh["my"]:
            n_s, bad_s = parity_windows(job1.d, path1, [sh["my"][-1]], max(2, args.parity_windows // 2), seed1 ^ (rank + 77),
                                        min_bq=job1.min_bq, fix_mate=job1.fix_mate)
            ok_s = not bad_s
        okt = torch.tensor([1.0 if ok_s else 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok_s = bool(okt[0].item() >= 1.0)
        strong = {"scaling": "strong", "steps": s_steps, "ms_per_step": round(sh["elapsed"] / s_steps * 1e3, 3),
                  "value": round(float(info1["reads"]) / (sh["elapsed"] / s_steps) / 1e6, 3), "unit": "Mreads/s", "parity_ok": ok_s,
                  "what": "the config's single-contig BAM cut into %d position slices (sambamba_amd.shard.plan_position_shards), every rank "
                          "fetches the reads of its slice through the BAI and clips its contributions to the slice; no data-path collective. "
                          "Bounded below by one residency of the lane-per-block Huffman kernel (DESIGN.md section 7)" % world}
        parity_ok = parity_ok and ok_s
        # (b) the alternative north_star names: reads partitioned by start position, counters summed with an all-reduce
        if args.config == 2:
            try:
                import numpy as np
                from sambamba_amd.dist_depth import allreduce_base_counters
                from sambamba_amd.shard import plan_position_shards
                from tests.util import oracle_base_counters
                # windows straddling the cuts between the owners: every position there is a sum over two ranks
                cuts = [iv[0][1] for iv in plan_position_shards(job1.ref_lengths, world, align=1024)[1:] if iv]
                wins = [(0, max(0, c - 25_000), min(job1.ref_lengths[0], c + 25_000)) for c in cuts[:3]] if args.parity_windows > 0 else []
                log("all-reduce option: reads partitioned by start position, counters summed over %d ranks" % world)
                allred = allreduce_base_counters(job1.d, dist, world, rank, red_dev, windows=wins, steps=max(2, min(5, args.steps)))
                got = allred.pop("windows")
                ok_a = True
                if rank == 0:
                    for (r, a0, b0) in wins:
                        want = oracle_base_counters(path1, r, a0, b0, n_samples=job1.d.n_samples_eff, ref_name=job1.d.ref_names[r])
                        ok_a = ok_a and (r, a0, b0) in got and bool(np.array_equal(got[(r, a0, b0)], want))
                okt = torch.tensor([1.0 if ok_a else 0.0], dtype=torch.float64, device=red_dev)
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                allred["parity_ok"] = bool(okt[0].item() >= 1.0)
                allred["parity_windows"] = [list(w) for w in wins]
                if not allred["parity_ok"]:
                    parity_ok = False
            except Exception as e:     # a side measurement must not take the headline down
                allred = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        d = job1.d

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        reads_per_s = total_reads / (elapsed / args.steps)
        read_len = 150
        names = {"huffman_decode": "ms_huffman", "lz77_resolve": "ms_lz77", "record_index": "ms_index", "decode_accumulate": "ms_accumulate"}
        # per-kernel time of one pass on rank 0: summed over the runs of a pass, averaged over the steps
        kern = {k: sum(sum(s[v] for s in p) for p in kstats) / len(kstats) for k, v in names.items()}
        dom = max(kern, key=kern.get)
        comp = sum(s["compressed_bytes"] for s in last)
        unc = sum(s["uncompressed_bytes"] for s in last)
        cnt = sum(s["counter_bytes"] for s in last)
        nrec = sum(s["n_records"] for s in last)
        tok = sum(s["token_bytes"] for s in last)
        acc_in = sum(s["accumulate_read_bytes"] for s in last)
        # ALGORITHMIC bytes per pass of rank 0: what each kernel has to move, measured by the run itself
        # (the K1a -> K1b token streams are an intermediate of THIS design -- SURVEY 8(d) names compressed bytes in and
        # inflated bytes / counters out -- so they are reported as `intermediate_bytes`, not counted in a kernel's achieved GB/s)
        alg = {"huffman_decode": comp,                        # compressed bytes in
               "lz77_resolve": unc,                           # inflated bytes out
               "record_index": nrec * (36 + 32),              # 36-B fixed part read + 32-B descriptor written per record
               "decode_accumulate": acc_in + cnt}             # descriptors + CIGAR / packed sequence (+ qualities when -q > 0) of the
                                                              # admitted records in; counters (+ span counts) out, once
        if args.config in (3, 4):
            # window / region modes print O(windows) numbers: per-position counters are an intermediate there (B_read ~ R_c)
            alg["decode_accumulate"] = acc_in
        inter = {"huffman_decode": tok, "lz77_resolve": tok, "record_index": 0,
                 "decode_accumulate": cnt if args.config in (3, 4) else 0}
        full_size = (args.config == 2 and args.length == CHR1_LEN) or (args.config != 2 and args.scale == 1.0)
        pmc, pmc_note = pmc_table(args.config) if (full_size and world == 1) else ({}, "not the full-size single-GPU workload")
        sq = sq_table(args.config) if (full_size and world == 1) else {}
        per_kernel = {}
        bad_frac = []
        for k in kern:
            if kern[k] <= 0:
                continue
            gbps = alg[k] / (kern[k] * 1e-3) / 1e9
            e = {"ms": round(kern[k], 4), "algorithmic_bytes": int(alg[k]), "algorithmic_GBps": round(gbps, 2),
                 "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBS, 5), "intermediate_bytes": int(inter[k])}
            if k in sq:
                e["issue_roofline"] = issue_roofline(sq[k], kern[k])
            if k in pmc:
                e["traffic_bytes"] = pmc[k]["traffic"]
                e["traffic_over_algorithmic"] = round(pmc[k]["traffic"] / max(1.0, alg[k]), 3)
            if not (0.0 < gbps / HBM_PEAK_GBS <= 1.0):
                bad_frac.append(k)
            per_kernel[k] = e
        # the fused-path figures of SURVEY.md 8(d): compressed bytes in + counters out over the whole pass
        pacc = path_accounting(comp, cnt, kern, pmc, args.config)
        fused = pacc["path_GBps"] or 0.0
        roof = {"bound": "hbm", "kernel": dom, "achieved": per_kernel[dom]["algorithmic_GBps"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "frac": per_kernel[dom]["frac_of_hbm_peak"],
                "algorithmic_bytes_per_launch": int(alg[dom]), "intermediate_bytes_per_launch": int(inter[dom]),
                "kernel_ms": round(kern[dom], 4),
                "path_what": "SURVEY 8(d) fused figure: (compressed bytes in + counter bytes out) / sum of the kernel times of a pass -- the "
                             "north star's 0.50 refers to path_frac; traffic_total / path_traffic_over_algorithmic: HBM bytes of the whole "
                             "pass from the counter files against those algorithmic bytes"}
        roof.update(pacc)
        if "issue_roofline" in per_kernel[dom]:
            roof["issue_roofline"] = per_kernel[dom]["issue_roofline"]
        if dom not in pmc:
            roof["traffic_note"] = pmc_note
        if dom in pmc:
            roof.update({"traffic": pmc[dom]["traffic"], "traffic_unit": "bytes per launch", "traffic_source": pmc[dom]["source"],
                         "traffic_over_algorithmic": per_kernel[dom]["traffic_over_algorithmic"]})
            if dom == "huffman_decode" and os.environ.get("SBX_K1A", "2") == "1" and os.environ.get("SBX_K1A_BURST", "1") == "1":
                roof["traffic_note"] = ("K1a is issue / latency bound (the memory system moves 1.5 of its 8 TB/s under it); its default writes the "
                                        "token streams in 16-byte stores, which costs 28.6 GB of partial-sector traffic per launch and saves "
                                        "1.7 ms; SBX_K1A_BURST=22 (32-byte nontemporal bursts): 29.2 ms, 14.2 GB = 1.11 x algorithmic "
                                        "(DESIGN.md section 3, K1a; profiles/round3/call_l_stdout_summary.txt)")
        cpu = None
        log("parity %s; cpu baseline" % par.get("ok"))
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(path, min(args.cpu_sample_reads, int(info["reads"])), mode_args)
        e2e = None
        log("cpu baseline done; e2e CLI")
        if not args.no_e2e and world == 1:
            e2e = cli_e2e(path, mode_args, int(info["reads"]), pause=args.e2e_pause)
            if cpu and e2e.get("Mreads_per_s"):
                e2e["vs_cpu_baseline"] = round(e2e["Mreads_per_s"] / cpu["value"], 1)
        what = {2: "depth base", 3: "depth window -w 1000", 4: "depth region -L exome.bed", 5: "depth base --fix-mate-overlaps -q20"}[args.config]
        full = (args.config == 2 and args.length == CHR1_LEN) or (args.config != 2 and args.scale == 1.0)
        sharding = {"single": "single GPU",
                    "shard": "ONE BAM (%d contig(s)) sharded over the ranks by reference position (BAI fetch per slice, the slice's BGZF blocks "
                             "resident in the rank's HBM, contributions clipped to the slice), no data-path collective" % len(ref_lengths),
                    "strong": "the config's single-contig BAM cut into N position slices (BAI fetch per slice, contributions clipped), "
                              "no data-path collective",
                    "replicas": "every GPU runs the full workload (same seeded BAM), no data-path collective"}[mode]
        line = {
            "metric": "depth_%s_Mreads_per_s" % mode_args[0], "value": round(reads_per_s / 1e6, 3), "unit": "Mreads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "strong" if mode == "strong" else "weak", "vs_baseline": None, "dtype": "u8/u32", "data": "synthetic",
            "config": {"workload": "%s on synthetic %dx coordinate-sorted BAM (%d contig(s), %d Mbp, %d reads of %d bp, BGZF %s level %d, "
                                   "ratio %.2f), compressed bytes resident in HBM" % (
                                       what, int(coverage), len(ref_lengths), sum(ref_lengths) // 1_000_000, int(info["reads"]), read_len,
                                       args.codec, args.level, unc / max(1, comp)),
                       "baseline_config": cfg_name if full else cfg_name + " scaled down (development)",
                       "sharding": sharding},
            "gbases_per_s": round(total_admitted * read_len / (elapsed / args.steps) / 1e9, 3),
            "reads_total": int(total_reads), "reads_admitted": int(total_admitted),
            "roofline": roof, "kernels": per_kernel,
            "fused_path": {"algorithmic_GBps": round(fused, 1), "frac_of_hbm_peak": round(fused / HBM_PEAK_GBS, 5),
                           "what": "(compressed bytes in + counter bytes out) / sum of the kernel times of a pass, rank 0"},
            "device_text": device_text, "rerun_cached": rerun_cached,
            "strong_one_contig": strong, "allreduce_option": allred, "collective": coll,
            "cpu_baseline": cpu, "parity_checked": par, "e2e": e2e,
            "e2e_Mreads_per_s": (e2e or {}).get("Mreads_per_s"),
            "e2e_vs_cpu_baseline": (e2e or {}).get("vs_cpu_baseline"),
            "kernel_only_vs_cpu_baseline_note": "`value` is the device pipeline with the compressed bytes resident; the like-for-like pair is "
                                                "e2e_Mreads_per_s (file -> text) against cpu_baseline.value (file -> text)",
            "host": {"nproc": os.cpu_count(), "bam_gen_seconds": round(info.get("gen_seconds", 0.0), 1),
                     "bam_gen_phases": info.get("gen_phases"), "h2d_ms": round(max(s.get("ms_h2d", 0.0) for s in last), 1),
                     "runs_per_pass": len(last), "chain_runs": int(sum(s["n_runs"] for s in last)),
                     "window_rows_per_pass": job.window_rows // max(1, args.steps + args.warmup)},
        }
        if bad_frac:
            line["accounting_error"] = "roofline fraction outside (0, 1] for: " + ", ".join(bad_frac)
        log("done")
        print(json.dumps(line))
        sys.stdout.flush()
        if bad_frac:
            parity_ok = False
    d.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        sys.exit(3)


PMC_KERNELS = {"huffman_decode": ["k_huffman_decode", "k_huffman_decode2", "k_translate_literals"],
               "lz77_resolve": ["k_lz77_resolve_exact"],
               "record_index": ["k_walk_blocks", "k_check_scan", "k_check_scan_mw", "k_describe_blocks", "k_tile_compact", "k_tile_compact_mw", "k_chain_repair",
                                "k_rewalk_mismatched"],
               "decode_accumulate": ["k_accumulate16", "k_accumulate16c", "k_accumulate", "k_accumulate_mates", "k_find_mates_join", "k_find_partners",
                                     "k_mates_columns", "k_max_u32"]}


KERNEL_SOURCES = ["inflate.hip", "inflate2_core.hpp", "lz77_copy.hpp", "index.hip", "depth.hip", "mates.hip", "reduce.hip", "common.hpp", "kernels.hpp"]
CLOCK_GHZ = 2.4          # what the SIMDs run at under these kernels (GRBM_GUI_ACTIVE / duration: 2.35-2.43, profiles/round4/README.md)
N_SIMD = 1024


def _code_only(text):
    """C / C++ source without comments and with white space collapsed: what the compiler sees, near enough -- a reworded comment must
    not make a counter file stale, a changed statement must."""
    import re
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":                      # string / character literal: copied as it is
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return re.sub(r"\s+", " ", "".join(out)).strip()


def kernel_sources_hash():
    """sha1 over the CODE (comments stripped, white space collapsed) of the device sources of the hot path: the stamp of a committed
    counter file (`# sources <hash>` in its first line).  A counter file measured on other kernel code is STALE and is not joined
    into the line."""
    h = hashlib.sha1()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "sambamba_amd", "csrc", f), "r", errors="replace") as fh:
            h.update(_code_only(fh.read()).encode())
            h.update(b"\0")
    return "c" + h.hexdigest()[:15]


PROFILE_ROUND = "round6"


def _stamped_csv(name):
    """rows of profiles/<round>/<name> if the file exists and carries the stamp of the current kernel sources, else (None, why)"""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, name)
    if not os.path.exists(path):
        return None, "no counter pass committed for this workload (profiles/%s/%s)" % (PROFILE_ROUND, name)
    with open(path) as fh:
        first = fh.readline().strip()
        rows = [ln.strip().split(",") for ln in fh if ln.strip() and not ln.startswith("#") and not ln.startswith("kernel,")]
    want = kernel_sources_hash()
    if not first.startswith("# sources ") or first.split()[2] != want:
        return None, "profiles/%s/%s was measured on other kernel sources (%s; now %s): stale, not joined" % (PROFILE_ROUND, name, first[:40], want)
    return rows, "profiles/%s/%s (%s)" % (PROFILE_ROUND, name, first)


def pmc_table(config=2):
    """HBM bytes per pass of every kernel group from the committed PMC passes of this same workload AND these same kernel sources
    (tools/pmc_pass.sh: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs, KiB, summed over the launches of the
    ONE pass a counter run makes).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE is taken as is.
    Returns ({group: {...}}, note); an unstamped or stale file gives ({}, why): `traffic` is then null in the line."""
    rows, note = _stamped_csv("pmc_fetch_write_config%d.csv" % config)
    if rows is None:
        return {}, note
    best = {}
    for k, c, v, _ in rows:
        best[(k, c)] = best.get((k, c), 0.0) + float(v)
    out = {}
    for group, kernels in PMC_KERNELS.items():
        fetch = sum(v for (k, c), v in best.items() if k in kernels and c == "FETCH_SIZE") * 1024
        write = sum(v for (k, c), v in best.items() if k in kernels and c == "WRITE_SIZE") * 1024
        if fetch or write:
            out[group] = {"traffic": int(2 * fetch + write),
                          "source": "%s: rocprofv3 PMC, separate passes; FETCH_SIZE raw %.2f GB doubled, WRITE_SIZE %.2f GB" % (
                              note, fetch / 1e9, write / 1e9)}
    return out, note


def sq_table(config=2):
    """SQ instruction counters per pass and kernel group (tools/pmc_pass.sh, `--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS ...`),
    same stamp rule as pmc_table."""
    rows, note = _stamped_csv("pmc_sq_config%d.csv" % config)
    if rows is None:
        return {}
    acc = {}
    for k, c, v, _ in rows:
        acc[(k, c)] = acc.get((k, c), 0.0) + float(v)
    out = {}
    for group, kernels in PMC_KERNELS.items():
        e = {c: sum(v for (k, c2), v in acc.items() if k in kernels and c2 == c) for c in
             ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH")}
        if any(e.values()):
            e["source"] = note
            out[group] = e
    return out


def path_accounting(comp, cnt, kern_ms, pmc, config):
    """The fused-path figures of SURVEY.md 8(d) -- what the north star's 0.50 refers to -- as first-class fields of `roofline`
    (VERDICT r5, next 6c):
      path_frac                       (compressed bytes in + counter bytes out) / sum of the kernel times / HBM peak; for configs 3 / 4
                                      the per-position counters are an intermediate of the design (window / region modes print O(windows)
                                      numbers) and are left out,
      path_frac_incl_counters         the same with the counter bytes always counted: the definition rounds 2-4 used for every config
                                      (the redefined figure must not be compared with the older profiles),
      traffic_total                   HBM bytes of the whole pass from the counter files (sum over the kernel groups; null unless every
                                      group with a kernel time has a stamped counter entry),
      path_traffic_over_algorithmic   traffic_total / (compressed bytes in + counter bytes out): the waste of the unfused design."""
    total_ms = sum(kern_ms.values())
    out = {"path_frac": None, "path_GBps": None, "path_frac_incl_counters": None, "traffic_total": None, "path_traffic_over_algorithmic": None,
           "path_algorithmic_bytes": int(comp + (0 if config in (3, 4) else cnt))}
    if total_ms <= 0:
        return out
    fused = (comp + (0 if config in (3, 4) else cnt)) / (total_ms * 1e-3) / 1e9
    out["path_GBps"] = round(fused, 1)
    out["path_frac"] = round(fused / HBM_PEAK_GBS, 5)
    out["path_frac_incl_counters"] = round((comp + cnt) / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    groups = [k for k, v in kern_ms.items() if v > 0]
    if groups and all(k in pmc for k in groups):
        tt = sum(pmc[k]["traffic"] for k in groups)
        out["traffic_total"] = int(tt)
        out["path_traffic_over_algorithmic"] = round(tt / max(1.0, comp + cnt), 2)
    return out


def issue_roofline(sq, kernel_ms):
    """The roof the inflate kernels hit: a SIMD of gfx950 issues integer code that is not all adds and shifts
    at ~one VALU wave-instruction per 4 cycles.  achieved = VALU wave-instructions per second; peak = 1,024 SIMDs x clock / 4."""
    valu = sq.get("SQ_INSTS_VALU", 0.0)
    allk = sum(v for k, v in sq.items() if k.startswith("SQ_INSTS_"))
    peak = N_SIMD * CLOCK_GHZ * 1e9 / 4.0
    ach = valu / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
    return {"bound": "valu_issue", "achieved": round(ach / 1e9, 2), "peak": round(peak / 1e9, 2), "unit": "G wave-instructions/s",
            "frac": round(ach / peak, 4), "valu_wave_instructions": int(valu), "all_wave_instructions": int(allk),
            "source": sq.get("source")}


if __name__ == "__main__":
    main()
