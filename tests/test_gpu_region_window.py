"""`depth region` and `depth window` through the device path (K5 range_reduce / count_reads) and the
sbx-depth CLI: byte-identical text against the CPU oracle (the literal restatement of
sambamba/depth.d:609-1077).  The reference ships no golden for these modes without -m
("parity unpinned by goldens"), so the oracle is the pin."""
import os

import pytest

from tests import bamgen as bg
from tests.util import GOLDEN, gen_bam, run_cli, run_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("args", [
    ["region", "-L", "2:166868600-166868813", "-T", "15", "-T", "20", "-T", "25", "issue_204.bam"],
    ["region", "-L", "2:166868600-166868813", "-T", "0", "-T", "1", "-q", "30", "issue_204.bam"],
    ["region", "-L", "mate_overlaps_1_3M_4M.bed", "-T", "1", "-T", "5", "mate_overlaps_1_3M_4M.bam"],   # unsorted, overlapping BED
    ["region", "-L", "mate_overlaps_1_3M_4M.bed", "-a", "-c", "2", "-C", "40", "mate_overlaps_1_3M_4M.bam"],
    ["region", "-L", "chrM", "-T", "3", "issue225.bam"],
    ["region", "-L", "chrM:1-100", "issue225.bam"],           # region without any column: header only
    ["window", "-w", "100", "issue225.bam"],
    ["window", "-w", "1000", "-T", "1", "-T", "10", "issue225.bam"],
    ["window", "-w", "37", "-q", "20", "-a", "-c", "0.5", "issue_193.bam"],
    ["window", "-w", "500", "--combined", "issue_193.bam"],
    ["window", "-w", "100000", "issue225.bam"],              # window longer than the contig: no rows
])
def test_cli_matches_oracle_on_fixtures(args):
    assert run_cli(args, cwd=GOLDEN) == run_oracle(args, cwd=GOLDEN)


@pytest.fixture(scope="module")
def synth(tmp_path_factory):
    d = tmp_path_factory.mktemp("rw")
    return gen_bam(str(d / "s.bam"), "chrEmpty0:30000,chrA:400000,chrEmpty:25000,chrB:150000,chrTiny:700", coverage=25, seed=21)


@pytest.fixture(scope="module")
def synth_ms(tmp_path_factory):
    d = tmp_path_factory.mktemp("rwms")
    return gen_bam(str(d / "s3.bam"), "c1:120000,c2:60000", coverage=20, seed=22, extra=["--samples", "3"])


def test_window_synthetic(synth, synth_ms):
    for args in (["window", "-w", "1000"], ["window", "-w", "1000", "-T", "10", "-T", "30", "-T", "0"],
                 ["window", "-w", "777", "-q", "13"], ["window", "-w", "5000", "-a", "-c", "24", "-C", "26"]):
        assert run_cli(args + [synth]) == run_oracle(args + [synth]), args
    for args in (["window", "-w", "2000", "-T", "5"], ["window", "-w", "2000", "--combined"]):
        assert run_cli(args + [synth_ms]) == run_oracle(args + [synth_ms]), args


def test_region_synthetic_bed(synth, synth_ms, tmp_path):
    import random
    rng = random.Random(3)
    lines = []
    for i in range(300):
        chrom = rng.choice(["chrA", "chrA", "chrB", "chrTiny", "chrEmpty"])
        n = {"chrA": 400000, "chrB": 150000, "chrTiny": 700, "chrEmpty": 25000}[chrom]
        a = rng.randrange(0, n - 2)
        b = min(n, a + rng.choice([1, 10, 150, 1000, 20000]))
        lines.append("%s\t%d\t%d\tname%d\t%d" % (chrom, a, b, i, rng.randrange(1000)))
    lines.append("chrA 10 20")             # whitespace separated, 3 columns
    lines.append("chrB\t500")              # 2 columns: one base
    bed = tmp_path / "r.bed"
    bed.write_text("\n".join(lines) + "\n")
    for args in (["region", "-L", str(bed)], ["region", "-L", str(bed), "-T", "10", "-T", "30", "-q", "20"],
                 ["region", "-L", str(bed), "-a", "-c", "20"]):
        assert run_cli(args + [synth]) == run_oracle(args + [synth]), args
    bed2 = tmp_path / "m.bed"
    bed2.write_text("c1\t100\t5000\nc2\t0\t60000\nc1\t4000\t4500\n")
    for args in (["region", "-L", str(bed2), "-T", "8"], ["region", "-L", str(bed2), "--combined", "-T", "8"]):
        assert run_cli(args + [synth_ms]) == run_oracle(args + [synth_ms]), args


def test_region_quality_threshold_edge(tmp_path):
    """n_reads counts a read only if it has a base with qual >= q INSIDE the region (depth.d:661-698)."""
    refs = [("c1", 2000)]
    raw = [
        bg.make_record(0, 100, "50M", "A" * 50, [40] * 10 + [5] * 40, name="a"),     # good bases only at 100..109
        bg.make_record(0, 105, "20M5D20M", "C" * 40, [5] * 20 + [40] * 20, name="b"),  # good bases after the deletion
        bg.make_record(0, 120, "10S30M", "G" * 40, [40] * 10 + [5] * 30, name="c"),  # good quals only in the soft clip
    ]
    p = str(tmp_path / "q.bam")
    bg.write_bam(p, refs, raw)
    bed = tmp_path / "q.bed"
    bed.write_text("c1\t100\t110\nc1\t110\t125\nc1\t125\t135\nc1\t130\t160\nc1\t0\t2000\n")
    for q in ("0", "20", "41"):
        args = ["region", "-L", str(bed), "-q", q, "-T", "1", "-T", "2"]
        assert run_cli(args + [p]) == run_oracle(args + [p]), q


def test_window_overlap_small_genome():
    args = ["window", "-w", "100", "--overlap", "10", os.path.join(GOLDEN, "issue225.bam")]
    assert run_cli(args) == run_oracle(args)


def test_contig_shards_reproduce_the_whole_run(synth):
    """Multi-GPU sharding (sambamba_amd/shard.py): a rank restricted to its contigs (sbx_set_regions ->
    only the BAI-listed BGZF block range is inflated) must produce exactly the whole-file numbers for
    the regions / windows / positions it owns."""
    import numpy as np
    import sambamba_amd
    from sambamba_amd.shard import plan_contig_shards, regions_of_shard, owner_of_region
    thr = (5, 20)
    with sambamba_amd.Depth(synth) as d:
        lens = d.ref_lengths
        rng = np.random.default_rng(5)
        bed = []
        for _ in range(200):
            r = int(rng.integers(0, len(lens)))
            a = int(rng.integers(0, max(1, lens[r] - 10)))
            bed.append((r, a, min(lens[r], a + int(rng.choice([5, 100, 3000])))))
        d.set_params(mode=sambamba_amd.SBX_MODE_REGION, thresholds=thr)
        whole_stats = d.run()
        w_reads, w_bases, w_cov, w_seen = d.region_stats(bed, len(thr))
        w_covered = {r: d.covered(r, 0, lens[r]) for r in range(len(lens))}
    with sambamba_amd.Depth(synth) as d:         # (the seven counters per position are what a `base` run keeps)
        d.set_params()
        d.run()
        w_counts = {r: d.base_counters(r, 0, lens[r]) for r in range(len(lens))}
    for world in (2, 3):
        shards = plan_contig_shards(lens, world)
        for rank, sh in enumerate(shards):
            regs = regions_of_shard(lens, sh)
            if not regs:
                continue
            with sambamba_amd.Depth(synth) as d:
                d.set_params(mode=sambamba_amd.SBX_MODE_REGION, thresholds=thr)
                d.set_regions(regs)
                st = d.run()
                assert st["uncompressed_bytes"] <= whole_stats["uncompressed_bytes"]
                mine = [i for i, b in enumerate(bed) if owner_of_region(shards, b[0]) == rank]
                reads, bases, cov, seen = d.region_stats([bed[i] for i in mine], len(thr))
                assert np.array_equal(reads, w_reads[mine]) and np.array_equal(bases, w_bases[mine])
                assert np.array_equal(cov, w_cov[mine]) and np.array_equal(seen, w_seen[mine])
                for r in range(sh[0], sh[1]):
                    assert np.array_equal(d.covered(r, 0, lens[r]), w_covered[r])
            with sambamba_amd.Depth(synth) as d:
                d.set_params()
                d.set_regions(regs)
                d.run()
                for r in range(sh[0], sh[1]):
                    assert np.array_equal(d.base_counters(r, 0, lens[r]), w_counts[r])
