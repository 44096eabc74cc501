"""`depth base --fix-mate-overlaps` on the device (K7, mates.hip) against the CPU oracle (literal
restatement of depth.d:319-399,495-556 with column-order tie breaking; no reference golden pins base
mode -m -- SURVEY.md 8c).  Fixture: the reference's own, otherwise unused, mate_overlaps_1_3M_4M.bam."""
import os
import random

import numpy as np
import pytest

from tests import bamgen as bg
from tests.util import GOLDEN, gen_bam, oracle_base_counters, run_cli, run_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("args", [
    ["base", "-m", "mate_overlaps_1_3M_4M.bam"],
    ["base", "-m", "-q", "20", "mate_overlaps_1_3M_4M.bam"],
    ["base", "-m", "-q", "30", "-a", "-c", "2", "mate_overlaps_1_3M_4M.bam"],
    ["base", "-m", "-L", "mate_overlaps_1_3M_4M.bed", "mate_overlaps_1_3M_4M.bam"],
    ["base", "-m", "issue_204.bam"],
    ["base", "-m", "-F", "mapping_quality >= 0", "issue_204.bam"],
    ["base", "-m", "issue225.bam"],
])
def test_cli_fix_mate_overlaps_matches_oracle(args):
    assert run_cli(args, cwd=GOLDEN) == run_oracle(args, cwd=GOLDEN)


@pytest.mark.parametrize("extra", [["--tie-free-overlaps"], []])
def test_synthetic_overlapping_mates(tmp_path, extra):
    import sambamba_amd
    p = gen_bam(str(tmp_path / "ov.bam"), "chrA:400000,chrB:100000", coverage=60, seed=31,
                extra=["--insert-mean", "250", "--insert-sd", "40"] + extra)
    for q in (0, 20):
        with sambamba_amd.Depth(p) as d:
            d.set_params(min_bq=q, fix_mate_overlaps=True)
            d.run()
            for ref in (0, 1):
                n = d.ref_lengths[ref]
                got = d.base_counters(ref, 0, n)
                want = oracle_base_counters(p, ref, 0, n, min_bq=q, fix_mate=True, ref_name=d.ref_names[ref])
                assert np.array_equal(got, want), (extra, q, ref)
    assert run_cli(["base", "-m", "-q", "20", p]) == run_oracle(["base", "-m", "-q", "20", p])


def test_mate_rules_on_hand_made_pairs(tmp_path):
    """indel vs base -> mapping quality decides; base vs base -> base quality decides; ties -> later record;
    different samples or different names never pair."""
    refs = [("c1", 3000)]
    rg = lambda g: bg.tag_z("RG", g)
    raw = [
        # pair 1: A has a deletion where B has bases; mapq decides (B higher)
        bg.make_record(0, 100, "30M5D30M", "A" * 60, 30, name="p1", mapq=20, flag=99, tags=rg("g1")),
        bg.make_record(0, 120, "60M", "C" * 60, 35, name="p1", mapq=50, flag=147, tags=rg("g1")),
        # pair 2: equal qualities everywhere -> ties -> the later record wins
        bg.make_record(0, 300, "50M", "G" * 50, 30, name="p2", mapq=60, flag=99, tags=rg("g1")),
        bg.make_record(0, 320, "50M", "T" * 50, 30, name="p2", mapq=60, flag=147, tags=rg("g1")),
        # pair 3: first mate better on some bases, second on others; N-skip inside
        bg.make_record(0, 500, "20M100N20M", "A" * 40, [10] * 20 + [40] * 20, name="p3", mapq=30, flag=99, tags=rg("g1")),
        bg.make_record(0, 510, "40M", "C" * 40, [40] * 10 + [5] * 30, name="p3", mapq=30, flag=147, tags=rg("g1")),
        # same name but different samples: not a pair
        bg.make_record(0, 800, "50M", "A" * 50, 30, name="p4", mapq=60, flag=99, tags=rg("g1")),
        bg.make_record(0, 810, "50M", "C" * 50, 30, name="p4", mapq=60, flag=147, tags=rg("g2")),
        # different names overlapping: not a pair
        bg.make_record(0, 1000, "50M", "A" * 50, 30, name="x1", mapq=60, tags=rg("g1")),
        bg.make_record(0, 1010, "50M", "C" * 50, 30, name="x2", mapq=60, tags=rg("g1")),
        # mates that do not overlap
        bg.make_record(0, 1200, "50M", "A" * 50, 30, name="p5", mapq=60, flag=99, tags=rg("g1")),
        bg.make_record(0, 1300, "50M", "C" * 50, 30, name="p5", mapq=60, flag=147, tags=rg("g1")),
        # one mate filtered out (mapq 0): the other counts alone
        bg.make_record(0, 1500, "50M", "A" * 50, 30, name="p6", mapq=0, flag=99, tags=rg("g1")),
        bg.make_record(0, 1510, "50M", "C" * 50, 30, name="p6", mapq=60, flag=147, tags=rg("g1")),
    ]
    p = str(tmp_path / "pairs.bam")
    bg.write_bam(p, refs, raw, read_groups=[("g1", "s1"), ("g2", "s2")])
    for args in (["base", "-m"], ["base", "-m", "-q", "20"], ["base", "-m", "--combined"], ["base"]):
        assert run_cli(args + [p]) == run_oracle(args + [p]), args


def test_name_groups_of_three_follow_the_reference_loop(tmp_path):
    """Same-name records beyond a pair (supplementary alignments): at a column they pair up in file order and the odd one is
    counted on its own (depth.d:343-377); a record that is paired, alone again and paired a second time is processed on its
    own AND through its pair (status `past`, depth.d:355-371,522-532).  Base mode follows that; region / window mode and four
    or more records over one column are rejected."""
    rng = random.Random(5)
    refs = [("c1", 4000)]
    seq = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    qual = lambda n: [rng.choice([5, 12, 22, 30, 38]) for _ in range(n)]
    raw = []
    def rec(pos, cigar, n, name, mapq=60):
        raw.append((pos, bg.make_record(0, pos, cigar, seq(n), qual(n), name=name, mapq=mapq)))
    # trio, all three overlapping: (1st, 2nd) pair, the third alone, then (2nd, 3rd) once the first has ended
    rec(100, "60M", 60, "trio"); rec(110, "60M", 60, "trio", 40); rec(120, "70M", 70, "trio", 50)
    # chain: X - R - Z, X and Z do not overlap: R is paired, alone, paired again
    rec(400, "50M", 50, "chain"); rec(420, "180M", 180, "chain", 30); rec(500, "150M", 150, "chain", 20)
    # the long record first, two short ones inside it, with an indel and a skip
    rec(900, "40M10D40M200N20M", 100, "nest"); rec(910, "30M", 30, "nest", 10); rec(1000, "60M", 60, "nest", 55)
    # trio where the second record ends first
    rec(1500, "100M", 100, "t2"); rec(1510, "20M", 20, "t2"); rec(1520, "100M", 100, "t2")
    # an ordinary pair and an unrelated read next to them
    rec(2000, "80M", 80, "pair"); rec(2030, "80M", 80, "pair"); rec(2040, "50M", 50, "solo")
    raw.sort(key=lambda t: t[0])
    p = str(tmp_path / "groups.bam")
    bg.write_bam(p, refs, [r for _, r in raw])
    import sambamba_amd
    for q in (0, 20):
        with sambamba_amd.Depth(p) as d:
            d.set_params(min_bq=q, fix_mate_overlaps=True)
            d.run()
            got = d.base_counters(0, 0, 4000)
        want = oracle_base_counters(p, 0, 0, 4000, min_bq=q, fix_mate=True)
        assert np.array_equal(got, want), q
    for args in (["base", "-m"], ["base", "-m", "-q", "20", "-c", "0"]):
        assert run_cli(args + [p]) == run_oracle(args + [p]), args
    # region / window statistics are derived for pairs only
    r = run_cli(["window", "-w", "500", "-m", p], check=False)
    assert r.returncode == 1 and b"same name" in r.stderr
    # four records over one column
    raw4 = [bg.make_record(0, 100 + 10 * i, "60M", "ACGT" * 15, 30, name="quad", mapq=60) for i in range(4)]
    p4 = str(tmp_path / "quad.bam")
    bg.write_bam(p4, refs, raw4)
    r = run_cli(["base", "-m", p4], check=False)
    assert r.returncode == 1 and b"same name" in r.stderr
    assert run_cli(["base", p4]) == run_oracle(["base", p4])      # without -m the file is fine


def test_region_mode_with_m_small_genome():
    args = ["region", "-m", "-L", "chrM", os.path.join(GOLDEN, "issue225.bam")]
    assert run_cli(args) == run_oracle(args)


# ---- region / window with --fix-mate-overlaps (closed form of depth.d:717-845 in reduce.hip) -----------------

def test_reference_golden_region_fix_mate_overlaps():
    """The reference's own golden for `depth region -m` (test/test_suite.sh:159), through the product CLI."""
    out = run_cli(["region", "issue_204.bam", "-L", "2:166868600-166868813", "-T", "15", "-T", "20", "-T", "25", "-m"], cwd=GOLDEN)
    with open(os.path.join(GOLDEN, "issue_204_expected_output.txt"), "rb") as fh:
        assert out == fh.read()


@pytest.mark.parametrize("args", [
    ["region", "-m", "-L", "mate_overlaps_1_3M_4M.bed", "mate_overlaps_1_3M_4M.bam"],
    ["region", "-m", "-q", "20", "-T", "1", "-T", "2", "-T", "3", "-L", "mate_overlaps_1_3M_4M.bed", "mate_overlaps_1_3M_4M.bam"],
    ["region", "-m", "-q", "30", "-T", "10", "-L", "2:166868600-166868813", "issue_204.bam"],
    ["region", "-m", "-F", "mapping_quality >= 0", "-T", "5", "-L", "2:166868700-166868750", "issue_204.bam"],
    ["window", "-m", "-w", "200", "-T", "2", "issue225.bam"],
    ["window", "-m", "-w", "97", "-q", "25", "issue225.bam"],
])
def test_cli_region_window_fix_mate_matches_oracle(args):
    assert run_cli(args, cwd=GOLDEN) == run_oracle(args, cwd=GOLDEN)


@pytest.fixture(scope="module")
def overlapping(tmp_path_factory):
    d = tmp_path_factory.mktemp("ovm")
    bam = gen_bam(str(d / "ov.bam"), "chrA:120000,chrB:40000", coverage=40, seed=77,
                  extra=["--insert-mean", "230", "--insert-sd", "45", "--tie-free-overlaps", "--samples", "2"])
    bed = str(d / "r.bed")
    with open(bed, "w") as fh:     # disjoint, nested, overlapping, abutting and unsorted regions
        fh.write("chrA\t1000\t1400\nchrA\t5000\t5001\nchrA\t1300\t2000\nchrA\t1350\t1360\nchrB\t0\t40000\n"
                 "chrA\t20000\t20150\nchrA\t20150\t20300\nchrA\t60000\t90000\nchrA\t119900\t120000\n")
    return bam, bed


# (--combined together with -m and several samples indexes samples[] with the read's own sample id in the
#  reference, depth.d:731,788: out of bounds -- undefined there, not tested here)
@pytest.mark.parametrize("extra", [[], ["-q", "20"], ["-q", "37", "-T", "1", "-T", "10", "-T", "30"], ["-T", "5", "-a", "-c", "20"]])
def test_synthetic_region_fix_mate(overlapping, extra):
    bam, bed = overlapping
    args = ["region", "-m", "-L", bed] + extra + [bam]
    assert run_cli(args) == run_oracle(args)


@pytest.mark.parametrize("extra", [["-w", "1000"], ["-w", "150", "-q", "20", "-T", "8"], ["-w", "37", "-T", "1", "-T", "40"],
                                   ["-w", "5000", "-q", "13"]])
def test_synthetic_window_fix_mate(overlapping, extra):
    bam, _ = overlapping
    args = ["window", "-m"] + extra + [bam]
    assert run_cli(args) == run_oracle(args)


@pytest.fixture(scope="module")
def overlapping_behind_a_readless_contig(tmp_path_factory):
    # (a first contig too short for a read pair: the first pileup column of the run lies on the SECOND contig, so no window of the
    #  run's first ring -- where is_first_occurrence matters, depth.d:1031-1032 -- is printed; every real genome is like that: the first
    #  reads of chr1 lie ten thousand positions in)
    d = tmp_path_factory.mktemp("ovw")
    return gen_bam(str(d / "ovw.bam"), "c0:250,chrA:90000,chrB:30000", coverage=40, seed=78,
                   extra=["--insert-mean", "230", "--insert-sd", "45", "--tie-free-overlaps", "--samples", "2"])


@pytest.mark.parametrize("extra", [["-w", "1000", "--overlap", "500"], ["-w", "300", "--overlap", "200", "-q", "20", "-T", "8"],
                                   ["-w", "64", "--overlap", "48", "-T", "1", "-T", "40"], ["-w", "5000", "--overlap", "2500", "-q", "13"]])
def test_window_fix_mate_with_overlapping_windows(overlapping_behind_a_readless_contig, extra):
    """`window -m --overlap n` when the step divides the window and the first ring of the run prints nothing: a window is the region
    [k step, k step + w) of the closed form (PerRegionPrinter with mate fixing, depth.d:717-845) -- against the oracle's literal
    PerWindowPrinter."""
    args = ["window", "-m"] + extra + [overlapping_behind_a_readless_contig]
    assert run_cli(args) == run_oracle(args)


def test_window_fix_mate_overlap_refusals(overlapping):
    """What stays refused, loudly: a step that does not divide the window (a ring slot then collects per-column mate terms in front of
    its window), and a run whose first pileup column lies in the first ring of windows of the first contig."""
    bam, _ = overlapping
    r = run_cli(["window", "-m", "-w", "300", "--overlap", "100", bam], check=False)
    assert r.returncode != 0 and b"does not divide" in r.stderr
    r = run_cli(["window", "-m", "-w", "300", "--overlap", "150", bam], check=False)
    assert r.returncode != 0 and b"first ring" in r.stderr
    assert r.stdout.count(b"\n") <= 1          # at most the header line


def test_region_fix_mate_in_batches(overlapping):
    from tests.test_gpu_batches import cli_batched
    bam, bed = overlapping
    for args in (["region", "-m", "-q", "20", "-T", "3", "-L", bed, bam], ["window", "-m", "-w", "500", bam]):
        assert cli_batched(args, 1) == run_cli(args)
