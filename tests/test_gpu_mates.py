"""`depth base --fix-mate-overlaps` on the device (K7, mates.hip) against the CPU oracle (literal
restatement of depth.d:319-399,495-556 with column-order tie breaking; no reference golden pins base
mode -m -- SURVEY.md 8c).  Fixture: the reference's own, otherwise unused, mate_overlaps_1_3M_4M.bam."""
import os

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
                want = oracle_base_counters(p, ref, 0, n, min_bq=q, fix_mate=True)
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


def test_three_overlapping_same_name_records_are_rejected(tmp_path):
    refs = [("c1", 2000)]
    raw = [bg.make_record(0, 100 + 10 * i, "60M", "ACGT" * 15, 30, name="trio", mapq=60) for i in range(3)]
    p = str(tmp_path / "trio.bam")
    bg.write_bam(p, refs, raw)
    r = run_cli(["base", "-m", p], check=False)
    assert r.returncode == 1 and b"same name" in r.stderr
    assert run_cli(["base", p]) == run_oracle(["base", p])      # without -m the file is fine


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


def test_region_fix_mate_in_batches(overlapping):
    from tests.test_gpu_batches import cli_batched
    bam, bed = overlapping
    for args in (["region", "-m", "-q", "20", "-T", "3", "-L", bed, bam], ["window", "-m", "-w", "500", bam]):
        assert cli_batched(args, 1) == run_cli(args)
