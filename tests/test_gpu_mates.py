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


def test_region_mode_with_m_is_rejected_loudly():
    r = run_cli(["region", "-m", "-L", "chrM", os.path.join(GOLDEN, "issue225.bam")], check=False)
    assert r.returncode == 1 and b"fix-mate-overlaps" in r.stderr
