"""-F filters on the device (K2 eval_filter): integer tag comparisons and tag existence
(IntegerTagFilter / TagExistenceFilter, filtering.d:216-252) against the oracle, on hand-made records that
put the tag behind every kind of aux field, and on the reference's fixtures."""
import os

import pytest

from tests import bamgen as bg
from tests.util import GOLDEN, run_cli, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tagged(tmp_path_factory):
    d = tmp_path_factory.mktemp("tags")
    path = str(d / "t.bam")
    recs = []
    seq = "ACGT" * 10
    variants = [
        bg.tag_num("NM", "c", -3), bg.tag_num("NM", "C", 200), bg.tag_num("NM", "s", -300), bg.tag_num("NM", "S", 40000),
        bg.tag_num("NM", "i", -70000), bg.tag_num("NM", "I", 3000000000), bg.tag_num("NM", "f", 2.5), bg.tag_num("NM", "f", 3.0),
        bg.tag_num("NM", "A", "x"), bg.tag_z("NM", "7"), b"",
        bg.tag_z("RG", "g1") + bg.tag_bytes("ZB", range(7)) + bg.tag_num("XS", "A", "q") + bg.tag_num("NM", "C", 3),
        bg.tag_num("XN", "i", 5) + bg.tag_num("NM", "i", 3) + bg.tag_num("NM", "i", 99),       # the first NM counts
        bg.tag_z("MD", "40") + bg.tag_num("AS", "i", 37) + bg.tag_num("NM", "C", 0),
        bg.tag_num("AS", "f", 36.5), bg.tag_num("nm", "C", 1),
    ]
    pos = 100
    for k in range(6):
        for i, t in enumerate(variants):
            recs.append(bg.make_record(0, pos, "40M", seq, 30, name="r%d_%d" % (k, i), tags=t))
            pos += 3
    bg.write_bam(path, [("c1", 5000)], recs, read_groups=[("g1", "s1")])
    return path


@pytest.mark.parametrize("flt", [
    "[NM] > 2", "[NM] >= 3", "[NM] < 0", "[NM] <= -300", "[NM] == 3", "[NM] != 3", "[NM] > 2999999999",
    "[NM] == null", "[NM] != null", "[XS] != null", "[XS] == null and [NM] >= 0",
    "not ([NM] > 2) and mapping_quality >= 0", "[AS] >= 37 or [NM] == 200", "[AS] < 37", "[nm] == 1", "[ZZ] > 0",
])
def test_tag_filters_synthetic(tagged, flt):
    args = ["base", "-F", flt, tagged]
    assert run_cli(args) == run_oracle(args)


@pytest.mark.parametrize("flt", ["[NM] <= 1", "[NM] > 1 and mapping_quality >= 30", "[XS] == null", "[AS] > 90 and not duplicate",
                                 "[MD] != null and [NM] == 0"])
@pytest.mark.parametrize("bam", ["issue_204.bam", "mate_overlaps_1_3M_4M.bam"])
def test_tag_filters_reference_fixtures(flt, bam):
    args = ["base", "-F", flt, bam]
    a, b = run_cli(args, cwd=GOLDEN), run_oracle(args, cwd=GOLDEN)
    assert a == b


@pytest.mark.parametrize("flt", [
    "[RG] == 'g1'", "[RG] != 'g1'", "[NM] == '7'", "[NM] >= '7'", "[XS] == 'q'", "[XS] < 'r'", "[XS] == 'qq'", "[MD] > '3'",
    "read_name == 'r2_5'", "read_name > 'r3'", "read_name <= 'r1_9' and [NM] != null", "read_name != 'it\\'s'",
    "ref_name == 'c1'", "ref_name != 'c1'", "ref_name == 'nope'", "ref_name != 'nope'", "mate_ref_name == '*'", "mate_ref_name != '*'",
    "strand == '+'", "strand == '-'", "strand != '+'", "strand == 'x'", "not (strand == '-' or [RG] == 'g1')",
    "cigar == '40M'", "cigar != '40M'", "cigar > '4'", "cigar < '40M1'", "sequence == 'ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT'",
    "sequence > 'ACGTACGTAC'", "sequence < 'ACGU'", "sequence != ''",
])
def test_string_filters_synthetic(tagged, flt):
    args = ["base", "-F", flt, tagged]
    assert run_cli(args) == run_oracle(args)


@pytest.mark.parametrize("flt", ["[RG] != null and ref_name == '2'", "strand == '-' and [MD] >= '5'", "mate_ref_name == '2' and read_name > 'H'",
                                 "cigar == '101M' or cigar > '5'", "sequence >= 'G' and cigar != '101M'"])
def test_string_filters_reference_fixture(flt):
    args = ["base", "-F", flt, "issue_204.bam"]
    assert run_cli(args, cwd=GOLDEN) == run_oracle(args, cwd=GOLDEN)


@pytest.mark.parametrize("flt", ["avg_base_quality >= 30", "avg_base_quality < 37 and mapping_quality > 10", "avg_base_quality == 30"])
@pytest.mark.parametrize("bam", ["issue_204.bam", "issue225.bam"])
def test_avg_base_quality(flt, bam):
    args = ["base", "-F", flt, bam]
    assert run_cli(args, cwd=GOLDEN) == run_oracle(args, cwd=GOLDEN)


@pytest.mark.parametrize("flt", [
    "read_name =~ /^r1_/", "read_name =~ /_1[0-5]$/", "not (read_name =~ /^r[0-2]/)", "read_name =~ /R3_/i", "[RG] =~ /^g/",
    "[NM] =~ /7/", "[XS] =~ /q/", "[ZZ] =~ /./", "cigar =~ /^40M$/", "cigar =~ /S/", "sequence =~ /^(ACGT)+$/", "sequence =~ /TT/",
    "ref_name =~ /^c[0-9]$/", "ref_name =~ /x/", "mate_ref_name =~ /^\\*$/", "read_name =~ /^r[0-9]+_(3|5)$/ and [NM] != null",
])
def test_regex_filters_synthetic(tagged, flt):
    args = ["base", "-F", flt, tagged]
    assert run_cli(args) == run_oracle(args)


@pytest.mark.parametrize("flt", ["read_name =~ /:1[0-9]{3}:/", "cigar =~ /[IDS]/ and ref_name =~ /^[0-9]+$/", "[MD] =~ /^[0-9]+$/",
                                 "sequence =~ /^[ACGT]+$/i and not (cigar =~ /^101M$/)"])
def test_regex_filters_reference_fixture(flt):
    args = ["base", "-F", flt, "issue_204.bam"]
    assert run_cli(args, cwd=GOLDEN) == run_oracle(args, cwd=GOLDEN)


def test_unsupported_regex_is_reported(tagged):
    r = run_cli(["base", "-F", "read_name =~ /(r)\\1/", tagged], check=False)
    assert r.returncode != 0 and b"back-references" in r.stderr


@pytest.mark.parametrize("flt", [None, "mapping_quality >= 30 and not duplicate", "proper_pair and not (secondary_alignment or supplementary)",
                                 "first_of_pair or mate_is_reverse_strand", "template_length > 100 and position < 3500000"])
def test_simple_evaluator_and_interpreter_agree(flt):
    """-F programs of flag tests, integer fields and and / or / not run through `eval_filter_simple` in a describe kernel of their own
    (index.hip: k_describe_blocks_simple); SBX_K2_SIMPLE_FILTER=0 sends the same program through the interpreter.  Same text either way,
    and the oracle's."""
    args = ["base"] + (["-F", flt] if flt else []) + ["mate_overlaps_1_3M_4M.bam"]
    simple = run_cli(args, cwd=GOLDEN)
    assert simple == run_cli(args, cwd=GOLDEN, env={"SBX_K2_SIMPLE_FILTER": "0"})
    assert simple == run_oracle(args, cwd=GOLDEN)
