"""Known answers the reference itself holds for the pileup, beyond the `depth` goldens:

* the pileup unittest of BioD/bio/std/hts/bam/pileup.d:698-857 -- ten reads of NA20828 (20:1127810-1127819) with I / D / S
  operations and the bases the reference expects per column (`column.bases`, asserted there for columns 796, 805, 806, 821,
  826 and 849), and four reads with a stretch of zero coverage between them;
* `Base5` of BioD/bio/core/base.d:163-186 -- `nt16_to_nt5`, the table that maps the sixteen 4-bit codes of a BAM sequence to
  A, C, G, T or "other", which is what depth.d's per-base counters are indexed by.

The reads are rebuilt as a BAM with tests/bamgen.py (the unittest builds them in memory; this file restates its DATA, not
its code).  CPU tests pin the oracle on these answers; the `gpu` tests pin the product -- C ABI and CLI -- on the same.
"""
import numpy as np
import pytest

from tests import bamgen as bg
from tests.util import oracle_base_counters, run_cli, run_oracle

A, C, G, T, OTHER, DEL, REFSKIP = range(7)

# pileup.d:706-745: names r0 .. r9, sequences, CIGARs, 0-based positions
SEQS = ["ATTATGGACATTGTTTCCGTTATCATCATCATCATCATCATCATCATTATCATC",
        "GACATTGTTTCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATC",
        "ATTGTTTCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATCACC",
        "TGTTTCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATCACCAC",
        "TCCGTTATCATCATCATCATCATCATCATCATCATCATCATCATCACCACCACC",
        "GTTATCATCATCATCATCATCATCATCATCATCATCATCATCATCGTCACCCTG",
        "TCATCATCATCATAATCATCATCATCATCATCATCATCGTCACCCTGTGTTGAG",
        "TCATCATCATCGTCACCCTGTGTTGAGGACAGAAGTAATTTCCCTTTCTTGGCT",
        "TCATCATCATCATCACCACCACCACCCTGTGTTGAGGACAGAAGTAATATCCCT",
        "CACCACCACCCTGTGTTGAGGACAGAAGTAATTTCCCTTTCTTGGCTGGTCACC"]
CIGARS = ["54M", "54M", "50M3I1M", "54M", "54M", "54M", "2S52M", "16M15D38M", "13M3I38M", "54M"]
POSITIONS = [758, 764, 767, 769, 773, 776, 785, 795, 804, 817]
# pileup.d:790-828: `column.bases` per asserted column, one character per read of the column ('-' = deletion)
EXPECTED_BASES = {796: "CCCCCCAC", 805: "TCCCCCCCC", 806: "AAAAAAAGA", 821: "AAGG-AA", 826: "CCCCCC", 849: "TAT"}

# pileup.d:834-857: the second read set -- positions 1039 .. 1045 are covered by no read
SEQS2 = ["CCCACATAGAAAGCTTGCTGTTTCTCTGTGGGAAGTTTTAACTTAGGTCAGCTT",
         "TAGAAAGCTTGCTGTTTCTCTGTGGGAAGTTTTAACTTAGGTTAGCTTCATCTA",
         "TTTTTCTTTCTTTCTTTGAAGAAGGCAGATTCCTGGTCCTGCCACTCAAATTTT",
         "TTTCTTTCTTTCTTTGAAGAAGGCAGATTCCTGGTCCTGCCACTCAAATTTTCA"]
POSITIONS2 = [979, 985, 1046, 1048]

REFS = [("20", 2000)]


def _bam(tmp_path, second=False):
    recs = []
    if not second:
        for i, (s, c, p) in enumerate(zip(SEQS, CIGARS, POSITIONS)):
            recs.append(bg.make_record(0, p, c, s, 30, name="r%d" % i))
    else:
        for i, (s, p) in enumerate(zip(SEQS2, POSITIONS2)):
            recs.append(bg.make_record(0, p, "54M", s, 30, name="r%d" % (i + 1)))
    path = str(tmp_path / ("pileup_unittest%d.bam" % (2 if second else 1)))
    bg.write_bam(path, REFS, recs)
    return path


def _want(bases):
    w = np.zeros(7, dtype=np.uint32)
    for ch in bases:
        w[{"A": A, "C": C, "G": G, "T": T, "-": DEL}[ch]] += 1
    return w


def _check_first_set(counters):
    """counters: u32[2000][1][7] of contig 20"""
    for pos, bases in EXPECTED_BASES.items():
        assert np.array_equal(counters[pos, 0], _want(bases)), (pos, bases, counters[pos, 0])
    # pileup.d:803-812: column 810 holds the read with the deletion in front of it, 817 the one with the insertion behind it --
    # what the counters can show of that is the coverage: reads r1 .. r8 at 810 (r0 ended at 811, r9 starts at 817)
    assert counters[810, 0].sum() == 9 and counters[810, 0, DEL] == 0
    # the deletion of r7 spans 811 .. 825
    assert [int(counters[p, 0, DEL]) for p in (810, 811, 825, 826)] == [0, 1, 1, 0]


def _rows(text):
    rows = {}
    for line in text.decode().splitlines()[1:]:
        f = line.split("\t")
        rows[int(f[1])] = [int(x) for x in f[2:9]]      # POS -> COV A C G T DEL REFSKIP
    return rows


def _check_text_first_set(text):
    rows = _rows(text)
    for pos, bases in EXPECTED_BASES.items():
        w = _want(bases)
        # depth.d:534-555 prints 0-based positions; COV is the column's coverage, a read with a deletion there included
        assert rows[pos] == [int(w.sum()), int(w[A]), int(w[C]), int(w[G]), int(w[T]), int(w[DEL]), 0], (pos, rows[pos])


def test_oracle_reproduces_the_pileup_unittest_columns(tmp_path):
    bam = _bam(tmp_path)
    _check_first_set(oracle_base_counters(bam, 0, 0, 2000))
    _check_text_first_set(run_oracle(["base", bam]))


def test_oracle_leaves_the_zero_coverage_stretch_of_the_second_read_set_empty(tmp_path):
    bam = _bam(tmp_path, second=True)
    c = oracle_base_counters(bam, 0, 0, 2000)
    cov = c[:, 0, :].sum(axis=1)
    covered = set(np.nonzero(cov)[0].tolist())
    assert covered == set(range(979, 1039)) | set(range(1046, 1102))
    rows = _rows(run_oracle(["base", bam]))
    assert set(rows) == covered            # -c 1, the default: no row where no read is
    rows0 = _rows(run_oracle(["base", "-c", "0", bam]))
    assert all(rows0[p][0] == 0 for p in range(1039, 1046))


# base.d:185: nt16_to_nt5 for the codes "=ACMGRSVTWYHKDBN"
NT16 = "=ACMGRSVTWYHKDBN"
NT16_TO_NT5 = [4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4]


def _bam16(tmp_path):
    # one read holding every 4-bit code once, a second one holding them in reverse: every code at an even and at an odd offset
    recs = [bg.make_record(0, 100, "16M", NT16, 30, name="a"), bg.make_record(0, 200, "16M", NT16[::-1], 30, name="b")]
    path = str(tmp_path / "nt16.bam")
    bg.write_bam(path, [("c", 1000)], recs)
    return path


def _check_nt16(counters):
    for i, code in enumerate(NT16_TO_NT5):
        for pos in (100 + i, 200 + 15 - i):
            w = np.zeros(7, dtype=np.uint32)
            w[code] = 1
            assert np.array_equal(counters[pos, 0], w), (NT16[i], pos, counters[pos, 0])


def test_oracle_maps_the_sixteen_sequence_codes_like_base5(tmp_path):
    assert bg.SEQ_CODES == {c: i for i, c in enumerate(NT16)}
    _check_nt16(oracle_base_counters(_bam16(tmp_path), 0, 0, 1000))


@pytest.mark.gpu
def test_device_reproduces_the_pileup_unittest_columns(tmp_path):
    import sambamba_amd
    bam = _bam(tmp_path)
    with sambamba_amd.Depth(bam) as d:
        d.set_params()
        d.run()
        got = d.base_counters(0, 0, 2000)
    _check_first_set(got)
    assert np.array_equal(got, oracle_base_counters(bam, 0, 0, 2000))
    text = run_cli(["base", bam])
    _check_text_first_set(text)
    assert text == run_oracle(["base", bam])


@pytest.mark.gpu
def test_device_leaves_the_zero_coverage_stretch_empty(tmp_path):
    bam = _bam(tmp_path, second=True)
    rows = _rows(run_cli(["base", bam]))
    assert set(rows) == set(range(979, 1039)) | set(range(1046, 1102))
    for args in (["base"], ["base", "-c", "0"]):
        assert run_cli(args + [bam]) == run_oracle(args + [bam])


@pytest.mark.gpu
def test_device_maps_the_sixteen_sequence_codes_like_base5(tmp_path):
    import sambamba_amd
    bam = _bam16(tmp_path)
    with sambamba_amd.Depth(bam) as d:
        d.set_params()
        d.run()
        _check_nt16(d.base_counters(0, 0, 1000))
