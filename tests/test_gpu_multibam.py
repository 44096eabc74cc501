"""Several BAM files in one run (MultiBamReader semantics that matter for depth): the
product on files A, B, C must print what the oracle prints on ONE BAM holding the same reads, merged by coordinate
with the union of the read groups -- depth only sees the multiset of reads and their samples."""
import numpy as np
import pytest

from tests import bamgen as bg
from tests.util import oracle_base_counters, run_cli, run_oracle

pytestmark = pytest.mark.gpu

REFS = [("c1", 30000), ("cEmpty", 2000), ("c2", 9000)]


def make_reads(seed, n, rg, long_cigars=True):
    rng = np.random.RandomState(seed)
    recs = []
    for k in range(n):
        ref = 0 if rng.rand() < 0.75 else 2
        L = REFS[ref][1]
        pos = int(rng.randint(0, L - 400))
        kind = rng.randint(0, 4) if long_cigars else 0
        cigar = ["100M", "30M4D70M", "5S60M200N35M", "50M3I47M"][kind]
        seq = "".join("ACGTN"[i] for i in rng.randint(0, 5, size=100))
        qual = [int(q) for q in rng.randint(2, 41, size=100)]
        flag = int(rng.choice([0, 16, 99, 147, 1024, 512]))
        recs.append((ref, pos, bg.make_record(ref, pos, cigar, seq, qual, name="r%d_%d" % (seed, k), mapq=int(rng.choice([0, 20, 60])),
                                              flag=flag, tags=bg.tag_z("RG", rg) if rg else b"")))
    recs.sort(key=lambda t: (t[0], t[1]))
    return recs


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("multibam")
    parts = [("a.bam", make_reads(1, 1500, "ga"), [("ga", "S1")]),
             ("b.bam", make_reads(2, 1200, "gb"), [("gb", "S2")]),
             ("c.bam", make_reads(3, 900, "ga"), [("ga", "S1")])]      # same RG id and sample as file a: coverage adds up
    paths = []
    for name, recs, rgs in parts:
        p = str(d / name)
        bg.write_bam(p, REFS, [r[2] for r in recs], read_groups=rgs)
        paths.append(p)
    # the merged stream: stable by (ref, pos), files in command-line order
    allr = []
    for fi, (_, recs, _) in enumerate(parts):
        allr += [(r[0], r[1], fi, r[2]) for r in recs]
    allr.sort(key=lambda t: (t[0], t[1], t[2]))
    merged = str(d / "merged.bam")
    bg.write_bam(merged, REFS, [r[3] for r in allr], read_groups=[("ga", "S1"), ("gb", "S2")])
    bed = str(d / "r.bed")
    with open(bed, "w") as fh:
        fh.write("c1\t100\t4000\nc2\t0\t9000\nc1\t3500\t3600\ncEmpty\t5\t50\nc1\t25000\t29999\n")
    return paths, merged, bed


@pytest.mark.parametrize("args", [
    ["base"], ["base", "-c", "0"], ["base", "-q", "25", "-a", "-c", "3"], ["base", "--combined", "-c", "2"],
    ["base", "-F", "mapping_quality >= 20 and not duplicate"],
    ["window", "-w", "500", "-T", "2", "-T", "6"], ["window", "-w", "333", "-q", "30", "--combined"],
    ["region", "-L", "BED", "-T", "3"], ["region", "-L", "BED", "-q", "20", "--combined", "-T", "1", "-T", "9"],
    ["base", "-L", "BED"],
])
def test_multibam_equals_merged_single_bam(files, args):
    paths, merged, bed = files
    a = [bed if x == "BED" else x for x in args]
    assert run_cli(a + paths) == run_oracle(a + [merged])


def test_multibam_api_counters_and_header(files):
    import sambamba_amd
    paths, merged, _ = files
    with sambamba_amd.Depth(paths) as d:
        assert d.sample_names == ["S1", "S2"]
        d.set_params(min_bq=10)
        st = d.run()
        assert st["n_records"] == 1500 + 1200 + 900
        for ref in range(3):
            got = d.base_counters(ref, 0, REFS[ref][1])
            want = oracle_base_counters(merged, ref, 0, REFS[ref][1], n_samples=2, min_bq=10)
            assert np.array_equal(got, want), ref


def test_multibam_batched(files):
    from tests.test_gpu_batches import cli_batched
    paths, _, bed = files
    for a in (["base", "-c", "0"], ["window", "-w", "500", "-T", "4"], ["region", "-L", bed, "-T", "2"]):
        assert cli_batched(a + paths, 1) == run_cli(a + paths)


def make_reads_on(refs, seed, n, rg):
    """like make_reads, on an arbitrary dictionary: (name of the contig, pos, record with the FILE's reference id)"""
    rng = np.random.RandomState(seed)
    recs = []
    usable = [i for i, (_, L) in enumerate(refs) if L > 500]
    for k in range(n):
        ref = int(rng.choice(usable))
        pos = int(rng.randint(0, refs[ref][1] - 400))
        cigar = ["100M", "30M4D70M", "5S60M200N35M", "50M3I47M"][rng.randint(0, 4)]
        seq = "".join("ACGTN"[i] for i in rng.randint(0, 5, size=100))
        qual = [int(q) for q in rng.randint(2, 41, size=100)]
        recs.append((ref, pos, bg.make_record(ref, pos, cigar, seq, qual, name="d%d_%d" % (seed, k), mapq=int(rng.choice([0, 20, 60])),
                                              flag=int(rng.choice([0, 16, 99, 147])), tags=bg.tag_z("RG", rg))))
    recs.sort(key=lambda t: (t[0], t[1]))
    return recs


def renumber(rec, new_ref):
    """the same record with another reference id (bytes 4..8 behind block_size) -- what adjustTagsInRange does to a read"""
    import struct
    return rec[:4] + struct.pack("<i", new_ref) + rec[8:]


@pytest.fixture(scope="module")
def mixed_dictionaries(tmp_path_factory):
    """Three files whose @SQ dictionaries differ but can be merged (multireader.d:218-236, samheadermerger.d:127-177): a contig
    missing in the middle, an extra contig at the end, a file that starts further down.  Merged: c1, cEmpty, c2, cNew."""
    d = tmp_path_factory.mktemp("mixed")
    dict_a = [("c1", 30000), ("cEmpty", 2000), ("c2", 9000)]
    dict_b = [("c1", 30000), ("c2", 9000), ("cNew", 5000)]
    dict_c = [("cEmpty", 2000), ("c2", 9000)]
    merged_dict = [("c1", 30000), ("cEmpty", 2000), ("c2", 9000), ("cNew", 5000)]
    parts = [("a.bam", dict_a, make_reads_on(dict_a, 11, 1200, "ga"), [("ga", "S1")]),
             ("b.bam", dict_b, make_reads_on(dict_b, 12, 1000, "gb"), [("gb", "S2")]),
             ("c.bam", dict_c, make_reads_on(dict_c, 13, 500, "ga"), [("ga", "S1")])]
    paths, allr = [], []
    for fi, (name, dic, recs, rgs) in enumerate(parts):
        p = str(d / name)
        bg.write_bam(p, dic, [r[2] for r in recs], read_groups=rgs)
        paths.append(p)
        for ref, pos, rec in recs:
            new = [n for n, _ in merged_dict].index(dic[ref][0])
            allr.append((new, pos, fi, renumber(rec, new)))
    allr.sort(key=lambda t: (t[0], t[1], t[2]))
    merged = str(d / "merged.bam")
    bg.write_bam(merged, merged_dict, [r[3] for r in allr], read_groups=[("ga", "S1"), ("gb", "S2")])
    bed = str(d / "r.bed")
    with open(bed, "w") as fh:
        fh.write("c1\t100\t4000\nc2\t0\t9000\ncNew\t1000\t3000\ncEmpty\t5\t50\nc1\t25000\t29999\n")
    return paths, merged, bed


@pytest.mark.parametrize("args", [
    ["base"], ["base", "-c", "0"], ["base", "-q", "25", "--combined", "-c", "2"],
    ["base", "-F", "mapping_quality >= 20 and ref_name =~ /^c[2N]/"], ["base", "-F", "ref_id == 3 or ref_id == 0"],
    ["window", "-w", "500", "-T", "2", "-T", "6"],
    ["region", "-L", "BED", "-T", "3"], ["base", "-L", "BED"], ["base", "-L", "cNew:100-2000"], ["base", "-L", "c2"],
    # (cNew has reads of sample S2 only: without --combined or -c 0 the reference prints nothing there -- its loop over the samples
    # returns at the first one below min_coverage, depth.d:540-541)
    ["base", "-L", "cNew:100-2000", "--combined"], ["base", "-L", "cNew", "-c", "0"],
])
def test_compatible_dictionaries_are_merged(mixed_dictionaries, args):
    paths, merged, bed = mixed_dictionaries
    a = [bed if x == "BED" else x for x in args]
    assert run_cli(a + paths) == run_oracle(a + [merged])
    if args == ["base"]:
        assert run_cli(a + paths[::-1]) != b""          # another order of the files: another (valid) merged order, it must run


def test_compatible_dictionaries_batched_and_api(mixed_dictionaries):
    import sambamba_amd
    from tests.test_gpu_batches import cli_batched
    paths, merged, bed = mixed_dictionaries
    for a in (["base", "-c", "0"], ["region", "-L", bed, "-T", "2"]):
        assert cli_batched(a + paths, 1) == run_cli(a + paths)
    with sambamba_amd.Depth(paths) as d:
        assert [n for n in d.ref_names] == ["c1", "cEmpty", "c2", "cNew"]
        d.set_params(min_bq=10)
        assert d.run()["n_records"] == 2700
        got = d.base_counters(3, 0, 5000)
        want = oracle_base_counters(merged, 3, 0, 5000, n_samples=2, min_bq=10)
        assert np.array_equal(got, want)


def test_dictionaries_that_cannot_be_merged_are_rejected(files, tmp_path):
    """What SamHeaderMerger refuses: one name with two lengths (samheadermerger.d:110-125); orders that contradict each other send
    it to a strategy MultiBamReader does not implement (multireader.d:226: "NYI")."""
    paths, _, _ = files
    other = str(tmp_path / "o.bam")
    bg.write_bam(other, [("c1", 30001), ("cX", 10)], [bg.make_record(0, 5, "10M", "ACGTACGTAC", 30)])
    r = run_cli(["base", paths[0], other], check=False)
    assert r.returncode != 0 and b"can't merge SAM headers: one of references with name c1 has length 30000" in r.stderr
    swapped = str(tmp_path / "s.bam")
    bg.write_bam(swapped, [("c2", 9000), ("c1", 30000)], [bg.make_record(0, 5, "10M", "ACGTACGTAC", 30)])
    r = run_cli(["base", paths[0], swapped], check=False)
    assert r.returncode != 0 and b"NYI" in r.stderr


# ---- -m with several files: the reference pairs across files (multireader.d:265-268 merges the streams before depth.d:338-377 sorts a
# column's reads by name hash); the engine pairs within a file and must refuse when that is not the same thing --------------------------
MREFS = [("c1", 6000)]


def _pairs(seed, n, rg, prefix):
    """n overlapping pairs: (pos, record) of both mates, same name"""
    rng = np.random.RandomState(seed)
    out = []
    for k in range(n):
        p1 = int(rng.randint(0, 5000))
        p2 = p1 + int(rng.randint(20, 90))
        for pos, flag in ((p1, 99), (p2, 147)):
            seq = "".join("ACGT"[i] for i in rng.randint(0, 4, size=100))
            qual = [int(q) for q in rng.randint(5, 41, size=100)]
            out.append((pos, bg.make_record(0, pos, "100M", seq, qual, name="%s%d" % (prefix, k), mapq=int(rng.choice([20, 60])), flag=flag,
                                            tags=bg.tag_z("RG", rg))))
    return out


def _write_sorted(path, recs, rgs):
    recs = sorted(recs, key=lambda t: t[0])
    bg.write_bam(path, MREFS, [r[1] for r in recs], read_groups=rgs)


def _merged(path, per_file, rgs):
    allr = []
    for fi, recs in enumerate(per_file):
        allr += [(r[0], fi, k, r[1]) for k, r in enumerate(sorted(recs, key=lambda t: t[0]))]
    allr.sort(key=lambda t: (t[0], t[1], t[2]))
    bg.write_bam(path, MREFS, [r[3] for r in allr], read_groups=rgs)


M_ARGS = [["base", "-m"], ["base", "-m", "-q", "20", "-c", "0"], ["window", "-w", "400", "-m", "-T", "2"],
          ["region", "-L", "c1:500-5500", "-m", "-T", "1", "-T", "3"]]


@pytest.mark.parametrize("args", M_ARGS)
def test_fix_mate_overlaps_with_pairs_inside_each_file(tmp_path, args):
    ra, rb = _pairs(11, 150, "ga", "a"), _pairs(12, 120, "ga", "b")         # same sample in both files, names differ
    a, b, m = str(tmp_path / "a.bam"), str(tmp_path / "b.bam"), str(tmp_path / "m.bam")
    _write_sorted(a, ra, [("ga", "S1")])
    _write_sorted(b, rb, [("ga", "S1")])
    _merged(m, [ra, rb], [("ga", "S1")])
    assert run_cli(args + [a, b]) == run_oracle(args + [m])


@pytest.mark.parametrize("args", M_ARGS)
def test_a_pair_split_over_two_files_is_refused_not_printed_differently(tmp_path, args):
    ra, rb = _pairs(21, 40, "ga", "a"), _pairs(22, 40, "ga", "b")
    split = _pairs(23, 1, "ga", "split")                                      # its first mate goes to file a, its second to file b
    a, b, m = str(tmp_path / "a.bam"), str(tmp_path / "b.bam"), str(tmp_path / "m.bam")
    _write_sorted(a, ra + split[:1], [("ga", "S1")])
    _write_sorted(b, rb + split[1:], [("ga", "S1")])
    _merged(m, [ra + split[:1], rb + split[1:]], [("ga", "S1")])
    r = run_cli(args + [a, b], check=False)
    # (the header line is out before the files are opened, as in the reference: depth.d:1152)
    assert r.returncode != 0 and b"different files" in r.stderr and r.stdout.count(b"\n") <= 1
    # without -m the same files are fine, and the oracle on the merged stream does pair the two (the outputs differ)
    plain = [x for x in args if x != "-m"]
    assert run_cli(plain + [a, b]) == run_oracle(plain + [m])
    assert run_oracle(args + [m]) != run_oracle(plain + [m])


def test_same_names_in_two_files_that_the_reference_would_not_pair(tmp_path):
    # (1) different samples: depth.d:352 compares sample ids; (2) same sample, but the two records do not overlap
    ra, rb = _pairs(31, 30, "ga", "a"), _pairs(32, 30, "gb", "b")
    s1 = _pairs(33, 1, "ga", "split")[:1] + []
    rng = np.random.RandomState(5)
    seq = "".join("ACGT"[i] for i in rng.randint(0, 4, size=100))
    s2 = [(s1[0][0] + 30, bg.make_record(0, s1[0][0] + 30, "100M", seq, 30, name="split0", flag=147, tags=bg.tag_z("RG", "gb")))]
    far = [(100, bg.make_record(0, 100, "100M", seq, 30, name="far", flag=99, tags=bg.tag_z("RG", "ga")))]
    far2 = [(900, bg.make_record(0, 900, "100M", seq, 30, name="far", flag=147, tags=bg.tag_z("RG", "ga")))]
    a, b, m = str(tmp_path / "a.bam"), str(tmp_path / "b.bam"), str(tmp_path / "m.bam")
    rgs = [("ga", "S1"), ("gb", "S2")]
    _write_sorted(a, ra + s1 + far, rgs)
    _write_sorted(b, rb + s2 + far2, rgs)
    _merged(m, [ra + s1 + far, rb + s2 + far2], rgs)
    for args in (["base", "-m"], ["window", "-w", "400", "-m"]):
        assert run_cli(args + [a, b]) == run_oracle(args + [m]), args
