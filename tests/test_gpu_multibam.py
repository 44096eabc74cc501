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


def test_different_reference_dictionaries_are_rejected(files, tmp_path):
    paths, _, _ = files
    other = str(tmp_path / "o.bam")
    bg.write_bam(other, [("c1", 30000), ("cX", 10)], [bg.make_record(0, 5, "10M", "ACGTACGTAC", 30)])
    r = run_cli(["base", paths[0], other], check=False)
    assert r.returncode != 0 and b"reference dictionaries" in r.stderr
