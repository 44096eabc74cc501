"""`sbx-depth --gpus N`: ONE process, one context per device, the job sharded by position behind the C ABI (cli.cpp `Sharded`,
sbx_plan_shards).  The test box has one GPU, so the contexts share it (SBX_DEVICES=0,0,0 -- the code path is the multi-device
one: N sbx_open calls with explicit ordinals, N threads, slices dealt to the contexts); the output must be byte for byte what one
context prints -- which the rest of the suite holds against the oracle.  No multi-GPU node has been available to the builder:
what these tests establish is correctness of the sharded path, not a scaling curve."""
import os

import pytest

from tests.util import gen_bam, run_cli, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_contig(tmp_path_factory):
    """ONE contig (BASELINE configs 2 / 5): shards only by position; most mates overlap, some straddle every cut."""
    d = tmp_path_factory.mktemp("shard1")
    return gen_bam(str(d / "one.bam"), "chrOne:200000", coverage=40, seed=43,
                   extra=["--insert-mean", "250", "--insert-sd", "40", "--tie-free-overlaps"])


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    d = tmp_path_factory.mktemp("shardg")
    bam = gen_bam(str(d / "g.bam"), "c1:90000,c2:30000,cNone:2500,c3:70000,c4:900,c5:50000", coverage=12, seed=41,
                  extra=["--samples", "2", "--insert-mean", "260", "--insert-sd", "40", "--tie-free-overlaps"])
    bed = str(d / "r.bed")
    with open(bed, "w") as fh:
        fh.write("c3\t100\t9000\tx\nc1\t5000\t5100\ty\nc5\t49000\t50000\tz\ncNone\t10\t500\tq\nc1\t80000\t89000\tw\nc2\t0\t30000\tv\n")
    return bam, bed


def sharded(args, n, slice_positions=None, **kw):
    env = {"SBX_DEVICES": ",".join(["0"] * n)}
    if slice_positions:
        env["SBX_SLICE_POSITIONS"] = str(slice_positions)
    return run_cli(args + ["--gpus", str(n)], env=env, **kw)


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("args", [
    ["base"],
    ["base", "-m", "-q", "20"],
    ["base", "-a", "-C", "45", "-c", "3"],
    ["window", "-w", "1000", "-T", "20"],
    ["window", "-w", "1000", "-m", "-T", "20"],
    ["region", "-L", "chrOne:30000-150000", "-m", "-T", "10"],
])
def test_one_contig_sharded_by_position_equals_one_context(one_contig, args, n):
    want = run_cli(args + [one_contig])
    assert len(want) > 100
    assert sharded(args + [one_contig], n, slice_positions=30000) == want


@pytest.mark.parametrize("n", [2, 4])
def test_base_to_a_file_every_context_writes_its_own_byte_range(one_contig, tmp_path, n):
    want = run_cli(["base", "-m", one_contig])
    out = str(tmp_path / "out.txt")
    assert sharded(["base", "-m", "-o", out, one_contig], n, slice_positions=20000) == b""
    with open(out, "rb") as fh:
        assert fh.read() == want


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("args", [
    ["base"], ["base", "-q", "10", "--combined"],
    ["region", "-L", "BED", "-T", "5", "-T", "20"],
    ["region", "-L", "BED", "-m", "-q", "20", "-T", "3"],
    ["window", "-w", "1000", "-T", "10"],
    ["window", "-w", "700", "-m", "-q", "13"],
    ["window", "-w", "333", "--combined", "-a", "-c", "5"],
])
def test_a_genome_sharded_equals_one_context(genome, args, n):
    bam, bed = genome
    a = [bed if x == "BED" else x for x in args]
    want = run_cli(a + [bam])
    assert sharded(a + [bam], n, slice_positions=16384) == want
    assert want == run_oracle(a + [bam])


def test_order_dependent_option_sets_run_on_one_device_and_say_so(genome):
    bam, bed = genome
    for args in (["base", "-c", "0"], ["base", "-L", bed], ["window", "-w", "1000", "--overlap", "500"]):
        r = sharded(args + [bam], 2, check=False)
        assert r.returncode == 0 and b"running on one device" in r.stderr
        assert r.stdout == run_cli(args + [bam])


def test_alignments_hanging_over_contig_ends_and_trailing_readless_contigs(tmp_path):
    """what the window printer does behind a contig's end and on the read-less contigs after the last one with reads (depth.d:1057-1076),
    from collected statistics"""
    import random
    from tests import bamgen as bg
    rng = random.Random(9)
    refs = [("c0", 5000), ("cE", 1200), ("c2", 4000), ("cT", 2600), ("cT2", 700)]
    recs = []
    for ref, n in ((0, 300), (2, 260)):
        L = refs[ref][1]
        ps = sorted(rng.randint(0, L - 20) for _ in range(n))
        for i, p in enumerate(ps):
            seq = "".join(rng.choice("ACGT") for _ in range(100))
            recs.append(bg.make_record(ref, p, "100M" if i % 9 else "40M30N60M", seq, [rng.choice([5, 30]) for _ in range(100)], name="r%d_%d" % (ref, i)))
    p = str(tmp_path / "tails.bam")
    bg.write_bam(p, refs, recs)
    for args in (["window", "-w", "500"], ["window", "-w", "300", "-T", "3"], ["base"], ["region", "-L", "c2:3000-4100", "-T", "1"]):
        want = run_oracle(args + [p])
        assert run_cli(args + [p]) == want, args
        for n in (2, 3):
            assert sharded(args + [p], n, slice_positions=2048) == want, (args, n)


def test_device_list_errors():
    r = run_cli(["base", "--gpus", "64", os.path.join(os.path.dirname(__file__), "golden", "issue225.bam")], check=False)
    assert r.returncode == 1 and b"HIP device(s) visible" in r.stderr
    r = run_cli(["base", os.path.join(os.path.dirname(__file__), "golden", "issue225.bam")], check=False, env={"SBX_DEVICES": "0,x"})
    assert r.returncode == 1 and b"SBX_DEVICES" in r.stderr
