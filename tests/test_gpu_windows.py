"""`depth window` corner cases of PerWindowPrinter (depth.d:933-1077) against the oracle: overlapping windows (ring of
n = ceil(w / step) slots: extended coverage range when w is not a multiple of the step, first ring counting only reads
that start inside), read-less contigs before / between / after contigs with reads -- the first one after the last
contig with reads continues that contig's window coordinates and shows what its unfinished windows held."""
import numpy as np
import pytest

from tests import bamgen as bg
from tests.util import gen_bam, run_cli, run_oracle

pytestmark = pytest.mark.gpu


def make(path, refs, spec, seed):
    rng = np.random.RandomState(seed)
    recs = []
    for ref, lo, hi, n in spec:
        recs += [(ref, int(p)) for p in rng.randint(lo, hi, size=n)]
    recs.sort()
    out = [bg.make_record(r, p, "60M" if i % 7 else "20M5D30M10S", "ACGT" * 15, [int(x) for x in rng.randint(5, 40, size=60)],
                          name="q%d" % i, mapq=60 if i % 11 else 0) for i, (r, p) in enumerate(recs)]
    bg.write_bam(path, refs, out)
    return path


@pytest.fixture(scope="module")
def bams(tmp_path_factory):
    d = tmp_path_factory.mktemp("win")
    return {
        # first column late on contig 0, read-less contig in the middle, one at the end
        "A": make(str(d / "a.bam"), [("c0", 2000), ("cE", 777), ("c2", 1500), ("cT", 260)], [(0, 150, 1500, 120), (2, 0, 1400, 90)], 3),
        # read-less contig first, two at the end (only the first continues the coordinates)
        "B": make(str(d / "b.bam"), [("cE0", 300), ("c1", 2000), ("cE", 55), ("c2", 900), ("cT1", 500), ("cT2", 410)],
                  [(1, 5, 1900, 150), (3, 100, 800, 40)], 4),
        # a single contig, first column inside the first ring
        "C": make(str(d / "c.bam"), [("c0", 1200)], [(0, 30, 1100, 80)], 5),
        # reads hanging over the end of the last contig with reads, then a read-less contig
        "D": make(str(d / "d.bam"), [("c0", 1000), ("cT", 900)], [(0, 0, 990, 140)], 6),
    }


@pytest.mark.parametrize("which", ["A", "B", "C", "D"])
@pytest.mark.parametrize("extra", [
    ["-w", "100"], ["-w", "100", "-T", "2", "-T", "5"], ["-w", "100", "--overlap", "50", "-T", "2"],
    ["-w", "100", "--overlap", "30", "-T", "3", "-q", "20"], ["-w", "90", "--overlap", "70", "-T", "1", "-T", "4"],
    ["-w", "64", "--overlap", "40", "-q", "13", "-a", "-c", "1"], ["-w", "250", "--overlap", "249", "-T", "2"],
])
def test_window_quirks(bams, which, extra):
    args = ["window"] + extra + [bams[which]]
    assert run_cli(args) == run_oracle(args)


@pytest.fixture(scope="module")
def synth(tmp_path_factory):
    d = tmp_path_factory.mktemp("winsynth")
    return gen_bam(str(d / "s.bam"), "chrA:60000,chrB:9000,chrC:31000", coverage=10, seed=51, extra=["--samples", "2"])


@pytest.mark.parametrize("extra", [["-w", "1000", "--overlap", "500", "-T", "5"], ["-w", "300", "--overlap", "100", "-T", "3", "-T", "12"],
                                   ["-w", "700", "--overlap", "650", "--combined", "-q", "20"]])
def test_overlapping_windows_synthetic(synth, extra):
    args = ["window"] + extra + [synth]
    assert run_cli(args) == run_oracle(args)


def test_overlapping_windows_in_batches(bams, synth):
    from tests.test_gpu_batches import cli_batched
    for args in (["window", "-w", "100", "--overlap", "30", "-T", "3", bams["A"]], ["window", "-w", "100", bams["B"]],
                 ["window", "-w", "300", "--overlap", "100", "-T", "3", synth]):
        assert cli_batched(args, 1) == run_cli(args)


def test_overlap_with_fix_mate_is_rejected(synth):
    r = run_cli(["window", "-w", "300", "--overlap", "100", "-m", synth], check=False)
    assert r.returncode != 0 and b"--overlap" in r.stderr


@pytest.mark.parametrize("w", ["1", "7", "250"])
def test_small_windows_fall_back_to_per_call_statistics_when_the_cache_would_not_fit(bams, w):
    """the all-windows-at-once cache is bounded; beyond its budget every call computes its own windows -- same text."""
    for which in ("A", "D"):
        args = ["window", "-w", w, "-T", "2", bams[which]]
        want = run_oracle(args)
        assert run_cli(args) == want
        assert run_cli(args, env={"SBX_WINDOW_CACHE_BYTES": "64"}) == want
