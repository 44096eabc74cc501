"""Parity of the device path (K1 inflate -> K2 record index -> K3 accumulate) through the C ABI and
the sbx-depth CLI: byte-identical text against the reference's goldens, bit-exact counters against
the CPU oracle on seeded synthetic BAMs."""
import os

import numpy as np
import pytest

from tests.util import GOLDEN, gen_bam, oracle_base_counters, run_cli, run_oracle

pytestmark = pytest.mark.gpu

GOLDEN_BASE_CASES = [
    (["base", "issue_193.bam"], "issue_193_expected_output.txt"),
    (["base", "-c", "1", "issue225.bam"], "issue225.out"),
    (["base", "-c", "0", "issue225.bam"], "issue225.z.out"),
    (["base", "-c", "1", "-L", "chrM", "issue225.bam"], "issue225.out"),
    (["base", "-c", "0", "-L", "chrM", "issue225.bam"], "issue225.z.out"),
]


@pytest.mark.parametrize("args,golden", GOLDEN_BASE_CASES)
def test_cli_base_reproduces_reference_golden(args, golden):
    out = run_cli(args, cwd=GOLDEN)
    with open(os.path.join(GOLDEN, golden), "rb") as fh:
        assert out == fh.read()


@pytest.mark.parametrize("args", [
    ["base", "issue_204.bam"],
    ["base", "-q", "20", "issue_204.bam"],
    ["base", "-q", "30", "-a", "issue_204.bam"],
    ["base", "-F", "mapping_quality >= 30 and not duplicate", "issue_204.bam"],
    ["base", "-F", "proper_pair and not (secondary_alignment or supplementary)", "mate_overlaps_1_3M_4M.bam"],
    ["base", "-L", "mate_overlaps_1_3M_4M.bed", "mate_overlaps_1_3M_4M.bam"],
    ["base", "-c", "0", "-L", "mate_overlaps_1_3M_4M.bed", "mate_overlaps_1_3M_4M.bam"],
    ["base", "-c", "3", "-C", "10", "issue_204.bam"],
    ["base", "--combined", "-a", "-c", "2", "issue225.bam"],
    ["base", "-L", "2:166868600-166868813", "issue_204.bam"],
])
def test_cli_base_matches_oracle_on_reference_fixtures(args):
    assert run_cli(args, cwd=GOLDEN) == run_oracle(args, cwd=GOLDEN)


@pytest.fixture(scope="module")
def synth_small(tmp_path_factory):
    d = tmp_path_factory.mktemp("synth")
    return gen_bam(str(d / "s1.bam"), "chrA:300000,chrEmpty:5000,chrB:120000,chrTiny:700", coverage=30, seed=11)


@pytest.fixture(scope="module")
def synth_multisample(tmp_path_factory):
    d = tmp_path_factory.mktemp("synthms")
    return gen_bam(str(d / "s3.bam"), "c1:150000,c2:90000", coverage=20, seed=12, extra=["--samples", "3"])


def test_counters_match_oracle_synthetic(synth_small):
    import sambamba_amd
    with sambamba_amd.Depth(synth_small) as d:
        d.set_params()
        st = d.run()
        assert st["n_records"] > 0 and st["n_admitted"] > 0
        for ref in range(d.info.n_ref):
            n = d.ref_lengths[ref]
            got = d.base_counters(ref, 0, n)
            want = oracle_base_counters(synth_small, ref, 0, n)
            assert np.array_equal(got, want), "ref %d" % ref


@pytest.mark.parametrize("min_bq", [0, 13, 24, 38])
def test_counters_min_base_quality(synth_small, min_bq):
    import sambamba_amd
    with sambamba_amd.Depth(synth_small) as d:
        d.set_params(min_bq=min_bq)
        d.run()
        got, cov = d.base_counters(0, 1000, 60000, with_covered=True)
        want = oracle_base_counters(synth_small, 0, 1000, 60000, min_bq=min_bq)
        assert np.array_equal(got, want)
        want0 = oracle_base_counters(synth_small, 0, 1000, 60000, min_bq=0)
        assert np.array_equal(cov.astype(bool), want0.sum(axis=(1, 2)) > 0)


@pytest.mark.parametrize("min_bq", [0, 20])
def test_deep_tiles_keep_32_bit_counters(synth_small, synth_multisample, min_bq, monkeypatch):
    """K3 accumulates tiles with fewer than 2^16 records in 16-bit LDS counters and leaves the others to the 32-bit
    kernel; the hook lowers the limit so that ordinary tiles (30x: ~230 records) are split between the two."""
    import sambamba_amd
    for bam, ns in ((synth_small, 1), (synth_multisample, 3)):
        for thr in ("1", "150"):
            monkeypatch.setenv("SBX_DEEP_TILE_RECORDS", thr)
            with sambamba_amd.Depth(bam) as d:
                d.set_params(min_bq=min_bq)
                d.run()
                got, cov = d.base_counters(0, 0, 90000, with_covered=True)
                want = oracle_base_counters(bam, 0, 0, 90000, n_samples=ns, min_bq=min_bq)
                assert np.array_equal(got, want), (bam, thr)
                want0 = oracle_base_counters(bam, 0, 0, 90000, n_samples=ns, min_bq=0)
                assert np.array_equal(cov.astype(bool), want0.sum(axis=(1, 2)) > 0)


@pytest.mark.parametrize("read_len", [36, 161, 250])
def test_read_lengths_around_the_fast_path_limit(tmp_path, read_len):
    """The fast path of K3 takes single-run reads with up to 160 bases inside the tile; longer runs go one read at a time."""
    import sambamba_amd
    bam = gen_bam(str(tmp_path / "rl.bam"), "c1:120000", coverage=25, seed=17,
                  extra=["--read-len", str(read_len), "--insert-mean", str(2 * read_len + 50), "--insert-sd", "20"])
    for min_bq in (0, 15):
        with sambamba_amd.Depth(bam) as d:
            d.set_params(min_bq=min_bq)
            d.run()
            got = d.base_counters(0, 0, 120000)
            want = oracle_base_counters(bam, 0, 0, 120000, min_bq=min_bq)
            assert np.array_equal(got, want), (read_len, min_bq)


def test_multisample_min_base_quality(synth_multisample):
    import sambamba_amd
    with sambamba_amd.Depth(synth_multisample) as d:
        d.set_params(min_bq=17)
        d.run()
        got = d.base_counters(1, 0, d.ref_lengths[1])
        want = oracle_base_counters(synth_multisample, 1, 0, d.ref_lengths[1], n_samples=3, min_bq=17)
        assert np.array_equal(got, want)


def test_cli_text_matches_oracle_synthetic(synth_small):
    for args in (["base"], ["base", "-c", "0", "-L", "chrB:1000-3000"], ["base", "-q", "20", "-a", "-c", "25"],
                 ["base", "-F", "mapping_quality >= 0"]):
        assert run_cli(args + [synth_small]) == run_oracle(args + [synth_small]), args


def test_multisample_counters_and_text(synth_multisample):
    import sambamba_amd
    with sambamba_amd.Depth(synth_multisample) as d:
        assert d.info.n_samples == 3
        d.set_params()
        d.run()
        got = d.base_counters(0, 0, d.ref_lengths[0])
        want = oracle_base_counters(synth_multisample, 0, 0, d.ref_lengths[0], n_samples=3)
        assert np.array_equal(got, want)
        d.set_params(combined=True)
        d.run()
        got1 = d.base_counters(1, 0, d.ref_lengths[1])
        want1 = oracle_base_counters(synth_multisample, 1, 0, d.ref_lengths[1], n_samples=1, combined=True)
        assert np.array_equal(got1, want1)
    for args in (["base"], ["base", "--combined"], ["base", "-a", "-c", "8"]):
        assert run_cli(args + [synth_multisample]) == run_oracle(args + [synth_multisample]), args


def test_errors_match_reference_messages(tmp_path):
    r = run_cli(["base", "/nonexistent/file.bam"], check=False)
    assert r.returncode == 1 and r.stderr.startswith(b"sambamba-depth: ")
    r = run_cli(["region", os.path.join(GOLDEN, "issue225.bam")], check=False)
    assert r.returncode == 1 and b"BED file or a region must be provided in region mode" in r.stderr
