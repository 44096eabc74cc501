"""K6 `format_base_rows` (device-side text of `depth base`): the rows the device
writes are byte-identical to the oracle's text and to the host emulation of PerBasePrinter that the CLI
keeps for -L with -c 0 (SBX_HOST_FORMAT=1 forces that path everywhere)."""
import os
import subprocess

import numpy as np
import pytest

from tests.util import GOLDEN, gen_bam, run_cli, run_oracle

pytestmark = pytest.mark.gpu

CASES = [
    ["base", "issue_193.bam"],
    ["base", "-c", "0", "issue225.bam"],
    ["base", "-c", "1", "issue225.bam"],
    ["base", "-c", "0", "-a", "issue225.bam"],
    ["base", "-c", "2", "-a", "issue225.bam"],
    ["base", "-c", "3", "-C", "10", "issue_204.bam"],
    ["base", "-c", "0", "-C", "5", "issue225.bam"],       # (-c 0 prints every position of every contig: small genomes only)
    ["base", "-q", "30", "-c", "0", "issue225.bam"],
    ["base", "-q", "30", "-a", "issue_204.bam"],
    ["base", "--combined", "-a", "-c", "2", "issue225.bam"],
    ["base", "--combined", "-c", "0", "issue225.bam"],
    ["base", "-L", "mate_overlaps_1_3M_4M.bed", "mate_overlaps_1_3M_4M.bam"],
    ["base", "-c", "0", "-L", "mate_overlaps_1_3M_4M.bed", "mate_overlaps_1_3M_4M.bam"],
    ["base", "-L", "2:166868600-166868813", "issue_204.bam"],
    ["base", "-m", "-q", "20", "mate_overlaps_1_3M_4M.bam"],
]


def run_cli_host_format(args, cwd):
    from sambamba_amd import cli_path
    env = dict(os.environ, SBX_HOST_FORMAT="1")
    return subprocess.run([cli_path()] + list(args), cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                          check=True).stdout


@pytest.mark.parametrize("args", CASES)
def test_device_rows_equal_oracle_and_host_emulation(args):
    dev = run_cli(args, cwd=GOLDEN)
    assert dev == run_oracle(args, cwd=GOLDEN)
    assert dev == run_cli_host_format(args, cwd=GOLDEN)


@pytest.fixture(scope="module")
def synth_ms(tmp_path_factory):
    d = tmp_path_factory.mktemp("fmtms")
    return gen_bam(str(d / "ms.bam"), "c1:150000,cEmpty:3000,c2:40000", coverage=12, seed=21, extra=["--samples", "3"])


@pytest.mark.parametrize("extra", [["-c", "0"], ["-c", "2", "-a"]])
def test_rows_of_many_samples_go_past_the_lds(tmp_path, extra):
    """Eight samples: the rows of a 256-position chunk (8 x ~40 bytes per position: > 80 KB) exceed the 48 KB of LDS a chunk may use,
    so the write kernel stores them straight to HBM -- RowSink with 8-byte global stores at any byte address (format_core.hpp)."""
    bam = gen_bam(str(tmp_path / "s8.bam"), "cL:70000,cS:9000", coverage=16, seed=31, extra=["--samples", "8"])
    args = ["base"] + extra + [bam]
    assert run_cli(args) == run_oracle(args)


@pytest.mark.parametrize("extra", [[], ["-c", "0"], ["-c", "5", "-C", "14"], ["-c", "4", "-a"], ["-q", "24", "-c", "0"],
                                   ["--combined", "-c", "0"], ["-q", "38", "-c", "2", "-a"]])
def test_multisample_rows(synth_ms, extra):
    args = ["base"] + extra + [synth_ms]
    assert run_cli(args) == run_oracle(args)


def expected_rows(name, samples, beg, counters, covered, min_cov, max_cov, annotate, combined):
    """The reference's writeColumn / writeEmptyColumns text from dense counters (depth.d:534-555,452-487)."""
    out = []
    for i in range(counters.shape[0]):
        if not covered[i] and min_cov > 0:
            continue
        for s in range(counters.shape[1]):
            v = counters[i, s]
            tot = int(v.sum())
            ok = min_cov <= tot <= max_cov
            if not ok and not annotate:
                break
            row = [name, str(beg + i), str(tot)] + [str(int(v[k])) for k in (0, 1, 2, 3, 5, 6)]
            if not combined:
                row.append(samples[s])
            if annotate:
                row.append("y" if ok else "n")
            out.append("\t".join(row) + "\n")
    return "".join(out).encode()


@pytest.mark.parametrize("min_bq,min_cov,max_cov,annotate", [(0, 1, 1e300, False), (0, 0, 1e300, False), (24, 0, 9, False),
                                                             (0, 6, 11, True), (38, 1, 1e300, True)])
def test_api_rows_from_counters(synth_ms, min_bq, min_cov, max_cov, annotate):
    import sambamba_amd
    with sambamba_amd.Depth(synth_ms) as d:
        d.set_params(min_bq=min_bq)
        d.run()
        samples = d.sample_names
        for ref, (b, e) in [(0, (0, 150000)), (0, (777, 70001)), (1, (0, 3000)), (2, (39000, 40000))]:
            cnt, cov = d.base_counters(ref, b, e, with_covered=True)
            want = expected_rows(d.ref_names[ref], samples, b, cnt, cov, min_cov, max_cov, annotate, False)
            got = d.format_base_rows(ref, b, e, min_cov=min_cov, max_cov=max_cov, annotate=annotate)
            assert got == want, (ref, b, e)


def test_api_small_buffer_reports_required_size(synth_ms):
    import ctypes as C
    import sambamba_amd
    from sambamba_amd import _lib
    with sambamba_amd.Depth(synth_ms) as d:
        d.set_params()
        d.run()
        need = C.c_size_t(0)
        buf = C.create_string_buffer(16)
        rc = d._L.sbx_format_base_rows(d._ctx, 0, 0, 5000, 1.0, 1e300, 0, buf, 16, C.byref(need))
        assert rc == _lib.ENOMEM and need.value > 16
        assert len(d.format_base_rows(0, 0, 5000)) == need.value


def test_zero_fill_skips_contigs_between_columns(tmp_path):
    """-c 0: a contig without reads is zero-filled before the first / after the last contig that has columns,
    but skipped when it lies between two that have (PerBasePrinter.push jumps straight to the new contig,
    depth.d:574-583)."""
    from tests import bamgen as bg
    refs = [("e0", 30), ("a", 400), ("e1", 25), ("e2", 7), ("b", 300), ("e3", 12), ("e4", 3)]
    recs = [bg.make_record(1, 50, "40M", "ACGT" * 10, 30, name="x"), bg.make_record(4, 10, "20M2D20M", "ACGT" * 10, 30, name="y")]
    p = str(tmp_path / "gaps.bam")
    bg.write_bam(p, refs, recs)
    for args in (["base", "-c", "0"], ["base", "-c", "0", "-a"], ["base"]):
        dev = run_cli(args + [p])
        assert dev == run_oracle(args + [p])
        assert dev == run_cli_host_format(args + [p], cwd=None)
    out = run_cli(["base", "-c", "0", p]).decode()
    assert "e0\t0\t" in out and "e3\t0\t" in out and "e4\t2\t" in out and "e1\t" not in out and "e2\t" not in out


@pytest.mark.parametrize("piece", ["256", "5000", "70000"])
def test_streamed_rows_equal_the_buffered_form(synth_ms, piece, monkeypatch):
    """sbx_stream_base_rows (pinned double-buffered pieces, what the CLI prints through) == sbx_format_base_rows, for piece
    sizes that cut the interval into many, a few and one piece; a failing writer surfaces as SBX_EIO."""
    import sambamba_amd
    monkeypatch.setenv("SBX_STREAM_PIECE", piece)
    with sambamba_amd.Depth(synth_ms) as d:
        d.set_params()
        d.run()
        for ref, beg, end, kw in ((0, 0, 150000, {}), (0, 777, 60001, dict(min_cov=0)), (2, 100, 40000, dict(min_cov=3, annotate=True)),
                                  (1, 0, 3000, dict(min_cov=0))):
            want = d.format_base_rows(ref, beg, end, **kw)
            got = []
            d.stream_base_rows(ref, beg, end, got.append, **kw)
            assert b"".join(got) == want, (ref, beg, end, kw)
            assert len(want) > 0
        def broken(_):
            raise IOError("disk full")
        with pytest.raises(sambamba_amd.SbxError) as e:
            d.stream_base_rows(0, 0, 150000, broken)
        assert e.value.code == -2 and "writer" in str(e.value)
        # the context stays usable
        got = []
        d.stream_base_rows(2, 0, 1000, got.append)
        assert b"".join(got) == d.format_base_rows(2, 0, 1000)
