"""The device work list (engine.cpp build_runs): with -L / sbx_run_interval only the BGZF block runs that hold the
merged BAI chunks are uploaded and inflated (RandomAccessManager.getChunks / getReads, randomaccessmanager.d:247-348),
every run starting at a record boundary the index names -- results must not depend on how the file was cut."""
import os
import random

import numpy as np
import pytest

from tests.util import gen_bam, oracle_base_counters, run_cli, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    d = tmp_path_factory.mktemp("wl")
    contigs = ",".join("c%d:%d" % (i + 1, 260000 - 9000 * i) for i in range(25))
    return gen_bam(str(d / "g25.bam"), contigs, coverage=20, seed=91)


def sparse_bed(path, n_contigs=25, per_contig=2, seed=5):
    rng = random.Random(seed)
    rows = []
    for i in range(n_contigs):
        L = 260000 - 9000 * i
        for _ in range(per_contig):
            a = rng.randrange(0, L - 400)
            rows.append("c%d\t%d\t%d\n" % (i + 1, a, a + rng.choice([60, 150, 300])))
    with open(path, "w") as fh:
        fh.writelines(rows)
    return path


def test_sparse_bed_touches_only_its_chunks(genome, tmp_path):
    """BASELINE config 4's premise: an exome-like BED over 25 contigs inflates the chunk-covered blocks, not the genome."""
    import sambamba_amd
    bed = sparse_bed(str(tmp_path / "sparse.bed"))
    with sambamba_amd.Depth(genome) as d:
        merged, raw, lines = d.parse_regions(bed)
        assert len(raw) == 50 and len(lines) == 50      # each costs about one 16 kb linear-index window of reads
        d.set_params(mode=sambamba_amd.SBX_MODE_REGION)
        d.set_regions(merged)
        st = d.run()
        total = d.info.uncompressed_bytes
        assert st["n_runs"] > 25                                   # many separate chain runs ...
        assert st["uncompressed_bytes"] < 0.30 * total             # ... covering a fraction of the file
        assert st["uploaded_bytes"] < 0.30 * d.info.compressed_bytes
    for args in (["region", "-L", bed, "-T", "5", "-T", "15"], ["region", "-L", bed, "-m", "-q", "13"], ["base", "-L", bed]):
        assert run_cli(args + [genome]) == run_oracle(args + [genome]), args


def test_run_interval_counters_equal_the_whole_run(genome):
    """Counters inside [beg, end) of an interval run are complete: equal to the whole-file run and to the oracle."""
    import sambamba_amd
    with sambamba_amd.Depth(genome) as d:
        d.set_params()
        whole = d.run()
        ref = 3
        L = d.ref_lengths[ref]
        full = d.base_counters(ref, 0, L)
        for beg, end in ((0, 50000), (70000 - 512, 140000 + 1024), (L - 30000, L), (100352, 101376)):
            st = d.run_interval(ref, beg, end)
            assert st["uncompressed_bytes"] < whole["uncompressed_bytes"] // 8
            got = d.base_counters(ref, beg, end)
            assert np.array_equal(got, full[beg:end]), (beg, end)
        assert np.array_equal(full[60000:90000], oracle_base_counters(genome, ref, 60000, 90000))


def test_preloaded_and_streamed_payload_agree(genome):
    import sambamba_amd
    with sambamba_amd.Depth(genome) as d:
        d.set_params()
        d.preload()
        a = d.run()
        ca = d.base_counters(7, 0, d.ref_lengths[7])
        ia = d.run_interval(7, 20000, 90000)
        cia = d.base_counters(7, 20000, 90000)
    with sambamba_amd.Depth(genome) as d:
        d.set_params()
        b = d.run()
        cb = d.base_counters(7, 0, d.ref_lengths[7])
        ib = d.run_interval(7, 20000, 90000)
        cib = d.base_counters(7, 20000, 90000)
    assert a["n_records"] == b["n_records"] and a["n_admitted"] == b["n_admitted"]
    assert np.array_equal(ca, cb) and np.array_equal(cia, cib) and np.array_equal(cia, ca[20000:90000])
    assert ia["n_records"] == ib["n_records"]
    assert ib["uploaded_bytes"] < b["uploaded_bytes"] // 8        # only the interval's blocks travel


def test_short_records_overflow_the_descriptor_estimate_and_retry(tmp_path):
    """Records far smaller than the sizing heuristic assumes (36 bp reads): K2 reports the overflow, the engine sizes the
    descriptor array exactly and repeats the pass."""
    import sambamba_amd
    p = gen_bam(str(tmp_path / "short.bam"), "c1:300000", coverage=40, seed=13, extra=["--read-len", "36", "--insert-mean", "120", "--insert-sd", "10"])
    with sambamba_amd.Depth(p) as d:
        d.set_params()
        st = d.run()
        assert st["uncompressed_bytes"] / st["n_records"] < 160
        assert np.array_equal(d.base_counters(0, 100000, 160000), oracle_base_counters(p, 0, 100000, 160000))
    assert run_cli(["window", "-w", "5000", p]) == run_oracle(["window", "-w", "5000", p])


def test_block_table_scanned_in_pieces_equals_the_serial_scan(genome, monkeypatch):
    """sbx_open cuts the serial BGZF header chain at block starts the BAI names and scans the pieces on separate threads
    (files above 64 MB; the hook lowers the limit): same block table, same results."""
    import sambamba_amd
    def run():
        with sambamba_amd.Depth(genome) as d:
            d.set_params()
            st = d.run()
            return d.info.n_bgzf_blocks, d.info.uncompressed_bytes, st["n_records"], d.base_counters(11, 0, d.ref_lengths[11])
    monkeypatch.setenv("SBX_SCAN_PARALLEL_MIN", "1000000000000")
    a = run()
    monkeypatch.setenv("SBX_SCAN_PARALLEL_MIN", "0")
    b = run()
    assert a[:3] == b[:3] and np.array_equal(a[3], b[3])
    assert a[0] > 50
