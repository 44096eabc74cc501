"""Streaming over contigs (sbx_plan_batches / sbx_run_batch): a BAM processed in several batches of
contigs gives byte-identical text and bit-identical counters to the single pass and to the oracle."""
import os
import subprocess

import numpy as np
import pytest

from tests.util import gen_bam, oracle_base_counters, run_cli, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def multi(tmp_path_factory):
    d = tmp_path_factory.mktemp("batches")
    bam = gen_bam(str(d / "g.bam"), "c1:120000,c2:50000,cEmpty:4000,c3:90000,c4:700,c5:60000", coverage=15, seed=31,
                  extra=["--samples", "2"])
    bed = str(d / "r.bed")
    with open(bed, "w") as fh:      # unsorted on purpose, one region per contig group, one on the empty contig
        fh.write("c3\t100\t9000\tx\nc1\t5000\t5100\ty\nc5\t59000\t60000\tz\ncEmpty\t10\t500\tq\nc1\t100000\t119000\tw\n")
    return bam, bed


def cli_batched(args, budget):
    from sambamba_amd import cli_path
    env = dict(os.environ, SBX_BATCH_BYTES=str(budget))
    return subprocess.run([cli_path()] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, check=True).stdout


def test_plan_covers_all_contigs_in_order(multi):
    import sambamba_amd
    bam, _ = multi
    with sambamba_amd.Depth(bam) as d:
        d.set_params()
        one = d.plan_batches(1 << 40)
        assert one == [(0, d.info.n_ref, one[0][2])]
        many = d.plan_batches(1)                 # nothing fits: one contig per batch
        assert [(f, n) for f, n, _ in many] == [(r, 1) for r in range(d.info.n_ref)]
        mid = d.plan_batches(many[0][2] + many[1][2] + (many[0][2] // 2))
        assert sum(n for _, n, _ in mid) == d.info.n_ref and 1 < len(mid) < d.info.n_ref
        nxt = 0
        for f, n, _ in mid:
            assert f == nxt and n >= 1
            nxt = f + n


@pytest.mark.parametrize("args", [
    ["base"], ["base", "-c", "0"], ["base", "-q", "24", "-c", "0", "-a"], ["base", "--combined", "-c", "3"],
    ["base", "-m", "-q", "13"],
    ["window", "-w", "1000"], ["window", "-w", "700", "-T", "5", "-T", "20"],
    ["region", "-L", "BED"], ["region", "-L", "BED", "-T", "10", "--combined"], ["base", "-L", "BED"],
    ["base", "-c", "0", "-L", "BED"], ["region", "-L", "c3:2000-30000"],
])
def test_cli_batched_equals_single_pass_and_oracle(multi, args):
    bam, bed = multi
    a = [bed if x == "BED" else x for x in args] + [bam]
    single = run_cli(a)
    assert single == run_oracle(a)
    assert cli_batched(a, 1) == single                       # one contig per batch
    assert cli_batched(a, 40_000_000) == single              # a few contigs per batch


def test_api_batch_counters(multi):
    import sambamba_amd
    bam, _ = multi
    with sambamba_amd.Depth(bam) as d:
        d.set_params(min_bq=13)
        for first, n, _ in d.plan_batches(1):
            st = d.run_batch(first, n)
            for ref in range(first, first + n):
                L = d.ref_lengths[ref]
                got = d.base_counters(ref, 0, L)
                want = oracle_base_counters(bam, ref, 0, L, n_samples=2, min_bq=13)
                assert np.array_equal(got, want), ref
