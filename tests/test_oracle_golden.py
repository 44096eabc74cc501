"""The CPU oracle must reproduce the reference's own golden outputs byte for byte.

Commands are the ones in the reference's test/test_suite.sh:151,159,178,182,187,191
(testIssue193, testIssue204, testIssue225); goldens are the files checked in next to them
(copied to tests/golden/, see tests/golden/README.md).
"""
import os
import subprocess

import pytest

CASES = [
    (["base", "issue_193.bam"], "issue_193_expected_output.txt"),
    (["base", "-c", "1", "issue225.bam"], "issue225.out"),
    (["base", "-c", "0", "issue225.bam"], "issue225.z.out"),
    (["base", "-c", "1", "-L", "chrM", "issue225.bam"], "issue225.out"),
    (["base", "-c", "0", "-L", "chrM", "issue225.bam"], "issue225.z.out"),
    (["region", "issue_204.bam", "-L", "2:166868600-166868813", "-T", "15", "-T", "20", "-T", "25", "-m"],
     "issue_204_expected_output.txt"),
]


@pytest.mark.parametrize("args,golden", CASES)
def test_oracle_reproduces_reference_golden(oracle_bin, golden_dir, args, golden):
    out = subprocess.run([oracle_bin] + args, cwd=golden_dir, stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, check=True).stdout
    with open(os.path.join(golden_dir, golden), "rb") as fh:
        want = fh.read()
    assert out == want


def test_oracle_deprecated_z_flag_equals_c0(oracle_bin, golden_dir):
    # depth.d:416-419: -z is an alias for --min-coverage=0
    a = subprocess.run([oracle_bin, "base", "-z", "issue225.bam"], cwd=golden_dir, stdout=subprocess.PIPE,
                       stderr=subprocess.DEVNULL, check=True).stdout
    with open(os.path.join(golden_dir, "issue225.z.out"), "rb") as fh:
        assert a == fh.read()


def test_oracle_requires_regions_in_region_mode(oracle_bin, golden_dir):
    r = subprocess.run([oracle_bin, "region", "issue225.bam"], cwd=golden_dir, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 1


@pytest.mark.parametrize("opts", [dict(), dict(min_bq=20), dict(fix_mate=True, min_bq=13)])
def test_oracle_counters_fetched_through_the_index_equal_the_whole_file_pass(tmp_path, opts):
    """bench.py samples windows of the full-size run with the indexed variant: inside the window it must give exactly what
    the pass over the whole file gives (every read covering a position of the window overlaps the fetched region)."""
    import numpy as np
    from tests.util import gen_bam, oracle_base_counters
    bam = gen_bam(str(tmp_path / "ix.bam"), "cA:90000,cB:70000", coverage=25, seed=23,
                  extra=["--insert-mean", "260", "--insert-sd", "40", "--tie-free-overlaps"])
    for ref, name, beg, end in ((0, "cA", 20000, 31000), (1, "cB", 0, 5000), (1, "cB", 66000, 70000)):
        whole = oracle_base_counters(bam, ref, beg, end, **opts)
        fetched = oracle_base_counters(bam, ref, beg, end, ref_name=name, **opts)
        assert whole.sum() > 0
        assert np.array_equal(whole, fetched), (name, beg, end)
