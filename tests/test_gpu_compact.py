"""Region / window runs keep ONE word per position and sample -- {bases counted, depth} -- instead of the seven base counters
(engine.cpp `compact_counters`, depth.hip write-out, reduce.hip k_range_reduce): what PerRegionPrinter / PerWindowPrinter print is a
sum over those two numbers (sambamba/depth.d:661-698,760-845,933-1077).  The form with the seven counters (SBX_COMPACT=0) is the
witness: same text; and the C ABI says what it no longer holds after such a run."""
import os

import numpy as np
import pytest

import sambamba_amd
from tests.util import gen_bam, run_cli, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def synth(tmp_path_factory):
    d = tmp_path_factory.mktemp("compact")
    return gen_bam(str(d / "s.bam"), "chrA:300000,chrEmpty:20000,chrB:90000", coverage=30, seed=77)


@pytest.fixture(scope="module")
def synth_ms(tmp_path_factory):
    d = tmp_path_factory.mktemp("compactms")
    return gen_bam(str(d / "s3.bam"), "c1:100000,c2:50000", coverage=20, seed=78, extra=["--samples", "3"])


@pytest.mark.parametrize("args", [
    ["window", "-w", "1000"], ["window", "-w", "1000", "-T", "10", "-T", "30", "-T", "0"], ["window", "-w", "613", "-q", "13", "-T", "20"],
    ["window", "-w", "2500", "--overlap", "500", "-T", "25"], ["region", "-L", "chrA:1000-250000", "-T", "5", "-T", "31"],
    ["region", "-L", "chrB", "-q", "20"],
])
def test_compact_and_full_counters_print_the_same(synth, args):
    want = run_oracle(args + [synth])
    assert run_cli(args + [synth]) == want
    assert run_cli(args + [synth], env={"SBX_COMPACT": "0"}) == want


def test_compact_with_samples(synth_ms):
    for args in (["window", "-w", "2000", "-T", "5"], ["window", "-w", "700", "-q", "13"], ["region", "-L", "c1:500-90000", "-T", "10"]):
        want = run_oracle(args + [synth_ms])
        assert run_cli(args + [synth_ms]) == want, args
        assert run_cli(args + [synth_ms], env={"SBX_COMPACT": "0"}) == want, args


def test_the_abi_says_what_a_window_run_keeps(synth):
    with sambamba_amd.Depth(synth) as d:
        d.set_params()
        d.run()
        base_cov = d.base_counters(0, 0, 300000, with_covered=True)[1]
    with sambamba_amd.Depth(synth) as d:
        d.set_params(mode=sambamba_amd.SBX_MODE_WINDOW, window=1000)
        d.run()
        assert np.array_equal(d.covered(0, 0, 300000), base_cov)          # `covered` is always there
        with pytest.raises(sambamba_amd.SbxError):
            d.base_counters(0, 0, 1000)                                    # the seven counters are not
        nr, nb, _ = d.window_stats(0, 0, 300)
        assert nb.sum() > 0
