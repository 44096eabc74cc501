"""The record-chain repair path (index.hip k_chain_repair) must be invisible in the results: forcing the
serial repair to start at various blocks (SBX_FORCE_REPAIR, a debug hook of the engine) has to give
exactly the record counts and counters of the all-guesses-right fast path."""
import os

import numpy as np
import pytest

from tests.util import gen_bam

pytestmark = pytest.mark.gpu


def test_forced_chain_repair_is_result_neutral(tmp_path):
    import sambamba_amd
    p = gen_bam(str(tmp_path / "rep.bam"), "chrA:2500000,chrB:600000", coverage=30, seed=3)

    def run():
        with sambamba_amd.Depth(p) as d:
            d.set_params()
            st = d.run()
            return st, d.base_counters(0, 0, 2500000), d.base_counters(1, 0, 600000)

    os.environ.pop("SBX_FORCE_REPAIR", None)
    s0, a0, b0 = run()
    assert s0["n_records"] > 500000
    try:
        for start in ("0", "1", "7", "130", "400"):
            os.environ["SBX_FORCE_REPAIR"] = start
            s1, a1, b1 = run()
            assert s1["n_records"] == s0["n_records"], start
            assert s1["n_admitted"] == s0["n_admitted"], start
            assert np.array_equal(a0, a1) and np.array_equal(b0, b1), start
    finally:
        os.environ.pop("SBX_FORCE_REPAIR", None)
