"""bench.py on one GPU, every BASELINE config at development scale: one JSON line, the contract's fields, and the parity of the
timed results against the oracle (`parity_checked.ok`; a mismatch makes bench.py exit non-zero)."""
import json
import os
import subprocess
import sys

import pytest

from tests.util import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [
    ["--config", "2", "--length", "4000000"],
    ["--config", "3", "--scale", "0.002"],
    ["--config", "4", "--scale", "0.002"],
    ["--config", "5", "--scale", "0.01", "--coverage", "60"],
])
def test_bench_line(tmp_path, extra):
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "20000", "--parity-windows", "4"] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity_checked"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    # the accounting checks itself: every kernel's algorithmic bytes / time stays below the HBM peak (bench.py exits non-zero otherwise)
    assert "accounting_error" not in d
    for k, e in d["kernels"].items():
        assert 0 < e["frac_of_hbm_peak"] <= 1, (k, e)
    assert d["kernels"]["decode_accumulate"]["algorithmic_bytes"] < d["kernels"]["lz77_resolve"]["algorithmic_bytes"]
    assert d["parity_checked"]["ok"] and d["parity_checked"]["windows"] >= 1
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    assert d["e2e"]["Mreads_per_s"] > 0
