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
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-sample-reads", "20000", "--parity-windows", "4", "--e2e-pause", "0.3"] + extra
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
    assert d["parity_checked"]["ok"] and d["parity_checked"]["windows"] >= 1
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    assert d["e2e"]["Mreads_per_s"] > 0
    # round 5: the whole share is checked, not samples -- every row of text (configs 2, 5), every window / BED region of a contig (3, 4)
    cfg = int(extra[1])
    if cfg in (2, 5):
        assert d["parity_checked"]["coverage"] == 1.0 and d["parity_checked"]["full_text"]["ok"]
        assert d["parity_checked"]["full_text"]["slices"] >= 64
    else:
        assert d["parity_checked"]["whole_contig"]["ok"] and d["parity_checked"]["whole_contig"]["regions"] >= 1
    # the token streams are an intermediate, not algorithmic bytes; the fused-path figure is a first-class field
    assert d["kernels"]["huffman_decode"]["intermediate_bytes"] > 0
    assert d["kernels"]["huffman_decode"]["algorithmic_bytes"] < d["kernels"]["lz77_resolve"]["algorithmic_bytes"]
    assert 0 < d["roofline"]["path_frac"] < 1
    # round 6: the older definition beside the redefined one; the pass including its text (configs 2, 5); config 4 timed with the work-list
    # cache off, the cached re-run beside it
    assert d["roofline"]["path_frac_incl_counters"] >= d["roofline"]["path_frac"] and "traffic_total" in d["roofline"]
    if cfg in (2, 5):
        dt = d["device_text"]
        assert "error" not in dt, dt
        assert dt["text_equals_host_copy_path"] and dt["text_bytes"] > 100000 and dt["ms_per_step"] >= d["ms_per_step"] * 0.8
    if cfg == 4:
        assert d["rerun_cached"]["ms_per_step"] > 0
    # counters are joined only from a pass stamped with these very kernel sources: a development-scale line has none
    assert d["roofline"]["traffic"] is None


def test_two_ranks_report_what_the_collective_layer_saw(tmp_path):
    """bench.py --gpus 2 (two ranks sharing the test box's GPU over gloo): the line carries `collective` -- backend, world size,
    an all-reduce of ones, per-rank device identity and kernel times -- so that a scaling record can be audited."""
    env = dict(os.environ, PYTHONPATH=ROOT, SBX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29611",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--length", "3000000", "--no-cpu-baseline", "--no-e2e",
           "--no-side-runs", "--parity-windows", "2"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    c = d["collective"]
    assert c["world_size"] == 2 and c["allreduce_of_ones"] == 2.0 and c["backend"] == "gloo" and len(c["ranks"]) == 2
    assert sorted(e["rank"] for e in c["ranks"]) == [0, 1] and all(e["kernel_ms"]["lz77_resolve"] > 0 for e in c["ranks"])
    assert d["n_gpus"] == 2 and d["parity_checked"]["ok_all_ranks"]
