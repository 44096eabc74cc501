"""Who checks the checker: bench.py's whole-share parity functions on the CPU, with a stand-in for the device whose answers come from the
oracle itself -- they must accept identical results with coverage 1.0 and name the slice / region of a single tampered byte or count."""
import os
import subprocess
import sys

import pytest

from tests.util import GOLDEN, ROOT, ensure_oracle, ORACLE_BIN

sys.path.insert(0, ROOT)
import bench  # noqa: E402

BAM = os.path.join(GOLDEN, "issue225.bam")
LEN = 16571          # chrM of the fixture


class OracleAsDevice:
    """ref_names + format_base_rows / rows of regions, answered by the oracle (optionally with one flaw)"""
    ref_names = ["chrM"]

    def __init__(self, flaw_at=None):
        self.flaw_at = flaw_at

    def format_base_rows(self, ref, a, b):
        out = subprocess.run([ORACLE_BIN, "base", "-L", "chrM:%d-%d" % (a + 1, b), BAM], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        rows = out.split(b"\n", 1)[1] if out else b""
        if self.flaw_at is not None and a <= self.flaw_at < b and rows:
            rows = rows[:-2] + (b"8" if rows[-2:-1] != b"8" else b"9") + rows[-1:]       # one digit of the slice's last row
        return rows


@pytest.fixture(scope="module", autouse=True)
def _oracle():
    ensure_oracle()


def test_full_text_accepts_identical_rows_and_covers_every_position():
    r = bench.parity_full_text(OracleAsDevice(), BAM, [(0, 0, LEN)], [], min_slices=8)
    assert r["ok"] and r["coverage"] == 1.0 and r["slices"] >= 4 and r["text_bytes"] > 0 and r["mismatching_slices"] == []


def test_full_text_names_the_slice_of_one_wrong_digit():
    r = bench.parity_full_text(OracleAsDevice(flaw_at=9000), BAM, [(0, 0, LEN)], [], min_slices=8)
    assert not r["ok"] and len(r["mismatching_slices"]) == 1
    ref, a, b = r["mismatching_slices"][0]
    assert a <= 9000 < b


def test_full_regions_checks_every_row():
    regs = [(0, k * 500, (k + 1) * 500) for k in range(LEN // 500)]
    bed = "".join("chrM\t%d\t%d\n" % (a, b) for _, a, b in regs)
    tmp = os.path.join(bench.tmp_dir(), "sbx_test_parity_%d.bed" % os.getpid())
    open(tmp, "w").write(bed)
    out = subprocess.run([ORACLE_BIN, "region", "-L", tmp, BAM], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    os.remove(tmp)
    rows = [(int(f[3]), f[4]) for f in (ln.split("\t") for ln in out.splitlines() if ln and not ln.startswith("#"))]
    assert len(rows) == len(regs)
    ok = bench.parity_full_regions(OracleAsDevice(), BAM, regs, rows)
    assert ok["ok"] and ok["regions"] == len(regs)
    bad_rows = list(rows)
    bad_rows[7] = (bad_rows[7][0] + 1, bad_rows[7][1])
    bad = bench.parity_full_regions(OracleAsDevice(), BAM, regs, bad_rows)
    assert not bad["ok"] and bad["mismatches"][0][1:3] == [regs[7][1], regs[7][2]]
