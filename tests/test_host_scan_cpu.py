"""Host logic without a device: the BGZF header scan of sbx_open cut into pieces at hinted block starts (host_io.hpp
scan_bgzf) against the serial scan, including hints that are not block starts (a stale index must only cost the speed-up)."""
import os
import subprocess

from tests.util import ROOT, gen_bam


def test_block_table_from_pieces_equals_serial(tmp_path):
    exe = str(tmp_path / "scan_check")
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "host", "scan_check.cpp")])
    bam = gen_bam(str(tmp_path / "s.bam"), "c1:400000,c2:150000", coverage=20, seed=3)
    r = subprocess.run([exe, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (r.stdout.decode(), r.stderr.decode())
    assert b"pieces ok, stale hints ok, mixed ok" in r.stdout
    for fixture in ("issue225.bam", "issue_204.bam"):
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", fixture)], stdout=subprocess.PIPE)
        assert r.returncode == 0, r.stdout.decode()
