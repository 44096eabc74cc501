"""`sbx-depth base` through the three-stage pipeline (file -> device | kernels | device -> text over two alternating contexts,
cli.cpp): the text must be byte-identical to the one-pass form and to the oracle, whatever the slice size -- cuts inside
contigs, slices without reads, alignments hanging over a contig end, -m pairs straddling cuts, several samples."""
import os
import subprocess

import pytest

from sambamba_amd import cli_path
from tests.util import gen_bam, run_cli, run_oracle

pytestmark = pytest.mark.gpu


def run_pipelined(args, slice_positions, orderly=False):
    env = dict(os.environ, SBX_FORCE_PIPELINE="1", SBX_SLICE_POSITIONS=str(slice_positions), SBX_STREAM_PIECE="30000")
    if orderly:
        env["SBX_ORDERLY_EXIT"] = "1"
    r = subprocess.run([cli_path()] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    return r.stdout


@pytest.fixture(scope="module")
def bam(tmp_path_factory):
    d = tmp_path_factory.mktemp("pipe")
    return gen_bam(str(d / "p.bam"), "c1:300000,cEmpty:5000,c2:70000,c3:1500,c4:120000", coverage=25, seed=91,
                   extra=["--samples", "2", "--insert-mean", "260", "--insert-sd", "40", "--tie-free-overlaps"])


@pytest.mark.parametrize("slice_positions", [1024, 40000, 100000, 10**9])
@pytest.mark.parametrize("args", [["base"], ["base", "-c", "3", "-C", "40", "-a"], ["base", "-m", "-q", "20"], ["base", "--combined", "-F", "mapping_quality > 10"]])
def test_pipelined_base_equals_one_pass_and_oracle(bam, args, slice_positions):
    want = run_oracle(args + [bam])
    env_off = dict(os.environ, SBX_NO_PIPELINE="1")
    one_pass = subprocess.run([cli_path()] + args + [bam], stdout=subprocess.PIPE, env=env_off, check=True).stdout
    assert one_pass == want
    assert run_pipelined(args + [bam], slice_positions) == want
    assert len(want) > 100000


def test_pipelined_output_file_and_orderly_exit(bam, tmp_path):
    out = str(tmp_path / "o.txt")
    want = run_oracle(["base", bam])
    run_pipelined(["base", "-o", out, bam], 50000)
    assert open(out, "rb").read() == want
    assert run_pipelined(["base", bam], 50000, orderly=True) == want


def test_pipeline_reports_errors(tmp_path, bam):
    env = dict(os.environ, SBX_FORCE_PIPELINE="1", SBX_SLICE_POSITIONS="50000")
    r = subprocess.run([cli_path(), "base", "-F", "[RG] =~ /(a)\\1/", bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert r.returncode == 1 and b"sambamba-depth:" in r.stderr
