"""`sbx-depth base` through the three-stage pipeline (file -> device | kernels | device -> text over two alternating contexts,
cli.cpp): the text must be byte-identical to the one-pass form and to the oracle, whatever the slice size -- cuts inside
contigs, slices without reads, alignments hanging over a contig end, -m pairs straddling cuts, several samples."""
import os
import subprocess

import pytest

from sambamba_amd import cli_path
from tests.util import gen_bam, run_cli, run_oracle

pytestmark = pytest.mark.gpu


def run_pipelined(args, slice_positions, orderly=False, contexts=None):
    """contexts: None = the form of the process (ONE context as one process, the default since round 5), 1 / 2 = forced"""
    env = dict(os.environ, SBX_FORCE_PIPELINE="1", SBX_SLICE_POSITIONS=str(slice_positions), SBX_STREAM_PIECE="30000")
    if contexts:
        env["SBX_PIPELINE_CONTEXTS"] = str(contexts)
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


PIPE_ARGS = [["base"], ["base", "-c", "3", "-C", "40", "-a"], ["base", "-m", "-q", "20"], ["base", "--combined", "-F", "mapping_quality > 10"]]


# (slices of one tile -- 1,024 positions, hundreds of slices -- with two of the option sets only: they cost 10 s each)
@pytest.mark.parametrize("args,slice_positions", [(a, sp) for a in PIPE_ARGS for sp in (40000, 100000, 10**9)] + [(PIPE_ARGS[0], 1024), (PIPE_ARGS[2], 1024)])
def test_pipelined_base_equals_one_pass_and_oracle(bam, args, slice_positions):
    want = run_oracle(args + [bam])
    env_off = dict(os.environ, SBX_NO_PIPELINE="1")
    one_pass = subprocess.run([cli_path()] + args + [bam], stdout=subprocess.PIPE, env=env_off, check=True).stdout
    assert one_pass == want
    assert run_pipelined(args + [bam], slice_positions) == want                 # one context: upload of slice k + 1 next to the text of slice k
    assert run_pipelined(args + [bam], slice_positions, contexts=2) == want     # two contexts: the detached child's form
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


# ---- process model (cli.cpp main): the work runs in a child, the command returns when the output is complete -----------------
@pytest.mark.parametrize("mode_args", [["base"], ["window", "-w", "500"], ["base", "-L", "c1:1000-90000"]])
def test_detached_and_single_process_print_the_same(bam, mode_args):
    want = run_oracle(mode_args + [bam])
    a = subprocess.run([cli_path()] + mode_args + [bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, SBX_DETACH="1"))
    b = subprocess.run([cli_path()] + mode_args + [bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-300:], b.stderr[-300:])
    assert a.stdout == want and b.stdout == want


def test_output_is_complete_when_the_command_returns(bam, tmp_path):
    """The parent may only return once the worker has written and closed everything: the file is read right after
    subprocess.run returns (only the parent is waited for), and a consumer on a pipe sees the whole text and end-of-file."""
    want = run_oracle(["base", bam])
    out = str(tmp_path / "o.txt")
    det = dict(os.environ, SBX_DETACH="1")
    for k in range(3):
        r = subprocess.run([cli_path(), "base", "-o", out, bam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=det)
        assert r.returncode == 0
        assert open(out, "rb").read() == want
        os.unlink(out)
    sh = subprocess.run("%s base %s | md5sum" % (cli_path(), bam), shell=True, stdout=subprocess.PIPE, check=True, env=det)
    import hashlib
    assert sh.stdout.split()[0].decode() == hashlib.md5(want).hexdigest()
    env = dict(os.environ, SBX_DETACH="1", SBX_FORCE_PIPELINE="1", SBX_SLICE_POSITIONS="50000")
    sh = subprocess.run("%s base %s | md5sum" % (cli_path(), bam), shell=True, stdout=subprocess.PIPE, check=True, env=env)
    assert sh.stdout.split()[0].decode() == hashlib.md5(want).hexdigest()


def test_failures_keep_their_status_and_message(tmp_path, bam):
    bad = str(tmp_path / "bad.bam")
    raw = bytearray(open(bam, "rb").read())
    raw[len(raw) // 2] ^= 0xFF                     # corrupt a deflate stream in the middle of the file
    raw[len(raw) // 2 + 1] ^= 0xFF
    open(bad, "wb").write(raw)
    open(bad + ".bai", "wb").write(open(bam + ".bai", "rb").read())
    a = subprocess.run([cli_path(), "base", bad], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, SBX_DETACH="1"))
    b = subprocess.run([cli_path(), "base", bad], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == b.returncode
    assert a.stderr == b.stderr
    if a.returncode:
        assert a.stderr.startswith(b"sambamba-depth: ")


def test_upload_pool_moves_every_byte(tmp_path):
    """A file of several staging buffers (32 MiB each) through the reader pool with different thread counts: the inflated
    stream is compared block by block inside the run (status per block), the text with the oracle's."""
    path = gen_bam(str(tmp_path / "big.bam"), "c1:4000000", coverage=40, seed=5)
    assert os.path.getsize(path) > 80 << 20
    want = run_oracle(["base", "-L", "c1:3990000-4000000", path])
    want_windows = run_oracle(["window", "-w", "100000", path])
    for thr in ("1", "3", "12"):
        env = dict(os.environ, SBX_UPLOAD_THREADS=thr, SBX_NO_PIPELINE="1")
        whole = subprocess.run([cli_path(), "window", "-w", "100000", path], stdout=subprocess.PIPE, env=env, check=True).stdout
        assert whole == want_windows
        got = subprocess.run([cli_path(), "base", "-L", "c1:3990000-4000000", path], stdout=subprocess.PIPE, env=env, check=True).stdout
        assert got == want
