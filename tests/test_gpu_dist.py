"""`depth base|region|window` of one BAM sharded over the ranks of a torch.distributed job (sambamba_amd.dist_depth,
BASELINE config 4's shape): every rank runs its contigs through the device, rank 0 prints -- byte-identical to the
single-GPU CLI.  Ranks share the one GPU of the test box and talk through gloo; on a multi-GPU node the same
module runs one rank per GPU over RCCL."""
import os
import subprocess
import sys

import pytest

from tests.util import ROOT, gen_bam, run_cli

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    d = tmp_path_factory.mktemp("dist")
    bam = gen_bam(str(d / "g.bam"), "c1:90000,c2:30000,cNone:2500,c3:70000,c4:900,c5:50000", coverage=12, seed=41,
                  extra=["--samples", "2", "--insert-mean", "260", "--insert-sd", "40", "--tie-free-overlaps"])
    bed = str(d / "r.bed")
    with open(bed, "w") as fh:
        fh.write("c3\t100\t9000\tx\nc1\t5000\t5100\ty\nc5\t49000\t50000\tz\ncNone\t10\t500\tq\nc1\t80000\t89000\tw\nc2\t0\t30000\tv\n")
    return bam, bed


def run_sharded(args, world, port, first=b"# "):
    env = dict(os.environ, SBX_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "sambamba_amd.dist_depth"] + args
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=600)
    err = r.stderr.decode()
    own = [ln for ln in err.splitlines() if "sambamba-depth" in ln or "Error" in ln or "error" in ln]
    assert r.returncode == 0, "\n".join(own[:30]) + "\n...\n" + err[-600:]
    # gloo announces its connections on stdout while the process group comes up (the ranks' lines interleave);
    # rank 0 prints afterwards, starting with the "# chrom ..." header
    at = r.stdout.find(first)
    return r.stdout[at:] if at >= 0 else r.stdout


@pytest.fixture(scope="module")
def one_contig(tmp_path_factory):
    """ONE contig (BASELINE config 5's shape): shards only by position; most mates overlap, some straddle every cut."""
    d = tmp_path_factory.mktemp("dist1")
    return gen_bam(str(d / "one.bam"), "chrOne:200000", coverage=40, seed=43,
                   extra=["--insert-mean", "250", "--insert-sd", "40", "--tie-free-overlaps"])


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("args", [
    ["base"],
    ["base", "-m", "-q", "20"],
    ["base", "-c", "0", "-a", "-C", "45"],
    ["base", "-L", "chrOne:30000-150000", "-m"],
    ["window", "-w", "1000", "-m", "-T", "20"],
])
def test_position_sharded_single_contig_equals_single_gpu_cli(one_contig, args, world):
    want = run_cli(args + [one_contig])
    assert len(want) > 1000
    port = 29500 + 10 * world + (sum(map(ord, " ".join(args))) % 10)
    got = run_sharded([args[0], one_contig] + args[1:], world, port, first=b"REF" if args[0] == "base" else b"# ")
    assert got == want


def test_position_sharded_base_on_a_genome(genome):
    bam, bed = genome
    for args in (["base"], ["base", "-c", "0"], ["base", "-L", bed, "-q", "10"]):
        want = run_cli(args + [bam])
        got = run_sharded([args[0], bam] + args[1:], 3, 29590 + len(args), first=b"REF")
        assert got == want, args


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("args", [
    ["region", "-L", "BED", "-T", "5", "-T", "20"],
    ["region", "-L", "BED", "-m", "-q", "20", "-T", "3"],
    ["window", "-w", "1000", "-T", "10"],
    ["window", "-w", "700", "-m", "-q", "13", "--combined"],
])
def test_sharded_equals_single_gpu_cli(genome, args, world):
    bam, bed = genome
    if "--combined" in args and "-m" in args:
        args = [a for a in args if a != "--combined"]      # (undefined in the reference with several samples)
    a = [bed if x == "BED" else x for x in args]
    want = run_cli([a[0]] + a[1:] + [bam])
    port = 29700 + 10 * world + (sum(map(ord, " ".join(args))) % 10)      # one port per case: no reuse while a socket lingers
    got = run_sharded([a[0], bam] + a[1:], world, port)
    assert got == want


@pytest.mark.parametrize("mode", ["auto", "strong", "replicas"])
def test_bench_multi_rank_line(tmp_path, mode):
    """bench.py as the driver launches it for N > 1 (here: two ranks sharing the box's GPU over gloo, 3 Mbp contigs):
    one JSON line, parity of the timed results against the oracle on every rank.  Default mode: ONE BAM with a contig per
    rank, sharded by position, plus the side measurements (the single contig cut in two; the all-reduce option)."""
    import json
    env = dict(os.environ, SBX_BENCH_BACKEND="gloo", PYTHONPATH=ROOT, TMPDIR=str(tmp_path))
    port = {"auto": "29871", "strong": "29872", "replicas": "29873"}[mode]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--length", "3000000", "--no-cpu-baseline", "--no-e2e", "--parity-windows", "3", "--mode", mode]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["parity_checked"]["ok"] and d["parity_checked"]["ok_all_ranks"]
    assert "accounting_error" not in d
    if mode == "auto":
        assert d["scaling"] == "weak" and d["reads_total"] == 2 * 600000
        assert "sharded over the ranks" in d["config"]["sharding"]
        assert d["strong_one_contig"]["parity_ok"] and d["strong_one_contig"]["value"] > 0
        ar = d["allreduce_option"]
        assert ar and "error" not in ar, ar
        assert ar["parity_ok"] and ar["allreduce_bytes_per_rank"] > 0
    elif mode == "strong":
        assert d["scaling"] == "strong" and d["strong_one_contig"] is None and d["reads_total"] == 600000
    else:
        assert d["scaling"] == "weak" and d["reads_total"] == 2 * 600000
