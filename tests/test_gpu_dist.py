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


ONE_CONTIG_ARGS = [
    ["base"],
    ["base", "-m", "-q", "20"],
    ["base", "-c", "0", "-a", "-C", "45"],
    ["base", "-L", "chrOne:30000-150000", "-m"],
    ["window", "-w", "1000", "-m", "-T", "20"],
]


# (round 6: every option set at two ranks, two of them at three -- a torch.distributed job takes seconds to come up; the one-process
# form of the same sharding, `sbx-depth --gpus N`, runs the full matrix in tests/test_gpu_cli_sharded.py)
@pytest.mark.parametrize("args,world", [(a, 2) for a in ONE_CONTIG_ARGS] + [(ONE_CONTIG_ARGS[1], 3), (ONE_CONTIG_ARGS[4], 3)])
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


GENOME_ARGS = [
    ["region", "-L", "BED", "-T", "5", "-T", "20"],
    ["region", "-L", "BED", "-m", "-q", "20", "-T", "3"],
    ["window", "-w", "1000", "-T", "10"],
    ["window", "-w", "700", "-m", "-q", "13", "--combined"],
]


@pytest.mark.parametrize("args,world", [(a, 2) for a in GENOME_ARGS] + [(GENOME_ARGS[1], 3), (GENOME_ARGS[2], 3)])
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
        assert ar["ms_export_and_allreduce"] is None and "gloo" in ar["collective_backend"]      # a collective through host memory is not a number
    elif mode == "strong":
        assert d["scaling"] == "strong" and d["strong_one_contig"] is None and d["reads_total"] == 600000
    else:
        assert d["scaling"] == "weak" and d["reads_total"] == 2 * 600000


@pytest.mark.parametrize("cfg", ["3", "4"])
def test_bench_multi_rank_whole_genome(tmp_path, cfg):
    """The north star's scaling input: `bench.py --gpus N --config 3` (whole genome, window mode) and `--config 4` (exome regions on
    it) -- ONE BAM sharded by position over the ranks (here two ranks on the box's GPU over gloo, a 1/500 genome)."""
    import json
    env = dict(os.environ, SBX_BENCH_BACKEND="gloo", PYTHONPATH=ROOT, TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "2987" + str(3 + int(cfg)), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--config", cfg, "--scale", "0.002", "--no-cpu-baseline", "--no-e2e", "--parity-windows", "4"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["parity_checked"]["ok"] and d["parity_checked"]["ok_all_ranks"]
    assert "accounting_error" not in d and "sharded over the ranks" in d["config"]["sharding"]
    assert d["metric"] == ("depth_window_Mreads_per_s" if cfg == "3" else "depth_region_Mreads_per_s")


# ---- round 3: per-rank output ranges, the all-reduce form, RCCL at world size 1, data-driven mate slack ------------------
def run_dist_to_file(args, world, port, out_path, env_extra=None):
    env = dict(os.environ, SBX_BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "sambamba_amd.dist_depth"] + args + ["-o", out_path]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    return open(out_path, "rb").read()


@pytest.mark.parametrize("args,world", [(["base"], 2), (["base", "-c", "0"], 3), (["base", "-m", "-q", "20", "-a"], 2)])
def test_every_rank_writes_its_own_byte_range(genome, one_contig, tmp_path, args, world):
    """`base -o`: no text funnel -- each rank pwrites at the offset the exclusive scan of the measured sizes gives it, in pieces
    (SBX_STREAM_PIECE makes the pieces small enough that every rank writes several)."""
    for k, bam in enumerate((genome[0], one_contig)):
        want = run_cli(args + [bam])
        got = run_dist_to_file([args[0], bam] + args[1:], world, 29900 + 10 * world + k + 3 * len(args), str(tmp_path / ("o%d.txt" % k)),
                               {"SBX_STREAM_PIECE": "20000"})
        assert got == want, (args, k)


@pytest.mark.parametrize("world", [3])
def test_allreduce_form_prints_the_same_text(genome, one_contig, tmp_path, world):
    """`base --reduce allreduce`: reads partitioned by start position, per-position counters summed by an all-reduce."""
    for k, bam in enumerate((one_contig, genome[0])):
        want = run_cli(["base", bam])
        got = run_dist_to_file(["base", bam, "--reduce", "allreduce"], world, 29940 + 10 * world + k, str(tmp_path / ("a%d.txt" % k)))
        assert got == want, k


def test_rccl_backend_at_world_size_one(one_contig, tmp_path):
    """The `nccl` (RCCL) branch of the driver on the box's one GPU: process group, all_reduce of flags, the all-reduce of the
    counter tensors on the device -- world size 1, so that the code path at least runs on hardware."""
    env = dict(os.environ, SBX_BENCH_BACKEND="nccl", SBX_DIST_FORCE_GROUP="1", PYTHONPATH=ROOT, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29983")
    for args, ref in ((["window", "-w", "1000", "-T", "10"], None), (["base", "--reduce", "allreduce"], ["base"])):
        out = str(tmp_path / "n.txt")
        r = subprocess.run([sys.executable, "-m", "sambamba_amd.dist_depth", args[0], one_contig] + args[1:] + ["-o", out],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        assert open(out, "rb").read() == run_cli((ref or args) + [one_contig])


def test_owned_runs_add_up_to_the_whole(one_contig):
    """sbx_run_interval_owned: the reads are partitioned by start position, so the counters of the owners' runs add up to the
    counters of the whole run, position by position (also across the cuts, where both owners contribute)."""
    import numpy as np
    import sambamba_amd
    with sambamba_amd.Depth(one_contig) as d:
        d.set_params()
        d.run()
        L = d.ref_lengths[0]
        whole = d.base_counters(0, 0, L)
        total = np.zeros_like(whole)
        cuts = [0, 50176, 50176 + 1024, 131072, L]
        n_owned = 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            st = d.run_interval_owned(0, a, b)
            n_owned += st["n_admitted"]
            total += d.base_counters(0, 0, L)
        assert np.array_equal(total, whole)
        d.run()
        assert n_owned == d.run()["n_admitted"]


def test_region_starting_beyond_its_contig_keeps_its_row(genome, tmp_path):
    """A BED region that starts at or beyond the end of its contig is owned by the rank holding the contig's last position
    (the single-GPU CLI prints a zero row for it)."""
    bam, _ = genome
    bed = str(tmp_path / "beyond.bed")
    with open(bed, "w") as fh:
        fh.write("c1\t100\t9000\nc2\t30000\t30500\nc3\t69990\t70100\nc5\t60000\t60010\nc1\t89999\t95000\n")
    want = run_cli(["region", "-L", bed, bam])
    for world in (2, 3):
        got = run_sharded(["region", bam, "-L", bed], world, 29960 + world)
        assert got == want


def test_mate_slack_follows_the_longest_alignment(tmp_path):
    """window -m across a cut with spliced reads: a read's mate ends tens of kilobases before the slice (an N operation
    longer than one linear-index window); the left margin of the fetch grows to the longest alignment of the run."""
    from tests import bamgen as bg
    import random
    rng = random.Random(5)
    L = 400000
    recs = []
    def seq(n):
        return "".join(rng.choice("ACGT") for _ in range(n))
    k = 0
    for pos in range(1000, L - 80000, 1500):
        k += 1
        # a spliced read (60M <skip>N 60M) and its mate: the mate overlaps the FIRST exon (so that, right of a cut inside the
        # skip, the spliced read is `past` only if the mate -- which ends far left of the cut -- is in the run) or the second
        skip = rng.choice([200, 30000, 52000])
        s1, s2 = seq(120), seq(100)
        recs.append((pos, bg.make_record(0, pos, "60M%dN60M" % skip, s1, 30, name="p%d" % k, flag=99)))
        at = pos + 20 if k % 2 else pos + 60 + skip + 20
        recs.append((at, bg.make_record(0, at, "100M", s2, 31, name="p%d" % k, flag=147)))
    recs.sort(key=lambda x: x[0])
    bam = str(tmp_path / "spliced.bam")
    bg.write_bam(bam, [("chrS", L)], [r[1] for r in recs])
    args = ["window", "-w", "1000", "-m", "-T", "1"]
    want = run_cli(args + [bam])
    for world in (2, 3):
        got = run_sharded([args[0], bam] + args[1:], world, 29970 + world)
        assert got == want
