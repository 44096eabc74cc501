#!/usr/bin/env python3
"""bench.py -- `sambamba depth base` hot path on MI355X: BGZF inflate -> BAM record index ->
CIGAR-walk per-position coverage counters, all on the device (libsbx_depth.so).

Contract (driver): python bench.py --gpus N --steps K --warmup W   (N>1: launched under
torch.distributed.run, one rank per GPU).  One "step" = one full pass of the hot path over the
synthetic coordinate-sorted BAM of BASELINE.json configs[1] (chr1, L=248,956,422, 30x, 2x150 bp
paired reads), whose compressed bytes are already resident in HBM when the timed
region starts.  Each rank processes its own contig-sized BAM (seed + rank): weak scaling, no
data-path collective (position-sharded outputs are disjoint).

Prints ONE JSON line on rank 0 with the driver's fields plus `roofline` (dominant kernel,
algorithmic bytes / HIP-event kernel time) and `cpu_baseline` (the CPU oracle -- a literal port
of the reference algorithm -- timed on a bounded sample of the same BAM on this box's host).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHR1_LEN = 248_956_422
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def ensure_built():
    from sambamba_amd import build as b
    b.build()
    for d in ("oracle", "tools"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, d)])


def workload_path(length, coverage, seed, level, codec):
    key = "chr1_%d_%g_%x_%d_%s" % (length, coverage, seed, level, codec)
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    return os.path.join(tmp, "sbx_bench_%s.bam" % hashlib.sha1(key.encode()).hexdigest()[:12]), key


def generate(path, length, coverage, seed, level, codec):
    """Seeded synthetic BAM+BAI (tools/gen_bam.cpp); cached on tmpfs across invocations on the same box."""
    meta = path + ".json"
    if os.path.exists(path) and os.path.exists(path + ".bai") and os.path.exists(meta):
        return json.load(open(meta))
    t0 = time.time()
    tmp_out = path + ".tmp%d" % os.getpid()
    out = subprocess.check_output([os.path.join(ROOT, "tools", "gen_bam"), "--out", tmp_out, "--contigs",
                                   "chr1:%d" % length, "--coverage", str(coverage), "--seed", hex(seed),
                                   "--level", str(level), "--codec", codec])
    info = json.loads(out.decode().strip().splitlines()[-1])
    info["gen_seconds"] = time.time() - t0
    os.replace(tmp_out + ".bai", path + ".bai")
    os.replace(tmp_out, path)
    json.dump(info, open(meta, "w"))
    return info


def cpu_baseline(bam, sample_reads):
    """Reference-algorithm CPU stand-in: the oracle CLI on the first `sample_reads` records of the same
    BAM, output to /dev/null.  Structured like the reference: zlib inflate on a pool of worker threads
    with in-order delivery (`-t`), then the single-threaded sweep-line pileup + text formatting.  Timed
    twice -- sambamba's default (`-t 0`: everything on one thread, depth.d:1081,1154) and with 16 inflate
    workers -- and the faster one is reported (the serial sweep bounds what more threads can buy)."""
    exe = os.path.join(ROOT, "oracle", "depth_oracle")
    env = dict(os.environ, ORC_STATS="1")
    runs = []
    for workers in (0, max(1, min(16, (os.cpu_count() or 2) - 1))):
        t0 = time.time()
        r = subprocess.run([exe, "base", "-t", str(workers), "--max-reads", str(sample_reads), bam],
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        dt = time.time() - t0
        if r.returncode != 0:
            continue
        secs, seen = dt, sample_reads
        for line in r.stderr.decode().splitlines():
            if line.startswith("[oracle]"):
                parts = line.split()
                secs = float(parts[1])
                seen = int(parts[3])
        runs.append({"workers": workers, "mreads_per_s": seen / secs / 1e6, "secs": secs, "seen": seen})
    if not runs:
        return None
    best = max(runs, key=lambda x: x["mreads_per_s"])
    return {"value": round(best["mreads_per_s"], 4), "unit": "Mreads/s", "cores": best["workers"] + 1, "kind": "port",
            "sample": "first %d records of the same BAM, depth base -> /dev/null, %.1f s wall" % (best["seen"], best["secs"]),
            "runs": [{"inflate_workers": x["workers"], "Mreads_per_s": round(x["mreads_per_s"], 4)} for x in runs],
            "threads_note": "sambamba depth default is 0 worker threads (fully serial); the sweep-line pileup and the "
                            "text output are single-threaded in the reference whatever -t says; nproc=%d" % (os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--length", type=int, default=int(os.environ.get("SBX_BENCH_LEN", CHR1_LEN)),
                    help="contig length (default: chr1; smaller values are for development only)")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--codec", default=os.environ.get("SBX_BENCH_CODEC", "zlib"))
    ap.add_argument("--cpu-sample-reads", type=int, default=int(os.environ.get("SBX_BENCH_CPU_READS", 3_000_000)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # one rank per GPU; SBX_BENCH_BACKEND=gloo lets the multi-rank path be exercised on a box with fewer GPUs
    # than ranks (ranks then share devices and the timing reductions travel through host memory)
    backend = os.environ.get("SBX_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if world > 1 else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if rank == 0:
        ensure_built()
    if dist:
        dist.barrier()
    import sambamba_amd

    # One seeded contig-sized BAM per node, generated once (rank 0, all host cores) and processed by every
    # rank on its own GPU: per-GPU work is fixed as N grows (weak scaling) and ranks never exchange data.
    seed = 0x5A4D0002
    path, key = workload_path(args.length, args.coverage, seed, args.level, args.codec)
    if rank == 0:
        info = generate(path, args.length, args.coverage, seed, args.level, args.codec)
    if dist:
        dist.barrier()
    if rank != 0:
        info = json.load(open(path + ".json"))

    d = sambamba_amd.Depth(path, device=dev_index)
    d.set_params()             # depth base, default filter, -q 0
    d.preload()                # compressed BAM resident in HBM before the timed region

    def sync():
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    last = None
    for _ in range(args.warmup):
        last = d.run()
    sync()
    t0 = time.perf_counter()
    kstats = []
    for _ in range(args.steps):
        last = d.run()
        kstats.append(last)
    sync()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([float(last["n_records"]), float(last["n_admitted"])], dtype=torch.float64, device=red_dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_reads, total_admitted = float(r[0].item()), float(r[1].item())
    else:
        total_reads, total_admitted = float(last["n_records"]), float(last["n_admitted"])

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        reads_per_s = total_reads / (elapsed / args.steps)
        read_len = 150
        avg = lambda k: sum(s[k] for s in kstats) / len(kstats)
        kern = {"huffman_decode": avg("ms_huffman"), "lz77_resolve": avg("ms_lz77"), "record_index": avg("ms_index"),
                "decode_accumulate": avg("ms_accumulate")}
        dom = max(kern, key=kern.get)
        comp, unc, cnt = last["compressed_bytes"], last["uncompressed_bytes"], last["counter_bytes"]
        # algorithmic bytes per launch
        alg = {"huffman_decode": comp + 0.57 * unc,          # compressed in; literal + match-entry streams out (~0.57 B per output byte)
               "lz77_resolve": 0.57 * unc + unc,              # token streams in; inflated bytes out
               "record_index": unc * 0 + last["n_records"] * (36 + 32),   # fixed part of each record read, 32-B descriptor written
               "decode_accumulate": unc + cnt}               # record bytes read once + 28 B/position/sample written once
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(alg[dom] / (kern[dom] * 1e-3) / 1e9, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                "algorithmic_bytes_per_launch": int(alg[dom]), "kernel_ms": round(kern[dom], 4)}
        roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 5)
        if args.length == CHR1_LEN:
            roof.update(pmc_traffic(dom))
        per_kernel = {k: {"ms": round(kern[k], 4), "algorithmic_GBps": round(alg[k] / (kern[k] * 1e-3) / 1e9, 2),
                          "frac_of_hbm_peak": round(alg[k] / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)} for k in kern}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(path, min(args.cpu_sample_reads, int(last["n_records"])))
        line = {
            "metric": "depth_base_Mreads_per_s", "value": round(reads_per_s / 1e6, 3), "unit": "Mreads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32", "data": "synthetic",
            "config": {"workload": "depth base on synthetic chr1 %dx coordinate-sorted BAM (L=%d, %d reads of %d bp per GPU, "
                                   "BGZF %s level %d, ratio %.2f), compressed bytes resident in HBM" % (
                                       int(args.coverage), args.length, int(last["n_records"]), read_len, args.codec,
                                       args.level, unc / max(1, comp)),
                       "baseline_config": "configs[1]" if args.length == CHR1_LEN else "configs[1] scaled down (development)",
                       "sharding": "every GPU runs the full contig-sized workload (same seeded BAM), no data-path collective"},
            "gbases_per_s": round(total_admitted * read_len / (elapsed / args.steps) / 1e9, 3),
            "reads_total": int(total_reads), "reads_admitted": int(total_admitted),
            "roofline": roof, "kernels": per_kernel, "cpu_baseline": cpu,
            "host": {"nproc": os.cpu_count(), "bam_gen_seconds": round(info.get("gen_seconds", 0.0), 1),
                     "h2d_ms": round(last.get("ms_h2d", 0.0), 1)},
        }
        print(json.dumps(line))
    d.close()
    if dist:
        dist.barrier()
        dist.destroy_process_group()


PMC_KERNELS = {"huffman_decode": ["k_huffman_decode"], "lz77_resolve": ["k_lz77_resolve"],
               "record_index": ["k_block_walk", "k_chain_check", "k_chain_repair", "k_count_scan", "k_describe", "k_tile_compact"],
               "decode_accumulate": ["k_accumulate"]}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same workload
    (tools/profile_round.sh: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs, KiB).
    FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE is taken as is."""
    path = os.path.join(ROOT, "profiles", "round1", "pmc_fetch_write_chr1_30x.csv")
    if not os.path.exists(path):
        return {}
    fetch = write = 0.0
    best = {}
    with open(path) as fh:
        next(fh)
        for ln in fh:
            k, c, v, _ = ln.strip().split(",")
            if k in PMC_KERNELS[kernel]:
                best[(k, c)] = max(best.get((k, c), 0.0), float(v))     # the full-size launch
    for (k, c), v in best.items():
        if c == "FETCH_SIZE":
            fetch += v * 1024
        else:
            write += v * 1024
    if not fetch and not write:
        return {}
    return {"traffic": int(2 * fetch + write), "traffic_unit": "bytes per launch",
            "traffic_source": "profiles/round1/pmc_fetch_write_chr1_30x.csv (rocprofv3 PMC, separate passes; "
                              "FETCH_SIZE raw %.2f GB doubled, WRITE_SIZE %.2f GB)" % (fetch / 1e9, write / 1e9)}


if __name__ == "__main__":
    main()
