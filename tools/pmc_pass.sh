#!/bin/bash
# Counter passes of one bench workload on a GPU box (run from the repo root), written as the stamped files bench.py joins:
#   profiles-style CSVs under $OUT: pmc_fetch_write_config<N>.csv (FETCH_SIZE, WRITE_SIZE: separate runs, KiB) and
#   pmc_sq_config<N>.csv (SQ instruction counters), each with `# sources <hash>` = bench.kernel_sources_hash() in its first line,
#   plus kernel_stats_config<N>.csv from `rocprofv3 --kernel-trace --stats` of the same command.
# Counters are never combined with traces.  usage: tools/pmc_pass.sh <outdir> <config> [extra bench.py arguments]
set -u
OUT=${1:-gpurun_out/pmc}
CFG=${2:-2}
EXTRA=${3:-}
REPO=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
RAW=/tmp/pmc_raw_$CFG
mkdir -p $RAW
export TMPDIR=/tmp
STAMP=$(python -c "import bench; print(bench.kernel_sources_hash())")
cd /tmp
B="python $REPO/bench.py --config $CFG $EXTRA --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --parity-windows 0 --no-full-parity --no-side-runs"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $RAW/fetch -o p -- $B > /dev/null 2> $RAW/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $RAW/write -o p -- $B > /dev/null 2> $RAW/write.err
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $RAW/sq1 -o p -- $B > /dev/null 2> $RAW/sq1.err
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $RAW/sq2 -o p -- $B > /dev/null 2> $RAW/sq2.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/kt -o kt -- python $REPO/bench.py --config $CFG $EXTRA --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --parity-windows 0 --no-full-parity --no-side-runs > $OUT/bench_config${CFG}_under_rocprofv3.json 2> $RAW/kt.err
cd $REPO
python - "$RAW" "$OUT" "$CFG" "$STAMP" <<'PY'
import csv, glob, collections, re, sys, subprocess
raw, out, cfg, stamp = sys.argv[1:5]
def collect(dirs):
    acc = collections.defaultdict(float)
    for d in dirs:
        for f in glob.glob("%s/%s/**/*counter_collection.csv" % (raw, d), recursive=True):
            for row in csv.DictReader(open(f)):
                m = re.search(r"(k_[a-z0-9_]+)", row["Kernel_Name"])
                acc[(m.group(1) if m else row["Kernel_Name"][:30], row["Counter_Name"])] += float(row["Counter_Value"])
    return acc
for name, dirs, floor in (("pmc_fetch_write_config%s.csv" % cfg, ("fetch", "write"), 1024), ("pmc_sq_config%s.csv" % cfg, ("sq1", "sq2"), 1)):
    acc = collect(dirs)
    with open("%s/%s" % (out, name), "w") as fh:
        fh.write("# sources %s (bench.kernel_sources_hash(): sha1 of the comment-stripped device sources these counters were measured on)\n" % stamp)
        fh.write("kernel,counter,value,launches_summed\n")
        for (k, c), v in sorted(acc.items()):
            if v >= floor:
                fh.write("%s,%s,%d,all\n" % (k, c, v))
    print(name, len(acc), "rows")
for f in glob.glob("%s/kt/**/*kernel_stats.csv" % raw, recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("%s/kernel_stats_config%s.csv" % (out, cfg), "w") as fh:
        fh.write("# sources %s\n" % stamp)
        w = csv.DictWriter(fh, fieldnames=rows[0].keys()); w.writeheader()
        for r in rows[:24]: w.writerow(r)
    for r in rows[:12]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
