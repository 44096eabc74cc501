#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the small files kept under profiles/<round>/:
kernel_launch_durations_ms.json (from the kernel trace), pmc_fetch_write.csv, pmc_sq.csv."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def find(pattern):
    r = glob.glob(os.path.join(out, pattern), recursive=True)
    return r[0] if r else None


kt = find("kt/**/*_kernel_trace.csv")
if kt:
    d = defaultdict(list)
    with open(kt) as fh:
        for row in csv.DictReader(fh):
            d[short(row["Kernel_Name"])].append(round((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6, 4))
    with open(os.path.join(out, "kernel_launch_durations_ms.json"), "w") as fh:
        json.dump(d, fh, indent=1)
    for k, v in d.items():
        if max(v) > 0.5:
            print("trace", k, "launches", len(v), "ms", v[:8])


def counters(sub):
    f = find(sub + "/**/*_counter_collection.csv")
    rows = []
    if not f:
        return rows
    seen = defaultdict(int)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            key = (k, row["Counter_Name"])
            seen[key] += 1
            rows.append((k, row["Counter_Name"], float(row["Counter_Value"]), seen[key]))
    return rows


fw = counters("fetch") + counters("write")
if fw:
    with open(os.path.join(out, "pmc_fetch_write.csv"), "w") as fh:
        fh.write("kernel,counter,value_KB,launch\n")
        for k, c, v, n in fw:
            if v >= 1024:
                fh.write("%s,%s,%d,%d\n" % (k, c, v, n))
                print("pmc", k, c, "%.2f GB" % (v * 1024 / 1e9), "launch", n)
sq = counters("sq1") + counters("sq2")
if sq:
    agg = defaultdict(float)
    for k, c, v, n in sq:
        agg[(k, c)] = max(agg[(k, c)], v)     # the full-size launch of each kernel
    with open(os.path.join(out, "pmc_sq.csv"), "w") as fh:
        fh.write("kernel,counter,value\n")
        for (k, c), v in sorted(agg.items()):
            if k.startswith(("k_huffman_decode", "k_lz77_resolve", "k_accumulate", "k_describe_blocks", "k_walk_blocks", "k_check_scan")):
                fh.write("%s,%s,%d\n" % (k, c, v))
                print("sq", k, c, int(v))
