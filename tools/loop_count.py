#!/usr/bin/env python3
"""Counts the instructions of the loop that holds the first occurrence of a marker instruction in a kernel of an AMDGPU .s file
(development tool).  usage: loop_count.py file.s kernel_name_substring [marker=v_dot2_u32_u16]"""
import collections
import re
import sys

path, kname = sys.argv[1], sys.argv[2]
args = [a for a in sys.argv[3:] if a != "-v"]
marker = args[0] if args else "v_dot2_u32_u16"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(kname), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
# blocks: (label line index, loop header name or None)
blocks = []
cur = None
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
    m2 = re.match(r"^; %bb\.\d+:(.*)$", l)
    if m or m2:
        cur = [i, (m.group(1) if m else None), (m.group(2) if m else m2.group(1)), []]
        blocks.append(cur)
    elif cur is not None:
        cur[3].append((i, l))
first = next(i for i, l in enumerate(body) if marker in l)
blk = max((b for b in blocks if b[0] <= first), key=lambda b: b[0])
def header_of(b, lines_after):
    txt = b[2] + " ".join(l for _, l in b[3][:3] if l.strip().startswith(";"))
    m = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", txt)
    if m: return m.group(1)
    m = re.search(r"Loop Header: Depth=(\d+)", txt)
    if m and b[1]: return b[1][2:]
    if "Parent Loop" in txt and b[1]: return b[1][2:]
    return None
hdr = header_of(blk, None)
cat = collections.Counter()
ops = collections.Counter()
n_blocks = 0
for b in blocks:
    if header_of(b, None) != hdr:
        continue
    n_blocks += 1
    for _, l in b[3]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        ops[op] += 1
        if op.startswith("v_"): cat["valu"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): cat["branch"] += 1
        elif op.startswith("s_nop"): cat["nop"] += 1
        elif op.startswith("s_waitcnt"): cat["wait"] += 1
        elif op.startswith("s_"): cat["salu"] += 1
        elif op.startswith("ds_"): cat["ds"] += 1
        else: cat["vmem"] += 1
print("loop", hdr, "blocks", n_blocks, dict(cat), "total", sum(cat.values()))
if "-v" in sys.argv:
    for k, v in ops.most_common(60): print(v, k)
