"""Build libsbx_depth.so (HIP kernels + C ABI) and the sbx-depth CLI for gfx950 with hipcc.

Usage: python -m sambamba_amd.build   (or sambamba_amd.build.build())
Outputs are written in-tree (sambamba_amd/csrc/) so they travel with gpurun snapshots.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsbx_depth.so")
CLI = os.path.join(CSRC, "sbx-depth")
SOURCES = ["inflate.hip", "index.hip", "depth.hip", "reduce.hip", "mates.hip", "format.hip", "deflate.hip", "engine.cpp"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join("..", "..", "include", "sbx_depth.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-Wno-unused-result"]
    if force or _stale(LIB, deps):
        objs, jobs = [], []
        for s in srcs:
            o = s.rsplit(".", 1)[0] + ".o"
            if force or _stale(o, deps):
                cmd = [_hipcc()] + flags + ["-x", "hip", "-c", s, "-o", o]
                if verbose:
                    print(" ".join(cmd))
                jobs.append(cmd)
            objs.append(o)
        # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
            for rc in ex.map(subprocess.call, jobs):
                if rc != 0:
                    raise subprocess.CalledProcessError(rc, "hipcc")
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    cli_src = os.path.join(CSRC, "cli.cpp")
    if os.path.exists(cli_src) and (force or _stale(CLI, [cli_src, LIB] + deps)):
        cmd = [_hipcc(), "-O2", "-std=c++17", "-o", CLI, cli_src, "-L" + CSRC, "-lsbx_depth",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
