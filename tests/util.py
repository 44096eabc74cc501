"""Shared helpers for the test-suite: the oracle (checker) and the product CLI."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_BIN = os.path.join(ROOT, "oracle", "depth_oracle")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
GEN_BAM = os.path.join(ROOT, "tools", "gen_bam")


def ensure_oracle():
    if not (os.path.exists(ORACLE_BIN) and os.path.exists(ORACLE_LIB)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


def ensure_gen():
    if not os.path.exists(GEN_BAM):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])


def run_oracle(args, cwd=None):
    """Text output of the CPU oracle for `depth <args>`."""
    ensure_oracle()
    return subprocess.run([ORACLE_BIN] + list(args), cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                          check=True).stdout


def run_cli(args, cwd=None, check=True, env=None):
    """Text output of the product CLI (sbx-depth) for `depth <args>` -- runs on the GPU.  env: extra environment variables."""
    from sambamba_amd import cli_path
    r = subprocess.run([cli_path()] + list(args), cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, **env) if env else None)
    if check and r.returncode != 0:
        raise RuntimeError("sbx-depth failed (%d): %s" % (r.returncode, r.stderr.decode()))
    return r.stdout if check else r


_olib = None


def oracle_lib():
    global _olib
    if _olib is None:
        ensure_oracle()
        _olib = C.CDLL(ORACLE_LIB)
        _olib.orc_base_counters.argtypes = [C.c_char_p, C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                            C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
        _olib.orc_base_counters_indexed.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int,
                                                    C.c_char_p, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
        _olib.orc_inflate_all.argtypes = [C.c_char_p, C.c_void_p, C.c_ulonglong]
        _olib.orc_inflate_all.restype = C.c_longlong
    return _olib


def oracle_base_counters(bam, ref_id, beg, end, n_samples=1, min_bq=0, fix_mate=False, combined=False, flt=None, ref_name=None):
    """Counters of [beg, end) from the oracle's literal column pipeline; with `ref_name` the reads are fetched through the
    BAI (`-L name:beg+1-end`) instead of a pass over the whole file."""
    L = oracle_lib()
    out = np.zeros((end - beg, n_samples, 7), dtype=np.uint32)
    err = C.create_string_buffer(512)
    if ref_name is not None:
        rc = L.orc_base_counters_indexed(bam.encode(), ref_name.encode(), ref_id, beg, end, min_bq, int(fix_mate), int(combined),
                                         flt.encode() if flt else None, n_samples, out.ctypes.data, err, 512)
    else:
        rc = L.orc_base_counters(bam.encode(), ref_id, beg, end, min_bq, int(fix_mate), int(combined),
                                 flt.encode() if flt else None, n_samples, out.ctypes.data, err, 512)
    if rc != 0:
        raise RuntimeError("oracle failed: " + err.value.decode())
    return out


def oracle_inflate_all(path):
    L = oracle_lib()
    n = L.orc_inflate_all(path.encode(), None, 0)
    assert n >= 0
    buf = np.zeros(int(n), dtype=np.uint8)
    got = L.orc_inflate_all(path.encode(), buf.ctypes.data, n)
    assert got == n
    return buf


def scan_bgzf(path):
    """Host-side BGZF block table (python restatement for tests): returns numpy arrays
    comp_off, comp_len, isize, out_off and the file bytes."""
    data = np.fromfile(path, dtype=np.uint8)
    b = data.tobytes()
    off, uo = 0, 0
    co, cl, isz, oo = [], [], [], []
    while off + 18 <= len(b):
        assert b[off:off + 4] == b"\x1f\x8b\x08\x04"
        xlen = int.from_bytes(b[off + 10:off + 12], "little")
        p, bsize = off + 12, None
        while p < off + 12 + xlen:
            slen = int.from_bytes(b[p + 2:p + 4], "little")
            if b[p] == 66 and b[p + 1] == 67:
                bsize = int.from_bytes(b[p + 4:p + 6], "little")
            p += 4 + slen
        cdata = bsize - xlen - 19
        n = int.from_bytes(b[off + 12 + xlen + cdata + 4:off + 12 + xlen + cdata + 8], "little")
        co.append(off + 12 + xlen)
        cl.append(cdata)
        isz.append(n)
        oo.append(uo)
        uo += n
        off += bsize + 1
    return data, np.array(co, np.uint64), np.array(cl, np.uint32), np.array(isz, np.uint32), np.array(oo, np.uint64), uo


def gen_bam(out, contigs, coverage=30, seed=1, extra=()):
    ensure_gen()
    subprocess.check_call([GEN_BAM, "--out", out, "--contigs", contigs, "--coverage", str(coverage), "--seed", str(seed)]
                          + list(extra), stdout=subprocess.DEVNULL)
    return out
