"""K1a `huffman_decode2` on the CPU: the lane program of sambamba_amd/csrc/inflate2_core.hpp is `__host__ __device__`; the
harness tests/cpp/inflate2_host.cpp runs it one lane at a time over every BGZF block of a file, applies the literal translation and
the LZ77 resolve in plain C++ and compares with zlib's inflate -- the library the reference calls (block.d:158-185).  The same
statements run on the GPU (tests/test_gpu_inflate.py checks them there through the C ABI)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from tests.util import GOLDEN, ROOT, gen_bam

SRC = os.path.join(ROOT, "tests", "cpp", "inflate2_host.cpp")
CORE = os.path.join(ROOT, "sambamba_amd", "csrc", "inflate2_core.hpp")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("inflate2") / "inflate2_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-o", exe, SRC, "-lz"])
    return exe


def _run(exe, path, lane=0, exact=False):
    p = subprocess.run([exe, path, str(lane)] + (["--exact"] if exact else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    assert p.returncode == 0, p.stderr
    blocks, fast, general, bad = (int(x) for x in p.stdout.split())
    assert bad == 0
    return blocks, fast, general


def _bgzf_block(payload, **kw):
    level = kw.get("level", 6)
    co = zlib.compressobj(level, zlib.DEFLATED, -15, kw.get("mem", 8), kw.get("strategy", zlib.Z_DEFAULT_STRATEGY))
    c = b""
    fe = kw.get("flush_every")
    if fe:
        for i in range(0, len(payload), fe):
            c += co.compress(payload[i:i + fe]) + co.flush(zlib.Z_FULL_FLUSH)
    else:
        c += co.compress(payload)
    c += co.flush()
    assert len(c) + 26 <= 65536
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(c) + 25)
    return hdr + c + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


@pytest.mark.parametrize("name", ["issue225.bam", "issue_193.bam", "issue_204.bam", "mate_overlaps_1_3M_4M.bam"])
def test_reference_fixtures_every_block(harness, name):
    blocks, fast, general = _run(harness, os.path.join(GOLDEN, name), lane=5)
    assert blocks >= 2 and general == 0          # dynamic blocks and the fixed-code EOF block: all the fast kernel's


def test_synthetic_streams(harness, tmp_path):
    rng = np.random.default_rng(4321)
    text = (b"ACGTTTGACCA" * 4000)[:40000]
    rand = rng.integers(0, 256, 50000, dtype=np.uint8).tobytes()
    quals = rng.choice(np.array([2, 12, 23, 37], dtype=np.uint8), 65280, p=[.02, .05, .13, .80]).tobytes()
    nib = rng.integers(0, 16, 40000, dtype=np.uint8).tobytes()      # incompressible by matches, compressible by the code: dynamic blocks
    fast_cases = [
        (b"", {}), (b"A", {}), (text[:2000], dict(strategy=zlib.Z_FIXED)), (text, dict(level=9)), (quals, {}),
        (b"\x00" * 65280, {}), (b"ab" * 30000, dict(strategy=zlib.Z_RLE)), (text, dict(strategy=zlib.Z_HUFFMAN_ONLY)),
        (nib[:32768] + nib[:32768 - 7], dict(level=9)),     # distances up to 32768
        (rng.choice(np.frombuffer(b"ACGT", np.uint8), 65280).tobytes(), {}),
        (bytes(rng.integers(0, 256, 700, dtype=np.uint8)) * 90, dict(level=1)),
        (quals[:20000], dict(mem=6)),                 # several deflate blocks in one BGZF block (<= kMaxSeg)
    ]
    path = str(tmp_path / "fast.bgzf")
    with open(path, "wb") as fh:
        for payload, kw in fast_cases:
            fh.write(_bgzf_block(payload, **kw))
    for lane in (0, 17, 63):
        blocks, fast, general = _run(harness, path, lane)
        assert blocks == len(fast_cases) and general == 0
    # what the fast kernel hands to the general one: stored blocks (level 0, full-flush markers), too many deflate blocks
    general_cases = [(rand[:30000], dict(level=0)), (rand[:40000], {}), (text, dict(flush_every=3000)), (quals, dict(mem=1))]
    path = str(tmp_path / "general.bgzf")
    with open(path, "wb") as fh:
        for payload, kw in general_cases:
            fh.write(_bgzf_block(payload, **kw))
    blocks, fast, general = _run(harness, path, 9)
    assert blocks == len(general_cases) and general == len(general_cases)


def test_bench_like_bam_every_block(harness, tmp_path):
    path = gen_bam(str(tmp_path / "b.bam"), "chrB:1500000", coverage=30, seed=5)
    blocks, fast, general = _run(harness, path, lane=42)
    assert blocks > 600 and general == 0


@pytest.mark.parametrize("level", [1, 6, 9])
def test_streams_of_the_device_encoder_are_the_fast_kernels(harness, tmp_path, level):
    """What sbx_write_bam / sbx_bgzf_compress write (deflate_core.hpp: fixed code at level 1, dynamic codes from level 4, complete
    codes with at least two symbols each) is read back by the fast kernel alone -- a BAM written by the device encoder costs the read
    path no detour through the general kernel."""
    import gzip
    enc = str(tmp_path / "deflate_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", enc, os.path.join(ROOT, "tests", "native", "deflate_host.cpp")])
    bam = gen_bam(str(tmp_path / "b.bam"), "chrB:120000", coverage=30, seed=9)
    raw = gzip.decompress(open(bam, "rb").read())
    src, dst = str(tmp_path / "raw"), str(tmp_path / "enc.bgzf")
    for data in (raw, b"ab" * 5 + bytes(70000), bytes(np.random.default_rng(1).integers(0, 16, 100000, dtype=np.uint8))):
        open(src, "wb").write(data)
        subprocess.check_call([enc, src, dst, str(level)])
        blocks, fast, general = _run(harness, dst, lane=11)
        assert blocks == (len(data) + 0xFF00 - 1) // 0xFF00 and general == 0


def test_exact_readiness_rule_of_k1b(harness, tmp_path):
    """K1b (inflate.hip kExact, the default kernel since round 5) lets a near match start once no unfinished match writes into its source
    range: the candidates are a contiguous range of lanes found by two branch-free binary searches, tested against the pending lanes.
    The harness runs that rule step by step -- batches, window base, phase A, dependency masks, rounds in lockstep -- on the token
    streams K1a's lane program produces and compares with zlib: the reference's fixtures, a bench-like BAM, streams with runs and
    short periods (self-overlapping matches), other compressors' streams, far and near matches mixed."""
    for name in ("issue225.bam", "issue_193.bam", "issue_204.bam", "mate_overlaps_1_3M_4M.bam"):
        blocks, fast, general = _run(harness, os.path.join(GOLDEN, name), lane=9, exact=True)
        assert general == 0
    bam = gen_bam(str(tmp_path / "b.bam"), "chrB:600000", coverage=30, seed=6)
    blocks, fast, general = _run(harness, bam, lane=2, exact=True)
    assert blocks > 100 and general == 0
    rng = np.random.default_rng(77)
    path = str(tmp_path / "s.bgzf")
    with open(path, "wb") as fh:
        fh.write(_bgzf_block(b"A" * 60000))                                                   # one run: every byte points one back
        fh.write(_bgzf_block((b"ACGTTTGACCA" * 6000)[:65000]))                                # a short period
        fh.write(_bgzf_block(bytes(rng.integers(0, 4, 65000, dtype=np.uint8)), level=9))      # dense short matches
        fh.write(_bgzf_block(bytes(rng.integers(0, 4, 65000, dtype=np.uint8)), level=1))
        fh.write(_bgzf_block(b"ab" * 20 + bytes(rng.integers(0, 256, 3000, dtype=np.uint8)) * 20, level=6))     # 3000-byte period: far and near mixed
        fh.write(_bgzf_block(b""))
    blocks, fast, general = _run(harness, path, lane=0, exact=True)
    assert blocks == 6 and general == 0


def test_lanes_use_disjoint_lds():
    """The lane-interleaved layout: no two lanes' bytes overlap, every lane's area lies inside the wavefront's."""
    import re
    src = open(CORE).read()
    consts = {}
    for name, expr in re.findall(r"constexpr int (k\w+) = ([^;]+);", src):
        try:
            consts[name] = eval(expr, {}, consts)
        except Exception:
            pass
    wave = consts["kWaveLds"]
    owner = np.full(wave, -1, np.int32)
    for lane in range(64):
        spans = []
        for j in range(consts["kRingDw"]):
            spans.append((consts["kOffRing"] + 256 * j + 4 * lane, 4))
        for off, n in (("kOffLitStage", 4), ("kOffEntStage", 8)):
            for j in range(n):
                spans.append((consts[off] + 256 * j + 4 * lane, 4))
        for e in range(16):
            spans.append((consts["kOffAux"] + 128 * e + 2 * lane, 2))
        for off in ("kOffLenSym", "kOffDistSym"):
            for e in range(consts["kSymEntries"]):
                spans.append((consts[off] + 64 * e + lane, 1))
        for a, n in spans:
            assert a + n <= wave
            assert (owner[a:a + n] == -1).all()
            owner[a:a + n] = lane
    assert (owner >= 0).all()        # and nothing is wasted
