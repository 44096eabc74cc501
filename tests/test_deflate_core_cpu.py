"""The BGZF block encoder the device runs one lane per block (sambamba_amd/csrc/deflate_core.hpp: stored / fixed Huffman code /
dynamic Huffman code by level, hash-table LZ77, CRC32, stored fallback), compiled for the host with g++ and checked against
zlib: every stream must inflate to its input with correct CRC32 / ISIZE trailers (gzip.decompress verifies both) -- no GPU
needed."""
import gzip
import heapq
import os
import random
import struct
import subprocess

import pytest

from tests.util import ROOT

SRC = os.path.join(ROOT, "tests", "native", "deflate_host.cpp")


@pytest.fixture(scope="module")
def host_encoder(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("defl") / "deflate_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, SRC])
    return exe


def bam_like(n, seed):
    rng = random.Random(seed)
    out = bytearray()
    pos = 1000
    while len(out) < n:
        pos += rng.randrange(0, 12)
        name = b"r%010d\0" % rng.randrange(10 ** 9)
        seq = bytes(rng.choice((0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88)) for _ in range(75))
        qual = bytes(rng.choices((2, 12, 23, 37), (2, 5, 13, 80), k=150))
        body = struct.pack("<iiBBHHHiiii", 0, pos, len(name), 60, 4681, 1, 99, 150, 0, pos + 200, 350) + name + struct.pack("<I", 150 << 4) + seq + qual + b"RGZS1\0"
        out += struct.pack("<i", len(body)) + body
    return bytes(out[:n])


CASES = {
    "empty": b"",
    "one_byte": b"A",
    "three_bytes": b"abc",
    "run": b"F" * 70000,
    "period3": b"abc" * 30000,
    "text": (b"the quick brown fox jumps over the lazy dog. " * 3000),
    "bam_like": None,
    "random": None,
    "exact_block": None,
    "block_plus_one": None,
    "all_bytes": bytes(range(256)) * 600,
    "long_matches": (bytes(range(200)) * 2 + b"x") * 400,
    "two_symbols": b"ab" * 5,                               # shorter than a match: literals only, no distance code in use
    "nibbles": None,                                        # 16 equally likely literals, hardly a match: the dynamic code's home ground
    "skewed": None,                                         # literal counts like Fibonacci numbers: code lengths hit the 15-bit limit
    "zeros_then_noise": None,
}
LEVELS = [0, 1, 3, 4, 6, 7, 9, -1]


@pytest.mark.parametrize("name", sorted(CASES))
def skewed(seed):
    rng = random.Random(seed)
    fib = [1, 1]
    while len(fib) < 22:
        fib.append(fib[-1] + fib[-2])
    pool = b"".join(bytes([40 + k]) * f for k, f in enumerate(fib))     # 46367 bytes, 22 symbols, the rarest once
    return bytes(rng.sample(pool, len(pool)))


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("level", LEVELS)
def test_host_encoder_round_trips_through_zlib(host_encoder, tmp_path, name, level):
    data = CASES[name]
    if name == "nibbles":
        data = bytes(random.Random(7).choices(range(16), k=150000))
    elif name == "skewed":
        data = skewed(8)
    elif name == "zeros_then_noise":
        data = bytes(30000) + random.Random(9).randbytes(30000) + bytes(30000)
    if name == "bam_like":
        data = bam_like(300000, 3)
    elif name == "random":
        data = random.Random(1).randbytes(200000)
    elif name == "exact_block":
        data = bam_like(0xFF00, 4)
    elif name == "block_plus_one":
        data = bam_like(0xFF00 + 1, 5)
    src, dst = str(tmp_path / "in"), str(tmp_path / "out")
    open(src, "wb").write(data)
    subprocess.check_call([host_encoder, src, dst, str(level)])
    comp = open(dst, "rb").read()
    n_blocks = (len(data) + 0xFF00 - 1) // 0xFF00
    # block structure: BGZF magic, BC subfield, BSIZE chain
    off, seen = 0, 0
    while off < len(comp):
        assert comp[off:off + 4] == b"\x1f\x8b\x08\x04" and comp[off + 12:off + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", comp, off + 16)[0] + 1
        assert bsize <= 65536
        off += bsize
        seen += 1
    assert off == len(comp) and seen == n_blocks
    assert (gzip.decompress(comp) if comp else b"") == data
    if level and name in ("run", "period3", "text", "bam_like", "long_matches"):
        assert len(comp) < len(data) * (0.75 if name == "bam_like" else 0.2), (len(comp), len(data))
    if name == "nibbles" and (level >= 4 or level == -1):
        assert len(comp) < len(data) * 0.52                 # 4 bits of entropy per byte; the fixed code spends 8 on a literal
    if name == "random" or level == 0:
        assert len(comp) <= len(data) + n_blocks * 31       # stored fallback: 5 + 26 bytes per block


def test_levels_trade_work_for_bytes(host_encoder, tmp_path):
    """bgzfCompress passes `level` to zlib (compress.d:34-103): here 1..3 = fixed code, 4..6 and -1 = dynamic code, 7..9 = dynamic
    code over four candidates per position with lazy evaluation; each step must pay on a BAM-like stream."""
    data = bam_like(400000, 11)
    src = str(tmp_path / "in")
    open(src, "wb").write(data)
    size = {}
    for level in (0, 1, 3, 4, 6, -1, 7, 9):
        dst = str(tmp_path / ("out%d" % level))
        subprocess.check_call([host_encoder, src, dst, str(level)])
        comp = open(dst, "rb").read()
        assert gzip.decompress(comp) == data
        size[level] = len(comp)
    assert size[1] == size[3] and size[4] == size[6] == size[-1] and size[7] == size[9]
    assert size[0] > size[1] > size[4] > size[7]
    assert size[4] < 0.85 * size[1]


def _optimal_cost(freqs):
    heap = [f for f in freqs if f]
    heapq.heapify(heap)
    cost = 0
    while len(heap) > 1:
        a, b = heapq.heappop(heap), heapq.heappop(heap)
        cost += a + b
        heapq.heappush(heap, a + b)
    return cost


def _lengths(host_encoder, max_len, freqs):
    out = subprocess.check_output([host_encoder, "--lengths", str(max_len)] + [str(f) for f in freqs]).decode().split()
    return [tuple(int(x) for x in t.split(":")) for t in out]


def test_code_lengths_are_minimum_redundancy_and_limited(host_encoder):
    """huffman_lengths: without the limit in play the cost equals Huffman's; with it the code stays complete (Kraft sum 1), within
    the limit, and prefix-free as canonical codes."""
    rng = random.Random(5)
    cases = [[1, 1], [5, 1, 1, 2, 3, 0, 8], [1] * 19, [1] * 286, [65280, 1], [0, 0, 7, 0, 0, 9]]
    fib = [1, 1]
    while len(fib) < 30:
        fib.append(fib[-1] + fib[-2])
    cases.append(fib[:22])                                                 # depth 21 without a limit
    for _ in range(40):
        n = rng.choice((2, 3, 19, 30, 286))
        cases.append([rng.choice((0, 0, 1, 2, 5, 40, 300, 5000)) for _ in range(n)])
    for freqs in cases:
        if sum(1 for f in freqs if f) < 2:
            continue
        for max_len in (15, 7):
            used = sum(1 for f in freqs if f)
            if used > (1 << max_len):
                continue
            res = _lengths(host_encoder, max_len, freqs)
            lens = [l for l, _ in res]
            assert all((l > 0) == (f > 0) for l, f in zip(lens, freqs))
            assert max(lens) <= max_len
            assert sum(2.0 ** -l for l in lens if l) == 1.0
            cost = sum(l * f for l, f in zip(lens, freqs))
            opt = _optimal_cost(freqs)
            assert cost >= opt
            unlimited = _lengths(host_encoder, 15, freqs) if max_len != 15 else res
            if max(l for l, _ in unlimited) < 15 and max_len == 15:
                assert cost == opt, (freqs, lens)
            # canonical codes (stored bit-reversed): no code is a prefix of another
            words = sorted(format(c, "0%db" % l)[::-1] for l, c in res if l)
            assert all(not b.startswith(a) for a, b in zip(words, words[1:]))


def test_random_mixtures_round_trip_at_every_level(host_encoder, tmp_path):
    rng = random.Random(2026)
    src, dst = str(tmp_path / "in"), str(tmp_path / "out")
    for trial in range(24):
        parts = []
        for _ in range(rng.randrange(1, 12)):
            kind = rng.randrange(5)
            if kind == 0:
                parts.append(rng.randbytes(rng.randrange(1, 30000)))
            elif kind == 1:
                parts.append(bytes([rng.randrange(256)]) * rng.randrange(1, 40000))
            elif kind == 2:
                unit = rng.randbytes(rng.randrange(1, 600))
                parts.append(unit * rng.randrange(1, 80))
            elif kind == 3:
                parts.append(bytes(rng.choices(range(rng.randrange(1, 40)), k=rng.randrange(1, 30000))))
            else:
                parts.append(bam_like(rng.randrange(1, 60000), trial))
        data = b"".join(parts)
        open(src, "wb").write(data)
        for level in (1, 6, 9):
            subprocess.check_call([host_encoder, src, dst, str(level)])
            assert gzip.decompress(open(dst, "rb").read()) == data, (trial, level)


def test_known_answers(host_encoder, tmp_path):
    """The encoder's bytes are part of its contract with the device (tests/test_gpu_writer.py compares the device's stream with this
    build's, byte for byte): a change of the matcher or of the code construction shows up here, on the CPU, as a changed digest --
    to be updated knowingly, together with a device run."""
    import hashlib
    data = bam_like(300000, 3)
    assert hashlib.md5(data).hexdigest() == "66160ee26db1622e0b86d2730c6efb1e"          # (the input itself is seeded python code)
    want = {0: (300155, "4bf1d93369de6a09fa6b55f2be383277"), 1: (155479, "57beebc3875118902a333e7e828e2d1c"),
            6: (110865, "091827342aea044d075a1dbd026ab259"), 9: (107102, "098d10085f80a36b9d64706986a60e06")}
    src, dst = str(tmp_path / "in"), str(tmp_path / "out")
    open(src, "wb").write(data)
    for level, (size, digest) in want.items():
        subprocess.check_call([host_encoder, src, dst, str(level)])
        comp = open(dst, "rb").read()
        assert (len(comp), hashlib.md5(comp).hexdigest()) == (size, digest), level
