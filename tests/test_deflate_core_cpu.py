"""The BGZF block encoder the device runs one lane per block (sambamba_amd/csrc/deflate_core.hpp: fixed Huffman code, greedy
LZ77, CRC32, stored fallback), compiled for the host with g++ and checked against zlib: every stream must inflate to its
input with correct CRC32 / ISIZE trailers (gzip.decompress verifies both) -- no GPU needed."""
import gzip
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
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("level", [0, 6, -1])
def test_host_encoder_round_trips_through_zlib(host_encoder, tmp_path, name, level):
    data = CASES[name]
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
    if name == "random" or level == 0:
        assert len(comp) <= len(data) + n_blocks * 31       # stored fallback: 5 + 26 bytes per block
