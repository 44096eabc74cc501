"""K1 parity: device inflate (sbx_inflate_blocks, the codec seam) vs zlib through the oracle.
Bit-exact on every BGZF block of the reference's fixtures and on synthetic streams covering the
deflate block types the fixtures lack (stored blocks, many blocks per BGZF block)."""
import os
import struct
import zlib

import numpy as np
import pytest

from tests.util import GOLDEN, oracle_inflate_all, scan_bgzf

pytestmark = pytest.mark.gpu

FIXTURES = ["issue225.bam", "issue_193.bam", "issue_204.bam", "mate_overlaps_1_3M_4M.bam"]


@pytest.mark.parametrize("name", FIXTURES)
def test_inflate_fixture_blocks_bit_exact(name):
    import sambamba_amd
    path = os.path.join(GOLDEN, name)
    data, co, cl, isz, oo, total = scan_bgzf(path)
    got = sambamba_amd.inflate_blocks(data, co, cl, isz, oo, total)
    want = oracle_inflate_all(path)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def _raw_deflate(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8, flush_every=None):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    out = b""
    if flush_every:
        for i in range(0, len(payload), flush_every):
            out += co.compress(payload[i:i + flush_every])
            out += co.flush(zlib.Z_FULL_FLUSH)   # emits an empty stored block: exercises BTYPE 0 mid-stream
    else:
        out += co.compress(payload)
    out += co.flush()
    return out


def _records(rng, n, size):
    """n records of `size` bytes, each a copy of the previous one with a few bytes changed (what a BAM stream looks like)."""
    rec = bytearray(rng.integers(0, 256, size, dtype=np.uint8).tobytes())
    out = bytearray()
    for _ in range(n):
        for _k in range(int(rng.integers(1, 6))):
            rec[int(rng.integers(0, size))] = int(rng.integers(0, 256))
        out += rec
    return bytes(out[:65280])


def _cases():
    rng = np.random.default_rng(1234)
    text = (b"ACGTTTGACCA" * 4000)[:40000]
    rand = rng.integers(0, 256, 50000, dtype=np.uint8).tobytes()
    quals = rng.choice(np.array([2, 12, 23, 37], dtype=np.uint8), 65280, p=[.02, .05, .13, .80]).tobytes()
    cases = [
        ("empty", b"", dict(level=6)),
        ("one_byte", b"A", dict(level=6)),
        ("stored_level0", rand[:30000], dict(level=0)),
        ("fixed_huffman", text[:2000], dict(level=6, strategy=zlib.Z_FIXED)),
        ("dynamic_text", text, dict(level=9)),
        ("dynamic_random", rand, dict(level=6)),
        ("quals_full_block", quals, dict(level=6)),
        ("rle_runs", b"\x00" * 65280, dict(level=6)),
        ("rle_strategy", b"ab" * 30000, dict(level=6, strategy=zlib.Z_RLE)),
        ("huffman_only", text, dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY)),
        ("many_blocks_memlevel1", quals, dict(level=6, mem=1)),
        ("full_flush_stored_markers", text, dict(level=6, flush_every=3000)),
        ("max_distance", rand[:32768] + rand[:32768 - 7], dict(level=9)),
        # match chains inside one batch of the resolver: every "record" copies most of its bytes from the previous one, which
        # copied them from the one before; short matches over a four-letter alphabet reference recent output densely
        ("record_chain", _records(rng, 230, 283), dict(level=6)),
        ("record_chain_short", _records(rng, 1200, 41), dict(level=9)),
        ("acgt_random", rng.choice(np.frombuffer(b"ACGT", np.uint8), 65280).tobytes(), dict(level=6)),
        ("packed_bases", rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88],
                                             dtype=np.uint8), 65280).tobytes(), dict(level=6)),
    ]
    return cases


def test_inflate_synthetic_deflate_streams_bit_exact():
    import sambamba_amd
    comp, co, cl, isz, oo = b"", [], [], [], []
    want = b""
    for name, payload, kw in _cases():
        c = _raw_deflate(payload, **kw)
        assert zlib.decompress(c, -15) == payload
        pad = (-len(comp)) % 1          # payloads start at arbitrary byte alignment
        comp += b"\0" * pad
        co.append(len(comp) + 0)
        cl.append(len(c))
        isz.append(len(payload))
        oo.append(len(want))
        comp += c + b"\xAA\xBB\xCC"      # 3 junk bytes so the next payload is misaligned
        want += payload
    got = sambamba_amd.inflate_blocks(np.frombuffer(comp, np.uint8), co, cl, isz, oo, len(want))
    assert got.tobytes() == want


def test_inflate_rejects_corrupt_stream():
    import sambamba_amd
    payload = b"hello hello hello hello" * 100
    c = bytearray(_raw_deflate(payload))
    c[len(c) // 2] ^= 0x5A
    try:
        ok = zlib.decompress(bytes(c), -15) == payload
    except zlib.error:
        ok = False
    if ok:
        pytest.skip("bit flip did not corrupt the stream")
    with pytest.raises(sambamba_amd.SbxError) as ei:
        sambamba_amd.inflate_blocks(np.frombuffer(bytes(c), np.uint8), [0], [len(c)], [len(payload)], [0], len(payload))
    assert ei.value.code == -3   # SBX_EFORMAT


def test_inflate_rejects_wrong_isize():
    import sambamba_amd
    payload = b"abcdefgh" * 64
    c = _raw_deflate(payload)
    with pytest.raises(sambamba_amd.SbxError):
        sambamba_amd.inflate_blocks(np.frombuffer(c, np.uint8), [0], [len(c)], [len(payload) - 1], [0], len(payload))


def test_inflate_bench_like_bam_bit_exact(tmp_path):
    """Every block of a BAM with the bench's statistics (tools/gen_bam: 41 % of the literal/length symbols are matches of 8
    bytes on average, record-to-record chains, short far matches): ~800 BGZF blocks against zlib."""
    import sambamba_amd
    from tests.util import gen_bam
    path = gen_bam(str(tmp_path / "b.bam"), "chrB:1200000", coverage=30, seed=77)
    data, co, cl, isz, oo, total = scan_bgzf(path)
    assert len(cl) > 500
    got = sambamba_amd.inflate_blocks(data, co, cl, isz, oo, total)
    want = oracle_inflate_all(path)
    assert got.shape == want.shape and np.array_equal(got, want)
