"""The write side: device BGZF compression (sbx_bgzf_compress, sbx_write_bam) and the BAI builder
(sbx_build_index, `sambamba index`).  Any valid deflate stream is acceptable to a BGZF reader, so the compressor is checked by
inflating with zlib (gzip.decompress verifies the CRC32 and ISIZE of every block) and -- bit for bit -- against the same encoder
compiled for the host; the index is checked through what it is for: region fetches through it must give the results the
harness-written index gives, and its bookkeeping (metadata pseudo-bin, linear index, no-coordinate count) must follow
IndexBuilder (BioD/bio/std/hts/bam/bai/indexing.d)."""
import gzip
import os
import random
import shutil
import struct
import subprocess

import numpy as np
import pytest

import sambamba_amd
from tests import bamgen as bg
from tests.test_deflate_core_cpu import SRC as HOST_SRC, bam_like, skewed
from tests.util import ROOT, gen_bam, run_cli, run_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host_encoder(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("defl") / "deflate_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, HOST_SRC])
    return exe


WRITER_CASES = {
    "empty": lambda: b"",
    "tiny": lambda: b"x",
    "run": lambda: b"F" * 200000,
    "bam_like": lambda: bam_like(1_500_000, 11),
    "random": lambda: random.Random(2).randbytes(300_000),
    "exact_blocks": lambda: bam_like(2 * 0xFF00, 12),
    "one_over": lambda: bam_like(0xFF00 + 1, 13),
    "nibbles": lambda: bytes(random.Random(3).choices(range(16), k=200_000)),        # literals only: the dynamic code's home ground
    "skewed": lambda: skewed(4) * 3,                                                  # code lengths hit the 15-bit limit
}


@pytest.mark.parametrize("name", sorted(WRITER_CASES))
@pytest.mark.parametrize("level", [1, 6, 9, 0, -1])
def test_device_bgzf_stream_inflates_and_equals_the_host_encoder(host_encoder, tmp_path, name, level):
    data = WRITER_CASES[name]()
    comp = sambamba_amd.bgzf_compress(data, level=level, with_eof=True)
    assert comp.endswith(bytes(bg.EOF_BLOCK))
    assert gzip.decompress(comp) == data
    src, dst = str(tmp_path / "in"), str(tmp_path / "out")
    open(src, "wb").write(data)
    subprocess.check_call([host_encoder, src, dst, str(level)])
    assert comp[:-28] == open(dst, "rb").read()          # the same bytes on the device and on the host
    if level and name in ("run", "bam_like"):
        assert len(comp) < 0.7 * len(data)


def test_default_level_compresses_and_bad_levels_are_rejected():
    """-1 is zlib's Z_DEFAULT_COMPRESSION and the reference's default (bgzfCompress(chunk, level = -1)): it must compress."""
    data = bam_like(400000, 11)
    assert len(sambamba_amd.bgzf_compress(data, level=-1)) == len(sambamba_amd.bgzf_compress(data, level=6)) < 0.7 * len(data)
    for bad in (-2, 10, 100):
        with pytest.raises(sambamba_amd.SbxError) as ei:
            sambamba_amd.bgzf_compress(data, level=bad)
        assert ei.value.code == -1        # SBX_EINVAL


def test_levels_trade_work_for_bytes():
    """compress.d:34-103 hands `level` to zlib; here 1..3 = fixed code, 4..6 / -1 = dynamic code, 7..9 = dynamic code over four
    candidates per position with lazy evaluation (deflate_core.hpp)."""
    data = bam_like(2_000_000, 31)
    size = {lv: len(sambamba_amd.bgzf_compress(data, level=lv)) for lv in (0, 1, 3, 4, 6, 7, 9)}
    assert size[1] == size[3] and size[4] == size[6] and size[7] == size[9]
    assert size[0] > size[1] > size[4] > size[7]
    assert size[4] < 0.85 * size[1]


def test_many_blocks_in_several_pieces():
    """More blocks than one launch piece holds is out of reach of a unit test; several thousand blocks exercise the scan + pack."""
    data = bam_like(40_000_000, 21)
    comp = sambamba_amd.bgzf_compress(data, with_eof=False)
    assert gzip.decompress(comp) == data
    assert len(comp) < 0.65 * len(data)


@pytest.fixture(scope="module")
def small_bam(tmp_path_factory):
    d = tmp_path_factory.mktemp("wr")
    return gen_bam(str(d / "s.bam"), "c1:120000,c2:40000,cE:3000,c3:60000", coverage=15, seed=77, extra=["--samples", "2"])


def test_write_bam_and_index_round_trip(small_bam, tmp_path):
    """inflate a BAM, write it again through the device writer + indexer: the oracle (zlib + its own BAI reader) and the product read
    the same records from it -- whole-file and through the new index."""
    stream = gzip.decompress(open(small_bam, "rb").read())
    out = str(tmp_path / "rewritten.bam")
    sambamba_amd.write_bam(out, stream, with_index=True)
    assert gzip.decompress(open(out, "rb").read()) == stream
    assert os.path.exists(out + ".bai")
    for args in (["base"], ["base", "-L", "c1:20000-60000"], ["window", "-w", "500"], ["region", "-L", "c3:100-50000", "-T", "5"]):
        want = run_oracle(args + [small_bam])
        assert run_oracle(args + [out]) == want, args
        assert run_cli(args + [out]) == want, args


def parse_bai(path):
    b = open(path, "rb").read()
    assert b[:4] == b"BAI\1"
    n_ref = struct.unpack_from("<i", b, 4)[0]
    p = 8
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", b, p)[0]; p += 4
        bins = {}
        for _ in range(n_bin):
            bid, n_ch = struct.unpack_from("<Ii", b, p); p += 8
            bins[bid] = [struct.unpack_from("<QQ", b, p + 16 * k) for k in range(n_ch)]
            p += 16 * n_ch
        n_intv = struct.unpack_from("<i", b, p)[0]; p += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, b, p)); p += 8 * n_intv
        refs.append((bins, lin))
    tail = b[p:]
    return refs, tail


def test_index_bookkeeping_follows_the_reference_indexer(tmp_path):
    """A hand-made BAM with unmapped-but-placed reads, reads without coordinates, an empty contig and reads that cross 16 kbp
    windows: metadata pseudo-bin, linear index, chunk merging rule and the no-coordinate trailer as IndexBuilder writes them."""
    rng = random.Random(3)
    recs = []
    def seq(n):
        return "".join(rng.choice("ACGT") for _ in range(n))
    for pos in range(100, 70000, 37):
        recs.append(bg.make_record(0, pos, "100M", seq(100), 30, name="a%d" % pos))
    recs.append(bg.make_record(0, 70000, "", seq(50), 30, name="placed_unmapped", flag=4))
    for pos in range(5, 20000, 211):
        recs.append(bg.make_record(2, pos, "30M5000N30M", seq(60), 30, name="s%d" % pos))
    for k in range(7):
        recs.append(bg.make_record(-1, -1, "", seq(40), 30, name="nocoor%d" % k, flag=4))
    bam = str(tmp_path / "h.bam")
    info = bg.write_bam(bam, [("k1", 100000), ("kEmpty", 5000), ("k3", 50000)], recs, block_size=9000)
    shutil.copy(bam + ".bai", bam + ".harness.bai")
    sambamba_amd.build_index(bam)
    refs, tail = parse_bai(bam + ".bai")
    assert len(refs) == 3 and struct.unpack("<Q", tail)[0] == 7            # n_no_coor
    assert refs[1] == ({}, [])                                             # empty reference: n_bin = 0, n_intv = 0
    n_mapped0 = len(range(100, 70000, 37))
    meta0 = refs[0][0].pop(37450)
    assert len(meta0) == 2 and meta0[1] == (n_mapped0, 1)                  # (mapped, unmapped) of the reference
    meta2 = refs[2][0].pop(37450)
    assert meta2[1] == (len(range(5, 20000, 211)), 0)
    # every chunk is a proper virtual-offset range, chunks of a bin are in file order, no bin id beyond 37449
    for bins, lin in (refs[0], refs[2]):
        assert bins and max(bins) <= 37449
        for chunks in bins.values():
            for (a, b2), nxt in zip(chunks, chunks[1:] + [None]):
                assert a < b2 and (nxt is None or b2 <= nxt[0])
        assert all(x <= y for x, y in zip(lin, lin[1:])) and lin[-1] > 0
    # linear index of k1: window 0 is reached by the first read, the window of position 70000 by the placed unmapped read at the latest
    assert len(refs[0][1]) == 70000 // 16384 + 1
    # reads of k3 span 5060 positions: the last one (pos 19838) reaches window 1
    assert len(refs[2][1]) == 2
    # the index does what an index is for: fetches through it equal fetches through the harness index and the whole-file pass
    for args in (["base", "-L", "k1:16000-17000"], ["base", "-L", "k3:1-50000", "-c", "0"], ["region", "-L", "k1:60000-70100"]):
        want = run_cli(args + [bam])
        shutil.copy(bam + ".harness.bai", bam + ".tmp.bai")
        os.replace(bam + ".bai", bam + ".device.bai")
        os.replace(bam + ".tmp.bai", bam + ".bai")
        assert run_cli(args + [bam]) == want == run_oracle(args + [bam]), args
        os.replace(bam + ".device.bai", bam + ".bai")


def test_index_of_a_bench_like_bam_serves_random_regions(small_bam, tmp_path):
    bam = str(tmp_path / "copy.bam")
    shutil.copy(small_bam, bam)
    sambamba_amd.build_index(bam)
    rng = random.Random(9)
    for _ in range(6):
        ref = rng.choice(["c1", "c2", "c3"])
        L = {"c1": 120000, "c2": 40000, "c3": 60000}[ref]
        a = rng.randrange(1, L - 2000)
        reg = "%s:%d-%d" % (ref, a, a + rng.randrange(50, 2000))
        shutil.copy(small_bam + ".bai", str(tmp_path / "ref.bai"))
        got = run_cli(["base", "-L", reg, bam])
        assert got == run_cli(["base", "-L", reg, small_bam]) and len(got) > 100


def test_unsorted_input_is_rejected(tmp_path):
    recs = [bg.make_record(0, 500, "50M", "A" * 50, 30, name="x"), bg.make_record(0, 100, "50M", "C" * 50, 30, name="y")]
    bam = str(tmp_path / "u.bam")
    bg.write_bam(bam, [("k", 10000)], recs, write_index=False)
    with pytest.raises(sambamba_amd.SbxError) as e:
        sambamba_amd.build_index(bam)
    assert "not coordinate-sorted" in str(e.value)


def test_product_reads_device_written_bam_at_scale(tmp_path):
    """A few hundred thousand reads through gen_bam -> inflate -> device writer (default level: dynamic-code blocks) -> K1a / K1b."""
    src = gen_bam(str(tmp_path / "g.bam"), "chrA:3000000", coverage=20, seed=5)
    stream = gzip.decompress(open(src, "rb").read())
    out = str(tmp_path / "dev.bam")
    sambamba_amd.write_bam(out, stream, with_index=True)
    want = run_cli(["window", "-w", "10000", src])
    assert run_cli(["window", "-w", "10000", out]) == want
    assert run_cli(["base", "-L", "chrA:1500000-1501000", out]) == run_cli(["base", "-L", "chrA:1500000-1501000", src])


def test_generator_with_the_device_codec_writes_the_same_reads(tmp_path):
    """tools/gen_bam --codec device: the harness generator compressing through the product's writer (dlopen of libsbx_depth.so).  Same
    records, same index, different deflate streams: the oracle and the product read the same depth from both files."""
    a = gen_bam(str(tmp_path / "zlib.bam"), "chrA:400000,chrB:90000", coverage=20, seed=17)
    b = gen_bam(str(tmp_path / "dev.bam"), "chrA:400000,chrB:90000", coverage=20, seed=17, extra=["--codec", "device"])
    assert gzip.decompress(open(a, "rb").read()) == gzip.decompress(open(b, "rb").read())
    assert os.path.getsize(b) > os.path.getsize(a)            # greedy matches from one candidate per position: a larger file than zlib-6's
    want = run_oracle(["base", a])
    assert run_oracle(["base", b]) == want and run_cli(["base", b]) == want
    reg = ["base", "-L", "chrA:100000-101000"]
    assert run_cli(reg + [b]) == run_oracle(reg + [a])


@pytest.mark.parametrize("batch", ["1", "70000", "300000"])
def test_index_is_built_in_batches_of_blocks(small_bam, tmp_path, monkeypatch, batch):
    """IndexBuilder consumes a stream (bai/indexing.d:262-316); sbx_build_index sends the file through the device in batches of
    whole BGZF blocks, each ending in front of the record that straddles its last block boundary.  Whatever the batch size -- one
    block at a time (records longer than a batch make it grow), a few blocks, the whole file --, the index is the same file."""
    one = str(tmp_path / "one.bam")
    shutil.copy(small_bam, one)
    sambamba_amd.build_index(one)                       # (default: the batch follows the free device memory -- one batch here)
    many = str(tmp_path / "many.bam")
    shutil.copy(small_bam, many)
    monkeypatch.setenv("SBX_INDEX_BATCH_BYTES", batch)
    sambamba_amd.build_index(many)
    assert open(many + ".bai", "rb").read() == open(one + ".bai", "rb").read()
    # the records of a batch are consumed on the device (bai_parallel.hpp: one lane per record); IndexBuilder's loop restated on the
    # host (SBX_BAI_HOST=1: the path of irregular input) writes the same file
    host = str(tmp_path / "host.bam")
    shutil.copy(small_bam, host)
    monkeypatch.setenv("SBX_BAI_HOST", "1")
    sambamba_amd.build_index(host)
    assert open(host + ".bai", "rb").read() == open(one + ".bai", "rb").read()


def _build_index_in_a_process(bam, **env):
    """sbx_build_index in a python of its own with SBX_TIMING=1: (stderr, bytes of the index)"""
    import sys
    e = dict(os.environ, SBX_TIMING="1", **env)
    r = subprocess.run([sys.executable, "-c", "import sys, sambamba_amd; sambamba_amd.build_index(sys.argv[1])", bam], env=e, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    return r.stderr, open(bam + ".bai", "rb").read()


def test_index_records_are_consumed_on_the_device(small_bam, tmp_path):
    """Which consumer ran is not visible in the file (they write the same bytes), so the library says it (SBX_TIMING): the device by
    default; the serial builder with SBX_BAI_HOST=1 and for input the device formulation calls irregular -- here a read that reaches
    90 kbp beyond the end of its reference (more linear-index windows than the reference's length asks for)."""
    bam = str(tmp_path / "d.bam")
    shutil.copy(small_bam, bam)
    err, dev = _build_index_in_a_process(bam)
    assert b"records consumed on the device" in err
    err, host = _build_index_in_a_process(bam, SBX_BAI_HOST="1")
    assert b"by the serial builder on the host" in err and host == dev
    odd = str(tmp_path / "odd.bam")
    bg.write_bam(odd, [("c", 1000), ("d", 50000)], [bg.make_record(0, 900, "10M90000N10M", "A" * 20, 30, name="a"),
                                                      bg.make_record(1, 5, "20M", "C" * 20, 30, name="b")], write_index=False)
    err, got = _build_index_in_a_process(odd)
    assert b"by the serial builder on the host" in err
    from tests.test_bai_cpu import SRC as BAI_SRC
    exe = str(tmp_path / "bai_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe, BAI_SRC, "-lz"])
    subprocess.check_call([exe, odd, odd + ".host.bai"])
    assert got == open(odd + ".host.bai", "rb").read()


def test_device_index_of_quirky_bams_equals_the_serial_builder(tmp_path, monkeypatch):
    """tests/test_bai_cpu.py's random BAMs -- reads with a reference but no position, placed reads with the unmapped flag, reads
    without CIGAR, long skips, empty references, block boundaries on record boundaries --: the index the device builds in one batch
    and in batches of a block or two, against the serial builder fed by zlib on the host (tests/native/bai_host.cpp)."""
    from tests.test_bai_cpu import SRC as BAI_SRC, _quirky_bam
    exe = str(tmp_path / "bai_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe, BAI_SRC, "-lz"])
    for seed in range(2000, 2016):
        bam = str(tmp_path / ("q%d.bam" % seed))
        _quirky_bam(bam, seed)
        subprocess.check_call([exe, bam, bam + ".host.bai"])
        want = open(bam + ".host.bai", "rb").read()
        for batch in (None, "1500"):
            if batch:
                monkeypatch.setenv("SBX_INDEX_BATCH_BYTES", batch)
            sambamba_amd.build_index(bam)
            monkeypatch.delenv("SBX_INDEX_BATCH_BYTES", raising=False)
            assert open(bam + ".bai", "rb").read() == want, (seed, batch)


def test_index_batches_with_records_longer_than_a_block(tmp_path, monkeypatch):
    """Records of ~100 kB (long reads) span two BGZF blocks each: a batch of one block holds no whole record and has to grow;
    batch boundaries fall inside records, block boundaries inside the 36 fixed bytes of a record."""
    rng = np.random.default_rng(5)
    recs = []
    pos = 100
    for i in range(40):
        n = int(rng.integers(90_000, 140_000)) if i % 3 else int(rng.integers(30, 200))
        recs.append(bg.make_record(0, pos, "%dM" % n, "ACGT"[i % 4] * n, 30, name="long%d" % i))
        pos += int(rng.integers(1, 5000))
    path = str(tmp_path / "long.bam")
    bg.write_bam(path, [("chrL", 5_000_000)], recs, write_index=False)
    whole = str(tmp_path / "whole.bam")
    shutil.copy(path, whole)
    sambamba_amd.build_index(whole)
    for batch in ("1", "65280", "200000"):
        part = str(tmp_path / ("part%s.bam" % batch))
        shutil.copy(path, part)
        monkeypatch.setenv("SBX_INDEX_BATCH_BYTES", batch)
        sambamba_amd.build_index(part)
        monkeypatch.delenv("SBX_INDEX_BATCH_BYTES")
        assert open(part + ".bai", "rb").read() == open(whole + ".bai", "rb").read(), batch
    # ... and it is the index the host build of the same IndexBuilder restatement writes from a zlib reader (tests/native/bai_host.cpp)
    from tests.test_bai_cpu import SRC as BAI_SRC
    exe = str(tmp_path / "bai_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe, BAI_SRC, "-lz"])
    subprocess.check_call([exe, path, str(tmp_path / "host.bai")])
    assert open(str(tmp_path / "host.bai"), "rb").read() == open(whole + ".bai", "rb").read()


@pytest.mark.parametrize("name", ["issue225", "issue_193", "issue_204", "mate_overlaps_1_3M_4M"])
def test_device_index_equals_the_reference_index(tmp_path, name):
    """sbx_build_index on the reference's fixtures against the .bai files the reference ships for them (the host half of
    the same comparison: tests/test_bai_cpu.py): equal as structures, byte-identical where the reference wrote ascending bins."""
    from tests.test_bai_cpu import parse_bai
    from tests.util import GOLDEN
    bam = str(tmp_path / (name + ".bam"))
    shutil.copy(os.path.join(GOLDEN, name + ".bam"), bam)
    sambamba_amd.build_index(bam)
    mine, tail_m = parse_bai(bam + ".bai")
    ref, tail_r = parse_bai(os.path.join(GOLDEN, name + ".bam.bai"))
    assert tail_m == tail_r and len(mine) == len(ref)
    for (bm, lm, _), (br, lr, _) in zip(mine, ref):
        assert bm == br and lm == lr
    if all(o == sorted(o) for _, _, o in ref):
        assert open(bam + ".bai", "rb").read() == open(os.path.join(GOLDEN, name + ".bam.bai"), "rb").read()
