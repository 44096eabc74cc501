"""The BAI builder (csrc/bai_writer.hpp: the restatement of IndexBuilder, BioD/bio/std/hts/bam/bai/indexing.d:52-346, and the
virtual-offset rule of sbx_build_index) against the index files the REFERENCE's own test-suite ships next to its BAMs
(tests/golden/*.bam.bai: checked into the reference next to the BAMs; which indexer wrote them is not recorded -- IndexBuilder
and samtools produce the same bytes for these files up to the order of the bins).  The builder is compiled for the host and fed by a zlib reader
(tests/native/bai_host.cpp); on the device the same two classes get their record fields from K2 (tests/test_gpu_writer.py).
Bins, chunks, linear index, the metadata pseudo-bin and the no-coordinate trailer must be equal; the ORDER of the bins of a
reference is unspecified in the reference (it iterates a D associative array), so files are compared as structures, and byte
for byte where the reference happened to write ascending bin ids."""
import os
import struct
import subprocess

import pytest

from tests.util import GOLDEN, ROOT

SRC = os.path.join(ROOT, "tests", "native", "bai_host.cpp")
NAMES = ["issue225", "issue_193", "issue_204", "mate_overlaps_1_3M_4M"]


@pytest.fixture(scope="module")
def bai_host(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bai") / "bai_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe, SRC, "-lz"])
    return exe


def parse_bai(path):
    b = open(path, "rb").read()
    assert b[:4] == b"BAI\1"
    n_ref = struct.unpack_from("<i", b, 4)[0]
    p = 8
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", b, p)[0]; p += 4
        bins, order = {}, []
        for _ in range(n_bin):
            bid, n_ch = struct.unpack_from("<Ii", b, p); p += 8
            assert bid not in bins
            bins[bid] = [struct.unpack_from("<QQ", b, p + 16 * k) for k in range(n_ch)]
            order.append(bid)
            p += 16 * n_ch
        n_intv = struct.unpack_from("<i", b, p)[0]; p += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, b, p)); p += 8 * n_intv
        refs.append((bins, lin, order))
    return refs, b[p:]


@pytest.mark.parametrize("name", NAMES)
def test_index_equals_the_reference_index(bai_host, tmp_path, name):
    bam = os.path.join(GOLDEN, name + ".bam")
    out = str(tmp_path / "x.bai")
    subprocess.check_call([bai_host, bam, out])
    mine, tail_m = parse_bai(out)
    ref, tail_r = parse_bai(bam + ".bai")
    assert tail_m == tail_r                                  # n_no_coor
    assert len(mine) == len(ref)
    n_bins = 0
    for r, ((bm, lm, om), (br, lr, orr)) in enumerate(zip(mine, ref)):
        assert bm == br, "reference %d: bins / chunks / metadata differ" % r
        assert lm == lr, "reference %d: linear index differs" % r
        assert om == sorted(om)
        n_bins += len(bm)
    assert os.path.getsize(out) == os.path.getsize(bam + ".bai")
    if all(o == sorted(o) for _, _, o in ref):
        assert open(out, "rb").read() == open(bam + ".bai", "rb").read()
    assert n_bins > 0


def test_three_of_the_fixtures_are_byte_identical(bai_host, tmp_path):
    same = 0
    for name in NAMES:
        out = str(tmp_path / (name + ".bai"))
        subprocess.check_call([bai_host, os.path.join(GOLDEN, name + ".bam"), out])
        same += open(out, "rb").read() == open(os.path.join(GOLDEN, name + ".bam.bai"), "rb").read()
    assert same >= 3


# ---- virtual offsets at block boundaries (VoffCursor): hand-made files --------------------------------------------------------
def _records(n, ref=0, start=100, step=50):
    import random
    from tests import bamgen as bg
    rng = random.Random(11)
    return [bg.make_record(ref, start + step * i, "40M", "".join(rng.choice("ACGT") for _ in range(40)), 30, name="r%d" % i)
            for i in range(n)]


def _blocks(path):
    b = open(path, "rb").read()
    out, p = [], 0
    while p + 18 <= len(b):
        bsize = struct.unpack_from("<H", b, p + 16)[0] + 1
        out.append((p, struct.unpack_from("<I", b, p + bsize - 4)[0]))
        p += bsize
    return out, len(b)


def test_header_that_fills_its_block_and_end_of_file_offsets(bai_host, tmp_path):
    """Block 0 holds exactly the header: the first read starts at offset 0 of block 1; the end of the last read is offset 0 of the
    EOF block (not the end of the file)."""
    from tests import bamgen as bg
    refs = [("k1", 100000)]
    recs = _records(300)
    probe = bg.write_bam(str(tmp_path / "probe.bam"), refs, recs, write_index=False)
    bam = str(tmp_path / "a.bam")
    bg.write_bam(bam, refs, recs, cuts=[probe["header_len"]], block_size=4000, write_index=False)
    out = str(tmp_path / "a.bai")
    subprocess.check_call([bai_host, bam, out])
    refs_b, tail = parse_bai(out)
    blocks, size = _blocks(bam)
    assert blocks[0][1] == probe["header_len"] and blocks[-1][1] == 0           # header block, EOF block
    bins, lin, _ = refs_b[0]
    meta = bins.pop(37450)
    assert meta[0] == (blocks[1][0] << 16, blocks[-1][0] << 16)                   # first read .. end of the last read
    assert min(c[0] for ch in bins.values() for c in ch) == blocks[1][0] << 16
    assert max(c[1] for ch in bins.values() for c in ch) == blocks[-1][0] << 16
    assert lin[0] == blocks[1][0] << 16


def test_file_without_eof_block_and_empty_block_in_the_middle(bai_host, tmp_path):
    """Without an EOF block the end of the last read is the end of the file; an empty block in the middle of the data is where
    the read in front of it ends (the block that starts there), while the read behind it starts in the next block that holds
    bytes."""
    from tests import bamgen as bg
    refs = [("k1", 100000)]
    recs = _records(200)
    info = bg.write_bam(str(tmp_path / "p.bam"), refs, recs, write_index=False)
    cut = info["records"][100][3]                                                  # block boundary in front of read 100
    bam = str(tmp_path / "b.bam")
    bg.write_bam(bam, refs, recs, cuts=[cut], block_size=3000, write_index=False)
    raw = open(bam, "rb").read()
    blocks, size = _blocks(bam)
    assert blocks[-1][1] == 0
    # the block that starts at `cut`: the first one whose inflated start equals cut
    u, at = 0, None
    for i, (coff, isz) in enumerate(blocks):
        if u == cut:
            at = i
            break
        u += isz
    assert at is not None
    eof = raw[blocks[-1][0]:]
    mod = raw[:blocks[at][0]] + eof + raw[blocks[at][0]:blocks[-1][0]]            # empty block inserted at the cut, EOF block dropped
    open(bam, "wb").write(mod)
    out = str(tmp_path / "b.bai")
    subprocess.check_call([bai_host, bam, out])
    refs_b, tail = parse_bai(out)
    blocks2, size2 = _blocks(bam)
    assert blocks2[at][1] == 0 and blocks2[-1][1] != 0
    bins, lin, _ = refs_b[0]
    meta = bins.pop(37450)
    assert meta[0][1] == size2 << 16                                               # no EOF block: the file's end
    assert meta[1] == (200, 0)
    # all reads share one bin here: one chunk from the first read to the end (chunks are only cut when the bin changes)
    chunks = [c for ch in bins.values() for c in ch]
    assert min(c[0] for c in chunks) >> 16 == blocks2[0][0] and max(c[1] for c in chunks) == size2 << 16
    # and the same file through a reader that needs the index: the oracle fetches reads on both sides of the empty block
    os.replace(out, bam + ".bai")
    from tests.util import run_oracle
    whole = run_oracle(["base", bam])
    pos100 = info["records"][100][1]
    part = run_oracle(["base", "-L", "k1:%d-%d" % (pos100 - 200, pos100 + 200), bam])
    rows = {ln.split(b"\t")[1]: ln for ln in whole.splitlines()[1:]}
    for ln in part.splitlines()[1:]:
        assert rows[ln.split(b"\t")[1]] == ln
    assert len(part.splitlines()) > 100


# ---- the data-parallel formulation (csrc/bai_parallel.hpp: what the device runs, one lane per record) against the serial builder ----
def _quirky_bam(path, seed):
    """Everything IndexBuilder treats specially, at random: reads with a reference but no position in front of / between / behind the
    placed ones, placed reads with the unmapped flag, reads without CIGAR, long reference skips across 16 kbp windows, empty
    references in front, in the middle and at the end, a tail of reads without coordinates, small BGZF blocks and block boundaries
    that fall on record boundaries."""
    import random

    from tests import bamgen as bg
    rng = random.Random(seed)
    n_ref = rng.randrange(1, 7)
    refs = [("q%d" % k, rng.choice((300, 20000, 70000, 400000))) for k in range(n_ref)]
    recs = []
    for r in range(n_ref):
        if rng.random() < 0.3:
            continue                                                            # an empty reference
        pos = rng.randrange(0, 200)
        L = refs[r][1]
        if rng.random() < 0.3:
            recs.append(bg.make_record(r, -1, "", "ACGT", 30, name="nopos%d" % r, flag=4))       # reference, no position
        while pos < L - 10:
            kind = rng.randrange(10)
            if kind == 0:
                recs.append(bg.make_record(r, pos, "", "ACGTAC", 30, name="pu%d_%d" % (r, pos), flag=4))      # placed, unmapped
            elif kind == 1:
                recs.append(bg.make_record(r, pos, "", "ACGTAC", 30, name="nc%d_%d" % (r, pos)))              # no CIGAR
            elif kind == 2 and pos + 1100 < L:
                n = rng.randrange(1000, min(40000, L - pos - 20))
                recs.append(bg.make_record(r, pos, "10M%dN10M" % n, "A" * 20, 30, name="sk%d_%d" % (r, pos)))
            elif kind == 3:
                recs.append(bg.make_record(r, pos, "5S20M2I8M3D10M", "C" * 45, 30, name="x%d_%d" % (r, pos)))
            else:
                n = rng.choice((36, 100, 150))
                recs.append(bg.make_record(r, pos, "%dM" % n, "G" * n, 30, name="m%d_%d" % (r, pos)))
            pos += rng.choice((0, 0, 1, 7, 40, 300, 5000))
        if rng.random() < 0.2:
            recs.append(bg.make_record(r, L - 1, "", "ACGT", 30, name="tail%d" % r, flag=4))
    for k in range(rng.randrange(0, 9)):
        recs.append(bg.make_record(-1, -1, "", "ACGT", 30, name="nocoor%d" % k, flag=4))
    cuts = None
    if rng.random() < 0.5 and recs:
        # block boundaries on record boundaries (the "behind the last byte of a block" rule of the virtual offsets)
        hdr_len = len(bg.bam_header("@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs), refs))
        offs, o = [], hdr_len
        for rec in recs:
            offs.append(o)
            o += len(rec)
        cuts = sorted(set(rng.sample(offs, min(len(offs), rng.randrange(1, 6)))))
    bg.write_bam(path, refs, recs, block_size=rng.choice((600, 3000, 0xFF00)), cuts=cuts, write_index=False)
    return len(recs)


@pytest.mark.parametrize("name", NAMES)
def test_parallel_formulation_equals_the_serial_builder_on_the_fixtures(bai_host, tmp_path, name):
    bam = os.path.join(GOLDEN, name + ".bam")
    serial = str(tmp_path / "s.bai")
    subprocess.check_call([bai_host, bam, serial])
    for seed in (1, 2, 3):                       # other batch sizes, another order of the records
        par = str(tmp_path / ("p%d.bai" % seed))
        subprocess.check_call([bai_host, bam, par, "--parallel", str(seed)])
        assert open(par, "rb").read() == open(serial, "rb").read()


def test_parallel_formulation_equals_the_serial_builder_on_quirky_bams(bai_host, tmp_path):
    n_total = 0
    for seed in range(60):
        bam = str(tmp_path / ("q%d.bam" % seed))
        n_total += _quirky_bam(bam, seed)
        serial, par = str(tmp_path / "s.bai"), str(tmp_path / "p.bai")
        subprocess.check_call([bai_host, bam, serial])
        subprocess.check_call([bai_host, bam, par, "--parallel", str(seed + 100)])
        assert open(par, "rb").read() == open(serial, "rb").read(), seed
    assert n_total > 5000


def test_parallel_formulation_refuses_what_the_loop_treats_specially(bai_host, tmp_path):
    """unsorted input: the step raises `irregular` (exit 3 of the harness); the engine then runs the serial builder, which words the
    reference's error"""
    from tests import bamgen as bg
    bam = str(tmp_path / "u.bam")
    recs = [bg.make_record(0, 500, "10M", "A" * 10, 30, name="a"), bg.make_record(0, 100, "10M", "A" * 10, 30, name="b")]
    bg.write_bam(bam, [("c", 1000)], recs, write_index=False)
    assert subprocess.run([bai_host, bam, str(tmp_path / "p.bai"), "--parallel", "1"], stderr=subprocess.DEVNULL).returncode == 3
    assert subprocess.run([bai_host, bam, str(tmp_path / "s.bai")], stderr=subprocess.DEVNULL).returncode == 1
    # a read far beyond the end of its reference: more linear-index windows than the reference's length asks for
    bam = str(tmp_path / "o.bam")
    bg.write_bam(bam, [("c", 1000)], [bg.make_record(0, 900, "10M90000N10M", "A" * 20, 30, name="a")], write_index=False)
    assert subprocess.run([bai_host, bam, str(tmp_path / "p.bai"), "--parallel", "1"], stderr=subprocess.DEVNULL).returncode == 3
    assert subprocess.run([bai_host, bam, str(tmp_path / "s.bai")], stderr=subprocess.DEVNULL).returncode == 0


def test_stateless_virtual_offsets_equal_the_cursor(bai_host):
    """bai_parallel.hpp finds the virtual offsets of a record by binary search in the block table; the serial builder moves a cursor
    (bai_writer.hpp VoffCursor, the rule of inputstream.d:497-530).  Random tables with empty blocks, 2,000 per seed."""
    for seed in (1, 2, 3):
        assert subprocess.run([bai_host, "--vo-selftest", str(seed)]).returncode == 0
