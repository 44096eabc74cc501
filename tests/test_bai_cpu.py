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
