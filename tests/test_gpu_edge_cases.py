"""Edge cases of the hot path on hand-made BAMs (tests/bamgen.py): CIGAR shapes, tile borders,
records straddling BGZF blocks, long reads that leave BGZF blocks without any record start, a
decoy that fools the record-start guesser (must be repaired, never change results), empty inputs,
stored/fixed deflate blocks inside a BAM.  Every case: CLI text == oracle text and device counters
== oracle counters (bit-exact)."""
import os
import random
import struct

import numpy as np
import pytest

from tests import bamgen as bg
from tests.util import oracle_base_counters, run_cli, run_oracle

pytestmark = pytest.mark.gpu


def _check(path, refs, min_bq_list=(0, 20), extra_cli=()):
    import sambamba_amd
    for args in [["base"], ["base", "-c", "0"], ["base", "-q", "20", "-a"]] + [list(a) for a in extra_cli]:
        got, want = run_cli(args + [path]), run_oracle(args + [path])
        if got != want:
            gl, wl = got.decode().splitlines(), want.decode().splitlines()
            for k in range(max(len(gl), len(wl))):
                a = gl[k] if k < len(gl) else "<missing>"
                b = wl[k] if k < len(wl) else "<missing>"
                if a != b:
                    raise AssertionError("%s: first difference at line %d:\n  device: %r\n  oracle: %r\n  (%d vs %d lines)" % (
                        args, k, a, b, len(gl), len(wl)))
    for q in min_bq_list:
        with sambamba_amd.Depth(path) as d:
            d.set_params(min_bq=q)
            d.run()
            for ref, (_, n) in enumerate(refs):
                got = d.base_counters(ref, 0, n)
                want = oracle_base_counters(path, ref, 0, n, min_bq=q)
                assert np.array_equal(got, want), (q, ref)


def _rand_seq(rng, n):
    return "".join(rng.choice("ACGTN") if rng.random() < 0.02 else rng.choice("ACGT") for _ in range(n))


def test_cigar_zoo(tmp_path):
    rng = random.Random(5)
    refs = [("c1", 5000), ("c2", 3000)]
    shapes = ["50M", "10S40M", "40M10S", "5H10S30M5S5H", "20M5I25M", "20M7D30M", "10M100N40M", "25=1X24=", "3I47M",
              "47M3I", "10M2P40M", "10M0D40M", "1M", "5S1M5S", "20M3D2I28M", "10M5N5D35M", "30M2000N20M", "50M",
              "5M1D1M1D1M1D42M", "2S3I45M"]
    recs = []
    pos = 100
    for i, sh in enumerate(shapes * 3):
        cig = bg.parse_cigar(sh)
        l_seq = sum(n for op, n in cig if op in "MIS=X")
        qual = [rng.choice([2, 12, 23, 37]) for _ in range(l_seq)]
        if i == 40:
            pos = 3       # second contig starts over near its beginning
        recs.append((0 if i < 40 else 1, pos, sh, _rand_seq(rng, l_seq), qual, "q%d" % i))
        pos += rng.randint(0, 60)
    recs.sort(key=lambda r: (r[0], r[1]))
    raw = [bg.make_record(r[0], r[1], r[2], r[3], r[4], name=r[5], flag=rng.choice([0, 16, 99, 147])) for r in recs]
    p = str(tmp_path / "zoo.bam")
    bg.write_bam(p, refs, raw)
    _check(p, refs, extra_cli=[["base", "-L", "c1:150-400"], ["base", "-c", "0", "-L", "c2:1-50"]])


def test_flags_mapq_filter_and_unmapped(tmp_path):
    rng = random.Random(6)
    refs = [("c1", 4000)]
    raw = []
    pos = 10
    for i in range(300):
        flag = rng.choice([0, 4, 0x400, 0x200, 0x100, 0x800, 99, 147, 16])
        mapq = rng.choice([0, 1, 20, 60])
        raw.append(bg.make_record(0, pos, "60M", _rand_seq(rng, 60), [rng.choice([2, 12, 23, 37]) for _ in range(60)],
                                  name="f%d" % i, flag=flag, mapq=mapq))
        pos += rng.randint(0, 12)
    # placed-unmapped read (flag 4 with a position) and a read without CIGAR
    raw.append(bg.make_record(0, pos, "", _rand_seq(rng, 30), 30, name="nocigar", flag=0))
    raw.append(bg.make_record(-1, -1, "", _rand_seq(rng, 30), 30, name="unmapped", flag=4))
    p = str(tmp_path / "flags.bam")
    bg.write_bam(p, refs, raw)
    _check(p, refs, extra_cli=[["base", "-F", "mapping_quality >= 20"], ["base", "-F", "not duplicate or secondary_alignment"],
                               ["base", "-F", "mapping_quality >= 0"]])


def test_tile_borders_and_contig_overhang(tmp_path):
    rng = random.Random(7)
    refs = [("c1", 4100), ("c2", 1024), ("c3", 1030)]
    raw = []
    for ref, (_, n) in enumerate(refs):
        starts = sorted([0, 1, 1020, 1021, 1022, 1023, 1024, 1025, 2047, 2048, 3000] + [rng.randint(0, n - 1) for _ in range(60)])
        for i, s in enumerate(starts):
            if s >= n:
                continue
            ln = rng.choice([1, 7, 50, 150, 300])
            raw.append(bg.make_record(ref, s, "%dM" % ln, _rand_seq(rng, ln), [rng.choice([2, 37]) for _ in range(ln)],
                                      name="t%d_%d" % (ref, i)))
    p = str(tmp_path / "tiles.bam")
    bg.write_bam(p, refs, raw)
    _check(p, refs)


def _overhang_bam(path, with_far=True):
    rng = random.Random(3)
    refs = [("c1", 3000), ("c2", 2000)]
    recs = [bg.make_record(0, p, "100M", _rand_seq(rng, 100), 30, name="a%d" % p) for p in (100, 2500, 2890)]
    if with_far:
        # ends 5,100 positions behind the contig's last one: five tiles beyond the spare tile the engine starts with
        recs.append(bg.make_record(0, 2950, "100M5000N50M", _rand_seq(rng, 150), [rng.choice([2, 37]) for _ in range(150)], name="long"))
    recs.append(bg.make_record(0, 2960, "60M", _rand_seq(rng, 60), 30, name="b"))
    if with_far:
        recs.append(bg.make_record(0, 4100, "50M", _rand_seq(rng, 50), 30, name="beyond"))     # STARTS beyond the contig's end
    recs.append(bg.make_record(1, 10, "50M", _rand_seq(rng, 50), 30, name="c"))
    bg.write_bam(path, refs, recs)
    return refs


def test_alignments_far_beyond_the_contig_end_have_all_their_columns(tmp_path):
    """The reference's sweep prints every column a read covers, whatever the contig's length says (pileup.d:345-397,
    depth.d:567-591): an alignment that ends thousands of positions behind the last one makes the engine lay out more spare
    tiles and repeat the pass -- nothing is clipped."""
    import sambamba_amd
    p = str(tmp_path / "over.bam")
    refs = _overhang_bam(p)
    for args in (["base"], ["base", "-c", "0"], ["base", "-q", "20", "-a"], ["base", "-m"], ["window", "-w", "500"],
                 ["window", "-w", "300", "-m"], ["region", "-L", "c1:2901-9000", "-T", "1"], ["base", "-L", "c1:2000-8060"]):
        got, want = run_cli(args + [p]), run_oracle(args + [p])
        assert got == want, args
    assert b"c1\t8099\t" in run_cli(["base", p]) and b"c1\t4149\t2\t" in run_cli(["base", p])
    with sambamba_amd.Depth(p) as d:
        d.set_params()
        d.run()
        assert d.active_end(0) >= 8100
        got = d.base_counters(0, 0, 8192)
        assert np.array_equal(got, oracle_base_counters(p, 0, 0, 8192))
        d.run()                  # the enlarged layout is kept: same answer from a second run of the context
        assert np.array_equal(d.base_counters(0, 0, 8192), got)


def test_several_files_share_the_enlarged_spare_tiles(tmp_path):
    a, b, m = str(tmp_path / "plain.bam"), str(tmp_path / "over.bam"), str(tmp_path / "merged.bam")
    _overhang_bam(a, with_far=False)
    _overhang_bam(b)
    # the merged stream of both files, by hand: a's reads come first among equal positions
    import struct
    from tests.util import oracle_inflate_all

    def records(path):
        u = bytes(oracle_inflate_all(path))
        l_text = struct.unpack_from("<i", u, 4)[0]
        o = 8 + l_text
        n_ref = struct.unpack_from("<i", u, o)[0]
        o += 4
        for _ in range(n_ref):
            o += 4 + struct.unpack_from("<i", u, o)[0] + 4
        out = []
        while o < len(u):
            bs = struct.unpack_from("<i", u, o)[0]
            out.append(u[o:o + 4 + bs])
            o += 4 + bs
        return out
    ra = [(struct.unpack_from("<ii", r, 4), 0, k, r) for k, r in enumerate(records(a))]
    rb = [(struct.unpack_from("<ii", r, 4), 1, k, r) for k, r in enumerate(records(b))]
    allr = sorted(ra + rb, key=lambda t: (t[0], t[1], t[2]))
    bg.write_bam(m, [("c1", 3000), ("c2", 2000)], [t[3] for t in allr])
    for args in (["base"], ["window", "-w", "500"]):
        assert run_cli(args + [a, b]) == run_oracle(args + [m]), args


def test_records_straddling_small_bgzf_blocks(tmp_path):
    rng = random.Random(8)
    refs = [("c1", 20000)]
    raw = []
    pos = 0
    for i in range(400):
        ln = rng.choice([36, 76, 151])
        raw.append(bg.make_record(0, pos, "%dM" % ln, _rand_seq(rng, ln), [rng.choice([2, 12, 23, 37]) for _ in range(ln)],
                                  name="s%05d" % i, tags=bg.tag_z("RG", "g1") + bg.tag_i("NM", i)))
        pos += rng.randint(5, 40)
    for bs, name in ((777, "b777"), (1, "never"), (4093, "b4093")):
        if bs == 1:
            continue
        p = str(tmp_path / (name + ".bam"))
        bg.write_bam(p, refs, raw, block_size=bs, levels=[0, 1, 6, 9], read_groups=[("g1", "sampleA")])
        _check(p, refs, min_bq_list=(0,))


def test_long_reads_leave_blocks_without_record_start(tmp_path):
    rng = random.Random(9)
    refs = [("c1", 600000)]
    raw = []
    pos = 1000
    for i in range(12):
        ln = rng.choice([70000, 150000, 9000, 200])
        raw.append(bg.make_record(0, pos, "%dM" % ln, _rand_seq(rng, ln), [rng.choice([12, 37]) for _ in range(ln)],
                                  name="long%d" % i))
        pos += rng.randint(100, 20000)
    p = str(tmp_path / "long.bam")
    bg.write_bam(p, refs, raw)
    _check(p, refs, min_bq_list=(0, 20))


def test_decoy_records_inside_a_tag_cannot_change_results(tmp_path):
    """A B:C aux array holding a chain of three perfectly plausible fake records, with a BGZF block
    cut exactly at the first fake record: the per-block guesser picks the decoy, chain_check flags
    the block and chain_repair restores the true chain."""
    rng = random.Random(10)
    refs = [("c1", 50000)]
    fake = b"".join(bg.make_record(0, 100 + 10 * k, "50M", _rand_seq(rng, 50), 30, name="fake%d" % k) for k in range(3))
    raw = []
    pos = 10
    for i in range(60):
        tags = bg.tag_bytes("XD", fake) if i == 20 else b""
        raw.append(bg.make_record(0, pos, "80M", _rand_seq(rng, 80), [rng.choice([2, 37]) for _ in range(80)],
                                  name="d%d" % i, tags=tags))
        pos += 25
    # uncompressed offset of the decoy inside record 20
    hdr_len = len(bg.bam_header("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:50000\n", refs))
    off = hdr_len + sum(len(r) for r in raw[:20])
    rec20 = raw[20]
    decoy_at = off + rec20.index(fake)
    p = str(tmp_path / "decoy.bam")
    bg.write_bam(p, refs, raw, cuts=[decoy_at])
    _check(p, refs, min_bq_list=(0,))
    import sambamba_amd
    with sambamba_amd.Depth(p) as d:
        d.set_params()
        st = d.run()
        assert st["n_records"] == 60


def test_header_only_and_unmapped_only(tmp_path):
    refs = [("c1", 1000), ("c2", 500)]
    p = str(tmp_path / "empty.bam")
    bg.write_bam(p, refs, [])
    for args in (["base"], ["base", "-c", "0"], ["base", "-c", "0", "-L", "c2:10-20"]):
        assert run_cli(args + [p]) == run_oracle(args + [p]), args
    p2 = str(tmp_path / "unm.bam")
    bg.write_bam(p2, refs, [bg.make_record(-1, -1, "", "ACGT", 30, name="u%d" % i, flag=4) for i in range(5)])
    for args in (["base"], ["base", "-c", "0"]):
        assert run_cli(args + [p2]) == run_oracle(args + [p2]), args


def test_unknown_read_group_is_an_error(tmp_path):
    refs = [("c1", 1000)]
    raw = [bg.make_record(0, 10, "20M", "A" * 20, 30, name="a", tags=bg.tag_z("RG", "nope"))]
    p = str(tmp_path / "rg.bam")
    bg.write_bam(p, refs, raw, read_groups=[("g1", "s1")])
    r = run_cli(["base", p], check=False)
    assert r.returncode == 1 and b"read group" in r.stderr and b"not present in the header" in r.stderr
