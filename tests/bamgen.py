"""Tiny pure-Python BAM / BGZF / BAI writer for hand-made edge-case inputs (test harness only).

Lets a test control exactly what the device path sees: CIGAR shapes, record sizes, where BGZF
blocks are cut, which deflate block types are used, what the aux bytes look like.
"""
import struct
import zlib

CIGAR_OPS = "MIDNSHP=X"
SEQ_CODES = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
EOF_BLOCK = bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0,
                   0, 0, 0, 0, 0, 0, 0, 0])


def reg2bin(beg, end):
    if end <= beg:
        end = beg + 1
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def ref_span(cigar):
    return sum(n for op, n in cigar if op in "MDN=X")


def parse_cigar(s):
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((ch, int(num)))
            num = ""
    return out


def make_record(ref, pos, cigar, seq, qual, name="r", mapq=60, flag=0, tags=b"", next_ref=-1, next_pos=-1, tlen=0):
    """One BAM record (block_size prefix included). cigar: "10M2D5M" or list of (op, len)."""
    if isinstance(cigar, str):
        cigar = parse_cigar(cigar)
    l_seq = len(seq)
    nm = name.encode() + b"\0"
    end = pos + (ref_span(cigar) if not (flag & 4) and cigar else 1)
    body = struct.pack("<iiBBHHHiiii", ref, pos, len(nm), mapq, reg2bin(pos, end), len(cigar), flag, l_seq, next_ref,
                       next_pos, tlen)
    body += nm
    for op, n in cigar:
        body += struct.pack("<I", (n << 4) | CIGAR_OPS.index(op))
    packed = bytearray((l_seq + 1) // 2)
    for i, c in enumerate(seq):
        code = SEQ_CODES[c]
        packed[i >> 1] |= code << 4 if (i & 1) == 0 else code
    body += bytes(packed)
    if isinstance(qual, int):
        qual = [qual] * l_seq
    body += bytes(qual)
    body += tags
    return struct.pack("<i", len(body)) + body


def tag_z(key, value):
    return key.encode() + b"Z" + value.encode() + b"\0"


def tag_i(key, value):
    return key.encode() + b"i" + struct.pack("<i", value)


def tag_num(key, ty, value):
    """Numeric aux field of BAM type ty in cCsSiIf, or A (one character)."""
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    if ty == "A":
        return key.encode() + b"A" + value.encode()
    return key.encode() + ty.encode() + struct.pack(fmt[ty], value)


def tag_bytes(key, payload):
    """B:C array tag carrying arbitrary bytes."""
    return key.encode() + b"B" + b"C" + struct.pack("<I", len(payload)) + bytes(payload)


def bam_header(text, refs):
    h = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for name, length in refs:
        nm = name.encode() + b"\0"
        h += struct.pack("<i", len(nm)) + nm + struct.pack("<i", length)
    return h


def bgzf_block(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    c = co.compress(payload) + co.flush()
    total = 18 + len(c) + 8
    assert total <= 65536, "payload does not fit a BGZF block"
    hdr = bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0]) + struct.pack("<H", total - 1)
    return hdr + c + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


def write_bam(path, refs, records, text=None, block_size=0xFF00, cuts=None, level=6, levels=None, read_groups=(),
              write_index=True):
    """records: list of (ref, pos, end, bytes) or raw bytes made by make_record (then ref/pos are parsed).
    cuts: optional list of uncompressed offsets where a new BGZF block must start."""
    if text is None:
        text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
        for rg_id, sm in read_groups:
            text += "@RG\tID:%s\tSM:%s\n" % (rg_id, sm)
    hdr = bam_header(text, refs)
    stream = bytearray(hdr)
    rec_info = []
    for r in records:
        b = r if isinstance(r, (bytes, bytearray)) else r[-1]
        ref, pos, l_name, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", b, 4)
        cig = [struct.unpack_from("<I", b, 36 + l_name + 4 * i)[0] for i in range(n_cig)]
        span = sum(c >> 4 for c in cig if CIGAR_OPS[c & 15] in "MDN=X") if not (flag & 4) else 0
        rec_info.append((ref, pos, pos + max(span, 1), len(stream), len(stream) + len(b)))
        stream += b
    # cut into blocks
    bounds = sorted(set([0, len(stream)] + [c for c in (cuts or []) if 0 < c < len(stream)]))
    pieces = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        o = a
        while o < b:
            n = min(block_size, b - o)
            pieces.append((o, n))
            o += n
    out = bytearray()
    blk_coff = []
    for i, (o, n) in enumerate(pieces):
        lv = levels[i % len(levels)] if levels else level
        blk_coff.append(len(out))
        out += bgzf_block(bytes(stream[o:o + n]), level=lv)
    blk_coff.append(len(out))
    out += EOF_BLOCK
    with open(path, "wb") as fh:
        fh.write(out)

    def voff(u):
        # index of the block containing uncompressed offset u
        lo, hi = 0, len(pieces)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if pieces[mid][0] <= u:
                lo = mid
            else:
                hi = mid
        if u >= pieces[lo][0] + pieces[lo][1]:
            return (blk_coff[lo + 1] << 16)
        return (blk_coff[lo] << 16) | (u - pieces[lo][0])

    if write_index:
        n_ref = len(refs)
        bins = [dict() for _ in range(n_ref)]
        lin = [dict() for _ in range(n_ref)]
        for ref, pos, end, ub, ue in rec_info:
            if ref < 0:
                continue
            vb, ve = voff(ub), voff(ue)
            b = reg2bin(pos, end)
            ch = bins[ref].setdefault(b, [])
            if ch and ch[-1][1] >= vb:
                ch[-1][1] = ve
            else:
                ch.append([vb, ve])
            for w in range(pos >> 14, ((end - 1) >> 14) + 1):
                if w not in lin[ref] or vb < lin[ref][w]:
                    lin[ref][w] = vb
        with open(path + ".bai", "wb") as fh:
            fh.write(b"BAI\1" + struct.pack("<i", n_ref))
            for r in range(n_ref):
                fh.write(struct.pack("<i", len(bins[r])))
                for b in sorted(bins[r]):
                    fh.write(struct.pack("<Ii", b, len(bins[r][b])))
                    for vb, ve in bins[r][b]:
                        fh.write(struct.pack("<QQ", vb, ve))
                n_intv = (max(lin[r]) + 1) if lin[r] else 0
                fh.write(struct.pack("<i", n_intv))
                prev = 0
                for w in range(n_intv):
                    prev = lin[r].get(w, prev)
                    fh.write(struct.pack("<Q", prev))
    return {"header_len": len(hdr), "records": rec_info, "pieces": pieces, "stream_len": len(stream)}
