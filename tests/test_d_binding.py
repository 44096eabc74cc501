"""The D binding (d/sbx_depth.d) is source only -- the image has no D compiler -- so its declarations are checked
mechanically against the C ABI: every struct is laid out with the C rules from its D field list and its size
compared with sbx_abi_sizeof() of the built library, and the extern(C) function list must be the header's."""
import os
import re

from tests.util import ROOT

D_FILE = os.path.join(ROOT, "d", "sbx_depth.d")
SCALARS = {"ubyte": 1, "byte": 1, "char": 1, "bool": 1, "ushort": 2, "short": 2, "uint": 4, "int": 4, "float": 4,
           "ulong": 8, "long": 8, "double": 8, "size_t": 8}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"/\+.*?\+/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _d_source():
    return _strip_comments(open(D_FILE).read())


def _enums(text):
    return {m.group(1): int(m.group(2), 0) for m in re.finditer(r"\benum\s+(\w+)\s*=\s*(-?\w+)\s*;", text)}


def _extern_block(text):
    m = re.search(r"extern\s*\(C\)[^{]*\{", text)
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1]


def _structs(block, enums):
    """name -> [(type, [dims innermost-first], field)] for `struct name { ... }` declarations."""
    out = {}
    for m in re.finditer(r"\bstruct\s+(\w+)\s*\{([^}]*)\}", block):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fm = re.match(r"^(\w+)((?:\s*\[\s*\w+\s*\])*)\s+(\w+)$", decl)
            assert fm, "unparsed D field declaration: %r" % decl
            dims = [enums[d] if d in enums else int(d, 0) for d in re.findall(r"\[\s*(\w+)\s*\]", fm.group(2))]
            fields.append((fm.group(1), dims, fm.group(3)))
        out[m.group(1)] = fields
    return out


def _layout(name, structs, cache):
    """(size, alignment) under the C ABI (natural alignment, tail padding) -- D's extern(C) structs follow it."""
    if name in SCALARS:
        return SCALARS[name], SCALARS[name]
    if name in cache:
        return cache[name]
    off, align = 0, 1
    for ty, dims, _ in structs[name]:
        sz, al = _layout(ty, structs, cache)
        n = 1
        for d in dims:
            n *= d
        off = (off + al - 1) // al * al
        off += sz * n
        align = max(align, al)
    cache[name] = ((off + align - 1) // align * align, align)
    return cache[name]


def test_d_struct_sizes_match_the_library():
    import sambamba_amd
    L = sambamba_amd.lib()
    text = _d_source()
    structs = _structs(_extern_block(text), _enums(text))
    checked = 0
    for name in ("sbx_region", "sbx_region_stats", "sbx_header_info", "sbx_regex_state", "sbx_regex", "sbx_filter_op",
                 "sbx_filter", "sbx_batch", "sbx_run_stats"):
        assert name in structs, "d/sbx_depth.d does not declare %s" % name
        want = L.sbx_abi_sizeof(name.encode())
        assert want > 0, name
        got = _layout(name, structs, {})[0]
        assert got == want, "%s: D declaration is %d bytes, the C struct is %d" % (name, got, want)
        checked += 1
    assert checked == 9
    assert L.sbx_abi_sizeof(b"no_such_struct") == 0


def test_d_struct_fields_match_the_header():
    """Same field names in the same order as include/sbx_depth.h (sizes alone would not catch a swap of equal-sized fields)."""
    header = _strip_comments(open(os.path.join(ROOT, "include", "sbx_depth.h")).read())
    text = _d_source()
    structs = _structs(_extern_block(text), _enums(text))
    for m in re.finditer(r"typedef\s+struct\s*\{([^}]*)\}\s*(\w+)\s*;", header):
        name = m.group(2)
        c_fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            body = re.sub(r"^\s*(?:const\s+)?\w+\s+", "", decl, count=1)      # drop the type
            for item in body.split(","):
                c_fields.append(re.match(r"\s*(\w+)", item).group(1))
        assert name in structs, name
        assert [f for _, _, f in structs[name]] == c_fields, name


def test_d_declares_every_function_of_the_header():
    header = _strip_comments(open(os.path.join(ROOT, "include", "sbx_depth.h")).read())
    want = sorted(set(re.findall(r"\b(sbx_[a-z_0-9]+)\s*\(", header)))
    block = _extern_block(_d_source())
    got = sorted(set(re.findall(r"\b(sbx_[a-z_0-9]+)\s*\(", block)))
    assert got == want


def test_d_glue_is_code_not_a_comment():
    text = _d_source()
    assert re.search(r"\bbool\s+sbxDepthRun\s*\(", text) and "sbx_stream_base_rows(ctx" in text and "sbx_depth_window_stats(ctx" in text


def test_d_glue_covers_or_refuses_every_stateful_corner_of_the_compiled_host():
    """d/sbx_depth.d cannot be compiled here; what can be checked is that, for every order-dependent rule cli.cpp implements
    (and the GPU tests pin against the oracle), the D glue either carries the same rule or hands the job back to the
    reference's CPU path (`return false`) -- never prints something else."""
    d = open(os.path.join(ROOT, "d", "sbx_depth.d")).read()
    cli = open(os.path.join(ROOT, "sambamba_amd", "csrc", "cli.cpp")).read()
    body = d[d.index("bool sbxDepthRun("):]
    # the one refusal left (round 6: the two base-mode corners are printed on the host from the device's counters, as cli.cpp does)
    assert "o.mode == SBX_MODE_WINDOW && o.overlap != 0) return false" in body                            # window --overlap
    assert body[:body.index("sbx_ctx* sbxOpen(")].count("return false") == 1
    # base -L with -c 0, and base -c 0 with over-hanging alignments: PerBasePrinter's rules, member for member of cli.cpp's BasePrinter
    for cli_name, d_name in (("void write_empty(", "void writeEmpty("), ("void write_column(", "void writeColumn("), ("void push(", "void push("),
                             ("void close()", "void close()"), ("bool output_required(", "bool outputRequired("), ("void init_tails()", "void initTails()")):
        assert cli_name in cli and d_name in d, (cli_name, d_name)
    assert "const bool base_host = o.mode == SBX_MODE_BASE && o.merged_bed.length && o.min_cov <= 0;" in body
    assert "if (base_host) { hp.runRefs(r0, r1); continue; }" in body and "if (base_host) hp.close();" in body
    assert "hp.writeColumn(cast(int) r, cast(long)(ob + x)" in body                                      # columns behind a contig's end
    # rules carried over (each named after the cli.cpp member that implements it)
    for cli_name, d_name in (("first_column(", "firstColumn("), ("last_column(", "lastColumn("), ("pending_empty_", "base_pending_empty"),
                             ("pending_empty", "win_pending_empty"), ("stale_st", "stale_st"), ("zero_windows(", "zeroWindows("),
                             ("last_nl", "last_nl")):
        assert cli_name in cli and d_name in body or d_name in d, (cli_name, d_name)
    assert "k * w + w <= fpos" in body                  # windows finished before the first column of the run print nothing
    assert "lastcol >= w ? (lastcol - w) / w + 1 : 0" in body      # windows finished by over-hanging columns
    # the writer callback is declared outside the nothrow @nogc block (its D implementation writes to a File)
    head = d[:d.index("extern (C) nothrow @nogc {")]
    assert "alias sbx_write_fn = extern (C) int function" in head
    assert "extern (C) int sbxFileSink" in d and "&sbxFileSink" in body
