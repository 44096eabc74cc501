"""The regular-expression engine behind `-F "read_name =~ /.../"` (sambamba_amd/csrc/regex_nfa.hpp): the same NFA
simulation runs on the device per record and on the host here -- compared with Python's `re.search` on the common
syntax of D's std.regex / ECMAScript."""
import itertools
import random
import re

import pytest

import sambamba_amd

PATTERNS = [
    r"abc", r"^abc", r"abc$", r"^abc$", r"a.c", r"a.*c", r"a.+c", r"ab?c", r"a|b", r"^(a|b)c", r"(ab)+", r"(?:ab)*c",
    r"[abc]x", r"[^abc]x", r"[a-c]+[0-9]$", r"\d+", r"^\d+$", r"\w+_\d", r"\s", r"\S\s\S", r"[\d_]+", r"[^\w]",
    r"a{3}", r"a{2,}", r"^a{1,2}b", r"(ab){2,3}c", r"x{0,2}y", r"colou?r", r"^r[0-9]+_[0-9]+$", r"^chr([0-9]+|X|Y|M)$",
    r"\.", r"a\+b", r"\/", r"^$", r"^", r"$", r"", r"(a|)b", r"a*?b", r"a+?", r"[]a]", r"[a-]", r"\D\d\D",
    r"^(?:[A-Z][a-z]+)+$", r"H[A-Z0-9]+:[0-9]:[0-9]+", r"(a|ab)(c|bcd)(d*)", r"x*", r"(x+x+)+y",
]
TEXTS = ["", "a", "abc", "xabcx", "abcabc", "ac", "abbc", "a1", "c9", "ab12", "r12_345", "r12_", "chr1", "chrX", "chr10x", "chrUn",
         "a+b", "a/b", "a.b", "hello world", "colour", "color", "colr", "aaa", "aa", "aaab", "ababab", "ababc", "xxy", "y", "]", "a-",
         "HWI-ST1234:7:1101", "HABC:1:22", "AbcDef", "abcdefg", "abcd", "_", "\t", "q r", "xxxxxxxxxxxxxxxxxxxx"]


@pytest.mark.parametrize("pat", PATTERNS)
def test_search_agrees_with_python_re(pat):
    rx = re.compile(pat)
    for t in TEXTS:
        assert sambamba_amd.regex_search(pat, t) == (rx.search(t) is not None), (pat, t)


def test_case_insensitive_option():
    for pat, t in itertools.product([r"abc", r"[a-c]+\d", r"^chrx$", r"Q"], ["ABC", "aBc1", "CHRX", "chrx", "q", "zzz"]):
        assert sambamba_amd.regex_search(pat, t, "i") == (re.search(pat, t, re.I) is not None), (pat, t)


def test_random_patterns():
    rng = random.Random(5)
    atoms = ["a", "b", "c", ".", "[ab]", "[^a]", r"\d", "(a|b)", "(ab)", "x"]
    quant = ["", "", "", "*", "+", "?", "{2}", "{1,2}"]
    for _ in range(300):
        pat = "".join(rng.choice(atoms) + rng.choice(quant) for _ in range(rng.randint(1, 4)))
        if rng.random() < 0.2:
            pat = "^" + pat
        if rng.random() < 0.2:
            pat += "$"
        rx = re.compile(pat)
        for _ in range(8):
            t = "".join(rng.choice("abcx1") for _ in range(rng.randint(0, 7)))
            assert sambamba_amd.regex_search(pat, t) == (rx.search(t) is not None), (pat, t)


@pytest.mark.parametrize("pat", [r"(a)\1", r"a(?=b)", r"\bword", r"(", r"a{40}", r"[z-a]", "a" * 70])
def test_outside_the_subset_is_an_error(pat):
    with pytest.raises(sambamba_amd.SbxError):
        sambamba_amd.regex_search(pat, "abc")


def test_filter_compiles_regex_conditions():
    f = sambamba_amd.compile_filter("read_name =~ /^r[0-9]+_/ and not ([RG] =~ /lane\\/1/i) and mapping_quality > 3")
    assert [f.ops[i].kind for i in range(f.n_ops)] == [15, 15, 5, 3, 2, 3] and f.n_regex == 2
    assert f.ops[0].field == 0 and f.ops[1].field == 3 and f.ops[1].mask == ord("R") | (ord("G") << 8)
    with pytest.raises(sambamba_amd.SbxError):
        sambamba_amd.compile_filter("read_name =~ /a/ and read_name =~ /b/ and read_name =~ /c/")      # three
    with pytest.raises(sambamba_amd.SbxError):
        sambamba_amd.compile_filter("read_name =~ /a/m")
