"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/sbx_depth.h declares, and it refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from tests.util import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sbx_depth.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbx_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import sambamba_amd
    L = sambamba_amd.lib()
    syms = _declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, "declared in sbx_depth.h but not exported: %s" % missing


def test_python_binding_lists_the_same_symbols():
    from sambamba_amd._lib import EXPORTS
    assert sorted(EXPORTS) == _declared_symbols()


def test_filter_compiler_default_and_errors():
    import sambamba_amd
    f = sambamba_amd.compile_filter(None)   # depth.d:1159 default
    kinds = [f.ops[i].kind for i in range(f.n_ops)]
    # mapping_quality > 0 ; duplicate ; not ; and ; failed_qc ; not ; and
    assert kinds == [2, 0, 5, 3, 0, 5, 3]
    assert f.ops[1].mask == 0x400 and f.ops[4].mask == 0x200
    g = sambamba_amd.compile_filter("not (unmapped or mate_is_unmapped) and mapping_quality >= 20 or chimeric")
    assert g.n_ops > 0
    h = sambamba_amd.compile_filter("[NM] < 3 and [XS] == null")      # integer tags and tag existence compile
    assert [h.ops[i].kind for i in range(h.n_ops)] == [7, 8, 3]
    assert h.ops[0].mask == ord("N") | (ord("M") << 8) and h.ops[0].cmp == 1 and h.ops[0].value == 3
    k = sambamba_amd.compile_filter("[RG] == 'lane\\'1' and strand == '-' and ref_name != 'chrM'")
    assert [k.ops[i].kind for i in range(k.n_ops)] == [9, 0, 3, 11, 3]
    assert k.strings[:6] == b"lane'1" and k.ops[0].value == (6 << 32)
    with pytest.raises(sambamba_amd.SbxError) as ei:
        sambamba_amd.compile_filter("[RG] =~ /(a)\\1/")
    assert ei.value.code == -5   # SBX_EUNSUPPORTED: back-references are outside the regular-expression subset
    assert sambamba_amd.compile_filter("read_name =~ /abc/").n_regex == 1


def test_no_cpu_fallback_without_device():
    import sambamba_amd
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(sambamba_amd.SbxError) as ei:
        sambamba_amd.Depth(os.path.join(ROOT, "tests", "golden", "issue225.bam"))
    assert "no HIP device" in str(ei.value) or "HIP" in str(ei.value)


def test_product_does_not_link_or_include_the_oracle():
    csrc = os.path.join(ROOT, "sambamba_amd", "csrc")
    for fn in os.listdir(csrc):
        if fn.endswith((".cpp", ".hpp", ".hip", ".h")):
            text = open(os.path.join(csrc, fn)).read()
            assert "oracle/" not in text and "liboracle" not in text and "zlib.h" not in text, fn
    for fn in os.listdir(os.path.join(ROOT, "sambamba_amd")):
        if fn.endswith(".py"):
            text = open(os.path.join(ROOT, "sambamba_amd", fn)).read()
            assert "oracle" not in text.replace("no CPU", ""), fn
